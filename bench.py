#!/usr/bin/env python3
"""Headline benchmark: streaming frames/s of TemporalCRN.realtime_process at batch 256 on MI355X.

A "step" is one realtime_process call (the hot path) over one batch of B synthetic 3 s, 16 kHz,
3-microphone utterances already resident in HBM: B x Nseg frames (Nseg = 34 windows of 3200 samples per stream,
including the reference's K/2 left pad and gap padding).  value = frames / s summed over all ranks.
N > 1: one process per GPU, streams sharded across ranks (independent units: no data-path collective, weak
scaling); barrier + synchronize on both sides of the timed region, MAX over ranks.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed on the launch stream in an extra
profiled step after the timed region) and `cpu_baseline` (the C oracle timed on the host cores, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4 dense peak


MODELS = {  # name -> (variant, num_channels, hidden): reference CRN.py / CRN_ELU.py / distillation_crn.py student
    "crn": (0, [16, 32, 64, 128], 512),
    "crn_elu": (1, [16, 32, 64, 128], 512),
    "student": (2, [16, 32, 64, 64], 128),
}


def crn_cfg(nfft, model="crn"):
    _, ch, hid = MODELS[model]
    return dict(num_channels=ch, num_freqs=nfft // 2 + 1, hidden=hid, segment_length=3200, num_layers=2,
                num_inputs=3, kernel_size=3, sample_rate=16000, win_length=25, hop_length=10, n_fft=nfft)


def cpu_baseline(cfg, sd, variant=0, seconds_budget=20.0):
    """Oracle (C restatement of the reference path, OpenMP) on the host cores, bounded sample."""
    from oracle import crn_oracle as orc
    from speech_enhancement_mi_amd import synth
    avail = len(os.sched_getaffinity(0))
    # one stream per thread (the oracle parallelises over streams and channels); more threads than streams only add
    # OpenMP overhead, so the baseline uses min(available cores, 32) threads and says so in `cores`
    cores = orc.lib().crn_oracle_set_threads(max(1, min(avail, 32)))
    o = orc.CrnOracle(**cfg, variant=variant)
    o.load_state_dict(sd)
    B, L = max(2, cores), 8000
    mix, _ = synth.synth_utterances(B, L, 3, seed=99)
    P, K = 1600, 3200
    Lp = L + P
    gap = K - (P + Lp % K) % K
    nseg = 2 * (Lp + gap + P) // K
    o.realtime_process(mix[:1, :, :3200])  # warm
    reps, t0 = 0, time.time()
    while True:
        o.realtime_process(mix)
        reps += 1
        if time.time() - t0 > seconds_budget / 2 or reps >= 8:
            break
    dt = time.time() - t0
    return dict(value=B * nseg * reps / dt, unit="frames/s", cores=cores, kind="port",
                sample=f"{reps} x realtime_process of {B} streams x {L} samples ({nseg} frames each), C oracle, OpenMP {cores} threads of {avail} available")


def pmc_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected
    in separate runs of this same command; FETCH_SIZE doubled per MI355X_MICROARCH.md 'HBM': on gfx950 it reports half
    of a wide coalesced read).  Only valid for the default workload; None otherwise or if the summary is missing."""
    path = os.path.join(ROOT, "profiles", "r01_v4_bench_b256_nfft512_pmc_hbm.csv")
    if not os.path.exists(path) or args.batch != 256 or args.nfft != 512 or args.model != "crn":
        return None
    import csv
    fetch = write = nf = nw = 0.0
    for row in csv.DictReader(open(path)):
        if kernel + "<" in row["kernel"] or row["kernel"].endswith("::" + kernel) or ("::" + kernel + "(") in row["kernel"]:
            if row["counter"] == "FETCH_SIZE":
                fetch += float(row["sum_KB"]); nf += float(row["launches"])
            elif row["counter"] == "WRITE_SIZE":
                write += float(row["sum_KB"]); nw += float(row["launches"])
    if nf == 0 or nw == 0:
        return None
    return (2.0 * fetch / nf + write / nw) * 1024.0


def bench_train(args, rank, local_rank, world):
    """BASELINE configs[3]: TemporalCRN data-parallel training, utterances sharded across ranks, ONE flat fp32 gradient
    all-reduce (24.5 MB) per optimizer step.  A step = forward + backward over `--utts` 3 s utterances per GPU (two
    micro-batches, grad accumulation 2 like config.yaml:99), all-reduce, clip, Adam.  value = utterances/s over all ranks."""
    import torch
    import torch.distributed as dist
    from speech_enhancement_mi_amd import synth
    from speech_enhancement_mi_amd.training import FlatBucket, TrainableCRN, train_step
    cfg = crn_cfg(400)  # the reference's training geometry (config.yaml:205-217)
    model = TrainableCRN(**cfg)
    spec = synth.crn_param_spec(cfg["num_channels"], cfg["num_freqs"], cfg["hidden"], cfg["num_layers"], 3, 3)
    sd = synth.make_state_dict(spec, seed=0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.cuda()
    bucket = FlatBucket(list(model.parameters()))
    opt = torch.optim.Adam(model.parameters(), lr=3e-4)
    U, L = args.utts, int(args.seconds * 16000)
    mix, clean = synth.synth_utterances(U, L, 3, seed=2000 + rank)
    mix, clean = torch.from_numpy(mix).cuda(), torch.from_numpy(clean).cuda()
    on_gpu = world > 1 and dist.get_backend() == "nccl"
    for _ in range(args.warmup):
        train_step(model, bucket, opt, mix, clean, accum=2)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = train_step(model, bucket, opt, mix, clean, accum=2)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert np.isfinite(loss)
    value = world * U * args.steps / dt
    result = dict(metric="DP training utterances/sec (TemporalCRN, 3 s utterances)", value=value, unit="utterances/s", n_gpus=world,
                  steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * dt / args.steps, higher_is_better=True, scaling="weak",
                  vs_baseline=value / 1.09, dtype="f32", data="synthetic",
                  config=dict(workload=f"TemporalCRN 400-pt training step: {U} utterances/GPU x {args.seconds:g} s, torch autograd forward/backward, "
                                       "flat 24.5 MB fp32 gradient all-reduce, clip 5, Adam 3e-4; loss = SI-SNR term only (STOI unpinned)",
                              utterances_per_gpu=U, parallelism=f"dp{world}", grad_bucket_bytes=int(bucket.flat.numel() * 4),
                              baseline_note="vs_baseline divides by the reference's 1.09 utterances/s at batch 1 on an unknown GPU (BASELINE.md 1) - indicative only"),
                  roofline=None, cpu_baseline=None)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def bench_fullsubnet(args, rank, local_rank, world):
    """BASELINE configs[2]: FullSubNet (fb + sb 2-layer LSTM) streaming inference, reference config.yaml:153-172."""
    import torch
    import torch.distributed as dist
    from speech_enhancement_mi_amd import engine, synth
    spec = synth.fsn_param_spec(201, 3, 512, 384, 2, 15, 0)
    eng = engine.FsnEngine(201, 3, 512, 384, 2, 15, 0, 0, 16000, 3200, 25, 10, 400, device=local_rank)
    eng.load_state_dict(synth.make_state_dict(spec, seed=0))
    B, L = args.batch, int(args.seconds * 16000)
    base, _ = synth.synth_utterances(min(B, 16), L, 3, seed=1000 + rank)
    mix = torch.from_numpy(np.ascontiguousarray(np.tile(base, (-(-B // base.shape[0]), 1, 1))[:B])).cuda()
    out = torch.empty((B, L), dtype=torch.float32, device="cuda")
    P, K = 1600, 3200
    Lp = L + P
    gap = K - (P + Lp % K) % K
    nseg = 2 * (Lp + gap + P) // K
    for _ in range(args.warmup):
        eng.realtime_process(mix, out=out)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.realtime_process(mix, out=out)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert bool(torch.isfinite(out).all())
    value = world * B * nseg * args.steps / dt
    tf = value / world * eng.flops_per_frame / 1e12
    result = dict(metric="streaming frames/sec @ b256 (FullSubNet, 3200-samp 16 kHz)", value=value, unit="frames/s", n_gpus=world,
                  steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * dt / args.steps, higher_is_better=True, scaling="weak",
                  vs_baseline=None, dtype="f32", data="synthetic",
                  config=dict(workload=f"FullSubNet realtime_process(train=False), batch {B} streams/GPU, 400-pt STFT / 201 bins, {args.seconds:g} s "
                                       f"utterances ({nseg} frames per stream), hash-generated weights", streams_per_gpu=B, frames_per_stream=nseg,
                              realtime_factor=value * 0.1, mflop_per_frame=eng.flops_per_frame / 1e6),
                  roofline=dict(bound="mfma", kernel="k_lstm_step_x6", achieved=tf, peak=FP32_MATRIX_PEAK_TFLOPS, unit="TFLOP/s",
                                frac=tf / FP32_MATRIX_PEAK_TFLOPS, traffic=None,
                                note="whole-path rate; the fused sub-band LSTM step GEMM holds 99 % of the FLOPs (bf16x6 MFMA, see DESIGN.md)"),
                  cpu_baseline=None)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="streams per GPU")
    ap.add_argument("--nfft", type=int, default=512, help="512 = BASELINE.json configs[1]; 400 = reference config.yaml default")
    ap.add_argument("--seconds", type=float, default=3.0, help="utterance length")
    ap.add_argument("--model", choices=sorted(MODELS) + ["fullsubnet"], default="crn", help="crn = BASELINE.json headline (default)")
    ap.add_argument("--mode", choices=["infer", "train"], default="infer",
                    help="train = BASELINE configs[3]: data-parallel training step (torch autograd + flat-bucket RCCL all-reduce)")
    ap.add_argument("--utts", type=int, default=8, help="--mode train: utterances per GPU per optimizer step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=["f32", "f16"], default="f32",
                    help="f32 = fp32-accurate contractions (headline); f16 = fp16 MFMA operands, fp32 accumulate (BASELINE config 5: --model student --dtype f16 --batch 1024)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from speech_enhancement_mi_amd import engine, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    ndev = torch.cuda.device_count()
    local_rank = local_rank % max(1, ndev)  # rehearsal with more ranks than GPUs shares devices (gloo, see below)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world <= ndev:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # RCCL over xGMI
        else:
            dist.init_process_group("gloo")  # several ranks per GPU: RCCL refuses duplicate devices

    if args.mode == "train":
        return bench_train(args, rank, local_rank, world)
    if args.model == "fullsubnet":
        return bench_fullsubnet(args, rank, local_rank, world)
    cfg = crn_cfg(args.nfft, args.model)
    variant = MODELS[args.model][0]
    spec = synth.crn_param_spec(cfg["num_channels"], cfg["num_freqs"], cfg["hidden"], cfg["num_layers"], 3, 3, variant=variant)
    sd = synth.make_state_dict(spec, seed=0)
    eng = engine.Engine(engine.make_config(cfg["num_channels"], cfg["num_freqs"], cfg["hidden"], 3200, 2, 3, 3, 16000, 25, 10, args.nfft,
                                           variant=variant, precision=1 if args.dtype == "f16" else 0), local_rank)
    eng.load_state_dict(sd)

    B, L = args.batch, int(args.seconds * 16000)
    base, _ = synth.synth_utterances(min(B, 16), L, 3, seed=1000 + rank)  # 16 distinct utterances, tiled over the batch
    mix = torch.from_numpy(np.ascontiguousarray(np.tile(base, (-(-B // base.shape[0]), 1, 1))[:B])).cuda()
    out = torch.empty((B, L), dtype=torch.float32, device="cuda")
    P, K = 1600, 3200
    Lp = L + P
    gap = K - (P + Lp % K) % K
    nseg = 2 * (Lp + gap + P) // K

    on_gpu = world > 1 and dist.get_backend() == "nccl"

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        eng.realtime_process(mix, out=out)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.realtime_process(mix, out=out)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert bool(torch.isfinite(out).all()), "non-finite output"

    frames = world * B * nseg * args.steps
    value = frames / dt

    # ---- roofline leg: one extra profiled step, HIP events around every launch on the launch stream ----
    eng.profile(True)
    eng.realtime_process(mix, out=out)
    recs = eng.profile_read()
    eng.profile(False)
    by_kernel = {}
    for r in recs:
        k = by_kernel.setdefault(r["kernel"], dict(ms=0.0, launches=0, flops=0.0))
        k["ms"] += r["ms"]
        k["launches"] += r["launches"]
        k["flops"] += r["flops_per_launch"] * r["launches"]
    dom = max(by_kernel, key=lambda k: by_kernel[k]["ms"])
    d = by_kernel[dom]
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
    # k_conv_x6 / k_gemm_bf16x6 produce fp32-accurate results from 6 bf16 MFMAs per product: `achieved`
    # counts ALGORITHMIC fp32 FLOPs and is priced against the fp32 matrix peak, the peak of the dtype the path computes in;
    # the bf16 matrix-core work actually executed is 6x that and is reported next to the bf16 dense peak.
    x6 = dom in ("k_conv_x6", "k_gemm_bf16x6") and args.dtype == "f32"
    peak = FP32_MATRIX_PEAK_TFLOPS if args.dtype == "f32" else 2500.0  # fp16 operands: priced against the dense fp16 matrix peak
    roofline = dict(bound="mfma", kernel=dom, achieved=achieved, peak=peak, unit="TFLOP/s",
                    frac=achieved / peak, traffic=pmc_traffic(dom, args) if args.dtype == "f32" else None,
                    executed_bf16_tflops=(6.0 * achieved if x6 else None), bf16_dense_peak=(2500.0 if x6 else None),
                    avg_launch_us=1e3 * d["ms"] / max(1, d["launches"]), launches_per_step=d["launches"],
                    flops_per_launch=d["flops"] / max(1, d["launches"]),
                    whole_path_tflops=value / world * eng.flops_per_frame / 1e12,
                    kernels={k: dict(ms=round(v["ms"], 3), launches=v["launches"],
                                     tflops=(v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0))
                             for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1]["ms"])},
                    labels={r["label"]: dict(ms=round(r["ms"], 3), launches=r["launches"]) for r in recs})

    result = dict(metric="streaming frames/sec @ b256 (CRN, 3200-samp 16 kHz)", value=value, unit="frames/s",
                  n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * dt / args.steps,
                  higher_is_better=True, scaling="weak", vs_baseline=None, dtype=args.dtype, data="synthetic",
                  config=dict(workload=f"TemporalCRN ({args.model}) realtime_process, batch {B} streams/GPU, {args.nfft}-pt STFT / {cfg['num_freqs']} bins, hop 160, "
                                       f"{args.seconds:g} s utterances ({nseg} frames of 3200 samples per stream), hash-generated weights",
                              streams_per_gpu=B, frames_per_stream=nseg, n_fft=args.nfft, parallelism=f"streams sharded x{world}, no collective",
                              realtime_factor=value * 0.1, mflop_per_frame=eng.flops_per_frame / 1e6),
                  roofline=roofline)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg, sd, variant)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
