#!/usr/bin/env python3
"""Headline benchmark: streaming frames/s of TemporalCRN.realtime_process at batch 256 on MI355X.

A "step" is one realtime_process call (the hot path) over one batch of B synthetic 3 s, 16 kHz,
3-microphone utterances already resident in HBM: B x Nseg frames (Nseg = 34 windows of 3200 samples per stream,
including the reference's K/2 left pad and gap padding).  value = frames / s summed over all ranks.

Multi-GPU (`--gpus N`, N > 1): one process per GPU, streams sharded across ranks (independent units: no data-path
collective, weak scaling); barrier + synchronize on both sides of the timed region, MAX over ranks over RCCL.
  * launched by the driver through `python -m torch.distributed.run ... bench.py --gpus N`: RANK is in the environment
    and this process IS rank RANK;
  * launched plainly as `python bench.py --gpus N`: this process is only the PARENT - it never touches the GPU (no
    torch.cuda / HIP call, no exec), spawns the N ranks through torch.distributed.run as a child process and exits
    with the child's code.  Fewer than N visible devices -> non-zero exit with a message, never a silent fallback.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed on the launch stream in an extra
profiled step after the timed region) and `cpu_baseline` (the C oracle timed on the host cores, N=1 only) plus
`cpu_baseline_torch` (the PyTorch-CPU restatement at B = 1 and 32, all cores, best of 3).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  One leg uses the caller's stream + three
# stage streams (CRN), or + two side streams (FullSubNet / training layer wavefront): with earlier legs' streams in
# the round-robin, two of a leg's own streams can land on one queue and serialise (training leg 289 -> 306 utt/s, student +1 %, CRN_ELU
# +2 % with 8 queues; the headline itself is unchanged).  Read by the runtime at initialisation, inherited by the rank processes.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

FP32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4 dense peak
BF16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak (spec)


MODELS = {  # name -> (variant, num_channels, hidden, label): reference CRN.py / CRN_ELU.py / distillation_crn.py student
    "crn": (0, [16, 32, 64, 128], 512, "CRN"),
    "crn_elu": (1, [16, 32, 64, 128], 512, "CRN_ELU"),
    "student": (2, [16, 32, 64, 64], 128, "distilled CRN_ELU student"),
}
PRECISIONS = {"f32": 0, "f16": 1, "bf16x3": 2}  # se_config.precision


def crn_cfg(nfft, model="crn"):
    _, ch, hid, _ = MODELS[model]
    return dict(num_channels=ch, num_freqs=nfft // 2 + 1, hidden=hid, segment_length=3200, num_layers=2,
                num_inputs=3, kernel_size=3, sample_rate=16000, win_length=25, hop_length=10, n_fft=nfft)


def seg_count(L, K=3200):
    """Frames per stream of realtime_process(flag=False): K/2 left pad + the reference's gap padding (utility.py:312-370)."""
    P = K // 2
    Lp = L + P
    gap = K - (P + Lp % K) % K
    return 2 * (Lp + gap + P) // K


# ---------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` with no RANK in the environment
# ---------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpus():
    """Device count WITHOUT initialising HIP in this process (torch.cuda.device_count() does not, on this image)."""
    import torch
    try:
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def launch_ranks(n, argv):
    """Parent of an N-rank run: spawn `torch.distributed.run` as a CHILD process (this process has not touched the GPU and
    never will) and return its exit code."""
    fake = os.environ.get("SE_BENCH_SELFTEST") == "1"  # CPU rehearsal of the launcher path (tests/test_bench_launcher.py)
    if not fake:
        ndev = visible_gpus()
        if ndev < n:
            print(f"bench.py: --gpus {n} requested but only {ndev} GPU(s) are visible; refusing to run "
                  f"(one process per GPU over RCCL, no silent fallback)", file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def init_world(args):
    """(rank, local_rank, world, backend).  N > 1: RCCL (`nccl`) with one device per rank; anything else is refused."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("SE_BENCH_SELFTEST") == "1":
        return rank, local_rank, world, "gloo"
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    ndev = torch.cuda.device_count()
    if ndev < world:
        raise SystemExit(f"bench.py: {world} ranks but only {ndev} GPU(s) visible; refusing (no gloo / shared-device fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # RCCL over xGMI
    return rank, local_rank, world, "nccl" if world > 1 else "none"


def launcher_selftest(args):
    """CPU-only rehearsal of the N-rank plumbing (SE_BENCH_SELFTEST=1): gloo world, barrier, MAX over ranks, one JSON line
    from rank 0.  Exercises exactly the spawn / env / reduction path of a real run; no engine, no timing claim."""
    import torch
    import torch.distributed as dist
    rank, _, world, backend = init_world(args)
    if os.environ.get("SE_BENCH_SELFTEST_FAIL_RANK") == str(rank):
        raise SystemExit(3)  # rehearses a dying rank: the parent must exit non-zero
    dist.init_process_group("gloo")
    dist.barrier()
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    from speech_enhancement_mi_amd.sharding import shard_streams
    lo, hi = shard_streams(args.batch * world, rank, world)
    dist.barrier()
    if rank == 0:
        print(json.dumps(dict(selftest=True, n_gpus=world, backend=backend, max_over_ranks=float(t.item()), rank0_streams=[lo, hi])))
    dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------
# baselines and roofline helpers
# ---------------------------------------------------------------------------------------------------------------
def host_cpu_share():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box hands a 1-GPU job a
    16-CPU share of a 256-thread host; 256 OpenMP threads on that share spin against each other and never finish)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def progress(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(cfg, sd, variant=0, seconds_budget=20.0):
    """Oracle (C restatement of the reference path, OpenMP) on the host cores, bounded sample."""
    from oracle import crn_oracle as orc
    from speech_enhancement_mi_amd import synth
    avail = host_cpu_share()
    # one stream per thread (the oracle parallelises over streams and channels); more threads than streams only add
    # OpenMP overhead, so the baseline uses min(available cores, 32) threads and says so in `cores`
    cores = orc.lib().crn_oracle_set_threads(max(1, min(avail, 32)))
    o = orc.CrnOracle(**cfg, variant=variant)
    o.load_state_dict(sd)
    B, L = max(2, cores), 8000
    mix, _ = synth.synth_utterances(B, L, 3, seed=99)
    nseg = seg_count(L)
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


def cpu_baseline_torch(cfg, sd, seconds_budget=30.0):
    """SURVEY.md 8d / BASELINE.md 3: the PyTorch-CPU restatement of the path (speech_enhancement_mi_amd.training.TrainableCRN,
    pinned against the oracle in tests/test_training_cpu.py) under no_grad on all host cores, B in {1, 32}, 3 s utterances,
    one warm-up utterance, best of 3.  Variant 0 (CRN.py) only."""
    import torch
    from speech_enhancement_mi_amd import synth
    from speech_enhancement_mi_amd.training import TrainableCRN
    avail = host_cpu_share()
    prev = torch.get_num_threads()
    torch.set_num_threads(avail)
    model = TrainableCRN(**cfg)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.eval()
    out = []
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    t_all = time.time()
    with torch.no_grad():
        for B in (1, 32):
            mix, _ = synth.synth_utterances(B, 48000, 3, seed=7)
            x = torch.from_numpy(mix)
            progress(f"cpu_baseline_torch: B={B}, {avail} threads")
            model.realtime_process_train(x[:1])  # warm-up utterance
            best = None
            for _ in range(3):
                t0 = time.time()
                model.realtime_process_train(x)
                dt = time.time() - t0
                best = dt if best is None else min(best, dt)
                if time.time() - t_all > seconds_budget:
                    break
            nseg = seg_count(48000)
            out.append(dict(batch=B, value=B * nseg / best, unit="frames/s", realtime_factor=B * 3.0 / best, cores=avail, kind="port",
                            cpu=cpu_model, sample=f"best of <=3 realtime_process of {B} x 3 s utterances ({nseg} frames each), torch {torch.__version__} CPU, {avail} threads"))
    torch.set_num_threads(prev)
    return out


def cpu_baseline_torch_guarded(args, timeout=240):
    """cpu_baseline_torch in a CHILD process (CPU only, never touches the GPU) under a hard timeout, so that a slow host
    cannot stall the bench line."""
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--nfft", str(args.nfft), "--model", args.model]
    try:
        env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        lines = [l for l in r.stdout.splitlines() if l.startswith("[")]
        if r.returncode == 0 and lines:
            return json.loads(lines[-1])
        return dict(error=f"worker exited {r.returncode}", stderr=r.stderr[-300:])
    except subprocess.TimeoutExpired:
        return dict(error=f"PyTorch-CPU baseline did not finish within {timeout} s on {host_cpu_share()} CPUs")


def kernel_source_sha():
    """sha256 over the kernel sources: stamps the committed PMC summary so a stale one is refused."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "speech_enhancement_mi_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, workload_key):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of THIS source state (FETCH_SIZE and
    WRITE_SIZE collected in separate runs of this same command; FETCH_SIZE doubled per MI355X_MICROARCH.md 'HBM': on gfx950
    it reports half of a wide coalesced read).  profiles/pmc_hbm_current.json = {"src_sha", "workload", "kernels": {name:
    {"fetch_kb", "write_kb"}}} is written by profiles/summarize.py; a summary made from other kernel sources (sha mismatch)
    or another workload is refused and traffic is null."""
    path = os.path.join(ROOT, "profiles", "pmc_hbm_current.json")
    if not os.path.exists(path):
        return None, "no PMC summary committed for this source state"
    d = json.load(open(path))
    if d.get("src_sha") != kernel_source_sha():
        return None, f"PMC summary is from kernel sources {d.get('src_sha')}, current {kernel_source_sha()}: refused"
    if d.get("workload") != workload_key:
        return None, f"PMC summary is for workload {d.get('workload')}"
    k = d["kernels"].get(kernel)
    if not k:
        return None, "kernel not in PMC summary"
    return (2.0 * k["fetch_kb"] + k["write_kb"]) * 1024.0, d.get("source", "")


def timed_region(fn, steps, warmup, world, backend):
    """W untimed calls, then K calls bracketed by barrier + synchronize on both sides; MAX over ranks (RCCL)."""
    import torch
    import torch.distributed as dist
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        assert backend == "nccl" and dist.get_backend() == "nccl"
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


# ---------------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------------
def train_measure(args, rank, local_rank, world, backend, steps, warmup, train_model=None):
    """BASELINE configs[3]: TemporalCRN data-parallel training, utterances sharded across ranks, ONE flat fp32 gradient
    all-reduce (24.5 MB) per optimizer step.  A step = forward + backward over `--utts` 3 s utterances per GPU (two
    micro-batches, grad accumulation 2 like config.yaml:99), all-reduce, clip, Adam.  value = utterances/s over all ranks."""
    import torch
    from speech_enhancement_mi_amd import synth
    train_model = train_model or args.train_model
    from speech_enhancement_mi_amd.training import FlatBucket, TrainableCRN, TrainableCRNELU, train_step
    cfg = crn_cfg(400)  # the reference's training geometry (config.yaml:205-217)
    variant = 1 if train_model == "crn_elu" else 0
    model = (TrainableCRNELU if variant else TrainableCRN)(**cfg)
    spec = synth.crn_param_spec(cfg["num_channels"], cfg["num_freqs"], cfg["hidden"], cfg["num_layers"], 3, 3, variant=variant)
    sd = synth.make_state_dict(spec, seed=0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.cuda()
    model.use_hip_kernels(args.train_kernels == "hip")
    bucket = FlatBucket(list(model.parameters()))
    opt = torch.optim.Adam(model.parameters(), lr=3e-4)
    U, L = args.utts, int(args.seconds * 16000)
    mix, clean = synth.synth_utterances(U, L, 3, seed=2000 + rank)
    mix, clean = torch.from_numpy(mix).cuda(), torch.from_numpy(clean).cuda()
    last = {}
    gen_ms = []
    if args.data == "gen":
        # SURVEY 8f-4 wired into the step: every step draws U fresh rooms (config.yaml:77-88 ranges) and simulates its own batch ON THE GPU
        # (image-source RIRs + diffuse tail, dry speech * RIR, SNR mix ~ U[-5, 25] dB, MAX_AMP guard) from a device-resident pool of dry
        # signals - the stand-in for data_c.LibriPartyDataset + multichannel.Single2Multi, which kept the reference single-GPU (README.md:24)
        from speech_enhancement_mi_amd.datagen import Single2Multi
        sim = Single2Multi(((3, 3, 2.5), (4, 5, 3)), (0.2, 1.0), ((0.5,) * 6, (1.0,) * 6), ((0.1, 0.1, 0.2), (0.9, 0.9, 0.7)),
                           ((0.06, 0.06, 0.06), (0.15, 0.15, 0.15)), ((0.0, 0.0, 0.3), (1.0, 1.0, 0.7)), num_src=1, num_mic=3)
        _, dry = synth.synth_utterances(64, L, 1, seed=3000 + rank)
        dry_pool = torch.from_numpy(dry).cuda()
        noise_pool = torch.randn(64, L, device="cuda") * 0.1
        rng = np.random.default_rng(17 + rank)

        def make_batch():
            rb = sim.sample(U, rng, snr_low=-5.0, snr_high=25.0)
            i1 = torch.from_numpy(rng.integers(0, 64, U)).cuda()
            i2 = torch.from_numpy(rng.integers(0, 64, U)).cuda()
            srcs = torch.stack([dry_pool[i1], noise_pool[i2]], dim=1).contiguous()
            mx, y, _ = sim.simulate(srcs, rb, rir=sim.rir(rb, "cuda", diffuse=True, seed=int(rng.integers(0, 2 ** 31))))
            return mx, y[:, 0, 0].contiguous()

        gen_stream = torch.cuda.Stream()
        pending = {}

        def prefetch():  # the NEXT step's batch is generated on a side stream while this step trains
            with torch.cuda.stream(gen_stream):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                pending["batch"] = make_batch()
                e1.record()
                pending["ready"] = e1
                gen_ms.append((e0, e1))

        prefetch()

    def step():
        if args.data == "gen":
            torch.cuda.current_stream().wait_event(pending["ready"])
            mx, tgt = pending["batch"]
            prefetch()
        else:
            mx, tgt = mix, clean
        last["loss"] = train_step(model, bucket, opt, mx, tgt, accum=args.accum, loss=args.train_loss, merge=not args.no_merge)

    progress(f"train: {warmup} + {steps} steps, kernels = {args.train_kernels}, loss = {args.train_loss}")
    dt = timed_region(step, steps, warmup, world, backend)
    assert np.isfinite(last["loss"])
    value = world * U * steps / dt
    roofline = None
    if args.train_kernels == "hip":  # one extra profiled step: events around every hand-written launch on the launch stream
        from speech_enhancement_mi_amd import train_ops
        train_ops.PROF = {}
        torch.cuda.synchronize()
        with torch.cuda.stream(torch.cuda.Stream()):  # timed events taint their queue (see measure_crn): keep the default stream clean
            step()
            prof = train_ops.profile_summary()
        torch.cuda.synchronize()
        train_ops.PROF = None
        bwd = {k: v for k, v in prof.items() if v["flops"] > 0}
        dom = max(bwd, key=lambda k: bwd[k]["ms"])
        d = prof[dom]
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
        roofline = dict(bound="mfma", kernel=dom, achieved=ach, peak=FP32_MATRIX_PEAK_TFLOPS, unit="TFLOP/s", frac=ach / FP32_MATRIX_PEAK_TFLOPS,
                        frac_of_executed_pipe=ach / FP32_MATRIX_PEAK_TFLOPS, executed_pipe="fp32 MFMA (v_mfma_f32_32x32x2_f32)",
                        traffic=None, avg_launch_us=1e3 * d["ms"] / d["launches"], launches_per_step=d["launches"],
                        note="dominant hand-written kernel of the training step (forward + backward); fp32-exact MFMA (v_mfma_f32_32x32x2_f32), "
                             "priced against the fp32 matrix peak; k_conv_igemm = conv / deconv forward AND their input gradients",
                        kernels={k: dict(ms=round(v["ms"], 3), launches=v["launches"], tflops=v["flops"] / max(v["ms"], 1e-9) / 1e9) for k, v in prof.items()},
                        step_ms_profiled=sum(v["ms"] for v in prof.values()))
    result = dict(metric=f"DP training utterances/sec ({'CRN_ELU' if variant else 'TemporalCRN'}, 3 s utterances)", value=value, unit="utterances/s", n_gpus=world,
                  steps=steps, warmup=warmup, ms_per_step=1e3 * dt / steps, higher_is_better=True, scaling="weak",
                  vs_baseline=None, dtype="f32", data="synthetic" if args.data != "gen" else "synthetic, generated on the GPU inside every step (rooms, RIRs, mix)",
                  config=dict(workload=f"{'CRN_ELU (CRN_ELU.py, the model train.py:16 trains)' if variant else 'TemporalCRN (CRN.py)'} 400-pt training step: {U} utterances/GPU x {args.seconds:g} s, forward/backward kernels = {args.train_kernels}, "
                                       f"loss = {args.train_loss}, accum {args.accum} ({'micro-batches share one forward/backward sweep, loss formed per micro-batch: same gradient' if not args.no_merge else 'micro-batches run one after the other'}), "
                                       f"flat 24.5 MB fp32 gradient all-reduce, clip 5, Adam 3e-4",
                              utterances_per_gpu=U, parallelism=f"dp{world}", grad_bucket_bytes=int(bucket.flat.numel() * 4),
                              reference_note="the reference logged 1.09 utterances/s at batch 1 on an unknown GPU (BASELINE.md 1): not this metric's baseline"),
                  roofline=roofline, cpu_baseline=None)
    if gen_ms:
        torch.cuda.synchronize()
        result["config"]["generator_ms_per_step"] = float(np.mean([a.elapsed_time(b) for a, b in gen_ms[warmup:]]))
    del model, bucket, opt, mix, clean
    torch.cuda.empty_cache()
    return result


def train_line(args, rank, local_rank, train_model="crn"):
    r = train_measure(args, rank, local_rank, 1, "none", steps=8, warmup=3, train_model=train_model)
    return {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "roofline")} | dict(workload=r["config"]["workload"])


def bench_train(args, rank, local_rank, world, backend):
    import torch.distributed as dist
    result = train_measure(args, rank, local_rank, world, backend, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def fullsubnet_measure(args, rank, local_rank, world, backend, B, dtype, steps, warmup):
    """BASELINE configs[2]: FullSubNet (fb + sb 2-layer LSTM) streaming inference, reference config.yaml:153-172."""
    import torch
    from speech_enhancement_mi_amd import engine, synth
    spec = synth.fsn_param_spec(201, 3, 512, 384, 2, 15, 0)
    kw = {} if dtype == "f32" else dict(precision=PRECISIONS[dtype])
    eng = engine.FsnEngine(201, 3, 512, 384, 2, 15, 0, 0, 16000, 3200, 25, 10, 400, device=local_rank, **kw)
    eng.load_state_dict(synth.make_state_dict(spec, seed=0))
    L = int(args.seconds * 16000)
    base, _ = synth.synth_utterances(min(B, 16), L, 3, seed=1000 + rank)
    mix = torch.from_numpy(np.ascontiguousarray(np.tile(base, (-(-B // base.shape[0]), 1, 1))[:B])).cuda()
    out = torch.empty((B, L), dtype=torch.float32, device="cuda")
    nseg = seg_count(L)
    progress(f"FullSubNet: {warmup} + {steps} steps, B={B}, {dtype}")
    dt = timed_region(lambda: eng.realtime_process(mix, out=out), steps, warmup, world, backend)
    assert bool(torch.isfinite(out).all())
    value = world * B * nseg * steps / dt
    tf = value / world * eng.flops_per_frame / 1e12
    terms = {"f32": 6.0, "bf16x3": 3.0}[dtype]
    result = dict(metric=f"streaming frames/sec @ b{B} (FullSubNet, 3200-samp 16 kHz)", value=value, unit="frames/s", n_gpus=world,
                  steps=steps, warmup=warmup, ms_per_step=1e3 * dt / steps, higher_is_better=True, scaling="weak",
                  vs_baseline=None, dtype=dtype, data="synthetic",
                  config=dict(workload=f"FullSubNet realtime_process(train=False), batch {B} streams/GPU, 400-pt STFT / 201 bins, {args.seconds:g} s "
                                       f"utterances ({nseg} frames per stream), hash-generated weights", streams_per_gpu=B, frames_per_stream=nseg,
                              realtime_factor=value * 0.1, audio_realtime_factor=world * B * args.seconds * steps / dt,
                              mflop_per_frame=eng.flops_per_frame / 1e6),
                  roofline=dict(bound="mfma", kernel="k_lstm_step_x6", achieved=tf, peak=FP32_MATRIX_PEAK_TFLOPS, unit="TFLOP/s",
                                frac=tf / FP32_MATRIX_PEAK_TFLOPS, frac_of_executed_pipe=terms * tf / BF16_DENSE_PEAK_TFLOPS, terms_per_mac=terms,
                                fp32_equivalent_roof=BF16_DENSE_PEAK_TFLOPS / terms, traffic=None,
                                note="whole-path rate (algorithmic FLOPs of the whole frame / wall time); the fused sub-band LSTM step GEMM holds 99 % of "
                                     "the FLOPs (split-bf16 MFMA, see DESIGN.md)"),
                  cpu_baseline=None)
    eng.close()
    del mix, out
    torch.cuda.empty_cache()
    return result


def fullsubnet_line(args, rank, local_rank, B, dtype, steps, warmup):
    r = fullsubnet_measure(args, rank, local_rank, 1, "none", B, dtype, steps, warmup)
    return {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "roofline")}


def bench_fullsubnet(args, rank, local_rank, world, backend):
    import torch.distributed as dist
    result = fullsubnet_measure(args, rank, local_rank, world, backend, args.batch, args.dtype, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def measure_crn(model, nfft, B, dtype, seconds, steps, warmup, rank, local_rank, world, backend, label_progress=""):
    """One CRN-family streaming workload: timed region + one extra profiled step (HIP events on the launch stream around every
    launch).  Returns (value frames/s, ms_per_step, roofline dict, cfg, sd, variant, eng-derived info)."""
    import torch
    from speech_enhancement_mi_amd import engine, synth
    cfg = crn_cfg(nfft, model)
    variant, _, _, label = MODELS[model]
    spec = synth.crn_param_spec(cfg["num_channels"], cfg["num_freqs"], cfg["hidden"], cfg["num_layers"], 3, 3, variant=variant)
    sd = synth.make_state_dict(spec, seed=0)
    eng = engine.Engine(engine.make_config(cfg["num_channels"], cfg["num_freqs"], cfg["hidden"], 3200, 2, 3, 3, 16000, 25, 10, nfft,
                                           variant=variant, precision=PRECISIONS[dtype]), local_rank)
    eng.load_state_dict(sd)

    L = int(seconds * 16000)
    base, _ = synth.synth_utterances(min(B, 16), L, 3, seed=1000 + rank)  # 16 distinct utterances, tiled over the batch
    mix = torch.from_numpy(np.ascontiguousarray(np.tile(base, (-(-B // base.shape[0]), 1, 1))[:B])).cuda()
    out = torch.empty((B, L), dtype=torch.float32, device="cuda")
    nseg = seg_count(L)

    progress(f"{label_progress}timed region: {warmup} + {steps} steps ({model}, B={B}, {nfft}-pt, {dtype})")
    dt = timed_region(lambda: eng.realtime_process(mix, out=out), steps, warmup, world, backend)
    assert bool(torch.isfinite(out).all()), "non-finite output"
    frames = world * B * nseg * steps
    value = frames / dt
    progress(f"{label_progress}{value:.0f} frames/s; profiled step")

    # ---- roofline leg: one extra profiled step, HIP events around every launch on the launch stream ----
    # On a stream of its own: timed events switch the stream's HSA queue to profiling mode for good, and everything launched on that
    # queue afterwards pays for it - the launch-dense secondary legs ran 14-16 % slower on the tainted default stream (FullSubNet 10.4 k
    # -> 9.0 k frames/s, training 335 -> 282 utt/s; profiles/leg_order_probe.py).  (Moving the LEGS to fresh streams instead costs the
    # pipelined CRN legs 5-13 %: a fifth stream next to the engine's three stage streams shares one of the four hardware queues.)
    torch.cuda.synchronize()
    with torch.cuda.stream(torch.cuda.Stream()):
        eng.profile(True)
        eng.realtime_process(mix, out=out)
        recs = eng.profile_read()
        eng.profile(False)
    torch.cuda.synchronize()
    by_kernel = {}
    for r in recs:
        k = by_kernel.setdefault(r["kernel"], dict(ms=0.0, launches=0, flops=0.0))
        k["ms"] += r["ms"]
        k["launches"] += r["launches"]
        k["flops"] += r["flops_per_launch"] * r["launches"]
    dom = max(by_kernel, key=lambda k: by_kernel[k]["ms"])
    d = by_kernel[dom]
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
    # The contraction kernels produce fp32-accurate results from split-bf16 MFMAs (6 products per MAC at precision f32, 3 at
    # bf16x3).  `achieved` counts ALGORITHMIC fp32 FLOPs.  Two prices are reported: `frac` against the fp32 matrix
    # peak (the peak of the dtype the path's RESULTS are in - the number the previous rounds quoted), and
    # `frac_of_executed_pipe` = executed bf16/fp16 MFMA FLOPs / the dense bf16 peak of the pipe the kernel actually runs on
    # (equivalently achieved / (2.5 PF / terms)): a kernel can exceed the former, never the latter.
    terms = {"f32": 6.0, "bf16x3": 3.0, "f16": 1.0}[dtype]
    split = dtype in ("f32", "bf16x3")
    peak = FP32_MATRIX_PEAK_TFLOPS if split else BF16_DENSE_PEAK_TFLOPS  # fp16 operands: priced against the dense fp16 matrix peak
    workload_key = f"{model}/b{B}/nfft{nfft}/{dtype}"
    traffic, traffic_note = pmc_traffic(dom, workload_key)
    pipe_peak_equiv = BF16_DENSE_PEAK_TFLOPS / terms
    roofline = dict(bound="mfma", kernel=dom, achieved=achieved, peak=peak, unit="TFLOP/s",
                    frac=achieved / peak, frac_of_executed_pipe=terms * achieved / BF16_DENSE_PEAK_TFLOPS,
                    executed_pipe="bf16 MFMA (v_mfma_f32_32x32x16_bf16)" if split else "fp16 MFMA",
                    executed_pipe_peak=BF16_DENSE_PEAK_TFLOPS, terms_per_mac=terms, fp32_equivalent_roof=pipe_peak_equiv,
                    traffic=traffic, traffic_note=traffic_note,
                    executed_bf16_tflops=(terms * achieved if split else None), bf16_dense_peak=(BF16_DENSE_PEAK_TFLOPS if split else None),
                    avg_launch_us=1e3 * d["ms"] / max(1, d["launches"]), launches_per_step=d["launches"],
                    flops_per_launch=d["flops"] / max(1, d["launches"]),
                    whole_path_tflops=value / world * eng.flops_per_frame / 1e12,
                    kernels={k: dict(ms=round(v["ms"], 3), launches=v["launches"],
                                     tflops=(v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0),
                                     frac_of_executed_pipe=(terms * v["flops"] / (v["ms"] * 1e-3) / 1e12 / BF16_DENSE_PEAK_TFLOPS if v["ms"] > 0 else 0.0))
                             for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1]["ms"])},
                    labels={r["label"]: dict(ms=round(r["ms"], 3), launches=r["launches"]) for r in recs})
    info = dict(cfg=cfg, sd=sd, variant=variant, label=label, nseg=nseg, mflop_per_frame=eng.flops_per_frame / 1e6, dt=dt)
    eng.close()
    del mix, out
    torch.cuda.empty_cache()
    return value, 1e3 * dt / steps, roofline, info


TOL_NOTE = {"f32": "fp32-accurate (6-term split-bf16 MFMA), parity <= 1e-6 rel vs reference",
            "bf16x3": "3-term split-bf16 MFMA (hi*hi + hi*mid + mid*hi), parity checked against the reference goldens at 1e-4 rel RMS / 0.02 dB",
            "f16": "fp16 MFMA operands, fp32 accumulate: OUTSIDE north_star's 1e-4 / 0.02 dB parity bar (1.6e-3..2.6e-3 rel, <= 0.05 dB), reported for reference only"}


def secondary_lines(args, rank, local_rank):
    """BASELINE configs 3, 4, 5 (+ CRN_ELU) under the same clock as the headline, a few steps each, AFTER the headline's timed
    region (N = 1 only).  Each entry: {config, metric, value, unit, ms_per_step, steps, roofline{kernel, achieved, frac (vs the fp32
    matrix peak), frac_of_executed_pipe (vs 2.5 PF / terms)}}.  A failing leg is recorded as {"error": ...}: it never takes the
    headline line down."""
    out = []

    def leg(name, fn):
        t0 = time.time()
        try:
            r = fn()
        except Exception as ex:  # noqa: BLE001 - a secondary leg must not lose the headline
            r = dict(error=f"{type(ex).__name__}: {ex}"[:400])
        r["config"] = name
        r["wall_s"] = round(time.time() - t0, 2)
        out.append(r)

    def crn_leg(model, B, dtype, nfft, steps=3, warmup=1):
        def fn():
            value, ms, roof, info = measure_crn(model, nfft, B, dtype, args.seconds, steps, warmup, rank, local_rank, 1, "none", label_progress="secondary: ")
            slim = {k: roof[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_of_executed_pipe", "terms_per_mac",
                                         "fp32_equivalent_roof", "avg_launch_us", "launches_per_step", "whole_path_tflops")}
            slim["traffic"] = roof["traffic"]
            slim["kernels"] = {k: v for k, v in list(roof["kernels"].items())[:6]}
            return dict(metric=f"streaming frames/sec @ b{B} ({info['label']}, 3200-samp 16 kHz)", value=value, unit="frames/s", ms_per_step=ms,
                        steps=steps, warmup=warmup, dtype=dtype, n_fft=nfft, mflop_per_frame=info["mflop_per_frame"], precision=TOL_NOTE[dtype], roofline=slim)
        return fn

    leg("BASELINE configs[1] (the headline workload) in bf16x3: 3-term split-bf16, pinned at 1e-4 RMS / 0.02 dB against the reference goldens", crn_leg("crn", 256, "bf16x3", 512))
    leg("BASELINE configs[2]: FullSubNet streaming, batch 256, f32", lambda: fullsubnet_line(args, rank, local_rank, 256, "f32", steps=3, warmup=1))
    leg("BASELINE configs[2]: FullSubNet streaming, batch 256, bf16x3", lambda: fullsubnet_line(args, rank, local_rank, 256, "bf16x3", steps=3, warmup=1))
    leg("BASELINE configs[4]: distilled CRN_ELU student, batch 1024, bf16x3 (inside the parity bar)", crn_leg("student", 1024, "bf16x3", 400))
    leg("BASELINE configs[4]: distilled CRN_ELU student, batch 1024, f16 (the config's named dtype; outside the parity bar)", crn_leg("student", 1024, "f16", 400))
    leg("CRN_ELU (the variant train.py trains) streaming, batch 256, f32", crn_leg("crn_elu", 256, "f32", 400))
    leg("BASELINE configs[3]: TemporalCRN training step, 8 x 3 s utterances per GPU, accum 2, full loss", lambda: train_line(args, rank, local_rank))
    leg("CRN_ELU training step (the model train.py:16 trains), 8 x 3 s utterances per GPU, accum 2, full loss", lambda: train_line(args, rank, local_rank, "crn_elu"))
    return out


def bench_crn(args, rank, local_rank, world, backend):
    import torch.distributed as dist
    B = args.batch
    value, ms_per_step, roofline, info = measure_crn(args.model, args.nfft, B, args.dtype, args.seconds, args.steps, args.warmup, rank, local_rank, world, backend)
    cfg, sd, variant, label, nseg = info["cfg"], info["sd"], info["variant"], info["label"], info["nseg"]
    dt = info["dt"]
    result = dict(metric=f"streaming frames/sec @ b{B} ({label}, 3200-samp 16 kHz)", value=value, unit="frames/s",
                  n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step,
                  higher_is_better=True, scaling="weak", vs_baseline=None, dtype=args.dtype, data="synthetic",
                  config=dict(workload=f"TemporalCRN ({args.model}) realtime_process, batch {B} streams/GPU, {args.nfft}-pt STFT / {cfg['num_freqs']} bins, hop 160, "
                                       f"{args.seconds:g} s utterances ({nseg} frames of 3200 samples per stream), hash-generated weights",
                              streams_per_gpu=B, frames_per_stream=nseg, n_fft=args.nfft, parallelism=f"streams sharded x{world}, no collective",
                              realtime_factor=value * 0.1,  # SURVEY 8d definition: every frame (incl. the reference's pad / gap frames) = 100 ms
                              audio_realtime_factor=world * B * args.seconds * args.steps / dt,  # real seconds of audio / wall second
                              mflop_per_frame=info["mflop_per_frame"], precision=TOL_NOTE[args.dtype], collective_backend=backend),
                  roofline=roofline)
    if rank == 0:
        result["cpu_baseline"] = None
        if world == 1 and not args.no_secondary:
            result["secondary"] = secondary_lines(args, rank, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            progress("cpu_baseline: C oracle")
            result["cpu_baseline"] = cpu_baseline(cfg, sd, variant)
            if variant == 0:
                result["cpu_baseline_torch"] = cpu_baseline_torch_guarded(args)
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="streams per GPU")
    ap.add_argument("--nfft", type=int, default=512, help="512 = BASELINE.json configs[1]; 400 = reference config.yaml default")
    ap.add_argument("--seconds", type=float, default=3.0, help="utterance length")
    ap.add_argument("--model", choices=sorted(MODELS) + ["fullsubnet"], default="crn", help="crn = BASELINE.json headline (default)")
    ap.add_argument("--mode", choices=["infer", "train"], default="infer",
                    help="train = BASELINE configs[3]: data-parallel training step (flat-bucket RCCL all-reduce)")
    ap.add_argument("--utts", type=int, default=8, help="--mode train: utterances per GPU per optimizer step")
    ap.add_argument("--train-kernels", choices=["hip", "torch"], default="hip",
                    help="--mode train: hip = hand-written forward/backward kernels for conv / transposed conv / GRU (default); torch = autograd checker path")
    ap.add_argument("--train-loss", choices=["full", "sisnr"], default="full",
                    help="--mode train: full = 0.7 * stoi_loss + 0.3 * (-SI-SNR) (CRN.py:609-611); sisnr = the SI-SNR term alone")
    ap.add_argument("--train-model", choices=["crn", "crn_elu"], default="crn", help="--mode train: CRN.py (BASELINE configs[3]) or CRN_ELU.py (what train.py imports)")
    ap.add_argument("--accum", type=int, default=2, help="--mode train: micro-batches per optimizer step (config.yaml:99 uses 2)")
    ap.add_argument("--data", choices=["fixed", "gen"], default="fixed",
                    help="--mode train: fixed = one resident synthetic batch; gen = a fresh batch of simulated rooms per step from the GPU generator")
    ap.add_argument("--no-merge", action="store_true", help="--mode train: run the accumulation micro-batches one after the other (default: one shared sweep, same gradient)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="headline line only: skip the FullSubNet / student / CRN_ELU / training legs")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dtype", choices=sorted(PRECISIONS), default="f32",
                    help="f32 = fp32-accurate contractions (headline); bf16x3 = 3-term split-bf16 (inside the 1e-4 parity bar; BASELINE config 5: "
                         "--model student --dtype bf16x3 --batch 1024); f16 = fp16 MFMA operands (outside the parity bar)")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.cpu_baseline_worker:  # child of cpu_baseline_torch_guarded: CPU only
        from speech_enhancement_mi_amd import synth
        cfg = crn_cfg(args.nfft, args.model)
        spec = synth.crn_param_spec(cfg["num_channels"], cfg["num_freqs"], cfg["hidden"], cfg["num_layers"], 3, 3, variant=0)
        print(json.dumps(cpu_baseline_torch(cfg, synth.make_state_dict(spec, seed=0))))
        return
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: become the parent of N ranks BEFORE anything touches the GPU
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if os.environ.get("SE_BENCH_SELFTEST") == "1":
        return launcher_selftest(args)
    rank, local_rank, world, backend = init_world(args)
    if args.mode == "train":
        return bench_train(args, rank, local_rank, world, backend)
    if args.model == "fullsubnet":
        return bench_fullsubnet(args, rank, local_rank, world, backend)
    return bench_crn(args, rank, local_rank, world, backend)


if __name__ == "__main__":
    main()
