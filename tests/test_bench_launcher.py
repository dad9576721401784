"""bench.py --gpus N launcher (CPU): the parent spawns N ranks through torch.distributed.run before touching any GPU,
rank 0 prints ONE JSON line with n_gpus == N, and a request for more GPUs than are visible exits non-zero."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(args, env_extra, timeout=300):
    env = dict(os.environ, **env_extra)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_launcher_spawns_two_rank_world():
    r = _run(["--gpus", "2", "--batch", "5"], {"SE_BENCH_SELFTEST": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["selftest"] is True
    assert d["max_over_ranks"] == 2.0          # MAX over ranks of (1 + rank) really went through the collective
    assert d["rank0_streams"] == [0, 5]        # streams sharded: rank 0 owns the first 5 of 10


def test_launcher_refuses_more_gpus_than_visible():
    import torch
    ndev = torch.cuda.device_count()
    r = _run(["--gpus", str(ndev + 2), "--steps", "1"], {})
    assert r.returncode != 0
    assert "visible" in r.stderr and "refusing" in r.stderr
    assert not any(l.startswith("{") for l in r.stdout.splitlines())  # no bench line from a run that did not happen


def test_child_failure_propagates():
    """A rank that dies makes the parent exit non-zero and print no bench line."""
    r = _run(["--gpus", "2"], {"SE_BENCH_SELFTEST": "1", "SE_BENCH_SELFTEST_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not any(l.startswith("{") for l in r.stdout.splitlines())
