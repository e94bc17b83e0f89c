"""bench.py's N > 1 control flow on the one GPU of the box (gloo instead of RCCL, ICZ_REHEARSE_ONE_GPU=1): launcher, ranks started
by bench.py itself, the contract line's fields, the launcher killing stragglers.  Child processes with their own torch import: slow, marked
gpu_slow (still part of -m gpu).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from _fullwidth import (ROOT, _check_dp_fields)  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.gpu_slow
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_rank_rehearsal_prints_the_contract_line(scaling):
    """bench.py's N > 1 control flow in fresh child processes (torch.distributed.run, two ranks on the one GPU of the box over
    gloo: ICZ_REHEARSE_ONE_GPU=1): normaliser all-reduce, gradient hook, barriers, MAX over ranks, rank-0 JSON.  The numbers
    mean nothing here; the line's shape and the rank count do."""
    env = dict(os.environ, ICZ_REHEARSE_ONE_GPU="1", MASTER_ADDR="127.0.0.1")
    port = 29500 + (os.getpid() % 400) + (0 if scaling == "weak" else 1)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--headline-only",
           "--scaling", scaling] + (["--dp-pieces", "1"] if scaling == "strong" else [])      # round 6: the timed region with ONE all-reduce behind the backward pass
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout[-2000:]
    j = json.loads(line[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["scaling"] == scaling and j["steps"] == 2 and j["warmup"] == 1
    assert j["config"]["global_batch"] == (128 if scaling == "weak" else 64) and j["config"]["parallelism"] == "dp2"
    assert j["value"] > 0 and j["unit"] == "captions/s" and j["higher_is_better"] is True and "grad_allreduce_ms" in j
    assert j["dp_pieces"] == (1 if scaling == "strong" else 4) and j["config"]["dp_pieces"] == j["dp_pieces"]
    _check_dp_fields(j)


@pytest.mark.gpu_slow
@pytest.mark.parametrize("n,extra", [(2, ["--scaling", "weak"]), (4, ["--scaling", "strong", "--batch", "32"])])
def test_bench_starts_its_own_ranks(n, extra):
    """`python bench.py --gpus N` with no launcher and no WORLD_SIZE (the form the driver uses at N = 1): bench.py must start the N
    ranks itself -- fresh child processes, the parent stays off the GPU -- and relay ONE line with n_gpus = ranks_seen = N
    (rehearsal mode: all ranks on the one GPU of the box over gloo; five ranks through the Engine itself: test_gpu_dist_two_ranks.py).  Four ranks x 8 rows: uneven launch timing, the <= 32-row
    decoder path, four gradient slices reduced from the library's callback under graph replay at ranks >= 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["ICZ_REHEARSE_ONE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--headline-only"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout[-2000:]
    j = json.loads(line[0])
    assert j["n_gpus"] == n and j["ranks_seen"] == n and j["config"]["parallelism"] == "dp%d" % n
    assert j["config"]["global_batch"] == (128 if n == 2 else 32) and j["value"] > 0 and "grad_allreduce_ms" in j
    _check_dp_fields(j)


@pytest.mark.gpu_slow
def test_bench_launcher_kills_the_other_ranks_when_one_dies():
    """A rank that exits early used to leave the others in the rendezvous until the process-group timeout, with their output
    discarded: now the launcher polls every child, kills the rest at the first failure (or at ICZ_BENCH_RANK_TIMEOUT), prints
    every rank's tail and exits non-zero -- within seconds."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(ICZ_REHEARSE_ONE_GPU="1", ICZ_BENCH_TEST_FAIL_RANK="1", ICZ_BENCH_RANK_TIMEOUT="300")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--headline-only"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and time.time() - t0 < 120
    assert "rank 1 exited with code" in r.stderr and "---- rank 1" in r.stderr and "told to fail" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu_slow
def test_bench_refuses_a_rank_count_that_is_not_the_one_asked_for():
    """--gpus 2 under WORLD_SIZE = 1 used to print a warning and an n_gpus = 1 line; now it is an error before any work starts."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--headline-only"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE = 1" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
