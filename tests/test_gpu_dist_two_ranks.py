"""Data parallelism through the real Engine: two ranks and five ranks (processes on this one GPU, gloo instead of RCCL) each take
their share of the golden batches (halves; 2 1 1 1 1 of six images); the normaliser is all-reduced before backward, the gradient
groups from the library's callback, and after every XE / SCST step every rank holds the parameters the reference Engine produced on
the whole batch (tests/dp_worker.py).  Five is the most a GPU box admits beside the test process (six GPU processes); the 8-rank
rehearsal of the host-side rules is tests/test_cpu_dist_gloo.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, pytest.param(5, marks=pytest.mark.gpu_slow)])
def test_ranks_reproduce_the_single_process_reference_steps(world):
    import socket
    here = os.path.dirname(os.path.abspath(__file__))
    with socket.socket() as sk:          # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(here, "dp_worker.py")], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out, err))
    for r, (rc, out, err) in enumerate(outs):
        assert rc == 0 and ("rank %d ok" % r) in out, err[-3000:]
