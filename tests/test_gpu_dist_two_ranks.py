"""Data parallelism through the real Engine: two ranks (two processes on this one GPU, gloo instead of RCCL) each take half
of the golden batches; the normaliser is all-reduced before backward, the gradient groups from the library's callback, and
after every XE / SCST step both ranks hold the parameters the reference Engine produced on the whole batch (tests/dp_worker.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_two_ranks_reproduce_the_single_process_reference_steps():
    import socket
    here = os.path.dirname(os.path.abspath(__file__))
    with socket.socket() as sk:          # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(here, "dp_worker.py")], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out, err))
    for r, (rc, out, err) in enumerate(outs):
        assert rc == 0 and ("rank %d ok" % r) in out, err[-3000:]
