"""BUTD SCST step through Engine.SCST_training_epoch with the pipelined clamp + Adam on / off (same process, alternating legs).
usage: perf_step.py [steps per leg]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng, opt, vocab, words = bench.build_engine("cuda:0", 64)
batches = bench.make_batches(n + 3, 64, words, "cuda:0", 0)
for bt in batches:
    eng.scorer().preload(bt[2])


def leg(flag):
    eng.pipeline_adam = flag          # only read by the rejected patch (tools/cxx/rejected/r4_pipelined_adam.patch)
    eng.SCST_training_epoch(batches[:3], opt, None, tqdm_visible=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.SCST_training_epoch(batches[3:], opt, None, tqdm_visible=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for r in range(3):
    for flag in (False, True):
        print("pipeline_adam=%-5s %.3f ms per step" % (flag, leg(flag)), flush=True)
