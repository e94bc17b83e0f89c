"""dev tool (round 6): the headline SCST step (B = 64) by phase and one greedy 64 x 20 decode, N timed steps after warm-up, for same-box
A/Bs of process-level switches (tools/ab_env.sh VAR rounds tools/perf_headline.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng, opt, vocab, words = bench.build_engine("cuda:0", 64)
batches = bench.make_batches(n + 3, 64, words, "cuda:0", 0)
for bt in batches:
    eng.scorer().preload(bt[2])
eng.SCST_training_epoch(batches[:3], opt, None, tqdm_visible=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.SCST_training_epoch(batches[3:], opt, None, tqdm_visible=False)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
eng.phase_events = []
eng.SCST_training_epoch(batches[3:13], opt, None, tqdm_visible=False)
torch.cuda.synchronize()
ph = {k: round(v, 3) for k, v in eng.phase_times(skip=2).items()}
h = eng._hot_handle()
feats = batches[0][3]["bu_feats"]
with torch.cuda.stream(eng.stream):
    for _ in range(3):
        h.greedy(feats, 20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        h.greedy(feats, 20)
    torch.cuda.synchronize()
    g_us = (time.perf_counter() - t0) / 20 / 20 * 1e6
print("SCST step %.3f ms  phases (ms) %s  greedy decode %.1f us per step" % (ms, ph, g_us), flush=True)
