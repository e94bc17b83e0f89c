"""dev tool (round 5): what a merged greedy + sampled chain could buy at small row counts.  Per batch b: the SCST step by phase
(two concurrent chains of b rows, today) and ONE greedy chain at b and at 2 b rows (graph replay) -- the merged chain's lower bound."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

for B in (8, 16, 32):
    eng, opt, vocab, words = bench.build_engine("cuda:0", 2 * B)
    batches = bench.make_batches(6, B, words, "cuda:0", 0)
    for bt in batches:
        eng.scorer().preload(bt[2])
    eng.SCST_training_epoch(batches, opt, None, tqdm_visible=False)
    eng.phase_events = []
    eng.SCST_training_epoch(batches * 3, opt, None, tqdm_visible=False)
    torch.cuda.synchronize()
    ph = eng.phase_times(skip=2)
    h = eng._hot_handle()
    out = {}
    with torch.cuda.stream(eng.stream):
        for rows in (B, 2 * B):
            f = torch.relu(torch.randn(rows, 36, 2048, device="cuda:0"))
            for _ in range(3):
                h.greedy(f, 20)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                h.greedy(f, 20)
            e1.record()
            torch.cuda.synchronize()
            out[rows] = e0.elapsed_time(e1) / 10
    print("b=%-3d rollouts %.3f reward %.3f backward %.3f adam %.3f | one greedy chain: %d rows %.3f ms, %d rows %.3f ms"
          % (B, ph["rollouts"], ph["reward"], ph["backward"], ph["adam"], B, out[B], 2 * B, out[2 * B]), flush=True)
    del eng, opt
