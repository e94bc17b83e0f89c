"""dev tool (round 5): one BUTD SCST step by phase with the LSTM weight gradients in 1 (= behind the loop, rounds 1-4) .. n time chunks
beside the reverse-time loop (option "wgrad_chunks"), alternating legs in ONE process.  usage: perf_bwd_chunks.py [chunk counts ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B = int(os.environ.get("ICZ_PERF_B", "64"))
counts = sys.argv[1:] or ["1", "2", "4"]       # "3" = three chunks, pieces on the 48 KB kernel; "3x" = on the regular kernels
eng, opt, vocab, words = bench.build_engine("cuda:0", B)
if os.environ.get("ICZ_NO_GRAPHS"):
    eng.use_graphs = False
batches = bench.make_batches(4, B, words, "cuda:0", 0)
eng.SCST_training_epoch(batches, opt, None, tqdm_visible=False)
torch.cuda.synchronize()
scorer = eng.scorer()
N = int(os.environ.get("ICZ_PERF_STEPS", "14"))
ROUNDS = int(os.environ.get("ICZ_PERF_ROUNDS", "3"))


def leg(n):
    h = eng._hot_handle()
    h.set_option("wgrad_chunks", int(n.rstrip("x")))
    h.set_option("wgrad_polite", 0 if n.endswith("x") else 1)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(N)]
    with torch.cuda.stream(eng.stream):
        eng.model.train()
        for i in range(N):
            ids, _, gts, supp = batches[i % 4]
            feats = eng._features(eng.modify_visual_inputs(img_tensors=None, supp_info_datas=supp))
            ev[i][0].record()
            g, s, lp = h.rollouts(feats, 20, eng.model._next_rng())
            ev[i][1].record()
            rew = scorer.reward(s, g, gts, ids)
            ev[i][2].record()
            grads = eng._grads()
            h.sample_backward(rew, grads, 0.0)
            ev[i][3].record()
            eng._apply(opt, 0.25)
            ev[i][4].record()
    torch.cuda.synchronize()
    ph = [sum(ev[i][j].elapsed_time(ev[i][j + 1]) for i in range(3, N)) / (N - 3) for j in range(4)]
    span = ev[3][0].elapsed_time(ev[N - 1][4]) / (N - 4)
    return ph, span


for r in range(ROUNDS):
    for n in counts:
        ph, span = leg(n)
        print("chunks=%-3s rollouts %.3f  reward %.3f  backward %.3f  adam %.3f  span %.3f ms" % (n, ph[0], ph[1], ph[2], ph[3], span), flush=True)
