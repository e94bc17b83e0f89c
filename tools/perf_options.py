"""dev tool (round 5): one BUTD SCST step by phase under different handle options (icz_butd_set_option), alternating legs in ONE
process (same box, same clocks: differences of 0.01 ms are visible).  usage: perf_options.py name=value[,name=value] ...
e.g. `perf_options.py early_out=0 early_out=1`; ICZ_PERF_BREAK=11 raises the <end> logit until the reference's break
(BUTD_Model.py:233) triggers after ~11 of the 20 steps (bench.end_bias)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B = int(os.environ.get("ICZ_PERF_B", "64"))
legs = sys.argv[1:] or ["early_out=1"]
eng, opt, vocab, words = bench.build_engine("cuda:0", B)
if os.environ.get("ICZ_NO_GRAPHS"):
    eng.use_graphs = False
batches = bench.make_batches(4, B, words, "cuda:0", 0)
eng.SCST_training_epoch(batches, opt, None, tqdm_visible=False)
torch.cuda.synchronize()
brk = float(os.environ.get("ICZ_PERF_BREAK", "0"))
if brk > 0:
    print("<end> bias: break steps", bench.end_bias(eng, batches, brk))
scorer = eng.scorer()
N = int(os.environ.get("ICZ_PERF_STEPS", "14"))
ROUNDS = int(os.environ.get("ICZ_PERF_ROUNDS", "3"))


def leg(n):
    h = eng._hot_handle()
    for kv in n.split(","):
        k, v = kv.split("=")
        h.set_option(k, int(v))
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


h0 = eng._hot_handle()
for r in range(ROUNDS):
    for n in legs:
        ph, span = leg(n)
        brk_now = bench.break_step(h0._bufs[("sample_seq", B, 20)])
        print("%-24s break %2d  rollouts %.3f  reward %.3f  backward %.3f  adam %.3f  span %.3f ms" % (n, brk_now, ph[0], ph[1], ph[2], ph[3], span), flush=True)
