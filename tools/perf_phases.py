"""dev tool: where one BUTD SCST step spends its time, by phase (HIP events on the Engine's stream, graphs on):
rollouts (greedy || sample) / CIDEr-D reward / REINFORCE backward (BPTT + weight gradients) / clamp + Adam."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from simpleimagecaptionzoo_amd.engine import BUTDDetection_Eng, init_optimizer
from simpleimagecaptionzoo_amd.synth import document_frequency, synthetic_references
from simpleimagecaptionzoo_amd.vocab import synthetic_vocab

B, V = 64, bench.V
vocab = synthetic_vocab(V)
words = [vocab.ix2word[i] for i in range(V)]
df = document_frequency(synthetic_references(2000, words, seed=0))
eng = BUTDDetection_Eng({"model_type": "BUTDDetection", "atten_dim": 1024, "embed_dim": 1024, "hidden_dim": 1024, "enc_dim": 2048},
                        "SYN", vocab, data_dir="/tmp/", device="cuda:0", cider_df=df, max_batch=B)
if os.environ.get("ICZ_NO_GRAPHS"):        # eager launches (host-issued, no captured graphs)
    eng.use_graphs = False
opt = init_optimizer("Adam", eng.model.get_param_groups({"lr": 2e-5}), 2e-5)
batches = bench.make_batches(4, B, words, "cuda:0", 0)
eng.SCST_training_epoch(batches, opt, None, tqdm_visible=False)
torch.cuda.synchronize()
scorer = eng.scorer()
N = 12
if os.environ.get("ICZ_EARLY_OUT") == "0":
    eng._hot_handle().set_option("early_out", 0)
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(N)]
with torch.cuda.stream(eng.stream):
    eng.model.train()
    for i in range(N):
        ids, _, gts, supp = batches[i % 4]
        feats = eng._features(eng.modify_visual_inputs(img_tensors=None, supp_info_datas=supp))
        h = eng._hot_handle()
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
names = ["rollouts", "reward", "backward", "clamp+adam"]
tot = 0.0
for j, n in enumerate(names):
    ms = sum(ev[i][j].elapsed_time(ev[i][j + 1]) for i in range(2, N)) / (N - 2)
    tot += ms
    print("%-12s %.3f ms" % (n, ms))
print("%-12s %.3f ms (sum of phases; the step also pays host gaps between them)" % ("total", tot))
span = ev[2][0].elapsed_time(ev[N - 1][4]) / (N - 2)
gaps = sum(ev[i][4].elapsed_time(ev[i + 1][0]) for i in range(2, N - 1)) / (N - 3)
print("%-12s %.3f ms per step on the device timeline; between two steps (features, handle, host) %.3f ms" % ("span", span, gaps))
