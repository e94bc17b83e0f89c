"""dev tool: AoADetection_Eng.SCST_training_epoch / training_epoch wall time per step at full size (B = 64, 36 regions):
what the handle-level numbers of perf_aoa.py become behind the Engine (reward, clamp + Adam, host glue)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from simpleimagecaptionzoo_amd.engine import AoADetection_Eng, init_optimizer
from simpleimagecaptionzoo_amd.synth import document_frequency, synthetic_references
from simpleimagecaptionzoo_amd.vocab import synthetic_vocab

B, V = 64, bench.V
vocab = synthetic_vocab(V)
words = [vocab.ix2word[i] for i in range(V)]
df = document_frequency(synthetic_references(2000, words, seed=0))
eng = AoADetection_Eng({"model_type": "AoADetection", "embed_dim": 1024, "hidden_dim": 1024}, "SYN", vocab, data_dir="/tmp/",
                       use_bu="fixed", device="cuda:0", cider_df=df, max_batch=B)
opt = init_optimizer("Adam", eng.model.get_param_groups({"lr": 2e-5}), 2e-5)
batches = bench.make_batches(2, B, words, "cuda:0", 0)


def run(n):
    eng.SCST_training_epoch([batches[i % 2] for i in range(n)], opt, None, tqdm_visible=False)


for rnd in range(3):
    for graphs in (False, True):          # alternating legs in one process (round 5: hipGraph replay of the rollout pair and the backward pass)
        eng.use_graphs = graphs
        if os.environ.get("ICZ_PERF_OPT_AB"):       # any 0 / 1 option of icz_aoa_set_option: off on the "eager" leg, on on the "graphs" leg
            eng.use_graphs = True
            eng.model._handle().set_option(os.environ["ICZ_PERF_OPT_AB"], 1 if graphs else 0)
        elif os.environ.get("ICZ_PERF_PAIR_AB"):      # second switch of round 5: both refiner passes as one (graphs leg) against two (eager leg)
            eng.use_graphs = True
            eng.model._handle().set_option("refine_pair", 1 if graphs else 0)
        run(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(10)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("AoA SCST step through the Engine, graphs=%-5s host issue %.2f ms, wall %.2f ms -> %.0f captions/s"
              % (graphs, (t1 - t0) / 10 * 1e3, (t2 - t0) / 10 * 1e3, B * 10 / (t2 - t0)), flush=True)
eng.phase_events = []
run(10)
torch.cuda.synchronize()
print("phases (ms):", {k: round(v, 3) for k, v in eng.phase_times(skip=2).items()})
