"""dev tool (round 6): a few EAGER AoADetection SCST steps (no hipGraph replay) for a kernel timeline under rocprofv3
(tools/trace_timeline.py: under the profiler a replayed graph runs its branches one after the other)."""
import os, sys
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
eng.use_graphs = False
eng.SCST_training_epoch([batches[i % 2] for i in range(5)], opt, None, tqdm_visible=False)
torch.cuda.synchronize()
print("done")
