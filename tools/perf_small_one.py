"""dev tool: N SCST steps at batch ICZ_PERF_B (default 8) through the Engine, for rocprofv3 kernel stats (tools/prof_any.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
B = int(os.environ.get("ICZ_PERF_B", "8"))
eng, opt, vocab, words = bench.build_engine("cuda:0", B)
batches = bench.make_batches(6, B, words, "cuda:0", 0)
for bt in batches:
    eng.scorer().preload(bt[2])
if os.environ.get("ICZ_MERGE_SMALL") is not None:
    eng._hot_handle().set_option("merge_small", int(os.environ["ICZ_MERGE_SMALL"]))
eng.SCST_training_epoch(batches * 3, opt, None, tqdm_visible=False)
torch.cuda.synchronize()
