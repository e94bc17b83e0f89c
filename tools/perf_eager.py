import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
eng, opt, vocab, words = bench.build_engine("cuda:0", 64)
batches = bench.make_batches(2, 64, words, "cuda:0", 0)
eng.use_graphs = False
h = eng._hot_handle(); h.set_concurrent(False)
def run(n):
    eng.SCST_training_epoch([batches[i % 2] for i in range(n)], opt, None, tqdm_visible=False)
run(2); torch.cuda.synchronize()
t0 = time.perf_counter(); run(3); torch.cuda.synchronize(); print("eager wall ms/step", (time.perf_counter() - t0) / 3 * 1e3)
