import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
eng, opt, vocab, words = bench.build_engine("cuda:0", 64)
batches = bench.make_batches(2, 64, words, "cuda:0", 0)
def run(n):
    eng.SCST_training_epoch([batches[i % 2] for i in range(n)], opt, None, tqdm_visible=False)
for use_side in (0, 1):
    for graphs in (0, 1):
        eng.use_graphs = bool(graphs)
        ctx = torch.cuda.stream(torch.cuda.Stream()) if use_side else torch.cuda.stream(torch.cuda.default_stream())
        with ctx:
            h = eng._hot_handle(); h.set_concurrent(False)
            run(2); torch.cuda.synchronize()
            t0 = time.perf_counter(); run(4); torch.cuda.synchronize()
            print("side_stream=%d graphs=%d: wall %.2f ms/step" % (use_side, graphs, (time.perf_counter() - t0) / 4 * 1e3))
