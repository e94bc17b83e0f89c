"""dev tool: SCST step wall time under (graphs, concurrent, prof) combinations"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from simpleimagecaptionzoo_amd._lib import lib
eng, opt, vocab, words = bench.build_engine("cuda:0", 64)
batches = bench.make_batches(2, 64, words, "cuda:0", 0)
def run(n):
    eng.SCST_training_epoch([batches[i % 2] for i in range(n)], opt, None, tqdm_visible=False)
for graphs, conc, prof in ((1, 1, 0), (0, 1, 0), (0, 0, 0), (0, 0, 1), (1, 0, 0)):
    eng.use_graphs = bool(graphs)
    h = eng._hot_handle()
    h.set_concurrent(bool(conc))
    run(3); torch.cuda.synchronize()
    if prof: lib().icz_prof_begin()
    t0 = time.perf_counter(); run(5); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    msg = ""
    if prof:
        a, b, f, n = C.c_double(), C.c_double(), C.c_double(), C.c_longlong()
        lib().icz_prof_end(C.byref(a), C.byref(b), C.byref(f), C.byref(n)); msg = "avg kernel %.1f us over %d" % (a.value, n.value)
    print("graphs=%d concurrent=%d prof=%d: host issue %.2f ms/step, wall %.2f ms/step %s" % (graphs, conc, prof, (t1 - t0) / 5 * 1e3, (t2 - t0) / 5 * 1e3, msg))
