"""Cold first epoch of the SCST step (every image unseen): where the host time of the reference store goes (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B = 64
eng, opt, vocab, words = bench.build_engine("cuda:0", B)
scorer = eng.scorer()
warm = bench.make_batches(6, B, words, "cuda:0", 0)
for bt in warm:
    scorer.preload(bt[2])
eng.SCST_training_epoch(warm, opt, None, tqdm_visible=False)
torch.cuda.synchronize()
acc = {"append": 0.0, "upload": 0.0, "prepare": 0.0}
for name in ("_append", "_upload", "prepare"):
    fn = getattr(scorer, name)
    def wrap(*a, _fn=fn, _n=name.strip("_"), **k):
        t0 = time.perf_counter()
        r = _fn(*a, **k)
        acc[_n] += time.perf_counter() - t0
        return r
    setattr(scorer, name, wrap)
import simpleimagecaptionzoo_amd.ciderd as cd
_orig_grow = scorer._grow
acc["grow"] = 0.0
def grow(*a, **k):
    t0 = time.perf_counter(); r = _orig_grow(*a, **k); acc["grow"] += time.perf_counter() - t0; return r
scorer._grow = grow
_orig_sync = torch.cuda.Event.synchronize
acc["evsync"] = 0.0
def evsync(self):
    t0 = time.perf_counter(); r = _orig_sync(self); acc["evsync"] += time.perf_counter() - t0; return r
torch.cuda.Event.synchronize = evsync
for rep in range(3):
    for k in acc:
        acc[k] = 0.0
    cold = bench.make_batches(20, B, words, "cuda:0", 0, id_base=(rep + 1) * 10_000_000)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.SCST_training_epoch(cold, opt, None, tqdm_visible=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    eng.SCST_training_epoch(cold, opt, None, tqdm_visible=False)
    torch.cuda.synchronize()
    dw = (time.perf_counter() - t0) / 20
    print("   grow %.3f ms, event sync %.3f ms per step" % (acc["grow"] / 20 * 1e3, acc["evsync"] / 20 * 1e3))
    print("cold %.3f ms/step, the same batches again (warm) %.3f ms/step; per step: _append %.3f ms (upload %.3f), prepare (any thread) %.3f ms"
          % (dt * 1e3, dw * 1e3, acc["append"] / 20 * 1e3, acc["upload"] / 20 * 1e3, acc["prepare"] / 20 * 1e3))
