"""In-kernel clock stamps of the 128-row resident GEMM (development build: make -C simpleimagecaptionzoo_amd/csrc DEV=1, then
ICZ_DEV_STAMPS=1): per workgroup the shader clock at entry, at the barrier, after every pipeline step, around the re-split and
after the two tile epilogues.  python tools/perf_m128_stamps.py M N K"""
import ctypes as C, os, sys
os.environ.setdefault("ICZ_DEV_STAMPS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from simpleimagecaptionzoo_amd.butd import gemm
from simpleimagecaptionzoo_amd._lib import lib
M, N, K = [int(x) for x in sys.argv[1:4]]
X = torch.randn(M, K, device="cuda"); Ws = [torch.randn(N, K, device="cuda") * 0.03 for _ in range(6)]
for i in range(12):
    gemm("nt", X, Ws[i % 6], None, 0)
torch.cuda.synchronize()
nwg = (N + 255) // 256 * (K // 256)
buf = (C.c_ulonglong * (32 * nwg))()
f = lib().icz_debug_skinny_stamps
f.restype = C.c_int
assert f(buf, nwg) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 32).astype(np.int64)
med = lambda a: int(np.median(a))
print("M %d N %d K %d: %d workgroups" % (M, N, K, nwg))
print("entry->barrier %d | barrier %d" % (med(st[:, 1] - st[:, 0]), med(st[:, 2] - st[:, 1])))
prev = st[:, 2]
steps = []
for i in range(8):
    if i == 4:
        steps.append("[resplit %d]" % med(st[:, 19] - prev))
        prev = st[:, 19]
    steps.append("%d" % med(st[:, 3 + i] - prev))
    prev = st[:, 3 + i]
print("steps: %s | tile 0 stores issued %d (after step 5), tile 1 %d (after step 7) | total %d (max %d) | launch span %d" % (
    " ".join(steps), med(st[:, 12] - st[:, 8]), med(st[:, 13] - st[:, 10]), med(st[:, 31] - st[:, 0]), int((st[:, 31] - st[:, 0]).max()),
    int(st[:, 31].max() - st[:, 0].min())))
