"""In-kernel time stamps of the skinny GEMM (ICZ_SKINNY_ABL=4): per workgroup the shader clock at entry, after the prologue,
after every stage and at the end -> where a launch spends its time.  python tools/perf_skinny_stamps.py M N K nsplit"""
import ctypes as C, os, sys
os.environ.setdefault("ICZ_SKINNY_ABL", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from simpleimagecaptionzoo_amd.butd import gemm
from simpleimagecaptionzoo_amd._lib import lib
M, N, K, ns = [int(x) for x in sys.argv[1:5]]
ns_arg = ns
X = torch.randn(M, K, device="cuda"); Ws = [torch.randn(N, K, device="cuda") * 0.03 for _ in range(6)]
for i in range(12):
    gemm("nt", X, Ws[i % 6], None, ns_arg, planes=bool(int(os.environ.get("PLANES", "0"))))
torch.cuda.synchronize()
if os.environ.get('RESIDENT'):
    ns = K // 256
tile = 128 if 2048 <= N <= 8192 else 64
nwg = (N + tile - 1) // tile * ns
buf = (C.c_ulonglong * (32 * nwg))()
L = lib()
L.icz_debug_skinny_stamps.restype = C.c_int
assert L.icz_debug_skinny_stamps(buf, nwg) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 32).astype(np.int64)
nst = K // 64 // ns
t0 = st[:, 0].min()
rel = st - t0
print("workgroups %d, stages per workgroup %d (clock ticks; 100 MHz if s_memtime counts the reference clock, else shader cycles)" % (nwg, nst))
print("entry spread (max - min of first stamp): %d" % (rel[:, 0].max()))
nst = min(nst, 9)
med = lambda a: int(np.median(a))
print("median ticks: prologue %d | per stage (issue loads / compute / barrier): %s | epilogue %d" % (
    med(st[:, 1] - st[:, 0]),
    "  ".join("%d/%d/%d" % (med(st[:, 2 + 3 * s] - st[:, 1 + 3 * s] if s else st[:, 2] - st[:, 1]), med(st[:, 3 + 3 * s] - st[:, 2 + 3 * s]),
                            med(st[:, 4 + 3 * s] - st[:, 3 + 3 * s])) for s in range(nst)),
    med(st[:, 31] - st[:, 1 + 3 * nst])))
print("workgroup total (last stamp - first) median %d max %d; launch span (max end - min start) %d" % (
    np.median(st[:, 31] - st[:, 0]), (st[:, 31] - st[:, 0]).max(), st[:, 31].max() - t0))
if os.environ.get("RESIDENT"):
    nwg = (N + 255) // 256 * (K // 256)
    buf = (C.c_ulonglong * (32 * nwg))()
    assert L.icz_debug_skinny_stamps(buf, nwg) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 32).astype(np.int64)
    print("resident kernel, %d workgroups: entry->x loads+split issued %d | ->planes complete (barrier) %d | steps: %s | total %d (max %d)" % (
        nwg, med(st[:, 1] - st[:, 0]), med(st[:, 2] - st[:, 1]),
        " ".join("%d%s" % (med(st[:, 3 + 2 * i] - (st[:, 2] if i == 0 else st[:, 1 + 2 * i] if (i % 4) else st[:, 2 + 2 * i] if i else st[:, 2])),
                           ("+epi %d" % med(st[:, 4 + 2 * i] - st[:, 3 + 2 * i])) if i % 4 == 3 else "") for i in range(8)),
        med(st[:, 31] - st[:, 0]), int((st[:, 31] - st[:, 0]).max())))
