"""In-kernel clock stamps of the resident-activation GEMM (development build: make -C simpleimagecaptionzoo_amd/csrc DEV=1, then
ICZ_DEV_STAMPS=1): per workgroup the shader clock at entry, through the prologue, after every pipeline step and at the end.
python tools/perf_skinny_stamps.py M N K"""
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
tot = K // 64
k512 = os.environ.get("ICZ_GEMM_RESIDENT_K512", "1") not in ("", "0") and M <= 64 and tot % 8 == 0      # round 6: 512-deep ranges on one column tile
nsr = 8 if k512 else 4
nwg = ((N + 127) // 128 if k512 else (N + 255) // 256) * (tot // nsr)
buf = (C.c_ulonglong * (32 * nwg))()
L = lib()
f = L.icz_debug_skinny_stamps
f.restype = C.c_int
assert f(buf, nwg) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 32).astype(np.int64)
med = lambda a: int(np.median(a))
nt = nsr if k512 else 2 * nsr
epi_every = nsr
print("M %d N %d K %d: %d workgroups x %d stages" % (M, N, K, nwg, nsr))
print("prologue: entry->x loads issued %d | ->W issued %d | ->x arrived %d | ->split+LDS writes issued %d | barrier %d" % (
    med(st[:, 20] - st[:, 0]), med(st[:, 21] - st[:, 20]), med(st[:, 22] - st[:, 21]), med(st[:, 1] - st[:, 22]), med(st[:, 2] - st[:, 1])))
steps = []
prev = st[:, 2]
for i in range(nt):
    steps.append("%d" % med(st[:, 3 + 2 * i] - prev))
    prev = st[:, 3 + 2 * i]
    if i % epi_every == epi_every - 1:
        steps[-1] += "+epi %d" % med(st[:, 4 + 2 * i] - st[:, 3 + 2 * i])
        prev = st[:, 4 + 2 * i]
print("steps: %s | total %d (max %d) | launch span %d" % (" ".join(steps), med(st[:, 31] - st[:, 0]), int((st[:, 31] - st[:, 0]).max()),
                                                         int(st[:, 31].max() - st[:, 0].min())))
