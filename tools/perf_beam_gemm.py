"""The three big GEMMs of a beam step at 640 rows in the step's own order with the step's own (fixed) weights -- TD gates, LM gates,
vocabulary projection, again and again -- for rocprofv3 --kernel-trace --stats.  usage: perf_beam_gemm.py ns_td ns_lm ns_pred
(ICZ_GEMM_PLANES_TEST=<config + 1> ICZ_GEMM_PLANES_STATIC_W=1: the planes GEMM with weight planes packed once)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simpleimagecaptionzoo_amd.butd import gemm
ns = [int(x) for x in sys.argv[1:4]]
shapes = [(640, 4096, 3072), (640, 4096, 4096), (640, 10112, 1024)]
Xs = [torch.randn(M, K, device="cuda") for M, N, K in shapes]
Ws = [torch.randn(N, K, device="cuda") * 0.03 for M, N, K in shapes]
for i in range(3):
    out = gemm("nt", Xs[i], Ws[i], None, ns[i])
    ref = Xs[i].double() @ Ws[i].double().t()
    print("shape %s rel err %.2e" % (shapes[i], ((out.double() - ref).abs().max() / ref.abs().max()).item()))
for it in range(40):
    for i in range(3):
        gemm("nt", Xs[i], Ws[i], None, ns[i])
torch.cuda.synchronize()
