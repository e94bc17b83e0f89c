import sys, os
sys.path.insert(0, "/root/repo")
import torch
from simpleimagecaptionzoo_amd.butd import gemm
torch.manual_seed(0)
for M, N, K, ns in ((64, 4096, 3072, 8), (128, 4096, 4096, 8), (50, 1024, 1024, 4), (100, 10102, 1024, 1), (64, 640, 128, 1)):
    X = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05
    want = X.double() @ W.double().t()
    for pl in (False, True):
        got = gemm("nt", X, W, None, ns, planes=pl).double()
        print(M, N, K, ns, "planes" if pl else "inline", float((got - want).abs().max() / want.abs().max()))
