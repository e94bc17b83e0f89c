"""dev: error of the vocabulary projection at 64 rows through the resident split-precision kernel (4 slabs) and through the un-split fp32-MFMA kernel, against float64"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simpleimagecaptionzoo_amd.butd import gemm
torch.manual_seed(0)
M, K, V, Vp = 64, 1024, 10102, 10112
v = (torch.rand(Vp, K, device="cuda") * 2 - 1) / 32
W = 6.0 * v                                          # sharpened weight-normed rows: |w_row| ~ 6 |v_row| / |v_row| ... magnitude as in the tests
W[V:] = 0
h = torch.tanh(torch.randn(M, K, device="cuda"))
X = h * 2.0 * (torch.rand(M, K, device="cuda") < 0.5)  # dropout(0.5) keep x 2
ref = (X.double() @ W.double().t())
a = gemm("nt", X, W, None, 4)          # N = Vp, split 4 -> resident kernel + slab reduce
b = gemm("nt", X, W[:V].contiguous(), None, 1)          # N = V, un-split -> fp32 MFMA kernel
ea, eb = (a.double() - ref)[:, :V], b.double() - ref[:, :V]
print("max |logit|", ref.abs().max().item())
print("resident x3 (4 slabs): max err %.3e  rms %.3e  mean %.3e" % (ea.abs().max().item(), ea.pow(2).mean().sqrt().item(), ea.mean().item()))
print("fp32 MFMA un-split   : max err %.3e  rms %.3e  mean %.3e" % (eb.abs().max().item(), eb.pow(2).mean().sqrt().item(), eb.mean().item()))
d = (a[:, :V] - b).double()
print("path difference      : max %.3e rms %.3e" % (d.abs().max().item(), d.pow(2).mean().sqrt().item()))
