import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simpleimagecaptionzoo_amd.butd import gemm
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 8
X = torch.randn(64, 4096, device="cuda"); W = torch.randn(4096, 4096, device="cuda")
for _ in range(20):
    gemm("nt", X, W, None, ns)
torch.cuda.synchronize()
