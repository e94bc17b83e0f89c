import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from simpleimagecaptionzoo_amd.butd import gemm
def bench(M, N, K, ns, it=20):
    X = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda")
    for _ in range(3): gemm("nt", X, W, None, ns)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(it): gemm("nt", X, W, None, ns)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / it
    print(f"nt M={M} N={N} K={K} ns={ns}: {dt*1e6:7.1f} us  {2*M*N*K/dt/1e12:6.1f} TF   tiles {((M+127)//128)*((N+127)//128)}", flush=True)
for (M, N, K) in [(2048, 2048, 2048), (2304, 2048, 2048), (4096, 2048, 2048), (2048, 3072, 1024), (2304, 3072, 1024), (2048, 1024, 1024), (2304, 1024, 1024), (4096, 1024, 1024), (2304,1024,2048),(2048,1024,2048)]:
    for ns in (1,):
        bench(M, N, K, ns)
