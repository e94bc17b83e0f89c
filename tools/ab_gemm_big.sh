#!/bin/bash
# same-box A/B of the 128 x 128 split-precision GEMM shapes: tools/ab_gemm_big.sh "VAR=val" "VAR=val" ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for v in "$@"; do
  echo "== $v"
  env $v python tools/perf_gemm_tn.py 2>/dev/null | head -4
  env $v python - <<'PY' 2>/dev/null
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "tools")); sys.path.insert(0, os.getcwd())
import importlib.util, torch, time
from simpleimagecaptionzoo_amd.butd import gemm
def bench(M, N, K, ns, it=30):
    X = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda")
    for _ in range(3): gemm("nt", X, W, None, ns)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(it): gemm("nt", X, W, None, ns)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / it
    print(f"nt M={M} N={N} K={K} ns={ns}: {dt*1e6:7.1f} us  {2*M*N*K/dt/1e12:6.1f} TF", flush=True)
for (M, N, K) in [(2304, 2048, 2048), (2304, 3072, 1024), (2304, 1024, 1024), (640, 10112, 1024)]:
    bench(M, N, K, 0)
PY
done
