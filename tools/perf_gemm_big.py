"""dev tool (round 5): the many-row split-precision GEMM shapes of the path through icz_gemm_f32, one process per kernel choice
(ICZ_GEMM_BIG=0: gemm_tn128_x3_kernel, 1..4: gemm_big_x3_kernel configs).  Prints us per launch (HIP events around `iters` launches,
operands rotated through `nrot` copies so that the warm column re-reads one pair and the cold column never repeats inside the
256 MB Infinity Cache) and the error against float64.

    for c in 0 1 2 3 4; do ICZ_GEMM_BIG=$c python tools/perf_gemm_big.py; done
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simpleimagecaptionzoo_amd.butd import gemm

CFG = os.environ.get("ICZ_GEMM_BIG", "0")


def mk(layout, M, N, K, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    if layout == "nt":
        return torch.randn(M, K, device="cuda", generator=g), torch.randn(N, K, device="cuda", generator=g)
    if layout == "nn":
        return torch.randn(M, K, device="cuda", generator=g), torch.randn(K, N, device="cuda", generator=g)
    return torch.randn(K, M, device="cuda", generator=g), torch.randn(K, N, device="cuda", generator=g)


def ref64(layout, X, W):
    X, W = X.double(), W.double()
    return X @ W.t() if layout == "nt" else (X @ W if layout == "nn" else X.t() @ W)


def timed(fn, iters):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def run(layout, M, N, K, ns, iters=20, check=True):
    byts = 4 * (M * K + N * K)
    nrot = max(2, min(12, int(600e6 // byts)))
    ops = [mk(layout, M, N, K, 1000 * i + M + N + K) for i in range(nrot)]
    out = gemm(layout, ops[0][0], ops[0][1], None, ns)
    err = float("nan")
    if check:
        r = ref64(layout, *ops[0])
        err = ((out.double() - r).abs().max() / r.abs().max()).item()
        del r
    for i in range(3):
        gemm(layout, ops[0][0], ops[0][1], None, ns)
    warm = timed(lambda i: gemm(layout, ops[0][0], ops[0][1], None, ns), iters)
    cold = timed(lambda i: gemm(layout, ops[i % nrot][0], ops[i % nrot][1], None, ns), iters)
    fl = 2.0 * M * N * K
    print("cfg %s %s M=%5d N=%5d K=%5d ns=%d: warm %7.1f us (%5.1f TF-eq = %.2f of 417)  cold %7.1f us   rel err %.1e"
          % (CFG, layout, M, N, K, ns, warm, fl / warm / 1e6, fl / warm / 1e6 / 417.0, cold, err), flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which == "check":
        for (lay, M, N, K, ns) in (("nt", 700, 4100, 1024, 1), ("nt", 700, 4100, 1024, 2), ("nt", 640, 1024, 1024, 4), ("nt", 2304, 1024, 1024, 1),
                                   ("nt", 129, 8200, 1152, 3),
                                   ("nn", 300, 260, 128, 1), ("nn", 1280, 1028, 10112, 4), ("nn", 130, 516, 2176, 1), ("nn", 1280, 4096, 4096, 2),
                                   ("tn", 2052, 2060, 96, 1), ("tn", 4096, 1024, 1280, 1), ("tn", 2048, 2048, 64, 1), ("tn", 4100, 2044, 1264, 1)):
            run(lay, M, N, K, ns, 2)
        sys.exit(0)
    shapes = []
    if which in ("all", "tn"):
        shapes += [("tn", 4096, 4096, 1280, (1,)), ("tn", 4096, 3072, 1280, (1,)), ("tn", 4096, 2048, 1280, (1,)), ("tn", 10112, 1024, 1280, (1,))]
    if which in ("all", "nn"):
        shapes += [("nn", 1280, 1024, 10112, (0, 2, 4, 8)), ("nn", 1280, 4096, 4096, (0, 1, 2)), ("nn", 1280, 3072, 4096, (0, 1, 2))]
    if which in ("all", "nt"):
        shapes += [("nt", 640, 4096, 4096, (0, 2, 3, 4, 5, 6)), ("nt", 640, 4096, 3072, (0, 2, 3, 4, 6)), ("nt", 640, 10112, 1024, (0, 1, 2)),
                   ("nt", 640, 1024, 1024, (0, 2, 4)),
                   ("nt", 2304, 1024, 1024, (0, 1, 2, 4)), ("nt", 2304, 2048, 2048, (0, 1, 2, 4)), ("nt", 2304, 3072, 1024, (0, 1, 2)),
                   ("nt", 1280, 10112, 1024, (0, 1))]
    for (lay, M, N, K, splits) in shapes:
        for ns in splits:
            run(lay, M, N, K, ns)
