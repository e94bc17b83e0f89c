import os, sys, random
sys.path.insert(0, os.getcwd())
import torch
from simpleimagecaptionzoo_amd.butd import gemm, gemm_set_big_cfg
random.seed(1)
bad = 0
for it in range(90):
    lay = random.choice(["nt", "nn", "tn"])
    cfg = random.choice([1, 2, 3, 4, 5])
    if lay == "tn":
        M = random.randrange(2048, 4200, 4); N = random.randrange(2048, 3000, 4); K = random.choice([64, 96, 160, 320, 608]); ns = 1
    elif lay == "nn":
        M = random.randrange(128, 1500); N = random.randrange(128, 2100, 4); K = 128 * random.randrange(1, 12); ns = random.choice([1, 2, 3])
    else:
        M = random.randrange(256, 2400); N = random.randrange(1024, 4200, 4); K = 128 * random.randrange(4, 16); ns = random.choice([1, 2, 3])
    g = torch.Generator(device="cuda").manual_seed(it)
    if lay == "nt":
        X = torch.randn(M, K, device="cuda", generator=g); W = torch.randn(N, K, device="cuda", generator=g); ref = X.double() @ W.double().t()
    elif lay == "nn":
        X = torch.randn(M, K, device="cuda", generator=g); W = torch.randn(K, N, device="cuda", generator=g); ref = X.double() @ W.double()
    else:
        X = torch.randn(K, M, device="cuda", generator=g); W = torch.randn(K, N, device="cuda", generator=g); ref = X.double().t() @ W.double()
    gemm_set_big_cfg(0); base = gemm(lay, X, W, None, ns)
    gemm_set_big_cfg(cfg); out = gemm(lay, X, W, None, ns)
    err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
    same = torch.equal(out, base)
    if err > 3e-6 or not same:
        bad += 1
        print("BAD", lay, M, N, K, ns, cfg, err, same, flush=True)
gemm_set_big_cfg(-2)
print("fuzz done, bad =", bad)
