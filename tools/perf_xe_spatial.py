"""BUTDSpatial XE training step (BASELINE config 2 at N = 1: 49 grid regions, batch 64) -- dev tool; run under rocprofv3 for
profiles/r03_xe_spatial49_kernel_stats.csv (tools/prof_any.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

B = 64
rs = np.random.RandomState(0)
xb = []
for i in range(2):
    lens = sorted(rs.randint(9, 19, size=B).tolist(), reverse=True)
    caps = torch.zeros(B, max(lens), dtype=torch.int64)
    for b, n in enumerate(lens):
        caps[b, 0] = 1
        caps[b, 1:n - 1] = torch.from_numpy(rs.randint(4, bench.V, size=n - 2))
        caps[b, n - 1] = 2
    xb.append((tuple(range(B)), None, caps, lens, None))
print(bench.xe_spatial("cuda:0", B, xb))
