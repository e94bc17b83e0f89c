"""Seeded synthetic inputs shared by the golden generator and the tests (numpy legacy RandomState
streams are frozen by numpy's compatibility policy, so only the seeds are stored in the fixtures)."""
import numpy as np


def feats_from_seed(seed, B, R, D):
    rng = np.random.RandomState(seed)
    return (np.maximum(rng.randn(B, R, D), 0) * 0.5).astype(np.float32)


def masks_from_seed(seed, T, B, R, E, A, H):
    rng = np.random.RandomState(seed)
    em = (rng.rand(T, B, E) < 0.5).astype(np.uint8)
    am = (rng.rand(T, B, R, A) < 0.5).astype(np.uint8)
    om = (rng.rand(T, B, H) < 0.5).astype(np.uint8)
    u = rng.rand(T, B)
    return em, am, om, u


def probe_indices(numel, n=256, seed=7):
    """Fixed pseudo-random positions used to pin large tensors by a sample + moments."""
    rng = np.random.RandomState(seed + numel % 1000003)
    return rng.randint(0, numel, size=min(n, numel))


def pin_tensor(a, full_below=4096):
    """Compact fingerprint of a tensor: full copy if small, else (sample, sum, sumsq)."""
    a = np.asarray(a)
    if a.size <= full_below:
        return {"full": a.copy()}
    f = a.reshape(-1)
    idx = probe_indices(f.size)
    return {"sample": f[idx].copy(), "sum": np.float64(f.astype(np.float64).sum()),
            "sumsq": np.float64((f.astype(np.float64) ** 2).sum())}
