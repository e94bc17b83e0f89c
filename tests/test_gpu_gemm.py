"""The GEMM kernels behind the path on their own: the large-tile split-precision kernel in every configuration, the
grouped weight-gradient launch, the vocabulary projection's slab path -- each against float64 / the kernel it replaces.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from _fullwidth import (BIG_SHAPES, H, _first_divergence, _gemm_operands, _gemm_ref64)  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5])
def test_large_tile_gemm_configurations_against_float64_and_the_128_tile_kernel(cfg):
    """Same arithmetic in the same order per accumulator: for one split-K decomposition every tile configuration returns the bits of
    gemm_tn128_x3_kernel; all of them within 3e-6 of max|C| of the float64 product (the bound of tests/test_gpu_butd.py)."""
    from simpleimagecaptionzoo_amd.butd import gemm, gemm_set_big_cfg
    try:
        for (lay, M, N, K, ns) in BIG_SHAPES:
            X, W = _gemm_operands(lay, M, N, K, 17 * M + N + K)
            gemm_set_big_cfg(0)
            base = gemm(lay, X, W, None, ns)
            gemm_set_big_cfg(cfg)
            out = gemm(lay, X, W, None, ns)
            ref = _gemm_ref64(lay, X, W)
            err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
            assert err < 3e-6, (lay, M, N, K, ns, cfg, err)
            assert torch.equal(out, base), (lay, M, N, K, ns, cfg, (out - base).abs().max().item())
    finally:
        gemm_set_big_cfg(-2)


def test_large_tile_gemm_with_bias_and_per_shape_choice():
    """The routed default (cfg -1) with a bias on the direct path, the shapes of an XE step's vocabulary projection and of the refiner."""
    from simpleimagecaptionzoo_amd.butd import gemm
    for (M, N, K) in ((1280, 10102, 1024), (2304, 2048, 2048), (1088, 10102, 1024)):
        X, W = _gemm_operands("nt", M, N, K, M + N)
        b = torch.randn(N, device="cuda")
        out = gemm("nt", X, W, b, 1)
        ref = X.double() @ W.double().t() + b.double()
        assert ((out.double() - ref).abs().max() / ref.abs().max()).item() < 3e-6


@pytest.mark.parametrize("cfg", [-1, 1, 4])
def test_grouped_weight_gradients_equal_the_separate_products(cfg):
    """icz_gemm_tn_grouped (the LSTM weight gradients of Butd::bptt as one launch over column groups) = the products one by one, bit for
    bit, into strided outputs (W_ih column blocks and W_hh), also with the row limit of an early-ended rollout."""
    from simpleimagecaptionzoo_amd.butd import gemm, gemm_set_big_cfg, gemm_tn_grouped
    H, K = 1024, 640
    g = torch.Generator(device="cuda").manual_seed(5)
    dY = torch.randn(K, 4 * H, device="cuda", generator=g)
    Xs = [torch.randn(K, 2 * H, device="cuda", generator=g), torch.randn(K, H, device="cuda", generator=g), torch.randn(K, H, device="cuda", generator=g)]
    try:
        gemm_set_big_cfg(cfg)
        w_ih = torch.full((4 * H, 3 * H + 8), 7.0, device="cuda")        # [ctx | h1] and 8 columns nobody may touch
        w_hh = torch.zeros(4 * H, H, device="cuda")
        outs = [w_ih[:, :2 * H], w_ih[:, 2 * H:3 * H], w_hh]
        gemm_tn_grouped(dY, Xs, outs)
        assert torch.all(w_ih[:, 3 * H:] == 7.0)
        for x, o in zip(Xs, outs):
            ref = dY.double().t() @ x.double()
            assert ((o.double() - ref).abs().max() / ref.abs().max()).item() < 3e-6
        if cfg > 0:
            for x, o in zip(Xs, outs):
                # the same kernel on the product alone (4096 x 2048 and 4096 x 1024 both have >= 256 tiles of 128 x 128)
                assert torch.equal(o.contiguous(), gemm("tn", dY, x, None, 1))
        # rows behind the live count hold finite values whose products must not be read: the sum stops at 200 -> 224 rows
        live = torch.tensor([200], device="cuda", dtype=torch.int32)
        outs2 = gemm_tn_grouped(dY, Xs, None, live)
        for x, o in zip(Xs, outs2):
            ref = dY[:224].double().t() @ x[:224].double()
            assert ((o.double() - ref).abs().max() / ref.abs().max()).item() < 3e-6
    finally:
        gemm_set_big_cfg(-2)


def test_predict_slab_path_matches_the_unsplit_gemm(tmp_path):
    """At 33 - 64 rows the vocabulary projection of a decoder step goes through the resident-activation kernel and leaves four
    split-K slabs that the argmax / multinomial kernels sum (gemm_predict, gemm_resident_x3.hip); ICZ_PREDICT_SLABS=0 keeps the
    un-split GEMM with finished logits.  Same 64 rows, same Philox seeds, BUTD / AoA / NIC at full width, one child process per
    setting (the switch is read once per process): identical greedy tokens; sampled tokens (explicit uniforms, Philox dropout)
    identical except where the draw's target u * sum(p) lies within 1e-6 of a CDF edge of the float64 softmax of that step's
    logits (the two GEMMs differ by 1.4e-6 rms in the logits, tools/dbg_pred_err.py) -- the criterion of the oracle tests, with the
    logits taken from a teacher-forced replay of the sampled rows, which is first checked to reproduce the rollout's log-probs;
    log-probs of the drawn tokens within 3e-5 on the rows that agree; the gradient of the output bias (built from the saved
    logits the multinomial kernel writes) within 1e-6 when every row agrees."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    for flag in ("1", "0"):
        out = str(tmp_path / ("slab%s.npz" % flag))
        env = dict(os.environ, ICZ_PREDICT_SLABS=flag)
        r = subprocess.run([sys.executable, os.path.join(here, "slab_ab_worker.py"), out], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        res[flag] = dict(np.load(out))
    a, b = res["1"], res["0"]
    u = a["u"]
    assert np.array_equal(u, b["u"])
    for fam in ("butd", "aoa", "nic"):
        assert np.array_equal(a[fam + "_greedy"], b[fam + "_greedy"]), fam
        sa, sb = a[fam + "_seq"], b[fam + "_seq"]
        # the replay reproduces the rollout: log-softmax of its logits at the drawn token = the rollout's log-prob (steps before a row finished)
        lg = torch.from_numpy(a[fam + "_logits"]).double()                      # [T, B, V]
        lsm = torch.log_softmax(lg, 2)
        live = np.concatenate([np.ones((sa.shape[0], 1), bool), np.cumsum(sa[:, :-1] == 0, 1) == 0], 1) & (sa > 0)
        got_lp = lsm.permute(1, 0, 2).gather(2, torch.from_numpy(sa).unsqueeze(2)).squeeze(2).numpy()
        np.testing.assert_allclose(got_lp[live], a[fam + "_lp"][live], atol=1e-4, err_msg=fam + ": replay")
        first = _first_divergence(sa, sb)
        rows = np.where(first >= 0)[0]
        assert len(rows) <= 4, (fam, rows)      # (each of them must sit on a CDF edge, below; 2 with 256-deep k ranges, 3 with the 512-deep ones of round 6)
        for r_ in rows:
            t = first[r_]
            c = torch.cumsum(torch.softmax(lg[t, r_], 0), 0)
            tgt = float(u[t, r_]) * float(c[-1])
            assert float((c - tgt).abs().min()) < 1e-6, (fam, r_, t, "draws differ away from a CDF edge", sa[r_], sb[r_])
        same = first < 0
        np.testing.assert_allclose(a[fam + "_lp"][same], b[fam + "_lp"][same], atol=3e-5, err_msg=fam)
        if same.all():
            np.testing.assert_allclose(a[fam + "_dbias"], b[fam + "_dbias"], atol=1e-6, err_msg=fam)
        assert (sa > 0).any() and np.isfinite(a[fam + "_lp"]).all()
