"""Behaviour of the AoA handle around the kernels: graph replay = eager launches, device-side early-out, the paired
refiner pass, the refiner's self-attention on the matrix pipe, graph caches against regrown training buffers.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from _fullwidth import (D, E, H, R, V)  # noqa: E402

pytestmark = pytest.mark.gpu


# ---- AoA: hipGraph replay of the SCST rollout pair and of the backward pass, shared feature projection -------------------------
def test_aoa_rollouts_and_backward_under_graph_replay_equal_eager_launches(golden_dir):
    """AoADetection SCST step (AoA_Model.py:698-753 behind Engine.py:256-270) at full width: the rollout pair and the REINFORCE
    backward as replayed hipGraphs against eager launches -- same Philox seeds, same inputs: tokens, log-probs, loss and every
    decoder gradient bit for bit; and the greedy ids / sampled rollout of rollouts() (ONE feature projection for both refiner
    passes) equal greedy() + sample() (each with its own)."""
    import os
    from simpleimagecaptionzoo_amd.aoa import AoADetection_Captioner, make_aoa_rng
    B, T = 16, 20
    torch.manual_seed(7)
    cap = AoADetection_Captioner(vocab_size=V, num_heads=8, hidden_dim=H, embed_dim=E, device="cuda:0", num_regions=36, enc_dim=D,
                                 max_batch=B)
    cap.to("cuda:0")
    feats = torch.relu(torch.randn(B, 36, D, device="cuda"))
    out = {}
    for graphs in (False, True):
        h = cap._handle()
        h.enable_graphs(graphs)
        res = []
        grads = h.new_grads()
        rew = torch.linspace(-1, 1, B, device="cuda").unsqueeze(1).repeat(1, T).contiguous()
        for rep in range(3):                                       # the first call captures, the others replay
            ids, seq, lp = h.rollouts(feats, T, make_aoa_rng(500 + rep))
            loss, _ = h.sample_backward(rew, grads)
            res.append((ids.clone(), seq.clone(), lp.clone(), loss.clone(), {k: v.clone() for k, v in grads.items()}))
        out[graphs] = res
    for a, b in zip(out[False], out[True]):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
        for k in a[4]:
            assert torch.equal(a[4][k], b[4][k]), k
    h = cap._handle()
    h.enable_graphs(False)
    ids0 = h.greedy(feats, T)
    seq0, lp0 = h.sample(feats, T, make_aoa_rng(500))
    # (since the paired refiner pass: at 16 images its GEMMs of 2 x 576 rows take another split-K decomposition than the single passes of
    # greedy() / sample() -- tokens equal, log-probs within fp32 rounding; test_aoa_paired_refiner_pass_equals_the_two_passes has the bits)
    assert torch.equal(ids0, out[False][0][0]) and torch.equal(seq0, out[False][0][1])
    assert (lp0 - out[False][0][2]).abs().max().item() < 1e-4


@pytest.mark.parametrize("B,bias", [(4, 4.0), (48, 7.0)])
def test_aoa_early_out_equals_running_every_step(golden_dir, B, bias):
    """AoA_Decoder.sample_rl's break (AoA_Model.py:400) on the device: with the steps behind it returning at entry (default) and with
    every step run as rounds 1 - 4 did (option early_out = 0; that form is pinned by the reference goldens in test_gpu_aoa.py):
    greedy prefix, sampled ids, log-probs, loss and the decoder gradients agree (the batched GEMMs stop behind the last live step:
    trailing all-zero rows dropped from the same sums)."""
    import os
    from simpleimagecaptionzoo_amd.aoa import AoaHandle, make_aoa_rng
    g = dict(np.load(os.path.join(golden_dir, "aoa_tiny.npz")))
    _, Rr, Dd, Hd, Ee, Vv, NH = [int(x) for x in g["dims"]]
    sd = {k[3:]: v.copy() for k, v in g.items() if k.startswith("sd.")}
    sd["decoder.predict.bias"][2] = bias
    params = {k: torch.tensor(np.asarray(v), dtype=torch.float32, device="cuda") for k, v in sd.items()}
    gen = torch.Generator(device="cpu")
    gen.manual_seed(B)
    feats = torch.relu(torch.randn(B, Rr, Dd, generator=gen)).cuda()
    T = 20
    out = {}
    for eo in (0, 1):
        h = AoaHandle(Rr, Dd, Hd, Ee, Vv, NH, B, T)
        h.bind(params)
        check = __import__("simpleimagecaptionzoo_amd._lib", fromlist=["check"])
        check.check(check.lib().icz_aoa_set_option(h._h, b"early_out", eo))
        ids, seq, lp = h.rollouts(feats, T, make_aoa_rng(77))
        grads = h.new_grads()
        for v in grads.values():
            v.fill_(float("nan"))
        rew = torch.linspace(-1, 1, B, device="cuda").unsqueeze(1).repeat(1, T).contiguous()
        loss, _ = h.sample_backward(rew, grads)
        out[eo] = (ids.cpu().numpy(), seq.cpu().numpy(), lp.cpu().numpy(), loss.item(), {k: v.cpu().numpy() for k, v in grads.items()})
        h.close()
    a, b = out[1], out[0]
    assert (b[1][:, -1] == 0).all() and (b[1][:, 0] != 0).any()          # every sampled row ended before the last step
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3]
    for r in range(B):
        e = np.nonzero(b[0][r] == 2)[0]
        n = e[0] + 1 if e.size else T
        assert np.array_equal(a[0][r, :n], b[0][r, :n]), r
    for k in a[4]:
        assert np.isfinite(a[4][k]).all(), k
        scale = float(np.abs(b[4][k]).max()) + 1e-12
        assert float(np.abs(a[4][k] - b[4][k]).max()) <= 1e-5 * scale, (k, float(np.abs(a[4][k] - b[4][k]).max()), scale)


@pytest.mark.parametrize("B", [64, 16])
def test_aoa_paired_refiner_pass_equals_the_two_passes(golden_dir, B):
    """icz_aoa_scst_rollouts with both refiner passes as ONE pass over [evaluation rows; training rows] (option refine_pair, default)
    against two passes: greedy ids and sampled ids equal; at the BASELINE batch (64 images: every GEMM of the pair takes the split-K
    decomposition of the single passes) log-probs, loss and every decoder gradient bit for bit, at 16 images within fp32 rounding."""
    from simpleimagecaptionzoo_amd.aoa import AoADetection_Captioner, make_aoa_rng
    T = 20
    torch.manual_seed(11)
    cap = AoADetection_Captioner(vocab_size=V, num_heads=8, hidden_dim=H, embed_dim=E, device="cuda:0", num_regions=36, enc_dim=D,
                                 max_batch=B)
    cap.to("cuda:0")
    feats = torch.relu(torch.randn(B, 36, D, device="cuda"))
    rew = torch.linspace(-1, 1, B, device="cuda").unsqueeze(1).repeat(1, T).contiguous()
    out = {}
    for pair in (0, 1):
        h = cap._handle()
        h.enable_graphs(False)
        h.set_option("refine_pair", pair)
        grads = h.new_grads()
        ids, seq, lp = h.rollouts(feats, T, make_aoa_rng(900))
        loss, _ = h.sample_backward(rew, grads)
        out[pair] = (ids.clone(), seq.clone(), lp.clone(), loss.clone(), {k: v.clone() for k, v in grads.items()})
    a, b = out[0], out[1]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    if B == 64:
        assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
        for k in a[4]:
            assert torch.equal(a[4][k], b[4][k]), k
    else:
        assert (a[2] - b[2]).abs().max().item() < 1e-4 and abs(a[3].item() - b[3].item()) < 1e-5
        for k in a[4]:
            d = (a[4][k] - b[4][k]).abs().max().item()
            assert d <= 2e-5 * max(a[4][k].abs().max().item(), 1e-3), (k, d)
    cap._handle().set_option("refine_pair", 1)


@pytest.mark.parametrize("R,counts", [(36, None), (49, None), (36, "ragged"), (64, "ragged")])
def test_aoa_refiner_self_attention_on_the_matrix_pipe_equals_the_blocked_kernel(golden_dir, R, counts):
    """mha_self_mfma_kernel (option mha_mfma, default for <= 64 regions) against the register-blocked kernel that the reference goldens
    of test_gpu_aoa.py / test_gpu_aoa_adaptive.py pinned in rounds 1 - 4: refined regions of six layers within 2e-5 of max|x|, with fixed
    region sets (36 boxes, 7 x 7 grid) and with per-image counts (packed rows, masked keys)."""
    from simpleimagecaptionzoo_amd.aoa import AoADetection_Captioner, RegionBatch
    B = 12
    torch.manual_seed(3)
    cap = AoADetection_Captioner(vocab_size=V, num_heads=8, hidden_dim=H, embed_dim=E, device="cuda:0", num_regions=R, enc_dim=D, max_batch=B)
    cap.to("cuda:0")
    feats = torch.relu(torch.randn(B, R, D, device="cuda"))
    batch = feats
    if counts == "ragged":
        c = [R, 10, R - 1, 17, 16, 15, 1, 33 if R > 33 else R, R, 12, 20, 2]
        m = (torch.arange(R, device="cuda").unsqueeze(0) < torch.tensor(c, device="cuda").unsqueeze(1))
        batch = RegionBatch(feats * m.unsqueeze(2), c)
    h = cap._handle()
    out = {}
    for on in (0, 1):
        h.set_option("mha_mfma", on)
        out[on] = h.refine(batch).clone()
    h.set_option("mha_mfma", 1)
    assert torch.isfinite(out[1]).all()
    assert (out[0] - out[1]).abs().max().item() <= 2e-5 * out[0].abs().max().item()


def test_aoa_graphs_survive_training_buffers_that_regrow(golden_dir):
    """Aoa::ensure_train frees and re-allocates every training buffer when a longer XE batch arrives; the captured rollout /
    backward graphs carried the freed addresses.  Capture an SCST step at (B, T), grow the buffers through xe_forward with captions
    longer than T, then repeat the first SCST call (an already-seen key): tokens, log-probs, loss and gradients must equal the eager
    handle's, bit for bit."""
    from simpleimagecaptionzoo_amd.aoa import AoADetection_Captioner, make_aoa_rng
    B, T, L = 6, 20, 31
    torch.manual_seed(11)
    cap = AoADetection_Captioner(vocab_size=V, num_heads=8, hidden_dim=H, embed_dim=E, device="cuda:0", num_regions=36, enc_dim=D,
                                 max_batch=B)
    cap.to("cuda:0")
    feats = torch.relu(torch.randn(B, 36, D, device="cuda"))
    caps = torch.randint(4, V, (B, L), device="cuda")
    caps[:, 0] = 1
    lengths = [L - 1, L - 2, L - 4, 12, 9, 5]
    rew = torch.linspace(-1, 1, B, device="cuda").unsqueeze(1).repeat(1, T).contiguous()

    def scst(h, seed):
        grads = h.new_grads()
        ids, seq, lp = h.rollouts(feats, T, make_aoa_rng(seed))
        loss, _ = h.sample_backward(rew, grads)
        return ids.clone(), seq.clone(), lp.clone(), loss.clone(), {k: v.clone() for k, v in grads.items()}
    out = {}
    for graphs in (True, False):                   # the captioner's handle is one object: the replayed pass first, while its buffers are small
        h = cap._handle()
        h.enable_graphs(graphs)
        first = scst(h, 900)                       # graphs: captured here
        scst(h, 901)                               # ... and replayed once
        h.xe_forward(feats, caps, lengths, make_aoa_rng(3))       # 30 steps > T: every training buffer is re-allocated
        h.xe_backward(h.new_grads())
        torch.cuda.synchronize()
        junk = [torch.full((1 << 22,), float("nan"), device="cuda") for _ in range(8)]      # land on the freed blocks if the allocator hands them out
        again = scst(h, 900)
        del junk
        out[graphs] = (first, again)
    for a, b in zip(out[False], out[True]):
        for i in range(4):
            assert torch.equal(a[i], b[i]), i
        for k in a[4]:
            assert torch.equal(a[4][k], b[4][k]), k
    for i in range(4):                              # and the repeated call reproduces the first one on each handle
        assert torch.equal(out[True][0][i], out[True][1][i]), i
