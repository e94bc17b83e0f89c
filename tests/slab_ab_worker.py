"""Child process of tests/test_gpu_gemm.py::test_predict_slab_path_matches_the_unsplit_gemm: decodes 64 rows at full width
with BUTD, AoA and NIC (greedy + Philox-seeded sampled rollout + REINFORCE gradients of the output layer) and writes the results to
an .npz -- together with the explicit uniforms of the draws and the logits of a teacher-forced replay of the sampled rows (same
Philox dropout streams), so that the parent can hold a differing draw to the CDF-edge criterion.  The parent runs it twice, with ICZ_PREDICT_SLABS=1 (default: the vocabulary projection leaves split-K slabs that
the argmax / multinomial kernels sum) and =0 (un-split GEMM, finished logits); the switch is read once per process."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
R, D, H, E, A, V = 36, 2048, 1024, 1024, 1024, 10102
B, T = 64, 6


def replay_logits(handle, feats, seq, rng):
    """[T, B, V] training-mode logits of the sampled rows: <sta> + the drawn tokens fed back (all rows T steps: packed order = time-major)"""
    caps = torch.cat([torch.ones(B, 1, dtype=torch.int64, device=seq.device), seq], 1)
    return handle.xe_forward(feats, caps, [T] * B, rng, train=True, want_logits=True).view(T, B, -1).cpu().numpy()


def main(out_path):
    from simpleimagecaptionzoo_amd.aoa import AoADetection_Captioner, make_aoa_rng
    from simpleimagecaptionzoo_amd.butd import ButdHandle, make_rng
    from simpleimagecaptionzoo_amd.nic import NICDecoder_Captioner
    from simpleimagecaptionzoo_amd.nic import make_rng as make_nic_rng
    from simpleimagecaptionzoo_amd.synth import random_butd_params
    out = {}
    torch.manual_seed(123)
    feats = torch.relu(torch.randn(B, R, D, device="cuda"))
    reward = torch.randn(B, T, device="cuda")
    # ---- BUTD
    params = random_butd_params(R, D, H, E, A, V, "cuda", seed=31)
    params["predict.weight_g"].mul_(6.0)
    h = ButdHandle(R, D, H, E, A, V, B, 20)
    h.bind(params)
    out["butd_greedy"] = h.greedy(feats, T).cpu().numpy()
    u = torch.rand(T, B, device="cuda")
    out["u"] = u.cpu().numpy()
    seq, lp = h.sample(feats, T, make_rng(77, u))
    out["butd_seq"], out["butd_lp"] = seq.cpu().numpy(), lp.cpu().numpy()
    g = h.new_grads()
    loss, _ = h.sample_backward(reward, g)
    out["butd_loss"] = np.float32(loss.item())
    out["butd_dbias"] = g["predict.bias"].cpu().numpy()
    out["butd_logits"] = replay_logits(h, feats, seq.clone(), make_rng(77))
    del h
    # ---- AoA
    cap = AoADetection_Captioner(V, max_batch=B, max_beam=1).cuda()
    with torch.no_grad():
        cap.decoder.predict.weight_g.mul_(6.0)
    ha = cap._handle()
    out["aoa_greedy"] = ha.greedy(feats, T).cpu().numpy()
    seq, lp = ha.sample(feats, T, make_aoa_rng(78, u))
    out["aoa_seq"], out["aoa_lp"] = seq.cpu().numpy(), lp.cpu().numpy()
    g = ha.new_grads()
    ha.sample_backward(reward, g)
    out["aoa_dbias"] = g["decoder.predict.bias"].cpu().numpy()
    out["aoa_logits"] = replay_logits(ha, feats, seq.clone(), make_aoa_rng(78))
    del ha, cap
    # ---- NIC
    nic = NICDecoder_Captioner(E, H, V, max_batch=B, max_beam=1).cuda()
    with torch.no_grad():
        nic.decoder.predict.weight_g.mul_(6.0)
    hn = nic._handle()
    img = torch.randn(B, E, device="cuda")
    out["nic_greedy"] = hn.greedy(img, T).cpu().numpy()
    seq, lp = hn.sample(img, T, make_nic_rng(79, u))
    out["nic_seq"], out["nic_lp"] = seq.cpu().numpy(), lp.cpu().numpy()
    g = hn.new_grads()
    hn.sample_backward(reward, g)
    out["nic_dbias"] = g["predict.bias"].cpu().numpy()
    out["nic_logits"] = replay_logits(hn, img, seq.clone(), make_nic_rng(79))
    np.savez(out_path, **out)


if __name__ == "__main__":
    main(sys.argv[1])
