#!/usr/bin/env python3
"""Benchmark of the hot path: captions/sec of the BUTDDetection SCST step (BASELINE.json metric).

One "step" = Engine.SCST_training_epoch on one batch of 64 images per GPU: greedy baseline (20 steps) + sampled
rollout (20 steps, dropout on) + CIDEr-D reward + REINFORCE backward + clamp 0.25 + Adam, at the reference sizes
(36 x 2048 bottom-up features, hidden = embed = attention = 1024, COCO14-size vocabulary 10102, fp32).  Synthetic
features / references / random-init weights (no network), already resident in HBM when the timed region starts.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (see the keys below).  `roofline` is measured live with HIP events around every launch of
the dominant kernel (gemm_nt_kernel<4>, the forward GEMMs of the decoder step) during the timed steps;
`cpu_baseline` times the CPU oracle (our port of the reference path, oracle/) on a bounded sample on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

R, D, H, E, A, V, T = 36, 2048, 1024, 1024, 1024, 10102, 20
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3


def build_engine(device, B):
    from simpleimagecaptionzoo_amd.engine import BUTDDetection_Eng, init_optimizer
    from simpleimagecaptionzoo_amd.synth import document_frequency, random_butd_params, synthetic_references
    from simpleimagecaptionzoo_amd.vocab import synthetic_vocab
    vocab = synthetic_vocab(V)
    words = [vocab.ix2word[i] for i in range(V)]
    df = document_frequency(synthetic_references(2000, words, seed=0))
    eng = BUTDDetection_Eng({"model_type": "BUTDDetection", "atten_dim": A, "embed_dim": E, "hidden_dim": H},
                            "SYN", vocab, data_dir="/tmp/", use_bu="fixed", device=device, cider_df=df, max_batch=B)
    params = random_butd_params(R, D, H, E, A, V, device, seed=1234)      # identical on every rank (replicas)
    eng.model.load_state_dict({"decoder." + k: v for k, v in params.items()})
    opt = init_optimizer("Adam", eng.model.get_param_groups({"lr": 2e-5}), 2e-5)
    return eng, opt, vocab, words


def make_batches(n, B, words, device, rank):
    from simpleimagecaptionzoo_amd.synth import synthetic_references
    g = torch.Generator(device="cpu")
    batches = []
    for i in range(n):
        g.manual_seed(1234 + 1000 * rank + i)
        feats = torch.relu(torch.randn(B, R, D, generator=g)).to(device)
        ids = tuple(range((rank * n + i) * B, (rank * n + i + 1) * B))
        refs = synthetic_references(B, words, seed=77 + rank * n + i)
        gts = {ids[j]: refs[j] for j in range(B)}
        batches.append((ids, None, gts, {"bu_feats": feats}))
    return batches


def secondary(eng, opt, words, device, B):
    """The other rates SURVEY.md 8d lists next to the headline metric, on the same engine and model size (N = 1, a few
    hundred ms each): XE training step (Engine.training_epoch), greedy and beam-5 decode (eval_captions_json_generation's
    decoders; BASELINE config 3 quotes beam 5 at batch 128), and the AoADetection SCST step of config 5."""
    import time as _t
    out = {}
    rs = np.random.RandomState(0)
    Vn = len(words)
    xb = []
    for i in range(2):
        lens = sorted(rs.randint(9, 19, size=B).tolist(), reverse=True)        # caption lengths incl. <sta>/<end>
        caps = torch.zeros(B, max(lens), dtype=torch.int64)
        for b, n in enumerate(lens):
            caps[b, 0] = 1
            caps[b, 1:n - 1] = torch.from_numpy(rs.randint(4, Vn, size=n - 2))
            caps[b, n - 1] = 2
        feats = torch.relu(torch.randn(B, R, D, device=device))
        xb.append((tuple(range(B)), None, caps, lens, {"bu_feats": feats}))

    class _Crit:
        smoothing = 0.1

    def xe(n):
        eng.training_epoch([xb[i % 2] for i in range(n)], opt, _Crit(), tqdm_visible=False)
    xe(3)
    torch.cuda.synchronize()
    t0 = _t.perf_counter()
    xe(10)
    torch.cuda.synchronize()
    dt = (_t.perf_counter() - t0) / 10
    out["xe_step"] = {"captions_per_s": B / dt, "ms_per_step": dt * 1e3, "batch": B,
                      "note": "Engine.training_epoch, captions of 9..18 tokens, label smoothing 0.1"}
    # decoding on freshly initialised weights (the trained-for-a-few-steps ones above may or may not emit <end>: the beam
    # search stops early when every beam has finished, which would make the number depend on the training state)
    from simpleimagecaptionzoo_amd.butd import ButdHandle
    from simpleimagecaptionzoo_amd.synth import random_butd_params
    h = ButdHandle(R, D, H, E, A, V, 5 * B, 20)
    h.bind(random_butd_params(R, D, H, E, A, V, device, seed=1234))
    h.enable_graphs(True)
    with torch.cuda.stream(eng.stream):
        f64 = torch.relu(torch.randn(B, R, D, device=device))
        for name, fn, nimg in (("greedy", lambda: h.greedy(f64, 20), B), ("beam5", lambda: h.beam_search(f64, 5, 20), B)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = _t.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            dt = (_t.perf_counter() - t0) / 10
            out[name] = {"captions_per_s": nimg / dt, "ms": dt * 1e3, "batch": nimg, "steps": 20}
        out["beam5"]["note"] = "beam 5 = 5 decoder rows per image; random-init weights do not emit <end>, so all 20 steps run"
    h.close()
    # BASELINE config 3: beam 5 at batch 128 (640 decoder rows)
    h = ButdHandle(R, D, H, E, A, V, 5 * 128, 20)
    h.bind(random_butd_params(R, D, H, E, A, V, device, seed=1234))
    h.enable_graphs(True)
    with torch.cuda.stream(eng.stream):
        f128 = torch.relu(torch.randn(128, R, D, device=device))
        for _ in range(3):
            h.beam_search(f128, 5, 20)
        torch.cuda.synchronize()
        t0 = _t.perf_counter()
        for _ in range(10):
            h.beam_search(f128, 5, 20)
        torch.cuda.synchronize()
        dt = (_t.perf_counter() - t0) / 10
        out["beam5_b128"] = {"captions_per_s": 128 / dt, "ms": dt * 1e3, "batch": 128, "steps": 20}
    h.close()
    del h, f128, f64
    out["aoa_scst_step"] = aoa_scst(words, device, B)
    return out


def aoa_scst(words, device, B):
    """BASELINE config 5 at N = 1: AoADetection (6-layer refiner + AoA decoder) SCST step through AoADetection_Eng, same
    synthetic inputs, random-init weights (the reference's own initialisation), reward + clamp + Adam included."""
    import time as _t
    from simpleimagecaptionzoo_amd.engine import AoADetection_Eng, init_optimizer
    from simpleimagecaptionzoo_amd.synth import document_frequency, synthetic_references
    from simpleimagecaptionzoo_amd.vocab import synthetic_vocab
    vocab = synthetic_vocab(V)
    df = document_frequency(synthetic_references(2000, words, seed=0))
    eng = AoADetection_Eng({"model_type": "AoADetection", "embed_dim": E, "hidden_dim": H}, "SYN", vocab, data_dir="/tmp/",
                           use_bu="fixed", device=device, cider_df=df, max_batch=B)
    opt = init_optimizer("Adam", eng.model.get_param_groups({"lr": 2e-5}), 2e-5)
    batches = make_batches(2, B, words, device, 0)

    def run(n):
        eng.SCST_training_epoch([batches[i % 2] for i in range(n)], opt, None, tqdm_visible=False)
    run(3)
    torch.cuda.synchronize()
    t0 = _t.perf_counter()
    run(10)
    torch.cuda.synchronize()
    dt = (_t.perf_counter() - t0) / 10
    return {"captions_per_s": B / dt, "ms_per_step": dt * 1e3, "batch": B,
            "note": "AoADetection_Eng.SCST_training_epoch: refiner x2 (eval + train mode), greedy + sampled rollout, CIDEr-D reward, "
                    "REINFORCE backward of the decoder, clamp + Adam"}


def cpu_baseline(words, rows=8):
    """One SCST step of the CPU oracle (port of the reference path) at full model size on `rows` images."""
    from oracle import butd as ob
    from oracle import ciderd as oc
    from simpleimagecaptionzoo_amd.synth import document_frequency, random_butd_params, synthetic_references
    torch.manual_seed(0)
    p = {k: v.clone().requires_grad_(True) for k, v in random_butd_params(R, D, H, E, A, V, "cpu", seed=1234).items()}
    feats = torch.relu(torch.randn(rows, R, D))
    refs = synthetic_references(rows, words, seed=5)
    docfreq = oc.DocFreq(document_frequency(synthetic_references(2000, words, seed=0))["document_frequency"], 2000)
    ix2word = dict(enumerate(words))
    rng = np.random.RandomState(3)
    em = (rng.rand(T, rows, E) < 0.5)
    am = (rng.rand(T, rows, R, A) < 0.5)
    om = (rng.rand(T, rows, H) < 0.5)
    u = rng.rand(T, rows)
    opt = ob.Adam(p, 2e-5)
    t0 = time.time()
    with torch.no_grad():
        gre, _, _ = ob.greedy(feats, p, T)
    seq, lp, _ = ob.sample_rl(feats, p, u, em, am, om, T, early_exit=False)
    rew = oc.self_critical_reward(seq.numpy(), gre.numpy(), {i: refs[i] for i in range(rows)}, list(range(rows)), ix2word, docfreq)
    loss = ob.reward_criterion(lp, seq, torch.from_numpy(rew))
    grads = dict(zip(p.keys(), torch.autograd.grad(loss, list(p.values()))))
    opt.step(grads, 0.25)
    dt = time.time() - t0
    return {"value": rows / dt, "unit": "captions/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "1 SCST step of the CPU oracle (torch-CPU fp32 port of Engine.SCST_training_epoch), %d of the 64 "
                      "images of a batch, full model size, %.1f s" % (rows, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step (BASELINE config: 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the PCIe-inclusive extra measurement")
    ap.add_argument("--cpu-rows", type=int, default=64)
    args = ap.parse_args()

    from simpleimagecaptionzoo_amd import dist as icz_dist
    from simpleimagecaptionzoo_amd._lib import lib
    # ICZ_REHEARSE_ONE_GPU=1 (development only): every rank on cuda:0 with the gloo backend, to rehearse the N > 1 control
    # flow (normaliser all-reduce, gradient hook, barriers, rank-0 JSON) on a one-GPU box; the numbers mean nothing then
    rehearse = os.environ.get("ICZ_REHEARSE_ONE_GPU") == "1"
    rank, world, local = icz_dist.init_from_env("gloo" if rehearse else None)
    if rehearse:
        local = 0
    if args.gpus != world:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    device = "cuda:%d" % local
    torch.cuda.set_device(local)
    B = args.batch
    eng, opt, vocab, words = build_engine(device, B)
    n_distinct = 2
    batches = make_batches(n_distinct, B, words, device, rank)

    def run(n):
        loader = [batches[i % n_distinct] for i in range(n)]
        eng.SCST_training_epoch(loader, opt, None, tqdm_visible=False)

    run(max(args.warmup, n_distinct))          # also cooks + caches the references of every distinct batch
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # Live kernel timing for the roofline line: the same steps are run once more right here with a HIP event pair around
    # every launch of the dominant kernel on its launch stream -- eagerly (kernel nodes of a replayed graph cannot be
    # bracketed one by one) and with the side streams switched off, so that a pair measures the kernel alone, which is
    # also how rocprofv3 sees it (it serialises concurrent branches).
    avg_us, bpl, fpl, nl = C.c_double(), C.c_double(), C.c_double(), C.c_longlong()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    # every rank runs the same extra steps (they contain the gradient all-reduce); only rank 0 records events
    eng.use_graphs = False
    eng._hot_handle().set_concurrent(False)
    run(1)
    torch.cuda.synchronize()
    if rank == 0:
        lib().icz_prof_begin()
    run(3)
    torch.cuda.synchronize()
    pair_us = C.c_double()
    if rank == 0:
        lib().icz_prof_end(C.byref(avg_us), C.byref(bpl), C.byref(fpl), C.byref(nl))
        with torch.cuda.stream(eng.stream):
            lib().icz_prof_pair_overhead(C.c_void_p(eng.stream.cuda_stream), 64, C.byref(pair_us))
    # PCIe-inclusive variant (never `value`): the same steps with the features starting in host memory, as the reference
    # boundary hands them over (BUTD_Engine.py:45), streamed through the pinned double-buffered prefetcher
    pcie = None
    if rank == 0 and world == 1 and not args.no_h2d:
        from simpleimagecaptionzoo_amd.features import DevicePrefetcher
        eng.use_graphs = True
        eng._hot_handle().set_concurrent(True)
        host = []
        for ids, _, gts, supp in batches:
            f = supp["bu_feats"].cpu().numpy()
            host.append((ids, None, gts, tuple({"bu_feat": f[j], "bu_bbox": None} for j in range(f.shape[0]))))

        def run_h2d(n):
            eng.SCST_training_epoch(DevicePrefetcher([host[i % n_distinct] for i in range(n)], device), opt, None, tqdm_visible=False)
        run_h2d(4)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_h2d(args.steps)
        torch.cuda.synchronize()
        dth = time.perf_counter() - t1
        pcie = {"value": B * args.steps / dth, "unit": "captions/s", "ms_per_step": dth / args.steps * 1e3,
                "note": "features start in host memory: gather into pinned buffers + async H2D (18.9 MB per batch) overlapped with the previous step"}
    if world > 1:
        torch.distributed.barrier()
    if rank != 0:
        return
    value = world * B * args.steps / dt
    # `achieved` uses the raw event-pair time (conservative: a pair also spans the dispatch of the bracketed kernel).  For
    # the comparison with rocprofv3's kernel trace (profiles/) the line also carries the pair time around an EMPTY kernel:
    # pair(empty) = dispatch + records + ~1.7 us of empty-kernel execution, so kernel time ~ pair - (pair(empty) - 1.7).
    kern_us = avg_us.value
    ach_gbs = bpl.value / (kern_us * 1e-6) / 1e9 if kern_us > 0 else 0.0
    ach_tf = fpl.value / (kern_us * 1e-6) / 1e12 if kern_us > 0 else 0.0
    # HBM bytes per launch of the same kernel from the PMC passes of this command (FETCH_SIZE x2 + WRITE_SIZE, gfx950
    # corrections applied by tools/pmc_summary.py); PMC counters cannot be read from inside the process
    traffic = None
    try:
        pmc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")))
        nt = [v for k, v in pmc["kernels"].items() if "gemm_nt_kernel" in k]
        traffic = max(nt, key=lambda v: v["launches"])["hbm_bytes_per_launch"]      # the decoder-step instantiation
    except Exception:
        pass
    # Both rooflines of the dominant kernel (report both fractions, name the binding one).  At 64 rows the gate
    # GEMMs do 32 flop per weight byte, above the fp32 ridge of the part (157.3 TF / 8 TB/s = 20 flop/B): the roofline time of
    # a launch is max(bytes / HBM peak, flops / fp32-MFMA peak) and `bound` names the larger term.
    t_hbm, t_mfma = bpl.value / (HBM_PEAK_GBS * 1e9), fpl.value / (MFMA_F32_PEAK_TFLOPS * 1e12)
    roof = {"kernel": "gemm_nt_kernel<4> (decoder-step forward GEMMs: LSTM gates, dec_att, predict, prologue)"}
    if t_mfma >= t_hbm:
        roof.update({"bound": "mfma", "achieved": ach_tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / MFMA_F32_PEAK_TFLOPS})
    else:
        roof.update({"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS})
    roof.update({"traffic": traffic,
                 "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc passes of this command)" if traffic else None,
                 "avg_launch_us": kern_us, "empty_kernel_pair_us": pair_us.value, "launches": nl.value,
                 "bytes_per_launch": bpl.value, "flops_per_launch": fpl.value,
                 "roofline_us_per_launch": {"hbm": t_hbm * 1e6, "mfma_f32": t_mfma * 1e6},
                 "measured": "HIP event pair around every launch; eager single-stream re-run of the same steps right after the timed region",
                 "hbm_gbs": ach_gbs, "hbm_frac": ach_gbs / HBM_PEAK_GBS, "mfma_f32_tflops": ach_tf,
                 "mfma_f32_frac": ach_tf / MFMA_F32_PEAK_TFLOPS})
    out = {
        "metric": "captions/sec (SCST step), BUTDDetection COCO14-size vocab",
        "value": value, "unit": "captions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "arithmetic": "fp32 throughout (fp32-input MFMA / VALU, float64 CIDEr-D); the 128 x 128-tile GEMMs (weight gradients, the "
                      "dgrad over all time steps, forward GEMMs of 128+ rows) multiply fp32 operands as three bf16 pieces each (24 "
                      "mantissa bits, six bf16 MFMAs per product, fp32 accumulation): fp32-level error, inside the same parity bounds "
                      "(tests/); ICZ_GEMM_TN_X3=0 ICZ_GEMM_NN_X3=0 ICZ_GEMM_NT_X3BIG=0 select the fp32-MFMA kernels",
        "config": {"workload": "BUTDDetection SCST step (greedy + sampled rollout + CIDEr-D reward + REINFORCE backward "
                               "+ clamp + Adam), batch %d per GPU, 36x2048 features, H=E=A=1024, V=10102, 20 decode steps" % B,
                   "global_batch": world * B, "parallelism": "dp%d" % world},
        "roofline": roof,
    }
    if pcie:
        out["pcie_inclusive"] = pcie
    if world == 1 and not args.no_h2d:
        try:
            eng.use_graphs = True
            eng._hot_handle().set_concurrent(True)
            out["secondary"] = secondary(eng, opt, words, device, B)
        except Exception as e:      # the headline line must not depend on the extras
            out["secondary"] = {"error": repr(e)}
    if not args.no_cpu_baseline and world == 1:       # rank 0 at N = 1 only: the N > 1 runs share the host with the other ranks
        out["cpu_baseline"] = cpu_baseline(words, args.cpu_rows)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
