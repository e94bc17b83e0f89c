"""Worker of tests/test_gpu_dist_two_ranks.py: one of WORLD_SIZE data-parallel ranks on the SAME GPU (gloo backend, development
rehearsal of the one-process-per-GPU RCCL path).  Each rank takes its dist.shard_range share of the golden Engine batches (two
ranks: halves; five ranks: 2 1 1 1 1 of the six images); after every step the parameters must equal what the reference Engine
produced on the whole batch in one process."""
import os
import sys

import numpy as np
import torch
import torch.distributed as td

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, HERE)

from synth import feats_from_seed, masks_from_seed  # noqa: E402
import test_gpu_engine as tge  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    td.init_process_group(backend="gloo", rank=rank, world_size=world)
    from simpleimagecaptionzoo_amd.butd import make_rng
    from simpleimagecaptionzoo_amd.engine import init_optimizer
    g, fx = tge._load(os.path.join(HERE, "golden"))
    eng, vocab = tge._engine(g, fx)
    B, R, D, H, E, A, V = [int(x) for x in g["dims"]]
    from simpleimagecaptionzoo_amd import dist as icz_dist
    lo, hi = icz_dist.shard_range(B, rank, world)
    assert hi > lo
    dev = "cuda"

    def rng_of(seed, T, with_u):
        em, am, om, u = masks_from_seed(seed, T, B, R, E, A, H)
        cut = lambda x: torch.tensor(np.ascontiguousarray(x[:, lo:hi]), device=dev)
        return make_rng(0, torch.tensor(np.ascontiguousarray(u[:, lo:hi]), dtype=torch.float32, device=dev) if with_u else None,
                        cut(em), cut(am), cut(om))

    def total(loss):
        t = loss.detach().float().cpu().view(1).clone()
        td.all_reduce(t)
        return float(t.item())

    # G3: evaluation sharded over the ranks batch by batch, rows all-gathered: every rank must return the JSON
    # the reference Engine produced in one process, in loader order (batches of 3 / 1 images: unequal shares)
    ef = feats_from_seed(int(g["eval_feats_seed"]), B, R, D)
    eids = tuple(int(i) for i in g["eval_img_ids"])
    loader3 = [(eids[i:i + 3], None, tge._supp(ef[i:i + 3])) for i in range(0, B, 3)]
    loader1 = [(eids[i:i + 1], None, tge._supp(ef[i:i + 1])) for i in range(B)]
    assert eng.eval_captions_json_generation(loader3, eval_beam_size=-1, tqdm_visible=False) == fx["eval_greedy_json"]
    assert eng.eval_captions_json_generation(loader1, eval_beam_size=3, tqdm_visible=False) == fx["eval_beam3_json"]

    opt = init_optimizer("Adam", eng.model.get_param_groups({"lr": 4e-4}), 4e-4)
    for s in range(2):
        pre = "xe%d_" % s
        feats = feats_from_seed(int(g[pre + "feats_seed"]), B, R, D)[lo:hi]
        caps = torch.tensor(g[pre + "captions"])[lo:hi]
        lens_all = [int(x) for x in g[pre + "lengths"]]
        lens = lens_all[lo:hi]
        # every rank runs max(lengths) - 1 steps of masks in the reference layout: the shard's own longest caption may be shorter
        rng = rng_of(int(g[pre + "mask_seed"]), max(lens_all) - 1, False)
        batch = (tuple(range(lo, hi)), None, caps, lens, tge._supp(feats))
        losses = eng.training_epoch([batch], opt, tge._Crit(), tqdm_visible=False, rngs=[rng])
        torch.cuda.synchronize()
        assert abs(total(losses[0]) - float(g[pre + "loss"])) < 1e-4, (pre, total(losses[0]), float(g[pre + "loss"]))
        tge._check_pinned(g, pre + "sd.", eng.model, slack=4e-4 * (s + 1))
    opt = init_optimizer("Adam", eng.model.get_param_groups({"lr": 2e-5}), 2e-5)
    for s in range(2):
        pre = "rl%d_" % s
        feats = feats_from_seed(int(g[pre + "feats_seed"]), B, R, D)[lo:hi]
        rng = rng_of(int(g[pre + "mask_seed"]), 20, True)
        img_ids = tuple(int(i) for i in g[pre + "img_ids"])[lo:hi]
        gts = {int(k): v for k, v in fx[pre + "gts"].items()}
        batch = (img_ids, None, gts, tge._supp(feats))
        losses = eng.SCST_training_epoch([batch], opt, None, tqdm_visible=False, rngs=[rng])
        torch.cuda.synchronize()
        assert abs(total(losses[0]) - float(g[pre + "loss"])) < 1e-4, (pre, total(losses[0]), float(g[pre + "loss"]))
        tge._check_pinned(g, pre + "sd.", eng.model, slack=8e-4 + 2e-5 * (s + 1))
    td.barrier()
    td.destroy_process_group()
    print("rank %d ok" % rank)


main()
