"""gloo tests (world 2 and world 8) of the data-parallel exchange rules using the CPU oracle as the per-rank
compute: shard the batch, all-reduce the normaliser BEFORE backward, all-reduce the gradients, clamp AFTER the
reduction -- the result must equal the single-process step on the concatenated batch.  The 8-rank case is the rehearsal of the
driver's N = 8 run that fits this container and the GPU box (a box admits six GPU processes: the ranks that go through the real
Engine on the card stop at five, tests/test_gpu_dist_two_ranks.py): 11 images over 8 ranks = uneven shards (2 2 2 1 1 1 1 1),
one Philox seed per rank, caption rows gathered back in loader order from ranks >= 2."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(B=6):
    from oracle import butd as ob
    from simpleimagecaptionzoo_amd.synth import random_butd_params
    R, D, H, E, A, V, T = 36, 32, 16, 16, 16, 40, 8
    p = random_butd_params(R, D, H, E, A, V, "cpu", seed=5)
    p["embed.0.weight"] = p["embed.0.weight"] * 30
    p["predict.weight_g"] = p["predict.weight_g"] * 10
    torch.manual_seed(0)
    feats = torch.relu(torch.randn(B, R, D))
    rs = np.random.RandomState(2)
    masks = ((rs.rand(T, B, E) < 0.5), (rs.rand(T, B, R, A) < 0.5), (rs.rand(T, B, H) < 0.5))
    u = rs.rand(T, B)
    reward = torch.from_numpy(rs.randn(B, 1).astype(np.float32)).repeat(1, T)
    return ob, p, feats, masks, u, reward, T


def _rl_grads(ob, p, feats, masks, u, reward, T, lo, hi, denom=None):
    q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    em, am, om = [m[:, lo:hi] for m in masks]
    seq, lp, _ = ob.sample_rl(feats[lo:hi], q, u[:, lo:hi], em, am, om, T, early_exit=False)
    mask = torch.cat([torch.ones(hi - lo, 1), (seq > 0).float()[:, :-1]], 1)
    d = mask.sum() if denom is None else denom
    loss = -(lp * reward[lo:hi] * mask).sum() / d
    g = torch.autograd.grad(loss, list(q.values()))
    return dict(zip(q.keys(), g)), float(mask.sum()), float(loss.detach())


def _worker(rank, world, port, out, B=6):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from simpleimagecaptionzoo_amd import dist as D
    D.init_from_env("gloo")
    assert D.is_distributed() and D.world_size() == world and D.rank() == rank
    ob, p, feats, masks, u, reward, T = _problem(B)
    lo, hi = D.shard_range(feats.shape[0])
    assert hi > lo and (lo, hi) == D.shard_range(B, rank, world)
    sizes = D.all_gather_rows(torch.tensor([[hi - lo]])).view(-1).tolist()       # every rank's share: they tile [0, B) in rank order
    assert sum(sizes) == B and max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    # G2: normaliser first
    _, msum_local, _ = _rl_grads(ob, p, feats, masks, u, reward, T, lo, hi)
    msum = D.all_reduce_scalar(msum_local)
    grads, _, loss_part = _rl_grads(ob, p, feats, masks, u, reward, T, lo, hi, denom=msum)
    flat = torch.cat([g.reshape(-1) for g in grads.values()])
    D.all_reduce_sum_(flat)                       # G1
    loss = D.all_reduce_scalar(loss_part)
    ids = D.all_gather_rows(torch.arange(lo, hi).view(-1, 1))   # G3 with unequal shard sizes
    # G3 as the Engine's evaluation uses it: rank r decoded the batches i with i % world == r (3 images per batch, ragged
    # token rows); every rank gets all captions back in loader order
    mine = [i for i in range(10) if (i // 3) % world == rank]
    cap_ids, cap_rows = D.gather_caption_rows(mine, [1000 + i for i in mine], [np.arange(4, 4 + 1 + i % 5) for i in mine], "cpu")
    assert cap_ids == [1000 + i for i in range(10)]
    assert all(r.tolist() == list(range(4, 4 + 1 + i % 5)) for i, r in enumerate(cap_rows))
    assert D.seed_for_rank(7) != 7 or rank == 0
    seeds = D.all_gather_rows(torch.tensor([[D.seed_for_rank(7) >> 40, D.seed_for_rank(7) & 0xFFFFFFFFFF]])).tolist()
    assert len({tuple(x) for x in seeds}) == world and [x[0] for x in seeds] == list(range(world))      # one Philox stream per rank
    if rank == 0:
        torch.save({"flat": flat, "loss": loss, "msum": msum, "ids": ids}, out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 6), (8, 11)])
def test_sharded_scst_gradient_equals_single_batch(tmp_path, world, B):
    import socket
    sys.path.insert(0, ROOT)
    out = str(tmp_path / "r0.pt")
    with socket.socket() as sk:          # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.start_processes(_worker, args=(world, port, out, B), nprocs=world, join=True, start_method="spawn")
    got = torch.load(out)
    ob, p, feats, masks, u, reward, T = _problem(B)
    grads, msum, loss = _rl_grads(ob, p, feats, masks, u, reward, T, 0, feats.shape[0])
    want = torch.cat([g.reshape(-1) for g in grads.values()])
    assert got["msum"] == msum
    assert abs(got["loss"] - loss) < 1e-6
    np.testing.assert_allclose(got["flat"].numpy(), want.numpy(), atol=1e-6, rtol=1e-5)
    assert got["ids"].view(-1).tolist() == list(range(feats.shape[0]))
    # the clamp is applied after the reduction (Engine.py:271): clamping per-rank partial gradients first would differ
    assert float((want.clamp(-1e-3, 1e-3) - want).abs().max()) > 0
