"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm) on the GPU,
gloo on the CPU (tests).  The reference is single-process (Main.py:24-25); the exchange steps below are the only
ones the path needs:
  * G2  all-reduce(sum) of the loss normaliser (XE: token count, SCST: mask sum) BEFORE backward, so that every rank
        scales its gradient by 1 / global normaliser and the summed gradient equals the single-batch one;
  * G1  all-reduce(sum) of the flat gradient buffer (one collective for all 21 tensors), THEN clamp + Adam on every
        rank (the clamp must follow the reduction to match Engine.py:186-188);
  * G3  all-gather of (B_r, T) caption ids for evaluation / corpus-level scoring.
"""
import os

import torch
import torch.distributed as td


def is_distributed():
    return td.is_available() and td.is_initialized() and td.get_world_size() > 1


def init_from_env(backend=None):
    """Initialise from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run); no-op for one process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, 0
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not td.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local)
            td.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            td.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def rank():
    return td.get_rank() if is_distributed() else 0


def world_size():
    return td.get_world_size() if is_distributed() else 1


def all_reduce_sum_(t):
    if is_distributed():
        td.all_reduce(t, op=td.ReduceOp.SUM)
    return t


def all_reduce_scalar(x):
    """Sum a python number or a 1-element tensor over ranks; returns a python float (host sync: it feeds a kernel
    argument -- the loss normaliser)."""
    if not is_distributed():
        return float(x)
    if torch.is_tensor(x):
        t = x.detach().clone().float().view(1)
    else:
        dev = torch.device("cuda", torch.cuda.current_device()) if td.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([float(x)], dtype=torch.float32, device=dev)
    td.all_reduce(t, op=td.ReduceOp.SUM)
    return float(t.item())


def seed_for_rank(seed):
    """Philox seed of this rank: replicas must not draw the same dropout masks / sampling uniforms for their local row b
    (the streams are keyed by (seed, stream, step, element) and every shard numbers its rows from 0)."""
    return (int(seed) + (rank() << 40)) & 0xFFFFFFFFFFFFFFFF


def all_gather_rows(t):
    """Concatenate per-rank (B_r, ...) tensors along dim 0 in rank order (B_r may differ between ranks): one all-gather of
    the row counts (read back once), one of the rows padded to the largest count."""
    if not is_distributed():
        return t
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    alln = torch.zeros(td.get_world_size(), dtype=torch.int64, device=t.device)
    td.all_gather_into_tensor(alln, n)
    sizes = alln.tolist()
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    outs = [torch.zeros_like(pad) for _ in sizes]
    td.all_gather(outs, pad)
    return torch.cat([o[:s] for o, s in zip(outs, sizes)], 0)


def gather_caption_rows(keys, image_ids, rows, device, width=52):
    """G3: all-gather the decoded captions of every rank and return (image ids, token rows) of ALL ranks ordered by `keys`
    (the image's position in the loader).  rows: per image a 1-D array of token ids (greedy: 20; beam search: <sta> + at most
    50 tokens + <end>), padded here with -1 to `width`."""
    import numpy as np
    pack = np.full((len(rows), 2 + width), -1, dtype=np.int64)
    for j, r in enumerate(rows):
        pack[j, 0], pack[j, 1] = keys[j], image_ids[j]
        pack[j, 2:2 + len(r)] = np.asarray(r, dtype=np.int64)
    dev = torch.device(device) if td.get_backend() == "nccl" else torch.device("cpu")
    allp = all_gather_rows(torch.from_numpy(pack).to(dev)).cpu().numpy()
    allp = allp[np.argsort(allp[:, 0], kind="stable")]
    return [int(x) for x in allp[:, 1]], [row[2:][row[2:] >= 0] for row in allp]


def shard_range(n, r=None, w=None):
    """Contiguous shard [lo, hi) of n items for rank r of w (images are independent: SURVEY.md 8e)."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    per, rem = divmod(n, w)
    lo = r * per + min(r, rem)
    return lo, lo + per + (1 if r < rem else 0)
