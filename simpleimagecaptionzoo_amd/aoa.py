"""Host-side owner of one libicz AoA handle (Models/AoA_Model.py) and the AoADetection Captioner on top of it.

Region sets: `'fixed'` (36 boxes, `bu_masks` None), the 7x7 grid (49) and `'adaptive'` (10..100 boxes per image, padded to
the batch's largest count with prefix `bu_masks`, AoA_Engine.py:33-46) -- the handle is created for the largest region
count it will see (`num_regions`) and every batch may be narrower."""
import collections
import copy
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .scheduled import ScheduledSamplingState, handle_set_scheduled_sampling
from ._lib import AOA_DECODER_KEYS, AOA_PARAM_KEYS, AoaDims, AoaParams, AoaRng, check, lib, ptr, stream_ptr

_MASKS = ("proj", "ref_att", "ref_aoa", "ref_sc", "emb", "ctx", "att", "out")


def make_aoa_rng(seed=0, uniforms=None, masks=None):
    """icz_aoa_rng: Philox streams from `seed` for everything that is not given explicitly.  `masks`: dict with any of
    proj / ref_att / ref_aoa / ref_sc / emb / ctx / att / out -> uint8 keep-mask CUDA tensors (parity tests)."""
    r = AoaRng()
    r.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    keep = []
    if uniforms is not None:
        if uniforms.dtype != torch.float32 or not uniforms.is_cuda:
            raise _lib.IczError("uniforms must be an fp32 CUDA tensor")
        uniforms = uniforms.contiguous()
        r.uniforms = uniforms.data_ptr()
        keep.append(uniforms)
    for name in _MASKS:
        m = (masks or {}).get(name)
        if m is None:
            continue
        if m.dtype != torch.uint8 or not m.is_cuda:
            raise _lib.IczError("mask %s must be a uint8 CUDA tensor" % name)
        m = m.contiguous()
        setattr(r, name + "_mask", m.data_ptr())
        keep.append(m)
    r._keep = keep
    return r


# features [B, R, D] + the valid region count of each image (host ints; None = all R valid)
RegionBatch = collections.namedtuple("RegionBatch", "feats counts")


def counts_from_masks(bu_masks):
    """bu_masks [B, R] (1 = valid) -> per-image counts.  The reference builds prefix masks (AoA_Engine.py:37-40) and itself
    reads them as lengths (pack_wrapper, AoA_Model.py:652); anything else is rejected."""
    m = bu_masks.detach()
    counts = m.long().sum(1)
    prefix = torch.arange(m.shape[1], device=m.device).unsqueeze(0) < counts.unsqueeze(1)
    if not bool(((m != 0) == prefix).all()):
        raise ValueError("bu_masks must mark a leading run of valid regions per image (AoA_Engine.py:37-40)")
    return counts.tolist()


class AoaHandle:
    def __init__(self, R, D, Hd, E, V, NH, max_rows, max_len=20, device="cuda:0"):
        self.R, self.D, self.Hd, self.E, self.V, self.NH = R, D, Hd, E, V, NH
        self.device = torch.device(device)
        self._h = C.c_void_p()
        self._params = None
        self._persistent = False
        self._regions = (R, None)
        self._counts_dev = None
        with torch.cuda.device(self.device):
            check(lib().icz_aoa_create(C.byref(AoaDims(R, D, Hd, E, V, NH, max_rows, max_len)), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().icz_aoa_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def enable_graphs(self, on):
        """Capture the SCST rollout pair and the REINFORCE backward pass into hipGraphs and replay them (include/icz.h:
        icz_aoa_set_option).  Implies persistent output buffers: the tensors those calls return are overwritten by the next call."""
        self._persistent = bool(on)
        check(lib().icz_aoa_set_option(self._h, b"graphs", 1 if on else 0))

    def set_option(self, name, value):
        """icz_aoa_set_option: "early_out", "refine_pair" (include/icz.h)."""
        check(lib().icz_aoa_set_option(self._h, name.encode(), int(value)))

    def _buf(self, name, shape, dtype):
        if not self._persistent:
            return torch.zeros(shape, dtype=dtype, device=self.device)
        bufs = self.__dict__.setdefault("_bufs", {})
        key = (name,) + tuple(shape)
        t = bufs.get(key)
        if t is None:
            t = bufs[key] = torch.zeros(shape, dtype=dtype, device=self.device)
        return t

    def set_grad_callback(self, fn):
        """fn(stage) is called while a backward call is being enqueued, each time a group of decoder gradients is complete in
        stream order (include/icz.h: icz_aoa_set_grad_callback); None removes it."""
        self._grad_cb = _lib.GRAD_READY_CB(lambda user, stage: fn(int(stage))) if fn is not None else _lib.GRAD_READY_CB()
        check(lib().icz_aoa_set_grad_callback(self._h, self._grad_cb, None))

    def bind(self, tensors):
        st = AoaParams()
        keep = {}
        for i, key in enumerate(AOA_PARAM_KEYS):
            t = tensors[key]
            if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
                raise _lib.IczError("parameter %s must be a contiguous fp32 CUDA tensor" % key)
            setattr(st, "p%d" % i, t.data_ptr())
            keep[key] = t
        self._params = keep
        check(lib().icz_aoa_bind_params(self._h, C.byref(st)))
        self.refresh()

    def refresh(self):
        check(lib().icz_aoa_refresh_weights(self._h, stream_ptr()))

    def new_grads(self):
        """Gradient buffers for the decoder parameters -- the only ones in the reference's optimizer (AoA_Model.py:669-674)."""
        return {k: torch.zeros_like(self._params[k]) for k in AOA_DECODER_KEYS}

    def _grad_struct(self, grads):
        st = AoaParams()
        for i, key in enumerate(AOA_PARAM_KEYS):
            if key in grads:
                setattr(st, "p%d" % i, grads[key].data_ptr())
        return st

    def set_regions(self, regions, counts=None):
        """icz_aoa_set_regions: the batches that follow are [B, regions, D]; counts = valid regions per image or None."""
        if counts is None:
            if self._regions != (regions, None):
                check(lib().icz_aoa_set_regions(self._h, int(regions), None, None, 0))
                self._regions, self._counts_dev = (regions, None), None
            return
        counts = [int(c) for c in counts]
        host = (C.c_int32 * len(counts))(*counts)
        dev = torch.tensor(counts, dtype=torch.int32, device=self.device)
        check(lib().icz_aoa_set_regions(self._h, int(regions), ptr(dev), host, len(counts)))
        # the kernels of the following calls read `dev`: it stays referenced until the next set_regions, and torch's
        # caching allocator hands its memory out again only in stream order
        self._regions, self._counts_dev = (regions, tuple(counts)), dev

    def _feats(self, f):
        counts = None
        if isinstance(f, RegionBatch):
            f, counts = f
        if f.dtype != torch.float32 or not f.is_cuda or f.dim() != 3 or f.shape[2] != self.D or not 1 <= f.shape[1] <= self.R:
            raise _lib.IczError("bu_feats must be an fp32 CUDA tensor (B, 1..%d, %d)" % (self.R, self.D))
        if counts is not None and len(counts) != f.shape[0]:
            raise _lib.IczError("%d region counts for %d images" % (len(counts), f.shape[0]))
        self.set_regions(f.shape[1], counts)
        return f.contiguous()

    def refine(self, feats):
        feats = self._feats(feats)
        out = torch.empty(feats.shape[0], feats.shape[1], self.Hd, device=feats.device)
        check(lib().icz_aoa_refine(self._h, ptr(feats), feats.shape[0], ptr(out), stream_ptr()))
        return out

    def greedy(self, feats, max_len=20):
        feats = self._feats(feats)
        ids = torch.empty(feats.shape[0], max_len, dtype=torch.int64, device=feats.device)
        check(lib().icz_aoa_greedy(self._h, ptr(feats), feats.shape[0], max_len, ptr(ids), stream_ptr()))
        return ids

    def beam_search(self, feats, beam_size=5, max_steps=50):
        feats = self._feats(feats)
        n = feats.shape[0]
        seqs = torch.zeros(n, max_steps + 1, dtype=torch.float32, device=feats.device)
        lens = torch.zeros(n, dtype=torch.int32, device=feats.device)
        check(lib().icz_aoa_beam_search(self._h, ptr(feats), n, beam_size, max_steps, ptr(seqs), ptr(lens), stream_ptr()))
        return seqs, lens

    def sample(self, feats, max_len=20, rng=None):
        feats = self._feats(feats)
        B = feats.shape[0]
        rng = rng or make_aoa_rng(0)
        seq = torch.zeros(B, max_len, dtype=torch.int64, device=feats.device)
        lp = torch.zeros(B, max_len, dtype=torch.float32, device=feats.device)
        check(lib().icz_aoa_sample(self._h, ptr(feats), B, max_len, C.byref(rng), ptr(seq), ptr(lp), stream_ptr()))
        self._live = (feats, rng, seq, lp)
        return seq, lp

    def rollouts(self, feats, max_len=20, rng=None):
        """Greedy baseline (eval mode) and sampled rollout (train mode) of one SCST step, Engine.py:256-261, as two concurrent
        chains on the device; identical to greedy() followed by sample()."""
        feats = self._feats(feats)
        B = feats.shape[0]
        rng = rng or make_aoa_rng(0)
        ids = self._buf("greedy_ids", (B, max_len), torch.int64)
        seq = self._buf("sample_seq", (B, max_len), torch.int64)
        lp = self._buf("sample_lp", (B, max_len), torch.float32)
        check(lib().icz_aoa_scst_rollouts(self._h, ptr(feats), B, max_len, C.byref(rng), ptr(ids), ptr(seq), ptr(lp), stream_ptr()))
        self._live = (feats, rng, seq, lp)
        return ids, seq, lp

    def sample_mask_sum(self):
        """Local sum of the REINFORCE mask (Utils.py:307-309) as a 1-element DEVICE tensor (no host round trip)."""
        seq = self._live[2]
        return ((seq[:, :-1] > 0).sum() + seq.shape[0]).float().view(1)

    def set_mask_sum_global(self, t):
        """DP: the all-reduced loss normaliser as a 1-element device tensor; then pass -1 as the global normaliser."""
        check(lib().icz_aoa_set_norm_global(self._h, ptr(t), stream_ptr()))

    def sample_backward(self, reward, grads, mask_sum_global=0.0):
        reward = reward.to(device=self.device, dtype=torch.float32).contiguous()
        loss = self._buf("rl_loss", (1,), torch.float32)
        msum = self._buf("rl_msum", (1,), torch.float32)
        gs = self._grad_struct(grads)
        check(lib().icz_aoa_sample_backward(self._h, ptr(reward), C.byref(gs), ptr(loss), ptr(msum), float(mask_sum_global), stream_ptr()))
        return loss, msum

    def saved_alphas(self, B, T, regions):
        """Head-averaged decoder attention [B, T, regions] of the forward pass the handle holds (last xe_forward / sample)."""
        out = torch.empty(B, T, regions, device=self.device)
        check(lib().icz_aoa_saved_alphas(self._h, ptr(out), stream_ptr()))
        return out

    def set_scheduled_sampling(self, ss_prob, gate=None, draw=None):
        """Scheduled sampling for the following xe_forward calls (AoA_Model.py:258-270 with the decoder's `ss_prob`)."""
        handle_set_scheduled_sampling(self, "icz_aoa_set_scheduled_sampling", ss_prob, gate, draw)

    def xe_forward(self, feats, captions, lengths, rng=None, train=True, want_logits=False):
        feats = self._feats(feats)
        B, L = captions.shape
        captions = captions.to(device=feats.device, dtype=torch.int64).contiguous()
        lens = (C.c_int32 * B)(*[int(x) for x in lengths])
        out = torch.empty(sum(int(x) for x in lengths), self.V, device=feats.device) if want_logits else None
        if train and rng is None:
            rng = make_aoa_rng(0)
        check(lib().icz_aoa_xe_forward(self._h, ptr(feats), ptr(captions), B, L, lens, C.byref(rng) if rng is not None else None,
                                       1 if train else 0, ptr(out), stream_ptr()))
        self._live = (feats, rng, captions)
        return out

    def xe_backward(self, grads, smoothing=0.1, n_tokens_global=0.0):
        loss = torch.zeros(1, device=self.device)
        gs = self._grad_struct(grads)
        check(lib().icz_aoa_xe_backward(self._h, float(smoothing), C.byref(gs), ptr(loss), float(n_tokens_global), stream_ptr()))
        return loss


# ---- parameter containers with the reference's module tree (state_dict keys and shapes equal the reference's) ----------
class _Norm(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.gain = nn.Parameter(torch.ones(n))
        self.bias = nn.Parameter(torch.zeros(n))


class _Block(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.linear_Q, self.linear_K, self.linear_V = nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d)
        self.aoa_module = nn.Sequential(nn.Linear(2 * d, 2 * d), nn.GLU())


class _RefineLayer(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.aoa_block = _Block(d)
        self.sublayer = nn.Module()
        self.sublayer.norm = _Norm(d)


class AoADetection_Captioner(nn.Module, ScheduledSamplingState):
    """AoADetection_Captioner (Models/AoA_Model.py:657-753) on libicz: same constructor arguments, state_dict and methods;
    every forward / sampling / search path runs in the HIP library (no torch compute, no CPU fallback)."""

    def __init__(self, vocab_size, num_heads=8, hidden_dim=1024, embed_dim=1024, dropout_aoa=0.3, dropout_prob=0.5, device="cuda:0",
                 num_regions=36, enc_dim=2048, max_batch=128, max_beam=5, max_len=20):
        super().__init__()
        if dropout_aoa != 0.3 or dropout_prob != 0.5:
            raise ValueError("the HIP path implements the reference's default drop probabilities (0.3 / 0.5 / 0.1)")
        Hd, E, V = hidden_dim, embed_dim, vocab_size
        self.img_feats_porjection = nn.Sequential(nn.Linear(enc_dim, Hd), nn.ReLU(), nn.Dropout(p=dropout_prob))
        self.aoa_refine = nn.Module()
        layer = _RefineLayer(Hd)
        self.aoa_refine.aoa_layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(6)])     # clones(): identical initial weights
        self.aoa_refine.norm = _Norm(Hd)
        dec = nn.Module()
        dec.lstm = nn.LSTMCell(E + Hd, Hd)
        dec.aoa_block = _Block(Hd)
        dec.embed = nn.Sequential(nn.Embedding(V, E), nn.ReLU(), nn.Dropout(p=dropout_prob))
        dec.h_norm = _Norm(Hd)
        dec.embed[0].weight.data.uniform_(-0.1, 0.1)
        v = torch.empty(V, Hd).uniform_(-0.1, 0.1)
        dec.predict = nn.Module()
        dec.predict.register_parameter("bias", nn.Parameter(torch.zeros(V)))
        dec.predict.register_parameter("weight_g", nn.Parameter(v.norm(dim=1, keepdim=True)))
        dec.predict.register_parameter("weight_v", nn.Parameter(v))
        self.decoder = dec
        self.dims = (num_regions, enc_dim, Hd, E, V, num_heads)
        self.max_rows, self.max_len = max_batch * max(1, max_beam), max_len
        self._h, self._bound = None, None
        self._seed = 0x5EED
        self._ss_init()                 # ss_prob (Engine.py:143) and its plumbing: scheduled.py

    def _named(self):
        sd = dict(self.named_parameters())
        return {k: sd[k] for k in AOA_PARAM_KEYS}

    def _trainable(self):
        sd = dict(self.named_parameters())
        return {k: sd[k] for k in AOA_DECODER_KEYS}

    def _next_rng(self):
        from .dist import seed_for_rank
        self._seed += 1
        return make_aoa_rng(seed_for_rank(self._seed))   # data-parallel replicas draw independent streams

    def _handle(self):
        named = self._named()
        ptrs = tuple(p.data_ptr() for p in named.values())
        dev = next(iter(named.values())).device
        if dev.type != "cuda":
            raise RuntimeError("AoADetection_Captioner (libicz) needs its parameters on a ROCm device; got %s" % dev)
        fresh = False
        if self._h is None or self._h.device != dev:
            R, D, Hd, E, V, NH = self.dims
            self._h = AoaHandle(R, D, Hd, E, V, NH, self.max_rows, max(self.max_len, 20), dev)
            self._bound = None
            fresh = True
        if ptrs != self._bound:
            self._h.bind({k: p.data for k, p in named.items()})
            self._bound = ptrs
        else:
            self._h.refresh()
        self._ss_push(self._h, fresh)
        return self._h

    def _replay_handle(self):
        """One-image handle for the teacher-forced replay behind eval_test_image's attention maps (as BUTDDetection_Captioner's):
        the training handle keeps its stored pass, graphs and buffers, and scheduled sampling is off on a fresh handle."""
        named = self._named()
        dev = next(iter(named.values())).device
        rh = getattr(self, "_rh", None)
        if rh is None or rh.device != dev:
            R, D, Hd, E, V, NH = self.dims
            rh = self._rh = AoaHandle(R, D, Hd, E, V, NH, 1, 52, dev)
        rh.bind({k: p.data for k, p in named.items()})
        return rh

    @staticmethod
    def _feats(visual_inputs):
        """bu_feats (+ the region counts behind bu_masks: `bu_counts` when the Engine supplies them, else read back from the
        mask)."""
        feats = visual_inputs["bu_feats"].detach()
        masks = visual_inputs.get("bu_masks")
        if masks is None:
            return feats
        counts = visual_inputs.get("bu_counts")
        return RegionBatch(feats, list(counts) if counts is not None else counts_from_masks(masks))

    def get_param_groups(self, lr_dict):
        """AoA_Model.py:669-674: only the decoder is optimised."""
        return [{"params": list(self.decoder.parameters()), "lr": lr_dict["lr"]}]

    def forward(self, visual_inputs, captions, lengths, rng=None):
        """AoA_Model.py:676-696: [0] of the result = packed logits (fused path: gradients come from the handle's xe_backward)."""
        train = self.training
        logits = self._handle().xe_forward(self._feats(visual_inputs), captions, list(lengths),
                                           (rng or self._next_rng()) if train else None, train=train, want_logits=True)
        return (logits, None)

    def sampler(self, visual_inputs, max_len=20):
        """AoA_Model.py:698-714."""
        return self._handle().greedy(self._feats(visual_inputs), max_len)

    def sampler_rl(self, visual_inputs, max_len=20, rng=None):
        """AoA_Model.py:716-734."""
        return self._handle().sample(self._feats(visual_inputs), max_len, rng or self._next_rng())

    def beam_search_sampler(self, visual_inputs, beam_size=5):
        """AoA_Model.py:736-753."""
        seqs, lens = self._handle().beam_search(self._feats(visual_inputs), beam_size, 50)
        lens = lens.tolist()
        out = [seqs[i:i + 1, :lens[i]] for i in range(len(lens))]
        return out[0] if len(out) == 1 else out

    def eval_test_image(self, visual_inputs, caption_vocab, max_len=20, eval_beam_size=-1):
        """AoA_Model.py:755-786 -> (caption words, [alphas (1, steps, regions)]): the decoder block's attention weights
        averaged over the heads (:118), taken from an evaluation-mode teacher-forced pass over the decoded sentence (the
        decoder state is a function of the token prefix, so these are the maps the reference records while decoding)."""
        feats = self._feats(visual_inputs)
        raw = feats[0] if isinstance(feats, RegionBatch) else feats
        assert raw.size(0) == 1
        h = self._handle()
        if eval_beam_size != -1:
            seqs, lens = h.beam_search(feats, eval_beam_size, 50)
            ids = seqs[:, :int(lens[0])].long()
            replay = ids                                     # <sta> w1 .. wn [<end>]
        else:
            ids = h.greedy(feats, max_len)
            replay = torch.cat([torch.ones(1, 1, dtype=torch.int64, device=ids.device), ids], 1)
        steps = replay.shape[1] - 1
        if steps > 0:
            rh = self._replay_handle()
            rh.xe_forward(feats, replay, [steps], None, train=False)
            alphas = rh.saved_alphas(1, steps, raw.shape[1])
        else:
            alphas = torch.zeros(1, 0, raw.shape[1], device=raw.device)
        caption = []
        for word_id in ids[0].cpu().numpy():
            word = caption_vocab.ix2word[int(word_id)]
            if word == "<end>":
                break
            elif word != "<sta>":
                caption.append(word)
        return caption, [alphas]
