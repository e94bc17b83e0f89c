"""Drop-in Captioner for the reference's Engine: same constructor arguments, same state_dict keys, same five
methods that Engine calls (reference class Models/BUTD_Model.py:443-544), all compute in libicz.

Two ways to train with it:
  * the reference's *unmodified* Engine.training_epoch / SCST_training_epoch work: forward() and sampler_rl()
    return tensors wired into autograd through _XEFunction / _SampleFunction, whose backward runs the HIP BPTT;
  * the Engine subclasses in engine.py call the fused entry points (loss + gradient + clamp + Adam on the device,
    no autograd graph) -- the fast path that bench.py measures.
"""
import math

import torch
import torch.nn as nn

from ._lib import BUTD_PARAM_KEYS
from .butd import ButdHandle, make_rng
from .scheduled import ScheduledSamplingState


class _Holder(nn.Module):
    """A module that only owns parameters (gives them the reference's dotted names)."""

    def __init__(self, **tensors):
        super().__init__()
        for k, v in tensors.items():
            self.register_parameter(k, nn.Parameter(v))


def _uniform(shape, bound):
    return (torch.rand(shape) * 2 - 1) * bound


class _DecoderParams(nn.Module):
    """Parameter tree of DecoderRNN (BUTD_Model.py:64-90) with the reference's initialisation."""

    def __init__(self, atten_dim, embed_dim, hidden_dim, vocab_size, enc_dim):
        super().__init__()
        A, E, H, V, D = atten_dim, embed_dim, hidden_dim, vocab_size, enc_dim

        def wn_linear(o, i, wbound=None, zero_bias=False):
            b = 1.0 / math.sqrt(i)
            v = _uniform((o, i), wbound if wbound is not None else b)
            bias = torch.zeros(o) if zero_bias else _uniform((o,), b)
            return _Holder(bias=bias, weight_g=v.norm(dim=1, keepdim=True), weight_v=v)

        self.atten = nn.Module()
        self.atten.enc_att = wn_linear(A, D)
        self.atten.dec_att = wn_linear(A, H)
        self.atten.affine = wn_linear(1, A)
        self.embed = nn.ModuleList([_Holder(weight=_uniform((V, E), 0.1))])      # key "embed.0.weight"
        k = 1.0 / math.sqrt(H)
        self.TD_atten = _Holder(weight_ih=_uniform((4 * H, H + D + E), k), weight_hh=_uniform((4 * H, H), k),
                                bias_ih=_uniform((4 * H,), k), bias_hh=_uniform((4 * H,), k))
        self.language_model = _Holder(weight_ih=_uniform((4 * H, D + H), k), weight_hh=_uniform((4 * H, H), k),
                                      bias_ih=_uniform((4 * H,), k), bias_hh=_uniform((4 * H,), k))
        self.predict = wn_linear(V, H, wbound=0.1, zero_bias=True)


class PackedLogits(tuple):
    """Stands in for the PackedSequence the reference returns: Engine only reads element [0] (Engine.py:182)."""
    @property
    def data(self):
        return self[0]


class _SampleFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cap, feats, max_len, rng, *params):
        seq, lp = cap._handle().sample(feats, max_len, rng)
        ctx.cap = cap
        ctx.mark_non_differentiable(seq)
        return seq, lp

    @staticmethod
    def backward(ctx, _gseq, glp):
        cap = ctx.cap
        grads = cap._grad_buffers()
        cap._handle().sample_backward_dlogp(glp, grads)
        return (None, None, None, None) + tuple(grads[k] for k in BUTD_PARAM_KEYS)


class _XEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cap, feats, captions, lengths, rng, train, *params):
        logits = cap._handle().xe_forward(feats, captions, lengths, rng, train=train, want_logits=True)
        ctx.cap = cap
        return logits

    @staticmethod
    def backward(ctx, glogits):
        cap = ctx.cap
        grads = cap._grad_buffers()
        cap._handle().xe_backward_dlogits(glogits, grads)
        return (None, None, None, None, None, None) + tuple(grads[k] for k in BUTD_PARAM_KEYS)


class BUTDDetection_Captioner(nn.Module, ScheduledSamplingState):
    """Models/BUTD_Model.py:443-544 on libicz.  enc_dim / num_regions default to the bottom-up 36 x 2048 layout
    (49 regions = BUTDSpatial's 7x7 grid features, the same decoder, BUTD_Model.py:321-440)."""

    def __init__(self, atten_dim, embed_dim, hidden_dim, vocab_size, dropout=0.5, device="cuda:0", enc_dim=2048,
                 num_regions=36, max_batch=128, max_beam=5, max_len=20):
        super().__init__()
        if dropout != 0.5:
            raise ValueError("the HIP path implements the reference's fixed nn.Dropout(p=0.5) (BUTD_Model.py:66)")
        self.decoder = _DecoderParams(atten_dim, embed_dim, hidden_dim, vocab_size, enc_dim)
        self.dims = dict(R=num_regions, D=enc_dim, H=hidden_dim, E=embed_dim, A=atten_dim, V=vocab_size)
        self.max_rows = max_batch * max(1, max_beam)
        self.max_len = max_len
        self._ss_init()                 # ss_prob (Engine.py:143) and its plumbing: scheduled.py
        self._h = None
        self._bound_ptrs = None
        self._grads = None
        self._seed = 0x5EED
        self._device = torch.device(device)

    # ---- plumbing --------------------------------------------------------------------------------
    def _named(self):
        sd = dict(self.decoder.named_parameters())
        return {k: sd[k] for k in BUTD_PARAM_KEYS}

    def _handle(self):
        """(Re)bind the handle when parameters moved (.to(device), load_state_dict keeps storage) and refresh the
        materialised weight-norm weights -- cheap (4 small kernels) and always correct after optimizer steps."""
        named = self._named()
        ptrs = tuple(p.data_ptr() for p in named.values())
        dev = next(iter(named.values())).device
        fresh = False
        if dev.type != "cuda":
            raise RuntimeError("BUTDDetection_Captioner (libicz) needs its parameters on a ROCm device; got %s" % dev)
        if self._h is None or self._h.device != dev:
            d = self.dims
            self._h = ButdHandle(d["R"], d["D"], d["H"], d["E"], d["A"], d["V"], self.max_rows, max(self.max_len, 20), dev)
            self._bound_ptrs = None
            fresh = True
        if ptrs != self._bound_ptrs:
            self._h.bind({k: p.data for k, p in named.items()})
            self._bound_ptrs = ptrs
        else:
            self._h.refresh()
        self._ss_push(self._h, fresh)
        return self._h

    def _grad_buffers(self):
        if self._grads is None or next(iter(self._grads.values())).device != next(self.parameters()).device:
            self._grads = {k: torch.zeros_like(p.data) for k, p in self._named().items()}
        return self._grads

    def _next_rng(self):
        from .dist import seed_for_rank
        self._seed += 1
        return make_rng(seed_for_rank(self._seed))       # data-parallel replicas draw independent streams

    def set_seed(self, seed):
        self._seed = int(seed)

    def get_param_groups(self, lr_dict):
        """BUTD_Model.py:451-456."""
        return [{"params": list(self.decoder.parameters()), "lr": lr_dict["lr"]}]

    # ---- the five methods Engine calls ---------------------------------------------------------------
    def forward(self, visual_inputs, captions, lengths, rng=None):
        """XE forward (BUTD_Model.py:458-476): returns an object whose [0] is the packed logits (sum(lengths), V)."""
        feats = visual_inputs["bu_feats"]
        train = self.training
        if train and rng is None:
            rng = self._next_rng()
        params = list(self._named().values())
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            logits = _XEFunction.apply(self, feats, captions, list(lengths), rng, train, *params)
        else:
            logits = self._handle().xe_forward(feats, captions, list(lengths), rng, train=train, want_logits=True)
        return PackedLogits((logits, None))

    def sampler(self, visual_inputs, max_len=20):
        """Greedy decode (BUTD_Model.py:478-489) -> LongTensor (B, max_len)."""
        return self._handle().greedy(visual_inputs["bu_feats"], max_len)

    def sampler_rl(self, visual_inputs, max_len=20, rng=None):
        """Multinomial rollout (BUTD_Model.py:491-503) -> (seq LongTensor (B,T), seqLogprobs (B,T) with grad)."""
        feats = visual_inputs["bu_feats"]
        rng = rng or self._next_rng()
        params = list(self._named().values())
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _SampleFunction.apply(self, feats, max_len, rng, *params)
        return self._handle().sample(feats, max_len, rng)

    def beam_search_sampler(self, visual_inputs, beam_size=5):
        """Beam search (BUTD_Model.py:505-517).  A batch of one image returns the reference's (1, L) float tensor;
        larger batches (an extension) return a list of (1, L_i) tensors."""
        seqs, lens = self._handle().beam_search(visual_inputs["bu_feats"], beam_size, 50)
        lens = lens.tolist()
        out = [seqs[i:i + 1, :lens[i]] for i in range(len(lens))]
        return out[0] if len(out) == 1 else out

    def _replay_handle(self):
        """One-row handle for the teacher-forced replay behind eval_test_image's beam-search attention maps: the training handle
        keeps its stored forward pass, its captured graphs and its buffers (a beam sentence of up to 50 steps would re-allocate
        them), and the replay never sees scheduled sampling (a fresh handle has it switched off)."""
        named = self._named()
        dev = next(iter(named.values())).device
        rh = getattr(self, "_rh", None)
        if rh is None or rh.device != dev:
            d = self.dims
            rh = self._rh = ButdHandle(d["R"], d["D"], d["H"], d["E"], d["A"], d["V"], 1, 52, dev)
        rh.bind({k: v.data for k, v in named.items()})          # binds and refreshes the weight-norm weights (parameters may have moved on)
        return rh

    def eval_test_image(self, visual_inputs, caption_vocab, max_len=20, eval_beam_size=-1):
        """BUTD_Model.py:519-544 -> (caption words, [alphas (1, steps, R)]).  Greedy: the attention maps come out of the decode
        itself.  Beam search: the decoder state of a beam is a function of its token prefix, so the maps of the returned
        sentence are those of an evaluation-mode teacher-forced pass over it.  (The reference means to carry them along beam
        by beam, :282,:309-317, but appends every step's maps without the beam permutation -- `alpha.unsqueeze(1)` is not
        indexed by prev_word_inds and the history is never compacted with incomplete_inds -- so for beam > 1 its result mixes
        the maps of different beams; for beam 1 both agree, tests/test_gpu_round2.py.)"""
        feats = visual_inputs["bu_feats"]
        assert feats.size(0) == 1
        h = self._handle()
        if eval_beam_size != -1:
            seqs, lens = h.beam_search(feats, eval_beam_size, 50)
            n = int(lens[0])
            ids = seqs[:, :n]
            steps = n - 1                                   # one map per generated token (the leading <sta> has none, :316)
            if steps > 0:
                rh = self._replay_handle()
                rh.xe_forward(feats, ids.long(), [steps], None, train=False)
                alphas = rh.saved_alphas(1, steps)
            else:
                alphas = torch.zeros(1, 0, self.dims["R"], device=feats.device)
        else:
            ids, alphas = h.greedy(feats, max_len, want_alphas=True)
            ids, alphas = ids.clone(), alphas.clone()
        caption = []
        for word_id in ids[0].cpu().numpy():
            word = caption_vocab.ix2word[int(word_id)]
            if word == "<end>":
                break
            elif word != "<sta>":
                caption.append(word)
        return caption, [alphas]
