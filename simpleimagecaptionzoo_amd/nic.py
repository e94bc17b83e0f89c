"""Host-side owner of one libicz NIC decoder handle (Models/NIC_Model.py:39-212) and the Captioner on top of it."""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from ._lib import NIC_PARAM_FIELDS, NIC_PARAM_KEYS, NicDims, NicParams, check, lib, ptr, stream_ptr
from .butd import make_rng
from .scheduled import ScheduledSamplingState, handle_set_scheduled_sampling


class NicHandle:
    def __init__(self, E, H, V, max_rows, max_len=20, device="cuda:0"):
        self.E, self.H, self.V = E, H, V
        self.device = torch.device(device)
        self._h = C.c_void_p()
        self._params = None
        self._persistent = False
        with torch.cuda.device(self.device):
            check(lib().icz_nic_create(C.byref(NicDims(E, H, V, max_rows, max_len)), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().icz_nic_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def enable_graphs(self, on):      # the NIC paths are launched eagerly
        self._persistent = bool(on)

    def rollouts(self, feats, max_len=20, rng=None):
        """Greedy baseline (eval mode) then the sampled rollout (train mode): Engine.py:256-261."""
        greedy = self.greedy(feats, max_len)
        seq, lp = self.sample(feats, max_len, rng)
        return greedy, seq, lp

    def sample_mask_sum(self):
        """Local sum of the REINFORCE mask (Utils.py:307-309) as a 1-element DEVICE tensor (no host round trip)."""
        seq = self._live[2]
        return ((seq[:, :-1] > 0).sum() + seq.shape[0]).float().view(1)

    def set_mask_sum_global(self, t):
        """DP: the all-reduced loss normaliser as a 1-element device tensor; then pass -1 as the global normaliser."""
        check(lib().icz_nic_set_norm_global(self._h, ptr(t), stream_ptr()))

    def bind(self, tensors):
        st = NicParams()
        keep = []
        for field, key in zip(NIC_PARAM_FIELDS, NIC_PARAM_KEYS):
            t = tensors[key]
            if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
                raise _lib.IczError("parameter %s must be a contiguous fp32 CUDA tensor" % key)
            setattr(st, field, t.data_ptr())
            keep.append(t)
        self._params = keep
        check(lib().icz_nic_bind_params(self._h, C.byref(st)))
        self.refresh()

    def refresh(self):
        check(lib().icz_nic_refresh_weights(self._h, stream_ptr()))

    def new_grads(self):
        return {k: torch.zeros_like(t) for k, t in zip(NIC_PARAM_KEYS, self._params)}

    def _grad_struct(self, grads):
        st = NicParams()
        for field, key in zip(NIC_PARAM_FIELDS, NIC_PARAM_KEYS):
            setattr(st, field, grads[key].data_ptr())
        return st

    def _feats(self, f):
        if f.dtype != torch.float32 or not f.is_cuda or f.dim() != 2 or f.shape[1] != self.E:
            raise _lib.IczError("features must be an fp32 CUDA tensor (B,%d)" % self.E)
        return f.contiguous()

    def greedy(self, feats, max_len=20):
        feats = self._feats(feats)
        ids = torch.empty(feats.shape[0], max_len, dtype=torch.int64, device=feats.device)
        check(lib().icz_nic_greedy(self._h, ptr(feats), feats.shape[0], max_len, ptr(ids), stream_ptr()))
        return ids

    def sample(self, feats, max_len=20, rng=None):
        feats = self._feats(feats)
        B = feats.shape[0]
        rng = rng or make_rng(0)
        seq = torch.zeros(B, max_len, dtype=torch.int64, device=feats.device)
        lp = torch.zeros(B, max_len, dtype=torch.float32, device=feats.device)
        check(lib().icz_nic_sample(self._h, ptr(feats), B, max_len, C.byref(rng), ptr(seq), ptr(lp), stream_ptr()))
        self._live = (feats, rng, seq, lp)
        return seq, lp

    def sample_backward(self, reward, grads, mask_sum_global=0.0, want_dfeats=False):
        feats = self._live[0]
        reward = reward.to(device=self.device, dtype=torch.float32).contiguous()
        loss = torch.zeros(1, device=self.device)
        msum = torch.zeros(1, device=self.device)
        dfe = torch.zeros_like(feats) if want_dfeats else None
        gs = self._grad_struct(grads)
        check(lib().icz_nic_sample_backward(self._h, ptr(reward), C.byref(gs), ptr(dfe), ptr(loss), ptr(msum),
                                            float(mask_sum_global), stream_ptr()))
        return (loss, msum, dfe) if want_dfeats else (loss, msum)

    def set_scheduled_sampling(self, ss_prob, gate=None, draw=None):
        """Scheduled sampling for the following xe_forward calls (NIC_Model.py:77-89 with the decoder's `ss_prob`)."""
        handle_set_scheduled_sampling(self, "icz_nic_set_scheduled_sampling", ss_prob, gate, draw)

    def xe_forward(self, feats, captions, lengths, rng=None, train=True, want_logits=False):
        feats = self._feats(feats)
        B, L = captions.shape
        captions = captions.to(device=feats.device, dtype=torch.int64).contiguous()
        lens = (C.c_int32 * B)(*[int(x) for x in lengths])
        out = torch.empty(sum(int(x) for x in lengths), self.V, device=feats.device) if want_logits else None
        if train and rng is None:
            rng = make_rng(0)
        check(lib().icz_nic_xe_forward(self._h, ptr(feats), ptr(captions), B, L, lens, C.byref(rng) if rng is not None else None,
                                       1 if train else 0, ptr(out), stream_ptr()))
        self._live = (feats, rng, captions)
        return out

    def xe_backward(self, grads, smoothing=0.1, n_tokens_global=0.0, want_dfeats=False):
        feats = self._live[0]
        loss = torch.zeros(1, device=self.device)
        dfe = torch.zeros_like(feats) if want_dfeats else None
        gs = self._grad_struct(grads)
        check(lib().icz_nic_xe_backward(self._h, float(smoothing), C.byref(gs), ptr(dfe), ptr(loss), float(n_tokens_global),
                                        stream_ptr()))
        return (loss, dfe) if want_dfeats else loss

    def beam_search(self, feats, beam_size=5, max_steps=50):
        feats = self._feats(feats)
        n = feats.shape[0]
        seqs = torch.zeros(n, max_steps + 1, dtype=torch.float32, device=feats.device)
        lens = torch.zeros(n, dtype=torch.int32, device=feats.device)
        check(lib().icz_nic_beam_search(self._h, ptr(feats), n, beam_size, max_steps, ptr(seqs), ptr(lens), stream_ptr()))
        return seqs, lens


class NICDecoder_Captioner(nn.Module, ScheduledSamplingState):
    """The decoder half of NIC_Captioner (Models/NIC_Model.py:214-332) on libicz.  The CNN encoder + img_embedding
    (NIC_Model.py:8-37) is outside the hot path: pass its output as visual_inputs['img_feats'] (B, embed_dim), or
    supply `encoder` (any nn.Module mapping visual_inputs['img_tensors'] to that embedding)."""

    def __init__(self, embed_dim, hidden_dim, vocab_size, dropout=0.5, device="cuda:0", encoder=None, max_batch=128,
                 max_beam=5, max_len=20):
        super().__init__()
        if dropout != 0.5:
            raise ValueError("the HIP path implements the reference's fixed nn.Dropout(p=0.5) (NIC_Model.py:50)")
        import math
        E, H, V = embed_dim, hidden_dim, vocab_size
        k = 1.0 / math.sqrt(H)
        U = lambda shape, b: (torch.rand(shape) * 2 - 1) * b
        self.decoder = nn.Module()
        self.decoder.embed = nn.Module()
        self.decoder.embed.register_parameter("weight", nn.Parameter(torch.randn(V, E)))       # nn.Embedding default N(0,1)
        self.decoder.lstm = nn.Module()
        for name, shape in (("weight_ih", (4 * H, E)), ("weight_hh", (4 * H, H)), ("bias_ih", (4 * H,)), ("bias_hh", (4 * H,))):
            self.decoder.lstm.register_parameter(name, nn.Parameter(U(shape, k)))
        v = U((V, H), k)
        self.decoder.predict = nn.Module()
        self.decoder.predict.register_parameter("bias", nn.Parameter(U((V,), k)))
        self.decoder.predict.register_parameter("weight_g", nn.Parameter(v.norm(dim=1, keepdim=True)))
        self.decoder.predict.register_parameter("weight_v", nn.Parameter(v))
        self.encoder = encoder
        self.dims = (E, H, V)
        self.max_rows, self.max_len = max_batch * max(1, max_beam), max_len
        self._h, self._bound = None, None
        self._seed = 0x5EED
        self._ss_init()                 # ss_prob (Engine.py:143) and its plumbing: scheduled.py

    def _named(self):
        sd = dict(self.decoder.named_parameters())
        return {k: sd[k] for k in NIC_PARAM_KEYS}

    def _handle(self):
        named = self._named()
        ptrs = tuple(p.data_ptr() for p in named.values())
        dev = next(iter(named.values())).device
        if dev.type != "cuda":
            raise RuntimeError("NICDecoder_Captioner (libicz) needs its parameters on a ROCm device; got %s" % dev)
        fresh = False
        if self._h is None or self._h.device != dev:
            E, H, V = self.dims
            self._h = NicHandle(E, H, V, self.max_rows, max(self.max_len, 20), dev)
            self._bound = None
            fresh = True
        if ptrs != self._bound:
            self._h.bind({k: p.data for k, p in named.items()})
            self._bound = ptrs
        else:
            self._h.refresh()
        self._ss_push(self._h, fresh)
        return self._h

    def _next_rng(self):
        from .dist import seed_for_rank
        self._seed += 1
        return make_rng(seed_for_rank(self._seed))       # data-parallel replicas draw independent streams

    def _features(self, visual_inputs):
        if "img_feats" in visual_inputs:
            return visual_inputs["img_feats"]
        if self.encoder is None:
            raise RuntimeError("no 'img_feats' in visual_inputs and no encoder module was supplied")
        return self.encoder(visual_inputs["img_tensors"])

    def get_param_groups(self, lr_dict):
        return [{"params": list(self.decoder.parameters()), "lr": lr_dict["lr"]}]

    def sampler(self, visual_inputs, max_len=20):
        """NIC_Model.py:262-273."""
        return self._handle().greedy(self._features(visual_inputs).detach(), max_len)

    def sampler_rl(self, visual_inputs, max_len=20, rng=None):
        """NIC_Model.py:275-287 (fused path: no autograd graph; use the handle's sample_backward)."""
        return self._handle().sample(self._features(visual_inputs).detach(), max_len, rng or self._next_rng())

    def beam_search_sampler(self, visual_inputs, beam_size=5):
        """NIC_Model.py:289-301."""
        seqs, lens = self._handle().beam_search(self._features(visual_inputs).detach(), beam_size, 50)
        lens = lens.tolist()
        out = [seqs[i:i + 1, :lens[i]] for i in range(len(lens))]
        return out[0] if len(out) == 1 else out

    def forward(self, visual_inputs, captions, lengths, rng=None):
        """NIC_Model.py:246-260: [0] of the result = packed logits (no autograd graph on this path)."""
        train = self.training
        logits = self._handle().xe_forward(self._features(visual_inputs).detach(), captions, list(lengths),
                                           (rng or self._next_rng()) if train else None, train=train, want_logits=True)
        return (logits, None)

    def eval_test_image(self, visual_inputs, caption_vocab, max_len=20, eval_beam_size=-1):
        """NIC_Model.py:306-331 -> (caption words, []): NIC has no attention maps."""
        feats = self._features(visual_inputs).detach()
        assert feats.size(0) == 1
        if eval_beam_size != -1:
            ids = self.beam_search_sampler(visual_inputs, eval_beam_size)
        else:
            ids = self.sampler(visual_inputs, max_len)
        caption = []
        for word_id in ids[0].cpu().numpy():
            word = caption_vocab.ix2word[int(word_id)]
            if word == "<end>":
                break
            elif word != "<sta>":
                caption.append(word)
        return caption, []
