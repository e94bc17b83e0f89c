"""Host-side owner of one libicz BUTD decoder handle (thin: shapes, pointers, stream)."""
import ctypes as C

import torch

from . import _lib
from ._lib import BUTD_PARAM_FIELDS, BUTD_PARAM_KEYS, ButdDims, ButdParams, Rng, check, lib, ptr, stream_ptr


class ButdHandle:
    """Wraps icz_butd_* for a fixed architecture (R, D, H, E, A, V) and row / step capacity."""

    def __init__(self, R, D, H, E, A, V, max_rows, max_len=20, device="cuda:0"):
        self.R, self.D, self.H, self.E, self.A, self.V = R, D, H, E, A, V
        self.max_rows, self.max_len = max_rows, max_len
        self.device = torch.device(device)
        self._h = C.c_void_p()
        self._params = None
        self._persistent = False
        self._bufs = {}
        dims = ButdDims(R, D, H, E, A, V, max_rows, max_len)
        with torch.cuda.device(self.device):
            check(lib().icz_butd_create(C.byref(dims), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().icz_butd_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def enable_graphs(self, on=True):
        """Capture greedy / sample / sample_backward into hipGraphs and replay them.  Implies persistent output
        buffers: the tensors returned by those calls are reused (overwritten) by the next call of the same shape."""
        self._persistent = bool(on)
        check(lib().icz_butd_set_option(self._h, b"graphs", 1 if on else 0))

    def set_grad_callback(self, fn):
        """fn(stage) is called while a backward call is being enqueued, each time a group of gradients is complete in
        stream order (include/icz.h: icz_butd_set_grad_callback); None removes it."""
        self._grad_cb = _lib.GRAD_READY_CB(lambda user, stage: fn(int(stage))) if fn is not None else _lib.GRAD_READY_CB()
        check(lib().icz_butd_set_grad_callback(self._h, self._grad_cb, None))

    def set_concurrent(self, on=True):
        check(lib().icz_butd_set_option(self._h, b"concurrent", 1 if on else 0))

    def set_option(self, name, value):
        """icz_butd_set_option (include/icz.h): "graphs", "concurrent"."""
        check(lib().icz_butd_set_option(self._h, name.encode(), int(value)))

    def _buf(self, name, shape, dtype):
        if not self._persistent:
            return torch.zeros(shape, dtype=dtype, device=self.device)
        key = (name,) + tuple(shape)
        t = self._bufs.get(key)
        if t is None:
            t = torch.zeros(shape, dtype=dtype, device=self.device)
            self._bufs[key] = t
        return t

    # ---- parameters ---------------------------------------------------------------------------
    def bind(self, tensors):
        """tensors: {reference state_dict key without 'decoder.': fp32 CUDA tensor}.  The tensors are used in
        place (no copy) and must stay alive; call refresh() after every update."""
        st = ButdParams()
        keep = []
        for field, key in zip(BUTD_PARAM_FIELDS, BUTD_PARAM_KEYS):
            t = tensors[key]
            if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
                raise _lib.IczError("parameter %s must be a contiguous fp32 CUDA tensor" % key)
            setattr(st, field, t.data_ptr())
            keep.append(t)
        self._params = keep
        check(lib().icz_butd_bind_params(self._h, C.byref(st)))
        self.refresh()

    def refresh(self):
        check(lib().icz_butd_refresh_weights(self._h, stream_ptr()))

    # ---- decode -------------------------------------------------------------------------------
    def _check_feats(self, feats):
        if feats.dtype != torch.float32 or not feats.is_cuda:
            raise _lib.IczError("feats must be an fp32 CUDA tensor")
        if feats.dim() != 3 or feats.shape[1] != self.R or feats.shape[2] != self.D:
            raise _lib.IczError("feats must be (B,%d,%d), got %s" % (self.R, self.D, tuple(feats.shape)))
        return feats.contiguous()

    def greedy(self, feats, max_len=20, want_alphas=False):
        """DecoderRNN.sample (Models/BUTD_Model.py:153-189) -> ids (B,max_len) int64 [, alphas (B,max_len,R)]."""
        feats = self._check_feats(feats)
        B = feats.shape[0]
        ids = self._buf("greedy_ids", (B, max_len), torch.int64)
        alphas = self._buf("greedy_alphas", (B, max_len, self.R), torch.float32) if want_alphas else None
        check(lib().icz_butd_greedy(self._h, ptr(feats), B, max_len, ptr(ids), ptr(alphas), stream_ptr()))
        return (ids, alphas) if want_alphas else ids

    def _grad_struct(self, grads):
        st = ButdParams()
        for field, key in zip(BUTD_PARAM_FIELDS, BUTD_PARAM_KEYS):
            t = grads[key]
            if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
                raise _lib.IczError("gradient buffer %s must be a contiguous fp32 CUDA tensor" % key)
            setattr(st, field, t.data_ptr())
        return st

    def new_grads(self):
        """Zeroed gradient buffers, one per bound parameter (same keys / shapes)."""
        return {k: torch.zeros_like(t) for k, t in zip(BUTD_PARAM_KEYS, self._params)}

    def sample(self, feats, max_len=20, rng=None):
        """DecoderRNN.sample_rl (BUTD_Model.py:191-234), dropout on -> (seq int64 (B,T), logprobs (B,T))."""
        feats = self._check_feats(feats)
        B = feats.shape[0]
        rng = rng or make_rng(0)
        seq = self._buf("sample_seq", (B, max_len), torch.int64)
        lp = self._buf("sample_lp", (B, max_len), torch.float32)
        check(lib().icz_butd_sample(self._h, ptr(feats), B, max_len, C.byref(rng), ptr(seq), ptr(lp), stream_ptr()))
        self._live = (feats, rng, seq, lp)
        return seq, lp

    def rollouts(self, feats, max_len=20, rng=None):
        """Greedy baseline + sampled rollout of one SCST step (Engine.py:258-262), run concurrently on the device.
        Returns (greedy_ids, seq, logprobs); identical to greedy() followed by sample()."""
        feats = self._check_feats(feats)
        B = feats.shape[0]
        rng = rng or make_rng(0)
        ids = self._buf("greedy_ids", (B, max_len), torch.int64)
        seq = self._buf("sample_seq", (B, max_len), torch.int64)
        lp = self._buf("sample_lp", (B, max_len), torch.float32)
        check(lib().icz_butd_scst_rollouts(self._h, ptr(feats), B, max_len, C.byref(rng), ptr(ids), ptr(seq), ptr(lp),
                                           stream_ptr()))
        self._live = (feats, rng, seq, lp)
        return ids, seq, lp

    def sample_mask_sum(self):
        out = torch.zeros(1, device=self.device)
        check(lib().icz_butd_sample_mask_sum(self._h, ptr(out), stream_ptr()))
        return out

    def set_mask_sum_global(self, t):
        """DP: hand the all-reduced loss normaliser (mask sum of the rollout / token count of the XE batch) over as a 1-element
        device tensor; then sample_backward(..., mask_sum_global=-1) / xe_backward(..., n_tokens_global=-1)."""
        check(lib().icz_butd_set_mask_sum_global(self._h, ptr(t), stream_ptr()))

    def sample_backward(self, reward, grads, mask_sum_global=0.0):
        """RewardCriterion + backward (Utils.py:295-317) for the last sample(); fills `grads`; returns
        (loss, local mask sum) as 1-element device tensors."""
        reward = reward.to(device=self.device, dtype=torch.float32).contiguous()
        loss = self._buf("rl_loss", (1,), torch.float32)
        msum = self._buf("rl_msum", (1,), torch.float32)
        gs = self._grad_struct(grads)
        check(lib().icz_butd_sample_backward(self._h, ptr(reward), C.byref(gs), ptr(loss), ptr(msum),
                                             float(mask_sum_global), stream_ptr()))
        return loss, msum

    def xe_forward(self, feats, captions, lengths, rng=None, train=True, want_logits=False):
        """DecoderRNN.forward (BUTD_Model.py:97-151).  lengths = caption lengths minus one (Engine.py:178),
        sorted descending.  Returns packed logits (sum(lengths), V) if want_logits."""
        feats = self._check_feats(feats)
        B, L = captions.shape
        captions = captions.to(device=feats.device, dtype=torch.int64).contiguous()
        lens = (C.c_int32 * B)(*[int(x) for x in lengths])
        out = torch.empty(sum(int(x) for x in lengths), self.V, device=feats.device) if want_logits else None
        if train and rng is None:
            rng = make_rng(0)
        check(lib().icz_butd_xe_forward(self._h, ptr(feats), ptr(captions), B, L, lens,
                                        C.byref(rng) if rng is not None else None, 1 if train else 0, ptr(out),
                                        stream_ptr()))
        self._live = (feats, rng, captions)
        return out

    def set_scheduled_sampling(self, ss_prob, gate=None, draw=None):
        """Scheduled sampling for the following xe_forward calls (BUTD_Model.py:120-132 with the decoder's `ss_prob`):
        gate / draw are optional explicit uniforms [T, B] (parity tests), default Philox."""
        from .scheduled import handle_set_scheduled_sampling
        handle_set_scheduled_sampling(self, "icz_butd_set_scheduled_sampling", ss_prob, gate, draw)

    def xe_backward(self, grads, smoothing=0.1, n_tokens_global=0.0):
        """LabelSmoothingLoss + backward (Utils.py:268-286) for the last xe_forward(); returns the loss."""
        loss = torch.zeros(1, device=self.device)
        gs = self._grad_struct(grads)
        check(lib().icz_butd_xe_backward(self._h, float(smoothing), C.byref(gs), ptr(loss), float(n_tokens_global),
                                         stream_ptr()))
        return loss

    def sample_backward_dlogp(self, dlogp, grads):
        """BPTT of the last sample() for an upstream gradient d loss / d logprobs (B,T)."""
        dlogp = dlogp.to(device=self.device, dtype=torch.float32).contiguous()
        gs = self._grad_struct(grads)
        check(lib().icz_butd_sample_backward_dlogp(self._h, ptr(dlogp), C.byref(gs), stream_ptr()))

    def saved_alphas(self, B, T):
        """Attention maps [B, T, R] of the forward pass the handle holds (the last xe_forward / sample of B rows, T steps)."""
        out = torch.empty(B, T, self.R, device=self.device)
        check(lib().icz_butd_saved_alphas(self._h, ptr(out), stream_ptr()))
        return out

    def xe_backward_dlogits(self, dpacked, grads):
        """BPTT of the last xe_forward() for an upstream gradient w.r.t. the packed logits (sum(lengths), V)."""
        dpacked = dpacked.to(device=self.device, dtype=torch.float32).contiguous()
        gs = self._grad_struct(grads)
        check(lib().icz_butd_xe_backward_dlogits(self._h, ptr(dpacked), C.byref(gs), stream_ptr()))

    def beam_search(self, feats, beam_size=5, max_steps=50):
        """DecoderRNN.beam_search_sample (BUTD_Model.py:236-318) for all images of `feats` at once.
        Returns (seqs float32 (n_img, max_steps+1) zero-padded, lens int32 (n_img,)); row i[:lens[i]] is what the
        reference returns for image i (leading <sta>, trailing <end> if finished)."""
        feats = self._check_feats(feats)
        n = feats.shape[0]
        seqs = torch.zeros(n, max_steps + 1, dtype=torch.float32, device=feats.device)
        lens = torch.zeros(n, dtype=torch.int32, device=feats.device)
        check(lib().icz_butd_beam_search(self._h, ptr(feats), n, beam_size, max_steps, ptr(seqs), ptr(lens), stream_ptr()))
        return seqs, lens

    def step(self, feats, it, h1, c1, h2, c2):
        """One decoder step from an explicit state (BUTD_Model.py:172-182); state tensors are updated in place.
        Returns (ctx, alpha, logits)."""
        feats = self._check_feats(feats)
        B = feats.shape[0]
        dev = feats.device
        ctx = torch.empty(B, self.D, device=dev)
        alpha = torch.empty(B, self.R, device=dev)
        logits = torch.empty(B, self.V, device=dev)
        check(lib().icz_butd_step(self._h, ptr(feats), B, ptr(it), ptr(h1), ptr(c1), ptr(h2), ptr(c2), ptr(ctx),
                                  ptr(alpha), ptr(logits), stream_ptr()))
        return ctx, alpha, logits


def make_rng(seed=0, uniforms=None, emb_mask=None, att_mask=None, out_mask=None):
    """icz_rng: explicit arrays (uint8 keep-masks / fp32 uniforms on the device) or Philox from `seed` for None."""
    r = Rng()
    r.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    keep = []
    for name, t, dt in (("uniforms", uniforms, torch.float32), ("emb_mask", emb_mask, torch.uint8),
                        ("att_mask", att_mask, torch.uint8), ("out_mask", out_mask, torch.uint8)):
        if t is not None:
            if t.dtype != dt or not t.is_cuda or not t.is_contiguous():
                raise _lib.IczError("%s must be a contiguous %s CUDA tensor" % (name, dt))
            setattr(r, name, t.data_ptr())
            keep.append(t)
    r._keep = keep
    return r


def gemm(layout, X, W, bias=None, nsplit=0):
    """Test/bench entry for icz_gemm_f32.  layout 'nt': X[M,K] W[N,K]; 'nn': X[M,K] W[K,N]; 'tn': X[K,M] W[K,N]."""
    code = {"nt": 0, "nn": 1, "tn": 2}[layout]
    if layout == "nt":
        M, K = X.shape
        N = W.shape[0]
    elif layout == "nn":
        M, K = X.shape
        N = W.shape[1]
    else:
        K, M = X.shape
        N = W.shape[1]
    out = torch.empty(M, N, device=X.device, dtype=torch.float32)
    ws = torch.empty(max(lib().icz_gemm_workspace_floats(M, N), max(nsplit, 1) * M * N), device=X.device,
                     dtype=torch.float32)
    check(lib().icz_gemm_f32(code, ptr(X), X.stride(0), ptr(W), W.stride(0), ptr(bias), ptr(out), N, M, N, K,
                             nsplit, ptr(ws), ws.numel(), stream_ptr()))
    return out


def gemm_set_big_cfg(cfg):
    """icz_gemm_set_big_cfg: -1 per shape (default), 0 the 128 x 128 two-barrier kernel, 1..5 one large-tile configuration, -2 environment."""
    check(lib().icz_gemm_set_big_cfg(int(cfg)))


def gemm_tn_grouped(dY, Xs, outs=None, rows_live=None):
    """Test/bench entry for icz_gemm_tn_grouped: out_j = dY[K,M]^T Xs[j][K,cols_j] in one launch.  Returns the list of outputs."""
    K, M = dY.shape
    n = len(Xs)
    if outs is None:
        outs = [torch.empty(M, x.shape[1], device=dY.device, dtype=torch.float32) for x in Xs]
    xp = (C.c_void_p * n)(*[x.data_ptr() for x in Xs])
    op = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
    ldx = (C.c_int32 * n)(*[x.stride(0) for x in Xs])
    cols = (C.c_int32 * n)(*[x.shape[1] for x in Xs])
    ldo = (C.c_int32 * n)(*[o.stride(0) for o in outs])
    check(lib().icz_gemm_tn_grouped(ptr(dY), dY.stride(0), M, K, n, xp, ldx, cols, op, ldo, ptr(rows_live), stream_ptr()))
    return outs
