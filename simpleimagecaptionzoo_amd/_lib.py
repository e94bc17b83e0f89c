"""ctypes binding of libicz.so (the C ABI declared in include/icz.h).

The HIP library is the product: there is no CPU or eager-PyTorch fallback.  Loading fails loudly if the
shared object is missing (build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C simpleimagecaptionzoo_amd/csrc`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libicz.so")


class IczError(RuntimeError):
    pass


GRAD_READY_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int32)       # icz_grad_ready_cb


class ButdDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("R", "D", "H", "E", "A", "V", "max_rows", "max_len")]


BUTD_PARAM_FIELDS = (
    "embed_weight",
    "td_w_ih", "td_w_hh", "td_b_ih", "td_b_hh",
    "lm_w_ih", "lm_w_hh", "lm_b_ih", "lm_b_hh",
    "enc_att_v", "enc_att_g", "enc_att_b",
    "dec_att_v", "dec_att_g", "dec_att_b",
    "affine_v", "affine_g", "affine_b",
    "predict_v", "predict_g", "predict_b",
)
# reference state_dict key (without the "decoder." prefix) of each field, Models/BUTD_Model.py:75-84
BUTD_PARAM_KEYS = (
    "embed.0.weight",
    "TD_atten.weight_ih", "TD_atten.weight_hh", "TD_atten.bias_ih", "TD_atten.bias_hh",
    "language_model.weight_ih", "language_model.weight_hh", "language_model.bias_ih", "language_model.bias_hh",
    "atten.enc_att.weight_v", "atten.enc_att.weight_g", "atten.enc_att.bias",
    "atten.dec_att.weight_v", "atten.dec_att.weight_g", "atten.dec_att.bias",
    "atten.affine.weight_v", "atten.affine.weight_g", "atten.affine.bias",
    "predict.weight_v", "predict.weight_g", "predict.bias",
)


class NicDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("E", "H", "V", "max_rows", "max_len")]


NIC_PARAM_FIELDS = ("embed_weight", "w_ih", "w_hh", "b_ih", "b_hh", "predict_v", "predict_g", "predict_b")
# reference state_dict keys of Models/NIC_Model.py DecoderRNN (:47-49)
NIC_PARAM_KEYS = ("embed.weight", "lstm.weight_ih", "lstm.weight_hh", "lstm.bias_ih", "lstm.bias_hh",
                  "predict.weight_v", "predict.weight_g", "predict.bias")


class AoaDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("R", "D", "Hd", "E", "V", "NH", "max_rows", "max_len")]


def _aoa_block_keys(block, norm):
    return tuple(block + n for n in ("linear_Q.weight", "linear_Q.bias", "linear_K.weight", "linear_K.bias", "linear_V.weight",
                                     "linear_V.bias", "aoa_module.0.weight", "aoa_module.0.bias")) + (norm + "gain", norm + "bias")


# reference state_dict keys of AoADetection_Captioner (Models/AoA_Model.py:657-667) in the pointer order of icz_aoa_params
AOA_PARAM_KEYS = (("img_feats_porjection.0.weight", "img_feats_porjection.0.bias")
                  + sum((_aoa_block_keys("aoa_refine.aoa_layers.%d.aoa_block." % l, "aoa_refine.aoa_layers.%d.sublayer.norm." % l)
                         for l in range(6)), ())
                  + ("aoa_refine.norm.gain", "aoa_refine.norm.bias",
                     "decoder.lstm.weight_ih", "decoder.lstm.weight_hh", "decoder.lstm.bias_ih", "decoder.lstm.bias_hh")
                  + _aoa_block_keys("decoder.aoa_block.", "decoder.h_norm.")
                  + ("decoder.embed.0.weight", "decoder.predict.weight_v", "decoder.predict.weight_g", "decoder.predict.bias"))
AOA_DECODER_KEYS = tuple(k for k in AOA_PARAM_KEYS if k.startswith("decoder."))


class AoaParams(C.Structure):      # icz_aoa_params is 82 pointers with no padding: addressed here as a flat table
    _fields_ = [("p%d" % i, C.c_void_p) for i in range(len(AOA_PARAM_KEYS))]


class AoaRng(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("uniforms", C.c_void_p)] + [(n, C.c_void_p) for n in (
        "proj_mask", "ref_att_mask", "ref_aoa_mask", "ref_sc_mask", "emb_mask", "ctx_mask", "att_mask", "out_mask")]


class NicParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in NIC_PARAM_FIELDS]


class ButdParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in BUTD_PARAM_FIELDS]


class Rng(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("uniforms", C.c_void_p), ("emb_mask", C.c_void_p),
                ("att_mask", C.c_void_p), ("out_mask", C.c_void_p)]


_lib = None


def lib():
    """Load libicz.so once; raise IczError (never fall back) if it is missing or lacks a symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IczError("libicz.so not found at %s -- the HIP extension is required (no fallback); run "
                       "__graft_entry__.build()" % LIB_PATH)
    # torch first: its wheel carries its own libamdhip64.so, and libicz.so (linked against the same soname) must bind to THAT runtime --
    # loaded the other way round the process holds two HIP runtimes and the second one finds no device ("no ROCm-capable device is
    # detected" from the first hipMalloc; seen with __graft_entry__.build() followed by smoke() in one process)
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    sig = {
        "icz_last_error": (C.c_char_p, []),
        "icz_version": (C.c_char_p, []),
        "icz_butd_create": (C.c_int, [C.POINTER(ButdDims), C.POINTER(vp)]),
        "icz_butd_destroy": (C.c_int, [vp]),
        "icz_butd_bind_params": (C.c_int, [vp, C.POINTER(ButdParams)]),
        "icz_butd_set_option": (C.c_int, [vp, C.c_char_p, i32]),
        "icz_butd_set_grad_callback": (C.c_int, [vp, GRAD_READY_CB, vp]),
        "icz_aoa_set_grad_callback": (C.c_int, [vp, GRAD_READY_CB, vp]),
        "icz_butd_set_mask_sum_global": (C.c_int, [vp, vp, vp]),
        "icz_butd_refresh_weights": (C.c_int, [vp, vp]),
        "icz_butd_greedy": (C.c_int, [vp, vp, i32, i32, vp, vp, vp]),
        "icz_butd_step": (C.c_int, [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "icz_butd_sample": (C.c_int, [vp, vp, i32, i32, C.POINTER(Rng), vp, vp, vp]),
        "icz_butd_scst_rollouts": (C.c_int, [vp, vp, i32, i32, C.POINTER(Rng), vp, vp, vp, vp]),
        "icz_butd_sample_backward": (C.c_int, [vp, vp, C.POINTER(ButdParams), vp, vp, f32, vp]),
        "icz_butd_sample_backward_dlogp": (C.c_int, [vp, vp, C.POINTER(ButdParams), vp]),
        "icz_butd_xe_backward_dlogits": (C.c_int, [vp, vp, C.POINTER(ButdParams), vp]),
        "icz_butd_beam_search": (C.c_int, [vp, vp, i32, i32, i32, vp, vp, vp]),
        "icz_butd_sample_mask_sum": (C.c_int, [vp, vp, vp]),
        "icz_butd_xe_forward": (C.c_int, [vp, vp, vp, i32, i32, C.POINTER(i32), C.POINTER(Rng), i32, vp, vp]),
        "icz_butd_xe_backward": (C.c_int, [vp, f32, C.POINTER(ButdParams), vp, f32, vp]),
        "icz_butd_set_scheduled_sampling": (C.c_int, [vp, f32, vp, vp]),
        "icz_nic_set_scheduled_sampling": (C.c_int, [vp, f32, vp, vp]),
        "icz_aoa_set_scheduled_sampling": (C.c_int, [vp, f32, vp, vp]),
        "icz_adam_clamp_step": (C.c_int, [vp, vp, vp, vp, i64, f32, f32, i32, vp]),
        "icz_nic_create": (C.c_int, [C.POINTER(NicDims), C.POINTER(vp)]),
        "icz_nic_destroy": (C.c_int, [vp]),
        "icz_nic_bind_params": (C.c_int, [vp, C.POINTER(NicParams)]),
        "icz_nic_refresh_weights": (C.c_int, [vp, vp]),
        "icz_nic_greedy": (C.c_int, [vp, vp, i32, i32, vp, vp]),
        "icz_nic_sample": (C.c_int, [vp, vp, i32, i32, C.POINTER(Rng), vp, vp, vp]),
        "icz_nic_sample_backward": (C.c_int, [vp, vp, C.POINTER(NicParams), vp, vp, vp, f32, vp]),
        "icz_nic_xe_forward": (C.c_int, [vp, vp, vp, i32, i32, C.POINTER(i32), C.POINTER(Rng), i32, vp, vp]),
        "icz_nic_xe_backward": (C.c_int, [vp, f32, C.POINTER(NicParams), vp, vp, f32, vp]),
        "icz_butd_saved_alphas": (C.c_int, [vp, vp, vp]),
        "icz_aoa_saved_alphas": (C.c_int, [vp, vp, vp]),
        "icz_nic_set_norm_global": (C.c_int, [vp, vp, vp]),
        "icz_aoa_set_norm_global": (C.c_int, [vp, vp, vp]),
        "icz_nic_beam_search": (C.c_int, [vp, vp, i32, i32, i32, vp, vp, vp]),
        "icz_aoa_create": (C.c_int, [C.POINTER(AoaDims), C.POINTER(vp)]),
        "icz_aoa_destroy": (C.c_int, [vp]),
        "icz_aoa_bind_params": (C.c_int, [vp, C.POINTER(AoaParams)]),
        "icz_aoa_refresh_weights": (C.c_int, [vp, vp]),
        "icz_aoa_set_regions": (C.c_int, [vp, i32, vp, vp, i32]),
        "icz_aoa_set_option": (C.c_int, [vp, C.c_char_p, i32]),
        "icz_nic_set_option": (C.c_int, [vp, C.c_char_p, i32]),
        "icz_aoa_refine": (C.c_int, [vp, vp, i32, vp, vp]),
        "icz_aoa_greedy": (C.c_int, [vp, vp, i32, i32, vp, vp]),
        "icz_aoa_beam_search": (C.c_int, [vp, vp, i32, i32, i32, vp, vp, vp]),
        "icz_aoa_sample": (C.c_int, [vp, vp, i32, i32, C.POINTER(AoaRng), vp, vp, vp]),
        "icz_aoa_scst_rollouts": (C.c_int, [vp, vp, i32, i32, C.POINTER(AoaRng), vp, vp, vp, vp]),
        "icz_aoa_sample_backward": (C.c_int, [vp, vp, C.POINTER(AoaParams), vp, vp, f32, vp]),
        "icz_aoa_xe_forward": (C.c_int, [vp, vp, vp, i32, i32, C.POINTER(i32), C.POINTER(AoaRng), i32, vp, vp]),
        "icz_aoa_xe_backward": (C.c_int, [vp, f32, C.POINTER(AoaParams), vp, f32, vp]),
        "icz_ciderd_create": (C.c_int, [vp, vp, i64, C.c_double, vp, C.POINTER(vp)]),
        "icz_ciderd_destroy": (C.c_int, [vp]),
        "icz_ciderd_reward": (C.c_int, [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "icz_ciderd_cook_host": (C.c_int, [vp, vp, i64, C.c_double, vp, vp, i32, i64, vp, vp, vp, vp, vp, vp, C.POINTER(i64)]),
        "icz_ciderd_vocab_create": (C.c_int, [C.c_char_p, vp, vp, i32, i32, C.POINTER(vp)]),
        "icz_ciderd_vocab_destroy": (C.c_int, [vp]),
        "icz_ciderd_vocab_oov_id": (C.c_int, [vp, C.c_char_p, i32, C.POINTER(i32)]),
        "icz_ciderd_cook_text": (C.c_int, [vp, vp, vp, i64, C.c_double, C.c_char_p, i64, i32, i64, vp, vp, vp, vp, vp, vp, C.POINTER(i64)]),
        "icz_ciderd_reward_indexed": (C.c_int, [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "icz_prof_begin": (C.c_int, []),
        "icz_prof_select": (C.c_int, [i32]),
        "icz_prof_pair_overhead": (C.c_int, [vp, i32, C.POINTER(C.c_double)]),
        "icz_prof_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
        "icz_kprof_begin": (C.c_int, []),
        "icz_kprof_end": (C.c_int, [i32, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
        "icz_prof_stream_rate": (C.c_int, [vp, C.c_size_t, i32, vp, C.POINTER(C.c_double)]),
        "icz_adam_clamp_multi": (C.c_int, [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), f32, f32, i32, vp]),
        "icz_gemm_f32": (C.c_int, [i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, C.c_size_t, vp]),
        "icz_gemm_workspace_floats": (C.c_size_t, [i32, i32]),
        "icz_gemm_set_big_cfg": (C.c_int, [i32]),
        "icz_gemm_big_cfg_for": (C.c_int, [i32, i32, i32, i32, i32]),
        "icz_gemm_tn_grouped": (C.c_int, [vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
    }
    for name, (res, args) in sig.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise IczError("libicz.so lacks symbol %s (stale build?)" % name) from e
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def check(status):
    if status != 0:
        raise IczError("libicz status %d: %s" % (status, lib().icz_last_error().decode()))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
