"""ctypes binding of libditto_hip.so (include/ditto_hip.h).

There is NO fallback: if the library is missing or a call fails this module raises.
Build it with `python -m ditto_tts_amd.build` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DITTO_HIP_LIB") or os.path.join(HERE, "libditto_hip.so")   # override: diagnostic builds (tools/)

OK, ERR_ARG, ERR_SHAPE, ERR_HIP, ERR_SIZE = range(5)
CFG_FP8_LINEAR = 1

KERNEL_CLASSES = ["layernorm", "gemm_qkv_rope", "gemm_q_proj", "gemm_out_proj", "gemm_gated_mlp", "gemm_fc2",
                  "gemm_final", "attn_self", "attn_cross", "adaln", "p_sample_update"]
KC_COUNT = len(KERNEL_CLASSES)


class DittoHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libditto_hip error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("hidden_dim", "num_layers", "num_heads", "time_dim", "text_dim", "diffusion_steps", "flags")]


_LAYER_FIELDS = ["norm1_weight", "norm1_bias", "attn_in_proj_weight", "attn_in_proj_bias",
                 "norm2_weight", "norm2_bias", "cross_in_proj_weight", "cross_in_proj_bias",
                 "cross_out_proj_weight", "cross_out_proj_bias", "norm3_weight", "norm3_bias",
                 "mlp_fc1_weight", "mlp_fc1_bias", "gate_weight", "gate_bias", "mlp_fc2_weight", "mlp_fc2_bias"]
_GLOBAL_FIELDS = ["t_embedding_weight", "time_embed_0_weight", "time_embed_0_bias", "time_embed_2_weight",
                  "time_embed_2_bias", "ada_time_mlp_weight", "ada_time_mlp_bias", "ada_text_mlp_weight",
                  "ada_text_mlp_bias", "proj_in_weight", "proj_in_bias", "proj_out_weight", "proj_out_bias",
                  "rotary_inv_freq"]

# struct field -> reference state_dict key (layer keys are relative to "blocks.{i}.")
LAYER_KEY = {
    "norm1_weight": "norm1.weight", "norm1_bias": "norm1.bias",
    "attn_in_proj_weight": "attn.in_proj_weight", "attn_in_proj_bias": "attn.in_proj_bias",
    "norm2_weight": "norm2.weight", "norm2_bias": "norm2.bias",
    "cross_in_proj_weight": "cross_attn.in_proj_weight", "cross_in_proj_bias": "cross_attn.in_proj_bias",
    "cross_out_proj_weight": "cross_attn.out_proj.weight", "cross_out_proj_bias": "cross_attn.out_proj.bias",
    "norm3_weight": "norm3.weight", "norm3_bias": "norm3.bias",
    "mlp_fc1_weight": "mlp_fc1.weight", "mlp_fc1_bias": "mlp_fc1.bias",
    "gate_weight": "gate.weight", "gate_bias": "gate.bias",
    "mlp_fc2_weight": "mlp_fc2.weight", "mlp_fc2_bias": "mlp_fc2.bias",
}
GLOBAL_KEY = {
    "t_embedding_weight": "t_embedding.weight",
    "time_embed_0_weight": "time_embed.0.weight", "time_embed_0_bias": "time_embed.0.bias",
    "time_embed_2_weight": "time_embed.2.weight", "time_embed_2_bias": "time_embed.2.bias",
    "ada_time_mlp_weight": "ada_ln.time_mlp.1.weight", "ada_time_mlp_bias": "ada_ln.time_mlp.1.bias",
    "ada_text_mlp_weight": "ada_ln.text_mlp.1.weight", "ada_text_mlp_bias": "ada_ln.text_mlp.1.bias",
    "proj_in_weight": "proj_in.weight", "proj_in_bias": "proj_in.bias",
    "proj_out_weight": "proj_out.weight", "proj_out_bias": "proj_out.bias",
    "rotary_inv_freq": "rotary.inv_freq",
}


class LayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _LAYER_FIELDS]


class Weights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _GLOBAL_FIELDS] + [("layers", C.POINTER(LayerWeights))]


# the speech-length predictor's decoder stack: struct field -> state_dict key relative to "transformer.layers.{i}."
SLP_LAYER_KEY = {
    "self_in_proj_weight": "self_attn.in_proj_weight", "self_in_proj_bias": "self_attn.in_proj_bias",
    "self_out_proj_weight": "self_attn.out_proj.weight", "self_out_proj_bias": "self_attn.out_proj.bias",
    "cross_in_proj_weight": "multihead_attn.in_proj_weight", "cross_in_proj_bias": "multihead_attn.in_proj_bias",
    "cross_out_proj_weight": "multihead_attn.out_proj.weight", "cross_out_proj_bias": "multihead_attn.out_proj.bias",
    "linear1_weight": "linear1.weight", "linear1_bias": "linear1.bias",
    "linear2_weight": "linear2.weight", "linear2_bias": "linear2.bias",
    "norm1_weight": "norm1.weight", "norm1_bias": "norm1.bias", "norm2_weight": "norm2.weight",
    "norm2_bias": "norm2.bias", "norm3_weight": "norm3.weight", "norm3_bias": "norm3.bias",
}


class SlpConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("d_model", "nhead", "num_layers", "dim_feedforward", "num_classes")]


class SlpLayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in SLP_LAYER_KEY]


class SlpWeights(C.Structure):
    _fields_ = [("layers", C.POINTER(SlpLayerWeights)), ("length_predictor_weight", C.c_void_p),
                ("length_predictor_bias", C.c_void_p)]


class CallOpts(C.Structure):
    """ditto_call_opts (ABI 9): the per-call switches that decide which bits an utterance gets.  -1 = inherit."""
    _fields_ = [("class_rows", C.c_int32), ("residual_bf16", C.c_int32), ("fr_mask", C.c_int32), ("lnq", C.c_int32),
                ("reserved", C.c_int32 * 4)]

    def __init__(self, class_rows=None, residual_bf16=None, fr_mask=None, lnq=None):
        super().__init__(-1 if class_rows is None else min(int(class_rows), 0x7FFFFFFF),
                         -1 if residual_bf16 is None else int(residual_bf16),
                         -1 if fr_mask is None else int(fr_mask), -1 if lnq is None else int(lnq))

    def __repr__(self):
        return f"CallOpts(class_rows={self.class_rows}, residual_bf16={self.residual_bf16}, fr_mask={self.fr_mask}, lnq={self.lnq})"


# ditto_layer_grads / ditto_grads have the layout of the weight structs (one pointer per state_dict key)
LayerGrads, Grads = LayerWeights, Weights

# every symbol include/ditto_hip.h declares: name -> (restype, argtypes)
_vp, _i, _sz, _f, _u64 = C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.c_uint64
SYMBOLS = {
    "ditto_abi_version": (_i, []),
    "ditto_last_error": (C.c_char_p, []),
    "ditto_arena_bytes": (_sz, [C.POINTER(Config)]),
    "ditto_cond_bytes": (_sz, [C.POINTER(Config), _i, _i]),
    "ditto_workspace_bytes": (_sz, [C.POINTER(Config), _i, _i, _i]),
    "ditto_model_create": (_i, [C.POINTER(Config), C.POINTER(Weights), _vp, _sz, _vp, C.POINTER(_vp)]),
    "ditto_model_destroy": (_i, [_vp]),
    "ditto_rope_tables": (_i, [_vp, _i, _vp, _vp, _vp]),
    "ditto_text_precompute": (_i, [_vp, _vp, _i, _i, _vp, _sz, _vp, _sz, _vp]),
    "ditto_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ditto_forward_opts": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp, C.POINTER(CallOpts)]),
    "ditto_call_opts_push": (_i, [C.POINTER(CallOpts)]),
    "ditto_call_opts_pop": (_i, []),
    "ditto_call_opts_current": (_i, [C.POINTER(CallOpts)]),
    "ditto_block_forward": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "ditto_block_forward_taps": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ditto_global_adaln_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "ditto_global_adaln": (_i, [_vp] * 7 + [_i] * 6 + [_vp, _vp, _sz, _vp]),
    "ditto_apply_rope_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ditto_p_sample_update": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _sz, _vp]),
    "ditto_p_sample": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "ditto_p_sample_opts": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp, C.POINTER(CallOpts)]),
    "ditto_denoise_steps": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ditto_denoise_steps_opts": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp,
                                      C.POINTER(CallOpts)]),
    "ditto_noise_normal": (_i, [_vp, _vp, C.c_uint32, _i, _sz, _vp]),
    "ditto_p_sample_seeded": (_i, [_vp, _vp, _vp, _vp, _vp, C.c_uint32, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "ditto_p_sample_seeded_opts": (_i, [_vp, _vp, _vp, _vp, _vp, C.c_uint32, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp,
                                        C.POINTER(CallOpts)]),
    "ditto_q_sample": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _sz, _vp]),
    "ditto_layernorm_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "ditto_gemm_bf16": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ditto_gemm_ln_bf16": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ditto_gemm_tn_bf16": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "ditto_attention_bf16": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _sz, _vp]),
    "ditto_attention_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "ditto_vq_argmin": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ditto_embedding_gather": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "ditto_code_embed_mean": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ditto_linear_update": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _sz, _vp]),
    "ditto_cfg_combine": (_i, [_vp, _vp, _f, _sz, _vp]),
    "ditto_train_arena_bytes": (_sz, [C.POINTER(Config)]),
    "ditto_tape_bytes": (_sz, [C.POINTER(Config), _i, _i, _i]),
    "ditto_train_workspace_bytes": (_sz, [C.POINTER(Config), _i, _i, _i]),
    "ditto_train_attach": (_i, [_vp, C.POINTER(Weights), _vp, _sz, _vp]),
    "ditto_train_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _f, _u64, _vp, _vp, _sz, _vp, _sz, _vp]),
    "ditto_train_backward": (_i, [_vp, C.POINTER(Weights), _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _f, _u64, _vp, _sz,
                                  C.POINTER(Grads), _vp, _sz, _vp]),
    "ditto_train_forward_opts": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _f, _u64, _vp, _vp, _sz, _vp, _sz, _vp,
                                      C.POINTER(CallOpts)]),
    "ditto_train_backward_opts": (_i, [_vp, C.POINTER(Weights), _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _f, _u64, _vp, _sz,
                                       C.POINTER(Grads), _vp, _sz, _vp, C.POINTER(CallOpts)]),
    "ditto_train_backward_layers": (_i, [_vp, C.POINTER(Weights), _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _f, _u64, _vp, _sz,
                                         C.POINTER(Grads), _vp, _sz, _vp, C.POINTER(CallOpts), _i, _i]),
    "ditto_layernorm_bwd_scratch_bytes": (_sz, [_i, _i, _i]),
    "ditto_layernorm_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _vp]),
    "ditto_attention_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "ditto_attention_bwd_bf16": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _i,
                                      _i, _i, _i, _i, _i, _f, _f, _u64, _i, _vp, _sz, _vp]),
    "ditto_attention_dropout_bf16": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _f, _u64, _i,
                                          _vp, _sz, _vp]),
    "ditto_quantize_rows_fp8": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "ditto_layernorm_fp8": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "ditto_gemm_fp8": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ditto_slp_arena_bytes": (_sz, [C.POINTER(SlpConfig)]),
    "ditto_slp_workspace_bytes": (_sz, [C.POINTER(SlpConfig), _i, _i, _i]),
    "ditto_slp_create": (_i, [C.POINTER(SlpConfig), C.POINTER(SlpWeights), _vp, _sz, _vp, C.POINTER(_vp)]),
    "ditto_slp_destroy": (None, [_vp]),
    "ditto_slp_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "ditto_attention_causal_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "ditto_attention_causal_bf16": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _sz, _vp]),
    "ditto_layernorm_dual": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "ditto_gemm_lnq_bf16": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ditto_set_option": (_i, [C.c_char_p, _i]),
    "ditto_get_option": (_i, [C.c_char_p, C.POINTER(C.c_int)]),
    "ditto_full_row_plan": (_i, [C.POINTER(Config), _i, _i, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ditto_full_row_plan_opts": (_i, [C.POINTER(Config), _i, _i, C.POINTER(CallOpts), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(C.c_int)]),
    "ditto_profile_enable": (_i, [_vp, _i]),
    "ditto_profile_read": (_i, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    "ditto_kernel_class_name": (C.c_char_p, [_i]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libditto_hip.so (once).  Raises if it has not been built — there is no other path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is the only compute path of ditto_tts_amd. "
                "Build it with `python -m ditto_tts_amd.build` (needs hipcc).")
        # torch first: libditto_hip.so and torch must share ONE HIP runtime (both want SONAME libamdhip64.so.7; the
        # first one loaded wins).  Loading ours first binds the process to /opt/rocm's runtime, under which torch's
        # device initialisation and ours disagreed on a GPU box ("no ROCm-capable device is detected" at the first launch).
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        frozen = bool(os.environ.get("DITTO_HIP_LIB"))   # a diagnostic or FROZEN earlier build for same-box A/Bs (tools/):
        for name, (res, args) in SYMBOLS.items():        # it may predate entry points added since (they are then absent)
            try:
                fn = getattr(l, name)  # AttributeError if the .so does not export a declared symbol
            except AttributeError:
                if frozen:
                    continue
                raise
            fn.restype, fn.argtypes = res, args
        if l.ditto_abi_version() != 10 and not (frozen and l.ditto_abi_version() in (7, 8, 9)):
            raise RuntimeError("libditto_hip.so ABI version mismatch")
        _lib = l
    return _lib


def check(rc: int):
    if rc != OK:
        raise DittoHipError(rc, lib().ditto_last_error().decode(errors="replace"))


def set_option(name: str, value: int):
    """ditto_set_option: process-wide tuning switches of libditto_hip.so (include/ditto_hip.h)."""
    check(lib().ditto_set_option(name.encode(), int(value)))


def get_option(name: str) -> int:
    """ditto_get_option: the current value of a switch."""
    v = C.c_int(0)
    check(lib().ditto_get_option(name.encode(), C.byref(v)))
    return int(v.value)


def _has_call_opts() -> bool:
    return hasattr(lib(), "ditto_call_opts_push") and lib().ditto_abi_version() >= 9


class call_opts:
    """Context manager: the per-call options (class_rows, residual_bf16, fr_mask, lnq) in force for every library call THIS
    THREAD makes inside the block (ditto_call_opts_push / _pop).  Nothing process-wide changes: other threads and streams keep
    their own options.  Nests; a field left None inherits the enclosing block, else the process default (set_option)."""

    def __init__(self, class_rows=None, residual_bf16=None, fr_mask=None, lnq=None):
        self.opts = CallOpts(class_rows, residual_bf16, fr_mask, lnq)
        self._legacy = None

    def __enter__(self):
        if _has_call_opts():
            check(lib().ditto_call_opts_push(C.byref(self.opts)))
        else:   # a frozen pre-ABI-9 library (DITTO_HIP_LIB, tools/ A/Bs): the old process-wide switches
            names = {"fr_class_rows": self.opts.class_rows, "residual_bf16": self.opts.residual_bf16,
                     "fr_mask": self.opts.fr_mask, "lnq": self.opts.lnq}
            self._legacy = {k: get_option(k) for k, v in names.items() if v >= 0}
            for k, v in names.items():
                if v >= 0:
                    set_option(k, v)
        return self

    def __exit__(self, *exc):
        if self._legacy is None:
            check(lib().ditto_call_opts_pop())
        else:
            for k, v in self._legacy.items():
                set_option(k, v)
            self._legacy = None
        return False


class batch_class(call_opts):
    """Context manager: every forward THIS THREAD makes inside decides its kernel class (low-latency / tiled GEMM + LayerNorm /
    full-row GEMM with the LayerNorms fused: different summation orders, so different last bits) as a batch of `rows` = B x N
    rows would.  A caller that splits one batch over several launches or GPUs wraps the pieces in `batch_class(rows of the
    unsplit batch)` — or passes CallOpts(class_rows=...) to the engine call itself: an utterance's bits are then the same however
    the batch was split (dist.sample_sharded and SpeechGenerator's seeds= path do).  Since ABI 9 this is a property of the
    CALLS, not of the process (ditto_call_opts): no other thread's forwards are affected.  Nests.  A launch that cannot take the
    pinned class (a full-row class with fewer than 64 rows in the launch) raises DittoHipError instead of silently running
    another class."""

    def __init__(self, rows: int):
        super().__init__(class_rows=int(rows))
        self.rows = int(rows)


def current_opts() -> CallOpts:
    """The options a call made by this thread NOW (without its own CallOpts) would run under, every field resolved."""
    o = CallOpts()
    if _has_call_opts():
        check(lib().ditto_call_opts_current(C.byref(o)))
    else:
        o.class_rows, o.residual_bf16 = get_option("fr_class_rows"), get_option("residual_bf16")
        o.fr_mask, o.lnq = get_option("fr_mask"), get_option("lnq")
    return o


def full_row_plan(cfg, B: int, N: int, opts: "CallOpts | None" = None):
    """(outproj, fc2): which of a block's two fused GEMM + LayerNorm launches a forward over B x N rows takes
    (ditto_full_row_plan; host arithmetic, no GPU call) under this thread's options, or under `opts`."""
    a, b = C.c_int(0), C.c_int(0)
    c = make_config(cfg)
    if opts is not None:
        check(lib().ditto_full_row_plan_opts(C.byref(c), B, N, C.byref(opts), C.byref(a), C.byref(b), None))
    else:
        check(lib().ditto_full_row_plan(C.byref(c), B, N, C.byref(a), C.byref(b)))
    return bool(a.value), bool(b.value)


def stream_is_bf16(cfg, B: int, N: int, opts: "CallOpts | None" = None) -> bool:
    """Does a forward of B x N rows carry its residual stream as bf16 (ditto_forward's rule, answered by the library)?"""
    a, b, h = C.c_int(0), C.c_int(0), C.c_int(0)
    c = make_config(cfg)
    check(lib().ditto_full_row_plan_opts(C.byref(c), B, N, C.byref(opts) if opts is not None else None, C.byref(a), C.byref(b),
                                         C.byref(h)))
    return bool(h.value)


def set_low_latency(on: bool = True):
    """The explicit split-K rule of rounds 1-3 (a workgroup target of 256) for the long-K GEMMs of 1-2 utterance batches.
    Since round 4 the DEFAULT already splits them (the low-latency class: a rule that depends on K only, so that inside the
    class an utterance's bits do not depend on its batch neighbours); this switch trades that invariance for one more split at
    B = 1.  set_option("splitk_wgs", -1) turns every split off."""
    set_option("splitk_wgs", 256 if on else 0)


def make_config(cfg) -> Config:
    return Config(cfg.hidden_dim, cfg.num_layers, cfg.num_heads, cfg.time_dim, cfg.text_dim, cfg.diffusion_steps,
                  CFG_FP8_LINEAR if getattr(cfg, "fp8_linear", False) else 0)
