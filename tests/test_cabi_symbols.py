"""The C-ABI library loads on a GPU-less host and exports every symbol include/ditto_hip.h declares.
No compute call is made here."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from ditto_tts_amd import hip
from ditto_tts_amd.config import PRESETS, DiTTOConfig


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "ditto_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ditto_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 20
    lib = hip.lib()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in ditto_hip.h but not exported by libditto_hip.so"
        assert n in hip.SYMBOLS, f"{n} has no ctypes prototype in ditto_tts_amd/hip.py"
    assert sorted(hip.SYMBOLS) == names
    assert lib.ditto_abi_version() == 10


def test_struct_layout_matches_header():
    # 14 model-level pointers + layers pointer; 18 per-layer pointers; 7 int32 config fields
    assert C.sizeof(hip.Config) == 28
    assert C.sizeof(hip.LayerWeights) == 18 * 8
    assert C.sizeof(hip.Weights) == 15 * 8
    assert C.sizeof(hip.CallOpts) == 32


def test_size_queries_and_errors():
    lib = hip.lib()
    c = hip.make_config(PRESETS["C2"]["cfg"])
    arena = lib.ditto_arena_bytes(C.byref(c))
    # bf16 live parameters of 12L/768 = 137.83 M -> ~276 MB, plus fp32 vectors and the time table, plus
    # (d = 768) the stage-major second copies of the cross out-proj and fc2 weights the full-row GEMM reads: 12 x 5.9 MB,
    # and of the cross q-projection for the fused norm2 + q-projection kernel (one image per MFMA shape: 12 x 2 x 1.2 MB)
    assert 340e6 < arena < 385e6
    ws = lib.ditto_workspace_bytes(C.byref(c), 32, 1024, 1024)
    cond = lib.ditto_cond_bytes(C.byref(c), 32, 1024)
    assert cond >= 32 * 1024 * 12 * 2 * 768 * 2
    assert ws >= 32 * 1024 * 768 * (4 + 2 + 6 + 8 + 4 + 4)
    bad = hip.Config(768, 12, 12, 256, 512, 50, 0)          # text_dim != hidden_dim
    assert lib.ditto_arena_bytes(C.byref(bad)) == 0
    assert b"text_dim" in lib.ditto_last_error()
    bad = hip.Config(96, 1, 1, 64, 96, 10, 0)                # not a multiple of 64
    assert lib.ditto_arena_bytes(C.byref(bad)) == 0
    bad = hip.Config(192, 1, 3, 64, 192, 10, hip.CFG_FP8_LINEAR)   # fp8 needs hidden_dim % 128 == 0
    assert lib.ditto_arena_bytes(C.byref(bad)) == 0
    # head_dim % 64 != 0 (the reference takes any hidden_dim % num_heads == 0, src/components/DiT.py:78-86) runs on PADDED heads:
    # the paper's XL shape 1152 / 16 = 72 -> heads at a stride of 128 inside the block, attention-side buffers 2048 wide
    xl = hip.Config(1152, 2, 16, 256, 1152, 50, 0)
    plain = hip.Config(1152, 2, 18, 256, 1152, 50, 0)          # 18 heads of 64: no padding
    assert lib.ditto_arena_bytes(C.byref(xl)) > lib.ditto_arena_bytes(C.byref(plain)) > 0
    assert lib.ditto_cond_bytes(C.byref(xl), 2, 64) >= 2 * 64 * 2 * 2 * 2048 * 2
    assert lib.ditto_workspace_bytes(C.byref(xl), 2, 64, 64) > lib.ditto_workspace_bytes(C.byref(plain), 2, 64, 64)
    bad = hip.Config(90, 1, 2, 64, 90, 10, 0)                # hidden_dim itself must stay a multiple of 64 (the GEMMs' K tile)
    assert lib.ditto_arena_bytes(C.byref(bad)) == 0 and b"hidden_dim" in lib.ditto_last_error()
    bad = hip.Config(192, 1, 64, 64, 192, 10, 0)             # head_dim 3: odd (no half-split rotation)
    assert lib.ditto_arena_bytes(C.byref(bad)) == 0 and b"even" in lib.ditto_last_error()
    bad = hip.Config(384, 1, 4, 64, 384, 10, hip.CFG_FP8_LINEAR)   # fp8 linears: no padded heads (head_dim 96)
    assert lib.ditto_arena_bytes(C.byref(bad)) == 0 and b"fp8" in lib.ditto_last_error()
    for kc, name in enumerate(hip.KERNEL_CLASSES):
        assert lib.ditto_kernel_class_name(kc).decode() == name
    assert lib.ditto_attention_workspace_bytes(1, 1, 64, 64, 64) == 0
    assert lib.ditto_attention_workspace_bytes(1, 1, 64, 64, 768) > 0


def test_null_arguments_are_refused_not_crashed():
    lib = hip.lib()
    assert lib.ditto_model_create(None, None, None, 0, None, None) == hip.ERR_ARG
    assert lib.ditto_forward(None, None, None, None, 1, 1, 1, None, None, None, None, 0, None) == hip.ERR_ARG
    assert lib.ditto_p_sample_update(None, None, None, None, None, None, None, 1, 4, None) == hip.ERR_ARG
    with pytest.raises(hip.DittoHipError):
        hip.check(lib.ditto_gemm_bf16(None, 0, None, None, None, None, 0, 1, 1, 1, 0, None))


def test_config_rejects_what_the_reference_cannot_run():
    with pytest.raises(ValueError):
        DiTTOConfig(768, 12, 12, 256, 512, 50)
    with pytest.raises(ValueError):
        DiTTOConfig(770, 12, 12, 256, 770, 50)


def test_full_row_plan_is_judged_per_launch_and_pinnable():
    """the full-row kernel addresses its A operand with 32-bit byte offsets, so each of a block's two fused
    launches must be admitted on the stride IT reads with (fc2: lda = 4 d), and the class must be pinnable from the
    unsplit batch so that shards of one batch agree (ditto_full_row_plan: host arithmetic, no GPU call)."""
    cfg = PRESETS["C2"]["cfg"]
    try:
        assert hip.full_row_plan(cfg, 32, 1024) == (True, True)      # the headline shape: 256 row tiles
        assert hip.full_row_plan(cfg, 17, 1024) == (True, True)      # 136 tiles of 128 rows: the 128-row kernel's threshold
        assert hip.full_row_plan(cfg, 15, 1024) == (True, True)      # below it the 64-row kernel, from 176 of its tiles on ...
        assert hip.full_row_plan(cfg, 11, 1024) == (True, True)
        assert hip.full_row_plan(cfg, 10, 1024) == (False, False)
        assert hip.full_row_plan(cfg, 16, 1024) == (False, False)    # ... except where the tiled GEMMs make one exact round:
                                                                     # a shard of the 32 batch on its own...
        with hip.batch_class(32 * 1024):                             # ...and pinned to the class of the unsplit batch
            assert hip.full_row_plan(cfg, 16, 1024) == (True, True)
            assert hip.full_row_plan(cfg, 1, 1024) == (True, True)
            assert hip.full_row_plan(cfg, 1, 64) == (True, True)     # one 64-row tile: the 64-row kernel (same bits)
            with pytest.raises(hip.DittoHipError, match="pinned"):   # fewer rows than that cannot take the pinned class:
                hip.full_row_plan(cfg, 1, 48)                        # an error (as from the forward), never another class
        assert hip.full_row_plan(cfg, 1, 48) == (False, False)       # unpinned: the tiled GEMMs
        assert hip.full_row_plan(cfg, 16, 1024) == (False, False)    # the pin is gone
        # M * lda * 2 >= 2^32 with lda = 4 d = 3072: M >= 699051 rows.  The out-projection (lda = d) still fits.
        assert hip.full_row_plan(cfg, 682, 1024) == (True, True)     # 698368 rows
        assert hip.full_row_plan(cfg, 683, 1024) == (True, False)    # 699392 rows: fc2's A offsets would wrap
        assert hip.full_row_plan(cfg, 2731, 1024) == (False, False)  # M * d * 2 >= 2^32 as well
        hip.set_option("fr_mask", 1)
        assert hip.full_row_plan(cfg, 32, 1024) == (True, False)
        hip.set_option("fr_mask", 2)
        assert hip.full_row_plan(cfg, 32, 1024) == (False, True)
        hip.set_option("fr_mask", 0)
        assert hip.full_row_plan(cfg, 32, 1024) == (False, False)
    finally:
        hip.set_option("fr_mask", 3)
        hip.set_option("fr_class_rows", 0)
    # d = 1024 (BASELINE config C5) runs the 64-row kernel (csrc/gemm_fr64.hip), judged on 64-row tiles (>= 192 of them).  Its
    # cross out-projection is bf16 in the fp8 configuration too (LayerNorm output written as fp8); fc2 only in the bf16 one.
    c5 = PRESETS["C5"]["cfg"]
    assert hip.full_row_plan(c5, 16, 1024) == (True, False)
    assert hip.full_row_plan(c5, 8, 1024) == (False, False)
    c5b = PRESETS["C5_bf16"]["cfg"]
    # round 5: at 16 x 1024 rows the tiled fc2 makes ONE whole round of 256 x 256 tiles (kernels.h gemm256_whole_rounds) and, with
    # a LayerNorm launch, beats the 64-row full-row fc2 (which re-streams all of W per 64 rows); other row counts keep the fusion
    assert hip.full_row_plan(c5b, 16, 1024) == (True, False)
    assert hip.full_row_plan(c5b, 14, 1024) == (True, True)
    with hip.batch_class(14 * 1024):
        assert hip.full_row_plan(c5b, 16, 1024) == (True, True)        # (decided on the CLASS rows)
    assert hip.full_row_plan(c5b, 11, 1024) == (False, False)
    with pytest.raises(hip.DittoHipError):
        hip.set_option("fr_class_rows", -1)


def test_attention_kernel_thresholds_are_validated_options():
    """Round 6: the grids from which the 64-queries-per-wave attention kernels run (attn64p / attn64q) are two process defaults,
    one for the residual form (self-attention) and one for the plain form (cross-attention); both are read back, both refuse
    values below 1.  Host arithmetic only: no GPU call."""
    a, b = hip.get_option("attn64p_min_wgs"), hip.get_option("attn64p_min_wgs_plain")
    assert (a, b) == (768, 192) or "DITTO_ATTN64P_MIN_WGS" in "".join(__import__("os").environ)
    try:
        hip.set_option("attn64p_min_wgs", 512)
        hip.set_option("attn64p_min_wgs_plain", 96)
        assert hip.get_option("attn64p_min_wgs") == 512 and hip.get_option("attn64p_min_wgs_plain") == 96
        for name in ("attn64p_min_wgs", "attn64p_min_wgs_plain"):
            with pytest.raises(hip.DittoHipError):
                hip.set_option(name, 0)
    finally:
        hip.set_option("attn64p_min_wgs", a)
        hip.set_option("attn64p_min_wgs_plain", b)


def test_call_options_are_per_call_and_per_thread():
    """ABI 9: the switches that decide which bits an utterance gets — kernel class pin, residual-stream
    type, fused launches — are a ditto_call_opts ARGUMENT or a scope of the CALLING THREAD (ditto_call_opts_push / _pop), not
    process state: a second thread never sees the first one's pin, nesting restores, fields at -1 inherit, bad values and an
    unbalanced pop are errors.  Host arithmetic only (ditto_full_row_plan_opts / ditto_call_opts_current): no GPU call."""
    import threading
    cfg = PRESETS["C2"]["cfg"]
    base = hip.current_opts()
    assert (base.class_rows, base.residual_bf16, base.fr_mask, base.lnq) == (0, 1, 3, 32)
    # per-call argument: the same question, two answers, no state in between
    assert hip.full_row_plan(cfg, 16, 1024) == (False, False)
    assert hip.full_row_plan(cfg, 16, 1024, hip.CallOpts(class_rows=32 * 1024)) == (True, True)
    assert hip.full_row_plan(cfg, 32, 1024, hip.CallOpts(fr_mask=1)) == (True, False)
    assert hip.full_row_plan(cfg, 16, 1024) == (False, False)
    assert hip.stream_is_bf16(cfg, 32, 1024) and not hip.stream_is_bf16(cfg, 32, 1024, hip.CallOpts(residual_bf16=0))
    assert not hip.stream_is_bf16(cfg, 16, 1024) and hip.stream_is_bf16(cfg, 16, 1024, hip.CallOpts(class_rows=32 * 1024))
    assert not hip.stream_is_bf16(cfg, 12, 1024)      # the 64-row full-row kernel's batches keep the fp32 stream
    # thread scope: nests, inherits, restores; another thread is untouched while this one is inside its scope
    seen = {}
    inside, done = threading.Event(), threading.Event()

    def other():
        inside.wait(10)
        seen["other"] = (hip.current_opts().class_rows, hip.full_row_plan(cfg, 16, 1024))
        with hip.batch_class(777):
            seen["other_in"] = hip.current_opts().class_rows
        done.set()

    th = threading.Thread(target=other)
    th.start()
    with hip.call_opts(class_rows=32 * 1024, residual_bf16=0):
        inside.set()
        assert done.wait(10)
        o = hip.current_opts()
        assert (o.class_rows, o.residual_bf16, o.fr_mask) == (32 * 1024, 0, 3)
        with hip.call_opts(fr_mask=2):                    # class_rows and residual_bf16 inherit the enclosing scope
            o = hip.current_opts()
            assert (o.class_rows, o.residual_bf16, o.fr_mask) == (32 * 1024, 0, 2)
            assert hip.full_row_plan(cfg, 16, 1024) == (False, True)
        assert hip.current_opts().fr_mask == 3
        assert hip.get_option("fr_class_rows") == 0 and hip.get_option("residual_bf16") == 1   # the process defaults never moved
    th.join()
    assert seen == {"other": (0, (False, False)), "other_in": 777}
    assert hip.current_opts().class_rows == 0
    # validation
    import ctypes as C
    lib = hip.lib()
    bad = hip.CallOpts(); bad.lnq = 7
    assert lib.ditto_call_opts_push(C.byref(bad)) == hip.ERR_ARG
    bad = hip.CallOpts(); bad.reserved[2] = 1
    assert lib.ditto_call_opts_push(C.byref(bad)) == hip.ERR_ARG
    bad = hip.CallOpts(); bad.class_rows = -5
    a, b = C.c_int(), C.c_int()
    c = hip.make_config(cfg)
    assert lib.ditto_full_row_plan_opts(C.byref(c), 1, 64, C.byref(bad), C.byref(a), C.byref(b), None) == hip.ERR_ARG
    assert lib.ditto_call_opts_pop() == hip.ERR_ARG and b"without a matching push" in lib.ditto_last_error()
    for _ in range(16):
        assert lib.ditto_call_opts_push(None) == hip.OK
    assert lib.ditto_call_opts_push(None) == hip.ERR_ARG          # depth limit
    for _ in range(16):
        assert lib.ditto_call_opts_pop() == hip.OK
