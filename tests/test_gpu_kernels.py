"""Each HIP kernel against a plain fp32 reference of the same op, through the C-ABI (libditto_hip.so).
Tolerances: fp32 outputs of bf16-operand products: rel-L2 <= 2e-3 against an fp32 product of the SAME
bf16-rounded operands (only accumulation order differs); bf16 outputs add one bf16 rounding (2^-9 relative)."""
import math

import pytest
import torch

from ditto_tts_amd import hip
from gpu_util import asym, bf16, max_abs, rel_l2, stream

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return hip.lib()


@pytest.mark.parametrize("M,d,affine", [(7, 256, True), (130, 768, True), (33, 1024, False), (5, 64, True),
                                        (9, 2048, True)])
def test_layernorm(lib, M, d, affine):
    x = (asym((M, d), 1) * 1.7 + 0.3).to(DEV)
    g = (1 + 0.1 * asym((d,), 2)).to(DEV) if affine else None
    b = (0.1 * asym((d,), 3)).to(DEV) if affine else None
    out = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
    hip.check(lib.ditto_layernorm_bf16(x.data_ptr(), g.data_ptr() if affine else None,
                                       b.data_ptr() if affine else None, out.data_ptr(), M, d, stream()))
    want = torch.nn.functional.layer_norm(x, (d,), g, b, 1e-5)
    assert max_abs(out.float(), want) < 3e-2           # bf16 output: |y| < ~6 -> half-ulp 1.6e-2
    assert rel_l2(out.float(), want) < 4e-3


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 192, 128), (1, 64, 64), (300, 2304, 768), (257, 768, 3072),
                                   (64, 16, 64)])
@pytest.mark.parametrize("epi", [0, 1])
def test_gemm(lib, M, N, K, epi):
    A = bf16(asym((M, K), 4).to(DEV))
    W = bf16((asym((N, K), 5) / math.sqrt(K)).to(DEV))
    bias = (0.1 * asym((N,), 6)).to(DEV)
    res = asym((M, N), 7).to(DEV)
    want = A.float() @ W.float().T + bias
    if epi == 0:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        hip.check(lib.ditto_gemm_bf16(A.data_ptr(), K, W.data_ptr(), bias.data_ptr(), None, out.data_ptr(), N, M, N, K,
                                      0, stream()))
        assert rel_l2(out.float(), want) < 4e-3
    else:
        out = res.clone()   # residual aliases out (the in-place residual-stream update of the model path)
        hip.check(lib.ditto_gemm_bf16(A.data_ptr(), K, W.data_ptr(), bias.data_ptr(), out.data_ptr(), out.data_ptr(), N,
                                      M, N, K, 1, stream()))
        assert rel_l2(out, want + res) < 1e-5
        assert max_abs(out, want + res) < 1e-4


@pytest.fixture(params=[(256, 73), (256, 73 + 256), (129, 73), (192, 73), (127, 73), (131, 321)])
def tile256(lib, request):
    """forces one structure: 256x256 eight-phase, 256x256 wide-phase, 256x128 ring, 256x192, 128x128 deep, 128x256 ping-pong"""
    hip.check(lib.ditto_set_option(b"gemm_tile", request.param[0]))
    hip.check(lib.ditto_set_option(b"gemm_flags", request.param[1]))
    yield
    hip.check(lib.ditto_set_option(b"gemm_tile", 0))
    hip.check(lib.ditto_set_option(b"gemm_flags", 321))


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 320, 64), (1000, 768, 768), (513, 2304, 192),
                                   (260, 272, 3072), (1, 16, 64), (2048, 1024, 320), (777, 144, 128)])
@pytest.mark.parametrize("epi", [0, 1])
def test_gemm_256_tile_structure(lib, tile256, M, N, K, epi):
    """The eight-phase 256x256 kernel forced on ragged M / N and odd / even K-tile counts (K = 64, 192, 320)."""
    A = bf16(asym((M, K), 4).to(DEV))
    W = bf16((asym((N, K), 5) / math.sqrt(K)).to(DEV))
    bias = (0.1 * asym((N,), 6)).to(DEV)
    res = asym((M, N), 7).to(DEV)
    want = A.float() @ W.float().T + bias
    for rep in range(3):   # repeat: a DMA/ds_read race would show as run-to-run differences
        if epi == 0:
            out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            hip.check(lib.ditto_gemm_bf16(A.data_ptr(), K, W.data_ptr(), bias.data_ptr(), None, out.data_ptr(), N, M, N,
                                          K, 0, stream()))
            assert rel_l2(out.float(), want) < 4e-3
        else:
            out = res.clone()
            hip.check(lib.ditto_gemm_bf16(A.data_ptr(), K, W.data_ptr(), bias.data_ptr(), out.data_ptr(), out.data_ptr(),
                                          N, M, N, K, 1, stream()))
            assert rel_l2(out, want + res) < 1e-5
            assert max_abs(out, want + res) < 2e-4
        if rep == 0:
            first = out.clone()
        else:
            assert torch.equal(out, first)


@pytest.mark.parametrize("tile", [256, 131, 192, 129, 127])
@pytest.mark.parametrize("flags", [321, 321 + 1024])
def test_every_epilogue_on_every_structure(lib, tile, flags):
    """All four C-ABI epilogues (0 bias->bf16, 1 bias+residual->fp32 in place, 3 gelu*sigmoid gate, 4 bias->fp32) on each
    tile structure, with the specialised straight-line epilogue (production flags) and without it (flag 1024), against
    the 128x128 kernel: interior and ragged tiles.  (Round 2: the fp32 fast path first shipped without the wait state its
    hand-written store needs; only epilogue 4 showed it.)"""
    try:
        for (M, N, K) in [(512, 1536, 768), (300, 768, 256), (1024, 768, 1536)]:
            A = bf16(asym((M, K), 4).to(DEV))
            W = bf16((asym((N, K), 5) / math.sqrt(K)).to(DEV))
            bias = (0.1 * asym((N,), 6)).to(DEV)
            res = asym((M, N), 7).to(DEV)
            for epi in (0, 1, 3, 4):
                def run(t, fl):
                    hip.check(lib.ditto_set_option(b"gemm_tile", t))
                    hip.check(lib.ditto_set_option(b"gemm_flags", fl))
                    ldo = N // 2 if epi == 3 else N
                    out = res.clone() if epi == 1 else torch.zeros(M, ldo, device=DEV,
                                                                   dtype=torch.bfloat16 if epi in (0, 3) else torch.float32)
                    hip.check(lib.ditto_gemm_bf16(A.data_ptr(), K, W.data_ptr(), bias.data_ptr(),
                                                  out.data_ptr() if epi == 1 else None, out.data_ptr(), ldo, M, N, K, epi,
                                                  stream()))
                    return out.float()
                ref, got = run(128, 321), run(tile, flags)
                if tile == 131 and N % 192 == 0 and epi != 3:           # the ping-pong kernel's opt-in 192-wide tiles
                    hip.check(lib.ditto_set_option(b"pp_nb", 3))
                    try:
                        got3 = run(tile, flags)
                    finally:
                        hip.check(lib.ditto_set_option(b"pp_nb", 0))
                    assert max_abs(got3, ref) <= (1e-2 if epi == 0 else 2e-5) * max(1.0, float(ref.abs().max())), (M, N, K, epi)
                tol = 1e-2 if epi in (0, 3) else 2e-5            # bf16 outputs may differ by one rounding; fp32 by summation order
                assert max_abs(got, ref) <= tol * max(1.0, float(ref.abs().max())), (M, N, K, epi)
    finally:
        hip.check(lib.ditto_set_option(b"gemm_tile", 0))
        hip.check(lib.ditto_set_option(b"gemm_flags", 321))


@pytest.mark.parametrize("shape", [(4096, 6144, 768), (5000, 2304, 768), (8192, 768, 3072), (4096, 2304, 192), (8200, 2336, 768)])
def test_flat_k_loop_across_the_tile_switch_is_bitwise(lib, shape):
    """gemm256's flat K loop (default for even K-tile counts; gemm_flags bit 16384 = the round-3 tile switch): the next tile's first
    six half-tiles ride the empty DMA slots of the last K iteration, no prologue between main loop and epilogue.  More tiles than CUs (384 / 180 / 96 / 144 / 330 tiles; the second and
    fourth shapes have ragged last row tiles, the fourth an ODD K-tile count that must fall back, the fifth ragged last row AND
    column tiles: the re-pointed source bases clamp both), every C-ABI epilogue:
    the arithmetic is unchanged, so the outputs must equal the default build's BIT FOR BIT."""
    M, N, K = shape
    A = bf16(asym((M, K), 24).to(DEV))
    W = bf16((asym((N, K), 25) / math.sqrt(K)).to(DEV))
    bias = (0.1 * asym((N,), 26)).to(DEV)
    res = asym((M, N), 27).to(DEV)
    try:
        hip.check(lib.ditto_set_option(b"gemm_tile", 256))
        for epi in (0, 1, 3, 4):
            outs = []
            for fl in (321, 321 + 16384):
                hip.check(lib.ditto_set_option(b"gemm_flags", fl))
                ldo = N // 2 if epi == 3 else N
                out = res.clone() if epi == 1 else torch.zeros(M, ldo, device=DEV,
                                                               dtype=torch.bfloat16 if epi in (0, 3) else torch.float32)
                hip.check(lib.ditto_gemm_bf16(A.data_ptr(), K, W.data_ptr(), bias.data_ptr(),
                                              out.data_ptr() if epi == 1 else None, out.data_ptr(), ldo, M, N, K, epi, stream()))
                torch.cuda.synchronize()
                outs.append(out)
            assert torch.equal(outs[0], outs[1]), (shape, epi, float((outs[0].float() - outs[1].float()).abs().max()))
            if epi == 4:
                want = A.float() @ W.float().T + bias
                assert rel_l2(outs[1], want) < 1e-5
    finally:
        hip.check(lib.ditto_set_option(b"gemm_tile", 0))
        hip.check(lib.ditto_set_option(b"gemm_flags", 321))


def test_gemm_identity_asymmetric_256(lib, tile256):
    K = N = 256
    A = bf16(torch.eye(K, device=DEV))
    W = bf16((torch.arange(N * K, device=DEV).reshape(N, K) % 251).float() - 125)
    out = torch.empty(K, N, dtype=torch.float32, device=DEV)
    hip.check(lib.ditto_gemm_bf16(A.data_ptr(), K, W.data_ptr(), None, None, out.data_ptr(), N, K, N, K, 1, stream()))
    assert torch.equal(out, W.float().T)


def test_gemm_identity_asymmetric(lib):
    """A = I with an asymmetric W catches a transposed C-write (guide §3)."""
    K = N = 128
    A = bf16(torch.eye(K, device=DEV))
    W = bf16((torch.arange(N * K, device=DEV).reshape(N, K) % 251).float() - 125)   # exact in bf16
    out = torch.empty(K, N, dtype=torch.float32, device=DEV)
    hip.check(lib.ditto_gemm_bf16(A.data_ptr(), K, W.data_ptr(), None, None, out.data_ptr(), N, K, N, K, 1, stream()))
    assert torch.equal(out, W.float().T)


def _attn_ref(q, k, v, B, H, Sq, Skv, dh, scale):
    qh = q.float().view(B, Sq, H, dh).permute(0, 2, 1, 3)
    kh = k.float().view(B, Skv, H, dh).permute(0, 2, 1, 3)
    vh = v.float().view(B, Skv, H, dh).permute(0, 2, 1, 3)
    a = torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1)
    return (a @ vh).permute(0, 2, 1, 3).reshape(B * Sq, H * dh)


@pytest.mark.parametrize("B,H,Sq,Skv,dh", [(1, 1, 32, 64, 64), (2, 3, 200, 96, 64), (1, 2, 128, 128, 64),
                                           (2, 2, 300, 200, 64), (1, 3, 1024, 1024, 64),
                                           (1, 4, 64, 1, 64), (2, 12, 256, 320, 64), (1, 1, 40, 50, 128),
                                           (1, 2, 70, 33, 192), (2, 3, 200, 256, 64), (1, 2, 70, 384, 64),
                                           (1, 2, 333, 2048, 64)])
@pytest.mark.parametrize("attn_flags", [0, 1, 3, 16, 16 + 64, 16 + 128, 16 + 32 + 3, 16 + 256, 16 + 512, 16 + 262144])
def test_attention(lib, B, H, Sq, Skv, dh, attn_flags):
    hip.check(lib.ditto_set_option(b"attn_flags", attn_flags))   # 1: K/V tiles by LDS-DMA; 16: pre-scaled q (attn64v3 when Skv % 128 == 0,
                                                                  # and the rule says so, else attn64v2); 256: never attn64v3; 512: attn64v3 wherever Skv % 128 == 0;
                                                                  # 262144: attn64p (round 6: 64 queries per wave) whatever the grid
    d = H * dh
    q = bf16(asym((B * Sq, d), 8).to(DEV))
    if attn_flags & 16 and dh == 64:
        # the model path: q leaves its projection multiplied by scale * log2(e) (folded into the packed weights);
        # 16 = the reduced-VALU kernel on such a q, 16+32 = the older kernels on it.  The reference is computed on
        # the same (rounded) q, un-scaled in fp32.
        qs = bf16(q.float() * (1.4426950408889634 / math.sqrt(dh)))
        q_ref, q = qs.float() / (1.4426950408889634 / math.sqrt(dh)), qs
    else:
        q_ref = q
    k = bf16(asym((B * Skv, d), 9).to(DEV))
    v = bf16(asym((B * Skv, d), 10).to(DEV))
    scale = 1.0 / math.sqrt(dh)
    out = torch.empty(B * Sq, d, dtype=torch.bfloat16, device=DEV)
    nws = lib.ditto_attention_workspace_bytes(B, H, Sq, Skv, dh)
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=DEV)
    hip.check(lib.ditto_attention_bf16(q.data_ptr(), d, k.data_ptr(), d, v.data_ptr(), d, out.data_ptr(), d, B, H, Sq,
                                       Skv, dh, scale, ws.data_ptr(), ws.numel(), stream()))
    hip.check(lib.ditto_set_option(b"attn_flags", 3))
    want = _attn_ref(q_ref, k, v, B, H, Sq, Skv, dh, scale)
    assert rel_l2(out.float(), want) < 1.5e-2     # P is rounded to bf16 before the PV product
    assert max_abs(out.float(), want) < 6e-2


@pytest.mark.parametrize("Mo,No,K", [(128, 128, 64), (256, 256, 96), (768, 768, 1000), (200, 328, 517), (2304, 768, 4096),
                                     (8, 16, 33), (520, 264, 2050)])
@pytest.mark.parametrize("tile", [128, 256])
@pytest.mark.parametrize("splits", [1, 5])
def test_weight_gradient_gemm_tn(lib, Mo, No, K, tile, splits):
    """ditto_gemm_tn_bf16 (csrc/gemm_tn.hip): out = X^T Y with both operands K-major — dW = dY^T X of nn.Linear, what the
    backward of reference src/TrainDiTTO.py:90 computes — on both tile structures (128x128 two workgroups per CU, 256x256
    one per CU), plain and split-K with the ordered reduce; ragged Mo / No / K (zero rows past K, clamped columns);
    repeated for run-to-run determinism.  Against the fp32 product of the bf16 operands."""
    X = bf16(asym((K, Mo), 31).to(DEV))
    Y = bf16(asym((K, No), 32).to(DEV))
    want = X.float().T @ Y.float()
    ws = torch.empty(256 + splits * Mo * No * 4, dtype=torch.uint8, device=DEV)
    first = None
    for rep in range(2):
        out = torch.full((Mo, No), float("nan"), device=DEV)
        hip.check(lib.ditto_gemm_tn_bf16(X.data_ptr(), Mo, Y.data_ptr(), No, out.data_ptr(), No, Mo, No, K, splits, tile,
                                         ws.data_ptr(), ws.numel(), stream()))
        assert rel_l2(out, want) < 1e-5 and max_abs(out, want) < 1e-3 * (1 + float(want.abs().max()))
        if first is None:
            first = out.clone()
        else:
            assert torch.equal(out, first)
    assert lib.ditto_gemm_tn_bf16(X.data_ptr(), Mo, Y.data_ptr(), No, out.data_ptr(), No, Mo, No, K, 1, 64, ws.data_ptr(),
                                  ws.numel(), stream()) == hip.ERR_ARG


@pytest.mark.parametrize("B,H,Sq,Skv", [(1, 2, 128, 128), (2, 3, 200, 256), (1, 12, 1024, 1024), (1, 2, 333, 2048),
                                        (2, 2, 4096, 4096)])
def test_attention_pipelined_kernel_is_bitwise_the_tile_loop_kernel(lib, B, H, Sq, Skv):
    """attn64v3 (software-pipelined, attn_flags 512) and attn64v2 (256) are chosen by shape AND grid size, so an utterance's
    bits may not depend on which one ran: same products in the same order, running-maximum raises by whole octaves in both.
    Rows with forced raises included."""
    dh, d = 64, H * 64
    q = asym((B * Sq, d), 21) * 0.7
    k = asym((B * Skv, d), 22) * 0.7
    v = asym((B * Skv, d), 23)
    k[Skv - 3, :dh] = q[5, :dh] * 40.0      # a late dominating key for one row: forced raise in the last tile
    k[70, :dh] = q[9, :dh] * 30.0           # and one in tile 1
    q = bf16((q * (1.4426950408889634 / math.sqrt(dh))).to(DEV))
    k, v = bf16(k.to(DEV)), bf16(v.to(DEV))
    outs = []
    variants = (16 + 256 + 131072, 16 + 512 + 2048 + 131072, 16 + 512 + 1024 + 131072)
    for flags in variants:   # attn64v2; attn64v3 with 4 and with 8 waves per workgroup (131072: never attn64p, whose row sum is another association)
        hip.check(lib.ditto_set_option(b"attn_flags", flags))
        out = torch.empty(B * Sq, d, dtype=torch.bfloat16, device=DEV)
        hip.check(lib.ditto_attention_bf16(q.data_ptr(), d, k.data_ptr(), d, v.data_ptr(), d, out.data_ptr(), d, B, H, Sq,
                                           Skv, dh, 1.0 / math.sqrt(dh), None, 0, stream()))
        outs.append(out)
    hip.check(lib.ditto_set_option(b"attn_flags", 3))
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("B,H,Sq,Skv", [(1, 2, 128, 128), (2, 3, 200, 256), (1, 12, 1024, 1024), (1, 2, 333, 2048),
                                        (2, 2, 1000, 1000), (1, 1, 40, 1), (3, 12, 512, 96), (8, 12, 1024, 1024),
                                        (2, 3, 300, 65), (1, 2, 256, 127), (2, 2, 100, 129), (1, 3, 512, 191), (1, 4, 700, 1023)])
def test_attention_64_queries_per_wave_kernel(lib, B, H, Sq, Skv):
    """attn64p and attn64q (round 6, csrc/attn64p.h / attn64q.h: 64 queries per wave, fp32 row sums, the wide epilogue; attn64q = one
    software-pipelined stream per wave with the optimistic softmax, from two key tiles on, the last one possibly partial) against attn64v2 on the same operands
    — same products, other associations of the row sum only: within bf16-output noise — against torch, and against themselves
    (bitwise repeatable).  Ragged query and key counts, rows with forced raises of the running maximum (for attn64q: logits of
    ~ 30 .. 60 in log2 units, still inside its range).  (The residual epilogues — the self-attention's in-place update on the fp32
    and on the bf16 stream — run in the model tests: tests/test_gpu_model.py taps `after_self` of G1 and the full-size checks.)"""
    dh, d = 64, H * 64
    q = asym((B * Sq, d), 51) * 0.7
    k = asym((B * Skv, d), 52) * 0.7
    v = asym((B * Skv, d), 53)
    if Skv > 70:
        k[Skv - 3, :dh] = q[5, :dh] * 40.0      # a late dominating key for one row: forced raise in the last tile
        k[70, :dh] = q[9, :dh] * 30.0           # and one in tile 1
    scale = 1.0 / math.sqrt(dh)
    qs = bf16((q * (1.4426950408889634 * scale)).to(DEV))
    q_ref = qs.float() / (1.4426950408889634 * scale)
    k, v = bf16(k.to(DEV)), bf16(v.to(DEV))
    outs = []
    try:
        # attn64v2; the round-6 kernels as dispatched (attn64q from 65 keys on — a partial last tile included —, else attn64p), twice; attn64p always
        for flags in (16 + 256 + 131072, 16 + 262144, 16 + 262144, 16 + 262144 + 1048576):
            hip.check(lib.ditto_set_option(b"attn_flags", flags))
            out = torch.full((B * Sq, d), float("nan"), dtype=torch.bfloat16, device=DEV)
            hip.check(lib.ditto_attention_bf16(qs.data_ptr(), d, k.data_ptr(), d, v.data_ptr(), d, out.data_ptr(), d, B, H, Sq,
                                               Skv, dh, scale, None, 0, stream()))
            outs.append(out)
    finally:
        hip.check(lib.ditto_set_option(b"attn_flags", 3))
    want = _attn_ref(q_ref, k, v, B, H, Sq, Skv, dh, scale)
    for o in outs:
        assert torch.isfinite(o.float()).all()
        assert rel_l2(o.float(), want) < 1.5e-2 and max_abs(o.float(), want) < 6e-2
    assert torch.equal(outs[1], outs[2]), "not repeatable"
    assert rel_l2(outs[1].float(), outs[0].float()) < 4e-3 and rel_l2(outs[3].float(), outs[0].float()) < 4e-3
    if Skv <= 64:
        assert torch.equal(outs[1], outs[3]), "a single key tile runs attn64p"


@pytest.mark.parametrize("Skv", [512, 471])
@pytest.mark.parametrize("case", ["overflow", "underflow", "nan", "mixed"])
def test_attention_optimistic_softmax_leaves_its_range(lib, case, Skv):
    """attn64q keeps no running maximum: P = exp2(S) unshifted, and a workgroup whose row sums left [2^-100, 2^100] (or are NaN) starts
    over on attn64p's exact loop before it stores anything.  Logits far above the range (every probability overflows), far below
    (every probability underflows: l = 0), NaN inputs, and a batch in which only some workgroups leave the range: the result is
    attn64p's, bit for bit where the workgroup fell back, and within rounding of it elsewhere."""
    B, H, Sq, dh = 2, 4, 512, 64       # Skv = 471: the partial last tile's copy of the loop (masked score chains) on both paths
    d = H * dh
    q = asym((B * Sq, d), 61) * 0.7
    k = asym((B * Skv, d), 62) * 0.7
    v = asym((B * Skv, d), 63)
    scale = 1.0 / math.sqrt(dh)
    qs = q * (1.4426950408889634 * scale)
    u = torch.full((dh,), 0.125)                           # |u|^2 = 1
    fell_back = torch.zeros(B, H, Sq // 256, dtype=torch.bool)     # workgroups (256 queries of one head) that must take the exact path
    if case == "overflow":
        qs = qs * 60.0                                      # row maxima of some hundreds of octaves
        fell_back[:] = True
    elif case == "underflow":
        k[:, :] = (u.repeat(H) + 0.02 * k)                  # every key ~ u, every query ~ -150 u: logits ~ -150 (log2 units), spread ~ 1
        qs = -150.0 * u.repeat(H) + 0.3 * qs
        fell_back[:] = True
    elif case == "nan":
        qs[300, dh:2 * dh] = float("nan")                   # one query row of head 1, batch 0: its workgroup only
        fell_back[0, 1, 1] = True
    else:
        qs[Sq + 256:Sq + 512, 2 * dh:3 * dh] *= 60.0        # batch 1, head 2, queries 256.. : one workgroup overflows
        qs[10, :dh] = -150.0 * u                            # batch 0, head 0, one row underflows against keys ~ u
        k[:Skv, :dh] = u + 0.02 * k[:Skv, :dh]
        fell_back[1, 2, 1] = True
        fell_back[0, 0, 0] = True
    qs, k, v = bf16(qs.to(DEV)), bf16(k.to(DEV)), bf16(v.to(DEV))
    outs = []
    try:
        for flags in (16 + 262144, 16 + 262144 + 1048576):   # attn64q (this shape), attn64p
            hip.check(lib.ditto_set_option(b"attn_flags", flags))
            out = torch.full((B * Sq, d), float("nan"), dtype=torch.bfloat16, device=DEV)
            hip.check(lib.ditto_attention_bf16(qs.data_ptr(), d, k.data_ptr(), d, v.data_ptr(), d, out.data_ptr(), d, B, H, Sq,
                                               Skv, dh, scale, None, 0, stream()))
            outs.append(out.float().view(B, Sq // 256, 256, H, dh).permute(0, 3, 1, 2, 4))   # [B, H, workgroup, 256, dh]
    finally:
        hip.check(lib.ditto_set_option(b"attn_flags", 3))
    got, exact = outs
    fb = fell_back.to(DEV)
    if case != "nan":
        assert torch.isfinite(exact).all() and torch.isfinite(got).all()
        want = _attn_ref(qs.float() / (1.4426950408889634 * scale), k, v, B, H, Sq, Skv, dh, scale)
        want = want.view(B, Sq // 256, 256, H, dh).permute(0, 3, 1, 2, 4)
        assert rel_l2(got, want) < 1.5e-2
    assert torch.equal(got[fb].nan_to_num(nan=7.0), exact[fb].nan_to_num(nan=7.0)), "a workgroup out of range is not the exact path's"
    ok = ~fb
    assert torch.isfinite(got[ok]).all()
    assert rel_l2(got[ok], exact[ok]) < 4e-3
    if case in ("nan", "mixed"):
        assert not torch.equal(got[ok], exact[ok]), "expected the optimistic path on the workgroups inside the range"


@pytest.mark.parametrize("attn_flags", [3, 16, 16 + 64, 16 + 128, 16 + 256, 16 + 512, 16 + 262144])
def test_attention_forced_rescale(lib, attn_flags):
    """Rule 26: force the online-softmax rescale branch — one key in the LAST tile dominates one query row."""
    hip.check(lib.ditto_set_option(b"attn_flags", attn_flags))
    B, H, Sq, Skv, dh = 1, 1, 64, 256, 64
    q = asym((Sq, dh), 11) * 0.5
    k = asym((Skv, dh), 12) * 0.5
    v = asym((Skv, dh), 13)
    k[Skv - 3] = q[5] * 40.0          # score ~ 40*|q5|^2*0.125 >> all others, arrives in tile 3
    k[70] = q[9] * 30.0               # and one in tile 1 for another row
    q, k, v = bf16(q.to(DEV)), bf16(k.to(DEV)), bf16(v.to(DEV))
    out = torch.empty(Sq, dh, dtype=torch.bfloat16, device=DEV)
    qk = bf16(q.float() * (1.4426950408889634 * 0.125)) if attn_flags & 16 else q
    if attn_flags & 16:
        q = qk.float() / (1.4426950408889634 * 0.125)           # what the kernel effectively sees, for the reference
    hip.check(hip.lib().ditto_attention_bf16(qk.data_ptr(), dh, k.data_ptr(), dh, v.data_ptr(), dh, out.data_ptr(), dh,
                                             B, H, Sq, Skv, dh, 0.125, None, 0, stream()))
    hip.check(lib.ditto_set_option(b"attn_flags", 3))
    want = _attn_ref(q, k, v, B, H, Sq, Skv, dh, 0.125)
    assert max_abs(out.float(), want) < 6e-2
    assert max_abs(out.float()[5], v.float()[Skv - 3]) < 6e-2   # row 5 is (almost) exactly that value row


def test_p_sample_update_and_q_sample(lib):
    from oracle import ditto_oracle as O
    B, n = 3, 16 * 768
    betas, alphas, acp = O.sampler_tables(50)
    x, eps, z = asym((B, 16, 768), 14), asym((B, 16, 768), 15), asym((B, 16, 768), 16)
    t = torch.tensor([49, 0, 17])
    want = O.p_sample_update(x, eps, t, betas, alphas, acp, z)
    # keep every device tensor referenced until the launch has been checked (a freed temporary's memory is
    # handed to the next allocation)
    xd, epsd, zd, td = x.to(DEV).clone(), eps.to(DEV), z.to(DEV), t.to(DEV)
    bd, ad, cd = betas.to(DEV), alphas.to(DEV), acp.to(DEV)
    hip.check(lib.ditto_p_sample_update(xd.data_ptr(), epsd.data_ptr(), zd.data_ptr(), td.data_ptr(), bd.data_ptr(),
                                        ad.data_ptr(), cd.data_ptr(), B, n, stream()))
    assert rel_l2(xd, want) < 1e-6
    assert max_abs(xd[1], want[1]) < 1e-6   # t == 0: no noise term
    buf = O.cosine_beta_schedule(1000)
    tq = torch.tensor([0, 500, 999])
    wantq = O.q_sample(buf, x, tq, z)
    out = torch.empty_like(xd)
    x0d, tqd, bufd = x.to(DEV), tq.to(DEV), buf.to(DEV)
    hip.check(lib.ditto_q_sample(x0d.data_ptr(), zd.data_ptr(), tqd.data_ptr(), bufd.data_ptr(), out.data_ptr(), B, n,
                                 stream()))
    assert rel_l2(out, wantq) < 1e-6
    torch.cuda.synchronize()


def test_component_modules_adaln_and_rope(golden):
    """GlobalAdaLN / RotaryEmbedding facade modules == golden G1 (reference outputs)."""
    from ditto_tts_amd.modules import GlobalAdaLN, RotaryEmbedding
    from ditto_tts_amd.config import DiTTOConfig
    from ditto_tts_amd.synth import synthetic_state_dict
    g = golden("G1_block_c1.npz")
    sd = synthetic_state_dict(DiTTOConfig(256, 1, 4, 256, 256, 50), seed=1)
    ada = GlobalAdaLN(256, 256, 256)
    ada.load_state_dict({k[len("ada_ln."):]: v for k, v in sd.items() if k.startswith("ada_ln.")})
    ada = ada.to(DEV)
    with torch.no_grad():
        h0 = ada(g["x"].to(DEV), g["temb"].to(DEV), g["text"].to(DEV))
    assert rel_l2(h0, g["after_adaln"]) < 1e-5          # all-fp32 kernel
    rot = RotaryEmbedding(64).to(DEV)
    pos = rot(64, DEV)
    assert rel_l2(pos, g["rotary_pos"]) < 1e-6
    from oracle import ditto_oracle as O
    tq = asym((2, 64, 4, 64), 17)
    want = O.apply_rope(g["rotary_pos"], tq)
    got = rot.apply_rope(pos, tq.to(DEV))
    assert rel_l2(got, want) < 1e-5


# ------------------------------------------------------------------------------------------------------------
# fp8 (OCP e4m3) building blocks of BASELINE config 5.  torch.float8_e4m3fn is the same format, used only to
# DEQUANTISE what the library produced.
# ------------------------------------------------------------------------------------------------------------
def _deq(t_u8, scales=None):
    f = t_u8.view(torch.float8_e4m3fn).float()
    return f if scales is None else f * scales[:, None]


def _quant(lib, w):
    rows, cols = w.shape
    q = torch.empty(rows, cols, dtype=torch.uint8, device=DEV)
    sc = torch.empty(rows, dtype=torch.float32, device=DEV)
    hip.check(lib.ditto_quantize_rows_fp8(w.data_ptr(), rows, cols, q.data_ptr(), sc.data_ptr(), stream()))
    return q, sc


def test_fp8_quantize_rows(lib):
    w = (asym((200, 384), 21) * torch.linspace(0.01, 30, 200)[:, None]).to(DEV)
    q, sc = _quant(lib, w)
    assert torch.allclose(sc, w.abs().amax(dim=1) / 448.0, rtol=1e-6)
    back = _deq(q, sc)
    assert rel_l2(back, w) < 4e-2          # e4m3: 3 mantissa bits -> ~2^-4 / sqrt(3) rms relative error
    assert float((back.abs().amax(dim=1) - w.abs().amax(dim=1)).abs().max()) < 1e-3 * float(w.abs().max())


def test_fp8_layernorm(lib):
    M, d = 70, 1024
    x = (asym((M, d), 22) * 1.7 + 0.3).to(DEV)
    g, b = (1 + 0.1 * asym((d,), 23)).to(DEV), (0.1 * asym((d,), 24)).to(DEV)
    out = torch.empty(M, d, dtype=torch.uint8, device=DEV)
    hip.check(lib.ditto_layernorm_fp8(x.data_ptr(), g.data_ptr(), b.data_ptr(), out.data_ptr(), M, d, stream()))
    want = torch.nn.functional.layer_norm(x, (d,), g, b, 1e-5)
    assert torch.equal(_deq(out), want.to(torch.float8_e4m3fn).float())   # same RNE rounding as torch's cast


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 320, 384), (1000, 768, 1024), (64, 16, 128),
                                   (513, 2304, 768)])
@pytest.mark.parametrize("epi", [0, 1, 4])
def test_fp8_gemm(lib, M, N, K, epi):
    """exact in the operands: the reference multiplies the DEQUANTISED fp8 values in fp32."""
    a32 = asym((M, K), 25).to(DEV)
    w32 = (asym((N, K), 26) / math.sqrt(K)).to(DEV)
    Aq = a32.to(torch.float8_e4m3fn).view(torch.uint8).contiguous()        # activations: scale 1
    Wq, ws = _quant(lib, w32)
    bias = (0.1 * asym((N,), 27)).to(DEV)
    res = asym((M, N), 28).to(DEV)
    want = _deq(Aq) @ _deq(Wq, ws).T + bias
    if epi == 0:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        hip.check(lib.ditto_gemm_fp8(Aq.data_ptr(), K, Wq.data_ptr(), ws.data_ptr(), bias.data_ptr(), None,
                                     out.data_ptr(), N, M, N, K, 0, stream()))
        assert rel_l2(out.float(), want) < 4e-3
    elif epi == 1:
        out = res.clone()
        hip.check(lib.ditto_gemm_fp8(Aq.data_ptr(), K, Wq.data_ptr(), ws.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                     out.data_ptr(), N, M, N, K, 1, stream()))
        assert rel_l2(out, want + res) < 5e-5
    else:
        out = torch.empty(M, N, dtype=torch.float32, device=DEV)
        hip.check(lib.ditto_gemm_fp8(Aq.data_ptr(), K, Wq.data_ptr(), ws.data_ptr(), bias.data_ptr(), None,
                                     out.data_ptr(), N, M, N, K, 4, stream()))
        assert rel_l2(out, want) < 5e-5   # the reference scales W before the product, the kernel scales acc after
        assert max_abs(out, want) < 1e-3


@pytest.mark.parametrize("shape", [(8192, 2304, 1024), (9000, 3072, 768)])
def test_fp8_flat_k_loop_across_the_tile_switch_is_bitwise(lib, shape):
    """The fp8 instantiations of gemm256 take the flat K loop as well (even K-tile counts of 128 bytes).  More tiles than CUs
    (288 / 432, the second with a ragged last row tile): gemm_flags bit 16384 (the round-3 tile switch) gives the same bits, and
    the fp32-output epilogue agrees with the dequantised fp32 product."""
    M, N, K = shape
    a32 = asym((M, K), 41).to(DEV)
    w32 = (asym((N, K), 42) / math.sqrt(K)).to(DEV)
    Aq = a32.to(torch.float8_e4m3fn).view(torch.uint8).contiguous()
    Wq, ws = _quant(lib, w32)
    bias = (0.1 * asym((N,), 43)).to(DEV)
    res = asym((M, N), 44).to(DEV)
    try:
        for epi in (0, 1, 4):
            outs = []
            for fl in (321, 321 + 16384):
                hip.check(lib.ditto_set_option(b"gemm_flags", fl))
                out = res.clone() if epi == 1 else torch.zeros(M, N, device=DEV, dtype=torch.bfloat16 if epi == 0 else torch.float32)
                hip.check(lib.ditto_gemm_fp8(Aq.data_ptr(), K, Wq.data_ptr(), ws.data_ptr(), bias.data_ptr(),
                                             out.data_ptr() if epi == 1 else None, out.data_ptr(), N, M, N, K, epi, stream()))
                torch.cuda.synchronize()
                outs.append(out)
            assert torch.equal(outs[0], outs[1]), (shape, epi, float((outs[0].float() - outs[1].float()).abs().max()))
            if epi == 4:
                assert rel_l2(outs[0], _deq(Aq) @ _deq(Wq, ws).T + bias) < 5e-5
    finally:
        hip.check(lib.ditto_set_option(b"gemm_flags", 321))


def test_fp8_gemm_gated(lib):
    """fp8 gated-MLP epilogue: interleaved fc1|gate rows, fp8 output."""
    M, d = 300, 256
    a32 = asym((M, d), 29).to(DEV)
    w1, wg = (asym((4 * d, d), 30) / math.sqrt(d)).to(DEV), (asym((4 * d, d), 31) / math.sqrt(d)).to(DEV)
    b1, bg = (0.1 * asym((4 * d,), 32)).to(DEV), (0.1 * asym((4 * d,), 33)).to(DEV)
    idx = torch.arange(4 * d, device=DEV)
    pos1, posg = (idx // 16) * 32 + idx % 16, (idx // 16) * 32 + idx % 16 + 16   # packed row of fc1 row i / gate row i
    wp = torch.empty(8 * d, d, device=DEV); wp[pos1], wp[posg] = w1, wg
    bp = torch.empty(8 * d, device=DEV); bp[pos1], bp[posg] = b1, bg
    Aq = a32.to(torch.float8_e4m3fn).view(torch.uint8).contiguous()
    Wq, ws = _quant(lib, wp)
    out = torch.empty(M, 4 * d, dtype=torch.uint8, device=DEV)
    hip.check(lib.ditto_gemm_fp8(Aq.data_ptr(), d, Wq.data_ptr(), ws.data_ptr(), bp.data_ptr(), None, out.data_ptr(),
                                 4 * d, M, 8 * d, d, 5, stream()))
    wdq = _deq(Wq, ws)
    h = _deq(Aq) @ wdq[pos1].T + b1
    g = _deq(Aq) @ wdq[posg].T + bg
    want = torch.nn.functional.gelu(h) * torch.sigmoid(g)
    assert rel_l2(_deq(out), want) < 4e-2      # one e4m3 rounding of the result
    assert max_abs(_deq(out), want.clamp(-448, 448)) < 0.07 * (1 + float(want.abs().max()))


@pytest.mark.parametrize("M,K,ln", [(128, 768, True), (300, 256, True), (1024, 3072, True), (257, 64, False), (1000, 128, True),
                                      (515, 192, True), (2048, 768, True)])
@pytest.mark.parametrize("rot", [0, 8, 3])
def test_full_row_gemm_with_fused_layernorm(lib, M, K, ln, rot):
    """ditto_gemm_ln_bf16 (csrc/gemm_fr.hip): out = residual + A W^T + bias in place, u = LayerNorm(out) * gamma + beta, in ONE
    kernel whose workgroups own whole rows — against the fp32 ops.  The model's cross out-projection + norm3 and fc2 + next
    norm1 (fr_mask).  Ragged M (row clamp / masks), one to 48 K slabs, with and without the K-loop rotation (period in
    tiles; 2048 rows = 16 tiles takes the XCD tile remap), repeated for run-to-run determinism."""
    N = 768
    hip.check(lib.ditto_set_option(b"fr_rot", rot))
    A = bf16(asym((M, K), 4).to(DEV))
    W = bf16((asym((N, K), 5) / math.sqrt(K)).to(DEV))
    Wp = W.view(N, K // 16, 16).permute(1, 0, 2).contiguous()      # the kernel's stage-major weight layout [K/16][N][16]
    bias = (0.1 * asym((N,), 6)).to(DEV)
    res = asym((M, N), 7).to(DEV)
    g = (1 + 0.1 * asym((N,), 8)).to(DEV)
    b = (0.1 * asym((N,), 9)).to(DEV)
    want = res + A.float() @ W.float().T + bias
    wu = torch.nn.functional.layer_norm(want, (N,), g, b, 1e-5)
    first = None
    for rep in range(3):
        h = res.clone()
        u = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
        hip.check(lib.ditto_gemm_ln_bf16(A.data_ptr(), K, Wp.data_ptr(), bias.data_ptr(), h.data_ptr(), h.data_ptr(), N,
                                         g.data_ptr() if ln else None, b.data_ptr() if ln else None,
                                         u.data_ptr() if ln else None, N, M, N, K, stream()))
        assert rel_l2(h, want) < 1e-5 and max_abs(h, want) < 2e-4
        if ln:
            assert max_abs(u.float(), wu) < 4e-2 and rel_l2(u.float(), wu) < 4e-3      # bf16 output
        if first is None:
            first = (h.clone(), u.clone())
        else:
            assert torch.equal(h, first[0]) and torch.equal(u, first[1])
    hip.check(lib.ditto_set_option(b"fr_rot", 1))
    assert lib.ditto_gemm_ln_bf16(A.data_ptr(), K, Wp.data_ptr(), None, None, h.data_ptr(), 512, None, None, None, 0, M, 512,
                                  K, stream()) == hip.ERR_SHAPE


@pytest.mark.parametrize("M,K,ln,res", [(128, 768, True, True), (300, 256, True, True), (1024, 3072, True, True),
                                          (257, 64, False, True), (1000, 128, True, False), (515, 192, True, True),
                                          (4096, 768, True, True), (2048, 6144, False, True)])
@pytest.mark.parametrize("rot", [0, 8, 3])
def test_full_row_gemm_with_weights_straight_into_registers(lib, M, K, ln, res, rot):
    """csrc/gemm_frd.hip (fr_tile 130: 128 x 768 tiles, a wave owns 128 rows x 192 columns and fetches ITS weight fragments
    with global_load_dwordx4 two stages ahead into a register ring — no W in the LDS) against its 64-row twin csrc/gemm_fr64.hip
    (rounds 2-3 also had an LDS-ring form, fr_tile 128, deleted in round 6): h must agree
    BIT FOR BIT (same K order incl. rotation, same accumulator init); u = LayerNorm(h) sums its statistics per quarter row,
    so it may differ from theirs in the last bf16 bit of a few elements — checked against the fp32 op and counted.
    Ragged M, one to 96 K slabs, with / without residual and LayerNorm, run-to-run determinism."""
    N = 768
    hip.check(lib.ditto_set_option(b"fr_rot", rot))
    A = bf16(asym((M, K), 44).to(DEV))
    W = bf16((asym((N, K), 45) / math.sqrt(K)).to(DEV))
    Wp = W.view(N, K // 16, 16).permute(1, 0, 2).contiguous()
    bias = (0.1 * asym((N,), 46)).to(DEV)
    r0 = asym((M, N), 47).to(DEV)
    g = (1 + 0.1 * asym((N,), 48)).to(DEV)
    b = (0.1 * asym((N,), 49)).to(DEV)
    want = A.float() @ W.float().T + bias + (r0 if res else 0)
    wu = torch.nn.functional.layer_norm(want, (N,), g, b, 1e-5)
    outs = {}
    try:
        for tile in (64, 130, 130):
            hip.check(lib.ditto_set_option(b"fr_tile", tile))
            h = r0.clone() if res else torch.full((M, N), 7.0, device=DEV)
            u = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
            hip.check(lib.ditto_gemm_ln_bf16(A.data_ptr(), K, Wp.data_ptr(), bias.data_ptr(), h.data_ptr() if res else None,
                                             h.data_ptr(), N, g.data_ptr() if ln else None, b.data_ptr() if ln else None,
                                             u.data_ptr() if ln else None, N, M, N, K, stream()))
            torch.cuda.synchronize()
            if tile in outs:
                assert torch.equal(h, outs[tile][0]) and torch.equal(u, outs[tile][1])      # run-to-run
            outs[tile] = (h, u)
    finally:
        hip.check(lib.ditto_set_option(b"fr_tile", 0))
        hip.check(lib.ditto_set_option(b"fr_rot", 1))
    (h64, u64), (hd, ud) = outs[64], outs[130]
    assert rel_l2(hd, want) < 1e-5 and max_abs(hd, want) < 3e-4
    assert torch.equal(hd, h64), float((hd - h64).abs().max())
    if ln:
        assert max_abs(ud.float(), wu) < 4e-2 and rel_l2(ud.float(), wu) < 4e-3
        assert float((ud != u64).float().mean()) < 1e-3           # a bf16 rounding tie here and there at most


def test_full_row_kernels_on_seeded_random_shapes(lib):
    """Twenty seeded (M, K, rotation, bias / residual / LayerNorm) draws — M anywhere in 128 .. 5000, K any multiple of 64 up to
    6144 (the training dgrad's depth) — through the N = 768 full-row kernels (fr_tile 130 / 64) and,
    at N = 1024, the 64-row kernel: against the fp32 ops, h of the N = 768 kernels bitwise equal, u of 128 and 64 bitwise equal."""
    import random
    rnd = random.Random(20261002)
    try:
        for case in range(20):
            N = 768 if case % 4 else 1024
            M = rnd.randint(128, 5000)
            K = 64 * rnd.randint(1, 96 if case % 5 else 24)
            rot = rnd.choice([0, 0, 3, 8])
            ln, res, has_bias = rnd.random() < 0.7, rnd.random() < 0.8, rnd.random() < 0.8
            hip.check(lib.ditto_set_option(b"fr_rot", rot))
            A = bf16(asym((M, K), 100 + case).to(DEV))
            W = bf16((asym((N, K), 200 + case) / math.sqrt(K)).to(DEV))
            Wp = W.view(N, K // 16, 16).permute(1, 0, 2).contiguous()
            bias = (0.1 * asym((N,), 300 + case)).to(DEV)
            r0 = asym((M, N), 400 + case).to(DEV)
            g = (1 + 0.1 * asym((N,), 500 + case)).to(DEV)
            b = (0.1 * asym((N,), 600 + case)).to(DEV)
            want = A.float() @ W.float().T + (bias if has_bias else 0) + (r0 if res else 0)
            wu = torch.nn.functional.layer_norm(want, (N,), g, b, 1e-5)
            outs = {}
            for tile in ((130, 64) if N == 768 else (0,)):
                hip.check(lib.ditto_set_option(b"fr_tile", tile))
                h = r0.clone() if res else torch.full((M, N), -3.0, device=DEV)
                u = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
                hip.check(lib.ditto_gemm_ln_bf16(A.data_ptr(), K, Wp.data_ptr(), bias.data_ptr() if has_bias else None,
                                                 h.data_ptr() if res else None, h.data_ptr(), N, g.data_ptr() if ln else None,
                                                 b.data_ptr() if ln else None, u.data_ptr() if ln else None, N, M, N, K, stream()))
                torch.cuda.synchronize()
                tag = (case, N, M, K, rot, ln, res, has_bias, tile)
                assert rel_l2(h, want) < 1e-5 and max_abs(h, want) < 5e-4, tag
                if ln:
                    assert max_abs(u.float(), wu) < 4e-2 and rel_l2(u.float(), wu) < 4e-3, tag
                outs[tile] = (h, u)
            if N == 768:
                assert torch.equal(outs[130][0], outs[64][0]), (case, M, K)
                if 128 in outs:
                    assert torch.equal(outs[64][0], outs[128][0]) and torch.equal(outs[64][1], outs[128][1]), (case, M, K)
    finally:
        hip.check(lib.ditto_set_option(b"fr_tile", 0))
        hip.check(lib.ditto_set_option(b"fr_rot", 1))


@pytest.mark.parametrize("M,K,ln,res", [(64, 1024, True, True), (300, 256, True, True), (1024, 4096, True, True), (257, 64, False, True),
                                          (1000, 128, True, False), (515, 192, True, True), (4096, 1024, True, True),
                                          (2048, 4096, False, True)])
@pytest.mark.parametrize("rot", [0, 8, 3])
def test_full_row_gemm_at_width_1024(lib, M, K, ln, res, rot):
    """csrc/gemm_fr64.hip at N = 1024 (BASELINE config C5: 64 x 1024 tiles, one workgroup per CU, W ring of four stages, all
    accumulators in AGPRs): out = residual + A W^T + bias in place and u = LayerNorm(out) against the fp32 ops; ragged M,
    one to 64 K slabs, K-loop rotation, with / without residual and LayerNorm, run-to-run determinism; the LayerNorm output
    also as fp8 e4m3 (the fp8 linear path's next operand) against the bf16 one."""
    N = 1024
    hip.check(lib.ditto_set_option(b"fr_rot", rot))
    A = bf16(asym((M, K), 34).to(DEV))
    W = bf16((asym((N, K), 35) / math.sqrt(K)).to(DEV))
    Wp = W.view(N, K // 16, 16).permute(1, 0, 2).contiguous()
    bias = (0.1 * asym((N,), 36)).to(DEV)
    r0 = asym((M, N), 37).to(DEV)
    g = (1 + 0.1 * asym((N,), 38)).to(DEV)
    b = (0.1 * asym((N,), 39)).to(DEV)
    want = A.float() @ W.float().T + bias + (r0 if res else 0)
    wu = torch.nn.functional.layer_norm(want, (N,), g, b, 1e-5)
    first = None
    try:
        for rep in range(3):
            h = r0.clone() if res else torch.full((M, N), 7.0, device=DEV)
            u = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
            hip.check(lib.ditto_gemm_ln_bf16(A.data_ptr(), K, Wp.data_ptr(), bias.data_ptr(), h.data_ptr() if res else None,
                                             h.data_ptr(), N, g.data_ptr() if ln else None, b.data_ptr() if ln else None,
                                             u.data_ptr() if ln else None, N, M, N, K, stream()))
            torch.cuda.synchronize()
            assert rel_l2(h, want) < 1e-5 and max_abs(h, want) < 3e-4
            if ln:
                assert max_abs(u.float(), wu) < 4e-2 and rel_l2(u.float(), wu) < 4e-3      # bf16 output
            if first is None:
                first = (h.clone(), u.clone())
            else:
                assert torch.equal(h, first[0]) and torch.equal(u, first[1])
        if ln:
            hip.check(lib.ditto_set_option(b"fr_u_fp8", 1))
            h8 = r0.clone() if res else torch.full((M, N), 7.0, device=DEV)
            u8 = torch.zeros(M, N, dtype=torch.uint8, device=DEV)
            hip.check(lib.ditto_gemm_ln_bf16(A.data_ptr(), K, Wp.data_ptr(), bias.data_ptr(), h8.data_ptr() if res else None,
                                             h8.data_ptr(), N, g.data_ptr(), b.data_ptr(), u8.data_ptr(), N, M, N, K, stream()))
            torch.cuda.synchronize()
            assert torch.equal(h8, first[0])
            got8 = u8.view(torch.float8_e4m3fn).float()
            ref8 = wu.clamp(-448, 448).to(torch.float8_e4m3fn).float()       # one e4m3 rounding of the fp32 LayerNorm
            # the kernel rounds ITS fp32 y (1e-6 from the reference's): a different e4m3 neighbour only at ties
            assert float((got8 != ref8).float().mean()) < 2e-3
            assert max_abs(got8, wu.clamp(-448, 448)) < 0.07 * (1 + float(wu.abs().max()))
    finally:
        hip.check(lib.ditto_set_option(b"fr_u_fp8", 0))
        hip.check(lib.ditto_set_option(b"fr_rot", 1))


@pytest.mark.parametrize("M,K,ln,res", [(128, 768, True, True), (300, 256, True, True), (1024, 3072, True, True),
                                          (257, 64, False, True), (1000, 128, True, False), (515, 192, True, True),
                                          (4096, 768, True, True), (2048, 3072, False, True)])
@pytest.mark.parametrize("rot", [0, 8, 3])
def test_full_row_gemm_on_64_row_tiles_is_bitwise_the_128_row_kernel(lib, M, K, ln, res, rot):
    """csrc/gemm_fr64.hip (64 x 768 tiles, two workgroups per CU, wave-private W ring) against the 128-row kernels on the same
    inputs (csrc/gemm_frd.hip: h bit for bit, u within a bf16 rounding tie): h and u must agree BIT FOR BIT (same K order incl. the rotation of the 128-row tile the rows belong to, same
    accumulator init, same association of the LayerNorm statistics) — the choice between the two is a speed rule, not a
    numerics class.  Ragged M, with / without residual, bias, LayerNorm, start delay of the second workgroup on and off."""
    N = 768
    hip.check(lib.ditto_set_option(b"fr_rot", rot))
    A = bf16(asym((M, K), 14).to(DEV))
    W = bf16((asym((N, K), 15) / math.sqrt(K)).to(DEV))
    Wp = W.view(N, K // 16, 16).permute(1, 0, 2).contiguous()
    bias = (0.1 * asym((N,), 16)).to(DEV)
    r0 = asym((M, N), 17).to(DEV)
    g = (1 + 0.1 * asym((N,), 18)).to(DEV)
    b = (0.1 * asym((N,), 19)).to(DEV)
    want = A.float() @ W.float().T + bias + (r0 if res else 0)
    outs = {}
    try:
        for tile, stagger in ((64, 0), (64, 700)) + (((130, 0),) if M >= 128 else ()):
            hip.check(lib.ditto_set_option(b"fr_tile", tile))
            hip.check(lib.ditto_set_option(b"fr_stagger", stagger))
            h = r0.clone() if res else torch.full((M, N), 7.0, device=DEV)
            u = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
            hip.check(lib.ditto_gemm_ln_bf16(A.data_ptr(), K, Wp.data_ptr(), bias.data_ptr(), h.data_ptr() if res else None,
                                             h.data_ptr(), N, g.data_ptr() if ln else None, b.data_ptr() if ln else None,
                                             u.data_ptr() if ln else None, N, M, N, K, stream()))
            torch.cuda.synchronize()
            outs[(tile, stagger)] = (h, u)
    finally:
        hip.check(lib.ditto_set_option(b"fr_tile", 0))
        hip.check(lib.ditto_set_option(b"fr_stagger", 0))
        hip.check(lib.ditto_set_option(b"fr_rot", 1))
    href, uref = outs[(64, 0)]
    assert rel_l2(href, want) < 1e-5 and max_abs(href, want) < 2e-4
    for key in [k for k in outs if k != (64, 0)]:
        h, u = outs[key]
        assert torch.equal(h, href), (key, float((h - href).abs().max()))
        if key[0] == 130:
            assert float((u != uref).float().mean()) < 1e-3, key           # statistics summed per quarter row: a rounding tie at most
        else:
            assert torch.equal(u, uref), (key, float((u.float() - uref.float()).abs().max()))


# ---------------------------------------------------------------------------------------------------------------
# csrc/gemm_lnq.hip: LayerNorm fused INTO the GEMM that consumes it (the model's norm2 + cross-attention q-projection)
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(params=[(0, 1), (-1, 5), (0, 16)])
def lnq_variant(lib, request):
    """(ring, rot): W register ring depth 0 = default (4 stages at shape 32, 2 at shape 16), -1 = the deep one (8 / 4);
    fr_rot > 1 = the unit entry rotates its K loop with that period in 64-row tiles (the model path rotates by tiles per
    utterance)."""
    yield request.param
    hip.set_option("lnq_ring", 0)
    hip.set_option("fr_rot", 1)


@pytest.mark.parametrize("shape", [32, 16])
@pytest.mark.parametrize("h_bf16", [False, True])
@pytest.mark.parametrize("M", [64, 1, 63, 65, 200, 1024, 2048 + 37])
def test_layernorm_fused_into_the_q_projection(lib, lnq_variant, M, h_bf16, shape):
    """ditto_gemm_lnq_bf16: out = (LayerNorm(h) * gamma + beta) W^T + bias with the normalised rows in the LDS only, both MFMA
    shapes, fp32 and bf16 rows, ragged M (a partial last 64-row tile, a single row).  Against (a) the fp32 ops on the SAME
    bf16-rounded normalised rows — the kernel's LayerNorm is the LayerNorm kernel's arithmetic statement for statement, so
    that rounding is reproduced exactly — tolerance 4e-3 (bf16 output), and (b) the two-launch path it replaces
    (ditto_layernorm_bf16 + ditto_gemm_bf16): same inputs to the MFMAs, so the difference is summation order + one
    output rounding (rel-L2 < 3e-3); rows do not depend on what else is in the launch (bitwise, against a 1-row launch)."""
    d = 768
    ring, rot = lnq_variant
    hip.set_option("lnq_ring", 0 if ring == 0 else (8 if shape == 32 else 4))
    hip.set_option("fr_rot", rot)
    h = (asym((M, d), 31) * 1.7 + 0.4).to(DEV)
    if h_bf16:
        hin = bf16(h)
        href = hin.float()
    else:
        hin, href = h, h
    gamma = (1 + 0.2 * asym((d,), 32)).to(DEV)
    beta = (0.1 * asym((d,), 33)).to(DEV)
    W = bf16((asym((d, d), 34) / math.sqrt(d)).to(DEV))
    bias = (0.1 * asym((d,), 35)).to(DEV)
    scratch = torch.empty(d * d * 2, dtype=torch.uint8, device=DEV)
    out = torch.full((M, d), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.check(lib.ditto_gemm_lnq_bf16(hin.data_ptr(), d, int(h_bf16), gamma.data_ptr(), beta.data_ptr(), W.data_ptr(),
                                      bias.data_ptr(), out.data_ptr(), d, M, d, shape, scratch.data_ptr(), stream()))
    assert torch.isfinite(out.float()).all()
    # (b) the two launches it replaces (fp32 rows: ditto_layernorm_bf16 takes fp32)
    u = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
    hip.check(lib.ditto_layernorm_bf16(href.contiguous().data_ptr(), gamma.data_ptr(), beta.data_ptr(), u.data_ptr(), M, d, stream()))
    two = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
    hip.check(lib.ditto_gemm_bf16(u.data_ptr(), d, W.data_ptr(), bias.data_ptr(), None, two.data_ptr(), d, M, d, d, 0, stream()))
    # (a) fp32 ops on the LayerNorm kernel's own bf16 output
    want = u.float() @ W.float().T + bias
    assert rel_l2(out.float(), want) < 4e-3, rel_l2(out.float(), want)
    assert rel_l2(out.float(), two.float()) < 3e-3
    # no bias; and one row alone gives that row's bits
    out0 = torch.empty_like(out)
    hip.check(lib.ditto_gemm_lnq_bf16(hin.data_ptr(), d, int(h_bf16), gamma.data_ptr(), beta.data_ptr(), W.data_ptr(),
                                      None, out0.data_ptr(), d, M, d, shape, scratch.data_ptr(), stream()))
    assert rel_l2(out0.float(), want - bias) < 4e-3
    r = M // 2
    one = torch.empty(1, d, dtype=torch.bfloat16, device=DEV)
    hip.check(lib.ditto_gemm_lnq_bf16(hin[r:r + 1].contiguous().data_ptr(), d, int(h_bf16), gamma.data_ptr(), beta.data_ptr(),
                                      W.data_ptr(), bias.data_ptr(), one.data_ptr(), d, 1, d, shape, scratch.data_ptr(), stream()))
    if rot <= 1 or (r // 64) % rot % 6 == 0:            # (a rotated K loop sums in another order: same phase only)
        assert torch.equal(one[0], out[r])
    else:
        assert rel_l2(one[0].float(), out[r].float()) < 4e-3


def test_fused_q_projection_normalises_exactly_like_the_layernorm_kernel(lib):
    """With W = identity the fused kernel's output IS bf16(its normalised rows) (products by 1 and 0 are exact, the fp32 sum
    of one non-zero term is exact): bit-identical to ditto_layernorm_bf16 for both MFMA shapes."""
    d, M = 768, 320
    h = (asym((M, d), 41) * 2.3 - 0.7).to(DEV)
    gamma = (1 + 0.3 * asym((d,), 42)).to(DEV)
    beta = (0.2 * asym((d,), 43)).to(DEV)
    W = bf16(torch.eye(d, device=DEV))
    scratch = torch.empty(d * d * 2, dtype=torch.uint8, device=DEV)
    u = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
    hip.check(lib.ditto_layernorm_bf16(h.data_ptr(), gamma.data_ptr(), beta.data_ptr(), u.data_ptr(), M, d, stream()))
    for shape in (32, 16):
        out = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
        hip.check(lib.ditto_gemm_lnq_bf16(h.data_ptr(), d, 0, gamma.data_ptr(), beta.data_ptr(), W.data_ptr(), None,
                                          out.data_ptr(), d, M, d, shape, scratch.data_ptr(), stream()))
        assert torch.equal(out, u), shape


@pytest.mark.parametrize("h_bf16", [False, True])
@pytest.mark.parametrize("M,rot", [(64, 0), (1, 0), (63, 5), (200, 0), (1024, 16), (2048 + 37, 5)])
def test_fused_q_projection_on_two_waves_per_simd_is_bitwise_the_four_wave_kernel(lib, M, rot, h_bf16):
    """ditto_set_option("lnq_waves", 8): the fused norm2 + q-projection with EIGHT waves per workgroup (two per SIMD; each wave
    owns 96 of the 768 columns and normalises 8 of the 64 rows) against the four-wave kernel: no weight byte is fetched twice,
    every output element is the same K-ordered MFMA chain and every row is normalised by the same one-wave-per-row arithmetic —
    so the outputs must agree BIT FOR BIT, ragged M and rotated K loops included."""
    d = 768
    h = (asym((M, d), 51) * 1.9 - 0.3).to(DEV)
    hin = bf16(h) if h_bf16 else h
    gamma = (1 + 0.2 * asym((d,), 52)).to(DEV)
    beta = (0.1 * asym((d,), 53)).to(DEV)
    W = bf16((asym((d, d), 54) / math.sqrt(d)).to(DEV))
    bias = (0.1 * asym((d,), 55)).to(DEV)
    scratch = torch.empty(d * d * 2, dtype=torch.uint8, device=DEV)
    outs = {}
    try:
        hip.set_option("fr_rot", rot if rot else 1)
        for waves in (4, 8, 8):
            hip.set_option("lnq_waves", waves)
            out = torch.full((M, d), float("nan"), dtype=torch.bfloat16, device=DEV)
            hip.check(lib.ditto_gemm_lnq_bf16(hin.data_ptr(), d, int(h_bf16), gamma.data_ptr(), beta.data_ptr(), W.data_ptr(),
                                              bias.data_ptr(), out.data_ptr(), d, M, d, 32, scratch.data_ptr(), stream()))
            torch.cuda.synchronize()
            if waves in outs:
                assert torch.equal(out, outs[waves])                    # run to run
            outs[waves] = out
    finally:
        hip.set_option("lnq_waves", 8)
        hip.set_option("fr_rot", 1)
    assert torch.isfinite(outs[4].float()).all()
    assert torch.equal(outs[8], outs[4]), float((outs[8].float() - outs[4].float()).abs().max())


@pytest.mark.parametrize("M", [64, 100, 1024 + 13])
def test_layernorm_fused_into_the_q_projection_at_width_1024(lib, M):
    """The same kernel at d = 1024 (BASELINE config C5: 64 x 1024 tiles, 256 accumulators per lane, 32x32x16, fp32 rows):
    against the fp32 ops on the LayerNorm kernel's own bf16 output (4e-3) and the two launches it replaces (3e-3); with an
    identity weight its normalised rows are the LayerNorm kernel's bit for bit."""
    d = 1024
    h = (asym((M, d), 51) * 1.3 - 0.2).to(DEV)
    gamma = (1 + 0.2 * asym((d,), 52)).to(DEV)
    beta = (0.1 * asym((d,), 53)).to(DEV)
    W = bf16((asym((d, d), 54) / math.sqrt(d)).to(DEV))
    bias = (0.1 * asym((d,), 55)).to(DEV)
    scratch = torch.empty(d * d * 2, dtype=torch.uint8, device=DEV)
    out = torch.full((M, d), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.check(lib.ditto_gemm_lnq_bf16(h.data_ptr(), d, 0, gamma.data_ptr(), beta.data_ptr(), W.data_ptr(), bias.data_ptr(),
                                      out.data_ptr(), d, M, d, 32, scratch.data_ptr(), stream()))
    u = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
    hip.check(lib.ditto_layernorm_bf16(h.data_ptr(), gamma.data_ptr(), beta.data_ptr(), u.data_ptr(), M, d, stream()))
    two = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
    hip.check(lib.ditto_gemm_bf16(u.data_ptr(), d, W.data_ptr(), bias.data_ptr(), None, two.data_ptr(), d, M, d, d, 0, stream()))
    want = u.float() @ W.float().T + bias
    assert torch.isfinite(out.float()).all()
    assert rel_l2(out.float(), want) < 4e-3 and rel_l2(out.float(), two.float()) < 3e-3
    eye = bf16(torch.eye(d, device=DEV))
    hip.check(lib.ditto_gemm_lnq_bf16(h.data_ptr(), d, 0, gamma.data_ptr(), beta.data_ptr(), eye.data_ptr(), None,
                                      out.data_ptr(), d, M, d, 32, scratch.data_ptr(), stream()))
    assert torch.equal(out, u)
    assert lib.ditto_gemm_lnq_bf16(h.data_ptr(), d, 0, gamma.data_ptr(), beta.data_ptr(), W.data_ptr(), None, out.data_ptr(), d, M,
                                   d, 16, scratch.data_ptr(), stream()) == hip.ERR_SHAPE      # 16x16x32 exists at d = 768 only
    # two waves per SIMD (the default since round 5) against one: the same bits
    res = {}
    try:
        for waves in (8, 4):
            hip.set_option("lnq_waves", waves)
            o = torch.empty_like(out)
            hip.check(lib.ditto_gemm_lnq_bf16(h.data_ptr(), d, 0, gamma.data_ptr(), beta.data_ptr(), W.data_ptr(), bias.data_ptr(),
                                              o.data_ptr(), d, M, d, 32, scratch.data_ptr(), stream()))
            res[waves] = o
    finally:
        hip.set_option("lnq_waves", 8)
    assert torch.equal(res[8], res[4])
