"""-m gpu: the speech-length predictor's decoder stack (reference src/model/SpeechLP.py) through
the C-ABI — its three new building blocks against plain fp32 references, then the whole stack against the oracle
(oracle.slp_decode, pinned by G7) and against the G7 fixtures captured from the reference's own SLP.forward.

Tolerances: bf16 operands / fp32 accumulate, a post-norm stack renormalises every sub-layer, so the decoded
sequence is held to rel-L2 <= 2e-2 (north_star's bf16 bound) and in practice sits near 5e-3; logits likewise."""
import ctypes as C
import math

import pytest
import torch

from ditto_tts_amd import hip
from ditto_tts_amd.slp import SLP
from ditto_tts_amd.synth import hash_normal, slp_state_shapes, synthetic_slp_state_dict
from gpu_util import asym, bf16, max_abs, rel_l2, stream
from oracle import ditto_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 2e-2


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return hip.lib()


# ------------------------------------------------------------------ building blocks
@pytest.mark.parametrize("tile", [0, 128, 127, 256, 129, 192])
@pytest.mark.parametrize("M,N,K", [(300, 320, 64), (513, 5888, 1472), (64, 16, 128), (2100, 1472, 192)])
def test_gemm_relu_epilogue(lib, tile, M, N, K):
    """epilogue 6 = relu(A W^T + b) -> bf16 (linear1 + activation of nn.TransformerDecoderLayer).  Structures without
    this epilogue (129, 192) must fall back to one that has it, not fail or drop the ReLU."""
    A = bf16(asym((M, K), 4).to(DEV))
    W = bf16((asym((N, K), 5) / math.sqrt(K)).to(DEV))
    bias = (0.1 * asym((N,), 6)).to(DEV)
    want = torch.relu(A.float() @ W.float().T + bias)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    hip.check(lib.ditto_set_option(b"gemm_tile", tile))
    try:
        hip.check(lib.ditto_gemm_bf16(A.data_ptr(), K, W.data_ptr(), bias.data_ptr(), None, out.data_ptr(), N, M, N, K,
                                      6, stream()))
    finally:
        hip.check(lib.ditto_set_option(b"gemm_tile", 0))
    assert float(out.float().min()) >= 0.0
    assert float((out == 0).float().mean()) > 0.3          # about half the pre-activations are negative
    assert rel_l2(out.float(), want) < 4e-3


@pytest.mark.parametrize("M,d", [(7, 128), (130, 1472), (33, 2048), (5, 64), (1025, 192)])
def test_layernorm_dual(lib, M, d):
    x = (asym((M, d), 1) * 1.7 + 0.3).to(DEV)
    g = (1 + 0.1 * asym((d,), 2)).to(DEV)
    b = (0.1 * asym((d,), 3)).to(DEV)
    want = torch.nn.functional.layer_norm(x, (d,), g, b, 1e-5)
    yf = torch.empty_like(x)
    yb = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
    hip.check(lib.ditto_layernorm_dual(x.data_ptr(), g.data_ptr(), b.data_ptr(), yf.data_ptr(), yb.data_ptr(), M, d,
                                       stream()))
    assert rel_l2(yf, want) < 1e-6 and max_abs(yf, want) < 1e-5
    assert torch.equal(yb, yf.to(torch.bfloat16))           # the bf16 copy is the rounding of the fp32 one
    # in place on the fp32 stream, bf16 output omitted; and no affine
    x2 = x.clone()
    hip.check(lib.ditto_layernorm_dual(x2.data_ptr(), g.data_ptr(), b.data_ptr(), x2.data_ptr(), None, M, d, stream()))
    assert torch.equal(x2, yf)
    hip.check(lib.ditto_layernorm_dual(x.data_ptr(), None, None, None, yb.data_ptr(), M, d, stream()))
    assert rel_l2(yb.float(), torch.nn.functional.layer_norm(x, (d,))) < 4e-3
    assert lib.ditto_layernorm_dual(x.data_ptr(), None, None, None, None, M, d, stream()) == hip.ERR_ARG


def _attn_ref(q, k, v, B, H, Sq, Skv, dh, scale, causal):
    qh = q.float().view(B, Sq, H, dh).transpose(1, 2)
    kh = k.float().view(B, Skv, H, dh).transpose(1, 2)
    vh = v.float().view(B, Skv, H, dh).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * scale
    if causal:
        i = torch.arange(Sq, device=q.device)[:, None]
        j = torch.arange(Skv, device=q.device)[None, :]
        s = s.masked_fill(j > i + (Skv - Sq), float("-inf"))
    return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B * Sq, H * dh)


@pytest.mark.parametrize("B,H,Sq,Skv,dh", [(2, 4, 40, 40, 64), (1, 2, 97, 97, 128), (2, 4, 130, 130, 384),
                                           (1, 1, 64, 64, 1472 + 64), (3, 2, 33, 70, 64), (1, 1, 1, 1, 64)])
def test_causal_attention(lib, B, H, Sq, Skv, dh):
    """The masked attention of the decoder stack: triu(diagonal=1) tgt_mask (src/model/SpeechLP.py:57-61); for
    Sq < Skv the mask is aligned to the END of the keys (query i sees keys <= i + Skv - Sq)."""
    D = H * dh
    q = bf16(asym((B * Sq, D), 11).to(DEV))
    k = bf16(asym((B * Skv, D), 12).to(DEV))
    v = bf16(asym((B * Skv, D), 13).to(DEV))
    scale = 1.0 / math.sqrt(dh)
    out = torch.empty(B * Sq, D, dtype=torch.bfloat16, device=DEV)
    nb = lib.ditto_attention_causal_workspace_bytes(B, H, Sq, Skv, dh)
    assert nb > 0
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    hip.check(lib.ditto_attention_causal_bf16(q.data_ptr(), D, k.data_ptr(), D, v.data_ptr(), D, out.data_ptr(), D,
                                              B, H, Sq, Skv, dh, scale, ws.data_ptr(), nb, stream()))
    want = _attn_ref(q, k, v, B, H, Sq, Skv, dh, scale, True)
    assert rel_l2(out.float(), want) < 8e-3
    # the first query of a square mask sees exactly one key: its output is that key's value row
    if Sq == Skv:
        o0 = out.float().view(B, Sq, D)[:, 0]
        assert max_abs(o0, v.float().view(B, Skv, D)[:, 0]) < 2e-2
    # and the mask matters: the unmasked result differs
    assert rel_l2(_attn_ref(q, k, v, B, H, Sq, Skv, dh, scale, False), want) > 1e-2 or Sq == 1
    assert lib.ditto_attention_causal_bf16(q.data_ptr(), D, k.data_ptr(), D, v.data_ptr(), D, out.data_ptr(), D, B, H,
                                           Sq, Skv, dh, scale, ws.data_ptr(), nb - 1, stream()) == hip.ERR_SIZE


# ------------------------------------------------------------------ the stack
def _slp(d, nhead, nl, ncls, seed):
    m = SLP(ncls, nhead, nl, hidden_size=d)
    sd = synthetic_slp_state_dict(d, nhead, nl, ncls, seed)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval(), sd


SLP_CASES = {"G7_slp_4head": (128, 4, 2, 11, 2, (2, 20), 24), "G7_slp_1head": (192, 1, 1, 11, 3, (2, 9), 16)}


@pytest.mark.parametrize("name", sorted(SLP_CASES))
def test_slp_matches_reference_fixture(name, golden):
    """HIP stack vs the outputs of the reference's own SLP.forward (G7, tests/golden/make_golden.py make_slp)."""
    d, nhead, nl, ncls, B, (ncb, nfr), T = SLP_CASES[name]
    g = golden(name + ".npz")
    m, _ = _slp(d, nhead, nl, ncls, 5)
    z_text = hash_normal((B, T, d), "slp_text", 5).to(DEV)
    z_audio = hash_normal((B, ncb, nfr, d), "slp_audio", 5).view(B, -1, d).to(DEV)
    logits, decoded = m.decode(z_text, z_audio, return_decoded=True)
    assert logits.shape == (B, ncls) and decoded.shape == (B, ncb * nfr, d)
    assert rel_l2(decoded, g["decoded"]) < TOL
    assert rel_l2(logits, g["logits"]) < TOL
    # the reference call form, encoders injected as pass-through modules (what make_slp did to the reference)
    class Text(torch.nn.Module):
        def forward(self, X):
            return X

    class Audio(torch.nn.Module):
        def forward(self, X):
            return X, None
    m.text_encoder, m.audio_encoder = Text(), Audio()
    again = m(z_text, z_audio.view(B, ncb, nfr, d))
    assert torch.equal(again, logits)


@pytest.mark.parametrize("d,nhead,nl,B,S,T", [
    (1472, 4, 4, 2, 96, 32),      # SLP() defaults on byt5-small: heads of 368 (packed to 384), dim_ff 5888
    (1472, 1, 1, 3, 70, 128),     # ConfigSLP: one layer, one head of 1472 (src/utils/Config.py:75-76,82)
    (64, 1, 1, 1, 1, 1),          # a single position: the causal row is one key
    (256, 2, 3, 2, 257, 5),       # ragged sequence, short memory
])
def test_slp_matches_oracle(d, nhead, nl, B, S, T):
    ncls = 11
    m, sd = _slp(d, nhead, nl, ncls, 3)
    z_text, z_audio = hash_normal((B, T, d), "t", 3), hash_normal((B, S, d), "a", 3)
    want_logits, want_dec = O.slp_decode(sd, nl, nhead, z_text, z_audio)
    logits, dec = m.decode(z_text.to(DEV), z_audio.to(DEV), return_decoded=True)
    assert rel_l2(dec, want_dec) < TOL
    assert rel_l2(logits, want_logits) < TOL
    assert int((logits.argmax(-1).cpu() == want_logits.argmax(-1)).sum()) >= B - 1   # the predicted class
    # deterministic, and causal: truncating the audio leaves the earlier positions' outputs unchanged
    logits2, dec2 = m.decode(z_text.to(DEV), z_audio.to(DEV), return_decoded=True)
    assert torch.equal(dec2, dec) and torch.equal(logits2, logits)
    if S > 8:
        _, dec_short = m.decode(z_text.to(DEV), z_audio[:, :S // 2].to(DEV), return_decoded=True)
        assert rel_l2(dec_short, dec[:, :S // 2]) < 2e-3     # same math; tile shapes (and so summation order) differ


def test_slp_surface_and_errors(lib):
    d, nhead, nl, ncls = 128, 4, 2, 11
    m, sd = _slp(d, nhead, nl, ncls, 1)
    keys = [k for k in m.state_dict() if k.startswith(("transformer.", "length_predictor."))]
    assert keys == list(slp_state_shapes(d, nhead, nl, ncls))       # the reference's key set and order
    assert m.hidden_size == d and m.transformer.layers[0].linear1.out_features == d * nhead
    assert torch.equal(SLP.generate_causal_mask(3, "cpu"),
                       torch.tensor([[0, 1, 1], [0, 0, 1], [0, 0, 0]], dtype=torch.bool))
    zt, za = hash_normal((1, 8, d), "t", 1).to(DEV), hash_normal((1, 12, d), "a", 1).to(DEV)
    first = m.decode(zt, za)
    # weights changed in place (optimizer step / load_state_dict) -> repacked lazily
    sd2 = synthetic_slp_state_dict(d, nhead, nl, ncls, 2)
    m.load_state_dict(sd2)
    second = m.decode(zt, za)
    want, _ = O.slp_decode(sd2, nl, nhead, zt.cpu(), za.cpu())
    assert not torch.equal(first, second) and rel_l2(second, want) < TOL
    with pytest.raises(RuntimeError, match="pretrained encoders"):
        m("some text", torch.zeros(1, 24000))
    with pytest.raises(NotImplementedError, match="inference-only"):
        m.train().decode(zt, za)
    m.eval()
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.decode(zt.cpu(), za.cpu())
    with pytest.raises(ValueError):
        m.decode(zt, za[:, :, :64])
    with pytest.raises(ValueError):
        SLP(ncls)                                                    # neither text_encoder nor hidden_size
    bad = hip.SlpConfig(100, 4, 1, 400, 5)                           # not a multiple of 64
    assert lib.ditto_slp_arena_bytes(C.byref(bad)) == 0 and b"multiples of 64" in lib.ditto_last_error()
    ok = hip.SlpConfig(128, 4, 1, 512, 5)
    assert lib.ditto_slp_workspace_bytes(C.byref(ok), 0, 4, 4) == 0
    handle = C.c_void_p()
    assert lib.ditto_slp_create(C.byref(ok), None, None, 0, stream(), C.byref(handle)) == hip.ERR_ARG


def test_speech_generator_builds_slp_from_checkpoint(tmp_path):
    """SpeechGenerator(slp_path=...) (reference src/model/SpeechGenerator.py:54-62): ConfigSLP geometry, checkpoint dict
    with "model_state_dict", encoder entries of the reference checkpoint skipped."""
    from ditto_tts_amd.shipped_config import ConfigSLP
    from ditto_tts_amd.config import DiTTOConfig
    from ditto_tts_amd.modules import DiTTO
    from ditto_tts_amd.sampler import SpeechGenerator
    from ditto_tts_amd.synth import synthetic_state_dict
    d, nh, nl, ncls = ConfigSLP.EMBEDDING_DIM, ConfigSLP.NUM_HEADS, ConfigSLP.NUM_LAYERS, ConfigSLP.NB_CLASSES
    sd = synthetic_slp_state_dict(d, nh, nl, ncls, 8)
    ckpt = dict(sd)
    ckpt["text_encoder.model.shared.weight"] = torch.zeros(4, 4)          # what a reference checkpoint also holds
    ckpt["audio_encoder.embedding_head.weight"] = torch.zeros(4, 4)
    path = tmp_path / "SLP_epoch_20.pth"
    torch.save({"epoch": 20, "model_state_dict": ckpt}, path)
    cfg = DiTTOConfig(128, 1, 2, 64, 128, 6)
    dm = DiTTO(128, 1, 2, 64, 128, 6)
    dm.load_state_dict(synthetic_state_dict(cfg, 3))
    sg = SpeechGenerator(ditto_model=dm, slp_path=str(path), device=DEV)
    assert sg.slp is not None and not sg.slp.training and sg.slp.num_classes == ncls
    zt, za = hash_normal((2, 24, d), "t", 8), hash_normal((2, 50, d), "a", 8)
    want, _ = O.slp_decode(sd, nl, nh, zt, za)
    assert rel_l2(sg.slp.decode(zt.to(DEV), za.to(DEV)), want) < TOL
    torch.save({"model_state_dict": {k: v for k, v in sd.items() if "norm3" not in k}}, path)
    with pytest.raises(KeyError, match="lacks"):
        SpeechGenerator(ditto_model=dm, slp_path=str(path), device=DEV)
