"""Oracle (oracle/ditto_oracle.py) == golden vectors captured from the reference's own code
(tests/golden/make_golden.py).  Runs everywhere, including the GPU box (no /root/reference needed).
Tolerance: fp32 restatement vs reference rel-L2 <= 1e-5, max-abs <= 1e-4 at unit scale."""
import pytest
import torch

from conftest import max_abs, rel_l2
from ditto_tts_amd.config import DiTTOConfig
from ditto_tts_amd.synth import hash_normal, synthetic_inputs, synthetic_state_dict
from oracle import ditto_oracle as O

RTOL = 1e-5


def test_g1_block_segments_and_adaln(golden):
    g = golden("G1_block_c1.npz")
    cfg = DiTTOConfig(256, 1, 4, 256, 256, 50)
    sd = synthetic_state_dict(cfg, seed=1)
    x, text, t = synthetic_inputs(cfg, 1, 64, 64, seed=11)
    assert torch.equal(x, g["x"]) and torch.equal(text, g["text"]) and torch.equal(t, g["t"])
    temb = O.time_embedding(sd, t)
    assert rel_l2(temb, g["temb"]) < RTOL
    pos = O.rotary_table(sd["rotary.inv_freq"], 64)
    assert torch.equal(pos, g["rotary_pos"])
    h0 = O.global_adaln(sd, x, temb, text)
    assert rel_l2(h0, g["after_adaln"]) < RTOL
    taps = {}
    out = O.ditto_forward(sd, 1, 4, x, text, t, taps)
    for k in ("after_self", "after_cross", "after_mlp"):
        assert rel_l2(taps["blocks.0." + k], g[k]) < RTOL, k
        assert max_abs(taps["blocks.0." + k], g[k]) < 1e-4, k
    assert rel_l2(out, g["out"]) < RTOL


def test_g2_full_ditto_s(golden):
    g = golden("G2_ditto_s.npz")
    cfg = DiTTOConfig(768, 12, 12, 256, 768, 50)
    sd = synthetic_state_dict(cfg, seed=2)
    x, text, t = synthetic_inputs(cfg, 2, 128, 96, seed=22)
    assert torch.equal(x.half(), g["x"]) and torch.equal(text.half(), g["text"])
    taps = {}
    out = O.ditto_forward(sd, 12, 12, x, text, t, taps)
    for i in (0, 5, 11):
        assert rel_l2(taps[f"blocks.{i}.after_mlp"], g[f"block{i}"]) < RTOL, i
    assert rel_l2(out, g["out"]) < RTOL
    assert max_abs(out, g["out"]) < 1e-4
    # the fixture is a meaningful unit-scale signal, not a collapsed one
    assert 0.3 < float(g["out"].std()) < 10.0


def test_g3_shipped_one_head(golden):
    g = golden("G3_shipped_1head.npz")
    cfg = DiTTOConfig(768, 5, 1, 256, 768, 1000)
    sd = synthetic_state_dict(cfg, seed=3)
    x, text, t = synthetic_inputs(cfg, 1, 64, 64, seed=33)
    out = O.ditto_forward(sd, 5, 1, x, text, t)
    assert rel_l2(out, g["out"]) < RTOL


def test_g4_schedule_and_qsample(golden):
    g = golden("G4_schedule_qsample.npz")
    assert torch.equal(O.cosine_beta_schedule(50), g["betas50"])
    assert torch.equal(O.cosine_beta_schedule(1000), g["betas1000"])
    # App. B-1: the `alphas_cumprod` buffer holds the clipped betas
    assert torch.equal(g["buffer1000"], g["betas1000"])
    x0, nz = hash_normal((3, 16, 768), "x0", 44), hash_normal((3, 16, 768), "qnoise", 44)
    qs = O.q_sample(g["buffer1000"], x0, g["t"], nz)
    assert rel_l2(qs, g["q_sample"]) < 1e-6


def test_g5_sampler_trajectory(golden):
    g = golden("G5_sampler_50.npz")
    cfg = DiTTOConfig(256, 2, 4, 256, 256, 50)
    sd = synthetic_state_dict(cfg, seed=5)
    B, N, T, S = 2, 64, 32, 50
    betas, alphas, ac = O.sampler_tables(S)
    assert torch.equal(betas, g["betas"]) and torch.equal(alphas, g["alphas"])
    assert torch.equal(ac, g["alphas_cumprod"])
    noises = [hash_normal((B, N, 256), f"z{i}", 55) for i in range(S)]
    x, kept = O.sample_latents(sd, 2, 4, g["xinit"], g["text"], S, noises, keep=(0, 1, 10, 49))
    for i in (0, 1, 10, 49):
        assert rel_l2(kept[i], g[f"x_step{i}"]) < 1e-4, i   # 50 chained steps of fp32 round-off


def test_g8_sampling_loop_at_the_timed_width(golden):
    """G8: the 50-step loop at d = 768 / 12 heads of 64 (the width of the timed kernel class), eps from the imported
    reference DiTTO.forward — the restated loop of the oracle reproduces it, and the regenerated inputs are the stored ones."""
    g = golden("G8_loop768.npz")
    cfg = DiTTOConfig(768, 2, 12, 256, 768, 50)
    sd = synthetic_state_dict(cfg, seed=8)
    B, N, T, S = 2, 128, 96, 50
    text, xinit = hash_normal((B, T, 768), "text", 88), hash_normal((B, N, 768), "xT", 88)
    assert torch.equal(text.half(), g["text16"]) and torch.equal(xinit.half(), g["xinit16"])
    noises = [hash_normal((B, N, 768), f"z{i}", 88) for i in range(S)]
    x, kept = O.sample_latents(sd, 2, 12, xinit, text, S, noises, keep=(0, 1, 10, 49))
    for i in (0, 1, 10, 49):
        assert rel_l2(kept[i], g[f"x_step{i}"]) < 1e-4, i


def _vq_inputs():
    cb = hash_normal((1024, 768), "codebook", 66) * 0.05
    lat = hash_normal((2, 2, 96, 768), "vq_latents", 66) * 0.06
    lat[0, 0, :8] = cb[:8] + 1e-3 * hash_normal((8, 768), "jit", 66)
    return cb, lat


def test_g6_vector_quantizer_indices(golden):
    """Integer output: bit-exact against the reference's VectorQuantizer.forward."""
    g = golden("G6_vq.npz")
    cb, lat = _vq_inputs()
    idx = O.vq_indices(cb, lat)
    assert idx.dtype == torch.int64 and idx.shape == (2, 2, 96)
    assert torch.equal(idx, g["indices"].long())


def test_strided_schedule_and_ddim_coefficients():
    """Published-formula anchors (the reference has no strided sampler): stride 1 + eta=1 reproduces the
    reference's posterior noise scale up to the beta-tilde/beta choice, eta=0 is deterministic, and the last step
    lands on x0 = (x - sqrt(1-ab) eps)/sqrt(ab)."""
    betas, alphas, ac = O.sampler_tables(50)
    assert O.strided_timesteps(50, 50) == list(range(49, -1, -1))
    taus = O.strided_timesteps(50, 25)
    assert taus[0] == 49 and len(taus) == 25 and all(a > b for a, b in zip(taus, taus[1:])) and taus[-1] >= 0
    a, ce, cz = O.ddim_coefficients(ac, taus[-1], -1, 0.0)
    ab = float(ac[taus[-1]])
    assert abs(a - ab ** -0.5) < 1e-6 and abs(ce + ((1 - ab) / ab) ** 0.5) < 1e-6 and cz == 0.0
    a, ce, cz = O.ddim_coefficients(ac, 30, 28, 1.0)
    assert cz > 0 and abs(cz ** 2 - (1 - float(ac[28])) / (1 - float(ac[30])) * (1 - float(ac[30] / ac[28]))) < 1e-6


# ------------------------------------------------------------------ G7: speech-length predictor decoder stack
SLP_CASES = {"G7_slp_4head": (128, 4, 2, 11, 2, (2, 20), 24), "G7_slp_1head": (192, 1, 1, 11, 3, (2, 9), 16)}


@pytest.mark.parametrize("name", sorted(SLP_CASES))
def test_g7_slp_decoder_oracle_matches_reference_forward(name, golden):
    """oracle.slp_decode vs the reference's SLP.forward outputs (tests/golden/make_golden.py make_slp)."""
    from ditto_tts_amd.synth import synthetic_slp_state_dict
    d, nhead, nl, ncls, B, (ncb, nfr), T = SLP_CASES[name]
    g = golden(name + ".npz")
    sd = synthetic_slp_state_dict(d, nhead, nl, ncls, 5)
    z_text = hash_normal((B, T, d), "slp_text", 5)
    z_audio = hash_normal((B, ncb, nfr, d), "slp_audio", 5).view(B, -1, d)    # src/model/SpeechLP.py:49
    logits, decoded = O.slp_decode(sd, nl, nhead, z_text, z_audio)
    assert rel_l2(decoded, g["decoded"]) < RTOL
    assert rel_l2(logits, g["logits"]) < RTOL


def test_slp_oracle_matches_torch_transformer_decoder_live():
    """The restated layer math against nn.TransformerDecoder itself (what src/model/SpeechLP.py:22-32 instantiates),
    byt5-small's head geometry in miniature: d = 4 * 92 (head width not a multiple of 64 or 8)."""
    import torch.nn as nn
    from ditto_tts_amd.synth import synthetic_slp_state_dict
    d, nhead, nl, ncls, B, S, T = 368, 4, 2, 7, 2, 19, 11
    sd = synthetic_slp_state_dict(d, nhead, nl, ncls, 9)
    dec = nn.TransformerDecoder(nn.TransformerDecoderLayer(d_model=d, nhead=nhead, dim_feedforward=d * nhead,
                                                           batch_first=True), num_layers=nl).eval()
    dec.load_state_dict({k[len("transformer."):]: v for k, v in sd.items() if k.startswith("transformer.")})
    z_text, z_audio = hash_normal((B, T, d), "t", 9), hash_normal((B, S, d), "a", 9)
    with torch.no_grad():
        want = dec(z_audio, z_text, tgt_mask=torch.triu(torch.ones(S, S), diagonal=1).bool())
    _, decoded = O.slp_decode(sd, nl, nhead, z_text, z_audio)
    assert rel_l2(decoded, want) < RTOL


def test_philox_known_answer_vectors():
    """The counter-based generator behind ditto_noise_normal is Philox4x32-10; oracle/philox.py (its numpy restatement,
    used by the GPU tests as the checker) is pinned by Random123's published known-answer vectors."""
    import numpy as np
    from oracle.philox import noise_normal, philox4x32_10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = tuple(int(v) for v in philox4x32_10(*[np.uint32(x) for x in ctr], *[np.uint32(x) for x in key]))
        assert got == want
    z = noise_normal(0x1234567890ABCDEF, 17, 1 << 18)
    assert abs(z.mean()) < 1e-2 and abs(z.std() - 1.0) < 1e-2 and np.isfinite(z).all()
    assert not np.array_equal(z, noise_normal(0x1234567890ABCDEF, 18, 1 << 18))         # the step is part of the counter
    assert not np.array_equal(z[:64], noise_normal(0x1234567890ABCDEE, 17, 64))          # the seed is the key
