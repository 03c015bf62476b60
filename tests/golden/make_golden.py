#!/usr/bin/env python3
"""Generate the golden fixtures tests/golden/G*.npz by RUNNING THE REFERENCE ITSELF.

Runs only where /root/reference exists (the build container).  It imports the reference's
`components.DiT` and `model.DiTTO`, fills them with the closed-form weights of
`ditto_tts_amd.synth`, runs them in eval mode on CPU fp32 and stores inputs + outputs.
No reference source text is stored: the fixtures are data only.

    python tests/golden/make_golden.py            # rewrites all G*.npz

Constructing `DiTTO` needs a stub for the neural audio codec: `NAC()` downloads GPT-2 /
EnCodec from the HF hub and `torch.load`s a private checkpoint (reference
src/model/DiTTO.py:22-34); neither exists here and neither is on the denoise path.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_SRC = "/root/reference/src"
sys.path.insert(0, ROOT)

from ditto_tts_amd.config import DiTTOConfig  # noqa: E402
from ditto_tts_amd.synth import hash_normal, synthetic_inputs, synthetic_state_dict  # noqa: E402


def import_reference():
    if not os.path.isdir(REF_SRC):
        raise SystemExit("reference not present; goldens can only be regenerated in the build container")
    sys.path.insert(0, REF_SRC)
    import components.DiT as ref_dit  # noqa
    import model.DiTTO as ref_ditto  # noqa
    return ref_dit, ref_ditto


def build_reference_ditto(ref_ditto, cfg: DiTTOConfig, sd):
    class _StubNAC(torch.nn.Module):
        def __init__(self, lambda_factor=0.1):
            super().__init__()
            self.language_model = torch.nn.Identity()
            self.audio_encoder = torch.nn.Identity()

        def load_state_dict(self, *a, **k):  # the private NAC checkpoint is not on the path
            return None

    real_nac, real_load = ref_ditto.NAC, torch.load
    ref_ditto.NAC = _StubNAC
    torch.load = lambda *a, **k: {"model_state_dict": {}}
    try:
        m = ref_ditto.DiTTO(hidden_dim=cfg.hidden_dim, num_layers=cfg.num_layers, num_heads=cfg.num_heads,
                            time_dim=cfg.time_dim, text_dim=cfg.text_dim, diffusion_steps=cfg.diffusion_steps,
                            nac_model_path="unused")
    finally:
        ref_ditto.NAC, torch.load = real_nac, real_load
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("nac.") for k in missing), missing
    # the reference's own key set (minus nac.*) must be exactly ours
    ref_keys = [k for k in m.state_dict().keys() if not k.startswith("nac.")]
    assert sorted(ref_keys) == sorted(sd.keys()), set(ref_keys) ^ set(sd.keys())
    return m.eval()


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1e6:.2f} MB)")


def block_taps(block, x, text, temb, pos):
    """Run one reference DiT block and tap its three segment outputs with forward hooks on
    norm2 / norm3 inputs (they receive the post-self / post-cross residual stream)."""
    taps = {}
    h2 = block.norm2.register_forward_hook(lambda m, i, o: taps.__setitem__("after_self", i[0].detach().clone()))
    h3 = block.norm3.register_forward_hook(lambda m, i, o: taps.__setitem__("after_cross", i[0].detach().clone()))
    with torch.no_grad():
        out = block(x, text, temb, pos)
    h2.remove(); h3.remove()
    taps["after_mlp"] = out
    return taps


@torch.no_grad()
def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_dit, ref_ditto = import_reference()

    # ---- G1 (C1): one block d=256,h=4,N=64,T=64,B=1 + GlobalAdaLN + rotary table ----
    cfg = DiTTOConfig(256, 1, 4, 256, 256, 50)
    sd = synthetic_state_dict(cfg, seed=1)
    m = build_reference_ditto(ref_ditto, cfg, sd)
    x, text, t = synthetic_inputs(cfg, 1, 64, 64, seed=11)
    temb = m.time_embed(m.t_embedding(t))
    pos = m.rotary(64, x.device)
    h0 = m.ada_ln(x, temb, text)
    taps = block_taps(m.blocks[0], h0, text, temb, pos)
    out = m(x, text, t)
    save("G1_block_c1.npz", x=x, text=text, t=t, temb=temb, rotary_pos=pos, after_adaln=h0,
         after_self=taps["after_self"], after_cross=taps["after_cross"], after_mlp=taps["after_mlp"], out=out)

    # ---- G2: full DiTTO 12L d=768 h=12, N=128, T=96, B=2: blocks 0,5,11 + output ----
    cfg = DiTTOConfig(768, 12, 12, 256, 768, 50)
    sd = synthetic_state_dict(cfg, seed=2)
    m = build_reference_ditto(ref_ditto, cfg, sd)
    x, text, t = synthetic_inputs(cfg, 2, 128, 96, seed=22)
    blk = {}
    hooks = [m.blocks[i].register_forward_hook(lambda mod, i_, o, i=i: blk.__setitem__(i, o.detach().clone()))
             for i in (0, 5, 11)]
    out = m(x, text, t)
    for h in hooks:
        h.remove()
    save("G2_ditto_s.npz", x=x.half(), text=text.half(), t=t,  # inputs are regenerated from synth; fp16 copy = checksum
         block0=blk[0], block5=blk[5], block11=blk[11], out=out)

    # ---- G3: shipped-config shape: 5L, ONE head (d_h = 768), N=64, T=64, B=1 ----
    cfg = DiTTOConfig(768, 5, 1, 256, 768, 1000)
    sd = synthetic_state_dict(cfg, seed=3)
    m = build_reference_ditto(ref_ditto, cfg, sd)
    x, text, t = synthetic_inputs(cfg, 1, 64, 64, seed=33)
    out = m(x, text, t)
    save("G3_shipped_1head.npz", x=x, text=text, t=t, out=out)

    # ---- G4: schedules + q_sample ----
    b50, b1000 = m.cosine_beta_schedule(50), m.cosine_beta_schedule(1000)
    x0 = hash_normal((3, 16, 768), "x0", 44)
    nz = hash_normal((3, 16, 768), "qnoise", 44)
    tq = torch.tensor([0, 500, 999])
    qs = m.q_sample(x0, tq, nz)
    save("G4_schedule_qsample.npz", betas50=b50, betas1000=b1000, buffer1000=m.alphas_cumprod,
         x0=x0, noise=nz, t=tq, q_sample=qs)

    # ---- G5: 50-step sampler trajectory, 2L d=256 h=4, N=64, T=32, B=2, injected noise ----
    # The loop + update are the RESTATED SpeechGenerator lines (it cannot be imported: torchaudio
    # and BigVGAN are missing); every eps comes from the imported reference DiTTO.forward.
    cfg = DiTTOConfig(256, 2, 4, 256, 256, 50)
    sd = synthetic_state_dict(cfg, seed=5)
    m = build_reference_ditto(ref_ditto, cfg, sd)
    B, N, T, S = 2, 64, 32, 50
    text = hash_normal((B, T, 256), "text", 55)
    xinit = hash_normal((B, N, 256), "xT", 55)
    betas = m.cosine_beta_schedule(S)                      # SpeechGenerator.py:70
    alphas = 1.0 - betas                                   # :71
    ac = torch.cumprod(alphas, dim=0)                      # :72
    x = xinit.clone()
    kept = {}
    for i, t_val in enumerate(reversed(range(S))):         # :161
        tt = torch.full((B,), t_val, dtype=torch.long)     # :162
        eps = m(x, text, tt)                               # :135
        z = hash_normal((B, N, 256), f"z{i}", 55)
        beta_t, alpha_t, ac_t = betas[tt].view(-1, 1, 1), alphas[tt].view(-1, 1, 1), ac[tt].view(-1, 1, 1)
        mask = (tt > 0).float().view(-1, 1, 1)
        x = (1 / torch.sqrt(alpha_t)) * (x - (1 - alpha_t) / torch.sqrt(1 - ac_t) * eps) \
            + mask * torch.sqrt(beta_t) * z                # :141-145
        if i in (0, 1, 10, 49):
            kept[i] = x.clone()
    save("G5_sampler_50.npz", text=text, xinit=xinit, betas=betas, alphas=alphas, alphas_cumprod=ac,
         x_step0=kept[0], x_step1=kept[1], x_step10=kept[10], x_step49=kept[49])


G8_SHAPE = dict(layers=2, d=768, heads=12, B=2, N=128, T=96, steps=50, seed=8, keep=(0, 1, 10, 49))


@torch.no_grad()
def make_loop768():
    """G8: the 50-step sampling loop at the dimensions of the TIMED kernel class — d = 768, 12 heads of 64 (the full-row
    GEMMs with fused LayerNorms, norm2 fused into the q-projection, the bf16 residual stream all exist only at this width;
    G5's d = 256 model can never reach them).  2 layers, N = 128, T = 96, B = 2.  Every eps comes from the imported
    reference DiTTO.forward (src/model/DiTTO.py:66-94); loop and update are the restated SpeechGenerator lines
    (src/model/SpeechGenerator.py:135-145,154-163: the class cannot be imported here), noise injected from hash_normal.
    Inputs are regenerated from ditto_tts_amd.synth by the tests (fp16 copies stored as a checksum only)."""
    _, ref_ditto = import_reference()
    s = G8_SHAPE
    cfg = DiTTOConfig(s["d"], s["layers"], s["heads"], 256, s["d"], s["steps"])
    m = build_reference_ditto(ref_ditto, cfg, synthetic_state_dict(cfg, seed=s["seed"]))
    B, N, T, S = s["B"], s["N"], s["T"], s["steps"]
    text = hash_normal((B, T, s["d"]), "text", 88)
    xinit = hash_normal((B, N, s["d"]), "xT", 88)
    betas = m.cosine_beta_schedule(S)                      # SpeechGenerator.py:70
    alphas = 1.0 - betas                                   # :71
    ac = torch.cumprod(alphas, dim=0)                      # :72
    x = xinit.clone()
    kept = {}
    for i, t_val in enumerate(reversed(range(S))):         # :161
        tt = torch.full((B,), t_val, dtype=torch.long)     # :162
        eps = m(x, text, tt)                               # :135
        z = hash_normal((B, N, s["d"]), f"z{i}", 88)
        beta_t, alpha_t, ac_t = betas[tt].view(-1, 1, 1), alphas[tt].view(-1, 1, 1), ac[tt].view(-1, 1, 1)
        mask = (tt > 0).float().view(-1, 1, 1)
        x = (1 / torch.sqrt(alpha_t)) * (x - (1 - alpha_t) / torch.sqrt(1 - ac_t) * eps) \
            + mask * torch.sqrt(beta_t) * z                # :141-145
        if i in s["keep"]:
            kept[i] = x.clone()
    save("G8_loop768.npz", text16=text.half(), xinit16=xinit.half(),
         **{f"x_step{i}": kept[i] for i in s["keep"]})


@torch.no_grad()
def make_vq():
    """G6: the reference's VectorQuantizer on a hashed codebook / latents (indices are the fixture)."""
    import_reference()
    import components.VectorQuantizer as ref_vq
    vq = ref_vq.VectorQuantizer(1024, 768)
    cb = hash_normal((1024, 768), "codebook", 66) * 0.05
    vq.codebook.data.copy_(cb)
    lat = hash_normal((2, 2, 96, 768), "vq_latents", 66) * 0.06
    lat[0, 0, :8] = cb[:8] + 1e-3 * hash_normal((8, 768), "jit", 66)     # near exact hits on known rows
    idx = vq(lat)
    assert idx[0, 0, :8].tolist() == list(range(8))
    save("G6_vq.npz", indices=idx.to(torch.int16))


SLP_CASES = {   # name: (d_model, nhead, num_layers, classes, B, codebooks x frames, T)
    "G7_slp_4head": (128, 4, 2, 11, 2, (2, 20), 24),      # head width 32: packed zero-padded to 64
    "G7_slp_1head": (192, 1, 1, 11, 3, (2, 9), 16),       # ConfigSLP's shape class: one layer, one head
}


@torch.no_grad()
def make_slp():
    """G7: the reference's own SLP.forward (src/model/SpeechLP.py:36-55) with its two pretrained encoders replaced by
    pass-through stand-ins (ByT5 / EnCodec need the HF hub), so the view(), the causal mask, the TransformerDecoder
    and the last-position head that run are the reference's.  The fixture holds the logits and the decoder output
    (z_audio_decoded, tapped with a forward hook); inputs and weights are
    closed-form (ditto_tts_amd.synth)."""
    import types

    import torch.nn as nn
    import_reference()
    import model.SpeechLP as ref_slp
    from ditto_tts_amd.synth import synthetic_slp_state_dict

    for name, (d, nhead, nl, ncls, B, (ncb, nfr), T) in SLP_CASES.items():
        class TextStandIn(nn.Module):
            def __init__(self):
                super().__init__()
                self.model = types.SimpleNamespace(config=types.SimpleNamespace(d_model=d))

            def forward(self, X):
                return X

        class AudioStandIn(nn.Module):
            def __init__(self, hidden_size):
                super().__init__()

            def forward(self, X):
                return X, None

        ref_slp.ByT5, ref_slp.EnCodec = TextStandIn, AudioStandIn
        slp = ref_slp.SLP(ncls, nhead, nl).eval()
        missing = slp.load_state_dict(synthetic_slp_state_dict(d, nhead, nl, ncls, 5), strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        z_text = hash_normal((B, T, d), "slp_text", 5)
        z_audio = hash_normal((B, ncb, nfr, d), "slp_audio", 5)       # EnCodec wrapper's [B, codebooks, frames, d]
        tap = {}
        slp.transformer.register_forward_hook(lambda mod, args, out: tap.__setitem__("decoded", out))
        logits = slp(z_text, z_audio)
        assert logits.shape == (B, ncls) and tap["decoded"].shape == (B, ncb * nfr, d)
        save(name + ".npz", logits=logits, decoded=tap["decoded"])


if __name__ == "__main__":
    if "--vq-only" in sys.argv:
        make_vq()
    elif "--slp-only" in sys.argv:
        make_slp()
    elif "--loop768-only" in sys.argv:
        make_loop768()
    else:
        main()
        make_vq()
        make_slp()
        make_loop768()
