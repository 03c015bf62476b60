#!/usr/bin/env python3
"""In-model A/B of GEMM variants: per-kernel-class average launch time of the real denoise step (C2, B=32),
variants interleaved round-robin in ONE process on ONE device (cross-run numbers differ by >10 % between boxes).
    python tools/step_ab.py --variants 128/73,256/73,0/73 [--rounds 4] [--steps 3]
variant = gemm_tile/gemm_flags[@attn_flags][#splitk_wgs][%gemm_group][^pp_mask][!pp_nb][~fr_mask][&fr_rot][+fr_tile.fr64_maxk.fr_stagger]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ditto_tts_amd import hip  # noqa: E402
from ditto_tts_amd.config import PRESETS  # noqa: E402
from ditto_tts_amd.modules import DiTTO  # noqa: E402
from ditto_tts_amd.sampler import SpeechGenerator  # noqa: E402
from ditto_tts_amd.synth import synthetic_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="128/73,256/73")
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--config", default="C2")
ap.add_argument("--batch", type=int, default=None)
a = ap.parse_args()
lib = hip.lib()
p = PRESETS[a.config]
cfg, N, T = p["cfg"], p["N"], p["T"]
B = a.batch or p["B"]
dev = torch.device("cuda")
m = DiTTO(cfg.hidden_dim, cfg.num_layers, cfg.num_heads, cfg.time_dim, cfg.text_dim, cfg.diffusion_steps, fp8_linear=cfg.fp8_linear)
m.load_state_dict(synthetic_state_dict(cfg, seed=1234))
m = m.to(dev).eval()
sg = SpeechGenerator(ditto_model=m, device=dev)
eng = m.engine(dev)
text = torch.randn(B, T, cfg.text_dim, device=dev)
x = torch.randn(B, N, cfg.hidden_dim, device=dev)
z = torch.randn_like(x)
t = torch.full((B,), cfg.diffusion_steps - 1, device=dev, dtype=torch.long)
cond = eng.prepare_text(text, N)
variants = a.variants.split(",")
acc = {v: {} for v in variants}
wall = {v: [] for v in variants}


GENERIC_OPTS = {}   # name -> the value found when a variant first named it (restored for variants without it)


def select(v):
    v, _, kv = v.partition(":")          # ...:name.value[,name.value] arbitrary ditto_set_option pairs (reset to 0 by variants without them)
    for name, dflt in GENERIC_OPTS.items():
        hip.check(lib.ditto_set_option(name.encode(), dflt))
    for pair in [x for x in kv.split(";") if x]:
        name, _, val = pair.partition(".")
        if name not in GENERIC_OPTS:
            GENERIC_OPTS[name] = hip.get_option(name)
        hip.check(lib.ditto_set_option(name.encode(), int(val)))
    v, _, cr = v.partition("=")          # ...=fr_class_rows (pin the full-row kernel class: rows of the batch to decide as)
    if cr or not os.environ.get("DITTO_HIP_LIB"):
        hip.check(lib.ditto_set_option(b"fr_class_rows", int(cr) if cr else 0))
    v, _, ft = v.partition("+")          # ...+fr_tile.fr64_maxk.fr_stagger (64-row full-row kernel: gemm_fr64.hip; default: the rule)
    ftv = [int(x) for x in ft.split(".")] if ft else []
    if ft or not os.environ.get("DITTO_HIP_LIB"):   # (an older library selected with DITTO_HIP_LIB does not know these)
        hip.check(lib.ditto_set_option(b"fr_tile", ftv[0] if len(ftv) > 0 else 0))
        hip.check(lib.ditto_set_option(b"fr64_maxk", ftv[1] if len(ftv) > 1 else 1 << 30))
        hip.check(lib.ditto_set_option(b"fr_stagger", ftv[2] if len(ftv) > 2 else 0))
    v, _, fr = v.partition("&")          # ...&fr_rot (full-row GEMM K-loop rotation: default 1 = on)
    hip.check(lib.ditto_set_option(b"fr_rot", int(fr) if fr else 1))
    v, _, fm = v.partition("~")          # ...~fr_mask (full-row GEMM + fused LayerNorm: 1 out-proj, 2 fc2)
    hip.check(lib.ditto_set_option(b"fr_mask", int(fm) if fm else 0))
    v, _, nb = v.partition("!")          # ...!pp_nb (ping-pong tile width: 3 = 192, 4 = 256; default 0 = rule)
    hip.check(lib.ditto_set_option(b"pp_nb", int(nb) if nb else 0))
    v, _, pm = v.partition("^")          # ...^pp_mask (GEMM classes on the ping-pong kernel; default -1 = rule)
    hip.check(lib.ditto_set_option(b"pp_mask", int(pm) if pm else -1))
    v, _, gg = v.partition("%")          # ...%gemm_group (forced super-column width of the tile order; 0 = rule)
    hip.check(lib.ditto_set_option(b"gemm_group", int(gg) if gg else 0))
    v, _, sk = v.partition("#")          # ...#splitk_wgs (small-batch split-K target; default 0 = off)
    hip.check(lib.ditto_set_option(b"splitk_wgs", int(sk) if sk else 0))
    v, _, af = v.partition("@")          # tile/gemm_flags@attn_flags
    hip.check(lib.ditto_set_option(b"attn_flags", int(af) if af else 0))
    tile, _, fl = v.partition("/")
    hip.check(lib.ditto_set_option(b"gemm_tile", int(tile)))
    hip.check(lib.ditto_set_option(b"gemm_flags", int(fl) if fl else 321))


with torch.no_grad():
    for v in variants:
        select(v)
        for _ in range(2):
            x.normal_()
            eng.p_sample_(x, cond, t, z, sg.betas, sg.alphas, sg.alphas_cumprod)
    torch.cuda.synchronize()
    for r in range(a.rounds):
        for v in variants:
            select(v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            x.normal_()
            e0.record()
            for _ in range(a.steps):
                eng.p_sample_(x, cond, t, z, sg.betas, sg.alphas, sg.alphas_cumprod)
            e1.record()
            e1.synchronize()
            wall[v].append(e0.elapsed_time(e1) / a.steps)
            eng.profile_enable(True)
            x.normal_()
            for _ in range(a.steps):
                eng.p_sample_(x, cond, t, z, sg.betas, sg.alphas, sg.alphas_cumprod)
            torch.cuda.synchronize()
            for k, (n, ms) in eng.profile_read().items():
                c = acc[v].setdefault(k, [0, 0.0])
                c[0] += n
                c[1] += ms
            eng.profile_enable(False)
names = [k for k in acc[variants[0]] if acc[variants[0]][k][0]]
print("variant      step_ms  " + "  ".join(f"{k[:12]:>12s}" for k in names))
for v in variants:
    print(f"{v:10s} {sorted(wall[v])[len(wall[v]) // 2]:8.2f}  " +
          "  ".join(f"{1e3 * acc[v][k][1] / max(acc[v][k][0], 1):12.1f}" for k in names), flush=True)
