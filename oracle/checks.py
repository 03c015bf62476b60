"""Checker routines shared by tests/ and bench.py's post-timing self-validation.  TEST INFRASTRUCTURE ONLY (like the
rest of oracle/): they run the HIP product path and compare it with the fp32 CPU oracle; nothing in ditto_tts_amd/ imports
this file.

train_grad_parity — SURVEY.md §8f row 1 at the TIMED model dimensions.  The C2 training step that bench.py times runs
kernels which the small-shape gradient tests never reach: d = 768 / 12 heads of 64 (whole 64-row attention tiles, the fused
attention backward with the rotation in its epilogues), the full-row forward GEMMs with the fused LayerNorms, the long-K
dgrads on the full-row kernel and the 256 x 256 weight-gradient tiles with the XCD map.  This check builds a 2-layer model
of exactly those widths, pins the kernel class to the timed batch (32 x 1024 rows: hip.call_opts(class_rows=...), ditto_call_opts), runs
the reference's training closure (src/TrainDiTTO.py:85-91: model.train() -> forward -> MSE -> backward; cross-attention
dropout p = 0.1 active, src/components/DiT.py:90-91) and compares EVERY parameter gradient with fp32 autograd over the
oracle, whose dropout mask is the restated counter hash (hash_dropout_mask)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import ditto_oracle as O

TIMED_SHAPE = dict(hidden_dim=768, num_layers=2, num_heads=12, time_dim=256, text_dim=768, diffusion_steps=50,
                   B=2, N=256, T=192, class_rows=32 * 1024)
GRAD_TOL = 3e-2          # per-tensor rel-L2: bf16 operands, fp32 accumulation, bf16-rounded activation gradients


def _rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float(torch.linalg.norm(a - b) / torch.linalg.norm(b).clamp_min(1e-30))


def train_grad_parity(dev="cuda", train_mode=True, pin_class=True, seed=21, shape=None):
    """-> {"worst_rel_l2", "tensor", "median_rel_l2", "loss_rel", "out_rel_l2", "n_tensors", "tol", "ok", "what"}"""
    from ditto_tts_amd import hip
    from ditto_tts_amd.config import DiTTOConfig
    from ditto_tts_amd.modules import DiTTO
    from ditto_tts_amd.synth import hash_normal, synthetic_inputs, synthetic_state_dict

    s = dict(TIMED_SHAPE)
    s.update(shape or {})
    cfg = DiTTOConfig(s["hidden_dim"], s["num_layers"], s["num_heads"], s["time_dim"], s["text_dim"], s["diffusion_steps"])
    B, N, T = s["B"], s["N"], s["T"]
    x, text, t = synthetic_inputs(cfg, B, N, T, seed=seed)
    target = hash_normal((B, N, cfg.hidden_dim), "noise", seed + 1)
    p = 0.1 if train_mode else 0.0
    torch.manual_seed(1000 + seed)
    drop_seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if train_mode else None   # what DiTTO.forward will draw

    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in synthetic_state_dict(cfg, seed).items()}
    want_out = O.ditto_forward(sd, cfg.num_layers, cfg.num_heads, x, text, t, dropout_p=p, dropout_seed=drop_seed)
    want_loss = F.mse_loss(want_out, target)
    want_loss.backward()

    m = DiTTO(cfg.hidden_dim, cfg.num_layers, cfg.num_heads, cfg.time_dim, cfg.text_dim, cfg.diffusion_steps)
    m.load_state_dict(synthetic_state_dict(cfg, seed))
    m = m.to(dev)
    m = m.train() if train_mode else m.eval()
    plan = None
    # the class pin is a scope of THIS thread's calls; the autograd Function carries it to the backward (run on autograd's thread)
    with hip.call_opts(class_rows=s["class_rows"] if pin_class else None):
        plan = hip.full_row_plan(cfg, B, N)
        torch.manual_seed(1000 + seed)
        out = m(x.to(dev), text.to(dev), t.to(dev))
        loss = F.mse_loss(out, target.to(dev))
    loss.backward()       # outside the scope on purpose
    rels, missing = [], []
    for name, prm in m.named_parameters():
        w = sd[name].grad
        if ".attn.out_proj." in name:          # the reference never calls it: no gradient on either side
            if prm.grad is not None or w is not None:
                missing.append(name)
            continue
        if prm.grad is None or w is None:
            missing.append(name)
            continue
        rels.append((_rel(prm.grad, w), name))
    rels.sort()
    worst = rels[-1]
    return {"worst_rel_l2": worst[0], "tensor": worst[1], "median_rel_l2": rels[len(rels) // 2][0],
            "loss_rel": abs(float(loss) - float(want_loss)) / float(want_loss),
            "out_rel_l2": _rel(out.detach(), want_out.detach()), "n_tensors": len(rels), "unexpected": missing,
            "full_row_forward": list(plan) if plan else None, "tol": GRAD_TOL,
            "ok": bool(worst[0] < GRAD_TOL and not missing),
            "what": f"{cfg.num_layers}L d={cfg.hidden_dim} h={cfg.num_heads} N={N} T={T} B={B}, "
                    f"{'train mode (cross-attention dropout 0.1, hashed mask)' if train_mode else 'eval mode'}, kernel class "
                    f"{'pinned to 32x1024 rows (the timed training step)' if pin_class else 'of its own rows'}: every parameter "
                    "gradient vs fp32 autograd of the CPU oracle"}
