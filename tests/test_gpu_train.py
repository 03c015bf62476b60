"""-m gpu parity of the BACKWARD pass through the C-ABI / the nn.Module surface.

Oracle: torch autograd over the fp32 CPU restatement (oracle/ditto_oracle.py) — the reference's own modules under
autograd are what tests/test_oracle_vs_reference.py pins that restatement to.  Tolerances (stated per test):
fp32 row kernels 1e-5; bf16-operand attention backward 2e-2; whole-model parameter gradients rel-L2 <= 3e-2 per
tensor (bf16 operands, fp32 accumulation, bf16-rounded activation gradients between GEMMs).
Train-mode dropout: the HIP path draws the cross-attention keep-mask from a counter-based hash, restated in the
oracle (hash_dropout_mask), so train-mode VALUES are compared, not only structure."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from ditto_tts_amd import hip
from ditto_tts_amd.config import DiTTOConfig
from ditto_tts_amd.modules import DiTTO
from ditto_tts_amd.synth import hash_normal, synthetic_inputs, synthetic_state_dict
from gpu_util import bf16, max_abs, rel_l2, stream
from oracle import ditto_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


# ----------------------------------------------------------------------------------------------- row kernels
@pytest.mark.parametrize("M,d,groups", [(37, 64, 1), (300, 768, 1), (96, 256, 4), (50, 2048, 2), (4096, 768, 8)])
def test_layernorm_backward_vs_autograd(M, d, groups):
    lib = hip.lib()
    x = (hash_normal((M, d), "x", M) * 1.7 + 0.3).requires_grad_(True)
    gamma = (1 + 0.2 * hash_normal((d,), "g", d)).requires_grad_(True)
    beta = (0.1 * hash_normal((d,), "b", d)).requires_grad_(True)
    dy = hash_normal((M, d), "dy", M + 1)
    F.layer_norm(x, (d,), gamma, beta, 1e-5).backward(dy)
    rpg = M // groups
    xd, dyd, gd = x.detach().to(DEV), dy.to(DEV), gamma.detach().to(DEV)
    acc0 = hash_normal((M, d), "acc", 3).to(DEV)
    acc = acc0.clone()
    dgb = torch.empty(groups, 2 * d, device=DEV)
    nb = lib.ditto_layernorm_bwd_scratch_bytes(rpg, groups, d)
    scratch = torch.empty(nb, dtype=torch.uint8, device=DEV)
    hip.check(lib.ditto_layernorm_bwd(dyd.data_ptr(), xd.data_ptr(), gd.data_ptr(), acc.data_ptr(), dgb.data_ptr(),
                                      scratch.data_ptr(), nb, rpg, groups, d, stream()))
    assert rel_l2(acc - acc0, x.grad) < 1e-5
    assert rel_l2(dgb[:, :d].sum(0), gamma.grad) < 1e-5 and rel_l2(dgb[:, d:].sum(0), beta.grad) < 1e-5
    if groups > 1:   # per-group partials (the GlobalAdaLN reduction): group g = rows [g*rpg, (g+1)*rpg)
        xh = F.layer_norm(x.detach(), (d,), None, None, 1e-5)
        want = (dy * xh).view(groups, rpg, d).sum(1)
        assert rel_l2(dgb[:, :d], want) < 1e-5
    # gamma = NULL (ones), dx only
    x2 = x.detach().clone().requires_grad_(True)
    F.layer_norm(x2, (d,), None, None, 1e-5).backward(dy)
    acc = torch.zeros(M, d, device=DEV)
    hip.check(lib.ditto_layernorm_bwd(dyd.data_ptr(), xd.data_ptr(), None, acc.data_ptr(), None, None, 0, rpg, groups,
                                      d, stream()))
    assert rel_l2(acc, x2.grad) < 1e-5


def _attn_ref(q, k, v, scale, mask=None, p=0.0):
    """fp32 reference on [B,H,S,dh] tensors; mask = keep-mask [B,H,Sq,Skv]"""
    a = torch.softmax(torch.matmul(q, k.transpose(-2, -1)) * scale, dim=-1)
    if mask is not None:
        a = a * mask / (1.0 - p)
    return torch.matmul(a, v)


def _heads(t, B, S, H, dh):
    return t.view(B, S, H, dh).permute(0, 2, 1, 3)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("B,H,Sq,Skv,dh,p", [(1, 2, 64, 64, 64, 0.0), (2, 2, 100, 72, 64, 0.1), (1, 1, 48, 80, 256, 0.1),
                                              (2, 3, 130, 130, 64, 0.0), (1, 2, 257, 321, 64, 0.1),
                                              (1, 2, 256, 192, 64, 0.1),     # whole tiles + dropout (no mask instantiation)
                                              (1, 1, 128, 640, 64, 0.0),     # 10 key tiles: the 4-buffer ring wraps twice
                                              (1, 1, 704, 64, 64, 0.1)])     # 11 query tiles for the dk,dv kernel's ring
def test_attention_dropout_forward_and_backward_vs_autograd(B, H, Sq, Skv, dh, p, fused):
    """bf16 operands: compare against fp32 autograd on the SAME bf16-rounded q/k/v/dO.  Tolerance 2e-2 rel-L2
    (P and dS are rounded to bf16 between the two products).  fused = the head_dim-64 flash kernels (forward with
    log-sum-exp, two-kernel backward); otherwise the GEMM-composed path."""
    if fused and dh != 64:
        pytest.skip("the fused kernels are head_dim 64 only")
    lib = hip.lib()
    d = H * dh
    seed, layer = 0x1234567890ABCDEF, 3
    q = bf16(hash_normal((B, Sq, d), "q", 1)).float().requires_grad_(True)
    k = bf16(hash_normal((B, Skv, d), "k", 2)).float().requires_grad_(True)
    v = bf16(hash_normal((B, Skv, d), "v", 3)).float().requires_grad_(True)
    do = bf16(hash_normal((B, Sq, d), "do", 4)).float()
    scale = dh ** -0.5
    mask = O.hash_dropout_mask(seed, layer, B, H, Sq, Skv, p) if p > 0 else None
    qh, kh, vh = _heads(q, B, Sq, H, dh), _heads(k, B, Skv, H, dh), _heads(v, B, Skv, H, dh)
    out = _attn_ref(qh, kh, vh, scale, mask, p)
    out = out.permute(0, 2, 1, 3).reshape(B, Sq, d)
    out.backward(do)
    qd, kd, vd, dod = (bf16(z.detach()).to(DEV) for z in (q, k, v, do))
    nb = lib.ditto_attention_bwd_workspace_bytes(B, H, Sq, Skv, dh)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    o = torch.empty(B, Sq, d, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B, H, Sq, dtype=torch.float32, device=DEV) if fused else None
    hip.check(lib.ditto_attention_dropout_bf16(qd.data_ptr(), d, kd.data_ptr(), d, vd.data_ptr(), d, o.data_ptr(), d,
                                               lse.data_ptr() if fused else None, B, H, Sq, Skv, dh, scale, p, seed,
                                               layer, ws.data_ptr(), nb, stream()))
    assert rel_l2(o.float(), out.detach()) < 1.5e-2
    if fused:   # log2-domain log-sum-exp of the scaled scores
        want = torch.logsumexp(torch.matmul(qh, kh.transpose(-2, -1)).detach() * scale, dim=-1) / 0.6931471805599453
        assert max_abs(lse, want) < 2e-3 * (1 + float(want.abs().max()))
    dq = torch.empty_like(qd); dk = torch.empty_like(kd); dv = torch.empty_like(vd)
    hip.check(lib.ditto_attention_bwd_bf16(qd.data_ptr(), d, kd.data_ptr(), d, vd.data_ptr(), d, dod.data_ptr(), d,
                                           o.data_ptr(), d, lse.data_ptr() if fused else None,
                                           dq.data_ptr(), d, dk.data_ptr(), d, dv.data_ptr(), d, B, H, Sq, Skv, dh,
                                           scale, p, seed, layer, ws.data_ptr(), nb, stream()))
    for name, got, want in (("dq", dq, q.grad), ("dk", dk, k.grad), ("dv", dv, v.grad)):
        r = rel_l2(got.float(), want)
        assert r < 2e-2, f"{name}: rel-L2 {r:.3e}"


# ----------------------------------------------------------------------------------------------- whole model
def _build(cfg, seed):
    m = DiTTO(cfg.hidden_dim, cfg.num_layers, cfg.num_heads, cfg.time_dim, cfg.text_dim, cfg.diffusion_steps)
    m.load_state_dict(synthetic_state_dict(cfg, seed))
    return m.to(DEV)


def _oracle_grads(cfg, seed, x, text, t, target, p=0.0, drop_seed=None):
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in synthetic_state_dict(cfg, seed).items()}
    out = O.ditto_forward(sd, cfg.num_layers, cfg.num_heads, x, text, t, dropout_p=p, dropout_seed=drop_seed)
    loss = F.mse_loss(out, target)
    loss.backward()
    return out.detach(), float(loss), {k: v.grad for k, v in sd.items() if v.requires_grad}


def _check_grads(m, want, tol):
    worst = (0.0, "")
    for name, p in m.named_parameters():
        if ".attn.out_proj." in name:
            assert p.grad is None, f"{name}: the reference never uses it, so it must get no gradient"
            assert want[name] is None
            continue
        assert p.grad is not None, f"{name}: no gradient"
        r = rel_l2(p.grad, want[name])
        worst = max(worst, (r, name))
        assert r < tol, f"{name}: rel-L2 {r:.3e}"
    return worst


@pytest.mark.parametrize("cfg,B,N,T", [
    (DiTTOConfig(128, 2, 2, 64, 128, 20), 2, 48, 40),          # d_h = 64: fused forward attention
    (DiTTOConfig(256, 1, 1, 64, 256, 20), 1, 40, 24),          # ONE head, d_h = 256 (the shipped config's shape class)
    (DiTTOConfig(256, 3, 4, 256, 256, 50), 3, 100, 72),        # ragged N / T, 3 layers
])
def test_parameter_gradients_vs_oracle_autograd(cfg, B, N, T):
    """eval-mode forward under autograd (no dropout): every live parameter's gradient vs fp32 autograd of the
    oracle.  Tolerance rel-L2 <= 3e-2 per tensor."""
    x, text, t = synthetic_inputs(cfg, B, N, T, seed=7)
    t[0] = t[-1]                                       # duplicate timestep rows: t_embedding grads must accumulate
    target = hash_normal((B, N, cfg.hidden_dim), "noise", 9)
    want_out, want_loss, want = _oracle_grads(cfg, 4, x, text, t, target)
    m = _build(cfg, 4).eval()
    out = m(x.to(DEV), text.to(DEV), t.to(DEV))
    assert out.requires_grad and rel_l2(out, want_out) < 2e-2
    loss = F.mse_loss(out, target.to(DEV))
    loss.backward()
    assert abs(float(loss) - want_loss) < 2e-2 * want_loss
    _check_grads(m, want, 3e-2)


@pytest.mark.parametrize("train_mode,pin_class", [(True, True), (False, True), (True, False)])
def test_parameter_gradients_at_the_timed_dimensions_vs_oracle_autograd(train_mode, pin_class):
    """SURVEY §8f row 1 at the dimensions bench.py TIMES (oracle/checks.py train_grad_parity): 2 layers of d = 768, 12 heads of
    64, N = 256, T = 192, B = 2, the reference's training closure (src/TrainDiTTO.py:85-91) with the cross-attention dropout
    active, the kernel class pinned to the timed batch of 32 x 1024 rows so that the full-row forward GEMMs + fused
    LayerNorms, the full-row dgrads, the fused attention backward and the 256 x 256 weight-gradient tiles are the kernels
    that run.  Every parameter gradient against fp32 autograd of the oracle: rel-L2 <= 3e-2 per tensor (bf16 operands, fp32
    accumulation).  (False, True) = eval mode; (True, False) = the same model on the kernels of its own 512 rows."""
    from oracle.checks import train_grad_parity
    r = train_grad_parity(DEV, train_mode=train_mode, pin_class=pin_class)
    assert not r["unexpected"], r["unexpected"]
    assert r["n_tensors"] > 40
    if pin_class:
        assert r["full_row_forward"] == [True, True], "the timed (full-row) forward kernels did not engage"
    assert r["out_rel_l2"] < 2e-2 and r["loss_rel"] < 2e-2, r
    assert r["worst_rel_l2"] < r["tol"], r


def test_train_mode_dropout_values_and_gradients_vs_oracle():
    """model.train(): cross-attention dropout p = 0.1 with the hashed mask; forward values and gradients."""
    cfg = DiTTOConfig(128, 2, 2, 64, 128, 20)
    B, N, T = 2, 64, 48
    x, text, t = synthetic_inputs(cfg, B, N, T, seed=8)
    target = hash_normal((B, N, 128), "noise", 10)
    torch.manual_seed(77)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())   # what DiTTO.forward will draw
    want_out, want_loss, want = _oracle_grads(cfg, 5, x, text, t, target, p=0.1, drop_seed=seed)
    nodrop_out, _, _ = _oracle_grads(cfg, 5, x, text, t, target)
    m = _build(cfg, 5).train()
    torch.manual_seed(77)
    out = m(x.to(DEV), text.to(DEV), t.to(DEV))
    assert rel_l2(out, want_out) < 2e-2
    assert rel_l2(out, nodrop_out) > 2 * rel_l2(out, want_out)        # the mask really was applied
    F.mse_loss(out, target.to(DEV)).backward()
    _check_grads(m, want, 3e-2)
    # same seed -> bit-identical, another seed -> different
    torch.manual_seed(77)
    assert torch.equal(m(x.to(DEV), text.to(DEV), t.to(DEV)), out)
    torch.manual_seed(78)
    assert not torch.equal(m(x.to(DEV), text.to(DEV), t.to(DEV)), out)


def test_training_closure_like_the_reference():
    """The loop of reference src/TrainDiTTO.py:55-95 (q_sample -> forward -> MSE -> backward -> AdamW step) on the
    HIP path against the same loop on the oracle: losses agree step by step and go down."""
    cfg = DiTTOConfig(128, 2, 2, 64, 128, 20)
    B, N, T, steps = 2, 32, 32, 4
    m = _build(cfg, 6).eval()            # eval: no dropout, so both loops see the same function
    opt = torch.optim.AdamW([p for n, p in m.named_parameters()], lr=2e-3)
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in synthetic_state_dict(cfg, 6).items()}
    live = [v for k, v in sd.items() if v.requires_grad and ".attn.out_proj." not in k and k != "alphas_cumprod"]
    opt_o = torch.optim.AdamW(live, lr=2e-3)
    losses, losses_o = [], []
    for i in range(steps):
        x0 = hash_normal((B, N, 128), f"x0{i}", 1); noise = hash_normal((B, N, 128), f"nz{i}", 2)
        text = hash_normal((B, T, 128), f"tx{i}", 3); t = torch.tensor([3 + i, 17 - i])
        xt = m.q_sample(x0.to(DEV), t.to(DEV), noise.to(DEV))
        loss = F.mse_loss(m(xt, text.to(DEV), t.to(DEV)), noise.to(DEV))
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(float(loss))
        xt_o = O.q_sample(sd["alphas_cumprod"], x0, t, noise)
        loss_o = F.mse_loss(O.ditto_forward(sd, 2, 2, xt_o, text, t), noise)
        opt_o.zero_grad(); loss_o.backward(); opt_o.step()
        losses_o.append(float(loss_o))
    for a, b in zip(losses, losses_o):
        assert abs(a - b) < 3e-2 * b, (losses, losses_o)
    # the same batch again after the updates: lower loss than before them
    with torch.no_grad():
        again = float(F.mse_loss(m(xt, text.to(DEV), t.to(DEV)), noise.to(DEV)))
    assert again < losses[-1]


@pytest.mark.parametrize("cfg,B,N,T", [(DiTTOConfig(256, 2, 4, 256, 256, 50), 4, 512, 256),
                                       (DiTTOConfig(768, 2, 12, 256, 768, 50), 3, 1024, 128)])
def test_gradients_are_bit_reproducible(cfg, B, N, T):
    """No atomics anywhere in the backward (split-K partials and column sums are reduced in a fixed order): the same
    step twice gives bitwise-identical gradients, at a shape that exercises split-K wgrad and the fused attention, and at one
    (d = 768, 3 x 1024 rows) where the gated MLP's derivative runs as the fc2 dgrad's epilogue with its per-half-tile partial
    rows of the bias gradients."""
    x, text, t = (z.to(DEV) for z in synthetic_inputs(cfg, B, N, T, seed=3))
    target = hash_normal((B, N, cfg.hidden_dim), "noise", 4).to(DEV)
    grads = []
    for _ in range(2):
        m = _build(cfg, 9).train()
        torch.manual_seed(123)
        F.mse_loss(m(x, text, t), target).backward()
        grads.append({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    assert grads[0].keys() == grads[1].keys() and len(grads[0]) > 40
    for n in grads[0]:
        assert torch.equal(grads[0][n], grads[1][n]), n


def test_rope_backward_in_the_attention_epilogue_vs_its_own_pass():
    """head_dim 64: the backward of the self-attention rotation runs inside the dq / dk epilogues of the attention backward
    (csrc/attention_bwd.hip) on the fp32 accumulators; train_flags 1 = the older separate in-place pass over the bf16 dq | dk.
    Both against fp32 autograd of the oracle (rel-L2 <= 3e-2 per tensor), the fused one no worse than the pass on the QKV
    weight gradient (it rounds to bf16 once instead of twice), and the two differ (the switch is live).  Ragged N = 200
    exercises the partial last 64-row tile of the epilogue."""
    cfg = DiTTOConfig(256, 2, 4, 64, 256, 20)
    B, N, T = 2, 200, 72
    x, text, t = synthetic_inputs(cfg, B, N, T, seed=11)
    target = hash_normal((B, N, cfg.hidden_dim), "noise", 12)
    _, _, want = _oracle_grads(cfg, 5, x, text, t, target)
    got = {}
    for flag in (0, 1):
        hip.set_option("train_flags", flag)
        try:
            m = _build(cfg, 5).eval()
            F.mse_loss(m(x.to(DEV), text.to(DEV), t.to(DEV)), target.to(DEV)).backward()
            _check_grads(m, want, 3e-2)
            got[flag] = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        finally:
            hip.set_option("train_flags", 0)
    name = "blocks.0.attn.in_proj_weight"
    r0, r1 = rel_l2(got[0][name], want[name]), rel_l2(got[1][name], want[name])
    assert r0 <= r1 * 1.05, (r0, r1)
    assert not torch.equal(got[0][name], got[1][name]), "train_flags did not switch the path"
    d = cfg.hidden_dim                                   # dv takes no rotation: its rows of the gradient are bitwise the same
    assert torch.equal(got[0]["blocks.1.attn.in_proj_weight"][2 * d:], got[1]["blocks.1.attn.in_proj_weight"][2 * d:])


def test_gated_mlp_backward_in_the_fc2_dgrad_epilogue():
    """From 144 tiles of 256 x 256 on, the backward of the gated MLP (reference src/components/DiT.py:152-154 differentiated) is
    the EPILOGUE of the fc2 dgrad GEMM (csrc/gemm_common.h epilogue_gated_bwd: dact stays in the accumulators, [da | dg] and the
    bias gradients' partial rows leave the kernel) and the training forward's gated GEMM is its own straight-line instantiation
    (EPI_GATED_PRE); train_flags 6 = the older two launches / general epilogue.  M = 4 x 1000 rows: the last 256-row tile is
    partial (the guarded epilogue).  (a) every parameter gradient against fp32 autograd of the oracle, rel-L2 <= 3e-2;
    (b) fused against unfused: the forward bit-identical, the MLP gradients within 2e-3 (same bf16 rounding of dact, the
    bias sums in another order) — and not all equal (the switch is live)."""
    from oracle.checks import train_grad_parity
    shape = {"B": 4, "N": 1000, "T": 192}
    r = train_grad_parity(DEV, train_mode=True, pin_class=False, shape=shape)
    assert not r["unexpected"], r["unexpected"]
    assert r["out_rel_l2"] < 2e-2 and r["worst_rel_l2"] < r["tol"], r
    cfg = DiTTOConfig(768, 2, 12, 256, 768, 50)
    x, text, t = (z.to(DEV) for z in synthetic_inputs(cfg, 4, 1000, 192, seed=5))
    target = hash_normal((4, 1000, 768), "noise", 6).to(DEV)
    res = {}
    for flag in (0, 6):
        hip.set_option("train_flags", flag)
        try:
            m = _build(cfg, 12).eval()
            out = m(x, text, t)
            F.mse_loss(out, target).backward()
            res[flag] = (out.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
        finally:
            hip.set_option("train_flags", 0)
    assert torch.equal(res[0][0], res[6][0]), "EPI_GATED_PRE must compute what EPI_GATED computes"
    differs = 0
    for n, g in res[0][1].items():
        assert torch.isfinite(g).all(), n
        assert rel_l2(g, res[6][1][n]) < 2e-3, (n, rel_l2(g, res[6][1][n]))
        differs += int(not torch.equal(g, res[6][1][n]))
    assert differs > 0, "train_flags did not switch the path"


@pytest.mark.parametrize("pin", [False, True])
def test_layernorm_input_gradient_travels_as_bf16(pin):
    """du — the gradient wrt a LayerNorm's output, written by a dgrad GEMM and read once by the LayerNorm backward — is bf16 in
    HBM (train_flags 8 = fp32, the round-3 form); the stream gradient dh it is folded into stays fp32.  Both forms (on the fp32
    tape: train_flags 16, so that nothing else differs) against each
    other: every parameter gradient within 4e-3 (one more bf16 rounding per segment), and not all equal.  pin = the kernel class
    of the timed step (32 x 1024 rows): the long-K dgrads then run on the full-row kernel, whose bf16-output instantiation
    without residual exists for this; unpinned: the tiled dgrad GEMMs' bf16 epilogue.  (Both forms against the oracle:
    test_parameter_gradients_at_the_timed_dimensions_vs_oracle_autograd runs the default; 3e-2.)"""
    cfg = DiTTOConfig(768, 2, 12, 256, 768, 50)
    B, N, T = 2, 256, 192
    x, text, t = (z.to(DEV) for z in synthetic_inputs(cfg, B, N, T, seed=15))
    target = hash_normal((B, N, 768), "noise", 16).to(DEV)
    res = {}
    if pin:
        hip.set_option("fr_class_rows", 32 * 1024)
    try:
        for flag in (16, 24):
            hip.set_option("train_flags", flag)
            m = _build(cfg, 13).eval()
            F.mse_loss(m(x, text, t), target).backward()
            res[flag] = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    finally:
        hip.set_option("train_flags", 0)
        if pin:
            hip.set_option("fr_class_rows", 0)
    differs = 0
    for n, g in res[16].items():
        assert torch.isfinite(g).all(), n
        assert rel_l2(g, res[24][n]) < 4e-3, (n, rel_l2(g, res[24][n]))
        differs += int(not torch.equal(g, res[24][n]))
    assert differs > 10, "train_flags 8 did not switch the path"


def test_training_step_on_the_bf16_residual_stream():
    """With the kernel class of the timed step pinned (32 x 1024 rows: both fused launches on the 128-row full-row kernel), the
    training forward keeps h as bf16 rows in the tape — AdaLN + block 0's norm1 from one kernel, the self-attention epilogue
    updating the bf16 row and leaving O beside it for the backward, the two full-row launches reading / writing bf16, the
    LayerNorm backward reading bf16 x — exactly the inference forward's stream.  train_flags 16 = the
    fp32 tape.  Both against fp32 autograd of the oracle (3e-2 per tensor; the stream costs the forward 3.4e-3 -> 7e-3 and the
    gradients about as much), against each other (2e-2), and the switch is live."""
    from oracle.checks import train_grad_parity
    got = {}
    for flag in (0, 16):
        hip.set_option("train_flags", flag)
        try:
            got[flag] = train_grad_parity(DEV, train_mode=True, pin_class=True)
        finally:
            hip.set_option("train_flags", 0)
        r = got[flag]
        assert not r["unexpected"] and r["full_row_forward"] == [True, True], r
        assert r["out_rel_l2"] < 2e-2 and r["worst_rel_l2"] < r["tol"], (flag, r)
    assert got[0]["out_rel_l2"] != got[16]["out_rel_l2"], "train_flags 16 did not switch the stream"
    cfg = DiTTOConfig(768, 2, 12, 256, 768, 50)
    x, text, t = (z.to(DEV) for z in synthetic_inputs(cfg, 2, 256, 192, seed=17))
    target = hash_normal((2, 256, 768), "noise", 18).to(DEV)
    res = {}
    hip.set_option("fr_class_rows", 32 * 1024)
    try:
        for flag in (0, 16):
            hip.set_option("train_flags", flag)
            m = _build(cfg, 14).train()
            torch.manual_seed(5)
            F.mse_loss(m(x, text, t), target).backward()
            res[flag] = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    finally:
        hip.set_option("train_flags", 0)
        hip.set_option("fr_class_rows", 0)
    for n, g in res[0].items():
        assert torch.isfinite(g).all(), n
        assert rel_l2(g, res[16][n]) < 2e-2, (n, rel_l2(g, res[16][n]))


def test_multi_step_training_on_the_bf16_tape_tracks_the_fp32_oracle():
    """the bf16 residual stream of the training tape had only been checked on ONE step's gradients.  Here the
    reference's loop (src/TrainDiTTO.py:55-95: q_sample -> forward -> MSE -> backward -> AdamW step) runs 12 steps at the timed
    width (d = 768, 12 heads, 2 layers, 2 x 256 frames, text 192) with the kernel class of the timed step pinned (32 x 1024 rows:
    full-row forward, bf16 tape) three times: HIP on the bf16 tape (default), HIP on the fp32 tape (train_flags 16) and fp32
    autograd of the oracle on the CPU, all from the same weights, data and AdamW settings.  Stated bars: every step's loss within
    5e-3 of the oracle's (measured 1e-4); the accumulated parameter UPDATE (theta_12 - theta_0, all tensors concatenated) within
    8e-2 rel-L2 of the oracle's for either tape (measured: bf16 tape 3.50e-2, fp32 tape 3.43e-2 — AdamW's normalised step amplifies
    gradient noise where |g| is tiny, bf16 OPERANDS alone cost this much), and the bf16 tape no further from the oracle than
    1.25 x the fp32 tape + 0.01 — the stream adds no drift of its own."""
    cfg = DiTTOConfig(768, 2, 12, 256, 768, 50)
    B, N, T, steps, lr = 2, 256, 192, 12, 5e-4
    data = []
    for i in range(steps):
        data.append((hash_normal((B, N, 768), f"x0{i}", 31), hash_normal((B, N, 768), f"nz{i}", 32),
                     hash_normal((B, T, 768), f"tx{i}", 33), torch.tensor([5 + 3 * i, 44 - 2 * i])))
    sd0 = synthetic_state_dict(cfg, 21)
    # the oracle's loop
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd0.items()}
    live = {k: v for k, v in sd.items() if v.requires_grad and k != "alphas_cumprod"}
    opt_o = torch.optim.AdamW(list(live.values()), lr=lr)
    losses_o = []
    for x0, noise, text, t in data:
        xt = O.q_sample(sd["alphas_cumprod"], x0, t, noise)
        loss = F.mse_loss(O.ditto_forward(sd, cfg.num_layers, cfg.num_heads, xt, text, t), noise)
        opt_o.zero_grad(); loss.backward(); opt_o.step()
        losses_o.append(float(loss))
    runs = {}
    with hip.batch_class(32 * 1024):
        for flag in (0, 16):
            hip.set_option("train_flags", flag)
            try:
                m = _build(cfg, 21).eval()           # eval: no dropout, the three loops see the same function
                assert hip.full_row_plan(cfg, B, N) == (True, True)
                opt = torch.optim.AdamW([p for _, p in m.named_parameters()], lr=lr)
                losses = []
                for x0, noise, text, t in data:
                    xt = m.q_sample(x0.to(DEV), t.to(DEV), noise.to(DEV))
                    loss = F.mse_loss(m(xt, text.to(DEV), t.to(DEV)), noise.to(DEV))
                    opt.zero_grad(); loss.backward(); opt.step()
                    losses.append(float(loss))
                runs[flag] = (losses, {n: p.detach().float().cpu().clone() for n, p in m.named_parameters()})
                del m, opt
            finally:
                hip.set_option("train_flags", 0)
    dist = {}
    for flag, (losses, params) in runs.items():
        for a, b in zip(losses, losses_o):
            assert abs(a - b) < 5e-3 * b, (flag, losses, losses_o)
        num = den = 0.0
        for n, p in params.items():
            if n not in live or sd0[n].shape != p.shape:
                continue
            du_o = live[n].detach() - sd0[n]
            if float(du_o.abs().max()) == 0.0:       # (a parameter the function does not depend on: no update to compare)
                continue
            num += float(((p - sd0[n]) - du_o).pow(2).sum()); den += float(du_o.pow(2).sum())
        dist[flag] = (num / den) ** 0.5
    print(f"12 AdamW steps, d=768 2L: losses oracle {losses_o[0]:.4f} -> {losses_o[-1]:.4f}, bf16 tape {runs[0][0][0]:.4f} -> {runs[0][0][-1]:.4f}; "
          f"update rel-L2 vs oracle: bf16 tape {dist[0]:.3e}, fp32 tape {dist[16]:.3e}")
    assert losses_o[-1] < losses_o[0] and runs[0][0][-1] < runs[0][0][0]
    assert dist[0] < 8e-2 and dist[16] < 8e-2, dist
    assert dist[0] < 1.25 * dist[16] + 0.01, dist
    assert runs[0][1].keys() == runs[16][1].keys() and any(not torch.equal(runs[0][1][n], runs[16][1][n]) for n in runs[0][1])


def test_large_batch_training_step_takes_the_full_row_forward():
    """From 160 row tiles on (B >= 20 at N = 1024) the training forward runs the cross out-projection + norm3 and fc2 + the
    next block's norm1 on the full-row kernel (csrc/gemm_fr.hip), its LayerNorm outputs landing in the tape slots the
    backward reads, and the backward its two long-K dgrads (fc1|gate, QKV) on the same kernel.  C2 (12 layers, d = 768) at B = 20: loss and every parameter gradient of the fused step against the
    unfused one (fr_mask 0) within bf16-path noise — the small-shape tests above pin the unfused step to the oracle."""
    from ditto_tts_amd.config import PRESETS
    cfg = PRESETS["C2"]["cfg"]
    x, text, t = (z.to(DEV) for z in synthetic_inputs(cfg, 20, 1024, 1024, seed=3))
    target = hash_normal((20, 1024, cfg.hidden_dim), "noise", 4).to(DEV)
    res = []
    for mask in (3, 0):
        hip.set_option("fr_mask", mask)
        hip.set_option("fr_dgrad", 3 if mask else 0)     # ... and the long-K dgrads of the backward on the same kernel
        try:
            m = _build(cfg, 9).eval()                    # eval: no dropout, both runs see the same function
            loss = F.mse_loss(m(x, text, t), target)
            loss.backward()
            res.append((float(loss), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
            del m, loss
        finally:
            hip.set_option("fr_mask", 3)
            hip.set_option("fr_dgrad", 3)
    (la, ga), (lb, gb) = res
    assert abs(la - lb) < 2e-3 * abs(lb), (la, lb)
    assert ga.keys() == gb.keys() and len(ga) > 200
    differs = 0
    for n in ga:
        assert torch.isfinite(ga[n]).all(), n
        assert rel_l2(ga[n], gb[n]) < 3e-2, (n, rel_l2(ga[n], gb[n]))
        differs += int(not torch.equal(ga[n], gb[n]))
    assert differs > 100, "the full-row forward did not run"


def test_training_surface_contract():
    cfg = DiTTOConfig(128, 1, 2, 64, 128, 20)
    m = _build(cfg, 1).train()
    x, text, t = (z.to(DEV) for z in synthetic_inputs(cfg, 1, 16, 8))
    with pytest.raises(NotImplementedError, match="x / text_emb"):
        m(x.clone().requires_grad_(True), text, t)
    out = m(x, text, t)
    out.sum().backward()
    with pytest.raises(RuntimeError):
        out.sum().backward()                            # the tape was released
    # frozen model: plain inference path even with grad mode on
    for p in m.parameters():
        p.requires_grad_(False)
    assert not m(x, text, t).requires_grad
    # gradient accumulation over two micro-batches == sum of the two gradients
    m2 = _build(cfg, 1).eval()
    a = m2(x, text, t).square().mean(); a.backward()
    g1 = m2.proj_out.weight.grad.clone()
    b = m2(x * 0.5, text, t).square().mean(); b.backward()
    m3 = _build(cfg, 1).eval()
    m3(x * 0.5, text, t).square().mean().backward()
    assert torch.allclose(m2.proj_out.weight.grad, g1 + m3.proj_out.weight.grad, rtol=1e-5, atol=1e-7)


def test_backward_reads_the_tape_as_the_forward_wrote_it():
    """the tape's residual-stream rows are bf16 or fp32 by a decision the FORWARD takes from the options in
    force around it; the backward used to take the same decision again from whatever the options were by then, and a change in
    between (`with hip.batch_class(rows): out = m(x)` ... `loss.backward()` outside, on autograd's own thread) read a bf16 tape
    as fp32: garbage gradients, no error.  Now the forward records its decision against the tape (ditto_model::tapes) and the
    autograd Function carries the forward's options to the backward: the gradients are the SAME BITS whether the scope is
    still open at backward time or not, whether process-wide switches moved in between or not."""
    cfg = DiTTOConfig(768, 2, 12, 256, 768, 50)
    B, N, T = 2, 256, 192
    x, text, t = (z.to(DEV) for z in synthetic_inputs(cfg, B, N, T, seed=21))
    target = hash_normal((B, N, 768), "noise", 22).to(DEV)

    def run(close_scope_first, meddle):
        m = _build(cfg, 15).eval()
        scope = hip.batch_class(32 * 1024)
        scope.__enter__()
        try:
            assert hip.stream_is_bf16(cfg, B, N)                  # the pinned class carries the bf16 stream (and tape)
            loss = F.mse_loss(m(x, text, t), target)
        finally:
            if close_scope_first:
                scope.__exit__(None, None, None)
        try:
            if meddle:                                            # process-wide switches that used to re-decide the tape's type
                hip.set_option("train_flags", 16)
                hip.set_option("residual_bf16", 0)
            loss.backward()
        finally:
            hip.set_option("train_flags", 0)
            hip.set_option("residual_bf16", 1)
            if not close_scope_first:
                scope.__exit__(None, None, None)
        return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    ref = run(False, False)
    for variant in ((True, False), (True, True)):
        got = run(*variant)
        assert got.keys() == ref.keys()
        for n, g in ref.items():
            assert torch.isfinite(got[n]).all(), (variant, n)
            assert torch.equal(got[n], g), (variant, n, rel_l2(got[n], g))
    # ... and the C-ABI refuses a tape no forward of the handle wrote, or one written for another shape
    m = _build(cfg, 15).eval()
    eng = m.engine(DEV, train=True)
    sd = {k: v for k, v in m.state_dict(keep_vars=True).items() if not k.startswith("nac.")}
    out, tape, xf, tt = eng.train_forward(x, text, t, 0.0, 0)
    stray = torch.empty_like(tape)
    with pytest.raises(hip.DittoHipError, match="no ditto_train_forward of this handle wrote the tape"):
        eng.train_backward(sd, torch.ones_like(out), xf, tt, T, stray, 0.0, 0)
    with pytest.raises(hip.DittoHipError, match="written for"):
        eng.train_backward(sd, torch.ones_like(out)[:1], xf[:1].contiguous(), tt[:1].contiguous(), T, tape, 0.0, 0)


@pytest.mark.parametrize("per_piece", [1, 2, 5])
def test_backward_in_layer_pieces_is_bitwise_the_single_call(per_piece):
    """ditto_train_backward_layers (the backward as successive calls over layer ranges, top first: what lets the data-parallel
    gradient exchange of the upper layers overlap the computation of the lower ones, dist.GradSync) == ditto_train_backward, bit
    for bit, for every piece size; the callback sees every gradient tensor exactly once, in backward order."""
    cfg = DiTTOConfig(256, 5, 4, 64, 256, 20)
    B, N, T = 2, 96, 40
    x, text, t = (z.to(DEV) for z in synthetic_inputs(cfg, B, N, T, seed=31))
    m = _build(cfg, 17).train()
    eng = m.engine(DEV, train=True)
    sd = {k: v for k, v in m.state_dict(keep_vars=True).items() if not k.startswith("nac.")}
    gout = hash_normal((B, N, 256), "gout", 32).to(DEV)
    out, tape, xf, tt = eng.train_forward(x, text, t, 0.1, 1234)
    whole = eng.train_backward(sd, gout, xf, tt, T, tape, 0.1, 1234)
    seen = []
    pieces = eng.train_backward(sd, gout, xf, tt, T, tape, 0.1, 1234, piece_cb=lambda ts: seen.append([z.data_ptr() for z in ts]),
                                layers_per_piece=per_piece)
    assert len(seen) == -(-5 // per_piece)
    flat = [p for piece in seen for p in piece]
    assert sorted(flat) == sorted(g.data_ptr() for g in pieces.values()) and len(set(flat)) == len(flat)
    assert seen[0][0] == pieces["proj_in.weight"].data_ptr() and seen[-1][-1] == pieces["ada_ln.text_mlp.1.bias"].data_ptr()
    for k, g in whole.items():
        assert torch.equal(pieces[k], g), k
