"""Oracle == the live reference (imported from /root/reference/src).  Skipped where the reference
is absent (the GPU box); the committed goldens cover that case."""
import os
import sys

import pytest
import torch

from conftest import REFERENCE_SRC, rel_l2
from ditto_tts_amd.config import DiTTOConfig
from ditto_tts_amd.synth import synthetic_inputs, synthetic_state_dict
from oracle import ditto_oracle as O

pytestmark = pytest.mark.skipif(not os.path.isdir(REFERENCE_SRC), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_golden as mg
    ref_dit, ref_ditto = mg.import_reference()
    return mg, ref_dit, ref_ditto


@pytest.mark.parametrize("shape", [(256, 1, 4, 1, 64, 64), (128, 2, 2, 3, 40, 24), (192, 2, 1, 2, 33, 17)])
def test_forward_matches_reference(ref, shape):
    mg, _, ref_ditto = ref
    d, L, H, B, N, T = shape
    cfg = DiTTOConfig(d, L, H, 64, d, 20)
    sd = synthetic_state_dict(cfg, seed=9)
    m = mg.build_reference_ditto(ref_ditto, cfg, sd)
    x, text, t = synthetic_inputs(cfg, B, N, T, seed=3)
    with torch.no_grad():
        want = m(x, text, t)
    got = O.ditto_forward(sd, L, H, x, text, t)
    assert rel_l2(got, want) < 1e-5


def test_components_match_reference(ref):
    _, ref_dit, _ = ref
    rot = ref_dit.RotaryEmbedding(64)
    assert torch.equal(rot.inv_freq, O.rotary_inv_freq(64))
    assert torch.equal(rot(100, "cpu"), O.rotary_table(rot.inv_freq, 100))
    q = torch.randn(2, 100, 3, 64)
    assert torch.equal(rot.apply_rope(rot(100, "cpu"), q), O.apply_rope(O.rotary_table(rot.inv_freq, 100), q))


def test_train_mode_structure(ref):
    """G6: eval is deterministic, train (cross-attn dropout) is not; attn.out_proj is dead."""
    mg, _, ref_ditto = ref
    cfg = DiTTOConfig(128, 2, 2, 64, 128, 20)
    sd = synthetic_state_dict(cfg, seed=9)
    m = mg.build_reference_ditto(ref_ditto, cfg, sd)
    x, text, t = synthetic_inputs(cfg, 2, 32, 16, seed=3)
    with torch.no_grad():
        assert torch.equal(m(x, text, t), m(x, text, t))
    m.train()
    m(x, text, t).square().mean().backward()
    for name, p in m.named_parameters():
        if name.startswith("nac."):
            continue
        if ".attn.out_proj." in name:
            assert p.grad is None, name
        else:
            assert p.grad is not None, name
