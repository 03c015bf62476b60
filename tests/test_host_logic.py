"""Host-side logic that needs no GPU: key set, synthetic recipe, FLOP model, loud failures."""
import os
import sys

import pytest
import torch

from conftest import REFERENCE_SRC
from ditto_tts_amd.config import PRESETS, DiTTOConfig
from ditto_tts_amd.modules import DiT, DiTTO, GlobalAdaLN, RotaryEmbedding, _ParamWatch
from ditto_tts_amd.synth import (cosine_betas, expected_state_shapes, hash_normal, hash_uniform,
                                 synthetic_inputs, synthetic_state_dict)
from ditto_tts_amd.dist import shard_bounds


def test_state_dict_keys_match_reference_list():
    cfg = DiTTOConfig(256, 2, 4, 256, 256, 50)
    m = DiTTO(256, 2, 4, 256, 256, 50)
    exp = expected_state_shapes(cfg)
    sd = m.state_dict()
    assert list(sd.keys()) == list(exp.keys())
    for k, shp in exp.items():
        assert tuple(sd[k].shape) == shp, k
    # dead parameters and per-block buffers are part of the on-disk format
    assert "blocks.0.attn.out_proj.weight" in sd and "blocks.1.rotary.inv_freq" in sd
    assert m.load_state_dict(synthetic_state_dict(cfg)).missing_keys == []
    # App. B-1: the buffer called alphas_cumprod holds the clipped betas
    assert torch.equal(m.alphas_cumprod, cosine_betas(50))


@pytest.mark.skipif(not os.path.isdir(REFERENCE_SRC), reason="reference tree not present")
def test_state_dict_keys_match_live_reference():
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_golden as mg
    _, ref_ditto = mg.import_reference()
    cfg = DiTTOConfig(128, 2, 2, 64, 128, 20)
    ref = mg.build_reference_ditto(ref_ditto, cfg, synthetic_state_dict(cfg))
    ours = DiTTO(128, 2, 2, 64, 128, 20)
    ref_keys = [k for k in ref.state_dict() if not k.startswith("nac.")]
    assert ref_keys == list(ours.state_dict().keys())
    # a reference checkpoint loads straight into the facade
    ours.load_state_dict({k: v for k, v in ref.state_dict().items() if not k.startswith("nac.")})


def test_cpu_inputs_fail_loudly_no_fallback():
    m = DiTTO(128, 1, 2, 64, 128, 10)
    x, text, t = torch.zeros(1, 8, 128), torch.zeros(1, 8, 128), torch.zeros(1, dtype=torch.long)
    with pytest.raises(RuntimeError, match="no CPU"):
        m(x, text, t)
    with pytest.raises(RuntimeError, match="no CPU"):
        DiT(128, 2, 64, 128)(x, text, None, torch.zeros(8, 64))
    with pytest.raises(RuntimeError, match="no CPU"):
        GlobalAdaLN(128, 64, 128)(x, torch.zeros(1, 64), text)
    with pytest.raises(RuntimeError, match="no CPU"):
        m.q_sample(x, t)


def test_missing_library_fails_loudly(monkeypatch):
    from ditto_tts_amd import hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", "/nonexistent/libditto_hip.so")
    with pytest.raises(RuntimeError, match="only compute path"):
        hip.lib()


def test_rotary_table_matches_reference_formula():
    r = RotaryEmbedding(64)
    tab = r(10, "cpu")
    assert tab.shape == (10, 64) and torch.equal(tab[:, :32], tab[:, 32:])
    assert torch.allclose(tab[3, 5], torch.tensor(3.0) / 10000 ** (10 / 64))


def test_param_watch_sees_inplace_updates():
    p = torch.nn.Parameter(torch.zeros(4))
    w = _ParamWatch()
    assert w.changed([p]) and not w.changed([p])
    with torch.no_grad():
        p.add_(1.0)
    assert w.changed([p])


def test_synth_is_deterministic_and_full_rank():
    a = hash_uniform((64, 64), "w", 1)
    assert (a == hash_uniform((64, 64), "w", 1)).all() and not (a == hash_uniform((64, 64), "w", 2)).all()
    assert -1.0 <= a.min() and a.max() < 1.0
    import numpy as np
    assert np.linalg.matrix_rank(a) == 64
    z = hash_normal((4096,), "z", 3)
    assert abs(float(z.mean())) < 0.05 and abs(float(z.std()) - 1.0) < 0.05
    x, text, t = synthetic_inputs(DiTTOConfig(128, 1, 2, 64, 128, 20), 3, 8, 5)
    assert x.shape == (3, 8, 128) and text.shape == (3, 5, 128) and t.dtype == torch.int64
    assert int(t.max()) < 20 and int(t.min()) >= 0


def test_flop_model_matches_survey():
    c2 = PRESETS["C2"]["cfg"]
    assert abs(c2.flops_per_utt_step(1024, 1024, cached_kv=False) / 1e9 - 355.14) < 0.01
    assert abs(c2.flops_per_utt_step(1024, 1024, cached_kv=True) / 1e9 - 326.15) < 0.01
    assert abs(PRESETS["C4"]["cfg"].flops_per_utt_step(4096, 1024, False) / 1e9 - 1797.45) < 0.01
    assert abs(PRESETS["C5"]["cfg"].flops_per_utt_step(1024, 1024, False) / 1e9 - 1189.71) < 0.01


def test_shard_bounds_cover_exactly():
    for total in (0, 1, 7, 32, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_slp_surface_matches_torch_decoder_keys_and_has_no_cpu_path():
    """The SLP mirror (src/model/SpeechLP.py): same parameter keys / shapes as the nn.TransformerDecoder + Linear the
    reference builds, reference import name, loud failure off-GPU and without the pretrained encoders."""
    import torch.nn as nn
    from ditto_tts_amd.compat.model.SpeechLP import SLP as SLPcompat
    from ditto_tts_amd.shipped_config import ConfigSLP
    from ditto_tts_amd.slp import SLP
    from ditto_tts_amd.synth import slp_state_shapes
    assert SLPcompat is SLP
    d, nhead, nl, ncls = 128, 4, 2, ConfigSLP.NB_CLASSES
    m = SLP(ncls, nhead, nl, hidden_size=d)
    ref = nn.ModuleDict({
        "transformer": nn.TransformerDecoder(nn.TransformerDecoderLayer(d_model=d, nhead=nhead, dim_feedforward=d * nhead,
                                                                       batch_first=True), num_layers=nl),
        "length_predictor": nn.Linear(d, ncls)})
    want = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == want
    assert dict(slp_state_shapes(d, nhead, nl, ncls)) == want
    assert ConfigSLP.NB_CLASSES == 11 and ConfigSLP.EMBEDDING_DIM == 1472          # src/utils/Config.py:74,77
    m.eval()
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.decode(torch.zeros(1, 4, d), torch.zeros(1, 6, d))
    with pytest.raises(RuntimeError, match="pretrained encoders"):
        m("text", torch.zeros(1, 100))
    # hidden size taken from the injected text encoder, as the reference does (:18)
    class Enc(nn.Module):
        def __init__(self):
            super().__init__()
            self.model = type("M", (), {"config": type("C", (), {"d_model": 192})()})()
    assert SLP(5, 1, 1, text_encoder=Enc()).hidden_size == 192


def test_bench_refuses_to_mislabel_a_multi_gpu_run():
    """`bench.py --gpus N` must never print a 1-GPU number as an N-GPU one: without WORLD_SIZE it
    becomes a launcher that needs N visible devices; under a launcher WORLD_SIZE must equal --gpus.  Both exits are
    non-zero and happen before any GPU call, so they are checkable on this GPU-less host."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True,
                           text=True, env=env, timeout=300)
        assert r.returncode != 0 and "refusing to measure fewer" in r.stderr and r.stdout.strip() == ""
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr and r.stdout.strip() == ""
