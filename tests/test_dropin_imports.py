"""The drop-in recipe of INTEGRATION.md §2, exercised with the reference's OWN import lines (CPU, skips without the
reference tree).  Each case runs in a fresh interpreter so sys.path / sys.modules start clean.  Only what is absent
from this image is stubbed: torchaudio, the BigVGAN checkout, `evaluate`."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import REFERENCE_SRC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(REFERENCE_SRC), reason="reference tree not present")

STUBS = """
import sys, types, importlib.machinery
import transformers                                   # before the stubs: its availability probes must see no torchaudio
from transformers import AutoProcessor, AutoTokenizer  # noqa: F401
for name in ("torchaudio", "torchaudio.transforms", "bigvgan_v2_24khz_100band_256x",
             "bigvgan_v2_24khz_100band_256x.bigvgan", "bigvgan_v2_24khz_100band_256x.meldataset", "evaluate"):
    m = types.ModuleType(name); m.__path__ = []; m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    sys.modules[name] = m
sys.modules["bigvgan_v2_24khz_100band_256x"].bigvgan = sys.modules["bigvgan_v2_24khz_100band_256x.bigvgan"]
sys.modules["bigvgan_v2_24khz_100band_256x.meldataset"].get_mel_spectrogram = lambda *a, **k: None
sys.modules["evaluate"].load = lambda *a, **k: None
"""


def run(body, cwd=None):
    code = STUBS + textwrap.dedent(body)
    env = dict(os.environ, PYTHONPATH="")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=cwd or ROOT, env=env,
                       timeout=600)
    assert r.returncode == 0, f"stdout:\n{r.stdout[-3000:]}\nstderr:\n{r.stderr[-3000:]}"
    return r.stdout


def test_traindito_import_block_under_the_recipe():
    """sys.path exactly as INTEGRATION.md §2, then reference src/TrainDiTTO.py:1-11 verbatim (read from the reference
    tree at test time), then the constructor call of :41-49 with the codec escape."""
    out = run(f"""
        import sys
        sys.path.insert(0, {REFERENCE_SRC!r})                      # the caller's tree (scripts run from src/)
        sys.path.insert(0, {ROOT!r})                               # the package
        sys.path.insert(0, {os.path.join(ROOT, "ditto_tts_amd", "compat")!r})
        lines = open({os.path.join(REFERENCE_SRC, "TrainDiTTO.py")!r}).read().splitlines()[:11]
        src = "\\n".join(lines)
        assert "from utils.Trainer import Trainer" in src and "ConfigDiTTO.display()" in src
        exec(compile(src, "TrainDiTTO.py:1-11", "exec"))
        import ditto_tts_amd.modules as M, utils.Config, utils.Trainer, utils.MLS
        assert DiTTO is M.DiTTO, DiTTO
        assert utils.Config.__file__.startswith({REFERENCE_SRC!r}), utils.Config.__file__
        assert utils.Trainer.__file__.startswith({REFERENCE_SRC!r}) and utils.MLS.__file__.startswith({REFERENCE_SRC!r})
        assert ConfigNAC.LAMBDA_FACTOR == 0.1
        # modules.py's own `from model.NeuralAudioCodec import NAC` (reference src/model/DiTTO.py:4) resolves too
        import model.NeuralAudioCodec as NACmod
        assert NACmod.__file__.startswith({REFERENCE_SRC!r})
        from components.DiT import DiT, GlobalAdaLN, RotaryEmbedding
        assert DiT is M.DiT and GlobalAdaLN is M.GlobalAdaLN
        from components.VectorQuantizer import VectorQuantizer
        import ditto_tts_amd.around as A
        assert VectorQuantizer is A.VectorQuantizer
        import components.EnCodec as E
        assert E.__file__.startswith({REFERENCE_SRC!r})
        m = DiTTO(hidden_dim=ConfigDiTTO.HIDDEN_DIM, num_layers=1, num_heads=ConfigDiTTO.NUM_HEADS,
                  time_dim=ConfigDiTTO.TIME_DIM, text_dim=ConfigDiTTO.TEXT_EMBED_DIM, diffusion_steps=20,
                  nac_model_path=None)
        t = Trainer().set_model(m, name=ConfigDiTTO.MODEL_NAME)
        print("ok-train")
    """)
    assert "DiTTO Settings" in out and "ok-train" in out


def test_speechgenerator_import_block_and_config_mutation():
    """reference src/model/SpeechGenerator.py:1-13 verbatim under the recipe; then the notebook's flow
    (src/Experiments.ipynb cells 1, 6): mutate ConfigDiTTO on the caller's module, build the sampler, and the loop
    length follows the caller's global at call time (reference :161)."""
    out = run(f"""
        import sys
        import ditto_tts_amd.compat as compat
        compat.install(reference_src={REFERENCE_SRC!r})
        lines = open({os.path.join(REFERENCE_SRC, "model", "SpeechGenerator.py")!r}).read().splitlines()[:13]
        src = "\\n".join(lines)
        assert "from model.SpeechLP import SLP" in src and "from utils.Config import ConfigDiTTO, ConfigSLP" in src
        exec(compile(src, "SpeechGenerator.py:1-13", "exec"))
        import ditto_tts_amd.modules as M, ditto_tts_amd.slp as S, ditto_tts_amd.sampler as P
        assert DiTTO is M.DiTTO and SLP is S.SLP
        from model.SpeechGenerator import SpeechGenerator
        assert SpeechGenerator is P.SpeechGenerator
        from utils.Config import ConfigSLP, ConfigNAC, ConfigDiTTO           # Experiments.ipynb cell 1
        ConfigSLP.display(); ConfigNAC.display(); ConfigDiTTO.display()      # cell 2
        ConfigDiTTO.DIFFUSION_STEPS = 40                                     # cell 6
        ConfigDiTTO.NUM_LAYERS = 1
        sg = SpeechGenerator(lambda_factor=ConfigNAC.LAMBDA_FACTOR, nac_model_path=None, ditto_model_path=None,
                             slp_path=None, sample_rate=ConfigNAC.SAMPLE_RATE, device="cpu")
        assert sg.ditto_model.cfg.diffusion_steps == 40 and len(sg.betas) == 40 and sg._loop_steps() == 40
        ConfigDiTTO.DIFFUSION_STEPS = 25                  # mutated after construction: read at call time
        assert sg._loop_steps() == 25
        ConfigDiTTO.DIFFUSION_STEPS = 1000
        try:
            sg._loop_steps(); raise SystemExit("expected IndexError")
        except IndexError:
            pass
        pinned = SpeechGenerator(ditto_model=sg.ditto_model, device="cpu", diffusion_steps=10)
        assert pinned._loop_steps() == 10
        print("ok-sampler")
    """)
    assert "ok-sampler" in out


def test_unmodified_script_through_the_launcher(tmp_path):
    """`python -m ditto_tts_amd.run_reference <script>`: a script that sits in the reference's src/ layout and uses
    the reference's import lines runs unchanged (sys.path[0] = its directory would otherwise make src/model/DiTTO.py
    win).  The script here is a 6-line stand-in placed in a COPY-FREE overlay: a temp dir whose model/, components/,
    utils/ are symlinks to the reference's."""
    src = tmp_path / "src"
    src.mkdir()
    for d in ("model", "components", "utils"):
        os.symlink(os.path.join(REFERENCE_SRC, d), src / d)
    (src / "script.py").write_text(textwrap.dedent("""
        from model.DiTTO import DiTTO
        from utils.Config import ConfigDiTTO, ConfigNAC
        from utils.Trainer import Trainer
        import model.DiTTO as md, sys
        print("DITTO_FROM", md.__file__)
        print("ARGS", sys.argv[1:])
    """))
    code = STUBS + f"import sys; sys.path.insert(0, {ROOT!r}); sys.argv = ['x', {str(src / 'script.py')!r}, '--flag']\n" \
                   "from ditto_tts_amd.run_reference import main; main(sys.argv[1:])\n"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert os.path.join("ditto_tts_amd", "compat", "model", "DiTTO.py") in r.stdout and "ARGS ['--flag']" in r.stdout
