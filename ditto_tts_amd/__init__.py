"""ditto_tts_amd — MI355X-native DiT denoise path for DiTTo-TTS.

`compat/` exposes the classes under the reference's import names as namespace-package portions: with
`ditto_tts_amd/compat` ahead of the reference's `src/` on sys.path (`compat.install()` / `python -m
ditto_tts_amd.run_reference script.py`) `from model.DiTTO import DiTTO` resolves here while `utils.*`,
`model.NeuralAudioCodec`, ... stay the reference's."""
from .config import DiTTOConfig, PRESETS  # noqa: F401

__all__ = ["DiTTOConfig", "PRESETS", "DiTTO", "DiT", "GlobalAdaLN", "RotaryEmbedding", "SpeechGenerator",
           "DenoiseEngine"]


def __getattr__(name):  # lazy: importing the package must not require torch.cuda or the built library
    if name in ("DiTTO", "DiT", "GlobalAdaLN", "RotaryEmbedding"):
        from . import modules
        return getattr(modules, name)
    if name == "SpeechGenerator":
        from .sampler import SpeechGenerator
        return SpeechGenerator
    if name == "DenoiseEngine":
        from .engine import DenoiseEngine
        return DenoiseEngine
    raise AttributeError(name)
