"""Import-name aliases for the reference tree.

`compat/model`, `compat/components` are NAMESPACE-package portions (no `__init__.py`): with this directory on
`sys.path` ahead of the reference's `src/`, `model.DiTTO`, `model.SpeechGenerator`, `model.SpeechLP`,
`components.DiT` and `components.VectorQuantizer` resolve to this package's classes, and every other submodule
(`model.NeuralAudioCodec`, `components.EnCodec`, `utils.Config`, `utils.MLS`, `utils.Trainer`, ...) falls through to
the reference's own files.  `utils` is not touched at all.

    import ditto_tts_amd.compat as compat; compat.install()          # notebooks / interactive
    python -m ditto_tts_amd.run_reference src/TrainDiTTO.py           # unmodified scripts
"""
import os
import sys

COMPAT_DIR = os.path.dirname(os.path.abspath(__file__))
ALIASED = ("model.DiTTO", "model.SpeechGenerator", "model.SpeechLP", "components.DiT", "components.VectorQuantizer")


def install(reference_src=None):
    """Put the alias directory at the FRONT of sys.path (and `reference_src` right behind it when given), and
    forget already-imported `model` / `components` modules so the next import re-resolves through it."""
    for p in (reference_src, COMPAT_DIR):
        if p is None:
            continue
        p = os.path.abspath(p)
        while p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    for name in list(sys.modules):
        if name in ("model", "components") or name in ALIASED:
            del sys.modules[name]
