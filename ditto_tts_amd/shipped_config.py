"""The reference's shipped hyper-parameters (reference src/utils/Config.py:69-82,101-128), used ONLY when the
caller's own `utils.Config` is not importable (stand-alone use of this package).  When the reference tree is on
`sys.path` — the drop-in case — `caller_config()` returns the caller's module, so a notebook's
`ConfigDiTTO.DIFFUSION_STEPS = 1000` (reference src/Experiments.ipynb cell 6) mutates the object the sampler reads,
exactly as in the reference.  Nothing here shadows `utils.Config`: `ConfigNAC`, the dataset paths and the
`display()` helpers stay the reference's."""
import importlib
import sys

import torch


class BaseConfig:
    SAMPLE_RATE = 24000
    MIN_AUDIO_DURATION = 10
    MAX_AUDIO_DURATION = 20
    DEVICE = "cuda" if torch.cuda.is_available() else "cpu"
    BETAS = [0.9, 0.999]


class ConfigDiTTO(BaseConfig):
    MODEL_NAME = "DiTTO"
    HIDDEN_DIM = 768
    NUM_LAYERS = 5
    NUM_HEADS = 1
    TIME_DIM = 256
    TEXT_EMBED_DIM = 768
    DIFFUSION_STEPS = 1000
    EPOCHS = 20
    LEARNING_RATE = 1e-4
    BATCH_SIZE = 8
    MAX_TOKEN_LENGTH = 1024
    NB_SAMPLES = 10000


class ConfigSLP(BaseConfig):
    """The attributes `SpeechGenerator` reads to build the length predictor (reference src/utils/Config.py:69-82,
    src/model/SpeechGenerator.py:54-58): ONE layer, ONE head over byt5-small's 1472-wide embeddings."""
    MODEL_NAME = "SLP"
    EMBEDDING_DIM = 1472
    NUM_LAYERS = 1
    NUM_HEADS = 1
    NB_CLASSES = int(BaseConfig.MAX_AUDIO_DURATION - BaseConfig.MIN_AUDIO_DURATION + 1)
    EPOCHS = 20
    LEARNING_RATE = 1e-4
    BATCH_SIZE = 8
    NB_SAMPLES = 10000
    MAX_TOKEN_LENGTH = 128


def caller_config(import_it: bool = True):
    """The module object the CALLER uses as `utils.Config` (already imported, or importable from its tree), or None.
    Never this file: the point is to share the caller's mutable class attributes."""
    mod = sys.modules.get("utils.Config")
    if mod is None and import_it:
        try:
            mod = importlib.import_module("utils.Config")
        except ImportError:
            mod = None
    return mod if mod is not None and hasattr(mod, "ConfigDiTTO") else None


def config_classes():
    """(ConfigDiTTO, ConfigSLP): the caller's when its tree is importable, else the shipped values above."""
    mod = caller_config()
    if mod is not None:
        return mod.ConfigDiTTO, getattr(mod, "ConfigSLP", ConfigSLP)
    return ConfigDiTTO, ConfigSLP
