"""`utils.Config` under the reference's import name (reference src/utils/Config.py): the attribute bag the
scripts and notebooks read (`ConfigDiTTO.HIDDEN_DIM`, `.DIFFUSION_STEPS`, ...), with the shipped values
(5 layers, ONE head).  Only the attributes; the dataset paths and printing helpers of the
reference are not part of the denoise path."""
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
