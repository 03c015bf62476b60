"""`model.SpeechGenerator` under the reference import name (reference src/model/SpeechGenerator.py)."""
from ditto_tts_amd.sampler import SpeechGenerator  # noqa: F401
