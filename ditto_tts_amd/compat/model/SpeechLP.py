"""`model.SpeechLP` under the reference import name (reference src/model/SpeechLP.py)."""
from ditto_tts_amd.slp import SLP  # noqa: F401
