"""`model.DiTTO` under the reference import name (reference src/model/DiTTO.py)."""
from ditto_tts_amd.modules import DiTTO  # noqa: F401
