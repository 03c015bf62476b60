"""`components.DiT` under the reference import name (reference src/components/DiT.py)."""
from ditto_tts_amd.modules import DiT, GlobalAdaLN, RotaryEmbedding  # noqa: F401
