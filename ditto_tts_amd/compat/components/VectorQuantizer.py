"""`components.VectorQuantizer` under the reference import name (reference src/components/VectorQuantizer.py)."""
from ditto_tts_amd.around import VectorQuantizer  # noqa: F401
