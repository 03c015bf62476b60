"""Shape configuration of the DiT denoise path.

The reference has no config object for the model: `DiTTO.__init__` takes keyword
arguments (reference src/model/DiTTO.py:10-19) and the scripts read a static
class `ConfigDiTTO` (reference src/utils/Config.py:102-121).  This dataclass is
the shape tuple the HIP library needs; `shipped_config.py` holds the reference's
shipped attribute values for stand-alone use (the caller's own `utils.Config` is never shadowed).
"""
from __future__ import annotations

from dataclasses import dataclass, asdict


@dataclass(frozen=True)
class DiTTOConfig:
    hidden_dim: int = 768
    num_layers: int = 12
    num_heads: int = 12
    time_dim: int = 256
    text_dim: int = 768
    diffusion_steps: int = 1000
    fp8_linear: bool = False   # QKV / fc1|gate / fc2 GEMMs with fp8 e4m3 operands (BASELINE config 5)

    def __post_init__(self):
        if self.hidden_dim % self.num_heads:
            raise ValueError("hidden_dim must be divisible by num_heads")
        if self.text_dim != self.hidden_dim:
            # nn.MultiheadAttention is built without kdim/vdim in the reference
            # (src/components/DiT.py:90-91), so cross-attention requires this.
            raise ValueError("text_dim must equal hidden_dim (reference cross-attn has no kdim/vdim)")
        if (self.hidden_dim // self.num_heads) % 2:
            raise ValueError("head_dim must be even (half-split RoPE)")
        if self.fp8_linear and self.hidden_dim % 128:
            raise ValueError("fp8_linear needs hidden_dim % 128 == 0")

    @property
    def head_dim(self) -> int:
        return self.hidden_dim // self.num_heads

    def as_dict(self):
        return asdict(self)

    # ---- algorithmic FLOPs, 2 FLOP per MAC -----------------
    def flops_per_utt_step(self, N: int, T: int, cached_kv: bool = True) -> float:
        d, L = self.hidden_dim, self.num_layers
        per_layer = 34 * N * d * d + 4 * N * N * d + 4 * N * T * d
        if not cached_kv:
            per_layer += 4 * T * d * d
        return float(L * per_layer + 4 * N * d * d)

    def flops_text_kv(self, T: int) -> float:
        return float(4 * self.num_layers * T * self.hidden_dim * self.hidden_dim)


# BASELINE.json `configs`, in order (SURVEY.md §8 shorthand C1..C5).
PRESETS = {
    "C1": dict(cfg=DiTTOConfig(256, 1, 4, 256, 256, 50), B=1, N=64, T=64),
    "C2": dict(cfg=DiTTOConfig(768, 12, 12, 256, 768, 50), B=32, N=1024, T=1024),
    "C3": dict(cfg=DiTTOConfig(768, 12, 12, 256, 768, 50), B=32, N=1024, T=1024),  # per GPU of 8
    "C4": dict(cfg=DiTTOConfig(768, 12, 12, 256, 768, 50), B=8, N=4096, T=1024),
    "C5": dict(cfg=DiTTOConfig(1024, 24, 16, 256, 1024, 50, fp8_linear=True), B=16, N=1024, T=1024),
    "C5_bf16": dict(cfg=DiTTOConfig(1024, 24, 16, 256, 1024, 50), B=16, N=1024, T=1024),
    # the shipped ConfigDiTTO of the reference (src/utils/Config.py:109-116): 5 layers, ONE head
    "shipped": dict(cfg=DiTTOConfig(768, 5, 1, 256, 768, 1000), B=1, N=64, T=64),
}
