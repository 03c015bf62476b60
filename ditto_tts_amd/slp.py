"""Speech-length predictor (SLP) — the reference's `model.SpeechLP.SLP` surface on the HIP decoder stack.

Reference: src/model/SpeechLP.py.  The module keeps the reference's constructor arguments, attribute names and
state_dict keys (`transformer.layers.{i}.*`, `length_predictor.*`; `text_encoder.*` / `audio_encoder.*` when those
modules are injected), so `SpeechGenerator`'s `slp_info["model_state_dict"]` loads unchanged
(src/model/SpeechGenerator.py:54-62).  What runs on the GPU is everything after the two pretrained encoders
(src/model/SpeechLP.py:50-54): causal-mask TransformerDecoder over the audio embeddings with cross-attention to the text
embeddings, and `length_predictor` on the last position — ditto_slp_forward (csrc/slp.hip).

Out of scope: ByT5 and EnCodec themselves (pretrained HF checkpoints, no network).  They are injected
(`text_encoder=`, `audio_encoder=`); without them `forward(text, audio)` raises and `decode(z_text, z_audio)` is the
entry point.  There is no CPU path and no eager fallback: nn.TransformerDecoder here only OWNS the parameters."""
from __future__ import annotations

import ctypes as C
from typing import Mapping, Optional

import torch
import torch.nn as nn

from . import hip
from .modules import _ParamWatch


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class SlpEngine:
    """Host-side owner of one packed decoder stack on one GPU (ditto_slp_* of include/ditto_hip.h)."""

    def __init__(self, d_model: int, nhead: int, num_layers: int, dim_feedforward: int, num_classes: int,
                 state: Mapping[str, torch.Tensor], device):
        self.lib = hip.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("SlpEngine needs an MI355X (torch.cuda.is_available() is False); "
                               "ditto_tts_amd has no CPU path")
        self.device = torch.device(device)
        self.cfg = hip.SlpConfig(d_model, nhead, num_layers, dim_feedforward, num_classes)
        nbytes = self.lib.ditto_slp_arena_bytes(C.byref(self.cfg))
        if nbytes == 0:
            raise hip.DittoHipError(hip.ERR_SHAPE, self.lib.ditto_last_error().decode())
        self.arena = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self.handle = C.c_void_p()
        self._ws: Optional[torch.Tensor] = None
        self.repack(state)

    def repack(self, state: Mapping[str, torch.Tensor]):
        keep = []

        def dev(key):
            if key not in state:
                raise KeyError(f"state_dict is missing '{key}'")
            t = state[key].detach().to(device=self.device, dtype=torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()

        L = self.cfg.num_layers
        layers = (hip.SlpLayerWeights * L)()
        for l in range(L):
            for f, k in hip.SLP_LAYER_KEY.items():
                setattr(layers[l], f, dev(f"transformer.layers.{l}.{k}"))
        w = hip.SlpWeights(layers, dev("length_predictor.weight"), dev("length_predictor.bias"))
        with torch.cuda.device(self.device):
            if self.handle:
                self.lib.ditto_slp_destroy(self.handle)
                self.handle = C.c_void_p()
            hip.check(self.lib.ditto_slp_create(C.byref(self.cfg), C.byref(w), self.arena.data_ptr(),
                                                self.arena.numel(), _stream(), C.byref(self.handle)))
            torch.cuda.current_stream().synchronize()   # the fp32 staging copies may now be freed

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.ditto_slp_destroy(self.handle)
        except Exception:
            pass

    def forward(self, z_text: torch.Tensor, z_audio: torch.Tensor, return_decoded: bool = False):
        """z_text [B,T,d], z_audio [B,S,d] (CUDA) -> logits fp32 [B,num_classes] (and the decoded sequence [B,S,d])."""
        if z_audio.dim() != 3 or z_text.dim() != 3 or z_audio.shape[0] != z_text.shape[0]:
            raise ValueError(f"expected z_text [B,T,d] and z_audio [B,S,d], got {tuple(z_text.shape)} / "
                             f"{tuple(z_audio.shape)}")
        d = self.cfg.d_model
        if z_audio.shape[2] != d or z_text.shape[2] != d:
            raise ValueError(f"embedding width must be d_model = {d}")
        B, S, _ = z_audio.shape
        T = z_text.shape[1]
        if S == 0 or T == 0:
            raise ValueError("empty audio / text sequence")
        za = z_audio.detach().to(device=self.device, dtype=torch.float32).contiguous()
        zt = z_text.detach().to(device=self.device, dtype=torch.float32).contiguous()
        need = self.lib.ditto_slp_workspace_bytes(C.byref(self.cfg), B, S, T)
        if need == 0:
            raise hip.DittoHipError(hip.ERR_SHAPE, self.lib.ditto_last_error().decode())
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        logits = torch.empty(B, self.cfg.num_classes, dtype=torch.float32, device=self.device)
        decoded = torch.empty(B, S, d, dtype=torch.float32, device=self.device) if return_decoded else None
        with torch.cuda.device(self.device):
            hip.check(self.lib.ditto_slp_forward(self.handle, za.data_ptr(), zt.data_ptr(), B, S, T, logits.data_ptr(),
                                                 None if decoded is None else decoded.data_ptr(),
                                                 self._ws.data_ptr(), self._ws.numel(), _stream()))
        return (logits, decoded) if return_decoded else logits


class SLP(nn.Module):
    """Drop-in for the reference `SLP(max_audio_token_length, nhead=4, num_layers=4)` (src/model/SpeechLP.py:15-34).

    Extra keyword arguments, all optional: `text_encoder` / `audio_encoder` (the pretrained ByT5 / EnCodec wrappers,
    src/components/ByT5.py, EnCodec.py) and `hidden_size` (needed only when no text encoder is given; the reference
    takes it from `text_encoder.model.config.d_model` = 1472 for byt5-small)."""

    def __init__(self, max_audio_token_length, nhead=4, num_layers=4, *, hidden_size: Optional[int] = None,
                 text_encoder: Optional[nn.Module] = None, audio_encoder: Optional[nn.Module] = None):
        super().__init__()
        if text_encoder is not None:
            self.text_encoder = text_encoder
            hidden_size = int(text_encoder.model.config.d_model) if hidden_size is None else hidden_size
        if hidden_size is None:
            raise ValueError("SLP needs text_encoder= (the ByT5 wrapper) or hidden_size= (1472 for byt5-small)")
        self.hidden_size = int(hidden_size)
        if audio_encoder is not None:
            self.audio_encoder = audio_encoder
        self.nhead, self.num_layers, self.num_classes = int(nhead), int(num_layers), int(max_audio_token_length)
        # parameter owner with the reference's keys; never called (see module docstring)
        self.transformer = nn.TransformerDecoder(
            nn.TransformerDecoderLayer(d_model=self.hidden_size, nhead=nhead,
                                       dim_feedforward=self.hidden_size * nhead, batch_first=True),
            num_layers=num_layers)
        self.length_predictor = nn.Linear(self.hidden_size, max_audio_token_length)
        self._engine: Optional[SlpEngine] = None
        self._watch = _ParamWatch()

    def _path_tensors(self):
        return list(self.transformer.parameters()) + list(self.length_predictor.parameters())

    def engine(self, device=None) -> SlpEngine:
        device = torch.device(device) if device is not None else self.length_predictor.weight.device
        if device.type != "cuda":
            raise RuntimeError("SLP parameters are not on a CUDA (ROCm) device: call .to('cuda') first; "
                               "ditto_tts_amd has no CPU path")
        tensors = self._path_tensors()
        sd = lambda: {k: v for k, v in self.state_dict().items()   # noqa: E731
                      if k.startswith(("transformer.", "length_predictor."))}
        if self._engine is None or self._engine.device != device:
            self._engine = SlpEngine(self.hidden_size, self.nhead, self.num_layers, self.hidden_size * self.nhead,
                                     self.num_classes, sd(), device)
            self._watch.changed(tensors)
        elif self._watch.changed(tensors):
            self._engine.repack(sd())
        return self._engine

    def decode(self, z_text: torch.Tensor, z_audio: torch.Tensor, return_decoded: bool = False):
        """src/model/SpeechLP.py:50-54 on encoder outputs: z_text [B,T,d], z_audio [B,S,d] -> length logits [B,C]."""
        if self.training:
            raise NotImplementedError("ditto_tts_amd: the SLP decoder stack is inference-only (eval mode: no dropout); "
                                      "call .eval() — training it (src/TrainSLP.py) is out of scope")
        if z_audio.device.type != "cuda":
            raise RuntimeError("z_audio is not on a CUDA (ROCm) device; ditto_tts_amd has no CPU path")
        out = self.engine(z_audio.device).forward(z_text, z_audio, return_decoded)
        if return_decoded:
            return out[0].to(z_audio.dtype), out[1].to(z_audio.dtype)
        return out.to(z_audio.dtype)

    @torch.no_grad()
    def forward(self, text, audio):
        """text: tokenizer output for ByT5, audio: waveforms (src/model/SpeechLP.py:36-55)."""
        if not hasattr(self, "text_encoder") or not hasattr(self, "audio_encoder"):
            raise RuntimeError("SLP.forward(text, audio) needs the pretrained encoders (text_encoder=, audio_encoder=); "
                               "with encoder outputs at hand call SLP.decode(z_text, z_audio)")
        z_text = self.text_encoder(text)
        z_audio, _ = self.audio_encoder(audio)
        z_audio = z_audio.view(z_audio.size(0), -1, z_audio.size(-1))
        return self.decode(z_text, z_audio)

    @staticmethod
    def generate_causal_mask(size, device):
        """The boolean tgt_mask the reference builds (src/model/SpeechLP.py:57-61); the HIP path applies it inside
        the softmax and never materialises it."""
        return torch.triu(torch.ones(size, size), diagonal=1).bool().to(device)
