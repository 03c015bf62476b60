"""nn.Module façade with the reference's class names, constructor signatures, attribute names and
state_dict keys whose forward() runs on libditto_hip.so.

    reference class                         this file
    components.DiT.GlobalAdaLN   (:8-40)    GlobalAdaLN
    components.DiT.RotaryEmbedding (:43-72) RotaryEmbedding
    components.DiT.DiT           (:75-157)  DiT
    model.DiTTO.DiTTO            (:7-126)   DiTTO

torch.nn modules are used only as PARAMETER CONTAINERS (so `state_dict()`, `.to()`, optimisers and the
reference's checkpoints work unchanged); none of their forward()s is ever called.  There is no CPU or
eager-PyTorch fallback: a non-CUDA input or a missing libditto_hip.so raises.

Training: `DiTTO.forward` under autograd runs ditto_train_forward / ditto_train_backward
through a torch.autograd.Function, so the reference's training closure (src/TrainDiTTO.py:55-95) drives the HIP
path unchanged: `model.train()`, `loss.backward()`, any torch optimizer over `model.parameters()`.  Gradients
reach every live parameter; `blocks.i.attn.out_proj.*` get none (dead in the reference too), and neither do `x`
and `text_emb` (frozen-encoder outputs in the reference) — asking for those raises.  The standalone
`DiT` / `GlobalAdaLN` modules stay forward-only.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
import torch.nn as nn

from . import hip
from .config import DiTTOConfig
from .engine import DenoiseEngine, TextCond, _stream
from .synth import cosine_betas


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what} is on {t.device}: ditto_tts_amd runs only on an MI355X through libditto_hip.so "
                           "(no CPU / eager fallback). Move the module and its inputs to 'cuda'.")


def _refuse_autograd(module: nn.Module):
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters(recurse=True)):
        raise NotImplementedError(
            "ditto_tts_amd: the standalone DiT / GlobalAdaLN modules are forward-only (the backward pass is built for "
            "the whole model: train through DiTTO.forward). Call under torch.no_grad() / torch.inference_mode().")


class _DiTTOTrainFn(torch.autograd.Function):
    """DiTTO.forward with a tape (ditto_train_forward) and its backward (ditto_train_backward)."""

    @staticmethod
    def forward(ctx, model, x, text_emb, t, dropout_p, seed, keys, *params):
        eng = model.engine(x.device, train=True)
        # the options in force where forward() is CALLED (a hip.batch_class / hip.call_opts scope of this thread) are this
        # step's options: autograd runs backward() on its own worker thread, where no such scope exists, so they travel in ctx
        ctx.opts = hip.current_opts() if hip._has_call_opts() else None
        out, tape, xf, tt = eng.train_forward(x, text_emb.to(x.device), t, dropout_p, seed, opts=ctx.opts)
        ctx.model, ctx.eng, ctx.tape, ctx.xf, ctx.tt = model, eng, tape, xf, tt
        ctx.T, ctx.dropout_p, ctx.seed, ctx.keys = text_emb.shape[1], dropout_p, seed, keys
        ctx.sig = model._watch.sig
        return out

    @staticmethod
    def backward(ctx, grad_out):
        model, eng = ctx.model, ctx.eng
        if model._watch.sig != ctx.sig or model._watch.changed(model._path_tensors()):
            raise RuntimeError("DiTTO parameters changed between forward and backward")
        if ctx.tape is None:
            raise RuntimeError("backward through the same DiTTO.forward twice (the activation tape was released)")
        sd = {k: v for k, v in model.state_dict(keep_vars=True).items() if not k.startswith("nac.")}
        sync = getattr(model, "_grad_sync", None)      # dist.GradSync: the gradient exchange overlapped with this backward
        try:
            grads = eng.train_backward(sd, grad_out, ctx.xf, ctx.tt, ctx.T, ctx.tape, ctx.dropout_p, ctx.seed, opts=ctx.opts,
                                       piece_cb=sync.reduce if sync is not None else None,
                                       layers_per_piece=getattr(model, "_grad_sync_layers", 1))
        except BaseException:
            if sync is not None:
                sync.abort()                           # no stale tensors in the next step's buckets
            raise
        if sync is not None:
            sync.finish()                              # the compute stream waits for the exchange: the gradients returned are the means
        B, N, _ = ctx.xf.shape
        eng.release_tape(ctx.tape, B, N, ctx.T)
        ctx.tape = None
        return (None, None, None, None, None, None, None) + tuple(grads.get(k) for k in ctx.keys)


class _ParamWatch:
    """Detects parameter changes (optimizer step, load_state_dict, .to()) so packed weights are rebuilt lazily."""

    def __init__(self):
        self.sig = None

    def changed(self, tensors) -> bool:
        sig = tuple((t.data_ptr(), t._version) for t in tensors)
        if sig != self.sig:
            self.sig = sig
            return True
        return False


class GlobalAdaLN(nn.Module):
    """Global Adaptive LayerNorm — reference src/components/DiT.py:8-40."""

    def __init__(self, hidden_dim, time_dim, text_dim):
        super().__init__()
        self.time_mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_dim, 2 * hidden_dim))
        self.text_mlp = nn.Sequential(nn.SiLU(), nn.Linear(text_dim, 2 * hidden_dim))
        self.norm = nn.LayerNorm(hidden_dim, elementwise_affine=False)
        self._dims = (hidden_dim, time_dim, text_dim)

    def forward(self, x, time_emb, text_emb):
        _require_cuda(x, "x")
        _refuse_autograd(self)
        lib = hip.lib()
        d, td, dt = self._dims
        xf = x.float().contiguous()
        te = time_emb.to(x.device).float().contiguous()
        tx = text_emb.to(x.device).float().contiguous()
        B, N, _ = xf.shape
        T = tx.shape[1]
        out = torch.empty_like(xf)
        scratch = torch.empty(lib.ditto_global_adaln_scratch_bytes(B, d, td, dt), dtype=torch.uint8, device=x.device)
        w = [self.time_mlp[1].weight, self.time_mlp[1].bias, self.text_mlp[1].weight, self.text_mlp[1].bias]
        w = [p.detach().float().contiguous() for p in w]
        hip.check(lib.ditto_global_adaln(xf.data_ptr(), te.data_ptr(), tx.data_ptr(), *[p.data_ptr() for p in w],
                                         B, N, T, d, td, dt, out.data_ptr(), scratch.data_ptr(), scratch.numel(),
                                         _stream()))
        return out.to(x.dtype)


class RotaryEmbedding(nn.Module):
    """RoPE — reference src/components/DiT.py:43-72."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        inv_freq = 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim))
        self.register_buffer("inv_freq", inv_freq)

    def _rotate_half(self, x):
        x1, x2 = x.chunk(2, dim=-1)
        return torch.cat((-x2, x1), dim=-1)

    def forward(self, seq_len, device):
        # the [N, dh] ANGLE table the reference hands to its blocks; an index outer product, not a hot op
        t = torch.arange(seq_len, device=device).type_as(self.inv_freq)
        freqs = torch.outer(t, self.inv_freq.to(device))
        return torch.cat((freqs, freqs), dim=-1)

    def apply_rope(self, pos, t):
        _require_cuda(t, "t")
        B, N, H, dh = t.shape
        tf = t.float().contiguous()
        out = torch.empty_like(tf)
        posf = pos.to(t.device).float().contiguous()
        hip.check(hip.lib().ditto_apply_rope_f32(posf.data_ptr(), tf.data_ptr(), out.data_ptr(), B, N, H, dh,
                                                 _stream()))
        return out.to(t.dtype)


class DiT(nn.Module):
    """Single DiT block — reference src/components/DiT.py:75-157.

    Parameter containers are the same torch modules the reference instantiates, so the key set is identical
    (including the dead `attn.out_proj.*`)."""

    def __init__(self, hidden_dim, num_heads, time_dim, text_dim):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = hidden_dim // num_heads
        self.norm1 = nn.LayerNorm(hidden_dim)
        self.attn = nn.MultiheadAttention(hidden_dim, num_heads)
        self.rotary = RotaryEmbedding(self.head_dim)
        self.norm2 = nn.LayerNorm(hidden_dim)
        self.cross_attn = nn.MultiheadAttention(hidden_dim, num_heads, dropout=0.1)
        self.norm3 = nn.LayerNorm(hidden_dim)
        self.mlp_fc1 = nn.Linear(hidden_dim, 4 * hidden_dim)
        self.act = nn.GELU()
        self.gate = nn.Linear(hidden_dim, 4 * hidden_dim)
        self.mlp_fc2 = nn.Linear(4 * hidden_dim, hidden_dim)
        self._cfg = DiTTOConfig(hidden_dim, 1, num_heads, time_dim, text_dim, 1)
        self._engine: Optional[DenoiseEngine] = None
        self._watch = _ParamWatch()

    def _standalone_engine(self, device) -> DenoiseEngine:
        tensors = list(self.parameters())
        if self._engine is None or self._watch.changed(tensors) or self._engine.device != device:
            sd = {"blocks.0." + k: v for k, v in self.state_dict().items()}
            self._engine = _BlocksOnlyEngine(self._cfg, sd, device)
            self._watch.changed(tensors)
        return self._engine

    def forward(self, x, text_emb, time_emb, rotary_pos, taps=None):
        """`time_emb` is accepted and ignored, exactly as in the reference.  `taps` (an extension, for
        segment-wise parity tests): a dict that receives the residual stream "after_self" / "after_cross"."""
        _require_cuda(x, "x")
        _refuse_autograd(self)
        eng = self._standalone_engine(x.device)
        half = self.head_dim // 2
        ang = rotary_pos.to(x.device).float()[:, :half].contiguous()   # cat(freqs, freqs): first half suffices
        rope = (torch.cos(ang).contiguous(), torch.sin(ang).contiguous())
        cond = eng.prepare_text(text_emb, x.shape[1])
        h = x.float().contiguous().clone()
        eng.block_forward_(0, h, cond, 0, rope, taps=taps)
        return h.to(x.dtype)


class _BlocksOnlyEngine(DenoiseEngine):
    """An engine over DiT blocks only (no model-level weights): ditto_model_create with NULL globals."""

    def _pack(self, state):
        keep = []

        def dev(key):
            t = state[key].detach().to(device=self.device, dtype=torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()

        L = self.cfg.num_layers
        layers = (hip.LayerWeights * L)()
        for l in range(L):
            for f, k in hip.LAYER_KEY.items():
                setattr(layers[l], f, dev(f"blocks.{l}.{k}"))
        w = hip.Weights()
        w.layers = layers
        with torch.cuda.device(self.device):
            if self.handle:
                hip.check(self.lib.ditto_model_destroy(self.handle))
                self.handle = C.c_void_p()
            hip.check(self.lib.ditto_model_create(C.byref(self._ccfg), C.byref(w), self.arena.data_ptr(),
                                                  self.arena.numel(), _stream(), C.byref(self.handle)))
            torch.cuda.current_stream().synchronize()
        self._rope.clear()
        self._generation += 1


class DiTTO(nn.Module):
    """Full DiT noise predictor — reference src/model/DiTTO.py:7-126.

    Same keyword arguments.  The reference unconditionally builds `NAC(lambda_factor)` (HF-hub downloads) and
    `torch.load(nac_model_path)`; neither is on the denoise path, so two OPTIONAL escapes are added:
    `nac=` (an already-built codec module to attach as `self.nac`) and `nac_model_path=None` (no codec:
    `self.nac` is None and only the denoise surface works).  With a path, the codec class is imported from the
    caller's own tree (`model.NeuralAudioCodec.NAC`, as the reference does, src/model/DiTTO.py:4)."""

    def __init__(self, hidden_dim=768, num_layers=12, num_heads=12, time_dim=256, text_dim=768,
                 diffusion_steps=1000, lambda_factor=0.1, nac_model_path=None, nac: Optional[nn.Module] = None,
                 fp8_linear: bool = False):
        super().__init__()
        # fp8_linear (extra, optional): QKV / FFN GEMMs with fp8 e4m3 operands (BASELINE config 5)
        self.cfg = DiTTOConfig(hidden_dim, num_layers, num_heads, time_dim, text_dim, diffusion_steps,
                               fp8_linear=fp8_linear)
        if nac is not None:
            self.nac = nac
        elif nac_model_path is not None:
            print("[INFO] Loading NAC model...")
            from model.NeuralAudioCodec import NAC  # the caller's (reference) tree; not part of this package
            self.nac = NAC(lambda_factor=lambda_factor)
            nac_info = torch.load(nac_model_path)
            self.nac.load_state_dict(nac_info["model_state_dict"])
            self.nac.eval()
            for p in self.nac.language_model.parameters():
                p.requires_grad = False
            for p in self.nac.audio_encoder.parameters():
                p.requires_grad = False
            print("[INFO] NAC Loaded.")
        else:
            self.nac = None

        self.t_embedding = nn.Embedding(diffusion_steps, time_dim)
        self.time_embed = nn.Sequential(nn.Linear(time_dim, time_dim), nn.SiLU(), nn.Linear(time_dim, time_dim))
        self.ada_ln = GlobalAdaLN(hidden_dim, time_dim, text_dim)
        self.blocks = nn.ModuleList([DiT(hidden_dim, num_heads, time_dim, text_dim) for _ in range(num_layers)])
        self.proj_in = nn.Linear(hidden_dim, hidden_dim)
        self.proj_out = nn.Linear(hidden_dim, hidden_dim)
        self.rotary = RotaryEmbedding(hidden_dim // num_heads)
        # bug-for-bug: the buffer named alphas_cumprod holds the clipped betas
        self.register_buffer("alphas_cumprod", self.cosine_beta_schedule(diffusion_steps))

        self._engine: Optional[DenoiseEngine] = None
        self._watch = _ParamWatch()
        self._cond_key = None
        self._cond: Optional[TextCond] = None
        self._cond_src: Optional[torch.Tensor] = None

    # ------------------------------------------------------------------ engine plumbing
    def _path_tensors(self):
        return [p for n, p in self.named_parameters() if not n.startswith("nac.")] + [self.rotary.inv_freq]

    def engine(self, device=None, train: bool = False) -> DenoiseEngine:
        """The packed HIP model for the current parameters (rebuilt lazily after they change); `train` also packs
        the transposed copies the backward's dgrad GEMMs read."""
        device = torch.device(device) if device is not None else self.proj_in.weight.device
        if device.type != "cuda":
            raise RuntimeError("DiTTO parameters are not on a CUDA (ROCm) device: call .to('cuda') first; "
                               "ditto_tts_amd has no CPU path")
        tensors = self._path_tensors()
        if self._engine is None or self._engine.device != device:
            sd = {k: v for k, v in self.state_dict().items() if not k.startswith("nac.")}
            self._engine = DenoiseEngine(self.cfg, sd, device)
            self._watch.changed(tensors)
            self._cond_key = None
            self._train_packed = False
        elif self._watch.changed(tensors):
            self._engine.repack({k: v for k, v in self.state_dict().items() if not k.startswith("nac.")})
            self._cond_key = None
            self._train_packed = False
        if train and not getattr(self, "_train_packed", False):
            self._engine.train_attach({k: v for k, v in self.state_dict().items() if not k.startswith("nac.")})
            self._train_packed = True
        return self._engine

    def set_grad_sync(self, sync, layers_per_piece: int = 1):
        """Data-parallel training: `sync` (ditto_tts_amd.dist.GradSync, or None to switch off) receives every piece of the
        backward's gradients as soon as that piece is enqueued (engine.train_backward piece_cb) and exchanges them on a side
        stream while the layers below are computed; `loss.backward()` then returns gradients that already are the mean over the
        data-parallel group — no allreduce_gradients call afterwards.  The reference has no distributed training
        (src/utils/Trainer.py:16 trains on one device); this is the hook a multi-GPU Trainer would use."""
        self._grad_sync, self._grad_sync_layers = sync, int(layers_per_piece)

    def text_cond(self, text_emb: torch.Tensor, N_hint: int = 1) -> TextCond:
        """Step-invariant text work, cached while the caller keeps passing the same text_emb tensor (the
        sampler does for all its steps, reference src/model/SpeechGenerator.py:161-163)."""
        eng = self.engine()
        key = (text_emb.data_ptr(), text_emb._version, tuple(text_emb.shape), text_emb.dtype, text_emb.device)
        if key != self._cond_key or self._cond is None:
            self._cond = eng.prepare_text(text_emb, N_hint)
            self._cond_key = key
            # the entry is keyed by ADDRESS: keep the keyed tensor alive for as long as the entry is, or the caching
            # allocator hands the same address (version 0, same shape) to the next utterance's temporary
            # (`text.to(device)` of a CPU tensor, a function-local `wte(tokens)`) and the stale K/V would hit
            self._cond_src = text_emb
        return self._cond

    def invalidate(self):
        """Force a repack of the weights and drop the cached text conditioning on the next call.  Needed only after
        updates the version counters cannot see: in-place writes through `.data` (`p.data.copy_(ema)`,
        some legacy optimizers) do not bump `Tensor._version`."""
        self._watch.sig = None
        self._cond_key = None
        self._cond = None
        self._cond_src = None

    # ------------------------------------------------------------------ reference surface
    def forward(self, x, text_emb, t):
        """x [B,N,d] noisy latents, text_emb [B,T,text_dim], t [B] long -> predicted noise [B,N,d]
        (reference src/model/DiTTO.py:66-94)."""
        _require_cuda(x, "x")
        if torch.is_grad_enabled() and (x.requires_grad or text_emb.requires_grad):
            raise NotImplementedError("ditto_tts_amd: gradients with respect to x / text_emb are not produced (the "
                                      "reference feeds frozen-encoder outputs, src/TrainDiTTO.py:66-73); detach them")
        named = [(n, p) for n, p in self.named_parameters() if not n.startswith("nac.") and ".attn.out_proj." not in n]
        if torch.is_grad_enabled() and any(p.requires_grad for _, p in named):
            if self.cfg.fp8_linear:
                raise NotImplementedError("ditto_tts_amd: training with fp8_linear=True is not supported")
            # cross-attention dropout is active in train mode only (nn.MultiheadAttention(dropout=0.1),
            # reference src/components/DiT.py:90-91); its mask seed comes from torch's CPU generator
            p_drop = float(self.blocks[0].cross_attn.dropout) if self.training else 0.0
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p_drop > 0 else 0
            keys = tuple(n for n, _ in named)
            out = _DiTTOTrainFn.apply(self, x, text_emb, t, p_drop, seed, keys, *[p for _, p in named])
            return out if x.dtype == torch.float32 else out.to(x.dtype)
        eng = self.engine(x.device)
        cond = self.text_cond(text_emb.to(x.device), x.shape[1])
        out = eng.forward(x, cond, t)
        return out if x.dtype == torch.float32 else out.to(x.dtype)

    def cosine_beta_schedule(self, timesteps, s=0.008):
        """Reference src/model/DiTTO.py:96-104 (a dozen-element host-side table, torch ops as in the reference)."""
        if s != 0.008:
            steps = timesteps + 1
            x = torch.linspace(0, timesteps, steps)
            ac = torch.cos(((x / timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
            ac = ac / ac[0]
            return torch.clip(1 - (ac[1:] / ac[:-1]), 0.0001, 0.9999)
        return cosine_betas(timesteps)

    def q_sample(self, x_start, t, noise=None):
        """Forward diffusion, reference src/model/DiTTO.py:106-126 (uses the mis-named buffer, bug-for-bug)."""
        _require_cuda(x_start, "x_start")
        if noise is None:
            noise = torch.randn_like(x_start)
        x0 = x_start.float().contiguous()
        nz = noise.to(x_start.device).float().contiguous()
        tt = t.to(x_start.device).long().contiguous()
        buf = self.alphas_cumprod.to(x_start.device).float().contiguous()
        out = torch.empty_like(x0)
        B = x0.shape[0]
        hip.check(hip.lib().ditto_q_sample(x0.data_ptr(), nz.data_ptr(), tt.data_ptr(), buf.data_ptr(), out.data_ptr(),
                                           B, x0.numel() // B, _stream()))
        return out.to(x_start.dtype)
