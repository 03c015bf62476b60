"""DenoiseEngine — host-side owner of one packed model on one GPU.

PyTorch is plumbing here: it owns device memory (arena / workspace / cond buffers are uint8 tensors) and
the stream; every FLOP of the path runs in libditto_hip.so.  One engine per process per GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Mapping, Optional, Tuple

import torch

from . import hip
from .config import DiTTOConfig


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class TextCond:
    """Step-invariant conditioning of one utterance batch: cached cross-attention K/V of every layer and the
    text half of the AdaLN modulation (ditto_text_precompute)."""

    def __init__(self, buf: torch.Tensor, B: int, T: int):
        self.buf, self.B, self.T = buf, B, T


class StepGraph:
    """One captured reverse-diffusion step.  The HIP graph holds raw device addresses; this object owns a reference
    to every buffer behind them (state, conditioning, t, noise, schedule tables, the engine's workspace and RoPE
    tables AT CAPTURE TIME), so the engine growing its workspace or dropping its RoPE cache later cannot hand that
    memory to someone else while the graph can still be replayed.  A repack of the weights (new model handle, arena
    rewritten) invalidates the graph: replay() then raises instead of running on a half-updated arena."""

    def __init__(self, engine, graph, keep):
        self._engine, self._graph, self._keep = engine, graph, keep
        self._generation = engine._generation

    def replay(self):
        if self._generation != self._engine._generation:
            raise RuntimeError("this step graph was captured before the engine's weights were repacked; capture again")
        self._graph.replay()


class DenoiseEngine:
    def __init__(self, cfg: DiTTOConfig, state: Mapping[str, torch.Tensor], device: Optional[torch.device] = None):
        """`state`: reference state_dict keys -> tensors; fp32 CUDA copies are made as needed.
        `nac.*`, `blocks.i.attn.out_proj.*`, `blocks.i.rotary.inv_freq` and `alphas_cumprod` are ignored here."""
        self.lib = hip.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("DenoiseEngine needs an MI355X (torch.cuda.is_available() is False); "
                               "ditto_tts_amd has no CPU path")
        self.cfg = cfg
        self.device = torch.device(device if device is not None else "cuda")
        self._ccfg = hip.make_config(cfg)
        nbytes = self.lib.ditto_arena_bytes(C.byref(self._ccfg))
        if nbytes == 0:
            raise hip.DittoHipError(hip.ERR_SHAPE, self.lib.ditto_last_error().decode())
        self.arena = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self.handle = C.c_void_p()
        self._ws: Optional[torch.Tensor] = None
        self._rope: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
        self._generation = 0            # bumped by every (re)pack: outstanding StepGraphs check it
        self._pack(state)

    # ------------------------------------------------------------------ weights
    def _pack(self, state: Mapping[str, torch.Tensor]):
        keep = []  # fp32 device staging copies must outlive the (stream-ordered) pack kernels

        def dev(key):
            if key not in state:
                raise KeyError(f"state_dict is missing '{key}'")
            t = state[key].detach().to(device=self.device, dtype=torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()

        L = self.cfg.num_layers
        layers = (hip.LayerWeights * L)()
        for l in range(L):
            for f, k in hip.LAYER_KEY.items():
                setattr(layers[l], f, dev(f"blocks.{l}.{k}"))
        w = hip.Weights()
        for f, k in hip.GLOBAL_KEY.items():
            setattr(w, f, dev(k))
        w.layers = layers
        with torch.cuda.device(self.device):
            if self.handle:
                hip.check(self.lib.ditto_model_destroy(self.handle))
                self.handle = C.c_void_p()
            hip.check(self.lib.ditto_model_create(C.byref(self._ccfg), C.byref(w), self.arena.data_ptr(),
                                                  self.arena.numel(), _stream(), C.byref(self.handle)))
            torch.cuda.current_stream().synchronize()  # staging copies may now be freed
        self._rope.clear()
        self._generation += 1
        self._train_attached = False    # a new handle: the transposed training copies must be re-attached

    def repack(self, state: Mapping[str, torch.Tensor]):
        self._pack(state)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.ditto_model_destroy(self.handle)
        except Exception:
            pass

    # ------------------------------------------------------------------ buffers
    def workspace(self, B: int, N: int, T: int) -> torch.Tensor:
        need = self.lib.ditto_workspace_bytes(C.byref(self._ccfg), B, N, T)
        if need == 0:
            raise hip.DittoHipError(hip.ERR_SHAPE, self.lib.ditto_last_error().decode())
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def rope_tables(self, N: int):
        if N not in self._rope:
            half = self.cfg.head_dim // 2
            c = torch.empty(N, half, dtype=torch.float32, device=self.device)
            s = torch.empty_like(c)
            hip.check(self.lib.ditto_rope_tables(self.handle, N, c.data_ptr(), s.data_ptr(), _stream()))
            self._rope[N] = (c, s)
        return self._rope[N]

    # ------------------------------------------------------------------ path
    def _f32(self, t: torch.Tensor, name: str) -> torch.Tensor:
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA (ROCm) tensor: ditto_tts_amd has no CPU path")
        return t.to(dtype=torch.float32).contiguous()

    def prepare_text(self, text_emb: torch.Tensor, N_hint: int = 1) -> TextCond:
        """text_emb [B, T, text_dim] -> TextCond (K/V cache of all layers + text modulation)."""
        text = self._f32(text_emb, "text_emb")
        B, T, dt = text.shape
        if dt != self.cfg.text_dim:
            raise ValueError(f"text_emb last dim {dt} != text_dim {self.cfg.text_dim}")
        nb = self.lib.ditto_cond_bytes(C.byref(self._ccfg), B, T)
        buf = torch.empty(nb, dtype=torch.uint8, device=self.device)
        ws = self.workspace(B, max(N_hint, 1), T)
        hip.check(self.lib.ditto_text_precompute(self.handle, text.data_ptr(), B, T, buf.data_ptr(), nb,
                                                 ws.data_ptr(), ws.numel(), _stream()))
        return TextCond(buf, B, T)

    def prepare_text_into(self, text_emb: torch.Tensor, N_hint: int, cond: TextCond) -> TextCond:
        """Recompute the conditioning of a new utterance batch into an EXISTING TextCond buffer (same B, T), so
        captured graphs bound to that buffer stay valid."""
        text = self._f32(text_emb, "text_emb")
        B, T, _ = text.shape
        if (B, T) != (cond.B, cond.T):
            raise ValueError("prepare_text_into needs the same (B, T) as the existing conditioning")
        ws = self.workspace(B, max(N_hint, 1), T)
        hip.check(self.lib.ditto_text_precompute(self.handle, text.data_ptr(), B, T, cond.buf.data_ptr(),
                                                 cond.buf.numel(), ws.data_ptr(), ws.numel(), _stream()))
        return cond

    def _t64(self, t: torch.Tensor, B: int) -> torch.Tensor:
        if t.shape != (B,):
            raise ValueError(f"t must have shape [{B}]")
        return t.to(device=self.device, dtype=torch.int64).contiguous()

    def _call(self, name, *args, opts=None):
        """`name`_opts(*args, opts) — the entry point with this call's ditto_call_opts (None = NULL: every field inherits the
        thread's scope / the process defaults); a frozen pre-ABI-9 library (DITTO_HIP_LIB: tools/ A/Bs) has only `name`."""
        if hip._has_call_opts():
            hip.check(getattr(self.lib, name + "_opts")(*args, None if opts is None else C.byref(opts)))
        elif opts is not None:
            raise RuntimeError("per-call options need an ABI-9 libditto_hip.so")
        else:
            hip.check(getattr(self.lib, name)(*args))

    def forward(self, x: torch.Tensor, cond: TextCond, t: torch.Tensor, out: Optional[torch.Tensor] = None,
                opts: Optional[hip.CallOpts] = None):
        """DiTTO.forward(x, text_emb, t) with text_emb pre-digested into `cond` -> eps fp32 [B,N,d].  `opts`: this call's
        hip.CallOpts (kernel class pin, residual-stream type ...: include/ditto_hip.h ditto_call_opts)."""
        xf = self._f32(x, "x")
        B, N, d = xf.shape
        if d != self.cfg.hidden_dim or B != cond.B:
            raise ValueError("x shape does not match the model / the conditioning batch")
        tt = self._t64(t, B)
        if out is None:
            out = torch.empty_like(xf)
        ws = self.workspace(B, N, cond.T)
        c, s = self.rope_tables(N)
        self._call("ditto_forward", self.handle, xf.data_ptr(), cond.buf.data_ptr(), tt.data_ptr(), B, N, cond.T,
                                              c.data_ptr(), s.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                              _stream(), opts=opts)
        return out

    def p_sample_(self, x: torch.Tensor, cond: TextCond, t: torch.Tensor, noise: Optional[torch.Tensor],
                  betas: torch.Tensor, alphas: torch.Tensor, alphas_cumprod: torch.Tensor,
                  opts: Optional[hip.CallOpts] = None):
        """One reverse-diffusion step IN PLACE on the fp32 CUDA state x [B,N,d]."""
        if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
            raise ValueError("p_sample_ needs a contiguous fp32 CUDA state tensor (it is updated in place)")
        B, N, d = x.shape
        tt = self._t64(t, B)
        ws = self.workspace(B, N, cond.T)
        c, s = self.rope_tables(N)
        if noise is not None:
            noise = self._f32(noise, "noise")
        self._call("ditto_p_sample", self.handle, x.data_ptr(), cond.buf.data_ptr(), tt.data_ptr(), _ptr(noise),
                                               betas.data_ptr(), alphas.data_ptr(), alphas_cumprod.data_ptr(), B, N, cond.T,
                                               c.data_ptr(), s.data_ptr(), ws.data_ptr(), ws.numel(), _stream(), opts=opts)
        return x

    def noise_normal_(self, out: torch.Tensor, seeds: torch.Tensor, step: int):
        """out[b] <- N(0,1) of (seeds[b], step): Philox4x32-10 + Box-Muller in libditto_hip (ditto_noise_normal).  A function
        of the utterance's seed, the step and the element index only — independent of batch composition and GPU."""
        if not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()):
            raise ValueError("noise_normal_ needs a contiguous fp32 CUDA tensor")
        B = out.shape[0]
        sd = self._t64(seeds, B)
        hip.check(self.lib.ditto_noise_normal(out.data_ptr(), sd.data_ptr(), int(step) & 0xFFFFFFFF, B,
                                              out.numel() // B, _stream()))
        return out

    def p_sample_seeded_(self, x: torch.Tensor, cond: TextCond, t: torch.Tensor, seeds: torch.Tensor, step: int,
                         betas: torch.Tensor, alphas: torch.Tensor, alphas_cumprod: torch.Tensor,
                         opts: Optional[hip.CallOpts] = None):
        """p_sample_ with the step's noise generated inside the update kernel from per-utterance seeds
        (bit-identical to noise_normal_(z, seeds, step) + p_sample_(x, ..., z))."""
        if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
            raise ValueError("p_sample_seeded_ needs a contiguous fp32 CUDA state tensor (it is updated in place)")
        B, N, d = x.shape
        tt, sd = self._t64(t, B), self._t64(seeds, B)
        ws = self.workspace(B, N, cond.T)
        c, s = self.rope_tables(N)
        self._call("ditto_p_sample_seeded", self.handle, x.data_ptr(), cond.buf.data_ptr(), tt.data_ptr(),
                                                      sd.data_ptr(), int(step) & 0xFFFFFFFF, betas.data_ptr(),
                                                      alphas.data_ptr(), alphas_cumprod.data_ptr(), B, N, cond.T, c.data_ptr(),
                                                      s.data_ptr(), ws.data_ptr(), ws.numel(), _stream(), opts=opts)
        return x

    def denoise_steps_(self, x: torch.Tensor, cond: TextCond, t_begin: int, t_end: int, noises: Optional[torch.Tensor],
                       betas: torch.Tensor, alphas: torch.Tensor, alphas_cumprod: torch.Tensor,
                       opts: Optional[hip.CallOpts] = None):
        """The sampling loop t_begin .. t_end (inclusive, descending) as ONE library call, in place on x;
        noises fp32 [t_begin - t_end + 1, B, N, d] in execution order."""
        if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
            raise ValueError("denoise_steps_ needs a contiguous fp32 CUDA state tensor (it is updated in place)")
        B, N, _ = x.shape
        nsteps = t_begin - t_end + 1
        if noises is not None:
            noises = self._f32(noises, "noises")
            if noises.shape != (nsteps, *x.shape):
                raise ValueError(f"noises must have shape {(nsteps, *x.shape)}")
        ws = self.workspace(B, N, cond.T)
        c, s = self.rope_tables(N)
        tt = torch.empty(B, dtype=torch.int64, device=self.device)
        self._call("ditto_denoise_steps", self.handle, x.data_ptr(), cond.buf.data_ptr(), int(t_begin), int(t_end),
                                                    _ptr(noises), betas.data_ptr(), alphas.data_ptr(),
                                                    alphas_cumprod.data_ptr(), B, N, cond.T, c.data_ptr(), s.data_ptr(),
                                                    tt.data_ptr(), ws.data_ptr(), ws.numel(), _stream(), opts=opts)
        return x

    def capture_p_sample(self, x: torch.Tensor, cond: TextCond, t: torch.Tensor, noise: torch.Tensor,
                         betas: torch.Tensor, alphas: torch.Tensor, alphas_cumprod: torch.Tensor,
                         opts: Optional[hip.CallOpts] = None):
        """Capture ONE reverse-diffusion step (the ~122 stream-ordered launches of ditto_p_sample) into a HIP
        graph bound to these exact tensors; returns a StepGraph (which keeps them, the workspace and the
        RoPE tables alive).  Replaying it advances `x` in place using whatever
        `t` and `noise` hold at replay time, so a sampling loop is: fill t, draw noise, graph.replay().
        Worth it when the step is launch-bound (small batches); the library calls neither allocate nor
        synchronise, so they are capturable as they are."""
        for name, v in (("x", x), ("noise", noise)):
            if not (v.is_cuda and v.dtype == torch.float32 and v.is_contiguous()):
                raise ValueError(f"{name} must be a contiguous fp32 CUDA tensor")
        if t.dtype != torch.int64 or not t.is_cuda:
            raise ValueError("t must be an int64 CUDA tensor")
        B, N, _ = x.shape
        ws = self.workspace(B, N, cond.T)      # allocate outside the capture
        rope = self.rope_tables(N)
        self.p_sample_(x, cond, t, noise, betas, alphas, alphas_cumprod, opts=opts)   # warm-up (lazy kernel attributes)
        torch.cuda.current_stream().synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.p_sample_(x, cond, t, noise, betas, alphas, alphas_cumprod, opts=opts)
        return StepGraph(self, g, (ws, rope, cond.buf, x, t, noise, betas, alphas, alphas_cumprod, self.arena))

    def block_forward_(self, layer: int, h: torch.Tensor, cond: TextCond, cond_layer: Optional[int] = None,
                       rope: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, taps: Optional[dict] = None):
        """DiT block `layer` in place on the fp32 residual stream h [B,N,d].  `taps` (optional dict) receives the
        stream after the self-attention and after the cross-attention segment ("after_self", "after_cross")."""
        if not (h.is_cuda and h.dtype == torch.float32 and h.is_contiguous()):
            raise ValueError("block_forward_ needs a contiguous fp32 CUDA tensor")
        B, N, _ = h.shape
        ws = self.workspace(B, N, cond.T)
        c, s = rope if rope is not None else self.rope_tables(N)
        ta = tb = None
        if taps is not None:
            ta, tb = torch.empty_like(h), torch.empty_like(h)
            taps.update(after_self=ta, after_cross=tb)
        hip.check(self.lib.ditto_block_forward_taps(self.handle, layer, h.data_ptr(), cond.buf.data_ptr(),
                                                    layer if cond_layer is None else cond_layer, B, N, cond.T,
                                                    c.data_ptr(), s.data_ptr(), _ptr(ta), _ptr(tb), ws.data_ptr(),
                                                    ws.numel(), _stream()))
        return h

    # ------------------------------------------------------------------ training
    def _weights_struct(self, state: Mapping[str, torch.Tensor]):
        """ditto_weights over the CURRENT fp32 CUDA tensors of `state` (no copies when they already are fp32,
        contiguous and on this device: the nn.Parameters themselves).  Returns (struct, keepalive)."""
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
        for f, k in hip.GLOBAL_KEY.items():
            setattr(w, f, dev(k))
        w.layers = layers
        keep.append(layers)
        return w, keep

    def train_attach(self, state: Mapping[str, torch.Tensor]):
        """Pack the transposed weight copies the dgrad GEMMs read (after every weight change)."""
        nb = self.lib.ditto_train_arena_bytes(C.byref(self._ccfg))
        if getattr(self, "_train_arena", None) is None:
            self._train_arena = torch.empty(nb, dtype=torch.uint8, device=self.device)
            self._tapes = {}
            self._train_ws = None
        w, keep = self._weights_struct(state)
        hip.check(self.lib.ditto_train_attach(self.handle, C.byref(w), self._train_arena.data_ptr(), nb, _stream()))
        torch.cuda.current_stream().synchronize()
        self._train_attached = True

    def _train_workspace(self, B, N, T):
        need = self.lib.ditto_train_workspace_bytes(C.byref(self._ccfg), B, N, T)
        if need == 0:
            raise hip.DittoHipError(hip.ERR_SHAPE, self.lib.ditto_last_error().decode())
        if self._train_ws is None or self._train_ws.numel() < need:
            self._train_ws = None
            self._train_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._train_ws

    def _take_tape(self, B, N, T):
        need = self.lib.ditto_tape_bytes(C.byref(self._ccfg), B, N, T)
        pool = self._tapes.setdefault((B, N, T), [])
        return pool.pop() if pool else torch.empty(need, dtype=torch.uint8, device=self.device)

    def release_tape(self, tape, B, N, T):
        self._tapes.setdefault((B, N, T), []).append(tape)

    def train_forward(self, x, text_emb, t, dropout_p: float, seed: int, opts: Optional[hip.CallOpts] = None):
        """DiTTO.forward in train mode: returns (eps fp32 [B,N,d], tape).  The library records against the tape how it wrote it
        (bf16 or fp32 stream rows): train_backward reads it that way whatever the options are by then."""
        if not getattr(self, "_train_attached", False):
            raise RuntimeError("train_attach() has not been called for the current weights")
        xf, text = self._f32(x, "x"), self._f32(text_emb, "text_emb")
        B, N, d = xf.shape
        T = text.shape[1]
        if d != self.cfg.hidden_dim or text.shape[0] != B or text.shape[2] != self.cfg.text_dim:
            raise ValueError("x / text_emb shapes do not match the model")
        tt = self._t64(t, B)
        tape = self._take_tape(B, N, T)
        ws = self._train_workspace(B, N, T)
        c, s = self.rope_tables(N)
        out = torch.empty_like(xf)
        self._call("ditto_train_forward", self.handle, xf.data_ptr(), text.data_ptr(), tt.data_ptr(), B, N, T,
                                                    c.data_ptr(), s.data_ptr(), float(dropout_p), int(seed), out.data_ptr(),
                                                    tape.data_ptr(), tape.numel(), ws.data_ptr(), ws.numel(), _stream(), opts=opts)
        return out, tape, xf, tt

    def train_backward(self, state: Mapping[str, torch.Tensor], grad_eps, xf, tt, T: int, tape, dropout_p: float,
                       seed: int, opts: Optional[hip.CallOpts] = None, piece_cb=None,
                       layers_per_piece: int = 1) -> Dict[str, torch.Tensor]:
        """Backward of train_forward: fp32 gradients keyed by the reference state_dict names (fresh tensors).

        `piece_cb` (optional): the backward runs in pieces of `layers_per_piece` layers, top layer first
        (ditto_train_backward_layers: bit-identical to the single call), and after each piece is ENQUEUED piece_cb(list of that
        piece's gradient tensors) is called — dist.GradSync.reduce starts their data-parallel exchange on a side stream while
        the layers below are still being computed."""
        B, N, d = xf.shape
        g = self._f32(grad_eps, "grad_output")
        w, keep = self._weights_struct(state)
        keys = [f"blocks.{l}.{k}" for l in range(self.cfg.num_layers) for k in hip.LAYER_KEY.values()] + \
               [k for f, k in hip.GLOBAL_KEY.items() if f != "rotary_inv_freq"]
        # one tensor of its own per parameter (the allocator aligns to 512 B): autograd's AccumulateGrad takes such a
        # gradient over as .grad without a copy; views of one flat buffer cost a copy kernel per parameter per step
        grads = {k: torch.empty(state[k].shape, dtype=torch.float32, device=self.device) for k in keys}
        L = self.cfg.num_layers
        layers = (hip.LayerGrads * L)()
        for l in range(L):
            for f, k in hip.LAYER_KEY.items():
                setattr(layers[l], f, grads[f"blocks.{l}.{k}"].data_ptr())
        gs = hip.Grads()
        for f, k in hip.GLOBAL_KEY.items():
            if f != "rotary_inv_freq":
                setattr(gs, f, grads[k].data_ptr())
        gs.layers = layers
        ws = self._train_workspace(B, N, T)
        c, s = self.rope_tables(N)
        args = (self.handle, C.byref(w), g.data_ptr(), xf.data_ptr(), tt.data_ptr(), B, N, T, c.data_ptr(), s.data_ptr(),
                float(dropout_p), int(seed), tape.data_ptr(), tape.numel(), C.byref(gs), ws.data_ptr(), ws.numel(), _stream())
        if piece_cb is None:
            self._call("ditto_train_backward", *args, opts=opts)
            return grads
        if not hip._has_call_opts():
            # a frozen pre-ABI-9 library (DITTO_HIP_LIB) has no ditto_train_backward_layers: one call, then every tensor as one piece
            # (the caller's exchange then simply runs after the backward instead of under it)
            self._call("ditto_train_backward", *args, opts=opts)
            piece_cb(list(grads.values()))
            return grads
        head = ("proj_in.weight", "proj_in.bias", "proj_out.weight", "proj_out.bias")     # written by the piece that starts at the top
        tail = [k for f, k in hip.GLOBAL_KEY.items() if f != "rotary_inv_freq" and k not in head]   # ... that ends at layer 0
        step = max(int(layers_per_piece), 1)
        hi = L - 1
        while hi >= 0:
            lo = max(hi - step + 1, 0)
            hip.check(self.lib.ditto_train_backward_layers(*args, None if opts is None else C.byref(opts), hi, lo))
            piece = [grads[k] for k in head] if hi == L - 1 else []
            piece += [grads[f"blocks.{l}.{k}"] for l in range(hi, lo - 1, -1) for k in hip.LAYER_KEY.values()]
            if lo == 0:
                piece += [grads[k] for k in tail]
            piece_cb(piece)
            hi = lo - 1
        return grads

    # ------------------------------------------------------------------ profiling (bench.py)
    def profile_enable(self, on: bool):
        hip.check(self.lib.ditto_profile_enable(self.handle, 1 if on else 0))

    def profile_read(self):
        n = (C.c_int32 * hip.KC_COUNT)()
        ms = (C.c_float * hip.KC_COUNT)()
        hip.check(self.lib.ditto_profile_read(self.handle, n, ms))
        return {hip.KERNEL_CLASSES[i]: (int(n[i]), float(ms[i])) for i in range(hip.KC_COUNT)}
