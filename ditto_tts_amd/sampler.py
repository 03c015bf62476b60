"""The DDPM sampling loop over the HIP denoise path.

Mirrors the sampling surface of the reference's `SpeechGenerator`
(reference src/model/SpeechGenerator.py): constructor arguments `:18-27`, attributes `ditto_model`, `betas`,
`alphas`, `alphas_cumprod`, `device` `:70-72`, and the two (name-mangled) methods the notebooks reach,
`_SpeechGenerator__p_sample` `:130-147` and `_SpeechGenerator__sample_latents` `:149-164`.

Out of scope: the EnCodec / GPT-2 / BigVGAN pieces around the loop and the SLP's encoders (its decoder
stack is ditto_tts_amd/slp.py).  They are taken from
`ditto_model.nac` and from injected `vocoder` / `text_tokenizer` / `audio_processor` objects when present and
raise a clear error otherwise; nothing here re-implements them.

Differences from the reference, both deliberate and documented in SURVEY.md App. B:
  * B-8: the loop length is the caller's `utils.Config.ConfigDiTTO.DIFFUSION_STEPS` read at call time, as in the
    reference, when that module is imported (bounds-checked against the frozen tables instead of overrunning them);
    with `diffusion_steps=` or stand-alone it is this object's own table length;
  * B-9: RoPE tables and the cross-attention K/V of the unchanging text are computed once per call to
    `__sample_latents`, not once per step.
RNG: `torch.randn_like` on the state's device, in the reference's call order (one draw for x_T unless
`cond_by_audio`, then one draw per step).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from .modules import DiTTO


class SpeechGenerator:
    def __init__(self, lambda_factor=0.1, nac_model_path=None, ditto_model_path=None, slp_path=None,
                 sample_rate=24000, device="cuda", *, ditto_model: Optional[DiTTO] = None, config=None,
                 diffusion_steps: Optional[int] = None, vocoder=None, mel_fn: Optional[Callable] = None,
                 text_tokenizer=None, audio_processor=None, slp=None):
        self.device = device
        if ditto_model is None:
            if config is None:
                # the CALLER's utils.Config.ConfigDiTTO when its tree is importable (so a notebook's mutation of the
                # class attributes is what gets read, reference src/Experiments.ipynb cell 6), else the shipped values
                from .shipped_config import config_classes
                config = config_classes()[0]
            ditto_model = DiTTO(hidden_dim=config.HIDDEN_DIM, num_layers=config.NUM_LAYERS,
                                num_heads=config.NUM_HEADS, time_dim=config.TIME_DIM,
                                text_dim=config.TEXT_EMBED_DIM, diffusion_steps=config.DIFFUSION_STEPS,
                                lambda_factor=lambda_factor, nac_model_path=nac_model_path)
            if ditto_model_path is not None:
                info = torch.load(ditto_model_path, map_location="cpu")
                ditto_model.load_state_dict(info["model_state_dict"], strict=ditto_model.nac is not None)
        self.ditto_model = ditto_model.to(self.device).eval()
        if slp is None and slp_path is not None:
            # reference :54-62: SLP(ConfigSLP.NB_CLASSES, NUM_HEADS, NUM_LAYERS) + its checkpoint.  Only the decoder
            # stack and the head are built here (ditto_tts_amd/slp.py); the checkpoint's text_encoder.* /
            # audio_encoder.* entries belong to the pretrained encoders and are skipped.
            from .shipped_config import config_classes
            ConfigSLP = config_classes()[1]
            from .slp import SLP
            slp = SLP(ConfigSLP.NB_CLASSES, ConfigSLP.NUM_HEADS, ConfigSLP.NUM_LAYERS,
                      hidden_size=ConfigSLP.EMBEDDING_DIM)
            info = torch.load(slp_path, map_location="cpu")
            res = slp.load_state_dict(info["model_state_dict"], strict=False)
            if res.missing_keys:
                raise KeyError(f"SLP checkpoint {slp_path} lacks {res.missing_keys[:3]}...")
            slp = slp.to(self.device).eval()
        self.sample_rate = sample_rate
        self.vocoder, self.mel_fn, self.slp = vocoder, mel_fn, slp
        self.text_tokenizer, self.audio_processor = text_tokenizer, audio_processor
        # `diffusion_steps=` (an extension) pins the loop length; without it the reference's rule applies: the tables
        # are frozen here from the model's step count (reference :70 reads ConfigDiTTO.DIFFUSION_STEPS, which is also
        # what the model was built from, :32-36) and the LOOP length is read from the caller's mutable
        # ConfigDiTTO.DIFFUSION_STEPS at call time (reference :161), see `_loop_steps`.
        self._pinned_steps = diffusion_steps is not None
        steps = diffusion_steps if diffusion_steps is not None else self.ditto_model.cfg.diffusion_steps
        if steps > self.ditto_model.cfg.diffusion_steps:
            raise ValueError("diffusion_steps exceeds the rows of the model's t_embedding")
        # reference src/model/SpeechGenerator.py:70-72
        self.betas = self.ditto_model.cosine_beta_schedule(steps).to(self.device)
        self.alphas = (1.0 - self.betas).to(self.device)
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0).to(self.device)

    @property
    def diffusion_steps(self) -> int:
        return int(self.betas.shape[0])

    def _loop_steps(self) -> int:
        """Number of reverse steps of one call to __sample_latents.  Reference src/model/SpeechGenerator.py:161 reads
        the global `ConfigDiTTO.DIFFUSION_STEPS` at CALL time while the tables were frozen at construction (SURVEY
        App. B-8).  When the caller's `utils.Config` module is imported and this object was not pinned with
        `diffusion_steps=`, that read is reproduced (a shorter count starts the loop at that t, as the reference does;
        a longer one would index past the tables there — here it raises with a message).  Otherwise the loop length is
        the table length."""
        if not self._pinned_steps:
            from .shipped_config import caller_config
            mod = caller_config(import_it=False)
            if mod is not None:
                n = int(mod.ConfigDiTTO.DIFFUSION_STEPS)
                if n > self.diffusion_steps:
                    raise IndexError(f"ConfigDiTTO.DIFFUSION_STEPS = {n} exceeds the {self.diffusion_steps}-entry "
                                     "schedule tables frozen when this SpeechGenerator was built")
                return n
        return self.diffusion_steps

    # ---------------------------------------------------------------- the hot loop
    @torch.no_grad()
    def __p_sample(self, x, t, text_emb, noise=None):
        """Reverse diffusion step (reference :130-147).  Returns a new tensor, like the reference."""
        m = self.ditto_model
        cond = m.text_cond(text_emb, x.shape[1])
        if noise is None:
            noise = torch.randn_like(x)
        x_prev = x.detach().float().contiguous().clone()
        m.engine().p_sample_(x_prev, cond, t, noise, self.betas, self.alphas, self.alphas_cumprod)
        return x_prev

    @torch.no_grad()
    def __sample_latents(self, text_emb, audio_emb, text_prompt=None, audio=None, is_slp=False, cond_by_audio=False,
                         noises=None, keep=None, use_graph=None, seeds=None, batch_class=None):
        """All reverse diffusion steps (reference :149-164).

        `noises` (optional): a sequence / callable giving the z of executed step i, for parity tests;
        `keep` (optional): dict filled with {i: state after step i} for the i it already has as keys;
        `seeds` (optional, int64 [B]): per-utterance seeds.  x_T and every step's z then come from the library's
        counter-based generator (Philox4x32-10 keyed by the utterance's seed, engine.noise_normal_), generated inside
        the update kernel: an utterance's trajectory is a function of (seed, text, weights) and of the KERNEL CLASS its
        launches take (batches of 11 .. 15 and >= 17 utterances of 1024 frames — csrc/kernels.h fr_rule_rows — run two GEMM + LayerNorm pairs per block on a full-row kernel,
        which sums over k in another order).  Default None = the reference's behaviour, torch.randn_like from the global
        generator;
        `batch_class` (optional int): the number of utterances of the UNSPLIT batch this call is a piece of.  The loop
        then passes hip.CallOpts(class_rows = batch_class * N) to every step's call, every launch picks the class that batch would pick, and an
        utterance's latents are the same bits whatever piece or GPU it is sampled on (dist.sample_sharded pins the
        class itself: do not pass it there);
        `use_graph`: replay the step from a HIP graph (default off: measured no gain even at B = 1, the step is
        bound by per-kernel latency, not by its 122 launches; bit-identical to eager either way)."""
        if is_slp:
            raise NotImplementedError("the is_slp branch is broken in the reference (it passes the predictor's logits "
                                      "as a tensor shape, SURVEY.md App. B-6); length logits are available from "
                                      "self.slp.decode(z_text, z_audio)")
        m = self.ditto_model
        if seeds is not None and (noises is not None or use_graph):
            raise ValueError("seeds= excludes noises= and use_graph=")
        if seeds is not None and not cond_by_audio:
            x = torch.empty(audio_emb.shape, dtype=torch.float32, device=self.device)
            seeds = seeds.to(self.device).long().contiguous()
            m.engine(x.device).noise_normal_(x, seeds, 0xFFFFFFFF)        # x_T: step index no loop step uses
        else:
            x = torch.randn_like(audio_emb) if not cond_by_audio else audio_emb.clone()
            x = x.to(self.device).float().contiguous()
            if seeds is not None:
                seeds = seeds.to(self.device).long().contiguous()
        eng = m.engine(x.device)
        cond = m.text_cond(text_emb.to(x.device), x.shape[1])
        B = x.shape[0]
        t_tensor = torch.empty(B, device=x.device, dtype=torch.long)
        use_graph = bool(use_graph)
        z = torch.empty_like(x)
        n_loop = self._loop_steps()
        opts = None
        if batch_class is not None:                 # every step of the loop is CALLED with the unsplit batch's kernel class: a
            from .hip import CallOpts               # per-call argument (ditto_call_opts), nothing process-wide changes
            opts = CallOpts(class_rows=int(batch_class) * x.shape[1])
        return self.__loop(x, cond, t_tensor, z, use_graph, n_loop, seeds, noises, keep, eng, opts)

    def __loop(self, x, cond, t_tensor, z, use_graph, n_loop, seeds, noises, keep, eng, opts=None):
        graph = None
        if use_graph:
            t_tensor.fill_(n_loop - 1)
            keep_x = x.clone()                      # capture runs one warm-up step on x: restore it afterwards
            z.zero_()
            graph = eng.capture_p_sample(x, cond, t_tensor, z, self.betas, self.alphas, self.alphas_cumprod, opts=opts)
            x.copy_(keep_x)
        for i, t_val in enumerate(reversed(range(n_loop))):
            t_tensor.fill_(t_val)
            if seeds is not None:
                eng.p_sample_seeded_(x, cond, t_tensor, seeds, t_val, self.betas, self.alphas, self.alphas_cumprod, opts=opts)
                if keep is not None and i in keep:
                    keep[i] = x.clone()
                continue
            if noises is None:
                z.normal_()                          # same generator stream as randn_like(x)
            else:
                z.copy_((noises(i) if callable(noises) else noises[i]).to(x.device))
            if graph is not None:
                graph.replay()
            else:
                eng.p_sample_(x, cond, t_tensor, z, self.betas, self.alphas, self.alphas_cumprod, opts=opts)
            if keep is not None and i in keep:
                keep[i] = x.clone()
        return x

    # ---------------------------------------------------------------- strided (DDIM) loop + CFG  
    @torch.no_grad()
    def sample_latents_strided(self, text_emb, audio_emb, n_steps=25, eta=0.0, cfg_scale=None, null_text_emb=None,
                               cond_by_audio=False, noises=None):
        """The serving configuration of the paper (App. A: 25 steps, guidance 5.0), which the reference lacks: a
        DDIM-style loop over `n_steps` evenly spaced timesteps, x' = a x + ce eps + cz z per step
        (ditto_linear_update), with optional classifier-free guidance: the step runs ONE forward on the doubled
        batch [x; x] x [text; null_text] and combines eps_u + w (eps_c - eps_u) (ditto_cfg_combine)."""
        from .around import cfg_combine, linear_update_
        m = self.ditto_model
        T = self.diffusion_steps
        if not 1 <= n_steps <= T:
            raise ValueError("n_steps must be in [1, diffusion_steps]")
        x = (torch.randn_like(audio_emb) if not cond_by_audio else audio_emb.clone()).to(self.device).float().contiguous()
        B = x.shape[0]
        eng = m.engine(x.device)
        text = text_emb.to(x.device).float()
        if cfg_scale is not None:
            if null_text_emb is None:
                raise ValueError("classifier-free guidance needs null_text_emb (the unconditional text embedding)")
            text = torch.cat([text, null_text_emb.to(x.device).float().expand_as(text)], dim=0).contiguous()
        cond = eng.prepare_text(text, x.shape[1])
        stride = T / n_steps
        taus = [int(round(T - 1 - i * stride)) for i in range(n_steps)]
        ac = self.alphas_cumprod.double().cpu()
        nb = 2 * B if cfg_scale is not None else B
        t_tensor = torch.empty(nb, device=x.device, dtype=torch.long)
        coef = torch.empty(3, B, device=x.device, dtype=torch.float32)
        z = torch.empty_like(x)
        x2 = torch.empty(nb, *x.shape[1:], device=x.device) if cfg_scale is not None else None
        for i, t_val in enumerate(taus):
            ab_t = ac[t_val]
            ab_p = ac[taus[i + 1]] if i + 1 < n_steps else torch.tensor(1.0, dtype=torch.float64)
            sigma = eta * torch.sqrt((1 - ab_p) / (1 - ab_t)) * torch.sqrt(1 - ab_t / ab_p)
            a = torch.sqrt(ab_p / ab_t)
            ce = torch.sqrt(torch.clamp(1 - ab_p - sigma ** 2, min=0.0)) - torch.sqrt(ab_p * (1 - ab_t) / ab_t)
            coef[0].fill_(float(a)); coef[1].fill_(float(ce)); coef[2].fill_(float(sigma))
            t_tensor.fill_(t_val)
            if cfg_scale is not None:
                x2[:B].copy_(x); x2[B:].copy_(x)
                eps = cfg_combine(eng.forward(x2, cond, t_tensor), cfg_scale)
            else:
                eps = eng.forward(x, cond, t_tensor)
            use_noise = float(sigma) != 0.0
            if use_noise:
                if noises is None:
                    z.normal_()
                else:
                    z.copy_((noises(i) if callable(noises) else noises[i]).to(x.device))
            linear_update_(x, eps, z if use_noise else None, coef[0], coef[1], coef[2])
        return x

    # public aliases (the mangled names above are what the reference's own code reaches)
    def p_sample(self, x, t, text_emb, noise=None):
        return self.__p_sample(x, t, text_emb, noise)

    def sample_latents(self, text_emb, audio_emb, **kw):
        return self.__sample_latents(text_emb, audio_emb, **kw)

    # ---------------------------------------------------------------- around the loop (out of scope: delegated)
    def _need(self, what, obj):
        if obj is None:
            raise RuntimeError(f"SpeechGenerator.{what} is not available: the codec / vocoder / tokenizer stack is "
                               "outside the MI355X denoise path (SURVEY.md §2) and was not injected")
        return obj

    @torch.no_grad()
    def generate_speech_from_audio_tensor(self, audio_tensor, padding_mask_audio, text_prompt, is_tokenized=False,
                                          is_slp=False, cond_by_audio=False):
        """Reference :93-111, using the caller-provided codec (`ditto_model.nac`) and vocoder."""
        nac = self._need("ditto_model.nac", self.ditto_model.nac)
        audio_latents, audio_scales = nac.audio_encoder(audio_tensor, padding_mask_audio)
        max_length = nac.language_model.config.n_positions
        audio_latents = audio_latents[:, :, :max_length].mean(dim=1)
        if not is_tokenized:
            tok = self._need("text_tokenizer", self.text_tokenizer)
            text_tokens = tok(text_prompt, return_tensors="pt").input_ids.to(self.device)
        else:
            text_tokens = text_prompt
        text_tokens = text_tokens[:, :max_length]
        text_embeddings = nac.language_model.transformer.wte(text_tokens)
        t = torch.full((audio_latents.size(0),), self._loop_steps() - 1, device=self.device, dtype=torch.long)  # :105
        audio_latents = self.ditto_model.q_sample(audio_latents, t)
        refined = self.__sample_latents(text_embeddings, audio_latents, text_tokens, audio_tensor, is_slp, cond_by_audio)
        return self.__generate_speech_from_latents(refined, audio_scales, padding_mask_audio)

    @torch.no_grad()
    def __generate_speech_from_latents(self, audio_latents, audio_scales, padding_mask_audio):
        """Reference :114-128 (VQ -> EnCodec decode -> mel -> vocoder), all delegated."""
        nac = self._need("ditto_model.nac", self.ditto_model.nac)
        vocoder, mel_fn = self._need("vocoder", self.vocoder), self._need("mel_fn", self.mel_fn)
        audio_latents = audio_latents.unsqueeze(1).repeat(1, 2, 1, 1)
        q = nac.vector_quantizer(audio_latents)
        waveform = nac.audio_decoder.decode(q.unsqueeze(0).detach(), audio_scales=audio_scales,
                                            padding_mask=padding_mask_audio)[0].squeeze(1)
        return vocoder(mel_fn(waveform, vocoder.h).to(self.device)).squeeze(0)

    def generate_speech_from_file(self, file_path, text_prompt, cond_by_audio=False):
        raise RuntimeError("generate_speech_from_file needs torchaudio + the EnCodec processor, which are outside the "
                           "denoise path; load the audio yourself and call generate_speech_from_audio_tensor")
