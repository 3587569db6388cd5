"""``GaussianDiffusion`` sampling API of the reference (text2motion/models/gaussian_diffusion.py) on HIP.

Kept: constructor keywords, the f64 numpy schedule attributes, ``p_sample_loop_with_cfg`` (:1100-1141, the loop the
trainer uses), ``ddim_sample_loop`` (:744-774), ``p_sample_loop`` (:616-646, with the evidently intended noise draw:
the reference calls ``randn_like`` with a shape and raises TypeError at :606), and the single-step methods.
Only the configuration the trainer builds is implemented on the device -- EPSILON mean, FIXED_SMALL / FIXED_LARGE
variance (ddpm_trainer.py:45-50); learned-variance / x0-prediction variants and the training losses are out of
scope and raise NotImplementedError.

One denoising step = one C call chain captured into a hipGraph:
   [cond | uncond] rows batched as 2B through mdm_denoiser_forward  ->  mdm_cfg_posterior_step (guidance on
   pred_xstart, posterior mean, noise)  ->  t -= 1 on the device.
The unconditional text embedding is encoded once and cached instead of re-running the text encoder on [""]*B every
step (gaussian_diffusion.py:1059-1062); in eval mode that is bit-identical.
"""
from __future__ import annotations

import ctypes as C
import enum
import math
from typing import Callable, Optional

import numpy as np
import torch

from . import _lib as L


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()
    START_X = enum.auto()
    EPSILON = enum.auto()


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()


class LossType(enum.Enum):
    MSE = enum.auto()
    RESCALED_MSE = enum.auto()
    KL = enum.auto()
    RESCALED_KL = enum.auto()

    def is_vb(self):
        return self in (LossType.KL, LossType.RESCALED_KL)


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    return np.array([min(1 - alpha_bar((i + 1) / num_diffusion_timesteps) / alpha_bar(i / num_diffusion_timesteps), max_beta)
                     for i in range(num_diffusion_timesteps)])


def get_named_beta_schedule(schedule_name: str, num_diffusion_timesteps: int) -> np.ndarray:
    """gaussian_diffusion.py:19-55 ('linear' is what the trainer uses; scaled by 1000/steps)."""
    n = num_diffusion_timesteps
    if schedule_name == "linear":
        scale = 1000 / n
        return np.linspace(scale * 0.0001, scale * 0.02, n, dtype=np.float64)
    if schedule_name == "cosine":
        return betas_for_alpha_bar(n, lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    if schedule_name == "sqrt":
        alphas = np.linspace(1.0, 0.0, n, dtype=np.float64)
        betas = 1 - alphas ** 2
        betas = (betas - betas.min()) / (betas.max() - betas.min())
        return betas * (0.02 - 0.0001) + 0.0001
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


class GaussianDiffusion:
    def __init__(self, *, betas, model_mean_type, model_var_type, loss_type, rescale_timesteps=False, cfg_scale=7.5):
        self.model_mean_type, self.model_var_type, self.loss_type = model_mean_type, model_var_type, loss_type
        self.rescale_timesteps, self.cfg_scale = rescale_timesteps, cfg_scale
        betas = np.array(betas, dtype=np.float64)
        self.betas = betas
        assert betas.ndim == 1, "betas must be 1-D"
        assert (betas > 0).all() and (betas <= 1).all()
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)
        self._tab_cache = {}

    # ---- host logic ---------------------------------------------------------------------------------
    def schedule_table(self) -> np.ndarray:
        """fp32 [7, steps] table handed to the step kernels: each f64 entry rounded to f32 exactly as
        _extract_into_tensor does (gaussian_diffusion.py:329-341)."""
        if self.model_var_type == ModelVarType.FIXED_SMALL:
            logvar = self.posterior_log_variance_clipped
        elif self.model_var_type == ModelVarType.FIXED_LARGE:
            logvar = np.log(np.append(self.posterior_variance[1], self.betas[1:]))
        else:
            raise NotImplementedError("learned-variance models are out of scope of the HIP sampler")
        rows = [self.sqrt_recip_alphas_cumprod, self.sqrt_recipm1_alphas_cumprod, self.posterior_mean_coef1,
                self.posterior_mean_coef2, logvar, self.alphas_cumprod, self.alphas_cumprod_prev]
        return np.stack(rows).astype(np.float32)

    def _device_table(self, device) -> torch.Tensor:
        key = str(device)
        if key not in self._tab_cache:
            self._tab_cache[key] = torch.from_numpy(self.schedule_table()).to(device).contiguous()
        return self._tab_cache[key]

    def _check_supported(self, denoised_fn=None, cond_fn=None):
        if self.model_mean_type != ModelMeanType.EPSILON:
            raise NotImplementedError("only epsilon-prediction models are implemented (ddpm_trainer.py:47)")
        if denoised_fn is not None or cond_fn is not None:
            raise NotImplementedError("denoised_fn / cond_fn hooks are not supported by the fused HIP step")
        if self.rescale_timesteps:
            raise NotImplementedError("rescale_timesteps would feed float timesteps; the denoiser takes int64 steps")

    def _scale_timesteps(self, t):
        return t

    # ---- elementwise helpers of the reference API (gaussian_diffusion.py:329-341,433-475,554-571) -----------------
    # Plumbing around the HIP forward: table lookups rounded f64 -> f32 exactly as _extract_into_tensor does, evaluated
    # with device tensor ops.  The sampling loops do NOT go through these (they use the fused step kernels).
    def _extract(self, arr: np.ndarray, t: torch.Tensor, shape) -> torch.Tensor:
        res = torch.from_numpy(np.asarray(arr)).to(t.device)[t.long()].float()
        while res.dim() < len(shape):
            res = res[..., None]
        return res.expand(shape)

    def q_mean_variance(self, x_start, t):
        mean = self._extract(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
        variance = self._extract(1.0 - self.alphas_cumprod, t, x_start.shape)
        log_variance = self._extract(self.log_one_minus_alphas_cumprod, t, x_start.shape)
        return mean, variance, log_variance

    def q_sample(self, x_start, t, noise=None):
        if noise is None:
            noise = torch.randn_like(x_start)
        assert noise.shape == x_start.shape
        return (self._extract(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
                + self._extract(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

    def q_posterior_mean_variance(self, x_start, x_t, t):
        assert x_start.shape == x_t.shape
        mean = (self._extract(self.posterior_mean_coef1, t, x_t.shape) * x_start
                + self._extract(self.posterior_mean_coef2, t, x_t.shape) * x_t)
        var = self._extract(self.posterior_variance, t, x_t.shape)
        logvar = self._extract(self.posterior_log_variance_clipped, t, x_t.shape)
        return mean, var, logvar

    def _predict_xstart_from_eps(self, x_t, t, eps):
        assert x_t.shape == eps.shape
        return (self._extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t
                - self._extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * eps)

    def _predict_eps_from_xstart(self, x_t, t, pred_xstart):
        return ((self._extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - pred_xstart)
                / self._extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape))

    @torch.no_grad()
    def p_mean_variance(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        """One denoiser forward (HIP) + the epsilon / fixed-variance posterior of gaussian_diffusion.py:481-552.
        Returns {"mean", "variance", "log_variance", "pred_xstart"}."""
        self._check_supported(denoised_fn)
        if model_kwargs is None:
            model_kwargs = {}
        B = x.shape[0]
        assert t.shape == (B,)
        eps = model(x, self._scale_timesteps(t), **model_kwargs)
        if self.model_var_type == ModelVarType.FIXED_SMALL:
            var_tab, logvar_tab = self.posterior_variance, self.posterior_log_variance_clipped
        elif self.model_var_type == ModelVarType.FIXED_LARGE:
            var_tab = np.append(self.posterior_variance[1], self.betas[1:])
            logvar_tab = np.log(var_tab)
        else:
            raise NotImplementedError("learned-variance models are out of scope of the HIP sampler")
        variance = self._extract(var_tab, t, x.shape)
        log_variance = self._extract(logvar_tab, t, x.shape)
        pred_xstart = self._predict_xstart_from_eps(x, t, eps)
        if clip_denoised:
            pred_xstart = pred_xstart.clamp(-1, 1)
        mean, _, _ = self.q_posterior_mean_variance(pred_xstart, x, t)
        assert mean.shape == log_variance.shape == pred_xstart.shape == x.shape
        return {"mean": mean, "variance": variance, "log_variance": log_variance, "pred_xstart": pred_xstart}

    @torch.no_grad()
    def training_losses(self, model, x_start, t, model_kwargs=None, noise=None):
        """The MSE branch of gaussian_diffusion.py:923-985 evaluated with the HIP forward: resets the MoE counters,
        diffuses ``x_start`` to ``x_t``, runs the denoiser and returns {"mse", "target", "pred", "moe_loss"}.  These are
        VALUES (validation loss, routing balance): the HIP path has no backward, training is outside this build."""
        if self.loss_type not in (LossType.MSE, LossType.RESCALED_MSE):
            raise NotImplementedError("only the MSE losses of the epsilon model are evaluated (ddpm_trainer.py:47)")
        self._check_supported()
        if model_kwargs is None:
            model_kwargs = {}
        if noise is None:
            noise = torch.randn_like(x_start)
        x_t = self.q_sample(x_start, t, noise=noise)
        terms = {}
        model.reset_all_moe_counters(model)
        model_output = model(x_t, self._scale_timesteps(t), **model_kwargs)
        target = noise  # ModelMeanType.EPSILON
        assert model_output.shape == target.shape == x_start.shape
        terms["mse"] = ((target - model_output) ** 2).mean(dim=list(range(1, x_start.dim()))).view(-1)
        terms["target"], terms["pred"] = target, model_output
        terms["moe_loss"] = model.get_moe_loss(model)
        return terms

    def _progressive(self, r, noise, step_noise):
        """Generator form of _StepRunner.run: yields {"sample", "pred_xstart"} after every step (eager launches)."""
        B = r.B
        r._prepare()
        if noise is None:
            noise = torch.randn((B, r.T, r.Fe), device=r.dev)
        r.xx[:B].copy_(noise.to(r.dev, torch.float32))
        r.t_dev.fill_(self.num_timesteps - 1)
        for i in range(self.num_timesteps):
            if r._needs_noise():
                if step_noise is not None:
                    r.noise.copy_(step_noise[i].to(r.dev, torch.float32))
                else:
                    r.noise.normal_()
            r._step(r._needs_noise())
            yield {"sample": r.xx[:B].clone(), "pred_xstart": r.x0.clone()}

    @torch.no_grad()
    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                  model_kwargs=None, device=None, progress=False, *, step_noise=None):
        self._check_supported(denoised_fn, cond_fn)
        r = self._runner(model, shape, model_kwargs, device, "ddpm", 0.0, 0.0, clip_denoised, False)
        yield from self._progressive(r, noise, step_noise)

    @torch.no_grad()
    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                     model_kwargs=None, device=None, progress=False, eta=0.0, *, step_noise=None):
        self._check_supported(denoised_fn, cond_fn)
        r = self._runner(model, shape, model_kwargs, device, "ddim", 0.0, eta, clip_denoised, False)
        yield from self._progressive(r, noise, step_noise)

    # ---- fused step drivers ----------------------------------------------------------------------------
    def _runner(self, model, shape, model_kwargs, device, mode: str, cfg_scale: float, eta: float, clip: bool,
                use_graph: bool, streams: int = 0):
        return _StepRunner(self, model, tuple(shape), model_kwargs or {}, device, mode, cfg_scale, eta, clip, use_graph,
                           streams)

    @torch.no_grad()
    def p_sample_loop_with_cfg(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, model_kwargs=None,
                               device=None, progress=False, cfg_scale=7.5, *, step_noise=None, use_graph=True,
                               callback: Optional[Callable] = None, seed: Optional[int] = None, sample_offset: int = 0):
        """Classifier-free-guided ancestral sampling.  ``step_noise``: optional list/tensor of per-step noise (the
        reference draws ``randn_like`` each step, :1094); ``callback(i, t, x)`` is called after every step."""
        self._check_supported(denoised_fn)
        r = self._runner(model, shape, model_kwargs, device, "cfg", cfg_scale, 0.0, clip_denoised, use_graph)
        return r.run(noise, step_noise, progress, callback, seed, sample_offset)

    @torch.no_grad()
    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                      model_kwargs=None, device=None, progress=False, before_step_fn=None, *, step_noise=None,
                      use_graph=True, seed: Optional[int] = None, sample_offset: int = 0):
        """Unguided ancestral sampling with the intended noise draw (the reference's version raises at :606)."""
        self._check_supported(denoised_fn, cond_fn)
        r = self._runner(model, shape, model_kwargs, device, "ddpm", 0.0, 0.0, clip_denoised, use_graph)
        cb = (lambda i, t, x: before_step_fn(t, x)) if before_step_fn is not None else None
        return r.run(noise, step_noise, progress, cb, seed, sample_offset)

    @torch.no_grad()
    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                         model_kwargs=None, device=None, progress=False, eta=0.0, *, step_noise=None, use_graph=True,
                         seed: Optional[int] = None, sample_offset: int = 0, callback: Optional[Callable] = None):
        self._check_supported(denoised_fn, cond_fn)
        r = self._runner(model, shape, model_kwargs, device, "ddim", 0.0, eta, clip_denoised, use_graph)
        return r.run(noise, step_noise, progress, callback, seed, sample_offset)

    # single steps (eager): same arithmetic, returns {"sample", "pred_xstart"}
    @torch.no_grad()
    def p_sample_with_cfg(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None, cfg_scale=7.5,
                          noise=None):
        self._check_supported(denoised_fn)
        r = self._runner(model, x.shape, model_kwargs, x.device, "cfg", cfg_scale, 0.0, clip_denoised, False)
        return r.single(x, t, noise)

    @torch.no_grad()
    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0,
                    noise=None):
        self._check_supported(denoised_fn, cond_fn)
        r = self._runner(model, x.shape, model_kwargs, x.device, "ddim", 0.0, eta, clip_denoised, False)
        return r.single(x, t, noise)

    @torch.no_grad()
    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, noise=None):
        self._check_supported(denoised_fn, cond_fn)
        r = self._runner(model, x.shape, model_kwargs, x.device, "ddpm", 0.0, 0.0, clip_denoised, False)
        return r.single(x, t, noise)


class _StepRunner:
    """Static buffers + (optionally) one captured hipGraph for a whole denoising step."""

    def __init__(self, diff: GaussianDiffusion, model, shape, kw, device, mode, cfg_scale, eta, clip, use_graph,
                 streams: int = 0):
        self.d, self.model, self.mode = diff, model, mode
        self.philox = None  # (seed, global index of row 0): per-step noise from the counter-based device generator
        self.ntok = None    # per-row text token counts when the cond / uncond captions tokenise to different lengths
        self.tcache = None
        self.nstreams = int(streams) if streams else int(getattr(model, "sampler_streams", 1))
        self.cfg_scale, self.eta, self.clip, self.use_graph = float(cfg_scale), float(eta), bool(clip), use_graph
        if device is None:
            device = next(model.parameters()).device
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise L.MdmError("the sampler runs on HIP kernels only: model and tensors must live on a GPU")
        B, T, Fe = shape
        self.B, self.T, self.Fe = B, T, Fe
        self.n = B * T * Fe
        length = kw.get("length")
        if length is None:
            raise ValueError("model_kwargs['length'] is required (ddpm_trainer.py:166-171)")
        length = torch.as_tensor(length).to(self.dev, torch.int32)
        if getattr(model, "ephemeral_mode", "frozen") == "resample":
            self.use_graph = False  # fresh random projections are drawn on the host before every forward
        xp, xo = kw.get("xf_proj"), kw.get("xf_out")
        if xp is None or xo is None:
            xp, xo = model.encode_text(kw["text"], self.dev)
        xp, xo = xp.to(self.dev, torch.float32), xo.to(self.dev, torch.float32)
        if mode == "cfg":  # cond rows then uncond rows of the same samples, one forward of 2B rows
            up, uo = kw.get("xf_proj_uncond"), kw.get("xf_out_uncond")
            if up is None or uo is None:
                up, uo = model.uncond_embedding(B, self.dev)
            up, uo = up.to(self.dev, torch.float32), uo.to(self.dev, torch.float32)
            self.len2 = torch.cat([length, length], 0)
            self.R = 2 * B
            # A real tokenizer gives the empty caption fewer tokens than the captions (N = 8 + 2 vs 8 + longest caption,
            # text_encoder.py:25-43) and the reference's cross-attention has no text mask, so the shorter side must NOT
            # see padding.  Default: the shorter half is padded with zero rows and the text cache carries a per-row token
            # count (MdmTextCache.ntok) under which those rows have weight exactly 0 in both cross-attentions -- still ONE
            # forward of 2B rows.  model.ragged_text = "split": two B-row forwards with their own text caches instead.
            ragged = uo.shape[1] != xo.shape[1]
            self.split_halves = ragged and (getattr(model, "ragged_text", "mask") == "split" or not hasattr(model, "prepare_text"))
            if self.split_halves:
                self.xp, self.xo = None, None
                self.halves = [(xp.contiguous(), xo.contiguous()), (up.contiguous(), uo.contiguous())]
            else:
                if ragged:
                    nmax = max(xo.shape[1], uo.shape[1])
                    self.ntok = [xo.shape[1]] * B + [uo.shape[1]] * B
                    xo = torch.nn.functional.pad(xo, (0, 0, 0, nmax - xo.shape[1]))
                    uo = torch.nn.functional.pad(uo, (0, 0, 0, nmax - uo.shape[1]))
                self.xp = torch.cat([xp, up], 0).contiguous()
                self.xo = torch.cat([xo, uo], 0).contiguous()
        else:
            self.xp, self.xo, self.len2, self.R = xp.contiguous(), xo.contiguous(), length, B
            self.split_halves = False
        self.xx = torch.empty((self.R, T, Fe), dtype=torch.float32, device=self.dev)  # model input rows
        self.eps = torch.empty_like(self.xx)
        self.noise = torch.empty((B, T, Fe), dtype=torch.float32, device=self.dev)
        self.x0 = torch.empty_like(self.noise)
        self.t_dev = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.ts = torch.zeros(self.R, dtype=torch.int64, device=self.dev)
        self.tab = diff._device_table(self.dev)
        self.graph = None
        # time-embedding chain tabulated per timestep + text half of the gated fusion: once per loop, not per step
        frozen = getattr(model, "ephemeral_mode", "frozen") == "frozen"
        can_cache = hasattr(model, "stem_cache") and frozen
        self.stem = model.stem_cache(diff.num_timesteps, self.xp) if can_cache and not self.split_halves else None
        # Samples never interact, so the R rows of a step can be cut into independent chunks whose forwards run
        # CONCURRENTLY on separate HIP streams (forked/joined inside the captured graph): most launches of a forward are
        # latency-bound, and two chains in flight overlap each other's prologues, DMA round trips and tails.
        self.chunks = None
        self.side = []
        if self.split_halves:
            if not hasattr(model, "prepare_text"):
                raise ValueError("cond and uncond text embeddings have different token counts; this model cannot run "
                                 "them as separate forwards")
            self.chunks = []
            for c, (xp_c, xo_c) in enumerate(self.halves):  # same stream, one after the other
                sl = slice(c * B, (c + 1) * B)
                self.chunks.append(dict(
                    sl=sl, xp=xp_c, xo=xo_c, len=self.len2[sl].contiguous(), tc=model.prepare_text(xo_c, private=True),
                    stem=model.stem_cache(diff.num_timesteps, xp_c) if can_cache else None,
                    ws=model.new_workspace(B, T, xo_c.shape[1]), stream=0))
        elif self.nstreams > 1 and frozen and hasattr(model, "new_workspace") and self.R % self.nstreams == 0:
            n = self.R // self.nstreams
            self.side = [torch.cuda.Stream(device=self.dev) for _ in range(self.nstreams - 1)]
            self.chunks = []
            for c in range(self.nstreams):
                sl = slice(c * n, (c + 1) * n)
                xo_c = self.xo[sl].contiguous()
                xp_c = self.xp[sl].contiguous()
                self.chunks.append(dict(
                    sl=sl, xp=xp_c, xo=xo_c, len=self.len2[sl].contiguous(),
                    tc=model.prepare_text(xo_c, private=True, ntok=self.ntok[sl] if self.ntok else None),
                    stem=model.stem_cache(diff.num_timesteps, xp_c),
                    ws=model.new_workspace(n, T, xo_c.shape[1]), stream=c))

    # one step on the current stream: reads self.xx[:B] (x_t), writes x_{t-1} back into it
    def _step(self, use_noise: bool):
        with torch.cuda.device(self.dev):  # kernels go to the current stream of the sampler's device
            self._step_on_device(use_noise)

    def _step_on_device(self, use_noise: bool):
        lib, s = L.lib(), C.c_void_p(L.stream_ptr())
        B, n = self.B, self.n
        if use_noise and self.philox is not None:  # step noise = f(seed, global sample, t, element); t read on the device
            self._philox_fill(self.noise, C.c_void_p(self.t_dev.data_ptr()), 0, s)
        x = self.xx[:B]
        if self.R == 2 * B:
            self.xx[B:].copy_(x)
        L.check(lib.mdm_fill_i64(C.c_void_p(self.ts.data_ptr()), C.c_int64(self.R), C.c_void_p(self.t_dev.data_ptr()), s))
        if self.chunks is not None:
            main = torch.cuda.current_stream()
            for i, ch in enumerate(self.chunks):
                st = main if ch["stream"] == 0 else self.side[ch["stream"] - 1]
                if st is not main:
                    st.wait_stream(main)  # fork: x_t rows and the timestep vector are ready
                with torch.cuda.stream(st):
                    self.model(self.xx[ch["sl"]], self.ts[ch["sl"]], ch["len"], xf_proj=ch["xp"], xf_out=ch["xo"],
                               out=self.eps[ch["sl"]], stem_cache=ch["stem"], text_cache=ch["tc"], workspace=ch["ws"])
            for st in self.side:
                main.wait_stream(st)  # join before the guidance / posterior update
        elif self.ntok is not None:  # ragged captions: a private text cache with per-row token counts
            if self.tcache is None:
                self.tcache = self.model.prepare_text(self.xo, private=True, ntok=self.ntok)
            # text_tokens as well: should the module's packed weights have been rebuilt since the cache was made, forward()
            # rebuilds the text side -- with these counts, not with the zero-padded rows taken for real tokens
            self.model(self.xx, self.ts, self.len2, xf_proj=self.xp, xf_out=self.xo, out=self.eps, stem_cache=self.stem,
                       text_cache=self.tcache, text_tokens=self.ntok)
        elif self.stem is not None:
            self.model(self.xx, self.ts, self.len2, xf_proj=self.xp, xf_out=self.xo, out=self.eps, stem_cache=self.stem)
        else:
            self.model(self.xx, self.ts, self.len2, xf_proj=self.xp, xf_out=self.xo, out=self.eps)
        noise = C.c_void_p(self.noise.data_ptr() if use_noise else 0)
        steps = C.c_int32(self.d.num_timesteps)
        if self.mode == "ddim":
            L.check(lib.mdm_ddim_step(C.c_void_p(x.data_ptr()), C.c_void_p(self.eps.data_ptr()), noise, C.c_int64(n),
                                      C.c_void_p(self.tab.data_ptr()), steps, C.c_void_p(self.t_dev.data_ptr()), C.c_int32(0),
                                      C.c_float(self.eta), C.c_int32(int(self.clip)), C.c_void_p(x.data_ptr()),
                                      C.c_void_p(self.x0.data_ptr()), s), "mdm_ddim_step")
        else:
            eps_u = self.eps[B:].data_ptr() if self.mode == "cfg" else 0
            L.check(lib.mdm_cfg_posterior_step(C.c_void_p(x.data_ptr()), C.c_void_p(self.eps.data_ptr()), C.c_void_p(eps_u),
                                               noise, C.c_int64(n), C.c_void_p(self.tab.data_ptr()), steps,
                                               C.c_void_p(self.t_dev.data_ptr()), C.c_int32(0), C.c_float(self.cfg_scale),
                                               C.c_int32(int(self.clip)), C.c_void_p(x.data_ptr()),
                                               C.c_void_p(self.x0.data_ptr()), s), "mdm_cfg_posterior_step")
        L.check(lib.mdm_add_i32(C.c_void_p(self.t_dev.data_ptr()), C.c_int32(-1), s))

    def _needs_noise(self) -> bool:
        return not (self.mode == "ddim" and self.eta == 0.0)

    def _prepare(self):
        """Pack weights, build the text cache, size the workspace -- everything that allocates -- before capture.
        The warm-up forward must not disturb the MoE counters the reference would show (switch_moe.py:71-92)."""
        with torch.cuda.device(self.dev):  # the synchronize below must be on the sampler's device as well
            saved = {k: v.clone() for k, v in self.model.moe_buffers().items()} if hasattr(self.model, "moe_buffers") else {}
            self.xx.zero_()
            self.t_dev.fill_(self.d.num_timesteps - 1)
            self.noise.zero_()
            self._step(self._needs_noise())
            for k, v in saved.items():
                self.model.moe_buffers()[k].copy_(v)
            torch.cuda.current_stream().synchronize()

    def _philox_fill(self, out, stream_dev, stream_imm, s):
        """out[row] = noise(seed, global sample of that row, stream): rows are consecutive samples (first + row) or carry
        explicit global indices (self.philox = (seed, int64 device tensor))."""
        seed, first = self.philox
        lib, per = L.lib(), C.c_int64(self.T * self.Fe)
        if torch.is_tensor(first):
            L.check(lib.mdm_noise_normal_ids(C.c_void_p(out.data_ptr()), per, C.c_int32(self.B), C.c_void_p(first.data_ptr()),
                                             C.c_uint64(seed), stream_dev, C.c_int32(stream_imm), s), "mdm_noise_normal_ids")
        else:
            L.check(lib.mdm_noise_normal(C.c_void_p(out.data_ptr()), per, C.c_int32(self.B), C.c_int64(first),
                                         C.c_uint64(seed), stream_dev, C.c_int32(stream_imm), s), "mdm_noise_normal")

    def draw_xT(self, seed: int, first=0):
        """x_T for rows [first, first + B) of a global batch (or the rows whose global indices `first` lists): the
        counter-based generator's MDM_NOISE_STREAM_XT stream."""
        out = torch.empty((self.B, self.T, self.Fe), dtype=torch.float32, device=self.dev)
        keep = self.philox
        self.philox = (int(seed) & 0xFFFFFFFFFFFFFFFF, self._ids(first))
        with torch.cuda.device(self.dev):
            self._philox_fill(out, C.c_void_p(0), L.NOISE_STREAM_XT, C.c_void_p(L.stream_ptr()))
        self.philox = keep
        return out

    def _ids(self, sample_offset):
        if torch.is_tensor(sample_offset) or isinstance(sample_offset, (list, tuple)):
            ids = torch.as_tensor(sample_offset).to(device=self.dev, dtype=torch.int64).contiguous()
            if ids.numel() != self.B:
                raise ValueError("sample_offset as a sequence must list one global sample index per row")
            return ids
        return int(sample_offset)

    def run(self, noise, step_noise, progress, callback, seed: Optional[int] = None, sample_offset: int = 0):
        """``seed``: draw x_T (when ``noise`` is None) and every step's noise (when ``step_noise`` is None) from the
        counter-based device generator keyed on (seed, sample_offset + row, timestep, element): the same global sample gets
        the same noise whatever the batch split.  Without a seed the torch generator is used, like the reference."""
        # everything below -- the warm-up step, the graph capture, its replays, the noise draws -- runs with the SAMPLER'S device
        # current: captured on another device's stream the graph would be empty and every replay a no-op (the reference's
        # tools use torch.device('cuda:N') without set_device, tools/visualization.py:57)
        with torch.cuda.device(self.dev):
            return self._run(noise, step_noise, progress, callback, seed, sample_offset)

    def _run(self, noise, step_noise, progress, callback, seed, sample_offset):
        d, B = self.d, self.B
        if seed is not None and step_noise is None:
            self.philox = (int(seed) & 0xFFFFFFFFFFFFFFFF, self._ids(sample_offset))
        self._prepare()
        if self.use_graph:
            g = torch.cuda.CUDAGraph()
            saved = {k: v.clone() for k, v in self.model.moe_buffers().items()}
            with torch.cuda.graph(g):
                self._step(self._needs_noise())
            for k, v in saved.items():  # capture does not execute, but keep the invariant explicit
                self.model.moe_buffers()[k].copy_(v)
            self.graph = g
        if noise is None:
            noise = self.draw_xT(seed, sample_offset) if seed is not None else torch.randn((B, self.T, self.Fe), device=self.dev)
        self.xx[:B].copy_(noise.to(self.dev, torch.float32))
        self.t_dev.fill_(d.num_timesteps - 1)
        it = range(d.num_timesteps)
        if progress:
            from tqdm.auto import tqdm
            it = tqdm(it, desc="Sampling")
        for i in it:
            if self._needs_noise() and self.philox is None:
                if step_noise is not None:
                    self.noise.copy_(step_noise[i].to(self.dev, torch.float32))
                else:
                    self.noise.normal_()
            if self.graph is not None:
                self.graph.replay()
            else:
                self._step(self._needs_noise())
            if callback is not None:
                callback(i, d.num_timesteps - 1 - i, self.xx[:B])
        return self.xx[:B].clone()

    def single(self, x, t, noise):
        t = torch.as_tensor(t)
        t0 = int(t.flatten()[0])
        if not 0 <= t0 < self.d.num_timesteps:
            raise ValueError(f"timestep {t0} outside the {self.d.num_timesteps}-step schedule")
        if t.numel() > 1 and not bool((t == t0).all()):
            raise NotImplementedError("per-sample timesteps within one sampler step are not supported; "
                                      "call the model directly for that")
        self.xx[:self.B].copy_(x.to(self.dev, torch.float32))
        self.t_dev.fill_(t0)
        use_noise = self._needs_noise()
        if use_noise:
            self.noise.copy_(noise.to(self.dev, torch.float32)) if noise is not None else self.noise.normal_()
        self._step(use_noise)
        return {"sample": self.xx[:self.B].clone(), "pred_xstart": self.x0.clone()}
