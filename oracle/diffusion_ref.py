"""ORACLE (test infrastructure only) -- CPU restatement of the reference sampling loops.

Same rules as oracle/denoiser_ref.py: imported only by tests/, smoke() and bench.py's
cpu_baseline leg.  Pinned by tests/golden/loop_*.npz generated from the reference itself.

Restates (paths under text2motion/models/gaussian_diffusion.py): linear betas :28-34,
the f64 tables :397-431, eps->x0 :554-558, posterior mean :462-475, the CFG step
:1042-1098 and loop :1100-1141, the DDIM step :699-742 and loop :776-818.
The model is any callable ``model(x, t, cond: bool) -> eps``.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import numpy as np
import torch


def linear_betas(num_steps: int) -> np.ndarray:
    scale = 1000 / num_steps
    return np.linspace(scale * 0.0001, scale * 0.02, num_steps, dtype=np.float64)


class Tables:
    def __init__(self, betas: np.ndarray):
        betas = np.asarray(betas, dtype=np.float64)
        a = 1.0 - betas
        self.betas = betas
        self.acp = np.cumprod(a, axis=0)
        self.acp_prev = np.append(1.0, self.acp[:-1])
        self.sqrt_recip_acp = np.sqrt(1.0 / self.acp)
        self.sqrt_recipm1_acp = np.sqrt(1.0 / self.acp - 1)
        self.post_var = betas * (1.0 - self.acp_prev) / (1.0 - self.acp)
        self.post_logvar_clipped = np.log(np.append(self.post_var[1], self.post_var[1:]))
        self.coef1 = betas * np.sqrt(self.acp_prev) / (1.0 - self.acp)
        self.coef2 = (1.0 - self.acp_prev) * np.sqrt(a) / (1.0 - self.acp)
        self.num_steps = len(betas)

    def f32(self, arr: np.ndarray, t: int) -> float:
        # _extract_into_tensor (:329-341): f64 table entry rounded to f32
        return float(np.float32(arr[t]))


def cfg_step(tb: Tables, t: int, x, eps_c, eps_u, noise, cfg_scale: float, clip: bool = False):
    """One p_sample_with_cfg update given the two model outputs (:1065-1096); eps_u None = unguided p_sample."""
    a = torch.tensor(tb.f32(tb.sqrt_recip_acp, t))
    b = torch.tensor(tb.f32(tb.sqrt_recipm1_acp, t))
    x0 = a * x - b * eps_c
    if clip:
        x0 = x0.clamp(-1, 1)
    if eps_u is not None:
        x0_u = a * x - b * eps_u
        if clip:
            x0_u = x0_u.clamp(-1, 1)
        x0 = x0_u + cfg_scale * (x0 - x0_u)
    mean = torch.tensor(tb.f32(tb.coef1, t)) * x0 + torch.tensor(tb.f32(tb.coef2, t)) * x
    nz = 0.0 if t == 0 else 1.0
    return mean + nz * torch.exp(0.5 * torch.tensor(tb.f32(tb.post_logvar_clipped, t))) * noise, x0


def ddim_step(tb: Tables, t: int, x, eps_in, noise, eta: float, clip: bool = True):
    """One ddim_sample update given the model output (:714-742)."""
    a = torch.tensor(tb.f32(tb.sqrt_recip_acp, t))
    b = torch.tensor(tb.f32(tb.sqrt_recipm1_acp, t))
    x0 = a * x - b * eps_in
    if clip:
        x0 = x0.clamp(-1, 1)
    eps = (a * x - x0) / b
    ab = torch.tensor(tb.f32(tb.acp, t))
    abp = torch.tensor(tb.f32(tb.acp_prev, t))
    sigma = eta * torch.sqrt((1 - abp) / (1 - ab)) * torch.sqrt(1 - ab / abp)
    mean = x0 * torch.sqrt(abp) + torch.sqrt(1 - abp - sigma ** 2) * eps
    nz = 0.0 if t == 0 else 1.0
    return mean + nz * sigma * noise, x0


def cfg_ddpm_loop(model: Callable, tb: Tables, x_T: torch.Tensor, step_noise: List[torch.Tensor],
                  cfg_scale: float = 7.5, keep: Optional[list] = None) -> torch.Tensor:
    """p_sample_loop_with_cfg with clip_denoised=False (as the trainer calls it, ddpm_trainer.py:161-173).
    step_noise[i] is the randn_like draw of the i-th executed step (t = steps-1-i)."""
    x = x_T
    B = x.shape[0]
    for i, t in enumerate(reversed(range(tb.num_steps))):
        tt = torch.full((B,), t, dtype=torch.int64)
        x, _ = cfg_step(tb, t, x, model(x, tt, True), model(x, tt, False), step_noise[i], cfg_scale, clip=False)
        if keep is not None:
            keep.append(x.clone())
    return x


def ddim_loop(model: Callable, tb: Tables, x_T: torch.Tensor, step_noise: List[torch.Tensor], eta: float = 0.0,
              clip_denoised: bool = True, cond: bool = True, keep: Optional[list] = None) -> torch.Tensor:
    """ddim_sample_loop; its default clip_denoised=True clamps pred_xstart to [-1,1] (:523-528)."""
    x = x_T
    B = x.shape[0]
    for i, t in enumerate(reversed(range(tb.num_steps))):
        tt = torch.full((B,), t, dtype=torch.int64)
        x, _ = ddim_step(tb, t, x, model(x, tt, cond), step_noise[i], eta, clip_denoised)
        if keep is not None:
            keep.append(x.clone())
    return x
