"""Device-side post-processing of generated motions.

Host mirror of what the reference does on the CPU after sampling (tools/visualization.py:21-27,89):
``motion * std + mean`` -> ``recover_from_ric`` (utils/motion_process.py:403-416) -> ``motion_temporal_filter``
(utils/utils.py:125-130).  All arithmetic is in ``mdm_motion_postprocess`` (csrc/motion_post.hip); no eager fallback."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib as L


def gaussian_taps(sigma: float, truncate: float = 4.0) -> np.ndarray:
    """w[0..radius] of scipy.ndimage.gaussian_filter1d's normalised kernel (w[k] == w[-k]), fp64."""
    if sigma is None or sigma <= 0:
        return np.zeros(1, dtype=np.float64)
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    phi = phi / phi.sum()
    return np.ascontiguousarray(phi[radius:], dtype=np.float64)


@torch.no_grad()
def motion_to_joints(motion: torch.Tensor, mean, std, lengths: Optional[torch.Tensor] = None, joints_num: int = 22,
                     sigma: float = 1.0) -> torch.Tensor:
    """motion (B, T, 263) normalised samples on a GPU -> joints (B, T, joints_num, 3); frames >= lengths[b] are zero."""
    L.require_cuda(motion)
    dev = motion.device
    x = motion.detach().to(torch.float32).contiguous()
    if x.dim() == 2:
        x = x[None]
    B, T, Fe = x.shape
    mean_t = torch.as_tensor(np.asarray(mean), dtype=torch.float32).to(dev).contiguous()
    std_t = torch.as_tensor(np.asarray(std), dtype=torch.float32).to(dev).contiguous()
    if mean_t.numel() != Fe or std_t.numel() != Fe:
        raise ValueError(f"mean/std must have {Fe} entries")
    w = gaussian_taps(sigma)
    radius = len(w) - 1
    w_t = torch.from_numpy(w).to(dev)
    ln = None if lengths is None else torch.as_tensor(lengths).to(dev, torch.int32).contiguous()
    scratch = torch.empty(B, T, joints_num, 3, device=dev)
    out = torch.empty_like(scratch)
    with torch.cuda.device(dev):
        L.check(L.lib().mdm_motion_postprocess(
            C.c_void_p(x.data_ptr()), C.c_void_p(L.ptr(ln)), C.c_void_p(mean_t.data_ptr()), C.c_void_p(std_t.data_ptr()),
            C.c_int32(B), C.c_int32(T), C.c_int32(Fe), C.c_int32(joints_num), C.c_int32(radius), C.c_void_p(w_t.data_ptr()),
            C.c_void_p(scratch.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(L.stream_ptr())), "mdm_motion_postprocess")
    return out


def recover_from_ric(data: torch.Tensor, joints_num: int = 22) -> torch.Tensor:
    """Same name / meaning as utils/motion_process.py:403: de-normalised 263-d rows (..., T, 263) -> (..., T, J, 3)."""
    lead = data.shape[:-2]
    x = data.reshape((-1,) + tuple(data.shape[-2:]))
    Fe = x.shape[-1]
    j = motion_to_joints(x, np.zeros(Fe, np.float32), np.ones(Fe, np.float32), None, joints_num, sigma=0.0)
    return j.reshape(lead + tuple(j.shape[1:]))
