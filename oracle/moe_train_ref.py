"""ORACLE (test infrastructure only -- never imported by the product path): reference gradients of the MoE feed-forward
block by torch autograd through the CPU restatement of the forward (oracle/denoiser_ref.py::moe_ffn, which follows
text2motion/models/multi_branch.py:52-61, switch_moe.py:44-111 and stylization.py:20-31), the load-balancing loss of
switch_moe.py:113-145, and the optimizer arithmetic of ddpm_trainer.py:228-244 (clip_grad_norm_ 1.0 + Adam).

Pinning: the forward restated here is the one pinned against the reference-generated goldens (tests/test_oracle_golden.py);
the gradients are autograd's of that forward, in fp32 (fp64 on request) -- the same thing loss.backward() computes in the
reference's training loop.
"""
from typing import Dict, Optional

import torch

import numpy as np
import torch.nn.functional as F

from . import denoiser_ref as R
from . import philox_ref as PH


def lb_loss(usage: torch.Tensor, importance: torch.Tensor, E: int, eps: float = 1e-8) -> torch.Tensor:
    """switch_moe.py:113-145."""
    fu = usage / usage.sum().clamp_min(eps)
    fi = importance / importance.sum().clamp_min(eps)
    return E * (1.0 - (fu * fi).sum())


def dropout_masks(seed: int, M: int, D: int, p: float):
    """The training step's dropout masks, one (M, D) tensor per site (0 / 1: the two branch outputs, multi_branch.py:57;
    2: after the SiLU of the stylization block, stylization.py:16): 1 / (1 - p') where kept, 0 where dropped, with
    p' = floor(p 2^24) / 2^24 and element (row, col) kept when the top 24 bits of word col % 4 of
    Philox4x32-10(counter (col // 4, row, 0, site), key seed) are >= floor(p 2^24) -- the rule of csrc/moe_train.hip."""
    thr = int(np.float32(p) * np.float32(16777216.0))
    keep = np.float32(16777216.0) / (np.float32(16777216.0) - np.float32(thr))
    out = []
    for site in range(3):
        bits = PH.uniform_bits(D, M, 0, seed, site).reshape(M, -1)[:, :D]
        out.append(torch.from_numpy(np.where((bits >> np.uint32(8)) >= thr, keep, np.float32(0.0)).astype(np.float32)))
    return out


def moe_ffn_train(x, emb, sd, prefix: str, E: int, eph_wb, masks=None, forced=None, trace=None):
    """MoEMultiBranchFFN.forward in TRAINING mode (multi_branch.py:52-61 with b["drop"], stylization.py:20-31 with its
    Dropout) for explicit dropout masks; masks=None is denoiser_ref.moe_ffn."""
    if masks is None:
        return R.moe_ffn(x, emb, sd, prefix, E, eph_wb, forced=forced, trace=trace)
    B, S, D = x.shape
    acc = 0
    for br in range(2):
        h = R._ln(x, sd, f"{prefix}.branches.{br}.layernorm").reshape(-1, D)
        o, idx, vals, usage, imp = R.switch_moe(h, sd, f"{prefix}.branches.{br}.moe", E, None if forced is None else forced[br])
        if trace is not None:
            trace[f"{prefix}.branches.{br}.top2_idx"], trace[f"{prefix}.branches.{br}.top2_val"] = idx, vals
            trace[f"{prefix}.branches.{br}.usage"], trace[f"{prefix}.branches.{br}.importance"] = usage, imp
        acc = acc + (o * masks[br].to(o.dtype)).view(B, S, D)
    acc = acc / 2
    sp = prefix + ".proj_out"
    te_dim = sd[sp + ".emb_layers.1.weight"].shape[1]
    e = emb if emb.shape[-1] == te_dim else F.linear(emb, eph_wb[0], eph_wb[1])
    eo = R._lin(F.silu(e), sd, sp + ".emb_layers.1").unsqueeze(1)
    h = R._ln(acc, sd, sp + ".norm") * (1 + eo[..., :D]) + eo[..., D:]
    return x + R._lin(F.silu(h) * masks[2].to(h.dtype).view(B, S, D), sd, sp + ".out_layers.2")


def moe_ffn_grads(sd: Dict[str, torch.Tensor], prefix: str, E: int, x: torch.Tensor, emb: torch.Tensor, eph_wb, dout: torch.Tensor,
                  forced=None, dtype=torch.float32, masks=None):
    """Returns (out, dx, demb, {state_dict key: gradient}, lb (2,), trace)."""
    p = {k: v.detach().to(dtype).clone().requires_grad_(v.dtype.is_floating_point and "expert_" not in k)
         for k, v in sd.items() if k.startswith(prefix + ".")}
    x = x.detach().to(dtype).clone().requires_grad_(True)
    emb = emb.detach().to(dtype).clone().requires_grad_(True)
    eph = None if eph_wb is None else (eph_wb[0].to(dtype), eph_wb[1].to(dtype))
    trace = {}
    out = moe_ffn_train(x, emb, p, prefix, E, eph, masks=masks, forced=forced, trace=trace)
    out.backward(dout.to(dtype))
    grads = {k: v.grad for k, v in p.items() if v.requires_grad and v.grad is not None}
    lb = torch.stack([lb_loss(trace[f"{prefix}.branches.{b}.usage"], trace[f"{prefix}.branches.{b}.importance"], E) for b in range(2)])
    return out.detach(), x.grad, emb.grad, grads, lb.detach(), trace


def adam_clip_step(params: torch.Tensor, grads: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr=2e-4, betas=(0.9, 0.999),
                   eps=1e-8, max_norm: Optional[float] = 1.0):
    """clip_grad_norm_ + torch.optim.Adam arithmetic on flat fp32 tensors (returns new params, m, v, the gradient norm)."""
    g = grads.clone()
    norm = g.norm()
    if max_norm is not None and max_norm > 0:
        g = g * torch.clamp(max_norm / (norm + 1e-6), max=1.0)
    m = betas[0] * m + (1 - betas[0]) * g
    v = betas[1] * v + (1 - betas[1]) * g * g
    bc1, bc2 = 1 - betas[0] ** step, 1 - betas[1] ** step
    params = params - lr * (m / bc1) / ((v / bc2).sqrt() + eps)
    return params, m, v, norm
