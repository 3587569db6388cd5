"""Batch-parallel sampling over the GPUs of one node.

Samples never interact inside the denoiser or the sampler, so the global batch is cut into contiguous per-rank
shards (each sample's cond+uncond pair stays on its rank), weights are replicated, there is no per-step
collective, and the only exchange is ONE all_gather of the final (B/world, T, feats) motion tensor
(RCCL over xGMI on GPUs; gloo in the CPU tests).  x_T and every step's noise come from a counter-based generator keyed
on (seed, GLOBAL sample index, timestep, element) -- csrc/noise.hip on the device, oracle/philox_ref.py restates it -- so
results do not depend on the world size and no rank ever materialises the global batch's noise.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.distributed as dist


def shard_range(B: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split: the first B % world ranks get one extra sample."""
    q, r = divmod(B, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_kwargs(kw: Dict, lo: int, hi: int) -> Dict:
    out = {}
    for k, v in kw.items():
        if isinstance(v, torch.Tensor) and v.dim() >= 1:
            out[k] = v[lo:hi]
        elif isinstance(v, (list, tuple)):
            out[k] = v[lo:hi]
        else:
            out[k] = v
    return out


def global_noise(shape, seed: int, steps: int = 0):
    """x_T (and optionally per-step noise) for the GLOBAL batch from a CPU generator.  Host-materialised: O(steps * batch)
    memory, kept for small offline comparisons only; the samplers use the counter-based device generator
    (``p_sample_loop_with_cfg(..., seed=, sample_offset=)``) instead."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    x_T = torch.randn(shape, generator=g)
    step = [torch.randn(shape, generator=g) for _ in range(steps)]
    return x_T, step


def all_gather_ragged(local: torch.Tensor, B: int, group=None) -> torch.Tensor:
    """all_gather of shards whose first dim may differ by one: pad to the max shard, gather once, trim."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    rank = dist.get_rank(group)
    sizes = [shard_range(B, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    parts = [out[r * mx:r * mx + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, 0)


def sample_sharded(sample_fn, shape, model_kwargs: Dict, seed: int, group=None):
    """Run ``sample_fn(local_shape, local_kwargs, seed, first_global_row) -> (b, T, F)`` on this rank's shard of the global
    batch and all_gather the result.  ``sample_fn`` draws its noise from the counter-based generator with
    ``sample_offset=first_global_row`` (e.g. ``diffusion.p_sample_loop_with_cfg(..., seed=seed, sample_offset=first)``), so
    the gathered tensor is the same for every world size.  Any backend (nccl == RCCL on ROCm, gloo on CPU)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = shape[0]
    lo, hi = shard_range(B, rank, world)
    local_kw = shard_kwargs(model_kwargs, lo, hi)
    if hi > lo:
        local = sample_fn((hi - lo,) + tuple(shape[1:]), local_kw, seed, lo)
    else:
        # more ranks than samples: this rank has nothing to denoise but still takes part in the gather (on the device the
        # others produce their shards on: that of the conditioning tensors)
        dev = next((v.device for v in model_kwargs.values() if isinstance(v, torch.Tensor)), torch.device("cpu"))
        local = torch.empty((0,) + tuple(shape[1:]), dtype=torch.float32, device=dev)
    return all_gather_ragged(local, B, group)


# ---- evaluation-scale generation: length-bucketed batches dealt over the ranks --------------
def plan_buckets(m_lens, batch_size: int, max_frames: int, unit: int = 1):
    """Replace the reference's serial ``while cur_idx < N`` batching (ddpm_trainer.py:176-199), which pads every batch to
    its longest sample, by batches of similar length: samples sorted by (clamped) length, longest first, cut every
    ``batch_size``.  Returns [(indices (LongTensor), T)], T = that batch's frame count (rounded up to ``unit``)."""
    lens = torch.as_tensor(m_lens).flatten().long().clamp(max=max_frames)
    order = torch.sort(lens, descending=True, stable=True).indices
    plan = []
    for lo in range(0, len(order), batch_size):
        idx = order[lo:lo + batch_size]
        T = int(lens[idx].max())
        T = min(max_frames, (T + unit - 1) // unit * unit)
        plan.append((idx, T))
    return plan


def padded_frames(plan) -> int:
    """sum over batches of B_i * T_i: the work the plan schedules (the serial plan's figure is its baseline)."""
    return sum(len(idx) * T for idx, T in plan)


def run_plan(plan, run_bucket, n: int, max_frames: int, feats: int, device, group=None):
    """Deal the plan's batches round-robin over the ranks (batches are sorted by cost, so every rank gets a similar mix),
    run ``run_bucket(bucket_id, indices, T) -> (len(indices), T, feats)`` on the local ones, and exchange the results
    with ONE all_gather of a padded (per-rank samples, max_frames, feats) buffer.  Returns the list of per-sample
    ``(T_bucket, feats)`` tensors in the ORIGINAL order, on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = [k for k in range(len(plan)) if k % world == rank]
    per_rank = [sum(len(plan[k][0]) for k in range(len(plan)) if k % world == r) for r in range(world)]
    cap = max(per_rank) if per_rank else 0
    buf = torch.zeros((cap, max_frames, feats), dtype=torch.float32, device=device)
    pos = 0
    for k in mine:
        idx, T = plan[k]
        y = run_bucket(k, idx, T)
        assert tuple(y.shape) == (len(idx), T, feats), (tuple(y.shape), len(idx), T, feats)
        buf[pos:pos + len(idx), :T] = y
        pos += len(idx)
    if world > 1:
        allb = torch.empty((world * cap, max_frames, feats), dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(allb, buf, group=group)
    else:
        allb = buf
    out = [None] * n
    cursor = [0] * world
    for k, (idx, T) in enumerate(plan):
        r = k % world
        base = r * cap + cursor[r]
        for j, i in enumerate(idx.tolist()):
            out[i] = allb[base + j, :T].clone()
        cursor[r] += len(idx)
    return out
