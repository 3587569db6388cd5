"""Deterministic synthetic weights / captured-random state / inputs.

The reference ships no checkpoints and its forward pass depends on 33 per-call
random ``nn.Linear`` layers and 2*2*L lazily drawn Performer feature matrices
(reference: text2motion/models/stylization.py:22-24, transformer.py:313-315,
fast_attention.py:19-36).  Parity is only defined once those are *captured and
injected*; this module is the single place that defines how we draw them so the
golden-fixture generator (oracle/make_golden.py), the tests, smoke() and
bench.py all agree bit-for-bit on every host.

Everything here is built from ``torch.randint`` on a CPU generator followed by
exact float arithmetic, so the values do not depend on the host's vector ISA
(``torch.randn`` goes through vectorised log/sin/cos which may differ by an ulp
between CPUs).
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, List, Tuple

import torch

_MASK63 = (1 << 63) - 1


def _seed_for(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) * 2654435761 + seed * 1000003 + 12345) & _MASK63


def uniform_pm1(shape, name: str, seed: int) -> torch.Tensor:
    """Exact, platform independent U(-1, 1) fp32 tensor keyed by (name, seed)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(_seed_for(name, seed))
    r = torch.randint(0, 1 << 23, tuple(shape), generator=g, dtype=torch.int64)
    # (r + 0.5) * 2^-22 - 1 is exactly representable in fp32 for r < 2^23
    return (r.to(torch.float32) + 0.5) * (1.0 / (1 << 22)) - 1.0


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int) -> torch.Tensor:
    """One synthetic parameter/buffer for state_dict key ``name``.

    Scales are chosen so activations stay O(1) through the stack and the MoE
    router is non-degenerate (the reference zero-inits gates and output heads,
    transformer.py:257, switch_moe.py:28-29, which would make every golden
    vector trivially zero)."""
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf in ("expert_usage", "expert_importance"):
        return torch.zeros(shape, dtype=torch.float32)
    u = uniform_pm1(shape, name, seed)
    if name == "sequence_embedding":
        return u
    if leaf == "gate" or leaf == "adaptive_gate":  # cross_attn.gate (D,), adaptive_gate (1,)
        return u
    if len(shape) == 1:
        if leaf == "weight":  # LayerNorm gains
            return 1.0 + 0.2 * u
        return 0.1 * u  # biases (Linear, LayerNorm, Conv)
    if len(shape) == 2:
        fan_in = shape[1]
        return u * float((3.0 / fan_in) ** 0.5)
    if len(shape) == 3:  # Conv1d (out,in,k) / ConvTranspose1d (in,out,k); D x D x 2 either way
        fan_in = shape[1] * shape[2] if name.startswith("downsample") else shape[0] * shape[2]
        return u * float((3.0 / fan_in) ** 0.5)
    raise ValueError(f"unexpected shape for {name}: {shape}")


def synth_state_dict(keys_shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int) -> Dict[str, torch.Tensor]:
    return {k: synth_tensor(k, tuple(s), seed) for k, s in keys_shapes}


def synth_linear(name: str, in_f: int, out_f: int, seed: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """An 'ephemeral' Linear drawn like nn.Linear's default init:
    weight, bias ~ U(-1/sqrt(in), 1/sqrt(in))."""
    b = float(in_f) ** -0.5
    w = uniform_pm1((out_f, in_f), name + ".weight", seed) * b
    bias = uniform_pm1((out_f,), name + ".bias", seed) * b
    return w, bias


def ephemeral_names(num_layers: int, with_text_proj: bool) -> List[str]:
    """Call order of the per-forward random Linears in the reference:
    text_proj (transformer.py:313-315) then, for the low blocks followed by the
    high blocks, per layer: local style, global style (fast_attention.py:175),
    cross-attn proj_out (fast_attention.py:257), ffn proj_out (multi_branch.py:60)."""
    names = ["text_proj"] if with_text_proj else []
    for scale in ("low", "high"):
        for i in range(num_layers):
            for sub in ("local_style", "global_style", "cross_style", "ffn_style"):
                names.append(f"{scale}.{i}.{sub}")
    return names


def synth_ephemerals(latent_dim: int, text_latent_dim: int, num_layers: int, seed: int):
    """Ordered list of (name, weight, bias) for one forward pass."""
    te = 4 * latent_dim
    out = []
    for n in ephemeral_names(num_layers, text_latent_dim != latent_dim):
        if n == "text_proj":
            w, b = synth_linear("ephemeral." + n, text_latent_dim, latent_dim, seed)
        else:
            w, b = synth_linear("ephemeral." + n, latent_dim, te, seed)
        out.append((n, w, b))
    return out


def synth_projection(name: str, head_dim: int, seed: int) -> torch.Tensor:
    """A Performer feature matrix with the reference's post-processing
    (fast_attention.py:26: column-normalise, scale by head_dim**-0.25) applied to
    an exact-uniform draw instead of QR(randn): shape (head_dim, min(head_dim,256)).
    The reference's matrix is unsaved per-process random state, so any injected
    matrix of this shape is a legitimate 'captured' value."""
    m = min(head_dim, 256)
    p = uniform_pm1((head_dim, m), "projection." + name, seed)
    p = p / p.norm(dim=0, keepdim=True).clamp_min(1e-12)
    return p * float(head_dim) ** -0.25


def projection_names(num_layers: int) -> List[str]:
    names = []
    for scale in ("low", "high"):
        for i in range(num_layers):
            names += [f"{scale}.{i}.local", f"{scale}.{i}.global"]
    return names


def synth_projections(head_dim: int, num_layers: int, seed: int):
    return [(n, synth_projection(n, head_dim, seed)) for n in projection_names(num_layers)]


def synth_inputs(B: int, T: int, feats: int, n_text: int, text_latent_dim: int, seed: int,
                 num_steps: int = 1000, min_len: int = 0):
    """HumanML3D-shaped synthetic inputs: x ~ U scaled to unit
    variance, lengths multiple of 4 with one row == T, text tokens, pooled = mean."""
    x = uniform_pm1((B, T, feats), "in.x", seed) * (3.0 ** 0.5)
    g = torch.Generator(device="cpu")
    g.manual_seed(_seed_for("in.len", seed))
    lo = max(4, min(min_len if min_len else 40, T))
    length = torch.randint(lo // 4, T // 4 + 1, (B,), generator=g, dtype=torch.int64) * 4
    length = length.clamp(max=T)
    length[0] = T
    t = torch.randint(0, num_steps, (B,), generator=g, dtype=torch.int64)
    xf_out = uniform_pm1((B, n_text, text_latent_dim), "in.xf_out", seed) * (3.0 ** 0.5)
    xf_proj = xf_out.mean(dim=1)
    return x, t, length, xf_proj, xf_out
