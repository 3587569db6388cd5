"""state_dict layout of the reference denoiser.

Pinned against the reference by tests/golden/state_dict_layout.json (dumped from the
reference's own ``MotionTransformer(...).state_dict()`` by oracle/make_golden.py).
Reference: text2motion/models/transformer.py:174-257 and the sub-module constructors
(fast_attention.py:95-135,186-206,228-240,261-267,279-299; multi_branch.py:32-50;
switch_moe.py:13-34; stylization.py:6-18; time.py:5-13; gate.py:5-14).
"""
from __future__ import annotations

from typing import List, Tuple

Key = Tuple[str, Tuple[int, ...]]

BUFFER_LEAVES = ("expert_usage", "expert_importance")


def resolve_dims(latent_dim: int, ff_size: int, text_latent_dim: int, model_size: str):
    """model_size == 'big' doubles the three widths (transformer.py:188-192)."""
    if model_size == "big":
        return latent_dim * 2, ff_size * 2, text_latent_dim * 2
    return latent_dim, ff_size, text_latent_dim


def _linear(p: str, out_f: int, in_f: int) -> List[Key]:
    return [(p + ".weight", (out_f, in_f)), (p + ".bias", (out_f,))]


def _norm(p: str, d: int) -> List[Key]:
    return [(p + ".weight", (d,)), (p + ".bias", (d,))]


def _style(p: str, D: int, Te: int) -> List[Key]:
    return _linear(p + ".emb_layers.1", 2 * D, Te) + _norm(p + ".norm", D) + _linear(p + ".out_layers.2", D, D)


def _performer(p: str, D: int, dh: int, Te: int) -> List[Key]:
    k = _norm(p + ".pre_norm", D) + _norm(p + ".post_norm", D)
    k += _linear(p + ".query", D, D) + _linear(p + ".key", D, D) + _linear(p + ".value", D, D)
    k += _norm(p + ".fast_attention.norm", dh)
    k += _linear(p + ".proj_out.0", D, D) + _linear(p + ".proj_out.3", D, D)
    k += _style(p + ".style_block", D, Te)
    return k


def layer_layout(p: str, D: int, F: int, Dt: int, H: int, E: int) -> List[Key]:
    Te, dh = 4 * D, D // H
    k: List[Key] = []
    d = p + ".dual_self_attn"
    k += _norm(d + ".pre_norm", D) + _norm(d + ".post_norm", D)
    k += _performer(d + ".local_attn", D, dh, Te) + _performer(d + ".global_attn", D, dh, Te)
    k += _linear(d + ".skip_proj.0", D, D)
    c = p + ".cross_attn"
    k += [(c + ".gate", (D,)), (c + ".base_ca.adaptive_gate", (1,))]
    k += _norm(c + ".base_ca.norm", D) + _norm(c + ".base_ca.text_norm", Dt)
    k += _linear(c + ".base_ca.query", D, D) + _linear(c + ".base_ca.key", D, Dt) + _linear(c + ".base_ca.value", D, Dt)
    k += _style(c + ".base_ca.proj_out", D, Te)
    f = p + ".ffn"
    for b in range(2):
        br = f"{f}.branches.{b}"
        k += _norm(br + ".layernorm", D)
        k += [(br + ".moe.expert_usage", (E,)), (br + ".moe.expert_importance", (E,))]
        k += _linear(br + ".moe.gate", E, D)
        for e in range(E):
            k += _linear(f"{br}.moe.experts.{e}.0", F, D) + _linear(f"{br}.moe.experts.{e}.2", D, F)
    k += _style(f + ".proj_out", D, Te)
    s = p + ".sd_cross_attn"
    k += _linear(s + ".query", D, D) + _linear(s + ".key", D, Dt) + _linear(s + ".value", D, Dt) + _linear(s + ".out", D, D)
    k += _norm(s + ".ffn.0", D) + _linear(s + ".ffn.1", 4 * D, D) + _linear(s + ".ffn.3", D, 4 * D)
    return k


def state_dict_layout(input_feats: int, num_frames: int = 60, latent_dim: int = 512, ff_size: int = 1024,
                      num_layers: int = 4, num_heads: int = 4, text_latent_dim: int = 256,
                      moe_num_experts: int = 4, model_size: str = "small", **_ignored) -> List[Key]:
    D, F, Dt = resolve_dims(latent_dim, ff_size, text_latent_dim, model_size)
    Te = 4 * D
    k: List[Key] = [("sequence_embedding", (num_frames, D))]
    k += _linear("learnable_time_embed.mlp.0", 2 * D, D) + _linear("learnable_time_embed.mlp.2", D, 2 * D)
    for n in ("proj_time", "proj_text", "post_mlp.0", "post_mlp.2"):
        k += _linear("gated_fusion." + n, D, D)
    k += _linear("time_embed.0", Te, D) + _linear("time_embed.2", Te, Te) + _linear("time_proj", D, Te)
    k += _linear("joint_embed", D, input_feats)
    k += [("downsample.weight", (D, D, 2)), ("downsample.bias", (D,))]
    k += [("upsample.weight", (D, D, 2)), ("upsample.bias", (D,))]
    for scale in ("low", "high"):
        for i in range(num_layers):
            k += layer_layout(f"decoder_blocks_{scale}.{i}.module", D, F, Dt, num_heads, moe_num_experts)
    k += _linear("out", input_feats, D)
    return k
