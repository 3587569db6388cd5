"""Weight packing: reference ``state_dict`` layout -> kernel layouts -> ``MdmModel`` (include/mdm_hip.h).

Two stages so the layout logic is checkable without a GPU:
  * ``kernel_layout(sd, cfg, eph, proj)``: pure tensor reshapes/concats (fp32, any device);
  * ``PackedModel(...)``: uploads, splits every matrix into bf16 hi/lo planes with the HIP pack kernel and
    fills the ctypes structs.  Runs once per (weights, captured randomness), never on the hot path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import torch

from . import _lib as L

STYLE_SLOTS = ("local_style", "global_style", "cross_style", "ffn_style")

# Weights that are always packed as bf16 hi + lo planes: they are multiplied with fp32 activations by the register-staged
# kernel (csrc/gemm.hip) in every mode -- the per-loop stem / text caches are always built in the bf16x3 arithmetic, and
# joint_embed reads the fp32 motion tensor itself (the root of the residual stream: fp32-grade in every mode).
_ALWAYS_X3 = ("tmlp0", "tmlp2", "te0", "te2", "tproj", "gf_time", "gf_text", "text_proj", "joint")
_ALWAYS_X3_LAYER = ("ca_k", "ca_v", "sd_k", "sd_v")
_MLP_LAYER = ("w1", "w2", "sd_f1", "sd_f2")  # the MFMA-bound GEMMs: expert MLPs and the 4x FFN


def format_class(precision: int) -> str:
    """Packed models are shared between precisions that read the same planes."""
    return {L.PREC_BF16: "bf16", L.PREC_X3: "bf16", L.PREC_F16: "f16", L.PREC_MIXED: "mixed", L.PREC_FP8: "f8"}[precision]


def weight_format(name: str, precision: int, head_dim: int) -> str:
    """Plane format of kernel_layout()'s matrix `name` for a run at `precision` (include/mdm_hip.h: MDM_PREC_*)."""
    leaf = name.split(".", 1)[1] if name.startswith("L") and "." in name else name
    if name in _ALWAYS_X3 or leaf in _ALWAYS_X3_LAYER:
        return "bf16x2"
    if leaf.endswith("feat") and head_dim not in (128, 256):  # no fused Performer core: the feature GEMM reads fp32 rows
        return "bf16x2"
    if precision == L.PREC_FP8:
        return "f8" if leaf in ("w1", "w2") else "f16"
    if precision == L.PREC_F16:
        return "f16"
    if precision == L.PREC_MIXED and leaf in _MLP_LAYER:
        return "f16"
    return "bf16x2"


def layer_tags(num_layers: int) -> List[Tuple[str, str]]:
    """(state_dict prefix, captured-randomness tag) for the 2L decoder layers: low blocks then high blocks."""
    out = []
    for scale in ("low", "high"):
        for i in range(num_layers):
            out.append((f"decoder_blocks_{scale}.{i}.module", f"{scale}.{i}"))
    return out


def kernel_layout(sd: Dict[str, torch.Tensor], cfg: dict, eph: Dict[str, Tuple[torch.Tensor, torch.Tensor]],
                  proj: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Matrices ("W:" keys, [N,K] fp32) and vectors ("V:" keys) in the layouts the kernels read."""
    D, L_, E = cfg["latent_dim"], cfg["num_layers"], cfg["moe_num_experts"]
    out: Dict[str, torch.Tensor] = {}

    def lin(dst, src):
        out["W:" + dst] = sd[src + ".weight"]
        out["V:" + dst + "_b"] = sd[src + ".bias"]

    lin("tmlp0", "learnable_time_embed.mlp.0"), lin("tmlp2", "learnable_time_embed.mlp.2")
    lin("te0", "time_embed.0"), lin("te2", "time_embed.2"), lin("tproj", "time_proj")
    lin("gf_time", "gated_fusion.proj_time"), lin("gf_text", "gated_fusion.proj_text")
    lin("gf_post0", "gated_fusion.post_mlp.0"), lin("gf_post2", "gated_fusion.post_mlp.2")
    lin("joint", "joint_embed"), lin("out", "out")
    if "text_proj" in eph:
        out["W:text_proj"], out["V:text_proj_b"] = eph["text_proj"]
    # Conv1d(k=2,s=2) weight (out,in,k) -> [out, k*D + in]: a Linear over the concatenated frame pair (transformer.py:223,334)
    out["W:down"] = sd["downsample.weight"].permute(0, 2, 1).reshape(D, 2 * D)
    out["V:down_b"] = sd["downsample.bias"]
    # ConvTranspose1d(k=2,s=2) weight (in,out,k) -> [k*D + out, in]: each coarse frame emits its two fine frames (:224,348)
    out["W:up"] = sd["upsample.weight"].permute(2, 1, 0).reshape(2 * D, D)
    out["V:up_b2"] = torch.cat([sd["upsample.bias"], sd["upsample.bias"]])
    out["V:seq_emb"] = sd["sequence_embedding"]
    eph_w, eph_b, emb_w, emb_b = [], [], [], []
    for li, (pre, tag) in enumerate(layer_tags(L_)):
        k = f"L{li}."
        d = pre + ".dual_self_attn"
        for nm in ("pre_norm", "post_norm"):
            out[f"V:{k}dual_{nm}_w"], out[f"V:{k}dual_{nm}_b"] = sd[f"{d}.{nm}.weight"], sd[f"{d}.{nm}.bias"]
        for which in ("local", "global"):
            a = f"{d}.{which}_attn"
            q = k + which + "."
            for nm in ("pre_norm", "post_norm"):
                out[f"V:{q}{nm}_w"], out[f"V:{q}{nm}_b"] = sd[f"{a}.{nm}.weight"], sd[f"{a}.{nm}.bias"]
            out["W:" + q + "qkv"] = torch.cat([sd[a + ".query.weight"], sd[a + ".key.weight"], sd[a + ".value.weight"]], 0)
            out["V:" + q + "qkv_b"] = torch.cat([sd[a + ".query.bias"], sd[a + ".key.bias"], sd[a + ".value.bias"]])
            out["V:" + q + "hn_w"], out["V:" + q + "hn_b"] = sd[a + ".fast_attention.norm.weight"], sd[a + ".fast_attention.norm.bias"]
            out["W:" + q + "feat"] = proj[f"{tag}.{which}"].t()  # [m, dh]
            out["W:" + q + "proj0"], out["V:" + q + "proj0_b"] = sd[a + ".proj_out.0.weight"], sd[a + ".proj_out.0.bias"]
            out["W:" + q + "proj3"], out["V:" + q + "proj3_b"] = sd[a + ".proj_out.3.weight"], sd[a + ".proj_out.3.bias"]
            _style(out, sd, q + "style.", a + ".style_block")
            emb_w.append(sd[a + ".style_block.emb_layers.1.weight"]), emb_b.append(sd[a + ".style_block.emb_layers.1.bias"])
        out["W:" + k + "skip"], out["V:" + k + "skip_b"] = sd[d + ".skip_proj.0.weight"], sd[d + ".skip_proj.0.bias"]
        c = pre + ".cross_attn.base_ca"
        out[f"V:{k}ca_norm_w"], out[f"V:{k}ca_norm_b"] = sd[c + ".norm.weight"], sd[c + ".norm.bias"]
        out[f"V:{k}ca_tnorm_w"], out[f"V:{k}ca_tnorm_b"] = sd[c + ".text_norm.weight"], sd[c + ".text_norm.bias"]
        for nm, src in (("ca_q", "query"), ("ca_k", "key"), ("ca_v", "value")):
            out["W:" + k + nm], out["V:" + k + nm + "_b"] = sd[f"{c}.{src}.weight"], sd[f"{c}.{src}.bias"]
        out["V:" + k + "ca_gate"] = sd[pre + ".cross_attn.gate"]
        out["V:" + k + "ca_adaptive"] = sd[c + ".adaptive_gate"]
        _style(out, sd, k + "ca_style.", c + ".proj_out")
        emb_w.append(sd[c + ".proj_out.emb_layers.1.weight"]), emb_b.append(sd[c + ".proj_out.emb_layers.1.bias"])
        f = pre + ".ffn"
        w1, b1, w2, b2 = [], [], [], []
        for b in range(2):
            br = f"{f}.branches.{b}"
            out[f"V:{k}moe_ln_w{b}"], out[f"V:{k}moe_ln_b{b}"] = sd[br + ".layernorm.weight"], sd[br + ".layernorm.bias"]
            out[f"V:{k}gate_w{b}"], out[f"V:{k}gate_b{b}"] = sd[br + ".moe.gate.weight"], sd[br + ".moe.gate.bias"]
            for e in range(E):
                w1.append(sd[f"{br}.moe.experts.{e}.0.weight"]), b1.append(sd[f"{br}.moe.experts.{e}.0.bias"])
                w2.append(sd[f"{br}.moe.experts.{e}.2.weight"]), b2.append(sd[f"{br}.moe.experts.{e}.2.bias"])
        out["W:" + k + "w1"], out["V:" + k + "b1"] = torch.cat(w1, 0), torch.cat(b1)  # [2*E*F, D]
        out["W:" + k + "w2"], out["V:" + k + "b2"] = torch.cat(w2, 0), torch.cat(b2)  # [2*E*D, F]
        _style(out, sd, k + "ffn_style.", f + ".proj_out")
        emb_w.append(sd[f + ".proj_out.emb_layers.1.weight"]), emb_b.append(sd[f + ".proj_out.emb_layers.1.bias"])
        s = pre + ".sd_cross_attn"
        for nm, src in (("sd_q", "query"), ("sd_k", "key"), ("sd_v", "value"), ("sd_out", "out"), ("sd_f1", "ffn.1"),
                        ("sd_f2", "ffn.3")):
            out["W:" + k + nm], out["V:" + k + nm + "_b"] = sd[f"{s}.{src}.weight"], sd[f"{s}.{src}.bias"]
        out[f"V:{k}sd_ln_w"], out[f"V:{k}sd_ln_b"] = sd[s + ".ffn.0.weight"], sd[s + ".ffn.0.bias"]
        # fp32 copies for the folded text cache (K' = K Wq, V' = V Wout^T are built once per caption batch)
        out[f"V:{k}sd_q_w32"], out[f"V:{k}sd_out_w32"] = sd[s + ".query.weight"], sd[s + ".out.weight"]
        # emb_w was appended in slot order local, global, cross, ffn -- same as STYLE_SLOTS
        for slot in STYLE_SLOTS:
            w, b = eph[f"{tag}.{slot}"]
            eph_w.append(w), eph_b.append(b)
    out["W:style_eph"], out["V:style_eph_b"] = torch.cat(eph_w, 0), torch.cat(eph_b)
    out["W:style_emb"], out["V:style_emb_b"] = torch.cat(emb_w, 0), torch.cat(emb_b)
    return out


def _style(out, sd, dst, src):
    out["V:" + dst + "norm_w"], out["V:" + dst + "norm_b"] = sd[src + ".norm.weight"], sd[src + ".norm.bias"]
    out["W:" + dst + "out"], out["V:" + dst + "out_b"] = sd[src + ".out_layers.2.weight"], sd[src + ".out_layers.2.bias"]


class PackedModel:
    """Device-resident packed weights + the ctypes ``MdmModel`` that points at them."""

    def __init__(self, sd: Dict[str, torch.Tensor], cfg: dict, eph, proj, device, with_lo: bool = True,
                 counters: Dict[str, torch.Tensor] = None, precision: int = L.PREC_X3):
        from .ops import PackedWeight  # HIP pack kernel

        self.cfg = dict(cfg)
        self.format_class = format_class(precision)
        head_dim = cfg["latent_dim"] // cfg["num_heads"]
        dev = torch.device(device)
        if dev.type != "cuda":
            raise L.MdmError("PackedModel needs a GPU device: the denoising path has no CPU fallback")
        lay = kernel_layout(sd, cfg, eph, proj)
        self._keep = []
        self.W: Dict[str, PackedWeight] = {}
        self.V: Dict[str, torch.Tensor] = {}
        with torch.cuda.device(dev):
            for k, t in lay.items():
                t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
                if k.startswith("W:"):
                    fmt = weight_format(k[2:], precision, head_dim)
                    self.W[k[2:]] = PackedWeight(t, fmt=fmt if (with_lo or fmt != "bf16x2") else "bf16")
                else:
                    self.V[k[2:]] = t
            D, L_ = cfg["latent_dim"], cfg["num_layers"]
            # weight streams of the expert MLPs for the streamed-weight fused kernel (csrc/mlp_stream.hip): the 16-bit
            # modes at the shapes it takes; MDM_MLP_STREAM=0 keeps the LDS-staged kernel (A/B runs)
            self.wstream = {}
            E2, F_ = 2 * cfg["moe_num_experts"], cfg["ff_size"]
            import os
            if (os.environ.get("MDM_MLP_STREAM", "1") != "0" and D in (512, 1024) and F_ % 256 == 0
                    and precision in (L.PREC_BF16, L.PREC_F16, L.PREC_MIXED)):
                from .ops import mlp_stream_pack
                for li in range(2 * L_):
                    k = f"L{li}."
                    fmt = weight_format(k + "w1", precision, head_dim)
                    dt = torch.float16 if fmt == "f16" else torch.bfloat16
                    self.wstream[k] = mlp_stream_pack(lay["W:" + k + "w1"].to(dev).reshape(E2, F_, D),
                                                      lay["W:" + k + "w2"].to(dev).reshape(E2, D, F_), dt)
            # ... and of the dense Linear-GELU-Linear pairs (Performer output projection, 4x FFN of the text cross-attention),
            # throughput modes only (their activations are 16-bit there); the mixed mode runs the FFN pair in fp16 as well
            if os.environ.get("MDM_MLP_STREAM", "1") != "0" and D == 512 and precision in (L.PREC_BF16, L.PREC_F16, L.PREC_MIXED):
                from .ops import mlp_stream_pack
                for li in range(2 * L_):
                    k = f"L{li}."
                    pairs = [(k + "sd_ffn", k + "sd_f1", k + "sd_f2")]
                    if precision != L.PREC_MIXED:
                        pairs += [(k + w + ".proj", k + w + ".proj0", k + w + ".proj3") for w in ("local", "global")]
                    for name, a, b in pairs:
                        fmt = weight_format(a, precision, head_dim)
                        if fmt not in ("f16", "bf16x2", "bf16"):
                            continue
                        dt = torch.float16 if fmt == "f16" else torch.bfloat16
                        self.wstream[name] = mlp_stream_pack(lay["W:" + a].to(dev), lay["W:" + b].to(dev), dt)
                    # out_layers.2 of the four StylizationBlocks, for the fused stylization kernel (csrc/style_gemm.hip)
                    if precision != L.PREC_MIXED:
                        from .ops import gemm_stream_pack
                        for st in (k + "local.style.", k + "global.style.", k + "ca_style.", k + "ffn_style."):
                            fmt = weight_format(st + "out", precision, head_dim)
                            dt = torch.float16 if fmt == "f16" else torch.bfloat16
                            ws = gemm_stream_pack(lay["W:" + st + "out"].to(dev), dt)
                            if ws is not None:
                                self.wstream[st + "out"] = ws
            # ... and its (hi, lo) pair streams for the fp32-grade form (csrc/style_gemm.hip style_gemm3): every format class whose
            # stylization Linears are packed as bf16 hi + lo planes (fp32-grade and mixed runs)
            if os.environ.get("MDM_MLP_STREAM", "1") != "0" and D == 512 and with_lo:
                from .ops import gemm_stream3_pack
                for li in range(2 * L_):
                    k = f"L{li}."
                    for st in (k + "local.style.", k + "global.style.", k + "ca_style.", k + "ffn_style."):
                        if weight_format(st + "out", precision, head_dim) != "bf16x2":
                            continue
                        ws = gemm_stream3_pack(lay["W:" + st + "out"].to(dev))
                        if ws is not None:
                            self.wstream[st + "out3"] = ws
            # fragment streams of the plain per-layer Linears for the streamed-weight GEMM (csrc/gemm_stream.hip): the 16-bit modes of the
            # big model, whose D x D launches are latency chains on the tile kernel; MDM_GEMM_STREAM=0 keeps the tile kernel (A/B runs)
            self.wstream1 = {}
            if (os.environ.get("MDM_GEMM_STREAM", "1") != "0" and D == 1024
                    and precision in (L.PREC_BF16, L.PREC_F16, L.PREC_FP8)):
                from .ops import gemm_stream1_pack
                for kk, t in lay.items():
                    if not kk.startswith("W:L") or t.dim() != 2 or t.shape[1] not in (512, 1024) or t.shape[0] % 256:
                        continue
                    fmt = weight_format(kk[2:], precision, head_dim)
                    if fmt not in ("f16", "bf16", "bf16x2"):
                        continue
                    ws = gemm_stream1_pack(t.to(dev), torch.float16 if fmt == "f16" else torch.bfloat16)
                    if ws is not None:
                        self.wstream1[kk[2:]] = ws
            # (hi, lo) fragment-pair streams of the expert matrices for the streamed-weight bf16x3 GEMM (csrc/gemm_stream3.hip): the
            # fp32-grade mode, whose expert GEMM pair is 45 % of its step on the tile kernel
            if os.environ.get("MDM_GEMM_STREAM", "1") != "0" and with_lo and precision == L.PREC_X3 and D % 128 == 0 and F_ % 128 == 0:
                from .ops import gemm_stream3x_pack
                for li in range(2 * L_):
                    k = f"L{li}."
                    for name, shape in ((k + "w1", (E2, F_, D)), (k + "w2", (E2, D, F_))):
                        if weight_format(name, precision, head_dim) != "bf16x2":
                            continue
                        ws = gemm_stream3x_pack(lay["W:" + name].to(dev).reshape(*shape))
                        if ws is not None:
                            self.wstream1[name] = ws
            self.layers = (L.Layer * (2 * L_))()
            for li, (pre, tag) in enumerate(layer_tags(L_)):
                self._fill_layer(self.layers[li], f"L{li}.", pre, counters)
            m = L.Model()
            m.D, m.F, m.Dt, m.H = D, cfg["ff_size"], cfg["text_latent_dim"], cfg["num_heads"]
            m.E, m.L, m.feats, m.num_frames = cfg["moe_num_experts"], L_, cfg["input_feats"], cfg["num_frames"]
            for n in L._MODEL_PACKED:
                if n in self.W:
                    setattr(m, n, self._packed(n))
            for n in L._MODEL_BIAS:
                if n in self.V:
                    setattr(m, n, self.V[n].data_ptr())
            m.seq_emb = self.V["seq_emb"].data_ptr()
            m.style_eph, m.style_eph_b = self._packed("style_eph"), self.V["style_eph_b"].data_ptr()
            m.style_emb, m.style_emb_b = self._packed("style_emb"), self.V["style_emb_b"].data_ptr()
            m.layers = C.cast(self.layers, C.POINTER(L.Layer))
            self.model = m
            torch.cuda.current_stream().synchronize()

    def _packed(self, name: str) -> L.Packed:
        w = self.W[name]
        p = L.Packed()
        p.hi, p.lo, p.ld = w.hi.data_ptr(), (w.lo.data_ptr() if w.lo is not None else 0), w.Kp
        if name in getattr(self, "wstream1", {}):
            p.ws = self.wstream1[name].data_ptr()
        return p

    def _style(self, st: L.Style, pre: str):
        st.norm_w, st.norm_b = self.V[pre + "norm_w"].data_ptr(), self.V[pre + "norm_b"].data_ptr()
        st.out, st.out_b = self._packed(pre + "out"), self.V[pre + "out_b"].data_ptr()
        if (pre + "out") in self.wstream:
            st.out_ws = self.wstream[pre + "out"].data_ptr()
        if (pre + "out3") in self.wstream:
            st.out_ws3 = self.wstream[pre + "out3"].data_ptr()

    def _fill_layer(self, l: L.Layer, k: str, sd_prefix: str, counters):
        V, D = self.V, self.cfg["latent_dim"]
        l.dual_pre_w, l.dual_pre_b = V[k + "dual_pre_norm_w"].data_ptr(), V[k + "dual_pre_norm_b"].data_ptr()
        l.dual_post_w, l.dual_post_b = V[k + "dual_post_norm_w"].data_ptr(), V[k + "dual_post_norm_b"].data_ptr()
        for which, p in (("local", l.local), ("global", l.global_)):
            q = k + which + "."
            p.pre_w, p.pre_b = V[q + "pre_norm_w"].data_ptr(), V[q + "pre_norm_b"].data_ptr()
            p.post_w, p.post_b = V[q + "post_norm_w"].data_ptr(), V[q + "post_norm_b"].data_ptr()
            p.qkv, p.qkv_b = self._packed(q + "qkv"), V[q + "qkv_b"].data_ptr()
            p.hn_w, p.hn_b = V[q + "hn_w"].data_ptr(), V[q + "hn_b"].data_ptr()
            p.feat = self._packed(q + "feat")
            p.proj0, p.proj0_b = self._packed(q + "proj0"), V[q + "proj0_b"].data_ptr()
            p.proj3, p.proj3_b = self._packed(q + "proj3"), V[q + "proj3_b"].data_ptr()
            if (q + "proj") in self.wstream:
                p.proj_ws = self.wstream[q + "proj"].data_ptr()
            self._style(p.style, q + "style.")
        l.skip, l.skip_b = self._packed(k + "skip"), V[k + "skip_b"].data_ptr()
        l.ca_norm_w, l.ca_norm_b = V[k + "ca_norm_w"].data_ptr(), V[k + "ca_norm_b"].data_ptr()
        l.ca_tnorm_w, l.ca_tnorm_b = V[k + "ca_tnorm_w"].data_ptr(), V[k + "ca_tnorm_b"].data_ptr()
        for nm in ("ca_q", "ca_k", "ca_v", "sd_q", "sd_k", "sd_v", "sd_out", "sd_f1", "sd_f2"):
            setattr(l, nm, self._packed(k + nm))
            setattr(l, nm + "_b", V[k + nm + "_b"].data_ptr())
        l.sd_q_w32, l.sd_out_w32 = V[k + "sd_q_w32"].data_ptr(), V[k + "sd_out_w32"].data_ptr()
        gvec = torch.empty(D, dtype=torch.float32, device=V[k + "ca_gate"].device)
        L.check(L.lib().mdm_xattn_gate(C.c_void_p(V[k + "ca_gate"].data_ptr()), C.c_void_p(V[k + "ca_adaptive"].data_ptr()),
                                       C.c_int32(D), C.c_void_p(gvec.data_ptr()), C.c_void_p(L.stream_ptr())),
                "mdm_xattn_gate")
        V[k + "ca_gvec"] = gvec
        l.ca_gvec = gvec.data_ptr()
        self._style(l.ca_style, k + "ca_style.")
        for b in range(2):
            l.moe_ln_w[b], l.moe_ln_b[b] = V[f"{k}moe_ln_w{b}"].data_ptr(), V[f"{k}moe_ln_b{b}"].data_ptr()
            l.gate_w[b], l.gate_b[b] = V[f"{k}gate_w{b}"].data_ptr(), V[f"{k}gate_b{b}"].data_ptr()
            if counters is not None:
                br = f"{sd_prefix}.ffn.branches.{b}.moe"
                l.usage[b] = counters[br + ".expert_usage"].data_ptr()
                l.importance[b] = counters[br + ".expert_importance"].data_ptr()
        l.w1, l.b1 = self._packed(k + "w1"), V[k + "b1"].data_ptr()
        l.w2, l.b2 = self._packed(k + "w2"), V[k + "b2"].data_ptr()
        if (k + "sd_ffn") in self.wstream:
            l.sd_ffn_ws = self.wstream[k + "sd_ffn"].data_ptr()
        ws = self.wstream.get(k)
        if ws is not None:
            l.wstream, l.wstream_gs = ws.data_ptr(), 2 * D * self.cfg["ff_size"]
        self._style(l.ffn_style, k + "ffn_style.")
        l.sd_ln_w, l.sd_ln_b = V[k + "sd_ln_w"].data_ptr(), V[k + "sd_ln_b"].data_ptr()

    def nbytes(self) -> int:
        n = sum(w.hi.numel() * w.hi.element_size() + (w.lo.numel() * w.lo.element_size() if w.lo is not None else 0)
                for w in self.W.values())
        return n + sum(v.numel() * 4 for v in self.V.values())
