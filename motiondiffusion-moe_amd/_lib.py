"""ctypes binding of libmdm_hip.so (include/mdm_hip.h).  The product path has NO fallback:
if the library is missing or a call fails this raises."""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MDM_LIB") or os.path.join(_HERE, "libmdm_hip.so")  # MDM_LIB: A/B benchmarking of builds
_lib = None

OP_F32_ROW, OP_F32_KSTRIDE, OP_BF16_ROW, OP_FP8_ROW, OP_X2_ROW = 0, 1, 2, 3, 4
ACT_NONE, ACT_GELU, ACT_SILU, ACT_FEAT, ACT_HEADNORM, ACT_HEADSOFTMAX = 0, 1, 2, 3, 4, 5
H16_BF16, H16_F16 = 1, 2  # MDM_H16_*
PREC_BF16, PREC_F16, PREC_X3, PREC_MIXED, PREC_FP8 = 1, 2, 3, 4, 5  # MDM_PREC_*
PRECISIONS = (PREC_BF16, PREC_F16, PREC_X3, PREC_MIXED, PREC_FP8)
PREC_NAMES = {1: "bf16", 2: "f16", 3: "bf16x3(fp32-grade)", 4: "mixed(bf16x3 + f16 expert/FFN GEMMs)",
              5: "f16 + fp8(e4m3) expert GEMMs"}


class MdmError(RuntimeError):
    pass


class Operand(C.Structure):
    _fields_ = [("p", C.c_void_p), ("p_lo", C.c_void_p), ("ld", C.c_int64), ("gstride", C.c_int64),
                ("gather", C.c_void_p), ("bs1", C.c_int64), ("bs2", C.c_int64), ("rpg", C.c_int32),
                ("kind", C.c_int32)]


class GemmDesc(C.Structure):
    _fields_ = [("A", Operand), ("W", Operand), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("batch", C.c_int32), ("nb2", C.c_int32), ("goff", C.c_void_p), ("ngroups", C.c_int32),
                ("act", C.c_int32), ("C", C.c_void_p), ("ldc", C.c_int64), ("c_bs1", C.c_int64),
                ("c_bs2", C.c_int64), ("C16", C.c_void_p), ("bias", C.c_void_p), ("bias_bs", C.c_int64), ("alpha", C.c_float),
                ("out_scale", C.c_float), ("r1_scale", C.c_float), ("r1_mod", C.c_int32),
                ("colscale", C.c_void_p), ("rowscale", C.c_void_p), ("R1", C.c_void_p), ("ldr1", C.c_int64),
                ("R2", C.c_void_p), ("ldr2", C.c_int64), ("feat_len", C.c_void_p), ("feat_S", C.c_int32),
                ("feat_rpt", C.c_int32), ("feat_kslot", C.c_int32), ("precision", C.c_int32), ("h16", C.c_int32),
                ("a_scale", C.c_void_p), ("w_scale", C.c_void_p), ("a_scale_u", C.c_float), ("C8", C.c_void_p),
                ("c8_scale", C.c_float), ("kgoff", C.c_void_p), ("hn_w", C.c_void_p), ("hn_b", C.c_void_p),
                ("C16_lo", C.c_void_p), ("hn_l2_tiles", C.c_int32), ("Cx2", C.c_void_p), ("w_stream", C.c_void_p), ("w_stream_gs", C.c_int64)]


class MoeTensors(C.Structure):
    """MdmMoeTensors: parameters (or gradients) of one MoE feed-forward block, branches stacked (include/mdm_hip.h)."""
    NAMES = ("ln_w", "ln_b", "gate_w", "gate_b", "w1", "b1", "w2", "b2", "st_emb_w", "st_emb_b", "st_norm_w", "st_norm_b",
             "st_out_w", "st_out_b")
    _fields_ = [(n, C.c_void_p) for n in NAMES]


class MlpDesc(C.Structure):
    _fields_ = [("X", C.c_void_p), ("ldx", C.c_int64), ("gather", C.c_void_p), ("M", C.c_int32), ("Din", C.c_int32),
                ("F", C.c_int32), ("Dout", C.c_int32), ("goff", C.c_void_p), ("ngroups", C.c_int32),
                ("w1", C.c_void_p), ("ldw1", C.c_int64), ("w1_gs", C.c_int64), ("b1", C.c_void_p), ("b1_gs", C.c_int64),
                ("w2", C.c_void_p), ("ldw2", C.c_int64), ("w2_gs", C.c_int64), ("b2", C.c_void_p), ("b2_gs", C.c_int64),
                ("rowscale", C.c_void_p), ("R1", C.c_void_p), ("ldr1", C.c_int64), ("r1_scale", C.c_float),
                ("R2", C.c_void_p), ("ldr2", C.c_int64), ("C", C.c_void_p), ("C16", C.c_void_p), ("ldc", C.c_int64),
                ("h16", C.c_int32), ("wstream", C.c_void_p), ("wstream_gs", C.c_int64)]


class Packed(C.Structure):
    _fields_ = [("hi", C.c_void_p), ("lo", C.c_void_p), ("ld", C.c_int64), ("ws", C.c_void_p)]


class Style(C.Structure):
    _fields_ = [("norm_w", C.c_void_p), ("norm_b", C.c_void_p), ("out", Packed), ("out_b", C.c_void_p), ("out_ws", C.c_void_p),
                ("out_ws3", C.c_void_p)]


class Performer(C.Structure):
    _fields_ = [("pre_w", C.c_void_p), ("pre_b", C.c_void_p), ("post_w", C.c_void_p), ("post_b", C.c_void_p),
                ("qkv", Packed), ("qkv_b", C.c_void_p), ("hn_w", C.c_void_p), ("hn_b", C.c_void_p),
                ("feat", Packed), ("proj0", Packed), ("proj3", Packed), ("proj0_b", C.c_void_p),
                ("proj3_b", C.c_void_p), ("proj_ws", C.c_void_p), ("style", Style)]


_P2 = C.c_void_p * 2


class Layer(C.Structure):
    _fields_ = [("dual_pre_w", C.c_void_p), ("dual_pre_b", C.c_void_p), ("dual_post_w", C.c_void_p),
                ("dual_post_b", C.c_void_p), ("local", Performer), ("global_", Performer), ("skip", Packed),
                ("skip_b", C.c_void_p),
                ("ca_norm_w", C.c_void_p), ("ca_norm_b", C.c_void_p), ("ca_tnorm_w", C.c_void_p),
                ("ca_tnorm_b", C.c_void_p), ("ca_q", Packed), ("ca_k", Packed), ("ca_v", Packed),
                ("ca_q_b", C.c_void_p), ("ca_k_b", C.c_void_p), ("ca_v_b", C.c_void_p), ("ca_gvec", C.c_void_p),
                ("ca_style", Style),
                ("moe_ln_w", _P2), ("moe_ln_b", _P2), ("gate_w", _P2), ("gate_b", _P2), ("w1", Packed), ("w2", Packed),
                ("wstream", C.c_void_p), ("wstream_gs", C.c_int64), ("b1", C.c_void_p), ("b2", C.c_void_p), ("usage", _P2), ("importance", _P2), ("ffn_style", Style),
                ("sd_q", Packed), ("sd_k", Packed), ("sd_v", Packed), ("sd_out", Packed), ("sd_f1", Packed),
                ("sd_f2", Packed), ("sd_q_b", C.c_void_p), ("sd_k_b", C.c_void_p), ("sd_v_b", C.c_void_p),
                ("sd_out_b", C.c_void_p), ("sd_ln_w", C.c_void_p), ("sd_ln_b", C.c_void_p), ("sd_f1_b", C.c_void_p),
                ("sd_f2_b", C.c_void_p), ("sd_q_w32", C.c_void_p), ("sd_out_w32", C.c_void_p), ("sd_ffn_ws", C.c_void_p)]


_MODEL_PACKED = ["tmlp0", "tmlp2", "te0", "te2", "tproj", "gf_time", "gf_text", "gf_post0", "gf_post2", "text_proj",
                 "joint", "down", "up", "out"]
_MODEL_BIAS = ["tmlp0_b", "tmlp2_b", "te0_b", "te2_b", "tproj_b", "gf_time_b", "gf_text_b", "gf_post0_b", "gf_post2_b",
               "text_proj_b", "joint_b", "down_b", "up_b2", "out_b"]


class Model(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("D", "F", "Dt", "H", "E", "L", "feats", "num_frames")]
                + [(n, Packed) for n in _MODEL_PACKED] + [(n, C.c_void_p) for n in _MODEL_BIAS]
                + [("seq_emb", C.c_void_p), ("style_eph", Packed), ("style_eph_b", C.c_void_p),
                   ("style_emb", Packed), ("style_emb_b", C.c_void_p), ("layers", C.POINTER(Layer))])


class TextCache(C.Structure):
    _fields_ = [("lin_at", C.c_void_p), ("sd_k", C.c_void_p), ("sd_v", C.c_void_p), ("B", C.c_int32),
                ("N", C.c_int32), ("sd_kfold", C.c_void_p), ("sd_cb", C.c_void_p), ("sd_vfold", C.c_void_p),
                ("ntok", C.c_void_p)]


class StemCache(C.Structure):
    _fields_ = [("time_table", C.c_void_p), ("gx", C.c_void_p), ("steps", C.c_int32)]


BLOCK_DUAL, BLOCK_CROSS, BLOCK_MOE, BLOCK_SDCROSS, BLOCK_LAYER = 0, 1, 2, 3, 4
NOISE_STREAM_XT = 0x7FFFFFFF  # MDM_NOISE_STREAM_XT
TAB_ROWS = 7  # sqrt_recip_acp, sqrt_recipm1_acp, coef1, coef2, post_logvar_clipped, acp, acp_prev


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MdmError(f"{LIB_PATH} not found: build it with `python motiondiffusion-moe_amd/build.py` "
                           "(hipcc, gfx950). There is no CPU/eager fallback for the denoising path.")
        L = C.CDLL(LIB_PATH)
        L.mdm_version.restype = C.c_char_p
        for name in EXPORTS:
            if name in ("mdm_workspace_bytes", "mdm_text_head_workspace_bytes", "mdm_moe_train_workspace_bytes", "mdm_mlp_stream_elems", "mdm_gemm_stream_elems", "mdm_gemm_stream3_elems", "mdm_gemm_stream1_elems", "mdm_gemm_stream3x_elems", "mdm_gemm_stream3x_group_elems"):
                getattr(L, name).restype = C.c_int64
            elif name != "mdm_version":
                getattr(L, name).restype = C.c_int
        _lib = L
    return _lib


# every symbol include/mdm_hip.h declares (checked by tests/test_abi.py)
EXPORTS = ["mdm_version", "mdm_gemm", "mdm_fused_mlp", "mdm_mlp_stream_elems", "mdm_mlp_stream_pack", "mdm_gemm_stream_elems", "mdm_gemm_stream_pack", "mdm_gemm_stream3_elems", "mdm_gemm_stream3_pack", "mdm_gemm_stream1_elems", "mdm_gemm_stream1_pack", "mdm_gemm_stream3x_elems", "mdm_gemm_stream3x_group_elems", "mdm_gemm_stream3x_pack", "mdm_pack_bf16", "mdm_pack_f16", "mdm_pack_fp8", "mdm_workspace_bytes", "mdm_text_cache_build",
           "mdm_denoiser_forward", "mdm_stem_cache_build", "mdm_block_forward", "mdm_moe_ffn_forward", "mdm_dual_self_attn_forward", "mdm_linear_xattn_forward",
           "mdm_softmax_xattn_ffn_forward", "mdm_performer_attn_forward", "mdm_stylization_forward", "mdm_stem_embeddings",
           "mdm_cfg_posterior_step", "mdm_ddim_step", "mdm_noise_normal", "mdm_noise_normal_ids", "mdm_text_head_workspace_bytes", "mdm_text_head_forward", "mdm_motion_postprocess", "mdm_xattn_gate", "mdm_fill_i64", "mdm_add_i32", "mdm_set_gemm_variant", "mdm_diag_build", "mdm_diag_mlp_counters", "mdm_debug_stamps", "mdm_probe_enable", "mdm_probe_read", "mdm_route_dump",
           "mdm_moe_train_workspace_bytes", "mdm_moe_ffn_train_forward", "mdm_moe_ffn_train_backward", "mdm_sumsq", "mdm_adam_step", "mdm_sd_fold_passes"]


def check(status: int, what: str = "mdm call"):
    if status != 0:
        names = {1: "MDM_ERR_ARG", 2: "MDM_ERR_LAUNCH", 3: "MDM_ERR_UNSUPPORTED"}
        raise MdmError(f"{what} failed: {names.get(status, status)}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MdmError("mdm HIP path needs tensors on a GPU device (no CPU fallback)")
