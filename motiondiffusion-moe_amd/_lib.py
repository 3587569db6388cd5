"""ctypes binding of libmdm_hip.so (include/mdm_hip.h).  The product path has NO fallback:
if the library is missing or a call fails this raises."""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmdm_hip.so")
_lib = None

OP_F32_ROW, OP_F32_KSTRIDE, OP_BF16_ROW = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_SILU, ACT_FEAT = 0, 1, 2, 3


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
                ("c_bs2", C.c_int64), ("bias", C.c_void_p), ("bias_bs", C.c_int64), ("alpha", C.c_float),
                ("out_scale", C.c_float), ("r1_scale", C.c_float), ("r1_mod", C.c_int32),
                ("colscale", C.c_void_p), ("rowscale", C.c_void_p), ("R1", C.c_void_p), ("ldr1", C.c_int64),
                ("R2", C.c_void_p), ("ldr2", C.c_int64), ("feat_len", C.c_void_p), ("feat_S", C.c_int32),
                ("feat_rpt", C.c_int32), ("feat_kslot", C.c_int32), ("precision", C.c_int32)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MdmError(f"{LIB_PATH} not found: build it with `python motiondiffusion-moe_amd/build.py` "
                           "(hipcc, gfx950). There is no CPU/eager fallback for the denoising path.")
        L = C.CDLL(LIB_PATH)
        L.mdm_version.restype = C.c_char_p
        for name in EXPORTS:
            if name != "mdm_version":
                getattr(L, name).restype = C.c_int
        _lib = L
    return _lib


# every symbol include/mdm_hip.h declares (checked by tests/test_abi.py)
EXPORTS = ["mdm_version", "mdm_gemm", "mdm_pack_bf16"]


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
