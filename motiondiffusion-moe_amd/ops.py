"""Thin host wrappers over the C ABI: one Python function per kernel family.
Only plumbing lives here (pointer/stride marshalling); all arithmetic is in csrc/."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L


class PackedWeight:
    """16-bit planes of an fp32 [N, K] weight, K zero-padded to a multiple of 32.
    fmt "bf16x2": bf16 hi + lo planes (`lo` = rn(w - hi), the bf16x3 fp32-grade mode; `hi` alone serves a single bf16 pass);
    "bf16": hi plane only; "f16": one IEEE fp16 plane in `hi` (single fp16 pass, include/mdm_hip.h MDM_H16_F16)."""

    def __init__(self, w: torch.Tensor, with_lo: bool = True, fmt: Optional[str] = None):
        L.require_cuda(w)
        fmt = fmt or ("bf16x2" if with_lo else "bf16")
        if fmt not in ("bf16x2", "bf16", "f16", "f8"):
            raise ValueError(f"unknown packed weight format {fmt!r}")
        self.fmt = fmt
        with_lo = fmt == "bf16x2"
        if fmt == "f8":  # e4m3 bytes (K padded to 128) in `hi`, per-row fp32 scales (amax / 448) in `lo`: csrc/gemm8.hip
            w = w.detach().to(torch.float32)
            self.lead = tuple(w.shape[:-2])
            n, k = w.shape[-2], w.shape[-1]
            w2 = w.reshape(-1, k).contiguous()
            kp = (k + 127) // 128 * 128
            self.N, self.K, self.Kp = n, k, kp
            self.hi = torch.empty((w2.shape[0], kp), dtype=torch.uint8, device=w.device)
            self.lo = torch.empty((w2.shape[0],), dtype=torch.float32, device=w.device)
            with torch.cuda.device(w.device):
                L.check(L.lib().mdm_pack_fp8(C.c_void_p(w2.data_ptr()), C.c_int64(k), C.c_int64(w2.shape[0]), C.c_int64(k),
                                             C.c_void_p(self.hi.data_ptr()), C.c_int64(kp), C.c_void_p(self.lo.data_ptr()),
                                             C.c_void_p(L.stream_ptr())), "mdm_pack_fp8")
            return
        w = w.detach().to(torch.float32)
        lead = w.shape[:-2]
        n, k = w.shape[-2], w.shape[-1]
        w2 = w.reshape(-1, k).contiguous()
        kp = (k + 31) // 32 * 32
        self.N, self.K, self.Kp = n, k, kp
        self.hi = torch.empty((w2.shape[0], kp), dtype=torch.float16 if fmt == "f16" else torch.bfloat16, device=w.device)
        self.lo = torch.empty_like(self.hi) if with_lo else None
        with torch.cuda.device(w.device):
            if fmt == "f16":
                L.check(L.lib().mdm_pack_f16(C.c_void_p(w2.data_ptr()), C.c_int64(k), C.c_int64(w2.shape[0]), C.c_int64(k),
                                             C.c_void_p(self.hi.data_ptr()), C.c_int64(kp), C.c_void_p(L.stream_ptr())),
                        "mdm_pack_f16")
            else:
                L.check(L.lib().mdm_pack_bf16(C.c_void_p(w2.data_ptr()), C.c_int64(k), C.c_int64(w2.shape[0]), C.c_int64(k),
                                              C.c_void_p(self.hi.data_ptr()), C.c_void_p(L.ptr(self.lo)), C.c_int64(kp),
                                              C.c_void_p(L.stream_ptr())), "mdm_pack_bf16")
        self.lead = tuple(lead)

    def operand(self, row_offset: int = 0) -> L.Operand:
        o = L.Operand()
        o.p = self.hi.data_ptr() + row_offset * self.Kp * 2
        o.p_lo = (self.lo.data_ptr() + row_offset * self.Kp * 2) if self.lo is not None else 0
        o.ld = self.Kp
        o.kind = L.OP_BF16_ROW
        return o


def f32_operand(t: torch.Tensor, ld: int, kind: int = L.OP_F32_ROW, offset: int = 0) -> L.Operand:
    o = L.Operand()
    o.p = t.data_ptr() + 4 * offset
    o.ld = ld
    o.kind = kind
    return o


def gemm_desc(precision: int) -> L.GemmDesc:
    d = L.GemmDesc()
    d.batch, d.nb2 = 1, 1
    d.alpha, d.out_scale, d.r1_scale = 1.0, 1.0, 1.0
    d.precision = precision
    return d


def run_gemm(d: L.GemmDesc):
    L.check(L.lib().mdm_gemm(C.byref(d), C.c_void_p(L.stream_ptr())), "mdm_gemm")


def linear(x: torch.Tensor, w: PackedWeight, bias: Optional[torch.Tensor] = None, *, act: int = L.ACT_NONE,
           alpha: float = 1.0, out_scale: float = 1.0, colscale=None, rowscale=None, r1=None, r1_scale: float = 1.0,
           r1_mod: int = 0, r2=None, precision: int = 3, out: Optional[torch.Tensor] = None,
           out16: Optional[torch.Tensor] = None, w_stream: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = epilogue(x @ w^T): the fused Linear used throughout the denoiser.  bf16 ``x`` selects the throughput
    kernel (gemm2.hip); ``out16`` receives an optional bf16 copy of the result; ``w_stream`` (gemm_stream1_pack of the same
    weight) lets an eligible 16-bit launch run on the streamed-weight kernel (gemm_stream.hip)."""
    L.require_cuda(x)
    K = x.shape[-1]
    assert K == w.K, (K, w.K)
    x2 = x.reshape(-1, K)
    assert x2.stride(-1) == 1
    M = x2.shape[0]
    if out is None:
        out = torch.empty((M, w.N), dtype=torch.float32, device=x.device)
    d = gemm_desc(precision)
    if x2.dtype in (torch.bfloat16, torch.float16):  # 16-bit rows: single-pass kernel in that format (w packed alike)
        assert (x2.dtype == torch.float16) == (w.fmt == "f16"), "activation and weight 16-bit formats must match"
        d.A.p, d.A.ld, d.A.kind = x2.data_ptr(), x2.stride(0), L.OP_BF16_ROW
        d.precision, d.h16 = 1, (L.H16_F16 if x2.dtype == torch.float16 else L.H16_BF16)
    else:
        d.A = f32_operand(x2, x2.stride(0))
    d.W = w.operand()
    d.w_stream = L.ptr(w_stream)
    d.M, d.N, d.K = M, w.N, K
    d.C, d.ldc = out.data_ptr(), out.stride(0)
    if out16 is not None:
        assert out16.stride(0) == out.stride(0)
        d.C16 = out16.data_ptr()
        d.h16 = L.H16_F16 if out16.dtype == torch.float16 else L.H16_BF16
    d.bias = L.ptr(bias)
    d.act, d.alpha, d.out_scale = act, alpha, out_scale
    d.colscale, d.rowscale = L.ptr(colscale), L.ptr(rowscale)
    if r1 is not None:
        r1 = r1.reshape(-1, w.N)
        d.R1, d.ldr1, d.r1_scale, d.r1_mod = r1.data_ptr(), r1.stride(0), r1_scale, r1_mod
    if r2 is not None:
        r2 = r2.reshape(-1, w.N)
        d.R2, d.ldr2 = r2.data_ptr(), r2.stride(0)
    run_gemm(d)
    return out.reshape(*x.shape[:-1], w.N)


def quantize_rows_fp8(x: torch.Tensor):
    """fp32 rows (M, K) -> (e4m3 bytes (M, Kp) uint8, scales (M,) fp32) with the library's own quantiser (amax / 448 per row,
    K zero-padded to a multiple of 128): what the router kernel writes for the experts in the fp8 mode."""
    L.require_cuda(x)
    pw = PackedWeight(x, fmt="f8")
    return pw.hi, pw.lo


def gemm_fp8(a8: torch.Tensor, a_scale: Optional[torch.Tensor], w: PackedWeight, bias: Optional[torch.Tensor] = None, *,
             act: int = L.ACT_NONE, a_scale_u: float = 1.0, rowscale=None, gather=None, goff=None, rows: Optional[int] = None,
             out: Optional[torch.Tensor] = None, out8: Optional[torch.Tensor] = None, c8_scale: float = 1.0) -> torch.Tensor:
    """C = act((A8 W8^T) * a_scale_u * a_scale[src row] * w_scale[n] + bias) * rowscale  (csrc/gemm8.hip).  ``goff`` selects
    grouped mode (w / bias carry a leading group axis); ``out8`` receives e4m3(C * c8_scale)."""
    assert w.fmt == "f8" and a8.dtype == torch.uint8 and a8.stride(-1) == 1
    M = a8.shape[0] if rows is None else rows
    d = gemm_desc(1)
    d.A.p, d.A.ld, d.A.kind, d.A.gather = a8.data_ptr(), a8.stride(0), L.OP_FP8_ROW, L.ptr(gather)
    d.W.p, d.W.ld, d.W.kind = w.hi.data_ptr(), w.Kp, L.OP_FP8_ROW
    d.M, d.N, d.K = M, w.N, w.Kp
    d.a_scale, d.w_scale, d.a_scale_u, d.c8_scale = L.ptr(a_scale), w.lo.data_ptr(), a_scale_u, c8_scale
    if goff is not None:
        d.goff, d.ngroups = goff.data_ptr(), goff.numel() - 1
        d.W.bs1, d.bias_bs = w.N * w.Kp, w.N
    d.bias, d.act, d.rowscale = L.ptr(bias), act, L.ptr(rowscale)
    if out is None and out8 is None:
        out = torch.empty((M, w.N), dtype=torch.float32, device=a8.device)
    if out is not None:
        d.C, d.ldc = out.data_ptr(), out.stride(0)
    if out8 is not None:
        d.C8, d.ldc = out8.data_ptr(), out8.stride(0)
        assert out is None or out.stride(0) == out8.stride(0)
    run_gemm(d)
    return out if out is not None else out8


def mlp_stream_pack(w1: torch.Tensor, w2: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Weight stream of the streamed-weight fused MLP (csrc/mlp_stream.hip, include/mdm_hip.h mdm_mlp_stream_pack):
    fp32 w1 (G, F, Din) / w2 (G, Dout, F) (or without the group axis) -> one 16-bit buffer holding, per (group, wave), the
    1-KiB MFMA fragments of both layers in the order the wave consumes them (+ 16 KiB of tail padding)."""
    L.require_cuda(w1, w2)
    if w1.dim() == 2:
        w1, w2 = w1[None], w2[None]
    G, F, Din = w1.shape
    Dout = w2.shape[1]
    assert w2.shape == (G, Dout, F) and dtype in (torch.float16, torch.bfloat16)
    w1 = w1.detach().to(torch.float32).contiguous()
    w2 = w2.detach().to(torch.float32).contiguous()
    n = L.lib().mdm_mlp_stream_elems(C.c_int32(G), C.c_int32(F), C.c_int32(Din), C.c_int32(Dout))
    out = torch.empty(n, dtype=dtype, device=w1.device)
    with torch.cuda.device(w1.device):
        L.check(L.lib().mdm_mlp_stream_pack(C.c_void_p(w1.data_ptr()), C.c_void_p(w2.data_ptr()), C.c_int32(G), C.c_int32(F),
                                            C.c_int32(Din), C.c_int32(Dout),
                                            C.c_int32(L.H16_F16 if dtype == torch.float16 else L.H16_BF16),
                                            C.c_void_p(out.data_ptr()), C.c_void_p(L.stream_ptr())), "mdm_mlp_stream_pack")
    return out


def gemm_stream_pack(w: torch.Tensor, dtype: torch.dtype) -> Optional[torch.Tensor]:
    """Weight stream of one fp32 Linear [N, K] for the fused stylization kernel (csrc/style_gemm.hip); None when the shape
    is not taken (N = K = 512 only)."""
    L.require_cuda(w)
    N, K = w.shape
    n = L.lib().mdm_gemm_stream_elems(C.c_int32(N), C.c_int32(K))
    if n <= 0:
        return None
    w = w.detach().to(torch.float32).contiguous()
    out = torch.empty(n, dtype=dtype, device=w.device)
    with torch.cuda.device(w.device):
        L.check(L.lib().mdm_gemm_stream_pack(C.c_void_p(w.data_ptr()), C.c_int32(N), C.c_int32(K),
                                             C.c_int32(L.H16_F16 if dtype == torch.float16 else L.H16_BF16),
                                             C.c_void_p(out.data_ptr()), C.c_void_p(L.stream_ptr())), "mdm_gemm_stream_pack")
    return out


def gemm_stream1_pack(w: torch.Tensor, dtype: torch.dtype) -> Optional[torch.Tensor]:
    """Fragment stream of one fp32 Linear [N, K] for the streamed-weight GEMM of the 16-bit modes (csrc/gemm_stream.hip;
    MdmGemmDesc.w_stream / MdmPacked.ws); None when the shape is not covered (N % 256, K % 256)."""
    L.require_cuda(w)
    N, K = w.shape
    n = L.lib().mdm_gemm_stream1_elems(C.c_int32(N), C.c_int32(K))
    if n <= 0:
        return None
    w = w.detach().to(torch.float32).contiguous()
    out = torch.empty(n, dtype=dtype, device=w.device)
    with torch.cuda.device(w.device):
        L.check(L.lib().mdm_gemm_stream1_pack(C.c_void_p(w.data_ptr()), C.c_int64(K), C.c_int32(N), C.c_int32(K),
                                              C.c_int32(L.H16_F16 if dtype == torch.float16 else L.H16_BF16),
                                              C.c_void_p(out.data_ptr()), C.c_void_p(L.stream_ptr())), "mdm_gemm_stream1_pack")
    return out


def gemm_stream3x_pack(w: torch.Tensor) -> Optional[torch.Tensor]:
    """(bf16 hi, lo) fragment-pair stream of G stacked fp32 Linears [G, N, K] (or one [N, K]) for the streamed-weight bf16x3 GEMM
    (csrc/gemm_stream3.hip; MdmGemmDesc.w_stream with pre-split activation rows); None when the shape is not covered."""
    L.require_cuda(w)
    w3 = w if w.dim() == 3 else w[None]
    G, N, K = w3.shape
    n = L.lib().mdm_gemm_stream3x_elems(C.c_int32(G), C.c_int32(N), C.c_int32(K))
    if n <= 0:
        return None
    w3 = w3.detach().to(torch.float32).contiguous()
    out = torch.empty(n, dtype=torch.bfloat16, device=w.device)
    with torch.cuda.device(w.device):
        L.check(L.lib().mdm_gemm_stream3x_pack(C.c_void_p(w3.data_ptr()), C.c_int64(K), C.c_int32(G), C.c_int32(N), C.c_int32(K),
                                               C.c_void_p(out.data_ptr()), C.c_void_p(L.stream_ptr())), "mdm_gemm_stream3x_pack")
    return out


def gemm_stream3_pack(w: torch.Tensor) -> Optional[torch.Tensor]:
    """(bf16 hi, lo) fragment-pair stream of one fp32 Linear [N, K] for the fp32-grade form of the fused stylization kernel
    (csrc/style_gemm.hip style_gemm3); None when the shape is not taken (N = K = 512 only)."""
    L.require_cuda(w)
    N, K = w.shape
    n = L.lib().mdm_gemm_stream3_elems(C.c_int32(N), C.c_int32(K))
    if n <= 0:
        return None
    w = w.detach().to(torch.float32).contiguous()
    out = torch.empty(n, dtype=torch.bfloat16, device=w.device)
    with torch.cuda.device(w.device):
        L.check(L.lib().mdm_gemm_stream3_pack(C.c_void_p(w.data_ptr()), C.c_int32(N), C.c_int32(K), C.c_void_p(out.data_ptr()),
                                              C.c_void_p(L.stream_ptr())), "mdm_gemm_stream3_pack")
    return out


def mlp_stream_pack_reference(w1: torch.Tensor, w2: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """The same layout written as torch reshapes (tests check the packer kernel against it); csrc/mlp_stream.hip documents
    the order: per (group, wave)  W1(0) H0(0) | W1(1) H1(0) H0(1) | ... | W1(C-1) H1(C-2) H0(C-1) | H1(C-1)."""
    if w1.dim() == 2:
        w1, w2 = w1[None], w2[None]
    G, F, Din = w1.shape
    Dout = w2.shape[1]
    NJ, Cn, KT = Dout // 128, F // 256, Din // 32
    # W1 fragments: [g, w, c, step, j, lane = (q, r), e]
    a = w1.reshape(G, Cn, 8, 2, 16, KT, 4, 8).permute(0, 2, 1, 5, 3, 6, 4, 7).reshape(G, 8, Cn, -1)
    # hidden-image position p = 16 w' + u of half h holds unit 32 w' + 16 h + u of the chunk: reorder w2's k axis to (c, h, p)
    w2p = w2.reshape(G, Dout, Cn, 8, 2, 16).permute(0, 1, 2, 4, 3, 5).reshape(G, Dout, Cn, 2, 128)
    # phase-2 fragments of one half: [g, w, c, h, s, j, lane = (q, r), e]
    b = w2p.reshape(G, 8, NJ, 16, Cn, 2, 4, 4, 8).permute(0, 1, 4, 5, 6, 2, 7, 3, 8).reshape(G, 8, Cn, 2, -1)
    parts = []
    for c in range(Cn):
        parts.append(a[:, :, c])
        if c > 0:
            parts.append(b[:, :, c - 1, 1])
        parts.append(b[:, :, c, 0])
    parts.append(b[:, :, Cn - 1, 1])
    s = torch.cat(parts, dim=2).to(dtype).reshape(-1)
    return torch.cat([s, torch.zeros(16 * 512, dtype=dtype, device=s.device)])


def fused_mlp(x16: torch.Tensor, w1: PackedWeight, b1: Optional[torch.Tensor], w2: PackedWeight,
              b2: Optional[torch.Tensor], *, gather=None, goff=None, rowscale=None, r1=None, r1_scale: float = 1.0,
              r2=None, rows: Optional[int] = None, out: Optional[torch.Tensor] = None,
              out16: Optional[torch.Tensor] = None, wstream: Optional[torch.Tensor] = None,
              only16: bool = False) -> torch.Tensor:
    """y = (GELU(x w1^T + b1) w2^T + b2) * rowscale + r1_scale * r1 + r2 with the hidden layer kept on chip (mlp.hip;
    mlp_stream.hip when ``wstream`` is given).
    ``goff`` (int32 [G+1], device) selects grouped mode: w1 / w2 / b1 / b2 then carry a leading group axis."""
    L.require_cuda(x16)
    assert x16.dtype in (torch.bfloat16, torch.float16) and x16.stride(-1) == 1
    assert (x16.dtype == torch.float16) == (w1.fmt == "f16") == (w2.fmt == "f16"), "16-bit formats of x, w1, w2 must match"
    x2 = x16.reshape(-1, x16.shape[-1])
    M = x2.shape[0] if rows is None else rows
    F, Dout = w1.N, w2.N
    assert w1.K == x2.shape[1] and w2.K == F
    if only16:  # 16-bit output only (what the model's expert MLPs write in the throughput modes): returns out16
        assert out16 is not None and out is None
    elif out is None:
        out = torch.empty((M, Dout), dtype=torch.float32, device=x16.device)
    d = L.MlpDesc()
    d.h16 = L.H16_F16 if x16.dtype == torch.float16 else L.H16_BF16
    d.X, d.ldx, d.gather = x2.data_ptr(), x2.stride(0), L.ptr(gather)
    d.M, d.Din, d.F, d.Dout = M, w1.K, F, Dout
    if goff is not None:
        d.goff, d.ngroups = goff.data_ptr(), goff.numel() - 1
        d.w1_gs, d.w2_gs = F * w1.Kp, Dout * w2.Kp
        d.b1_gs, d.b2_gs = F, Dout
    d.w1, d.ldw1, d.b1 = w1.hi.data_ptr(), w1.Kp, L.ptr(b1)
    d.w2, d.ldw2, d.b2 = w2.hi.data_ptr(), w2.Kp, L.ptr(b2)
    d.rowscale = L.ptr(rowscale)
    if wstream is not None:  # from mlp_stream_pack: selects the streamed-weight kernel when the shape fits
        assert wstream.dtype == x16.dtype
        d.wstream, d.wstream_gs = wstream.data_ptr(), F * w1.K + Dout * F
    d.r1_scale = r1_scale
    if r1 is not None:
        d.R1, d.ldr1 = r1.data_ptr(), r1.stride(0)
    if r2 is not None:
        d.R2, d.ldr2 = r2.data_ptr(), r2.stride(0)
    if out is not None:
        d.C, d.ldc = out.data_ptr(), out.stride(0)
    if out16 is not None:
        assert out is None or out16.stride(0) == out.stride(0)
        d.C16, d.ldc = out16.data_ptr(), out16.stride(0)
    L.check(L.lib().mdm_fused_mlp(C.byref(d), C.c_void_p(L.stream_ptr())), "mdm_fused_mlp")
    return out if out is not None else out16
