// Common device helpers for the gfx950 (MI355X) denoiser kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mdm_hip.h"

namespace mdm {

// status codes: include/mdm_hip.h (MDM_OK, MDM_ERR_*)

#define MDM_RETURN_IF_LAUNCH_FAILED()                 \
  do {                                                \
    hipError_t e__ = hipGetLastError();               \
    if (e__ != hipSuccess) return (int)MDM_ERR_LAUNCH; \
  } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only, so the "already set" state of a launcher
// is kept per device ordinal: a process that drives a second GPU sets the attribute there too.  DevOnce replaces a
// `static bool`, DevInt a `static int` (largest size set so far) in the launchers; both read hipGetDevice() per use.
inline int dev_ordinal() {
  int d = 0;
  return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64) ? d : 0;
}
struct DevOnce {
  unsigned long long mask = 0;
  bool operator!() const { return !((mask >> dev_ordinal()) & 1ull); }
  DevOnce& operator=(bool v) {
    if (v) mask |= 1ull << dev_ordinal();
    return *this;
  }
};
struct DevInt {
  int v[64] = {};
  operator int() const { return v[dev_ordinal()]; }
  DevInt& operator=(int x) {
    v[dev_ordinal()] = x;
    return *this;
  }
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));  // MFMA A/B fragment: 8 bf16 in 4 VGPRs
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

// round-to-nearest-even f32 -> bf16 pair (lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  f32x2 v = {a, b};
  bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float bf16_lo_f32(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16_hi_f32(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

// split (a,b) into bf16 "hi" and bf16 "lo" = rn(x - hi): hi+lo carries ~16 mantissa bits
__device__ __forceinline__ void split_bf16(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = pack_bf16(a, b);
  lo = pack_bf16(a - bf16_lo_f32(hi), b - bf16_hi_f32(hi));
}

// MDM_OP_X2_ROW rows (include/mdm_hip.h): 16-bit index of the "hi" element of column k in a row (its "lo" sits 32 elements behind);
// four consecutive columns starting at a multiple of 4 are 8 contiguous bytes in either half
__device__ __forceinline__ int x2_col(int k) { return ((k >> 5) << 6) + (k & 31); }
__device__ __forceinline__ void store_x2_4(uint16_t* row, int k, float a, float b, float c, float d) {
  uint32_t h0, h1, l0, l1;
  split_bf16(a, b, h0, l0);
  split_bf16(c, d, h1, l1);
  uint16_t* p = row + x2_col(k);
  *(uint2*)p = make_uint2(h0, h1);
  *(uint2*)(p + 32) = make_uint2(l0, l1);
}

// The same when the lanes l and l ^ 1 hold the column groups k and k ^ 4 of ONE row (row kernels, GEMM epilogues): the pair trades
// halves through a quad-permute (a VALU move) so that the even lane stores the 8 hi values and the odd lane the 8 lo values of their
// 8 columns as ONE 16-byte store each (two 8-byte stores per lane move the same bytes in twice the store instructions).
__device__ __forceinline__ void store_x2_4p(uint16_t* row, int k, float a, float b, float c, float d) {
  uint32_t h0, h1, l0, l1;
  split_bf16(a, b, h0, l0);
  split_bf16(c, d, h1, l1);
  const bool odd = (k >> 2) & 1;
  const uint32_t s0 = odd ? h0 : l0, s1 = odd ? h1 : l1;  // what the partner needs: the even lane's lo, the odd lane's hi
  const uint32_t r0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
  const uint32_t r1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xf, 0xf, false);
  uint16_t* p = row + x2_col(k & ~7) + (odd ? 32 : 0);
  *(uint4*)p = odd ? make_uint4(r0, r1, l0, l1) : make_uint4(h0, h1, r0, r1);
}

// ---- the 16-bit operand format of the throughput kernels -----------------------------------------------------------
// Every single-pass MFMA kernel is templated on one of these two traits: bf16 (8 significand bits, fp32 range) or IEEE
// fp16 (11 significand bits, |x| <= 65504).  Both MFMA forms run at the same rate on gfx950 and both conversions are
// one packed instruction, so fp16 buys 8x smaller operand rounding for free wherever the value range is known (LayerNorm
// outputs, GELU hidden units, softmax weights, weights of trained Linears).  Host-side format codes: MDM_H16_*.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t pack_f16(float a, float b) {
  f32x2 v = {a, b};
  f16x2_t r = __builtin_convertvector(v, f16x2_t);  // v_cvt_pk_f16_f32 (round to nearest even)
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float f16_lo_f32(uint32_t p) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(p & 0xffffu)); }
__device__ __forceinline__ float f16_hi_f32(uint32_t p) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(p >> 16)); }
struct HB {  // bf16
  typedef bf16x8_t frag_t;
  static constexpr int FMT = 1;
  static __device__ __forceinline__ uint32_t pack(float a, float b) { return pack_bf16(a, b); }
  static __device__ __forceinline__ float lo(uint32_t p) { return bf16_lo_f32(p); }
  static __device__ __forceinline__ float hi(uint32_t p) { return bf16_hi_f32(p); }
  static __device__ __forceinline__ float one(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
  static __device__ __forceinline__ f32x4 mfma16(frag_t a, frag_t b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
struct HF {  // fp16
  typedef f16x8_t frag_t;
  static constexpr int FMT = 2;
  static __device__ __forceinline__ uint32_t pack(float a, float b) { return pack_f16(a, b); }
  static __device__ __forceinline__ float lo(uint32_t p) { return f16_lo_f32(p); }
  static __device__ __forceinline__ float hi(uint32_t p) { return f16_hi_f32(p); }
  static __device__ __forceinline__ float one(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
  static __device__ __forceinline__ f32x4 mfma16(frag_t a, frag_t b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
// run-time format (row-wise kernels, generic epilogues): fmt 1 = bf16, 2 = fp16
__device__ __forceinline__ uint32_t pack_h16(int fmt, float a, float b) { return fmt == 2 ? pack_f16(a, b) : pack_bf16(a, b); }
__device__ __forceinline__ float h16_lo_f32(int fmt, uint32_t p) { return fmt == 2 ? f16_lo_f32(p) : bf16_lo_f32(p); }
__device__ __forceinline__ float h16_hi_f32(int fmt, uint32_t p) { return fmt == 2 ? f16_hi_f32(p) : bf16_hi_f32(p); }

// erf(x) as the odd rational P(x^2) x / Q(x^2) on [-4, 4] (|erf| = 1 beyond to fp32 precision): max abs error
// 3.8e-7 against libm over [-6, 6] (checked in tests/test_host_logic.py), ~16 instructions, no branches.  The
// libm erff inlined 64x per thread dominated the GEMM epilogues.
__device__ __forceinline__ float erf_fast(float x) {
  x = fminf(fmaxf(x, -4.f), 4.f);
  const float x2 = x * x;
  float p = -2.72614225801306e-10f;
  p = fmaf(p, x2, 2.77068142495902e-08f);
  p = fmaf(p, x2, -2.10102402082508e-06f);
  p = fmaf(p, x2, -5.69250639462346e-05f);
  p = fmaf(p, x2, -7.34990630326855e-04f);
  p = fmaf(p, x2, -2.95459980854025e-03f);
  p = fmaf(p, x2, -1.60960333262415e-02f);
  float q = -1.45660718464996e-05f;
  q = fmaf(q, x2, -2.13374055278905e-04f);
  q = fmaf(q, x2, -1.68282697438203e-03f);
  q = fmaf(q, x2, -7.37332916720468e-03f);
  q = fmaf(q, x2, -1.42647390514189e-02f);
  return x * p * __builtin_amdgcn_rcpf(q);
}
// exact (erf) GELU, nn.GELU() default
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }
// Two GELUs at once on the packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32: two lanes of work per issue slot).
// Element for element the same operations as gelu_erf, so results are bit-identical; ~10 instructions per value instead
// of ~22.  The GELU epilogues (expert hidden layer, FFN, projections) evaluate 80 M of these per sampling step.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 v) {
  f32x2 x = v * 0.70710678118654752440f;
  x = __builtin_elementwise_min(__builtin_elementwise_max(x, (f32x2){-4.f, -4.f}), (f32x2){4.f, 4.f});
  const f32x2 x2 = x * x;
  f32x2 p = (f32x2){-2.72614225801306e-10f, -2.72614225801306e-10f};
  p = __builtin_elementwise_fma(p, x2, (f32x2){2.77068142495902e-08f, 2.77068142495902e-08f});
  p = __builtin_elementwise_fma(p, x2, (f32x2){-2.10102402082508e-06f, -2.10102402082508e-06f});
  p = __builtin_elementwise_fma(p, x2, (f32x2){-5.69250639462346e-05f, -5.69250639462346e-05f});
  p = __builtin_elementwise_fma(p, x2, (f32x2){-7.34990630326855e-04f, -7.34990630326855e-04f});
  p = __builtin_elementwise_fma(p, x2, (f32x2){-2.95459980854025e-03f, -2.95459980854025e-03f});
  p = __builtin_elementwise_fma(p, x2, (f32x2){-1.60960333262415e-02f, -1.60960333262415e-02f});
  f32x2 q = (f32x2){-1.45660718464996e-05f, -1.45660718464996e-05f};
  q = __builtin_elementwise_fma(q, x2, (f32x2){-2.13374055278905e-04f, -2.13374055278905e-04f});
  q = __builtin_elementwise_fma(q, x2, (f32x2){-1.68282697438203e-03f, -1.68282697438203e-03f});
  q = __builtin_elementwise_fma(q, x2, (f32x2){-7.37332916720468e-03f, -7.37332916720468e-03f});
  q = __builtin_elementwise_fma(q, x2, (f32x2){-1.42647390514189e-02f, -1.42647390514189e-02f});
  const f32x2 r = {__builtin_amdgcn_rcpf(q[0]), __builtin_amdgcn_rcpf(q[1])};
  const f32x2 e = x * p * r;
  return 0.5f * v * (1.0f + e);
}
// GELU as x * sigmoid(q(x)), q(x) = x (c0 + c1 x^2 + c2 x^4 + c3 x^6 + c4 x^8) fitted to the exact erf form on |x| <= 6.5;
// the coefficients carry the factor -log2(e) so that the hardware exp2 takes them directly.  Beyond |x| = 6.5 the sigmoid's
// ARGUMENT is clamped (there q = -+63.3 in exp2 units: the sigmoid is 1 - 8.7e-20 / 8.7e-20) while the multiplier stays the
// unclamped x, so a large negative x gives x * 8.7e-20 instead of 0: an absolute error that grows linearly with |x| and stays
// below the fit's own error for every |x| <= 1e13 (8.7e-8 at 1e12); x = -inf gives -inf where the erf form gives NaN.
// |error| <= 3.5e-6 absolute against the erf form over |x| <= 1e12 (tests/test_host_logic.py checks [-12, 12] densely, +-1e4 and the decades up to 1e12), 70x
// below the fp16 rounding of the hidden unit it produces, so it is used only where the result is rounded to 16 bits right
// after (the fused expert MLP).  14 instructions per PAIR of values (4 packed FMAs, 2 exp2, 2 rcp) against 24 for gelu_erf2.
// (The degree-7 fit, 1.2e-5, is a SYSTEMATIC error: summed over 1024 hidden units it showed as 1.4e-4 of the block output.)
__device__ __forceinline__ f32x2 gelu_sig2(f32x2 v) {
  const f32x2 xc = {__builtin_amdgcn_fmed3f(v[0], -6.5f, 6.5f), __builtin_amdgcn_fmed3f(v[1], -6.5f, 6.5f)};
  const f32x2 x2 = xc * xc;
  f32x2 p = __builtin_elementwise_fma(x2, (f32x2){-3.229054982512025e-06f, -3.229054982512025e-06f},
                                      (f32x2){8.82392268977128e-05f, 8.82392268977128e-05f});
  p = __builtin_elementwise_fma(p, x2, (f32x2){0.00036026936140842736f, 0.00036026936140842736f});
  p = __builtin_elementwise_fma(p, x2, (f32x2){-0.10522668063640594f, -0.10522668063640594f});
  p = __builtin_elementwise_fma(p, x2, (f32x2){-2.3020453453063965f, -2.3020453453063965f});
  const f32x2 t = p * xc;
  const f32x2 d = (f32x2){__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + 1.0f;
  const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  return v * r;
}
// exp on the hardware exp2 unit (v_exp_f32): ~1e-6 relative error for |x| <= 15, 2 instructions instead of ~25.
// Used only where the result is rounded to bf16 right after (throughput-mode attention cores).
__device__ __forceinline__ float exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + __expf(-x)); }

// Cross-lane reductions.  Inside a row of 16 lanes they use DPP modifiers (quad_perm, row_half_mirror, row_mirror: plain
// VALU instructions); __shfl_xor compiles to ds_bpermute_b32, an LDS-crossbar round trip of ~100+ cycles per step, and the
// row-wise kernels (LayerNorm chains, stylization inputs, router) are chains of such reductions.  Rows are combined with
// v_readlane (whole wave: the row totals are wave-uniform after the DPP steps) or one __shfl_xor per doubling (sub-wave
// groups of 32 / 64 lanes).
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
struct OpSum {
  static __device__ __forceinline__ float f(float a, float b) { return a + b; }
};
struct OpMax {
  static __device__ __forceinline__ float f(float a, float b) { return fmaxf(a, b); }
};
// reduce over aligned groups of W <= 16 lanes (every lane of the group gets the result)
template <int W, typename Op>
__device__ __forceinline__ float row_reduce(float v) {
  if constexpr (W >= 2) v = Op::f(v, dpp_move<0xB1>(v));   // quad_perm [1,0,3,2]
  if constexpr (W >= 4) v = Op::f(v, dpp_move<0x4E>(v));   // quad_perm [2,3,0,1]
  if constexpr (W >= 8) v = Op::f(v, dpp_move<0x141>(v));  // row_half_mirror: lane i <- 7 - i within each 8
  if constexpr (W >= 16) v = Op::f(v, dpp_move<0x140>(v)); // row_mirror: lane i <- 15 - i within each 16
  return v;
}
template <typename Op>
__device__ __forceinline__ float wave_reduce(float v) {
  v = row_reduce<16, Op>(v);
  const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return Op::f(Op::f(a, b), Op::f(c, d));
}
__device__ __forceinline__ float wave_sum(float v) { return wave_reduce<OpSum>(v); }
__device__ __forceinline__ float wave_max(float v) { return wave_reduce<OpMax>(v); }
// reductions over aligned sub-groups of W lanes (W power of two <= 64)
template <int W, typename Op>
__device__ __forceinline__ float group_reduce(float v) {
  if constexpr (W == 64) {
    return wave_reduce<Op>(v);
  } else {
    v = row_reduce<(W < 16 ? W : 16), Op>(v);
    if constexpr (W == 32) v = Op::f(v, __shfl_xor(v, 16, 64));
    return v;
  }
}
template <int W>
__device__ __forceinline__ float group_sum(float v) {
  return group_reduce<W, OpSum>(v);
}
template <int W>
__device__ __forceinline__ float group_max(float v) {
  return group_reduce<W, OpMax>(v);
}

// Bijective XCD-aware remap: consecutive logical tiles land on the same XCD (8 XCDs, round-robin dispatch).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

}  // namespace mdm
