// Text cross-attention cores of the fp32-grade (bf16x3) modes, head_dim 128: one launch each instead of the chains of batched
// contractions and row-wise kernels of the precision-3 path.  One workgroup (8 waves) per (batch, head); every product is three
// MFMAs on bf16 hi / lo splits (hi*lo + lo*hi + hi*hi, fp32 accumulate) like every other GEMM of these modes.
//
//   lin_xattn3: y[t] = softmax_dh(q[t]) A[b, h]                         (LinearTemporalCrossAttention, fast_attention.py:248-253)
//     The softmax over head_dim is the query projection's epilogue (csrc/gemm3.hip ACT_HEADSOFTMAX: hi / lo planes).  A^T[b, h]
//     (128 x 128 fp32, text cache) is split once into hi / lo LDS images; the frame tiles take q fragments global -> registers.
//   sd_attn3:   o[t] = softmax_n(q[t] . k[n]) v[n], n < ntok[b] <= N <= 128    (MemoryEfficientCrossAttentionBlock, :305-325)
//     q = the query projection as hi / lo planes (scaled by head_dim^-1/2 in its epilogue); the sample's key / value rows of this
//     head (fp32, text cache) are split once into LDS: K rows as they are, V transposed ([d][n], n in the k-slot order of the
//     second MFMA).  scores^T[n][t] come out with the frame on the lane, so the softmax over n is in registers (+ two shuffles)
//     and the probability accumulators ARE the B operand of o^T[d][t] = V^T[d][n] P[n][t].
// Both walk 16-frame tiles (wave w: tiles w, w + 8, ...), prefetch the next tile's fragments behind the current tile's MFMAs,
// and write fp32 rows.  LDS images: 256-byte rows of 16-bit elements, 16-byte chunk c of row r at slot c ^ (r & 15).
#include "kernels.h"

namespace mdm {
namespace {

typedef bf16x8_t bfx;
constexpr int X3_DH = 128, X3_NT = 512, X3_IMG = 128 * 256;

__device__ __forceinline__ void x3_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ f32x4 xm(bfx a, bfx b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 xm3(bfx ah, bfx al, bfx bh, bfx bl, f32x4 c) {
  c = xm(ah, bl, c);
  c = xm(al, bh, c);
  return xm(ah, bh, c);
}
__device__ __forceinline__ float x3_quad_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float x3_quad_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
typedef uint32_t u32x4x __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void x3_split8(const f32x4& a, const f32x4& b, bfx& hi, bfx& lo) {
  uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
  split_bf16(a[0], a[1], h0, l0);
  split_bf16(a[2], a[3], h1, l1);
  split_bf16(b[0], b[1], h2, l2);
  split_bf16(b[2], b[3], h3, l3);
  const u32x4x h = {h0, h1, h2, h3}, l = {l0, l1, l2, l3};
  hi = __builtin_bit_cast(bfx, h);
  lo = __builtin_bit_cast(bfx, l);
}
// fp32 [rows x 128] (row stride ld) -> hi / lo row images; rows >= nrows are zero
__device__ __forceinline__ void fill_rows(const float* __restrict__ src, int64_t ld, int nrows, int rows_pad, uint8_t* hi, uint8_t* lo,
                                          int tid) {
  for (int id = tid; id < rows_pad * 32; id += X3_NT) {  // 32 float4 per row
    const int row = id >> 5, c4 = id & 31;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < nrows) v = *(const f32x4*)(src + (int64_t)row * ld + 4 * c4);
    uint32_t h0, h1, l0, l1;
    split_bf16(v[0], v[1], h0, l0);
    split_bf16(v[2], v[3], h1, l1);
    const int off = row * 256 + (((c4 >> 1) ^ (row & 15)) << 4) + (c4 & 1) * 8;
    *(uint2*)(hi + off) = make_uint2(h0, h1);
    *(uint2*)(lo + off) = make_uint2(l0, l1);
  }
}

struct LinX3Args {
  const uint16_t* qh;  // softmax_dh(query) hi plane [B S, D]
  const uint16_t* ql;
  const float* at;     // A^T [B, H, 128 (l), 128 (d)] fp32 of this layer
  int S, H;
  float* out;          // fp32 [B S, D]
};

__global__ __launch_bounds__(X3_NT) void lin_xattn3_kernel(const LinX3Args g) {
  extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
  uint8_t* const ath = smem;
  uint8_t* const atl = smem + X3_IMG;
  const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, q = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / g.H, h = blockIdx.x - b * g.H;
  const int S = g.S, D = g.H * X3_DH, ntile = (S + 15) >> 4;
  const int64_t rowbase = (int64_t)b * S;
  bfx qh[4], ql[4];
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    const int t = tile * 16 + r16;
    const int64_t ro = (rowbase + (t < S ? t : S - 1)) * D + h * X3_DH + 8 * q;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qh[ks] = *(const bfx*)(g.qh + ro + 32 * ks), ql[ks] = *(const bfx*)(g.ql + ro + 32 * ks);
  };
  fetch(wid < ntile ? wid : ntile - 1);
  fill_rows(g.at + (int64_t)blockIdx.x * X3_DH * X3_DH, X3_DH, X3_DH, X3_DH, ath, atl, tid);
  x3_barrier();
#pragma unroll 1
  for (int tile = wid; tile < ntile; tile += 8) {
    const int t = tile * 16 + r16;
    int ro16 = r16 * 256, rx = r16;  // opaque per tile: the image reads must not be hoisted out of the loop (256 registers)
    asm volatile("" : "+v"(ro16), "+v"(rx));
    f32x4 acc[8];
#pragma unroll
    for (int lt = 0; lt < 8; ++lt) acc[lt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int lt = 0; lt < 8; ++lt) {
        const int off = 4096 * lt + ro16 + (((4 * ks + q) ^ rx) << 4);
        const bfx ah = *(const bfx*)(ath + off), al = *(const bfx*)(atl + off);
        acc[lt] = xm3(ah, al, qh[ks], ql[ks], acc[lt]);  // D[l][t]: t = r16, l = 16 lt + 4 q + reg      (:253)
      }
    if (tile + 8 < ntile) fetch(tile + 8);
    if (t < S) {
      float* orow = g.out + (rowbase + t) * D + h * X3_DH;
#pragma unroll
      for (int lt = 0; lt < 8; ++lt) *(f32x4*)(orow + 16 * lt + 4 * q) = acc[lt];
    }
  }
}

struct SdX3Args {
  const uint16_t* qh;  // query hi plane [B S, D], already scaled by head_dim^-1/2
  const uint16_t* ql;
  const float* kc;     // keys   [B, N, D] fp32 of this layer (text cache)
  const float* vc;     // values [B, N, D]
  const int32_t* ntok; // optional per-sample token counts
  int N, S, H;
  float* out;          // fp32 [B S, D], or (out_x2) the same rows pre-split (MDM_OP_X2_ROW) for the output projection
  int out_x2;
};

template <int NS>  // k steps of 32 keys: ceil(N / 32), compiled in (with a run-time bound hipcc spills the score accumulators)
__global__ __launch_bounds__(X3_NT) void sd_attn3_kernel(const SdX3Args g) {
  extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
  uint8_t* const kh = smem;               // K rows [n][128 d]
  uint8_t* const kl = smem + X3_IMG;
  uint8_t* const vh = smem + 2 * X3_IMG;  // V^T [d][128 n positions]
  uint8_t* const vl = smem + 3 * X3_IMG;
  const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, q = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / g.H, h = blockIdx.x - b * g.H;
  const int S = g.S, N = g.N, D = g.H * X3_DH, ntile = (S + 15) >> 4;
  constexpr int NP = 32 * NS, NT = 2 * NS;  // keys padded to whole k steps of the second product; 16-key tiles
  const int nv = g.ntok ? min(max(g.ntok[b], 1), N) : N;  // this sample attends to its first nv text tokens (as row_softmax)
  const int64_t rowbase = (int64_t)b * S;
  bfx qh[4], ql[4];
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    const int t = tile * 16 + r16;
    const int64_t ro = (rowbase + (t < S ? t : S - 1)) * D + h * X3_DH + 8 * q;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qh[ks] = *(const bfx*)(g.qh + ro + 32 * ks), ql[ks] = *(const bfx*)(g.ql + ro + 32 * ks);
  };
  fetch(wid < ntile ? wid : ntile - 1);
  fill_rows(g.kc + ((int64_t)b * N) * D + h * X3_DH, D, N, NP, kh, kl, tid);
  // V^T: element (n, d) at row d, position 32 (n >> 5) + 8 ((n & 15) >> 2) + 4 ((n >> 4) & 1) + (n & 3): the 8 k slots a lane of
  // the second product multiplies are one 16-byte read; positions of n >= N are zero
  for (int id = tid; id < NP * 32; id += X3_NT) {
    const int n = id >> 5, c4 = id & 31;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (n < N) v = *(const f32x4*)(g.vc + ((int64_t)b * N + n) * D + h * X3_DH + 4 * c4);
    const int pos = 32 * (n >> 5) + 8 * ((n & 15) >> 2) + 4 * ((n >> 4) & 1) + (n & 3);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int d = 4 * c4 + r;
      uint32_t hh, ll;
      split_bf16(v[r], 0.f, hh, ll);
      const int off = d * 256 + (((pos >> 3) ^ (d & 15)) << 4) + (pos & 7) * 2;
      *(uint16_t*)(vh + off) = (uint16_t)(hh & 0xffffu);
      *(uint16_t*)(vl + off) = (uint16_t)(ll & 0xffffu);
    }
  }
  x3_barrier();
#pragma unroll 1
  for (int tile = wid; tile < ntile; tile += 8) {
    const int t = tile * 16 + r16;
    int ro16 = r16 * 256, rx = r16;
    asm volatile("" : "+v"(ro16), "+v"(rx));
    // scores^T[n][t] = k[n] . q[t]   (:317-319; the scale is in q)
    f32x4 sc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) sc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int off = 4096 * nt + ro16 + (((4 * ks + q) ^ rx) << 4);
        const bfx ah = *(const bfx*)(kh + off), al = *(const bfx*)(kl + off);
        sc[nt] = xm3(ah, al, qh[ks], ql[ks], sc[nt]);  // D[n][t]: t = r16, n = 16 nt + 4 q + reg
      }
    if (tile + 8 < ntile) fetch(tile + 8);
    // softmax over the sample's nv keys (:320): arithmetic as rowwise.hip row_softmax_kernel
    float mx = -INFINITY;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool in = 16 * nt + 4 * q + r < nv;
        sc[nt][r] = in ? sc[nt][r] : -INFINITY;
        mx = fmaxf(mx, sc[nt][r]);
      }
    mx = x3_quad_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = (16 * nt + 4 * q + r < nv) ? expf(sc[nt][r] - mx) : 0.f;
        sc[nt][r] = e;
        sum += e;
      }
    sum = x3_quad_sum(sum);
    f32x4 o[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        f32x4 p0, p1;
#pragma unroll
        for (int r = 0; r < 4; ++r) p0[r] = sc[2 * s][r] / sum, p1[r] = sc[2 * s + 1][r] / sum;
        bfx ph, pl;  // k slots 0..3 <-> n = 32 s + 4 q + j, 4..7 <-> n = 32 s + 16 + 4 q + (j - 4)
        x3_split8(p0, p1, ph, pl);
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
          const int off = 4096 * dt + ro16 + (((4 * s + q) ^ rx) << 4);
          const bfx ah = *(const bfx*)(vh + off), al = *(const bfx*)(vl + off);
          o[dt] = xm3(ah, al, ph, pl, o[dt]);  // D[d][t]: t = r16, d = 16 dt + 4 q + reg      (:321)
        }
      }
    if (t < S) {
      float* orow = g.out + (rowbase + t) * D + h * X3_DH;
      uint16_t* xrow = (uint16_t*)g.out + (rowbase + t) * 2 * D;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        if (g.out_x2) {
          store_x2_4(xrow, h * X3_DH + 16 * dt + 4 * q, o[dt][0], o[dt][1], o[dt][2], o[dt][3]);
        } else {
          *(f32x4*)(orow + 16 * dt + 4 * q) = o[dt];
        }
      }
    }
  }
}

}  // namespace

bool xattn3_supported(int dh, int N) { return dh == X3_DH && N >= 1 && N <= 128; }

int lin_xattn3(const uint16_t* qh, const uint16_t* ql, const float* at, int B, int S, int H, int dh, float* out, hipStream_t s) {
  if (dh != X3_DH) return MDM_ERR_UNSUPPORTED;
  if (!qh || !ql || !at || !out || B <= 0 || S <= 0 || H <= 0 ||
      ((((uintptr_t)qh) | ((uintptr_t)ql) | ((uintptr_t)at) | ((uintptr_t)out)) & 15))
    return MDM_ERR_ARG;
  constexpr int smem = 2 * X3_IMG;
  static DevOnce attr;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)lin_xattn3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return MDM_ERR_LAUNCH;
    attr = true;
  }
  const LinX3Args g = {qh, ql, at, S, H, out};
  hipLaunchKernelGGL(lin_xattn3_kernel, dim3(B * H), dim3(X3_NT), smem, s, g);
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

int sd_attn3(const uint16_t* qh, const uint16_t* ql, const float* kc, const float* vc, const int32_t* ntok, int B, int S, int H,
             int dh, int N, float* out, int out_x2, hipStream_t s) {
  if (!xattn3_supported(dh, N)) return MDM_ERR_UNSUPPORTED;
  if (!qh || !ql || !kc || !vc || !out || B <= 0 || S <= 0 || H <= 0 ||
      ((((uintptr_t)qh) | ((uintptr_t)ql) | ((uintptr_t)kc) | ((uintptr_t)vc) | ((uintptr_t)out)) & 15))
    return MDM_ERR_ARG;
  constexpr int smem = 4 * X3_IMG;
  static DevOnce attr;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)sd_attn3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess ||
        hipFuncSetAttribute((const void*)sd_attn3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess ||
        hipFuncSetAttribute((const void*)sd_attn3_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess ||
        hipFuncSetAttribute((const void*)sd_attn3_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return MDM_ERR_LAUNCH;
    attr = true;
  }
  const SdX3Args g = {qh, ql, kc, vc, ntok, N, S, H, out, out_x2};
  switch ((N + 31) >> 5) {
    case 1: hipLaunchKernelGGL(sd_attn3_kernel<1>, dim3(B * H), dim3(X3_NT), smem, s, g); break;
    case 2: hipLaunchKernelGGL(sd_attn3_kernel<2>, dim3(B * H), dim3(X3_NT), smem, s, g); break;
    case 3: hipLaunchKernelGGL(sd_attn3_kernel<3>, dim3(B * H), dim3(X3_NT), smem, s, g); break;
    default: hipLaunchKernelGGL(sd_attn3_kernel<4>, dim3(B * H), dim3(X3_NT), smem, s, g); break;
  }
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

}  // namespace mdm
