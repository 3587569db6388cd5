// Fused text cross-attention cores for head_dim = 128 (throughput mode), one workgroup per (batch, head):
//   softmax cross attention (fast_attention.py:305-325): o = softmax_n(q k^T) v over N <= 96 text tokens
//   linear cross attention  (fast_attention.py:248,253):  y = softmax_dh(q) A          with A^T[b,h] cached per text
// Same register choreography as perf_attn.hip: rows of q are loaded straight into MFMA fragments (lane = row,
// 8 consecutive k per lane group), products are taken with the operands swapped so that each lane ends up with
// 4 consecutive output features of ONE row; the score accumulator is reused as the B operand of the value product.
#include "kernels.h"
#include "proj_phase.h"

namespace mdm {
namespace {

constexpr int DH = 128, PS = 136, NP = 96, NS = NP + 8;  // NS: row stride (elements) of the v^T image
constexpr int XNT = 512;  // threads per (batch, head) workgroup: 8 waves share the <= 14 row tiles


template <typename HT>
__device__ __forceinline__ typename HT::frag_t make_frag(const float* x) {
  typedef typename HT::frag_t frag_t;
  u32x4 u = {HT::pack(x[0], x[1]), HT::pack(x[2], x[3]), HT::pack(x[4], x[5]), HT::pack(x[6], x[7])};
  return __builtin_bit_cast(frag_t, u);
}
__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}
// row (t0 + lane&15) of a (M, D) fp32 / bf16 matrix, head h: x[32] at k = 32*ks + 8*q + j
template <typename HT, bool IN16, int DHT = DH>
__device__ __forceinline__ void load_row(const void* __restrict__ base, int64_t row, int D, int h, int q, float (&x)[DHT / 4]) {
  if constexpr (IN16) {
    const uint16_t* p = (const uint16_t*)base + row * D + h * DHT + 8 * q;
#pragma unroll
    for (int ks = 0; ks < DHT / 32; ++ks) {
      const uint4 u = *(const uint4*)(p + 32 * ks);
      x[8 * ks + 0] = HT::lo(u.x), x[8 * ks + 1] = HT::hi(u.x);
      x[8 * ks + 2] = HT::lo(u.y), x[8 * ks + 3] = HT::hi(u.y);
      x[8 * ks + 4] = HT::lo(u.z), x[8 * ks + 5] = HT::hi(u.z);
      x[8 * ks + 6] = HT::lo(u.w), x[8 * ks + 7] = HT::hi(u.w);
    }
  } else {
    const float* p = (const float*)base + row * D + h * DHT + 8 * q;
#pragma unroll
    for (int ks = 0; ks < DHT / 32; ++ks) {
      const f32x4 a = *(const f32x4*)(p + 32 * ks), c = *(const f32x4*)(p + 32 * ks + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) x[8 * ks + j] = a[j], x[8 * ks + 4 + j] = c[j];
    }
  }
}

template <typename HT, int NT32, bool IN16, int DHT>  // ceil(N / 32); q stored as 16-bit or fp32; head_dim 128 or 256
__global__ __launch_bounds__(XNT) void sd_attn_kernel(const void* __restrict__ qm, const float* __restrict__ kc,
                                                      const float* __restrict__ vc, int S, int H, int N,
                                                      uint16_t* __restrict__ out16, float* __restrict__ out32,
                                                      const int32_t* __restrict__ ntok) {
  typedef typename HT::frag_t frag_t;
  constexpr int DH = DHT, PS = DHT + 8;  // (shadow the file-level head_dim-128 constants)
  extern __shared__ __attribute__((aligned(16))) uint16_t sd_smem[];
  uint16_t* kL = sd_smem;                 // k [n][d], NT32 * 32 rows of PS
  uint16_t* vT = sd_smem + NT32 * 32 * PS;  // v^T [d][n], DH rows of NS
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r16 = lane & 15, q = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H, D = H * DH;
  const int nv = ntok ? min(max(ntok[b], 1), N) : N;  // this sample's own text tokens (the rest of its N rows is padding)
  for (int i = tid; i < NT32 * 32 * (DH / 4); i += XNT) {
    const int n = i / (DH / 4), c = i - n * (DH / 4);
    f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
      kv = *(const f32x4*)(kc + ((int64_t)(b * N + n)) * D + h * DH + 4 * c);
      vv = *(const f32x4*)(vc + ((int64_t)(b * N + n)) * D + h * DH + 4 * c);
    }
    *(uint2*)(kL + n * PS + 4 * c) = make_uint2(HT::pack(kv[0], kv[1]), HT::pack(kv[2], kv[3]));
#pragma unroll
    for (int j = 0; j < 4; ++j) vT[(4 * c + j) * NS + n] = (uint16_t)(HT::pack(vv[j], 0.f) & 0xffff);
  }
  __syncthreads();
  const int ntile = (S + 15) >> 4;
  for (int tile = wid; tile < ntile; tile += XNT / 64) {
    const int t = tile * 16 + r16, tc = t < S ? t : S - 1;
    float x[DH / 4];
    load_row<HT, IN16, DHT>(qm, (int64_t)b * S + tc, D, h, q, x);
    frag_t qf[DH / 32];
#pragma unroll
    for (int ks = 0; ks < DH / 32; ++ks) qf[ks] = make_frag<HT>(x + 8 * ks);
    f32x4 sc[2 * NT32];
#pragma unroll
    for (int nt = 0; nt < 2 * NT32; ++nt) sc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < DH / 32; ++ks)
#pragma unroll
      for (int nt = 0; nt < 2 * NT32; ++nt) {
        const frag_t kf = *(const frag_t*)(kL + (16 * nt + r16) * PS + 32 * ks + 8 * q);
        sc[nt] = HT::mfma16(kf, qf[ks], sc[nt]);  // D[n][t]
      }
    // lane: t, n = 16*nt + 4q + r.  softmax over n (no text mask in the reference, :317-320)
    float mx = -INFINITY;
#pragma unroll
    for (int nt = 0; nt < 2 * NT32; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (16 * nt + 4 * q + r >= nv) sc[nt][r] = -INFINITY;
        mx = fmaxf(mx, sc[nt][r]);
      }
    mx = quad_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int nt = 0; nt < 2 * NT32; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sc[nt][r] = exp_fast(sc[nt][r] - mx);
        sum += sc[nt][r];
      }
    const float inv = 1.f / quad_sum(sum);
    f32x4 o[DH / 16];
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NT32; ++s) {
      const u32x4 ub = {HT::pack(sc[2 * s][0] * inv, sc[2 * s][1] * inv), HT::pack(sc[2 * s][2] * inv, sc[2 * s][3] * inv),
                        HT::pack(sc[2 * s + 1][0] * inv, sc[2 * s + 1][1] * inv),
                        HT::pack(sc[2 * s + 1][2] * inv, sc[2 * s + 1][3] * inv)};
      const frag_t pf = __builtin_bit_cast(frag_t, ub);
#pragma unroll
      for (int dt = 0; dt < DH / 16; ++dt) {
        const uint2 lo = *(const uint2*)(vT + (16 * dt + r16) * NS + 32 * s + 4 * q);
        const uint2 hi = *(const uint2*)(vT + (16 * dt + r16) * NS + 32 * s + 16 + 4 * q);
        const u32x4 ua = {lo.x, lo.y, hi.x, hi.y};
        o[dt] = HT::mfma16(__builtin_bit_cast(frag_t, ua), pf, o[dt]);  // D[d][t]
        // keep hipcc from hoisting all NT32 * DH / 16 value-fragment reads (246 spilled VGPRs at head_dim 256, 6 at 128)
        if ((dt & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (t < S) {
      const int64_t off = ((int64_t)b * S + t) * D + h * DH;
#pragma unroll
      for (int dt = 0; dt < DH / 16; ++dt) {
        if (out16) *(uint2*)(out16 + off + 16 * dt + 4 * q) = make_uint2(HT::pack(o[dt][0], o[dt][1]), HT::pack(o[dt][2], o[dt][3]));
        if (out32) *(f32x4*)(out32 + off + 16 * dt + 4 * q) = o[dt];
      }
    }
  }
}

template <typename HT, bool IN16>
__global__ __launch_bounds__(XNT) void lin_xattn_kernel(const void* __restrict__ ql, const float* __restrict__ at, int S,
                                                        int H, float* __restrict__ out, uint16_t* __restrict__ out16) {
  typedef typename HT::frag_t frag_t;
  __shared__ __attribute__((aligned(16))) uint16_t aL[DH * PS];  // A^T [l][d]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r16 = lane & 15, q = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H, D = H * DH;
  const float* ab = at + (int64_t)blockIdx.x * DH * DH;
  for (int i = tid; i < DH * (DH / 4); i += XNT) {
    const int l = i / (DH / 4), c = i - l * (DH / 4);
    const f32x4 v = *(const f32x4*)(ab + l * DH + 4 * c);
    *(uint2*)((uint8_t*)aL + l * 256 + (((c >> 1) ^ (l & 15)) << 4) + (c & 1) * 8) = make_uint2(HT::pack(v[0], v[1]), HT::pack(v[2], v[3]));
  }
  __syncthreads();
  const int ntile = (S + 15) >> 4;
  for (int tile = wid; tile < ntile; tile += XNT / 64) {
    const int t = tile * 16 + r16, tc = t < S ? t : S - 1;
    float x[32];
    load_row<HT, IN16>(ql, (int64_t)b * S + tc, D, h, q, x);
    float mx = -INFINITY;  // softmax over head_dim (:248)
#pragma unroll
    for (int i = 0; i < 32; ++i) mx = fmaxf(mx, x[i]);
    mx = quad_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      x[i] = exp_fast(x[i] - mx);
      sum += x[i];
    }
    const float inv = 1.f / quad_sum(sum);
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] *= inv;
    frag_t qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = make_frag<HT>(x + 8 * ks);
    f32x4 y[8];
#pragma unroll
    for (int lt = 0; lt < 8; ++lt) y[lt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int lt = 0; lt < 8; ++lt) {
        const int ar = 16 * lt + r16;  // 256-B rows, 16-B chunk c at slot c ^ (row & 15): conflict-free ds_read_b128
        const frag_t af = *(const frag_t*)((const uint8_t*)aL + ar * 256 + (((4 * ks + q) ^ (ar & 15)) << 4));
        y[lt] = HT::mfma16(af, qf[ks], y[lt]);  // D[l][t]
      }
    if (t < S && out16) {
      uint16_t* orow = out16 + ((int64_t)b * S + t) * D + h * DH;
#pragma unroll
      for (int lt = 0; lt < 8; ++lt)
        *(uint2*)(orow + 16 * lt + 4 * q) = make_uint2(HT::pack(y[lt][0], y[lt][1]), HT::pack(y[lt][2], y[lt][3]));
    } else if (t < S) {
      float* orow = out + ((int64_t)b * S + t) * D + h * DH;
#pragma unroll
      for (int lt = 0; lt < 8; ++lt) *(f32x4*)(orow + 16 * lt + 4 * q) = y[lt];
    }
  }
}


// The linear cross attention with its query projection inside (16-bit modes, head_dim 128, D = 512): the workgroup of (batch,
// head) first multiplies its sample's normed rows with its head's 128 rows of the query weight (proj_phase.h: one column tile
// per wave, all MT row tiles), leaves q + b as 16-bit rows in LDS (the rounding point of the separate GEMM launch), and every wave
// then takes its own row tiles through softmax over head_dim and the product with A^T as above.  The [M, D] query tensor and its
// launch are gone.
template <typename HT, int MT>
__global__ __launch_bounds__(XNT) void lin_xattn_q_kernel(const uint16_t* __restrict__ xn, const uint16_t* __restrict__ wq, int ldw,
                                                          const float* __restrict__ bq, const float* __restrict__ at, int S, int H,
                                                          float* __restrict__ out, uint16_t* __restrict__ out16) {
  typedef typename HT::frag_t frag_t;
  constexpr int R = QkvGeo<MT>::R, NCH = QkvGeo<MT>::NCH;
  constexpr int WORK_B = 2 * R * 128 > R * 256 ? 2 * R * 128 : R * 256;  // two row stages, then the q image [R][128] (256-B rows)
  extern __shared__ __attribute__((aligned(1024))) uint8_t xq_smem[];
  uint8_t* const aL = xq_smem + WORK_B;  // A^T [l][d], 256-B rows
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r16 = lane & 15, q = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H, D = H * DH;
  const float* ab = at + (int64_t)blockIdx.x * DH * DH;
  for (int i = tid; i < DH * (DH / 4); i += XNT) {
    const int l = i / (DH / 4), c = i - l * (DH / 4);
    const f32x4 v = *(const f32x4*)(ab + l * DH + 4 * c);
    *(uint2*)(aL + l * 256 + (((c >> 1) ^ (l & 15)) << 4) + (c & 1) * 8) = make_uint2(HT::pack(v[0], v[1]), HT::pack(v[2], v[3]));
  }
  // ---- query rows of this head: [R x 128] = xn[b] . Wq_h^T --------------------------------------------------------------------
  {
    const uint16_t* xrow0 = xn + (int64_t)b * S * D;
    const uint16_t* wrow[1] = {wq + (int64_t)(h * DH + 16 * wid + r16) * ldw + 8 * q};
    frag_t wr[2];
    wr[0] = *(const frag_t*)(wrow[0]);
    wr[1] = *(const frag_t*)(wrow[0] + 32);
    f32x4 acc[MT][1];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 sa[NCH];
    qkv_fetch<NCH>(sa, tid, R, S, D, 0, xrow0);
#pragma unroll 1
    for (int kk = 0; kk < 8; ++kk) qkv_slice<HT, MT, 1>(kk, sa, acc, wr, wrow, xq_smem + (kk & 1) * (R * 128), tid, r16, q, S, D, xrow0);
    __syncthreads();  // the stages are dead: their LDS takes the q rows
    const int col = 16 * wid + 4 * q;
    const f32x4 bb = *(const f32x4*)(bq + h * DH + col);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = 16 * mt + r16;
      const f32x4 v = acc[mt][0];
      *(uint2*)(xq_smem + row * 256 + ((((col >> 3)) ^ (row & 15)) << 4) + ((col >> 2) & 1) * 8) =
          make_uint2(HT::pack(v[0] + bb[0], v[1] + bb[1]), HT::pack(v[2] + bb[2], v[3] + bb[3]));
    }
  }
  __syncthreads();
  const int ntile = (S + 15) >> 4;
  for (int tile = wid; tile < ntile; tile += XNT / 64) {
    const int row = tile * 16 + r16, t = row;
    float x[32];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {  // the row in the fragment layout: k = 32 ks + 8 q + j
      const uint4 u = *(const uint4*)(xq_smem + row * 256 + (((4 * ks + q) ^ (row & 15)) << 4));
      x[8 * ks + 0] = HT::lo(u.x), x[8 * ks + 1] = HT::hi(u.x);
      x[8 * ks + 2] = HT::lo(u.y), x[8 * ks + 3] = HT::hi(u.y);
      x[8 * ks + 4] = HT::lo(u.z), x[8 * ks + 5] = HT::hi(u.z);
      x[8 * ks + 6] = HT::lo(u.w), x[8 * ks + 7] = HT::hi(u.w);
    }
    float mx = -INFINITY;  // softmax over head_dim (:248)
#pragma unroll
    for (int i = 0; i < 32; ++i) mx = fmaxf(mx, x[i]);
    mx = quad_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      x[i] = exp_fast(x[i] - mx);
      sum += x[i];
    }
    const float inv = 1.f / quad_sum(sum);
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] *= inv;
    frag_t qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = make_frag<HT>(x + 8 * ks);
    f32x4 y[8];
#pragma unroll
    for (int lt = 0; lt < 8; ++lt) y[lt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int lt = 0; lt < 8; ++lt) {
        const int ar = 16 * lt + r16;
        const frag_t af = *(const frag_t*)(aL + ar * 256 + (((4 * ks + q) ^ (ar & 15)) << 4));
        y[lt] = HT::mfma16(af, qf[ks], y[lt]);  // D[l][t]
      }
    if (t < S && out16) {
      uint16_t* orow = out16 + ((int64_t)b * S + t) * D + h * DH;
#pragma unroll
      for (int lt = 0; lt < 8; ++lt)
        *(uint2*)(orow + 16 * lt + 4 * q) = make_uint2(HT::pack(y[lt][0], y[lt][1]), HT::pack(y[lt][2], y[lt][3]));
    } else if (t < S) {
      float* orow = out + ((int64_t)b * S + t) * D + h * DH;
#pragma unroll
      for (int lt = 0; lt < 8; ++lt) *(f32x4*)(orow + 16 * lt + 4 * q) = y[lt];
    }
  }
}

// the same at head_dim 256 (big model): A^T [256][264] fills 132 KiB of LDS (dynamic), 64 elements of q per lane
constexpr int DH2 = 256, PS2 = 264;
template <typename HT, bool IN16>
__global__ __launch_bounds__(XNT, 2) void lin_xattn256_kernel(const void* __restrict__ ql, const float* __restrict__ at, int S,
                                                              int H, float* __restrict__ out, uint16_t* __restrict__ out16) {
  typedef typename HT::frag_t frag_t;
  extern __shared__ __attribute__((aligned(16))) uint16_t aL2[];  // A^T [l][d], row stride PS2
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r16 = lane & 15, q = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H, D = H * DH2;
  const float* ab = at + (int64_t)blockIdx.x * DH2 * DH2;
  for (int i = tid; i < DH2 * (DH2 / 4); i += XNT) {
    const int l = i / (DH2 / 4), c = i - l * (DH2 / 4);
    const f32x4 v = *(const f32x4*)(ab + l * DH2 + 4 * c);
    *(uint2*)((uint8_t*)aL2 + l * 512 + (((c >> 1) ^ (l & 15)) << 4) + (c & 1) * 8) = make_uint2(HT::pack(v[0], v[1]), HT::pack(v[2], v[3]));
  }
  __syncthreads();
  const int ntile = (S + 15) >> 4;
  for (int tile = wid; tile < ntile; tile += XNT / 64) {
    const int t = tile * 16 + r16, tc = t < S ? t : S - 1;
    float x[64];  // k = 32 ks + 8 q + j
    if constexpr (IN16) {
      const uint16_t* p = (const uint16_t*)ql + ((int64_t)b * S + tc) * D + h * DH2 + 8 * q;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const uint4 u = *(const uint4*)(p + 32 * ks);
        x[8 * ks + 0] = HT::lo(u.x), x[8 * ks + 1] = HT::hi(u.x), x[8 * ks + 2] = HT::lo(u.y), x[8 * ks + 3] = HT::hi(u.y);
        x[8 * ks + 4] = HT::lo(u.z), x[8 * ks + 5] = HT::hi(u.z), x[8 * ks + 6] = HT::lo(u.w), x[8 * ks + 7] = HT::hi(u.w);
      }
    } else {
      const float* p = (const float*)ql + ((int64_t)b * S + tc) * D + h * DH2 + 8 * q;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const f32x4 a = *(const f32x4*)(p + 32 * ks), c = *(const f32x4*)(p + 32 * ks + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) x[8 * ks + j] = a[j], x[8 * ks + 4 + j] = c[j];
      }
    }
    float mx = -INFINITY;  // softmax over head_dim (:248)
#pragma unroll
    for (int i = 0; i < 64; ++i) mx = fmaxf(mx, x[i]);
    mx = quad_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      x[i] = exp_fast(x[i] - mx);
      sum += x[i];
    }
    const float inv = 1.f / quad_sum(sum);
    frag_t qf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[8 * ks + j] *= inv;
      qf[ks] = make_frag<HT>(x + 8 * ks);
    }
    f32x4 y[16];
#pragma unroll
    for (int lt = 0; lt < 16; ++lt) y[lt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int lt = 0; lt < 16; ++lt) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int ar = 16 * lt + r16;
        const frag_t af = *(const frag_t*)((const uint8_t*)aL2 + ar * 512 + (((4 * ks + q) ^ (ar & 15)) << 4));
        y[lt] = HT::mfma16(af, qf[ks], y[lt]);  // D[l][t]
      }
      if ((lt & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    if (t < S && out16) {
      uint16_t* orow = out16 + ((int64_t)b * S + t) * D + h * DH2;
#pragma unroll
      for (int lt = 0; lt < 16; ++lt)
        *(uint2*)(orow + 16 * lt + 4 * q) = make_uint2(HT::pack(y[lt][0], y[lt][1]), HT::pack(y[lt][2], y[lt][3]));
    } else if (t < S) {
      float* orow = out + ((int64_t)b * S + t) * D + h * DH2;
#pragma unroll
      for (int lt = 0; lt < 16; ++lt) *(f32x4*)(orow + 16 * lt + 4 * q) = y[lt];
    }
  }
}

template <typename HT, int NT32, bool IN16, int DHT>
int launch_sd(const void* q, const float* kc, const float* vc, int B, int S, int H, int N, uint16_t* out16, float* out32,
              hipStream_t s, const int32_t* ntok) {
  constexpr int smem = (NT32 * 32 * (DHT + 8) + DHT * NS) * 2;
  static DevOnce attr;
  if (smem > 65536 && !attr) {
    if (hipFuncSetAttribute((const void*)sd_attn_kernel<HT, NT32, IN16, DHT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return MDM_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL((sd_attn_kernel<HT, NT32, IN16, DHT>), dim3(B * H), dim3(XNT), smem, s, q, kc, vc, S, H, N, out16, out32, ntok);
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

}  // namespace

bool xattn_supported(int dh, int N) { return (dh == DH || dh == DH2) && N >= 1 && N <= NP; }
bool lin_xattn_supported(int dh) { return dh == DH || dh == DH2; }

// q_fmt: 0 = fp32 q rows, 1 / 2 = bf16 / fp16 rows; h16: operand format of the MFMAs and of out16 (MDM_H16_*)
int sd_attn(const void* q, int q_fmt, const float* kc, const float* vc, int B, int S, int H, int dh, int N, uint16_t* out16,
            float* out32, int h16, hipStream_t s, const int32_t* ntok) {
  const int q_bf16 = q_fmt != 0;
  if (q_fmt && q_fmt != h16) return MDM_ERR_ARG;
  if (!xattn_supported(dh, N)) return MDM_ERR_UNSUPPORTED;
  if (!q || !kc || !vc || (!out16 && !out32)) return MDM_ERR_ARG;
  const bool f16 = h16 == MDM_H16_F16;
#define MDM_SD(NT, DHT)                                                                                                     \
  (f16 ? (q_bf16 ? launch_sd<HF, NT, true, DHT>(q, kc, vc, B, S, H, N, out16, out32, s, ntok)                                     \
                 : launch_sd<HF, NT, false, DHT>(q, kc, vc, B, S, H, N, out16, out32, s, ntok))                                   \
       : (q_bf16 ? launch_sd<HB, NT, true, DHT>(q, kc, vc, B, S, H, N, out16, out32, s, ntok)                                     \
                 : launch_sd<HB, NT, false, DHT>(q, kc, vc, B, S, H, N, out16, out32, s, ntok)))
  if (dh == DH2) return N <= 32 ? MDM_SD(1, 256) : (N <= 64 ? MDM_SD(2, 256) : MDM_SD(3, 256));
  return N <= 32 ? MDM_SD(1, 128) : (N <= 64 ? MDM_SD(2, 128) : MDM_SD(3, 128));
#undef MDM_SD
}

int lin_xattn(const void* ql, int ql_fmt, const float* at, int B, int S, int H, int dh, float* out, uint16_t* out16,
              int h16, hipStream_t s) {
  if (dh != DH && dh != DH2) return MDM_ERR_UNSUPPORTED;
  if (!ql || !at || (!out && !out16) || (ql_fmt && ql_fmt != h16)) return MDM_ERR_ARG;
  const dim3 grid(B * H), block(XNT);
  if (dh == DH2) {
    constexpr int smem = DH2 * PS2 * 2;
    static DevOnce attr;
    if (!attr) {
      if (hipFuncSetAttribute((const void*)lin_xattn256_kernel<HF, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess ||
          hipFuncSetAttribute((const void*)lin_xattn256_kernel<HF, false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess ||
          hipFuncSetAttribute((const void*)lin_xattn256_kernel<HB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess ||
          hipFuncSetAttribute((const void*)lin_xattn256_kernel<HB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
        return MDM_ERR_LAUNCH;
      attr = true;
    }
    if (h16 == MDM_H16_F16) {
      if (ql_fmt) {
        hipLaunchKernelGGL((lin_xattn256_kernel<HF, true>), grid, block, smem, s, ql, at, S, H, out, out16);
      } else {
        hipLaunchKernelGGL((lin_xattn256_kernel<HF, false>), grid, block, smem, s, ql, at, S, H, out, out16);
      }
    } else if (ql_fmt) {
      hipLaunchKernelGGL((lin_xattn256_kernel<HB, true>), grid, block, smem, s, ql, at, S, H, out, out16);
    } else {
      hipLaunchKernelGGL((lin_xattn256_kernel<HB, false>), grid, block, smem, s, ql, at, S, H, out, out16);
    }
    MDM_RETURN_IF_LAUNCH_FAILED();
    return MDM_OK;
  }
  if (h16 == MDM_H16_F16) {
    if (ql_fmt) {
      hipLaunchKernelGGL((lin_xattn_kernel<HF, true>), grid, block, 0, s, ql, at, S, H, out, out16);
    } else {
      hipLaunchKernelGGL((lin_xattn_kernel<HF, false>), grid, block, 0, s, ql, at, S, H, out, out16);
    }
  } else if (ql_fmt) {
    hipLaunchKernelGGL((lin_xattn_kernel<HB, true>), grid, block, 0, s, ql, at, S, H, out, out16);
  } else {
    hipLaunchKernelGGL((lin_xattn_kernel<HB, false>), grid, block, 0, s, ql, at, S, H, out, out16);
  }
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

// the same with the query projection inside (16-bit modes, head_dim 128, H = 4): xn 16-bit [B S, D] (format h16), wq the 16-bit plane
// [D][ldw] of the query weight, bq [D]
bool lin_xattn_q_supported(int dh, int S, int H) { return dh == DH && H == 4 && S >= 1 && S <= 208; }

namespace {
template <int MT>
int launch_lin_xattn_q(const uint16_t* xn, const uint16_t* wq, int ldw, const float* bq, const float* at, int B, int S, int H, float* out,
                       uint16_t* out16, int h16, hipStream_t s) {
  constexpr int R = QkvGeo<MT>::R;
  constexpr int smem = (2 * R * 128 > R * 256 ? 2 * R * 128 : R * 256) + DH * 256;
  static DevOnce attr;
  if (smem > 65536 && !attr) {
    if (hipFuncSetAttribute((const void*)lin_xattn_q_kernel<HF, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess ||
        hipFuncSetAttribute((const void*)lin_xattn_q_kernel<HB, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return MDM_ERR_LAUNCH;
    attr = true;
  }
  if (h16 == MDM_H16_F16) {
    hipLaunchKernelGGL((lin_xattn_q_kernel<HF, MT>), dim3(B * H), dim3(XNT), smem, s, xn, wq, ldw, bq, at, S, H, out, out16);
  } else {
    hipLaunchKernelGGL((lin_xattn_q_kernel<HB, MT>), dim3(B * H), dim3(XNT), smem, s, xn, wq, ldw, bq, at, S, H, out, out16);
  }
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}
}  // namespace

int lin_xattn_q(const uint16_t* xn, const uint16_t* wq, int ldw, const float* bq, const float* at, int B, int S, int H, int dh, float* out,
                uint16_t* out16, int h16, hipStream_t s) {
  if (!lin_xattn_q_supported(dh, S, H) || (h16 != MDM_H16_BF16 && h16 != MDM_H16_F16)) return MDM_ERR_UNSUPPORTED;
  if (!xn || !wq || !bq || !at || (!out && !out16) || (ldw & 7) || ((((uintptr_t)xn) | ((uintptr_t)wq)) & 15)) return MDM_ERR_ARG;
  if (S <= 112) return launch_lin_xattn_q<7>(xn, wq, ldw, bq, at, B, S, H, out, out16, h16, s);
  return launch_lin_xattn_q<13>(xn, wq, ldw, bq, at, B, S, H, out, out16, h16, s);
}

}  // namespace mdm
