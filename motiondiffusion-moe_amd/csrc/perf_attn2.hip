// Fused Performer-style linear attention core (fast_attention.py:29-92) for head_dim = 256 (the BIG model: latent 1024,
// 4 heads, m = min(head_dim, 256) = 256 features), throughput modes.
//
// At head_dim 128 (perf_attn.hip) the projection P^T, the feature image kphi^T, v^T and the KV state all fit the 160 KiB of
// LDS at once and one workgroup per (batch, head) does everything.  At 256 each of them alone is 112-135 KB, and a single
// workgroup per (batch, head) that pages them through LDS recomputes the LayerNorms / feature maps per page and is one long
// latency chain (a first version took 239 us per launch, 4 % of the MFMA peak).  The work is therefore split where the
// data dependencies are, into two launches that keep ONE big operand resident each:
//   perf_feat256_kernel   P^T [256 m][256 k] resident in LDS, persistent over (batch, head, 16-token tile) units, one unit
//                         per wave: q, k rows -> LN(dh) -> L2 norm -> MFMA with P^T -> 0.1 exp(clamp) -> key mask.  Every P^T
//                         fragment read feeds three MFMAs: kphi in the D[t][m] arrangement (written TRANSPOSED, kphi^T [m][t],
//                         the K-contiguous operand of the KV product), qphi in the D[m][t] arrangement (written row-major,
//                         the K-contiguous operand of the num product) and kphi again as D[m][t] for the same-t denominator
//                         <qphi[t], kphi[t]> (:81), reduced in fp32 registers and written per token.
//   perf_kvnum256_kernel  one workgroup per (batch, head): v rows -> LN -> LDS, row-major [t][d]; KV^T = 0.1 sum_t kphi^T v
//                         with kphi^T fragments straight from L2 and the v operand through the transposing LDS read
//                         (ds_read_b64_tr_b16, conflict-free XOR image); the 256 x 256 state goes registers -> LDS (over the
//                         dead v image) and never to memory; num = qphi KV with qphi fragments from L2, then
//                         out = LN_dh(0.1 num / max(den, 1e-6)) as 16-bit rows.
// Scratch between the launches (L2 / MALL resident): qphi [BH][S][256], kphi^T [BH][256][TP] (16-bit), den [BH][S] (fp32).
#include "kernels.h"

namespace mdm {
namespace {

constexpr int DH2 = 256, PS2 = 264, NW2 = 8, NTH2 = 64 * NW2;
constexpr int R_BIG = DH2 * PS2 * 2;             // 135168 >= P^T / KV^T images [256][256] with 512-B rows whose 16-B chunk c sits at
                                                 // slot c ^ (row & 15): conflict-free for the measured ds_read_b128 lane groups (a row
                                                 // padded by 16 B, 264 elements, costs one 2-way collision per 16-lane group = 2x cycles)
constexpr int SMEM_PA2 = R_BIG + 2 * DH2 * 4;    // + LN gain | bias

typedef uint32_t u32x4b __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float quad_sum2(float v) {  // across the 4 lanes that share (lane & 15)
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

struct Raw { uint4 u[8]; };  // row t0 + r16 of one head, elements k = 32 ks + 8 q + j

__device__ __forceinline__ Raw raw_load(const uint16_t* __restrict__ qkv, int b, int S, int D, int which, int h, int tile, int r16,
                                        int q) {
  Raw r;
  const int t = tile * 16 + r16;
  const int tc = t < S ? t : S - 1;
  const uint16_t* p = qkv + ((int64_t)(b * S + tc)) * 3 * D + which * D + h * DH2 + 8 * q;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) r.u[ks] = *(const uint4*)(p + 32 * ks);
  return r;
}

// LN over head_dim (+ L2 normalise) -> x[64]  (x[8 ks + j] = element 32 ks + 8 q + j); gain = LDS hn_w[256] | hn_b[256]
template <typename HT>
__device__ __forceinline__ void normalize(const Raw& r, const float* gain, int q, bool l2, float (&x)[64]) {
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    x[8 * ks + 0] = HT::lo(r.u[ks].x), x[8 * ks + 1] = HT::hi(r.u[ks].x);
    x[8 * ks + 2] = HT::lo(r.u[ks].y), x[8 * ks + 3] = HT::hi(r.u[ks].y);
    x[8 * ks + 4] = HT::lo(r.u[ks].z), x[8 * ks + 5] = HT::hi(r.u[ks].z);
    x[8 * ks + 6] = HT::lo(r.u[ks].w), x[8 * ks + 7] = HT::hi(r.u[ks].w);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) s += x[i];
  const float mean = quad_sum2(s) * (1.f / DH2);
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    x[i] -= mean;
    v += x[i] * x[i];
  }
  const float rstd = rsqrtf(quad_sum2(v) * (1.f / DH2) + 1e-5f);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const f32x4 w0 = *(const f32x4*)(gain + 32 * ks + 8 * q), w1 = *(const f32x4*)(gain + 32 * ks + 8 * q + 4);
    const f32x4 b0 = *(const f32x4*)(gain + DH2 + 32 * ks + 8 * q), b1 = *(const f32x4*)(gain + DH2 + 32 * ks + 8 * q + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x[8 * ks + j] = x[8 * ks + j] * rstd * w0[j] + b0[j];
      x[8 * ks + 4 + j] = x[8 * ks + 4 + j] * rstd * w1[j] + b1[j];
    }
  }
  if (l2) {
    float n = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) n += x[i] * x[i];
    const float inv = 1.f / fmaxf(sqrtf(quad_sum2(n)), 1e-12f);
#pragma unroll
    for (int i = 0; i < 64; ++i) x[i] *= inv;
  }
}

template <typename HT>
__device__ __forceinline__ typename HT::frag_t make_frag8(const float* x) {
  const u32x4b u = {HT::pack(x[0], x[1]), HT::pack(x[2], x[3]), HT::pack(x[4], x[5]), HT::pack(x[6], x[7])};
  return __builtin_bit_cast(typename HT::frag_t, u);
}

__device__ __forceinline__ float feat(float z) { return 0.1f * exp_fast(fminf(fmaxf(z, -15.f), 15.f)); }  // (:58-66)

// ---- launch 1: feature maps ----------------------------------------------------------------------------------------------
template <typename HT>
__global__ __launch_bounds__(NTH2, 2) void perf_feat256_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ PT,
                                                                int ldp, const float* __restrict__ hn_w,
                                                                const float* __restrict__ hn_b, const int* __restrict__ len, int S,
                                                                int H, int nbh, uint16_t* __restrict__ qphi,
                                                                uint16_t* __restrict__ kphiT, float* __restrict__ den) {
  typedef typename HT::frag_t frag_t;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  uint16_t* P = (uint16_t*)smem_raw;        // P^T [256 m][264]
  float* gain = (float*)(smem_raw + R_BIG);  // hn_w[256] | hn_b[256]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r16 = lane & 15, q = lane >> 4;
  const int D = H * DH2;
  const int ntile = (S + 15) >> 4, SP = ntile * 16, TP = (S + 31) & ~31;
  for (int i = tid; i < 2 * DH2; i += NTH2) gain[i] = i < DH2 ? hn_w[i] : hn_b[i - DH2];
#pragma unroll 1
  for (int k0 = 0; k0 < 16; k0 += 8) {  // 256 rows x 256 k = 8192 16-B chunks, 16 per thread in two batches
    uint4 tmp[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = tid + NTH2 * (k0 + k);
      tmp[k] = *(const uint4*)(PT + (int64_t)(i >> 5) * ldp + (i & 31) * 8);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = tid + NTH2 * (k0 + k);
      *(uint4*)((uint8_t*)P + (i >> 5) * 512 + ((((i & 31) ^ ((i >> 5) & 15))) << 4)) = tmp[k];  // 512-B rows, chunk ^ (row & 15)
    }
  }
  __syncthreads();

  const int units = nbh * ntile;
#pragma unroll 1
  for (int u = blockIdx.x * NW2 + wid; u < units; u += gridDim.x * NW2) {
    const int bh = u / ntile, tile = u - bh * ntile, b = bh / H, h = bh - b * H;
    int nvalid = len[b];
    nvalid = nvalid < S ? nvalid : S;
    const int t0 = tile * 16, tq = t0 + r16;
    frag_t kf[8], qf[8];
    {
      float x[64];
      normalize<HT>(raw_load(qkv, b, S, D, 1, h, tile, r16, q), gain, q, true, x);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) kf[ks] = make_frag8<HT>(x + 8 * ks);
    }
    {
      float x[64];
      normalize<HT>(raw_load(qkv, b, S, D, 0, h, tile, r16, q), gain, q, true, x);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) qf[ks] = make_frag8<HT>(x + 8 * ks);
    }
    uint16_t* kT = kphiT + (int64_t)bh * DH2 * TP;
    uint16_t* qo = qphi + (int64_t)bh * S * DH2;
    const bool liveq = tq < nvalid;
    float dsum = 0.f;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {  // 8 of the 16 feature tiles at a time: 96 accumulator registers
      f32x4 aT[8], aQ[8], aK[8];
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) aT[mt] = aQ[mt] = aK[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const int pr = 128 * half + 16 * mt + r16;
          const frag_t p = *(const frag_t*)((const uint8_t*)P + pr * 512 + (((4 * ks + q) ^ (pr & 15)) << 4));
          aT[mt] = HT::mfma16(kf[ks], p, aT[mt]);  // D[t][m]: lane col m = r16, rows t = 4 q + r
          aQ[mt] = HT::mfma16(p, qf[ks], aQ[mt]);  // D[m][t]: lane col t = r16, rows m = 4 q + r
          aK[mt] = HT::mfma16(p, kf[ks], aK[mt]);
        }
        __builtin_amdgcn_sched_barrier(0);  // one feature tile's fragment reads at a time (hipcc otherwise hoists them all)
      }
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        const int m0 = 128 * half + 16 * mt;
        float v[4], fq[4], fk[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = (t0 + 4 * q + r) < nvalid ? feat(aT[mt][r]) : 0.f;  // key mask (:69-74)
          fq[r] = feat(aQ[mt][r]);
          fk[r] = liveq ? feat(aK[mt][r]) : 0.f;
          dsum += fq[r] * fk[r];  // same-t dot (:81)
        }
        *(uint2*)(kT + (int64_t)(m0 + r16) * TP + t0 + 4 * q) = make_uint2(HT::pack(v[0], v[1]), HT::pack(v[2], v[3]));
        if (tq < S) *(uint2*)(qo + (int64_t)tq * DH2 + m0 + 4 * q) = make_uint2(HT::pack(fq[0], fq[1]), HT::pack(fq[2], fq[3]));
      }
    }
    dsum = quad_sum2(dsum);
    if (q == 0 && tq < S) den[(int64_t)bh * S + tq] = dsum;
    if (tile == ntile - 1 && TP > SP) {  // zero the K-padding columns [SP, TP) of kphi^T
#pragma unroll
      for (int mt = 0; mt < 16; ++mt) *(uint2*)(kT + (int64_t)(16 * mt + r16) * TP + SP + 4 * q) = make_uint2(0u, 0u);
    }
  }
}

// ---- launch 2: KV state, numerator, normalisation --------------------------------------------------------------------------
// v image: two [TP][128 d] halves with plain 256-byte rows; 16-byte chunk ch of row `row` sits at chunk ch ^ f(row): the
// row-major chunk writes and the transposed 4 x 16 block reads are both conflict-free
__device__ __forceinline__ int voff(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

template <typename HT, int NT>
__global__ __launch_bounds__(NTH2, 2) void perf_kvnum256_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ qphi,
                                                                 const uint16_t* __restrict__ kphiT, const float* __restrict__ den,
                                                                 const float* __restrict__ hn_w, const float* __restrict__ hn_b,
                                                                 int S, int H, uint16_t* __restrict__ out) {
  typedef typename HT::frag_t frag_t;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  uint16_t* kv = (uint16_t*)smem_raw;        // KV^T [256 d][264]  (after the v images are dead)
  float* gain = (float*)(smem_raw + R_BIG);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r16 = lane & 15, q = lane >> 4;
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int D = H * DH2;
  const int ntile = (S + 15) >> 4, SP = ntile * 16, TP = (S + 31) & ~31;
  const int IMG = TP * 256;  // bytes per v half image
  for (int i = tid; i < 2 * DH2; i += NTH2) gain[i] = i < DH2 ? hn_w[i] : hn_b[i - DH2];
  __syncthreads();

  // ---- v rows -> LN -> LDS, row-major, rows >= S zero -------------------------------------------------------------------------
#pragma unroll 1
  for (int it = 0; it < 2; ++it) {
    const int tile = wid + NW2 * it;
    if (tile >= ntile) break;
    const int t = tile * 16 + r16;
    float x[64];
    normalize<HT>(raw_load(qkv, b, S, D, 2, h, tile, r16, q), gain, q, false, x);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {  // d = 32 ks + 8 q + j: half ks >> 2, chunk 4 (ks & 3) + q
      u32x4b u = {HT::pack(x[8 * ks], x[8 * ks + 1]), HT::pack(x[8 * ks + 2], x[8 * ks + 3]),
                  HT::pack(x[8 * ks + 4], x[8 * ks + 5]), HT::pack(x[8 * ks + 6], x[8 * ks + 7])};
      if (t >= S) u = (u32x4b){0u, 0u, 0u, 0u};
      *(u32x4b*)(smem_raw + (ks >> 2) * IMG + voff(t, 4 * (ks & 3) + q)) = u;
    }
  }
  if (TP > SP)  // the 16 K-padding rows of both halves
    for (int i = tid; i < 2 * 16 * 16; i += NTH2)
      *(u32x4b*)(smem_raw + (i >> 8) * IMG + 256 * (SP + ((i >> 4) & 15)) + 16 * (i & 15)) = (u32x4b){0u, 0u, 0u, 0u};
  __syncthreads();

  // ---- KV^T[d][m] = 0.1 sum_t v[t][d] kphi^T[m][t]  (:77).  Wave = (m quarter, d half): 4 x 8 MFMA tiles --------------------
  {
    const int wmq = wid & 3, wdh = wid >> 2;
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const uint16_t* kT = kphiT + (int64_t)bh * DH2 * TP + (int64_t)(64 * wmq + r16) * TP + 8 * q;
    const uint8_t* img = smem_raw + wdh * IMG;
    const int qp = r16 >> 2, pp = r16 & 3;  // transposing read: this lane addresses row qp, columns 4 pp .. 4 pp + 3 of its group's block
    const int nks = TP >> 5;
    frag_t a[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *(const frag_t*)(kT + (int64_t)16 * i * TP);
#pragma unroll 1
    for (int ks = 0; ks < nks; ++ks) {
      frag_t an[4];
      const int kn = ks + 1 < nks ? ks + 1 : ks;
#pragma unroll
      for (int i = 0; i < 4; ++i) an[i] = *(const frag_t*)(kT + (int64_t)16 * i * TP + 32 * kn);
      const int row0 = 32 * ks + 8 * q + qp;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ch = 2 * j + (pp >> 1);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(img + voff(row0, ch) + 8 * (pp & 1)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(img + voff(row0 + 4, ch) + 8 * (pp & 1)));
        const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        const frag_t vf = __builtin_bit_cast(frag_t, both);  // column d = 16 j + r16, k = t = 32 ks + 8 q + 0..7
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = HT::mfma16(a[i], vf, acc[i][j]);  // D[m][d]: lane col d, rows m = 4 q + r
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = an[i];
    }
    __syncthreads();  // every wave is done with the v images
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
      {
        const int kr = 128 * wdh + 16 * j + r16, kc = 64 * wmq + 16 * i + 4 * q;  // row d, first column m of this lane's 4
        *(uint2*)((uint8_t*)kv + kr * 512 + (((kc >> 3) ^ (kr & 15)) << 4) + ((kc >> 2) & 1) * 8) =
            make_uint2(HT::pack(0.1f * acc[i][j][0], 0.1f * acc[i][j][1]), HT::pack(0.1f * acc[i][j][2], 0.1f * acc[i][j][3]));
      }
  }
  __syncthreads();

  // ---- num[t][d] = sum_m qphi[t][m] KV^T[d][m]  (:78): wave w owns NT token tiles, D[d][t] accumulators -----------------------
  if (wid * NT >= ntile) return;
  frag_t bq[NT][8];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int t = (wid * NT + n) * 16 + r16;
    const uint16_t* src = qphi + ((int64_t)bh * S + (t < S ? t : S - 1)) * DH2 + 8 * q;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) bq[n][ks] = *(const frag_t*)(src + 32 * ks);
  }
  f32x4 accn[NT][16];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) accn[n][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int dt = 0; dt < 16; ++dt) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int kr = 16 * dt + r16;
      const frag_t av = *(const frag_t*)((const uint8_t*)kv + kr * 512 + (((4 * ks + q) ^ (kr & 15)) << 4));
#pragma unroll
      for (int n = 0; n < NT; ++n) accn[n][dt] = HT::mfma16(av, bq[n][ks], accn[n][dt]);  // lane col t = r16, rows d = 4 q + r
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- tail: out = LN_dh(0.1 num / max(den, 1e-6))   (:78,85-90) --------------------------------------------------------------
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int tile = wid * NT + n;
    if (tile >= ntile) break;
    const int t = tile * 16 + r16;
    const float dn = fmaxf(den[(int64_t)bh * S + (t < S ? t : S - 1)], 1e-6f);
    const float sc = 0.1f / dn;
    float s1 = 0.f;
#pragma unroll
    for (int dt = 0; dt < 16; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        accn[n][dt][r] *= sc;
        s1 += accn[n][dt][r];
      }
    const float mean = quad_sum2(s1) * (1.f / DH2);
    float s2 = 0.f;
#pragma unroll
    for (int dt = 0; dt < 16; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        accn[n][dt][r] -= mean;
        s2 += accn[n][dt][r] * accn[n][dt][r];
      }
    const float rstd = rsqrtf(quad_sum2(s2) * (1.f / DH2) + 1e-5f);
    if (t < S) {
      uint16_t* orow = out + ((int64_t)(b * S + t)) * D + h * DH2;
#pragma unroll
      for (int dt = 0; dt < 16; ++dt) {
        const f32x4 w = *(const f32x4*)(gain + 16 * dt + 4 * q), bb = *(const f32x4*)(gain + DH2 + 16 * dt + 4 * q);
        const float y0 = accn[n][dt][0] * rstd * w[0] + bb[0], y1 = accn[n][dt][1] * rstd * w[1] + bb[1];
        const float y2 = accn[n][dt][2] * rstd * w[2] + bb[2], y3 = accn[n][dt][3] * rstd * w[3] + bb[3];
        *(uint2*)(orow + 16 * dt + 4 * q) = make_uint2(HT::pack(y0, y1), HT::pack(y2, y3));
      }
    }
  }
}

inline int64_t up256(int64_t n) { return (n + 255) & ~(int64_t)255; }

template <typename HT>
int launch256(const uint16_t* qkv, const uint16_t* PT, int ldp, const float* hn_w, const float* hn_b, const int* len, int B, int S,
              int H, uint16_t* out, uint8_t* scratch, hipStream_t s) {
  static DevOnce attr;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)perf_feat256_kernel<HT>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_PA2) != hipSuccess ||
        hipFuncSetAttribute((const void*)perf_kvnum256_kernel<HT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_PA2) != hipSuccess ||
        hipFuncSetAttribute((const void*)perf_kvnum256_kernel<HT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_PA2) != hipSuccess)
      return MDM_ERR_LAUNCH;
    attr = true;
  }
  const int nbh = B * H, ntile = (S + 15) >> 4, TP = (S + 31) & ~31;
  uint16_t* qphi = (uint16_t*)scratch;
  uint16_t* kphiT = (uint16_t*)(scratch + up256((int64_t)nbh * S * DH2 * 2));
  float* den = (float*)((uint8_t*)kphiT + up256((int64_t)nbh * DH2 * TP * 2));
  const int units = nbh * ntile;
  int grid = (units + NW2 - 1) / NW2;
  grid = grid < 256 ? grid : 256;  // persistent: one workgroup per CU (P^T fills the LDS), waves stride over the units
  hipLaunchKernelGGL(perf_feat256_kernel<HT>, dim3(grid), dim3(NTH2), SMEM_PA2, s, qkv, PT, ldp, hn_w, hn_b, len, S, H, nbh, qphi,
                     kphiT, den);
  MDM_RETURN_IF_LAUNCH_FAILED();
  if (ntile <= NW2) {
    hipLaunchKernelGGL((perf_kvnum256_kernel<HT, 1>), dim3(nbh), dim3(NTH2), SMEM_PA2, s, qkv, qphi, kphiT, den, hn_w, hn_b, S, H, out);
  } else {
    hipLaunchKernelGGL((perf_kvnum256_kernel<HT, 2>), dim3(nbh), dim3(NTH2), SMEM_PA2, s, qkv, qphi, kphiT, den, hn_w, hn_b, S, H, out);
  }
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

}  // namespace

bool perf_attn256_supported(int dh, int S) { return dh == DH2 && S >= 1 && S <= 224; }
int64_t perf_attn256_scratch_bytes(int B, int H, int S) {
  const int64_t nbh = (int64_t)B * H, TP = (S + 31) & ~31;
  return up256(nbh * S * DH2 * 2) + up256(nbh * DH2 * TP * 2) + up256(nbh * S * 4);
}

int perf_attn256(const void* qkv, int h16, const uint16_t* PT, int ldp, const float* hn_w, const float* hn_b, const int* len,
                 int B, int S, int H, uint16_t* out, void* scratch, hipStream_t s) {
  if (!perf_attn256_supported(DH2, S) || (h16 != MDM_H16_BF16 && h16 != MDM_H16_F16)) return MDM_ERR_UNSUPPORTED;
  if (!qkv || !PT || !hn_w || !hn_b || !len || !out || !scratch || (ldp & 7) || B < 1 || H < 1) return MDM_ERR_ARG;
  if (((S + 31) & ~31) * 256 * 2 > R_BIG) return MDM_ERR_UNSUPPORTED;
  return h16 == MDM_H16_F16
             ? launch256<HF>((const uint16_t*)qkv, PT, ldp, hn_w, hn_b, len, B, S, H, out, (uint8_t*)scratch, s)
             : launch256<HB>((const uint16_t*)qkv, PT, ldp, hn_w, hn_b, len, B, S, H, out, (uint8_t*)scratch, s);
}

}  // namespace mdm
