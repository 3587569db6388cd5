// Fused Performer-style linear attention core (fast_attention.py:29-92) for head_dim = 256 (the BIG model: latent 1024,
// 4 heads, m = min(head_dim, 256) = 256 features), throughput modes.  One workgroup (8 waves) per (batch, head).
//
// At head_dim 128 (perf_attn.hip) the feature image kphi^T, v^T and the KV state all fit the 160 KiB of LDS at once; at 256
// each of kphi^T [m x T] and v^T [d x T] alone is 119 KB.  The work is therefore blocked over FEATURE halves (mh: 128 of
// the 256 features) and VALUE halves (dv: 128 of the 256 head dims), and split in two parts:
//   part 1, per mh:  P^T half -> LDS;  K phase: k rows -> LN(dh) -> L2 norm -> MFMA with P^T half -> 0.1 exp(clamp) -> mask
//                    -> kphi^T half [128 m][T] in LDS;  per dv: v rows -> LN -> v^T half [128 d][T] in LDS (over the dead
//                    P^T image);  KV^T[dv][mh] = 0.1 sum_t v^T kphi^T (both operands t-contiguous) -> global scratch
//                    (16-bit, 136-element rows: the LDS image of part 2; 139 KB per (batch, head), L2 resident);
//   part 2, per mh:  P^T half + KV^T half -> LDS;  per token tile: k features again (for the same-t denominator, kept as
//                    packed 16-bit), q features with the operands swapped so that the accumulator IS the B operand of the
//                    next MFMA;  den += <qphi, kphi>;  num += qphi KV  (num accumulators of the wave's <= 2 token tiles
//                    stay in registers across the two feature halves);
//   tail:            out = LN_dh(0.1 num / max(den, 1e-6)), 16-bit rows.
// Rows of q / k / v are re-read from L2 per phase (10 row reads of 512 B per token in all) instead of being kept: the
// register file holds the 128 num accumulators.  LDS: P^T half 66 KiB + (kphi^T half | KV^T half) 68 KiB + LN gain / bias.
#include "kernels.h"

namespace mdm {
namespace {

constexpr int DH2 = 256, MH = 128, PS2 = 264, KS2 = 136, NW2 = 8, NTH2 = 64 * NW2;
constexpr int R_P = MH * PS2 * 2;    // 67584: P^T half [128][264]; later v^T half [128][TS]
constexpr int R_KV = DH2 * KS2 * 2;  // 69632: KV^T half [256][136]; in part 1 kphi^T half [128][TS]
constexpr int SMEM_PA2 = R_P + R_KV + 2 * DH2 * 4;

typedef uint32_t u32x4b __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float quad_sum2(float v) {  // across the 4 lanes that share (lane & 15)
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

template <typename HT>
__global__ __launch_bounds__(NTH2, 2) void perf_attn256_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ PT,
                                                                int ldp, const float* __restrict__ hn_w,
                                                                const float* __restrict__ hn_b, const int* __restrict__ len,
                                                                int S, int H, uint16_t* __restrict__ out,
                                                                uint16_t* __restrict__ scratch) {
  typedef typename HT::frag_t frag_t;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  uint16_t* rP = (uint16_t*)smem_raw;             // P^T half, or v^T half
  uint16_t* rKV = (uint16_t*)(smem_raw + R_P);    // kphi^T half (part 1), KV^T half (part 2)
  float* gain = (float*)(smem_raw + R_P + R_KV);  // hn_w[256] | hn_b[256]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r16 = lane & 15, q = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int D = H * DH2;
  const int ntile = (S + 15) >> 4, SP = ntile * 16, TP = (S + 31) & ~31, TS = TP + 8;
  uint16_t* kvs = scratch + (int64_t)blockIdx.x * (2 * DH2 * KS2);  // [mh][256 d][136]
  int nvalid = len[b];
  nvalid = nvalid < S ? nvalid : S;
  for (int i = tid; i < 2 * DH2; i += NTH2) gain[i] = i < DH2 ? hn_w[i] : hn_b[i - DH2];

  struct Raw { uint4 u[8]; };  // row t0 + r16, elements k = 32 ks + 8 q + j
  auto raw_load = [&](int which, int tile) {
    Raw r;
    const int t = tile * 16 + r16;
    const int tc = t < S ? t : S - 1;
    const uint16_t* p = qkv + ((int64_t)(b * S + tc)) * 3 * D + which * D + h * DH2 + 8 * q;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) r.u[ks] = *(const uint4*)(p + 32 * ks);
    return r;
  };
  // LN over head_dim (+ L2 normalise) -> x[64]  (x[8 ks + j] = element 32 ks + 8 q + j)
  auto normalize = [&](const Raw& r, bool l2, float (&x)[64]) {
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
  };
  auto make_frag8 = [&](const float* x) {
    const u32x4b u = {HT::pack(x[0], x[1]), HT::pack(x[2], x[3]), HT::pack(x[4], x[5]), HT::pack(x[6], x[7])};
    return __builtin_bit_cast(frag_t, u);
  };
  auto load_PT_half = [&](int mh) {  // 128 rows x 256 k = 4096 16-B chunks, 8 per thread
    uint4 tmp[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = tid + NTH2 * k;
      tmp[k] = *(const uint4*)(PT + (int64_t)(MH * mh + (i >> 5)) * ldp + (i & 31) * 8);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = tid + NTH2 * k;
      *(uint4*)(rP + (i >> 5) * PS2 + (i & 31) * 8) = tmp[k];
    }
  };
  constexpr int MAXT = 2;  // S <= 224 -> <= 14 tiles, wave w owns tiles w and w + 8

  // =============================== part 1: KV^T halves -> scratch ============================================================
#pragma unroll 1
  for (int mh = 0; mh < 2; ++mh) {
    __syncthreads();  // previous half's KV reads of rP / rKV are done
    load_PT_half(mh);
    __syncthreads();
    // ---- K phase: kphi^T half [m][t] -> rKV (row stride TS) -----------------------------------------------------------------
#pragma unroll 1
    for (int it = 0; it < MAXT; ++it) {
      const int tile = wid + NW2 * it;
      if (tile >= ntile) break;
      const int t0 = tile * 16;
      float x[64];
      normalize(raw_load(1, tile), true, x);
      frag_t a[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) a[ks] = make_frag8(x + 8 * ks);
      f32x4 acc[8];
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {  // feature tile outermost + a scheduling fence per tile: hipcc otherwise hoists all
#pragma unroll                          // 64 fragment reads of the phase and spills (390 VGPRs in the first version)
        for (int ks = 0; ks < 8; ++ks) {
          const frag_t p = *(const frag_t*)(rP + (16 * mt + r16) * PS2 + 32 * ks + 8 * q);
          acc[mt] = HT::mfma16(a[ks], p, acc[mt]);  // D[t][m]: lane col m = r16, rows t = 4 q + r
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int t = t0 + 4 * q + r;
          v[r] = t < nvalid ? 0.1f * exp_fast(fminf(fmaxf(acc[mt][r], -15.f), 15.f)) : 0.f;  // key mask (:69-74)
        }
        *(uint2*)(rKV + (16 * mt + r16) * TS + t0 + 4 * q) = make_uint2(HT::pack(v[0], v[1]), HT::pack(v[2], v[3]));
      }
    }
    if (TP > SP)  // zero the K-padding columns of kphi^T
      for (int i = tid; i < MH * 16; i += NTH2) rKV[(i >> 4) * TS + SP + (i & 15)] = 0;
    __syncthreads();  // kphi^T half complete, P^T reads done
#pragma unroll 1
    for (int dv = 0; dv < 2; ++dv) {
      // ---- V phase: v^T half [d][t] -> rP (the P^T image is dead until the next feature half) ---------------------------------
#pragma unroll 1
      for (int it = 0; it < MAXT; ++it) {
        const int tile = wid + NW2 * it;
        if (tile >= ntile) break;
        const int t = tile * 16 + r16;
        float x[64];
        normalize(raw_load(2, tile), false, x);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)  // this half's head dims: ks = 4 dv + kk, d_local = 32 kk + 8 q + j
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float xv = dv ? x[8 * (4 + kk) + j] : x[8 * kk + j];
            rP[(32 * kk + 8 * q + j) * TS + t] = t < S ? (uint16_t)(HT::pack(xv, 0.f) & 0xffff) : (uint16_t)0;
          }
      }
      if (TP > SP)
        for (int i = tid; i < MH * 16; i += NTH2) rP[(i >> 4) * TS + SP + (i & 15)] = 0;
      __syncthreads();
      // ---- KV^T[d (this half)][m (this half)] = 0.1 sum_t v^T[d][t] kphi^T[m][t]   (:77); wave w owns the m tile w -------------
      {
        f32x4 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < TP / 32; ++ks) {
          const frag_t a = *(const frag_t*)(rKV + (16 * wid + r16) * TS + 32 * ks + 8 * q);
          frag_t bf[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) bf[j] = *(const frag_t*)(rP + (16 * j + r16) * TS + 32 * ks + 8 * q);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = HT::mfma16(a, bf[j], acc[j]);  // D[m][d]: col d = 16 j + r16, rows m = 16 w + 4 q + r
        }
        uint16_t* dst = kvs + (int64_t)mh * (DH2 * KS2);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          *(uint2*)(dst + (MH * dv + 16 * j + r16) * KS2 + 16 * wid + 4 * q) =
              make_uint2(HT::pack(0.1f * acc[j][0], 0.1f * acc[j][1]), HT::pack(0.1f * acc[j][2], 0.1f * acc[j][3]));
      }
      __syncthreads();  // v^T half consumed before the next V phase / P^T load overwrites it
    }
  }

  // =============================== part 2: features, denominator, num, tail ====================================================
  // Two passes (the wave's first / second token tile): keeping the num accumulators of BOTH tiles across the two feature
  // halves needs 128 registers more than the file has (a first version spilled 390 VGPRs); per pass the P^T and KV^T halves
  // are re-read from L2 instead (2 x 137 KB per (batch, head)).
#pragma unroll 1
  for (int it = 0; it < MAXT; ++it) {
    const int tile = wid + NW2 * it;
    const bool mine = tile < ntile;
    const int t = tile * 16 + r16;
    f32x4 accn[16];
    float den = 0.f;
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) accn[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int mh = 0; mh < 2; ++mh) {
      __syncthreads();  // every wave wrote its KV^T tiles / finished the previous reads of rP and rKV
      load_PT_half(mh);
      {  // KV^T half: 256 rows x 136 elements, straight copy of the scratch image
        const uint16_t* src = kvs + (int64_t)mh * (DH2 * KS2);
        for (int i = tid; i < DH2 * KS2 / 8; i += NTH2) *(uint4*)(rKV + 8 * i) = *(const uint4*)(src + 8 * i);
      }
      __syncthreads();
      if (!mine) continue;
      uint32_t fk[16];  // kphi[t][m] of this half, packed 16-bit: lane column t = r16, rows m = 16 mt + 4 q + r
      {
        float x[64];
        normalize(raw_load(1, tile), true, x);
        frag_t kf[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) kf[ks] = make_frag8(x + 8 * ks);
        f32x4 ak[8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) ak[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
          for (int ks = 0; ks < 8; ++ks) {
            const frag_t p = *(const frag_t*)(rP + (16 * mt + r16) * PS2 + 32 * ks + 8 * q);
            ak[mt] = HT::mfma16(p, kf[ks], ak[mt]);  // D[m][t]: lane col t = r16, rows m = 4 q + r
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        const bool live = t < nvalid;
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
          float f[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) f[r] = live ? 0.1f * exp_fast(fminf(fmaxf(ak[mt][r], -15.f), 15.f)) : 0.f;
          fk[2 * mt] = HT::pack(f[0], f[1]), fk[2 * mt + 1] = HT::pack(f[2], f[3]);
        }
      }
      f32x4 aq[8];
      {
        float x[64];
        normalize(raw_load(0, tile), true, x);
        frag_t qf[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = make_frag8(x + 8 * ks);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) aq[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
          for (int ks = 0; ks < 8; ++ks) {
            const frag_t p = *(const frag_t*)(rP + (16 * mt + r16) * PS2 + 32 * ks + 8 * q);
            aq[mt] = HT::mfma16(p, qf[ks], aq[mt]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) aq[mt][r] = 0.1f * exp_fast(fminf(fmaxf(aq[mt][r], -15.f), 15.f));
        den += aq[mt][0] * HT::lo(fk[2 * mt]) + aq[mt][1] * HT::hi(fk[2 * mt]) + aq[mt][2] * HT::lo(fk[2 * mt + 1]) +
               aq[mt][3] * HT::hi(fk[2 * mt + 1]);  // same-t dot (:81), rows m = 16 mt + 4 q + r
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        // k-slots j = 0..3 <-> m = 32 s + 4 q + j ; j = 4..7 <-> m = 32 s + 16 + 4 q + (j - 4)
        const u32x4b ub = {HT::pack(aq[2 * s][0], aq[2 * s][1]), HT::pack(aq[2 * s][2], aq[2 * s][3]),
                           HT::pack(aq[2 * s + 1][0], aq[2 * s + 1][1]), HT::pack(aq[2 * s + 1][2], aq[2 * s + 1][3])};
        const frag_t bq = __builtin_bit_cast(frag_t, ub);
#pragma unroll
        for (int dt = 0; dt < 16; ++dt) {
          const uint2 lo = *(const uint2*)(rKV + (16 * dt + r16) * KS2 + 32 * s + 4 * q);
          const uint2 hi = *(const uint2*)(rKV + (16 * dt + r16) * KS2 + 32 * s + 16 + 4 * q);
          const u32x4b ua = {lo.x, lo.y, hi.x, hi.y};
          accn[dt] = HT::mfma16(__builtin_bit_cast(frag_t, ua), bq, accn[dt]);  // D[d][t]
          if ((dt & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (!mine) continue;
    // ---- tail: out = LN_dh(0.1 num / max(den, 1e-6))   (:78,85-90) ------------------------------------------------------------
    const float dn = fmaxf(quad_sum2(den), 1e-6f);
    const float sc = 0.1f / dn;
    float s1 = 0.f;
#pragma unroll
    for (int dt = 0; dt < 16; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        accn[dt][r] *= sc;
        s1 += accn[dt][r];
      }
    const float mean = quad_sum2(s1) * (1.f / DH2);
    float s2 = 0.f;
#pragma unroll
    for (int dt = 0; dt < 16; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        accn[dt][r] -= mean;
        s2 += accn[dt][r] * accn[dt][r];
      }
    const float rstd = rsqrtf(quad_sum2(s2) * (1.f / DH2) + 1e-5f);
    if (t < S) {
      uint16_t* orow = out + ((int64_t)(b * S + t)) * D + h * DH2;
#pragma unroll
      for (int dt = 0; dt < 16; ++dt) {
        const f32x4 w = *(const f32x4*)(gain + 16 * dt + 4 * q), bb = *(const f32x4*)(gain + DH2 + 16 * dt + 4 * q);
        const float y0 = accn[dt][0] * rstd * w[0] + bb[0], y1 = accn[dt][1] * rstd * w[1] + bb[1];
        const float y2 = accn[dt][2] * rstd * w[2] + bb[2], y3 = accn[dt][3] * rstd * w[3] + bb[3];
        *(uint2*)(orow + 16 * dt + 4 * q) = make_uint2(HT::pack(y0, y1), HT::pack(y2, y3));
      }
    }
  }
}

}  // namespace

bool perf_attn256_supported(int dh, int S) { return dh == DH2 && S >= 1 && S <= 224; }
int64_t perf_attn256_scratch_bytes(int B, int H) { return (int64_t)B * H * 2 * DH2 * KS2 * 2; }

int perf_attn256(const void* qkv, int h16, const uint16_t* PT, int ldp, const float* hn_w, const float* hn_b, const int* len,
                 int B, int S, int H, uint16_t* out, void* scratch, hipStream_t s) {
  if (!perf_attn256_supported(DH2, S) || (h16 != MDM_H16_BF16 && h16 != MDM_H16_F16)) return MDM_ERR_UNSUPPORTED;
  if (!qkv || !PT || !hn_w || !hn_b || !len || !out || !scratch || (ldp & 7)) return MDM_ERR_ARG;
  const int TP = (S + 31) & ~31, TS = TP + 8;
  if (MH * TS * 2 > R_P || MH * TS * 2 > R_KV) return MDM_ERR_UNSUPPORTED;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)perf_attn256_kernel<HB>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_PA2) != hipSuccess ||
        hipFuncSetAttribute((const void*)perf_attn256_kernel<HF>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_PA2) != hipSuccess)
      return MDM_ERR_LAUNCH;
    attr = true;
  }
  if (h16 == MDM_H16_F16) {
    hipLaunchKernelGGL(perf_attn256_kernel<HF>, dim3(B * H), dim3(NTH2), SMEM_PA2, s, (const uint16_t*)qkv, PT, ldp, hn_w, hn_b, len,
                       S, H, out, (uint16_t*)scratch);
  } else {
    hipLaunchKernelGGL(perf_attn256_kernel<HB>, dim3(B * H), dim3(NTH2), SMEM_PA2, s, (const uint16_t*)qkv, PT, ldp, hn_w, hn_b, len,
                       S, H, out, (uint16_t*)scratch);
  }
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

}  // namespace mdm
