// Text cross-attention with the query / output projections folded into the cached text side (throughput mode, D = 512).
//
// MemoryEfficientCrossAttentionBlock (fast_attention.py:305-325) computes, per head h,
//     o = softmax_n( (x Wq_h^T + bq_h) K_h^T / sqrt(dh) ) V_h ,   out = concat_h(o) Wout^T + bout .
// K_h, V_h depend only on the text, so both projections fold into per-text matrices (built once per caption batch by
// mdm_text_cache_build):   K'[h n, :] = K_h[n, :] Wq_h / sqrt(dh),   cb[h n] = K_h[n, :] . bq_h / sqrt(dh),
//                          V'[h n, :] = V_h[n, :] Wout[:, h]^T
// and the block becomes   out = softmax_n( x K'^T + cb ) V' + bout   -- two small GEMMs with H*N "hidden" columns, taken in
// passes of <= 128 columns (whole heads: hpp = min(H, 128 / N) heads per pass; one pass up to N = 32 text tokens at H = 4, two
// up to 64, four up to 128 -- the reference pads captions to 8 + 77 = 85 tokens, text_encoder.py:19,26); the output
// accumulators stay in registers across the passes.
// This kernel does that and the LayerNorm that follows (ffn.0, fast_attention.py:293,329) in ONE launch instead of four
// (query GEMM, attention core, output GEMM, LayerNorm): a workgroup of 8 waves owns <= 64 rows of one sample and all 512
// output columns, so the row statistics are local.
//   phase 1: scores[64 x 128] = x tile . K'^T over K = 512 (x and K' tiles by LDS-DMA, 4-stage ring, 3 pieces per wave)
//   softmax over each head's N columns (fp32, hardware exp2), probabilities to LDS as bf16 A-fragments
//   phase 2: out[64 x 512] = P[64 x 128] . V'^T in four 128-column slabs of V' (32 KiB each, LDS-DMA)
//   epilogue: + bout, two-pass LayerNorm across the 8 waves through LDS, fp32 pre-norm rows and bf16 normalised rows.
#include "kernels.h"

namespace mdm {
namespace {

constexpr int FD = 512, HNP = 128, BM = 64, NT = 512;
constexpr int ST_B = (BM + HNP) * 128;  // 24576: x tile [64][128 B] + K' tile [128][128 B]
constexpr int NST = 4;                  // phase-1 ring: 96 KiB; phase 2 reuses it as three 32-KiB slab slots
constexpr int SLAB_B = 128 * 256;       // V' slab: 128 output columns x 128 k (256-B rows)
constexpr int SC_LD = 132;              // fp32 score row stride (floats)
constexpr int RING_B = NST * ST_B, SC_B = BM * SC_LD * 4, P_B = BM * 256, RED_B = BM * 8 * 4;
constexpr int SMEM_B = RING_B + SC_B + P_B + 2 * RED_B;

__device__ __forceinline__ void glds16(const void* g, uint8_t* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename HT>
__global__ __launch_bounds__(NT) void sd_fold_kernel(const uint16_t* __restrict__ x16, const uint16_t* __restrict__ kfold,
                                                     const float* __restrict__ cb, const uint16_t* __restrict__ vfold,
                                                     const float* __restrict__ bout, const float* __restrict__ ln_w,
                                                     const float* __restrict__ ln_b, float* __restrict__ out32,
                                                     uint16_t* __restrict__ out16, int S, int H, int N, int ntile, int rpt, int hpp,
                                                     int npass) {
  typedef typename HT::frag_t frag_t;
  extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
  uint8_t* ring = smem;
  float* sc = (float*)(smem + RING_B);
  uint8_t* pim = smem + RING_B + SC_B;
  float* red1 = (float*)(smem + RING_B + SC_B + P_B);
  float* red2 = red1 + BM * 8;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r16 = lane & 15, q = lane >> 4;
  const int b = blockIdx.x / ntile, tl = blockIdx.x - b * ntile;
  const int t0 = tl * rpt;
  int nrows = S - t0;
  nrows = nrows < rpt ? nrows : rpt;
  if (nrows <= 0) return;
  const int64_t row0 = (int64_t)b * S + t0;

  // ---- phase 1 sources: 24 pieces of 8 rows x 128 B per stage (8 of x, 16 of K'), 3 per wave --------------------
  const int sub8 = lane >> 3, c8 = ((lane & 7) ^ sub8) * 8;
  const uint16_t* src[3];
  int dst[3];
  int64_t kfoff[3];  // K' pieces: element offset inside one pass's [128][512] image (x pieces: -1)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int p = 3 * wid + i;
    if (p < 8) {
      int r = p * 8 + sub8;
      r = r < nrows ? r : nrows - 1;
      src[i] = x16 + (row0 + r) * FD + c8;
      dst[i] = p * 1024;
      kfoff[i] = -1;
    } else {
      const int r = (p - 8) * 8 + sub8;
      src[i] = nullptr;
      kfoff[i] = (int64_t)r * FD + c8;
      dst[i] = BM * 128 + (p - 8) * 1024;
    }
  }
  f32x4 o[4][4];
#pragma unroll
  for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
    for (int i = 0; i < 4; ++i) o[sidx][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
  for (int pass = 0; pass < npass; ++pass) {
  const int hcnt = (H - pass * hpp) < hpp ? (H - pass * hpp) : hpp;  // heads of this pass
  const uint16_t* kf_p = kfold + ((int64_t)b * npass + pass) * HNP * FD;
  const float* cb_p = cb + ((int64_t)b * npass + pass) * HNP;
  const uint16_t* vf_p = vfold + ((int64_t)b * npass + pass) * FD * HNP;
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (kfoff[i] >= 0) src[i] = kf_p + kfoff[i];
  // zero the probability image (its columns >= hcnt * N stay zero: they multiply the zero padding of V'); the previous
  // pass's readers are behind the barrier that ends it
  *(uint4*)(pim + tid * 32) = make_uint4(0, 0, 0, 0);
  *(uint4*)(pim + tid * 32 + 16) = make_uint4(0, 0, 0, 0);
  auto stage1 = [&](int kt) {
    uint8_t* s = ring + (kt % NST) * ST_B;
#pragma unroll
    for (int i = 0; i < 3; ++i) glds16(src[i] + kt * 64, s + dst[i]);
  };
  constexpr int NK = FD / 64;
  stage1(0), stage1(1), stage1(2);
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int kt = 0; kt < NK; ++kt) {
    const int younger = NK - 1 - kt < 2 ? NK - 1 - kt : 2;
    if (younger == 2) {
      wait_vm<6>();
    } else if (younger == 1) {
      wait_vm<3>();
    } else {
      wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (kt + 3 < NK) stage1(kt + 3);
    const uint8_t* s = ring + (kt % NST) * ST_B;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int rb = 16 * wid + r16;  // this wave's 16 score columns (rows of K')
      const frag_t kf = *(const frag_t*)(s + BM * 128 + rb * 128 + (((ks * 4 + q) ^ (rb & 7)) << 4));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ra = 16 * i + r16;
        const frag_t xf = *(const frag_t*)(s + ra * 128 + (((ks * 4 + q) ^ (ra & 7)) << 4));
        acc[i] = HT::mfma16(kf, xf, acc[i]);  // lane: row 16i + r16, cols 16w + 4q..
      }
    }
  }
  {
    const int n0 = 16 * wid + 4 * q;
    const f32x4 cbv = *(const f32x4*)(cb_p + n0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v = acc[i];
      v[0] += cbv[0], v[1] += cbv[1], v[2] += cbv[2], v[3] += cbv[3];
      *(f32x4*)(sc + (16 * i + r16) * SC_LD + n0) = v;
    }
  }
  __syncthreads();  // scores complete, every wave is done reading the phase-1 ring

  // ---- V' slabs: 32 pieces of 4 rows x 256 B, 4 per wave; 16-B chunk slot (lane & 15) holds chunk slot ^ (row & 15)
  const uint16_t* vsrc;
  {
    const int r = 16 * wid + (lane >> 4);  // + 4 * i
    vsrc = vf_p + (int64_t)r * HNP;
  }
  auto slab = [&](int sidx) {
    uint8_t* s = ring + (sidx % 3) * SLAB_B;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 16 * wid + 4 * i + (lane >> 4);  // row inside the slab
      glds16(vsrc + ((int64_t)sidx * 128 + 4 * i) * HNP + (((lane & 15) ^ (r & 15)) << 3), s + (4 * wid + i) * 1024);
    }
  };
  slab(0), slab(1), slab(2);

  // ---- softmax over each head's N key columns (fast_attention.py:318), probabilities as bf16 -----------------------
  {
    // G lanes per (row, head), a power of two with 64 * hcnt * G <= 512 threads: 8 with one head in the pass, 4 with two,
    // 2 with three or four, 1 beyond
    const int gq = 8 / hcnt;
    const int G = gq >= 8 ? 8 : (gq >= 4 ? 4 : (gq >= 2 ? 2 : 1));
    const int unit = tid / G, sub = tid - unit * G;
    const bool on = unit < BM * hcnt;
    const int row = on ? unit / hcnt : 0, h = on ? unit - row * hcnt : 0;
    const float* p = sc + row * SC_LD + h * N;
    float mx = -INFINITY;
    for (int i = sub; i < N; i += G) mx = fmaxf(mx, p[i]);
    for (int o2 = 1; o2 < G; o2 <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o2, 64));
    float sum = 0.f;
    for (int i = sub; i < N; i += G) sum += exp_fast(p[i] - mx);
    for (int o2 = 1; o2 < G; o2 <<= 1) sum += __shfl_xor(sum, o2, 64);
    const float inv = 1.f / sum;
    if (on)
      for (int i = sub; i < N; i += G) {
        const int c = h * N + i;
        const float e = exp_fast(p[i] - mx) * inv;
        *(uint16_t*)(pim + row * 256 + ((((c >> 3) ^ (row & 15))) << 4) + (c & 7) * 2) = (uint16_t)(HT::pack(e, 0.f) & 0xffff);
      }
  }
  __syncthreads();

  // ---- phase 2: out[64 x 512] = P . V'^T ; wave w owns columns 16w.. of every slab ------------------------------------
  frag_t pf[4][4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ra = 16 * i + r16;
      pf[ks][i] = *(const frag_t*)(pim + ra * 256 + (((ks * 4 + q) ^ (ra & 15)) << 4));
    }
#pragma unroll
  for (int sidx = 0; sidx < 4; ++sidx) {
    // slabs 0..2 were issued together (4 pieces each per wave), slab 3 after slab 0's slot was released
    if (sidx == 0) {
      wait_vm<8>();
    } else if (sidx == 1) {
      wait_vm<4>();  // outstanding: slab 2 only (slab 3 is issued below, after this wait)
    } else if (sidx == 2) {
      wait_vm<4>();  // outstanding: slab 3
    } else {
      wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (sidx == 1) slab(3);  // slot 0 is free: every wave passed the barrier after consuming slab 0
    const uint8_t* s = ring + (sidx % 3) * SLAB_B;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int rb = 16 * wid + r16;
      const frag_t vf = *(const frag_t*)(s + rb * 256 + (((ks * 4 + q) ^ (rb & 15)) << 4));
#pragma unroll
      for (int i = 0; i < 4; ++i) o[sidx][i] = HT::mfma16(vf, pf[ks][i], o[sidx][i]);
    }
  }

  __syncthreads();  // every wave is done with this pass's slabs and probability image
  }  // pass

  // ---- epilogue: + bout, LayerNorm over the 512 columns (two passes through LDS across the 8 waves) -------------------
  // lane: rows m = 16 i + r16 (i < 4), columns j = 128 s + 16 w + 4 q + r
  float ps[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) ps[i] = 0.f;
#pragma unroll
  for (int sidx = 0; sidx < 4; ++sidx) {
    const f32x4 bo = *(const f32x4*)(bout + 128 * sidx + 16 * wid + 4 * q);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[sidx][i][0] += bo[0], o[sidx][i][1] += bo[1], o[sidx][i][2] += bo[2], o[sidx][i][3] += bo[3];
      ps[i] += (o[sidx][i][0] + o[sidx][i][1]) + (o[sidx][i][2] + o[sidx][i][3]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ps[i] += __shfl_xor(ps[i], 16, 64);
    ps[i] += __shfl_xor(ps[i], 32, 64);
    if (q == 0) red1[(16 * i + r16) * 8 + wid] = ps[i];
  }
  __syncthreads();
  float mean[4], rstd[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x4 a = *(const f32x4*)(red1 + (16 * i + r16) * 8), c = *(const f32x4*)(red1 + (16 * i + r16) * 8 + 4);
    mean[i] = (((a[0] + a[1]) + (a[2] + a[3])) + ((c[0] + c[1]) + (c[2] + c[3]))) * (1.f / FD);
    float v = 0.f;
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = o[sidx][i][r] - mean[i];
        v += d * d;
      }
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (q == 0) red2[(16 * i + r16) * 8 + wid] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x4 a = *(const f32x4*)(red2 + (16 * i + r16) * 8), c = *(const f32x4*)(red2 + (16 * i + r16) * 8 + 4);
    rstd[i] = rsqrtf((((a[0] + a[1]) + (a[2] + a[3])) + ((c[0] + c[1]) + (c[2] + c[3]))) * (1.f / FD) + 1e-5f);
  }
#pragma unroll
  for (int sidx = 0; sidx < 4; ++sidx) {
    const int j = 128 * sidx + 16 * wid + 4 * q;
    const f32x4 w = *(const f32x4*)(ln_w + j), bb = *(const f32x4*)(ln_b + j);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = 16 * i + r16;
      if (m >= nrows) continue;
      const f32x4 v = o[sidx][i];
      *(f32x4*)(out32 + (row0 + m) * FD + j) = v;
      const float y0 = (v[0] - mean[i]) * rstd[i] * w[0] + bb[0], y1 = (v[1] - mean[i]) * rstd[i] * w[1] + bb[1];
      const float y2 = (v[2] - mean[i]) * rstd[i] * w[2] + bb[2], y3 = (v[3] - mean[i]) * rstd[i] * w[3] + bb[3];
      *(uint2*)(out16 + (row0 + m) * FD + j) = make_uint2(HT::pack(y0, y1), HT::pack(y2, y3));
    }
  }
}

}  // namespace

bool sd_fold_supported(int D, int H, int N) { return D == FD && H >= 1 && H <= 8 && N >= 1 && N <= HNP; }
// heads per pass / number of passes of <= 128 folded columns
int sd_fold_heads_per_pass(int H, int N) {
  const int hpp = HNP / (N > 0 ? N : 1);
  return hpp < 1 ? 1 : (hpp > H ? H : hpp);
}
int sd_fold_passes(int H, int N) {
  const int hpp = sd_fold_heads_per_pass(H, N);
  return (H + hpp - 1) / hpp;
}

int sd_fold(const uint16_t* x16, const uint16_t* kfold, const float* cb, const uint16_t* vfold, const float* bout,
            const float* ln_w, const float* ln_b, int B, int S, int D, int H, int N, float* out32, uint16_t* out16,
            int h16, hipStream_t s) {
  if (!sd_fold_supported(D, H, N)) return MDM_ERR_UNSUPPORTED;
  if (!x16 || !kfold || !cb || !vfold || !bout || !ln_w || !ln_b || !out32 || !out16 || B <= 0 || S <= 0) return MDM_ERR_ARG;
  static DevOnce attr;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)sd_fold_kernel<HB>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B) != hipSuccess ||
        hipFuncSetAttribute((const void*)sd_fold_kernel<HF>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B) != hipSuccess)
      return MDM_ERR_LAUNCH;
    attr = true;
  }
  const int ntile = (S + BM - 1) / BM, rpt = (S + ntile - 1) / ntile;
  const int hpp = sd_fold_heads_per_pass(H, N), npass = sd_fold_passes(H, N);
  if (h16 == MDM_H16_F16) {
    hipLaunchKernelGGL(sd_fold_kernel<HF>, dim3(B * ntile), dim3(NT), SMEM_B, s, x16, kfold, cb, vfold, bout, ln_w, ln_b,
                       out32, out16, S, H, N, ntile, rpt, hpp, npass);
  } else {
    hipLaunchKernelGGL(sd_fold_kernel<HB>, dim3(B * ntile), dim3(NT), SMEM_B, s, x16, kfold, cb, vfold, bout, ln_w, ln_b,
                       out32, out16, S, H, N, ntile, rpt, hpp, npass);
  }
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

}  // namespace mdm
