// Fused MFMA GEMM for gfx950: 128x128x32 block tile, 4 waves (2x2), each wave 64x64 as 4x4
// v_mfma_f32_16x16x32_bf16 tiles, fp32 accumulate.  Operands are staged global -> registers -> LDS
// as bf16 planes (double buffered); fp32 activations are converted on the fly, optionally split
// into hi+lo planes so that hi*hi + hi*lo + lo*hi recovers fp32-grade products on the bf16 pipe
// (gfx950 has no xf32/TF32 path; the native f32 MFMA runs at 1/16 of the bf16 rate).
#include "gemm.h"

namespace mdm {

extern int g_bf16_variant;

namespace {

constexpr int BM = 128, BN = 128, BK = 32, NT = 256;
constexpr int PLANE = BM * BK * 2;  // bytes of one bf16 plane (A and W tiles have the same shape)

typedef __bf16 frag_t __attribute__((ext_vector_type(8)));

// 64-B rows; the four 16-B chunks of a row are rotated by 2*(row>>2) so that every 16-lane group of a
// ds_read_b128 fragment read (16 rows x one k-chunk, lane groups per MI355X_MICROARCH LDS table) hits
// 16 distinct 16-B slots of the 256-B bank row.
__device__ __forceinline__ int lds_off(int row, int k) {
  return row * 64 + ((((k >> 3) + 2 * (row >> 2)) & 3) << 4) + ((k & 7) << 1);
}

template <int KIND, int NPL>
struct TileLoader {
  // staging registers
  f32x4 f[4];
  uint4 q[NPL][2];
  const float* rp[4];
  const uint16_t* wp[NPL][2];
  const float* kbase;
  int64_t ld;
  int rows_left;  // KSTRIDE: valid rows from this thread's first row
  bool vec;

  __device__ __forceinline__ void init(const Operand& op, int64_t boff, int row0, int row_end, int tid) {
    ld = op.ld;
    if constexpr (KIND == OP_F32_ROW) {
      const float* base = (const float*)op.p + boff;
      vec = ((ld & 3) == 0) && ((op.gstride & 3) == 0) && ((((uintptr_t)base) & 15) == 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int r = row0 + (tid >> 3) + 32 * j;
        if (r < row_end) {
          int64_t src = op.gather ? (int64_t)op.gather[r] : (int64_t)r;
          int64_t off = op.rpg ? (src / op.rpg) * op.gstride + (src % op.rpg) * ld : src * ld;
          rp[j] = base + off;
        } else {
          rp[j] = nullptr;
        }
      }
    } else if constexpr (KIND == OP_F32_KSTRIDE) {
      const float* base = (const float*)op.p + boff;
      int r = row0 + 4 * (tid & 31);
      kbase = base + r;
      rows_left = row_end - r;
      vec = ((ld & 3) == 0) && ((((uintptr_t)kbase) & 15) == 0);
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int r = row0 + (tid >> 2) + 64 * j;
        bool ok = r < row_end;
        wp[0][j] = ok ? (const uint16_t*)op.p + boff + (int64_t)r * ld : nullptr;
        if constexpr (NPL == 2) wp[1][j] = ok ? (const uint16_t*)op.p_lo + boff + (int64_t)r * ld : nullptr;
      }
    }
  }

  __device__ __forceinline__ void load(int k0, int K, int tid) {
    if constexpr (KIND == OP_F32_ROW) {
      const int k = k0 + 4 * (tid & 7);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const float* p = rp[j];
        if (p) {
          if (vec && k + 4 <= K) {
            v = *(const f32x4*)(p + k);
          } else {
            if (k + 0 < K) v[0] = p[k + 0];
            if (k + 1 < K) v[1] = p[k + 1];
            if (k + 2 < K) v[2] = p[k + 2];
            if (k + 3 < K) v[3] = p[k + 3];
          }
        }
        f[j] = v;
      }
    } else if constexpr (KIND == OP_F32_KSTRIDE) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + (tid >> 5) + 8 * j;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < K && rows_left > 0) {
          const float* p = kbase + (int64_t)k * ld;
          if (vec && rows_left >= 4) {
            v = *(const f32x4*)p;
          } else {
            v[0] = p[0];
            if (rows_left > 1) v[1] = p[1];
            if (rows_left > 2) v[2] = p[2];
            if (rows_left > 3) v[3] = p[3];
          }
        }
        f[j] = v;
      }
    } else {
      const int k = k0 + 8 * (tid & 3);
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          uint4 v = {0u, 0u, 0u, 0u};
          if (wp[pl][j]) v = *(const uint4*)(wp[pl][j] + k);
          q[pl][j] = v;
        }
    }
  }

  // planes: hi at s, lo at s + PLANE
  __device__ __forceinline__ void store(uint8_t* s, int tid) {
    if constexpr (KIND == OP_F32_ROW) {
      const int k = 4 * (tid & 7);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = (tid >> 3) + 32 * j;
        uint32_t h0, h1, l0, l1;
        if constexpr (NPL == 2) {
          split_bf16(f[j][0], f[j][1], h0, l0);
          split_bf16(f[j][2], f[j][3], h1, l1);
          *(uint2*)(s + PLANE + lds_off(r, k)) = make_uint2(l0, l1);
        } else {
          h0 = pack_bf16(f[j][0], f[j][1]);
          h1 = pack_bf16(f[j][2], f[j][3]);
        }
        *(uint2*)(s + lds_off(r, k)) = make_uint2(h0, h1);
      }
    } else if constexpr (KIND == OP_F32_KSTRIDE) {
      const int r = 4 * (tid & 31);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = (tid >> 5) + 8 * j;
        uint32_t h0, h1, l0 = 0, l1 = 0;
        if constexpr (NPL == 2) {
          split_bf16(f[j][0], f[j][1], h0, l0);
          split_bf16(f[j][2], f[j][3], h1, l1);
        } else {
          h0 = pack_bf16(f[j][0], f[j][1]);
          h1 = pack_bf16(f[j][2], f[j][3]);
        }
        *(uint16_t*)(s + lds_off(r + 0, k)) = (uint16_t)(h0 & 0xffff);
        *(uint16_t*)(s + lds_off(r + 1, k)) = (uint16_t)(h0 >> 16);
        *(uint16_t*)(s + lds_off(r + 2, k)) = (uint16_t)(h1 & 0xffff);
        *(uint16_t*)(s + lds_off(r + 3, k)) = (uint16_t)(h1 >> 16);
        if constexpr (NPL == 2) {
          *(uint16_t*)(s + PLANE + lds_off(r + 0, k)) = (uint16_t)(l0 & 0xffff);
          *(uint16_t*)(s + PLANE + lds_off(r + 1, k)) = (uint16_t)(l0 >> 16);
          *(uint16_t*)(s + PLANE + lds_off(r + 2, k)) = (uint16_t)(l1 & 0xffff);
          *(uint16_t*)(s + PLANE + lds_off(r + 3, k)) = (uint16_t)(l1 >> 16);
        }
      }
    } else {
      const int k = 8 * (tid & 3);
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int r = (tid >> 2) + 64 * j;
          *(uint4*)(s + pl * PLANE + lds_off(r, k)) = q[pl][j];
        }
    }
  }
};

template <int AK, int WK, int NPL, int ACT>
__global__ __launch_bounds__(NT) void gemm_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr int STAGE = 2 * NPL * PLANE;  // A planes then W planes
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  const int ntn = (g.N + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = tile % ntn;
  const int mt = tile / ntn;
  int row0, row_end, grp = 0;
  if (g.goff) {
    int acc_t = 0, found = -1;
    for (int e = 0; e < g.ngroups; ++e) {
      const int b = g.goff[e], en = g.goff[e + 1];
      const int t = (en - b + BM - 1) / BM;
      if (mt < acc_t + t) {
        found = e;
        row0 = b + (mt - acc_t) * BM;
        row_end = en;
        break;
      }
      acc_t += t;
    }
    if (found < 0) return;
    grp = found;
  } else {
    row0 = mt * BM;
    row_end = g.M;
    if (row0 >= row_end) return;
  }
  const int z = blockIdx.z;
  const int z1 = z / g.nb2, z2 = z % g.nb2;
  int64_t offA = (int64_t)z1 * g.A.bs1 + (int64_t)z2 * g.A.bs2;
  int64_t offW = g.goff ? (int64_t)grp * g.W.bs1 : (int64_t)z1 * g.W.bs1 + (int64_t)z2 * g.W.bs2;
  int K = g.K;
  if (g.kgoff) {  // weight-gradient mode: this batch reduces over its own K range of the k-strided operands
    const int k0 = g.kgoff[z];
    K = g.kgoff[z + 1] - k0;
    offA += (int64_t)k0 * g.A.ld, offW += (int64_t)k0 * g.W.ld;
  }
  const int64_t offC = (int64_t)z1 * g.c_bs1 + (int64_t)z2 * g.c_bs2;
  const int64_t offB = g.goff ? (int64_t)grp * g.bias_bs : (int64_t)z * g.bias_bs;

  TileLoader<AK, NPL> ta;
  TileLoader<WK, NPL> tw;
  ta.init(g.A, offA, row0, row_end, tid);
  tw.init(g.W, offW, nt * BN, g.N, tid);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (K + BK - 1) / BK;
  ta.load(0, K, tid);
  tw.load(0, K, tid);
  ta.store(smem, tid);
  tw.store(smem + NPL * PLANE, tid);
  __syncthreads();

  const int frow = lane & 15, fk = 8 * (lane >> 4);
  for (int kt = 0; kt < nk; ++kt) {
    uint8_t* cur = smem + (kt & 1) * STAGE;
    uint8_t* nxt = smem + ((kt & 1) ^ 1) * STAGE;
    const bool more = kt + 1 < nk;
    if (more) {
      ta.load((kt + 1) * BK, K, tid);
      tw.load((kt + 1) * BK, K, tid);
    }
    frag_t a[NPL][4], b[NPL][4];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[pl][i] = *(const frag_t*)(cur + pl * PLANE + lds_off(wm * 64 + i * 16 + frow, fk));
        b[pl][i] = *(const frag_t*)(cur + (NPL + pl) * PLANE + lds_off(wn * 64 + i * 16 + frow, fk));
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (NPL == 2) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][i], b[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][i], b[1][j], acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
      }
    if (more) {
      ta.store(nxt, tid);
      tw.store(nxt + NPL * PLANE, tid);
    }
    __syncthreads();
  }

  // epilogue: C/D map of 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg.  Loads are hoisted ahead of the
  // stores and pointers restrict-qualified (outputs never alias bias / residual inputs).
  float* __restrict__ C = g.C ? g.C + offC : nullptr;
  uint16_t* __restrict__ C16 = g.C16 ? g.C16 + offC : nullptr;
  const float* __restrict__ bias = g.bias ? g.bias + offB : nullptr;
  const float* __restrict__ colscale = g.colscale;
  const float* __restrict__ R1 = g.R1;
  const float* __restrict__ R2 = g.R2;
  float bv[4], cv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = nt * BN + wn * 64 + j * 16 + (lane & 15);
    const int nn = n < g.N ? n : g.N - 1;
    bv[j] = bias ? bias[nn] : 0.f;
    cv[j] = g.out_scale * (colscale ? colscale[nn] : 1.f);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float q1[4][4], q2[4][4], rs[4];
    bool km[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = row0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
      const bool ok = m < row_end;
      rs[r] = (ok && g.rowscale) ? g.rowscale[m] : 1.f;
      km[r] = false;
      if (ok && ACT == ACT_FEAT && g.feat_len) {
        const int tok = m / g.feat_rpt, slot = m - tok * g.feat_rpt;
        if (slot >= g.feat_kslot) {
          const int bb = tok / g.feat_S, t = tok - bb * g.feat_S;
          km[r] = t >= g.feat_len[bb];
        }
      }
      const int64_t mr = g.r1_mod ? (m % g.r1_mod) : m;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = nt * BN + wn * 64 + j * 16 + (lane & 15);
        const bool in = ok && n < g.N;
        q1[r][j] = (in && R1) ? R1[mr * g.ldr1 + n] : 0.f;
        q2[r][j] = (in && R2) ? R2[(int64_t)m * g.ldr2 + n] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = row0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
      if (m >= row_end) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = nt * BN + wn * 64 + j * 16 + (lane & 15);
        if (n >= g.N) continue;
        float v = g.alpha * (acc[i][j][r] + bv[j]);
        if constexpr (ACT == ACT_GELU) {
          v = gelu_erf(v);
        } else if constexpr (ACT == ACT_SILU) {
          v = silu(v);
        } else if constexpr (ACT == ACT_FEAT) {
          v = km[r] ? 0.f : 0.1f * expf(fminf(fmaxf(v, -15.f), 15.f));
        }
        v = v * (cv[j] * rs[r]) + g.r1_scale * q1[r][j] + q2[r][j];
        if (C) C[(int64_t)m * g.ldc + n] = v;
        if (C16) C16[(int64_t)m * g.ldc + n] = (uint16_t)(pack_h16(g.h16, v, 0.f) & 0xffff);
      }
    }
  }
}

template <int AK, int WK, int ACT>
int launch_act(const GemmArgs& a, dim3 grid, hipStream_t s) {
  if (a.precision == 3) {
    hipLaunchKernelGGL((gemm_kernel<AK, WK, 2, ACT>), grid, dim3(NT), 2 * 2 * 2 * PLANE, s, a);
  } else {
    hipLaunchKernelGGL((gemm_kernel<AK, WK, 1, ACT>), grid, dim3(NT), 2 * 2 * 1 * PLANE, s, a);
  }
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

template <int AK, int WK, bool ALL_ACTS>
int launch(const GemmArgs& a, dim3 grid, hipStream_t s) {
  if (a.act == ACT_NONE) return launch_act<AK, WK, ACT_NONE>(a, grid, s);
  if constexpr (ALL_ACTS) {
    if (a.act == ACT_GELU) return launch_act<AK, WK, ACT_GELU>(a, grid, s);
    if (a.act == ACT_SILU) return launch_act<AK, WK, ACT_SILU>(a, grid, s);
    if (a.act == ACT_FEAT) return launch_act<AK, WK, ACT_FEAT>(a, grid, s);
  }
  return MDM_ERR_UNSUPPORTED;
}

}  // namespace

int gemm(const GemmArgs& a_in, hipStream_t stream) {
  GemmArgs a = a_in;
  if (a.precision == 2) a.precision = 1, a.h16 = MDM_H16_F16;  // "2" = single pass with fp16 operands
  if (a.h16 != MDM_H16_F16) a.h16 = MDM_H16_BF16;
  if (a.M <= 0 || a.N <= 0 || a.batch <= 0) return MDM_OK;
  if (a.A.kind == MDM_OP_FP8_ROW || a.W.kind == MDM_OP_FP8_ROW) {
    if (a.a_scale_u == 0.f) a.a_scale_u = 1.f;  // zero-initialised descriptors
    if (a.c8_scale == 0.f) a.c8_scale = 1.f;
    return gemm_fp8(a, stream);
  }
  if (a.K <= 0 || !a.A.p || !a.W.p || (!a.C && !a.C16 && !a.Cx2)) return MDM_ERR_ARG;
  if (a.A.kind == OP_BF16_ROW) {
    // a Linear that carries a weight stream: streamed-weight kernel (gemm_stream.hip); knob 63: the tile kernel as before
    // (knob 68: wherever it is eligible, not only where it was measured faster)
    if (a.w_stream && g_bf16_variant != 63 && (g_bf16_variant >= 64 && g_bf16_variant <= 68 ? gemm_stream1_eligible(a) : gemm_stream1_wanted(a)))
      return gemm_stream1(a, stream);
    return gemm_bf16(a, stream);  // bf16 activations: throughput kernel (gemm2.hip)
  }
  if (a.precision != 1 && a.precision != 3) return MDM_ERR_ARG;
  if (a.W.kind == OP_BF16_ROW) {
    if ((a.W.ld & 31) || (((uintptr_t)a.W.p) & 15) || (a.W.bs1 & 7) || (a.W.bs2 & 7)) return MDM_ERR_ARG;
    if (a.W.ld < ((a.K + 31) & ~31)) return MDM_ERR_ARG;
    if (a.precision == 3 && (!a.W.p_lo || (((uintptr_t)a.W.p_lo) & 15))) return MDM_ERR_ARG;
  }
  if (a.A.kind == OP_BF16_ROW) return MDM_ERR_UNSUPPORTED;
  if (a.goff && (a.batch != 1 || a.ngroups <= 0)) return MDM_ERR_ARG;
  if (a.kgoff && (a.goff || a.A.kind != OP_F32_KSTRIDE || a.W.kind != OP_F32_KSTRIDE)) return MDM_ERR_ARG;
  // pre-split rows x a (hi, lo) pair stream: streamed-weight bf16x3 kernel (gemm_stream3.hip) where it measured faster; knob 69: the
  // tile kernel as before, 70: the streamed kernel wherever it is eligible
  if (a.w_stream && a.A.kind == OP_X2_ROW && g_bf16_variant != 69 && (g_bf16_variant == 70 ? gemm_stream3x_eligible(a) : gemm_stream3x_wanted(a)))
    return gemm_stream3x(a, stream);
  // plain Linears of the fp32-grade mode: LDS-DMA staged bf16x3 kernel (gemm3.hip); knob 36 keeps the register-staged one
  if ((g_bf16_variant != 36 || a.act == ACT_HEADNORM || a.act == ACT_HEADSOFTMAX || a.C16_lo || a.Cx2 || a.A.kind == OP_X2_ROW) &&
      gemm_x3_dma_eligible(a))
    return gemm_x3_dma(a, stream);
  if (a.C16_lo || a.Cx2 || a.A.kind == OP_X2_ROW) return MDM_ERR_UNSUPPORTED;  // pre-split rows / plane outputs exist on that kernel only
  const int tm = (a.M + BM - 1) / BM + (a.goff ? a.ngroups : 0);
  const int tn = (a.N + BN - 1) / BN;
  dim3 grid((unsigned)(tm * tn), 1, (unsigned)a.batch);
  const int ak = a.A.kind, wk = a.W.kind;
  if (ak == OP_F32_ROW && wk == OP_BF16_ROW) return launch<OP_F32_ROW, OP_BF16_ROW, true>(a, grid, stream);
  if (ak == OP_F32_ROW && wk == OP_F32_ROW) return launch<OP_F32_ROW, OP_F32_ROW, false>(a, grid, stream);
  if (ak == OP_F32_ROW && wk == OP_F32_KSTRIDE) return launch<OP_F32_ROW, OP_F32_KSTRIDE, false>(a, grid, stream);
  if (ak == OP_F32_KSTRIDE && wk == OP_F32_KSTRIDE) return launch<OP_F32_KSTRIDE, OP_F32_KSTRIDE, false>(a, grid, stream);
  return MDM_ERR_UNSUPPORTED;
}

}  // namespace mdm
