// 256 x 256 x 64 bf16 tile GEMM (8 waves, one workgroup per CU) for the large launches of the denoiser: expert
// W1 / W2 grouped GEMMs, fused QKV, the 4x FFN.
//
// The 128 x 128 kernel (gemm2.hip) is bound by the per-CU L2 -> LDS DMA rate (~64 GB/s: 32 KiB per 2.1 MFLOP);
// a 256 x 256 tile moves 64 KiB per 8.4 MFLOP -- half the bytes per FLOP -- and gives every wave 64 MFMAs per 8
// LDS-DMA issues and 24 fragment reads per K-tile.  Same LDS image / swizzle / swapped-operand conventions as
// gemm2.hip; 2-stage ring (128 KiB); waves 2 (M) x 4 (N), each 128 x 64 = 8 x 4 accumulator tiles (128 VGPRs);
// epilogue staged through LDS in four 64-row slabs, written as full 1-KiB rows.
#include "gemm.h"

namespace mdm {
namespace {

constexpr int BM = 256, BN = 256, BK = 64, NT = 512;
constexpr int ROWB = 2 * BK, TILE_B = BM * ROWB, STAGE_B = 2 * TILE_B;  // 32 KiB per operand per stage

__device__ __forceinline__ void glds16(const void* g, uint8_t* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <typename HT, int ACT>
__global__ __launch_bounds__(NT, 2) void gemm_bf16_256_kernel(const GemmArgs g) {
  typedef typename HT::frag_t frag_t;
  extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3;
  const int ntn = g.N / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = tile % ntn, mt = tile / ntn;
  int row0, row_end, grp = 0;
  if (g.goff) {
    int acc_t = 0, found = -1;
    for (int e = 0; e < g.ngroups; ++e) {
      const int b = g.goff[e], en = g.goff[e + 1];
      const int t = (en - b + BM - 1) / BM;
      if (mt < acc_t + t) {
        found = e, row0 = b + (mt - acc_t) * BM, row_end = en;
        break;
      }
      acc_t += t;
    }
    if (found < 0) return;
    grp = found;
  } else {
    row0 = mt * BM, row_end = g.M;
    if (row0 >= row_end) return;
  }
  const int64_t offW = g.goff ? (int64_t)grp * g.W.bs1 : 0;
  const int64_t offB = g.goff ? (int64_t)grp * g.bias_bs : 0;

  // LDS-DMA sources: 32 pieces (8 rows x 128 B) per operand per stage, 4 + 4 per wave; edge rows clamped
  const int sub = lane >> 3, cswz = ((lane & 7) ^ sub) * 8;
  const uint16_t* pa[4];
  const uint16_t* pw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int tr = (wid * 4 + i) * 8 + sub;
    int r = row0 + tr;
    r = r < row_end ? r : row_end - 1;
    const int64_t src = g.A.gather ? (int64_t)g.A.gather[r] : (int64_t)r;
    pa[i] = (const uint16_t*)g.A.p + src * g.A.ld + cswz;
    pw[i] = (const uint16_t*)g.W.p + offW + (int64_t)(nt * BN + tr) * g.W.ld + cswz;
  }
  auto stage = [&](int kt, int buf) {
    uint8_t* sa = smem + buf * STAGE_B + wid * 4096;
    uint8_t* sw = sa + TILE_B;
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(pa[i] + k0, sa + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(pw[i] + k0, sw + i * 1024);
  };

  float* __restrict__ C = g.C;
  uint16_t* __restrict__ C16 = g.C16;
  const float* __restrict__ bias = g.bias ? g.bias + offB : nullptr;
  const float* __restrict__ R1 = g.R1;
  const float* __restrict__ R2 = g.R2;
  const int frow = lane & 15, fq = lane >> 4;
  const int nbase = nt * BN + wn * 64 + fq * 4;
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    const uint8_t* sa = smem + (kt & 1) * STAGE_B;
    const uint8_t* sw = sa + TILE_B;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      frag_t b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rb = wn * 64 + j * 16 + frow;
        b[j] = *(const frag_t*)(sw + rb * ROWB + (((ks * 4 + fq) ^ (rb & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int ra = wm * 128 + i * 16 + frow;
        const frag_t a = *(const frag_t*)(sa + ra * ROWB + (((ks * 4 + fq) ^ (ra & 7)) << 4));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = HT::mfma16(b[j], a, acc[i][j]);
      }
    }
  }

  // epilogue: four 64-row slabs through a [64][256] fp32 staging image (chunks XOR-swizzled by the row)
  float* stg = (float*)smem;
  const int cl = tid & 63, n = nt * BN + 4 * cl;
  f32x4 bv[4], cv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bv[j] = bias ? *(const f32x4*)(bias + nbase + j * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
    cv[j] = g.colscale ? *(const f32x4*)(g.colscale + nbase + j * 16) : (f32x4){1.f, 1.f, 1.f, 1.f};
  }
  for (int p = 0; p < 4; ++p) {
    // this slab's four row scales, requested together and before the barrier (rows clamped, not predicated: predicated they were
    // four dependent round trips per slab; rows past row_end are never stored)
    float rs4[4] = {1.f, 1.f, 1.f, 1.f};
    if (g.rowscale) {
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int m = row0 + 64 * p + ii * 16 + frow;
        rs4[ii] = g.rowscale[m < row_end ? m : row_end - 1];
      }
    }
    __syncthreads();  // K loop / previous slab's readers done with the staging region
    if (wm == (p >> 1)) {
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int ml = ii * 16 + frow;
        const float rs = rs4[ii];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float a0 = acc[ii][j][r], a1 = acc[4 + ii][j][r];  // static indices; the slab half is a select
            float x = g.alpha * (((p & 1) ? a1 : a0) + bv[j][r]);
            if constexpr (ACT == ACT_GELU) {
              x = gelu_erf(x);
            } else if constexpr (ACT == ACT_SILU) {
              x = silu(x);
            }
            v[r] = x * (g.out_scale * cv[j][r] * rs);
          }
          const int chunk = wn * 16 + j * 4 + fq;
          *(f32x4*)(stg + ml * 256 + ((chunk ^ (ml & 63)) << 2)) = v;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 q1[4], q2[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int m = row0 + 64 * p + (tid >> 6) + 8 * (4 * half + k);
        q1[k] = (f32x4){0.f, 0.f, 0.f, 0.f}, q2[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (m < row_end) {
          const int64_t mr = g.r1_mod ? (m % g.r1_mod) : m;
          if (R1) q1[k] = *(const f32x4*)(R1 + mr * g.ldr1 + n);
          if (R2) q2[k] = *(const f32x4*)(R2 + (int64_t)m * g.ldr2 + n);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int ml = (tid >> 6) + 8 * (4 * half + k), m = row0 + 64 * p + ml;
        if (m >= row_end) continue;
        f32x4 v = *(const f32x4*)(stg + ml * 256 + ((cl ^ (ml & 63)) << 2));
        v[0] += g.r1_scale * q1[k][0] + q2[k][0], v[1] += g.r1_scale * q1[k][1] + q2[k][1];
        v[2] += g.r1_scale * q1[k][2] + q2[k][2], v[3] += g.r1_scale * q1[k][3] + q2[k][3];
        if (C) *(f32x4*)(C + (int64_t)m * g.ldc + n) = v;
        if (C16) *(uint2*)(C16 + (int64_t)m * g.ldc + n) = make_uint2(HT::pack(v[0], v[1]), HT::pack(v[2], v[3]));
      }
    }
  }
}

template <typename HT, int ACT>
int launch256(const GemmArgs& a, hipStream_t stream) {
  constexpr int smem = 2 * STAGE_B;  // 128 KiB
  static DevOnce attr;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)gemm_bf16_256_kernel<HT, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) !=
        hipSuccess)
      return MDM_ERR_LAUNCH;
    attr = true;
  }
  const int tm = (a.M + BM - 1) / BM + (a.goff ? a.ngroups : 0);
  hipLaunchKernelGGL((gemm_bf16_256_kernel<HT, ACT>), dim3((unsigned)(tm * (a.N / BN))), dim3(NT), smem, stream, a);
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

}  // namespace

bool gemm_bf16_256_eligible(const GemmArgs& a) {
  if (!(a.precision == 1 && a.A.kind == OP_BF16_ROW && a.W.kind == OP_BF16_ROW)) return false;
  if (a.batch != 1 || a.A.rpg || a.K < BK || (a.K % BK) || (a.N % BN)) return false;
  if ((a.A.ld % 8) || (a.W.ld % 8) || (a.W.bs1 % 8) || ((((uintptr_t)a.A.p) | ((uintptr_t)a.W.p)) & 15)) return false;
  if ((a.ldc & 3) || (a.R1 && (a.ldr1 & 3)) || (a.R2 && (a.ldr2 & 3))) return false;
  if (a.act != ACT_NONE && a.act != ACT_GELU && a.act != ACT_SILU) return false;
  if (a.bias && ((((uintptr_t)a.bias) & 15) || (a.bias_bs & 3))) return false;
  if (a.colscale && (((uintptr_t)a.colscale) & 15)) return false;
  return true;
}

int gemm_bf16_256(const GemmArgs& a, hipStream_t stream) {
  if (!gemm_bf16_256_eligible(a)) return MDM_ERR_UNSUPPORTED;
  if (!a.C && !a.C16) return MDM_ERR_ARG;
  const bool f16 = a.h16 == MDM_H16_F16;
  switch (a.act) {
    case ACT_NONE: return f16 ? launch256<HF, ACT_NONE>(a, stream) : launch256<HB, ACT_NONE>(a, stream);
    case ACT_GELU: return f16 ? launch256<HF, ACT_GELU>(a, stream) : launch256<HB, ACT_GELU>(a, stream);
    default: return f16 ? launch256<HF, ACT_SILU>(a, stream) : launch256<HB, ACT_SILU>(a, stream);
  }
}

}  // namespace mdm
