// fp8 (OCP e4m3) grouped GEMM for the expert MLPs (BASELINE configs[4]: "fp8 MFMA expert GEMMs"):
//     C = epilogue( (A8[M,K] W8[N,K]^T) * a_scale[m] * w_scale[n] )
// Operands are 1 byte per element: activations quantised per ROW (scale = amax / 448, written by the router kernel next to
// the fp8 rows), weights per OUTPUT CHANNEL at pack time (mdm_pack_fp8); products run on the block-scaled matrix instruction
// v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (K = 128 per instruction: twice the bf16 MFMA rate, half the
// operand bytes), fp32 accumulation, and the two scale vectors are applied to the accumulator in the epilogue -- exact for
// per-row / per-channel scaling.  The router (LayerNorm, gate logits, top-2) stays fp32: routing is unchanged by this mode.
//
// Structure = gemm2.hip: 128(or 64) x 128 tile, 4 waves (2 x 2), both operands by LDS-DMA into a 2-stage ring of 128-BYTE
// rows (= 128 k per K tile), 16-B chunks XOR-swizzled by (row & 7) on the DMA source and on the fragment read, MFMA operands
// swapped so each lane owns 4 consecutive output columns, epilogue staged through LDS and written as whole rows.  A lane's
// fragment is the 32 consecutive k of its row = two swizzled 16-B chunks (two ds_read_b128).  Both operands use the same
// byte -> k-slot map, so the instruction's internal k order never matters.
// Outputs: fp32 (C), 16-bit (C16, format h16) and / or fp8 (C8 = e4m3(v * c8_scale): the hidden layer of the expert MLP,
// consumed by the second GEMM with a uniform activation scale 1 / c8_scale).
#include "gemm.h"

namespace mdm {
namespace {

constexpr int BN8 = 128, NT8 = 256, BKB = 128;  // BKB: bytes (= k) per K tile

typedef int v8i __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16c(const void* g, uint8_t* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// e4m3fn has no infinity: a value beyond +-448 converts to NaN and would poison the row through the next GEMM, so the static-
// scale outputs (the hidden layer at 8 x GELU) saturate first
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
  a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f), b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
  c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f), d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
  uint32_t r = 0;
  r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, r, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return r;
}

template <int BM, int ACT>
__global__ __launch_bounds__(NT8, 2) void gemm_fp8_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
  constexpr int TILE_A = BM * BKB, TILE_W = BN8 * BKB, STAGE_B = TILE_A + TILE_W;
  constexpr int PPA = BM / 8 / 4, PPW = BN8 / 8 / 4;  // 1-KiB pieces (8 rows) per wave per stage
  constexpr int MI = BM / 32;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int ntn = (g.N + BN8 - 1) / BN8;
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
  const int64_t offW = g.goff ? (int64_t)grp * g.W.bs1 : 0;   // bytes == elements
  const int64_t offB = g.goff ? (int64_t)grp * g.bias_bs : 0;  // bias / w_scale rows of this group

  const int sub = lane >> 3;
  const uint8_t* pa[PPA];
  const uint8_t* pw[PPW];
#pragma unroll
  for (int i = 0; i < PPA; ++i) {
    const int tr = (wid * PPA + i) * 8 + sub;
    int r = row0 + tr;
    r = r < row_end ? r : row_end - 1;
    const int64_t src = g.A.gather ? (int64_t)g.A.gather[r] : (int64_t)r;
    pa[i] = (const uint8_t*)g.A.p + src * g.A.ld + (((lane & 7) ^ (tr & 7)) << 4);
  }
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int tr = (wid * PPW + i) * 8 + sub;
    int n = nt * BN8 + tr;
    n = n < g.N ? n : g.N - 1;
    pw[i] = (const uint8_t*)g.W.p + offW + (int64_t)n * g.W.ld + (((lane & 7) ^ (tr & 7)) << 4);
  }
  auto stage = [&](int kt, int buf) {
    uint8_t* sa = smem + buf * STAGE_B + wid * PPA * 1024;
    uint8_t* sw = smem + buf * STAGE_B + TILE_A + wid * PPW * 1024;
    const int k0 = kt * BKB;
#pragma unroll
    for (int i = 0; i < PPA; ++i) glds16c(pa[i] + k0, sa + i * 1024);
#pragma unroll
    for (int i = 0; i < PPW; ++i) glds16c(pw[i] + k0, sw + i * 1024);
  };
  const int nk = g.K / BKB;
  stage(0, 0);

  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    const uint8_t* sa = smem + (kt & 1) * STAGE_B;
    const uint8_t* sw = sa + TILE_A;
    v8i a[MI], b[4];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int ra = wm * (BM / 2) + i * 16 + frow;
      const uint4 lo = *(const uint4*)(sa + ra * BKB + (((2 * fq) ^ (ra & 7)) << 4));
      const uint4 hi = *(const uint4*)(sa + ra * BKB + (((2 * fq + 1) ^ (ra & 7)) << 4));
      a[i] = (v8i){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rb = wn * 64 + j * 16 + frow;
      const uint4 lo = *(const uint4*)(sw + rb * BKB + (((2 * fq) ^ (rb & 7)) << 4));
      const uint4 hi = *(const uint4*)(sw + rb * BKB + (((2 * fq + 1) ^ (rb & 7)) << 4));
      b[j] = (v8i){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)  // D = W A^T: lane (col m = frow, rows n = 4 fq + r); formats 0 = e4m3, unit block scales
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b[j], a[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }

  // ---- epilogue: dequantise, bias, activation, scales -> LDS [BM][128] fp32 (chunks XOR-swizzled by the row) -> whole rows
  const float* __restrict__ bias = g.bias ? g.bias + offB : nullptr;
  const float* __restrict__ wsc = g.w_scale ? g.w_scale + offB : nullptr;
  const int nbase = nt * BN8 + wn * 64 + fq * 4;
  __syncthreads();
  float* stg = (float*)smem;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int ml = wm * (BM / 2) + i * 16 + frow, m = row0 + ml;
    const int mc = m < row_end ? m : row_end - 1;
    float as = g.a_scale_u;
    if (g.a_scale) as *= g.a_scale[g.A.gather ? g.A.gather[mc] : mc];
    const float rs = g.out_scale * (g.rowscale ? g.rowscale[mc] : 1.f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nbase + j * 16 + r;
        const int nn = n < g.N ? n : g.N - 1;
        float x = acc[i][j][r] * as * (wsc ? wsc[nn] : 1.f) + (bias ? bias[nn] : 0.f);
        if constexpr (ACT == ACT_GELU) x = gelu_erf(x);
        v[r] = x * rs;
      }
      const int chunk = wn * 16 + j * 4 + fq;
      *(f32x4*)(stg + ml * 128 + ((chunk ^ (ml & 31)) << 2)) = v;
    }
  }
  __syncthreads();
  const int cl = tid & 31, n = nt * BN8 + 4 * cl;
  const bool full = n + 4 <= g.N && (g.ldc & 3) == 0;
#pragma unroll
  for (int k = 0; k < BM / 8; ++k) {
    const int ml = (tid >> 5) + 8 * k, m = row0 + ml;
    if (m >= row_end || n >= g.N) continue;
    const f32x4 v = *(const f32x4*)(stg + ml * 128 + ((cl ^ (ml & 31)) << 2));
    if (full) {
      if (g.C) *(f32x4*)(g.C + (int64_t)m * g.ldc + n) = v;
      if (g.C16) *(uint2*)(g.C16 + (int64_t)m * g.ldc + n) = make_uint2(pack_h16(g.h16, v[0], v[1]), pack_h16(g.h16, v[2], v[3]));
      if (g.C8) *(uint32_t*)(g.C8 + (int64_t)m * g.ldc + n) = pack_fp8x4(v[0] * g.c8_scale, v[1] * g.c8_scale, v[2] * g.c8_scale, v[3] * g.c8_scale);
    } else {
      for (int r = 0; r < 4 && n + r < g.N; ++r) {
        if (g.C) g.C[(int64_t)m * g.ldc + n + r] = v[r];
        if (g.C16) g.C16[(int64_t)m * g.ldc + n + r] = (uint16_t)(pack_h16(g.h16, v[r], 0.f) & 0xffff);
        if (g.C8) g.C8[(int64_t)m * g.ldc + n + r] = (uint8_t)(pack_fp8x4(v[r] * g.c8_scale, 0.f, 0.f, 0.f) & 0xff);
      }
    }
  }
}

// fp32 [rows, K] -> e4m3 [rows, ld_dst] (zero padded) + per-row scale = amax / 448 (1 for an all-zero row); one wave per row
__global__ __launch_bounds__(256) void pack_fp8_kernel(const float* __restrict__ src, int64_t ld_src, int64_t rows, int64_t K,
                                                       uint8_t* __restrict__ dst, int64_t ld_dst, float* __restrict__ scales) {
  const int lane = threadIdx.x & 63;
  const int64_t row = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = src + row * ld_src;
  float amax = 0.f;
  for (int64_t k = lane; k < K; k += 64) amax = fmaxf(amax, fabsf(p[k]));
  amax = wave_max(amax);
  const float scale = amax > 0.f ? amax * (1.f / 448.f) : 1.f;
  const float inv = 1.f / scale;
  if (lane == 0) scales[row] = scale;
  for (int64_t k4 = 4 * (int64_t)lane; k4 < ld_dst; k4 += 256) {
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = k4 + r < K ? p[k4 + r] * inv : 0.f;
    *(uint32_t*)(dst + row * ld_dst + k4) = pack_fp8x4(v[0], v[1], v[2], v[3]);
  }
}

template <int BM>
int launch8(const GemmArgs& a, hipStream_t stream) {
  constexpr int smem = 2 * (BM + BN8) * BKB;
  const int tm = (a.M + BM - 1) / BM + (a.goff ? a.ngroups : 0);
  const int tn = (a.N + BN8 - 1) / BN8;
  const dim3 grid((unsigned)(tm * tn));
  if (a.act == ACT_GELU) {
    hipLaunchKernelGGL((gemm_fp8_kernel<BM, ACT_GELU>), grid, dim3(NT8), smem, stream, a);
  } else {
    hipLaunchKernelGGL((gemm_fp8_kernel<BM, ACT_NONE>), grid, dim3(NT8), smem, stream, a);
  }
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

}  // namespace

bool gemm_fp8_eligible(const GemmArgs& a) {
  return a.A.kind == MDM_OP_FP8_ROW && a.W.kind == MDM_OP_FP8_ROW && a.K >= BKB && (a.K % BKB) == 0 && (a.A.ld % 16) == 0 &&
         (a.W.ld % 16) == 0 && (a.W.bs1 % 16) == 0 && ((((uintptr_t)a.A.p) | ((uintptr_t)a.W.p)) & 15) == 0 && a.batch == 1 &&
         !a.R1 && !a.R2 && !a.colscale && (a.act == ACT_NONE || a.act == ACT_GELU) && a.alpha == 1.f;
}

int gemm_fp8(const GemmArgs& a, hipStream_t stream) {
  if (!gemm_fp8_eligible(a)) return MDM_ERR_UNSUPPORTED;
  if (!a.C && !a.C16 && !a.C8) return MDM_ERR_ARG;
  const int64_t tiles128 = (int64_t)((a.M + 127) / 128) * ((a.N + BN8 - 1) / BN8);
  if (!a.goff && tiles128 <= 256) return launch8<64>(a, stream);
  return launch8<128>(a, stream);
}

int pack_fp8(const float* src, int64_t ld_src, int64_t rows, int64_t K, uint8_t* dst, int64_t ld_dst, float* scales,
             hipStream_t stream) {
  if (!src || !dst || !scales || rows < 0 || K <= 0 || ld_dst < K || (ld_dst & 127)) return MDM_ERR_ARG;
  if (rows == 0) return MDM_OK;
  hipLaunchKernelGGL(pack_fp8_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, src, ld_src, rows, K, dst, ld_dst,
                     scales);
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

}  // namespace mdm

extern "C" int mdm_pack_fp8(const float* src, int64_t ld_src, int64_t rows, int64_t K, uint8_t* dst, int64_t ld_dst,
                            float* scales, void* stream) {
  return mdm::pack_fp8(src, ld_src, rows, K, dst, ld_dst, scales, (hipStream_t)stream);
}
