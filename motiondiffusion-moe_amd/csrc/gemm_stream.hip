// Streamed-weight GEMM of the 16-bit modes:   C = epilogue( A16[M, K] * W[N, K]^T ),   K in {512, 1024}, N % 256 == 0.
// For the Linears whose tile-GEMM launch is a latency chain rather than a throughput problem: the big model's D x D layers
// (transformer.py:188-192 -> D = 1024: 12544 x 1024 x 1024 is 784 tiles of 128 x 128 = 1.5 rounds of two co-resident
// workgroups, 60 us = 0.18 of the MFMA peak in csrc/gemm2.hip) and the same layers at 8 samples per GPU (configs[4]: 200-block grids).
//
// One workgroup = 8 waves owns RT x 16 rows and 8 x NJ x 16 output columns; the waves split the columns only:
//   * the weights never touch LDS: wave w streams the fragments of its 16 NJ columns global -> registers from a packed fragment
//     stream (mdm_gemm_stream1_pack: [N / 16][K / 32] fragments of 1 KiB in MFMA operand order) through a 16-fragment ring, as
//     csrc/mlp_stream.hip does; a fragment feeds RT MFMAs, so at RT = 7 the stream (1 KiB per 112 matrix-pipe cycles and wave) sits
//     just under the ~100 GB/s a CU draws from L2 (profiles/r04_dma_rate.txt);
//   * the activation rows are the shared operand: they pass through LDS in K slices of 256 (RT x 8 KiB per slice, two buffers,
//     LDS-DMA with the 16-B-chunk XOR swizzle applied on the source side, one workgroup barrier per slice);
//   * epilogue straight from the accumulators: a lane holds 4 consecutive columns of one row (operands swapped in the MFMA), so
//     the bias / activation / scale / residual arithmetic and the 16-B stores need no staging.
// The arithmetic of an output element is the tile kernel's (same MFMA, same operand order, k ascending; same epilogue formula).
#include "gemm.h"
#include "kernels.h"

namespace mdm {
namespace {

constexpr int GS_NT = 512, GS_KS = 256, GS_NR = 16;
// fragments between the streams of neighbouring column groups beyond K / 32: with a power-of-two distance every wave of a workgroup
// (and every workgroup) reads the same address bits at the same time
constexpr int GS_SKEW = 1;

struct GsArgs {
  const uint16_t* A;
  int64_t lda;
  int64_t M;
  int N, K;
  const uint16_t* ws;
  const float* bias;
  float alpha, out_scale, r1_scale;
  const float* colscale;
  const float* R1;
  int64_t ldr1;
  const float* R2;
  int64_t ldr2;
  float* C;
  uint16_t* C16;
  int64_t ldc;
};

// LDS-DMA of 16 B per lane to lds + lane * 16, written as inline assembly ON PURPOSE: with the builtin, hipcc's wait-count pass
// sees LDS-DMA writes pending next to the fragment reads of the other buffer and puts s_waitcnt vmcnt(0) in front of the first MFMA
// of every slice -- which drains the weight ring AND waits for the slice that was just requested.  Hidden from the compiler, the
// copies only make its counted waits on the ring a little stricter than needed (the counter is in order); the kernel's own
// s_waitcnt vmcnt(NR) + barrier per slice is what orders the copies against the reads.
__device__ __forceinline__ void gs_glds16(const void* g, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds) : "memory", "m0");
}

// LDS: the two slice buffers (RT x 16 KiB); the epilogue's fp32 staging wants whole tiles (16 RT rows x 512 NJ B) up to the 160 KiB of a CU
constexpr int gs_smem(int rt, int nj) {
  const int ring = 2 * 16 * rt * 512, stage = 16 * rt * 512 * nj;
  return stage <= ring ? ring : (stage < 160 * 1024 ? stage : 160 * 1024);
}

// NSL = K / 256 is a template parameter: with the slice loop rolled the compiler cannot count the ring's loads across the back edge and
// drains the ring (vmcnt(0)) in front of every slice
template <typename HT, int RT, int NJ, int NSL, int ACT>
__global__ __launch_bounds__(GS_NT, 2) void gemm_stream_kernel(const GsArgs g) {
  constexpr int ROWS = 16 * RT, SLICE = ROWS * 512, NR = GS_NR, AHEAD = NR / NJ;  // AHEAD: K steps the ring runs ahead
  constexpr int SMEM = gs_smem(RT, NJ);
  static_assert(NR % NJ == 0 && (8 * NJ) % NR == 0, "ring slots must be compile-time per slice");
  typedef typename HT::frag_t frag_t;
  extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
  int tid = threadIdx.x;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ncb = g.N / (128 * NJ);
  const int rt = blockIdx.x / ncb, cb = blockIdx.x - rt * ncb;
  const int64_t row0 = (int64_t)rt * ROWS;
  constexpr int ks32 = NSL * 8, nsl = NSL;

  // this wave's NJ fragment streams (one per 16 columns): fragment (j, s) at wp + j * jst + s * 1024
  const int cg0 = (cb * 8 + wn) * NJ;
  const int64_t jst = (int64_t)(ks32 + GS_SKEW) * 1024;
  const uint8_t* wp = (const uint8_t*)g.ws + cg0 * jst + (tid & 63) * 16;

  // X slice sl -> buffer: the slice is RT x 8 KiB; one LDS-DMA instruction fills 1 KiB = two rows; wave w issues pieces w, w + 8, ...
  // physical 16-B slot p of row r holds chunk p ^ (r & 15)
  auto issue = [&](int sl, int buf) {  // (addresses recomputed per slice: seven more live registers would spill at RT = 7)
    const uint32_t dst = (uint32_t)(uintptr_t)smem + buf * SLICE + wn * 1024;
    const int lane = tid & 63, slot = lane & 31;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const int r = 2 * (wn + 8 * t) + (lane >> 5);
      int64_t row = row0 + r;
      row = row < g.M ? row : g.M - 1;  // rows past the end: copies of the last row, never stored
      gs_glds16(g.A + row * g.lda + ((slot ^ (r & 15)) << 3) + sl * GS_KS, dst + t * 8192);
    }
  };
  issue(0, 0);
  frag_t R[NR];
#pragma unroll
  for (int f = 0; f < NR; ++f) R[f] = *(const frag_t*)(wp + (f % NJ) * jst + (f / NJ) * 1024);
  wp += AHEAD * 1024;

  f32x4 y[RT][NJ];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) y[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int sl = 0; sl < nsl; ++sl) {
    // my pieces of slice sl have landed (everything older than the ring's NR most recent loads has), then everybody's; behind the
    // barrier every wave is also done with slice sl - 1, whose buffer takes slice sl + 1
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NR) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // unconditional (the last pass re-reads its own slice into the idle buffer): behind a branch the compiler cannot count the
    // loads in flight and drains the weight ring (vmcnt(0)) in front of every slice
    issue(sl + 1 < nsl ? sl + 1 : sl, (sl + 1) & 1);
    asm volatile("" : "+v"(tid));  // lane-constant LDS addresses are recomputed per slice, not kept in registers across the loop
    const int lane = tid & 63, frow = lane & 15, fq = lane >> 4;
    const uint8_t* xb = smem + (sl & 1) * SLICE + frow * 512;
    frag_t A[2][RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) A[0][i] = *(const frag_t*)(xb + i * 8192 + ((fq ^ frow) << 4));
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s + 1 < 8) {
#pragma unroll
        for (int i = 0; i < RT; ++i) A[(s + 1) & 1][i] = *(const frag_t*)(xb + i * 8192 + (((4 * (s + 1) + fq) ^ frow) << 4));
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int slot = (s * NJ + j) % NR;
#pragma unroll
        for (int i = 0; i < RT; ++i) y[i][j] = HT::mfma16(R[slot], A[s & 1][i], y[i][j]);
        R[slot] = *(const frag_t*)(wp + j * jst + s * 1024);  // fragment (j, s + AHEAD); the last AHEAD steps read the next stream / the pad
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    wp += 8 * 1024;
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring's read-ahead and the last (idle) slice copy: nothing may land in LDS after exit
  const int lane = tid & 63, frow = lane & 15, fq = lane >> 4;
  f32x4 bb[NJ], cs[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = (cg0 + j) * 16 + 4 * fq;
    bb[j] = g.bias ? *(const f32x4*)(g.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    cs[j] = (f32x4){g.out_scale, g.out_scale, g.out_scale, g.out_scale};
    if (g.colscale) {
      const f32x4 q = *(const f32x4*)(g.colscale + n);
      cs[j][0] *= q[0], cs[j][1] *= q[1], cs[j][2] *= q[2], cs[j][3] *= q[3];
    }
  }
  auto finish = [&](int i, int j) -> f32x4 {  // (acc + bias) * alpha, activation, scales: lane (frow, fq) holds 4 columns of row 16 i + frow
    f32x4 v;
    if constexpr (ACT == ACT_GELU) {
      const f32x2 a = gelu_erf2((f32x2){g.alpha * (y[i][j][0] + bb[j][0]), g.alpha * (y[i][j][1] + bb[j][1])});
      const f32x2 b = gelu_erf2((f32x2){g.alpha * (y[i][j][2] + bb[j][2]), g.alpha * (y[i][j][3] + bb[j][3])});
      v = (f32x4){a[0], a[1], b[0], b[1]};
    } else {
      v = (f32x4){g.alpha * (y[i][j][0] + bb[j][0]), g.alpha * (y[i][j][1] + bb[j][1]), g.alpha * (y[i][j][2] + bb[j][2]), g.alpha * (y[i][j][3] + bb[j][3])};
    }
    v[0] *= cs[j][0], v[1] *= cs[j][1], v[2] *= cs[j][2], v[3] *= cs[j][3];
    return v;
  };
  if (!g.C && !g.R1 && !g.R2) {
    // ---- 16-bit output only: straight from the accumulators (8 B per lane, 32 B per row and instruction: the bytes are few) ------
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = (cg0 + j) * 16 + 4 * fq;
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const int64_t m = row0 + i * 16 + frow;
        const f32x4 v = finish(i, j);
        if (m < g.M) *(uint2*)(g.C16 + m * g.ldc + n) = make_uint2(HT::pack(v[0], v[1]), HT::pack(v[2], v[3]));
      }
    }
    return;
  }
  // ---- fp32 output / residuals: EPT row tiles at a time staged as fp32 [16 EPT][NB] in LDS (16-B chunks XOR-swizzled by the row), then
  // whole rows: the residual is read and the outputs are written as complete lines (from the accumulators a lane touches 64-B
  // pieces of 16 rows: measured +18 us at 12544 x 1024 against +7 us for this form)
  constexpr int NB = 128 * NJ, EPT = (SMEM / (64 * NB)) < RT ? (SMEM / (64 * NB)) : RT, NPASS = (RT + EPT - 1) / EPT;
  constexpr int TPR = NB / 4, RPS = GS_NT / TPR, NK = 16 * EPT / RPS;  // threads per row, rows per step, steps per pass
  float* const stg = (float*)smem;
  const int cl = tid % TPR, rq = tid / TPR, n = cb * NB + 4 * cl;
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    __syncthreads();
#pragma unroll
    for (int ii = 0; ii < EPT; ++ii) {
      const int i = p * EPT + ii;
      if (i < RT) {
        const int ml = ii * 16 + frow;
#pragma unroll
        for (int j = 0; j < NJ; ++j) *(f32x4*)(stg + ml * NB + (((wn * 4 * NJ + 4 * j + fq) ^ (ml & 31)) << 2)) = finish(i, j);
      }
    }
    // stores count in vmcnt on gfx950: a wait for a residual row that was requested after a store is a wait for that store.  So the
    // pass's residual rows are all requested HERE, before its first store (the staged tiles' accumulators are dead by now), and the
    // store loop below loads nothing
    if (!g.R2) {
      constexpr int KS = 4;
      static_assert(NK % KS == 0, "row batches");
      f32x4 q1[NK];
      if (g.R1) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          int64_t m = row0 + p * EPT * 16 + rq + RPS * k;
          m = m < g.M ? m : g.M - 1;
          q1[k] = *(const f32x4*)(g.R1 + m * g.ldr1 + n);
        }
      } else {
#pragma unroll
        for (int k = 0; k < NK; ++k) q1[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      __syncthreads();
#pragma unroll
      for (int k0 = 0; k0 < NK; k0 += KS) {
        f32x4 o[KS];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const int ml = rq + RPS * (k0 + k);
          o[k] = *(const f32x4*)(stg + ml * NB + ((cl ^ (ml & 31)) << 2));
        }
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const int ml = rq + RPS * (k0 + k);
          const int64_t m = row0 + p * EPT * 16 + ml;
          if (p * EPT * 16 + ml >= 16 * RT || m >= g.M) continue;
          f32x4 v = o[k];
          v[0] += g.r1_scale * q1[k0 + k][0] + 0.f, v[1] += g.r1_scale * q1[k0 + k][1] + 0.f;
          v[2] += g.r1_scale * q1[k0 + k][2] + 0.f, v[3] += g.r1_scale * q1[k0 + k][3] + 0.f;
          if (g.C) *(f32x4*)(g.C + m * g.ldc + n) = v;
          if (g.C16) *(uint2*)(g.C16 + m * g.ldc + n) = make_uint2(HT::pack(v[0], v[1]), HT::pack(v[2], v[3]));
        }
      }
      continue;
    }
    __syncthreads();
    constexpr int KB = RT > 4 ? 2 : 4;  // rows per thread in flight (the accumulators of the later passes are still live)
#pragma unroll
    for (int k0 = 0; k0 < NK; k0 += KB) {
      f32x4 v[KB], q1[KB], q2[KB];
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        const int ml = rq + RPS * (k0 + k);
        int64_t m = row0 + p * EPT * 16 + ml;
        m = m < g.M ? m : g.M - 1;
        v[k] = *(const f32x4*)(stg + ml * NB + ((cl ^ (ml & 31)) << 2));
        q1[k] = (f32x4){0.f, 0.f, 0.f, 0.f}, q2[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (g.R1) q1[k] = *(const f32x4*)(g.R1 + m * g.ldr1 + n);
        if (g.R2) q2[k] = *(const f32x4*)(g.R2 + m * g.ldr2 + n);
      }
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        const int ml = rq + RPS * (k0 + k);
        const int64_t m = row0 + p * EPT * 16 + ml;
        if (k0 + k >= NK || p * EPT * 16 + ml >= 16 * RT || m >= g.M) continue;
        f32x4 o = v[k];
        o[0] += g.r1_scale * q1[k][0] + q2[k][0], o[1] += g.r1_scale * q1[k][1] + q2[k][1];
        o[2] += g.r1_scale * q1[k][2] + q2[k][2], o[3] += g.r1_scale * q1[k][3] + q2[k][3];
        if (g.C) *(f32x4*)(g.C + m * g.ldc + n) = o;
        if (g.C16) *(uint2*)(g.C16 + m * g.ldc + n) = make_uint2(HT::pack(o[0], o[1]), HT::pack(o[2], o[3]));
      }
    }
  }
}

// stream[cg = n / 16][s = k / 32][lane l][8]:  W[16 cg + (l & 15)][32 s + 8 (l >> 4) + e]
template <typename HT>
__global__ __launch_bounds__(256) void gemm_stream1_pack_kernel(const float* w, int64_t ldw, int N, int K, uint16_t* out) {
  const int ks32 = K >> 5, nf = (N >> 4) * ks32;
  for (int fi = blockIdx.x * 4 + (threadIdx.x >> 6); fi < nf; fi += gridDim.x * 4) {
    const int l = threadIdx.x & 63, cg = fi / ks32, s = fi - cg * ks32;
    const float* src = w + (int64_t)(16 * cg + (l & 15)) * ldw + 32 * s + 8 * (l >> 4);
    uint4 o;
    o.x = HT::pack(src[0], src[1]), o.y = HT::pack(src[2], src[3]), o.z = HT::pack(src[4], src[5]), o.w = HT::pack(src[6], src[7]);
    *(uint4*)(out + ((int64_t)cg * (ks32 + GS_SKEW) + s) * 512 + l * 8) = o;
  }
}

int gs_device_cus() {
  static DevInt cus;
  if (!cus) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev_ordinal()) != hipSuccess || n <= 0) n = 256;
    cus = n;
  }
  return cus;
}

constexpr int64_t GS_PAD = 16 * 512;  // elements behind the last fragment: what the ring reads ahead of the last K step

template <typename HT, int RT, int NJ, int NSL>
int launch_gs(const GsArgs& g, int act, hipStream_t s) {
  constexpr int smem = gs_smem(RT, NJ);
  static DevOnce attr;
  if (smem > 65536 && !attr) {
    if (hipFuncSetAttribute((const void*)gemm_stream_kernel<HT, RT, NJ, NSL, ACT_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess ||
        hipFuncSetAttribute((const void*)gemm_stream_kernel<HT, RT, NJ, NSL, ACT_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return MDM_ERR_LAUNCH;
    attr = true;
  }
  const dim3 grid((unsigned)(((g.M + 16 * RT - 1) / (16 * RT)) * (g.N / (128 * NJ))));
  if (act == ACT_GELU) hipLaunchKernelGGL((gemm_stream_kernel<HT, RT, NJ, NSL, ACT_GELU>), grid, dim3(GS_NT), smem, s, g);
  else hipLaunchKernelGGL((gemm_stream_kernel<HT, RT, NJ, NSL, ACT_NONE>), grid, dim3(GS_NT), smem, s, g);
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

template <typename HT, int NSL>
int launch_gs_shape(const GsArgs& g, int act, int rt, int nj, hipStream_t s) {
  if (rt == 7 && nj == 4) return launch_gs<HT, 7, 4, NSL>(g, act, s);
  if (rt == 4 && nj == 4) return launch_gs<HT, 4, 4, NSL>(g, act, s);
  if (rt == 4 && nj == 2) return launch_gs<HT, 4, 2, NSL>(g, act, s);
  return launch_gs<HT, 2, 2, NSL>(g, act, s);
}
template <typename HT>
int launch_gs_k(const GsArgs& g, int act, int rt, int nj, hipStream_t s) {
  return g.K == 1024 ? launch_gs_shape<HT, 4>(g, act, rt, nj, s) : launch_gs_shape<HT, 2>(g, act, rt, nj, s);
}

}  // namespace

extern int g_bf16_variant;

int64_t gemm_stream1_elems(int N, int K) {
  return (N > 0 && K > 0 && N % 256 == 0 && K % 256 == 0) ? (int64_t)(N / 16) * (K / 32 + GS_SKEW) * 512 + GS_PAD : 0;
}

int gemm_stream1_pack(const float* w, int64_t ldw, int N, int K, int h16, uint16_t* out, hipStream_t stream) {
  if (!w || !out || !gemm_stream1_elems(N, K) || ldw < K) return MDM_ERR_UNSUPPORTED;
  if (hipMemsetAsync(out, 0, gemm_stream1_elems(N, K) * sizeof(uint16_t), stream) != hipSuccess) return MDM_ERR_LAUNCH;
  if (h16 == MDM_H16_F16) {
    hipLaunchKernelGGL(gemm_stream1_pack_kernel<HF>, dim3(256), dim3(256), 0, stream, w, ldw, N, K, out);
  } else {
    hipLaunchKernelGGL(gemm_stream1_pack_kernel<HB>, dim3(256), dim3(256), 0, stream, w, ldw, N, K, out);
  }
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

// plain Linear on 16-bit rows with a weight stream: no batch / groups / gather / row scales / feature maps
bool gemm_stream1_eligible(const GemmArgs& a) {
  if (!a.w_stream || a.precision != 1 || a.A.kind != OP_BF16_ROW || a.A.gather || a.A.rpg || a.batch != 1 || a.nb2 != 1 || a.goff || a.kgoff) return false;
  if (a.act != ACT_NONE && a.act != ACT_GELU) return false;
  if (a.rowscale || a.r1_mod || a.C8 || a.C16_lo || a.Cx2 || a.a_scale || a.w_scale) return false;
  if (!gemm_stream1_elems(a.N, a.K) || (a.K != 512 && a.K != 1024) || (a.A.ld & 7) || (a.ldc & 3) || (a.R1 && (a.ldr1 & 3)) || (a.R2 && (a.ldr2 & 3))) return false;
  if ((((uintptr_t)a.A.p) | ((uintptr_t)a.w_stream) | ((uintptr_t)a.C) | ((uintptr_t)a.R1) | ((uintptr_t)a.R2) | ((uintptr_t)a.bias) |
       ((uintptr_t)a.colscale)) & 15)
    return false;
  if (((uintptr_t)a.C16) & 7) return false;
  return a.M > 0 && (int64_t)a.M * a.A.ld < (1ll << 31) && (int64_t)a.M * a.N < (1ll << 40);
}

// Where the streamed kernel is the faster one (tools/gemm_stream_bench.py, same process, 12544 / 6272 / 3136 / 1568 rows x K = 1024;
// profiles/r04_gemm_stream_bench*.txt): with a 16-bit-only epilogue everywhere up to 1024 columns (37 vs 45, 22 vs 24, 13 vs 16, 9 vs
// 13.5 us) and at >= 10 000 rows for 3072 / 4096 columns (105 vs 109, 137 vs 151 us; at 6 272 rows the tile kernel wins, 60 vs 65);
// with fp32 residual / output traffic both kernels sit on the same memory phase, and the streamed one is ahead only at the ends:
// >= 10 000 rows (50 vs 52, 146 vs 159, 185 vs 216 us) and <= 2 000 rows (14.1 vs 15.4)
bool gemm_stream1_wanted(const GemmArgs& a) {
  if (!gemm_stream1_eligible(a)) return false;
  const bool light = !a.C && !a.R1 && !a.R2;
  if (a.M >= 10000) return true;
  return light ? a.N <= 1024 : a.M <= 2000;
}

// tile shape: the largest (rows, columns) whose grid still covers ~3/4 of the CUs; a workgroup streams 32 NJ x K weight bytes
// whatever its height, so tall tiles are what makes the stream cheap per row
void gemm_stream1_shape(int64_t M, int N, int& rt, int& nj) {
  static const int cand[4][2] = {{7, 4}, {4, 4}, {4, 2}, {2, 2}};
  const int64_t want = (int64_t)gs_device_cus() * 3 / 4;
  for (int c = 0; c < 4; ++c) {
    rt = cand[c][0], nj = cand[c][1];
    if (N % (128 * nj)) continue;
    if (((M + 16 * rt - 1) / (16 * rt)) * (N / (128 * nj)) >= want) return;
  }
  rt = 2, nj = 2;
}

int gemm_stream1(const GemmArgs& a, hipStream_t stream) {
  if (!gemm_stream1_eligible(a)) return MDM_ERR_UNSUPPORTED;
  GsArgs g = {};
  g.A = (const uint16_t*)a.A.p, g.lda = a.A.ld, g.M = a.M, g.N = a.N, g.K = a.K, g.ws = a.w_stream, g.bias = a.bias;
  g.alpha = a.alpha, g.out_scale = a.out_scale, g.r1_scale = a.r1_scale, g.colscale = a.colscale;
  g.R1 = a.R1, g.ldr1 = a.ldr1, g.R2 = a.R2, g.ldr2 = a.ldr2, g.C = a.C, g.C16 = a.C16, g.ldc = a.ldc;
  int rt, nj;
  gemm_stream1_shape(a.M, a.N, rt, nj);
  // A/B knobs 64..67 force a tile shape
  if (g_bf16_variant == 64) rt = 7, nj = 4;
  if (g_bf16_variant == 65) rt = 4, nj = 4;
  if (g_bf16_variant == 66) rt = 4, nj = 2;
  if (g_bf16_variant == 67) rt = 2, nj = 2;
  if (a.N % (128 * nj)) nj = 2;
  return a.h16 == MDM_H16_F16 ? launch_gs_k<HF>(g, a.act, rt, nj, stream) : launch_gs_k<HB>(g, a.act, rt, nj, stream);
}

}  // namespace mdm
