// Streamed-weight GEMM in the fp32-grade (bf16x3) arithmetic:   C = epilogue( A[M, K] * (W_hi + W_lo)[N, K]^T ),  precision 3,
// A as pre-split rows (MDM_OP_X2_ROW), optionally gathered and grouped -- the two expert GEMMs of the fp32-grade mode
// (switch_moe.py:19-25,97-109: W1 + GELU, W2 * gate probability), 45 % of that mode's step on the 128 x 128 tile kernel of
// csrc/gemm3.hip, whose matrix pipes wait on LDS fragment reads of BOTH operands (16 KiB per wave and K block for 48 MFMAs).
//
// Same structure as csrc/gemm_stream.hip: one workgroup = 8 waves = RT x 16 rows x 512 columns, the waves split the columns;
//   * the weights never touch LDS: wave w streams the (hi, lo) fragment PAIRS of its 64 columns global -> registers from a packed
//     stream (mdm_gemm_stream3x_pack: per group [N / 16][K / 32] pairs of 1-KiB MFMA fragments) through an 8-fragment ring (one K step ahead); a pair
//     feeds 3 RT MFMAs;
//   * the pre-split activation rows are the shared operand: K slices of 128 (RT x 8 KiB of hi | lo blocks per slice, two buffers)
//     by LDS-DMA with the 16-B-chunk XOR swizzle on the source side, one workgroup barrier per slice; a fragment is two LDS reads
//     and no arithmetic (hi k = chunk fq of the 32-column block, lo = chunk 4 + fq);
//   * per 32 k and output tile the three products in the tile kernel's order (W_hi A_lo, W_lo A_hi, W_hi A_hi), k ascending: an
//     output element is bit-identical to csrc/gemm3.hip's;
//   * epilogue staged through LDS as fp32 rows (bias, activation, column / row scales, residuals), results as fp32 and / or as
//     pre-split rows for the next GEMM.
#include "gemm.h"
#include "kernels.h"

namespace mdm {
namespace {

constexpr int G3_NT = 512, G3_KS = 128, G3_NR = 8, G3_NJ = 4, G3_SKEW = 1;  // ring of 8 fragments = the 4 pairs of one K step (252 MFMA issue slots ahead); 16 spills 70 - 85 registers

struct G3Args {
  const uint16_t* A;  // pre-split rows; lda in 16-bit elements (2 K for dense rows)
  int64_t lda;
  const int32_t* gather;
  int64_t M;
  int N, K;
  const int32_t* goff;
  int ngroups;
  const uint16_t* ws;
  int64_t ws_gs;  // elements per group
  const float* bias;
  int64_t bias_bs;
  float alpha, out_scale, r1_scale;
  const float* colscale;
  const float* rowscale;
  const float* R1;
  int64_t ldr1;
  const float* R2;
  int64_t ldr2;
  float* C;
  uint16_t* Cx2;
  int64_t ldc;
};

// LDS-DMA from inline assembly (see csrc/gemm_stream.hip: with the builtin hipcc drains the weight ring in front of every slice)
__device__ __forceinline__ void g3_glds16(const void* g, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds) : "memory", "m0");
}

constexpr int g3_smem(int rt) {  // two slice buffers + the tile's source-row table; the epilogue's fp32 staging wants whole tiles up to 160 KiB
  const int ring = 2 * 16 * rt * 512 + 16 * rt * 4, stage = 16 * rt * 2048;
  return stage <= ring ? ring : (stage < 160 * 1024 ? stage : 160 * 1024);
}

// row range of logical row tile rt: groups are cut into ceil(rows / ROWS) tiles; lane e of the wave looks at group e (<= 64 groups)
__device__ __forceinline__ bool g3_find(const G3Args& g, int rows, int rt, int lane, int64_t& row0, int64_t& row_end, int& grp) {
  if (!g.goff) {
    row0 = (int64_t)rt * rows, row_end = row0 + rows < g.M ? row0 + rows : g.M, grp = 0;
    return row0 < g.M;
  }
  const int ng = g.ngroups;
  const int e = lane < ng ? lane : ng - 1;
  const int b = g.goff[e], en = g.goff[e + 1];
  const int t = lane < ng ? (en - b + rows - 1) / rows : 0;
  int incl = t;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(incl, d, 64);
    if (lane >= d) incl += v;
  }
  const unsigned long long m = __ballot(rt < incl);
  if (m == 0) return false;
  const int ge = __ffsll((long long)m) - 1;
  const int gb = __builtin_amdgcn_readlane(b, ge), gen = __builtin_amdgcn_readlane(en, ge);
  const int gt = __builtin_amdgcn_readlane(t, ge), gi = __builtin_amdgcn_readlane(incl, ge);
  row0 = gb + (int64_t)(rt - (gi - gt)) * rows;
  row_end = row0 + rows < gen ? row0 + rows : gen;
  grp = ge;
  return row0 < row_end;
}

// NSL = K / 128 as a template argument (fully unrolled slice loop: csrc/gemm_stream.hip says why)
template <int RT, int NSL, int ACT>
__global__ __launch_bounds__(G3_NT, 2) void gemm_stream3_kernel(const G3Args g) {
  constexpr int NJ = G3_NJ, ROWS = 16 * RT, SLICE = ROWS * 512, NR = G3_NR, AHEAD = NR / (2 * NJ);  // K steps the ring runs ahead
  constexpr int SMEM = g3_smem(RT), ks32 = NSL * 4;
  static_assert(NR % (2 * NJ) == 0 && (4 * 2 * NJ) % NR == 0, "ring slots must be compile-time per slice");
  typedef HB::frag_t frag_t;
  extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
  int tid = threadIdx.x;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ncb = g.N / (128 * NJ);
  const int lt = xcd_remap(blockIdx.x, gridDim.x);  // consecutive logical tiles (one group's) on one XCD: its weights stay in that L2
  const int rt = lt / ncb, cb = lt - rt * ncb;
  int64_t row0, row_end;
  int grp;
  if (!g3_find(g, ROWS, rt, tid & 63, row0, row_end, grp)) return;

  // this wave's NJ pair streams (one per 16 columns): fragment (j, s, h) at wp + j * jst + s * 2048 + h * 1024
  const int cg0 = (cb * 8 + wn) * NJ;
  const int64_t jst = (int64_t)(ks32 + G3_SKEW) * 2048;
  const uint8_t* wp = (const uint8_t*)(g.ws + grp * g.ws_gs) + cg0 * jst + (tid & 63) * 16;

  // X slice sl -> buffer: RT x 8 KiB; one LDS-DMA instruction fills 1 KiB = two rows of 512 B (4 blocks of 64 B hi | 64 B lo);
  // physical 16-B slot p of row r holds chunk p ^ (r & 15)
  // source row of every tile row, once, into LDS behind the slice buffers (rows past the tile's end: copies of its last row, never
  // stored).  Read per slice from LDS and not from global memory: a gather index fetched inside the K loop would be the youngest
  // load in flight, and waiting for it (in-order counter) drains the weight ring once per slice.
  int* const srow = (int*)(smem + 2 * SLICE);
  if (tid < ROWS) {
    int64_t row = row0 + tid;
    row = row < row_end ? row : row_end - 1;
    srow[tid] = g.gather ? g.gather[row] : (int)row;
  }
  __syncthreads();
  auto issue = [&](int sl, int buf) {
    const uint32_t dst = (uint32_t)(uintptr_t)smem + buf * SLICE + wn * 1024;
    const int lane = tid & 63, slot = lane & 31;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const int r = 2 * (wn + 8 * t) + (lane >> 5);
      g3_glds16(g.A + (int64_t)srow[r] * g.lda + ((slot ^ (r & 15)) << 3) + sl * (2 * G3_KS), dst + t * 8192);
    }
  };
  issue(0, 0);
  frag_t R[NR];
#pragma unroll
  for (int f = 0; f < NR; ++f) R[f] = *(const frag_t*)(wp + ((f >> 1) % NJ) * jst + ((f >> 1) / NJ) * 2048 + (f & 1) * 1024);
  wp += AHEAD * 2048;

  f32x4 y[RT][NJ];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) y[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int sl = 0; sl < NSL; ++sl) {
    // my pieces of slice sl have landed (everything older than the ring's NR most recent loads has), then everybody's; behind the
    // barrier every wave is also done with slice sl - 1, whose buffer takes slice sl + 1 (unconditionally: the last pass re-reads
    // its own slice into the idle buffer)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NR) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(sl + 1 < NSL ? sl + 1 : sl, (sl + 1) & 1);
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, frow = lane & 15, fq = lane >> 4;
    const uint8_t* xb = smem + (sl & 1) * SLICE + frow * 512;
    frag_t Ah[RT], Al[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      Ah[i] = *(const frag_t*)(xb + i * 8192 + ((fq ^ frow) << 4));
      Al[i] = *(const frag_t*)(xb + i * 8192 + (((4 + fq) ^ frow) << 4));
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int slot = ((s * NJ + j) * 2) % NR;
        // the three products of one (W pair, row tile) in the tile kernel's order; the row tiles interleaved so that consecutive
        // MFMAs never wait on each other's accumulator
#pragma unroll
        for (int i = 0; i < RT; ++i) y[i][j] = HB::mfma16(R[slot], Al[i], y[i][j]);
        if (j == NJ - 1 && s + 1 < 4) {  // next K step's lo fragments into the registers that just died
#pragma unroll
          for (int i = 0; i < RT; ++i) Al[i] = *(const frag_t*)(xb + i * 8192 + (((8 * (s + 1) + 4 + fq) ^ frow) << 4));
        }
#pragma unroll
        for (int i = 0; i < RT; ++i) y[i][j] = HB::mfma16(R[slot + 1], Ah[i], y[i][j]);
#pragma unroll
        for (int i = 0; i < RT; ++i) y[i][j] = HB::mfma16(R[slot], Ah[i], y[i][j]);
        if (j == NJ - 1 && s + 1 < 4) {
#pragma unroll
          for (int i = 0; i < RT; ++i) Ah[i] = *(const frag_t*)(xb + i * 8192 + (((8 * (s + 1) + fq) ^ frow) << 4));
        }
        R[slot] = *(const frag_t*)(wp + j * jst + s * 2048);  // pair (j, s + AHEAD); the last AHEAD steps read the next stream / the pad
        R[slot + 1] = *(const frag_t*)(wp + j * jst + s * 2048 + 1024);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    wp += 4 * 2048;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring's read-ahead and the last (idle) slice copy: nothing may land in LDS after exit

  // ---- epilogue: EPT row tiles at a time staged as fp32 [16 EPT][512] in LDS (16-B chunks XOR-swizzled by the row), then whole rows
  const int lane = tid & 63, frow = lane & 15, fq = lane >> 4;
  const float* bias = g.bias ? g.bias + grp * g.bias_bs : nullptr;
  f32x4 bb[NJ], cs[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = (cg0 + j) * 16 + 4 * fq;
    bb[j] = bias ? *(const f32x4*)(bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 q = g.colscale ? *(const f32x4*)(g.colscale + n) : (f32x4){1.f, 1.f, 1.f, 1.f};
    cs[j] = (f32x4){g.out_scale * q[0], g.out_scale * q[1], g.out_scale * q[2], g.out_scale * q[3]};
  }
  float rsv[RT];  // row scales of this lane's RT rows, requested together
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    int64_t m = row0 + i * 16 + frow;
    m = m < row_end ? m : row_end - 1;
    rsv[i] = g.rowscale ? g.rowscale[m] : 1.f;
  }
  constexpr int NB = 128 * NJ, EPT = (SMEM / (64 * NB)) < RT ? (SMEM / (64 * NB)) : RT, NPASS = (RT + EPT - 1) / EPT;
  constexpr int TPR = NB / 4, RPS = G3_NT / TPR, NK = 16 * EPT / RPS;  // threads per row, rows per step, steps per pass
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
        const float rs = rsv[i];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          f32x4 v;
          if constexpr (ACT == ACT_GELU) {
            const f32x2 a = gelu_erf2((f32x2){g.alpha * (y[i][j][0] + bb[j][0]), g.alpha * (y[i][j][1] + bb[j][1])});
            const f32x2 b = gelu_erf2((f32x2){g.alpha * (y[i][j][2] + bb[j][2]), g.alpha * (y[i][j][3] + bb[j][3])});
            v = (f32x4){a[0], a[1], b[0], b[1]};
          } else {
            v = (f32x4){g.alpha * (y[i][j][0] + bb[j][0]), g.alpha * (y[i][j][1] + bb[j][1]), g.alpha * (y[i][j][2] + bb[j][2]), g.alpha * (y[i][j][3] + bb[j][3])};
          }
          v[0] *= cs[j][0] * rs, v[1] *= cs[j][1] * rs, v[2] *= cs[j][2] * rs, v[3] *= cs[j][3] * rs;
          *(f32x4*)(stg + ml * NB + (((wn * 4 * NJ + 4 * j + fq) ^ (ml & 31)) << 2)) = v;
        }
      }
    }
    __syncthreads();
    if (!g.R1 && !g.R2) {
      // no residuals (both expert launches): nothing is loaded between the stores -- on gfx950 stores count in vmcnt too, so a wait for
      // a residual row (even one that a null pointer skips) is a wait for every store issued before it: the general path below
      // serialised the tile's 40 store batches on the write latency (80 of 199 us at 50 176 x 1024 x 512)
      constexpr int KS = 4;  // staged rows read together, then stored
      static_assert(NK % KS == 0, "row batches");
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
          if (p * EPT * 16 + ml >= 16 * RT || m >= row_end) continue;  // (whole rows: both lanes of a column pair agree)
          if (g.C) *(f32x4*)(g.C + m * g.ldc + n) = o[k];
          if (g.Cx2) store_x2_4p(g.Cx2 + m * 2 * g.ldc, n, o[k][0], o[k][1], o[k][2], o[k][3]);
        }
      }
      continue;
    }
    constexpr int KB = 2;
#pragma unroll
    for (int k0 = 0; k0 < NK; k0 += KB) {
      f32x4 v[KB], q1[KB], q2[KB];
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        const int ml = rq + RPS * (k0 + k);
        int64_t m = row0 + p * EPT * 16 + ml;
        m = m < row_end ? m : row_end - 1;
        v[k] = *(const f32x4*)(stg + ml * NB + ((cl ^ (ml & 31)) << 2));
        q1[k] = (f32x4){0.f, 0.f, 0.f, 0.f}, q2[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (g.R1) q1[k] = *(const f32x4*)(g.R1 + m * g.ldr1 + n);
        if (g.R2) q2[k] = *(const f32x4*)(g.R2 + m * g.ldr2 + n);
      }
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        const int ml = rq + RPS * (k0 + k);
        const int64_t m = row0 + p * EPT * 16 + ml;
        if (k0 + k >= NK || p * EPT * 16 + ml >= 16 * RT || m >= row_end) continue;  // (whole rows: both lanes of a column pair agree)
        f32x4 o = v[k];
        o[0] += g.r1_scale * q1[k][0] + q2[k][0], o[1] += g.r1_scale * q1[k][1] + q2[k][1];
        o[2] += g.r1_scale * q1[k][2] + q2[k][2], o[3] += g.r1_scale * q1[k][3] + q2[k][3];
        if (g.C) *(f32x4*)(g.C + m * g.ldc + n) = o;
        if (g.Cx2) store_x2_4p(g.Cx2 + m * 2 * g.ldc, n, o[0], o[1], o[2], o[3]);
      }
    }
  }
}

// stream3x[group][cg = n / 16][s = k / 32][hi | lo][lane l][8]:  bf16 hi / lo of W[group][16 cg + (l & 15)][32 s + 8 (l >> 4) + e]
__global__ __launch_bounds__(256) void gemm_stream3x_pack_kernel(const float* w, int64_t ldw, int G, int N, int K, uint16_t* out, int64_t gs) {
  const int ks32 = K >> 5, nf = (N >> 4) * ks32;
  for (int64_t fi = blockIdx.x * 4 + (threadIdx.x >> 6); fi < (int64_t)G * nf; fi += gridDim.x * 4) {
    const int l = threadIdx.x & 63, grp = (int)(fi / nf), f = (int)(fi - (int64_t)grp * nf), cg = f / ks32, s = f - cg * ks32;
    const float* src = w + ((int64_t)grp * N + 16 * cg + (l & 15)) * ldw + 32 * s + 8 * (l >> 4);
    uint4 h, lo;
    split_bf16(src[0], src[1], h.x, lo.x);
    split_bf16(src[2], src[3], h.y, lo.y);
    split_bf16(src[4], src[5], h.z, lo.z);
    split_bf16(src[6], src[7], h.w, lo.w);
    uint16_t* dst = out + grp * gs + ((int64_t)cg * (ks32 + G3_SKEW) + s) * 1024 + l * 8;
    *(uint4*)dst = h;
    *(uint4*)(dst + 512) = lo;
  }
}

constexpr int64_t G3_PAD = 8 * 512;  // elements behind the last group's last pair: what the ring reads ahead of the last K step

template <int RT, int NSL>
int launch_g3(const G3Args& g, int act, int tiles, hipStream_t s) {
  constexpr int smem = g3_smem(RT);
  static DevOnce attr;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)gemm_stream3_kernel<RT, NSL, ACT_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess ||
        hipFuncSetAttribute((const void*)gemm_stream3_kernel<RT, NSL, ACT_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return MDM_ERR_LAUNCH;
    attr = true;
  }
  const dim3 grid((unsigned)tiles);
  if (act == ACT_GELU) hipLaunchKernelGGL((gemm_stream3_kernel<RT, NSL, ACT_GELU>), grid, dim3(G3_NT), smem, s, g);
  else hipLaunchKernelGGL((gemm_stream3_kernel<RT, NSL, ACT_NONE>), grid, dim3(G3_NT), smem, s, g);
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

}  // namespace

extern int g_bf16_variant;

int64_t gemm_stream3x_group_elems(int N, int K) {
  return (N > 0 && K > 0 && N % 512 == 0 && K % 128 == 0) ? (int64_t)(N / 16) * (K / 32 + G3_SKEW) * 1024 : 0;
}
int64_t gemm_stream3x_elems(int G, int N, int K) { return G > 0 && gemm_stream3x_group_elems(N, K) ? G * gemm_stream3x_group_elems(N, K) + G3_PAD : 0; }

int gemm_stream3x_pack(const float* w, int64_t ldw, int G, int N, int K, uint16_t* out, hipStream_t stream) {
  const int64_t n = gemm_stream3x_elems(G, N, K);
  if (!w || !out || !n || ldw < K) return MDM_ERR_UNSUPPORTED;
  if (hipMemsetAsync(out, 0, n * sizeof(uint16_t), stream) != hipSuccess) return MDM_ERR_LAUNCH;
  hipLaunchKernelGGL(gemm_stream3x_pack_kernel, dim3(512), dim3(256), 0, stream, w, ldw, G, N, K, out, gemm_stream3x_group_elems(N, K));
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

// pre-split rows x a (hi, lo) pair stream; dense or grouped (<= 64 groups), optional row gather
bool gemm_stream3x_eligible(const GemmArgs& a) {
  if (!a.w_stream || a.precision != 3 || a.A.kind != OP_X2_ROW || a.A.rpg || a.batch != 1 || a.nb2 != 1 || a.kgoff) return false;
  if (a.act != ACT_NONE && a.act != ACT_GELU) return false;
  if (a.r1_mod || a.C8 || a.C16 || a.C16_lo || a.a_scale || a.w_scale || (!a.C && !a.Cx2)) return false;
  if ((a.K != 512 && a.K != 1024) || a.N % 512 || (a.A.ld & 7) || (a.ldc & 3) || (a.R1 && (a.ldr1 & 3)) || (a.R2 && (a.ldr2 & 3))) return false;
  if (a.goff ? (a.ngroups <= 0 || a.ngroups > 64 || a.w_stream_gs != gemm_stream3x_group_elems(a.N, a.K) || (a.bias && (a.bias_bs & 3))) : false) return false;
  if ((((uintptr_t)a.A.p) | ((uintptr_t)a.w_stream) | ((uintptr_t)a.C) | ((uintptr_t)a.Cx2) | ((uintptr_t)a.R1) | ((uintptr_t)a.R2) |
       ((uintptr_t)a.bias) | ((uintptr_t)a.colscale)) & 15)
    return false;
  return a.M > 0 && (int64_t)a.M * a.N < (1ll << 40);
}

// Measured against the tile kernel (tools/x3_stream_bench.py, 16 balanced groups, same process, profiles/r04_x3_stream_bench.txt):
// K = 1024 (expert W2) 146 vs 183 us at 50 176 rows, 70 vs 92 us at 25 088 = 0.43 - 0.45 of the 833 TFLOP/s a bf16x3 product stream
// can reach; K = 512 (W1 + GELU -> pre-split rows) 190 vs 206 us and 89 vs 104 us (0.33 - 0.36: with 16 K steps per tile the
// prologue and the staged GELU epilogue, which one workgroup per CU cannot overlap with the next tile, are a third of the tile).
// Taken wherever it is eligible.
bool gemm_stream3x_wanted(const GemmArgs& a) { return gemm_stream3x_eligible(a); }

int gemm_stream3x(const GemmArgs& a, hipStream_t stream) {
  if (!gemm_stream3x_eligible(a)) return MDM_ERR_UNSUPPORTED;
  G3Args g = {};
  g.A = (const uint16_t*)a.A.p, g.lda = 2 * a.A.ld, g.gather = a.A.gather, g.M = a.M, g.N = a.N, g.K = a.K;
  g.goff = a.goff, g.ngroups = a.ngroups, g.ws = a.w_stream, g.ws_gs = a.goff ? a.w_stream_gs : 0;
  g.bias = a.bias, g.bias_bs = a.goff ? a.bias_bs : 0;
  g.alpha = a.alpha, g.out_scale = a.out_scale, g.r1_scale = a.r1_scale, g.colscale = a.colscale, g.rowscale = a.rowscale;
  g.R1 = a.R1, g.ldr1 = a.ldr1, g.R2 = a.R2, g.ldr2 = a.ldr2, g.C = a.C, g.Cx2 = a.Cx2, g.ldc = a.ldc;
  constexpr int RT = 7;
  const int64_t rtiles = (a.M + 16 * RT - 1) / (16 * RT) + (a.goff ? a.ngroups : 0);  // upper bound: a partial tile per group
  const int tiles = (int)(rtiles * (a.N / 512));
  return a.K == 512 ? launch_g3<RT, 4>(g, a.act, tiles, stream) : launch_g3<RT, 8>(g, a.act, tiles, stream);
}

}  // namespace mdm
