// Throughput GEMM for gfx950: bf16 activations x bf16 weights, fp32 accumulate.
//   * both operands stream global -> LDS with global_load_lds_dwordx4 (no VGPR staging, no conversion): each
//     wave-instruction lands 8 rows x 128 B; the LDS image stays lane-linear and the 16-B chunks of a row are
//     XOR-swizzled on the SOURCE address (chunk ^ (row & 7)), the same involution is applied on the fragment read,
//     which makes every ds_read_b128 of a 16x16x32 fragment conflict-free;
//   * 128x128x64 tile, 4 waves (2x2), NSTAGE-deep ring, one barrier per K-tile, 2 blocks per CU;
//   * MFMA operands are swapped (W fragment as A, activation fragment as B) so each lane ends up with 4 CONSECUTIVE
//     output columns of one row: 16-byte stores / residual loads in the epilogue instead of 4-byte ones.
// Row gather (MoE expert inputs), grouped and batched modes as in gemm.hip.
#include "gemm.h"

namespace mdm {
namespace {

constexpr int BN = 128, NT = 256;

__device__ __forceinline__ void glds16(const void* g, uint8_t* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__device__ unsigned long long g_stamps[16];
#define MDM_STAMP(i)                                                             \
  do {                                                                       \
    if (dbg) {                                                               \
      unsigned long long t__;                                                \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory"); \
      if (threadIdx.x == 0) g_stamps[i] = t__;                               \
    }                                                                        \
  } while (0)

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS image of one operand tile: 128 rows x (2*BK) bytes, 16-B chunks swizzled so that the ds_read_b128 of a
// 16x16x32 fragment (16 rows x one chunk per 16-lane group) is conflict-free:
//   BK = 64 (128-B rows, 8 chunks):  phys = chunk ^ (row & 7)
//   BK = 32 ( 64-B rows, 4 chunks):  phys = (chunk + 2 * (row >> 2)) & 3
template <int BK>
__device__ __forceinline__ int chunk_phys(int row, int c) {
  if constexpr (BK == 64) return c ^ (row & 7);
  return (c + 2 * (row >> 2)) & 3;
}
template <int BK>
__device__ __forceinline__ int chunk_logical(int row, int phys) {
  if constexpr (BK == 64) return phys ^ (row & 7);
  return (phys - 2 * (row >> 2)) & 3;
}

template <typename HT, int BM, int BK, int NSTAGE, int ACT>
__global__ __launch_bounds__(NT, 2) void gemm_bf16_kernel(const GemmArgs g) {
  typedef typename HT::frag_t frag_t;
  extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
  constexpr int ROWB = 2 * BK;             // bytes per LDS row
  constexpr int TILE_A = BM * ROWB, TILE_W = BN * ROWB;  // bytes per operand per stage
  constexpr int STAGE_B = TILE_A + TILE_W;
  constexpr int RPP = 1024 / ROWB;         // rows per 1-KiB LDS-DMA piece (8 or 16)
  constexpr int PPA = BM / RPP / 4;        // A pieces per wave per stage
  constexpr int PPWW = BN / RPP / 4;       // W pieces per wave per stage
  constexpr int MI = BM / 32;              // 16-row fragments per wave along M (waves 2x2)
  constexpr int CPR = ROWB / 16;           // 16-B chunks per row
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const bool dbg = (g.feat_S == -77) && blockIdx.x == 0;
  MDM_STAMP(0);

  const int ntn = (g.N + BN - 1) / BN;
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
  const int z = blockIdx.z, z1 = z / g.nb2, z2 = z % g.nb2;
  const int64_t offA = (int64_t)z1 * g.A.bs1 + (int64_t)z2 * g.A.bs2;
  const int64_t offW = g.goff ? (int64_t)grp * g.W.bs1 : (int64_t)z1 * g.W.bs1 + (int64_t)z2 * g.W.bs2;
  const int64_t offC = (int64_t)z1 * g.c_bs1 + (int64_t)z2 * g.c_bs2;
  const int64_t offB = g.goff ? (int64_t)grp * g.bias_bs : (int64_t)z * g.bias_bs;

  // per-lane source rows of this wave's LDS-DMA pieces (RPP rows x ROWB bytes each); rows past the edge are
  // clamped to a valid row (their products are discarded in the epilogue): pad, don't mask
  const int sub = lane / CPR;  // row inside the piece
  const uint16_t* pa[PPA];
  const uint16_t* pw[PPWW];
#pragma unroll
  for (int i = 0; i < PPA; ++i) {
    const int tr = (wid * PPA + i) * RPP + sub;  // row inside the tile
    const int koff = chunk_logical<BK>(tr, lane % CPR) * 8;
    int r = row0 + tr;
    r = r < row_end ? r : row_end - 1;
    const int64_t src = g.A.gather ? (int64_t)g.A.gather[r] : (int64_t)r;
    pa[i] = (const uint16_t*)g.A.p + offA + src * g.A.ld + koff;
  }
#pragma unroll
  for (int i = 0; i < PPWW; ++i) {
    const int tr = (wid * PPWW + i) * RPP + sub;
    const int koff = chunk_logical<BK>(tr, lane % CPR) * 8;
    int n = nt * BN + tr;
    n = n < g.N ? n : g.N - 1;
    pw[i] = (const uint16_t*)g.W.p + offW + (int64_t)n * g.W.ld + koff;
  }
  auto stage = [&](int kt, int buf) {
    uint8_t* sa = smem + buf * STAGE_B + wid * PPA * 1024;
    uint8_t* sw = smem + buf * STAGE_B + TILE_A + wid * PPWW * 1024;
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < PPA; ++i) glds16(pa[i] + k0, sa + i * 1024);
#pragma unroll
    for (int i = 0; i < PPWW; ++i) glds16(pw[i] + k0, sw + i * 1024);
  };

  // the first K tiles go out BEFORE anything else touches the memory pipeline: the epilogue constants below are not
  // needed for ~20 K cycles, the tiles are needed at once
  const int nk = g.K / BK;
  MDM_STAMP(1);
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) stage(s, s);
  MDM_STAMP(2);

  // column / row constants of this lane's outputs, fetched before the K loop (they ride in 40 registers)
  float* __restrict__ C = g.C ? g.C + offC : nullptr;
  uint16_t* __restrict__ C16 = g.C16 ? g.C16 + offC : nullptr;
  const float* __restrict__ bias = g.bias ? g.bias + offB : nullptr;
  const float* __restrict__ colscale = g.colscale;
  const float* __restrict__ R1 = g.R1;
  const float* __restrict__ R2 = g.R2;
  const int fq = lane >> 4;
  const int nbase = nt * BN + wn * 64 + fq * 4;
  float bv[4][4], cv[4][4], rs[MI];
  bool keymask[MI];
  const bool vec_n = nt * BN + BN <= g.N && (!bias || ((((uintptr_t)bias) & 15) == 0 && (offB & 3) == 0)) &&
                     (!colscale || (((uintptr_t)colscale) & 15) == 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (vec_n) {  // four consecutive columns per lane: one 16-byte load each instead of four dword loads
      const f32x4 b4 = bias ? *(const f32x4*)(bias + nbase + j * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
      const f32x4 c4 = colscale ? *(const f32x4*)(colscale + nbase + j * 16) : (f32x4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[j][r] = b4[r], cv[j][r] = g.out_scale * c4[r];
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nbase + j * 16 + r;
        const int nn = n < g.N ? n : g.N - 1;
        bv[j][r] = bias ? bias[nn] : 0.f;
        cv[j][r] = g.out_scale * (colscale ? colscale[nn] : 1.f);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = row0 + wm * (BM / 2) + i * 16 + (lane & 15);
    const bool ok = m < row_end;
    rs[i] = (ok && g.rowscale) ? g.rowscale[m] : 1.f;
    keymask[i] = false;
    if (ok && ACT == ACT_FEAT && g.feat_len) {
      const int tok = m / g.feat_rpt, slot = m - tok * g.feat_rpt;
      if (slot >= g.feat_kslot) {
        const int bb = tok / g.feat_S, t = tok - bb * g.feat_S;
        keymask[i] = t >= g.feat_len[bb];
      }
    }
  }

  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr int PIECES = PPA + PPWW;  // LDS-DMA instructions per wave per K-tile
  const int frow = lane & 15;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed once each wave's pieces for it have: wait for all but the younger tiles, then barrier
    const int younger = min(NSTAGE - 2, nk - 1 - kt);
    if (younger >= 2) {
      wait_vm<2 * PIECES>();
    } else if (younger == 1) {
      wait_vm<PIECES>();
    } else {
      wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (kt == 0) MDM_STAMP(3);
    if (kt + NSTAGE - 1 < nk) stage(kt + NSTAGE - 1, (kt + NSTAGE - 1) % NSTAGE);
    const uint8_t* sa = smem + (kt % NSTAGE) * STAGE_B;
    const uint8_t* sw = sa + TILE_A;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      frag_t a[MI], b[4];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int ra = wm * (BM / 2) + i * 16 + frow;
        a[i] = *(const frag_t*)(sa + ra * ROWB + (chunk_phys<BK>(ra, ks * 4 + fq) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rb = wn * 64 + j * 16 + frow;
        b[j] = *(const frag_t*)(sw + rb * ROWB + (chunk_phys<BK>(rb, ks * 4 + fq) << 4));
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = HT::mfma16(b[j], a[i], acc[i][j]);  // D = W A^T: (n, m)
    }
  }

  MDM_STAMP(4);
  // Epilogue in two passes through the (now idle) LDS ring:
  //  1. registers -> LDS: bias, activation, scales applied; D tile (j,i) lane holds n = nb + 4*(lane>>4) + r for
  //     m = mb + (lane & 15) -> one 16-B write, 16-B chunks of a row XOR-swizzled by the row index;
  //  2. LDS -> HBM by full rows (a wave instruction covers two whole 512-B rows): residuals are read and outputs
  //     written as complete cache lines instead of 64-B pieces of 16 different rows.
  // residual rows of pass 2 are requested NOW (the accumulators' registers are about to die): their latency hides
  // behind pass 1 and the barrier instead of being paid once per row group
  const bool vec = ((g.ldc & 3) == 0) && (!R1 || (g.ldr1 & 3) == 0) && (!R2 || (g.ldr2 & 3) == 0);
  const int cl = tid & 31, n = nt * BN + 4 * cl;
  const bool fast = vec && (nt * BN + BN <= g.N);
  constexpr int NR = BM / 8;
  f32x4 q1[NR], q2[NR];
  if (fast) {
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int m = row0 + (tid >> 5) + 8 * k;
      q1[k] = (f32x4){0.f, 0.f, 0.f, 0.f}, q2[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (m < row_end) {
        const int64_t mr = g.r1_mod ? (m % g.r1_mod) : m;
        if (R1) q1[k] = *(const f32x4*)(R1 + mr * g.ldr1 + n);
        if (R2) q2[k] = *(const f32x4*)(R2 + (int64_t)m * g.ldr2 + n);
      }
    }
  }
  __syncthreads();  // every wave is done with the last K tile
  MDM_STAMP(6);
  float* stg = (float*)smem;  // [BM][128] fp32
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int ml = wm * (BM / 2) + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 v;
      f32x2 ge[2];
      if constexpr (ACT == ACT_GELU) {  // packed-fp32 evaluation, two columns per instruction
        ge[0] = gelu_erf2((f32x2){g.alpha * (acc[i][j][0] + bv[j][0]), g.alpha * (acc[i][j][1] + bv[j][1])});
        ge[1] = gelu_erf2((f32x2){g.alpha * (acc[i][j][2] + bv[j][2]), g.alpha * (acc[i][j][3] + bv[j][3])});
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = g.alpha * (acc[i][j][r] + bv[j][r]);
        if constexpr (ACT == ACT_GELU) {
          x = ge[r >> 1][r & 1];
        } else if constexpr (ACT == ACT_SILU) {
          x = silu(x);
        } else if constexpr (ACT == ACT_FEAT) {
          x = keymask[i] ? 0.f : 0.1f * expf(fminf(fmaxf(x, -15.f), 15.f));
        }
        v[r] = x * (cv[j][r] * rs[i]);
      }
      const int chunk = wn * 16 + j * 4 + fq;
      *(f32x4*)(stg + ml * 128 + ((chunk ^ (ml & 31)) << 2)) = v;
    }
  }
  __syncthreads();
  MDM_STAMP(7);
  if (fast) {
    // all staged rows out of LDS first (their reads need no row test): behind the per-row `continue` every read was its own block,
    // `R w S | S |` sixteen times -- one exposed LDS latency per row
    f32x4 vv[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int ml = (tid >> 5) + 8 * k;
      vv[k] = *(const f32x4*)(stg + ml * 128 + ((cl ^ (ml & 31)) << 2));
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int ml = (tid >> 5) + 8 * k, m = row0 + ml;
      if (m >= row_end) continue;
      f32x4 v = vv[k];
      v[0] += g.r1_scale * q1[k][0] + q2[k][0], v[1] += g.r1_scale * q1[k][1] + q2[k][1];
      v[2] += g.r1_scale * q1[k][2] + q2[k][2], v[3] += g.r1_scale * q1[k][3] + q2[k][3];
      if (C) *(f32x4*)(C + (int64_t)m * g.ldc + n) = v;
      if (C16) *(uint2*)(C16 + (int64_t)m * g.ldc + n) = make_uint2(HT::pack(v[0], v[1]), HT::pack(v[2], v[3]));
    }
  } else {
    for (int k = 0; k < NR; ++k) {  // ragged N edge / unaligned leading dimensions: element-wise
      const int ml = (tid >> 5) + 8 * k, m = row0 + ml;
      if (m >= row_end || n >= g.N) continue;
      const f32x4 v = *(const f32x4*)(stg + ml * 128 + ((cl ^ (ml & 31)) << 2));
      const int64_t mr = g.r1_mod ? (m % g.r1_mod) : m;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (n + r >= g.N) break;
        float x = v[r];
        if (R1) x += g.r1_scale * R1[mr * g.ldr1 + n + r];
        if (R2) x += R2[(int64_t)m * g.ldr2 + n + r];
        if (C) C[(int64_t)m * g.ldc + n + r] = x;
        if (C16) C16[(int64_t)m * g.ldc + n + r] = (uint16_t)(HT::pack(x, 0.f) & 0xffff);
      }
    }
  }
  MDM_STAMP(8);
  if (dbg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MDM_STAMP(5);
  }
}

}  // namespace

int debug_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16) == hipSuccess ? MDM_OK : MDM_ERR_LAUNCH;
}

int g_big_min_tiles = 352;
int g_bf16_variant = 0;  // tuning knob (mdm_set_gemm_variant): 0 = default

bool gemm_bf16_eligible(const GemmArgs& a) {
  return a.precision == 1 && a.A.kind == OP_BF16_ROW && a.W.kind == OP_BF16_ROW && a.K > 0 && (a.K % 64) == 0 &&
         (a.A.ld % 8) == 0 && (a.W.ld % 8) == 0 && (a.A.bs1 % 8) == 0 && (a.A.bs2 % 8) == 0 &&
         ((((uintptr_t)a.A.p) | ((uintptr_t)a.W.p)) & 15) == 0 && a.A.rpg == 0;
}

template <typename HT, int BM, int BK, int NS, int ACT>
static int launch_bf16_act(const GemmArgs& a, hipStream_t stream) {
  constexpr int smem = NS * (BM + BN) * 2 * BK;
  static DevOnce attr_set;
  if (smem > 65536 && !attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_bf16_kernel<HT, BM, BK, NS, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            smem) != hipSuccess)
      return MDM_ERR_LAUNCH;
    attr_set = true;
  }
  const int tm = (a.M + BM - 1) / BM + (a.goff ? a.ngroups : 0);
  const int tn = (a.N + BN - 1) / BN;
  dim3 grid((unsigned)(tm * tn), 1, (unsigned)a.batch);
  hipLaunchKernelGGL((gemm_bf16_kernel<HT, BM, BK, NS, ACT>), grid, dim3(NT), smem, stream, a);
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

template <typename HT, int BM, int BK, int NS>
static int launch_bf16_h(const GemmArgs& a, hipStream_t stream) {
  switch (a.act) {
    case ACT_NONE: return launch_bf16_act<HT, BM, BK, NS, ACT_NONE>(a, stream);
    case ACT_GELU: return launch_bf16_act<HT, BM, BK, NS, ACT_GELU>(a, stream);
    case ACT_SILU: return launch_bf16_act<HT, BM, BK, NS, ACT_SILU>(a, stream);
    case ACT_FEAT: return launch_bf16_act<HT, BM, BK, NS, ACT_FEAT>(a, stream);
    default: return MDM_ERR_ARG;
  }
}
// h16 == MDM_H16_F16: the 16-bit operands (and the optional 16-bit output copy) are IEEE fp16 instead of bf16
template <int BM, int BK, int NS>
static int launch_bf16(const GemmArgs& a, hipStream_t stream) {
  return a.h16 == MDM_H16_F16 ? launch_bf16_h<HF, BM, BK, NS>(a, stream) : launch_bf16_h<HB, BM, BK, NS>(a, stream);
}

int gemm_bf16(const GemmArgs& a, hipStream_t stream) {
  if (!gemm_bf16_eligible(a)) return MDM_ERR_UNSUPPORTED;
  if (!a.C && !a.C16) return MDM_ERR_ARG;
  // long-K, many-tile launches: 256x256 tiles halve the L2->LDS bytes per FLOP (1090 vs 865 TFLOP/s at 8192^3), but
  // with one workgroup per CU their prologue / 4-slab epilogue is not hidden: measured slower than the 128^2 tiles at
  // K = 512 / 1024 (e.g. 50176x1024x512: 142 vs 128 us), so they are used for K >= 2048 only (variant 6 forces, 7 forbids)
  if (g_bf16_variant != 7 && g_bf16_variant != 1 && g_bf16_variant != 2 && gemm_bf16_256_eligible(a)) {
    const int64_t t256 = (int64_t)((a.M + 255) / 256 + (a.goff ? a.ngroups : 0)) * (a.N / 256);
    if (g_bf16_variant == 6 || (a.K >= 2048 && t256 >= g_big_min_tiles)) return gemm_bf16_256(a, stream);
  }
  // few 128x128 tiles => the launch is a latency chain on <= 2 blocks per CU: halve the tile height so that every
  // CU holds 3+ independent blocks (variant 1 / 2 force the 128- / 64-row tile for benchmarking).  The 64-row tile runs a
  // 3-stage ring (72 KiB: still two blocks per CU; K tiles arrive two steps ahead): 5.90 -> 5.79 ms per step; for the
  // 128-row tile a third stage costs the second resident block, and 32-wide K tiles in 3- / 4-deep rings change nothing
  const int64_t tiles128 = (int64_t)((a.M + 127) / 128) * ((a.N + BN - 1) / BN) * a.batch;
  bool small = !a.goff && (tiles128 <= 256 || a.M <= 64);
  if (g_bf16_variant == 1) small = false;
  if (g_bf16_variant == 2) small = true;
  if (small && g_bf16_variant != 28) return launch_bf16<64, 64, 3>(a, stream);  // 3-stage ring: still 2 blocks/CU, tiles arrive 2 ahead (knob 28: 2-stage)
  return small ? launch_bf16<64, 64, 2>(a, stream) : launch_bf16<128, 64, 2>(a, stream);
}

}  // namespace mdm
