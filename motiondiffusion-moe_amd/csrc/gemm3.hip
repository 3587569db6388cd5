// fp32-grade GEMM (bf16x3) with LDS-DMA staging:   C = epilogue( A_f32[M,K] * (W_hi + W_lo)[N,K]^T ),  precision 3.
//
// gfx950 has no TF32 path and its f32 MFMA runs at 1/16 of the bf16 rate, so an fp32-grade product is three bf16 MFMAs on
// hi / lo splits (hi*hi + hi*lo + lo*hi, fp32 accumulate; csrc/gemm.hip).  That kernel stages both operands through
// registers (global -> VGPR -> split -> LDS, one __syncthreads per 32-wide K step): the parity-grade mode ran 3.1x slower than
// the single-pass modes although most of its launches are latency-bound, not MFMA-bound.  Here:
//   * the fp32 activation tile goes global -> LDS by LDS-DMA AS fp32 (128-B rows = 32 k; the same 16-B-chunk XOR swizzle on
//     the DMA source and on the read as gemm2.hip) -- same bytes as a bf16 hi + lo pair, no conversion pass, no staging VGPRs;
//   * the split happens on the FRAGMENT: a lane reads its 8 fp32 (two ds_read_b128), rounds them to bf16 hi and to
//     lo = rn(x - hi) in registers (24 VALU per fragment, shared by 12 MFMAs) -- bit-identical operands to gemm.hip's;
//   * the weight tile arrives as its two pre-split bf16 planes (64-B rows, rotate swizzle), also by LDS-DMA;
//   * 2-stage ring (32 KiB per stage at BM = 128: two workgroups per CU), counted vmcnt + raw barrier, swapped MFMA operands,
//     epilogue staged through LDS and written as whole rows, row gather + grouped mode for the expert GEMMs.
// Results differ from gemm.hip's only by accumulation order (tests/test_gemm_gpu.py compares both with fp64).
// ACT_HEADNORM (the q | k | v projection in front of csrc/perf_attn3.hip): a 128-column tile IS one attention head of one of
// q / k / v, so the row-store loop -- 32 lanes hold a row of the staged tile -- applies the shared LayerNorm over head_dim and
// the L2 normalisation of q and k (fast_attention.py:44-55) and writes the rows as bf16 hi / lo planes: the head_norm launch,
// its fp32 round trip and the eight-fold re-splitting of the rows in the attention core are gone.
#include "gemm.h"

namespace mdm {
namespace {

// tile: BM x BN outputs, waves as 2 (M) x BN / 64 (N), each wave BM / 2 rows x 64 columns: 128-column tiles run 4 waves (two
// workgroups per CU), 256-column tiles 8 waves (one per CU, 3-stage ring: a third fewer operand bytes per output)

typedef __bf16 frag3_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4_3 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16d(const void* g, uint8_t* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm3() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// AX2: the activation rows arrive pre-split (MDM_OP_X2_ROW: per row and 32-k block 64 B of hi + 64 B of lo, the same 128 bytes and the
// same LDS-DMA pattern as the fp32 block): the fragment is two reads and NO arithmetic.  Splitting fp32 fragments in the loop costs
// 11 - 18 % of the kernel (24 VALU instructions per fragment in every wave that reads it: a timing build without them ran 30.1 vs
// 33.8 us at 12544 x 512 x 512 and 166 vs 202 us at the expert W1 shape), and the same row is re-split by every column tile.
template <int BM, int BN3, int NSTAGE, int ACT, bool AX2>
__global__ __launch_bounds__(2 * BN3, 2) void gemm_x3_kernel(const GemmArgs g) {
  constexpr int NWN = BN3 / 64, NW = 2 * NWN, NT3 = 64 * NW;
  extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
  constexpr int TILE_A = BM * 128;              // fp32 [BM][32 k]: 128-B rows, 8 rows per 1-KiB piece
  constexpr int TILE_W = BN3 * 64;              // one bf16 plane [128][32 k]: 64-B rows, 16 rows per piece
  constexpr int STAGE_B = TILE_A + 2 * TILE_W;  // A, W hi, W lo
  constexpr int PPA = BM / 8 / NW, PPW = BN3 / 16 / NW;
  static_assert(PPA >= 1 && PPW >= 1, "every wave stages at least one piece of each operand");
  constexpr int MI = BM / 32;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / NWN, wn = wid % NWN;
  const int ntn = (g.N + BN3 - 1) / BN3;
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

  // LDS-DMA sources.  A piece: 8 rows x 128 B, lane -> row (lane >> 3), physical chunk (lane & 7) holds logical chunk
  // phys ^ (row & 7).  W piece: 16 rows x 64 B, lane -> row (lane >> 2), physical chunk (lane & 3) holds (phys - 2 (row >> 2)) & 3.
  const float* pa[PPA];
  const uint16_t* ph[PPW];
  const uint16_t* pl[PPW];
#pragma unroll
  for (int i = 0; i < PPA; ++i) {
    const int tr = (wid * PPA + i) * 8 + (lane >> 3);
    int r = row0 + tr;
    r = r < row_end ? r : row_end - 1;
    const int64_t src = g.A.gather ? (int64_t)g.A.gather[r] : (int64_t)r;
    pa[i] = (const float*)g.A.p + src * g.A.ld + (((lane & 7) ^ (tr & 7)) << 2);
  }
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int tr = (wid * PPW + i) * 16 + (lane >> 2);
    int n = nt * BN3 + tr;
    n = n < g.N ? n : g.N - 1;
    const int64_t o = offW + (int64_t)n * g.W.ld + ((((lane & 3) - 2 * (tr >> 2)) & 3) << 3);
    ph[i] = (const uint16_t*)g.W.p + o;
    pl[i] = (const uint16_t*)g.W.p_lo + o;
  }
  auto stage = [&](int kt, int buf) {
    uint8_t* s = smem + buf * STAGE_B;
    const int k0 = kt * 32;
#pragma unroll
    for (int i = 0; i < PPA; ++i) glds16d(pa[i] + k0, s + (wid * PPA + i) * 1024);
#pragma unroll
    for (int i = 0; i < PPW; ++i) glds16d(ph[i] + k0, s + TILE_A + (wid * PPW + i) * 1024);
#pragma unroll
    for (int i = 0; i < PPW; ++i) glds16d(pl[i] + k0, s + TILE_A + TILE_W + (wid * PPW + i) * 1024);
  };
  constexpr int PIECES = PPA + 2 * PPW;
  const int nk = g.K / 32;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) stage(s, s);

  // column / row constants of this lane's outputs, requested behind the first K tiles and long before the epilogue needs them
  // (as gemm2.hip: loaded in the epilogue they were 8 + dependent round trips of dword loads per wave)
  const int frow = lane & 15, fq = lane >> 4;
  const float* __restrict__ bias = g.bias ? g.bias + offB : nullptr;
  const float* __restrict__ colscale = g.colscale;
  const int nbase = nt * BN3 + wn * 64 + fq * 4;
  float bv[4][4], cv[4][4], rs[MI];
  const bool vec_n = nt * BN3 + BN3 <= g.N && (!bias || ((((uintptr_t)bias) & 15) == 0 && (offB & 3) == 0)) &&
                     (!colscale || (((uintptr_t)colscale) & 15) == 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (vec_n) {
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
    const int m = row0 + wm * (BM / 2) + i * 16 + frow;
    rs[i] = (m < row_end && g.rowscale) ? g.rowscale[m] : 1.f;
  }

  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int kt = 0; kt < nk; ++kt) {
    const int younger = min(NSTAGE - 2, nk - 1 - kt);  // stages issued after this one that may stay in flight
    if (younger >= 2) {
      wait_vm3<2 * PIECES>();
    } else if (younger == 1) {
      wait_vm3<PIECES>();
    } else {
      wait_vm3<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (kt + NSTAGE - 1 < nk) stage(kt + NSTAGE - 1, (kt + NSTAGE - 1) % NSTAGE);
    const uint8_t* sa = smem + (kt % NSTAGE) * STAGE_B;
    const uint8_t* sh = sa + TILE_A;
    const uint8_t* sl = sh + TILE_W;
    frag3_t bh[4], bl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rb = wn * 64 + j * 16 + frow;
      const int off = rb * 64 + (((fq + 2 * (rb >> 2)) & 3) << 4);
      bh[j] = *(const frag3_t*)(sh + off);
      bl[j] = *(const frag3_t*)(sl + off);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int ra = wm * (BM / 2) + i * 16 + frow;
      frag3_t ah, al;
      if constexpr (AX2) {  // hi k = 8 fq .. 8 fq + 7 = 16-B chunk fq of the row's 128 bytes, lo = chunk 4 + fq (conflict-free as well)
        ah = *(const frag3_t*)(sa + ra * 128 + ((fq ^ (ra & 7)) << 4));
        al = *(const frag3_t*)(sa + ra * 128 + (((4 + fq) ^ (ra & 7)) << 4));
      } else {
        const f32x4 x0 = *(const f32x4*)(sa + ra * 128 + (((2 * fq) ^ (ra & 7)) << 4));
        const f32x4 x1 = *(const f32x4*)(sa + ra * 128 + (((2 * fq + 1) ^ (ra & 7)) << 4));
        uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
        split_bf16(x0[0], x0[1], h0, l0);
        split_bf16(x0[2], x0[3], h1, l1);
        split_bf16(x1[0], x1[1], h2, l2);
        split_bf16(x1[2], x1[3], h3, l3);
        const u32x4_3 uh = {h0, h1, h2, h3}, ul = {l0, l1, l2, l3};
        ah = __builtin_bit_cast(frag3_t, uh), al = __builtin_bit_cast(frag3_t, ul);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // D = W A^T; small terms first, as gemm.hip does
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah, acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue (as gemm2.hip): registers -> LDS [BM][128] fp32 with bias / activation / scales, then whole rows out ------
  float* __restrict__ C = g.C;
  uint16_t* __restrict__ C16 = g.C16;
  const float* __restrict__ R1 = g.R1;
  const float* __restrict__ R2 = g.R2;
  // 256 x 256 tiles stage their fp32 outputs in two passes of 128 rows (the rows of the waves with wm == pass): 128 KiB each
  constexpr int EP = (BM * BN3 * 4 > 128 * 1024) ? 2 : 1, RP = BM / EP;
  static_assert(EP == 1 || RP == BM / 2, "a pass = the rows of one wave row group");
  float* stg = (float*)smem;
  constexpr int TPR = BN3 / 4;  // threads per staged row (4 columns each); NT3 / TPR = 8 rows per sweep
  const int cl = tid % TPR, n = nt * BN3 + 4 * cl;
  const bool vec = ((g.ldc & 3) == 0) && (!R1 || (g.ldr1 & 3) == 0) && (!R2 || (g.ldr2 & 3) == 0) && n + 4 <= g.N;
#pragma unroll
  for (int ps = 0; ps < EP; ++ps) {
    __syncthreads();
    if (EP == 1 || wm == ps) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int ml = (EP == 1 ? wm * (BM / 2) : 0) + i * 16 + frow;  // row inside this pass
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = g.alpha * (acc[i][j][r] + bv[j][r]);
            if constexpr (ACT == ACT_GELU) {
              x = gelu_erf(x);
            } else if constexpr (ACT == ACT_SILU) {
              x = silu(x);
            }
            v[r] = x * (cv[j][r] * rs[i]);
          }
          const int chunk = wn * 16 + j * 4 + fq;
          *(f32x4*)(stg + ml * BN3 + ((chunk ^ (ml & 31)) << 2)) = v;
        }
      }
    }
    __syncthreads();
    if constexpr (ACT == ACT_HEADSOFTMAX) {
      // softmax over each 128-column slice (one head), arithmetic as rowwise.hip head_softmax_kernel<32>; rows out as planes
#pragma unroll
      for (int k = 0; k < RP / 8; ++k) {
        const int ml = tid / TPR + 8 * k, m = row0 + ps * RP + ml;
        f32x4 v = *(const f32x4*)(stg + ml * BN3 + ((cl ^ (ml & 31)) << 2));
        const float mx = group_max<32>(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
        v = (f32x4){expf(v[0] - mx), expf(v[1] - mx), expf(v[2] - mx), expf(v[3] - mx)};
        const float sm = group_sum<32>(v[0] + v[1] + v[2] + v[3]);
        v = (f32x4){v[0] / sm, v[1] / sm, v[2] / sm, v[3] / sm};
        if (m >= row_end) continue;
        uint32_t h0, h1, l0, l1;
        split_bf16(v[0], v[1], h0, l0);
        split_bf16(v[2], v[3], h1, l1);
        *(uint2*)(C16 + (int64_t)m * g.ldc + n) = make_uint2(h0, h1);
        *(uint2*)(g.C16_lo + (int64_t)m * g.ldc + n) = make_uint2(l0, l1);
      }
    } else if constexpr (ACT == ACT_HEADNORM) {
      // a 128-column slice of the tile is one head: LayerNorm over it (weights by position inside the head), q and k slices
      // L2-normalised, rows out as bf16 hi / lo planes.  Arithmetic as rowwise.hip head_norm_kernel<32> (32 lanes x 4 columns).
      const f32x4 ww = *(const f32x4*)(g.hn_w + 4 * (cl & 31)), bb = *(const f32x4*)(g.hn_b + 4 * (cl & 31));
      const bool l2 = nt * (BN3 / 128) + (cl >> 5) < g.hn_l2_tiles;  // the head slice of these 32 lanes
#pragma unroll
      for (int k = 0; k < RP / 8; ++k) {
        const int ml = tid / TPR + 8 * k, m = row0 + ps * RP + ml;  // (uniform per half wave: the reductions stay inside the 32 lanes)
        f32x4 v = *(const f32x4*)(stg + ml * BN3 + ((cl ^ (ml & 31)) << 2));
        const float mean = group_sum<32>(v[0] + v[1] + v[2] + v[3]) / 128.f;
        const f32x4 d = {v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean};
        const float rstd = rsqrtf(group_sum<32>(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) / 128.f + 1e-5f);
        v = (f32x4){d[0] * rstd * ww[0] + bb[0], d[1] * rstd * ww[1] + bb[1], d[2] * rstd * ww[2] + bb[2], d[3] * rstd * ww[3] + bb[3]};
        if (l2) {
          const float nrm = sqrtf(group_sum<32>(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]));
          const float inv = 1.f / fmaxf(nrm, 1e-12f);
          v *= inv;
        }
        if (m >= row_end) continue;
        uint32_t h0, h1, l0, l1;
        split_bf16(v[0], v[1], h0, l0);
        split_bf16(v[2], v[3], h1, l1);
        *(uint2*)(C16 + (int64_t)m * g.ldc + n) = make_uint2(h0, h1);
        *(uint2*)(g.C16_lo + (int64_t)m * g.ldc + n) = make_uint2(l0, l1);
      }
    } else {
      // residual rows of this pass, all requested before its first store: stores count in vmcnt on gfx950 and the residual is often
      // updated in place, so a load left inside the row loop waits for the previous row's stores (RP / 8 serial write round trips)
      f32x4 q1[RP / 8], q2[RP / 8];
      if (vec && (R1 || R2)) {
#pragma unroll
        for (int k = 0; k < RP / 8; ++k) {
          int m = row0 + ps * RP + tid / TPR + 8 * k;
          m = m < row_end ? m : row_end - 1;
          const int64_t mr = g.r1_mod ? (m % g.r1_mod) : m;
          if (R1) q1[k] = *(const f32x4*)(R1 + mr * g.ldr1 + n);
          if (R2) q2[k] = *(const f32x4*)(R2 + (int64_t)m * g.ldr2 + n);
        }
      }
#pragma unroll
      for (int k = 0; k < RP / 8; ++k) {
        const int ml = tid / TPR + 8 * k, m = row0 + ps * RP + ml;
        if (m >= row_end || n >= g.N) continue;
        f32x4 v = *(const f32x4*)(stg + ml * BN3 + ((cl ^ (ml & 31)) << 2));
        const int64_t mr = g.r1_mod ? (m % g.r1_mod) : m;
        if (vec) {
          if (R1) {
            const f32x4 q = q1[k];
            v[0] += g.r1_scale * q[0], v[1] += g.r1_scale * q[1], v[2] += g.r1_scale * q[2], v[3] += g.r1_scale * q[3];
          }
          if (R2) {
            const f32x4 q = q2[k];
            v[0] += q[0], v[1] += q[1], v[2] += q[2], v[3] += q[3];
          }
          if (C) *(f32x4*)(C + (int64_t)m * g.ldc + n) = v;
          if (g.Cx2) store_x2_4p(g.Cx2 + (int64_t)m * 2 * g.ldc, n, v[0], v[1], v[2], v[3]);  // pre-split rows for the next GEMM (lane pairs: N % 32 == 0)
          if (g.C16_lo) {  // bf16 hi / lo planes (the operand format of the fp32-grade attention cores)
            uint32_t h0, h1, l0, l1;
            split_bf16(v[0], v[1], h0, l0);
            split_bf16(v[2], v[3], h1, l1);
            *(uint2*)(C16 + (int64_t)m * g.ldc + n) = make_uint2(h0, h1);
            *(uint2*)(g.C16_lo + (int64_t)m * g.ldc + n) = make_uint2(l0, l1);
          } else if (C16) {
            *(uint2*)(C16 + (int64_t)m * g.ldc + n) = make_uint2(pack_h16(g.h16, v[0], v[1]), pack_h16(g.h16, v[2], v[3]));
          }
        } else {
          for (int r = 0; r < 4 && n + r < g.N; ++r) {
            float x = v[r];
            if (R1) x += g.r1_scale * R1[mr * g.ldr1 + n + r];
            if (R2) x += R2[(int64_t)m * g.ldr2 + n + r];
            if (C) C[(int64_t)m * g.ldc + n + r] = x;
            if (C16) C16[(int64_t)m * g.ldc + n + r] = (uint16_t)(pack_h16(g.h16, x, 0.f) & 0xffff);
          }
        }
      }
    }
  }
}

template <int BM, int BN, int NS, int ACT, bool AX2>
int launch3_k(const GemmArgs& a, hipStream_t stream) {
  constexpr int ring = NS * (BM * 128 + 2 * BN * 64), stgb = (BM * BN * 4 > 128 * 1024) ? BM * BN * 2 : BM * BN * 4;
  constexpr int smem = ring > stgb ? ring : stgb;
  static_assert(smem <= 160 * 1024, "LDS");
  static DevOnce attr;
  if (smem > 65536 && !attr) {
    if (hipFuncSetAttribute((const void*)gemm_x3_kernel<BM, BN, NS, ACT, AX2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return MDM_ERR_LAUNCH;
    attr = true;
  }
  const int tm = (a.M + BM - 1) / BM + (a.goff ? a.ngroups : 0);
  const int tn = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL((gemm_x3_kernel<BM, BN, NS, ACT, AX2>), dim3((unsigned)(tm * tn)), dim3(2 * BN), smem, stream, a);
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}
template <int BM, int BN, int NS, int ACT>
int launch3_act(const GemmArgs& a, hipStream_t stream) {
  return a.A.kind == OP_X2_ROW ? launch3_k<BM, BN, NS, ACT, true>(a, stream) : launch3_k<BM, BN, NS, ACT, false>(a, stream);
}
template <int BM, int BN, int NS>
int launch3(const GemmArgs& a, hipStream_t stream) {
  switch (a.act) {
    case ACT_NONE: return launch3_act<BM, BN, NS, ACT_NONE>(a, stream);
    case ACT_GELU: return launch3_act<BM, BN, NS, ACT_GELU>(a, stream);
    case ACT_SILU: return launch3_act<BM, BN, NS, ACT_SILU>(a, stream);
    case ACT_HEADNORM: return launch3_act<BM, BN, NS, ACT_HEADNORM>(a, stream);
    case ACT_HEADSOFTMAX: return launch3_act<BM, BN, NS, ACT_HEADSOFTMAX>(a, stream);
    default: return MDM_ERR_UNSUPPORTED;
  }
}

}  // namespace

extern int g_bf16_variant;

bool gemm_x3_dma_eligible(const GemmArgs& a) {
  const bool base = a.precision == 3 && (a.A.kind == OP_F32_ROW || a.A.kind == OP_X2_ROW) && a.W.kind == OP_BF16_ROW && a.W.p_lo && a.batch == 1 && a.A.rpg == 0 &&
                    a.K >= 32 && (a.K % 32) == 0 && (a.A.ld % 4) == 0 && (a.W.ld % 8) == 0 && (a.W.bs1 % 8) == 0 &&
                    ((((uintptr_t)a.A.p) | ((uintptr_t)a.W.p) | ((uintptr_t)a.W.p_lo)) & 15) == 0 && a.M >= 1;
  if (!base) return false;
  // hi / lo plane outputs: whole 128-column slices, vector stores, no fp32 copy
  const bool planes_ok = a.C16 && a.C16_lo && !a.C && a.N % 128 == 0 && (a.ldc % 4) == 0 && (!a.R1 || (a.ldr1 & 3) == 0) &&
                         (!a.R2 || (a.ldr2 & 3) == 0) && ((((uintptr_t)a.C16) | ((uintptr_t)a.C16_lo)) & 7) == 0;
  if (a.C16_lo && !planes_ok) return false;
  if (a.Cx2 && ((a.N % 32) || (a.ldc % 4) || (((uintptr_t)a.Cx2) & 7) || (a.R1 && (a.ldr1 & 3)) || (a.R2 && (a.ldr2 & 3)) ||
                a.act == ACT_HEADNORM || a.act == ACT_HEADSOFTMAX))
    return false;
  if (a.act == ACT_HEADNORM)
    return planes_ok && a.hn_w && a.hn_b && !a.R1 && !a.R2 && !a.goff && ((((uintptr_t)a.hn_w) | ((uintptr_t)a.hn_b)) & 15) == 0;
  if (a.act == ACT_HEADSOFTMAX) return planes_ok && !a.R1 && !a.R2 && !a.goff;
  return a.act == ACT_NONE || a.act == ACT_GELU || a.act == ACT_SILU;
}

int gemm_x3_dma(const GemmArgs& a, hipStream_t stream) {
  if (!gemm_x3_dma_eligible(a)) return MDM_ERR_UNSUPPORTED;
  if (!a.C && !a.C16 && !a.Cx2) return MDM_ERR_ARG;
  const int64_t tiles128 = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128);
  const bool small = !a.goff && (tiles128 <= 256 || a.M <= 64);
  if (g_bf16_variant == 37) return launch3<128, 128, 3>(a, stream);  // A/B knobs: ring depth at the 128-row tile
  if (g_bf16_variant == 38) return launch3<128, 128, 4>(a, stream);
  if (g_bf16_variant == 39) return launch3<64, 128, 3>(a, stream);
  return small ? launch3<64, 128, 3>(a, stream) : launch3<128, 128, 2>(a, stream);
}

}  // namespace mdm
