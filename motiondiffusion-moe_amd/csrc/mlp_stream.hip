// Fused two-layer MLP with STREAMED weights for gfx950:   Y = epilogue( GELU(X W1^T + b1) W2^T + b2 )
// the expert MLPs of the MoE feed-forward (switch_moe.py:19-25,97-109: grouped by expert, gathered rows, gate-probability
// row scale) and the dense Linear-GELU-Linear pairs (fast_attention.py:121-126,293-299).
//
// Decomposition (what differs from csrc/mlp.hip, whose weights cross LDS behind one barrier per K step):
//   * one workgroup = 8 waves owns a tile of up to RT*16 rows and ALL Dout output columns; it is PERSISTENT: one workgroup per
//     CU walks the tiles, the stores of a tile drain under the next tile's work.  The waves split the N axis only: wave w owns
//     hidden units [32 w, 32 w + 32) of every 256-unit hidden chunk (phase 1) and output columns [16 NJ w, 16 NJ (w + 1))
//     (phase 2), for all rows of the tile.
//   * the ACTIVATION operand is what the waves share, so it lives in LDS and every wave reads all of it (XOR images,
//     conflict-free ds_read_b128 fragments): the tile's X rows stay RESIDENT ([rows][Din] 16-bit, loaded once per tile), so
//     phase 1 stages nothing and has no barrier; the GELU'd hidden chunk is published in two halves of 128 units through one
//     28-KiB image (112 KiB of X + 56 KiB of hidden chunk would be 8 KiB over the 160 KiB of a CU): four barriers per chunk.
//   * the WEIGHTS are private to a wave, so they never touch LDS: packed once as ONE linear stream of 1-KiB MFMA fragments
//     per (group, wave) in exactly the order the wave consumes them (mdm_mlp_stream_pack), they go global -> registers with
//     plain 16-byte loads through an 8-fragment register ring that runs 8 fragments (~0.9 k MFMA cycles) ahead.  No LDS-DMA,
//     no barrier and no LDS read for 94 % of the bytes a tile moves.
//   * the order of a chunk is  phase 1 (c) -> phase 2, second half of chunk c - 1 -> GELU half 0 -> publish -> phase 2, first
//     half of chunk c -> GELU half 1 -> publish.  The GELU is x * sigmoid(q(x)) on SCALAR fp32 instructions (beside MFMAs a
//     packed-fp32 instruction costs several plain ones; this file is compiled with -fno-slp-vectorize).  Interleaving its
//     pieces with the phase-2 MFMAs of the same wave (knob 48) measured 1.5 % slower than back to back, storing the 16-bit
//     outputs straight from the accumulators 3 % slower than staging full rows through LDS, requesting the next tile's X rows
//     between the two halves of the epilogue 4 % slower (the wait moves, it does not shrink): all three tried and dropped.
//   * every { MFMAs of one A fragment, 1 fragment read } is pinned by a scheduling barrier, the A fragments go
//     through a small register ring that runs ahead of the MFMAs: left alone hipcc sinks the weight refills of an unrolled
//     body to its end (the ring's run-ahead collapses) and, once registers are tight, reads one fragment, waits, issues its
//     MFMAs (one exposed LDS round trip per 32 MFMA cycles).
//   * tiles are balanced: the launch picks the tile height so that the tiles of all groups fill the CUs in whole rounds
//     (112-row tiles: 50176 routed rows = 448..464 tiles = 2 rounds at 7/8 of the rows a round could hold; the 128-row
//     tiles of mlp.hip made 392 tiles = 2 rounds, the second 53 % full).
// Register budget per lane (RT = 7, NJ = 4): y 112 + h 56 + weight ring 32 + A ring 16 + GELU temporaries.
#include <utility>

#include "gemm.h"
#include "kernels.h"
#include "row.h"

namespace mdm {
namespace {

constexpr int NT = 512, FC = 256;

template <typename T>
__device__ __forceinline__ T ldg(const uint8_t* p) {
  return *(const T*)p;
}

// Pins what is written before it: nothing may be scheduled across this point (a mask that lets LDS reads, MFMAs and ALU work
// through is no pin at all: the scheduler then moves exactly those above it).
__device__ __forceinline__ void pin() { __builtin_amdgcn_sched_barrier(0); }

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Which tile this workgroup owns: groups are cut into ceil(rows / tile_h) tiles of equal height (a multiple of 16).  Lane e of
// every wave looks at group e (at most 64 groups); mt = tile index; false = no such tile.
// gtab: the group offsets goff[0 .. ng] (or {0, M}) in LDS, copied once per workgroup (a persistent workgroup looks tiles up many
// times; from global memory every lookup is a dependent L2 round trip in front of the tile)
__device__ __forceinline__ bool find_tile(const MdmMlpDesc& g, const int* gtab, int tile_h, int lane, int mt, int& row0, int& row_end,
                                          int& grp) {
  const int ng = g.goff ? g.ngroups : 1;
  const int e = lane < ng ? lane : ng - 1;
  const int b = gtab[e], en = gtab[e + 1];
  const int rows = lane < ng ? en - b : 0;
  const int t = (rows + tile_h - 1) / tile_h;
  int incl = t;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(incl, d, 64);
    if (lane >= d) incl += v;
  }
  const unsigned long long m = __ballot(mt < incl);
  if (m == 0) return false;
  const int ge = __ffsll((long long)m) - 1;
  const int gb = __builtin_amdgcn_readlane(b, ge), gen = __builtin_amdgcn_readlane(en, ge);
  const int gt = __builtin_amdgcn_readlane(t, ge), gi = __builtin_amdgcn_readlane(incl, ge);
  const int hg = (((gen - gb + gt - 1) / gt) + 15) & ~15;
  row0 = gb + (mt - (gi - gt)) * hg;
  row_end = row0 + hg < gen ? row0 + hg : gen;
  grp = ge;
  return row0 < row_end;
}

// Epilogue shared by both forms: EPT row tiles at a time staged as fp32 [16 EPT][DOUT] in LDS (all of it is free by then),
// written as full rows with the row scale, the residuals and the optional 16-bit copy.
template <typename HT, int RT, int NJ, int SMEM, int KO>
__device__ __forceinline__ void store_tile(const MdmMlpDesc& g, f32x4 (&y)[RT][NJ], uint8_t* smem, int row0, int row_end, int tid,
                                           int wn, int frow, int fq) {
  constexpr int DOUT = NJ * 128;
  if (!g.C && !g.R1 && !g.R2 && RT * 16 * DOUT * 2 <= SMEM) {
    // 16-bit output only (the expert MLPs of the throughput modes): the whole tile staged once as 16-bit [rows][DOUT], one
    // wave-instruction stores one full row (64 lanes x 16 B)
    constexpr int ROWB = DOUT * 2, CH = ROWB / 16;  // 16-B chunks per row
    lds_barrier();
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int ml = i * 16 + frow, m = row0 + ml;
      const float rs = g.rowscale ? g.rowscale[m < row_end ? m : row_end - 1] : 1.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int n = wn * (16 * NJ) + 16 * j + 4 * fq;  // 4 consecutive columns = 8 bytes
        const f32x4 v = y[i][j];
        *(uint2*)(smem + ml * ROWB + ((((n >> 3) ^ (ml & (CH - 1))) << 4)) + ((n >> 2) & 1) * 8) =
            make_uint2(HT::pack(v[0] * rs, v[1] * rs), HT::pack(v[2] * rs, v[3] * rs));
      }
    }
    lds_barrier();
    constexpr int TPR = CH, RPS = NT / TPR;  // threads per row, rows per sweep
    const int cl = tid % TPR;
#pragma unroll
    for (int k = 0; k < (RT * 16 + RPS - 1) / RPS; ++k) {
      const int ml = tid / TPR + RPS * k, m = row0 + ml;
      if (ml >= RT * 16 || m >= row_end) continue;
      const uint4 v = *(const uint4*)(smem + ml * ROWB + ((cl ^ (ml & (CH - 1))) << 4));
      if (KO == 6 && v.x != 0x12345678u) continue;
      *(uint4*)(g.C16 + (int64_t)m * g.ldc + cl * 8) = v;
    }
    return;
  }
  // ---- general form: EPT row tiles at a time staged as fp32 [16 EPT][DOUT] in LDS, written as full rows -------------------
  constexpr int EPT = SMEM / (16 * DOUT * 4) < RT ? SMEM / (16 * DOUT * 4) : RT;
  constexpr int NPASS = (RT + EPT - 1) / EPT;
  constexpr int CPR = DOUT / 4;  // float4 chunks per row
  float* stg = (float*)smem;
  const float* __restrict__ R1 = g.R1;
  const float* __restrict__ R2 = g.R2;
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    float rs[EPT];
#pragma unroll
    for (int ii = 0; ii < EPT; ++ii) {
      rs[ii] = 1.f;
      if (g.rowscale && p * EPT + ii < RT) {
        const int m = row0 + (p * EPT + ii) * 16 + frow;
        rs[ii] = g.rowscale[m < row_end ? m : row_end - 1];
      }
    }
    lds_barrier();
#pragma unroll
    for (int ii = 0; ii < EPT; ++ii) {
      if (p * EPT + ii < RT) {
        const int ml = ii * 16 + frow;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int c4 = wn * (4 * NJ) + 4 * j + fq;  // float4 chunk of columns wn 16 NJ + 16 j + 4 fq
          f32x4 v = y[p * EPT + ii][j];
          v[0] *= rs[ii], v[1] *= rs[ii], v[2] *= rs[ii], v[3] *= rs[ii];
          *(f32x4*)(stg + ml * DOUT + ((c4 ^ (ml & 31)) << 2)) = v;
        }
      }
    }
    constexpr int RPP = NT / CPR;  // rows per sweep of the workgroup
    const int cl = tid % CPR, n = 4 * cl;
    // the pass's residual rows, all requested before its first store (stores count in vmcnt on gfx950 and the residual is usually
    // updated in place: a load inside the row loop waits for the previous row's stores)
    // (the 32- / 64-row forms of the dense pairs: that is where the residuals are; at RT = 7 and at the big widths the accumulators of
    // the later passes leave no room for them -- 276 / 368 B of scratch -- and those launches have no residual)
    constexpr bool PRE = RT <= 4 && NJ == 4;
    f32x4 q1[PRE ? 16 * EPT / RPP : 1], q2[PRE ? 16 * EPT / RPP : 1];
    if constexpr (PRE) {
      if (R1 || R2) {
#pragma unroll
        for (int k = 0; k < 16 * EPT / RPP; ++k) {
          int m = row0 + p * EPT * 16 + tid / CPR + RPP * k;
          m = m < row_end ? m : row_end - 1;
          if (R1) q1[k] = *(const f32x4*)(R1 + (int64_t)m * g.ldr1 + n);
          if (R2) q2[k] = *(const f32x4*)(R2 + (int64_t)m * g.ldr2 + n);
        }
      }
    }
    lds_barrier();
#pragma unroll
    for (int k = 0; k < 16 * EPT / RPP; ++k) {
      const int ml = tid / CPR + RPP * k, tl = p * EPT * 16 + ml, m = row0 + tl;
      if (tl >= RT * 16 || m >= row_end) continue;
      f32x4 v = *(const f32x4*)(stg + ml * DOUT + ((cl ^ (ml & 31)) << 2));
      if (R1) {
        f32x4 q;
        if constexpr (PRE) q = q1[k];
        else q = *(const f32x4*)(R1 + (int64_t)m * g.ldr1 + n);
        v[0] += g.r1_scale * q[0], v[1] += g.r1_scale * q[1], v[2] += g.r1_scale * q[2], v[3] += g.r1_scale * q[3];
      }
      if (R2) {
        f32x4 q;
        if constexpr (PRE) q = q2[k];
        else q = *(const f32x4*)(R2 + (int64_t)m * g.ldr2 + n);
        v[0] += q[0], v[1] += q[1], v[2] += q[2], v[3] += q[3];
      }
      if (KO == 6 && v[0] != 123.456f) continue;
      if (g.C) *(f32x4*)(g.C + (int64_t)m * g.ldc + n) = v;
      if (g.C16) *(uint2*)(g.C16 + (int64_t)m * g.ldc + n) = make_uint2(HT::pack(v[0], v[1]), HT::pack(v[2], v[3]));
    }
  }
}

// Diagnostic build KO == 9 (-DMDM_DIAG library only, knob 49, tools/mlp_stamps.py): wave 0 of every workgroup sums s_memtime
// differences per phase and adds them to the eight 64-bit counters handed over through mdm_diag_mlp_counters() (a buffer of
// their own: no output or residual pointer is reused for them; the launch is refused while none is set).
// Read the SHARES of this build, not its run time.
__device__ __forceinline__ unsigned long long stamp_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define XSTAMP(k)                                  \
  do {                                             \
    if constexpr (KO == 9) {                       \
      const unsigned long long n__ = stamp_now();  \
      acc[k] += n__ - last;                        \
      last = n__;                                  \
    }                                              \
  } while (0)

template <int RT, int DIN, int NJ = 4, bool TAIL = false>
struct XGeo {
  static constexpr int ROWS = RT * 16;
  static constexpr int XROW_B = DIN * 2;     // X image: 16-B chunk c of row m at slot c ^ (m & 15)
  static constexpr int XIMG_B = ROWS * XROW_B;
  static constexpr int HID_B = ROWS * 256;   // half hidden chunk, 16-bit [ROWS][128]: chunk c of row m at slot c ^ (m & 15)
  static constexpr int IMG_B = XIMG_B + HID_B;
  // epilogue staging: the whole 16-bit tile [ROWS][128 NJ], or at least one fp32 row tile [16][128 NJ]
  static constexpr int STG_B = ROWS * NJ * 256 > NJ * 8192 ? ROWS * NJ * 256 : NJ * 8192;
  static constexpr int TAIL_B = ROWS * NJ * 512;  // the stylization tail stages whole fp32 rows [ROWS][128 NJ]
  static constexpr int SMEM0 = IMG_B > STG_B ? IMG_B : STG_B;
  static constexpr int SMEM = (TAIL && TAIL_B > SMEM0) ? TAIL_B : SMEM0;
};

// The GELU of one value pair in two pieces (so that a piece fits beside the four MFMAs of one A fragment): gelu_sig2 of
// mdm_common.h cut after the exp2, written on SCALAR fp32 operations: beside MFMAs a packed-fp32 instruction (v_pk_fma_f32)
// costs several times two plain ones (MI355X guide, cycle constants), and this file is compiled with -fno-slp-vectorize so
// that hipcc does not re-pack them.
__device__ __forceinline__ float gelu_a1(float v) {
  const float xc = __builtin_amdgcn_fmed3f(v, -6.5f, 6.5f);
  const float x2 = xc * xc;
  float p = fmaf(x2, -3.229054982512025e-06f, 8.82392268977128e-05f);
  p = fmaf(p, x2, 0.00036026936140842736f);
  p = fmaf(p, x2, -0.10522668063640594f);
  p = fmaf(p, x2, -2.3020453453063965f);
  return __builtin_amdgcn_exp2f(p * xc);
}
__device__ __forceinline__ f32x2 gelu_part_a(f32x2 v) { return (f32x2){gelu_a1(v[0]), gelu_a1(v[1])}; }
__device__ __forceinline__ f32x2 gelu_part_b(f32x2 v, f32x2 e) {
  return (f32x2){v[0] * __builtin_amdgcn_rcpf(e[0] + 1.0f), v[1] * __builtin_amdgcn_rcpf(e[1] + 1.0f)};
}

// ---- the Performer tail behind the proj_out pair (fast_attention.py:165-178), in the same launch ---------------------------
// Per 16 RT-row tile, with the pair's result y (+ b2) in the accumulators:
//   rows as 16-bit (what the unfused path stores and reloads) -> LDS image -> row phase exactly as csrc/style_gemm.hip / rowwise.hip
//   style_in (one wave per row: post_norm, L2 norm * sqrt(D), style norm, (1 + scale) / shift, SiLU) in place -> out_layers.2 with
//   Wout streamed global -> registers -> (y + b) * out_scale + resid staged as fp32 rows -> one wave per row: fp32 row out and,
//   optionally, LayerNorm(row) as 16-bit (the pre_norm of the block that follows: one more launch saved).
// Every row goes through the same arithmetic as in the unfused launches; the results agree to rounding, not bit for bit (the two
// compilations contract multiply-adds differently, and a last-bit difference before the 16-bit image flips a rounding there).

template <typename HT, int RT, int NR>
__device__ __forceinline__ void pair_tail(const PairTail& st, f32x4 (&y)[RT][4], typename HT::frag_t (&R)[NR], uint8_t* smem, int row0,
                                          int row_end, int tid, int wn) {
  typedef typename HT::frag_t frag_t;
  typedef Row<8, true> R8;
  constexpr int D = 512, FMT = HT::FMT, RPW = 2 * RT;  // rows per wave in the row phases
  const int lane = tid & 63, frow = lane & 15, fq = lane >> 4;
  // the ring's first fragments of Wout are requested now: they land during the row phase
  const uint8_t* wp = (const uint8_t*)st.ws + (int64_t)wn * 64 * 1024 + lane * 16;
#pragma unroll
  for (int f = 0; f < NR; ++f) R[f] = *(const frag_t*)(wp + f * 1024);
  wp += NR * 1024;
  lds_barrier();  // every wave is past its last reads of the X rows and of the hidden image: the LDS is free
  // rows as 16-bit in the MFMA image layout (16-B chunk c of row r at slot c ^ (r & 15))
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const int ml = i * 16 + frow;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = wn * 64 + 16 * j + 4 * fq;
      const f32x4 v = y[i][j];
      *(uint2*)(smem + ml * 1024 + ((((n >> 3) ^ (ml & 15))) << 4) + ((n >> 2) & 1) * 8) =
          make_uint2(HT::pack(v[0], v[1]), HT::pack(v[2], v[3]));
    }
  }
  lds_barrier();
  {
    R8 pww, pbb, sww, sbb;
    pww.load(st.pw, D, lane), pbb.load(st.pb, D, lane);
    sww.load(st.sw, D, lane), sbb.load(st.sb, D, lane);
#pragma unroll 2
    for (int q = 0; q < RPW; ++q) {
      const int rl = RPW * wn + q;
      int row = row0 + rl;
      row = row < row_end ? row : row_end - 1;  // rows past the tile's end: never stored
      const float* scb = st.sc + (int64_t)(row / st.S) * 2 * D;
      R8 r, scale, shift;
      scale.load(scb, D, lane);
      shift.load(scb + D, D, lane);
      // lane holds columns 4 l .. 4 l + 3 and 256 + 4 l ..: 8 bytes each, 16-B chunks (l >> 1) and 32 + (l >> 1), half l & 1
      uint8_t* ir = smem + rl * 1024 + (lane & 1) * 8;
      const uint2 u0 = *(const uint2*)(ir + ((((lane >> 1)) ^ (rl & 15)) << 4));
      const uint2 u1 = *(const uint2*)(ir + (((32 + (lane >> 1)) ^ (rl & 15)) << 4));
      r.e[0] = h16_lo_f32(FMT, u0.x), r.e[1] = h16_hi_f32(FMT, u0.x), r.e[2] = h16_lo_f32(FMT, u0.y), r.e[3] = h16_hi_f32(FMT, u0.y);
      r.e[4] = h16_lo_f32(FMT, u1.x), r.e[5] = h16_hi_f32(FMT, u1.x), r.e[6] = h16_lo_f32(FMT, u1.y), r.e[7] = h16_hi_f32(FMT, u1.y);
      r.layernorm(pww, pbb, D, lane);
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += r.e[j] * r.e[j];
      const float inv = sqrtf((float)D) / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
#pragma unroll
      for (int j = 0; j < 8; ++j) r.e[j] *= inv;
      r.layernorm(sww, sbb, D, lane);
#pragma unroll
      for (int j = 0; j < 8; ++j) r.e[j] = silu(r.e[j] * (1.f + scale.e[j]) + shift.e[j]);
      *(uint2*)(ir + ((((lane >> 1)) ^ (rl & 15)) << 4)) = make_uint2(HT::pack(r.e[0], r.e[1]), HT::pack(r.e[2], r.e[3]));
      *(uint2*)(ir + (((32 + (lane >> 1)) ^ (rl & 15)) << 4)) = make_uint2(HT::pack(r.e[4], r.e[5]), HT::pack(r.e[6], r.e[7]));
    }
  }
  lds_barrier();
  // out_layers.2: y[rows x 64 columns of this wave] = image . Wout^T
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) y[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const int xb = frow * 1024 + ((fq ^ frow) << 4);  // (row frow, K step 0); step s reads chunk (4 s) ^ (fq ^ frow)
    frag_t A[2][RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) A[0][i] = *(const frag_t*)(smem + xb + i * 16384);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s + 1 < 16) {
#pragma unroll
        for (int i = 0; i < RT; ++i) A[(s + 1) & 1][i] = *(const frag_t*)(smem + (xb ^ (64 * ((s + 1) & 3))) + ((s + 1) >> 2) * 256 + i * 16384);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int slot = (s * 4 + j) & (NR - 1);
#pragma unroll
        for (int i = 0; i < RT; ++i) y[i][j] = HT::mfma16(R[slot], A[s & 1][i], y[i][j]);
        R[slot] = *(const frag_t*)(wp + slot * 1024);  // the last NR refills read (and discard) the next wave's / the padding
        if (slot == NR - 1) wp += NR * 1024;
      }
      pin();
    }
  }
  // (y + b) * out_scale staged as fp32 [rows][512] (16-B group c of row r at slot c ^ (r & 31))
  lds_barrier();
  float* stg = (float*)smem;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = wn * 64 + 16 * j + 4 * fq;
    const f32x4 bb = *(const f32x4*)(st.bias + n);
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int ml = i * 16 + frow;
      f32x4 v = y[i][j];
      v[0] = (v[0] + bb[0]) * st.out_scale, v[1] = (v[1] + bb[1]) * st.out_scale, v[2] = (v[2] + bb[2]) * st.out_scale,
      v[3] = (v[3] + bb[3]) * st.out_scale;
      *(f32x4*)(stg + ml * D + (((n >> 2) ^ (ml & 31)) << 2)) = v;
    }
  }
  lds_barrier();
  {
    // the rows' residual (and skip) operands are all requested first: one memory round trip per wave, not one per row
    R8 xr[RPW], sk[RPW];
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      int64_t m = (int64_t)row0 + RPW * wn + q;
      m = m < row_end ? m : row_end - 1;
      xr[q].load(st.resid + m * D, D, lane);
      if (st.skip) sk[q].load(st.skip + m * D, D, lane);
    }
    R8 lww, lbb, l2ww, l2bb;
    if (st.lw) lww.load(st.lw, D, lane), lbb.load(st.lb, D, lane);
    if (st.l2w) l2ww.load(st.l2w, D, lane), l2bb.load(st.l2b, D, lane);
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      const int rl = RPW * wn + q;
      const int64_t m = (int64_t)row0 + rl;
      if (m >= row_end) continue;  // (wave-uniform)
      R8 r;
      const R8& x = xr[q];
      const f32x4 v0 = *(const f32x4*)(stg + rl * D + ((lane ^ (rl & 31)) << 2));
      const f32x4 v1 = *(const f32x4*)(stg + rl * D + (((64 + lane) ^ (rl & 31)) << 2));
      r.e[0] = v0[0] + x.e[0], r.e[1] = v0[1] + x.e[1], r.e[2] = v0[2] + x.e[2], r.e[3] = v0[3] + x.e[3];
      r.e[4] = v1[0] + x.e[4], r.e[5] = v1[1] + x.e[5], r.e[6] = v1[2] + x.e[6], r.e[7] = v1[3] + x.e[7];
      if (st.skip) {  // the block's tail: LN(skip + skip_scale * r) out, LN of that for the next block
#pragma unroll
        for (int j = 0; j < 8; ++j) r.e[j] = sk[q].e[j] + st.skip_scale * r.e[j];
        r.layernorm(lww, lbb, D, lane);
        r.store(st.out + m * D, D, lane);
        if (st.l2w) {
          r.layernorm(l2ww, l2bb, D, lane);
          r.template store_h16<FMT>(st.ln16 + m * D, D, lane);
        }
      } else {
        r.store(st.out + m * D, D, lane);
        if (st.ln16) {
          r.layernorm(lww, lbb, D, lane);
          r.template store_h16<FMT>(st.ln16 + m * D, D, lane);
        }
      }
    }
  }
}

// KO: 0 = the real kernel, 10 = the real kernel + the Performer tail.  Every other value exists only in the diagnostic library
// (built with -DMDM_DIAG by `python motiondiffusion-moe_amd/build.py --diag`; the shipped libmdm_hip.so neither instantiates them
// nor accepts their knobs): timing-only knock-outs for tools/mlp_ko.py whose results are wrong -- 1 no GELU arithmetic, 2 no
// weight refills, 4 no phase-1 MFMAs, 5 no phase-2 MFMAs, 6 no output stores -- and A/B arms: 7 = the erf-form GELU of the
// LDS-staged kernel, 8 = GELU pieces interleaved with the phase-2 MFMAs of the same wave, 9 = stamped build
template <typename HT, int RT, int NJ, int DIN, int KO>
__global__ __launch_bounds__(NT, 2) void fused_mlp_stream_kernel(const MdmMlpDesc g, const int tile_h, const PairTail st) {
  typedef typename HT::frag_t frag_t;
  constexpr bool TAIL = KO == 10;  // the Performer tail (post-norm, stylization, out_layers.2, residual) behind the pair
  static_assert(!TAIL || (NJ == 4 && DIN == 512 && RT <= 4), "the tail is written for D = 512 rows");
  typedef XGeo<RT, DIN, NJ, TAIL> G;
  constexpr int NKO = DIN / 128, NLINE = DIN / 64;
  constexpr int NA = 4, PD = NA - 1;  // A-fragment ring: NA registers, PD fragments ahead of the MFMAs
  // weight ring: fragments in flight per wave, 16 where the accumulators leave room.  The 32-row form runs one workgroup per CU
  // like the others (its launches have at most one tile per CU: <= 8192 rows), so it takes the 256-register budget and the deep
  // ring as well: what bounds these short-tile launches is how many weight bytes a CU has in flight
  constexpr int NR = (RT <= 4 && NJ == 4 && DIN % 256 == 0) ? 16 : 8;
  constexpr int NF = 4 * RT;          // A fragments per unrolled body (4 K steps x RT row tiles)
  constexpr bool ILV = KO == 8;  // knob 48: GELU pieces interleaved with phase-2 MFMAs (measured 1.5 % SLOWER than back to back)
  static_assert(NF % NA == 0, "the ring must close over the unrolled body");
  static_assert(NF >= 4 * RT, "one GELU pair piece per fragment: 2 RT pairs x 2 pieces");
  extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
  uint8_t* const ximg = smem;
  uint8_t* const hid = smem + G::XIMG_B;
  int* const gtab = (int*)(smem + G::SMEM);  // group offsets (<= 65 ints), behind everything the tiles use
  {
    const int ng = g.goff ? g.ngroups : 1;
    if ((int)threadIdx.x <= ng) gtab[threadIdx.x] = g.goff ? g.goff[threadIdx.x] : (threadIdx.x ? g.M : 0);
    lds_barrier();
  }

  unsigned long long acc[8] = {}, last = 0;
  if constexpr (KO == 9) last = stamp_now();
  // Persistent over tiles: workgroup b takes tiles remap(b), remap(b) + grid, ...; round k hands an XCD a contiguous range of
  // tiles (~1 group: its weights stay in that XCD's L2).
  for (int mt = xcd_remap(blockIdx.x, gridDim.x);; mt += gridDim.x) {
    // the thread id is made opaque per tile (and again per chunk / for the epilogue): otherwise every lane-constant address of
    // the body is hoisted out of this loop and lives in (spilled) registers across it; recomputing is a few instructions
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    int row0, row_end, grp;
    if (!find_tile(g, gtab, tile_h, lane, mt, row0, row_end, grp)) break;
    XSTAMP(0);

    const int nchunk = g.F / FC;
    constexpr int fpc = (DIN / 32) * 2 + 8 * NJ;
    const uint8_t* wp = (const uint8_t*)g.wstream + ((int64_t)grp * g.wstream_gs + (int64_t)wn * nchunk * fpc * 512) * 2 + lane * 16;

    frag_t R[NR];
#pragma unroll
    for (int f = 0; f < NR; ++f) R[f] = ldg<frag_t>(wp + f * 1024);
    wp += NR * 1024;  // wp + 1024 f is now the fragment that refills slot f

    // ---- X tile -> LDS: wave w < RT brings rows 16 w .. 16 w + 15; one instruction = 8 rows x one 128-B line ---------------
    if (wn < RT) {
      const uint8_t* xp[2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int r = 16 * wn + 8 * hf + (lane >> 3);
        int srow = row0 + r;
        srow = srow < row_end ? srow : row_end - 1;
        const int64_t src = g.gather ? (int64_t)g.gather[srow] : (int64_t)srow;
        xp[hf] = (const uint8_t*)(g.X + src * g.ldx) + (lane & 7) * 16;
      }
      constexpr int LB = NLINE < 8 ? NLINE : 8;  // lines in flight per half row: the accumulators are not live yet
#pragma unroll
      for (int c0 = 0; c0 < NLINE; c0 += LB) {
        uint4 v[2][LB];
#pragma unroll
        for (int c = 0; c < LB; ++c)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) v[hf][c] = ldg<uint4>(xp[hf] + (c0 + c) * 128);
#pragma unroll
        for (int c = 0; c < LB; ++c)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            // 16-B chunk 8 (c0 + c) + (lane & 7) of row r goes to slot chunk ^ (r & 15)
            const int r = 16 * wn + 8 * hf + (lane >> 3);
            const int ch = (8 * (c0 + c) + (lane & 7)) ^ (r & 15);
            *(uint4*)(ximg + r * G::XROW_B + (ch << 4)) = v[hf][c];
          }
      }
    }
    f32x4 y[RT][NJ];
    {
      const float* b2 = g.b2 ? g.b2 + (int64_t)grp * g.b2_gs + wn * (16 * NJ) + (lane >> 4) * 4 : nullptr;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const f32x4 bb = b2 ? *(const f32x4*)(b2 + j * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < RT; ++i) y[i][j] = bb;
      }
    }
    lds_barrier();
    XSTAMP(1);

    f32x4 h[RT][2];
    uint2 pk[RT];  // the GELU'd half that is about to be published, packed
    // One half of phase 2 (K = the 128 hidden units of the image) with the GELU of the h[.][GJ] accumulators into pk[]
    // interleaved (GJ < 0: none).  hb: byte offset of (row fr, K step 0) in the half image.
    auto phase2 = [&](auto hf_c, auto gj_c, int hb) __attribute__((always_inline)) {
      constexpr int HF = decltype(hf_c)::value, GJ = decltype(gj_c)::value;
      frag_t A[NA];
      f32x2 ge = {0.f, 0.f};
      uint32_t plo = 0;
#pragma unroll
      for (int k = 0; k < PD; ++k) A[k % NA] = *(const frag_t*)(hid + (hb ^ (64 * (k / RT))) + (k % RT) * 4096);
#pragma unroll
      for (int k = 0; k < NF; ++k) {
        const int sq = k / RT, i = k % RT;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int slot = ((4 * HF + sq) * NJ + j) & (NR - 1);
          if constexpr (KO == 5) {
            asm volatile("" ::"v"(A[k % NA]), "v"(R[slot]));
          } else {
            y[i][j] = HT::mfma16(R[slot], A[k % NA], y[i][j]);
          }
        }
        if (k + PD < NF) A[(k + PD) % NA] = *(const frag_t*)(hid + (hb ^ (64 * ((k + PD) / RT))) + ((k + PD) % RT) * 4096);
        if constexpr (GJ >= 0 && ILV) {
          // fragment k carries piece k & 1 of value pair k >> 1: pair p = elements 2 (p & 1), 2 (p & 1) + 1 of h[p >> 1][GJ]
          const int p = k >> 1, ig = p >> 1, e0 = 2 * (p & 1);
          const f32x2 v = {h[ig][GJ][e0], h[ig][GJ][e0 + 1]};
          if ((k & 1) == 0) {
            ge = KO == 1 ? v : gelu_part_a(v);
          } else {
            const f32x2 gv = KO == 1 ? v : gelu_part_b(v, ge);
            const uint32_t w = HT::pack(gv[0], gv[1]);
            if ((p & 1) == 0) plo = w;
            else pk[ig] = make_uint2(plo, w);
          }
        }
        if (i == RT - 1) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int slot = ((4 * HF + sq) * NJ + j) & (NR - 1);
            if constexpr (KO != 2) R[slot] = ldg<frag_t>(wp + slot * 1024);
            if (slot == NR - 1) wp += NR * 1024;
          }
        }
        pin();
      }
    };
    // GELU of h[.][GJ] -> pk[] on its own (first chunk: there is no phase 2 to hide it under; knobs 7 / 8: always)
    auto gelu_alone = [&](auto gj_c) __attribute__((always_inline)) {
      constexpr int GJ = decltype(gj_c)::value;
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        f32x2 g01 = {h[i][GJ][0], h[i][GJ][1]}, g23 = {h[i][GJ][2], h[i][GJ][3]};
        if constexpr (KO == 7) g01 = gelu_erf2(g01), g23 = gelu_erf2(g23);
        else if constexpr (KO != 1) g01 = gelu_part_b(g01, gelu_part_a(g01)), g23 = gelu_part_b(g23, gelu_part_a(g23));
        pk[i] = make_uint2(HT::pack(g01[0], g01[1]), HT::pack(g23[0], g23[1]));
      }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, -1> IN;

    // phase 1 of one chunk (no barrier): h[rows x 32 units of this wave] = X . W1 chunk^T + b1
    auto phase1 = [&](int chunk, int ln) __attribute__((always_inline)) {
      const int fr = ln & 15, fc = (ln >> 4) ^ fr;  // fc: the XOR-swizzled chunk of K step 0; step u reads chunk (4 u) ^ fc
      {
        const float* b1 = g.b1 ? g.b1 + (int64_t)grp * g.b1_gs + wn * 32 + (ln >> 4) * 4 + chunk * FC : nullptr;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f32x4 bb = b1 ? *(const f32x4*)(b1 + j * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < RT; ++i) h[i][j] = bb;
        }
      }
      const int xb = fr * G::XROW_B + (fc << 4);  // byte offset of (row fr, K step 0) in the X image; bits 6..7 flip per step
      frag_t A[NA];
#pragma unroll
      for (int k = 0; k < PD; ++k) A[k % NA] = *(const frag_t*)(ximg + (xb ^ (64 * (k / RT))) + (k % RT) * 16 * G::XROW_B);
      constexpr int KB = NR / 8;  // unrolled bodies (4 K steps = 8 weight fragments each) per turn of the weight ring
      static_assert(NKO % KB == 0, "the weight ring must close over phase 1");
#pragma unroll 1
      for (int ko = 0; ko < NKO; ko += KB) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          const int kc = ko + kb;
          const int kn = kc + 1 == NKO ? 0 : kc + 1;  // the body's last prefetches belong to the next body (the last ones are not used)
#pragma unroll
          for (int k = 0; k < NF; ++k) {
            const int u = k / RT, i = k % RT, s0 = 8 * kb + 2 * u;
            if constexpr (KO == 4) {
              asm volatile("" ::"v"(A[k % NA]), "v"(R[s0]), "v"(R[s0 + 1]));
            } else {
              h[i][0] = HT::mfma16(R[s0], A[k % NA], h[i][0]);
              h[i][1] = HT::mfma16(R[s0 + 1], A[k % NA], h[i][1]);
            }
            const int kp = k + PD, up = (kp % NF) / RT, ip = kp % RT;
            A[kp % NA] = *(const frag_t*)(ximg + (xb ^ (64 * up)) + (kp < NF ? kc : kn) * 256 + ip * 16 * G::XROW_B);
            if (i == RT - 1) {
              if constexpr (KO != 2) {
                R[s0] = ldg<frag_t>(wp + s0 * 1024);
                R[s0 + 1] = ldg<frag_t>(wp + (s0 + 1) * 1024);
              }
            }
            pin();
          }
        }
        wp += NR * 1024;
      }
    };
    // publish pk[] as the half image: [image free] write [published]
    auto publish = [&](int ln) __attribute__((always_inline)) {
      const int fr = ln & 15;
      // image position of this lane's four units of a half: 16 wn + 4 fq (+ r): chunk 2 wn + (fq >> 1), byte 8 (fq & 1)
      uint8_t* const hw = hid + fr * 256 + (((2 * wn + (ln >> 5)) ^ fr) << 4) + ((ln >> 4) & 1) * 8;
      lds_barrier();  // the image is free: every wave is past its reads of the half published before
#pragma unroll
      for (int i = 0; i < RT; ++i) *(uint2*)(hw + i * 4096) = pk[i];
      lds_barrier();  // published
    };
    // Lane-constant addresses are recomputed per phase from an opaque copy of the lane id, so that nothing but the accumulators,
    // the weight ring and the stream pointer lives across the phases (the register file is the limit here).
    auto olane = [&]() __attribute__((always_inline)) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      return ln;
    };
    auto hbase = [](int ln) { return (ln & 15) * 256 + ((((ln >> 4) ^ (ln & 15))) << 4); };  // (row fr, K step 0) of the half image

    // chunk 0: nothing to hide its first GELU under
    {
      const int ln = olane();
      phase1(0, ln);
      XSTAMP(2);
      gelu_alone(I0());
      XSTAMP(3);
      publish(ln);
      XSTAMP(4);
      if constexpr (ILV) {
        phase2(I0(), I1(), hbase(ln));
      } else {
        phase2(I0(), IN(), hbase(ln));
        gelu_alone(I1());
      }
      XSTAMP(5);
      publish(ln);
      XSTAMP(4);
    }
#pragma unroll 1
    for (int chunk = 1; chunk < nchunk; ++chunk) {
      const int ln = olane();
      phase1(chunk, ln);
      XSTAMP(2);
      // the second half of the PREVIOUS chunk's phase 2 (published before this chunk's phase 1) with the GELU of this chunk's
      // half 0 under it
      if constexpr (ILV) {
        phase2(I1(), I0(), hbase(ln));
      } else {
        phase2(I1(), IN(), hbase(ln));
        gelu_alone(I0());
      }
      XSTAMP(5);
      publish(ln);
      XSTAMP(4);
      if constexpr (ILV) {
        phase2(I0(), I1(), hbase(ln));
      } else {
        phase2(I0(), IN(), hbase(ln));
        gelu_alone(I1());
      }
      XSTAMP(5);
      publish(ln);
      XSTAMP(4);
    }
    phase2(I1(), IN(), hbase(olane()));  // the last chunk's second half
    XSTAMP(5);

    // the epilogue derives its addresses from an opaque thread id of its own (see above)
    int te = threadIdx.x;
    asm volatile("" : "+v"(te));
    if constexpr (TAIL) {
      pair_tail<HT, RT, NR>(st, y, R, smem, row0, row_end, te, wn);
    } else if constexpr (KO == 9) {
      store_tile<HT, RT, NJ, G::SMEM, KO>(g, y, smem, row0, row_end, te, wn, te & 15, (te & 63) >> 4);
      XSTAMP(6);
      if (te == 0) atomicAdd((unsigned long long*)st.out + 7, 1ull);  // st.out: the diagnostic counters (launcher below)
    } else {
      store_tile<HT, RT, NJ, G::SMEM, KO>(g, y, smem, row0, row_end, te, wn, te & 15, (te & 63) >> 4);
    }
    lds_barrier();  // the staging reads of this tile are done before the next tile's X rows land in the same LDS
  }
  if constexpr (KO == 9) {
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < 7; ++q) atomicAdd((unsigned long long*)st.out + q, acc[q]);
    }
  }
}

// ---- weight stream packing: fp32 row-major expert weights -> the per-(group, wave) fragment stream ----------------------
// One fragment = 64 lanes x 8 elements: lane l holds row (l & 15), k = 8 (l >> 4) + e of a 16 x 32 block (the MFMA A operand).
// With f1 = 2 Din/32 (phase-1 fragments per chunk), q = 4 NJ (phase-2 fragments per half), C = F / 256 chunks, the stream of
// (group g, wave w) is, in this order:
//     W1(0)  H0(0)  |  W1(1)  H1(0)  H0(1)  |  W1(2)  H1(1)  H0(2)  | ... |  W1(C-1)  H1(C-2)  H0(C-1)  |  H1(C-1)
//   W1(c), fragment f: step = f >> 1, j = f & 1:   W1[g][256 c + 32 w + 16 j + (l & 15)][32 step + 8 (l >> 4) + e]
//   Hh(c), fragment f: s = f / NJ, j = f % NJ, hidden-image position p = 32 s + 8 (l >> 4) + e of half h:
//                                                   W2[g][16 NJ w + 16 j + (l & 15)][256 c + 32 (p >> 4) + 16 h + (p & 15)]
//   (the half image holds the 16 units wave 0 produced for that half, then wave 1's, ...: unit 32 w' + 16 h + u at position
//   16 w' + u)
template <typename HT>
__global__ __launch_bounds__(256) void mlp_stream_pack_kernel(const float* w1, const float* w2, int G, int F, int Din, int Dout,
                                                              uint16_t* out) {
  const int NJ = Dout / 128, nchunk = F / FC, f1 = 2 * (Din / 32), q = 4 * NJ, fpc = f1 + 2 * q;
  const int64_t nfrag = (int64_t)G * 8 * nchunk * fpc;
  for (int64_t fi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); fi < nfrag; fi += (int64_t)gridDim.x * 4) {
    const int l = threadIdx.x & 63;
    const int f = (int)(fi % ((int64_t)nchunk * fpc));  // position in the wave's stream
    const int w = (int)((fi / ((int64_t)nchunk * fpc)) % 8);
    const int gi = (int)(fi / ((int64_t)nchunk * fpc) / 8);
    int c, loc;  // block (chunk) and position inside it
    if (f < f1 + q) {
      c = 0, loc = f;
    } else {
      c = 1 + (f - (f1 + q)) / fpc, loc = (f - (f1 + q)) % fpc;
    }
    const float* src;
    int hh = -1, cc = c, fp = 0;  // phase-2 fragment: half, chunk, index in the half
    if (c == nchunk) {
      hh = 1, cc = nchunk - 1, fp = loc;  // the tail H1(C-1)
    } else if (loc >= f1) {
      if (c == 0) hh = 0, fp = loc - f1;
      else if (loc < f1 + q) hh = 1, cc = c - 1, fp = loc - f1;
      else hh = 0, fp = loc - f1 - q;
    }
    if (hh < 0) {
      const int step = loc >> 1, j = loc & 1;
      src = w1 + ((int64_t)gi * F + 256 * c + 32 * w + 16 * j + (l & 15)) * Din + 32 * step + 8 * (l >> 4);
    } else {
      const int s = fp / NJ, j = fp % NJ, p = 32 * s + 8 * (l >> 4);
      src = w2 + ((int64_t)gi * Dout + 16 * NJ * w + 16 * j + (l & 15)) * F + 256 * cc + 32 * (p >> 4) + 16 * hh + (p & 15);
    }
    uint4 o;
    o.x = HT::pack(src[0], src[1]), o.y = HT::pack(src[2], src[3]), o.z = HT::pack(src[4], src[5]), o.w = HT::pack(src[6], src[7]);
    *(uint4*)(out + fi * 512 + l * 8) = o;
  }
}

int device_cus() {
  static int cus[64] = {};
  const int dev = dev_ordinal();
  if (!cus[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}

}  // namespace

int64_t mlp_stream_elems(int G, int F, int Din, int Dout) {
  // + 16 fragments: the ring's last refills of the last wave of the last group read (and discard) past the stream's end
  return (int64_t)G * ((int64_t)F * Din + (int64_t)Dout * F) + 16 * 512;
}

int mlp_stream_pack(const float* w1, const float* w2, int G, int F, int Din, int Dout, int h16, uint16_t* out, hipStream_t stream) {
  if (!w1 || !w2 || !out || G < 1 || F < FC || (F % FC) || Din < 128 || (Din % 128) || (Dout != 512 && Dout != 1024)) return MDM_ERR_UNSUPPORTED;
  const int64_t body = (int64_t)G * ((int64_t)F * Din + (int64_t)Dout * F);
  if (hipMemsetAsync(out + body, 0, 16 * 512 * sizeof(uint16_t), stream) != hipSuccess) return MDM_ERR_LAUNCH;
  const int blocks = (int)((body / 512 + 3) / 4 < 4096 ? (body / 512 + 3) / 4 : 4096);
  if (h16 == MDM_H16_F16) {
    hipLaunchKernelGGL(mlp_stream_pack_kernel<HF>, dim3(blocks), dim3(256), 0, stream, w1, w2, G, F, Din, Dout, out);
  } else {
    hipLaunchKernelGGL(mlp_stream_pack_kernel<HB>, dim3(blocks), dim3(256), 0, stream, w1, w2, G, F, Din, Dout, out);
  }
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

bool fused_mlp_stream_supported(const MdmMlpDesc& a) {
  const bool small = a.Dout == 512 && (a.Din == 128 || a.Din == 256 || a.Din == 512), big = a.Dout == 1024 && a.Din == 1024;
  if (!a.wstream || !(small || big) || a.F < FC || (a.F % FC) || a.M < 1) return false;
  if (a.goff && (a.ngroups < 1 || a.ngroups > 64)) return false;
  if ((a.ldx % 8) || ((((uintptr_t)a.X) | ((uintptr_t)a.wstream)) & 15) || (a.wstream_gs % 8)) return false;
  if ((a.ldc & 3) || (a.R1 && (a.ldr1 & 3)) || (a.R2 && (a.ldr2 & 3))) return false;
  if ((a.b1 && ((((uintptr_t)a.b1) & 15) || (a.b1_gs & 3))) || (a.b2 && ((((uintptr_t)a.b2) & 15) || (a.b2_gs & 3)))) return false;
  return true;
}

// tile height: the tiles of all groups should fill the CUs in whole rounds
int mlp_stream_tile_h(int64_t M, int rt_max) {
  const int cus = device_cus();
  const int64_t cap = (int64_t)cus * rt_max * 16;
  const int64_t rounds = (M + cap - 1) / cap;
  int64_t h = (M + cus * rounds - 1) / (cus * rounds);
  h = (h + 15) & ~15ll;
  if (h < 16) h = 16;
  if (h > rt_max * 16) h = rt_max * 16;
  return (int)h;
}

extern int g_bf16_variant;
#ifdef MDM_DIAG
unsigned long long* g_diag_counters = nullptr;  // mdm_diag_mlp_counters(): eight 64-bit device counters of the stamped build
#endif

template <int RT, int DIN, int KO, int NJ = 4>
static int launch_stream(const MdmMlpDesc& a, int th, hipStream_t stream, const PairTail& tail = PairTail()) {
  constexpr int smem = XGeo<RT, DIN, NJ, KO == 10>::SMEM + 512;  // + the group-offset table
  static DevOnce attr;
  if (smem > 65536 && !attr) {
    if (hipFuncSetAttribute((const void*)fused_mlp_stream_kernel<HB, RT, NJ, DIN, KO>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess ||
        hipFuncSetAttribute((const void*)fused_mlp_stream_kernel<HF, RT, NJ, DIN, KO>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return MDM_ERR_LAUNCH;
    attr = true;
  }
  const int tiles = (int)(a.M / th) + (a.goff ? a.ngroups : 1);  // upper bound of the tile count
  // persistent: the resident workgroups (one per CU) walk the tiles
  const int res = device_cus();
  const int grid = tiles < res ? tiles : res;
  if (a.h16 == MDM_H16_F16) {
    hipLaunchKernelGGL((fused_mlp_stream_kernel<HF, RT, NJ, DIN, KO>), dim3(grid), dim3(NT), smem, stream, a, th, tail);
  } else {
    hipLaunchKernelGGL((fused_mlp_stream_kernel<HB, RT, NJ, DIN, KO>), dim3(grid), dim3(NT), smem, stream, a, th, tail);
  }
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

// row tiles per workgroup by tile height: 32-row (RT 2) and 64-row (RT 4) forms for launches with few rows per CU (the dense
// Linear-GELU-Linear pairs: 6272 / 12544 rows), 112-row form (RT 7) for the expert MLPs
template <int DIN>
static int launch_by_height(const MdmMlpDesc& a, hipStream_t stream) {
  const int th = mlp_stream_tile_h(a.M, 7);
  if (th <= 32) return launch_stream<2, DIN, 0>(a, th, stream);
  if (th <= 64) return launch_stream<4, DIN, 0>(a, th, stream);
  return launch_stream<7, DIN, 0>(a, th, stream);
}

int fused_mlp_stream(const MdmMlpDesc& a, hipStream_t stream) {
  if (!a.X || (!a.C && !a.C16)) return MDM_ERR_ARG;
  if (!fused_mlp_stream_supported(a)) return MDM_ERR_UNSUPPORTED;
  if (a.Dout == 1024) {  // big model (transformer.py:188-192 doubles the widths): 64-row tiles, 8 output fragments per wave and K step
    const int th = mlp_stream_tile_h(a.M, 4);
    return launch_stream<4, 1024, 0, 8>(a, th, stream);
  }
  if (a.Din == 128) return launch_by_height<128>(a, stream);
  if (a.Din == 256) return launch_by_height<256>(a, stream);
#ifdef MDM_DIAG
  const int th = mlp_stream_tile_h(a.M, 7);
  if (th > 64) {
    switch (g_bf16_variant) {  // knobs 41..49: knock-out / diagnostic builds for tools/mlp_ko.py, tools/mlp_stamps.py (timing only)
      case 41: return launch_stream<7, 512, 1>(a, th, stream);
      case 42: return launch_stream<7, 512, 2>(a, th, stream);
      case 44: return launch_stream<7, 512, 4>(a, th, stream);
      case 45: return launch_stream<7, 512, 5>(a, th, stream);
      case 46: return launch_stream<7, 512, 6>(a, th, stream);
      case 47: return launch_stream<7, 512, 7>(a, th, stream);
      case 48: return launch_stream<7, 512, 8>(a, th, stream);
      case 49: {
        if (!g_diag_counters) return MDM_ERR_ARG;  // mdm_diag_mlp_counters() first: the stamps need a buffer of their own
        PairTail t = PairTail();
        t.out = (float*)g_diag_counters;
        return launch_stream<7, 512, 9>(a, th, stream, t);
      }
      default: break;
    }
  }
#endif
  return launch_by_height<512>(a, stream);
}

// The Performer's proj_out pair AND its tail (post_norm, stylization, out_layers.2, residual; optionally the LayerNorm of the
// block that follows) in one launch: a = the pair (dense, D = 512, no residuals / row scale / outputs of its own), t = the tail.
bool fused_pair_style_supported(const MdmMlpDesc& a) {
  return fused_mlp_stream_supported(a) && a.Din == 512 && a.Dout == 512 && !a.goff && !a.gather && !a.rowscale && !a.R1 && !a.R2;
}

int fused_pair_style(const MdmMlpDesc& a, const PairTail& t, hipStream_t stream) {
  if (!a.X || !fused_pair_style_supported(a)) return MDM_ERR_UNSUPPORTED;
  if (!t.pw || !t.pb || !t.sw || !t.sb || !t.sc || t.S <= 0 || !t.ws || !t.bias || !t.resid || !t.out || (t.ln16 && (!t.lw || !t.lb)) ||
      ((uintptr_t)t.ws & 15) || (t.skip && (!t.lw || !t.lb)) || (t.l2w && (!t.l2b || !t.ln16 || !t.skip)))
    return MDM_ERR_ARG;
  const int th = mlp_stream_tile_h(a.M, 4);  // the tail stages fp32 rows: 64-row tiles at most
  if (th <= 32) return launch_stream<2, 512, 10>(a, th, stream, t);
  return launch_stream<4, 512, 10>(a, th, stream, t);
}

}  // namespace mdm
