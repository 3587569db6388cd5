// StylizationBlock in ONE launch (stylization.py:20-31 behind its callers' row-wise heads):
//     s   = SiLU( LN_style(a) * (1 + scale[b]) + shift[b] )                       a per row, as csrc/rowwise.hip style_in:
//           a = x | normalize(LN_post(x)) * sqrt(D) (Performer tail, fast_attention.py:169-172) | 0.5 * sum of the token's four
//           routed expert rows (MoE combine, switch_moe.py:109 + multi_branch.py:58-59)
//     out = resid + out_scale * colscale * ( s Wout^T + b )                        (out_layers.2 and the caller's residual)
// replaces the style_in launch + the D x D GEMM launch behind it (4 pairs per decoder layer, 32 per forward) and the 16-bit
// [M, D] tensor between them.  D = 512, 16-bit modes.
//
// One workgroup (8 waves, <= 128 registers: two resident per CU) owns 32 whole rows:
//   * row phase: wave w brings rows 4 w .. 4 w + 3 exactly as the row-wise kernel does (one wave per row, DPP reductions) and
//     leaves them as 16-bit MFMA rows in a 32-KiB LDS image (16-B chunk c of row r at slot c ^ (r & 15));
//   * GEMM phase: wave w owns output columns [64 w, 64 w + 64): the image is the shared operand, Wout is streamed global ->
//     registers from a packed fragment stream (mdm_gemm_stream_pack: per wave the 64 fragments of its columns in K order) through
//     an 8-fragment ring, as in csrc/mlp_stream.hip.  With 32 rows a weight fragment feeds only two MFMAs, so the launch is bound
//     by the 512 KiB of Wout every workgroup streams (98 / 196 MB per launch, L2-resident), not by the matrix pipe;
//   * epilogue: staged as fp32 rows through LDS, full-row stores, residual read coalesced.
#include "gemm.h"
#include "kernels.h"
#include "row.h"

namespace mdm {
namespace {

constexpr int SG_NT = 512, SG_D = 512, SG_NJ = 4;
// RT row tiles (16 rows each) per workgroup: 2 -> 32 rows, <= 128 registers, two workgroups per CU; 4 -> 64 rows, one per CU (half
// the weight bytes per row).  LDS: the fp32 staging of the epilogue (rows x 2 KiB) covers the 16-bit image (rows x 1 KiB).

__device__ __forceinline__ void sg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

struct StyleGemmArgs {
  const void* src;      // fp32 [*, D] or 16-bit rows (format of the launch)
  int64_t M;
  int S;                // frames per sample: row m belongs to sample m / S
  const float *pw, *pb; // optional post_norm (+ L2 normalisation * sqrt(D))
  const float *sw, *sb; // style norm
  const float* sc;      // (B, 2 D) scale | shift
  const int* pos4;      // optional (M, 4): the row is 0.5 * the sum of these four source rows
  const uint16_t* ws;   // weight stream of out_layers.2
  const float* bias;
  const float* resid;   // optional [M, D]
  float out_scale;
  const float* colscale;  // optional [D]
  float* out;
  uint16_t* out16;      // optional 16-bit copy
  // fp32-grade form only (style_gemm3): what may follow the row while it is still on chip.  With r = resid + out_scale * colscale * (...)
  //   skip == NULL: out = r;  ln_out = LN(r; lw, lb) when lw is set                  (the pre-norm of the block that follows)
  //   skip != NULL: out = LN(skip + skip_scale * r; lw, lb)  (r itself is not written);  ln_out = LN(out; l2w, l2b) when l2w is set
  //                 (the tail of DualSelfAttentionBlock behind its second Performer, fast_attention.py:219-225)
  const float *lw, *lb;
  float* ln_out;
  const float* skip;
  float skip_scale;
  const float *l2w, *l2b;
  int ln_x2;  // ln_out as pre-split rows (MDM_OP_X2_ROW) for the GEMM that reads it, instead of fp32
#ifdef MDM_DIAG
  int ko;  // diagnostic library only (knobs 74..77, tools/style_ko.sh): 1 no row phase, 2 no K loop, 3 no output stores, 4 no weight refills
#endif
};

template <typename HT, bool SRC16, int SG_RT>
__global__ __launch_bounds__(SG_NT, (SG_RT <= 2 ? 4 : 2)) void style_gemm_kernel(const StyleGemmArgs g) {
  constexpr int SG_ROWS = 16 * SG_RT, RPW = SG_ROWS / 8;  // rows per wave in the row phase
  typedef typename HT::frag_t frag_t;
  typedef Row<8, true> R8;
  constexpr int D = SG_D, FMT = HT::FMT;
  extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t row0 = (int64_t)blockIdx.x * SG_ROWS;

  // the wave's weight stream: 64 fragments of 1 KiB; the ring's first 8 are requested before the row phase
  const uint8_t* wp = (const uint8_t*)g.ws + (int64_t)wn * 64 * 1024 + lane * 16;
  constexpr int NR = SG_RT <= 2 ? 16 : 8;  // ring depth: with 32 accumulator registers there is room for 16 fragments in flight
  frag_t R[NR];  // the first 8 are requested before the row phase, the rest behind it (the row phase needs the registers)
#pragma unroll
  for (int f = 0; f < 8; ++f) R[f] = *(const frag_t*)(wp + f * 1024);

  // ---- row phase (csrc/rowwise.hip style_in_rows, one wave per row) ------------------------------------------------------
#ifdef MDM_DIAG
  if (g.ko != 1)
#endif
  {
    R8 pww, pbb, sww, sbb;
    if (g.pw) pww.load(g.pw, D, lane), pbb.load(g.pb, D, lane);
    sww.load(g.sw, D, lane), sbb.load(g.sb, D, lane);
#pragma unroll 2
    for (int q = 0; q < RPW; ++q) {
      const int rl = RPW * wn + q;
      int64_t row = row0 + rl;
      row = row < g.M ? row : g.M - 1;  // rows past the end: recomputed copies of the last row, never stored
      const float* scb = g.sc + (row / g.S) * 2 * (int64_t)D;
      R8 r, scale, shift;
      scale.load(scb, D, lane);
      shift.load(scb + D, D, lane);
      if (g.pos4) {
        const int p0 = g.pos4[row * 4 + 0], p1 = g.pos4[row * 4 + 1], p2 = g.pos4[row * 4 + 2], p3 = g.pos4[row * 4 + 3];
        R8 a, b, c, d;
        if constexpr (SRC16) {
          const uint16_t* xh = (const uint16_t*)g.src;
          a.template load_h16<FMT>(xh + (int64_t)p0 * D, D, lane);
          b.template load_h16<FMT>(xh + (int64_t)p1 * D, D, lane);
          c.template load_h16<FMT>(xh + (int64_t)p2 * D, D, lane);
          d.template load_h16<FMT>(xh + (int64_t)p3 * D, D, lane);
        } else {
          const float* x = (const float*)g.src;
          a.load(x + (int64_t)p0 * D, D, lane);
          b.load(x + (int64_t)p1 * D, D, lane);
          c.load(x + (int64_t)p2 * D, D, lane);
          d.load(x + (int64_t)p3 * D, D, lane);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) r.e[j] = ((a.e[j] + b.e[j]) + (c.e[j] + d.e[j])) * 0.5f;
      } else {
        if constexpr (SRC16) {
          r.template load_h16<FMT>((const uint16_t*)g.src + row * D, D, lane);
        } else {
          r.load((const float*)g.src + row * D, D, lane);
        }
      }
      if (g.pw) {
        r.layernorm(pww, pbb, D, lane);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += r.e[j] * r.e[j];
        const float inv = sqrtf((float)D) / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
#pragma unroll
        for (int j = 0; j < 8; ++j) r.e[j] *= inv;
      }
      r.layernorm(sww, sbb, D, lane);
#pragma unroll
      for (int j = 0; j < 8; ++j) r.e[j] = silu(r.e[j] * (1.f + scale.e[j]) + shift.e[j]);
      // lane holds columns 4 l .. 4 l + 3 and 256 + 4 l ..: 8 bytes each, 16-B chunks (l >> 1) and 32 + (l >> 1), half l & 1
      uint8_t* ir = smem + rl * 1024 + (lane & 1) * 8;
      *(uint2*)(ir + ((((lane >> 1)) ^ (rl & 15)) << 4)) = make_uint2(HT::pack(r.e[0], r.e[1]), HT::pack(r.e[2], r.e[3]));
      *(uint2*)(ir + (((32 + (lane >> 1)) ^ (rl & 15)) << 4)) = make_uint2(HT::pack(r.e[4], r.e[5]), HT::pack(r.e[6], r.e[7]));
    }
  }
#pragma unroll
  for (int f = 8; f < NR; ++f) R[f] = *(const frag_t*)(wp + f * 1024);
  wp += NR * 1024;
  sg_barrier();

  // ---- GEMM phase: y[32 rows x 64 columns of this wave] = image . Wout^T ---------------------------------------------------
  const int frow = lane & 15, fq = lane >> 4;
  f32x4 y[SG_RT][SG_NJ];
#pragma unroll
  for (int i = 0; i < SG_RT; ++i)
#pragma unroll
    for (int j = 0; j < SG_NJ; ++j) y[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const int xb = frow * 1024 + ((fq ^ frow) << 4);  // (row frow, K step 0); step s reads chunk (4 s) ^ (fq ^ frow)
    frag_t A[2][SG_RT];
#pragma unroll
    for (int i = 0; i < SG_RT; ++i) A[0][i] = *(const frag_t*)(smem + xb + i * 16384);
#ifdef MDM_DIAG
    const int ksteps = g.ko == 2 ? 0 : 16;
    const bool refill = g.ko != 4;
#pragma unroll 1
    for (int s0 = 0; s0 < ksteps; s0 += 16)
#endif
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s + 1 < 16) {
#pragma unroll
        for (int i = 0; i < SG_RT; ++i) A[(s + 1) & 1][i] = *(const frag_t*)(smem + (xb ^ (64 * ((s + 1) & 3))) + ((s + 1) >> 2) * 256 + i * 16384);
      }
#pragma unroll
      for (int j = 0; j < SG_NJ; ++j) {
        const int slot = (s * SG_NJ + j) & (NR - 1);
#pragma unroll
        for (int i = 0; i < SG_RT; ++i) y[i][j] = HT::mfma16(R[slot], A[s & 1][i], y[i][j]);
#ifdef MDM_DIAG
        if (refill)
#endif
        R[slot] = *(const frag_t*)(wp + slot * 1024);  // the last NR refills read (and discard) the next wave's / the padding
        if (slot == NR - 1) wp += NR * 1024;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: (y + b) * out_scale * colscale staged as fp32 [rows][512], full rows out with the residual -----------------
  sg_barrier();
  float* stg = (float*)smem;
#pragma unroll
  for (int j = 0; j < SG_NJ; ++j) {
    const int n = wn * 64 + 16 * j + 4 * fq;
    f32x4 bb = *(const f32x4*)(g.bias + n);
    f32x4 cs = {g.out_scale, g.out_scale, g.out_scale, g.out_scale};
    if (g.colscale) {
      const f32x4 q = *(const f32x4*)(g.colscale + n);
      cs[0] *= q[0], cs[1] *= q[1], cs[2] *= q[2], cs[3] *= q[3];
    }
#pragma unroll
    for (int i = 0; i < SG_RT; ++i) {
      const int ml = i * 16 + frow;
      f32x4 v = y[i][j];
      v[0] = (v[0] + bb[0]) * cs[0], v[1] = (v[1] + bb[1]) * cs[1], v[2] = (v[2] + bb[2]) * cs[2], v[3] = (v[3] + bb[3]) * cs[3];
      *(f32x4*)(stg + ml * D + (((n >> 2) ^ (ml & 31)) << 2)) = v;
    }
  }
  // the thread's residual pieces are all requested before its first store: the residual is usually updated in place (out == resid), so
  // the compiler may not move a row's load above the previous row's store, and on gfx950 stores count in vmcnt -- left in the loop
  // every row waited for the row before it to be written (8 serial write round trips per tile)
  const int cl = tid & 127, n = 4 * cl;
  f32x4 q[SG_ROWS / 4];
#pragma unroll
  for (int k = 0; k < SG_ROWS / 4; ++k) {
    int64_t m = row0 + (tid >> 7) + 4 * k;
    m = m < g.M ? m : g.M - 1;
    q[k] = g.resid ? *(const f32x4*)(g.resid + m * D + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  sg_barrier();
#pragma unroll
  for (int k = 0; k < SG_ROWS / 4; ++k) {
    const int ml = (tid >> 7) + 4 * k;
    const int64_t m = row0 + ml;
    if (m >= g.M) continue;
    f32x4 v = *(const f32x4*)(stg + ml * D + ((cl ^ (ml & 31)) << 2));
    if (g.resid) v[0] += q[k][0], v[1] += q[k][1], v[2] += q[k][2], v[3] += q[k][3];
#ifdef MDM_DIAG
    if (g.ko == 3 && v[0] != 12345.678f) continue;
#endif
    *(f32x4*)(g.out + m * D + n) = v;
    if (g.out16) *(uint2*)(g.out16 + m * D + n) = make_uint2(HT::pack(v[0], v[1]), HT::pack(v[2], v[3]));
  }
}

// ---- the same launch in the fp32-grade (bf16x3) arithmetic ---------------------------------------------------------------------
// Row phase in fp32 exactly as above (the source rows are fp32 in these modes); the rows are left as TWO 16-bit images (bf16 hi and
// lo = rn(x - hi)) and every product is three MFMAs (Wh Al + Wl Ah + Wh Ah, small terms first as csrc/gemm3.hip).  Wout streams as
// (hi, lo) fragment PAIRS (mdm_gemm_stream3_pack: per wave the 64 pairs of its 64 columns in K order, 2 KiB per pair).  A weight
// fragment pair feeds 3 RT MFMAs instead of RT: at the same stream rate three times the matrix work of the 16-bit launch, which is
// what this mode's arithmetic costs anyway -- the launch replaces style_in + a 128 x 128-tile bf16x3 GEMM (whose fp32 operand
// tiles cross L2 -> LDS four times) and the fp32 [M, D] tensor between them.
template <int SG_RT>
__global__ __launch_bounds__(SG_NT, (SG_RT <= 2 ? 4 : 2)) void style_gemm3_kernel(const StyleGemmArgs g) {
  constexpr int SG_ROWS = 16 * SG_RT, RPW = SG_ROWS / 8;
  typedef HB::frag_t frag_t;
  typedef Row<8, true> R8;
  constexpr int D = SG_D, IMG = SG_ROWS * 1024;
  extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
  uint8_t* const imh = smem;
  uint8_t* const iml = smem + IMG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t row0 = (int64_t)blockIdx.x * SG_ROWS;

  // the wave's stream: 64 (hi, lo) pairs = 128 fragments of 1 KiB; fragment f of the stream refills ring slot f % NR
  const uint8_t* wp = (const uint8_t*)g.ws + (int64_t)wn * 128 * 1024 + lane * 16;
  constexpr int NR = SG_RT <= 2 ? 8 : 16;
  frag_t R[NR];
#pragma unroll
  for (int f = 0; f < (NR < 8 ? NR : 8); ++f) R[f] = *(const frag_t*)(wp + f * 1024);

  {
    R8 pww, pbb, sww, sbb;
    if (g.pw) pww.load(g.pw, D, lane), pbb.load(g.pb, D, lane);
    sww.load(g.sw, D, lane), sbb.load(g.sb, D, lane);
#pragma unroll 2
    for (int q = 0; q < RPW; ++q) {
      const int rl = RPW * wn + q;
      int64_t row = row0 + rl;
      row = row < g.M ? row : g.M - 1;
      const float* scb = g.sc + (row / g.S) * 2 * (int64_t)D;
      R8 r, scale, shift;
      scale.load(scb, D, lane);
      shift.load(scb + D, D, lane);
      const float* x = (const float*)g.src;
      if (g.pos4) {
        const int p0 = g.pos4[row * 4 + 0], p1 = g.pos4[row * 4 + 1], p2 = g.pos4[row * 4 + 2], p3 = g.pos4[row * 4 + 3];
        R8 a, b, c, d;
        a.load(x + (int64_t)p0 * D, D, lane);
        b.load(x + (int64_t)p1 * D, D, lane);
        c.load(x + (int64_t)p2 * D, D, lane);
        d.load(x + (int64_t)p3 * D, D, lane);
#pragma unroll
        for (int j = 0; j < 8; ++j) r.e[j] = ((a.e[j] + b.e[j]) + (c.e[j] + d.e[j])) * 0.5f;
      } else {
        r.load(x + row * D, D, lane);
      }
      if (g.pw) {
        r.layernorm(pww, pbb, D, lane);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += r.e[j] * r.e[j];
        const float inv = sqrtf((float)D) / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
#pragma unroll
        for (int j = 0; j < 8; ++j) r.e[j] *= inv;
      }
      r.layernorm(sww, sbb, D, lane);
#pragma unroll
      for (int j = 0; j < 8; ++j) r.e[j] = silu(r.e[j] * (1.f + scale.e[j]) + shift.e[j]);
      uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
      split_bf16(r.e[0], r.e[1], h0, l0);
      split_bf16(r.e[2], r.e[3], h1, l1);
      split_bf16(r.e[4], r.e[5], h2, l2);
      split_bf16(r.e[6], r.e[7], h3, l3);
      const int o0 = rl * 1024 + ((((lane >> 1)) ^ (rl & 15)) << 4) + (lane & 1) * 8;
      const int o1 = rl * 1024 + (((32 + (lane >> 1)) ^ (rl & 15)) << 4) + (lane & 1) * 8;
      *(uint2*)(imh + o0) = make_uint2(h0, h1), *(uint2*)(imh + o1) = make_uint2(h2, h3);
      *(uint2*)(iml + o0) = make_uint2(l0, l1), *(uint2*)(iml + o1) = make_uint2(l2, l3);
    }
  }
#pragma unroll
  for (int f = 8; f < NR; ++f) R[f] = *(const frag_t*)(wp + f * 1024);
  wp += NR * 1024;
  sg_barrier();

  const int frow = lane & 15, fq = lane >> 4;
  f32x4 y[SG_RT][SG_NJ];
#pragma unroll
  for (int i = 0; i < SG_RT; ++i)
#pragma unroll
    for (int j = 0; j < SG_NJ; ++j) y[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const int xb = frow * 1024 + ((fq ^ frow) << 4);
    frag_t Ah[2][SG_RT], Al[2][SG_RT];
#pragma unroll
    for (int i = 0; i < SG_RT; ++i) Ah[0][i] = *(const frag_t*)(imh + xb + i * 16384), Al[0][i] = *(const frag_t*)(iml + xb + i * 16384);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s + 1 < 16) {
#pragma unroll
        for (int i = 0; i < SG_RT; ++i) {
          const int off = (xb ^ (64 * ((s + 1) & 3))) + ((s + 1) >> 2) * 256 + i * 16384;
          Ah[(s + 1) & 1][i] = *(const frag_t*)(imh + off), Al[(s + 1) & 1][i] = *(const frag_t*)(iml + off);
        }
      }
#pragma unroll
      for (int j = 0; j < SG_NJ; ++j) {
        const int slot = (2 * (s * SG_NJ + j)) & (NR - 1);
#pragma unroll
        for (int i = 0; i < SG_RT; ++i) {
          y[i][j] = HB::mfma16(R[slot], Al[s & 1][i], y[i][j]);
          y[i][j] = HB::mfma16(R[slot + 1], Ah[s & 1][i], y[i][j]);
          y[i][j] = HB::mfma16(R[slot], Ah[s & 1][i], y[i][j]);
        }
        R[slot] = *(const frag_t*)(wp + slot * 1024);  // the last NR refills read (and discard) the next wave's / the padding
        R[slot + 1] = *(const frag_t*)(wp + (slot + 1) * 1024);
        if (slot + 2 == NR) wp += NR * 1024;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  sg_barrier();
  float* stg = (float*)smem;
#pragma unroll
  for (int j = 0; j < SG_NJ; ++j) {
    const int n = wn * 64 + 16 * j + 4 * fq;
    f32x4 bb = *(const f32x4*)(g.bias + n);
    f32x4 cs = {g.out_scale, g.out_scale, g.out_scale, g.out_scale};
    if (g.colscale) {
      const f32x4 q = *(const f32x4*)(g.colscale + n);
      cs[0] *= q[0], cs[1] *= q[1], cs[2] *= q[2], cs[3] *= q[3];
    }
#pragma unroll
    for (int i = 0; i < SG_RT; ++i) {
      const int ml = i * 16 + frow;
      f32x4 v = y[i][j];
      v[0] = (v[0] + bb[0]) * cs[0], v[1] = (v[1] + bb[1]) * cs[1], v[2] = (v[2] + bb[2]) * cs[2], v[3] = (v[3] + bb[3]) * cs[3];
      *(f32x4*)(stg + ml * D + (((n >> 2) ^ (ml & 31)) << 2)) = v;
    }
  }
  sg_barrier();
  {
    // one wave per row (lane: columns 4 l .. 4 l + 3 and 256 + 4 l ..): the residual / skip operands of the wave's rows are all
    // requested first (one memory round trip per wave), then row by row: residual, optional block tail and LayerNorms, stores
    R8 xr[RPW], sk[RPW];
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      int64_t m = row0 + RPW * wn + q;
      m = m < g.M ? m : g.M - 1;
      if (g.resid) xr[q].load(g.resid + m * D, D, lane);
      if (g.skip) sk[q].load(g.skip + m * D, D, lane);
    }
    R8 lww, lbb, l2ww, l2bb;
    if (g.lw) lww.load(g.lw, D, lane), lbb.load(g.lb, D, lane);
    if (g.l2w) l2ww.load(g.l2w, D, lane), l2bb.load(g.l2b, D, lane);
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      const int rl = RPW * wn + q;
      const int64_t m = row0 + rl;
      if (m >= g.M) continue;  // (wave-uniform)
      const f32x4 v0 = *(const f32x4*)(stg + rl * D + ((lane ^ (rl & 31)) << 2));
      const f32x4 v1 = *(const f32x4*)(stg + rl * D + (((64 + lane) ^ (rl & 31)) << 2));
      R8 r;
      r.e[0] = v0[0], r.e[1] = v0[1], r.e[2] = v0[2], r.e[3] = v0[3], r.e[4] = v1[0], r.e[5] = v1[1], r.e[6] = v1[2], r.e[7] = v1[3];
      if (g.resid) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r.e[j] += xr[q].e[j];
      }
      if (g.skip) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r.e[j] = sk[q].e[j] + g.skip_scale * r.e[j];
        r.layernorm(lww, lbb, D, lane);
        r.store(g.out + m * D, D, lane);
        if (g.l2w) {
          r.layernorm(l2ww, l2bb, D, lane);
          if (g.ln_x2) r.store_x2((uint16_t*)g.ln_out + m * 2 * D, D, lane);
          else r.store(g.ln_out + m * D, D, lane);
        }
      } else {
        r.store(g.out + m * D, D, lane);
        if (g.lw) {
          r.layernorm(lww, lbb, D, lane);
          if (g.ln_x2) r.store_x2((uint16_t*)g.ln_out + m * 2 * D, D, lane);
          else r.store(g.ln_out + m * D, D, lane);
        }
      }
    }
  }
}

// stream3[wave w][pair p = 4 s + j][plane hi | lo][lane l][8]:  bf16 hi / lo of W[64 w + 16 j + (l & 15)][32 s + 8 (l >> 4) + e]
__global__ __launch_bounds__(256) void gemm_stream3_pack_kernel(const float* w, uint16_t* out) {
  for (int fi = blockIdx.x * 4 + (threadIdx.x >> 6); fi < 8 * 64; fi += gridDim.x * 4) {
    const int l = threadIdx.x & 63, wv = fi >> 6, f = fi & 63, s = f >> 2, j = f & 3;
    const float* src = w + (int64_t)(64 * wv + 16 * j + (l & 15)) * SG_D + 32 * s + 8 * (l >> 4);
    uint4 h, lo;
    split_bf16(src[0], src[1], h.x, lo.x);
    split_bf16(src[2], src[3], h.y, lo.y);
    split_bf16(src[4], src[5], h.z, lo.z);
    split_bf16(src[6], src[7], h.w, lo.w);
    *(uint4*)(out + (int64_t)fi * 1024 + l * 8) = h;
    *(uint4*)(out + (int64_t)fi * 1024 + 512 + l * 8) = lo;
  }
}

// stream[wave w][fragment f = 4 s + j][lane l][8]:  W[64 w + 16 j + (l & 15)][32 s + 8 (l >> 4) + e]   (N = 512, K = 512)
template <typename HT>
__global__ __launch_bounds__(256) void gemm_stream_pack_kernel(const float* w, uint16_t* out) {
  for (int fi = blockIdx.x * 4 + (threadIdx.x >> 6); fi < 8 * 64; fi += gridDim.x * 4) {
    const int l = threadIdx.x & 63, wv = fi >> 6, f = fi & 63, s = f >> 2, j = f & 3;
    const float* src = w + (int64_t)(64 * wv + 16 * j + (l & 15)) * SG_D + 32 * s + 8 * (l >> 4);
    uint4 o;
    o.x = HT::pack(src[0], src[1]), o.y = HT::pack(src[2], src[3]), o.z = HT::pack(src[4], src[5]), o.w = HT::pack(src[6], src[7]);
    *(uint4*)(out + (int64_t)fi * 512 + l * 8) = o;
  }
}

}  // namespace

int64_t gemm_stream_elems(int N, int K) { return (N == SG_D && K == SG_D) ? (int64_t)N * K + 16 * 512 : 0; }  // + the ring's overrun

int gemm_stream_pack(const float* w, int N, int K, int h16, uint16_t* out, hipStream_t stream) {
  if (!w || !out || N != SG_D || K != SG_D) return MDM_ERR_UNSUPPORTED;
  if (hipMemsetAsync(out + (int64_t)N * K, 0, 16 * 512 * sizeof(uint16_t), stream) != hipSuccess) return MDM_ERR_LAUNCH;
  if (h16 == MDM_H16_F16) {
    hipLaunchKernelGGL(gemm_stream_pack_kernel<HF>, dim3(128), dim3(256), 0, stream, w, out);
  } else {
    hipLaunchKernelGGL(gemm_stream_pack_kernel<HB>, dim3(128), dim3(256), 0, stream, w, out);
  }
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

bool style_gemm_supported(int D, int64_t M) { return D == SG_D && M > 0 && M / 32 < (1ll << 30); }

extern int g_bf16_variant;

template <int RT>
static int launch_style_gemm(const StyleGemmArgs& g, bool src16, int h16, hipStream_t s) {
  constexpr int smem = 16 * RT * SG_D * 4;
  static DevOnce attr;
  if (!attr) {
    const void* fns[4] = {(const void*)style_gemm_kernel<HB, true, RT>, (const void*)style_gemm_kernel<HB, false, RT>,
                          (const void*)style_gemm_kernel<HF, true, RT>, (const void*)style_gemm_kernel<HF, false, RT>};
    for (const void* fn : fns)
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return MDM_ERR_LAUNCH;
    attr = true;
  }
  const dim3 grid((unsigned)((g.M + 16 * RT - 1) / (16 * RT)));
  if (h16 == MDM_H16_F16) {
    if (src16) hipLaunchKernelGGL((style_gemm_kernel<HF, true, RT>), grid, dim3(SG_NT), smem, s, g);
    else hipLaunchKernelGGL((style_gemm_kernel<HF, false, RT>), grid, dim3(SG_NT), smem, s, g);
  } else {
    if (src16) hipLaunchKernelGGL((style_gemm_kernel<HB, true, RT>), grid, dim3(SG_NT), smem, s, g);
    else hipLaunchKernelGGL((style_gemm_kernel<HB, false, RT>), grid, dim3(SG_NT), smem, s, g);
  }
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

// src_fmt: 0 = fp32 source rows, else the launch's 16-bit format (must equal h16)
int style_gemm(const void* src, int src_fmt, int64_t M, int D, int S, const float* pw, const float* pb, const float* sw, const float* sb,
               const float* sc, const int* pos4, const uint16_t* ws, const float* bias, const float* resid, float out_scale,
               const float* colscale, float* out, uint16_t* out16, int h16, hipStream_t s) {
  if (M <= 0) return MDM_OK;
  if (!style_gemm_supported(D, M)) return MDM_ERR_UNSUPPORTED;
  if (!src || !sw || !sb || !sc || !ws || !bias || !out || S <= 0 || (pw && !pb)) return MDM_ERR_ARG;
  if ((h16 != MDM_H16_BF16 && h16 != MDM_H16_F16) || (src_fmt != 0 && src_fmt != h16) || ((uintptr_t)ws & 15)) return MDM_ERR_ARG;
  StyleGemmArgs g = {};
  g.src = src, g.M = M, g.S = S, g.pw = pw, g.pb = pb, g.sw = sw, g.sb = sb, g.sc = sc, g.pos4 = pos4, g.ws = ws, g.bias = bias;
  g.resid = resid, g.out_scale = out_scale, g.colscale = colscale, g.out = out, g.out16 = out16;
#ifdef MDM_DIAG
  g.ko = (g_bf16_variant >= 74 && g_bf16_variant <= 77) ? g_bf16_variant - 73 : 0;
#endif
  // knob 29: 64-row tiles (one workgroup per CU, half the weight bytes per row) -- measured 1 % of a step SLOWER than two
  // co-resident 32-row workgroups per CU at 12544 rows; a row's arithmetic does not depend on the tile height
  if (g_bf16_variant == 29) return launch_style_gemm<4>(g, src_fmt != 0, h16, s);
  return launch_style_gemm<2>(g, src_fmt != 0, h16, s);
}

// ---- fp32-grade form ---------------------------------------------------------------------------------------------------------
int64_t gemm_stream3_elems(int N, int K) { return (N == SG_D && K == SG_D) ? 2 * (int64_t)N * K + 16 * 512 : 0; }

int gemm_stream3_pack(const float* w, int N, int K, uint16_t* out, hipStream_t stream) {
  if (!w || !out || N != SG_D || K != SG_D) return MDM_ERR_UNSUPPORTED;
  if (hipMemsetAsync(out + 2 * (int64_t)N * K, 0, 16 * 512 * sizeof(uint16_t), stream) != hipSuccess) return MDM_ERR_LAUNCH;
  hipLaunchKernelGGL(gemm_stream3_pack_kernel, dim3(128), dim3(256), 0, stream, w, out);
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

template <int RT>
static int launch_style_gemm3(const StyleGemmArgs& g, hipStream_t s) {
  constexpr int smem = 16 * RT * SG_D * 4;  // hi | lo images = the fp32 staging of the epilogue
  static DevOnce attr;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)style_gemm3_kernel<RT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return MDM_ERR_LAUNCH;
    attr = true;
  }
  const dim3 grid((unsigned)((g.M + 16 * RT - 1) / (16 * RT)));
  hipLaunchKernelGGL((style_gemm3_kernel<RT>), grid, dim3(SG_NT), smem, s, g);
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

// fp32 source rows, fp32 output; ws3 = the (hi, lo) pair stream of mdm_gemm_stream3_pack.  32-row tiles (two workgroups per CU) at
// every size: 64-row tiles (one per CU, half the weight bytes per row; knob 58) measured 1 % of a step slower at the full time scale
// (10.18 vs 10.06 ms) -- as in the 16-bit form, a second resident workgroup hides more than the halved stream saves.  A row's
// arithmetic does not depend on the tile height.
int style_gemm3(const float* src, int64_t M, int D, int S, const float* pw, const float* pb, const float* sw, const float* sb,
                const float* sc, const int* pos4, const uint16_t* ws3, const float* bias, const float* resid, float out_scale,
                const float* colscale, float* out, const StyleTail3& t, hipStream_t s) {
  if (M <= 0) return MDM_OK;
  if (!style_gemm_supported(D, M)) return MDM_ERR_UNSUPPORTED;
  if (!src || !sw || !sb || !sc || !ws3 || !bias || !out || S <= 0 || (pw && !pb) || ((uintptr_t)ws3 & 15)) return MDM_ERR_ARG;
  if ((t.lw && !t.lb) || (t.skip && (!t.lw || !t.lb)) || (t.l2w && (!t.l2b || !t.skip)) || ((t.skip ? t.l2w != nullptr : t.lw != nullptr) != (t.ln_out != nullptr)))
    return MDM_ERR_ARG;
  StyleGemmArgs g = {};
  g.src = src, g.M = M, g.S = S, g.pw = pw, g.pb = pb, g.sw = sw, g.sb = sb, g.sc = sc, g.pos4 = pos4, g.ws = ws3, g.bias = bias;
  g.resid = resid, g.out_scale = out_scale, g.colscale = colscale, g.out = out;
  g.lw = t.lw, g.lb = t.lb, g.ln_out = t.ln_out, g.skip = t.skip, g.skip_scale = t.skip_scale, g.l2w = t.l2w, g.l2b = t.l2b, g.ln_x2 = t.ln_x2;
  if (g_bf16_variant == 58) return launch_style_gemm3<4>(g, s);  // A/B knob: 64-row tiles
  return launch_style_gemm3<2>(g, s);
}

}  // namespace mdm
