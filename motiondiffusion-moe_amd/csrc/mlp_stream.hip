// Fused two-layer MLP with STREAMED weights for gfx950:   Y = epilogue( GELU(X W1^T + b1) W2^T + b2 )
// the expert MLPs of the MoE feed-forward (switch_moe.py:19-25,97-109: grouped by expert, gathered rows, gate-probability
// row scale) and the dense Linear-GELU-Linear pairs (fast_attention.py:121-126,293-299).
//
// Decomposition (what differs from csrc/mlp.hip, whose weights cross LDS behind one barrier per K step):
//   * one workgroup = 8 waves owns a tile of up to RT*16 rows and ALL Dout output columns.  The waves split the N axis
//     only: wave w owns hidden units [32 w, 32 w + 32) of every 256-unit hidden chunk (phase 1) and output columns
//     [16 NJ w, 16 NJ (w + 1)) (phase 2), for all rows of the tile.
//   * the ACTIVATION operand (X k-tiles in phase 1, the GELU'd hidden chunk in phase 2) is what the waves share: it lives in
//     LDS and every wave reads all of it (conflict-free XOR images, ds_read_b128 fragments);
//   * the WEIGHTS are private to a wave, so they never touch LDS: packed once as ONE linear stream of 1-KiB MFMA fragments
//     per (group, wave) in exactly the order the wave consumes them (mdm_mlp_stream_pack), they go global -> registers with
//     plain 16-byte loads through an 8-fragment register ring that runs 8 fragments (~0.9 k MFMA cycles) ahead.  No LDS-DMA,
//     no barrier and no LDS read for 94 % of the bytes a tile moves; phase 2 has no barrier at all;
//   * X k-tiles (64 k) are register-staged into a 3-stage LDS ring with ONE barrier per tile placed between the tile's two
//     32-wide K steps, so no barrier is followed by an LDS round trip that an MFMA waits for;
//   * tiles are balanced: the launch picks the tile height so that the tiles of all groups fill the CUs in whole rounds
//     (112-row tiles: 50176 routed rows = 448..464 tiles = 2 rounds at 7/8 of the rows a round could hold; the 128-row
//     tiles of mlp.hip made 392 tiles = 2 rounds, the second 53 % full).
// Register budget per lane (RT = 7, NJ = 4): y 112 + h 56 + weight ring 32 + A fragments + X staging 8.
#include "gemm.h"
#include "kernels.h"

namespace mdm {
namespace {

constexpr int NT = 512, FC = 256;

template <int RT>
struct SGeo {
  static constexpr int ROWS = RT * 16;
  static constexpr int HID_B = ROWS * FC * 2;  // one hidden chunk image, 16-bit [ROWS][256]: 512-B rows
  static constexpr int XS_B = ROWS * 128;      // one X k-tile, 16-bit [ROWS][64]: 128-B rows
  static constexpr int NXS = 3;
  static constexpr int SMEM = 2 * HID_B + NXS * XS_B;
};

template <typename T>
__device__ __forceinline__ T ldg(const uint8_t* p) {
  return *(const T*)p;
}

// Pins the weight refills where they are written: nothing may be scheduled across this point (a mask that lets LDS
// reads, MFMAs and ALU work through is no pin at all: the scheduler then moves exactly those above it).  Without it hipcc sinks all refills of an unrolled body to its end and the run-ahead
// (8 fragments) collapses to about one K step.
__device__ __forceinline__ void pin_vmem() { __builtin_amdgcn_sched_barrier(0); }

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Which tile this workgroup owns: groups are cut into ceil(rows / tile_h) tiles of equal height (a multiple of 16).  Lane e of
// every wave looks at group e (at most 64 groups); mt = tile index; false = no such tile.
__device__ __forceinline__ bool find_tile(const MdmMlpDesc& g, int tile_h, int lane, int mt, int& row0, int& row_end, int& grp) {
  const int ng = g.goff ? g.ngroups : 1;
  const int e = lane < ng ? lane : ng - 1;
  int b = 0, en = g.M;
  if (g.goff) b = g.goff[e], en = g.goff[e + 1];
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
    lds_barrier();
    constexpr int RPP = NT / CPR;  // rows per sweep of the workgroup
    const int cl = tid % CPR, n = 4 * cl;
#pragma unroll
    for (int k = 0; k < 16 * EPT / RPP; ++k) {
      const int ml = tid / CPR + RPP * k, tl = p * EPT * 16 + ml, m = row0 + tl;
      if (tl >= RT * 16 || m >= row_end) continue;
      f32x4 v = *(const f32x4*)(stg + ml * DOUT + ((cl ^ (ml & 31)) << 2));
      if (R1) {
        const f32x4 q = *(const f32x4*)(R1 + (int64_t)m * g.ldr1 + n);
        v[0] += g.r1_scale * q[0], v[1] += g.r1_scale * q[1], v[2] += g.r1_scale * q[2], v[3] += g.r1_scale * q[3];
      }
      if (R2) {
        const f32x4 q = *(const f32x4*)(R2 + (int64_t)m * g.ldr2 + n);
        v[0] += q[0], v[1] += q[1], v[2] += q[2], v[3] += q[3];
      }
      if (KO == 6 && v[0] != 123.456f) continue;
      if (g.C) *(f32x4*)(g.C + (int64_t)m * g.ldc + n) = v;
      if (g.C16) *(uint2*)(g.C16 + (int64_t)m * g.ldc + n) = make_uint2(HT::pack(v[0], v[1]), HT::pack(v[2], v[3]));
    }
  }
}

// KO: timing-only knock-outs for tools/mlp_bench.py (0 = the real kernel; results are wrong otherwise): 1 no GELU arithmetic,
// 2 no weight refills, 3 no X staging, 4 no phase-1 MFMAs, 5 no phase-2 MFMAs, 6 no output stores; 7 = the real kernel with the erf-form GELU of the LDS-staged kernel
template <typename HT, int RT, int NJ, int KO>
__global__ __launch_bounds__(NT, 2) void fused_mlp_stream_kernel(const MdmMlpDesc g, const int tile_h) {
  typedef typename HT::frag_t frag_t;
  typedef SGeo<RT> G;
  constexpr int DOUT = NJ * 128;
  extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
  uint8_t* const hid = smem;
  uint8_t* const xs = smem + 2 * G::HID_B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fq = lane >> 4;

  int row0, row_end, grp;
  // contiguous tile ranges per XCD: an XCD's L2 serves ~2 groups
  if (!find_tile(g, tile_h, lane, xcd_remap(blockIdx.x, gridDim.x), row0, row_end, grp)) return;

  const int nchunk = g.F / FC, nko = g.Din / 128, nkt = g.Din / 64;
  const int fpc = (g.Din / 32) * 2 + 8 * NJ;  // fragments per (wave, chunk): phase 1 then phase 2
  // the wave's weight stream: per-lane pointer at fragment 0 (+ 16 B per lane)
  const uint8_t* wp = (const uint8_t*)g.wstream + ((int64_t)grp * g.wstream_gs + (int64_t)wn * nchunk * fpc * 512) * 2 + lane * 16;
  const float* b1 = g.b1 ? g.b1 + (int64_t)grp * g.b1_gs + wn * 32 + fq * 4 : nullptr;
  const float* b2 = g.b2 ? g.b2 + (int64_t)grp * g.b2_gs + wn * (16 * NJ) + fq * 4 : nullptr;

  // ---- X staging: thread -> row tid >> 2, 32 B of each 128-B k-tile row; waves >= RT have no rows ---------------------
  const bool xact = wn < RT;
  const int xr = tid >> 2;
  const uint8_t* xp;
  {
    int srow = row0 + xr;
    srow = srow < row_end ? srow : row_end - 1;
    const int64_t src = g.gather ? (int64_t)g.gather[srow] : (int64_t)srow;
    xp = (const uint8_t*)(g.X + src * g.ldx) + (tid & 3) * 32;
  }
  const int xw0 = xr * 128 + ((((tid & 3) * 2) ^ (xr & 7)) << 4);
  const int xw1 = xr * 128 + ((((tid & 3) * 2 + 1) ^ (xr & 7)) << 4);
  // fragment read bases: X tile row frow + 16 i, 16-B chunk (4 ks + fq) ^ (row & 7); hidden row, chunk (4 s + fq) ^ (row & 15)
  const int xa0 = frow * 128 + ((fq ^ (frow & 7)) << 4), xa1 = frow * 128 + (((4 + fq) ^ (frow & 7)) << 4);
  const int hrow = frow * 512;

  // ---- prologue: weight ring, X tiles 0 (LDS stage 0) and 1 (registers) ------------------------------------------------
  frag_t R[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) R[f] = ldg<frag_t>(wp + f * 1024);
  wp += 8192;  // wp + 1024 f is now the fragment that refills slot f
  constexpr bool XD2 = KO == 8;  // X tiles requested two tiles ahead (two register sets) instead of one
  uint4 xq0 = {}, xq1 = {}, xr0 = {}, xr1 = {};  // set q: odd tiles (XD2) / every tile; set r: even tiles (XD2 only)
  if (xact) xq0 = ldg<uint4>(xp), xq1 = ldg<uint4>(xp + 16);
  f32x4 y[RT][NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const f32x4 bb = b2 ? *(const f32x4*)(b2 + j * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < RT; ++i) y[i][j] = bb;
  }
  if (xact) {
    *(uint4*)(xs + xw0) = xq0, *(uint4*)(xs + xw1) = xq1;
    xq0 = ldg<uint4>(xp + 128), xq1 = ldg<uint4>(xp + 128 + 16);
    if constexpr (XD2) xr0 = ldg<uint4>(xp + (nkt > 2 ? 256 : 0)), xr1 = ldg<uint4>(xp + (nkt > 2 ? 256 : 0) + 16);
  }
  lds_barrier();
  int st = 0;  // LDS stage of the current X tile
  int kn = XD2 ? (nkt > 3 ? 3 : 3 % nkt) : (nkt > 2 ? 2 : 0);  // k-tile index (mod nkt) of the next X tile to request

  for (int chunk = 0; chunk < nchunk; ++chunk) {
    // ---- phase 1: h[rows x 32 units of this wave] = X . W1 chunk^T -----------------------------------------------------
    f32x4 h[RT][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4 bb = b1 ? *(const f32x4*)(b1 + chunk * FC + j * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < RT; ++i) h[i][j] = bb;
    }
    for (int ko = 0; ko < nko; ++ko) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {  // K step 4 ko + u: X tile 2 ko + (u >> 1), half u & 1; ring slots 2u, 2u + 1
        const uint8_t* sa = xs + st * G::XS_B + ((u & 1) ? xa1 : xa0);
        frag_t a[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) a[i] = *(const frag_t*)(sa + i * 2048);
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          if constexpr (KO == 4) {
            asm volatile("" ::"v"(a[i]), "v"(R[2 * u]), "v"(R[2 * u + 1]));
          } else {
            h[i][0] = HT::mfma16(R[2 * u], a[i], h[i][0]);
            h[i][1] = HT::mfma16(R[2 * u + 1], a[i], h[i][1]);
          }
        }
        if constexpr (KO != 2) {
          R[2 * u] = ldg<frag_t>(wp + (2 * u) * 1024);
          R[2 * u + 1] = ldg<frag_t>(wp + (2 * u + 1) * 1024);
        }
        if ((u & 1) == 0) {
          // between the tile's two K steps: publish the next X tile (its stage was last read two tiles ago, and every wave
          // has passed the previous barrier since), request the one after it
          const int sn = st == 2 ? 0 : st + 1;
          if (KO != 3 && xact) {
            if (XD2 && u == 2) {  // odd tile: the next one (even) is in set r
              *(uint4*)(xs + sn * G::XS_B + xw0) = xr0, *(uint4*)(xs + sn * G::XS_B + xw1) = xr1;
              xr0 = ldg<uint4>(xp + kn * 128), xr1 = ldg<uint4>(xp + kn * 128 + 16);
            } else {
              *(uint4*)(xs + sn * G::XS_B + xw0) = xq0, *(uint4*)(xs + sn * G::XS_B + xw1) = xq1;
              xq0 = ldg<uint4>(xp + kn * 128), xq1 = ldg<uint4>(xp + kn * 128 + 16);
            }
          }
          kn = kn + 1 == nkt ? 0 : kn + 1;
          lds_barrier();
        } else {
          st = st == 2 ? 0 : st + 1;
        }
      }
      wp += 8192;
    }
    // ---- exact GELU (bias already in h) -> 16-bit hidden chunk image.  Every wave has passed a phase-1 barrier since it
    // finished phase 2 of the previous chunk, so the image it read there is free.
    {
      uint8_t* hw = hid + (chunk & 1) * G::HID_B + hrow + (fq & 1) * 8;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c16 = 16 * j + 2 * wn + (fq >> 1);  // image position of unit 32 w + 16 j + u is 128 j + 16 w + u (stream k order)
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          f32x2 g01 = {h[i][j][0], h[i][j][1]}, g23 = {h[i][j][2], h[i][j][3]};
          if constexpr (KO == 7) g01 = gelu_erf2(g01), g23 = gelu_erf2(g23);
          else if constexpr (KO != 1) g01 = gelu_sig2(g01), g23 = gelu_sig2(g23);
          *(uint2*)(hw + i * 8192 + ((c16 ^ frow) << 4)) = make_uint2(HT::pack(g01[0], g01[1]), HT::pack(g23[0], g23[1]));
        }
      }
    }
    lds_barrier();
    // ---- phase 2: y += hidden chunk . W2[:, chunk]^T, 8 K steps of 32, NJ fragments each; no barrier ---------------------
    const uint8_t* hb = hid + (chunk & 1) * G::HID_B + hrow;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      frag_t a[RT];
#pragma unroll
      for (int i = 0; i < RT; ++i) a[i] = *(const frag_t*)(hb + i * 8192 + (((4 * s + fq) ^ frow) << 4));
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int slot = (s * NJ + j) & 7;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          if constexpr (KO == 5) {
            asm volatile("" ::"v"(a[i]), "v"(R[slot]));
          } else {
            y[i][j] = HT::mfma16(R[slot], a[i], y[i][j]);
          }
        }
        if constexpr (KO != 2) R[slot] = ldg<frag_t>(wp + slot * 1024);
        if (slot == 7) wp += 8192;
      }
    }
  }

  store_tile<HT, RT, NJ, G::SMEM, KO>(g, y, smem, row0, row_end, tid, wn, frow, fq);
}

// =====================================================================================================================
// Second form (the default): the tile's X rows stay RESIDENT in LDS ([rows][Din] 16-bit, loaded once per tile instead of
// once per hidden chunk), so phase 1 stages nothing and has NO barrier: 16 K steps of { 7 fragment reads, 14 MFMAs, 2 weight
// refills } per wave, the waves free to drift apart.  What pays for it: the hidden chunk is published in two halves of 128
// units through one 28-KiB image (112 KiB of X + 56 KiB of hidden chunk would be 8 KiB over the 160 KiB of a CU), i.e.
// four barriers per chunk: [image free] write half 0 [published] phase 2a [read] write half 1 [published] phase 2b.  The
// half that waits is held as packed 16-bit values (14 registers).
template <int RT, int DIN>
struct XGeo {
  static constexpr int ROWS = RT * 16;
  static constexpr int XROW_B = DIN * 2;     // X image: 16-B chunk c of row m at slot c ^ (m & 15)
  static constexpr int XIMG_B = ROWS * XROW_B;
  static constexpr int HID_B = ROWS * 256;   // half hidden chunk, 16-bit [ROWS][128]: chunk c of row m at slot c ^ (m & 15)
  static constexpr int SMEM = XIMG_B + HID_B;
};

// Diagnostic build KO == 9 (knob 49, tools/mlp_stamps.py): wave 0 of every workgroup sums s_memtime differences per phase and
// adds them to the eight 64-bit counters that the R2 pointer of the descriptor points at (R2 is not read as a residual then).
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

template <typename HT, int RT, int NJ, int DIN, int KO>
__global__ __launch_bounds__(NT, 2) void fused_mlp_xres_kernel(const MdmMlpDesc g, const int tile_h) {
  typedef typename HT::frag_t frag_t;
  typedef XGeo<RT, DIN> G;
  constexpr int DOUT = NJ * 128, NKO = DIN / 128, NLINE = DIN / 64;
  constexpr int NA = KO == 8 ? 4 : RT, PD = NA - 1;  // A-fragment ring (knob 48: 4 registers, 3 ahead; default RT, RT - 1 ahead)
  static_assert((4 * RT) % NA == 0, "the ring must close over the unrolled body");
  extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
  uint8_t* const ximg = smem;
  uint8_t* const hid = smem + G::XIMG_B;

  unsigned long long acc[8] = {}, last = 0;
  if constexpr (KO == 9) last = stamp_now();
  // Persistent over tiles: workgroup b takes tiles remap(b), remap(b) + grid, ...; round k hands an XCD a contiguous range of
  // tiles (~1 group: its weights stay in that XCD's L2).  The stores of a tile drain under the next tile's work.
  for (int mt = xcd_remap(blockIdx.x, gridDim.x);; mt += gridDim.x) {
  // the thread id is made opaque per tile: otherwise every lane-constant address of the body is hoisted out of this loop and
  // lives in (spilled) registers across it -- 108 spilled registers; recomputing them per tile is a few dozen instructions
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fq = lane >> 4;
  int row0, row_end, grp;
  if (!find_tile(g, tile_h, lane, mt, row0, row_end, grp)) break;
  XSTAMP(0);

  const int nchunk = g.F / FC;
  constexpr int fpc = (DIN / 32) * 2 + 8 * NJ;
  const uint8_t* wp = (const uint8_t*)g.wstream + ((int64_t)grp * g.wstream_gs + (int64_t)wn * nchunk * fpc * 512) * 2 + lane * 16;

  frag_t R[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) R[f] = ldg<frag_t>(wp + f * 1024);
  wp += 8192;

  // ---- X tile -> LDS: wave w < RT brings rows 16 w .. 16 w + 15; one instruction = 8 rows x one 128-B line ----------------
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
#pragma unroll
    for (int c0 = 0; c0 < NLINE; c0 += 4) {
      uint4 v[2][4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) v[hf][c] = ldg<uint4>(xp[hf] + (c0 + c) * 128);
#pragma unroll
      for (int c = 0; c < 4; ++c)
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
    const float* b2 = g.b2 ? g.b2 + (int64_t)grp * g.b2_gs + wn * (16 * NJ) + fq * 4 : nullptr;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const f32x4 bb = b2 ? *(const f32x4*)(b2 + j * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < RT; ++i) y[i][j] = bb;
    }
  }
  lds_barrier();
  XSTAMP(1);

#pragma unroll 1
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    // Lane-constant addresses are recomputed per chunk from an opaque copy of the lane id, so that nothing but the
    // accumulators, the weight ring and the stream pointer lives across the phases (the register file is the limit here).
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int fr = ln & 15, fc = (ln >> 4) ^ fr;  // fc: the XOR-swizzled chunk of K step 0; step u reads chunk (4 u) ^ fc
    // ---- phase 1 (no barrier): h[rows x 32 units of this wave] = X . W1 chunk^T ---------------------------------------------
    f32x4 h[RT][2];
    {
      const float* b1 = g.b1 ? g.b1 + (int64_t)grp * g.b1_gs + wn * 32 + (ln >> 4) * 4 + chunk * FC : nullptr;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x4 bb = b1 ? *(const f32x4*)(b1 + j * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < RT; ++i) h[i][j] = bb;
      }
    }
    {
      // The 7 A fragments of a K step go through a ring of NA registers that runs PD = NA - 1 fragments ahead of the MFMAs,
      // every { 2 MFMAs, 1 fragment read } pinned by a scheduling barrier: left alone, hipcc issues one read, waits for it and
      // issues its two MFMAs (one exposed LDS round trip per 32 MFMA cycles) as soon as registers get tight.
      const int xb = fr * G::XROW_B + (fc << 4);  // byte offset of (row fr, K step 0) in the X image; bits 6..7 flip per step
      constexpr int NF = 4 * RT;                   // fragments per unrolled body (4 K steps)
      frag_t A[NA];
#pragma unroll
      for (int k = 0; k < PD; ++k) A[k % NA] = *(const frag_t*)(ximg + (xb ^ (64 * (k / RT))) + (k % RT) * 16 * G::XROW_B);
#pragma unroll 1
      for (int ko = 0; ko < NKO; ++ko) {
        const int kn = ko + 1 == NKO ? 0 : ko + 1;  // the body's last prefetches belong to the next body (the last ones are not used)
#pragma unroll
        for (int k = 0; k < NF; ++k) {
          const int u = k / RT, i = k % RT;
          if constexpr (KO == 4) {
            asm volatile("" ::"v"(A[k % NA]), "v"(R[2 * u]), "v"(R[2 * u + 1]));
          } else {
            h[i][0] = HT::mfma16(R[2 * u], A[k % NA], h[i][0]);
            h[i][1] = HT::mfma16(R[2 * u + 1], A[k % NA], h[i][1]);
          }
          const int kp = k + PD, up = (kp % NF) / RT, ip = kp % RT;
          A[kp % NA] = *(const frag_t*)(ximg + (xb ^ (64 * up)) + (kp < NF ? ko : kn) * 256 + ip * 16 * G::XROW_B);
          if (i == RT - 1) {
            if constexpr (KO != 2) {
              R[2 * u] = ldg<frag_t>(wp + (2 * u) * 1024);
              R[2 * u + 1] = ldg<frag_t>(wp + (2 * u + 1) * 1024);
            }
          }
          pin_vmem();
        }
        wp += 8192;
      }
    }
    XSTAMP(2);
    // ---- GELU (bias already in h) -> packed 16-bit, both halves -------------------------------------------------------------
    uint2 pk[2][RT];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        f32x2 g01 = {h[i][j][0], h[i][j][1]}, g23 = {h[i][j][2], h[i][j][3]};
        if constexpr (KO == 7) g01 = gelu_erf2(g01), g23 = gelu_erf2(g23);
        else if constexpr (KO != 1) g01 = gelu_sig2(g01), g23 = gelu_sig2(g23);
        pk[j][i] = make_uint2(HT::pack(g01[0], g01[1]), HT::pack(g23[0], g23[1]));
      }
    XSTAMP(3);
    // ---- phase 2 in two halves through the one half-chunk image ------------------------------------------------------------
    // image position of this lane's four units of half j: 16 wn + 4 fq (+ r): chunk 2 wn + (fq >> 1), byte 8 (fq & 1)
    uint8_t* const hw = hid + fr * 256 + (((2 * wn + (ln >> 5)) ^ fr) << 4) + ((ln >> 4) & 1) * 8;
    const int hb = fr * 256 + (fc << 4);  // (row fr, K step 0) of the half image; step s reads chunk (4 s) ^ fc
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      lds_barrier();  // the image is free: every wave is past its reads of the previous half
#pragma unroll
      for (int i = 0; i < RT; ++i) *(uint2*)(hw + i * 4096) = pk[hf][i];
      lds_barrier();  // published
      XSTAMP(4);
      {
        constexpr int NF = 4 * RT;  // fragments of this half: 4 K steps x RT row tiles
        frag_t A[NA];
#pragma unroll
        for (int k = 0; k < PD; ++k) A[k % NA] = *(const frag_t*)(hid + (hb ^ (64 * (k / RT))) + (k % RT) * 4096);
#pragma unroll
        for (int k = 0; k < NF; ++k) {
          const int sq = k / RT, i = k % RT;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int slot = ((4 * hf + sq) * NJ + j) & 7;
            if constexpr (KO == 5) {
              asm volatile("" ::"v"(A[k % NA]), "v"(R[slot]));
            } else {
              y[i][j] = HT::mfma16(R[slot], A[k % NA], y[i][j]);
            }
          }
          if (k + PD < NF) A[(k + PD) % NA] = *(const frag_t*)(hid + (hb ^ (64 * ((k + PD) / RT))) + ((k + PD) % RT) * 4096);
          if (i == RT - 1) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
              const int slot = ((4 * hf + sq) * NJ + j) & 7;
              if constexpr (KO != 2) R[slot] = ldg<frag_t>(wp + slot * 1024);
              if (slot == 7) wp += 8192;
            }
          }
          pin_vmem();
        }
      }
      XSTAMP(5);
    }
  }

  // the epilogue derives its addresses from an opaque thread id of its own (see above)
  int te = threadIdx.x;
  asm volatile("" : "+v"(te));
  if constexpr (KO == 9) {
    MdmMlpDesc g2 = g;
    g2.R2 = nullptr;
    store_tile<HT, RT, NJ, G::SMEM, KO>(g2, y, smem, row0, row_end, te, wn, te & 15, (te & 63) >> 4);
    XSTAMP(6);
    if (tid == 0) atomicAdd((unsigned long long*)g.R2 + 7, 1ull);
  } else {
    store_tile<HT, RT, NJ, G::SMEM, KO>(g, y, smem, row0, row_end, te, wn, te & 15, (te & 63) >> 4);
  }
  lds_barrier();  // the staging reads of this tile are done before the next tile's X rows land in the same LDS
  }
  if constexpr (KO == 9) {
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < 7; ++q) atomicAdd((unsigned long long*)g.R2 + q, acc[q]);
    }
  }
}

// ---- weight stream packing: fp32 / 16-bit row-major expert weights -> the per-(group, wave) fragment stream ------------
// stream[g][wave w][chunk c][fragment f][lane l][8]:
//   f <  2 Din/32 (phase 1):  step = f >> 1, j = f & 1:   W1[g][256 c + 32 w + 16 j + (l & 15)][32 step + 8 (l >> 4) + e]
//   f >= 2 Din/32 (phase 2):  f' = f - 2 Din/32, s = f' / NJ, j = f' % NJ, hidden-image position p = 32 s + 8 (l >> 4) + e:
//                                                         W2[g][16 NJ w + 16 j + (l & 15)][256 c + unit(p)]
//   unit(p) = 32 ((p >> 4) & 7) + 16 (p >> 7) + (p & 15): the image holds, per half, the 16 units wave 0 produced for that
//   half, then wave 1's, ... (the hidden chunk can then be published and consumed one half at a time)
template <typename HT>
__global__ __launch_bounds__(256) void mlp_stream_pack_kernel(const float* w1, const float* w2, int G, int F, int Din, int Dout,
                                                              uint16_t* out) {
  const int NJ = Dout / 128, nchunk = F / FC, f1 = 2 * (Din / 32), fpc = f1 + 8 * NJ;
  const int64_t nfrag = (int64_t)G * 8 * nchunk * fpc;
  for (int64_t fi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); fi < nfrag; fi += (int64_t)gridDim.x * 4) {
    const int l = threadIdx.x & 63;
    const int f = (int)(fi % fpc);
    const int c = (int)((fi / fpc) % nchunk);
    const int w = (int)((fi / fpc / nchunk) % 8);
    const int gi = (int)(fi / fpc / nchunk / 8);
    const float* src;
    if (f < f1) {
      const int step = f >> 1, j = f & 1;
      src = w1 + ((int64_t)gi * F + 256 * c + 32 * w + 16 * j + (l & 15)) * Din + 32 * step + 8 * (l >> 4);
    } else {
      const int fp = f - f1, s = fp / NJ, j = fp % NJ;
      const int kp = 32 * s + 8 * (l >> 4);  // position in the hidden image: half kp >> 7, wave (kp >> 4) & 7, unit kp & 15 (+ e)
      src = w2 + ((int64_t)gi * Dout + 16 * NJ * w + 16 * j + (l & 15)) * F + 256 * c + 32 * ((kp >> 4) & 7) + 16 * (kp >> 7) + (kp & 15);
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
  // + 8 fragments: the ring's last refills of the last wave of the last group read (and discard) past the stream's end
  return (int64_t)G * ((int64_t)F * Din + (int64_t)Dout * F) + 8 * 512;
}

int mlp_stream_pack(const float* w1, const float* w2, int G, int F, int Din, int Dout, int h16, uint16_t* out, hipStream_t stream) {
  if (!w1 || !w2 || !out || G < 1 || (F % FC) || (Din % 128) || (Dout != 512 && Dout != 1024)) return MDM_ERR_UNSUPPORTED;
  const int64_t body = (int64_t)G * ((int64_t)F * Din + (int64_t)Dout * F);
  if (hipMemsetAsync(out + body, 0, 8 * 512 * sizeof(uint16_t), stream) != hipSuccess) return MDM_ERR_LAUNCH;
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
  if (!a.wstream || a.Dout != 512 || a.Din < 128 || (a.Din % 128) || a.F < FC || (a.F % FC) || a.M < 1) return false;
  if (a.goff && (a.ngroups < 1 || a.ngroups > 64)) return false;
  if ((a.ldx % 8) || ((((uintptr_t)a.X) | ((uintptr_t)a.wstream)) & 15) || (a.wstream_gs % 8)) return false;
  if ((a.ldc & 3) || (a.R1 && (a.ldr1 & 3)) || (a.R2 && (a.ldr2 & 3))) return false;
  if ((a.b1 && ((((uintptr_t)a.b1) & 15) || (a.b1_gs & 3))) || (a.b2 && ((((uintptr_t)a.b2) & 15) || (a.b2_gs & 3)))) return false;
  return true;
}

// tile height: the tiles of all groups should fill the CUs in whole rounds
int mlp_stream_tile_h(int64_t M, int ngroups, int rt_max) {
  const int cus = device_cus();
  const int64_t cap = (int64_t)cus * rt_max * 16;
  const int64_t rounds = (M + cap - 1) / cap;
  int64_t h = (M + cus * rounds - 1) / (cus * rounds);
  h = (h + 15) & ~15ll;
  if (h < 16) h = 16;
  if (h > rt_max * 16) h = rt_max * 16;
  (void)ngroups;
  return (int)h;
}

extern int g_bf16_variant;

template <int KO>
static int launch_stream(const MdmMlpDesc& a, hipStream_t stream) {
  constexpr int RT = 7, NJ = 4;
  const int th = mlp_stream_tile_h(a.M, a.goff ? a.ngroups : 1, RT);
  const int tiles = (int)(a.M / th) + (a.goff ? a.ngroups : 1);
  if (a.Din == 512 && g_bf16_variant != 40) {  // X rows resident in LDS (knob 40: the staged-X form, for A/B runs)
    constexpr int smem = XGeo<RT, 512>::SMEM;
    static DevOnce attr;
    if (!attr) {
      if (hipFuncSetAttribute((const void*)fused_mlp_xres_kernel<HB, RT, NJ, 512, KO>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess ||
          hipFuncSetAttribute((const void*)fused_mlp_xres_kernel<HF, RT, NJ, 512, KO>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
        return MDM_ERR_LAUNCH;
      attr = true;
    }
    const int grid = tiles < device_cus() ? tiles : device_cus();  // persistent: one workgroup per CU walks the tiles
    if (a.h16 == MDM_H16_F16) {
      hipLaunchKernelGGL((fused_mlp_xres_kernel<HF, RT, NJ, 512, KO>), dim3(grid), dim3(NT), smem, stream, a, th);
    } else {
      hipLaunchKernelGGL((fused_mlp_xres_kernel<HB, RT, NJ, 512, KO>), dim3(grid), dim3(NT), smem, stream, a, th);
    }
    MDM_RETURN_IF_LAUNCH_FAILED();
    return MDM_OK;
  }
  constexpr int smem = SGeo<RT>::SMEM;
  static DevOnce attr;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)fused_mlp_stream_kernel<HB, RT, NJ, KO>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess ||
        hipFuncSetAttribute((const void*)fused_mlp_stream_kernel<HF, RT, NJ, KO>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return MDM_ERR_LAUNCH;
    attr = true;
  }
  if (a.h16 == MDM_H16_F16) {
    hipLaunchKernelGGL((fused_mlp_stream_kernel<HF, RT, NJ, KO>), dim3(tiles), dim3(NT), smem, stream, a, th);
  } else {
    hipLaunchKernelGGL((fused_mlp_stream_kernel<HB, RT, NJ, KO>), dim3(tiles), dim3(NT), smem, stream, a, th);
  }
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

int fused_mlp_stream(const MdmMlpDesc& a, hipStream_t stream) {
  if (!a.X || (!a.C && !a.C16)) return MDM_ERR_ARG;
  if (!fused_mlp_stream_supported(a)) return MDM_ERR_UNSUPPORTED;
  switch (g_bf16_variant) {  // knobs 41..46: knock-out builds; 47: erf-form GELU (bit-identical to csrc/mlp.hip) for tools/mlp_bench.py (wrong results, timing only)
    case 41: return launch_stream<1>(a, stream);
    case 42: return launch_stream<2>(a, stream);
    case 43: return launch_stream<3>(a, stream);
    case 44: return launch_stream<4>(a, stream);
    case 45: return launch_stream<5>(a, stream);
    case 46: return launch_stream<6>(a, stream);
    case 47: return launch_stream<7>(a, stream);
    case 48: return launch_stream<8>(a, stream);
    case 49: return launch_stream<9>(a, stream);
    default: return launch_stream<0>(a, stream);
  }
}

}  // namespace mdm
