// Fused Performer-style linear attention core (fast_attention.py:29-92) for head_dim = 128, throughput mode.
// One workgroup (8 waves) per (batch, head); everything between the QKV projection and the output projection
// happens on chip:
//   K:  k rows -> LN(dh) -> L2 norm -> bf16 A-fragments in registers -> MFMA with P^T (LDS) -> 0.1*exp(clamp) -> mask
//       -> kphi^T [m][t] in LDS (the accumulator's 4 consecutive t per lane are one 8-byte store)
//   V:  v rows -> LN(dh) -> v [t][d] in LDS, row-major (four 16-byte stores per row and lane; 16-B chunk c of row t at slot
//       c ^ f(t): the stores and the transposing reads below are conflict-free)
//   KV: KV^T [d][m] = 0.1 * sum_t v[t][d] kphi^T[m][t]   (kphi^T fragments: plain ds_read_b128; the v operand, K-contiguous in t,
//       comes out of the row-major image through the transposing read ds_read_b64_tr_b16)
//   Q:  q rows -> LN -> L2 -> MFMA with P^T with the operands swapped, so the feature accumulator (4 consecutive m per
//       lane, one t per lane) IS the B operand of the next MFMA after exp + bf16 packing (the k-slot order is matched on
//       the KV^T side by two 8-byte reads); same-t denominator: the lane's four kphi[t][m .. m + 3] come out of kphi^T [m][t] through
//       the transposing read as well (8 reads per tile instead of 32 two-byte reads); num / den; LN(dh); 16-bit rows out.
// LDS: kphi^T 128 x (TP+8) 16-bit (row stride/16 B odd => conflict-free fragment reads), v TP x 128, P^T 128 x 136 resident;
// KV^T (128 x 136) takes the v region once v is dead.  151,552 B at T = 196.
#include "kernels.h"
#include "proj_phase.h"

namespace mdm {
namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
// v image: 256-byte rows [t][128 d]; 16-byte chunk ch of row `row` sits at chunk ch ^ f(row) (as in perf_attn2.hip)
__device__ __forceinline__ int voff(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

constexpr int DH = 128, MF = 128, PS = 136, NW = 8, NTH = 64 * NW;  // 8 waves: <= 2 row tiles per wave per phase  // PS: padded row stride (elements) of the 128-wide LDS images


template <typename HT>
__device__ __forceinline__ typename HT::frag_t make_frag(const float* x) {
  typedef typename HT::frag_t frag_t;
  u32x4 u = {HT::pack(x[0], x[1]), HT::pack(x[2], x[3]), HT::pack(x[4], x[5]), HT::pack(x[6], x[7])};
  return __builtin_bit_cast(frag_t, u);
}

__device__ __forceinline__ float quad_sum(float v) {  // across the 4 lanes that share (lane & 15)
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// QKV = true: the launch also computes its (batch, head)'s q | k | v rows, 0.1 * (xn W^T + b) (fast_attention.py:145-157), instead of
// reading them: phase 0 below.  xn = the pre-normed 16-bit rows [B S, D]; wqkv = the 16-bit plane of query|key|value stacked [3 D][ldw];
// the q rows pass through `qkv` (the q third of the [B S, 3 D] buffer, written and re-read by the same lanes), k and v never leave the CU.
struct PerfQkv {
  const uint16_t* xn;
  const uint16_t* w;
  int ldw;
  const float* bias;  // [3 D]
  float alpha;
};

template <typename HT, int QMT>  // QMT: 0 = q | k | v rows read from memory; 7 / 13 = computed here for that many row tiles
__global__ __launch_bounds__(NTH) void perf_attn_kernel(uint16_t* __restrict__ qkv, const uint16_t* __restrict__ PT,
                                                           int ldp, const float* __restrict__ hn_w,
                                                           const float* __restrict__ hn_b, const int* __restrict__ len,
                                                           int S, int H, uint16_t* __restrict__ out, const PerfQkv qa) {
  typedef typename HT::frag_t frag_t;
  constexpr bool QKV = QMT > 0;
  extern __shared__ __attribute__((aligned(1024))) uint8_t smem_raw[];
  uint16_t* smem = (uint16_t*)smem_raw;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r16 = lane & 15, q = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int D = H * DH;
  const int ntile = (S + 15) >> 4, SP = ntile * 16, TP = (S + 31) & ~31, TS = TP + 8;
  const int vreg = (TP * DH > DH * PS) ? TP * DH : DH * PS;  // the v region takes the KV^T state once v is dead
  uint16_t* kT = smem;
  uint16_t* vT = smem + MF * TS;
  uint8_t* const vimg = (uint8_t*)vT;  // v [TP][128], row-major, swizzled chunks (voff)
  uint16_t* PTl = vT + vreg;  // P^T stays resident: the q features need it again (a second load from L2 sat between two
  uint16_t* KV = vT;          // barriers in the middle of the kernel)
  const int nvalid = min(len[b], S);
  float gw[32], gb[32];

  // 128 x 128 16-bit = 2048 16-B chunks, 4 per thread: all loads in flight, then the LDS writes (named registers: an array
  // captured by a lambda keeps its stack slot, and a kernel with a private segment pays for it at every wave launch)
  static_assert(2048 / NTH == 4, "four chunks per thread");
  auto load_PT = [&]() __attribute__((always_inline)) {
    const uint16_t* src = PT + (int64_t)(tid >> 4) * ldp + (tid & 15) * 8;
    const uint4 t0 = *(const uint4*)(src), t1 = *(const uint4*)(src + (int64_t)(NTH >> 4) * ldp);
    const uint4 t2 = *(const uint4*)(src + (int64_t)(2 * NTH >> 4) * ldp), t3 = *(const uint4*)(src + (int64_t)(3 * NTH >> 4) * ldp);
    uint16_t* dst = PTl + (tid >> 4) * PS + (tid & 15) * 8;
    *(uint4*)(dst) = t0;
    *(uint4*)(dst + (NTH >> 4) * PS) = t1;
    *(uint4*)(dst + (2 * NTH >> 4) * PS) = t2;
    *(uint4*)(dst + (3 * NTH >> 4) * PS) = t3;
  };
  // A wave owns tiles wid, wid+8 (S <= 224 -> 14 tiles).  All of its k and v rows are requested up front and
  // its q rows as soon as the k registers are free: one global round trip per phase instead of one per tile.
  constexpr int MAXT = 2;
  // raw row t0 + r16, elements k = 32*ks + 8*q + j, 16-bit: uint4 u[4]
  auto raw_load = [&](int which, int tile, uint4 (&u)[4]) __attribute__((always_inline)) {
    const int t = tile * 16 + r16;
    const int tc = t < S ? t : S - 1;
    const uint16_t* p = qkv + ((int64_t)(b * S + tc)) * 3 * D + which * D + h * DH + 8 * q;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) u[ks] = *(const uint4*)(p + 32 * ks);
  };
  // LN over head_dim (+ L2 normalise) -> x[32]
  auto normalize = [&](const uint4 (&u)[4], bool l2, float (&x)[32]) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      x[8 * ks + 0] = HT::lo(u[ks].x), x[8 * ks + 1] = HT::hi(u[ks].x);
      x[8 * ks + 2] = HT::lo(u[ks].y), x[8 * ks + 3] = HT::hi(u[ks].y);
      x[8 * ks + 4] = HT::lo(u[ks].z), x[8 * ks + 5] = HT::hi(u[ks].z);
      x[8 * ks + 6] = HT::lo(u[ks].w), x[8 * ks + 7] = HT::hi(u[ks].w);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += x[i];
    const float mean = quad_sum(s) * (1.f / DH);
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      x[i] -= mean;
      v += x[i] * x[i];
    }
    const float rstd = rsqrtf(quad_sum(v) * (1.f / DH) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = x[i] * rstd * gw[i] + gb[i];
    if (l2) {
      float n = 0.f;
#pragma unroll
      for (int i = 0; i < 32; ++i) n += x[i] * x[i];
      const float inv = 1.f / fmaxf(sqrtf(quad_sum(n)), 1e-12f);
#pragma unroll
      for (int i = 0; i < 32; ++i) x[i] *= inv;
    }
  };
  uint4 kq[MAXT][4], vr[MAXT][4];
  if constexpr (QKV) {
    // ---- phase 0: [S x 384] = xn[b] [S x 512] . (Wq_h | Wk_h | Wv_h)^T -------------------------------------------------------------
    // The rows are what the waves share: K slices of 64 columns go global -> registers -> LDS one slice ahead (two LDS stages; a
    // second register set would not fit beside the 39 accumulators).  The weights are private to a wave (waves split the 24 column tiles, 3 each = 48 of the 384 columns, and
    // multiply all row tiles: 39 accumulators at 13 row tiles): its 48 fragments stream global -> registers through a ring that
    // runs two K steps ahead and never touch LDS.
    constexpr int MT = QKV ? QMT : 1, NCH = QkvGeo<MT>::NCH, R = QkvGeo<MT>::R;
    const uint16_t* xrow0 = qa.xn + (int64_t)b * S * D;
    const uint16_t* wrow[3];  // this lane's row of the wave's three weight fragments, at k = 8 q
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int ct = 3 * wid + j, g = ct >> 3;
      wrow[j] = qa.w + (int64_t)(g * D + h * DH + ((16 * ct) & 127) + r16) * qa.ldw + 8 * q;
    }
    frag_t wr[6];
#pragma unroll
    for (int f = 0; f < 6; ++f) wr[f] = *(const frag_t*)(wrow[f % 3] + 32 * (f / 3));
    f32x4 acc[MT][3];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 sa[NCH];
    qkv_fetch<NCH>(sa, tid, R, S, D, 0, xrow0);
#pragma unroll 1
    for (int kk = 0; kk < 8; ++kk) qkv_slice<HT, MT, 3>(kk, sa, acc, wr, wrow, smem_raw + (kk & 1) * (R * 128), tid, r16, q, S, D, xrow0);
    __syncthreads();  // the stages are dead: their LDS takes the q | k | v rows
    // rows as 16-bit [R][384] (768-B rows; 16-B chunk c at slot c ^ (row & 15), inside its group of 16 chunks)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int col = 16 * (3 * wid + j) + 4 * q, g = col >> 7;  // column in q | k | v, its group
      const f32x4 bb = *(const f32x4*)(qa.bias + g * D + h * DH + (col & 127));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = 16 * mt + r16;
        const f32x4 v = acc[mt][j];
        *(uint2*)(smem_raw + row * 768 + ((((col >> 3)) ^ (row & 15)) << 4) + ((col >> 2) & 1) * 8) =
            make_uint2(HT::pack(qa.alpha * (v[0] + bb[0]), qa.alpha * (v[1] + bb[1])),
                       HT::pack(qa.alpha * (v[2] + bb[2]), qa.alpha * (v[3] + bb[3])));
      }
    }
    __syncthreads();
    // every wave takes the k and v rows of its own tiles (the register layout of raw_load) and hands its q rows to memory
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      const int tile = wid + NW * i;
      if (tile < ntile) {
        const int row = tile * 16 + r16;
        const uint8_t* ir = smem_raw + row * 768;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          kq[i][ks] = *(const uint4*)(ir + (((16 + 4 * ks + q) ^ (row & 15)) << 4));
          vr[i][ks] = *(const uint4*)(ir + (((32 + 4 * ks + q) ^ (row & 15)) << 4));
          const uint4 qv = *(const uint4*)(ir + (((4 * ks + q) ^ (row & 15)) << 4));
          if (row < S) *(uint4*)(qkv + ((int64_t)(b * S + row)) * 3 * D + h * DH + 32 * ks + 8 * q) = qv;
        }
      }
    }
    __syncthreads();  // the image is dead: P^T and the feature images land in the same LDS
  } else {
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      const int tile = wid + NW * i;
      if (tile < ntile) {
        raw_load(1, tile, kq[i]);
        raw_load(2, tile, vr[i]);
      }
    }
  }

  // LayerNorm gain/bias at this lane's input positions k = 32*ks + 8*q + j (loaded behind phase 0, which needs the registers)
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const f32x4 a = *(const f32x4*)(hn_w + 32 * ks + 8 * q), c = *(const f32x4*)(hn_w + 32 * ks + 8 * q + 4);
    const f32x4 d = *(const f32x4*)(hn_b + 32 * ks + 8 * q), e = *(const f32x4*)(hn_b + 32 * ks + 8 * q + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gw[8 * ks + j] = a[j], gw[8 * ks + 4 + j] = c[j];
      gb[8 * ks + j] = d[j], gb[8 * ks + 4 + j] = e[j];
    }
  }


  load_PT();
  __syncthreads();

  // ---- K: kphi^T[m][t] ---------------------------------------------------------------------------
#pragma unroll
  for (int it = 0; it < MAXT; ++it) {
    const int tile = wid + NW * it;
    if (tile >= ntile) continue;  // (not break: the loop must unroll completely, or kq[] / vr[] live in scratch memory)
    const int t0 = tile * 16;
    float x[32];
    normalize(kq[it], true, x);
    frag_t a[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a[ks] = make_frag<HT>(x + 8 * ks);
    f32x4 acc[8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        const frag_t p = *(const frag_t*)(PTl + (16 * mt + r16) * PS + 32 * ks + 8 * q);
        acc[mt] = HT::mfma16(a[ks], p, acc[mt]);  // D[t][m]
      }
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = t0 + 4 * q + r;
        v[r] = t < nvalid ? 0.1f * exp_fast(fminf(fmaxf(acc[mt][r], -15.f), 15.f)) : 0.f;  // key mask (:69-74)
      }
      *(uint2*)(kT + (16 * mt + r16) * TS + t0 + 4 * q) = make_uint2(HT::pack(v[0], v[1]), HT::pack(v[2], v[3]));
    }
  }
#pragma unroll
  for (int i = 0; i < MAXT; ++i)  // k registers are free: request the q rows now, they land during the V / KV phases
    if (wid + NW * i < ntile) raw_load(0, wid + NW * i, kq[i]);
  __syncthreads();  // kphi^T complete

  // ---- V: v^T[d][t] ------------------------------------------------------------------------------
#pragma unroll
  for (int it = 0; it < MAXT; ++it) {
    const int tile = wid + NW * it;
    if (tile >= ntile) continue;  // (not break: the loop must unroll completely, or kq[] / vr[] live in scratch memory)
    const int t0 = tile * 16, t = t0 + r16;
    float x[32];
    normalize(vr[it], false, x);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {  // d = 32 ks + 8 q + j: 16-byte chunk 4 ks + q of row t
      u32x4 u = {HT::pack(x[8 * ks], x[8 * ks + 1]), HT::pack(x[8 * ks + 2], x[8 * ks + 3]), HT::pack(x[8 * ks + 4], x[8 * ks + 5]),
                 HT::pack(x[8 * ks + 6], x[8 * ks + 7])};
      if (t >= S) u = (u32x4){0u, 0u, 0u, 0u};
      *(u32x4*)(vimg + voff(t, 4 * ks + q)) = u;
    }
  }
  if (TP > SP) {  // zero the K padding: 16 columns of kphi^T, 16 rows of v
    for (int i = tid; i < DH * 16; i += NTH) kT[(i >> 4) * TS + SP + (i & 15)] = 0;
    for (int i = tid; i < 16 * 16; i += NTH) *(u32x4*)(vimg + 256 * (SP + (i >> 4)) + 16 * (i & 15)) = (u32x4){0u, 0u, 0u, 0u};
  }
  __syncthreads();

  // ---- KV^T[d][m] = 0.1 * sum_t v^T[d][t] kphi^T[m][t]  (:77) ----------------------------------------
  {
    f32x4 acc[8];  // wave w owns the m tile w (8 waves x 16 = 128 features), all 8 d tiles
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int qp = r16 >> 2, pp = r16 & 3;  // transposing read: this lane addresses row qp, columns 4 pp .. 4 pp + 3 of its group's block
    for (int ks = 0; ks < TP / 32; ++ks) {
      frag_t bf[8];
      const frag_t a = *(const frag_t*)(kT + (16 * wid + r16) * TS + 32 * ks + 8 * q);
      const int tr0 = 32 * ks + 8 * q + qp;
#pragma unroll
      for (int j = 0; j < 8; ++j) {  // column d = 16 j + r16, k = t = 32 ks + 8 q + 0..7
        const int ch = 2 * j + (pp >> 1);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(vimg + voff(tr0, ch) + 8 * (pp & 1)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(vimg + voff(tr0 + 4, ch) + 8 * (pp & 1)));
        bf[j] = __builtin_bit_cast(frag_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = HT::mfma16(a, bf[j], acc[j]);
    }
    __syncthreads();  // every wave is done reading v^T: its region takes the state
    // D[m][d]: col d = 16j + r16, rows m = 16w + 4q + r  ->  KV^T[d][m..m+3]
#pragma unroll
    for (int j = 0; j < 8; ++j)
      *(uint2*)(KV + (16 * j + r16) * PS + 16 * wid + 4 * q) =
          make_uint2(HT::pack(0.1f * acc[j][0], 0.1f * acc[j][1]), HT::pack(0.1f * acc[j][2], 0.1f * acc[j][3]));
  }
  __syncthreads();

  // ---- Q: features -> denominator -> num = qphi KV -> LN -> out -------------------------------------
#pragma unroll
  for (int it = 0; it < MAXT; ++it) {
    const int tile = wid + NW * it;
    if (tile >= ntile) continue;  // (not break: the loop must unroll completely, or kq[] / vr[] live in scratch memory)
    const int t0 = tile * 16, t = t0 + r16;
    float x[32];
    normalize(kq[it], true, x);
    frag_t qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = make_frag<HT>(x + 8 * ks);
    f32x4 accf[8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) accf[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        const frag_t p = *(const frag_t*)(PTl + (16 * mt + r16) * PS + 32 * ks + 8 * q);
        accf[mt] = HT::mfma16(p, qf[ks], accf[mt]);  // D[m][t]
      }
    // lane: t = t0 + r16, m = 16*mt + 4q + r
    float den = 0.f;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      // kphi[t][16 mt + 4 q + 0..3] out of kphi^T [m][t]: the 16 lanes of a group read a 4 x 16 block, each lane gets a column
      const s16x4 kk = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(kT + (16 * mt + 4 * q + (r16 >> 2)) * TS + t0 + 4 * (r16 & 3)));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float f = 0.1f * exp_fast(fminf(fmaxf(accf[mt][r], -15.f), 15.f));
        accf[mt][r] = f;
        den += f * HT::one((uint16_t)kk[r]);  // same-t dot (:81)
      }
    }
    den = fmaxf(quad_sum(den), 1e-6f);
    f32x4 accn[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) accn[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      // k-slots j=0..3 <-> m = 32s + 4q + j ; j=4..7 <-> m = 32s + 16 + 4q + (j-4)
      const u32x4 ub = {HT::pack(accf[2 * s][0], accf[2 * s][1]), HT::pack(accf[2 * s][2], accf[2 * s][3]),
                        HT::pack(accf[2 * s + 1][0], accf[2 * s + 1][1]), HT::pack(accf[2 * s + 1][2], accf[2 * s + 1][3])};
      const frag_t bq = __builtin_bit_cast(frag_t, ub);
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        const uint2 lo = *(const uint2*)(KV + (16 * dt + r16) * PS + 32 * s + 4 * q);
        const uint2 hi = *(const uint2*)(KV + (16 * dt + r16) * PS + 32 * s + 16 + 4 * q);
        const u32x4 ua = {lo.x, lo.y, hi.x, hi.y};
        accn[dt] = HT::mfma16(__builtin_bit_cast(frag_t, ua), bq, accn[dt]);  // D[d][t]
      }
    }
    // lane: t, d = 16*dt + 4q + r.  out = LN_dh(0.1 * num / den)   (:78,85-90)
    const float sc = 0.1f / den;
    float s1 = 0.f;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        accn[dt][r] *= sc;
        s1 += accn[dt][r];
      }
    const float mean = quad_sum(s1) * (1.f / DH);
    float s2 = 0.f;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        accn[dt][r] -= mean;
        s2 += accn[dt][r] * accn[dt][r];
      }
    const float rstd = rsqrtf(quad_sum(s2) * (1.f / DH) + 1e-5f);
    if (t < S) {
      uint16_t* orow = out + ((int64_t)(b * S + t)) * D + h * DH;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        const f32x4 w = *(const f32x4*)(hn_w + 16 * dt + 4 * q), bb = *(const f32x4*)(hn_b + 16 * dt + 4 * q);
        const float y0 = accn[dt][0] * rstd * w[0] + bb[0], y1 = accn[dt][1] * rstd * w[1] + bb[1];
        const float y2 = accn[dt][2] * rstd * w[2] + bb[2], y3 = accn[dt][3] * rstd * w[3] + bb[3];
        *(uint2*)(orow + 16 * dt + 4 * q) = make_uint2(HT::pack(y0, y1), HT::pack(y2, y3));
      }
    }
  }
}

}  // namespace

bool perf_attn_supported(int dh, int S) { return dh == DH && S >= 1 && S <= 224; }
// the form that computes q | k | v itself holds 13 row tiles (its [R][384] row image must fit the 160 KiB of LDS) and D = 512
bool perf_attn_qkv_supported(int dh, int S, int H) { return dh == DH && H == 4 && S >= 1 && S <= 208; }

namespace {
template <int QMT>
int launch_perf_attn(uint16_t* qkv, int h16, const uint16_t* PT, int ldp, const float* hn_w, const float* hn_b, const int* len, int B,
                     int S, int H, uint16_t* out, const PerfQkv& qa, hipStream_t s) {
  const int TP = (S + 31) & ~31, TS = TP + 8;
  const int vreg = (TP * DH > MF * PS) ? TP * DH : MF * PS;
  int smem = (MF * TS + vreg + DH * PS) * 2;
  if (QMT > 0) {  // phase 0: two row stages, then the [R][384] 16-bit row image
    const int img = QMT * 16 * 768;
    smem = smem > img ? smem : img;
  }
  static DevInt attr;
  if (smem > attr) {
    if (hipFuncSetAttribute((const void*)perf_attn_kernel<HB, QMT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess ||
        hipFuncSetAttribute((const void*)perf_attn_kernel<HF, QMT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return MDM_ERR_LAUNCH;
    attr = smem;
  }
  if (h16 == MDM_H16_F16) {
    hipLaunchKernelGGL((perf_attn_kernel<HF, QMT>), dim3(B * H), dim3(NTH), smem, s, qkv, PT, ldp, hn_w, hn_b, len, S, H, out, qa);
  } else {
    hipLaunchKernelGGL((perf_attn_kernel<HB, QMT>), dim3(B * H), dim3(NTH), smem, s, qkv, PT, ldp, hn_w, hn_b, len, S, H, out, qa);
  }
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}
}  // namespace

// h16: format (MDM_H16_*) of the qkv rows, of the feature matrix P^T and of the output rows
int perf_attn(const void* qkv, int h16, const uint16_t* PT, int ldp, const float* hn_w, const float* hn_b, const int* len, int B,
              int S, int H, int dh, uint16_t* out, hipStream_t s) {
  if (!perf_attn_supported(dh, S) || (h16 != MDM_H16_BF16 && h16 != MDM_H16_F16)) return MDM_ERR_UNSUPPORTED;
  if (!qkv || !PT || !hn_w || !hn_b || !len || !out || (ldp & 7)) return MDM_ERR_ARG;
  return launch_perf_attn<0>((uint16_t*)qkv, h16, PT, ldp, hn_w, hn_b, len, B, S, H, out, PerfQkv(), s);
}

// the same with the q | k | v projection inside: xn 16-bit [B S, D] (format h16), wqkv the 16-bit plane [3 D][ldw] of the stacked
// query | key | value weights, bias [3 D], q | k | v = alpha * (xn W^T + b); qscratch = a [B S, 3 D] 16-bit buffer (its q third is used)
int perf_attn_qkv(const uint16_t* xn, const uint16_t* wqkv, int ldw, const float* bias, float alpha, uint16_t* qscratch, int h16,
                  const uint16_t* PT, int ldp, const float* hn_w, const float* hn_b, const int* len, int B, int S, int H, int dh,
                  uint16_t* out, hipStream_t s) {
  if (!perf_attn_qkv_supported(dh, S, H) || (h16 != MDM_H16_BF16 && h16 != MDM_H16_F16)) return MDM_ERR_UNSUPPORTED;
  if (!xn || !wqkv || !bias || !qscratch || !PT || !hn_w || !hn_b || !len || !out || (ldp & 7) || (ldw & 7) ||
      ((((uintptr_t)xn) | ((uintptr_t)wqkv) | ((uintptr_t)qscratch)) & 15))
    return MDM_ERR_ARG;
  const PerfQkv qa = {xn, wqkv, ldw, bias, alpha};
  // 7 row tiles cover the coarse scale (S <= 112), 13 the full one: the count is compiled in (see QkvGeo)
  if (S <= 112) return launch_perf_attn<7>(qscratch, h16, PT, ldp, hn_w, hn_b, len, B, S, H, out, qa, s);
  return launch_perf_attn<13>(qscratch, h16, PT, ldp, hn_w, hn_b, len, B, S, H, out, qa, s);
}

}  // namespace mdm
