// Projection phase shared by the fused attention cores (perf_attn.hip: q | k | v of one head, xattn.hip: the query of one head):
//   [R x 16 NJ 8] = rows [R x 512] . W^T, one workgroup of 8 waves per (batch, head).
// The rows are what the waves share: K slices of 64 columns go global -> registers -> LDS one slice ahead (two LDS stages).  The
// weights are private to a wave (the waves split the column tiles, NJ each, and multiply all MT row tiles): they stream global ->
// registers through a ring two K steps ahead and never touch LDS.
#pragma once
#include "kernels.h"

namespace mdm {
namespace {

constexpr int PROJ_NT = 512;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// phase-0 staging of one K slice (64 columns) of the sample's xn rows: global -> registers -> LDS (free functions on purpose: an
// array captured by a lambda keeps its stack slot = scratch memory).  Branch-free loads: chunk ids past the end re-read the last
// chunk and are not landed.  16-B chunk c of row r at slot c ^ ((r >> 1) & 7): two 128-B rows span the 64 banks, so the 16 rows of
// a fragment read are conflict-free when the 8 rows of either parity take 8 different slots.
template <int NCH>
__device__ __forceinline__ void qkv_fetch(u32x4 (&st)[NCH], int tid, int R, int S, int D, int kk, const uint16_t* __restrict__ xrow0) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int id = tid + PROJ_NT * i;
    id = id < R * 8 ? id : R * 8 - 1;
    int row = id >> 3;
    row = row < S ? row : S - 1;
    st[i] = *(const u32x4*)(xrow0 + (int64_t)row * D + kk * 64 + (id & 7) * 8);
  }
}
template <int NCH>
__device__ __forceinline__ void qkv_land(const u32x4 (&st)[NCH], int tid, int R, uint8_t* stage) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int id = tid + PROJ_NT * i;
    if (id < R * 8) {
      const int row = id >> 3;
      *(u32x4*)(stage + row * 128 + ((((id & 7)) ^ ((row >> 1) & 7)) << 4)) = st[i];
    }
  }
}

// Row tiles the fused form holds: 13 (S <= 208: the [R][384] row image must fit the LDS) or 7 (S <= 112, the coarse scale).  The
// count is a template argument: with a run-time bound every row tile is its own basic block and its fragment read is waited for
// right in front of its three MFMAs.
template <int MT>
struct QkvGeo {
  static constexpr int R = MT * 16;                       // rows staged and multiplied (rows >= S: copies of the last row)
  static constexpr int NCH = (R * 8 + PROJ_NT - 1) / PROJ_NT;     // 16-B chunks per thread and K slice
};

// one K slice (two K steps of 32): land its rows in `stage`, barrier, request the next slice into the same registers, MFMAs.
// wr: the wave's weight ring, NJ fragments (column tiles of the wave) per K step x 2 steps; a slot is refilled with the fragment two steps ahead right after use.
// The row fragments go through two registers, one read ahead of the MFMAs, pinned per row tile.
template <typename HT, int MT, int NJ, bool PIN = true>
__device__ __forceinline__ void qkv_slice(int kk, u32x4 (&st)[QkvGeo<MT>::NCH], f32x4 (&acc)[MT][NJ], typename HT::frag_t (&wr)[2 * NJ],
                                          const uint16_t* const (&wrow)[NJ], uint8_t* stage, int tid, int r16, int q, int S, int D,
                                          const uint16_t* __restrict__ xrow0) {
  typedef typename HT::frag_t frag_t;
  constexpr int R = QkvGeo<MT>::R, NCH = QkvGeo<MT>::NCH;
  int to = tid;  // opaque per slice: hoisted out of the K loop the staging addresses would be spilled
  asm volatile("" : "+v"(to));
  qkv_land<NCH>(st, to, R, stage);
  __syncthreads();
  if (kk + 1 < 8) qkv_fetch<NCH>(st, to, R, S, D, kk + 1, xrow0);
  const uint8_t* xb = stage + r16 * 128;
  constexpr int NA = 4, PD = NA - 1;  // row-fragment ring: PD fragments ahead of the MFMAs (one ahead leaves half the LDS latency exposed)
  frag_t xf[NA];
#pragma unroll
  for (int n = 0; n < PD; ++n) xf[n] = *(const frag_t*)(xb + (n % MT) * 2048 + ((((n / MT) * 4 + q) ^ ((r16 >> 1) & 7)) << 4));
#pragma unroll
  for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int c = k2 * MT + mt, n = c + PD;  // this fragment, and the one requested now: (step n / MT, row tile n % MT)
      if (n < 2 * MT) xf[n % NA] = *(const frag_t*)(xb + (n % MT) * 2048 + ((((n / MT) * 4 + q) ^ ((r16 >> 1) & 7)) << 4));
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[mt][j] = HT::mfma16(wr[NJ * k2 + j], xf[c % NA], acc[mt][j]);  // lane: row 16 mt + r16, cols 16 ct + 4 q ..
      if (PIN) __builtin_amdgcn_sched_barrier(0);
    }
    int nxt = 2 * kk + k2 + 2;  // the K step this slot holds next (past the end: re-read the last one, never used)
    nxt = nxt < 16 ? nxt : 15;
#pragma unroll
    for (int j = 0; j < NJ; ++j) wr[NJ * k2 + j] = *(const frag_t*)(wrow[j] + 32 * nxt);
  }
}

}  // namespace
}  // namespace mdm
