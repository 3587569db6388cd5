// Fused Performer-style linear attention core (fast_attention.py:29-92) in the fp32-grade (bf16x3) arithmetic, head_dim 128.
// One launch replaces the five of the precision-3 chain (head_norm, feature GEMM, KV-state GEMM, numerator GEMM, den_ln) and the
// fp32 [M, 2 H m] feature tensor between them.  One workgroup (8 waves) per (batch, head).
//
// Input: q^ | k^ | v^ = the q | k | v projection AFTER the shared LayerNorm over head_dim and the L2 normalisation of q and k, as
// bf16 hi / lo planes [B S, 3 D] (written by the projection GEMM's epilogue, csrc/gemm3.hip ACT_HEADNORM).  Every product below
// is three MFMAs on the hi / lo splits (hi*lo + lo*hi + hi*hi, fp32 accumulate), like every other GEMM of this mode.
//
// hi / lo images cost twice the LDS of the 16-bit kernel (csrc/perf_attn.hip), so all of kphi^T, v and P^T cannot be resident.
// Two passes over the frames instead:
//   pass 1 (feature-stationary): wave w owns features m in [16 w, 16 w + 16); its 16 rows of P^T (hi + lo) live in registers.
//     Chunks of 32 frames of k^ and v^ stream global -> LDS by LDS-DMA through a 3-stage ring (one barrier per chunk).  Per
//     chunk: z = k^ P (two 16-frame tiles, D[t][m]: frames in the accumulator registers) -> kphi = 0.1 exp(clamp z), masked
//     past the length -> the accumulator IS the B operand of the next MFMA (k = the 32 frames) -> KV^T[d][m] += v^T[d][t]
//     kphi[t][m], v^T fragments out of the row-major v image through the transposing read.  kphi never touches LDS.
//   KV^T (x 0.1) -> LDS as hi / lo planes, columns stored in the k-slot order of pass 2's MFMA.
//   pass 2 (frame-stationary, as perf_attn.hip's Q phase): wave w owns frame tiles w, w + 8; q^ and k^ fragments global ->
//     registers; P^T (hi + lo) resident in LDS since the start: qphi AND kphi of the tile (the same P^T fragment feeds both),
//     same-t denominator in registers, the qphi accumulator is the B operand of num = qphi KV, LayerNorm over head_dim, fp32
//     rows out.  Recomputing kphi here (one third more MFMAs in this pass) is what keeps kphi out of memory.
// LDS: [0, 96 KiB) the ring, then KV^T hi | lo (64 KiB); [96, 160 KiB) P^T hi | lo.  All images are 256-byte rows of 16-bit
// elements with the 16-byte chunk c of row r at slot c ^ f(r): f = r & 15 where fragments are read by rows (conflict-free
// ds_read_b128 for the 16x16x32 operand maps), f = ((r & 7) << 1) | ((r >> 3) & 1) for v (the transposing read takes 8 rows x
// 32 bytes per half wave).
#include "kernels.h"

namespace mdm {
namespace {

typedef bf16x8_t bfr;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4a __attribute__((ext_vector_type(4)));

constexpr int A3_DH = 128, A3_NT = 512, A3_CH = 32;         // head_dim, threads, frames per chunk
constexpr int A3_PLANE = A3_CH * 256;                        // one plane of one tensor of a chunk: 8 KiB
constexpr int A3_STAGE = 4 * A3_PLANE, A3_NSTAGE = 3;        // k hi | k lo | v hi | v lo
constexpr int A3_RING = A3_NSTAGE * A3_STAGE;                // 96 KiB
constexpr int A3_IMG = 128 * 256;                            // one 128 x 128 16-bit plane: 32 KiB
constexpr int A3_SMEM = A3_RING + 2 * A3_IMG;                // 160 KiB
static_assert(2 * A3_IMG <= A3_RING, "KV^T hi | lo take the ring's place");

__device__ __forceinline__ int fk(int r) { return r & 15; }
__device__ __forceinline__ int fv(int r) { return ((r & 7) << 1) | ((r >> 3) & 1); }

__device__ __forceinline__ void a3_glds(const void* g, uint8_t* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void a3_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void a3_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ f32x4 mm(bfr a, bfr b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
// fp32-grade product of two hi / lo pairs, small terms first (as csrc/gemm3.hip)
__device__ __forceinline__ f32x4 mm3(bfr ah, bfr al, bfr bh, bfr bl, f32x4 c) {
  c = mm(ah, bl, c);
  c = mm(al, bh, c);
  return mm(ah, bh, c);
}
__device__ __forceinline__ float quad_sum3(float v) {  // across the 4 lanes that share (lane & 15)
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
// 8 fp32 (two accumulators' worth, in k-slot order) -> bf16 hi and lo fragments
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bfr& hi, bfr& lo) {
  uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
  split_bf16(a[0], a[1], h0, l0);
  split_bf16(a[2], a[3], h1, l1);
  split_bf16(b[0], b[1], h2, l2);
  split_bf16(b[2], b[3], h3, l3);
  const u32x4a h = {h0, h1, h2, h3}, l = {l0, l1, l2, l3};
  hi = __builtin_bit_cast(bfr, h);
  lo = __builtin_bit_cast(bfr, l);
}
// feature map 0.1 exp(clamp(z, +-15)) (:58-66) on the hardware exp2: z = q^ . P column with |q^| = 1 and |P column| = dh^-1/4, so
// |z| <= 0.3 and v_exp_f32's error (~1 ulp of the result, plus the rounding of z log2 e: < 2e-8 relative here) is at the level of
// libm's; the clamp is kept for inputs that are not normalised.  libm's expf costs ~25 instructions, 64 of them per frame tile.
__device__ __forceinline__ float feat(float z) {
  return 0.1f * __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(z, -15.f, 15.f) * 1.44269504088896340736f);
}

struct Attn3Args {
  const uint16_t* xh;  // q^ | k^ | v^ hi plane [B S, 3 D]
  const uint16_t* xl;  // lo plane
  const uint16_t* ph;  // P^T hi [128][ldp]
  const uint16_t* pl;  // P^T lo
  int ldp;
  const float* hn_w;   // shared LayerNorm over head_dim (the output's: fast_attention.py:85-90)
  const float* hn_b;
  const int* len;
  int S, H;
  float* out;          // fp32 [B S, D], or (out_x2) the same rows pre-split: MDM_OP_X2_ROW, 2 D 16-bit elements per row
  int out_x2;
};

__global__ __launch_bounds__(A3_NT) void perf_attn3_kernel(const Attn3Args g) {
  extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
  uint8_t* const ring = smem;
  uint8_t* const kvh = smem;                  // after pass 1
  uint8_t* const kvl = smem + A3_IMG;
  uint8_t* const pth = smem + A3_RING;
  uint8_t* const ptl = smem + A3_RING + A3_IMG;
  const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, q = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / g.H, h = blockIdx.x - b * g.H;
  const int S = g.S, D = g.H * A3_DH, D3 = 3 * D;
  const int nvalid = min(g.len[b], S);
  const int nchunk = (S + A3_CH - 1) / A3_CH;
  const int64_t rowbase = (int64_t)b * S;

  // ---- LDS-DMA sources of this wave: one tensor plane (wave >> 1: k hi, k lo, v hi, v lo), 16 of a chunk's 32 frames ----------
  // One instruction lands 4 rows x 256 B; lane -> row (lane >> 4), physical chunk (lane & 15) <- logical chunk phys ^ f(row).
  const int tp = wid >> 1;
  const uint16_t* const plane = (tp & 1) ? g.xl : g.xh;
  const int which = 1 + (tp >> 1);  // 1 = k, 2 = v
  int drow[4], dcol[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    drow[u] = 16 * (wid & 1) + 4 * u + (lane >> 4);
    dcol[u] = (((lane & 15) ^ ((tp >> 1) ? fv(drow[u]) : fk(drow[u]))) << 3) + which * D + h * A3_DH;
  }
  auto stage = [&](int c) __attribute__((always_inline)) {
    uint8_t* dst = ring + (c % A3_NSTAGE) * A3_STAGE + tp * A3_PLANE + (wid & 1) * 4096;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int t = c * A3_CH + drow[u];
      t = t < S ? t : S - 1;
      a3_glds(plane + (rowbase + t) * D3 + dcol[u], dst + u * 1024);
    }
  };
  stage(0);
  if (nchunk > 1) stage(1);

  // ---- P^T: this wave's 16 feature rows into registers (pass 1), all 128 rows into LDS (pass 2) ---------------------------------
  bfr Ph[4], Pl[4];
  {
    const int64_t po = (int64_t)(16 * wid + r16) * g.ldp + 8 * q;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      Ph[ks] = *(const bfr*)(g.ph + po + 32 * ks);
      Pl[ks] = *(const bfr*)(g.pl + po + 32 * ks);
    }
    // 2 planes x 128 rows x 16 chunks = 4096 chunks of 16 B, 8 per thread
    uint4 tmp[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int id = tid + A3_NT * i, pln = id >> 11, row = (id >> 4) & 127, ch = id & 15;
      tmp[i] = *(const uint4*)((pln ? g.pl : g.ph) + (int64_t)row * g.ldp + ch * 8);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int id = tid + A3_NT * i, pln = id >> 11, row = (id >> 4) & 127, ch = id & 15;
      *(uint4*)((pln ? ptl : pth) + row * 256 + ((ch ^ fk(row)) << 4)) = tmp[i];
    }
  }

  // ---- pass 1: KV^T[d][m] = sum_t v^[t][d] kphi[t][m] over the chunks ------------------------------------------------------
  f32x4 kv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) kv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int qp = r16 >> 2, pp = r16 & 3;  // transposing read: this lane addresses row qp, columns 4 pp .. 4 pp + 3 of its group's block
#pragma unroll 1
  for (int c = 0; c < nchunk; ++c) {
    // chunk c has landed: each wave waits for its own pieces (all but the 4 of the younger chunk), then everybody's (barrier)
    if (c + 1 < nchunk) {
      a3_wait_vm<4>();
    } else {
      a3_wait_vm<0>();
    }
    a3_barrier();  // ... and every wave is done with chunk c - 1, whose stage takes chunk c + 2
    if (c + 2 < nchunk) stage(c + 2);
    const uint8_t* sb = ring + (c % A3_NSTAGE) * A3_STAGE;
    f32x4 kf[2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const int row = 16 * ti + r16;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int off = row * 256 + (((4 * ks + q) ^ fk(row)) << 4);
        const bfr kh = *(const bfr*)(sb + off), kl = *(const bfr*)(sb + A3_PLANE + off);
        z = mm3(kh, kl, Ph[ks], Pl[ks], z);  // D[t][m]: m = r16, t = 4 q + reg
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) kf[ti][r] = (c * A3_CH + 16 * ti + 4 * q + r < nvalid) ? feat(z[r]) : 0.f;  // key mask (:69-74)
    }
    bfr xh, xl;  // k slots 0..3 <-> t = 4 q + j of tile 0, slots 4..7 <-> t = 16 + 4 q + (j - 4)
    split8(kf[0], kf[1], xh, xl);
    const int tr0 = 4 * q + qp;
#pragma unroll
    for (int j = 0; j < 8; ++j) {  // rows d = 16 j + r16 of v^T, k = the chunk's frames in slot order
      const int ch = 2 * j + (pp >> 1);
      const int o0 = tr0 * 256 + ((ch ^ fv(tr0)) << 4) + 8 * (pp & 1);
      const int o1 = (tr0 + 16) * 256 + ((ch ^ fv(tr0 + 16)) << 4) + 8 * (pp & 1);
      const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sb + 2 * A3_PLANE + o0));
      const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sb + 2 * A3_PLANE + o1));
      const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sb + 3 * A3_PLANE + o0));
      const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sb + 3 * A3_PLANE + o1));
      const bfr vh = __builtin_bit_cast(bfr, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
      const bfr vl = __builtin_bit_cast(bfr, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
      kv[j] = mm3(vh, vl, xh, xl, kv[j]);  // D[d][m]: m = r16, d = 16 j + 4 q + reg
    }
  }
  // pass 2's operands of this wave's first frame tile are requested now: they land under the state's write and its barriers
  const int ntile = (S + 15) >> 4;
  bfr qh[4], ql[4], kh[4], kl[4];
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    const int t = tile * 16 + r16;
    const int64_t ro = (rowbase + (t < S ? t : S - 1)) * D3 + h * A3_DH + 8 * q;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qh[ks] = *(const bfr*)(g.xh + ro + 32 * ks);
      ql[ks] = *(const bfr*)(g.xl + ro + 32 * ks);
      kh[ks] = *(const bfr*)(g.xh + ro + D + 32 * ks);
      kl[ks] = *(const bfr*)(g.xl + ro + D + 32 * ks);
    }
  };
  fetch(wid < ntile ? wid : ntile - 1);
  a3_barrier();  // every wave is done reading the ring: its LDS takes the state
  // KV^T x 0.1 (:77) as hi / lo planes [d][128]; column m sits at position 32 (m >> 5) + 8 ((m & 15) >> 2) + 4 ((m >> 4) & 1) + (m & 3):
  // the 8 k slots a lane of pass 2 multiplies are then one 16-byte read
  {
    const int pos = 32 * (wid >> 1) + 8 * (r16 >> 2) + 4 * (wid & 1) + (r16 & 3);
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int d = 16 * j + 4 * q + r;
        const float v = 0.1f * kv[j][r];
        uint32_t hh, ll;
        split_bf16(v, 0.f, hh, ll);
        const int off = d * 256 + (((pos >> 3) ^ fk(d)) << 4) + (pos & 7) * 2;
        *(uint16_t*)(kvh + off) = (uint16_t)(hh & 0xffffu);
        *(uint16_t*)(kvl + off) = (uint16_t)(ll & 0xffffu);
      }
  }
  a3_barrier();

  // ---- pass 2: per frame tile  qphi, kphi -> den;  num = qphi KV;  LN(0.1 num / den) -> out ---------------------------------
#pragma unroll 1
  for (int tile = wid; tile < ntile; tile += 8) {
    const int t0 = tile * 16, t = t0 + r16;
    // The LDS addresses of the P^T and KV^T fragments do not depend on the tile: made opaque per tile, or hipcc hoists all 128
    // fragment reads (512 registers) out of this loop and parks them in scratch memory
    int ro16 = r16 * 256, rx = r16;
    asm volatile("" : "+v"(ro16), "+v"(rx));
    f32x4 aq[8], ak[8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) aq[mt] = (f32x4){0.f, 0.f, 0.f, 0.f}, ak[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        const int off = 4096 * mt + ro16 + (((4 * ks + q) ^ rx) << 4);
        const bfr ah = *(const bfr*)(pth + off), al = *(const bfr*)(ptl + off);
        aq[mt] = mm3(ah, al, qh[ks], ql[ks], aq[mt]);  // D[m][t]: t = r16, m = 16 mt + 4 q + reg
        ak[mt] = mm3(ah, al, kh[ks], kl[ks], ak[mt]);
      }
    if (tile + 8 < ntile) fetch(tile + 8);  // the fragments are consumed: the next tile's land under the rest of this one
    const bool valid = t < nvalid;
    float den = 0.f;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float fq = feat(aq[mt][r]);
        const float fkk = valid ? feat(ak[mt][r]) : 0.f;
        aq[mt][r] = fq;
        den += fq * fkk;  // same-t dot (:81)
      }
    den = fmaxf(quad_sum3(den), 1e-6f);
    f32x4 an[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) an[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bfr bh, bl;  // k slots 0..3 <-> m = 32 s + 4 q + j, 4..7 <-> m = 32 s + 16 + 4 q + (j - 4)
      split8(aq[2 * s], aq[2 * s + 1], bh, bl);
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        const int off = 4096 * dt + ro16 + (((4 * s + q) ^ rx) << 4);
        const bfr ah = *(const bfr*)(kvh + off), al = *(const bfr*)(kvl + off);
        an[dt] = mm3(ah, al, bh, bl, an[dt]);  // D[d][t]: t = r16, d = 16 dt + 4 q + reg
      }
    }
    // out = LN_dh(0.1 * num / den)   (:78,85-90)
    const float sc = 0.1f / den;
    float s1 = 0.f;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        an[dt][r] *= sc;
        s1 += an[dt][r];
      }
    const float mean = quad_sum3(s1) * (1.f / A3_DH);
    float s2 = 0.f;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        an[dt][r] -= mean;
        s2 += an[dt][r] * an[dt][r];
      }
    const float rstd = rsqrtf(quad_sum3(s2) * (1.f / A3_DH) + 1e-5f);
    if (t < S) {
      float* orow = g.out + (rowbase + t) * D + h * A3_DH;
      uint16_t* xrow = (uint16_t*)g.out + (rowbase + t) * 2 * D;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        const f32x4 w = *(const f32x4*)(g.hn_w + 16 * dt + 4 * q), bb = *(const f32x4*)(g.hn_b + 16 * dt + 4 * q);
        const f32x4 y = {an[dt][0] * rstd * w[0] + bb[0], an[dt][1] * rstd * w[1] + bb[1], an[dt][2] * rstd * w[2] + bb[2],
                         an[dt][3] * rstd * w[3] + bb[3]};
        if (g.out_x2) {
          store_x2_4(xrow, h * A3_DH + 16 * dt + 4 * q, y[0], y[1], y[2], y[3]);
        } else {
          *(f32x4*)(orow + 16 * dt + 4 * q) = y;
        }
      }
    }
  }
}

}  // namespace

bool perf_attn3_supported(int dh, int S) { return dh == A3_DH && S >= 1 && S <= 224; }

// xh / xl: hi / lo planes of the normalised q | k | v rows [B S, 3 D] (D = H * 128); ph / pl: P^T planes [128][ldp]; out fp32 [B S, D]
int perf_attn3(const uint16_t* xh, const uint16_t* xl, const uint16_t* ph, const uint16_t* pl, int ldp, const float* hn_w,
               const float* hn_b, const int* len, int B, int S, int H, int dh, float* out, int out_x2, hipStream_t s) {
  if (!perf_attn3_supported(dh, S)) return MDM_ERR_UNSUPPORTED;
  if (!xh || !xl || !ph || !pl || !hn_w || !hn_b || !len || !out || B <= 0 || H <= 0 || (ldp & 7) ||
      ((((uintptr_t)xh) | ((uintptr_t)xl) | ((uintptr_t)ph) | ((uintptr_t)pl) | ((uintptr_t)hn_w) | ((uintptr_t)hn_b) | ((uintptr_t)out)) & 15))
    return MDM_ERR_ARG;
  static DevOnce attr;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)perf_attn3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, A3_SMEM) != hipSuccess)
      return MDM_ERR_LAUNCH;
    attr = true;
  }
  const Attn3Args g = {xh, xl, ph, pl, ldp, hn_w, hn_b, len, S, H, out, out_x2};
  hipLaunchKernelGGL(perf_attn3_kernel, dim3(B * H), dim3(A3_NT), A3_SMEM, s, g);
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

}  // namespace mdm
