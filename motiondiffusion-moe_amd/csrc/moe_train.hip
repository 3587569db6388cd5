// Training step of the MoE feed-forward block: forward with saved activations and the full
// backward of
//     out = x + proj_out( mean_b SwitchMoE_b(LayerNorm_b(x)), emb )        multi_branch.py:52-61
// i.e. both SwitchMoELayers (gate Linear -> softmax -> top-2 -> expert MLPs -> probability-weighted sum, switch_moe.py:44-111)
// and the StylizationBlock (stylization.py:20-31, including its emb_layers Linear), plus the load-balancing loss of
// switch_moe.py:113-145 from this forward's device-side counters (it is built from buffers: it has a value, no gradient).
//
// fp32 master parameters in the reference's layouts; every GEMM is the bf16x3 (fp32-grade) MFMA kernel of gemm.hip:
//   forward        Linear                     C = A W^T            A row-major, W row-major
//   data gradient  dA = dC W                  C = A W'^T           W' = W read k-strided
//   weight gradient dW = dC^T A               C = A'^T-form        both operands k-strided, K = rows; per expert group the K
//                                                                  range comes from the routing offsets (MdmGemmDesc.kgoff)
// so the expert weight gradients are 2E-batched MFMA GEMMs over exactly the routed rows of each expert, with no host sync.
// Row-wise backward kernels (one wave per row) cover LayerNorm, the stylization gate, softmax/top-2 and the gathers; column
// sums (biases, LayerNorm gains, per-sample scale/shift) are accumulated with float atomics.  Dropout (multi_branch.py:57 on
// each branch's output, stylization.py:16 after the SiLU) uses counter-based masks: element (row, col) of site s keeps its
// value when the top 24 bits of word col % 4 of Philox4x32-10(counter = (col / 4, row, 0, s), key = seed) are >= p * 2^24, and is
// scaled by 1 / (1 - p) -- recomputed in the backward, never stored; p = 0 skips the generator.  The step is deterministic up
// to the atomics' summation order.
#include "gemm.h"
#include "kernels.h"
#include "philox.h"
#include "row.h"

#define MDM_TRY(expr)                \
  do {                               \
    int st__ = (expr);               \
    if (st__ != MDM_OK) return st__; \
  } while (0)

namespace mdm {
namespace {

template <int NE, bool VEC>
__device__ __forceinline__ int row_col(int j, int lane) {
  if constexpr (VEC) {
    return 4 * (lane + 64 * (j >> 2)) + (j & 3);
  } else {
    return lane + 64 * j;
  }
}

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752440f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}
__device__ __forceinline__ float silu_grad(float z) {
  const float sg = 1.f / (1.f + expf(-z));
  return sg * (1.f + z * (1.f - sg));
}

// dropout masks for the row image of a wave: m.e[j] = 1 / (1 - p) where the element is kept, 0 where it is dropped
struct Drop {
  uint32_t thr;  // p * 2^24 (0 = no dropout)
  float keep;    // 1 / (1 - p)
  uint32_t k0, k1;
};
template <int NE, bool VEC>
__device__ __forceinline__ void drop_mask(const Drop& d, int site, int64_t row, int lane, Row<NE, VEC>& m) {
  static_assert(VEC, "dropout masks are generated per float4 chunk");
#pragma unroll
  for (int c = 0; c < NE / 4; ++c) {
    uint32_t ctr[4] = {(uint32_t)(lane + 64 * c), (uint32_t)row, (uint32_t)((uint64_t)row >> 32), (uint32_t)site};
    philox4x32_10(ctr, d.k0, d.k1);
#pragma unroll
    for (int w = 0; w < 4; ++w) m.e[4 * c + w] = (ctr[w] >> 8) >= d.thr ? d.keep : 0.f;
  }
}
enum { DROP_BRANCH0 = 0, DROP_BRANCH1 = 1, DROP_STYLE = 2 };

// training-mode stylization input (style_in_kernel with the two dropout sites): a = 0.5 * sum_b drop_b(y2[b,0] + y2[b,1]),
// s = drop_s(SiLU(LN(a) (1 + scale) + shift))
template <int NE, bool VEC>
__global__ __launch_bounds__(256) void style_fwd_train_kernel(const float* __restrict__ y2, const int* __restrict__ pos4,
                                                              const float* __restrict__ sc, const float* __restrict__ nw,
                                                              const float* __restrict__ nb, int64_t M, int D, int S, Drop dr,
                                                              float* __restrict__ sact) {
  const int lane = threadIdx.x & 63;
  for (int64_t row = blockIdx.x * (int64_t)WPB + (threadIdx.x >> 6); row < M; row += (int64_t)gridDim.x * WPB) {
    Row<NE, VEC> a, b, c, d, m0, m1, ms;
    a.load(y2 + (int64_t)pos4[row * 4 + 0] * D, D, lane);
    b.load(y2 + (int64_t)pos4[row * 4 + 1] * D, D, lane);
    c.load(y2 + (int64_t)pos4[row * 4 + 2] * D, D, lane);
    d.load(y2 + (int64_t)pos4[row * 4 + 3] * D, D, lane);
    if constexpr (VEC) {
      drop_mask(dr, DROP_BRANCH0, row, lane, m0);
      drop_mask(dr, DROP_BRANCH1, row, lane, m1);
      drop_mask(dr, DROP_STYLE, row, lane, ms);
    } else {  // (the host only launches this kernel at the vectorised widths)
#pragma unroll
      for (int j = 0; j < NE; ++j) m0.e[j] = m1.e[j] = ms.e[j] = 1.f;
    }
#pragma unroll
    for (int j = 0; j < NE; ++j) a.e[j] = ((a.e[j] + b.e[j]) * m0.e[j] + (c.e[j] + d.e[j]) * m1.e[j]) * 0.5f;
    a.layernorm(nw, nb, D, lane);
    const float* scb = sc + (row / S) * 2 * (int64_t)D;
    Row<NE, VEC> scale, shift;
    scale.load(scb, D, lane);
    shift.load(scb + D, D, lane);
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const float z = a.e[j] * (1.f + scale.e[j]) + shift.e[j];
      a.e[j] = z / (1.f + expf(-z)) * ms.e[j];
    }
    a.store(sact + row * D, D, lane);
  }
}

// op 0: y = gelu(x); 1: y *= gelu'(x); 2: y = silu(x); 3: y *= silu'(x)
template <int OP>
__global__ __launch_bounds__(256) void ew_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    if constexpr (OP == 0) y[i] = gelu_exact(v);
    if constexpr (OP == 1) y[i] *= gelu_grad(v);
    if constexpr (OP == 2) y[i] = v / (1.f + expf(-v));
    if constexpr (OP == 3) y[i] *= silu_grad(v);
  }
}

template <int NE, bool VEC>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ perm,
                                                          float* __restrict__ dst, int64_t rows, int D) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = blockIdx.x * (int64_t)WPB + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * WPB) {
    Row<NE, VEC> v;
    v.load(src + (int64_t)perm[r] * D, D, lane);
    v.store(dst + r * D, D, lane);
  }
}

// stylization backward (stylization.py:26-30): a = mean of the branches (rebuilt from the 4 routed rows), n = LN(a),
// z = n (1 + scale) + shift, s = SiLU(z); given ds = dL/ds writes dzz = [dz * n | dz] (per-sample sums give d scale | d shift),
// da_half = 0.5 * dL/da (what each branch receives), and accumulates the LayerNorm gain / bias gradients
template <int NE, bool VEC>
__global__ __launch_bounds__(256) void style_bwd_kernel(const float* __restrict__ y2, const int* __restrict__ pos4,
                                                        const float* ds, const float* __restrict__ sc,
                                                        const float* __restrict__ nw, const float* __restrict__ nb, int64_t M,
                                                        int D, int S, Drop dr, float* __restrict__ dzz, float* da_half,
                                                        float* __restrict__ g_nw, float* __restrict__ g_nb) {
  const int lane = threadIdx.x & 63;
  __shared__ float red[2 * 1024];  // the block's column sums (gain | bias gradients): one global atomic per column and block
  for (int i = threadIdx.x; i < 2 * D; i += 256) red[i] = 0.f;
  __syncthreads();
  float gw[NE], gb[NE];
#pragma unroll
  for (int j = 0; j < NE; ++j) gw[j] = gb[j] = 0.f;
  Row<NE, VEC> w, bsr;
  w.load(nw, D, lane);
  bsr.load(nb, D, lane);
  for (int64_t row = blockIdx.x * (int64_t)WPB + (threadIdx.x >> 6); row < M; row += (int64_t)gridDim.x * WPB) {
    Row<NE, VEC> a, b, c, d, g;
    a.load(y2 + (int64_t)pos4[row * 4 + 0] * D, D, lane);
    b.load(y2 + (int64_t)pos4[row * 4 + 1] * D, D, lane);
    c.load(y2 + (int64_t)pos4[row * 4 + 2] * D, D, lane);
    d.load(y2 + (int64_t)pos4[row * 4 + 3] * D, D, lane);
    g.load(ds + row * D, D, lane);
    const float* scb = sc + (row / S) * 2 * (int64_t)D;
    Row<NE, VEC> scale, shift;
    scale.load(scb, D, lane);
    shift.load(scb + D, D, lane);
    if constexpr (VEC) {
      if (dr.thr) {  // the forward's masks, regenerated: branch outputs and the incoming gradient (taken after the SiLU dropout)
        Row<NE, VEC> m0, m1, ms;
        drop_mask(dr, DROP_BRANCH0, row, lane, m0);
        drop_mask(dr, DROP_BRANCH1, row, lane, m1);
        drop_mask(dr, DROP_STYLE, row, lane, ms);
#pragma unroll
        for (int j = 0; j < NE; ++j) {
          a.e[j] = (a.e[j] + b.e[j]) * m0.e[j], b.e[j] = 0.f;
          c.e[j] = (c.e[j] + d.e[j]) * m1.e[j], d.e[j] = 0.f;
          g.e[j] *= ms.e[j];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NE; ++j) a.e[j] = ((a.e[j] + b.e[j]) + (c.e[j] + d.e[j])) * 0.5f;
    const float mean = a.sum() / D;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const bool in = VEC || (lane + 64 * j < D);
      a.e[j] = in ? a.e[j] - mean : 0.f;
      s += a.e[j] * a.e[j];
    }
    const float rstd = rsqrtf(wave_sum(s) / D + 1e-5f);
    Row<NE, VEC> dz_n, dz;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const float nh = a.e[j] * rstd;                      // normalised
      const float n = nh * w.e[j] + bsr.e[j];              // LayerNorm output
      const float z = n * (1.f + scale.e[j]) + shift.e[j];
      const float dzv = g.e[j] * silu_grad(z);
      dz.e[j] = dzv, dz_n.e[j] = dzv * n;
      const float dn = dzv * (1.f + scale.e[j]);
      gw[j] += dn * nh, gb[j] += dn;
      const float dnh = dn * w.e[j];
      a.e[j] = nh, g.e[j] = dnh;
      s1 += dnh, s2 += dnh * nh;
    }
    const float m1 = wave_sum(s1) / D, m2 = wave_sum(s2) / D;
#pragma unroll
    for (int j = 0; j < NE; ++j) g.e[j] = 0.5f * rstd * (g.e[j] - m1 - a.e[j] * m2);
    dz_n.store(dzz + row * 2 * (int64_t)D, D, lane);
    dz.store(dzz + row * 2 * (int64_t)D + D, D, lane);
    g.store(da_half + row * D, D, lane);
  }
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    const int col = row_col<NE, VEC>(j, lane);
    if (col < D) atomicAdd(red + col, gw[j]), atomicAdd(red + D + col, gb[j]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += 256) atomicAdd((i < D ? g_nw : g_nb - D) + i, red[i]);
}

// routed rows: y2[r] = p_r * expert(h)[r] was added to its token's branch output (switch_moe.py:109):
//   dy[r] = p_r * dout_b[token];   dp[r] = <dout_b[token], expert output> = <dout_b[token], y2[r]> / p_r
template <int NE, bool VEC>
__global__ __launch_bounds__(256) void routed_bwd_kernel(const float* __restrict__ da_half, const float* __restrict__ y2,
                                                         const int* __restrict__ perm, const float* __restrict__ rowscale,
                                                         int64_t M, int64_t rows, int D, Drop dr, float* __restrict__ dy,
                                                         float* __restrict__ dp) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = blockIdx.x * (int64_t)WPB + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * WPB) {
    const int64_t tok = perm[r] % M;
    const float p = rowscale[r];
    Row<NE, VEC> d, y;
    d.load(da_half + tok * D, D, lane);
    y.load(y2 + r * D, D, lane);
    if constexpr (VEC) {
      if (dr.thr) {  // this branch's output dropout (multi_branch.py:57): the gradient passes through the same mask
        Row<NE, VEC> mb;
        drop_mask(dr, perm[r] >= M ? DROP_BRANCH1 : DROP_BRANCH0, tok, lane, mb);
#pragma unroll
        for (int j = 0; j < NE; ++j) d.e[j] *= mb.e[j];
      }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      s += d.e[j] * y.e[j];
      d.e[j] *= p;
    }
    s = wave_sum(s);
    d.store(dy + r * D, D, lane);
    if (lane == 0) dp[r] = p > 1e-30f ? s / p : 0.f;
  }
}

// per token, both branches: gate softmax / top-2 backward (switch_moe.py:53-57), the gather of the expert input
// gradients, the gate Linear's data gradient and the branch LayerNorm backward (multi_branch.py:55); dx = dout + both
template <int NE, bool VEC>
__global__ __launch_bounds__(256) void gate_ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dout,
                                                          const float* __restrict__ dxg, const float* __restrict__ dp,
                                                          const int* __restrict__ pos4, const int* __restrict__ top_idx,
                                                          const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                          const float* __restrict__ gate_w, const float* __restrict__ gate_b,
                                                          int64_t M, int D, int E, float* __restrict__ dlogits,
                                                          float* __restrict__ dx, float* __restrict__ g_ln_w,
                                                          float* __restrict__ g_ln_b) {
  const int lane = threadIdx.x & 63;
  extern __shared__ __attribute__((aligned(16))) float gl_smem[];  // gate rows [2][E][D], then the block's column sums [4][D]
  float* gws = gl_smem;
  float* red = gl_smem + 2 * E * D;
  for (int i = threadIdx.x; i < 2 * E * D; i += 256) gws[i] = gate_w[i];
  for (int i = threadIdx.x; i < 4 * D; i += 256) red[i] = 0.f;
  __syncthreads();
  float gw[2][NE], gb[2][NE];
#pragma unroll
  for (int br = 0; br < 2; ++br)
#pragma unroll
    for (int j = 0; j < NE; ++j) gw[br][j] = gb[br][j] = 0.f;
  for (int64_t row = blockIdx.x * (int64_t)WPB + (threadIdx.x >> 6); row < M; row += (int64_t)gridDim.x * WPB) {
    Row<NE, VEC> xh, acc;
    xh.load(x + row * D, D, lane);
    acc.load(dout + row * D, D, lane);
    const float mean = xh.sum() / D;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const bool in = VEC || (lane + 64 * j < D);
      xh.e[j] = in ? xh.e[j] - mean : 0.f;
      s += xh.e[j] * xh.e[j];
    }
    const float rstd = rsqrtf(wave_sum(s) / D + 1e-5f);
#pragma unroll
    for (int j = 0; j < NE; ++j) xh.e[j] *= rstd;
#pragma unroll 1
    for (int br = 0; br < 2; ++br) {
      Row<NE, VEC> w, b, h;
      w.load(ln_w + br * D, D, lane);
      b.load(ln_b + br * D, D, lane);
#pragma unroll
      for (int j = 0; j < NE; ++j) h.e[j] = xh.e[j] * w.e[j] + b.e[j];
      const float* gwb = gws + br * E * D;
      float logit[16], mx = -3.0e38f;
      for (int e = 0; e < E; ++e) {
        Row<NE, VEC> g;
        g.load(gwb + (int64_t)e * D, D, lane);
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < NE; ++j) t += h.e[j] * g.e[j];
        logit[e] = wave_sum(t) + gate_b[br * E + e];
        mx = fmaxf(mx, logit[e]);
      }
      float den = 0.f;
      for (int e = 0; e < E; ++e) logit[e] = expf(logit[e] - mx), den += logit[e];
      const int i1 = top_idx[(int64_t)br * 2 * M + 2 * row], i2 = top_idx[(int64_t)br * 2 * M + 2 * row + 1];
      const int r1 = pos4[row * 4 + 2 * br], r2 = pos4[row * 4 + 2 * br + 1];
      const float dp1 = dp[r1], dp2 = dp[r2];
      float c = 0.f;
      for (int e = 0; e < E; ++e) {
        logit[e] /= den;  // probabilities
        c += logit[e] * ((e == i1 ? dp1 : 0.f) + (e == i2 ? dp2 : 0.f));
      }
      Row<NE, VEC> dh, t2;
      dh.load(dxg + (int64_t)r1 * D, D, lane);
      t2.load(dxg + (int64_t)r2 * D, D, lane);
#pragma unroll
      for (int j = 0; j < NE; ++j) dh.e[j] += t2.e[j];
      for (int e = 0; e < E; ++e) {
        const float dl = logit[e] * ((e == i1 ? dp1 : 0.f) + (e == i2 ? dp2 : 0.f) - c);
        if (lane == 0) dlogits[((int64_t)br * M + row) * E + e] = dl;
        Row<NE, VEC> g;
        g.load(gwb + (int64_t)e * D, D, lane);
#pragma unroll
        for (int j = 0; j < NE; ++j) dh.e[j] += dl * g.e[j];
      }
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        gw[br][j] += dh.e[j] * xh.e[j], gb[br][j] += dh.e[j];
        dh.e[j] *= w.e[j];
        s1 += dh.e[j], s2 += dh.e[j] * xh.e[j];
      }
      const float m1 = wave_sum(s1) / D, m2 = wave_sum(s2) / D;
#pragma unroll
      for (int j = 0; j < NE; ++j) acc.e[j] += rstd * (dh.e[j] - m1 - xh.e[j] * m2);
    }
    acc.store(dx + row * D, D, lane);
  }
#pragma unroll
  for (int br = 0; br < 2; ++br)
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int col = row_col<NE, VEC>(j, lane);
      if (col < D) atomicAdd(red + br * D + col, gw[br][j]), atomicAdd(red + (2 + br) * D + col, gb[br][j]);
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * D; i += 256) atomicAdd((i < 2 * D ? g_ln_w : g_ln_b - 2 * D) + i, red[i]);
}

// out[g][c] += sum of X[r][c] over the rows r of group g; groups = row ranges goff[g]..goff[g+1], or uniform group_rows, or
// everything (both null / 0).  One thread per column, 256 rows per block: coalesced reads, one atomic per (block, group, col)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int64_t ld, int64_t rows, int C,
                                                     const int* __restrict__ goff, int ngroups, int64_t group_rows,
                                                     float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.y * 256;
  int64_t r1 = r0 + 256;
  r1 = r1 < rows ? r1 : rows;
  if (c >= C || r0 >= rows) return;
  int g = 0;
  int64_t gend = rows;
  if (goff) {
    while (g + 1 < ngroups && r0 >= goff[g + 1]) ++g;
    gend = goff[g + 1];
  } else if (group_rows > 0) {
    g = (int)(r0 / group_rows);
    gend = (int64_t)(g + 1) * group_rows;
  }
  float acc = 0.f;
  for (int64_t r = r0; r < r1; ++r) {
    while (r >= gend) {
      atomicAdd(out + (int64_t)g * C + c, acc);
      acc = 0.f, ++g;
      gend = goff ? goff[g + 1] : (int64_t)(g + 1) * group_rows;
    }
    acc += X[r * ld + c];
  }
  atomicAdd(out + (int64_t)g * C + c, acc);
}

// fp32 master weights -> bf16 hi / lo planes for the LDS-DMA staged bf16x3 GEMM (gemm3.hip), once per step and orientation:
// groups of [R][C] matrices, straight ([R][C]: forward Linear) or transposed ([C][R]: the data-gradient GEMM reads W^T rows)
__global__ __launch_bounds__(256) void pack_planes_kernel(const float* __restrict__ src, int R, int C, int transpose,
                                                          uint16_t* __restrict__ hi, uint16_t* __restrict__ lo) {
  __shared__ float tile[32][33];
  const int g = blockIdx.z, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const float* sg = src + (int64_t)g * R * C;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) tile[i][tx] = (r0 + i < R && c0 + tx < C) ? sg[(int64_t)(r0 + i) * C + c0 + tx] : 0.f;
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    float v;
    int64_t o;
    bool ok;
    if (transpose) {  // dst[g][c0 + i][r0 + tx]
      v = tile[tx][i], ok = c0 + i < C && r0 + tx < R, o = (int64_t)g * R * C + (int64_t)(c0 + i) * R + r0 + tx;
    } else {
      v = tile[i][tx], ok = r0 + i < R && c0 + tx < C, o = (int64_t)g * R * C + (int64_t)(r0 + i) * C + c0 + tx;
    }
    if (ok) {
      const uint32_t h = pack_bf16(v, 0.f) & 0xffffu;
      hi[o] = (uint16_t)h;
      lo[o] = (uint16_t)(pack_bf16(v - bf16_lo_f32(h), 0.f) & 0xffffu);
    }
  }
}

// K ranges of a split-K weight-gradient GEMM: `reps` consecutive row spaces of `total` rows, each cut into `ns` chunks:
// out[r * ns + i] = r * total + min(i * chunk, total), out[reps * ns] = reps * total
__global__ void splitk_offsets_kernel(int* __restrict__ out, int ns, int chunk, int total, int reps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > reps * ns) return;
  const int r = i / ns, c = i - r * ns;
  out[i] = i == reps * ns ? reps * total : r * total + (c * chunk < total ? c * chunk : total);
}

// get_load_balancing_loss (switch_moe.py:113-145) of both SwitchMoE layers from this forward's counters [2][E] each
__global__ void lb_loss_kernel(const float* __restrict__ usage, const float* __restrict__ imp, int E, float* __restrict__ out) {
  const int br = threadIdx.x;
  if (br >= 2) return;
  float tu = 0.f, ti = 0.f;
  for (int e = 0; e < E; ++e) tu += usage[br * E + e], ti += imp[br * E + e];
  tu = fmaxf(tu, 1e-8f), ti = fmaxf(ti, 1e-8f);
  float al = 0.f;
  for (int e = 0; e < E; ++e) al += (usage[br * E + e] / tu) * (imp[br * E + e] / ti);
  out[br] = E * (1.f - al);
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  float s = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += x[i] * x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

// Adam (torch.optim.Adam defaults: no weight decay, no amsgrad) with the gradient-norm clip of ddpm_trainer.py:239 folded in:
// g <- g * min(1, max_norm / (sqrt(*sumsq) + 1e-6)) like torch.nn.utils.clip_grad_norm_
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2, const float* __restrict__ sumsq, float max_norm) {
  float clip = 1.f;
  if (sumsq && max_norm > 0.f) clip = fminf(1.f, max_norm / (sqrtf(*sumsq) + 1e-6f));
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * clip;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi, v[i] = vi;
    p[i] -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
  }
}

inline int ew_grid(int64_t n) {
  const int64_t b = (n + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

struct TrainWork {
  float *hn, *xg, *pre, *hid, *y2, *dy, *sact, *ds, *dzz, *embp, *semb, *sc, *demb_out, *dembp, *dlogits, *dp, *cnt;
  float *top_val, *rowscale, *uimp;
  int *top_idx, *perm, *pos4, *hist, *goff, *cursor, *kso, *ksg;
  float *part_o, *part_g;  // split-K partials of dWo [NSK][D*D] and dWg [2*NSK][E*D]
  uint16_t* wp[8];         // bf16 hi / lo planes of W1, W2, W1^T, W2^T (packed per step when D and F are multiples of 32)
  int64_t bytes;
};
constexpr int NSK = 32;  // K chunks of the long-K weight gradients (K = B*S rows, outputs of only D x D / E x D)

struct Bump2 {
  uint8_t* base;
  int64_t off = 0;
  template <typename T>
  T* take(int64_t n) {
    T* p = base ? (T*)(base + off) : nullptr;
    off = (off + n * (int64_t)sizeof(T) + 255) & ~(int64_t)255;
    return p;
  }
};

TrainWork carve_train(int B, int S, int D, int F, int E, int Te, void* ws) {
  TrainWork w;
  Bump2 b{(uint8_t*)ws};
  const int64_t M = (int64_t)B * S;
  w.hn = b.take<float>(2 * M * D), w.xg = b.take<float>(4 * M * D), w.pre = b.take<float>(4 * M * F);
  w.hid = b.take<float>(4 * M * F), w.y2 = b.take<float>(4 * M * D), w.dy = b.take<float>(4 * M * D);
  w.sact = b.take<float>(M * D), w.ds = b.take<float>(M * D), w.dzz = b.take<float>(2 * M * D);
  w.embp = b.take<float>((int64_t)B * Te), w.semb = b.take<float>((int64_t)B * Te), w.sc = b.take<float>((int64_t)B * 2 * D);
  w.demb_out = b.take<float>((int64_t)B * 2 * D), w.dembp = b.take<float>((int64_t)B * Te);
  w.dlogits = b.take<float>(2 * M * E), w.dp = b.take<float>(4 * M), w.cnt = b.take<float>(4 * 16);
  w.top_val = b.take<float>(4 * M), w.rowscale = b.take<float>(4 * M), w.uimp = b.take<float>(1024 * 64);
  w.top_idx = b.take<int>(4 * M), w.perm = b.take<int>(4 * M), w.pos4 = b.take<int>(4 * M), w.hist = b.take<int>(1024 * 32);
  w.goff = b.take<int>(2 * E + 1), w.cursor = b.take<int>(2 * E);
  w.kso = b.take<int>(NSK + 1), w.ksg = b.take<int>(2 * NSK + 1);
  w.part_o = b.take<float>((int64_t)NSK * D * D), w.part_g = b.take<float>((int64_t)2 * NSK * E * D);
  for (int i = 0; i < 8; ++i) w.wp[i] = b.take<uint16_t>((int64_t)2 * E * F * D);
  w.bytes = b.off;
  return w;
}

bool shape_ok(int B, int S, int D, int F, int E, int Te, int De) {
  return B >= 1 && S >= 1 && D >= 4 && D <= 1024 && F >= 1 && E >= 2 && E <= 16 && Te >= 1 && De >= 1;
}

GemmArgs x3() { return gemm_defaults(3); }
inline bool planes_ok(int D, int F) { return (D % 32) == 0 && (F % 32) == 0; }
int pack_planes(const float* src, int G, int R, int C, bool transpose, uint16_t* hi, uint16_t* lo, hipStream_t s) {
  hipLaunchKernelGGL(pack_planes_kernel, dim3((C + 31) / 32, (R + 31) / 32, G), dim3(256), 0, s, src, R, C, transpose ? 1 : 0, hi, lo);
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

// p in [0, 1): threshold on the top 24 random bits; dropout needs the vectorised row widths
bool make_drop(float p, uint64_t seed, int D, Drop& d) {
  d.thr = 0, d.keep = 1.f, d.k0 = (uint32_t)seed, d.k1 = (uint32_t)(seed >> 32);
  if (p == 0.f) return true;
  if (!(p > 0.f && p < 1.f) || (D != 256 && D != 512 && D != 1024)) return false;
  d.thr = (uint32_t)(p * 16777216.0f);
  d.keep = 16777216.0f / (16777216.0f - (float)d.thr);  // exactly 1 / P(keep)
  return true;
}

#define ROWK(KERNEL, grid, ...)                                                              \
  do {                                                                                       \
    if (D == 512) {                                         \
      hipLaunchKernelGGL((KERNEL<8, true>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);       \
    } else if (D == 1024) {                                                                  \
      hipLaunchKernelGGL((KERNEL<16, true>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);      \
    } else if (D == 256) {                                                                   \
      hipLaunchKernelGGL((KERNEL<4, true>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);       \
    } else if (D <= 256) {                                                                   \
      hipLaunchKernelGGL((KERNEL<4, false>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);      \
    } else {                                                                                 \
      hipLaunchKernelGGL((KERNEL<16, false>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);     \
    }                                                                                        \
  } while (0)

// row-wise backward kernels that end in column-sum atomics: few, fat workgroups (every wave walks ~12 rows) so that the
// atomics per address stay in the hundreds
inline int acc_grid(int64_t M) {
  const int g = row_grid(M);
  return g > 256 ? 256 : g;
}

template <int NE, bool VEC>
int launch_gate_ln_one(int D, int E, int grid, hipStream_t s, const float* x, const float* dout, const float* dxg, const float* dp,
                       const int* pos4, const int* top_idx, const float* ln_w, const float* ln_b, const float* gate_w,
                       const float* gate_b, int64_t M, float* dlogits, float* dx, float* g_ln_w, float* g_ln_b) {
  const int smem = (2 * E * D + 4 * D) * 4;
  static DevInt attr;
  if (smem > 65536 && smem > attr) {
    if (hipFuncSetAttribute((const void*)gate_ln_bwd_kernel<NE, VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return MDM_ERR_LAUNCH;
    attr = smem;
  }
  hipLaunchKernelGGL((gate_ln_bwd_kernel<NE, VEC>), dim3(grid), dim3(256), smem, s, x, dout, dxg, dp, pos4, top_idx, ln_w, ln_b, gate_w,
                     gate_b, M, D, E, dlogits, dx, g_ln_w, g_ln_b);
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}
#define GLB_ARGS D, E, grid, s, x, dout, dxg, dp, pos4, top_idx, ln_w, ln_b, gate_w, gate_b, M, dlogits, dx, g_ln_w, g_ln_b
int launch_gate_ln_bwd(int D, int E, int grid, hipStream_t s, const float* x, const float* dout, const float* dxg, const float* dp,
                       const int* pos4, const int* top_idx, const float* ln_w, const float* ln_b, const float* gate_w,
                       const float* gate_b, int64_t M, float* dlogits, float* dx, float* g_ln_w, float* g_ln_b) {
  if (D == 512) return launch_gate_ln_one<8, true>(GLB_ARGS);
  if (D == 1024) return launch_gate_ln_one<16, true>(GLB_ARGS);
  if (D == 256) return launch_gate_ln_one<4, true>(GLB_ARGS);
  if (D <= 256) return launch_gate_ln_one<4, false>(GLB_ARGS);
  return launch_gate_ln_one<16, false>(GLB_ARGS);
}
#undef GLB_ARGS

int colsum(const float* X, int64_t ld, int64_t rows, int C, const int* goff, int ngroups, int64_t group_rows, float* out,
           hipStream_t s) {
  if (rows <= 0 || C <= 0) return MDM_OK;
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)((rows + 255) / 256)), dim3(256), 0, s, X, ld, rows,
                     C, goff, ngroups, group_rows, out);
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

int zero(float* p, int64_t n, hipStream_t s) {
  return hipMemsetAsync(p, 0, n * sizeof(float), s) == hipSuccess ? MDM_OK : MDM_ERR_LAUNCH;
}

}  // namespace
}  // namespace mdm

using namespace mdm;

extern "C" int64_t mdm_moe_train_workspace_bytes(int32_t B, int32_t S, int32_t D, int32_t F, int32_t E, int32_t Te) {
  if (!shape_ok(B, S, D, F, E, Te, 1)) return -1;
  return carve_train(B, S, D, F, E, Te, nullptr).bytes;
}

// forward in training mode: saves LN outputs, routing, gathered expert inputs, pre-activations, hidden and routed outputs
extern "C" int mdm_moe_ffn_train_forward(const MdmMoeTensors* P, int32_t D, int32_t F, int32_t E, int32_t Te, int32_t De,
                                         const float* eph_w, const float* eph_b, const float* x, const float* emb, int32_t B,
                                         int32_t S, float dropout_p, uint64_t seed, float* out, float* lb_loss,
                                         int32_t* route_out, void* ws, int64_t ws_bytes, void* stream) {
  if (!P || !x || !emb || !out || !ws || !shape_ok(B, S, D, F, E, Te, De)) return MDM_ERR_ARG;
  if (De != Te && (!eph_w || !eph_b)) return MDM_ERR_ARG;
  Drop dr;
  if (!make_drop(dropout_p, seed, D, dr)) return dropout_p >= 0.f && dropout_p < 1.f ? MDM_ERR_UNSUPPORTED : MDM_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const TrainWork w = carve_train(B, S, D, F, E, Te, ws);
  if (ws_bytes < w.bytes) return MDM_ERR_ARG;
  const int64_t M = (int64_t)B * S;
  MDM_TRY(zero(w.cnt, 64, s));
  // router (shared with the inference path): LN of both branches, gate, softmax, top-2, expert-sorted row lists
  MoeGateParams p = {};
  for (int b = 0; b < 2; ++b) {
    p.ln_w[b] = P->ln_w + b * D, p.ln_b[b] = P->ln_b + b * D;
    p.gate_w[b] = P->gate_w + (int64_t)b * E * D, p.gate_b[b] = P->gate_b + b * E;
    p.usage[b] = w.cnt + b * E, p.importance[b] = w.cnt + 32 + b * E;
  }
  p.hn = w.hn, p.hn_bf16 = 0, p.top_idx = w.top_idx, p.top_val = w.top_val, p.hist = w.hist, p.uimp = w.uimp;
  MDM_TRY(moe_route(x, M, D, E, p, w.goff, w.cursor, w.perm, w.rowscale, w.pos4, s));
  if (route_out && hipMemcpyAsync(route_out, w.top_idx, 4 * M * sizeof(int32_t), hipMemcpyDeviceToDevice, s) != hipSuccess)
    return MDM_ERR_LAUNCH;
  ROWK(gather_rows_kernel, row_grid(4 * M), (const float*)w.hn, (const int*)w.perm, w.xg, 4 * M, (int)D);
  const bool planes = planes_ok(D, F);
  if (planes) {  // this step's weights as bf16 hi / lo planes, both orientations (the backward reads the transposed ones)
    MDM_TRY(pack_planes(P->w1, 2 * E, F, D, false, w.wp[0], w.wp[1], s));
    MDM_TRY(pack_planes(P->w2, 2 * E, D, F, false, w.wp[2], w.wp[3], s));
    MDM_TRY(pack_planes(P->w1, 2 * E, F, D, true, w.wp[4], w.wp[5], s));
    MDM_TRY(pack_planes(P->w2, 2 * E, D, F, true, w.wp[6], w.wp[7], s));
  }
  {
    GemmArgs g = x3();  // pre = xg W1_e^T + b1_e                      (switch_moe.py:19-21)
    g.A = op_f32(w.xg, D), g.W = planes ? op_bf16(w.wp[0], w.wp[1], D) : op_f32(P->w1, D), g.W.bs1 = (int64_t)F * D;
    g.goff = w.goff, g.ngroups = 2 * E, g.M = (int)(4 * M), g.N = F, g.K = D;
    g.bias = P->b1, g.bias_bs = F, g.C = w.pre, g.ldc = F;
    MDM_TRY(gemm(g, s));
  }
  hipLaunchKernelGGL(ew_kernel<0>, dim3(ew_grid(4 * M * F)), dim3(256), 0, s, (const float*)w.pre, w.hid, 4 * M * F);
  {
    GemmArgs g = x3();  // y2 = p * (hid W2_e^T + b2_e)                (:24,108-109)
    g.A = op_f32(w.hid, F), g.W = planes ? op_bf16(w.wp[2], w.wp[3], F) : op_f32(P->w2, F), g.W.bs1 = (int64_t)D * F;
    g.goff = w.goff, g.ngroups = 2 * E, g.M = (int)(4 * M), g.N = D, g.K = F;
    g.bias = P->b2, g.bias_bs = D, g.rowscale = w.rowscale, g.C = w.y2, g.ldc = D;
    MDM_TRY(gemm(g, s));
  }
  const float* embp = emb;
  if (De != Te) {  // the per-call projection of stylization.py:22-24 (captured weights, not trained)
    GemmArgs g = x3();
    g.A = op_f32(emb, De), g.W = op_f32(eph_w, De), g.M = B, g.N = Te, g.K = De, g.bias = eph_b, g.C = w.embp, g.ldc = Te;
    MDM_TRY(gemm(g, s));
    embp = w.embp;
  }
  hipLaunchKernelGGL(ew_kernel<2>, dim3(ew_grid((int64_t)B * Te)), dim3(256), 0, s, embp, w.semb, (int64_t)B * Te);
  {
    GemmArgs g = x3();  // scale | shift = SiLU(emb) We^T + be          (stylization.py:10-13,26)
    g.A = op_f32(w.semb, Te), g.W = op_f32(P->st_emb_w, Te), g.M = B, g.N = 2 * D, g.K = Te;
    g.bias = P->st_emb_b, g.C = w.sc, g.ldc = 2 * D;
    MDM_TRY(gemm(g, s));
  }
  if (dr.thr) {
    ROWK(style_fwd_train_kernel, row_grid(M), (const float*)w.y2, (const int*)w.pos4, (const float*)w.sc, (const float*)P->st_norm_w,
         (const float*)P->st_norm_b, M, (int)D, (int)S, dr, w.sact);
  } else {
    MDM_TRY(style_in(w.y2, M, D, S, nullptr, nullptr, P->st_norm_w, P->st_norm_b, w.sc, w.pos4, 0, w.sact, 0, s));
  }
  {
    GemmArgs g = x3();  // out = x + SiLU(...) Wo^T + bo               (stylization.py:29, multi_branch.py:60)
    g.A = op_f32(w.sact, D), g.W = op_f32(P->st_out_w, D), g.M = (int)M, g.N = D, g.K = D;
    g.bias = P->st_out_b, g.R1 = x, g.ldr1 = D, g.C = out, g.ldc = D;
    MDM_TRY(gemm(g, s));
  }
  if (lb_loss) hipLaunchKernelGGL(lb_loss_kernel, dim3(1), dim3(64), 0, s, (const float*)w.cnt, (const float*)(w.cnt + 32), (int)E, lb_loss);
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

// backward: consumes the workspace of the matching forward.  G = every gradient tensor (overwritten); dx, demb optional
extern "C" int mdm_moe_ffn_train_backward(const MdmMoeTensors* P, int32_t D, int32_t F, int32_t E, int32_t Te, int32_t De,
                                          const float* eph_w, const float* x, const float* emb, int32_t B, int32_t S,
                                          float dropout_p, uint64_t seed, const float* dout, float* dx, float* demb,
                                          const MdmMoeTensors* G, void* ws, int64_t ws_bytes, void* stream) {
  if (!P || !G || !x || !emb || !dout || !dx || !ws || !shape_ok(B, S, D, F, E, Te, De)) return MDM_ERR_ARG;
  if (De != Te && !eph_w) return MDM_ERR_ARG;
  Drop dr;
  if (!make_drop(dropout_p, seed, D, dr)) return dropout_p >= 0.f && dropout_p < 1.f ? MDM_ERR_UNSUPPORTED : MDM_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const TrainWork w = carve_train(B, S, D, F, E, Te, ws);
  if (ws_bytes < w.bytes) return MDM_ERR_ARG;
  const int64_t M = (int64_t)B * S;
  MDM_TRY(zero(G->st_norm_w, D, s));
  MDM_TRY(zero(G->st_norm_b, D, s));
  MDM_TRY(zero(G->ln_w, 2 * D, s));
  MDM_TRY(zero(G->ln_b, 2 * D, s));
  MDM_TRY(zero(G->b1, (int64_t)2 * E * F, s));
  MDM_TRY(zero(G->b2, (int64_t)2 * E * D, s));
  MDM_TRY(zero(G->gate_b, 2 * E, s));
  MDM_TRY(zero(G->st_out_b, D, s));
  MDM_TRY(zero(G->st_emb_b, 2 * D, s));
  MDM_TRY(zero(w.demb_out, (int64_t)B * 2 * D, s));
  // ---- stylization block --------------------------------------------------------------------------------------------------
  {
    GemmArgs g = x3();  // ds = dout Wo
    g.A = op_f32(dout, D), g.W = op_f32_kstride(P->st_out_w, D), g.M = (int)M, g.N = D, g.K = D, g.C = w.ds, g.ldc = D;
    MDM_TRY(gemm(g, s));
  }
  const int chunk = (int)((M + NSK - 1) / NSK);
  {
    // dWo = dout^T SiLU(z): a D x D output reduced over K = M rows would run on 16 workgroups; split K into NSK chunks (the
    // per-batch K ranges of the weight-gradient mode), partials summed by the column-sum kernel
    hipLaunchKernelGGL(splitk_offsets_kernel, dim3(1), dim3(128), 0, s, w.kso, NSK, chunk, (int)M, 1);
    GemmArgs g = x3();
    g.A = op_f32_kstride(dout, D), g.W = op_f32_kstride(w.sact, D), g.M = D, g.N = D, g.K = (int)M;
    g.batch = NSK, g.kgoff = w.kso, g.C = w.part_o, g.ldc = D, g.c_bs1 = (int64_t)D * D;
    MDM_TRY(gemm(g, s));
    MDM_TRY(zero(G->st_out_w, (int64_t)D * D, s));
    MDM_TRY(colsum(w.part_o, (int64_t)D * D, NSK, D * D, nullptr, 0, 0, G->st_out_w, s));
  }
  MDM_TRY(colsum(dout, D, M, D, nullptr, 0, 0, G->st_out_b, s));
  ROWK(style_bwd_kernel, acc_grid(M), (const float*)w.y2, (const int*)w.pos4, (const float*)w.ds, (const float*)w.sc,
       (const float*)P->st_norm_w, (const float*)P->st_norm_b, M, (int)D, (int)S, dr, w.dzz, w.ds, G->st_norm_w, G->st_norm_b);
  MDM_TRY(colsum(w.dzz, 2 * D, M, 2 * D, nullptr, 0, S, w.demb_out, s));  // per-sample d(scale | shift)
  {
    GemmArgs g = x3();  // dWe = d(scale|shift)^T SiLU(emb)
    g.A = op_f32_kstride(w.demb_out, 2 * D), g.W = op_f32_kstride(w.semb, Te), g.M = 2 * D, g.N = Te, g.K = B;
    g.C = G->st_emb_w, g.ldc = Te;
    MDM_TRY(gemm(g, s));
  }
  MDM_TRY(colsum(w.demb_out, 2 * D, B, 2 * D, nullptr, 0, 0, G->st_emb_b, s));
  if (demb) {
    float* dst = De != Te ? w.dembp : demb;
    GemmArgs g = x3();  // d SiLU(emb) = d(scale|shift) We, then through the SiLU
    g.A = op_f32(w.demb_out, 2 * D), g.W = op_f32_kstride(P->st_emb_w, Te), g.M = B, g.N = Te, g.K = 2 * D, g.C = dst, g.ldc = Te;
    MDM_TRY(gemm(g, s));
    const float* embp = De != Te ? w.embp : emb;
    hipLaunchKernelGGL(ew_kernel<3>, dim3(ew_grid((int64_t)B * Te)), dim3(256), 0, s, embp, dst, (int64_t)B * Te);
    if (De != Te) {
      GemmArgs g2 = x3();  // back through the captured projection
      g2.A = op_f32(w.dembp, Te), g2.W = op_f32_kstride(eph_w, De), g2.M = B, g2.N = De, g2.K = Te, g2.C = demb, g2.ldc = De;
      MDM_TRY(gemm(g2, s));
    }
  }
  // ---- experts ------------------------------------------------------------------------------------------------------------
  ROWK(routed_bwd_kernel, row_grid(4 * M), (const float*)w.ds, (const float*)w.y2, (const int*)w.perm, (const float*)w.rowscale, M,
       4 * M, (int)D, dr, w.dy, w.dp);
  {
    GemmArgs g = x3();  // dW2[g] = dy_g^T hid_g over the routed rows of group g
    g.A = op_f32_kstride(w.dy, D), g.W = op_f32_kstride(w.hid, F), g.M = D, g.N = F, g.K = (int)(4 * M);
    g.batch = 2 * E, g.kgoff = w.goff, g.C = G->w2, g.ldc = F, g.c_bs1 = (int64_t)D * F;
    MDM_TRY(gemm(g, s));
  }
  MDM_TRY(colsum(w.dy, D, 4 * M, D, w.goff, 2 * E, 0, G->b2, s));
  {
    GemmArgs g = x3();  // d hid = dy W2_e  (over the hidden buffer, dead after dW2)
    // W2_e^T rows (f, k = d): the transposed planes packed by the forward, or the fp32 weight read k-strided
    g.A = op_f32(w.dy, D), g.W = planes_ok(D, F) ? op_bf16(w.wp[6], w.wp[7], D) : op_f32_kstride(P->w2, F), g.W.bs1 = (int64_t)D * F;
    g.goff = w.goff, g.ngroups = 2 * E, g.M = (int)(4 * M), g.N = F, g.K = D, g.C = w.hid, g.ldc = F;
    MDM_TRY(gemm(g, s));
  }
  hipLaunchKernelGGL(ew_kernel<1>, dim3(ew_grid(4 * M * F)), dim3(256), 0, s, (const float*)w.pre, w.hid, 4 * M * F);  // d pre
  {
    GemmArgs g = x3();  // dW1[g] = dpre_g^T xg_g
    g.A = op_f32_kstride(w.hid, F), g.W = op_f32_kstride(w.xg, D), g.M = F, g.N = D, g.K = (int)(4 * M);
    g.batch = 2 * E, g.kgoff = w.goff, g.C = G->w1, g.ldc = D, g.c_bs1 = (int64_t)F * D;
    MDM_TRY(gemm(g, s));
  }
  MDM_TRY(colsum(w.hid, F, 4 * M, F, w.goff, 2 * E, 0, G->b1, s));
  {
    GemmArgs g = x3();  // d xg = dpre W1_e  (over the dy buffer, dead by now)
    g.A = op_f32(w.hid, F), g.W = planes_ok(D, F) ? op_bf16(w.wp[4], w.wp[5], F) : op_f32_kstride(P->w1, D), g.W.bs1 = (int64_t)F * D;
    g.goff = w.goff, g.ngroups = 2 * E, g.M = (int)(4 * M), g.N = D, g.K = F, g.C = w.dy, g.ldc = D;
    MDM_TRY(gemm(g, s));
  }
  // ---- gate + branch LayerNorms ---------------------------------------------------------------------------------------------
  MDM_TRY(launch_gate_ln_bwd(D, E, acc_grid(M), s, x, dout, (const float*)w.dy, (const float*)w.dp, (const int*)w.pos4, (const int*)w.top_idx,
                             (const float*)P->ln_w, (const float*)P->ln_b, (const float*)P->gate_w, (const float*)P->gate_b, M, w.dlogits,
                             dx, G->ln_w, G->ln_b));
  {
    // dWg[b] = dlogits_b^T LN_b(x): E x D outputs over K = M rows -> split K; both branches are one [2M]-row space
    hipLaunchKernelGGL(splitk_offsets_kernel, dim3(1), dim3(128), 0, s, w.ksg, NSK, chunk, (int)M, 2);
    GemmArgs g = x3();
    g.A = op_f32_kstride(w.dlogits, E), g.W = op_f32_kstride(w.hn, D);
    g.M = E, g.N = D, g.K = (int)(2 * M), g.batch = 2 * NSK, g.kgoff = w.ksg, g.C = w.part_g, g.ldc = D, g.c_bs1 = (int64_t)E * D;
    MDM_TRY(gemm(g, s));
    MDM_TRY(zero(G->gate_w, (int64_t)2 * E * D, s));
    MDM_TRY(colsum(w.part_g, (int64_t)E * D, 2 * NSK, E * D, nullptr, 0, NSK, G->gate_w, s));
  }
  MDM_TRY(colsum(w.dlogits, E, 2 * M, E, nullptr, 0, M, G->gate_b, s));
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

// sum of squares of a flat gradient buffer into *out (zeroed here): the norm of ddpm_trainer.py:239's clip
extern "C" int mdm_sumsq(const float* x, int64_t n, float* out, void* stream) {
  if (!x || !out || n < 0) return MDM_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  MDM_TRY(zero(out, 1, s));
  if (n > 0) hipLaunchKernelGGL(sumsq_kernel, dim3(ew_grid(n) > 1024 ? 1024 : ew_grid(n)), dim3(256), 0, s, x, n, out);
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

// one Adam step on flat fp32 buffers; step >= 1; sumsq / max_norm: optional global-norm clip read from device memory
extern "C" int mdm_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                             int32_t step, const float* sumsq, float max_norm, void* stream) {
  if (!p || !g || !m || !v || n < 0 || step < 1) return MDM_ERR_ARG;
  if (n == 0) return MDM_OK;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps, bc1, bc2,
                     sumsq, max_norm);
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}
