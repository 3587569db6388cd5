// Fused two-layer MLP for gfx950 (throughput mode):   Y = epilogue( GELU(X W1^T + b1) W2^T + b2 )
// used for the expert MLPs (grouped by expert, gathered rows, gate-probability row scale: switch_moe.py:19-25,97-109);
// the descriptor also covers the dense Linear-GELU-Linear pairs (fast_attention.py:121-126,293-299).
//
// One workgroup = 8 waves owns 128 rows and ALL Dout = 512 output columns:
//   * the output accumulators (128 x 512 fp32) live in registers for the whole kernel: waves 2 (M) x 4 (N), each
//     64 x 128 = 4 x 8 MFMA tiles = 128 VGPRs;
//   * the hidden layer is produced in chunks of 256 units: phase 1 (K = Din, 64-wide K tiles: X tile + W1 tile by LDS-DMA)
//     -> bias + GELU -> bf16 chunk [128 x 256] in LDS -> phase 2 (K = 256 in 32-wide slabs of W2 [512 x 32]) accumulates
//     into Y.  The hidden activations (2 x 4M x F bf16 per MoE block = 206 MB at the bench shape) never touch HBM and there
//     is one prologue / epilogue per 0.27 GFLOP instead of one per 4 MFLOP tile.
//   * LDS (160 KiB exactly): hidden chunk 64 KiB + 2 ring stages of 48 KiB (phase 1: X 16 KiB + W1 32 KiB; phase 2: W2
//     slab 32 KiB); every image uses the conflict-free XOR / rotate swizzles of gemm2.hip, applied on the DMA source.
//   * epilogue staged through LDS in four 32-row slabs and written as full 2-KiB rows with coalesced residual reads.
// Measured (MI355X, 16 experts x 3136 rows, F = 1024): 199 us = 528 TFLOP/s vs 210 us for the two-GEMM chain; one tile
// takes 96 us with every CU busy: MFMA-only 27 us, LDS-DMA-only 66 us, compute-only (MFMA + fragment reads + GELU) 71 us.
// With few row tiles (dense M = 12544 -> 98 workgroups on 256 CUs) the two-GEMM chain is faster; callers choose.
#include "gemm.h"
#include "kernels.h"

namespace mdm {
namespace {

constexpr int BM = 128, DOUT = 512, FC = 256, NT = 512;
constexpr int HC_B = BM * FC * 2;             // 65536: hidden chunk, bf16 [128][256]
constexpr int XT_B = BM * 128, W1T_B = FC * 128;  // phase-1 tiles: 128-B rows (64 k): X 16 KiB + W1 32 KiB
constexpr int STAGE_B = XT_B + W1T_B;         // 49152; a phase-2 stage (W2 slab [512][32 k], 32 KiB) uses part of it
constexpr int NST = 2;

__device__ __forceinline__ void glds16(const void* g, uint8_t* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <typename HT>
__global__ __launch_bounds__(NT, 2) void fused_mlp_kernel(const MdmMlpDesc g) {
  typedef typename HT::frag_t frag_t;
  extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
  uint8_t* hc = smem;
  uint8_t* ring = smem + HC_B;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3, frow = lane & 15, fq = lane >> 4;

  int row0, row_end, grp = 0;
  {
    const int mt = xcd_remap(blockIdx.x, gridDim.x);  // contiguous tile ranges per XCD: one XCD L2 serves ~2 experts
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
  }
  const uint16_t* W1 = g.w1 + (int64_t)grp * g.w1_gs;
  const uint16_t* W2 = g.w2 + (int64_t)grp * g.w2_gs;
  const float* b1 = g.b1 ? g.b1 + (int64_t)grp * g.b1_gs : nullptr;
  const float* b2 = g.b2 ? g.b2 + (int64_t)grp * g.b2_gs : nullptr;

  // ---- LDS-DMA sources ------------------------------------------------------------------------------------------
  // phase 1, 128-B rows, pieces of 8 rows: lane -> row (lane >> 3) of the piece, slot (lane & 7) holds chunk slot ^ (row & 7)
  const int sub8 = lane >> 3, c8 = ((lane & 7) ^ sub8) * 8;
  const uint16_t* px[2];  // X tile: 16 pieces, 2 per wave
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int r = row0 + (wid * 2 + i) * 8 + sub8;
    r = r < row_end ? r : row_end - 1;
    const int64_t src = g.gather ? (int64_t)g.gather[r] : (int64_t)r;
    px[i] = g.X + src * g.ldx + c8;
  }
  // W1 tile: 32 pieces, 4 per wave: hidden unit f = chunk * 256 + (wid * 4 + i) * 8 + sub8
  const uint16_t* pw1 = W1 + (int64_t)(wid * 32 + sub8) * g.ldw1 + c8;
  // phase 2, 64-B rows (32 k), pieces of 16 rows: slot (lane & 3) holds chunk (slot - 2 * (row >> 2)) & 3
  const int sub16 = lane >> 2;
  const int c4 = (((lane & 3) - 2 * (sub16 >> 2)) & 3) * 8;
  const uint16_t* pw2 = W2 + (int64_t)(wid * 64 + sub16) * g.ldw2 + c4;  // slab: 32 pieces, 4 per wave
  const int nkt = g.Din / 64, nchunk = g.F / FC;

  auto stage_p1 = [&](int chunk, int kt, int buf) {
    uint8_t* s = ring + buf * STAGE_B;
    const int k0 = kt * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(px[i] + k0, s + (wid * 2 + i) * 1024);
    const uint16_t* w = pw1 + (int64_t)chunk * FC * g.ldw1 + k0;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(w + (int64_t)(8 * i) * g.ldw1, s + XT_B + (wid * 4 + i) * 1024);
  };
  auto stage_p2 = [&](int chunk, int sl, int buf) {
    uint8_t* s = ring + buf * STAGE_B;
    const uint16_t* w = pw2 + chunk * FC + sl * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(w + (int64_t)(16 * i) * g.ldw2, s + (wid * 4 + i) * 1024);
  };

  // The accumulators START from the biases (b2 here, b1 at the top of every hidden chunk): those loads then complete under the
  // wait the first K step has anyway.  Loaded where they are added -- after the MFMAs -- every one of them is a separate round
  // trip behind the LDS-DMA queue (vmcnt is one in-order counter): 16 per tile in phase 1 and 64 in the epilogue.
  f32x4 y[4][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const f32x4 bb = b2 ? *(const f32x4*)(b2 + wn * 128 + j * 16 + fq * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i][j] = bb;
  }

  // One DMA stream through the 2-stage ring: per chunk, nkt phase-1 steps then 8 phase-2 steps; step st uses stage st & 1
  // and prefetches step st + 1 right after the barrier that retires step st - 1.
  int st = 0;
  stage_p1(0, 0, 0);
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    {
      // ---- phase 1: h[128 x 256 chunk] = X (128 x Din) . W1 chunk (256 x Din)^T ; wave: rows wm*64.., units wn*64..
      f32x4 h[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 bb = b1 ? *(const f32x4*)(b1 + chunk * FC + wn * 64 + j * 16 + fq * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i][j] = bb;
      }
      for (int kt = 0; kt < nkt; ++kt, ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nkt) {
          stage_p1(chunk, kt + 1, (st + 1) & 1);
        } else {
          stage_p2(chunk, 0, (st + 1) & 1);
        }
        const uint8_t* s = ring + (st & 1) * STAGE_B;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          frag_t a[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int ra = wm * 64 + i * 16 + frow;
            a[i] = *(const frag_t*)(s + ra * 128 + (((ks * 4 + fq) ^ (ra & 7)) << 4));
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int rb = wn * 64 + j * 16 + frow;
            const frag_t b = *(const frag_t*)(s + XT_B + rb * 128 + (((ks * 4 + fq) ^ (rb & 7)) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i) h[i][j] = HT::mfma16(b, a[i], h[i][j]);
          }
        }
      }
      // exact GELU (bias already in h) -> 16-bit hidden chunk in LDS.  Lane: row m = wm*64 + 16 i + frow, units f = wn*64 + 16 j + 4 fq + r.
      // (the previous chunk's phase-2 readers of hc are behind at least one barrier of the phase-1 loop)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int f = wn * 64 + j * 16 + fq * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = wm * 64 + i * 16 + frow;
          const f32x2 g01 = gelu_erf2((f32x2){h[i][j][0], h[i][j][1]});
          const f32x2 g23 = gelu_erf2((f32x2){h[i][j][2], h[i][j][3]});
          const float v0 = g01[0], v1 = g01[1], v2 = g23[0], v3 = g23[1];
          // hidden chunk image: 512-B rows, 16-B chunk index (f >> 3) swizzled by (m & 15); 8-B half (f >> 2) & 1
          *(uint2*)(hc + m * 512 + (((f >> 3) ^ (m & 15)) << 4) + ((f >> 2) & 1) * 8) =
              make_uint2(HT::pack(v0, v1), HT::pack(v2, v3));
        }
      }
    }
    // ---- phase 2: y[128 x 512] += hidden[128 x 256] . W2[:, chunk]^T in 32-wide slabs ; wave: rows wm*64.., cols wn*128..
    for (int sl = 0; sl < 8; ++sl, ++st) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // also publishes the hidden chunk written above (sl == 0)
      if (sl + 1 < 8) {
        stage_p2(chunk, sl + 1, (st + 1) & 1);
      } else if (chunk + 1 < nchunk) {
        stage_p1(chunk + 1, 0, (st + 1) & 1);
      }
      const uint8_t* s = ring + (st & 1) * STAGE_B;
      frag_t a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = wm * 64 + i * 16 + frow;
        a[i] = *(const frag_t*)(hc + m * 512 + (((sl * 4 + fq) ^ (m & 15)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int n = wn * 128 + j * 16 + frow;
        const frag_t b = *(const frag_t*)(s + n * 64 + (((fq + 2 * (n >> 2)) & 3) << 4));
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i][j] = HT::mfma16(b, a[i], y[i][j]);
      }
    }
  }

  // ---- epilogue: four 32-row slabs staged as [32][512] fp32 in the ring (64 KiB), full-row writes ----------------
  float* stg = (float*)ring;
  const int cl = tid & 127, n = 4 * cl;       // 128 threads cover one 512-column row as float4
  const float* __restrict__ R1 = g.R1;
  const float* __restrict__ R2 = g.R2;
#pragma unroll 1
  for (int p = 0; p < 4; ++p) {
    // this pass's two row scales, requested before the barrier (rows clamped, not predicated: no branch around the loads)
    float rs2[2] = {1.f, 1.f};
    if (g.rowscale) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int m = row0 + 32 * p + ii * 16 + frow;
        rs2[ii] = g.rowscale[m < row_end ? m : row_end - 1];
      }
    }
    __syncthreads();
    if (wm == (p >> 1)) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int ml = ii * 16 + frow;
        const float rs = rs2[ii];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int nn = wn * 128 + j * 16 + fq * 4;
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float a0 = y[ii][j][r], a1 = y[2 + ii][j][r];
            v[r] = ((p & 1) ? a1 : a0) * rs;
          }
          const int chunk = nn >> 2;  // 0..127
          *(f32x4*)(stg + ml * 512 + ((chunk ^ (ml & 31)) << 2)) = v;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ml = (tid >> 7) + 4 * k, m = row0 + 32 * p + ml;
      if (m >= row_end) continue;
      f32x4 v = *(const f32x4*)(stg + ml * 512 + ((cl ^ (ml & 31)) << 2));
      if (R1) {
        const f32x4 q = *(const f32x4*)(R1 + (int64_t)m * g.ldr1 + n);
        v[0] += g.r1_scale * q[0], v[1] += g.r1_scale * q[1], v[2] += g.r1_scale * q[2], v[3] += g.r1_scale * q[3];
      }
      if (R2) {
        const f32x4 q = *(const f32x4*)(R2 + (int64_t)m * g.ldr2 + n);
        v[0] += q[0], v[1] += q[1], v[2] += q[2], v[3] += q[3];
      }
      if (g.C) *(f32x4*)(g.C + (int64_t)m * g.ldc + n) = v;
      if (g.C16) *(uint2*)(g.C16 + (int64_t)m * g.ldc + n) = make_uint2(HT::pack(v[0], v[1]), HT::pack(v[2], v[3]));
    }
  }
}

}  // namespace

bool fused_mlp_supported(const MdmMlpDesc& a) {
  if (fused_mlp_stream_supported(a)) return true;
  if (!a.w1 || !a.w2) return false;
  if (a.Dout != DOUT || a.Din < 64 || (a.Din % 64) || a.F < FC || (a.F % FC) || a.M < 1) return false;
  if ((a.ldx % 8) || (a.ldw1 % 8) || (a.ldw2 % 8) || (a.w1_gs % 8) || (a.w2_gs % 8)) return false;
  if (((((uintptr_t)a.X) | ((uintptr_t)a.w1) | ((uintptr_t)a.w2)) & 15)) return false;
  if ((a.ldc & 3) || (a.R1 && (a.ldr1 & 3)) || (a.R2 && (a.ldr2 & 3))) return false;
  if ((a.b1 && ((((uintptr_t)a.b1) & 15) || (a.b1_gs & 3))) || (a.b2 && ((((uintptr_t)a.b2) & 15) || (a.b2_gs & 3)))) return false;
  return true;
}

extern int g_bf16_variant;

int fused_mlp(const MdmMlpDesc& a, hipStream_t stream) {
  // streamed-weight kernel (csrc/mlp_stream.hip) whenever the caller packed a weight stream and the shape fits; knob 34
  // keeps this kernel for A/B runs
  if (g_bf16_variant != 34 && fused_mlp_stream_supported(a)) return fused_mlp_stream(a, stream);
  if (!a.X || !a.w1 || !a.w2 || (!a.C && !a.C16)) return MDM_ERR_ARG;
  MdmMlpDesc plain = a;
  plain.wstream = nullptr;
  if (!fused_mlp_supported(plain)) return MDM_ERR_UNSUPPORTED;
  constexpr int smem = HC_B + NST * STAGE_B;  // 163840
  static DevOnce attr;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)fused_mlp_kernel<HB>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess ||
        hipFuncSetAttribute((const void*)fused_mlp_kernel<HF>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return MDM_ERR_LAUNCH;
    attr = true;
  }
  const int tiles = (a.M + BM - 1) / BM + (a.goff ? a.ngroups : 0);
  if (a.h16 == MDM_H16_F16) {
    hipLaunchKernelGGL(fused_mlp_kernel<HF>, dim3(tiles), dim3(NT), smem, stream, a);
  } else {
    hipLaunchKernelGGL(fused_mlp_kernel<HB>, dim3(tiles), dim3(NT), smem, stream, a);
  }
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

}  // namespace mdm
