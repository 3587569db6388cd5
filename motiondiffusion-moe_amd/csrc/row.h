// One wave (64 lanes) owns one row of D fp32 elements: the register image, LayerNorm and the width dispatch shared by the
// row-wise kernels (rowwise.hip) and the MoE-block training kernels (moe_train.hip).
#pragma once
#include "mdm_common.h"
#include "mdm_hip.h"

namespace mdm {
namespace {

constexpr int WPB = 4;  // waves (rows) per 256-thread block

// ---- a row of D floats spread over a wave ------------------------------------------------------
// VEC: D == 4*64*NV4 exactly, lane holds NV4 float4 (coalesced 16 B/lane); else generic strided scalars.
template <int NE, bool VEC>
struct Row {
  float e[NE];
  __device__ __forceinline__ void load(const float* __restrict__ p, int D, int lane) {
    if constexpr (VEC) {
#pragma unroll
      for (int c = 0; c < NE / 4; ++c) {
        f32x4 v = *(const f32x4*)(p + 4 * (lane + 64 * c));
        e[4 * c + 0] = v[0], e[4 * c + 1] = v[1], e[4 * c + 2] = v[2], e[4 * c + 3] = v[3];
      }
    } else {
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        const int i = lane + 64 * j;
        e[j] = i < D ? p[i] : 0.f;
      }
    }
  }
  // 16-bit rows.  FMT (1 = bf16, 2 = fp16) is a template argument on purpose: with the format as a run-time value every
  // converted element carries a uniform branch, and hipcc then waits for each 8-byte load before it issues the next one
  // (the loads of a row, and of the rows a kernel gathers, must all be in flight together).
  template <int FMT>
  __device__ __forceinline__ void load_h16(const uint16_t* __restrict__ p, int D, int lane) {
    if constexpr (VEC) {
      uint2 u[NE / 4];
#pragma unroll
      for (int c = 0; c < NE / 4; ++c) u[c] = *(const uint2*)(p + 4 * (lane + 64 * c));
#pragma unroll
      for (int c = 0; c < NE / 4; ++c) {
        e[4 * c + 0] = h16_lo_f32(FMT, u[c].x), e[4 * c + 1] = h16_hi_f32(FMT, u[c].x);
        e[4 * c + 2] = h16_lo_f32(FMT, u[c].y), e[4 * c + 3] = h16_hi_f32(FMT, u[c].y);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        const int i = lane + 64 * j;
        e[j] = i < D ? h16_lo_f32(FMT, (uint32_t)p[i]) : 0.f;
      }
    }
  }
  __device__ __forceinline__ void load_bf16(const uint16_t* __restrict__ p, int D, int lane, int fmt = 1) {
    if (fmt == 2) {
      load_h16<2>(p, D, lane);
    } else {
      load_h16<1>(p, D, lane);
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ p, int D, int lane) const {
    if constexpr (VEC) {
#pragma unroll
      for (int c = 0; c < NE / 4; ++c) {
        f32x4 v = {e[4 * c + 0], e[4 * c + 1], e[4 * c + 2], e[4 * c + 3]};
        *(f32x4*)(p + 4 * (lane + 64 * c)) = v;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        const int i = lane + 64 * j;
        if (i < D) p[i] = e[j];
      }
    }
  }
  // 16-bit (round-to-nearest-even) copy of the row, for tensors whose only consumer is a 16-bit MFMA GEMM
  template <int FMT>
  __device__ __forceinline__ void store_h16(uint16_t* __restrict__ p, int D, int lane) const {
    if constexpr (VEC) {
#pragma unroll
      for (int c = 0; c < NE / 4; ++c)
        *(uint2*)(p + 4 * (lane + 64 * c)) = make_uint2(pack_h16(FMT, e[4 * c + 0], e[4 * c + 1]), pack_h16(FMT, e[4 * c + 2], e[4 * c + 3]));
    } else {
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        const int i = lane + 64 * j;
        if (i < D) p[i] = (uint16_t)(pack_h16(FMT, e[j], 0.f) & 0xffff);
      }
    }
  }
  __device__ __forceinline__ void store_bf16(uint16_t* __restrict__ p, int D, int lane, int fmt = 1) const {
    if (fmt == 2) {
      store_h16<2>(p, D, lane);
    } else {
      store_h16<1>(p, D, lane);
    }
  }
  // MDM_OP_X2_ROW row (include/mdm_hip.h): bf16 hi / lo halves per block of 32 columns, the pre-split operand of the fp32-grade GEMM
  __device__ __forceinline__ void store_x2(uint16_t* __restrict__ p, int D, int lane) const {
    static_assert(VEC, "x2 rows exist for the vector widths only");
#pragma unroll
    for (int c = 0; c < NE / 4; ++c) store_x2_4p(p, 4 * (lane + 64 * c), e[4 * c + 0], e[4 * c + 1], e[4 * c + 2], e[4 * c + 3]);
  }
  // mode 0 fp32, 1 the launch's 16-bit format, 2 x2 rows (2 D 16-bit elements per row)
  template <int FMT>
  __device__ __forceinline__ void store_mode(void* __restrict__ p, int64_t row, int D, int lane, int mode) const {
    if constexpr (VEC) {
      if (mode == 2) {
        store_x2((uint16_t*)p + row * 2 * D, D, lane);
        return;
      }
    }
    store_to<FMT>(p, row, D, lane, mode == 1);
  }
  // fp32 or 16-bit destination, format fixed at compile time
  template <int FMT>
  __device__ __forceinline__ void store_to(void* __restrict__ p, int64_t row, int D, int lane, bool h16) const {
    if (h16) {
      store_h16<FMT>((uint16_t*)p + row * D, D, lane);
    } else {
      store((float*)p + row * D, D, lane);
    }
  }
  __device__ __forceinline__ void store_as(void* __restrict__ p, int64_t row, int D, int lane, int bf16) const {
    if (bf16) {  // format code: 1 = bf16, 2 = fp16
      store_bf16((uint16_t*)p + row * D, D, lane, bf16);
    } else {
      store((float*)p + row * D, D, lane);
    }
  }
  __device__ __forceinline__ float sum() const {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NE; ++j) s += e[j];
    return wave_sum(s);
  }
  // LayerNorm in place (eps 1e-5, biased variance, two-pass like ATen); padded lanes stay 0
  // (weight / bias rows already in registers: the row kernels request them before the data rows arrive)
  __device__ __forceinline__ void layernorm(const Row<NE, VEC>& ww, const Row<NE, VEC>& bb, int D, int lane) {
    const float mean = sum() / D;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const bool in = VEC || (lane + 64 * j < D);
      const float d = in ? e[j] - mean : 0.f;
      s += d * d;
    }
    const float rstd = rsqrtf(wave_sum(s) / D + 1e-5f);
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const bool in = VEC || (lane + 64 * j < D);
      e[j] = in ? (e[j] - mean) * rstd * ww.e[j] + bb.e[j] : 0.f;
    }
  }
  __device__ __forceinline__ void layernorm(const float* __restrict__ w, const float* __restrict__ b, int D, int lane) {
    Row<NE, VEC> ww, bb;
    ww.load(w, D, lane);
    bb.load(b, D, lane);
    layernorm(ww, bb, D, lane);
  }
};

#define MDM_ROW_DISPATCH(D, CALL)                      \
  do {                                                 \
    if ((D) == 512) {                                  \
      CALL(8, true);                                   \
    } else if ((D) == 1024) {                          \
      CALL(16, true);                                  \
    } else if ((D) == 256) {                           \
      CALL(4, true);                                   \
    } else if ((D) <= 256) {                           \
      CALL(4, false);                                  \
    } else if ((D) <= 1024) {                          \
      CALL(16, false);                                 \
    } else {                                           \
      return MDM_ERR_UNSUPPORTED;                      \
    }                                                  \
  } while (0)

inline int row_grid(int64_t M) {
  int64_t g = (M + WPB - 1) / WPB;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace
}  // namespace mdm
