// Philox4x32-10 (Salmon et al., SC'11), the counter-based generator behind the sampler's noise (noise.hip) and the training
// step's dropout masks (moe_train.hip).  oracle/philox_ref.py restates it in numpy.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace mdm {

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
  c[0] = n0, c[1] = n1, c[2] = n2, c[3] = n3;
}

// 4 x 32 random bits for counter (c0, c1, c2, c3) under key (k0, k1): the standard 10-round Philox4x32
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
  }
}

}  // namespace mdm
