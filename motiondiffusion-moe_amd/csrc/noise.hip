// Counter-based gaussian noise for the sampler (Philox4x32-10 + Box-Muller), keyed on
//   (seed, GLOBAL sample index, stream id, element index)
// so that x_T and every step's noise are identical however the batch is sharded over GPUs, with O(1) host memory: the
// reference draws th.randn(*shape) / th.randn_like(x) from the process-global generator (gaussian_diffusion.py:1119,1094),
// which ties the sample to the batch composition; SURVEY.md section 8(e) asks for world-size-invariant noise instead.
// The stream id is the timestep t of the step that consumes the noise (read from device memory, so one captured hipGraph
// serves the whole loop) or MDM_NOISE_STREAM_XT for the initial x_T.
// oracle/philox_ref.py restates the generator in numpy; tests compare bit patterns of the uniforms and the normals to 1e-6.
#include "kernels.h"
#include "philox.h"

namespace mdm {
namespace {

// u in (0, 1]: (bits + 1) * 2^-32 evaluated exactly in fp32 steps that the numpy oracle repeats
__device__ __forceinline__ float u01(uint32_t b) { return ((float)(b >> 8) + 1.0f) * (1.0f / 16777216.0f); }

__global__ void philox_normal_kernel(float* __restrict__ out, int64_t per_sample, int nsamples, int64_t sample0,
                                     const int64_t* __restrict__ sample_ids, uint64_t seed,
                                     const int* __restrict__ stream_dev, int stream_imm) {
  const uint32_t stream = (uint32_t)(stream_dev ? *stream_dev : stream_imm);
  const int64_t quads = (per_sample + 3) >> 2;  // 4 normals per Philox call
  const int64_t total = quads * nsamples;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = i / quads, qd = i - s * quads;
    const uint64_t gs = sample_ids ? (uint64_t)sample_ids[s] : (uint64_t)(sample0 + s);
    uint32_t c[4] = {(uint32_t)qd, (uint32_t)gs, (uint32_t)(gs >> 32), stream};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    float z[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // Box-Muller on (u1, u2) = (c[2h], c[2h+1])
      const float u1 = u01(c[2 * h]), u2 = u01(c[2 * h + 1]);
      const float r = sqrtf(-2.0f * logf(u1));
      float sn, cs;
      sincosf(6.283185307179586f * u2, &sn, &cs);
      z[2 * h] = r * cs, z[2 * h + 1] = r * sn;
    }
    float* o = out + s * per_sample + 4 * qd;
    const int64_t left = per_sample - 4 * qd;
    if (left >= 4 && ((((uintptr_t)o) & 15) == 0)) {
      *(f32x4*)o = (f32x4){z[0], z[1], z[2], z[3]};
    } else {
      for (int k = 0; k < 4 && k < left; ++k) o[k] = z[k];
    }
  }
}

}  // namespace

int philox_normal(float* out, int64_t per_sample, int nsamples, int64_t sample0, const int64_t* sample_ids, uint64_t seed,
                  const int* stream_dev, int stream_imm, hipStream_t s) {
  if (per_sample <= 0 || nsamples <= 0) return MDM_OK;
  if (!out || sample0 < 0) return MDM_ERR_ARG;
  const int64_t total = ((per_sample + 3) >> 2) * nsamples;
  int64_t blocks = (total + 255) / 256;
  hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, s, out, per_sample,
                     nsamples, sample0, sample_ids, seed, stream_dev, stream_imm);
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

}  // namespace mdm

extern "C" int mdm_noise_normal(float* out, int64_t per_sample, int32_t nsamples, int64_t sample0, uint64_t seed,
                                const int32_t* stream_dev, int32_t stream_imm, void* stream) {
  return mdm::philox_normal(out, per_sample, nsamples, sample0, nullptr, seed, stream_dev, stream_imm, (hipStream_t)stream);
}

// the same with an explicit GLOBAL sample index per row (device int64 [nsamples]): rows of a length-bucketed batch
extern "C" int mdm_noise_normal_ids(float* out, int64_t per_sample, int32_t nsamples, const int64_t* sample_ids, uint64_t seed,
                                    const int32_t* stream_dev, int32_t stream_imm, void* stream) {
  if (!sample_ids) return MDM_ERR_ARG;
  return mdm::philox_normal(out, per_sample, nsamples, 0, sample_ids, seed, stream_dev, stream_imm, (hipStream_t)stream);
}
