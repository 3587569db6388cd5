// On-device post-processing of generated motions: HumanML3D 263-d rows -> 22 x 3 joint positions.
//   de-normalise  x * std + mean                                           tools/visualization.py:89
//   root: Y rotation = cumsum of its velocity, XZ position = cumsum of the rotated velocity, height as is
//                                                                          utils/motion_process.py:362-382
//   joints: rotation-invariant coordinates rotated back by the inverse root rotation, root XZ added
//                                                                          utils/motion_process.py:403-416, utils/quaternion.py:16-20,54-73
//   temporal smoothing: gaussian filter (sigma, radius = int(4 sigma + 0.5), "nearest" edges) per coordinate
//                                                                          utils/utils.py:125-130 (scipy.ndimage.gaussian_filter)
// One workgroup per sample; the two prefix sums run in one thread with double accumulators rounded to fp32 per element
// (what torch.cumsum does on fp32 CPU tensors), products/sums of the rotation keep the reference's operation order with
// contraction disabled, and the filter accumulates symmetric pairs in double like scipy's correlate1d: results agree with
// the reference to the last bits of the libm sin/cos.  HBM-bound and tiny (B x T x 263 floats in, B x T x 66 out).
#include "kernels.h"

namespace mdm {
namespace {

__device__ __forceinline__ float mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub(float a, float b) { return __fsub_rn(a, b); }

// v rotated by the quaternion (w, 0, qy, 0): qrot with cross products written out (utils/quaternion.py:70-73)
__device__ __forceinline__ void rot_y(float w, float qy, float vx, float vy, float vz, float& ox, float& oy, float& oz) {
  const float uvx = mul(qy, vz), uvz = -mul(qy, vx);        // uv = cross(qvec, v), qvec = (0, qy, 0)
  const float uuvx = mul(qy, uvz), uuvz = -mul(qy, uvx);    // uuv = cross(qvec, uv)
  ox = add(vx, mul(2.f, add(mul(w, uvx), uuvx)));
  oy = vy;                                                  // + 2 * (w * 0 + 0)
  oz = add(vz, mul(2.f, add(mul(w, uvz), uuvz)));
}

__global__ __launch_bounds__(256) void motion_post_kernel(const float* __restrict__ x, const int* __restrict__ len,
                                                          const float* __restrict__ mean, const float* __restrict__ sd,
                                                          int T, int feats, int J, int radius,
                                                          const double* __restrict__ wts, float* __restrict__ raw,
                                                          float* __restrict__ out) {
  extern __shared__ float sh[];  // cw[T], sw[T], px[T], pz[T], py[T]
  float* cw = sh;
  float* sw = cw + T;
  float* px = sw + T;
  float* pz = px + T;
  float* py = pz + T;
  const int b = blockIdx.x, tid = threadIdx.x;
  int n = len ? len[b] : T;
  n = n < 0 ? 0 : (n > T ? T : n);
  const float* xb = x + (int64_t)b * T * feats;
  auto val = [&](int t, int c) { return add(mul(xb[(int64_t)t * feats + c], sd[c]), mean[c]); };
  // root rotation angle: exclusive prefix sum of the rotation velocity (:364-367), then cos / sin
  if (tid == 0) {
    double acc = 0.0;
    for (int t = 0; t < n; ++t) {
      if (t > 0) acc += (double)val(t - 1, 0);
      const float a = (float)acc;
      cw[t] = cosf(a), sw[t] = sinf(a);
    }
  }
  __syncthreads();
  // root XZ: previous frame's velocity rotated by the inverse rotation (:373-376), prefix-summed (:378)
  for (int t = tid; t < n; t += 256) {
    float vx = 0.f, vz = 0.f;
    if (t > 0) vx = val(t - 1, 1), vz = val(t - 1, 2);
    float ox, oy, oz;
    rot_y(cw[t], -sw[t], vx, 0.f, vz, ox, oy, oz);
    px[t] = ox, pz[t] = oz, py[t] = val(t, 3);
  }
  __syncthreads();
  if (tid == 0) {
    double ax = 0.0, az = 0.0;
    for (int t = 0; t < n; ++t) {
      ax += (double)px[t], az += (double)pz[t];
      px[t] = (float)ax, pz[t] = (float)az;
    }
  }
  __syncthreads();
  // joints (:403-416)
  float* rb = raw + (int64_t)b * T * J * 3;
  for (int i = tid; i < n * J; i += 256) {
    const int t = i / J, j = i - t * J;
    float ox, oy, oz;
    if (j == 0) {
      ox = px[t], oy = py[t], oz = pz[t];
    } else {
      const int c = 4 + 3 * (j - 1);
      rot_y(cw[t], -sw[t], val(t, c), val(t, c + 1), val(t, c + 2), ox, oy, oz);
      ox = add(ox, px[t]), oz = add(oz, pz[t]);
    }
    float* o = rb + (int64_t)i * 3;
    o[0] = ox, o[1] = oy, o[2] = oz;
  }
  __syncthreads();
  // temporal gaussian filter over the valid frames, "nearest" edges; frames past the length are zeroed
  float* ob = out + (int64_t)b * T * J * 3;
  const int W = J * 3;
  for (int i = tid; i < T * W; i += 256) {
    const int t = i / W, c = i - t * W;
    float r = 0.f;
    if (t < n) {
      if (radius <= 0) {
        r = rb[i];
      } else {
        double acc = (double)rb[(int64_t)t * W + c] * wts[0];
        for (int k = 1; k <= radius; ++k) {
          const int lo = t - k < 0 ? 0 : t - k, hi = t + k > n - 1 ? n - 1 : t + k;
          acc += ((double)rb[(int64_t)lo * W + c] + (double)rb[(int64_t)hi * W + c]) * wts[k];
        }
        r = (float)acc;
      }
    }
    ob[i] = r;
  }
}

}  // namespace

int motion_post(const float* x, const int* len, const float* mean, const float* sd, int B, int T, int feats, int J,
                int radius, const double* wts, float* raw, float* out, hipStream_t s) {
  const int smem = 5 * T * (int)sizeof(float);
  if (smem > 64 * 1024) return MDM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(motion_post_kernel, dim3(B), dim3(256), smem, s, x, len, mean, sd, T, feats, J, radius, wts, raw, out);
  MDM_RETURN_IF_LAUNCH_FAILED();
  return MDM_OK;
}

}  // namespace mdm
