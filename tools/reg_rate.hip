// Micro-benchmark beside tools/dma_rate.hip: the same streams with plain 16-byte global loads into REGISTERS (what the streamed-weight
// kernels do: csrc/mlp_stream.hip, style_gemm.hip) instead of LDS-DMA.  DEPTH loads in flight per wave, 8 waves per CU, no compute.
//   hipcc --offload-arch=gfx950 -O3 tools/reg_rate.hip -o /tmp/reg_rate && /tmp/reg_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void reg_kernel(const uint8_t* src, size_t window, size_t stride_per_block, int iters, uint32_t* sink) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const uint8_t* base = src + (size_t)blockIdx.x * stride_per_block;
  size_t off = (size_t)wid * 1024 + lane * 16;
  u32x4 R[DEPTH];
  uint32_t acc = 0;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    R[d] = *(const u32x4*)(base + off);
    off += WAVES * 1024;
    if (off >= window) off -= window;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      acc += R[d][0] ^ R[d][3];  // consume slot d (waits for it only), then refill it: DEPTH - 1 loads stay in flight
      R[d] = *(const u32x4*)(base + off);
      off += WAVES * 1024;
      if (off >= window) off -= window;
    }
  }
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) acc += R[d][1];
  if (acc == 0x12345678u) sink[threadIdx.x] = acc;
}

template <int DEPTH, int WAVES>
void run(const char* name, const uint8_t* src, size_t window, size_t stride, int blocks, int iters, uint32_t* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((reg_kernel<DEPTH, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, src, window, stride, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  double bytes_per_block = (double)(iters + 1) * DEPTH * WAVES * 1024;
  printf("%-34s depth %2d waves %2d blocks %3d: %7.1f us, %6.2f TB/s aggregate, %5.1f GB/s per block\n", name, DEPTH, WAVES, blocks, ms * 1e3,
         bytes_per_block * blocks / (ms * 1e-3) / 1e12, bytes_per_block / (ms * 1e-3) / 1e9);
}

int main() {
  const size_t total = (size_t)1 << 30;
  uint8_t* buf;
  uint32_t* sink;
  hipMalloc(&buf, total);
  hipMalloc(&sink, 4096);
  hipMemset(buf, 1, total);
  run<8, 8>("shared 1 MiB window (L2 hits)", buf, 1 << 20, 0, 256, 64, sink);
  run<16, 8>("shared 1 MiB window (L2 hits)", buf, 1 << 20, 0, 256, 32, sink);
  run<16, 8>("shared 1 MiB, 2 blocks per CU", buf, 1 << 20, 0, 512, 32, sink);
  run<32, 8>("shared 1 MiB window (L2 hits)", buf, 1 << 20, 0, 256, 16, sink);
  run<16, 4>("shared 1 MiB window, 4 waves", buf, 1 << 20, 0, 256, 64, sink);
  run<16, 8>("shared 2 MiB window (L2 hits)", buf, 2 << 20, 0, 256, 32, sink);
  run<16, 8>("shared 16 MiB window", buf, 16 << 20, 0, 256, 32, sink);
  run<16, 8>("private 4 MiB windows", buf, 4 << 20, 4 << 20, 256, 32, sink);
  run<16, 8>("one block alone, shared window", buf, 1 << 20, 0, 1, 64, sink);
  return 0;
}
