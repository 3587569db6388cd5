// Micro-benchmark: what global -> LDS rate can ONE CU sustain with global_load_lds_dwordx4 (LDS-DMA)?
// Every workgroup (8 waves, one per CU: 160 KiB LDS requested) streams its own window of a source buffer into LDS with
// DEPTH instructions in flight per wave (counted vmcnt), no compute.  Source windows are either private per CU and larger
// than L2 (fabric / Infinity-Cache / HBM path) or one small window shared by all CUs (L2 hits).
//   hipcc --offload-arch=gfx950 -O3 tools/dma_rate.hip -o /tmp/dma_rate && /tmp/dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ void glds16(const void* g, uint8_t* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int DEPTH>
__global__ __launch_bounds__(512) void dma_kernel(const uint8_t* src, size_t window, size_t stride_per_block, int iters,
                                                  unsigned long long* cycles) {
  extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const uint8_t* base = src + (size_t)blockIdx.x * stride_per_block;
  unsigned long long t0 = 0, t1 = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  size_t off = (size_t)wid * 1024 + lane * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      glds16(base + off, smem + ((wid * DEPTH + d) & 127) * 1024);
      off += 8 * 1024;
      if (off >= window) off -= window;
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");  // keep half of them in flight
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int DEPTH>
void run(const char* name, const uint8_t* src, size_t window, size_t stride, int blocks, int iters) {
  unsigned long long* d;
  hipMalloc(&d, blocks * sizeof(unsigned long long));
  hipFuncSetAttribute((const void*)dma_kernel<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(dma_kernel<DEPTH>, dim3(blocks), dim3(512), 160 * 1024, 0, src, window, stride, iters, d);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), d, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double bytes_per_block = (double)iters * DEPTH * 8 * 1024;
  double avg = 0;
  for (auto c : h) avg += (double)c;
  avg /= blocks;
  printf("%-34s depth %2d  blocks %3d: %7.1f us, %6.2f TB/s aggregate, %5.1f GB/s per CU, s_memtime ticks/KiB %.1f\n", name, DEPTH, blocks,
         ms * 1e3, bytes_per_block * blocks / (ms * 1e-3) / 1e12, bytes_per_block / (ms * 1e-3) / 1e9, avg / (bytes_per_block / 1024));
  hipFree(d);
}

int main() {
  const size_t total = (size_t)1 << 30;
  uint8_t* buf;
  hipMalloc(&buf, total);
  hipMemset(buf, 1, total);
  const int iters = 256;
  // private 4-MiB windows (256 CUs x 4 MiB = 1 GiB: nothing is re-read from L2)
  run<4>("private 4 MiB windows", buf, 4 << 20, 4 << 20, 256, iters);
  run<8>("private 4 MiB windows", buf, 4 << 20, 4 << 20, 256, iters / 2);
  run<16>("private 4 MiB windows", buf, 4 << 20, 4 << 20, 256, iters / 4);
  // one 1-MiB window shared by every CU (L2-resident weights)
  run<4>("shared 1 MiB window (L2 hits)", buf, 1 << 20, 0, 256, iters);
  run<8>("shared 1 MiB window (L2 hits)", buf, 1 << 20, 0, 256, iters / 2);
  run<16>("shared 1 MiB window (L2 hits)", buf, 1 << 20, 0, 256, iters / 4);
  // private 256-KiB windows re-read (a CU's own tile, L2 / MALL resident after the first pass)
  run<8>("private 256 KiB windows, re-read", buf, 256 << 10, 256 << 10, 256, iters / 2);
  run<8>("one CU alone, shared window", buf, 1 << 20, 0, 1, iters / 2);
  return 0;
}
