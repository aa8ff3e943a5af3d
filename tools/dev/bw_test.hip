// Bandwidth calibration: (1) plain uint4 copy, (2) copy staged through LDS with global_load_lds in the conv kernel's
// 2-stage / one-barrier-per-step structure (no MFMA), per block 128 rows x 128 B per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
__global__ __launch_bounds__(256) void copy_k(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__device__ __forceinline__ void glds16(const void* src, unsigned lds_wave_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_wave_byte_addr) : "memory");
}
// each block streams `steps` chunks of 16 KB (1024 uint4) through 2 LDS stages and writes them back out
template <int NST>
__global__ __launch_bounds__(256) void dma_copy_k(const uint4* __restrict__ a, uint4* __restrict__ b, int steps, int nblk_chunks) {
  __shared__ uint4 lds[NST * 1024];
  const int tid = threadIdx.x, wave = tid >> 6;
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds;
  for (int c = blockIdx.x; c < nblk_chunks; c += gridDim.x) {
    const uint4* src = a + (size_t)c * steps * 1024;
    uint4* dst = b + (size_t)c * steps * 1024;
    for (int pre = 0; pre < NST - 1 && pre < steps; ++pre)
      for (int i = 0; i < 4; ++i) glds16(src + pre * 1024 + i * 256 + tid, __builtin_amdgcn_readfirstlane(base + (pre * 1024 + i * 256 + wave * 64) * 16));
    for (int s = 0; s < steps; ++s) {
      if (NST == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // leave the newest step (4 DMAs) in flight
      __syncthreads();
      const int nx = s + NST - 1;
      if (nx < steps)
        for (int i = 0; i < 4; ++i) glds16(src + nx * 1024 + i * 256 + tid, __builtin_amdgcn_readfirstlane(base + ((nx % NST) * 1024 + i * 256 + wave * 64) * 16));
      else if (NST == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      uint4 v[4];
      for (int i = 0; i < 4; ++i) v[i] = lds[(s % NST) * 1024 + i * 256 + tid];
      for (int i = 0; i < 4; ++i) dst[s * 1024 + i * 256 + tid] = v[i];
    }
    __syncthreads();
  }
}
int main() {
  const size_t bytes = (size_t)1 << 30; const size_t n = bytes / 16;
  uint4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](auto f, const char* name) {
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 10; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-40s %.3f ms  %.0f GB/s (read+write)\n", name, ms, 2.0 * bytes / ms / 1e6);
  };
  for (int g : {2048, 8192}) timeit([&] { hipLaunchKernelGGL(copy_k, dim3(g), dim3(256), 0, 0, a, b, n); }, g == 2048 ? "plain copy grid 2048" : "plain copy grid 8192");
  timeit([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, "hipMemcpy D2D");
  for (int steps : {1, 4, 16}) {
    const int chunks = (int)(n / 1024 / steps);
    char nm[64];
    for (int grid : {512, 768, 1024}) {
      snprintf(nm, 64, "dma 2-stage steps=%d grid=%d", steps, grid);
      timeit([&] { hipLaunchKernelGGL(dma_copy_k<2>, dim3(grid), dim3(256), 0, 0, a, b, steps, chunks); }, nm);
    }
    snprintf(nm, 64, "dma 3-stage steps=%d grid=768", steps);
    timeit([&] { hipLaunchKernelGGL(dma_copy_k<3>, dim3(768), dim3(256), 0, 0, a, b, steps, chunks); }, nm);
    snprintf(nm, 64, "dma 2-stage steps=%d grid=chunks", steps);
    timeit([&] { hipLaunchKernelGGL(dma_copy_k<2>, dim3(chunks), dim3(256), 0, 0, a, b, steps, chunks); }, nm);
  }
  return 0;
}
