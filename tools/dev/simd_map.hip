// Which SIMD does wave w of a 512-thread (8-wave) workgroup land on?  HW_REG_HW_ID (id 4): simd_id = bits [5:4], cu_id = [11:8]
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // id 4, offset 0, size 32
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = hw;
}
int main() {
  unsigned* d; hipMalloc(&d, 64 * 16 * 4); hipMemset(d, 0, 64 * 16 * 4);
  for (int nt : {256, 512}) {
    hipLaunchKernelGGL(k, dim3(6), dim3(nt), 0, 0, d);
    unsigned h[96]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 6; ++b) {
      printf("nt=%d block %d: simd of waves:", nt, b);
      for (int w = 0; w < nt / 64; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3);
      printf("   (cu %u)\n", (h[b * 16] >> 8) & 15);
    }
  }
  return 0;
}
