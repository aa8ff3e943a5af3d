// micro-test: __builtin_amdgcn_global_load_lds(16 B) writes LDS[base + lane*16]; per-lane global source.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint4* src, const int* perm, uint4* out) {
  __shared__ uint4 lds[256];
  const int tid = threadIdx.x, wave = tid >> 6;
  const uint4* g = src + perm[tid];
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)(lds + wave * 64), 16, 0, 0);
  __syncthreads();
  out[tid] = lds[tid];
}
int main() {
  std::vector<uint4> h(256); std::vector<int> p(256);
  for (int i = 0; i < 256; ++i) { h[i] = make_uint4(i, i * 2, i * 3, i * 4); p[i] = (i * 37 + 5) % 256; }
  uint4 *d, *o; int* dp;
  hipMalloc(&d, 4096); hipMalloc(&o, 4096); hipMalloc(&dp, 1024);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice); hipMemcpy(dp, p.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, dp, o);
  std::vector<uint4> r(256); hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) if (r[i].x != (unsigned)p[i] || r[i].w != (unsigned)p[i] * 4) ++bad;
  printf("glds bad=%d (%s)\n", bad, hipGetErrorString(hipGetLastError()));
  return bad != 0;
}
