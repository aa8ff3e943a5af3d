// dev microbenchmark: what a wave's batch of eight ds_read_b128 costs (issue -> all data back) for the two address patterns of conv_tile64.hip,
// with 4 or 8 waves per CU doing the same.   hipcc --offload-arch=gfx950 -O3 tools/dev/lds_rate.hip -o /tmp/lds_rate && /tmp/lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int PAT, int NR>
__global__ __launch_bounds__(512) void k(unsigned long long* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  char* b = reinterpret_cast<char*>(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 160 * 1024 / 16; i += blockDim.x) lds[i] = make_uint4(i, i + 1, i + 2, i + 3);
  __syncthreads();
  const int fr = lane & 15, fg = lane >> 4;
  unsigned addr[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    if (PAT == 0) addr[j] = j * 1024 + lane * 16 + (wave & 3) * 8192;                       // fragment-order weights: one contiguous KB per read
    else if (PAT == 1) { const int q = (wave & 3) * 68 + (j >> 1) * 34 + (j & 1) * 16 + fr + 1, v = q * 128 + fg * 16; addr[j] = 73984 + (v ^ ((v >> 3) & 0x60)); }   // patch pixels, 128-byte rows, swizzled
    else { const int q = (wave & 3) * 68 + (j >> 1) * 34 + (j & 1) * 16 + fr + 1; addr[j] = 73984 + q * 128 + fg * 16; }   // the same without the swizzle (conflicts)
  }
  uint4 acc = make_uint4(0, 0, 0, 0);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    uint4 v[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) v[j] = *reinterpret_cast<const uint4*>(b + addr[j]);
#pragma unroll
    for (int j = 0; j < NR; ++j) { acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w; }
#pragma unroll
    for (int j = 0; j < NR; ++j) asm volatile("" : "+v"(addr[j]));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  if (acc.x == 0x12345678u && acc.y == 1 && acc.z == 2 && acc.w == 3) out[0] = 0;
}
template <int PAT, int NR> static void run(const char* name, int threads) {
  unsigned long long* d; hipMalloc(&d, 256 * 8 * 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<PAT, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<PAT, NR>), dim3(256), dim3(threads), 160 * 1024, 0, d, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256 * 8);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0; int nw = threads / 64;
  for (int bI = 0; bI < 256; ++bI) for (int w = 0; w < nw; ++w) s += (double)h[bI * 8 + w];
  s /= 256.0 * nw * iters;
  printf("%-46s %d waves/CU, %d reads per batch: %7.1f s_memtime ticks per batch = %5.1f per ds_read_b128 per wave -> %6.1f B/tick/CU\n", name, nw, NR, s, s / NR, nw * NR * 1024.0 / s);
  hipFree(d);
}
int main() {
  run<0, 8>("contiguous KB (weights)", 256); run<0, 8>("contiguous KB (weights)", 512);
  run<1, 8>("patch pixels, swizzled 128-byte rows", 256); run<1, 8>("patch pixels, swizzled 128-byte rows", 512);
  run<2, 8>("patch pixels, linear 128-byte rows", 256);
  run<0, 4>("contiguous KB (weights)", 256); run<0, 12>("contiguous KB (weights)", 256); run<1, 12>("patch pixels, swizzled", 256);
  return 0;
}
