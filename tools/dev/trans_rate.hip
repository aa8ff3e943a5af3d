// dev microbench (gfx950): issue rate of the transcendental VALU ops in f32 and f16 - is v_exp_f16 / v_rcp_f16 cheaper than the f32 forms
// the epilogues use?   hipcc --offload-arch=gfx950 -O3 tools/dev/trans_rate.hip -o /tmp/trans_rate && /tmp/trans_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int OP> __global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a = threadIdx.x * 1e-3f + 0.5f, b = a + 0.1f, c = a + 0.2f, d = a + 0.3f;
  _Float16 ha = (_Float16)a, hb = (_Float16)b, hc = (_Float16)c, hd = (_Float16)d;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if constexpr (OP == 0) { a = __builtin_amdgcn_exp2f(a); b = __builtin_amdgcn_exp2f(b); c = __builtin_amdgcn_exp2f(c); d = __builtin_amdgcn_exp2f(d); }
      if constexpr (OP == 1) { a = __builtin_amdgcn_rcpf(a); b = __builtin_amdgcn_rcpf(b); c = __builtin_amdgcn_rcpf(c); d = __builtin_amdgcn_rcpf(d); }
      if constexpr (OP == 2) { asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3" : "+v"(ha), "+v"(hb), "+v"(hc), "+v"(hd)); }
      if constexpr (OP == 3) { asm volatile("v_rcp_f16 %0, %0\n v_rcp_f16 %1, %1\n v_rcp_f16 %2, %2\n v_rcp_f16 %3, %3" : "+v"(ha), "+v"(hb), "+v"(hc), "+v"(hd)); }
      if constexpr (OP == 4) { a = __builtin_fmaf(a, 1.0001f, 0.5f); b = __builtin_fmaf(b, 1.0001f, 0.5f); c = __builtin_fmaf(c, 1.0001f, 0.5f); d = __builtin_fmaf(d, 1.0001f, 0.5f); }
      if constexpr (OP == 5) { asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + (float)ha + (float)hb + (float)hc + (float)hd;
}
template <int OP> void run(const char* name) {
  float* out; hipMalloc(&out, 1024 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  k<OP><<<1024, 256>>>(out, 10); hipDeviceSynchronize();
  hipEventRecord(e0); k<OP><<<1024, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double insts = 1024.0 * 4 /*waves per block*/ * iters * 32;   // wave-instructions
  printf("%-12s %8.3f ms  %6.2f wave-instr per ns (chip)  -> %.2f cycles per wave-instr per SIMD at 2.4 GHz\n", name, ms, insts / (ms * 1e6), (ms * 1e6 * 2.4) / (insts / 1024.0));
  hipFree(out);
}
int main() { run<4>("v_fma_f32"); run<0>("exp2f builtin"); run<5>("v_exp_f32"); run<1>("v_rcp_f32"); run<2>("v_exp_f16"); run<3>("v_rcp_f16"); return 0; }
