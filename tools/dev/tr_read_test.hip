// dev: what does ds_read_b64_tr_b16 deliver?  lds[i] = i (16-bit); every lane supplies its own 8-byte-aligned address.
// hipcc --offload-arch=gfx950 -O3 tools/dev/tr_read_test.hip -o /tmp/tr && /tmp/tr
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int pitch) {
  __shared__ short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15;
  // hypothesis: the 16 lanes of a group read a [4 rows][16 cols] block - lane i points at row (i>>2), cols 4*(i&3)..+3 - and lane i
  // receives column i of the block, rows 0..3.  Rows are `pitch` elements apart; group g reads rows 4g..4g+3.
  __attribute__((address_space(3))) s16x4* p = (__attribute__((address_space(3))) s16x4*)(lds + (4 * g + (i >> 2)) * pitch + (i & 3) * 4);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  for (int pitch : {16, 80}) {
    k<<<1, 64>>>(d, pitch);
    short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { int g = l >> 4, i = l & 15; int want = (4 * g + j) * pitch + i; if (h[l * 4 + j] != want) ok = 0; }
    printf("pitch %d: hypothesis %s; lane 0: %d %d %d %d  lane 1: %d %d %d %d  lane 5: %d %d %d %d  lane 17: %d %d %d %d\n", pitch, ok ? "HOLDS" : "fails",
           h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[20], h[21], h[22], h[23], h[68], h[69], h[70], h[71]);
  }
  return 0;
}
