// Probe of gfx950 v_permlane16_swap / v_permlane32_swap semantics (prints what every lane receives).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
__global__ void k(unsigned* out) {
  const unsigned lane = threadIdx.x;
  unsigned a = 100 + lane, b = 200 + lane;
  u32x2 r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  u32x2 r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[lane * 4 + 0] = r16[0]; out[lane * 4 + 1] = r16[1];
  out[lane * 4 + 2] = r32[0]; out[lane * 4 + 3] = r32[1];
}
int main() {
  unsigned *d, h[256];
  (void)hipMalloc(&d, sizeof(h));
  k<<<1, 64>>>(d);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 1) printf("lane %2d: p16 (%3u,%3u)  p32 (%3u,%3u)\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
}
