// Which CUs does a CU-masked stream use?  Each block records HW_REG_HW_ID and HW_REG_XCC_ID; the host counts distinct CUs.
// build: hipcc --offload-arch=gfx950 -O2 cumask.hip -o _bin/cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
#include <cstdint>
__global__ void probe(uint32_t* out, int spin) {
  const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
  // keep the CU busy for a while so the blocks spread over everything the mask allows
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main() {
  const int NB = 4096;
  uint32_t* d;
  hipMalloc(&d, NB * 8);
  std::vector<uint32_t> h(NB * 2);
  const int patterns = 7;
  for (int p = 0; p < patterns; ++p) {
    uint32_t words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const char* name = "";
    switch (p) {
      case 0: name = "no mask"; break;
      case 1: name = "every 2nd bit"; for (int i = 0; i < 256; i += 2) words[i / 32] |= 1u << (i % 32); break;
      case 2: name = "every 8th bit"; for (int i = 0; i < 256; i += 8) words[i / 32] |= 1u << (i % 32); break;
      case 3: name = "every 32nd bit"; for (int i = 0; i < 256; i += 32) words[i / 32] |= 1u << (i % 32); break;
      case 4: name = "bits 0..63"; words[0] = words[1] = 0xffffffffu; break;
      case 5: name = "bits 0..31"; words[0] = 0xffffffffu; break;
      case 6: name = "bits 0..7"; words[0] = 0xffu; break;
    }
    hipStream_t s;
    hipError_t rc = p == 0 ? hipStreamCreate(&s) : hipExtStreamCreateWithCUMask(&s, 8, words);
    if (rc != hipSuccess) { printf("%s: create failed %d\n", name, (int)rc); continue; }
    hipMemsetAsync(d, 0xff, NB * 8, s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    hipLaunchKernelGGL(probe, dim3(NB), dim3(256), 0, s, d, 20000);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d, NB * 8, hipMemcpyDeviceToHost);
    std::set<uint32_t> cus, xccs;
    for (int b = 0; b < NB; ++b) {
      const uint32_t hw = h[2 * b], x = h[2 * b + 1];
      const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
      cus.insert((x << 16) | (se << 8) | (sh << 4) | cu);
      xccs.insert(x);
    }
    printf("%-16s: %zu distinct CUs on %zu XCDs, %.3f ms; first few:", name, cus.size(), xccs.size(), ms);
    int n = 0;
    for (uint32_t c : cus) { if (n++ < 10) printf(" x%u.se%u.cu%u", c >> 16, (c >> 8) & 0xff, c & 0xf); }
    printf("\n");
  }
  return 0;
}
