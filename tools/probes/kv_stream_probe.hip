// Read bandwidth of the decode step's K / V access pattern (csrc/attn.hip attn_decode_kernel): a workgroup of 256 threads streams `nkeys`
// segments of 128 B (one head's 64 dims of one key row), 8 lanes x 16 B per segment, 32 segments per sweep position, `U` loads in flight per
// thread -- (a) in place in the prefill's packed qkv rows (pitch 4608 B, the 12 heads' segments of a row adjacent: what the kernel reads today),
// (b) from a per-(image, head) contiguous copy (pitch 128 B).  One workgroup per (image, head), K pass then V pass, as the kernel does.
//   hipcc --offload-arch=gfx950 -O3 -o _bin/kv_stream_probe kv_stream_probe.hip && _bin/kv_stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int U>
__global__ __launch_bounds__(256) void probe(const char* __restrict__ base, size_t img_stride, size_t head_stride, size_t row_pitch,
                                             size_t v_off, int nkeys, int heads, float* __restrict__ out) {
  const int tid = threadIdx.x, sub = tid & 7, kslot = tid >> 3;
  const int img = blockIdx.x / heads, h = blockIdx.x % heads;
  const char* p = base + img * img_stride + h * head_stride + sub * 16;
  float acc = 0.f;
  for (int pass = 0; pass < 2; ++pass) {
    const char* q = p + pass * v_off;
    for (int kb = kslot; kb < nkeys; kb += 32 * U) {
      uint4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int k = kb + u * 32;
        k = k < nkeys ? k : nkeys - 1;
        v[u] = *(const uint4*)(q + (size_t)k * row_pitch);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc += __uint_as_float(v[u].x) + __uint_as_float(v[u].y) + __uint_as_float(v[u].z) + __uint_as_float(v[u].w);
    }
    __syncthreads();
  }
  if (acc == 123.456f) out[blockIdx.x] = acc;
}

template <int U>
static float run(const char* base, size_t copy_stride, int copies, size_t is, size_t hs, size_t rp, size_t vo, int nkeys, int heads, int B, float* out, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // every launch reads ANOTHER of the `copies` layers (as the decode step's four layers do): 4 x 114 MB at B = 64 do not fit the 256 MB Infinity Cache
  for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(probe<U>, dim3(B * heads), dim3(256), 0, 0, base + (i % copies) * copy_stride, is, hs, rp, vo, nkeys, heads, out);
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(probe<U>, dim3(B * heads), dim3(256), 0, 0, base + (i % copies) * copy_stride, is, hs, rp, vo, nkeys, heads, out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters * 1e3f;
}

int main() {
  const int S = 578, H = 12, L = 4;
  for (int B : {64, 512}) {
    // four layers' buffers so that successive launches do not re-read a cache-resident copy: the launch loop walks them round robin
    const size_t packed = (size_t)B * S * 2304 * 2;          // one layer's packed qkv
    char* buf;
    CK(hipMalloc(&buf, packed * L));
    CK(hipMemset(buf, 1, packed * L));
    float* out;
    CK(hipMalloc(&out, 1 << 20));
    const double bytes = (double)B * H * S * 128 * 2;       // K + V segments
    for (int layout = 0; layout < 2; ++layout) {
      // 0: in place (row pitch 4608 B, head stride 128 B, V 1536 B behind K); 1: contiguous per (image, head): [K 578 x 128 B | V 578 x 128 B]
      const size_t rp = layout == 0 ? 4608 : 128;
      const size_t hs = layout == 0 ? 128 : (size_t)S * 128 * 2;
      const size_t is = layout == 0 ? (size_t)S * 4608 : (size_t)H * S * 128 * 2;
      const size_t vo = layout == 0 ? 1536 : (size_t)S * 128;
      const char* base = buf + (layout == 0 ? 1536 : 0);     // K columns start 1536 B into a packed row
      const float t4 = run<4>(base, packed, L, is, hs, rp, vo, S, H, B, out, 32) * L;
      const float t8 = run<8>(base, packed, L, is, hs, rp, vo, S, H, B, out, 32) * L;
      const float t16 = run<16>(base, packed, L, is, hs, rp, vo, S, H, B, out, 32) * L;
      printf("B=%3d %-34s  4 loads in flight %7.1f us %5.2f TB/s | 8: %7.1f us %5.2f TB/s | 16: %7.1f us %5.2f TB/s\n", B,
             layout == 0 ? "in place (pitch 4608 B)" : "contiguous per (image, head)", t4 / L, bytes / (t4 / L) / 1e6, t8 / L, bytes / (t8 / L) / 1e6,
             t16 / L, bytes / (t16 / L) / 1e6);
    }
    CK(hipFree(buf)); CK(hipFree(out));
  }
  return 0;
}
