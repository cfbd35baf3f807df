// Probe: how fast can ONE CU retire an epilogue-shaped store stream, as a function of how many CUs store at once?
// Build: hipcc -O3 --offload-arch=gfx950 store_rate.hip -o store_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// one workgroup = 8 waves; a "tile" = 256 rows x 256 bf16 columns at row pitch `pitch` bytes (the WIDE epilogue's pattern:
// a wave-instruction covers 8 rows x 128 bytes) or x 256 fp32 columns (16 lanes x 16 B = one 256-byte row piece).
template <int F32>
__global__ __launch_bounds__(512) void store_tiles(char* out, size_t pitch, int tiles_n, int reps, int stride_tiles) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = w >> 2, wn = w & 3;
  uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
  for (int r = 0; r < reps; ++r) {
    const int tile = blockIdx.x + r * stride_tiles;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    if (!F32) {
      char* base = out + (size_t)(tm * 256 + wm * 128) * pitch + (size_t)(tn * 256 + wn * 64) * 2;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = g * 32 + it * 8 + (lane >> 3);
          *(uint4*)(base + (size_t)row * pitch + (lane & 7) * 16) = v;
        }
    } else {
      char* base = out + (size_t)(tm * 256 + wm * 128) * pitch + (size_t)(tn * 256 + wn * 64) * 4;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = g * 32 + it * 4 + (lane >> 4);
          *(uint4*)(base + (size_t)row * pitch + (lane & 15) * 16) = v;
        }
    }
  }
}

int main() {
  const int tiles_n = 9, tiles_m = 145;           // the qkv output at B=64: 36928 x 2304
  const size_t pitch_bf = 2304 * 2, pitch_f = 2304 * 4;
  char* buf;
  hipMalloc(&buf, (size_t)tiles_m * 256 * pitch_f + (1 << 20));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int f32 = 0; f32 < 2; ++f32)
    for (int G : {256, 128, 64, 32, 16, 8}) {
      const int reps = 5;                        // 5 tiles per workgroup, all distinct
      auto launch = [&]() {
        if (f32) hipLaunchKernelGGL(store_tiles<1>, dim3(G), dim3(512), 0, 0, buf, pitch_f, tiles_n, reps, 256);
        else hipLaunchKernelGGL(store_tiles<0>, dim3(G), dim3(512), 0, 0, buf, pitch_bf, tiles_n, reps, 256);
      };
      for (int i = 0; i < 3; ++i) launch();
      hipDeviceSynchronize();
      const int iters = 20;
      hipEventRecord(e0);
      for (int i = 0; i < iters; ++i) launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / iters;
      const double bytes_wg = (double)reps * 256 * 256 * (f32 ? 4 : 2);
      printf("%s G=%3d: %.1f us per launch, %.2f us per tile, %.1f GB/s per CU, %.2f TB/s total\n", f32 ? "f32 " : "bf16", G, us,
             us / reps, bytes_wg / us / 1e3, bytes_wg * G / us / 1e6);
    }
  return 0;
}
