// Probe (not product): the matrix pipe alone on random register operands, per MFMA shape -- TFLOP/s from hipEvents; board power and
// sclk are sampled beside it by tools/mfma_power.py.  Question (round 6): at the board's power cap, is bf16 32x32x16 (half the operand
// register reads per flop) cheaper per flop than the 16x16x32 form the GEMM main loops use?
//   mfma_power <variant 0..3> <seconds> <waves per SIMD>
//   0: v_mfma_f32_16x16x32_bf16, 16 independent accumulators   1: v_mfma_f32_32x32x16_bf16, 4 independent accumulators
//   2 / 3: the same on ZERO operands
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int V>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned seed, int zero) {
  unsigned s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
  bf16x8 a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    union { unsigned u[4]; bf16x8 v; } ua, ub;
    for (int j = 0; j < 4; ++j) {
      // two bf16 in (-2, 2) with random mantissas
      unsigned r = rnd(), q = rnd();
      ua.u[j] = zero ? 0u : ((r & 0x80ff80ffu) | 0x3f003f00u);
      ub.u[j] = zero ? 0u : ((q & 0x80ff80ffu) | 0x3f003f00u);
    }
    a[i] = ua.v; b[i] = ub.v;
  }
  if constexpr (V == 0) {
    f32x4 c[16];
    for (int i = 0; i < 16; ++i) c[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 7], b[(i >> 1) & 7], c[i], 0, 0, 0);
    }
    f32x4 t = c[0];
    for (int i = 1; i < 16; ++i) t += c[i];
    out[blockIdx.x * 256 + threadIdx.x] = t[0] + t[1] + t[2] + t[3];
  } else {
    f32x16 c[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + 4 * r) & 7], b[(i * 2 + r) & 7], c[i], 0, 0, 0);
    }
    f32x16 t = c[0] + c[1] + c[2] + c[3];
    float sum = 0;
    for (int j = 0; j < 16; ++j) sum += t[j];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
  }
}

int main(int argc, char** argv) {
  const int v = argc > 1 ? atoi(argv[1]) : 0;
  const double secs = argc > 2 ? atof(argv[2]) : 3.0;
  const int wps = argc > 3 ? atoi(argv[3]) : 1;
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * 4);
  const int iters = 4096, blocks = 256 * wps;
  // flops per launch: V0: 16 MFMA x 16*16*32*2; V1: 8 MFMA x 32*32*16*2 -- the same 262144 per wave and iteration
  const double flop = (double)blocks * 4 * iters * 262144.0;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto launch = [&]() {
    if ((v & 1) == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 17u, v >> 1);
    else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 17u, v >> 1);
  };
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  long n = 0;
  hipEventRecord(e0, 0);
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    for (int i = 0; i < 20; ++i) launch();
    n += 20;
    hipDeviceSynchronize();
  }
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("variant %d (%s, %s operands, %d wave(s) per SIMD): %.1f us per launch, %.0f TFLOP/s\n", v, (v & 1) ? "32x32x16" : "16x16x32",
         (v >> 1) ? "zero" : "random", wps, ms * 1e3 / n, flop * n / (ms * 1e-3) * 1e-12);
  return 0;
}
