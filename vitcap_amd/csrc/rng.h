// Counter-based random bits shared by the sampling decoder and attention dropout: a pure function of
// (seed, coordinates), so a CPU restatement (oracle/vitcap_oracle.py: rng_u32) reproduces every draw.
#pragma once
#include <stdint.h>

__host__ __device__ __forceinline__ uint32_t vc_lowbias32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
// fold one more coordinate into a running hash
__host__ __device__ __forceinline__ uint32_t vc_mix(uint32_t h, uint32_t v) {
  return vc_lowbias32(h ^ (v + 0x9e3779b9U + (h << 6) + (h >> 2)));
}
// uniform in (0,1): 23 random bits + 0.5, exactly representable in fp32
__host__ __device__ __forceinline__ float vc_uniform(uint32_t r) { return ((float)(r >> 9) + 0.5f) * (1.0f / 8388608.0f); }
