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

// Attention dropout (BertSelfAttention in training, modeling_bert.py:330-333): decision for the probability of query
// q and key k (both < 1024, row indices in the [visual | caption] layout) of one (layer, image, head) stream.
// stream = vc_mix(vc_mix(layer_seed, image), head);  keep iff hash >= thr,  thr = p * 2^32.
__host__ __device__ __forceinline__ uint32_t vc_drop_stream(uint32_t layer_seed, uint32_t b, uint32_t h) {
  return vc_mix(vc_mix(layer_seed, b), h);
}
__host__ __device__ __forceinline__ bool vc_drop_keep(uint32_t stream, uint32_t q, uint32_t k, uint32_t thr) {
  return vc_lowbias32(stream ^ (q << 10) ^ k) >= thr;
}
