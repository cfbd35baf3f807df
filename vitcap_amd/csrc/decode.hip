// Greedy bookkeeping and tag top-k: row-wise scans over the 30522-wide vocabulary (HBM/L2-bound).
#include "common.h"

namespace {

struct ArgMax {
  float v;
  int i;
};
__device__ __forceinline__ ArgMax am_better(ArgMax a, ArgMax b) {
  // larger value wins; on ties the LOWER index wins (torch.argmax / topk behaviour on CPU)
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ ArgMax wave_argmax(ArgMax a) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ArgMax b;
    b.v = __shfl_xor(a.v, o, 64);
    b.i = __shfl_xor(a.i, o, 64);
    a = am_better(a, b);
  }
  return a;
}

__global__ void greedy_init_kernel(int64_t* ids, int32_t* unf, float* sum_lp, float* cnt, int B, int max_len, int bos,
                                   int pad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * max_len) ids[i] = (i % max_len == 0) ? bos : pad;
  if (i < B) {
    unf[i] = 1;
    sum_lp[i] = 0.f;
    cnt[i] = 0.f;
  }
}

// one 1024-thread workgroup per sequence
__global__ __launch_bounds__(1024) void greedy_step_kernel(const float* __restrict__ logits, int ldl, int V,
                                                           int64_t* __restrict__ ids, int32_t* __restrict__ unf,
                                                           float* __restrict__ sum_lp, float* __restrict__ cnt,
                                                           float* __restrict__ logprob_out,
                                                           float* __restrict__ margin_out, int t, int max_len, int eos,
                                                           int pad) {
  __shared__ ArgMax s_am[16];
  __shared__ float s_second[16];
  __shared__ float s_sum[16];
  __shared__ ArgMax s_best;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* row = logits + (size_t)b * ldl;
  ArgMax best{-INFINITY, 0x7fffffff};
  float second = -INFINITY;   // runner-up value (for the top-2 margin tap)
  for (int i = tid; i < V; i += 1024) {
    const float v = row[i];
    if (v > best.v || (v == best.v && i < best.i)) {
      second = fmaxf(second, best.v);
      best.v = v;
      best.i = i;
    } else {
      second = fmaxf(second, v);
    }
  }
  // wave reduce keeping the runner-up
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ArgMax ob;
    ob.v = __shfl_xor(best.v, o, 64);
    ob.i = __shfl_xor(best.i, o, 64);
    const float os = __shfl_xor(second, o, 64);
    const ArgMax nb = am_better(best, ob);
    second = fmaxf(fmaxf(second, os), (nb.i == best.i) ? ob.v : best.v);
    best = nb;
  }
  if (lane == 0) { s_am[w] = best; s_second[w] = second; }
  __syncthreads();
  if (tid == 0) {
    ArgMax bb = s_am[0];
    float ss = s_second[0];
    for (int k = 1; k < 16; ++k) {
      const ArgMax nb = am_better(bb, s_am[k]);
      ss = fmaxf(fmaxf(ss, s_second[k]), (nb.i == bb.i) ? s_am[k].v : bb.v);
      bb = nb;
    }
    s_best = bb;
    if (margin_out) margin_out[(size_t)b * max_len + t] = bb.v - ss;
  }
  __syncthreads();
  const ArgMax bb = s_best;
  float se = 0.f;
  for (int i = tid; i < V; i += 1024) se += expf(row[i] - bb.v);
  se = wave_sum(se);
  if (lane == 0) s_sum[w] = se;
  __syncthreads();
  if (tid == 0) {
    float tot = 0.f;
    for (int k = 0; k < 16; ++k) tot += s_sum[k];
    const float lp = -logf(tot);                    // logit[tok] - max - log(sum exp(x - max)), tok is the max
    const int u = unf[b];
    const int add = u ? bb.i : pad;
    float s = sum_lp[b] + lp * (float)u;
    float c = cnt[b] + (float)u;
    int nu = u * (add != eos ? 1 : 0);
    int64_t outtok = add;
    if (t == max_len - 1) {
      if (nu) outtok = eos;                          // modeling_utils.py:870-871
      logprob_out[b] = s / c;                        // modeling_utils.py:873-877
    }
    ids[(size_t)b * max_len + t] = outtok;
    sum_lp[b] = s;
    cnt[b] = c;
    unf[b] = nu;
  }
}

// sigmoid + top-k (k <= 64), one 1024-thread workgroup per row; each thread keeps up to 32 candidates.
constexpr int TK_PER_THREAD = 32;
__global__ __launch_bounds__(1024) void sigmoid_topk_kernel(const float* __restrict__ logits, int ldl, int V, int k,
                                                            float thresh, int64_t* __restrict__ out_ids,
                                                            float* __restrict__ out_prob,
                                                            int64_t* __restrict__ out_len) {
  __shared__ ArgMax s_am[16];
  __shared__ ArgMax s_best;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* row = logits + (size_t)b * ldl;
  float pv[TK_PER_THREAD];
#pragma unroll
  for (int j = 0; j < TK_PER_THREAD; ++j) {
    const int i = tid + j * 1024;
    pv[j] = i < V ? 1.0f / (1.0f + expf(-row[i])) : -1.0f;   // sigmoid in (0,1); -1 = absent/taken
  }
  int nlen = 0;
  for (int r = 0; r < k; ++r) {
    ArgMax best{-2.0f, 0x7fffffff};
#pragma unroll
    for (int j = 0; j < TK_PER_THREAD; ++j) {
      ArgMax c{pv[j], tid + j * 1024};
      best = am_better(best, c);
    }
    best = wave_argmax(best);
    if (lane == 0) s_am[w] = best;
    __syncthreads();
    if (tid == 0) {
      ArgMax bb = s_am[0];
      for (int q = 1; q < 16; ++q) bb = am_better(bb, s_am[q]);
      s_best = bb;
      out_ids[(size_t)b * k + r] = bb.i;
      out_prob[(size_t)b * k + r] = bb.v;
      if (bb.v >= thresh) ++nlen;
    }
    __syncthreads();
    const int wi = s_best.i;
    if ((wi & 1023) == tid) {
      const int jj = wi >> 10;
#pragma unroll
      for (int j = 0; j < TK_PER_THREAD; ++j)
        if (j == jj) pv[j] = -1.0f;
    }
  }
  if (tid == 0) out_len[b] = nlen;
}

}  // namespace

extern "C" int vitcap_greedy_init(int64_t* ids, int32_t* unfinished, float* sum_lp, float* cnt, int B, int max_len,
                                  int bos, int pad, void* stream) {
  VC_REQUIRE(ids && unfinished && sum_lp && cnt && B > 0 && max_len > 1, "greedy_init: bad arguments");
  const int n = B * max_len;
  hipLaunchKernelGGL(greedy_init_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ids, unfinished,
                     sum_lp, cnt, B, max_len, bos, pad);
  VC_LAUNCH_CHECK("greedy_init");
  return VITCAP_OK;
}

extern "C" int vitcap_greedy_step(const float* logits, int ldl, int V, int64_t* ids, int32_t* unfinished,
                                  float* sum_lp, float* cnt, float* logprob_out, float* margin_out, int B, int t,
                                  int max_len, int eos, int pad, void* stream) {
  VC_REQUIRE(logits && ids && unfinished && sum_lp && cnt && logprob_out, "greedy_step: null pointer");
  VC_REQUIRE(B > 0 && V > 0 && ldl >= V && t >= 1 && t < max_len, "greedy_step: bad sizes (t=%d)", t);
  hipLaunchKernelGGL(greedy_step_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, logits, ldl, V, ids, unfinished,
                     sum_lp, cnt, logprob_out, margin_out, t, max_len, eos, pad);
  VC_LAUNCH_CHECK("greedy_step");
  return VITCAP_OK;
}

extern "C" int vitcap_sigmoid_topk(const float* logits, int ldl, int V, int k, float thresh, int64_t* out_ids,
                                   float* out_prob, int64_t* out_len, int B, void* stream) {
  VC_REQUIRE(logits && out_ids && out_prob && out_len && B > 0, "sigmoid_topk: bad arguments");
  VC_REQUIRE(k >= 1 && k <= 64 && V <= TK_PER_THREAD * 1024 && ldl >= V, "sigmoid_topk: k=%d V=%d unsupported", k, V);
  hipLaunchKernelGGL(sigmoid_topk_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, logits, ldl, V, k, thresh,
                     out_ids, out_prob, out_len);
  VC_LAUNCH_CHECK("sigmoid_topk");
  return VITCAP_OK;
}
