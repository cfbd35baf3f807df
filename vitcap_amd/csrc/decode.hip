// Greedy bookkeeping and tag top-k: row-wise scans over the 30522-wide vocabulary (HBM/L2-bound).
#include "common.h"
#include "rng.h"

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
                                   int pad, int32_t* live) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && live) *live = B;            // sequences still unfinished (decode kernels return at entry once it is 0)
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
                                                           float* __restrict__ margin_out, int64_t* __restrict__ raw_last,
                                                           int t, int max_len, int eos, int pad, int32_t* __restrict__ live,
                                                           VcEosExtra eos_x) {
  __shared__ ArgMax s_am[16];
  __shared__ float s_second[16];
  __shared__ float s_sum[16];
  __shared__ ArgMax s_best;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (unf[b] == 0) {
    // finished sequence: tokens_to_add = pad, the score is frozen (modeling_utils.py:855-858, 873-877); the logits are not
    // read -- once every sequence has finished the step's other kernels have not even produced them
    if (tid == 0) {
      ids[(size_t)b * max_len + t] = pad;
      if (t == max_len - 1) {
        if (raw_last) raw_last[b] = pad;
        logprob_out[b] = sum_lp[b] / cnt[b];
      }
    }
    return;
  }
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
    int nu = u * (vc_is_eos(add, eos, eos_x) ? 0 : 1);
    int64_t outtok = add;
    if (t == max_len - 1) {
      if (raw_last) raw_last[b] = add;               // the token actually chosen, before the forced [SEP]
      if (nu) outtok = eos;                          // modeling_utils.py:870-871
      logprob_out[b] = s / c;                        // modeling_utils.py:873-877
    }
    ids[(size_t)b * max_len + t] = outtok;
    sum_lp[b] = s;
    cnt[b] = c;
    unf[b] = nu;
    if (live && u && !nu) atomicSub(live, 1);
  }
}


// ---- sampling step (do_sample): temperature, top-k / top-p filtering, one draw per sequence ----------------------
// One 1024-thread workgroup per sequence; the whole 30522-wide row lives in registers (30 values per thread).
constexpr int SM_NPT = 30;
__device__ __forceinline__ uint32_t order_key(float v) {
  const uint32_t b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);          // unsigned order == float order
}

// MSB-first radix selection over the order keys: returns the largest key `thr` such that the total weight of the
// elements with key > thr is <= T while adding the elements equal to thr would exceed T -- i.e. "keep key >= thr".
// With unit weights and T = k-1 this is the k-th largest value (top-k keeps ties, like `logits < kth` removes);
// with weights = probability mass and T = top_p it is the nucleus boundary (an element is kept iff the mass ranked
// strictly above it is <= top_p, modeling_utils.py:1119-1131).  Integer weights make the sums order-independent.
#define SM_RADIX_SELECT(THR, WEIGHT_EXPR, T)                                                   \
  {                                                                                            \
    uint32_t prefix_ = 0, pmask_ = 0;                                                          \
    unsigned long long acc_ = 0;                                                               \
    for (int pass_ = 0; pass_ < 4; ++pass_) {                                                  \
      const int shift_ = 24 - 8 * pass_;                                                       \
      if (tid < 256) s_hist[tid] = 0;                                                          \
      __syncthreads();                                                                         \
      _Pragma("unroll") for (int j = 0; j < SM_NPT; ++j) {                                     \
        if ((key[j] & pmask_) == prefix_ && key[j] != 0) {                                     \
          const unsigned long long w_ = (WEIGHT_EXPR);                                         \
          if (w_) atomicAdd(&s_hist[(key[j] >> shift_) & 255u], w_);                           \
        }                                                                                      \
      }                                                                                        \
      __syncthreads();                                                                         \
      if (tid == 0) {                                                                          \
        unsigned long long a_ = acc_;                                                          \
        int d_ = 255;                                                                          \
        for (; d_ > 0; --d_) {                                                                 \
          if (a_ + s_hist[d_] > (T)) break;                                                    \
          a_ += s_hist[d_];                                                                    \
        }                                                                                      \
        s_sel = (uint32_t)d_;                                                                  \
        s_acc = a_;                                                                            \
      }                                                                                        \
      __syncthreads();                                                                         \
      prefix_ |= s_sel << shift_;                                                              \
      pmask_ |= 255u << shift_;                                                                \
      acc_ = s_acc;                                                                            \
      __syncthreads();                                                                         \
    }                                                                                          \
    THR = prefix_;                                                                             \
  }

// BEAM = true: the do_sample branch of beam search (modeling_utils.py:966-985).  Same temperature / filter, with
// min_tokens_to_keep = `min_keep` (k = max(top_k, min_keep); ranks 0..min_keep always survive top-p, 1125-1130), then TWO draws
// without replacement (the two largest of x + Gumbel noise == torch.multinomial(p, 2)); written per row: the two words, their
// temperature-scaled logits and the log-sum-exp of the surviving set (so log_softmax(filtered)[word] = value - lse).
template <bool BEAM>
__global__ __launch_bounds__(1024) void sample_step_kernel(const float* __restrict__ logits, int ldl, int V,
                                                           int64_t* __restrict__ ids, int32_t* __restrict__ unf,
                                                           float* __restrict__ sum_lp, float* __restrict__ cnt,
                                                           float* __restrict__ logprob_out,
                                                           float* __restrict__ margin_out, int64_t* __restrict__ raw_last,
                                                           int t, int max_len, int eos, int pad, float temperature, int top_k,
                                                           float top_p, uint32_t seed, int seq_off, int32_t* __restrict__ live,
                                                           int min_keep, float* __restrict__ cand_val,
                                                           int32_t* __restrict__ cand_idx, float* __restrict__ cand_lse,
                                                           VcEosExtra eos_x) {
  __shared__ unsigned long long s_hist[256];
  __shared__ unsigned long long s_acc;
  __shared__ uint32_t s_sel;
  __shared__ float s_f[16];
  __shared__ float s_bcast;
  __shared__ ArgMax s_am[16];
  __shared__ float s_second[16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (BEAM) {
    if (live && *live == 0) return;     // every image's search has closed
  } else if (unf[b] == 0) {             // finished sequence: pad, frozen score (see greedy_step_kernel)
    if (tid == 0) {
      ids[(size_t)b * max_len + t] = pad;
      if (t == max_len - 1) {
        if (raw_last) raw_last[b] = pad;
        logprob_out[b] = sum_lp[b] / cnt[b];
      }
    }
    return;
  }
  const float* row = logits + (size_t)b * ldl;
  float x[SM_NPT];
  uint32_t key[SM_NPT];      // 0 = absent (column >= V)
#pragma unroll
  for (int j = 0; j < SM_NPT; ++j) {
    const int i = tid + j * 1024;
    float v = i < V ? row[i] : -INFINITY;
    if (temperature != 1.0f) v = v / temperature;
    x[j] = v;
    key[j] = i < V ? order_key(v) : 0u;
  }
  // ---- top-k -------------------------------------------------------------------------------------------------
  uint32_t thr = 1u;                                       // keep every present element
  if (top_k > 0) {
    if (top_k < min_keep) top_k = min_keep;
    const unsigned long long kk = (unsigned long long)(top_k < V ? top_k : V) - 1ull;
    uint32_t tk;
    SM_RADIX_SELECT(tk, 1ull, kk);
    thr = tk > thr ? tk : thr;
  }
  // ---- softmax statistics of the surviving set ------------------------------------------------------------------
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < SM_NPT; ++j) if (key[j] >= thr) m = fmaxf(m, x[j]);
  m = wave_max(m);
  if (lane == 0) s_f[w] = m;
  __syncthreads();
  if (tid == 0) { float mm = s_f[0]; for (int k = 1; k < 16; ++k) mm = fmaxf(mm, s_f[k]); s_bcast = mm; }
  __syncthreads();
  m = s_bcast;
  __syncthreads();
  float z = 0.f;
#pragma unroll
  for (int j = 0; j < SM_NPT; ++j) if (key[j] >= thr) z += expf(x[j] - m);
  z = wave_sum(z);
  if (lane == 0) s_f[w] = z;
  __syncthreads();
  if (tid == 0) { float zz = 0.f; for (int k = 0; k < 16; ++k) zz += s_f[k]; s_bcast = zz; }
  __syncthreads();
  z = s_bcast;
  __syncthreads();
  // ---- top-p (nucleus) ---------------------------------------------------------------------------------------
  if (top_p < 1.0f) {
    const float scale = 1099511627776.0f / z;              // 2^40 / Z: probabilities as 40-bit fixed point
    const unsigned long long P = (unsigned long long)((double)top_p * 1099511627776.0);
    uint32_t tp;
    const uint32_t thr_k = thr;
    SM_RADIX_SELECT(tp, (key[j] >= thr_k ? (unsigned long long)(expf(x[j] - m) * scale) : 0ull), P);
    if (min_keep > 1) {                                    // ranks 0..min_keep stay whatever their mass
      uint32_t tm;
      SM_RADIX_SELECT(tm, (key[j] >= thr_k ? 1ull : 0ull), (unsigned long long)min_keep);
      tp = tm < tp ? tm : tp;
    }
    if (tp > thr) {
      thr = tp;
      z = 0.f;                                             // renormalise over the nucleus
#pragma unroll
      for (int j = 0; j < SM_NPT; ++j) if (key[j] >= thr) z += expf(x[j] - m);
      z = wave_sum(z);
      if (lane == 0) s_f[w] = z;
      __syncthreads();
      if (tid == 0) { float zz = 0.f; for (int k = 0; k < 16; ++k) zz += s_f[k]; s_bcast = zz; }
      __syncthreads();
      z = s_bcast;
    }
  }
  const uint32_t hrow = vc_mix(vc_mix(seed, (uint32_t)(b + seq_off)), (uint32_t)t);   // stream of sequence b + seq_off of the call
  if (BEAM) {
    // ---- two draws without replacement: the largest, then the largest of the rest, of x + Gumbel noise ----------------
    float sc[SM_NPT];
#pragma unroll
    for (int j = 0; j < SM_NPT; ++j) {
      const int i = tid + j * 1024;
      sc[j] = key[j] >= thr ? x[j] - logf(-logf(vc_uniform(vc_mix(hrow, (uint32_t)i)))) : -INFINITY;
    }
    int excl = -1;
    for (int d = 0; d < 2; ++d) {
      ArgMax best{-INFINITY, 0x7fffffff};
      float bx = 0.f;
#pragma unroll
      for (int j = 0; j < SM_NPT; ++j) {
        const int i = tid + j * 1024;
        if (key[j] >= thr && i != excl && (sc[j] > best.v || (sc[j] == best.v && i < best.i))) { best.v = sc[j]; best.i = i; bx = x[j]; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        ArgMax ob;
        ob.v = __shfl_xor(best.v, o, 64);
        ob.i = __shfl_xor(best.i, o, 64);
        const float obx = __shfl_xor(bx, o, 64);
        const ArgMax nb = am_better(best, ob);
        bx = (nb.i == best.i) ? bx : obx;
        best = nb;
      }
      __syncthreads();
      if (lane == 0) { s_am[w] = best; s_f[w] = bx; }
      __syncthreads();
      ArgMax bb = s_am[0];
      float xx = s_f[0];
      for (int k = 1; k < 16; ++k) {
        const ArgMax nb = am_better(bb, s_am[k]);
        xx = (nb.i == bb.i) ? xx : s_f[k];
        bb = nb;
      }
      excl = bb.i;
      if (tid == 0) {
        cand_val[(size_t)b * 2 + d] = xx;
        cand_idx[(size_t)b * 2 + d] = bb.i;
      }
    }
    if (tid == 0) cand_lse[b] = m + logf(z);
    return;
  }
  // ---- one draw: argmax(x + Gumbel noise) over the surviving set == multinomial(softmax(filtered)) ----------
  ArgMax best{-INFINITY, 0x7fffffff};
  float bx = 0.f, second = -INFINITY;
#pragma unroll
  for (int j = 0; j < SM_NPT; ++j) {
    const int i = tid + j * 1024;
    if (key[j] >= thr) {
      const float u = vc_uniform(vc_mix(hrow, (uint32_t)i));
      const float sc = x[j] - logf(-logf(u));
      if (sc > best.v || (sc == best.v && i < best.i)) {
        second = fmaxf(second, best.v);
        best.v = sc; best.i = i; bx = x[j];
      } else {
        second = fmaxf(second, sc);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ArgMax ob;
    ob.v = __shfl_xor(best.v, o, 64);
    ob.i = __shfl_xor(best.i, o, 64);
    const float obx = __shfl_xor(bx, o, 64);
    const float os = __shfl_xor(second, o, 64);
    const ArgMax nb = am_better(best, ob);
    second = fmaxf(fmaxf(second, os), (nb.i == best.i) ? ob.v : best.v);
    bx = (nb.i == best.i) ? bx : obx;
    best = nb;
  }
  if (lane == 0) { s_am[w] = best; s_second[w] = second; s_f[w] = bx; }
  __syncthreads();
  if (tid == 0) {
    ArgMax bb = s_am[0];
    float ss = s_second[0], xx = s_f[0];
    for (int k = 1; k < 16; ++k) {
      const ArgMax nb = am_better(bb, s_am[k]);
      ss = fmaxf(fmaxf(ss, s_second[k]), (nb.i == bb.i) ? s_am[k].v : bb.v);
      xx = (nb.i == bb.i) ? xx : s_f[k];
      bb = nb;
    }
    if (margin_out) margin_out[(size_t)b * max_len + t] = bb.v - ss;
    const float lp = xx - m - logf(z);                   // log_softmax of the FILTERED, temperature-scaled logits
    const int u = unf[b];
    const int add = u ? bb.i : pad;
    float s = sum_lp[b] + lp * (float)u;
    float c = cnt[b] + (float)u;
    int nu = u * (vc_is_eos(add, eos, eos_x) ? 0 : 1);
    int64_t outtok = add;
    if (t == max_len - 1) {
      if (raw_last) raw_last[b] = add;               // the token actually chosen, before the forced [SEP]
      if (nu) outtok = eos;
      logprob_out[b] = s / c;
    }
    ids[(size_t)b * max_len + t] = outtok;
    sum_lp[b] = s;
    cnt[b] = c;
    unf[b] = nu;
    if (live && u && !nu) atomicSub(live, 1);
  }
}

// sigmoid + top-k (k <= 64), one 1024-thread workgroup per row; each thread keeps up to 32 candidates as sortable 64-bit keys
// (tk_key below: larger value first, lower index on ties -- torch.topk's CPU order) and its own running maximum, so a round is
// one workgroup-wide max over 1024 cached values behind ONE barrier and a 32-element rescan by the one thread whose candidate was
// taken (k rescans of every thread's 32 values behind two barriers were 122 us per batch of 64 rows).
constexpr int TK_PER_THREAD = 32;
__device__ __forceinline__ unsigned long long tk_key(float f, int i);
__device__ __forceinline__ float tk_val(unsigned long long k);
__device__ __forceinline__ int tk_idx(unsigned long long k);
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long a);
__global__ __launch_bounds__(1024) void sigmoid_topk_kernel(const float* __restrict__ logits, int ldl, int V, int k,
                                                            float thresh, int64_t* __restrict__ out_ids,
                                                            float* __restrict__ out_prob,
                                                            int64_t* __restrict__ out_len) {
  __shared__ unsigned long long s_wmax[2][16];      // double-buffered by round: ONE barrier per round (a wave can be at most one round ahead)
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* row = logits + (size_t)b * ldl;
  unsigned long long key[TK_PER_THREAD];
  unsigned long long tmax = 0ull;                                   // below every real key (sigmoid > 0 maps above 2^63)
#pragma unroll
  for (int j = 0; j < TK_PER_THREAD; ++j) {
    const int i = tid + j * 1024;
    key[j] = i < V ? tk_key(1.0f / (1.0f + expf(-row[i])), i) : 0ull;   // sigmoid in (0,1); 0 = absent / taken
    tmax = key[j] > tmax ? key[j] : tmax;
  }
  int nlen = 0;
  for (int r = 0; r < k; ++r) {
    const unsigned long long wm = wave_max_u64(tmax);
    if (lane == 0) s_wmax[r & 1][w] = wm;
    __syncthreads();
    unsigned long long best = s_wmax[r & 1][0];
#pragma unroll
    for (int q = 1; q < 16; ++q) {
      const unsigned long long c = s_wmax[r & 1][q];
      best = c > best ? c : best;
    }
    if (tid == 0) {
      const float pv = tk_val(best);
      out_ids[(size_t)b * k + r] = tk_idx(best);
      out_prob[(size_t)b * k + r] = pv;
      if (pv >= thresh) ++nlen;
    }
    if (tmax == best) {                 // keys are unique (they carry the index): exactly one thread owns the winner
      tmax = 0ull;
#pragma unroll
      for (int j = 0; j < TK_PER_THREAD; ++j) {
        if (key[j] == best) key[j] = 0ull;
        tmax = key[j] > tmax ? key[j] : tmax;
      }
    }
  }
  if (tid == 0) out_len[b] = nlen;
}


// ------------------------------------------------------------------------------------------------
// Beam search (a13, modeling_utils.py:888-1100).  Per step:
//   row_topk_lse_kernel : per sequence row, logsumexp over the vocabulary and its 2*beams largest logits
//                         (the top 2*beams of a row are the only candidates that row can contribute, because
//                         log_softmax + beam_score is a per-row constant shift);
//   beam_step_kernel    : one thread per image merges beams x 2*beams candidates, sorts them by total score and
//                         replays the reference's python candidate loop + BeamHypotheses (n_hyp = 1) on the device;
//   beam_reorder_kernel : gathers the text K/V cache rows of the chosen parent beams.
// ------------------------------------------------------------------------------------------------
// (value, index) as ONE sortable 64-bit key: high word = the float mapped monotonically onto unsigned, low word = ~index, so
// "larger value first, lower index on ties" (torch.topk on CPU) is a plain unsigned max -- one compare per element instead of
// the two-field comparison (the k rescans of 32 register values per thread are what this kernel spends its time on).
__device__ __forceinline__ unsigned long long tk_key(float f, int i) {
  unsigned u = __float_as_uint(f);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)u << 32) | (unsigned)(~i);
}
__device__ __forceinline__ float tk_val(unsigned long long k) {
  unsigned u = (unsigned)(k >> 32);
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return __uint_as_float(u);
}
__device__ __forceinline__ int tk_idx(unsigned long long k) { return (int)(~(unsigned)k); }
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long a) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)a, o, 64), hi = __shfl_xor((unsigned)(a >> 32), o, 64);
    const unsigned long long b = ((unsigned long long)hi << 32) | lo;
    a = b > a ? b : a;
  }
  return a;
}

__global__ __launch_bounds__(1024) void row_topk_lse_kernel(const float* __restrict__ logits, int ldl, int V, int k,
                                                            float* __restrict__ out_val, int* __restrict__ out_idx,
                                                            float* __restrict__ out_lse, const int32_t* __restrict__ live) {
  VC_LIVE_EXIT(live);
  // Threshold selection -- three scans of the 32 register values per thread instead of 2 per extracted element (k rounds of
  // rescanning + removing made this kernel VALU-bound: 150-210 us for 1280 rows):
  //   1. every thread's maximum; the k largest of the 1024 thread maxima (cheap: one value per thread);
  //   2. T = the k-th of those.  Keys are unique, so exactly k threads own a key >= T and every key of the row's top k is
  //      >= T: all keys >= T (at most 32 per such thread, <= 32 k in total) go to a list in LDS;
  //   3. wave 0 orders the list's k best.
  constexpr int CAP = TK_PER_THREAD * 16;
  __shared__ unsigned long long s_cand[16][16];      // [wave][round], k <= 16
  __shared__ unsigned long long s_list[CAP];
  __shared__ unsigned long long s_thr;
  __shared__ float s_sum[16];
  __shared__ int s_n;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* row = logits + (size_t)b * ldl;
  unsigned long long key[TK_PER_THREAD];
  unsigned long long tmax = 0ull;
#pragma unroll
  for (int j = 0; j < TK_PER_THREAD; ++j) {
    const int i = tid + j * 1024;
    key[j] = i < V ? tk_key(row[i], i) : 0ull;        // 0 = nothing here (below every real key)
    tmax = key[j] > tmax ? key[j] : tmax;
  }
  if (tid == 0) s_n = 0;
  {
    unsigned long long mine = tmax;                    // k best thread maxima of this wave
    for (int r = 0; r < k; ++r) {
      const unsigned long long best = wave_max_u64(mine);
      if (lane == 0) s_cand[w][r] = best;
      if (mine == best) mine = 0ull;
    }
  }
  __syncthreads();
  if (w == 0) {
    unsigned long long c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = lane + u * 64;
      c[u] = idx < 16 * k ? s_cand[idx / k][idx % k] : 0ull;
    }
    unsigned long long best = 0ull;
    for (int r = 0; r < k; ++r) {
      best = c[0] > c[1] ? c[0] : c[1];
      best = c[2] > best ? c[2] : best;
      best = c[3] > best ? c[3] : best;
      best = wave_max_u64(best);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (c[u] == best) c[u] = 0ull;
      if (r == 0 && lane == 0) s_cand[0][0] = best;    // the row maximum (slot reused; everyone has read the candidates)
    }
    if (lane == 0) s_thr = best;                       // k-th largest thread maximum (0 when the row has fewer than k values)
  }
  __syncthreads();
  const unsigned long long thr = s_thr;
  const float rowmax = tk_val(s_cand[0][0]);
  float se = 0.f;                                      // logsumexp with the row max
#pragma unroll
  for (int j = 0; j < TK_PER_THREAD; ++j) {
    if (tid + j * 1024 < V) se += expf(tk_val(key[j]) - rowmax);
    if (key[j] != 0ull && key[j] >= thr) {
      const int pos = atomicAdd(&s_n, 1);
      if (pos < CAP) s_list[pos] = key[j];
    }
  }
  se = wave_sum(se);
  if (lane == 0) s_sum[w] = se;
  __syncthreads();
  if (w == 0) {
    const int n = s_n < CAP ? s_n : CAP;
    unsigned long long c[CAP / 64];
#pragma unroll
    for (int u = 0; u < CAP / 64; ++u) {
      const int idx = lane + u * 64;
      c[u] = idx < n ? s_list[idx] : 0ull;
    }
    for (int r = 0; r < k; ++r) {
      unsigned long long best = 0ull;
#pragma unroll
      for (int u = 0; u < CAP / 64; ++u) best = c[u] > best ? c[u] : best;
      best = wave_max_u64(best);
      if (lane == 0) {
        out_val[(size_t)b * k + r] = tk_val(best);
        out_idx[(size_t)b * k + r] = tk_idx(best);
      }
#pragma unroll
      for (int u = 0; u < CAP / 64; ++u)
        if (c[u] == best) c[u] = 0ull;
    }
    if (lane == 0) {
      float tot = 0.f;
      for (int q = 0; q < 16; ++q) tot += s_sum[q];
      out_lse[b] = rowmax + logf(tot);
    }
  }
}

// The same candidates from the vocabulary GEMM's row statistics (vitcap_gemm_desc.rowstat: per row and 32-column piece the
// maximum, its column and sum exp(x - max)) instead of from the 30522-wide rows.  With T = the k-th largest (piece maximum,
// column) key, every element of the row's top k has a key >= T and therefore lives in one of the k pieces whose maximum key
// is >= T (those k maxima are k distinct elements, so the k-th largest element is >= T): only k x 32 logits are read back.
// logsumexp = rowmax + log(sum_p se_p * exp(max_p - rowmax)).  One 256-thread workgroup per row.
__global__ __launch_bounds__(256) void row_topk_pieces_kernel(const float* __restrict__ logits, int ldl, int V,
                                                              const float* __restrict__ rowstat, int pieces, int k,
                                                              float* __restrict__ out_val, int* __restrict__ out_idx,
                                                              float* __restrict__ out_lse, const int32_t* __restrict__ live) {
  VC_LIVE_EXIT(live);
  constexpr int PPT = 4;                               // pieces per thread (956 pieces / 256 threads)
  __shared__ unsigned long long s_cand[4][16];
  __shared__ unsigned long long s_sel[16];             // the k largest piece keys, descending
  __shared__ unsigned long long s_list[512];           // k <= 16 pieces x 32 columns
  __shared__ float s_sum[4];
  __shared__ int s_n;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const f32x4* rs = (const f32x4*)rowstat + (size_t)b * pieces;
  unsigned long long key[PPT], mine[PPT];
  float pm[PPT], ps[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int i = tid + j * 256;
    if (i < pieces) {
      const f32x4 r = rs[i];
      key[j] = tk_key(r[0], __float_as_int(r[1]));
      pm[j] = r[0];
      ps[j] = r[2];
    } else {
      key[j] = 0ull; pm[j] = -INFINITY; ps[j] = 0.f;
    }
    mine[j] = key[j];
  }
  if (tid == 0) s_n = 0;
  for (int r = 0; r < k; ++r) {                        // the k best piece keys of this wave
    unsigned long long best = mine[0];
#pragma unroll
    for (int j = 1; j < PPT; ++j) best = mine[j] > best ? mine[j] : best;
    best = wave_max_u64(best);
    if (lane == 0) s_cand[w][r] = best;
#pragma unroll
    for (int j = 0; j < PPT; ++j)
      if (mine[j] == best) mine[j] = 0ull;
  }
  __syncthreads();
  if (w == 0) {
    unsigned long long c = lane < 4 * k ? s_cand[lane / k][lane % k] : 0ull;
    for (int r = 0; r < k; ++r) {
      const unsigned long long best = wave_max_u64(c);
      if (lane == 0) s_sel[r] = best;
      if (c == best) c = 0ull;
    }
  }
  __syncthreads();
  const unsigned long long thr = s_sel[k - 1];
  const float rowmax = tk_val(s_sel[0]);
  float se = 0.f;
#pragma unroll
  for (int j = 0; j < PPT; ++j)
    if (ps[j] > 0.f) se += ps[j] * expf(pm[j] - rowmax);
  se = wave_sum(se);
  if (lane == 0) s_sum[w] = se;
  // the k selected pieces' 32 columns each: keep what is >= the threshold key
  for (int e = tid; e < k * 32; e += 256) {
    const unsigned long long sk = s_sel[e >> 5];
    if (sk == 0ull) continue;
    const int col = ((tk_idx(sk) >> 5) << 5) + (e & 31);
    if (col < V) {
      const unsigned long long kk = tk_key(logits[(size_t)b * ldl + col], col);
      if (kk >= thr) s_list[atomicAdd(&s_n, 1)] = kk;
    }
  }
  __syncthreads();
  if (w == 0) {
    const int n = s_n;
    unsigned long long c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = lane + u * 64;
      c[u] = idx < n ? s_list[idx] : 0ull;
    }
    for (int r = 0; r < k; ++r) {
      unsigned long long best = 0ull;
#pragma unroll
      for (int u = 0; u < 8; ++u) best = c[u] > best ? c[u] : best;
      best = wave_max_u64(best);
      if (lane == 0) {
        out_val[(size_t)b * k + r] = tk_val(best);
        out_idx[(size_t)b * k + r] = tk_idx(best);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (c[u] == best) c[u] = 0ull;
    }
    if (lane == 0) out_lse[b] = rowmax + logf((s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]));
  }
}

constexpr int MAXBEAM = 8;

struct BeamState {
  int64_t* ids_in;      // [B*K][max_len]
  int64_t* ids_out;     // [B*K][max_len]
  float* beam_scores;   // [B*K]
  int* parent;          // [B*K]
  int* done;            // [B]
  int* has_hyp;         // [B]               number of hypotheses kept (0..n_keep)
  float* hyp_score;     // [B][n_keep]       in insertion order, like BeamHypotheses.hyp
  int* hyp_len;         // [B][n_keep]
  int64_t* hyp_tok;     // [B][n_keep][max_len]
  int n_keep;           // BeamHypotheses.n_hyp = num_keep_best
};

__global__ void beam_init_kernel(BeamState st, int B, int K, int max_len, int bos, int pad, int32_t* live) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && live) *live = B;     // images whose search is still open
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * K * max_len) {
    st.ids_in[i] = (i % max_len == 0) ? bos : pad;
    st.ids_out[i] = (i % max_len == 0) ? bos : pad;
  }
  if (i < B * K) {
    st.beam_scores[i] = (i % K == 0) ? 0.f : -1e9f;
    st.parent[i] = i;
  }
  if (i < B) {
    st.done[i] = 0;
    st.has_hyp[i] = 0;
  }
  if (i < B * st.n_keep) {
    st.hyp_score[i] = -1e30f;
    st.hyp_len[i] = 0;
  }
  if (i < B * st.n_keep * max_len) st.hyp_tok[i] = pad;
}

// One WAVE per image (before: one thread per image, ~150 dependent global accesses in series = 108 us per step at 256 images).
//   1. lanes load the beams x 2*beams candidates (score = value - lse + beam score, flat index = beam * V + word) into LDS;
//   2. every candidate computes its RANK under (score desc, flat index asc) against all others: rank r < 2*beams = position r
//      of the reference's sorted topk;
//   3. lane 0 replays the reference's candidate loop + BeamHypotheses on metadata only (scores, lengths, and for every kept
//      hypothesis WHERE its tokens come from: an old slot or a beam of this step);
//   4. all lanes materialise the token rows (hypotheses first read into registers, then written: slots may permute) and the
//      new beams' prefixes.
__global__ __launch_bounds__(64) void beam_step_kernel(const float* __restrict__ cval, const int* __restrict__ cidx,
                                                       const float* __restrict__ lse, BeamState st, int B, int K, int V, int t,
                                                       int max_len, int eos, int pad, float length_penalty, int32_t* __restrict__ live,
                                                       int sampled) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= B) return;
  // every image done (`if all(done): break`, modeling_utils.py:1072): beams, hypotheses and scores stay as they are
  if (live && *live == 0) return;
  const int C = 2 * K, n = sampled ? C : K * C, NH = st.n_keep;
  __shared__ float s_sc[MAXBEAM * 2 * MAXBEAM];
  __shared__ int s_fl[MAXBEAM * 2 * MAXBEAM];
  __shared__ float s_top_sc[2 * MAXBEAM];
  __shared__ int s_top_fl[2 * MAXBEAM];
  __shared__ float s_hsc[MAXBEAM];       // kept hypotheses after this step
  __shared__ int s_hln[MAXBEAM];
  __shared__ int s_hsrc[MAXBEAM];        // >= 0: old slot; < 0: -(1 + beam) of this step
  __shared__ int s_nh;
  __shared__ float s_nsc[MAXBEAM];
  __shared__ int s_nword[MAXBEAM], s_npar[MAXBEAM];
  __shared__ float s_best_sc;
  if (sampled) {
    // do_sample (modeling_utils.py:966-985): the candidates are the 2 draws of each beam IN POSITION ORDER, not ranked.
    // Position p holds draw p % 2 of beam p / 2 (score = its log-prob + THAT beam's score), but the reference's "match shape of
    // greedy beam search" step adds `arange(K) * V` repeated twice as beam offsets, so position p is ATTRIBUTED to beam p % K:
    // the prefix continued (or finished) with the word is beam p % K's.  Reproduced as written.
    for (int i = lane; i < n; i += 64) {
      const int row = b * K + (i >> 1);
      s_top_sc[i] = cval[(size_t)row * 2 + (i & 1)] + (st.beam_scores[row] - lse[row]);
      s_top_fl[i] = (i % K) * V + cidx[(size_t)row * 2 + (i & 1)];
    }
    __syncthreads();
    if (lane == 0) {
      float mx = s_top_sc[0];
      for (int i = 1; i < n; ++i) mx = fmaxf(mx, s_top_sc[i]);
      s_best_sc = mx;                                          // is_done(next_scores.max())
    }
  } else {
    for (int i = lane; i < n; i += 64) {
      const int k = i / C, row = b * K + k;
      s_sc[i] = cval[(size_t)row * C + (i - k * C)] + (st.beam_scores[row] - lse[row]);   // log_softmax + beam score
      s_fl[i] = k * V + cidx[(size_t)row * C + (i - k * C)];                              // index into the (beams*V) view
    }
    __syncthreads();
    for (int i = lane; i < n; i += 64) {
      const float v = s_sc[i];
      const int f = s_fl[i];
      int rank = 0;
      for (int j = 0; j < n; ++j) rank += (s_sc[j] > v || (s_sc[j] == v && s_fl[j] < f)) ? 1 : 0;
      if (rank < C) { s_top_sc[rank] = v; s_top_fl[rank] = f; }
      if (rank == 0) s_best_sc = v;
    }
  }
  float* hsc = st.hyp_score + (size_t)b * NH;
  int* hln = st.hyp_len + (size_t)b * NH;
  int64_t* htk = st.hyp_tok + (size_t)b * NH * max_len;
  // old hypothesis rows into registers (NH * max_len <= 8 * 40 = 320 values, <= 5 per lane)
  int64_t old_tok[5];
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int i = lane + u * 64;
    old_tok[u] = i < NH * max_len ? htk[i] : 0;
  }
  __syncthreads();
  if (lane == 0) {
    int nh = st.has_hyp[b];
    for (int i = 0; i < nh; ++i) { s_hsc[i] = hsc[i]; s_hln[i] = hln[i]; s_hsrc[i] = i; }
    int done = st.done[b];
    const int was_done = done;
    if (!done && nh >= NH) {
      // BeamHypotheses.is_done: the list is full and its worst score already beats what the best open beam can reach
      float worst = s_hsc[0];
      for (int i = 1; i < nh; ++i) worst = fminf(worst, s_hsc[i]);
      if (worst >= s_best_sc / powf((float)(max_len - 1), length_penalty)) done = 1;
    }
    int cnt = 0;
    if (!done) {
      const bool last = (t + 1 == max_len);
      for (int r = 0; r < C && cnt < K; ++r) {
        const int beam = s_top_fl[r] / V, word = s_top_fl[r] - beam * V;
        if (word == eos || last) {
          // BeamHypotheses.add (modeling_utils.py:1157-1170): append when the list is not full or the score beats the worst
          // kept one; when that makes n_hyp + 1 entries, delete the lowest (first of equals) and keep the insertion order
          const float hs = s_top_sc[r] / powf((float)t, length_penalty);     // len(hyp) == cur_len == t
          int slot = -1;
          if (nh < NH) {
            slot = nh++;
          } else {
            int lo = 0;
            for (int i = 1; i < nh; ++i)
              if (s_hsc[i] < s_hsc[lo]) lo = i;
            if (hs > s_hsc[lo]) {
              for (int i = lo; i + 1 < nh; ++i) { s_hsc[i] = s_hsc[i + 1]; s_hln[i] = s_hln[i + 1]; s_hsrc[i] = s_hsrc[i + 1]; }
              slot = nh - 1;
            }
          }
          if (slot >= 0) { s_hsc[slot] = hs; s_hln[slot] = t; s_hsrc[slot] = -(1 + beam); }
        } else {
          s_nsc[cnt] = s_top_sc[r];
          s_nword[cnt] = word;
          s_npar[cnt] = b * K + beam;
          ++cnt;
        }
      }
    }
    if (cnt < K) {   // finished image (or the last step): filler beams, ignored from here on
      for (int i = 0; i < K; ++i) { s_nsc[i] = 0.f; s_nword[i] = pad; s_npar[i] = b * K; }
    }
    s_nh = nh;
    if (live && done && !was_done) atomicSub(live, 1);
    st.done[b] = done;
    st.has_hyp[b] = nh;
    for (int i = 0; i < nh; ++i) { hsc[i] = s_hsc[i]; hln[i] = s_hln[i]; }
  }
  __syncthreads();
  // hypothesis token rows: slot i <- old slot s_hsrc[i] (from the registers, via LDS exchange) or beam -(s_hsrc[i]+1)'s prefix
  __shared__ int64_t s_old[MAXBEAM * 40];
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int i = lane + u * 64;
    if (i < NH * max_len && i < MAXBEAM * 40) s_old[i] = old_tok[u];
  }
  __syncthreads();
  const int nh = s_nh;
  for (int i = lane; i < nh * max_len; i += 64) {
    const int slot = i / max_len, q = i - slot * max_len;
    const int src = s_hsrc[slot];
    int64_t v;
    if (src >= 0) v = s_old[src * max_len + q];
    else v = q < t ? st.ids_in[(size_t)(b * K + (-src - 1)) * max_len + q] : (int64_t)pad;
    htk[i] = v;
  }
  // the K beams leaving the step: parent prefix + the chosen word
  for (int i = lane; i < K * (t + 1); i += 64) {
    const int k = i / (t + 1), q = i - k * (t + 1);
    st.ids_out[(size_t)(b * K + k) * max_len + q] = q < t ? st.ids_in[(size_t)s_npar[k] * max_len + q] : (int64_t)s_nword[k];
  }
  if (lane < K) {
    st.beam_scores[b * K + lane] = s_nsc[lane];
    st.parent[b * K + lane] = s_npar[lane];
  }
}

// dst[l][s][0..t) = src[l][parent[s]][0..t)   (rows of 2*768 bf16 = 3 KiB)
__global__ __launch_bounds__(192) void beam_reorder_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                           const int* __restrict__ parent, int NS, int max_len, int t,
                                                           const int32_t* __restrict__ live) {
  VC_LIVE_EXIT(live);
  const int s = blockIdx.x, l = blockIdx.y;
  const size_t rowv = 2 * 768 * 2 / 16;   // 192 uint4 per position
  const uint4* sp = src + ((size_t)l * NS + parent[s]) * max_len * rowv;
  uint4* dp = dst + ((size_t)l * NS + s) * max_len * rowv;
  for (int q = 0; q < t; ++q) dp[q * rowv + threadIdx.x] = sp[q * rowv + threadIdx.x];
}

// decoded[b][j] = the j-th best kept hypothesis + EOS, padded; logprobs -1e5 where fewer than n_keep finished
// (modeling_utils.py:1076-1100: topk over the kept scores, best first; equal scores: earlier insertion first)
__global__ void beam_finalize_kernel(BeamState st, int64_t* out_ids, float* out_lp, int B, int max_len, int eos, int pad) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int NH = st.n_keep, nh = st.has_hyp[b];
  const float* hsc = st.hyp_score + (size_t)b * NH;
  unsigned used = 0;
  for (int j = 0; j < NH; ++j) {
    int best = -1;
    for (int i = 0; i < nh; ++i)
      if (!(used >> i & 1) && (best < 0 || hsc[i] > hsc[best])) best = i;
    int64_t* dst = out_ids + ((size_t)b * NH + j) * max_len;
    if (best < 0) {
      for (int i = 0; i < max_len; ++i) dst[i] = pad;
      out_lp[(size_t)b * NH + j] = -1e5f;
      continue;
    }
    used |= 1u << best;
    const int Lh = st.hyp_len[(size_t)b * NH + best];
    const int64_t* src = st.hyp_tok + ((size_t)b * NH + best) * max_len;
    for (int i = 0; i < max_len; ++i) dst[i] = i < Lh ? src[i] : (i == Lh ? (int64_t)eos : (int64_t)pad);
    out_lp[(size_t)b * NH + j] = hsc[best];
  }
}

// CTRL repetition penalty as generate() applies it (modeling_utils.py:828-836 greedy / sampling, 955-963 beam): for every
// DISTINCT token already in the sequence (`set(input_ids[i].tolist())`, [CLS] included), logit < 0 ? logit * p : logit / p.
// One wave per row; lane j owns prefix position j and acts if no earlier position holds the same token.
__global__ __launch_bounds__(256) void repetition_penalty_kernel(float* __restrict__ logits, int ldl, int V,
                                                                 const int64_t* __restrict__ ids, int ld_ids, int t,
                                                                 float penalty, int rows, const int32_t* __restrict__ live) {
  VC_LIVE_EXIT(live);
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows || lane >= t) return;
  const int64_t* r = ids + (size_t)row * ld_ids;
  const int64_t tok = r[lane];
  for (int j = 0; j < lane; ++j)
    if (r[j] == tok) return;
  if (tok < 0 || tok >= V) return;
  float* x = logits + (size_t)row * ldl + tok;
  const float v = *x;
  *x = v < 0.f ? v * penalty : v / penalty;
}

}  // namespace

extern "C" int vitcap_repetition_penalty(float* logits, int ldl, int V, const int64_t* ids, int ld_ids, int t, float penalty,
                                         int rows, void* stream) {
  VC_REQUIRE(logits && ids && rows > 0 && V > 0 && ldl >= V && t >= 1 && t <= ld_ids && t <= 64 && penalty > 0.f,
             "repetition_penalty: bad arguments (t=%d, penalty=%g)", t, (double)penalty);
  hipLaunchKernelGGL(repetition_penalty_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ldl, V, ids,
                     ld_ids, t, penalty, rows, vc_tls_live);
  VC_LAUNCH_CHECK("repetition_penalty");
  return VITCAP_OK;
}

extern "C" int vitcap_greedy_init(int64_t* ids, int32_t* unfinished, float* sum_lp, float* cnt, int B, int max_len,
                                  int bos, int pad, void* stream) {
  VC_REQUIRE(ids && unfinished && sum_lp && cnt && B > 0 && max_len > 1, "greedy_init: bad arguments");
  const int n = B * max_len;
  hipLaunchKernelGGL(greedy_init_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ids, unfinished,
                     sum_lp, cnt, B, max_len, bos, pad, (int32_t*)vc_tls_live);
  VC_LAUNCH_CHECK("greedy_init");
  return VITCAP_OK;
}

extern "C" int vitcap_greedy_step(const float* logits, int ldl, int V, int64_t* ids, int32_t* unfinished,
                                  float* sum_lp, float* cnt, float* logprob_out, float* margin_out, int64_t* raw_last,
                                  int B, int t, int max_len, int eos, int pad, void* stream) {
  VC_REQUIRE(logits && ids && unfinished && sum_lp && cnt && logprob_out, "greedy_step: null pointer");
  VC_REQUIRE(B > 0 && V > 0 && ldl >= V && t >= 1 && t < max_len, "greedy_step: bad sizes (t=%d)", t);
  hipLaunchKernelGGL(greedy_step_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, logits, ldl, V, ids, unfinished,
                     sum_lp, cnt, logprob_out, margin_out, raw_last, t, max_len, eos, pad, (int32_t*)vc_tls_live, vc_tls_eos_extra);
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


extern "C" int vitcap_row_topk_lse(const float* logits, int ldl, int V, int k, float* out_val, int32_t* out_idx,
                                   float* out_lse, int rows, void* stream) {
  VC_REQUIRE(logits && out_val && out_idx && out_lse && rows > 0, "row_topk_lse: bad arguments");
  VC_REQUIRE(k >= 1 && k <= 16 && V <= TK_PER_THREAD * 1024 && ldl >= V, "row_topk_lse: k=%d V=%d unsupported", k, V);
  hipLaunchKernelGGL(row_topk_lse_kernel, dim3(rows), dim3(1024), 0, (hipStream_t)stream, logits, ldl, V, k, out_val,
                     out_idx, out_lse, vc_tls_live);
  VC_LAUNCH_CHECK("row_topk_lse");
  return VITCAP_OK;
}

extern "C" int vitcap_row_topk_pieces(const float* logits, int ldl, int V, const float* rowstat, int pieces, int k,
                                      float* out_val, int32_t* out_idx, float* out_lse, int rows, void* stream) {
  VC_REQUIRE(logits && rowstat && out_val && out_idx && out_lse && rows > 0, "row_topk_pieces: bad arguments");
  VC_REQUIRE(k >= 1 && k <= 16 && ldl >= V && pieces >= k && pieces <= 1024 && pieces * 32 >= V,
             "row_topk_pieces: k=%d V=%d pieces=%d unsupported", k, V, pieces);
  hipLaunchKernelGGL(row_topk_pieces_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, ldl, V, rowstat, pieces, k,
                     out_val, out_idx, out_lse, vc_tls_live);
  VC_LAUNCH_CHECK("row_topk_pieces");
  return VITCAP_OK;
}

static BeamState make_state(const vitcap_beam_state* s) {
  BeamState st;
  st.ids_in = s->ids_in; st.ids_out = s->ids_out; st.beam_scores = s->beam_scores; st.parent = s->parent;
  st.done = s->done; st.has_hyp = s->has_hyp; st.hyp_score = s->hyp_score; st.hyp_len = s->hyp_len;
  st.hyp_tok = s->hyp_tok;
  st.n_keep = s->n_keep;
  return st;
}

extern "C" int vitcap_beam_init(const vitcap_beam_state* s, int B, int K, int max_len, int bos, int pad, void* stream) {
  VC_REQUIRE(s && B > 0 && K >= 1 && K <= MAXBEAM && max_len > 1, "beam_init: bad arguments (beams <= %d)", MAXBEAM);
  VC_REQUIRE(s->n_keep >= 1 && s->n_keep <= MAXBEAM, "beam_init: n_keep must be 1..%d (got %d)", MAXBEAM, s->n_keep);
  const int n = B * (K > s->n_keep ? K : s->n_keep) * max_len;
  hipLaunchKernelGGL(beam_init_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, make_state(s), B, K,
                     max_len, bos, pad, (int32_t*)vc_tls_live);
  VC_LAUNCH_CHECK("beam_init");
  return VITCAP_OK;
}

extern "C" int vitcap_beam_step(const float* cand_val, const int32_t* cand_idx, const float* lse,
                                const vitcap_beam_state* s, int B, int K, int V, int t, int max_len, int eos, int pad,
                                float length_penalty, void* stream) {
  VC_REQUIRE(cand_val && cand_idx && lse && s && B > 0 && K >= 1 && K <= MAXBEAM, "beam_step: bad arguments");
  VC_REQUIRE(t >= 1 && t < max_len && max_len <= 40, "beam_step: t=%d out of range (max_len %d <= 40)", t, max_len);
  VC_REQUIRE(s->n_keep >= 1 && s->n_keep <= MAXBEAM, "beam_step: n_keep must be 1..%d (got %d)", MAXBEAM, s->n_keep);
  hipLaunchKernelGGL(beam_step_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, cand_val, cand_idx, lse,
                     make_state(s), B, K, V, t, max_len, eos, pad, length_penalty, (int32_t*)vc_tls_live, 0);
  VC_LAUNCH_CHECK("beam_step");
  return VITCAP_OK;
}

// The same step on SAMPLED candidates (vitcap_beam_sample_candidates: cand_val / cand_idx [B*K][2], lse [B*K]): consumed in
// position order with the reference's beam attribution (see beam_step_kernel).
extern "C" int vitcap_beam_step_sampled(const float* cand_val, const int32_t* cand_idx, const float* lse,
                                        const vitcap_beam_state* s, int B, int K, int V, int t, int max_len, int eos, int pad,
                                        float length_penalty, void* stream) {
  VC_REQUIRE(cand_val && cand_idx && lse && s && B > 0 && K >= 2 && K <= MAXBEAM, "beam_step_sampled: bad arguments");
  VC_REQUIRE(t >= 1 && t < max_len && max_len <= 40, "beam_step_sampled: t=%d out of range (max_len %d <= 40)", t, max_len);
  VC_REQUIRE(s->n_keep >= 1 && s->n_keep <= MAXBEAM, "beam_step_sampled: n_keep must be 1..%d (got %d)", MAXBEAM, s->n_keep);
  hipLaunchKernelGGL(beam_step_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, cand_val, cand_idx, lse,
                     make_state(s), B, K, V, t, max_len, eos, pad, length_penalty, (int32_t*)vc_tls_live, 1);
  VC_LAUNCH_CHECK("beam_step_sampled");
  return VITCAP_OK;
}

extern "C" int vitcap_beam_reorder_cache(const void* src, void* dst, const int32_t* parent, int layers, int NS,
                                         int max_len, int t, void* stream) {
  VC_REQUIRE(src && dst && parent && layers > 0 && NS > 0 && t >= 1 && t <= max_len, "beam_reorder: bad arguments");
  hipLaunchKernelGGL(beam_reorder_kernel, dim3(NS, layers), dim3(192), 0, (hipStream_t)stream, (const uint4*)src,
                     (uint4*)dst, parent, NS, max_len, t, vc_tls_live);
  VC_LAUNCH_CHECK("beam_reorder");
  return VITCAP_OK;
}

extern "C" int vitcap_beam_finalize(const vitcap_beam_state* s, int64_t* out_ids, float* out_logprobs, int B,
                                    int max_len, int eos, int pad, void* stream) {
  VC_REQUIRE(s && out_ids && out_logprobs && B > 0, "beam_finalize: bad arguments");
  VC_REQUIRE(s->n_keep >= 1 && s->n_keep <= MAXBEAM, "beam_finalize: n_keep must be 1..%d (got %d)", MAXBEAM, s->n_keep);
  hipLaunchKernelGGL(beam_finalize_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, make_state(s),
                     out_ids, out_logprobs, B, max_len, eos, pad);
  VC_LAUNCH_CHECK("beam_finalize");
  return VITCAP_OK;
}

extern "C" int vitcap_sample_step_offset(const float* logits, int ldl, int V, int64_t* ids, int32_t* unfinished,
                                         float* sum_lp, float* cnt, float* logprob_out, float* margin_out, int64_t* raw_last,
                                         int B, int t, int max_len, int eos, int pad, const vitcap_sample_params* sp, int seq_offset,
                                         void* stream) {
  VC_REQUIRE(logits && ids && unfinished && sum_lp && cnt && logprob_out && sp, "sample_step: null pointer");
  VC_REQUIRE(B > 0 && V > 0 && V <= SM_NPT * 1024 && ldl >= V && t >= 1 && t < max_len && seq_offset >= 0,
             "sample_step: bad sizes (V=%d t=%d)", V, t);
  VC_REQUIRE(sp->temperature > 0.f && sp->top_k >= 0 && sp->top_p > 0.f,
             "sample_step: temperature %g / top_k %d / top_p %g out of range", (double)sp->temperature, sp->top_k,
             (double)sp->top_p);
  hipLaunchKernelGGL(sample_step_kernel<false>, dim3(B), dim3(1024), 0, (hipStream_t)stream, logits, ldl, V, ids, unfinished,
                     sum_lp, cnt, logprob_out, margin_out, raw_last, t, max_len, eos, pad, sp->temperature, sp->top_k,
                     sp->top_p, sp->seed, seq_offset, (int32_t*)vc_tls_live, 1, (float*)nullptr, (int32_t*)nullptr,
                     (float*)nullptr, vc_tls_eos_extra);
  VC_LAUNCH_CHECK("sample_step");
  return VITCAP_OK;
}

extern "C" int vitcap_sample_step(const float* logits, int ldl, int V, int64_t* ids, int32_t* unfinished,
                                  float* sum_lp, float* cnt, float* logprob_out, float* margin_out, int64_t* raw_last,
                                  int B, int t, int max_len, int eos, int pad, const vitcap_sample_params* sp, void* stream) {
  return vitcap_sample_step_offset(logits, ldl, V, ids, unfinished, sum_lp, cnt, logprob_out, margin_out, raw_last, B, t, max_len, eos,
                                   pad, sp, 0, stream);
}

// Candidates of one beam-search step WITH sampling (modeling_utils.py:966-985): per row (= beam) two words drawn without
// replacement from softmax(filter(logits / temperature)), filter with min_tokens_to_keep = 2.  out_val / out_idx: [rows][2],
// out_lse: [rows]; vitcap_beam_step_sampled consumes them.  The noise stream of row r is (seed, r + row_offset, t).
extern "C" int vitcap_beam_sample_candidates(const float* logits, int ldl, int V, int rows, int t,
                                             const vitcap_sample_params* sp, int row_offset, float* out_val, int32_t* out_idx,
                                             float* out_lse, void* stream) {
  VC_REQUIRE(logits && sp && out_val && out_idx && out_lse, "beam_sample_candidates: null pointer");
  VC_REQUIRE(rows > 0 && V > 2 && V <= SM_NPT * 1024 && ldl >= V && t >= 1 && row_offset >= 0,
             "beam_sample_candidates: bad sizes (V=%d t=%d)", V, t);
  VC_REQUIRE(sp->temperature > 0.f && sp->top_k >= 0 && sp->top_p > 0.f,
             "beam_sample_candidates: temperature %g / top_k %d / top_p %g out of range", (double)sp->temperature, sp->top_k,
             (double)sp->top_p);
  hipLaunchKernelGGL(sample_step_kernel<true>, dim3(rows), dim3(1024), 0, (hipStream_t)stream, logits, ldl, V,
                     (int64_t*)nullptr, (int32_t*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                     (int64_t*)nullptr, t, 0, 0, 0, sp->temperature, sp->top_k, sp->top_p, sp->seed, row_offset,
                     (int32_t*)vc_tls_live, 2, out_val, out_idx, out_lse, VcEosExtra{{-1, -1, -1}});
  VC_LAUNCH_CHECK("beam_sample_candidates");
  return VITCAP_OK;
}
