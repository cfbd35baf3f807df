// Constrained beam search bookkeeping on the device: ConstrainedBeamSearch.search as ViTCAP.generate drives it
// (use_hypo = False, no decoding_constraint_flag, no bad_ending_ids; src/tools/captioning/utils_cbs.py:26-374) and
// select_best_beam_with_constraints (:377-443).  An image carries G = S * K sequences: slot (s, k) = beam k of FSM state s.
//
// The reference's step is two nested top-k per target state i: per slot the K best words of the masked log-probabilities, then the K
// best of (those + the slot's running score) over all S*K*K candidates.  Both are plain maxima under one total order -- larger value
// first, lower flat index on ties -- so the slot stage runs one workgroup per (slot, target state) over the 30522 words and the merge
// stage one workgroup per (image, target state) over the G*K survivors; nothing is sorted.  HBM-bound byte/float streaming: the
// slot stage reads its logits row S times (L2 serves the repeats) and every fsm byte of the batch once per step.
#include "common.h"

namespace {

constexpr int CBS_MAXK = 8;          // beams per FSM state (vitcap_gen_opts.num_beams)
constexpr int CBS_MAXS = 32;         // states: 2**3 main states x 4 words per constraint (utils_cbs.py:727-728)
constexpr float CBS_MASKED = -1e20f; // utils_cbs.py:240: a transition the machine does not allow (NOT -inf)

// (value, index) as one unsigned key: larger value first, lower index on ties = plain unsigned max
__device__ __forceinline__ unsigned long long ck_key(float f, int i) {
  unsigned u = __float_as_uint(f);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)u << 32) | (unsigned)(~i);
}
__device__ __forceinline__ float ck_val(unsigned long long k) {
  unsigned u = (unsigned)(k >> 32);
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return __uint_as_float(u);
}
__device__ __forceinline__ int ck_idx(unsigned long long k) { return (int)(~(unsigned)k); }
__device__ __forceinline__ unsigned long long ck_wave_max(unsigned long long a) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)a, o, 64), hi = __shfl_xor((unsigned)(a >> 32), o, 64);
    const unsigned long long b = ((unsigned long long)hi << 32) | lo;
    a = b > a ? b : a;
  }
  return a;
}

struct BadEnding { int32_t id[16]; };

// a thread's K best keys, descending; 0 = empty (below every real key: real keys have a non-zero high word)
struct TopK {
  unsigned long long k[CBS_MAXK];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int j = 0; j < CBS_MAXK; ++j) k[j] = 0ull;
  }
  __device__ __forceinline__ void push(unsigned long long x) {      // always keeps CBS_MAXK entries (static register indices)
    if (x <= k[CBS_MAXK - 1]) return;
#pragma unroll
    for (int j = 0; j < CBS_MAXK; ++j) {            // one bubble pass keeps the array sorted
      if (x > k[j]) {
        const unsigned long long tmp = k[j];
        k[j] = x;
        x = tmp;
      }
    }
  }
  __device__ __forceinline__ void pop() {
#pragma unroll
    for (int j = 0; j + 1 < CBS_MAXK; ++j) k[j] = k[j + 1];
    k[CBS_MAXK - 1] = 0ull;
  }
};

// the K best keys of the workgroup (256 threads), descending, into out[0..K): K rounds of "largest head", the owner pops
__device__ __forceinline__ void block_topk(TopK& mine, int K, unsigned long long* s_w, unsigned long long* out) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int r = 0; r < K; ++r) {
    const unsigned long long wbest = ck_wave_max(mine.k[0]);
    if (lane == 0) s_w[w] = wbest;
    __syncthreads();
    unsigned long long best = s_w[0];
#pragma unroll
    for (int u = 1; u < 4; ++u) best = s_w[u] > best ? s_w[u] : best;
    if (best != 0ull && mine.k[0] == best) mine.pop();     // keys are unique (the index is part of the key)
    if (threadIdx.x == 0) out[r] = best;
    __syncthreads();
  }
}

__global__ void cbs_init_kernel(int64_t* ids_in, int64_t* ids_out, float* sc_in, float* sc_out, int32_t* parent, int32_t* unf,
                                int32_t* n_pred, int32_t* live, int NS, int max_len, int bos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < NS * max_len) {
    const int64_t v = (i % max_len) == 0 ? (int64_t)bos : 0;
    ids_in[i] = v;
    ids_out[i] = v;
  }
  if (i < NS) { sc_in[i] = 0.f; sc_out[i] = 0.f; parent[i] = i; }
  if (i < max_len) unf[i] = 0;
  if (i == 0) { *n_pred = max_len - 1; *live = 1; }
}

// first step: grid (S, B).  Image b reads row b of the logits (as written, utils_cbs.py:134).
__global__ __launch_bounds__(256) void cbs_start_kernel(const float* __restrict__ logits, int ldl, int V, const float* __restrict__ lse,
                                                        const uint8_t* __restrict__ fsm, int64_t* __restrict__ ids_out,
                                                        float* __restrict__ sc_out, int32_t* __restrict__ parent,
                                                        int32_t* __restrict__ unf, int S, int K, int max_len, int eos,
                                                        VcEosExtra ex) {
  __shared__ unsigned long long s_w[4], s_out[CBS_MAXK];
  const int i = blockIdx.x, b = blockIdx.y, G = S * K;
  const float* row = logits + (size_t)b * ldl;
  const float l = lse[b];
  const uint8_t* m = fsm + ((size_t)(b * S + 0) * S + i) * V;
  TopK top;
  top.clear();
  for (int v = threadIdx.x; v < V; v += 256) top.push(ck_key(m[v] ? row[v] - l : -INFINITY, v));
  block_topk(top, K, s_w, s_out);
  if (threadIdx.x < K) {
    const unsigned long long key = s_out[threadIdx.x];
    const int slot = b * G + i * K + threadIdx.x, word = ck_idx(key);
    ids_out[(size_t)slot * max_len + 1] = word;               // column 0 holds BOS since cbs_init
    sc_out[slot] = ck_val(key);
    parent[slot] = slot;
    if (!vc_is_eos(word, eos, ex)) atomicAdd(&unf[1], 1);
  }
}

// once per call: flags[b][s][i] = does ANY word move image b's machine from state s to state i.  Most (s, i) pairs of a machine have
// no transition at all (8 main states: 8 loops + 12 steps out of 64 pairs); their candidates are K fillers of -1e20 whatever the
// logits say, and the slot stage need not scan 30522 words to find that out.
__global__ __launch_bounds__(256) void cbs_pair_flags_kernel(const uint8_t* __restrict__ fsm, int V, uint8_t* __restrict__ flags) {
  __shared__ int s_any;
  const uint8_t* m = fsm + (size_t)blockIdx.x * V;
  if (threadIdx.x == 0) s_any = 0;
  __syncthreads();
  int any = 0;
  for (int v = threadIdx.x; v < V; v += 256) any |= m[v];
  if (any) s_any = 1;
  __syncthreads();
  if (threadIdx.x == 0) flags[blockIdx.x] = (uint8_t)s_any;
}

// later steps, slot stage: grid (S targets, B*G slots)
__global__ __launch_bounds__(256) void cbs_candidates_kernel(const float* __restrict__ logits, int ldl, int V,
                                                             const float* __restrict__ lse, const uint8_t* __restrict__ fsm,
                                                             const int64_t* __restrict__ ids_in, int S, int K, int t, int max_len,
                                                             int eos, VcEosExtra ex, int no_repeat, BadEnding bad,
                                                             float* __restrict__ cand_val, int32_t* __restrict__ cand_word,
                                                             const int32_t* __restrict__ live, const uint8_t* __restrict__ pair_flags) {
  VC_LIVE_EXIT(live);
  __shared__ unsigned long long s_w[4], s_out[CBS_MAXK];
  const int i = blockIdx.x, slot = blockIdx.y, G = S * K;
  const int b = slot / G, s = (slot - b * G) / K;
  if (pair_flags && !pair_flags[(b * S + s) * S + i]) {
    // no word leads from s to i: every candidate is the -1e20 filler, the K lowest word ids under the tie rule -- what the scan below
    // would return
    if (threadIdx.x < K) {
      const size_t o = ((size_t)slot * S + i) * K + threadIdx.x;
      cand_val[o] = CBS_MASKED;
      cand_word[o] = threadIdx.x;
    }
    return;
  }
  const float* row = logits + (size_t)slot * ldl;
  const float l = lse[slot];
  const int lastw = (int)ids_in[(size_t)slot * max_len + t - 1];
  const bool fin = vc_is_eos(lastw, eos, ex);
  bool prev_bad = false;                      // bad_ending_ids: no EOS right behind these words (utils_cbs.py:192-198)
#pragma unroll
  for (int q = 0; q < 16; ++q) prev_bad |= bad.id[q] >= 0 && bad.id[q] == lastw;
  const uint8_t* m = fsm + ((size_t)(b * S + s) * S + i) * V;
  TopK top;
  top.clear();
  for (int v = threadIdx.x; v < V; v += 256) {
    float x;
    if (fin) {
      x = vc_is_eos(v, eos, ex) ? 0.f : -INFINITY;
    } else {
      x = row[v] - l;
      if ((no_repeat && v == lastw) || (prev_bad && vc_is_eos(v, eos, ex))) x = -INFINITY;      // :187-198, before the finished override
    }
    if (!m[v]) x = CBS_MASKED;
    top.push(ck_key(x, v));
  }
  block_topk(top, K, s_w, s_out);
  if (threadIdx.x < K) {
    const unsigned long long key = s_out[threadIdx.x];
    const size_t o = ((size_t)slot * S + i) * K + threadIdx.x;
    cand_val[o] = ck_val(key);
    cand_word[o] = ck_idx(key);
  }
}

// later steps, merge stage: grid (S targets, B images)
__global__ __launch_bounds__(256) void cbs_select_kernel(const float* __restrict__ cand_val, const int32_t* __restrict__ cand_word,
                                                         const int64_t* __restrict__ ids_in, int64_t* __restrict__ ids_out,
                                                         const float* __restrict__ sc_in, float* __restrict__ sc_out,
                                                         int32_t* __restrict__ parent, int32_t* __restrict__ unf, int S, int K,
                                                         int t, int max_len, int eos, VcEosExtra ex,
                                                         const int32_t* __restrict__ live) {
  __shared__ unsigned long long s_w[4], s_out[CBS_MAXK];
  const int i = blockIdx.x, b = blockIdx.y, G = S * K;
  if (*live == 0) {                 // the search has stopped: carry the state over to the buffers the next call reads
    if (threadIdx.x < K) {
      const int slot = b * G + i * K + threadIdx.x;
      for (int p = 0; p < max_len; ++p) ids_out[(size_t)slot * max_len + p] = ids_in[(size_t)slot * max_len + p];
      sc_out[slot] = sc_in[slot];
      parent[slot] = slot;
    }
    return;
  }
  TopK top;
  top.clear();
  const int n = G * K;
  for (int c = threadIdx.x; c < n; c += 256) {
    const int r = c / K, j = c - r * K;
    const float x = cand_val[((size_t)(b * G + r) * S + i) * K + j] + sc_in[b * G + r];      // fp32, as `top + last` (:245-247)
    top.push(ck_key(x, c));
  }
  block_topk(top, K, s_w, s_out);
  if (threadIdx.x < K) {
    const unsigned long long key = s_out[threadIdx.x];
    const int c = ck_idx(key), r = c / K, j = c - r * K;
    const int slot = b * G + i * K + threadIdx.x, src = b * G + r;
    const int word = cand_word[((size_t)src * S + i) * K + j];
    for (int p = 0; p < t; ++p) ids_out[(size_t)slot * max_len + p] = ids_in[(size_t)src * max_len + p];
    ids_out[(size_t)slot * max_len + t] = word;
    sc_out[slot] = ck_val(key);
    parent[slot] = src;
    if (!vc_is_eos(word, eos, ex)) atomicAdd(&unf[t], 1);
  }
}

// after a step's words are out: `if cur_finished.all(): break` (utils_cbs.py:177-181), evaluated for the NEXT iteration
__global__ void cbs_commit_kernel(const int32_t* unf, int32_t* n_pred, int32_t* live, int t) {
  if (*live != 0 && unf[t] == 0) {
    *live = 0;
    *n_pred = t;
  }
}

// grid (B): one wave per image
__global__ __launch_bounds__(64) void cbs_finalize_kernel(const int64_t* __restrict__ ids, const float* __restrict__ sc,
                                                          const int32_t* __restrict__ n_pred, const int64_t* __restrict__ ncons,
                                                          int min_c, int S, int K, int max_len, int eos, VcEosExtra ex, int pad,
                                                          int64_t* __restrict__ out_ids, float* __restrict__ out_lp) {
  const int b = blockIdx.x, lane = threadIdx.x, G = S * K, T = *n_pred;
  int given = (int)ncons[b];
  given = given < 0 ? 0 : (given > 5 ? 5 : given);          // 2**given main states, at most CBS_MAXS = 32 of them
  const int need = given < min_c ? given : min_c;
  const int nmain = 1 << given;
  // lane s < 2**given: the state's best beam (index 0) if the state satisfies enough constraints
  unsigned long long key = 0ull;
  if (lane < nmain && lane < S && __popc(lane) >= need) {
    const int slot = b * G + lane * K;
    int words = 0;
    for (int p = 1; p <= T; ++p) words += vc_is_eos((int)ids[(size_t)slot * max_len + p], eos, ex) ? 0 : 1;
    key = ck_key(sc[slot] / (float)(words + 1), lane);       // torch.argmax: first maximum = lowest state on ties
  }
  const unsigned long long best = ck_wave_max(key);
  if (best == 0ull) {               // no main state of this machine can satisfy the request (2**given > S): the reference indexes out of range
    for (int p = lane; p < max_len; p += 64) out_ids[(size_t)b * max_len + p] = (int64_t)pad;
    if (lane == 0) out_lp[b] = -INFINITY;
    return;
  }
  const int slot = b * G + ck_idx(best) * K;
  for (int p = lane; p < max_len; p += 64)
    out_ids[(size_t)b * max_len + p] = p < T ? ids[(size_t)slot * max_len + 1 + p] : (int64_t)pad;
  if (lane == 0) out_lp[b] = ck_val(best);
}

BadEnding make_bad(const int32_t* e) {
  BadEnding x;
  for (int i = 0; i < 16; ++i) x.id[i] = e ? e[i] : -1;
  return x;
}

VcEosExtra make_extra(const int32_t* e) {
  VcEosExtra x{{-1, -1, -1}};
  if (e)
    for (int i = 0; i < 3; ++i) x.id[i] = e[i];
  return x;
}

bool shape_ok(int B, int S, int K, int max_len) {
  return B > 0 && S >= 1 && S <= CBS_MAXS && K >= 1 && K <= CBS_MAXK && max_len >= 2 && max_len <= VITCAP_MAXLEN_CAP;
}

}  // namespace

extern "C" int vitcap_cbs_init(const vitcap_cbs_state* s, int B, int S, int K, int max_len, int bos, void* stream) {
  VC_REQUIRE(s && s->ids_in && s->ids_out && s->scores_in && s->scores_out && s->parent && s->unfinished && s->n_pred && s->live,
             "cbs_init: null state");
  VC_REQUIRE(shape_ok(B, S, K, max_len), "cbs_init: B=%d S=%d (1..%d) K=%d (1..%d) max_len=%d out of range", B, S, CBS_MAXS, K,
             CBS_MAXK, max_len);
  const int NS = B * S * K, n = NS * max_len;
  hipLaunchKernelGGL(cbs_init_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, s->ids_in, s->ids_out, s->scores_in,
                     s->scores_out, s->parent, s->unfinished, s->n_pred, s->live, NS, max_len, bos);
  VC_LAUNCH_CHECK("cbs_init");
  return VITCAP_OK;
}

extern "C" int vitcap_cbs_start(const float* logits, int ldl, int V, const float* lse, const uint8_t* fsm, const vitcap_cbs_state* s,
                                int B, int S, int K, int max_len, int eos, const int32_t* eos_extra, void* stream) {
  VC_REQUIRE(logits && lse && fsm && s && shape_ok(B, S, K, max_len) && V >= K && ldl >= V, "cbs_start: bad arguments");
  hipLaunchKernelGGL(cbs_start_kernel, dim3(S, B), dim3(256), 0, (hipStream_t)stream, logits, ldl, V, lse, fsm, s->ids_out,
                     s->scores_out, s->parent, s->unfinished, S, K, max_len, eos, make_extra(eos_extra));
  VC_LAUNCH_CHECK("cbs_start");
  hipLaunchKernelGGL(cbs_commit_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, s->unfinished, s->n_pred, s->live, 1);
  VC_LAUNCH_CHECK("cbs_commit");
  return VITCAP_OK;
}

extern "C" int vitcap_cbs_candidates(const float* logits, int ldl, int V, const float* lse, const uint8_t* fsm,
                                     const vitcap_cbs_state* s, int B, int S, int K, int t, int max_len, int eos,
                                     const int32_t* eos_extra, int no_repeat, const int32_t* bad_ending, const uint8_t* pair_flags,
                                     float* cand_val, int32_t* cand_word, void* stream) {
  VC_REQUIRE(logits && lse && fsm && s && cand_val && cand_word && shape_ok(B, S, K, max_len) && V >= K && ldl >= V,
             "cbs_candidates: bad arguments");
  VC_REQUIRE(t >= 2 && t < max_len, "cbs_candidates: t=%d out of 2..%d", t, max_len - 1);
  hipLaunchKernelGGL(cbs_candidates_kernel, dim3(S, B * S * K), dim3(256), 0, (hipStream_t)stream, logits, ldl, V, lse, fsm,
                     s->ids_in, S, K, t, max_len, eos, make_extra(eos_extra), no_repeat, make_bad(bad_ending), cand_val, cand_word, s->live, pair_flags);
  VC_LAUNCH_CHECK("cbs_candidates");
  return VITCAP_OK;
}

extern "C" int vitcap_cbs_pair_flags(const uint8_t* fsm, int B, int S, int V, uint8_t* flags, void* stream) {
  VC_REQUIRE(fsm && flags && B > 0 && S >= 1 && S <= CBS_MAXS && V > 0, "cbs_pair_flags: bad arguments");
  hipLaunchKernelGGL(cbs_pair_flags_kernel, dim3(B * S * S), dim3(256), 0, (hipStream_t)stream, fsm, V, flags);
  VC_LAUNCH_CHECK("cbs_pair_flags");
  return VITCAP_OK;
}

extern "C" int vitcap_cbs_select(const float* cand_val, const int32_t* cand_word, const vitcap_cbs_state* s, int B, int S, int K,
                                 int t, int max_len, int eos, const int32_t* eos_extra, void* stream) {
  VC_REQUIRE(cand_val && cand_word && s && shape_ok(B, S, K, max_len), "cbs_select: bad arguments");
  VC_REQUIRE(t >= 2 && t < max_len, "cbs_select: t=%d out of 2..%d", t, max_len - 1);
  hipLaunchKernelGGL(cbs_select_kernel, dim3(S, B), dim3(256), 0, (hipStream_t)stream, cand_val, cand_word, s->ids_in, s->ids_out,
                     s->scores_in, s->scores_out, s->parent, s->unfinished, S, K, t, max_len, eos, make_extra(eos_extra), s->live);
  VC_LAUNCH_CHECK("cbs_select");
  hipLaunchKernelGGL(cbs_commit_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, s->unfinished, s->n_pred, s->live, t);
  VC_LAUNCH_CHECK("cbs_commit");
  return VITCAP_OK;
}

extern "C" int vitcap_cbs_finalize(const vitcap_cbs_state* s, const int64_t* num_constraints, int min_constraints, int B, int S,
                                   int K, int max_len, int eos, const int32_t* eos_extra, int pad, int64_t* out_ids,
                                   float* out_logprobs, void* stream) {
  VC_REQUIRE(s && num_constraints && out_ids && out_logprobs && shape_ok(B, S, K, max_len), "cbs_finalize: bad arguments");
  VC_REQUIRE(min_constraints >= 0, "cbs_finalize: min_constraints_to_satisfy must be >= 0 (got %d)", min_constraints);
  hipLaunchKernelGGL(cbs_finalize_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, s->ids_in, s->scores_in, s->n_pred,
                     num_constraints, min_constraints, S, K, max_len, eos, make_extra(eos_extra), pad, out_ids, out_logprobs);
  VC_LAUNCH_CHECK("cbs_finalize");
  return VITCAP_OK;
}
