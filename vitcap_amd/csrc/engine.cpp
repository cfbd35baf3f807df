// Engine: enqueues the whole greedy-captioning forward of ViTCAP on one HIP stream.
//
//   encode  : patch embed -> 8 shared ViT blocks -> fork -> 4 caption blocks + 4 tag blocks   (a1-a5)
//             tag head (pooler, transform, 30522-way classifier, sigmoid top-50)                (a6)
//   prefill : [tag CLS | 577 visual] rows through the 4 post-LN decoder layers ONCE; the per-layer packed
//             qkv buffers stay resident as the visual K/V cache                                  (a8/a9)
//   decode  : 19 steps x (2 query rows per sequence through 4 layers against the caches, LM head on the
//             [MASK] row, device-side greedy bookkeeping) -- no host synchronisation             (a9-a12)
//
// The reference recomputes encode+prefill at every step (modeling_utils.py:798-867 with past=None,
// SURVEY.md headline 4); the result is the same because visual rows never attend text rows under the
// seq2seq mask (..._bertemb.py:57-85) and text rows attend only earlier text rows (dataset.py:377-390).
// The 50 tag slots of the text segment are attended by nothing and their outputs are discarded
// (SURVEY.md headline 5), so they are not materialised; the tag head itself is still computed and exposed.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stddef.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <utility>
#include <new>

#include "../../include/vitcap_hip.h"

void vitcap_set_error(const char* fmt, ...);
extern thread_local const int32_t* vc_tls_live;   // csrc/common.h: early-exit counter handed to the decode-step launchers
struct VcEosExtra { int32_t id[3]; };              // csrc/common.h: eos_token_ids[1..3] of the call being enqueued (-1 = unused)
extern thread_local VcEosExtra vc_tls_eos_extra;
extern thread_local hipEvent_t vc_tls_kev_start, vc_tls_kev_stop;   // csrc/common.h: kernel-bound timing events (timing runs only)
extern thread_local bool vc_tls_kev_used;
extern thread_local bool vc_tls_walk_rev;       // csrc/common.h: walk direction of the streaming kernels
extern thread_local bool vc_tls_zigzag;

#include <vector>

// Optional per-launch timing of the large-tile GEMM launches (bench.py's live roofline measurement):
// hipEvents are recorded on the SAME stream right before/after each launch; the pool is grown outside
// the timed region by vitcap_engine_timing_begin().
struct GemmTiming {
  hipEvent_t start, stop;     // stream markers right before / after the launch
  hipEvent_t kstart, kstop;   // bound to the kernel dispatch itself (hipExtLaunchKernelGGL): what rocprofv3 --kernel-trace reports
  bool kernel_bound;          // the launcher took kstart / kstop
  int variant;      // act*4 + out_f32*2 + has_res
  double flops;
};

struct GraphEntry {
  int B;
  void* ws;
  vitcap_gen_opts opts;
  hipGraph_t graph;
  hipGraphExec_t exec;
};

struct vitcap_engine {
  vitcap_weights w;
  bool bound = false;
  bool timing = false;
  int timing_stride = 1;      // the large-GEMM launches of every timing_stride-th STEP (encode call) are timed (vitcap_engine_timing_sample)
  long long timing_seen = 0;  // steps since timing_begin
  bool timing_this_step = true;
  // one enqueue at a time per engine: the side stream / fork-join events and the graph cache are shared by all callers
  std::mutex mu;
  // the tag branch of the encoder (4 tag blocks + tag head) runs on this side stream next to caption blocks 8-11
  hipStream_t side = nullptr;
  hipStream_t cap = nullptr;          // the decode loop is CAPTURED on this engine-owned stream (capture executes nothing), so the
                                      // caller's stream may be any stream, the legacy default stream included
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipStream_t dec2 = nullptr;         // second stream of the split decode loop (vitcap_gen_opts.decode_streams = 2)
  hipEvent_t ev_dfork = nullptr, ev_djoin = nullptr;
  hipStream_t part[3] = {nullptr, nullptr, nullptr};   // streams of encoder parts 1..3 (part 0 runs on the caller's stream)
  hipEvent_t ev_pfork = nullptr, ev_pjoin[3] = {nullptr, nullptr, nullptr};
  bool full_last_tag_block = false;   // VITCAP_FULL_TAG_BLOCK=1: compute all 577 rows of tag_blocks[3] (parity taps / measurements)
  bool fork_tag_branch = true;
  std::vector<GemmTiming> pool;
  size_t used = 0;
  std::vector<GraphEntry> graphs;     // captured decode loops (vitcap_gen_opts.use_graph)
};

namespace {

constexpr int D = VITCAP_HID;
constexpr int NV = VITCAP_NVIS;        // 577
constexpr int SV = VITCAP_NVIS + 1;    // 578 decoder visual rows (tag CLS first)
constexpr int VP = VITCAP_VOCAB_PAD;
constexpr int TOPK = 50;
constexpr int JROWS = 640;           // rows per image of the joint [visual | tag] buffer (578 + 50, padded to 5 x 128)
constexpr int SPLIT_AO = 6, SPLIT_FC2 = 12, SPLIT_MAX = 12;   // split-K of the K=768 / K=3072 decode GEMMs with N=768

vitcap_gen_opts default_opts() {
  vitcap_gen_opts o;
  memset(&o, 0, sizeof(o));
  o.abi = VITCAP_ABI_VERSION;
  o.num_beams = 1;
  o.seqs_per_image = 1;
  o.num_keep_best = 1;
  o.max_length = VITCAP_MAXLEN;
  o.bos_token_id = 101; o.eos_token_id = 102; o.pad_token_id = 0; o.mask_token_id = 103;
  o.length_penalty = 1.0f;
  o.repetition_penalty = 1.0f;
  o.sampling.do_sample = 0; o.sampling.temperature = 1.0f; o.sampling.top_k = 0; o.sampling.top_p = 1.0f; o.sampling.seed = 0u;
  o.gemm_mode = VITCAP_GEMM_AUTO;
  o.early_exit = 1;
  o.use_graph = 0;
  o.tag_visible = 0;
  o.tagemb_cls = 1;
  o.decode_streams = 0;
  o.encode_parts = 0;
  o.eos_extra[0] = o.eos_extra[1] = o.eos_extra[2] = -1;
  o.tag_pos0 = VITCAP_MAXLEN;
  o.use_cbs = 0; o.cbs_states = 0; o.min_constraints_to_satisfy = 2; o.cbs_no_repeat = 0;
  for (int i = 0; i < 16; ++i) o.cbs_bad_ending[i] = -1;
  o.fsm = nullptr; o.num_constraints = nullptr;
  return o;
}

int check_opts(const vitcap_gen_opts& o) {
#define OPT_REQ(cond, ...) do { if (!(cond)) { vitcap_set_error(__VA_ARGS__); return VITCAP_EINVAL; } } while (0)
  OPT_REQ(o.abi == VITCAP_ABI_VERSION, "gen_opts: built against ABI %d, this library is ABI %d (use vitcap_gen_opts_init)", o.abi, VITCAP_ABI_VERSION);
  OPT_REQ(o.num_beams >= 1 && o.num_beams <= 8, "gen_opts: num_beams must be 1..8 (got %d)", o.num_beams);
  OPT_REQ(o.seqs_per_image >= 1 && o.seqs_per_image <= 8, "gen_opts: seqs_per_image must be 1..8 (got %d)", o.seqs_per_image);
  OPT_REQ(o.num_keep_best >= 1 && o.num_keep_best <= 8, "gen_opts: num_keep_best must be 1..8 (got %d)", o.num_keep_best);
  OPT_REQ(!(o.num_beams > 1 && o.seqs_per_image > 1), "gen_opts: seqs_per_image > 1 needs num_beams == 1");
  OPT_REQ(!(o.num_beams == 1 && o.num_keep_best > 1), "gen_opts: cannot generate >1 sentences in greedy search (num_keep_best > 1 needs num_beams > 1)");
  OPT_REQ(o.max_length >= 2 && o.max_length <= VITCAP_MAXLEN_CAP, "gen_opts: max_length must be 2..%d (got %d)", VITCAP_MAXLEN_CAP, o.max_length);
  const int32_t toks[4] = {o.bos_token_id, o.eos_token_id, o.pad_token_id, o.mask_token_id};
  for (int i = 0; i < 4; ++i) OPT_REQ(toks[i] >= 0 && toks[i] < VITCAP_VOCAB, "gen_opts: token id %d out of the vocabulary", toks[i]);
  OPT_REQ(o.repetition_penalty > 0.f, "gen_opts: repetition_penalty must be > 0 (got %g)", (double)o.repetition_penalty);
  OPT_REQ(!o.sampling.do_sample || (o.sampling.temperature > 0.f && o.sampling.top_k >= 0 && o.sampling.top_p > 0.f),
          "gen_opts: temperature %g / top_k %d / top_p %g out of range", (double)o.sampling.temperature, o.sampling.top_k, (double)o.sampling.top_p);
  OPT_REQ(o.gemm_mode == VITCAP_GEMM_AUTO || o.gemm_mode == VITCAP_GEMM_TILES, "gen_opts: gemm_mode %d unknown", o.gemm_mode);
  OPT_REQ(o.tag_visible >= 0 && o.tag_visible <= 50, "gen_opts: tag_visible must be 0..50 (got %d)", o.tag_visible);
  OPT_REQ(o.tag_pos0 >= VITCAP_MAXLEN && o.tag_pos0 <= 512 - 50, "gen_opts: tag_pos0 must be %d..462 (got %d)", VITCAP_MAXLEN, o.tag_pos0);
  OPT_REQ(o.tag_visible == 0 || o.max_length == VITCAP_MAXLEN, "gen_opts: tag_visible > 0 needs max_length == %d", VITCAP_MAXLEN);
  OPT_REQ(o.encode_parts >= 0 && o.encode_parts <= 4, "gen_opts: encode_parts must be 0 (auto) .. 4 (got %d)", o.encode_parts);
  OPT_REQ(o.decode_streams >= 0 && o.decode_streams <= 2, "gen_opts: decode_streams must be 0 (auto), 1 or 2 (got %d)", o.decode_streams);
  for (int i = 0; i < 3; ++i) {
    OPT_REQ(o.eos_extra[i] >= -1 && o.eos_extra[i] < VITCAP_VOCAB, "gen_opts: eos_extra[%d] = %d is neither -1 nor a token id", i, o.eos_extra[i]);
    // the reference's own beam search does not survive several EOS ids: more than num_beams of the 2*num_beams candidates can then
    // be EOS words and `assert len(next_sent_beam) == num_beams` fires (modeling_utils.py:1037)
    OPT_REQ(o.eos_extra[i] < 0 || o.num_beams == 1, "gen_opts: several eos_token_ids need num_beams == 1 (the reference's beam search asserts with them)");
  }
  if (o.use_cbs) {
    // ViTCAP.generate(use_cbs=True): utils_cbs.py keeps num_keep_best = 1 and a plain log-softmax (modeling_bert.py:1038-1042)
    OPT_REQ(o.use_cbs == 1, "gen_opts: use_cbs must be 0 or 1 (got %d)", o.use_cbs);
    OPT_REQ(o.fsm && o.num_constraints, "gen_opts: use_cbs needs fsm [B][S][S][%d] uint8 and num_constraints [B] int64 on the device "
            "(the reference reads fsm.shape, modeling_bert.py:952)", VITCAP_VOCAB);
    OPT_REQ(o.cbs_states >= 1 && o.cbs_states <= 32, "gen_opts: cbs_states must be 1..32 (got %d)", o.cbs_states);
    OPT_REQ(o.cbs_states * o.num_beams <= 256, "gen_opts: cbs_states * num_beams must be <= 256 sequences per image (got %d)",
            o.cbs_states * o.num_beams);
    OPT_REQ(o.min_constraints_to_satisfy >= 0, "gen_opts: min_constraints_to_satisfy must be >= 0 (got %d)", o.min_constraints_to_satisfy);
    OPT_REQ(o.num_keep_best == 1, "gen_opts: not supported n_best > 1 for CBS (modeling_bert.py:1038)");
    OPT_REQ(o.seqs_per_image == 1 && !o.sampling.do_sample && o.repetition_penalty == 1.0f,
            "gen_opts: use_cbs runs neither sampling, num_return_sequences > 1 nor the repetition penalty (utils_cbs.py:184-185)");
    OPT_REQ(o.tag_visible == 0, "gen_opts: use_cbs with tag tokens visible to the caption is not built");
    OPT_REQ(o.max_length >= 3, "gen_opts: use_cbs needs max_length >= 3 (got %d)", o.max_length);
    OPT_REQ(o.cbs_no_repeat == 0 || o.cbs_no_repeat == 1, "gen_opts: cbs_no_repeat must be 0 or 1 (got %d)", o.cbs_no_repeat);
    for (int i = 0; i < 16; ++i)
      OPT_REQ(o.cbs_bad_ending[i] >= -1 && o.cbs_bad_ending[i] < VITCAP_VOCAB, "gen_opts: cbs_bad_ending[%d] = %d is neither -1 nor a token id", i, o.cbs_bad_ending[i]);
  }
#undef OPT_REQ
  return VITCAP_OK;
}

constexpr size_t VT_BYTES = (size_t)12 * 64 * 608 * 2;   // one image's transposed visual V rows of one layer (include/vitcap_hip.h: vitcap_attn_beam_vt)

struct Layout {
  size_t off = 0;
  size_t take(size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  }
  // image-sized buffers (B images) first, so their offsets do not depend on the number of decode sequences
  size_t patches, x, x2, xt, h, qkv, mlp, th, tqkv, tmlp, vis_f, vis_b, dqkv[4], da_f, da_b, dtmp;
  size_t pool_in, pooled, tg_f, tg_b, tag_logits, tag_ids, tag_prob, tag_len;
  // sequence-sized buffers (NS = B * seqs_per_image for greedy / sampling, B * beams for beam search)
  size_t xs_f, xs_b, sqkv, sctx, spart, sa_f, sa_b, smlp, tcache, tcache2;
  size_t hd_f, hd_b, logits, rowstat;
  size_t ids, ids2, unf, sum_lp, cnt, margins, logprob, last_tok, live;
  size_t cand_val, cand_idx, lse, beam_scores, parent, done, has_hyp, hyp_score, hyp_len, hyp_tok, fin_ids, fin_lp;
  size_t ln_cnt, ln_cnt_tag;   // row-block ticket counters of the GEMMs that normalise their own rows (main chain / tag branch); (4 B + 16) ints each
  size_t vt[4];      // beam search: per decoder layer the visual V rows transposed per (image, head) for vitcap_attn_decode_beams (0: unused)
  // tag rows visible to the caption (vitcap_gen_opts.tag_visible = n > 0): per embedding branch v in {A, B}
  size_t tagx_f[2], tagx_b[2], tqkv_c[2][4], jqkv, jout, jlse, tg_ctx, tg_sa_f, tg_sa_b, tg_mlp, tg_tmp;
  int NT;
  int L, NS, K;
  int group_k;       // K > 8: the largest divisor of K that is <= 8 (sequences per attention workgroup), 1 if K is a prime above 8
  bool beam;
  bool cbs;          // constrained beam search: K = cbs_states * num_beams sequences per image
  size_t cbs_val, cbs_word, cbs_sc, cbs_sc2, cbs_unf, cbs_npred, cbs_flags;
  // the same layout seen from image i0 on: every image-major buffer of the encoder / prefill advanced by i0 images
  Layout from_image(int i0) const {
    Layout v = *this;
    const size_t i = (size_t)i0;
    v.patches += i * 576 * D * 2;
    v.x += i * NV * D * 4; v.x2 += i * NV * D * 4; v.xt += i * NV * D * 4;
    v.h += i * SV * D * 2; v.qkv += i * NV * 3 * D * 2; v.mlp += i * SV * 4 * D * 2;
    v.th += i * NV * D * 2; v.tqkv += i * NV * 3 * D * 2; v.tmlp += i * NV * 4 * D * 2;
    v.vis_f += i * SV * D * 4; v.vis_b += i * SV * D * 2;
    for (int l = 0; l < 4; ++l) v.dqkv[l] += i * SV * 3 * D * 2;
    v.da_f += i * SV * D * 4; v.da_b += i * SV * D * 2; v.dtmp += i * SV * D * 4;
    for (int l = 0; l < 4; ++l) if (v.vt[l]) v.vt[l] += i * VT_BYTES;
    v.pool_in += i * D * 2; v.pooled += i * D * 2; v.tg_f += i * D * 4; v.tg_b += i * D * 2;
    v.tag_logits += i * VP * 4; v.tag_ids += i * TOPK * 8; v.tag_prob += i * TOPK * 4; v.tag_len += i * 8;
    v.ln_cnt += i * 4 * 4; v.ln_cnt_tag += i * 4 * 4;      // a part of n images owns <= 3.01 n + 1 row blocks of 192+ rows
    return v;
  }
  Layout(int B, const vitcap_gen_opts& o) {
    L = o.max_length;
    cbs = o.use_cbs != 0;
    beam = !cbs && o.num_beams > 1;
    K = cbs ? o.cbs_states * o.num_beams : (beam ? o.num_beams : o.seqs_per_image);
    NS = B * K;
    group_k = 1;
    for (int g = 8; g >= 2; --g)
      if (K % g == 0) { group_k = g; break; }
    const bool two = K > 1 || cbs;                // layouts with several sequences per image carry the second cache / id buffer
    const size_t b = (size_t)B, n = (size_t)NS, l = (size_t)L;
    patches = take(b * 576 * D * 2);
    x = take(b * NV * D * 4);
    x2 = take(b * NV * D * 4);          // caption branch after the fork (blocks 8-11); x keeps the fork state, read by both branches
    xt = take(b * NV * D * 4);
    h = take(b * SV * D * 2);
    qkv = take(b * NV * 3 * D * 2);
    mlp = take(b * SV * 4 * D * 2);
    th = take(b * NV * D * 2);          // tag branch's own LN / qkv / MLP temporaries (it runs concurrently)
    tqkv = take(b * NV * 3 * D * 2);
    tmlp = take(b * NV * 4 * D * 2);
    vis_f = take(b * SV * D * 4);
    vis_b = take(b * SV * D * 2);
    for (int i = 0; i < 4; ++i) dqkv[i] = take(b * SV * 3 * D * 2);
    da_f = take(b * SV * D * 4);
    da_b = take(b * SV * D * 2);
    dtmp = take(b * SV * D * 4);
    pool_in = take(b * D * 2);
    pooled = take(b * D * 2);
    tg_f = take(b * D * 4);
    tg_b = take(b * D * 2);
    tag_logits = take(b * VP * 4);
    tag_ids = take(b * TOPK * 8);
    tag_prob = take(b * TOPK * 4);
    tag_len = take(b * 8);
    ln_cnt = take((b * 4 + 16) * 4);
    ln_cnt_tag = take((b * 4 + 16) * 4);
    xs_f = take(n * 2 * D * 4);
    xs_b = take(n * 2 * D * 2);
    sqkv = take(n * 2 * 3 * D * 2);
    sctx = take(n * 2 * D * 2);
    spart = take((size_t)SPLIT_MAX * n * 2 * D * 4);   // split-K partial slabs of the decode-step GEMMs
    sa_f = take(n * 2 * D * 4);
    sa_b = take(n * 2 * D * 2);
    smlp = take(n * 2 * 4 * D * 2);
    tcache = take(4 * n * l * 2 * D * 2);
    tcache2 = two ? take(4 * n * l * 2 * D * 2) : 0;
    hd_f = take(n * D * 4);
    hd_b = take(n * D * 2);
    logits = take(n * VP * 4);
    rowstat = take(n * (size_t)(2 * (VP / 64)) * 16);    // {max, argmax, sum exp, -} per row and 32-column piece of the logits
    ids = take(n * l * 8);
    ids2 = two ? take(n * l * 8) : 0;
    unf = take(n * 4);
    sum_lp = take(n * 4);
    cnt = take(n * 4);
    margins = take(n * l * 4);
    logprob = take(n * 4);
    last_tok = take(n * 8);
    live = take(256);
    NT = o.tag_visible;
    for (int v = 0; v < 2; ++v) {
      tagx_f[v] = tagx_b[v] = 0;
      for (int i = 0; i < 4; ++i) tqkv_c[v][i] = 0;
    }
    jqkv = jout = jlse = tg_ctx = tg_sa_f = tg_sa_b = tg_mlp = tg_tmp = 0;
    if (NT > 0) {
      const size_t r = b * (size_t)NT;            // tag rows of the batch
      for (int v = 0; v < 2; ++v) {
        tagx_f[v] = take(r * D * 4);
        tagx_b[v] = take(r * D * 2);
        for (int i = 0; i < 4; ++i) tqkv_c[v][i] = take(r * 3 * D * 2);      // the tag rows' packed q|k|v = their K/V cache
      }
      jqkv = take(b * JROWS * 3 * D * 2);         // per image [578 visual K/V | n tag rows] for the tag rows' attention
      jout = take(b * JROWS * D * 2);
      jlse = take(b * 12 * JROWS * 4);
      tg_ctx = take(r * D * 2);
      tg_sa_f = take(r * D * 4);
      tg_sa_b = take(r * D * 2);
      tg_mlp = take(r * 4 * D * 2);
      tg_tmp = take(r * D * 4);
    }
    cand_val = cand_idx = lse = beam_scores = parent = done = has_hyp = hyp_score = hyp_len = hyp_tok = fin_ids = fin_lp = 0;
    for (int i = 0; i < 4; ++i) vt[i] = 0;
    if ((beam || cbs) && NT == 0)
      for (int i = 0; i < 4; ++i) vt[i] = take(b * VT_BYTES);
    cbs_val = cbs_word = cbs_sc = cbs_sc2 = cbs_unf = cbs_npred = cbs_flags = 0;
    if (cbs) {
      const size_t per = (size_t)o.cbs_states * o.num_beams;        // candidates per slot: K words for each of the S target states
      cbs_val = take(n * per * 4);
      cbs_word = take(n * per * 4);
      cbs_sc = take(n * 4);
      cbs_sc2 = take(n * 4);
      cbs_unf = take(l * 4);
      cbs_npred = take(256);
      cbs_flags = take(b * (size_t)o.cbs_states * o.cbs_states);
      lse = take(n * 4);
      cand_val = take(n * 4);              // row maxima next to the log-sum-exp (vitcap_row_topk_lse with k = 1)
      cand_idx = take(n * 4);
      parent = take(n * 4);
      fin_ids = take(b * l * 8);
      fin_lp = take(b * 4);
    }
    if (beam) {
      cand_val = take(n * 16 * 4);
      cand_idx = take(n * 16 * 4);
      lse = take(n * 4);
      beam_scores = take(n * 4);
      parent = take(n * 4);
      done = take(b * 4);
      has_hyp = take(b * 4);
      hyp_score = take(b * 8 * 4);          // up to 8 kept hypotheses per image (num_keep_best)
      hyp_len = take(b * 8 * 4);
      hyp_tok = take(b * 8 * l * 8);
      fin_ids = take(b * 8 * l * 8);
      fin_lp = take(b * 8 * 4);
    }
  }
};

thread_local vitcap_engine* g_cur = nullptr;   // engine whose launches are being enqueued (timing hook)
thread_local int g_gemm_mode = VITCAP_GEMM_AUTO; // vitcap_gen_opts.gemm_mode of the call being enqueued
thread_local bool g_light_decode = false;        // the other stream's decode chain is a plain greedy / sampling loop of <= 512 sequences
thread_local const int32_t* g_live = nullptr;    // live counter handed to the decode-step GEMMs (vitcap_gemm_desc.live)

// sets the per-call context (timing hook, GEMM launch form, early-exit counter) for the duration of one engine call
// Zig-zag walk of the encoder / prefill chain (common.h: vc_tls_walk_rev): `zz()` after every streaming launch flips the direction
// for the next one, so that each kernel starts on the rows its producer wrote last (still in the Infinity Cache).
// VITCAP_ZIGZAG=0 keeps every kernel first-to-last (A/B measurements).
static bool zigzag_on() {
  static const bool on = [] { const char* e = getenv("VITCAP_ZIGZAG"); return e ? atoi(e) != 0 : true; }();
  return on;
}
static inline void zz() {
  if (zigzag_on()) vc_tls_walk_rev = !vc_tls_walk_rev;
}
struct WalkScope {        // the direction never leaks out of an engine call
  WalkScope() { vc_tls_walk_rev = false; vc_tls_zigzag = zigzag_on(); }
  ~WalkScope() { vc_tls_walk_rev = false; vc_tls_zigzag = false; }
};

struct CallScope {
  CallScope(vitcap_engine* e, int gemm_mode, const int32_t* live, const vitcap_gen_opts* o = nullptr) {
    g_cur = e;
    g_gemm_mode = gemm_mode;
    g_live = live;
    vc_tls_live = live;
    g_light_decode = o && o->num_beams <= 1 && o->cbs_states <= 1;
    if (o) vc_tls_eos_extra = VcEosExtra{{o->eos_extra[0], o->eos_extra[1], o->eos_extra[2]}};
  }
  ~CallScope() {
    g_live = nullptr;
    vc_tls_live = nullptr;
    vc_tls_eos_extra = VcEosExtra{{-1, -1, -1}};
    g_gemm_mode = VITCAP_GEMM_AUTO;
    g_light_decode = false;
  }
};

int gemm_desc(const void* A, const void* W, const float* bias, const float* res, void* C, vitcap_gemm_desc d, void* s) {
  vitcap_engine* e = g_cur;
  // the big-tile launches of the encoder / prefill (the decode-step GEMMs of large batches are a different, latency-bound population)
  bool eligible = e && e->timing && d.M >= 2048;
  if (eligible) {      // a launch that is being captured into a hipGraph cannot carry events that are queried afterwards
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) eligible = false;
  }
  const bool timed = eligible && e->timing_this_step && e->used < e->pool.size();
  GemmTiming* t = timed ? &e->pool[e->used++] : nullptr;
  // one tile per workgroup for the large GEMMs when the caller overlaps a second stream (vitcap_gen_opts.gemm_mode)
  if (d.tile_hint == 0 && g_gemm_mode == VITCAP_GEMM_TILES && d.M >= 2048 && d.act != VITCAP_ACT_TANH && d.split_k <= 1) {
    d.tile_hint = 5;
    // Round 6: from 64k rows per launch on (B = 512) the bf16-output GEMMs (qkv, fc1) run the PERSISTENT 4-wave form, whose A-panel
    // prefetch exists for that size class (gemm4w.hip PF) -- +0.4-0.9 % images/s, -1 % joules per step at B = 512
    // (profiles/r06_prefetch_ab_b512.txt) -- but only next to a greedy decode chain: beside the 1 280-sequence chain of beam 5 x 256 the
    // persistent grids cost -3.4 % (3 688 -> 3 562 img/s, profiles/r06_decode_forms_and_beam_ab.txt).  VITCAP_GEMM_4W_MIX_BIG=0 turns it off.
    static const int mix_big = [] { const char* v = getenv("VITCAP_GEMM_4W_MIX_BIG"); return v ? atoi(v) : 1; }();
    if (mix_big && g_light_decode && d.M >= 65536 && d.out_dtype == VITCAP_OUT_BF16 && !res && (d.N & 255) == 0 && d.K >= 192 && d.row_group == 0 &&
        !d.colsum && !d.rowstat)
      d.tile_hint = 42;
  }
  if (d.M <= 4096) d.live = g_live;           // decode-step shapes only; the encoder / prefill GEMMs never carry it
  if (t) {
    t->variant = d.act * 4 + d.out_dtype * 2 + (res ? 1 : 0);
    t->flops = 2.0 * d.M * d.N * d.K;
    (void)hipEventRecord(t->start, (hipStream_t)s);
    vc_tls_kev_start = t->kstart;
    vc_tls_kev_stop = t->kstop;
    vc_tls_kev_used = false;
  }
  const int rc = vitcap_gemm_bias_act(A, W, bias, res, C, &d, s);
  if (t) {
    t->kernel_bound = vc_tls_kev_used;
    vc_tls_kev_start = vc_tls_kev_stop = nullptr;
    (void)hipEventRecord(t->stop, (hipStream_t)s);
  }
  return rc;
}

int gemm(const void* A, int lda, const void* W, const float* bias, const float* res, int ldr, void* C, int ldc, int M,
         int N, int K, int act, int out, void* s) {
  vitcap_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.abi = VITCAP_ABI_VERSION;
  d.M = M; d.N = N; d.K = K;
  d.lda = lda; d.ldw = K; d.ldc = ldc; d.ldr = ldr;
  d.act = act; d.out_dtype = out;
  return gemm_desc(A, W, bias, res, C, d, s);
}

// residual GEMM (N = 768, fp32 out) + LayerNorm of its finished rows -> ln_b (bf16) / ln_f (fp32, optional): one launch where the
// kernel normalises its own rows (vitcap_gemm_desc.ln_*), GEMM + LayerNorm launches otherwise; same bits either way
int gemm_ln(const void* A, int lda, const void* W, const float* bias, const float* res, void* C, int M, int K, const float* g,
            const float* beta, float eps, void* ln_b, float* ln_f, int32_t* cnt, void* s) {
  vitcap_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.abi = VITCAP_ABI_VERSION;
  d.M = M; d.N = D; d.K = K;
  d.lda = lda; d.ldw = K; d.ldc = D; d.ldr = D;
  d.act = VITCAP_ACT_NONE; d.out_dtype = VITCAP_OUT_F32;
  d.ln_gamma = g; d.ln_beta = beta; d.ln_eps = eps;
  // the in-kernel LayerNorm (last arriver of a row block) is built, bit-identical and SLOWER than the separate launch (docs/LAB_r01_r04.md 4.3):
  // the counters are handed over only under VITCAP_GEMM_LN_FUSE=1, otherwise vitcap_gemm_ex launches the LayerNorm kernel behind the GEMM
  static const int fuse = [] { const char* e = getenv("VITCAP_GEMM_LN_FUSE"); return e ? atoi(e) : 0; }();
  d.ln_out_bf16 = ln_b; d.ln_out_f32 = ln_f; d.ln_counters = fuse ? cnt : nullptr;
  return gemm_desc(A, W, bias, res, C, d, s);
}

int gemm_split(const void* A, int lda, const void* W, void* partials, int M, int N, int K, int split, void* s) {
  vitcap_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.abi = VITCAP_ABI_VERSION;
  d.M = M; d.N = N; d.K = K;
  d.lda = lda; d.ldw = K; d.ldc = N;
  d.act = VITCAP_ACT_NONE; d.out_dtype = VITCAP_OUT_F32;
  d.split_k = split;
  return gemm_desc(A, W, nullptr, nullptr, partials, d, s);
}

// Decode-step GEMMs with few rows (M <= 1024).  `hint` 20 / 21 / 22 name the "resident" kernel form (the whole 768-long k range of a
// tile requested at once); see below for which form actually runs.
int gemm_small(const void* A, int lda, const void* W, const float* bias, void* C, int ldc, int M, int N, int K, int act, int out,
               int hint, void* s) {
  vitcap_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.abi = VITCAP_ABI_VERSION;
  d.M = M; d.N = N; d.K = K;
  d.lda = lda; d.ldw = K; d.ldc = ldc;
  d.act = act; d.out_dtype = out;
  d.tile_hint = hint;
  // K = 768: the 4-stage LDS-DMA ring on 64x32 (32x32 for a handful of rows) tiles.  Round 2 used the resident whole-K form here
  // too -- a workaround for the ring's counted waits having silently become vmcnt(0) (docs/LAB_r01_r04.md 4.2 i); with the waits real the
  // ring wins at every batch size: decode phase 5.54 -> 5.28 ms at 64 images, 3.84 -> 3.59 at one, 8.65 -> 7.78 at 128.
  if (K == 768 && hint >= 20 && hint <= 22) d.tile_hint = M <= 32 ? 14 : 13;
  // K = 3072 (`output.dense`): the same ring with one raw fp32 slab per 768-long k range (hints 23 / 24 = the resident form's
  // contract): decode phase 5.24 -> 5.13 ms at 64 images, 3.58 -> 3.48 at one
  if (K > 768 && K % 768 == 0 && hint >= 20 && hint <= 22) d.tile_hint = M <= 32 ? 24 : 23;
  return gemm_desc(A, W, bias, nullptr, C, d, s);
}

#define CK(call)             \
  do {                       \
    int rc_ = (call);        \
    if (rc_ != 0) return rc_; \
  } while (0)

#define HIPCK(call, what)                                                              \
  do {                                                                                 \
    hipError_t he_ = (call);                                                           \
    if (he_ != hipSuccess) {                                                           \
      vitcap_set_error("%s: %s", what, hipGetErrorString(he_));                        \
      return VITCAP_ELAUNCH;                                                           \
    }                                                                                  \
  } while (0)

void drop_graphs(vitcap_engine* e) {
  for (auto& g : e->graphs) {
    (void)hipGraphExecDestroy(g.exec);
    (void)hipGraphDestroy(g.graph);
  }
  e->graphs.clear();
}

}  // namespace


extern "C" void vitcap_gen_opts_init(vitcap_gen_opts* o) {
  if (o) *o = default_opts();
}
extern "C" int vitcap_gen_opts_check(const vitcap_gen_opts* o) {
  if (!o) return VITCAP_OK;
  return check_opts(*o);
}

extern "C" int vitcap_engine_create(vitcap_engine** out) {
  if (!out) return VITCAP_EINVAL;
  *out = new (std::nothrow) vitcap_engine();
  if (*out) {
    const char* f = getenv("VITCAP_TAG_FORK");     // 0: keep the tag branch on the caller's stream
    (*out)->fork_tag_branch = f ? atoi(f) != 0 : true;
    const char* t = getenv("VITCAP_FULL_TAG_BLOCK");
    (*out)->full_last_tag_block = t ? atoi(t) != 0 : false;
  }
  return *out ? VITCAP_OK : VITCAP_EINVAL;
}
// The engine's helper streams are PROCESS-wide, one per role and device, created on first use and never destroyed.  HIP maps streams onto a
// few hardware queues (GPU_MAX_HW_QUEUES, default 4) in creation order, and two chains that share a queue block each other at every
// event wait.  With streams owned by the engine object, every new model of a process (pipeline_eval_multi over several test sets) drew a
// new arrangement: some put the encoder and the decode chain on one queue and the 2-slot pipeline ran at HALF its rate (measured: the
// second and third predict() of a process 1 935 instead of 3 750 images/s, all fine with 8 queues; profiles/r05_hw_queue_aliasing.txt).
// Shared streams add ordering between two engines used at the same time from two threads, never a hazard: every use is fenced by the
// engine's own events.
enum { ROLE_SIDE = 0, ROLE_DEC2 = 1, ROLE_PART0 = 2 /* .. +2 */, ROLE_COUNT = 5 };
static hipStream_t role_stream(int role) {
  static std::mutex mu;
  static hipStream_t pool[64][ROLE_COUNT] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || role < 0 || role >= ROLE_COUNT) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (!pool[dev][role] && hipStreamCreateWithFlags(&pool[dev][role], hipStreamNonBlocking) != hipSuccess) pool[dev][role] = nullptr;
  return pool[dev][role];
}

extern "C" void vitcap_engine_destroy(vitcap_engine* e) {
  if (!e) return;
  for (auto& t : e->pool) {
    (void)hipEventDestroy(t.start);
    (void)hipEventDestroy(t.stop);
  }
  drop_graphs(e);
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->ev_join) (void)hipEventDestroy(e->ev_join);
  if (e->cap) (void)hipStreamDestroy(e->cap);          // side / dec2 / part streams belong to the process (role_stream)
  if (e->ev_dfork) (void)hipEventDestroy(e->ev_dfork);
  if (e->ev_djoin) (void)hipEventDestroy(e->ev_djoin);
  for (int i = 0; i < 3; ++i) {
    if (e->ev_pjoin[i]) (void)hipEventDestroy(e->ev_pjoin[i]);
  }
  if (e->ev_pfork) (void)hipEventDestroy(e->ev_pfork);
  delete e;
}
extern "C" int vitcap_engine_graph_count(vitcap_engine* e) { return e ? (int)e->graphs.size() : 0; }

extern "C" int vitcap_engine_timing_begin(vitcap_engine* e, int max_launches) {
  if (!e || max_launches < 0) return VITCAP_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  while ((int)e->pool.size() < max_launches) {
    GemmTiming t;
    if (hipEventCreate(&t.start) != hipSuccess || hipEventCreate(&t.stop) != hipSuccess ||
        hipEventCreate(&t.kstart) != hipSuccess || hipEventCreate(&t.kstop) != hipSuccess) {
      vitcap_set_error("timing_begin: hipEventCreate failed");
      return VITCAP_ELAUNCH;
    }
    t.variant = 0;
    t.flops = 0;
    t.kernel_bound = false;
    e->pool.push_back(t);
  }
  e->used = 0;
  e->timing_seen = 0;
  e->timing = max_launches > 0;
  return VITCAP_OK;
}

extern "C" int vitcap_engine_timing_sample(vitcap_engine* e, int stride) {
  if (!e || stride < 1) return VITCAP_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  e->timing_stride = stride;
  return VITCAP_OK;
}

// Sums per GEMM epilogue variant (index = act*4 + out_f32*2 + has_res, 12 slots): milliseconds, flops, launches.
// Synchronises on the recorded events; call after the timed region.  busy_ms (optional): per variant, the length of the UNION of
// its launches' [start, stop] intervals -- with several chains in flight (batch pipeline, encoder parts) launches of one kernel
// overlap each other and their summed durations count that time twice.
static double union_ms(std::vector<std::pair<float, float>>& iv) {
  std::sort(iv.begin(), iv.end());
  float end = -1e30f;
  double tot = 0;
  for (auto& p : iv) {
    if (p.first > end) { tot += p.second - p.first; end = p.second; }
    else if (p.second > end) { tot += p.second - end; end = p.second; }
  }
  return tot;
}

// kernel_ms / kernel_busy_ms (optional): the same sums from the events BOUND TO THE KERNEL DISPATCHES (hipExtLaunchKernelGGL start /
// stop events: kernel begins executing -> kernel complete, the interval rocprofv3 --kernel-trace reports).  The stream-marker
// brackets (ms / busy_ms) also hold the time a dispatch waited behind another stream's kernels, so with two streams in flight they
// are longer than the kernel ran.  Launches whose launcher does not take kernel events fall back to their bracket.
extern "C" int vitcap_engine_timing_end_kernel(vitcap_engine* e, double* ms, double* flops, int* launches, double* busy_ms,
                                               double* kernel_ms, double* kernel_busy_ms) {
  if (!e || !ms || !flops || !launches) return VITCAP_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  for (int i = 0; i < 12; ++i) {
    ms[i] = 0; flops[i] = 0; launches[i] = 0;
    if (busy_ms) busy_ms[i] = 0;
    if (kernel_ms) kernel_ms[i] = 0;
    if (kernel_busy_ms) kernel_busy_ms[i] = 0;
  }
  std::vector<std::pair<float, float>> iv[12], kiv[12];
  hipEvent_t korigin = nullptr;     // time axis of the kernel-bound intervals: the first kernel-bound launch's start event
  for (size_t i = 0; i < e->used && !korigin; ++i)
    if (e->pool[i].kernel_bound) korigin = e->pool[i].kstart;
  for (size_t i = 0; i < e->used; ++i) {
    GemmTiming& t = e->pool[i];
    float el = 0.f;
    if (hipEventSynchronize(t.stop) != hipSuccess || hipEventElapsedTime(&el, t.start, t.stop) != hipSuccess) {
      vitcap_set_error("timing_end: event query failed");
      return VITCAP_ELAUNCH;
    }
    ms[t.variant] += el;
    flops[t.variant] += t.flops;
    launches[t.variant] += 1;
    float a = 0.f;
    if ((busy_ms || kernel_busy_ms) && hipEventElapsedTime(&a, e->pool[0].start, t.start) != hipSuccess) a = 0.f;   // relative to the first launch
    if (busy_ms) iv[t.variant].push_back({a, a + el});
    if (kernel_ms || kernel_busy_ms) {
      float kel = el, ka = a;
      if (t.kernel_bound) {
        float x = 0.f, y = 0.f;
        if (hipEventSynchronize(t.kstop) == hipSuccess && hipEventElapsedTime(&x, t.kstart, t.kstop) == hipSuccess &&
            hipEventElapsedTime(&y, korigin, t.kstart) == hipSuccess) {
          kel = x;
          ka = y;
        }
      }
      if (kernel_ms) kernel_ms[t.variant] += kel;
      if (kernel_busy_ms) kiv[t.variant].push_back({ka, ka + kel});
    }
  }
  for (int v = 0; v < 12; ++v) {
    if (busy_ms) busy_ms[v] = union_ms(iv[v]);
    if (kernel_busy_ms) kernel_busy_ms[v] = union_ms(kiv[v]);
  }
  e->timing = false;
  e->used = 0;
  return VITCAP_OK;
}

// Sums per GEMM epilogue variant (index = act*4 + out_f32*2 + has_res, 12 slots): milliseconds, flops, launches.
// Synchronises on the recorded events; call after the timed region.  busy_ms (optional): per variant, the length of the UNION of
// its launches' [start, stop] intervals -- with several chains in flight (batch pipeline, encoder parts) launches of one kernel
// overlap each other and their summed durations count that time twice.
extern "C" int vitcap_engine_timing_end_ex(vitcap_engine* e, double* ms, double* flops, int* launches, double* busy_ms) {
  return vitcap_engine_timing_end_kernel(e, ms, flops, launches, busy_ms, nullptr, nullptr);
}
extern "C" int vitcap_engine_timing_end(vitcap_engine* e, double* ms, double* flops, int* launches) {
  return vitcap_engine_timing_end_ex(e, ms, flops, launches, nullptr);
}

extern "C" int vitcap_engine_bind_weights(vitcap_engine* e, const vitcap_weights* w) {
  if (!e || !w) { vitcap_set_error("bind_weights: null"); return VITCAP_EINVAL; }
  const void* const* p = (const void* const*)w;
  const size_t required = offsetof(vitcap_weights, xword_emb) / sizeof(void*);      // bert.extra_embeddings is optional
  for (size_t i = 0; i < required; ++i)
    if (!p[i]) { vitcap_set_error("bind_weights: pointer #%zu of vitcap_weights is NULL", i); return VITCAP_EINVAL; }
  std::lock_guard<std::mutex> lk(e->mu);
  e->w = *w;
  e->bound = true;
  drop_graphs(e);          // captured loops hold the old weight pointers
  return VITCAP_OK;
}

extern "C" size_t vitcap_engine_workspace_bytes(int B, const vitcap_gen_opts* opts) {
  const vitcap_gen_opts o = opts ? *opts : default_opts();
  if (B <= 0 || check_opts(o) != VITCAP_OK) return 0;
  return Layout(B, o).off;
}

static int check(vitcap_engine* e, int B, const vitcap_gen_opts& o, void* ws, size_t ws_bytes, size_t need) {
  if (!e || !e->bound) { vitcap_set_error("engine: weights not bound"); return VITCAP_ESTATE; }
  if (B <= 0 || !ws) { vitcap_set_error("engine: bad batch/workspace"); return VITCAP_EINVAL; }
  if (((uintptr_t)ws & 255) != 0) { vitcap_set_error("engine: workspace must be 256-byte aligned"); return VITCAP_EINVAL; }
  CK(check_opts(o));
  if (ws_bytes < need) {
    vitcap_set_error("engine: workspace %zu < required %zu bytes", ws_bytes, need);
    return VITCAP_EWORKSPACE;
  }
  return VITCAP_OK;
}

// x_in: the block's input (read by LN1 and as the residual of proj); x: its output buffer, updated in place from proj on.
// x_in != x only at the fork (block 8 and tag block 0 both read the output of block 7 and write their own stream), which
// replaces a 113 MB device-to-device copy of the fork state per batch.
// have_ln1: `h` already holds norm1(x_in) (written by the previous block's fc2, below).  next: the block that consumes this one's
// output on the same chain, or null -- its norm1 then rides in this block's fc2 (-> h); norm2 always rides in proj.  Each fused
// LayerNorm is the separate vitcap_layernorm_fwd launch's arithmetic on the same fp32 rows (vitcap_gemm_desc.ln_*).
static int vit_block(const vitcap_vit_block_w& w, const float* x_in, float* x, void* h, void* qkv, void* mlp, int B, void* s,
                     int32_t* cnt = nullptr, bool have_ln1 = false, const vitcap_vit_block_w* next = nullptr) {
  const int M = B * NV;
  // zz(): each streaming kernel walks the rows the other way round than the one before it (common.h: vc_tls_walk_rev); gemm_ln is two
  // launches (GEMM, then the LayerNorm of its rows) and flips between them itself
  if (!have_ln1) { CK(vitcap_layernorm_fwd(x_in, D, w.n1_g, w.n1_b, 1e-6f, h, nullptr, M, D, s)); zz(); }
  CK(gemm(h, D, w.qkv_w, w.qkv_b, nullptr, 0, qkv, 3 * D, M, 3 * D, D, VITCAP_ACT_NONE, VITCAP_OUT_BF16, s));
  zz();
  CK(vitcap_attn_dense_fwd(qkv, h, B, NV, 0.125f, s));
  zz();
  // proj reads h (the attention output) as A and its norm2 writes h: a row block's A rows are read by its own three tiles only
  CK(gemm_ln(h, D, w.proj_w, w.proj_b, x_in, x, M, D, w.n2_g, w.n2_b, 1e-6f, h, nullptr, cnt, s));      // GEMM, LayerNorm: two flips = none
  CK(gemm(h, D, w.fc1_w, w.fc1_b, nullptr, 0, mlp, 4 * D, M, 4 * D, D, VITCAP_ACT_GELU_ERF, VITCAP_OUT_BF16, s));
  zz();
  if (next) {
    CK(gemm_ln(mlp, 4 * D, w.fc2_w, w.fc2_b, x, x, M, 4 * D, next->n1_g, next->n1_b, 1e-6f, h, nullptr, cnt, s));
  } else {
    CK(gemm(mlp, 4 * D, w.fc2_w, w.fc2_b, x, D, x, D, M, D, 4 * D, VITCAP_ACT_NONE, VITCAP_OUT_F32, s));
    zz();
  }
  return VITCAP_OK;
}

// The LAST tag block: only row 0 (CLS) of its output is ever read -- the pooler takes tag_hidden[:, 0]
// (modeling_bert.py:1424) and the joint sequence takes tag_hidden[:, 0] as its first visual token (1493).  So: LN1 and the
// K/V projections on all 577 rows (the CLS query attends every key), Q / attention / proj / LN2 / MLP for the CLS rows only
// (strided views of the same buffers: row b*577).  Rows 1..576 of `x` keep the previous block's output.
static int vit_block_cls_only(const vitcap_vit_block_w& w, float* x, void* h, void* qkv, void* mlp, void* cls_h, int B, void* s,
                              bool have_ln1 = false) {
  const int M = B * NV;
  const int RS = NV * D;                      // row stride between CLS rows
  if (!have_ln1) CK(vitcap_layernorm_fwd(x, D, w.n1_g, w.n1_b, 1e-6f, h, nullptr, M, D, s));
  CK(gemm(h, D, (const char*)w.qkv_w + (size_t)D * D * 2, w.qkv_b + D, nullptr, 0, (char*)qkv + (size_t)D * 2, 3 * D, M, 2 * D, D,
          VITCAP_ACT_NONE, VITCAP_OUT_BF16, s));                                         // K | V of every row
  CK(gemm(h, RS, w.qkv_w, w.qkv_b, nullptr, 0, qkv, NV * 3 * D, B, D, D, VITCAP_ACT_NONE, VITCAP_OUT_BF16, s));   // Q of the CLS rows
  CK(vitcap_attn_dense_fwd_rows(qkv, h, B, NV, 1, 0.125f, s));
  CK(gemm(h, RS, w.proj_w, w.proj_b, x, RS, x, RS, B, D, D, VITCAP_ACT_NONE, VITCAP_OUT_F32, s));
  CK(vitcap_layernorm_fwd(x, RS, w.n2_g, w.n2_b, 1e-6f, cls_h, nullptr, B, D, s));
  CK(gemm(cls_h, D, w.fc1_w, w.fc1_b, nullptr, 0, mlp, 4 * D, B, 4 * D, D, VITCAP_ACT_GELU_ERF, VITCAP_OUT_BF16, s));
  CK(gemm(mlp, 4 * D, w.fc2_w, w.fc2_b, x, RS, x, RS, B, D, 4 * D, VITCAP_ACT_NONE, VITCAP_OUT_F32, s));
  return VITCAP_OK;
}

static int tag_branch(vitcap_engine* e, const Layout& lo, char* ws, int B, void* s);

static int encode_part(vitcap_engine* e, const void* image, int image_is_bf16, int B, const vitcap_gen_opts& o, const Layout& lo,
                       char* ws, bool allow_fork, void* s) {
  const vitcap_weights& w = e->w;
  float* x = (float*)(ws + lo.x);
  // a1: patch embed as GEMM (+bias +pos_embed[1+p]) into rows b*577+1+p; cls rows separately
  CK(vitcap_patch_gather(image, image_is_bf16, ws + lo.patches, B, s));
  {
    vitcap_gemm_desc d;
    memset(&d, 0, sizeof(d));
  d.abi = VITCAP_ABI_VERSION;
    d.M = B * 576; d.N = D; d.K = D;
    d.lda = D; d.ldw = D; d.ldc = D; d.ldr = D;
    d.act = VITCAP_ACT_NONE; d.out_dtype = VITCAP_OUT_F32;
    d.row_group = 576; d.out_group_rows = NV; d.out_row_off = 1; d.res_periodic = 1;
    CK(gemm_desc(ws + lo.patches, w.patch_w, w.patch_b, w.pos_embed + D, x, d, s));
    zz();
  }
  CK(vitcap_cls_rows(w.cls_token, w.pos_embed, x, B, NV, s));
  // a5: 12 blocks, fork before block 8, 4 tag blocks on the fork.  Run the fork on a side stream when the large GEMMs are
  // in their one-tile-per-workgroup form (batch pipeline) and the batch is small enough for tile-quantisation gaps to
  // matter: B=64 pipelined +2.3 %; with persistent GEMMs or at B=512 it costs 1-2 % (measured), so it stays serial there.
  const bool fork = allow_fork && e->fork_tag_branch && o.gemm_mode == VITCAP_GEMM_TILES && B <= 128;
  float* x2 = (float*)(ws + lo.x2);
  // ticket counters of the GEMMs that normalise their own rows: zero before the first of them (the workspace is the caller's)
  int32_t* cnt = (int32_t*)(ws + lo.ln_cnt);
  // (exactly this part's 4 B counters: a batch part on another stream owns the ints behind them)
  HIPCK(hipMemsetAsync(cnt, 0, (size_t)B * 4 * 4, (hipStream_t)s), "encode: counters");
  HIPCK(hipMemsetAsync(ws + lo.ln_cnt_tag, 0, (size_t)B * 4 * 4, (hipStream_t)s), "encode: counters");
  for (int i = 0; i < 12; ++i) {
    if (i == 8 && fork) {
      // fork: the tag branch depends only on x (the output of block 7), which nobody writes from here on
      if (!e->side) {
        if ((e->side = role_stream(ROLE_SIDE)) == nullptr ||
            hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) != hipSuccess) {
          vitcap_set_error("encode: side stream creation failed");
          return VITCAP_ELAUNCH;
        }
      }
      HIPCK(hipEventRecord(e->ev_fork, (hipStream_t)s), "encode: fork record");
      HIPCK(hipStreamWaitEvent(e->side, e->ev_fork, 0), "encode: fork wait");
      CK(tag_branch(e, lo, ws, B, e->side));
      HIPCK(hipEventRecord(e->ev_join, e->side), "encode: join record");
    }
    // norm1 of block i+1 rides in block i's fc2 (block 7 feeds block 8 that way; the tag branch normalises the fork state itself)
    const vitcap_vit_block_w* next = i + 1 < 12 ? &w.blocks[i + 1] : nullptr;
    if (i < 8) CK(vit_block(w.blocks[i], x, x, ws + lo.h, ws + lo.qkv, ws + lo.mlp, B, s, cnt, i > 0, next));
    else CK(vit_block(w.blocks[i], i == 8 ? x : x2, x2, ws + lo.h, ws + lo.qkv, ws + lo.mlp, B, s, cnt, true, next));
  }
  if (fork) {
    HIPCK(hipStreamWaitEvent((hipStream_t)s, e->ev_join, 0), "encode: join wait");
  } else {
    CK(tag_branch(e, lo, ws, B, s));
  }
  return VITCAP_OK;
}

static int prefill_part(vitcap_engine* e, int B, const vitcap_gen_opts& o, const Layout& lo, char* ws, void* s);

// vitcap_gen_opts.encode_parts: the batch is cut into parts whose encoder + prefill run as independent chains on separate
// streams (part 0 on the caller's), so that the tile-quantisation tail of one part's GEMM (qkv: 5.1 rounds of 256 CUs cost 6 at
// B = 64) is filled by the other part's kernels; every row's arithmetic is unchanged (bit-identical results).  Measured,
// 2-slot pipeline, images/s without / with 2 parts: B = 16 2039 / 2034, 32 2825 / 2854, 64 3560 / 3635, 128 3724 / 3808,
// 512 3975 / 3990; 3 and 4 parts lose (B = 64: 3318 / 3460).  With two GEMM chains in flight every launch of the dominant kernel
// shares the chip with the other chain: its per-launch rate (the bench's roofline.frac) drops from 0.24 to 0.17 of peak although
// throughput rises -- roofline.frac_busy (flops / union of the launches' intervals) is the figure that stays comparable.
// VITCAP_ENCODE_SPLIT overrides for experiments.
static int encode_parts(const vitcap_gen_opts& o, int B, const Layout& lo) {
  static const int env = [] { const char* e = getenv("VITCAP_ENCODE_SPLIT"); return e ? atoi(e) : -1; }();
  int p = env >= 0 ? env : o.encode_parts;
  // auto (round 5): two parts inside the batch pipeline from 32 images on -- images/s is the metric (+2.1 % at B = 64, +2.3 % at 128,
  // +0.4 % at 512, measured above); the bench reports the dominant kernel's busy-interval rate (frac_busy) beside the per-launch one
  if (p == 0) p = (o.gemm_mode == VITCAP_GEMM_TILES && B >= 32) ? 2 : 1;
  if (p > 4) p = 4;
  if (B < 8 || lo.NT > 0) p = 1;
  return p;
}

static int ensure_dec2(vitcap_engine* e) {
  if (!e->dec2) {
    if ((e->dec2 = role_stream(ROLE_DEC2)) == nullptr ||
        hipEventCreateWithFlags(&e->ev_dfork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_djoin, hipEventDisableTiming) != hipSuccess) {
      vitcap_set_error("engine: second stream creation failed");
      return VITCAP_ELAUNCH;
    }
  }
  return VITCAP_OK;
}

static int encode_locked(vitcap_engine* e, const void* image, int image_is_bf16, int B, const vitcap_gen_opts& o, const Layout& lo,
                         char* ws, void* s) {
  if (!image) { vitcap_set_error("encode: null image"); return VITCAP_EINVAL; }
  CallScope scope(e, o.gemm_mode, nullptr, &o);
  WalkScope walk;
  // timing runs: a step is sampled WHOLE (its encoder and prefill launches), so that the union of the sampled launches' intervals
  // still sees which of them ran next to each other (tag branch beside caption blocks 8-11, batch parts)
  if (e->timing) e->timing_this_step = (e->timing_seen++ % e->timing_stride) == 0;
  const int np = encode_parts(o, B, lo);
  if (np >= 2) {
    if (!e->ev_pfork) HIPCK(hipEventCreateWithFlags(&e->ev_pfork, hipEventDisableTiming), "encode: event");
    for (int i = 0; i < np - 1; ++i)
      if (!e->part[i]) {
        if ((e->part[i] = role_stream(ROLE_PART0 + i)) == nullptr) { vitcap_set_error("encode: part stream creation failed"); return VITCAP_ELAUNCH; }
        HIPCK(hipEventCreateWithFlags(&e->ev_pjoin[i], hipEventDisableTiming), "encode: event");
      }
    HIPCK(hipEventRecord(e->ev_pfork, (hipStream_t)s), "encode: split fork record");
    const size_t img_bytes = (size_t)3 * 384 * 384 * (image_is_bf16 ? 2 : 4);
    int i0[5];
    for (int i = 0; i <= np; ++i) i0[i] = (int)((long long)B * i / np);
    for (int i = 0; i < np; ++i) {
      void* ps = i == 0 ? s : (void*)e->part[i - 1];
      if (i > 0) HIPCK(hipStreamWaitEvent((hipStream_t)ps, e->ev_pfork, 0), "encode: split fork wait");
      const Layout lv = lo.from_image(i0[i]);
      CK(encode_part(e, (const char*)image + (size_t)i0[i] * img_bytes, image_is_bf16, i0[i + 1] - i0[i], o, lv, ws, false, ps));
    }
    // the prefill of each part follows on its own stream (prefill_locked then has nothing left to do)
    for (int i = 0; i < np; ++i) {
      void* ps = i == 0 ? s : (void*)e->part[i - 1];
      CK(prefill_part(e, i0[i + 1] - i0[i], o, lo.from_image(i0[i]), ws, ps));
      if (i > 0) {
        HIPCK(hipEventRecord(e->ev_pjoin[i - 1], (hipStream_t)ps), "encode: split join record");
        HIPCK(hipStreamWaitEvent((hipStream_t)s, e->ev_pjoin[i - 1], 0), "encode: split join wait");
      }
    }
    return VITCAP_OK;
  }
  return encode_part(e, image, image_is_bf16, B, o, lo, ws, true, s);
}

extern "C" int vitcap_engine_encode(vitcap_engine* e, const void* image, int image_is_bf16, int B, const vitcap_gen_opts* opts,
                                    void* workspace, size_t workspace_bytes, void* s) {
  const vitcap_gen_opts o = opts ? *opts : default_opts();
  if (B <= 0 || check_opts(o) != VITCAP_OK) { if (B <= 0) vitcap_set_error("engine: bad batch"); return VITCAP_EINVAL; }
  const Layout lo(B, o);
  CK(check(e, B, o, workspace, workspace_bytes, lo.off));
  std::lock_guard<std::mutex> lk(e->mu);
  return encode_locked(e, image, image_is_bf16, B, o, lo, (char*)workspace, s);
}

// a5 (tag fork) + a6: 4 tag blocks on the forked stream, then the tag head on the tag branch CLS row
static int tag_branch(vitcap_engine* e, const Layout& lo, char* ws, int B, void* s) {
  const vitcap_weights& w = e->w;
  float* xt = (float*)(ws + lo.xt);
  const float* xf = (const float*)(ws + lo.x);       // fork state (output of block 7)
  int32_t* cnt = (int32_t*)(ws + lo.ln_cnt_tag);      // its own counters: the branch may run next to caption blocks 8-11
  for (int i = 0; i < 3; ++i)
    CK(vit_block(w.tag_blocks[i], i == 0 ? xf : xt, xt, ws + lo.th, ws + lo.tqkv, ws + lo.tmlp, B, s, cnt, i > 0, &w.tag_blocks[i + 1]));
  if (e->full_last_tag_block)
    CK(vit_block(w.tag_blocks[3], xt, xt, ws + lo.th, ws + lo.tqkv, ws + lo.tmlp, B, s, cnt, true, nullptr));
  else
    CK(vit_block_cls_only(w.tag_blocks[3], xt, ws + lo.th, ws + lo.tqkv, ws + lo.tmlp, ws + lo.pool_in, B, s, true));
  CK(vitcap_gather_rows_bf16(xt, NV, ws + lo.pool_in, B, D, s));
  CK(gemm(ws + lo.pool_in, D, w.pooler_w, w.pooler_b, nullptr, 0, ws + lo.pooled, D, B, D, D, VITCAP_ACT_TANH,
          VITCAP_OUT_BF16, s));
  CK(gemm(ws + lo.pooled, D, w.tag_logit.dense_w, w.tag_logit.dense_b, nullptr, 0, ws + lo.tg_f, D, B, D, D,
          VITCAP_ACT_GELU_ERF, VITCAP_OUT_F32, s));
  CK(vitcap_layernorm_fwd((const float*)(ws + lo.tg_f), D, w.tag_logit.ln_g, w.tag_logit.ln_b, 1e-12f, ws + lo.tg_b,
                          nullptr, B, D, s));
  CK(gemm(ws + lo.tg_b, D, w.tag_logit.dec_w, w.tag_logit.dec_b, nullptr, 0, ws + lo.tag_logits, VP, B, VP, D,
          VITCAP_ACT_NONE, VITCAP_OUT_F32, s));
  CK(vitcap_sigmoid_topk((const float*)(ws + lo.tag_logits), VP, VITCAP_VOCAB, TOPK, 0.2f, (int64_t*)(ws + lo.tag_ids),
                         (float*)(ws + lo.tag_prob), (int64_t*)(ws + lo.tag_len), B, s));
  return VITCAP_OK;
}

// the predicted tag tokens as real rows of the joint sequence.  With the mask tensorize_ab builds for a
// text_b of n tokens (dataset.py:240-252, 387-390) the n tag rows attend each other and the 578 visual rows, and every caption
// row attends them; nothing they attend depends on the caption, so their hidden states -- hence their K/V in every decoder
// layer -- are computed ONCE here, for BOTH embedding branches of modeling_bert.py:1435-1489 (the reference re-evaluates
// `topk_len[0] + 20 <= L` at every step: the decode attention picks the branch per step, vitcap_attn_decode_step_tags).
// Per layer: tag q|k|v (compact rows = the cache) -> joint buffer [visual K/V | tag rows] per image -> dense MFMA attention on
// the query range that covers the tag rows -> BertSelfOutput / BertIntermediate / BertOutput on the tag rows.
static int prefill_tags(vitcap_engine* e, int B, const vitcap_gen_opts& o, const Layout& lo, char* ws, void* s) {
  const vitcap_weights& w = e->w;
  const int n = lo.NT, R = B * n, S2 = SV + n;
  if (!o.tagemb_cls && !(w.xword_emb && w.xpos_emb && w.xtype_emb && w.xemb_ln_g && w.xemb_ln_b)) {
    vitcap_set_error("prefill: tag_visible with tagemb != 'cls' needs bert.extra_embeddings bound (vitcap_weights.x*)");
    return VITCAP_ESTATE;
  }
  for (int v = 0; v < 2; ++v)
    CK(vitcap_tag_embed((const int64_t*)(ws + lo.tag_ids), n, o.tag_pos0, v == 0, o.tagemb_cls, w.cls.dec_w, w.word_emb, w.pos_emb, w.type_emb,
                        w.emb_ln_g, w.emb_ln_b, w.xword_emb, w.xpos_emb, w.xtype_emb, w.xemb_ln_g, w.xemb_ln_b, 1e-12f,
                        (float*)(ws + lo.tagx_f[v]), ws + lo.tagx_b[v], B, s));
  for (int l = 0; l < 4; ++l) {
    const vitcap_bert_layer_w& lw = w.dec[l];
    if (l < 3)        // visual K | V of this layer into the joint buffer (the Q columns of those rows are never read as queries we keep)
      CK(vitcap_copy_row_blocks(ws + lo.dqkv[l], SV, 0, 3 * D, D, ws + lo.jqkv, JROWS, 0, 3 * D, D, SV, 2 * D, B, s));
    for (int v = 0; v < 2; ++v) {
      char* tq = ws + lo.tqkv_c[v][l];
      float* xf = (float*)(ws + lo.tagx_f[v]);
      char* xb = ws + lo.tagx_b[v];
      if (l == 3) {   // the last layer's tag-row outputs feed nothing: K | V only
        CK(gemm(xb, D, (const char*)lw.qkv_w + (size_t)D * D * 2, lw.qkv_b + D, nullptr, 0, tq + (size_t)D * 2, 3 * D, R, 2 * D, D,
                VITCAP_ACT_NONE, VITCAP_OUT_BF16, s));
        continue;
      }
      CK(gemm(xb, D, lw.qkv_w, lw.qkv_b, nullptr, 0, tq, 3 * D, R, 3 * D, D, VITCAP_ACT_NONE, VITCAP_OUT_BF16, s));
      CK(vitcap_copy_row_blocks(tq, n, 0, 3 * D, 0, ws + lo.jqkv, JROWS, SV, 3 * D, 0, n, 3 * D, B, s));
      CK(vitcap_attn_dense_fwd_train_rows(ws + lo.jqkv, ws + lo.jout, (float*)(ws + lo.jlse), B, S2, JROWS, 0.125f, 0.f, 0u, 0, 0, 512, S2, s));
      CK(vitcap_copy_row_blocks(ws + lo.jout, JROWS, SV, D, 0, ws + lo.tg_ctx, n, 0, D, 0, n, D, B, s));
      CK(gemm(ws + lo.tg_ctx, D, lw.ao_w, lw.ao_b, xf, D, ws + lo.tg_tmp, D, R, D, D, VITCAP_ACT_NONE, VITCAP_OUT_F32, s));
      CK(vitcap_layernorm_fwd((const float*)(ws + lo.tg_tmp), D, lw.ao_g, lw.ao_beta, 1e-12f, ws + lo.tg_sa_b, (float*)(ws + lo.tg_sa_f), R, D, s));
      CK(gemm(ws + lo.tg_sa_b, D, lw.i_w, lw.i_b, nullptr, 0, ws + lo.tg_mlp, 4 * D, R, 4 * D, D, VITCAP_ACT_GELU_ERF, VITCAP_OUT_BF16, s));
      CK(gemm(ws + lo.tg_mlp, 4 * D, lw.o_w, lw.o_b, (const float*)(ws + lo.tg_sa_f), D, ws + lo.tg_tmp, D, R, D, 4 * D, VITCAP_ACT_NONE,
              VITCAP_OUT_F32, s));
      CK(vitcap_layernorm_fwd((const float*)(ws + lo.tg_tmp), D, lw.o_g, lw.o_beta, 1e-12f, xb, xf, R, D, s));
    }
  }
  return VITCAP_OK;
}

static int prefill_locked(vitcap_engine* e, int B, const vitcap_gen_opts& o, const Layout& lo, char* ws, void* s) {
  CallScope scope(e, o.gemm_mode, nullptr, &o);
  WalkScope walk;
  if (encode_parts(o, B, lo) >= 2) return VITCAP_OK;       // done by encode_locked, per part
  return prefill_part(e, B, o, lo, ws, s);
}

static int prefill_part(vitcap_engine* e, int B, const vitcap_gen_opts& o, const Layout& lo, char* ws, void* s) {
  const vitcap_weights& w = e->w;
  const int M = B * SV;
  float* vis_f = (float*)(ws + lo.vis_f);
  void* vis_b = ws + lo.vis_b;
  CK(vitcap_assemble_visual((const float*)(ws + lo.x2), (const float*)(ws + lo.xt), vis_f, vis_b, B, NV, s));
  // the ticket counters of the GEMMs that normalise their own rows are zero on exit of every completed launch; zero them here as well,
  // so that a launch that was aborted (or a caller that runs prefill without encode) cannot poison the next one
  HIPCK(hipMemsetAsync(ws + lo.ln_cnt, 0, (size_t)B * 4 * 4, (hipStream_t)s), "prefill: counters");
  for (int l = 0; l < 4; ++l) {
    const vitcap_bert_layer_w& lw = w.dec[l];
    void* dq = ws + lo.dqkv[l];
    if (l == 3) {        // the last layer's visual-row outputs feed nothing: only its K/V are needed (no Q either)
      CK(gemm(vis_b, D, (const char*)lw.qkv_w + (size_t)D * D * 2, lw.qkv_b + D, nullptr, 0, (char*)dq + (size_t)D * 2, 3 * D, M,
              2 * D, D, VITCAP_ACT_NONE, VITCAP_OUT_BF16, s));
      if (lo.vt[l]) CK(vitcap_attn_beam_vt(dq, ws + lo.vt[l], B, SV, s));
      break;
    }
    CK(gemm(vis_b, D, lw.qkv_w, lw.qkv_b, nullptr, 0, dq, 3 * D, M, 3 * D, D, VITCAP_ACT_NONE, VITCAP_OUT_BF16, s));
    zz();
    if (lo.vt[l]) CK(vitcap_attn_beam_vt(dq, ws + lo.vt[l], B, SV, s));
    CK(vitcap_attn_dense_fwd(dq, ws + lo.h, B, SV, 0.125f, s));
    zz();
    // BertSelfOutput / BertOutput: dense + residual, then LayerNorm (post-LN) -- the LayerNorm rides in the GEMM
    int32_t* cnt = (int32_t*)(ws + lo.ln_cnt);
    CK(gemm_ln(ws + lo.h, D, lw.ao_w, lw.ao_b, vis_f, ws + lo.dtmp, M, D, lw.ao_g, lw.ao_beta, 1e-12f, ws + lo.da_b,
               (float*)(ws + lo.da_f), cnt, s));
    CK(gemm(ws + lo.da_b, D, lw.i_w, lw.i_b, nullptr, 0, ws + lo.mlp, 4 * D, M, 4 * D, D, VITCAP_ACT_GELU_ERF,
            VITCAP_OUT_BF16, s));
    zz();
    CK(gemm_ln(ws + lo.mlp, 4 * D, lw.o_w, lw.o_b, (const float*)(ws + lo.da_f), ws + lo.dtmp, M, 4 * D, lw.o_g, lw.o_beta, 1e-12f,
               vis_b, vis_f, cnt, s));
  }
  if (lo.NT > 0) CK(prefill_tags(e, B, o, lo, ws, s));
  return VITCAP_OK;
}

extern "C" int vitcap_engine_prefill(vitcap_engine* e, int B, const vitcap_gen_opts* opts, void* workspace, size_t workspace_bytes,
                                     void* s) {
  const vitcap_gen_opts o = opts ? *opts : default_opts();
  if (B <= 0 || check_opts(o) != VITCAP_OK) { if (B <= 0) vitcap_set_error("engine: bad batch"); return VITCAP_EINVAL; }
  const Layout lo(B, o);
  CK(check(e, B, o, workspace, workspace_bytes, lo.off));
  std::lock_guard<std::mutex> lk(e->mu);
  return prefill_locked(e, B, o, lo, (char*)workspace, s);
}

// A contiguous slice of the decode batch: sequences [s0, s0 + ns) = images [i0, i0 + ns / K).  The greedy loop can be cut into
// two such slices that run on two streams (vitcap_gen_opts.decode_streams = 2).  Every decode-step kernel costs ~4.5 us of
// dispatch-to-drain latency whatever its size (31 of them per step: 140 us of a 295 us step at 64 sequences); the experiment
// showed that a second chain does NOT hide it (see greedy_loop).  Results are bit-identical to the unsplit loop.
struct Part {
  int s0, ns, i0;
};

// One decode step for the sequences of `pt` (K sequences share one image's visual K/V): embeddings of (token t-1, [MASK]) ->
// 4 decoder layers against the caches -> LM head on the [MASK] rows -> fp32 logits [ns, VOCAB_PAD] (+ row statistics).
static int step_forward(const vitcap_weights& w, const Layout& lo, const vitcap_gen_opts& o, char* ws, int t, const int64_t* ids_all,
                        char* tcache, bool embed, bool rowstat, const Part& pt, void* s) {
  const int NS = lo.NS, K = lo.K, L = lo.L;
  const int ns = pt.ns, R = 2 * ns;
  const size_t r0 = (size_t)pt.s0 * 2;                       // first step-buffer row of the slice
  const int64_t* ids = ids_all + (size_t)pt.s0 * L;
  float* xs_f = (float*)(ws + lo.xs_f) + r0 * D;
  char* xs_b = ws + lo.xs_b + r0 * D * 2;
  char* sqkv = ws + lo.sqkv + r0 * 3 * D * 2;
  char* sctx = ws + lo.sctx + r0 * D * 2;
  float* sa_f = (float*)(ws + lo.sa_f) + r0 * D;
  char* sa_b = ws + lo.sa_b + r0 * D * 2;
  char* smlp = ws + lo.smlp + r0 * 4 * D * 2;
  char* hd_b = ws + lo.hd_b + (size_t)pt.s0 * D * 2;
  float* part = (float*)(ws + lo.spart) + (size_t)SPLIT_MAX * r0 * D;      // the slice's own slab region
  if (embed)          // otherwise the previous step's vitcap_greedy_select_embed already wrote this step's x
    CK(vitcap_embed_step(ids, L, t, o.mask_token_id, w.word_emb, w.pos_emb, w.type_emb, w.emb_ln_g, w.emb_ln_b, 1e-12f, xs_f, xs_b,
                         ns, s));
  static const int force_old = [] { const char* e = getenv("VITCAP_DECODE_SPLITK"); return e ? atoi(e) : 0; }();   // A/B measurements
  // small-tile LDS-DMA ring kernels (gemm_small) for batches of few rows; larger ones take the big-tile / split-K path.  The choice
  // follows the WHOLE batch, so that a sequence's arithmetic does not depend on how the batch is sliced.
  // up to 1024 rows (512 sequences) the small-tile ring forms of gemm_small win (decode phase 13.2 -> 11.9 ms at 256 images,
  // 20.4 -> 19.9 at 512); at 2560 rows (5 beams x 256 images) the 128x128 / 256x256 tiles do (21.4 against 23.8 ms)
  const bool small = 2 * NS <= 1024 && !force_old;
  static const int no_beam_attn = [] { const char* e = getenv("VITCAP_BEAM_ATTN_VALU"); return e ? atoi(e) : 0; }();   // A/B measurements
  for (int l = 0; l < 4; ++l) {
    const vitcap_bert_layer_w& lw = w.dec[l];
    char* tc = tcache + ((size_t)l * NS + pt.s0) * L * 2 * D * 2;
    const char* vis = ws + lo.dqkv[l] + (size_t)pt.i0 * SV * 3 * D * 2;
    if (small)
      CK(gemm_small(xs_b, D, lw.qkv_w, lw.qkv_b, sqkv, 3 * D, R, 3 * D, D, VITCAP_ACT_NONE, VITCAP_OUT_BF16, 20, s));
    else
      CK(gemm(xs_b, D, lw.qkv_w, lw.qkv_b, nullptr, 0, sqkv, 3 * D, R, 3 * D, D, VITCAP_ACT_NONE, VITCAP_OUT_BF16, s));
    if (lo.NT > 0)
      CK(vitcap_attn_decode_step_tags(sqkv, vis, tc, sctx, ns, SV, t, L, K, 0.125f,
                                      ws + lo.tqkv_c[0][l] + (size_t)pt.i0 * lo.NT * 3 * D * 2, ws + lo.tqkv_c[1][l] + (size_t)pt.i0 * lo.NT * 3 * D * 2,
                                      lo.NT, (const int64_t*)(ws + lo.tag_len), s));
    else if (lo.vt[l] && K >= 2 && K <= 8 && !no_beam_attn)
      // several sequences per image (beam search): all of an image's query rows against its visual rows on the matrix pipe
      CK(vitcap_attn_decode_beams(sqkv, vis, ws + lo.vt[l] + (size_t)pt.i0 * VT_BYTES, tc, sctx, ns / K, K, SV, t, L, 0.125f, s));
    else if (lo.vt[l] && K > 8 && lo.group_k > 1 && !no_beam_attn)
      // more than 8 sequences per image (constrained beam search: states x beams): groups of group_k sequences, K / group_k per image
      CK(vitcap_attn_decode_beam_groups(sqkv, vis, ws + lo.vt[l] + (size_t)pt.i0 * VT_BYTES, tc, sctx, ns / K, lo.group_k, K / lo.group_k, SV, t,
                                        L, 0.125f, s));
    else
      CK(vitcap_attn_decode_step(sqkv, vis, tc, sctx, ns, SV, t, L, K, 0.125f, s));
    // attention.output.dense and output.dense: fp32 partial slabs (one per 768-long k range; split-K 6 / 12 for beam batches),
    // reduced inside the fused bias + residual + LayerNorm kernel (BertSelfOutput / BertOutput, modeling_bert.py:353-357, 415-419)
    // split-K of the two N = 768 GEMMs by the WHOLE batch's rows (a sequence's sums must not depend on how the batch is sliced):
    // 6 / 12 slabs fill the chip at a few hundred rows; from ~1000 rows on the output tiles alone do, and the fp32 slabs (47 /
    // 94 MB per GEMM at 2560 rows) cost more than they buy -- decode phase at 5 beams x 256 images 24.6 -> 22.0 ms, 512 greedy
    // sequences 21.5 -> 20.8 ms, 256 sequences unchanged (measured)
    const int rows_all = 2 * NS;
    int s_ao = rows_all >= 2048 ? 1 : (rows_all >= 1024 ? 2 : SPLIT_AO), s_fc2 = rows_all >= 1024 ? 4 : SPLIT_FC2;
    if (small) {
      CK(gemm_small(sctx, D, lw.ao_w, nullptr, part, D, R, D, D, VITCAP_ACT_NONE, VITCAP_OUT_F32, 21, s));
      s_ao = 1;
    } else {
      CK(gemm_split(sctx, D, lw.ao_w, part, R, D, D, s_ao, s));
    }
    CK(vitcap_sum_layernorm(part, s_ao, (size_t)R * D, lw.ao_b, xs_f, D, 0, lw.ao_g, lw.ao_beta, 1e-12f, sa_b, sa_f, R, D, s));
    if (small)
      CK(gemm_small(sa_b, D, lw.i_w, lw.i_b, smlp, 4 * D, R, 4 * D, D, VITCAP_ACT_GELU_ERF, VITCAP_OUT_BF16, 20, s));
    else
      CK(gemm(sa_b, D, lw.i_w, lw.i_b, nullptr, 0, smlp, 4 * D, R, 4 * D, D, VITCAP_ACT_GELU_ERF, VITCAP_OUT_BF16, s));
    if (small) {
      CK(gemm_small(smlp, 4 * D, lw.o_w, nullptr, part, D, R, D, 4 * D, VITCAP_ACT_NONE, VITCAP_OUT_F32, 20, s));
      s_fc2 = 4;
    } else {
      CK(gemm_split(smlp, 4 * D, lw.o_w, part, R, D, 4 * D, s_fc2, s));
    }
    CK(vitcap_sum_layernorm(part, s_fc2, (size_t)R * D, lw.o_b, sa_f, D, 0, lw.o_g, lw.o_beta, 1e-12f, xs_b, xs_f, R, D, s));
  }
  // LM head on the [MASK] rows (row 1 of every pair): A = xs_b + 768, lda = 1536
  if (small) {
    CK(gemm_small(xs_b + D * 2, 2 * D, w.cls.dense_w, nullptr, part, D, ns, D, D, VITCAP_ACT_NONE, VITCAP_OUT_F32, 21, s));
    CK(vitcap_sum_layernorm(part, 1, (size_t)ns * D, w.cls.dense_b, nullptr, 0, 1, w.cls.ln_g, w.cls.ln_b, 1e-12f, hd_b, nullptr, ns, D, s));
  } else {
    const int s_hd = NS >= 2048 ? 1 : (NS >= 1024 ? 2 : SPLIT_AO);      // as above, by the whole batch's [MASK] rows
    CK(gemm_split(xs_b + D * 2, 2 * D, w.cls.dense_w, part, ns, D, D, s_hd, s));
    CK(vitcap_sum_layernorm(part, s_hd, (size_t)ns * D, w.cls.dense_b, nullptr, 0, 1, w.cls.ln_g, w.cls.ln_b, 1e-12f, hd_b, nullptr,
                            ns, D, s));
  }
  {
    // vocabulary GEMM: 47 MB of weights streamed once per step.  With few rows (greedy: NS <= 128) the 64x64-tile kernel
    // moves them at 3.8 TB/s against 2.3 TB/s for the 32x32 tiles the small-M dispatch would pick (12.5 vs 20.8 us at NS = 64)
    vitcap_gemm_desc d;
    memset(&d, 0, sizeof(d));
  d.abi = VITCAP_ABI_VERSION;
    d.M = ns; d.N = VP; d.K = D;
    d.lda = D; d.ldw = D; d.ldc = VP;
    d.act = VITCAP_ACT_NONE; d.out_dtype = VITCAP_OUT_F32;
    d.tile_hint = NS <= 128 ? 1 : 0;
    // greedy: argmax / log-softmax pieces next to the logits
    d.rowstat = rowstat ? (float*)(ws + lo.rowstat) + (size_t)pt.s0 * (2 * (VP / 64)) * 4 : nullptr;
    CK(gemm_desc(hd_b, w.cls.dec_w, w.cls.dec_b, nullptr, ws + lo.logits + (size_t)pt.s0 * VP * 4, d, s));
  }
  return VITCAP_OK;
}

// Greedy / sampled decode loop of NS = B * K sequences, K per image (K > 1: ViTCAP.generate with num_return_sequences = K
// expands every input K times, modeling_bert.py:976-994; the K copies of an image share its encoder output and visual K/V
// here, as the beams of a beam search do).  Results stay in the workspace (lo.ids, lo.logprob, lo.last_tok).
static int greedy_loop(vitcap_engine* e, const Layout& lo, const vitcap_gen_opts& o, char* ws, void* s) {
  const vitcap_weights& w = e->w;
  const int NS = lo.NS, L = lo.L, K = lo.K, B = NS / K;
  int64_t* ids = (int64_t*)(ws + lo.ids);
  int32_t* unf = (int32_t*)(ws + lo.unf);
  float* sum_lp = (float*)(ws + lo.sum_lp);
  float* cnt = (float*)(ws + lo.cnt);
  CK(vitcap_greedy_init(ids, unf, sum_lp, cnt, NS, L, o.bos_token_id, o.pad_token_id, s));
  // Plain greedy decoding of a small batch: the vocabulary GEMM also emits per-piece (max, argmax, sum exp) of its rows, and ONE
  // kernel turns them into the token, its log-prob, the bookkeeping and the NEXT step's embedded rows -- instead of reading the
  // 30522-wide fp32 rows back (greedy_step 18.7 us) and a separate embedding launch per step.
  static const int no_fuse = [] { const char* e = getenv("VITCAP_DECODE_NOFUSE"); return e ? atoi(e) : 0; }();
  const bool fused = !o.sampling.do_sample && o.repetition_penalty == 1.0f && NS <= 128 && !no_fuse;
  // two slices on two streams (decode_streams = 2): measured at 64 sequences, eager and graph-replayed: 5.96 ms per batch against
  // 5.66 ms for one chain -- the ~4.5 us per dependent small kernel is not hidden by a second chain (the dispatch path is the
  // shared resource), so auto = 1; the option stays for experiments and is covered by tests (bit-identical results)
  int nparts = o.decode_streams;
  if (nparts == 0) nparts = 1;
  if (B < 2) nparts = 1;
  Part parts[2] = {{0, NS, 0}, {0, 0, 0}};
  void* st[2] = {s, s};
  if (nparts == 2) {
    const int b0 = B / 2;
    parts[0] = Part{0, b0 * K, 0};
    parts[1] = Part{b0 * K, (B - b0) * K, b0};
    if (!e->dec2) {
      if ((e->dec2 = role_stream(ROLE_DEC2)) == nullptr ||
          hipEventCreateWithFlags(&e->ev_dfork, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&e->ev_djoin, hipEventDisableTiming) != hipSuccess) {
        vitcap_set_error("decode: second stream creation failed");
        return VITCAP_ELAUNCH;
      }
    }
    HIPCK(hipEventRecord(e->ev_dfork, (hipStream_t)s), "decode: fork record");
    HIPCK(hipStreamWaitEvent(e->dec2, e->ev_dfork, 0), "decode: fork wait");
    st[1] = (void*)e->dec2;
  }
  for (int t = 1; t < L; ++t) {
    for (int p = 0; p < nparts; ++p) {            // the slices' launches alternate so that both streams are fed evenly
      const Part& pt = parts[p];
      void* ps = st[p];
      const size_t s0 = (size_t)pt.s0;
      CK(step_forward(w, lo, o, ws, t, ids, ws + lo.tcache, !fused || t == 1, fused, pt, ps));
      float* logits = (float*)(ws + lo.logits) + s0 * VP;
      if (fused) {
        CK(vitcap_greedy_select_embed((const float*)(ws + lo.rowstat) + s0 * (2 * (VP / 64)) * 4, 2 * (VP / 64), ids + s0 * L, unf + s0,
                                      sum_lp + s0, cnt + s0, (float*)(ws + lo.logprob) + s0, (int64_t*)(ws + lo.last_tok) + s0, pt.ns, t,
                                      L, o.eos_token_id, o.pad_token_id, o.mask_token_id, w.word_emb, w.pos_emb, w.type_emb,
                                      w.emb_ln_g, w.emb_ln_b, 1e-12f, (float*)(ws + lo.xs_f) + s0 * 2 * D, ws + lo.xs_b + s0 * 2 * D * 2, ps));
        continue;
      }
      if (o.repetition_penalty != 1.0f)
        CK(vitcap_repetition_penalty(logits, VP, VITCAP_VOCAB, ids + s0 * L, L, t, o.repetition_penalty, pt.ns, ps));
      if (o.sampling.do_sample) {
        // the draws are keyed by (seed, sequence index within the call): a slice passes its first sequence as the stream offset
        vitcap_sample_params sp = o.sampling;
        CK(vitcap_sample_step_offset(logits, VP, VITCAP_VOCAB, ids + s0 * L, unf + s0, sum_lp + s0, cnt + s0,
                                     (float*)(ws + lo.logprob) + s0, (float*)(ws + lo.margins) + s0 * L, (int64_t*)(ws + lo.last_tok) + s0,
                                     pt.ns, t, L, o.eos_token_id, o.pad_token_id, &sp, pt.s0, ps));
      } else {
        CK(vitcap_greedy_step(logits, VP, VITCAP_VOCAB, ids + s0 * L, unf + s0, sum_lp + s0, cnt + s0, (float*)(ws + lo.logprob) + s0,
                              (float*)(ws + lo.margins) + s0 * L, (int64_t*)(ws + lo.last_tok) + s0, pt.ns, t, L, o.eos_token_id,
                              o.pad_token_id, ps));
      }
    }
  }
  if (nparts == 2) {
    HIPCK(hipEventRecord(e->ev_djoin, e->dec2), "decode: join record");
    HIPCK(hipStreamWaitEvent((hipStream_t)s, e->ev_djoin, 0), "decode: join wait");
  }
  return VITCAP_OK;
}

// Beam search loop (a13): B*beams sequences, all bookkeeping on device; the final n-best lists land in lo.fin_ids / lo.fin_lp.
static int beam_loop(vitcap_engine* e, int B, const Layout& lo, const vitcap_gen_opts& o, char* ws, void* s) {
  const vitcap_weights& w = e->w;
  const int NS = lo.NS, L = lo.L, beams = o.num_beams;
  vitcap_beam_state st;
  st.ids_in = (int64_t*)(ws + lo.ids);
  st.ids_out = (int64_t*)(ws + lo.ids2);
  st.beam_scores = (float*)(ws + lo.beam_scores);
  st.parent = (int32_t*)(ws + lo.parent);
  st.done = (int32_t*)(ws + lo.done);
  st.has_hyp = (int32_t*)(ws + lo.has_hyp);
  st.hyp_score = (float*)(ws + lo.hyp_score);
  st.hyp_len = (int32_t*)(ws + lo.hyp_len);
  st.hyp_tok = (int64_t*)(ws + lo.hyp_tok);
  st.n_keep = o.num_keep_best;
  CK(vitcap_beam_init(&st, B, beams, L, o.bos_token_id, o.pad_token_id, s));
  char* tc_cur = ws + lo.tcache;
  char* tc_alt = ws + lo.tcache2;
  const int C = 2 * beams;
  // plain beam search: the candidates come from the vocabulary GEMM's row statistics (the 30522-wide rows are not read back:
  // row_topk_lse 139 us -> 12 us per step at 256 images x 5 beams); with a repetition penalty the logits change after the
  // GEMM, and the sampled form draws from the whole filtered row, so both keep the row scan
  const bool from_pieces = !o.sampling.do_sample && o.repetition_penalty == 1.0f;
  for (int t = 1; t < L; ++t) {
    CK(step_forward(w, lo, o, ws, t, st.ids_in, tc_cur, true, from_pieces, Part{0, NS, 0}, s));
    if (o.repetition_penalty != 1.0f)
      CK(vitcap_repetition_penalty((float*)(ws + lo.logits), VP, VITCAP_VOCAB, st.ids_in, L, t, o.repetition_penalty, NS, s));
    if (o.sampling.do_sample) {   // modeling_utils.py:966-985: two sampled words per beam instead of the 2*beams best
      CK(vitcap_beam_sample_candidates((const float*)(ws + lo.logits), VP, VITCAP_VOCAB, NS, t, &o.sampling, 0,
                                       (float*)(ws + lo.cand_val), (int32_t*)(ws + lo.cand_idx), (float*)(ws + lo.lse), s));
      CK(vitcap_beam_step_sampled((const float*)(ws + lo.cand_val), (const int32_t*)(ws + lo.cand_idx),
                                  (const float*)(ws + lo.lse), &st, B, beams, VITCAP_VOCAB, t, L, o.eos_token_id,
                                  o.pad_token_id, o.length_penalty, s));
    } else {
      if (from_pieces)
        CK(vitcap_row_topk_pieces((const float*)(ws + lo.logits), VP, VITCAP_VOCAB, (const float*)(ws + lo.rowstat), 2 * (VP / 64), C,
                                  (float*)(ws + lo.cand_val), (int32_t*)(ws + lo.cand_idx), (float*)(ws + lo.lse), NS, s));
      else
        CK(vitcap_row_topk_lse((const float*)(ws + lo.logits), VP, VITCAP_VOCAB, C, (float*)(ws + lo.cand_val),
                               (int32_t*)(ws + lo.cand_idx), (float*)(ws + lo.lse), NS, s));
      CK(vitcap_beam_step((const float*)(ws + lo.cand_val), (const int32_t*)(ws + lo.cand_idx),
                          (const float*)(ws + lo.lse), &st, B, beams, VITCAP_VOCAB, t, L, o.eos_token_id, o.pad_token_id,
                          o.length_penalty, s));
    }
    if (t + 1 < L) {   // re-order the text K/V history (positions 0..t-1) by parent beam for the next step
      CK(vitcap_beam_reorder_cache(tc_cur, tc_alt, st.parent, 4, NS, L, t, s));
      char* tmp = tc_cur; tc_cur = tc_alt; tc_alt = tmp;
    }
    int64_t* ti = st.ids_in; st.ids_in = st.ids_out; st.ids_out = ti;
  }
  // finalize reads hypotheses only; it runs whether or not the loop ended early
  CK(vitcap_beam_finalize(&st, (int64_t*)(ws + lo.fin_ids), (float*)(ws + lo.fin_lp), B, L, o.eos_token_id, o.pad_token_id, s));
  return VITCAP_OK;
}

// Constrained beam search loop (ViTCAP.generate with use_cbs, modeling_bert.py:1035-1057): B * S * num_beams
// sequences through the same decode step as beam search, bookkeeping of utils_cbs.py:26-443 on the device (csrc/cbs.hip).  The
// reference re-runs the whole model on every prefix (`state` stays None); here the text K/V caches follow the back-pointers.
static int cbs_loop(vitcap_engine* e, int B, const Layout& lo, const vitcap_gen_opts& o, char* ws, void* s) {
  const vitcap_weights& w = e->w;
  const int NS = lo.NS, L = lo.L, S = o.cbs_states, K = o.num_beams;
  vitcap_cbs_state st;
  st.ids_in = (int64_t*)(ws + lo.ids);
  st.ids_out = (int64_t*)(ws + lo.ids2);
  st.scores_in = (float*)(ws + lo.cbs_sc);
  st.scores_out = (float*)(ws + lo.cbs_sc2);
  st.parent = (int32_t*)(ws + lo.parent);
  st.unfinished = (int32_t*)(ws + lo.cbs_unf);
  st.n_pred = (int32_t*)(ws + lo.cbs_npred);
  st.live = (int32_t*)(ws + lo.live);
  CK(vitcap_cbs_init(&st, B, S, K, L, o.bos_token_id, s));
  CK(vitcap_cbs_pair_flags(o.fsm, B, S, VITCAP_VOCAB, (uint8_t*)(ws + lo.cbs_flags), s));
  char* tc_cur = ws + lo.tcache;
  char* tc_alt = ws + lo.tcache2;
  const float* logits = (const float*)(ws + lo.logits);
  float* lse = (float*)(ws + lo.lse);
  auto swap_state = [&] {
    int64_t* ti = st.ids_in; st.ids_in = st.ids_out; st.ids_out = ti;
    float* tf = st.scores_in; st.scores_in = st.scores_out; st.scores_out = tf;
  };
  for (int t = 1; t < L; ++t) {
    CK(step_forward(w, lo, o, ws, t, st.ids_in, tc_cur, true, false, Part{0, NS, 0}, s));
    CK(vitcap_row_topk_lse(logits, VP, VITCAP_VOCAB, 1, (float*)(ws + lo.cand_val), (int32_t*)(ws + lo.cand_idx), lse, NS, s));
    if (t == 1) {
      CK(vitcap_cbs_start(logits, VP, VITCAP_VOCAB, lse, o.fsm, &st, B, S, K, L, o.eos_token_id, o.eos_extra, s));
    } else {
      CK(vitcap_cbs_candidates(logits, VP, VITCAP_VOCAB, lse, o.fsm, &st, B, S, K, t, L, o.eos_token_id, o.eos_extra, o.cbs_no_repeat,
                               o.cbs_bad_ending, (const uint8_t*)(ws + lo.cbs_flags), (float*)(ws + lo.cbs_val),
                               (int32_t*)(ws + lo.cbs_word), s));
      CK(vitcap_cbs_select((const float*)(ws + lo.cbs_val), (const int32_t*)(ws + lo.cbs_word), &st, B, S, K, t, L, o.eos_token_id,
                           o.eos_extra, s));
    }
    if (t + 1 < L) {   // the text K/V history (positions 0..t-1) follows the back-pointers
      CK(vitcap_beam_reorder_cache(tc_cur, tc_alt, st.parent, 4, NS, L, t, s));
      char* tmp = tc_cur; tc_cur = tc_alt; tc_alt = tmp;
    }
    swap_state();
  }
  CK(vitcap_cbs_finalize(&st, o.num_constraints, o.min_constraints_to_satisfy, B, S, K, L, o.eos_token_id, o.eos_extra, o.pad_token_id,
                         (int64_t*)(ws + lo.fin_ids), (float*)(ws + lo.fin_lp), s));
  return VITCAP_OK;
}

static int decode_loop(vitcap_engine* e, int B, const Layout& lo, const vitcap_gen_opts& o, char* ws, void* s) {
  CallScope scope(e, o.gemm_mode, o.early_exit ? (const int32_t*)(ws + lo.live) : nullptr, &o);
  if (lo.cbs) return cbs_loop(e, B, lo, o, ws, s);
  return lo.beam ? beam_loop(e, B, lo, o, ws, s) : greedy_loop(e, lo, o, ws, s);
}

static int decode_locked(vitcap_engine* e, int B, const vitcap_gen_opts& o, const Layout& lo, char* ws, int64_t* out_ids,
                         float* out_logprobs, int64_t* out_last_tok, void* s) {
  if (!out_ids || !out_logprobs) { vitcap_set_error("decode: null outputs"); return VITCAP_EINVAL; }
  hipStream_t st = (hipStream_t)s;
  const bool graph = o.use_graph && !o.sampling.do_sample;
  if (!graph) {
    CK(decode_loop(e, B, lo, o, ws, s));
  } else {
    GraphEntry* hit = nullptr;
    for (auto& g : e->graphs)
      if (g.B == B && g.ws == (void*)ws && memcmp(&g.opts, &o, sizeof(o)) == 0) { hit = &g; break; }
    if (!hit) {
      // capture the loop once: every launch below becomes a kernel node with its arguments frozen (workspace pointers,
      // step index, option values), which is why the key holds all of them
      GraphEntry g;
      g.B = B; g.ws = (void*)ws; g.opts = o; g.graph = nullptr; g.exec = nullptr;
      if (!e->cap) HIPCK(hipStreamCreateWithFlags(&e->cap, hipStreamNonBlocking), "decode: capture stream");
      HIPCK(hipStreamBeginCapture(e->cap, hipStreamCaptureModeThreadLocal), "decode: begin capture");
      const int rc = decode_loop(e, B, lo, o, ws, (void*)e->cap);
      const hipError_t he = hipStreamEndCapture(e->cap, &g.graph);
      if (rc != VITCAP_OK) { if (g.graph) (void)hipGraphDestroy(g.graph); return rc; }
      HIPCK(he, "decode: end capture");
      HIPCK(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0), "decode: graph instantiate");
      if (e->graphs.size() >= 16) drop_graphs(e);        // bounded cache
      e->graphs.push_back(g);
      hit = &e->graphs.back();
    }
    HIPCK(hipGraphLaunch(hit->exec, st), "decode: graph launch");
  }
  const size_t L = (size_t)lo.L;
  if (lo.cbs) {       // [B][1][max_length]: the n_pred words of the selected beam (no BOS column), then pad; tap "cbs_npred" = n_pred
    HIPCK(hipMemcpyAsync(out_ids, ws + lo.fin_ids, (size_t)B * L * 8, hipMemcpyDeviceToDevice, st), "decode: output copy");
    HIPCK(hipMemcpyAsync(out_logprobs, ws + lo.fin_lp, (size_t)B * 4, hipMemcpyDeviceToDevice, st), "decode: output copy");
  } else if (lo.beam) {
    const size_t n = (size_t)B * o.num_keep_best;
    HIPCK(hipMemcpyAsync(out_ids, ws + lo.fin_ids, n * L * 8, hipMemcpyDeviceToDevice, st), "decode: output copy");
    HIPCK(hipMemcpyAsync(out_logprobs, ws + lo.fin_lp, n * 4, hipMemcpyDeviceToDevice, st), "decode: output copy");
  } else {
    HIPCK(hipMemcpyAsync(out_ids, ws + lo.ids, (size_t)lo.NS * L * 8, hipMemcpyDeviceToDevice, st), "decode: output copy");
    HIPCK(hipMemcpyAsync(out_logprobs, ws + lo.logprob, (size_t)lo.NS * 4, hipMemcpyDeviceToDevice, st), "decode: output copy");
    // the token chosen at the last position before the forced [SEP] (its log-probability is what the score holds)
    if (out_last_tok)
      HIPCK(hipMemcpyAsync(out_last_tok, ws + lo.last_tok, (size_t)lo.NS * 8, hipMemcpyDeviceToDevice, st), "decode: last-token copy");
  }
  return VITCAP_OK;
}

extern "C" int vitcap_engine_decode(vitcap_engine* e, int B, const vitcap_gen_opts* opts, void* workspace, size_t workspace_bytes,
                                    int64_t* out_ids, float* out_logprobs, int64_t* out_last_tok, void* s) {
  const vitcap_gen_opts o = opts ? *opts : default_opts();
  if (B <= 0 || check_opts(o) != VITCAP_OK) { if (B <= 0) vitcap_set_error("engine: bad batch"); return VITCAP_EINVAL; }
  const Layout lo(B, o);
  CK(check(e, B, o, workspace, workspace_bytes, lo.off));
  std::lock_guard<std::mutex> lk(e->mu);
  return decode_locked(e, B, o, lo, (char*)workspace, out_ids, out_logprobs, out_last_tok, s);
}

static int tags_copy(const Layout& lo, int B, char* ws, float* tag_logits_out, int64_t* tag_topk_out, void* s) {
  if (tag_logits_out)
    HIPCK(hipMemcpy2DAsync(tag_logits_out, (size_t)VITCAP_VOCAB * 4, ws + lo.tag_logits, (size_t)VP * 4, (size_t)VITCAP_VOCAB * 4,
                           B, hipMemcpyDeviceToDevice, (hipStream_t)s), "tags: logits copy");
  if (tag_topk_out)
    HIPCK(hipMemcpyAsync(tag_topk_out, ws + lo.tag_ids, (size_t)B * TOPK * 8, hipMemcpyDeviceToDevice, (hipStream_t)s), "tags: topk copy");
  return VITCAP_OK;
}

extern "C" int vitcap_engine_tags(vitcap_engine* e, int B, const vitcap_gen_opts* opts, void* workspace, float* tag_logits_out,
                                  int64_t* tag_topk_out, void* s) {
  const vitcap_gen_opts o = opts ? *opts : default_opts();
  if (!e || B <= 0 || !workspace || check_opts(o) != VITCAP_OK) { vitcap_set_error("tags: bad arguments"); return VITCAP_EINVAL; }
  return tags_copy(Layout(B, o), B, (char*)workspace, tag_logits_out, tag_topk_out, s);
}

extern "C" int vitcap_engine_generate(vitcap_engine* e, const void* image, int image_is_bf16, int B, const vitcap_gen_opts* opts,
                                      void* workspace, size_t workspace_bytes, int64_t* out_ids, float* out_logprobs,
                                      float* tag_logits_out, int64_t* tag_topk_out, void* s) {
  const vitcap_gen_opts o = opts ? *opts : default_opts();
  if (B <= 0 || check_opts(o) != VITCAP_OK) { if (B <= 0) vitcap_set_error("engine: bad batch"); return VITCAP_EINVAL; }
  const Layout lo(B, o);
  CK(check(e, B, o, workspace, workspace_bytes, lo.off));
  char* ws = (char*)workspace;
  std::lock_guard<std::mutex> lk(e->mu);
  CK(encode_locked(e, image, image_is_bf16, B, o, lo, ws, s));
  CK(prefill_locked(e, B, o, lo, ws, s));
  CK(decode_locked(e, B, o, lo, ws, out_ids, out_logprobs, nullptr, s));
  return tags_copy(lo, B, ws, tag_logits_out, tag_topk_out, s);
}

extern "C" const void* vitcap_engine_tap(vitcap_engine* e, const char* name, void* workspace, int B, const vitcap_gen_opts* opts) {
  const vitcap_gen_opts o = opts ? *opts : default_opts();
  if (!e || !name || !workspace || B <= 0 || check_opts(o) != VITCAP_OK) return nullptr;
  const Layout lo(B, o);
  char* ws = (char*)workspace;
  if (!strcmp(name, "last_token")) return ws + lo.last_tok;
  if (!strcmp(name, "hidden")) return ws + lo.x2;
  if (!strcmp(name, "tag_hidden")) return ws + lo.xt;
  if (!strcmp(name, "vis")) return ws + lo.vis_f;
  if (!strcmp(name, "logits_last")) return ws + lo.logits;
  if (!strcmp(name, "margins")) return ws + lo.margins;
  if (!strcmp(name, "tag_logits")) return ws + lo.tag_logits;
  if (!strcmp(name, "tag_prob")) return ws + lo.tag_prob;
  if (!strcmp(name, "tag_len")) return ws + lo.tag_len;
  if (!strcmp(name, "ids")) return ws + lo.ids;
  if (!strcmp(name, "live")) return ws + lo.live;
  if (!strcmp(name, "cbs_npred")) return lo.cbs ? ws + lo.cbs_npred : nullptr;
  return nullptr;
}
