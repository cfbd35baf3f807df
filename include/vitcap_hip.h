/*
 * vitcap_hip.h -- C ABI of libvitcap_hip.so, the MI355X (gfx950) implementation of the ViTCAP
 * captioning hot path (BASELINE.json north_star).
 *
 * The reference (jacobswan1/ViTCAP) is 100 % Python: it has no FFI/operator boundary at all, every
 * "kernel" is a stock torch op.  Each entry point below therefore cites the
 * reference *Python* code it replaces (file:line relative to the reference root), which is what a
 * maintainer would swap for a ctypes call.
 *
 * Conventions
 *   - plain `extern "C"`; raw device pointers; explicit sizes; no torch types.
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*): no allocation, no
 *     synchronisation, no ownership transfer -> hipGraph-capturable.
 *   - returns 0 on success, a negative VITCAP_E* code otherwise; vitcap_last_error() returns a
 *     thread-local message for the last failure.
 *   - matrices are row-major; "bf16" is the 16-bit truncated-exponent brain float stored as
 *     uint16_t; weights keep the nn.Linear layout W[N][K] (out_features x in_features).
 */
#ifndef VITCAP_HIP_H
#define VITCAP_HIP_H

#include <stddef.h>
#include <stdint.h>
#include "vitcap_jpeg.h"

#ifdef __cplusplus
extern "C" {
#endif

#define VITCAP_OK 0
#define VITCAP_EINVAL (-1)   /* bad shape / alignment / null pointer        */
#define VITCAP_ELAUNCH (-2)  /* hipLaunch / HIP runtime failure              */
#define VITCAP_EWORKSPACE (-3) /* caller workspace too small                 */
#define VITCAP_ESTATE (-4)   /* engine used before weights were bound        */

/* epilogue activation of vitcap_gemm_bias_act */
#define VITCAP_ACT_NONE 0
#define VITCAP_ACT_GELU_ERF 1 /* x*0.5*(1+erf(x/sqrt2)): nn.GELU (timm vision_transformer.py:142-158) and
                                 activations.py:16-24 `_gelu_python`                                        */
#define VITCAP_ACT_TANH 2     /* BertPooler (modeling_bert.py:515-527)                                       */

/* output dtype */
#define VITCAP_OUT_BF16 0
#define VITCAP_OUT_F32 1

const char* vitcap_last_error(void);
/* ABI version of this header: bumped whenever a struct layout or a function signature changes (6: vitcap_gen_opts.cbs_no_repeat / cbs_bad_ending, vitcap_cbs_candidates' two
 * arguments; 5: vitcap_gemm_desc.ln_*; 4:
 * vitcap_gen_opts' constrained-beam-search block + the vitcap_cbs_* entry points; 3: `abi` heads vitcap_gemm_desc and
 * vitcap_gen_opts; 2 -> 3 also covers round 3's additions: gemm_desc.colsum, gen_opts.eos_extra / tag_pos0, vitcap_tag_embed's pos0,
 * vitcap_layernorm_bwd's extra pointer, zout / aux carrying gelu').  vitcap_version() returns the library's value: a binding checks
 * the two for equality at load time, and every call that takes one of the two option structs rejects a struct whose first field is
 * not VITCAP_ABI_VERSION (a caller built against an older header passes a shorter struct: its fields would be misread). */
#define VITCAP_ABI_VERSION 6
int vitcap_version(void);

/* ------------------------------------------------------------------------------------------------
 * GEMM with fused epilogue:  C[M,N] = act( A[M,K] . W[N,K]^T + bias[N] ) (+ residual)
 *   replaces nn.Linear + activation + residual add in
 *     timm Attention.qkv/proj, Mlp.fc1/fc2 (vision_transformer.py:150-158, 176-200, 246-247),
 *     BertSelfAttention.query/key/value, BertSelfOutput.dense, BertIntermediate, BertOutput
 *     (modeling_bert.py:307-313, 353-357, 402-405, 415-419), BertPooler, BertPredictionHeadTransform,
 *     BertLMPredictionHead.decoder (modeling_bert.py:515-563) and the PatchEmbed conv written as a GEMM
 *     (vision_transformer.py:253-275).
 *   A, W: bf16, K % 64 == 0, lda/ldw in elements (multiples of 8, 16-byte aligned bases).
 *   bias: fp32[N] or NULL.  residual: fp32, same row mapping as C, ldr in elements, or NULL.
 *   C: bf16 or fp32 (out_dtype), N % 4 == 0, ldc % 4 == 0.
 *   Row remap (patch embed writes rows b*577+1+p, adds pos_embed[1+p]):
 *     if row_group > 0: out_row = (r / row_group) * out_group_rows + out_row_off + r % row_group
 *                       res_row = res_periodic ? (r % row_group) : out_row
 *     else out_row = res_row = r.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int abi;       /* VITCAP_ABI_VERSION (checked by every entry point that takes this struct) */
  int M, N, K;
  int lda, ldw, ldc, ldr;
  int act;
  int out_dtype;
  int row_group, out_group_rows, out_row_off, res_periodic;
  int tile_hint; /* 0 = auto; 1 = 64x64, 2 = 128x128, 3 = 256x128 three-stage, 4 = skinny (tuning/tests);
                    5 = 256x256 tiles, one per workgroup (what a caller that overlaps a second stream asks for: the other
                    stream's kernels fill the partial last round, and CU-time, not the launch's length, is what counts);
                    33 = the same with 192- or 128-row tiles for the rows behind the last whole round of the chip's CUs where
                    that shortens the launch (vitcap_gemm_tile_plan; auto picks it for M >= 2048 on an otherwise idle GPU);
                    30 / 31 = every tile 192 / 128 rows (measurements, tests); 12 = 256x256 persistent workgroups;
                    13 / 14 / 15 = 64x32 / 32x32 / 32x64 tiles on the 4-stage ring (decode-step shapes); 23 / 24 = 64x32 / 32x32
                    ring tiles writing K/768 raw fp32 slabs C[K/768][M][ldc] (K a multiple of 768 above it, no bias /
                    residual / activation: the consumer sums them, vitcap_sum_layernorm); 20 / 21 / 22 = the whole-K
                    "resident" forms of the same (finished output at K = 768, slabs above).
                    Tile shape never changes a result bit. */
  int split_k;   /* > 1: C is fp32 [split_k][M][ldc] partial slabs (no bias/act/residual applied); the consumer
                    (vitcap_sum_layernorm) reduces them.  K must be a multiple of 128*split_k. */
  const int32_t* live; /* optional device counter: the kernel returns at entry when *live == 0 (decode loop after every
                          sequence has finished -- the reference's `if cur_unfinished.max() == 0: break`,
                          modeling_utils.py:866; NULL = always run).  Honoured by the small-M (decode) kernels. */
  float* rowstat;      /* optional (any M), fp32 [M][2*ceil(N/64)][4]: per row and 32-column piece {max, column of the max (int bits,
                          lowest on ties), sum exp(x - max), 0} of the finished values -- what argmax / log_softmax over the
                          row are assembled from (vitcap_greedy_select_embed) without reading C back.  Needs fp32 output,
                          no activation / residual; 64x64 tiles. */
  float* colsum;       /* optional, fp32 [N]: colsum[n] += sum over the M rows of the finished bf16 OUTPUT values (atomic
                          accumulation, as vitcap_colsum_bf16 of C would add them) -- the bias gradient of the layer whose output
                          gradient this GEMM produces, without another pass over C.  Needs bf16 output, no residual / row remap /
                          split-K, M >= 2048, N and ldc multiples of 8 (the 256x256 kernel's 16-byte-store epilogue). */
  /* optional LayerNorm of the finished rows (what vitcap_layernorm_fwd over C would write, bit for bit; timm Block.norm1 / norm2
   * behind the residual adds, vision_transformer.py:246-247, BertSelfOutput / BertOutput.LayerNorm, modeling_bert.py:356, 418):
   * ln_out_bf16 [M][768] and / or ln_out_f32 [M][768] = LayerNorm(C[row]; ln_gamma, ln_beta, ln_eps).  Needs N == 768, ldc == 768, fp32
   * output, plain rows, no activation.  Where the launch form supports it the pass runs INSIDE the GEMM: the last of a 256-row
   * block's three column tiles to finish normalises the block (write-through output stores, one counter per row block in
   * ln_counters -- int32 [ceil(M / 128) + 8], zero before the first launch, left zero by every launch); otherwise a LayerNorm launch
   * follows the GEMM on the same stream.  The outputs may alias A (a row block's A rows are read by its own tiles only). */
  const float* ln_gamma;
  const float* ln_beta;
  float ln_eps;
  int ln_reserved;
  void* ln_out_bf16;
  float* ln_out_f32;
  int32_t* ln_counters;
} vitcap_gemm_desc;

int vitcap_gemm_bias_act(const void* A, const void* W, const float* bias, const float* residual,
                         void* C, const vitcap_gemm_desc* d, void* stream);
/* The tile plan the 256-column kernel uses for an M x N x K problem on the current device (host-side query, no launch):
 * plan[0] = number of 256-row m-tiles, plan[1] = height class of the tiles behind them (0 = none, 3 = 192 rows, 2 = 128 rows),
 * plan[2] = number of those m-tiles. */
int vitcap_gemm_tile_plan(int M, int N, int K, int* plan3);
/* Which kernel family runs an M x N x K GEMM with M >= 2048 under `tile_hint` (0 = auto, 5) on the current device (host-side query):
 * -1 = the 8-wave 256x256 kernel (gemm.hip); otherwise form + 10 * MI of the 4-wave kernel (gemm4w.hip): form 1 = one tile per
 * workgroup, 2 = persistent pipeline; MI = 8 / 7 / 6 -> 256- / 224- / 192-row tiles.  Results do not depend on the choice.
 * tile_hint | 0x100 asks about a bf16-output GEMM without residual (qkv, fc1): under tile_hint 5 those run the 4-wave one-tile form,
 * the fp32 + residual GEMMs the 8-wave kernel. */
int vitcap_gemm_large_form(int M, int N, int K, int tile_hint);
/* CUs the persistent large-GEMM grids (one 512-register workgroup per CU) leave free from now on, for launches of this process
 * (0 = none; rounded up to a multiple of 8 so that every XCD keeps the same number of workgroups; at most the device's CUs - 8).
 * Returns the previous value.  Why it exists: a collective's kernel (RCCL over xGMI, the data-parallel gradient exchange of
 * reference trainer.py:119-126 / uni_pipeline.py:497-505) cannot start while persistent workgroups own every CU -- stream priority
 * orders dispatch, it does not evict a resident workgroup -- so the training engine reserves a few CUs between the first
 * bucket's launch and the exchange's end (vitcap_amd/dist_util.py).  Host-side state (an atomic), no launch, no synchronisation. */
int vitcap_gemm_reserve_cus(int cus);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm over the last dim (768): y = (x-mean)/sqrt(var+eps)*gamma+beta, fp32 statistics.
 *   replaces nn.LayerNorm in timm Block.norm1/norm2 (eps 1e-6, vision_transformer.py:352, 233-247)
 *   and BertSelfOutput/BertOutput/BertEmbeddings/BertPredictionHeadTransform LayerNorm (eps 1e-12,
 *   modeling_bert.py:235, 356, 418, 543).
 *   x: fp32 [M, ldx]; y_bf16 and/or y_f32 may be NULL (at least one non-NULL), both [M,768] dense.
 * ---------------------------------------------------------------------------------------------- */
int vitcap_layernorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, float eps,
                         void* y_bf16, float* y_f32, int M, int D, void* stream);

/* Fused split-K reduction + bias + residual (+ GELU) + LayerNorm for the decode-step GEMMs:
 *   v[row] = sum_s partials[s][row] + bias (+ residual[row]);  if act_before_ln: v = gelu_erf(v);
 *   y = LayerNorm(v)   -- BertSelfOutput / BertOutput (modeling_bert.py:353-357, 415-419) and
 *   BertPredictionHeadTransform (modeling_bert.py:540-544) with the dense layer computed by split-K.
 *   partials fp32 [S][M][768] (slab_stride elements apart); residual fp32 [M, ldr] or NULL. */
int vitcap_sum_layernorm(const float* partials, int S, size_t slab_stride, const float* bias, const float* residual,
                         int ldr, int act_before_ln, const float* gamma, const float* beta, float eps,
                         void* y_bf16, float* y_f32, int M, int D, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Patch gather: image [B,3,384,384] (fp32 or bf16) -> A matrix [B*576, 768] bf16 with
 *   K index = c*256 + kh*16 + kw (the Conv2d(3,768,16,16) weight flattened), rows in (h,w) order.
 *   replaces the implicit im2col of PatchEmbed.proj (vision_transformer.py:273-274).
 *   Also writes the cls rows: x[b*577 + 0] = cls_token + pos_embed[0]   (vision_transformer.py:423-426)
 * ---------------------------------------------------------------------------------------------- */
int vitcap_patch_gather(const void* image, int image_is_bf16, void* patches_bf16, int B, void* stream);
int vitcap_cls_rows(const float* cls_token, const float* pos_embed, float* x, int B, int rows_per_image,
                    void* stream);

/* ------------------------------------------------------------------------------------------------
 * Dense (unmasked) multi-head self-attention over packed qkv, flash style.
 *   qkv: bf16 [B*S, 3*768], column = which*768 + head*64 + d  (timm Attention.forward reshape
 *   (B,N,3,12,64), vision_transformer.py:176-177; decoder Q/K/V weights concatenated give the same layout)
 *   out: bf16 [B*S, 768] = softmax(q k^T * scale) v per head, heads concatenated
 *   (vision_transformer.py:179-200 with the all-zero mask of modeling_bert.py:1415 dropped;
 *    modeling_bert.py:320-340 for the visual rows of the decoder, which attend visual rows only).
 * ---------------------------------------------------------------------------------------------- */
int vitcap_attn_dense_fwd(const void* qkv, void* out, int B, int S, float scale, void* stream);
/* Same, but only the first q_rows query rows of every image are wanted (rounded up to blocks of 128; all S keys count).
 * Used for the LAST tag block of TIMMVitSplitEncoder: of its output only the CLS row is ever read
 * (modeling_bert.py:1424 pooler(tag_hidden), 1493 tag_hidden[:, 0] as the first visual token). */
int vitcap_attn_dense_fwd_rows(const void* qkv, void* out, int B, int S, int q_rows, float scale, void* stream);

/* Training forms of the dense attention (ViT blocks and the visual rows of the decoder in
 * ViTCAP.encode_forward(is_training=True), modeling_bert.py:751-807): the forward additionally stores the log2-domain
 * logsumexp lse[B][12][S]; the backward returns dqkv (bf16, packed like qkv) from dout, recomputing the probabilities.
 * `dsum` is scratch fp32 [B][12][S]; `extra_dkv` (optional, bf16 [B*S][2][768]) is added to dK/dV. */
int vitcap_attn_dense_fwd_train(const void* qkv, void* out, float* lse, int B, int S, int ld_rows, float scale,
                                float p_drop, uint32_t drop_seed, int causal_from, int mask_from, void* stream);   /* ld_rows >= S: rows per image in qkv/out (decoder: 598 = 578 visual + 20 text) */
int vitcap_attn_dense_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* dsum,
                          const void* extra_dkv, void* dqkv, int B, int S, int ld_rows, float scale, float p_drop,
                          uint32_t drop_seed, int causal_from, int mask_from, void* stream);
/* The same two with the QUERY rows restricted to [q_lo, q_hi) (q_lo a multiple of 128; every key still counts): for layers
 * of which only some output rows are ever read -- the last tag block (CLS row) and the last decoder layer (caption rows).
 * Forward: out / lse are written for the 128-row blocks covering the range only.  Backward: dQ is written for those blocks
 * only (the caller zero-fills dqkv's Q columns elsewhere); dK / dV for every key, accumulated over the 64-row query tiles
 * that intersect the range -- rows of those tiles outside [q_lo, q_hi) must carry dout = 0 and finite out / lse. */
int vitcap_attn_dense_fwd_train_rows(const void* qkv, void* out, float* lse, int B, int S, int ld_rows, float scale,
                                     float p_drop, uint32_t drop_seed, int causal_from, int mask_from, int q_lo, int q_hi,
                                     void* stream);
int vitcap_attn_dense_bwd_rows(const void* qkv, const void* out, const void* dout, const float* lse, float* dsum,
                               const void* extra_dkv, void* dqkv, int B, int S, int ld_rows, float scale, float p_drop,
                               uint32_t drop_seed, int causal_from, int mask_from, int q_lo, int q_hi, void* stream);
/* causal_from > 0: the decoder's joint sequence under teacher forcing, rows [causal_from visual | S - causal_from caption]
 * per image with the seq2seq mask of dataset.py:377-390 + ..._bertemb.py:57-85: a visual row attends visual rows only,
 * caption row q attends every visual row and caption rows <= q.  All caption keys must fall in the last 64-key tile
 * (578 = 9*64 + 2 for ViT-B/16-384).  causal_from = 0: unmasked (ViT blocks).
 * mask_from > 0 (sampled-sequence log-probabilities with gradient, the SCST step): rows [mask_from, S) are [MASK] probe rows,
 * probe j sees the visual rows, caption tokens 0..j and itself, and is seen by nobody else -- the rows one decode
 * step's [MASK] query attends (modeling_bert.py:846-876), for all 19 steps in one pass. */
/* p_drop > 0 (decoder layers in training, attention_probs_dropout_prob of BertSelfAttention, modeling_bert.py:330-333):
 * the probabilities entering P.V are dropped with probability p_drop and the survivors scaled by 1/(1-p_drop); the
 * keep decision of (query row q, key row k) of image b, head h is the counter hash vc_drop_keep() of csrc/rng.h on
 * stream (drop_seed, b, h) -- the same function in the forward and both backward kernels, and
 * restated on the CPU by oracle.dropout_keep().  drop_seed must differ per layer and per step.  p_drop = 0: no dropout. */

/* ------------------------------------------------------------------------------------------------
 * Incremental decoder attention for one greedy step t (1..19), 2 query rows per sequence:
 *   row 0 = last real token (position t-1), row 1 = [MASK] (position t).
 *   qkv_step: bf16 [B*2, 2304]; vis_qkv: bf16 [B*S_vis, 2304] (K/V sections are the prefill cache);
 *   text_kv: bf16 [B, max_len, 2, 768] cache of text-row K/V for this layer, row t-1 is written here.
 *   out: bf16 [B*2, 768].
 *   Row 0 attends visual + text rows 0..t-1; row 1 additionally attends itself
 *   (mask of dataset.py:377-390 sliced as in modeling_bert.py:853-871; scores / 8 then softmax,
 *   modeling_bert.py:320-336).  `vis_image_stride_rows`: rows between consecutive images in vis_qkv;
 *   `seq_per_image` > 1 shares one image's visual K/V between several sequences (beams).
 * ---------------------------------------------------------------------------------------------- */
int vitcap_attn_decode_step(const void* qkv_step, const void* vis_qkv, void* text_kv, void* out,
                            int B, int S_vis, int t, int max_len, int seq_per_image, float scale,
                            void* stream);
/* The same with the predicted tag tokens visible to the caption rows (the mask CaptionTensorizer builds when
 * a text_b of n_tag tag tokens is attached, dataset.py:240-252, 387-390): n_tag more keys per image between the visual and the
 * text keys.  tag_qkv_a / tag_qkv_b: bf16 [B/seq_per_image][n_tag][2304] packed like vis_qkv, the tag rows' K/V under the two
 * embedding branches of modeling_bert.py:1435-1489; the step uses branch A iff tag_len[0] + 20 <= t + 51 -- the reference's own
 * test `topk_len[0] + 20 <= input_ids.shape[1]` (tag_len: int64 device array, element 0 is read). */
int vitcap_attn_decode_step_tags(const void* qkv_step, const void* vis_qkv, void* text_kv, void* out, int B, int S_vis, int t,
                                 int max_len, int seq_per_image, float scale, const void* tag_qkv_a, const void* tag_qkv_b,
                                 int n_tag, const int64_t* tag_len, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Text embedding for step t: rows (b,0)=word[ids[b][t-1]]+pos[t-1]+type[0], (b,1)=word[MASK]+pos[t]+type[0],
 * followed by LayerNorm(eps)   (BertEmbeddings.forward, modeling_bert.py:222-237).
 *   word/pos/type tables bf16; ids int64 [B, max_len]; outputs x_f32 [B*2,768], x_bf16 [B*2,768].
 * ---------------------------------------------------------------------------------------------- */
int vitcap_embed_step(const int64_t* ids, int max_len, int t, int mask_token,
                      const void* word_emb, const void* pos_emb, const void* type_emb,
                      const float* gamma, const float* beta, float eps,
                      float* x_f32, void* x_bf16, int B, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Greedy bookkeeping for step t over fp32 logits [B, ldl] (first V columns valid):
 *   tok = argmax (lowest index on ties); lp = log_softmax(logits)[tok];
 *   add = unfinished ? tok : pad; ids[b][t] = add; sum_lp += lp*unf; cnt += unf; unf &= (add != eos);
 *   at t == max_len-1 additionally ids[b][t] = eos where still unfinished, and
 *   logprob[b] = sum_lp / cnt is written          (modeling_utils.py:836-877).
 *   state: int32 unfinished[B]; float sum_lp[B]; float cnt[B].  t == 1 expects the caller to have run
 *   vitcap_greedy_init.
 * ---------------------------------------------------------------------------------------------- */
int vitcap_greedy_init(int64_t* ids, int32_t* unfinished, float* sum_lp, float* cnt, int B, int max_len,
                       int bos, int pad, void* stream);
int vitcap_greedy_step(const float* logits, int ldl, int V, int64_t* ids, int32_t* unfinished,
                       float* sum_lp, float* cnt, float* logprob_out, float* margin_out, int64_t* raw_last, int B, int t,
                       int max_len, int eos, int pad, void* stream);
/* raw_last (optional, int64 [B]): at t == max_len-1 the token actually chosen at the last position, before it is
 * overwritten by [SEP] for unfinished rows -- the returned log-prob is that of the chosen token (modeling_utils.py:
 * 850-877), which a teacher-forced re-computation of the sequence probability needs. */

/* Embeddings of the predicted tag tokens written over the last 50 text slots (ViTSplitCLSEmbModel.forward,
 * modeling_bert.py:1435-1489, encode_tag_to_embedding 1381-1406): rows (b, j < n), token = tag_ids[b][j] (int64 [B][50]; slot 49
 * forced to 102).  branch_a / tagemb_cls select the four forms: raw rows of the caption head's decoder matrix (A, 'cls'),
 * LN(word + pos[20+j] + type) (A, other), LN(cls_w + pos[20+j] + type) (B, 'cls') -- encode_tag_to_embedding's literal caption_len = 20 --
 * and bert.extra_embeddings LN_x(xword + xpos[pos0+j] + xtype) (B, other; the x* tables), the one form that takes the caller's
 * position_ids: pos0 = max(od_labels_start_posid, max_length) (modeling_bert.py:958-959, 983-992, 1484-1485), pos0 + n <= 512.
 * Outputs fp32 and bf16 [B*n][768]. */
int vitcap_tag_embed(const int64_t* tag_ids, int n, int pos0, int branch_a, int tagemb_cls, const void* cls_w, const void* word_emb,
                     const void* pos_emb, const void* type_emb, const float* gamma, const float* beta, const void* xword_emb,
                     const void* xpos_emb, const void* xtype_emb, const float* xgamma, const float* xbeta, float eps, float* x_f32,
                     void* x_bf16, int B, void* stream);
/* bf16 row blocks between per-image layouts: dst[(b*dst_img_rows + dst_row0 + r)*ld_dst + dst_col0 + c] =
 * src[(b*src_img_rows + src_row0 + r)*ld_src + src_col0 + c], r < rows, c < cols (multiples of 8 elements). */
int vitcap_copy_row_blocks(const void* src, int src_img_rows, int src_row0, int ld_src, int src_col0, void* dst, int dst_img_rows,
                           int dst_row0, int ld_dst, int dst_col0, int rows, int cols, int B, void* stream);

/* The same step taken from the vocabulary GEMM's row statistics (vitcap_gemm_desc.rowstat, `pieces` = 2*ceil(V_pad/64) per row)
 * instead of the logits, fused with BertEmbeddings.forward for step t+1 (vitcap_embed_step): token choice, log-prob and
 * bookkeeping exactly as vitcap_greedy_step; then x rows (b,0) = LN(word[ids[b][t]] + pos[t] + type[0]),
 * (b,1) = LN(word[mask] + pos[t+1] + type[0]) unless t == max_len-1.  One launch instead of three per decode step. */
int vitcap_greedy_select_embed(const float* rowstat, int pieces, int64_t* ids, int32_t* unfinished, float* sum_lp, float* cnt,
                               float* logprob_out, int64_t* raw_last, int B, int t, int max_len, int eos, int pad,
                               int mask_token, const void* word_emb, const void* pos_emb, const void* type_emb,
                               const float* gamma, const float* beta, float eps, float* x_f32, void* x_bf16, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Sampling variant of the step above (do_sample=True, modeling_utils.py:839-846 + top_k_top_p_filtering
 * 1103-1135): x = logits / temperature; top-k keeps x >= k-th largest (ties kept); top-p keeps a token iff the
 * probability mass ranked strictly above it is <= top_p; tok ~ softmax(filtered x); lp = log_softmax(filtered x)[tok]
 * (the reference scores the FILTERED distribution, 850-851).  The draw is argmax(x + Gumbel(seed, b, t, column)),
 * a counter-based pure function of its coordinates (csrc/rng.h), so runs are reproducible and checkable on the CPU.
 * Book-keeping identical to vitcap_greedy_step.  V <= 30720.
 * ---------------------------------------------------------------------------------------------- */
typedef struct vitcap_sample_params {
  int do_sample;       /* 0 = greedy */
  float temperature;   /* > 0 */
  int top_k;           /* 0 = off */
  float top_p;         /* 1 = off */
  uint32_t seed;
} vitcap_sample_params;
int vitcap_sample_step(const float* logits, int ldl, int V, int64_t* ids, int32_t* unfinished,
                       float* sum_lp, float* cnt, float* logprob_out, float* margin_out, int64_t* raw_last, int B, int t,
                       int max_len, int eos, int pad, const vitcap_sample_params* sp, void* stream);
/* the same for rows that are sequences seq_offset .. seq_offset + B - 1 of a larger call (row b draws from the random stream of
 * sequence seq_offset + b): a batch processed in slices draws what the whole batch would */
int vitcap_sample_step_offset(const float* logits, int ldl, int V, int64_t* ids, int32_t* unfinished,
                              float* sum_lp, float* cnt, float* logprob_out, float* margin_out, int64_t* raw_last, int B, int t,
                              int max_len, int eos, int pad, const vitcap_sample_params* sp, int seq_offset, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Tag head tail: prob = sigmoid(logit); top-k (largest, sorted, lowest index first on ties);
 *   topk_len = #(prob_topk >= thresh)        (modeling_bert.py:1428-1432)
 *   logits fp32 [B, ldl]; out_ids int64 [B,k]; out_prob fp32 [B,k]; out_len int64 [B].  k <= 64.
 * ---------------------------------------------------------------------------------------------- */
int vitcap_sigmoid_topk(const float* logits, int ldl, int V, int k, float thresh, int64_t* out_ids,
                        float* out_prob, int64_t* out_len, int B, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Beam search, device side (ViTCAP._generate_beam_search + BeamHypotheses, modeling_utils.py:888-1100, 1138-1180,
 * early_stopping = False):
 *   vitcap_row_topk_lse   per row: logsumexp over V and the k = 2*beams largest logits (sorted, lowest index first
 *                         on ties) -- the candidates `log_softmax(scores) + beam_scores` -> view(B, beams*V) ->
 *                         topk(2*beams) can take from that row                       (modeling_utils.py:988-996)
 *   vitcap_beam_step      per image: merge, sort, `is_done`, EOS / last-step candidates -> hypotheses, next beams,
 *                         re-ordered + extended input_ids, parent indices            (modeling_utils.py:1003-1054)
 *   vitcap_beam_reorder_cache   text K/V cache rows gathered by parent (the `past` re-ordering, 1056-1068)
 *   vitcap_beam_finalize  the n_keep best hypotheses + EOS, best first, padded to max_len -> out_ids [B][n_keep][max_len],
 *                         out_logprobs [B][n_keep] = sum_logprobs / len**lp, -1e5 where fewer finished (1076-1100)
 * All state lives in caller-provided device arrays.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int64_t* ids_in;    /* [B*beams][max_len] prefixes entering the step */
  int64_t* ids_out;   /* [B*beams][max_len] prefixes leaving the step (caller swaps) */
  float* beam_scores; /* [B*beams] */
  int32_t* parent;    /* [B*beams] row each new beam was expanded from */
  int32_t* done;      /* [B] */
  int32_t* has_hyp;   /* [B] number of hypotheses kept so far (0..n_keep) */
  float* hyp_score;   /* [B][n_keep] length-normalised scores of the kept hypotheses, insertion order */
  int32_t* hyp_len;   /* [B][n_keep] */
  int64_t* hyp_tok;   /* [B][n_keep][max_len] */
  int32_t n_keep;     /* BeamHypotheses.n_hyp = num_keep_best (modeling_utils.py:1138-1180), 1..8 */
} vitcap_beam_state;

int vitcap_row_topk_lse(const float* logits, int ldl, int V, int k, float* out_val, int32_t* out_idx,
                        float* out_lse, int rows, void* stream);
/* the same outputs from the vocabulary GEMM's row statistics (vitcap_gemm_desc.rowstat, `pieces` = 2*ceil(V_pad/64) per row): the k
 * pieces with the largest maxima hold the row's k largest logits, so only k x 32 of them are read back (vitcap_row_topk_lse reads
 * the 30522-wide row); identical values and indices, logsumexp assembled from the pieces (same value up to fp32 summation order) */
int vitcap_row_topk_pieces(const float* logits, int ldl, int V, const float* rowstat, int pieces, int k, float* out_val,
                           int32_t* out_idx, float* out_lse, int rows, void* stream);
int vitcap_beam_init(const vitcap_beam_state* s, int B, int beams, int max_len, int bos, int pad, void* stream);
int vitcap_beam_step(const float* cand_val, const int32_t* cand_idx, const float* lse, const vitcap_beam_state* s,
                     int B, int beams, int V, int t, int max_len, int eos, int pad, float length_penalty,
                     void* stream);
/* Beam search WITH sampling (num_beams > 1 and do_sample, modeling_utils.py:966-985).  vitcap_beam_sample_candidates: per row
 * (= beam) temperature, top_k_top_p_filtering with min_tokens_to_keep = 2 (k = max(top_k, 2); ranks 0..2 survive top-p), then TWO
 * words drawn without replacement from the softmax of what is left -- the two largest of (logit + Gumbel noise), the draw
 * torch.multinomial(p, 2) makes, on the counter-based stream (seed, row + row_offset, t).  out_val / out_idx: [rows][2] (the words'
 * temperature-scaled logits), out_lse: [rows] log-sum-exp of the surviving set.  vitcap_beam_step_sampled: vitcap_beam_step on
 * those candidates, consumed in position order with the reference's own beam attribution (position p of the (B, 2*beams) view is
 * scored with beam p / 2 but continues beam p % beams -- as written at modeling_utils.py:979-984). */
int vitcap_beam_sample_candidates(const float* logits, int ldl, int V, int rows, int t, const vitcap_sample_params* sp,
                                  int row_offset, float* out_val, int32_t* out_idx, float* out_lse, void* stream);
int vitcap_beam_step_sampled(const float* cand_val, const int32_t* cand_idx, const float* lse, const vitcap_beam_state* s,
                             int B, int beams, int V, int t, int max_len, int eos, int pad, float length_penalty,
                             void* stream);
int vitcap_beam_reorder_cache(const void* src, void* dst, const int32_t* parent, int layers, int n_seq, int max_len,
                              int t, void* stream);
int vitcap_beam_finalize(const vitcap_beam_state* s, int64_t* out_ids, float* out_logprobs, int B, int max_len,
                         int eos, int pad, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Constrained beam search, device side (ConstrainedBeamSearch.search with use_hypo = False and
 * select_best_beam_with_constraints, src/tools/captioning/utils_cbs.py:26-443).  G = S * K sequences per image, slot (s, k) =
 * beam k of FSM state s, global slot = b * G + s * K + k.
 *   vitcap_cbs_init        prefixes = [BOS], counters cleared
 *   vitcap_cbs_start       first step (:127-152): image b reads ROW b of the (B*G)-row logits (`[:batch_size]` of an image-major
 *                          batch, as written), words that fsm[b][0][i] does not allow score -inf, K best words per state i
 *   vitcap_cbs_candidates  later steps (:184-247): per slot and target state i the K best words of log_softmax(logits) -- minus the
 *                          slot's last word (no_repeat) and the EOS ids behind a bad-ending word; a slot ending in an EOS id
 *                          continues with an EOS id at cost 0 only -- masked by fsm[b][s][i] (-1e20)
 *   vitcap_cbs_select      (:245-319) per image and target state the K best of (candidate + the slot's running score) over all
 *                          slots; prefixes re-ordered by parent and extended, parents for the K/V cache re-ordering.  Ties go to
 *                          the lower flat index.  Once every slot of the batch ends in EOS the search stops (:177-181):
 *                          *n_pred = words per sequence so far, later calls pass the state through unchanged
 *   vitcap_cbs_finalize    (:377-443) per image the best beam of the main state s < 2**num_constraints[b] with at least
 *                          min(num_constraints[b], min_constraints) bits set that maximises score / (non-EOS words + 1); first
 *                          maximum on ties.  out_ids [B][max_len]: the n_pred words (no BOS column, like the reference's output),
 *                          then pad; out_logprobs [B].
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int64_t* ids_in;      /* [B*G][max_len] prefixes entering the step (BOS first) */
  int64_t* ids_out;     /* [B*G][max_len] prefixes leaving it (caller swaps) */
  float* scores_in;     /* [B*G] running log-probabilities (last_log_probabilities) */
  float* scores_out;
  int32_t* parent;      /* [B*G] global slot each slot's prefix was copied from */
  int32_t* unfinished;  /* [max_len] slots not ending in EOS after step t */
  int32_t* n_pred;      /* [1] words per sequence when the search stopped (max_len - 1 if it never did) */
  int32_t* live;        /* [1] 1 while the search runs (the step kernels of a stopped search return at entry) */
} vitcap_cbs_state;
int vitcap_cbs_init(const vitcap_cbs_state* s, int B, int S, int K, int max_len, int bos, void* stream);
int vitcap_cbs_start(const float* logits, int ldl, int V, const float* lse, const uint8_t* fsm, const vitcap_cbs_state* s,
                     int B, int S, int K, int max_len, int eos, const int32_t* eos_extra, void* stream);
/* flags [B][S][S] uint8 (device): 1 iff any word moves image b's machine from s1 to s2 -- computed once per call; handed to
 * vitcap_cbs_candidates (or NULL) it lets the pairs without a transition skip the scan of the vocabulary (same output) */
int vitcap_cbs_pair_flags(const uint8_t* fsm, int B, int S, int V, uint8_t* flags, void* stream);
/* no_repeat / bad_ending (host, 16 ids, -1 = unused, or NULL): vitcap_gen_opts.cbs_no_repeat / cbs_bad_ending */
int vitcap_cbs_candidates(const float* logits, int ldl, int V, const float* lse, const uint8_t* fsm, const vitcap_cbs_state* s,
                          int B, int S, int K, int t, int max_len, int eos, const int32_t* eos_extra, int no_repeat,
                          const int32_t* bad_ending, const uint8_t* pair_flags, float* cand_val, int32_t* cand_word, void* stream);
int vitcap_cbs_select(const float* cand_val, const int32_t* cand_word, const vitcap_cbs_state* s, int B, int S, int K, int t,
                      int max_len, int eos, const int32_t* eos_extra, void* stream);
int vitcap_cbs_finalize(const vitcap_cbs_state* s, const int64_t* num_constraints, int min_constraints, int B, int S, int K,
                        int max_len, int eos, const int32_t* eos_extra, int pad, int64_t* out_ids, float* out_logprobs,
                        void* stream);

/* Decode-step attention for SEVERAL sequences per image (beam search), one workgroup per (image, head): the 2*K <= 16 query rows
 * of an image are scored against its 578 visual key rows on the matrix pipe (K rows loaded once for all beams), the <= 41 text
 * keys of each sequence on the vector ALU; same softmax as vitcap_attn_decode_step (BertSelfAttention, modeling_bert.py:320-340),
 * results equal up to fp32 summation order.  vis_vt: per (image, head) transposed copy of the visual V rows, [n_images][12][64][608]
 * bf16 (keys padded with zeros), written once per batch by vitcap_attn_beam_vt from the packed visual q|k|v rows. */
int vitcap_attn_beam_vt(const void* vis_qkv, void* vis_vt, int n_images, int S_vis, void* stream);
int vitcap_attn_decode_beams(const void* qkv_step, const void* vis_qkv, const void* vis_vt, void* text_kv, void* out, int n_images,
                             int seq_per_image, int S_vis, int t, int max_len, float scale, void* stream);
/* the same for images that own SEVERAL groups of <= 8 sequences (constrained beam search: states x beams sequences per image, cut
 * into groups_per_image groups of seq_per_group): group g = sequences [g * seq_per_group, (g + 1) * seq_per_group) of the batch,
 * image g / groups_per_image */
int vitcap_attn_decode_beam_groups(const void* qkv_step, const void* vis_qkv, const void* vis_vt, void* text_kv, void* out,
                                   int n_images, int seq_per_group, int groups_per_image, int S_vis, int t, int max_len, float scale,
                                   void* stream);

/* small data movers used by the engine and exposed for tests */
int vitcap_assemble_visual(const float* hidden, const float* tag_hidden, float* vis_f32, void* vis_bf16,
                           int B, int n_tok, void* stream);  /* modeling_bert.py:1493 */
int vitcap_gather_rows_bf16(const float* x, int ldx_rows, void* out_bf16, int B, int D, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Engine: the whole greedy captioning forward (ImageCaptioning.forward test branch,
 *   ..._bertemb.py:87-184 -> ViTCAP.generate, modeling_bert.py:928-1059 ->
 *   _generate_no_beam_search, modeling_utils.py:768-886) enqueued by ONE call.
 * Weights are bound once as device pointers (packed by the host side: matrices bf16 in nn.Linear
 * layout, decoder query/key/value concatenated to [2304,768], vocab padded to VITCAP_VOCAB_PAD rows,
 * vectors fp32).
 * ---------------------------------------------------------------------------------------------- */
#define VITCAP_VOCAB 30522
#define VITCAP_VOCAB_PAD 30592 /* 239 * 128 */
#define VITCAP_HID 768
#define VITCAP_NVIS 577
#define VITCAP_MAXLEN 20

typedef struct {
  const void* qkv_w;  const float* qkv_b;    /* [2304,768] bf16, [2304] */
  const void* proj_w; const float* proj_b;   /* [768,768]               */
  const void* fc1_w;  const float* fc1_b;    /* [3072,768]              */
  const void* fc2_w;  const float* fc2_b;    /* [768,3072]              */
  const float* n1_g; const float* n1_b; const float* n2_g; const float* n2_b;
} vitcap_vit_block_w;

typedef struct {
  const void* qkv_w;  const float* qkv_b;    /* query|key|value stacked: [2304,768] */
  const void* ao_w;   const float* ao_b;     /* attention.output.dense [768,768]   */
  const float* ao_g;  const float* ao_beta;  /* attention.output.LayerNorm        */
  const void* i_w;    const float* i_b;      /* intermediate.dense [3072,768]     */
  const void* o_w;    const float* o_b;      /* output.dense [768,3072]           */
  const float* o_g;   const float* o_beta;   /* output.LayerNorm                  */
} vitcap_bert_layer_w;

typedef struct {
  const void* dense_w; const float* dense_b; /* transform.dense [768,768]          */
  const float* ln_g;   const float* ln_b;    /* transform.LayerNorm                */
  const void* dec_w;                          /* decoder.weight [VOCAB_PAD,768], pad rows zero */
  const float* dec_b;                         /* predictions.bias [VOCAB_PAD], pad = -1e30    */
} vitcap_lm_head_w;

typedef struct {
  const void* patch_w; const float* patch_b;       /* [768, 768(c,kh,kw)] bf16, [768]            */
  const float* cls_token; const float* pos_embed;  /* fp32 [768], [577,768]                      */
  vitcap_vit_block_w blocks[12];
  vitcap_vit_block_w tag_blocks[4];
  const void* pooler_w; const float* pooler_b;     /* bert.pooler.dense                          */
  vitcap_lm_head_w tag_logit;
  const void* word_emb; const void* pos_emb; const void* type_emb; /* bf16 [VOCAB_PAD|512|2, 768] */
  const float* emb_ln_g; const float* emb_ln_b;
  vitcap_bert_layer_w dec[4];
  vitcap_lm_head_w cls;
  /* bert.extra_embeddings (tag rows under branch B when tagemb != 'cls', modeling_bert.py:1484-1485): OPTIONAL, may be NULL
   * unless vitcap_gen_opts.tag_visible > 0 with tagemb_cls == 0 */
  const void* xword_emb; const void* xpos_emb; const void* xtype_emb;
  const float* xemb_ln_g; const float* xemb_ln_b;
} vitcap_weights;

typedef struct vitcap_engine vitcap_engine;

int vitcap_engine_create(vitcap_engine** out);
void vitcap_engine_destroy(vitcap_engine* e);
int vitcap_engine_bind_weights(vitcap_engine* e, const vitcap_weights* w);

/* Options of one generate() call -- the kwargs ViTCAP.generate takes (modeling_bert.py:928-933), which the pipeline passes
 * through test_extra_input (..._bertemb.py:588-608), plus how the work is launched.  Passed BY VALUE SEMANTICS to every
 * engine call (the engine keeps no option state between calls; two host threads may drive one engine with different
 * options on different streams and workspaces).  NULL anywhere below means vitcap_gen_opts_init() defaults. */
#define VITCAP_MAXLEN_CAP 40     /* largest max_length the decode kernels are sized for (max_seq_a_length default 40) */
#define VITCAP_GEMM_AUTO 0       /* large GEMMs as persistent workgroups where that wins (the GEMM has the GPU to itself) */
#define VITCAP_GEMM_TILES 1      /* one tile per workgroup: a second stream's small kernels interleave (batch pipeline) */
typedef struct vitcap_gen_opts {
  int32_t abi;                /* VITCAP_ABI_VERSION: set by vitcap_gen_opts_init, checked by vitcap_gen_opts_check and every engine call */
  int32_t num_beams;          /* 1 = greedy / sampling (_generate_no_beam_search); 2..8 = beam search                  */
  int32_t seqs_per_image;     /* num_return_sequences, 1..8; > 1 needs num_beams == 1 (inputs expanded,
                                 modeling_bert.py:976-994; the copies share the image's encoder output and visual K/V)  */
  int32_t num_keep_best;      /* BeamHypotheses.n_hyp, 1..8; > 1 needs num_beams > 1 (modeling_utils.py:790)            */
  int32_t max_length;         /* 2..VITCAP_MAXLEN_CAP; output rows are max_length wide                                  */
  int32_t bos_token_id, eos_token_id, pad_token_id, mask_token_id;   /* 101 / 102 / 0 / 103; eos_token_id = eos_token_ids[0]
                                                                        (further ids: eos_extra below)                     */
  float length_penalty;       /* beam search                                                                            */
  float repetition_penalty;   /* CTRL penalty, 1 = off (modeling_utils.py:828-836, 955-963)                             */
  vitcap_sample_params sampling;   /* do_sample / temperature / top_k / top_p / seed (with num_beams > 1: beam sampling) */
  int32_t gemm_mode;          /* VITCAP_GEMM_AUTO | VITCAP_GEMM_TILES                                                   */
  int32_t early_exit;         /* 1 (default): once every sequence (beam search: image) has finished, the remaining
                                 steps' kernels return at entry -- `if cur_unfinished.max() == 0: break`
                                 (modeling_utils.py:866, 1072) without a host synchronisation; results are identical   */
  int32_t use_graph;          /* 1: the decode loop is captured once per (B, workspace, options) into a hipGraph owned
                                 by the engine and replayed by later calls (greedy and beam search; not sampling, whose
                                 seed changes per call)                                                                 */
  int32_t tag_visible;        /* n in 0..50: the first n predicted tag tokens are VISIBLE to the caption rows and to each other
                                 (the mask tensorize_ab builds for a text_b of n tokens, dataset.py:240-252, 387-390; SURVEY 8f
                                 rank 4).  0 = the shipped test mask (nothing attends the tag slots).  Needs max_length == 20.   */
  int32_t tagemb_cls;         /* model config `tagemb == 'cls'` (YAML tagemb: cls): which embedding the tag rows take
                                 (modeling_bert.py:1454-1489); only read when tag_visible > 0                                   */
  int32_t decode_streams;     /* greedy / sampling loop: 2 = the batch is cut into two slices that decode on two streams (the
                                 second one engine-owned, forked from and joined to the caller's); 1 = one chain; 0 = auto
                                 (= 1: measured, the second chain does not hide the per-kernel latency).  Same results.   */
  int32_t encode_parts;       /* encoder + prefill of the batch as 1..4 independent chains of batch parts on separate streams
                                 (engine-owned, forked from / joined to the caller's): one part's GEMM tails are filled by
                                 the other's kernels (2 parts: +1..2 % images/s at 32..128 images); 0 = auto = 1.  Same results.  */
  int32_t eos_extra[3];       /* eos_token_ids[1..3], -1 = unused: the greedy / sampling loop finishes a sequence at ANY of the ids
                                 (modeling_utils.py:862-865) and forces eos_token_ids[0] at the last position (:870-871).  Needs
                                 num_beams == 1: the reference's beam search asserts with several ids (modeling_utils.py:1037).  */
  int32_t tag_pos0;           /* position id of tag slot 0 = max(od_labels_start_posid, max_length) (modeling_bert.py:958-959,
                                 983-992; the pipeline passes od_labels_start_posid = max_seq_a_length, ..._bertemb.py:597: 20 in
                                 the shipped YAML, 40 by the pipeline's default).  20 (default) .. 462.  Reaches the model through
                                 ONE path only, as in the reference: bert.extra_embeddings of the tag rows (tagemb != 'cls',
                                 branch B, modeling_bert.py:1484-1485; every other tag embedding uses the literal 20 of
                                 encode_tag_to_embedding), so it is read when tag_visible > 0 and tagemb_cls == 0             */
  /* constrained beam search: ViTCAP.generate(use_cbs=True, fsm=..., num_constraints=..., min_constraints_to_satisfy=...)
     (modeling_bert.py:932-933, 949-953, 1035-1057 -> src/tools/captioning/utils_cbs.py:26-443).  Every image then decodes
     cbs_states * num_beams sequences (num_beams per state of its finite-state machine) */
  int32_t use_cbs;            /* 1: constrained beam search; needs fsm, num_constraints, no sampling / penalty / n-best        */
  int32_t cbs_states;         /* S = fsm.shape[1] (2**max_given_constraints main states + the sub-states in use), 1..32        */
  int32_t min_constraints_to_satisfy;   /* the pipeline passes 2 (..._bertemb.py:175-179)                                    */
  int32_t cbs_no_repeat;      /* generate's decoding_constraint_flag: a live sequence may not repeat its last word (utils_cbs.py:187-190) */
  const uint8_t* fsm;         /* device, uint8 [B][S][S][30522]: fsm[b][s1][s2][w] != 0 iff word w moves image b's machine from
                                 state s1 to s2 (FiniteStateMachineBuilder.build, utils_cbs.py:733-871)                        */
  const int64_t* num_constraints;       /* device, int64 [B]: constraints given per image (2**n main states can be valid)      */
  int32_t cbs_bad_ending[16]; /* generate's bad_ending_ids, -1 = unused: a live sequence whose last word is one of them may not end --
                                 every EOS id scores -inf (utils_cbs.py:192-198)                                               */
} vitcap_gen_opts;
void vitcap_gen_opts_init(vitcap_gen_opts* o);
/* VITCAP_OK or VITCAP_EINVAL with the offending field in vitcap_last_error() */
int vitcap_gen_opts_check(const vitcap_gen_opts* o);

/* bytes of caller-provided workspace for B images under `opts` (beams / seqs_per_image / max_length size it) */
size_t vitcap_engine_workspace_bytes(int B, const vitcap_gen_opts* opts);

/* The whole captioning forward (ImageCaptioning.forward test branch, ..._bertemb.py:87-184 -> ViTCAP.generate) by ONE call.
 * image: [B,3,384,384] fp32 or bf16, normalised as the reference's Normalize(.5,.5).
 * Outputs (what ViTCAP.generate returns, modeling_utils.py:883-886 / 1097-1100), rows max_length wide:
 *   num_beams == 1: out_ids int64 [B*seqs_per_image, 1, max_length], out_logprobs fp32 [B*seqs_per_image, 1], image-major;
 *   num_beams  > 1: out_ids int64 [B, num_keep_best, max_length], out_logprobs fp32 [B, num_keep_best], best first, rows of
 *                   images with fewer finished hypotheses padded with pad / -1e5.
 * Optional taps (may be NULL): tag_logits fp32 [B,30522], tag_topk int64 [B,50] (modeling_bert.py:1424-1432). */
int vitcap_engine_generate(vitcap_engine* e, const void* image, int image_is_bf16, int B, const vitcap_gen_opts* opts,
                           void* workspace, size_t workspace_bytes, int64_t* out_ids, float* out_logprobs,
                           float* tag_logits_out, int64_t* tag_topk_out, void* stream);

/* The three stages of vitcap_engine_generate on one workspace, for callers that overlap them across batches (encoder +
 * prefill of batch i+1 on one stream, decode of batch i on another) and for profiling / per-stage parity:
 *   encode  a1-a6: patch embed, 12 + 4 ViT blocks, tag head;  prefill  a8/a9: visual rows through the decoder once, K/V kept;
 *   decode  a9-a13: the step loop + bookkeeping; out_last_tok (optional, num_beams == 1) [B*seqs] = the token chosen at the
 *   last position, which the returned ids overwrite with the forced [SEP] (modeling_utils.py:870-871). */
int vitcap_engine_encode(vitcap_engine* e, const void* image, int image_is_bf16, int B, const vitcap_gen_opts* opts,
                         void* workspace, size_t workspace_bytes, void* stream);
int vitcap_engine_prefill(vitcap_engine* e, int B, const vitcap_gen_opts* opts, void* workspace, size_t workspace_bytes,
                          void* stream);
int vitcap_engine_decode(vitcap_engine* e, int B, const vitcap_gen_opts* opts, void* workspace, size_t workspace_bytes,
                         int64_t* out_ids, float* out_logprobs, int64_t* out_last_tok, void* stream);
/* copies the tag head's outputs of the last encode on this workspace (either pointer may be NULL) */
int vitcap_engine_tags(vitcap_engine* e, int B, const vitcap_gen_opts* opts, void* workspace, float* tag_logits_out,
                       int64_t* tag_topk_out, void* stream);
/* number of decode-loop hipGraphs the engine currently holds (tests) */
int vitcap_engine_graph_count(vitcap_engine* e);

/* CTRL repetition penalty kernel alone: rows of logits [rows][ldl], prefixes ids[rows][ld_ids][0..t). */
int vitcap_repetition_penalty(float* logits, int ldl, int V, const int64_t* ids, int ld_ids, int t, float penalty, int rows,
                              void* stream);

/* Debug/parity taps into the workspace after a generate call (device pointers, valid until the next call): name in
 * {"hidden","tag_hidden","vis","logits_last","margins","tag_logits","tag_prob","tag_len","ids","last_token","live"} */
const void* vitcap_engine_tap(vitcap_engine* e, const char* name, void* workspace, int B, const vitcap_gen_opts* opts);

/* ================================================================================================
 * Cross-entropy TRAINING step (ViTCAP.encode_forward(is_training=True) + backward + optimizer:
 * modeling_bert.py:751-807, 661-690; loss.py:5-22; trainer.py:95-142; optimization.py:151-210).
 * Backward GEMMs reuse vitcap_gemm_*: dX = dY . W is the NT kernel on a transposed weight copy, dW = dY^T . X is the
 * NT kernel on transposed activations with split-K into fp32 slabs.
 * ============================================================================================== */
/* GEMM with training extras.  Forward, act == gelu_erf: `zout` (bf16 [M][ldz]) receives gelu'(pre-activation) = Phi(z) + z phi(z),
 * evaluated in fp32 from the same erfc as the activation itself.  Backward (nn.GELU / `_gelu_python`): `aux` (bf16 [M][ldaux]) =
 * that stored factor multiplies the result -- the input-gradient GEMM's epilogue is one multiply per element instead of a gelu'
 * evaluation from a stored z (measured: docs/LAB_r01_r04.md 7).  With
 * d->split_k > 1 and M > 256 (weight gradients) K is split raggedly and C is fp32 [split_k][M][ldc]. */
int vitcap_gemm_ex(const void* A, const void* W, const float* bias, const float* residual, void* C,
                   const vitcap_gemm_desc* d, const void* aux_bf16, int ldaux, void* zout_bf16, int ldz, void* stream);
/* XT[c][r] = X[r][c] (bf16), r padded with zeros up to ldt (multiple of 64); colsum[c] += sum_r X[r][c] (bias grads) */
int vitcap_transpose_colsum(const void* x, int ldx, void* xt, int ldt, float* colsum, int R, int C, void* stream);
/* LayerNorm backward (nn.LayerNorm): dx = LNbwd(dy; x, gamma) + dres;  dgamma/dbeta accumulated with atomics */
int vitcap_layernorm_bwd(const float* x, int ldx, const void* dy, int dy_is_f32, const float* gamma, float eps,
                         const float* dres, float* dx_f32, void* dx_bf16, float* dgamma, float* dbeta, float* dx_bf16_colsum,
                         int M, int D, void* stream);
/* dx_bf16_colsum (optional, fp32 [768]): += column sums of the bf16-rounded dx (the bias gradient of the linear layer whose
 * output gradient dx is; what vitcap_colsum_bf16(dx_bf16) would add). */
int vitcap_reduce_slabs(const float* slabs, size_t slab_stride, int S, float* out, size_t n, int accumulate, void* stream);
int vitcap_cast_bf16(const float* x, void* y, size_t n, void* stream);
/* Hidden-state dropout of the BERT parts in training mode -- nn.Dropout(config.hidden_dropout_prob) in BertEmbeddings
 * (modeling_bert.py:236), BertSelfOutput (:355) and BertOutput (:417); the pipeline's `drop_out`:
 *   out[m][d] = x[m][d] * keep / (1 - p) (+ residual[m][d]); keep is a counter-based function of (seed, m / rows_per_seq,
 *   row0 + m % rows_per_seq, d), recomputed (not stored) by the backward pass, which is the same call on the gradient: dx = dy * keep / (1 - p).
 * x, residual (or NULL), out: fp32 [M][768]; out may alias x. */
int vitcap_hidden_dropout(const float* x, const float* residual, float* out, int M, int D, int rows_per_seq, int row0,
                          uint32_t seed, float p, void* stream);
/* Dropout salt: a device-resident uint32 that the training kernels XOR into every dropout seed they are launched with (attention
 * dropout of vitcap_attn_dense_fwd_train* / vitcap_attn_dense_bwd*, vitcap_hidden_dropout; modeling_bert.py:236, 330-333, 356, 418).
 * Seeds are launch arguments and therefore frozen inside a captured hipGraph; a captured training step draws new keep decisions on
 * every replay by rewriting this word before the replay.  Per calling thread; NULL (the default) = no salt.  Returns 0. */
int vitcap_set_dropout_salt(const void* device_u32);

/* the same over [M][768] rows, with colsum[c] += sum over rows of the ROUNDED values (cast + vitcap_colsum_bf16 in one pass) */
int vitcap_cast_bf16_colsum(const float* x, void* y, float* colsum, int M, int D, void* stream);
/* BertEmbeddings backward: scatter-add into word / position / token-type gradient tables (modeling_bert.py:230-234) */
int vitcap_embed_bwd(const float* d, const int64_t* ids, int rows_per_seq, float* gword, float* gpos, float* gtype,
                     int rows, int pos_wrap, void* stream);
/* BertCaptioningLoss (label-smoothed KL, mean over rows; modeling_bert.py:661-690): loss_sum += loss; dlogits (bf16,
 * [rows][ldd], columns >= V zeroed) = d loss / d logits */
int vitcap_ls_kl_loss(const float* logits, int ldl, int V, const int64_t* target, float eps, int rows,
                      const float* row_weight, float* loss_sum, void* dlogits_bf16, int ldd, void* stream);
/* row_weight (optional, fp32 [rows]): row i contributes row_weight[i] * loss_i instead of loss_i / rows -- with eps = 0
 * the self-critical policy-gradient loss -mean_s(reward_s * mean_t log p(token_st)) (ScstRewardCriterion.forward,
 * src/tools/captioning/utils_caption_evaluate.py:172-202) with row_weight = reward_s / (n_s * S). */
/* FocalLossWithLogitsNegLoss(alpha, gamma=1).sum()  (loss.py:5-22, modeling_bert.py:789-791) */
int vitcap_focal_loss_sum(const float* logits, int ldl, int V, const float* label, float alpha, float* out, int B,
                          void* stream);
/* torch.nn.BCEWithLogitsLoss() (mean over B x V) -- the tag loss when the configuration's `loss` is not 'focal'
 * (modeling_bert.py:713-717): out[0] += mean(max(x,0) - x*y + log1p(exp(-|x|))) */
int vitcap_bce_logits_mean(const float* logits, int ldl, int V, const float* label, float* out, int B, void* stream);
/* out[0] = sum g[i]^2 (overwritten, not accumulated), fixed summation order: bit-reproducible, so that data-parallel ranks
 * derive the same clip coefficient from the same all-reduced gradient (torch.nn.utils.clip_grad_norm_, trainer.py:124). */
int vitcap_sumsq(const float* g, size_t n, float* out, void* stream);
/* clip_grad_norm_(max_norm=clip) + solver.AdamW.step over a flat parameter buffer in 1024-element chunks; chunk_lr = 0
 * marks chunks the optimizer does not own (trainer.py:124-142, optimization.py:187-208, ..._bertemb.py:306-356) */
int vitcap_adamw_multi(float* p, const float* g, float* m, float* v, const float* chunk_lr, const float* chunk_wd,
                       const float* gsumsq, float clip, float lr_scale, int step, float b1, float b2, float eps,
                       size_t nchunks, void* stream);
/* Weight gradient of an nn.Linear, dW[N][K] = dY^T X (autograd of F.linear, e.g. modeling_bert.py:283-287 /
 * vision_transformer.py:147-150), straight from the row-major backward operands: Y = dY bf16 [M][ldy] (N columns),
 * X = saved input bf16 [M][ldx] (K columns), reduction over the M = batch x tokens rows.  The M range is cut into
 * `splits` pieces, piece s writes the fp32 slab C_slabs[s][N][K] (deterministic; sum them with vitcap_reduce_slabs).
 * N and K multiples of 256. */
int vitcap_gemm_tn(const void* Y, int ldy, const void* X, int ldx, float* C_slabs, int M, int N, int K, int splits,
                   void* stream);
/* The same with the sum over the splits done by the launch itself: out[N][K] (+)= sum_s C_slabs[s] in slab order (the result of
 * vitcap_gemm_tn followed by vitcap_reduce_slabs, bit for bit; C_slabs is scratch).  The workgroups of a tile wait for each other
 * (slabs published by write-through `sc1` stores + drained vmcnt + a relaxed agent-scope ticket; the readers take an agent-scope
 * acquire), so tiles x splits must not exceed the number of CUs (an error otherwise) AND the launch needs those CUs to itself: a caller
 * must not run it beside kernels of another stream that hold CUs for long (a collective, a persistent GEMM grid) -- a workgroup that is
 * not resident yet is waited for by its resident siblings.  Off in the training step (measured slower than reduce_slabs, LAB r05 4). */
int vitcap_gemm_tn_sum(const void* Y, int ldy, const void* X, int ldx, float* C_slabs, float* out, int accumulate, int M, int N,
                       int K, int splits, void* stream);
/* bias gradient: out[n] += sum_m y[m][n]  (bf16 [M][ldy] -> fp32 [N], atomic accumulation into `out`) */
int vitcap_colsum_bf16(const void* y, int ldy, int M, int N, float* out, void* stream);
/* fp32 master W[N][K] -> bf16 W and bf16 W^T[K][N] (the operands of the forward and the dgrad GEMMs) */
int vitcap_cast_transpose(const float* w, void* w_bf16, void* wt_bf16, int N, int K, int ldt, void* stream);
/* The same for a table of matrices in one launch (all GEMM weights after an optimizer step).  `items` lives in DEVICE memory,
 * sorted by tile0 = the running sum of ceil(N/64) * (K/64) over the preceding items (item 0: 0); total_tiles = that sum over all
 * items.  Per item: K a multiple of 64, ldt >= N and a multiple of 8, w_bf16 or wt_bf16 may be NULL.  The caller validates the
 * table (it is read on the device). */
typedef struct {
  const float* w;
  void* w_bf16;
  void* wt_bf16;
  int32_t N, K, ldt, tile0;
} vitcap_ct_item;
int vitcap_cast_transpose_multi(const vitcap_ct_item* items_dev, int n_items, int total_tiles, void* stream);
/* dz = dg * f, f (bf16) = the gelu'(pre-activation) factor a forward vitcap_gemm_ex stored through `zout` */
int vitcap_gelu_bwd(const float* dg, const void* gelu_grad_bf16, void* dz_bf16, size_t n, void* stream);
int vitcap_sum_over_batch(const float* x, size_t stride, int B, float* out, size_t n, void* stream);
/* BertEmbeddings.forward on all rows of the teacher-forced caption (modeling_bert.py:222-237): optional pre-LN sum */
int vitcap_embed_rows(const int64_t* ids, int rows_per_seq, const void* word_emb, const void* pos_emb,
                      const void* type_emb, const float* gamma, const float* beta, float eps, float* pre_f32,
                      float* x_f32, void* x_bf16, int rows, int pos_wrap, void* stream);
/* pos_wrap > 0: rows r >= pos_wrap of a sequence are [MASK] probe rows at positions r - pos_wrap + 1 (0 = positions = r) */
/* Live per-launch timing of the large-tile GEMM kernel (bench.py roofline): hipEvents recorded on the launch
 * stream around every GEMM launch with M > 256.  begin() sizes the event pool (outside the timed region);
 * end() synchronises and returns sums per epilogue variant (index = act*4 + out_f32*2 + has_residual). */
int vitcap_engine_timing_begin(vitcap_engine* e, int max_launches);
/* time the large-GEMM launches of every stride-th STEP (vitcap_engine_encode call + the prefill behind it) from the next timing_begin
 * on (default 1 = every step): four event records per timed launch cost 2.6 % of a pipelined B = 64 step when every launch is timed
 * (bench.py `unarmed_ms_per_step`).  Whole steps, so that the busy-interval union of the sampled launches stays meaningful */
int vitcap_engine_timing_sample(vitcap_engine* e, int stride);
int vitcap_engine_timing_end(vitcap_engine* e, double* ms12, double* flops12, int* launches12);
/* the same plus busy_ms12: per variant the length of the union of its launches' [start, stop] intervals (launches of one kernel
 * overlap when several chains are in flight; their summed durations count that time twice) */
int vitcap_engine_timing_end_ex(vitcap_engine* e, double* ms12, double* flops12, int* launches12, double* busy_ms12);
/* the same plus kernel_ms12 / kernel_busy_ms12: sums and union from HIP events bound to the KERNEL DISPATCHES themselves
 * (hipExtLaunchKernelGGL start / stop: the kernel begins executing -> has completed, the interval rocprofv3 --kernel-trace
 * reports); the stream-marker brackets above additionally hold the time a dispatch waited behind another stream's kernels */
int vitcap_engine_timing_end_kernel(vitcap_engine* e, double* ms12, double* flops12, int* launches12, double* busy_ms12,
                                    double* kernel_ms12, double* kernel_busy_ms12);

/* ------------------------------------------------------------------------------------------------
 * Input side of the path: the reference's test-time image transform, get_transform_vit_default
 * (src/pipelines/uni_pipeline.py:1233-1256) = torchvision Resize(int(384 / crop_pct), PIL BICUBIC) -> CenterCrop(384)
 * -> ToTensor -> Normalize(.5, .5), on decoded RGB uint8 HWC images that already sit in device memory.
 * The resize is Pillow's 8-bit two-pass fixed-point resampling (libImaging/Resample.c, third-party, not under
 * /root/reference) reproduced bit for bit; torchvision's size / crop arithmetic
 * (transforms/functional.py resize, center_crop) is restated in vitcap_resized_geometry.
 *   out: [B][3][crop][crop] fp32 or bf16 -- what vitcap_engine_greedy takes; out_u8 (optional) the cropped bytes.
 *   The descriptor array is host memory; `rgb` members are device pointers (pitch in bytes, >= 3*width).
 * ---------------------------------------------------------------------------------------------- */
typedef struct vitcap_image {
  const uint8_t* rgb;
  int height, width, pitch;
} vitcap_image;
size_t vitcap_image_preproc_workspace_bytes(const vitcap_image* imgs, int B, int resize_short, int crop);
int vitcap_image_preproc(const vitcap_image* imgs, int B, int resize_short, int crop, int out_bf16, void* out,
                         uint8_t* out_u8, void* workspace, size_t workspace_bytes, void* stream);
/* host-only helpers (no GPU): Pillow's bicubic weights for one axis -- bounds[out_size][2] = (first tap, tap count),
 * kk[out_size][ksize] 22-bit fixed point -- and torchvision's resized size / centre-crop origin */
int vitcap_resample_coeffs(int in_size, int out_size, int* ksize_out, int* bounds, int* kk, int kk_capacity);
int vitcap_resized_geometry(int height, int width, int resize_short, int crop, int* out_h, int* out_w, int* crop_y0,
                            int* crop_x0);

/* Train-time image transform, get_inception_train_transform (src/data_layer/transform.py:52-81; selected by
 * get_transform_vit_default, uni_pipeline.py:1258-1264): RandomResizedCrop(size, PIL BILINEAR) -> ColorJitter(brightness,
 * contrast, saturation) -> RandomHorizontalFlip -> ToTensor -> Normalize(.5, .5).
 * The random parameters are drawn on the host (vitcap_amd/augment.py restates torchvision 0.7's get_params) and passed in;
 * the image arithmetic is Pillow's (Resample.c triangle filter on the cropped image, Blend.c float32 blend with the
 * black / mean-gray / grayscale degenerate image, Convert.c rgb2l, left-right transpose) and is reproduced bit for bit.
 *   op[3]: colour operations in application order, 0 = brightness, 1 = contrast, 2 = saturation, -1 = none;
 *   factor[3]: the enhancement factors (>= 0);  out: [B][3][size][size] fp32 or bf16;  out_u8 optional bytes. */
typedef struct vitcap_train_aug {
  int32_t top, left, height, width; /* crop box in source pixels (RandomResizedCrop.get_params' i, j, h, w) */
  int32_t op[3];
  float factor[3];
  int32_t flip;
} vitcap_train_aug;
size_t vitcap_image_train_preproc_workspace_bytes(const vitcap_image* imgs, const vitcap_train_aug* aug, int B, int size);
int vitcap_image_train_preproc(const vitcap_image* imgs, const vitcap_train_aug* aug, int B, int size, int out_bf16,
                               void* out, uint8_t* out_u8, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Device back half of the JPEG decoder ("rocJPEG + HIP resize" -- built as own kernels because the parity bar is
 * bit-exact pixels): replaces the part of cv2.imdecode (src/tools/common.py:23-31 img_from_base64, src/data_layer/transform.py:106-136)
 * that follows the entropy decoder.  The host front half (include/vitcap_jpeg.h, libvitcap_jpeg.so: vitcap_jpeg_parse /
 * vitcap_jpeg_decode_coefs, run by the loader's worker processes) yields a vitcap_jpeg_info and int16 coefficient blocks; this call
 * dequantises them, runs libjpeg's ISLOW 8x8 inverse DCT, the "fancy" h2v1 / h2v2 chroma upsampling and the fixed-point YCbCr -> RGB
 * conversion, and writes the uint8 HWC RGB image that vitcap_image_preproc takes.  Bit-identical to Pillow's decoder (libjpeg-turbo,
 * default settings) on every stream vitcap_jpeg_parse accepts; other streams never reach this call (the caller decodes them with Pillow).
 *   imgs: host array; `coefs` (info.nblocks * 64 int16, as vitcap_jpeg_decode_coefs wrote them) and `rgb` (height rows of `pitch`
 *   >= 3 * width bytes) are DEVICE pointers.  Enqueue-only on `stream`; the workspace holds the component planes.
 * ---------------------------------------------------------------------------------------------- */
typedef struct vitcap_jpeg_image {
  vitcap_jpeg_info info;
  const int16_t* coefs;
  uint8_t* rgb;
  int32_t pitch;
} vitcap_jpeg_image;
size_t vitcap_jpeg_backhalf_workspace_bytes(const vitcap_jpeg_image* imgs, int B);
int vitcap_jpeg_backhalf(const vitcap_jpeg_image* imgs, int B, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VITCAP_HIP_H */
