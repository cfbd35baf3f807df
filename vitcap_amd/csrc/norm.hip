// LayerNorm / embedding+LayerNorm / small data movers (HBM-bound kernels, one wave per 768-wide row).
//
// D = 768 = 64 lanes x 12 elements: every lane owns three float4 (lane*4 + {0,256,512}) so each wave
// instruction reads a contiguous 1 KiB.  Statistics are two-pass in registers (mean, then centred sum
// of squares) in fp32, like ATen's CPU LayerNorm, then y = (x-mean)*rstd*gamma+beta.
#include "common.h"

namespace {


__global__ __launch_bounds__(256) void layernorm768_kernel(const float* __restrict__ x, int ldx,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps,
                                                           bf16_t* __restrict__ yb, float* __restrict__ yf, int M, int rev) {
  const int lane = threadIdx.x & 63;
  // rev: workgroups are dispatched in blockIdx order -- the rows are then visited last-to-first (common.h: vc_tls_walk_rev)
  const int row = (rev ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (size_t)row * ldx;
  f32x4 v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) v[i] = *(const f32x4*)(xr + i * 256 + lane * 4);
  ln_row(v, gamma, beta, eps, lane, yb ? yb + (size_t)row * D768 : nullptr, yf ? yf + (size_t)row * D768 : nullptr);
}

// split-K consumer: v = sum_s P[s][row] + bias (+ residual) (-> gelu) -> LayerNorm.
// One 64-thread workgroup per row (M is 64..256 here: many small workgroups beat 4 rows per workgroup), slab loads
// unrolled four deep so 12 float4 loads are in flight per lane.
__global__ __launch_bounds__(64) void sum_layernorm768_kernel(const float* __restrict__ part, int S, size_t slab,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ res, int ldr, int act,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps,
                                                              bf16_t* __restrict__ yb, float* __restrict__ yf, int M,
                                                              const int32_t* __restrict__ live) {
  VC_LIVE_EXIT(live);
  const int lane = threadIdx.x;
  const int row = blockIdx.x;
  const float* p0 = part + (size_t)row * D768 + lane * 4;
  f32x4 v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) v[i] = *(const f32x4*)(bias + i * 256 + lane * 4);
  int s = 0;
  for (; s + 4 <= S; s += 4) {
    f32x4 t[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 3; ++i) t[u][i] = *(const f32x4*)(p0 + (size_t)(s + u) * slab + i * 256);
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] += (t[0][i] + t[1][i]) + (t[2][i] + t[3][i]);
  }
  for (; s < S; ++s)
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] += *(const f32x4*)(p0 + (size_t)s * slab + i * 256);
  if (act) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
      v[i] = gelu_erf4(v[i]);
  }
  if (res) {
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] += *(const f32x4*)(res + (size_t)row * ldr + i * 256 + lane * 4);
  }
  ln_row(v, gamma, beta, eps, lane, yb ? yb + (size_t)row * D768 : nullptr, yf ? yf + (size_t)row * D768 : nullptr);
}

__device__ __forceinline__ f32x4 ld_bf4(const bf16_t* p) {
  const uint2 u = *(const uint2*)p;
  return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
               __uint_as_float(u.y & 0xffff0000u)};
}

// rows (b,0): word[ids[b][t-1]] + pos[t-1] + type[0];  rows (b,1): word[mask] + pos[t] + type[0]; then LN.
__global__ __launch_bounds__(256) void embed_step_kernel(const int64_t* __restrict__ ids, int max_len, int t,
                                                         int mask_token, const bf16_t* __restrict__ word,
                                                         const bf16_t* __restrict__ pos,
                                                         const bf16_t* __restrict__ type,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps,
                                                         float* __restrict__ xf, bf16_t* __restrict__ xb, int rows,
                                                         const int32_t* __restrict__ live) {
  VC_LIVE_EXIT(live);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int b = row >> 1, which = row & 1;
  const int64_t tok = which ? (int64_t)mask_token : ids[(size_t)b * max_len + (t - 1)];
  const int p = which ? t : t - 1;
  f32x4 v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = i * 256 + lane * 4;
    const f32x4 a = ld_bf4(word + (size_t)tok * D768 + c);
    const f32x4 q = ld_bf4(pos + (size_t)p * D768 + c);
    const f32x4 ty = ld_bf4(type + c);
    v[i] = (a + q) + ty;   // same association as `words + position + token_type` (modeling_bert.py:234)
  }
  ln_row(v, gamma, beta, eps, lane, xb + (size_t)row * D768, xf + (size_t)row * D768);
}

// Greedy token choice + bookkeeping for step t from the vocabulary GEMM's row statistics, fused with the text embedding of
// step t+1.  One 128-thread workgroup per sequence.
//   rowstat [B][pieces][4] = {max, argmax column (int bits), sum exp(x - max), -} per 32-column piece of the logits row:
//   tok = column of the overall maximum, lowest column on ties (torch.argmax);  lp = log_softmax(row)[tok] =
//   -log(sum_pieces s_p * exp(m_p - M));  then exactly greedy_step_kernel's bookkeeping (modeling_utils.py:850-877);
//   rows (b,0) = word[ids[b][t]] + pos[t] + type[0], (b,1) = word[MASK] + pos[t+1] + type[0], LayerNorm -> x of step t+1.
__global__ __launch_bounds__(128) void greedy_select_embed_kernel(const float* __restrict__ rowstat, int pieces,
                                                                  int64_t* __restrict__ ids, int32_t* __restrict__ unf,
                                                                  float* __restrict__ sum_lp, float* __restrict__ cnt,
                                                                  float* __restrict__ logprob_out, int64_t* __restrict__ raw_last,
                                                                  int t, int max_len, int eos, int pad, int32_t* __restrict__ live,
                                                                  int mask_token, const bf16_t* __restrict__ word,
                                                                  const bf16_t* __restrict__ pos, const bf16_t* __restrict__ type,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  float eps, float* __restrict__ xf, bf16_t* __restrict__ xb,
                                                                  VcEosExtra eos_x) {
  const bool last = t == max_len - 1;
  if (live != nullptr && *live == 0 && !last) return;       // every sequence finished: ids stay PAD, x is not needed any more
  __shared__ float s_m[2], s_s[2];
  __shared__ int s_i[2];
  __shared__ long long s_tok;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int u = unf[b];
  float M = 0.f, lp = 0.f;
  int I = pad;
  if (u) {
    const f32x4* rs = (const f32x4*)rowstat + (size_t)b * pieces;
    float bm = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < pieces; i += 128) {
      const f32x4 r = rs[i];
      const int idx = __float_as_int(r[1]);
      if (r[0] > bm || (r[0] == bm && idx < bi)) { bm = r[0]; bi = idx; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(bm, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (om > bm || (om == bm && oi < bi)) { bm = om; bi = oi; }
    }
    if (lane == 0) { s_m[w] = bm; s_i[w] = bi; }
    __syncthreads();
    M = s_m[0];
    I = s_i[0];
    if (s_m[1] > M || (s_m[1] == M && s_i[1] < I)) { M = s_m[1]; I = s_i[1]; }
    float se = 0.f;
    for (int i = tid; i < pieces; i += 128) {
      const f32x4 r = rs[i];
      se += r[2] * expf(r[0] - M);
    }
    se = wave_sum(se);
    if (lane == 0) s_s[w] = se;
    __syncthreads();
    lp = -logf(s_s[0] + s_s[1]);                              // logit[tok] - M - log(sum exp(x - M)), tok is the maximum
  }
  if (tid == 0) {
    const int add = u ? I : pad;
    const float s = sum_lp[b] + lp * (float)u;
    const float c = cnt[b] + (float)u;
    const int nu = u * (vc_is_eos(add, eos, eos_x) ? 0 : 1);
    long long outtok = add;
    if (last) {
      if (raw_last) raw_last[b] = add;               // the token actually chosen, before the forced [SEP]
      if (nu) outtok = eos;                          // modeling_utils.py:870-871
      logprob_out[b] = s / c;                        // modeling_utils.py:873-877
    }
    ids[(size_t)b * max_len + t] = outtok;
    sum_lp[b] = s;
    cnt[b] = c;
    unf[b] = nu;
    if (live && u && !nu) atomicSub(live, 1);
    s_tok = outtok;
  }
  if (last) return;
  __syncthreads();
  // embedding of step t+1 (BertEmbeddings.forward, modeling_bert.py:222-237): wave 0 -> row (b,0), wave 1 -> row (b,1)
  const long long tok = w ? (long long)mask_token : s_tok;
  const int p = w ? t + 1 : t;
  const int row = b * 2 + w;
  f32x4 v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = i * 256 + lane * 4;
    const f32x4 a = ld_bf4(word + (size_t)tok * D768 + c);
    const f32x4 q = ld_bf4(pos + (size_t)p * D768 + c);
    const f32x4 ty = ld_bf4(type + c);
    v[i] = (a + q) + ty;   // same association as `words + position + token_type` (modeling_bert.py:234)
  }
  ln_row(v, gamma, beta, eps, lane, xb + (size_t)row * D768, xf + (size_t)row * D768);
}

// Embeddings of the predicted tag tokens that ViTSplitCLSEmbModel.forward writes over the last 50 text slots
// (modeling_bert.py:1435-1489); one wave per (image, slot j < n).  tok = pred_topk[b][j], slot 49 forced to 102 (:1452, :1477).
//   branch A (topk_len[0] + 20 <= L), tagemb == 'cls': the raw row of the caption head's decoder matrix (F.embedding, :1456-1462)
//   branch A, otherwise        : LN_emb(word[tok] + pos[20 + j] + type[0])                       (encode_tag_to_embedding, :1381-1406)
//   branch B, tagemb == 'cls'  : LN_emb(cls_w[tok] + pos[20 + j] + type[0])                      (:1481)
//   branch B, otherwise        : LN_x(xword[tok] + xpos[pos0 + j] + xtype[0])   (bert.extra_embeddings, :1484-1485)
__global__ __launch_bounds__(256) void tag_embed_kernel(const int64_t* __restrict__ tag_ids, int n, int pos0, int branch_a, int tagemb_cls,
                                                        const bf16_t* __restrict__ cls_w, const bf16_t* __restrict__ word,
                                                        const bf16_t* __restrict__ pos, const bf16_t* __restrict__ type,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const bf16_t* __restrict__ xword, const bf16_t* __restrict__ xpos,
                                                        const bf16_t* __restrict__ xtype, const float* __restrict__ xgamma,
                                                        const float* __restrict__ xbeta, float eps, float* __restrict__ xf,
                                                        bf16_t* __restrict__ xb, int rows) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int b = row / n, j = row - b * n;
  const int64_t tok = j == 49 ? (int64_t)102 : tag_ids[(size_t)b * 50 + j];
  const bool extra = !branch_a && !tagemb_cls;
  const bf16_t* tab = tagemb_cls ? cls_w : (extra ? xword : word);
  const bf16_t* ptab = extra ? xpos : pos;
  const bf16_t* ttab = extra ? xtype : type;
  const int pbase = extra ? pos0 : 20;     // encode_tag_to_embedding's literal caption_len = 20 (:1381, :1396); the caller's position_ids reach bert.extra_embeddings only (:1484-1485)
  f32x4 v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = i * 256 + lane * 4;
    const f32x4 a = ld_bf4(tab + (size_t)tok * D768 + c);
    if (branch_a && tagemb_cls) {
      v[i] = a;
    } else {
      const f32x4 q = ld_bf4(ptab + (size_t)(pbase + j) * D768 + c);
      const f32x4 ty = ld_bf4(ttab + c);
      v[i] = (a + q) + ty;
    }
  }
  if (branch_a && tagemb_cls) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int c = i * 256 + lane * 4;
      *(f32x4*)(xf + (size_t)row * D768 + c) = v[i];
      uint2 o;
      o.x = pack2bf(v[i][0], v[i][1]);
      o.y = pack2bf(v[i][2], v[i][3]);
      *(uint2*)(xb + (size_t)row * D768 + c) = o;
    }
    return;
  }
  ln_row(v, extra ? xgamma : gamma, extra ? xbeta : beta, eps, lane, xb + (size_t)row * D768, xf + (size_t)row * D768);
}

// dst[(b * dst_img_rows + dst_row0 + r) * ld_dst + dst_col0 + c] = src[(b * src_img_rows + src_row0 + r) * ld_src + src_col0 + c]
// for r < rows, c < cols (bf16 elements, cols and all offsets multiples of 8): moves row blocks between per-image layouts
__global__ __launch_bounds__(256) void copy_row_blocks_kernel(const bf16_t* __restrict__ src, int src_img_rows, int src_row0, int ld_src,
                                                              int src_col0, bf16_t* __restrict__ dst, int dst_img_rows, int dst_row0,
                                                              int ld_dst, int dst_col0, int rows, int cols8) {
  const int b = blockIdx.y;
  const int per = rows * cols8;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < per; i += gridDim.x * 256) {
    const int r = i / cols8, c = (i - r * cols8) * 8;
    const uint4 v = *(const uint4*)(src + ((size_t)b * src_img_rows + src_row0 + r) * ld_src + src_col0 + c);
    *(uint4*)(dst + ((size_t)b * dst_img_rows + dst_row0 + r) * ld_dst + dst_col0 + c) = v;
  }
}

// teacher-forced text rows: row r = word[ids[r]] + pos[r % rows_per_seq] + type[0] -> (pre-LN sum fp32, LN fp32, LN bf16)
__global__ __launch_bounds__(256) void embed_rows_kernel(const int64_t* __restrict__ ids, int rows_per_seq,
                                                         const bf16_t* __restrict__ word, const bf16_t* __restrict__ pos,
                                                         const bf16_t* __restrict__ type, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, float* __restrict__ pre,
                                                         float* __restrict__ xf, bf16_t* __restrict__ xb, int rows, int pos_wrap) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t tok = ids[row];
  int p = row % rows_per_seq;
  if (pos_wrap > 0 && p >= pos_wrap) p = p - pos_wrap + 1;     // probe rows: [MASK] at positions 1, 2, ...
  f32x4 v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = i * 256 + lane * 4;
    v[i] = (ld_bf4(word + (size_t)tok * D768 + c) + ld_bf4(pos + (size_t)p * D768 + c)) + ld_bf4(type + c);
    if (pre) *(f32x4*)(pre + (size_t)row * D768 + c) = v[i];
  }
  ln_row(v, gamma, beta, eps, lane, xb ? xb + (size_t)row * D768 : nullptr, xf ? xf + (size_t)row * D768 : nullptr);
}

// image [B,3,384,384] -> patches [B*576, 768], k = c*256 + kh*16 + kw.  One thread = 8 kw (16 B out).
template <bool IN_BF16>
__global__ __launch_bounds__(256) void patch_gather_kernel(const void* __restrict__ image,
                                                           bf16_t* __restrict__ out, int total_chunks) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= total_chunks) return;
  // chunk id -> (row = b*576 + ph*24 + pw, c, kh, half)
  const int row = gid / 96, ck = gid - row * 96;      // 96 chunks of 8 per 768-wide row
  const int c = ck >> 5, kh = (ck >> 1) & 15, half = ck & 1;
  const int b = row / 576, pr = row - b * 576;
  const int ph = pr / 24, pw = pr - ph * 24;
  const size_t src = (((size_t)b * 3 + c) * 384 + (ph * 16 + kh)) * 384 + pw * 16 + half * 8;
  uint4 o;
  if (IN_BF16) {
    o = *(const uint4*)((const bf16_t*)image + src);
  } else {
    const f32x4 lo = *(const f32x4*)((const float*)image + src);
    const f32x4 hi = *(const f32x4*)((const float*)image + src + 4);
    o.x = pack2bf(lo[0], lo[1]);
    o.y = pack2bf(lo[2], lo[3]);
    o.z = pack2bf(hi[0], hi[1]);
    o.w = pack2bf(hi[2], hi[3]);
  }
  *(uint4*)(out + (size_t)row * 768 + ck * 8) = o;
}

__global__ void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ x,
                                int rows_per_image) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < D768; c += blockDim.x) x[(size_t)b * rows_per_image * D768 + c] = cls[c] + pos[c];
}

// vis[b] = [tag_hidden[b,0], hidden[b,0..n_tok-1]]  (modeling_bert.py:1493), fp32 + bf16 copies
__global__ __launch_bounds__(256) void assemble_visual_kernel(const float* __restrict__ hidden,
                                                              const float* __restrict__ tag_hidden,
                                                              float* __restrict__ vf, bf16_t* __restrict__ vb,
                                                              int n_tok, size_t total4) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total4) return;
  const size_t e = gid * 4;
  const size_t row = e / D768;
  const int c = (int)(e - row * D768);
  const int S = n_tok + 1;
  const size_t b = row / S;
  const int r = (int)(row - b * S);
  const float* src = r == 0 ? tag_hidden + (b * n_tok) * D768 + c : hidden + (b * n_tok + (r - 1)) * D768 + c;
  const f32x4 v = *(const f32x4*)src;
  *(f32x4*)(vf + e) = v;
  uint2 o;
  o.x = pack2bf(v[0], v[1]);
  o.y = pack2bf(v[2], v[3]);
  *(uint2*)(vb + e) = o;
}

__global__ void gather_rows_bf16_kernel(const float* __restrict__ x, int ldx_rows, bf16_t* __restrict__ out, int D) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < D; c += blockDim.x) out[(size_t)b * D + c] = f2bf(x[(size_t)b * ldx_rows * D + c]);
}

}  // namespace

extern "C" int vitcap_layernorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, float eps,
                                    void* y_bf16, float* y_f32, int M, int D, void* stream) {
  VC_REQUIRE(x && gamma && beta && (y_bf16 || y_f32), "layernorm: null pointer");
  VC_REQUIRE(D == D768, "layernorm: only D=768 is built (got %d)", D);
  VC_REQUIRE(M > 0 && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0, "layernorm: bad M/ldx/alignment");
  hipLaunchKernelGGL(layernorm768_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta,
                     eps, (bf16_t*)y_bf16, y_f32, M, vc_tls_walk_rev ? 1 : 0);
  VC_LAUNCH_CHECK("layernorm");
  return VITCAP_OK;
}

extern "C" int vitcap_sum_layernorm(const float* partials, int S, size_t slab_stride, const float* bias,
                                    const float* residual, int ldr, int act_before_ln, const float* gamma,
                                    const float* beta, float eps, void* y_bf16, float* y_f32, int M, int D,
                                    void* stream) {
  VC_REQUIRE(partials && bias && gamma && beta && (y_bf16 || y_f32), "sum_layernorm: null pointer");
  VC_REQUIRE(D == D768 && S >= 1 && M > 0, "sum_layernorm: only D=768, S>=1 (got D=%d S=%d)", D, S);
  VC_REQUIRE(!(act_before_ln && residual), "sum_layernorm: activation and residual are mutually exclusive");
  hipLaunchKernelGGL(sum_layernorm768_kernel, dim3(M), dim3(64), 0, (hipStream_t)stream, partials, S,
                     slab_stride, bias, residual, ldr, act_before_ln, gamma, beta, eps, (bf16_t*)y_bf16, y_f32, M, vc_tls_live);
  VC_LAUNCH_CHECK("sum_layernorm");
  return VITCAP_OK;
}

extern "C" int vitcap_embed_step(const int64_t* ids, int max_len, int t, int mask_token, const void* word_emb,
                                 const void* pos_emb, const void* type_emb, const float* gamma, const float* beta,
                                 float eps, float* x_f32, void* x_bf16, int B, void* stream) {
  VC_REQUIRE(ids && word_emb && pos_emb && type_emb && gamma && beta && x_f32 && x_bf16, "embed_step: null pointer");
  VC_REQUIRE(t >= 1 && t < max_len && B > 0, "embed_step: bad t=%d (max_len %d) or B=%d", t, max_len, B);
  const int rows = 2 * B;
  hipLaunchKernelGGL(embed_step_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, ids, max_len, t,
                     mask_token, (const bf16_t*)word_emb, (const bf16_t*)pos_emb, (const bf16_t*)type_emb, gamma,
                     beta, eps, x_f32, (bf16_t*)x_bf16, rows, vc_tls_live);
  VC_LAUNCH_CHECK("embed_step");
  return VITCAP_OK;
}

extern "C" int vitcap_greedy_select_embed(const float* rowstat, int pieces, int64_t* ids, int32_t* unfinished, float* sum_lp,
                                          float* cnt, float* logprob_out, int64_t* raw_last, int B, int t, int max_len, int eos,
                                          int pad, int mask_token, const void* word_emb, const void* pos_emb,
                                          const void* type_emb, const float* gamma, const float* beta, float eps, float* x_f32,
                                          void* x_bf16, void* stream) {
  VC_REQUIRE(rowstat && ids && unfinished && sum_lp && cnt && logprob_out, "greedy_select_embed: null pointer");
  VC_REQUIRE(B > 0 && pieces > 0 && t >= 1 && t < max_len, "greedy_select_embed: bad sizes (t=%d, pieces=%d)", t, pieces);
  VC_REQUIRE(t == max_len - 1 || (word_emb && pos_emb && type_emb && gamma && beta && x_f32 && x_bf16),
             "greedy_select_embed: the embedding of step t+1 needs the tables and outputs");
  hipLaunchKernelGGL(greedy_select_embed_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, rowstat, pieces, ids, unfinished,
                     sum_lp, cnt, logprob_out, raw_last, t, max_len, eos, pad, (int32_t*)vc_tls_live, mask_token,
                     (const bf16_t*)word_emb, (const bf16_t*)pos_emb, (const bf16_t*)type_emb, gamma, beta, eps, x_f32,
                     (bf16_t*)x_bf16, vc_tls_eos_extra);
  VC_LAUNCH_CHECK("greedy_select_embed");
  return VITCAP_OK;
}

extern "C" int vitcap_tag_embed(const int64_t* tag_ids, int n, int pos0, int branch_a, int tagemb_cls, const void* cls_w, const void* word_emb,
                                const void* pos_emb, const void* type_emb, const float* gamma, const float* beta,
                                const void* xword_emb, const void* xpos_emb, const void* xtype_emb, const float* xgamma,
                                const float* xbeta, float eps, float* x_f32, void* x_bf16, int B, void* stream) {
  VC_REQUIRE(tag_ids && x_f32 && x_bf16 && B > 0 && n >= 1 && n <= 50, "tag_embed: bad arguments (n=%d)", n);
  VC_REQUIRE(pos0 >= 0 && pos0 + n <= 512, "tag_embed: positions %d..%d are outside the 512-row position table", pos0, pos0 + n - 1);
  const bool need_extra = !tagemb_cls && !branch_a, raw = tagemb_cls && branch_a;
  VC_REQUIRE(!tagemb_cls || cls_w, "tag_embed: tagemb == 'cls' needs the caption head's decoder matrix");
  VC_REQUIRE(tagemb_cls || need_extra || word_emb, "tag_embed: missing word embedding table");
  VC_REQUIRE(raw || need_extra || (pos_emb && type_emb && gamma && beta), "tag_embed: missing position / type tables or LayerNorm");
  VC_REQUIRE(!need_extra || (xword_emb && xpos_emb && xtype_emb && xgamma && xbeta), "tag_embed: bert.extra_embeddings tables missing");
  const int rows = B * n;
  hipLaunchKernelGGL(tag_embed_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, tag_ids, n, pos0, branch_a, tagemb_cls,
                     (const bf16_t*)cls_w, (const bf16_t*)word_emb, (const bf16_t*)pos_emb, (const bf16_t*)type_emb, gamma, beta,
                     (const bf16_t*)xword_emb, (const bf16_t*)xpos_emb, (const bf16_t*)xtype_emb, xgamma, xbeta, eps, x_f32,
                     (bf16_t*)x_bf16, rows);
  VC_LAUNCH_CHECK("tag_embed");
  return VITCAP_OK;
}

extern "C" int vitcap_copy_row_blocks(const void* src, int src_img_rows, int src_row0, int ld_src, int src_col0, void* dst,
                                      int dst_img_rows, int dst_row0, int ld_dst, int dst_col0, int rows, int cols, int B,
                                      void* stream) {
  VC_REQUIRE(src && dst && B > 0 && rows > 0 && cols > 0, "copy_row_blocks: bad arguments");
  VC_REQUIRE(((cols | ld_src | ld_dst | src_col0 | dst_col0) & 7) == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0,
             "copy_row_blocks: columns / leading dimensions must be multiples of 8 bf16 elements");
  const int per = rows * (cols / 8);
  int gx = (per + 255) / 256;
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL(copy_row_blocks_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, src_img_rows,
                     src_row0, ld_src, src_col0, (bf16_t*)dst, dst_img_rows, dst_row0, ld_dst, dst_col0, rows, cols / 8);
  VC_LAUNCH_CHECK("copy_row_blocks");
  return VITCAP_OK;
}

extern "C" int vitcap_embed_rows(const int64_t* ids, int rows_per_seq, const void* word_emb, const void* pos_emb,
                                 const void* type_emb, const float* gamma, const float* beta, float eps, float* pre_f32,
                                 float* x_f32, void* x_bf16, int rows, int pos_wrap, void* stream) {
  VC_REQUIRE(ids && word_emb && pos_emb && type_emb && gamma && beta && rows > 0 && rows_per_seq > 0, "embed_rows: bad arguments");
  hipLaunchKernelGGL(embed_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, ids, rows_per_seq,
                     (const bf16_t*)word_emb, (const bf16_t*)pos_emb, (const bf16_t*)type_emb, gamma, beta, eps, pre_f32,
                     x_f32, (bf16_t*)x_bf16, rows, pos_wrap);
  VC_LAUNCH_CHECK("embed_rows");
  return VITCAP_OK;
}

extern "C" int vitcap_patch_gather(const void* image, int image_is_bf16, void* patches_bf16, int B, void* stream) {
  VC_REQUIRE(image && patches_bf16 && B > 0, "patch_gather: bad arguments");
  const int total = B * 576 * 96;
  if (image_is_bf16)
    hipLaunchKernelGGL(patch_gather_kernel<true>, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, image,
                       (bf16_t*)patches_bf16, total);
  else
    hipLaunchKernelGGL(patch_gather_kernel<false>, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, image,
                       (bf16_t*)patches_bf16, total);
  VC_LAUNCH_CHECK("patch_gather");
  return VITCAP_OK;
}

extern "C" int vitcap_cls_rows(const float* cls_token, const float* pos_embed, float* x, int B, int rows_per_image,
                               void* stream) {
  VC_REQUIRE(cls_token && pos_embed && x && B > 0, "cls_rows: bad arguments");
  hipLaunchKernelGGL(cls_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, cls_token, pos_embed, x,
                     rows_per_image);
  VC_LAUNCH_CHECK("cls_rows");
  return VITCAP_OK;
}

extern "C" int vitcap_assemble_visual(const float* hidden, const float* tag_hidden, float* vis_f32, void* vis_bf16,
                                      int B, int n_tok, void* stream) {
  VC_REQUIRE(hidden && tag_hidden && vis_f32 && vis_bf16 && B > 0 && n_tok > 0, "assemble_visual: bad arguments");
  const size_t total4 = (size_t)B * (n_tok + 1) * D768 / 4;
  hipLaunchKernelGGL(assemble_visual_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     hidden, tag_hidden, vis_f32, (bf16_t*)vis_bf16, n_tok, total4);
  VC_LAUNCH_CHECK("assemble_visual");
  return VITCAP_OK;
}

extern "C" int vitcap_gather_rows_bf16(const float* x, int ldx_rows, void* out_bf16, int B, int D, void* stream) {
  VC_REQUIRE(x && out_bf16 && B > 0 && D > 0, "gather_rows: bad arguments");
  hipLaunchKernelGGL(gather_rows_bf16_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, ldx_rows,
                     (bf16_t*)out_bf16, D);
  VC_LAUNCH_CHECK("gather_rows");
  return VITCAP_OK;
}
