// Training-step kernels other than GEMM/attention (gfx950): transposes for the weight-gradient GEMMs, LayerNorm
// backward, split-K slab reduction, embedding backward, losses, global-norm clip + fused AdamW, weight re-packing.
// All are HBM-bound streaming kernels (vectorised 8/16-byte accesses, one wave per 768-wide row where rows matter).
#include <stdlib.h>

#include "common.h"
#include "rng.h"

namespace {


// ---------------------------------------------------------------------------------------------------------------
// XT[c][r] = X[r][c] (bf16), rows padded with zeros to Rp (multiple of 64); optional column sums (bias gradients).
// 64x64 tile per 256-thread workgroup through LDS.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_colsum_kernel(const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ xt,
                                                               int ldt, float* __restrict__ colsum, int R, int C) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + i * 256;            // 512 pieces of 8 elements
    const int r = e >> 3, cc = (e & 7) * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r0 + r < R) v = *(const uint4*)(x + (size_t)(r0 + r) * ldx + c0 + cc);
    const bf16_t* pv = (const bf16_t*)&v;
#pragma unroll
    for (int j = 0; j < 8; ++j) tile[r][cc + j] = pv[j];
  }
  __syncthreads();
  if (colsum && tid < 64) {
    float s = 0.f;
    for (int r = 0; r < 64; ++r) s += bf2f(tile[r][tid]);
    atomicAdd(colsum + c0 + tid, s);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + i * 256;
    const int c = e >> 3, rr = (e & 7) * 8;
    bf16_t o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = tile[rr + j][c];
    *(uint4*)(xt + (size_t)(c0 + c) * ldt + r0 + rr) = *(const uint4*)o;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm backward over 768-wide rows.  x: the forward input (fp32, statistics recomputed), dy: bf16 or fp32 gradient
// of the output; dres: optional fp32 gradient arriving through the residual path (added to the result).
//   dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)) + dres ;  dgamma += sum dy*xhat ; dbeta += sum dy
// ---------------------------------------------------------------------------------------------------------------
//   CS: additionally dxb_colsum[c] += sum over rows of the bf16-ROUNDED dx (what vitcap_colsum_bf16 of dx_bf16 would add: the bias
//   gradient of the linear layer whose output gradient dx is -- one pass over dx less per layer)
template <bool DY_F32, int NW, bool CS>
__global__ __launch_bounds__(NW * 64) void layernorm_bwd_kernel(const float* __restrict__ x, int ldx, const void* __restrict__ dyv,
                                                            const float* __restrict__ gamma, float eps,
                                                            const float* __restrict__ dres, float* __restrict__ dxf,
                                                            bf16_t* __restrict__ dxb, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ dxb_colsum, int M,
                                                            int rows_per_wave) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * NW + (threadIdx.x >> 6);
  f32x4 g[3], ag[3], ab[3], ac[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    g[i] = *(const f32x4*)(gamma + i * 256 + lane * 4);
    ag[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    ab[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    ac[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // rows are dealt round-robin over ALL waves of the grid (rows_per_wave = the stride = total waves): the grid is a whole
  // number of workgroups per CU, so nobody waits for a ragged last round (289 workgroups on 256 CUs cost two rounds)
  for (int row = wave; row < M; row += rows_per_wave) {   // (no early return: every wave reaches the reduction below)
    f32x4 v[3], dy[3];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int c = i * 256 + lane * 4;
      v[i] = *(const f32x4*)(x + (size_t)row * ldx + c);
      if (DY_F32) {
        dy[i] = *(const f32x4*)((const float*)dyv + (size_t)row * D768 + c);
      } else {
        const uint2 u = *(const uint2*)((const bf16_t*)dyv + (size_t)row * D768 + c);
        dy[i] = f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                      __uint_as_float(u.y & 0xffff0000u)};
      }
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) * (1.0f / D768);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[i][e] - mean;
        q += d * d;
      }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D768) + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (v[i][e] - mean) * rstd;
        const float gd = g[i][e] * dy[i][e];
        s1 += gd;
        s2 += gd * xh;
        ag[i][e] += dy[i][e] * xh;
        ab[i][e] += dy[i][e];
        v[i][e] = xh;
        dy[i][e] = gd;
      }
    s1 = wave_sum(s1) * (1.0f / D768);
    s2 = wave_sum(s2) * (1.0f / D768);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int c = i * 256 + lane * 4;
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rstd * (dy[i][e] - s1 - v[i][e] * s2);
      if (dres) o += *(const f32x4*)(dres + (size_t)row * D768 + c);
      if (dxf) *(f32x4*)(dxf + (size_t)row * D768 + c) = o;
      if (dxb) {
        uint2 u;
        u.x = pack2bf(o[0], o[1]);
        u.y = pack2bf(o[2], o[3]);
        *(uint2*)(dxb + (size_t)row * D768 + c) = u;
        if (CS) ac[i] += f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                               __uint_as_float(u.y & 0xffff0000u)};
      }
    }
  }
  // workgroup-level reduction of the NW waves' partial dgamma/dbeta through LDS, then ONE atomic per column per workgroup
  // (per-wave atomics: 3.5 M atomics on 1536 addresses per call made this kernel 7x slower than its HBM time).
  // Measured at M = 36928 (tools/lnbwd_bench.py): 4 waves x 32 consecutive rows in 289 workgroups 130 us (two rounds on 256
  // CUs); 8 waves, 2 workgroups per CU, rows dealt round-robin 90 us = 5.0 TB/s
  extern __shared__ float red[];            // [2 (+1 with CS)][NW][768]
  const int wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    *(f32x4*)(&red[(0 * NW + wv) * D768 + i * 256 + lane * 4]) = ag[i];
    *(f32x4*)(&red[(1 * NW + wv) * D768 + i * 256 + lane * 4]) = ab[i];
    if (CS) *(f32x4*)(&red[(2 * NW + wv) * D768 + i * 256 + lane * 4]) = ac[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D768; c += NW * 64) {
    float sg = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      sg += red[(0 * NW + q) * D768 + c];
      sb += red[(1 * NW + q) * D768 + c];
      if (CS) sc += red[(2 * NW + q) * D768 + c];
    }
    atomicAdd(dgamma + c, sg);
    atomicAdd(dbeta + c, sb);
    if (CS) atomicAdd(dxb_colsum + c, sc);
  }
}

// y (bf16) = x (fp32) over 768-wide rows, and colsum[c] += sum over rows of the ROUNDED values -- the cast of a residual-stream
// gradient to the bf16 operand of the backward GEMMs, fused with the bias gradient of the layer it is the output gradient of
// (vitcap_cast_bf16 + vitcap_colsum_bf16 in one pass).  Rows dealt round-robin over the grid's waves as in layernorm_bwd_kernel.
template <int NW>
__global__ __launch_bounds__(NW * 64) void cast_bf16_colsum_kernel(const float* __restrict__ x, bf16_t* __restrict__ y,
                                                                  float* __restrict__ colsum, int M, int rows_per_wave) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * NW + (threadIdx.x >> 6);
  f32x4 ac[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) ac[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int row = wave; row < M; row += rows_per_wave) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int c = i * 256 + lane * 4;
      const f32x4 v = *(const f32x4*)(x + (size_t)row * D768 + c);
      uint2 u;
      u.x = pack2bf(v[0], v[1]);
      u.y = pack2bf(v[2], v[3]);
      *(uint2*)(y + (size_t)row * D768 + c) = u;
      ac[i] += f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u)};
    }
  }
  extern __shared__ float red[];            // [NW][768]
  const int wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 3; ++i) *(f32x4*)(&red[wv * D768 + i * 256 + lane * 4]) = ac[i];
  __syncthreads();
  for (int c = threadIdx.x; c < D768; c += NW * 64) {
    float sc = 0.f;
#pragma unroll
    for (int q = 0; q < NW; ++q) sc += red[q * D768 + c];
    atomicAdd(colsum + c, sc);
  }
}

// out[i] (+)= sum_s slab[s][i]   (split-K reduction of the weight-gradient GEMMs)
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, size_t slab_stride, int S,
                                                           float* __restrict__ out, size_t n4, int accumulate) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 a = accumulate ? *(const f32x4*)(out + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < S; ++s) a += *(const f32x4*)(slabs + (size_t)s * slab_stride + i * 4);
  *(f32x4*)(out + i * 4) = a;
}

// fp32 [R][C] (+ optional bf16 copy)
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = *(const f32x4*)(x + i * 4);
  uint2 u;
  u.x = pack2bf(v[0], v[1]);
  u.y = pack2bf(v[2], v[3]);
  *(uint2*)(y + i * 4) = u;
}

// embedding backward: gword[ids[row]] += d[row]; gpos[pos[row]] += d[row]; gtype[0] += d[row]
// Waves [0, rows): the word-table scatter of one row.  Waves [rows, rows + rows_per_seq): slot r of every sequence summed first
// (the position and token-type rows are shared by all sequences: as per-row atomics the token-type row took rows x 768 atomics
// on 768 addresses, 190 us per step), then one atomic per column into the position row and the token-type row.
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ d, const int64_t* __restrict__ ids,
                                                        int rows_per_seq, float* __restrict__ gword, float* __restrict__ gpos,
                                                        float* __restrict__ gtype, int rows, int pos_wrap) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row < rows) {
    const int64_t tok = ids[row];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int c = i * 256 + lane * 4;
      const f32x4 v = *(const f32x4*)(d + (size_t)row * D768 + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(gword + (size_t)tok * D768 + c + e, v[e]);
    }
    return;
  }
  const int r = row - rows;
  if (r >= rows_per_seq) return;
  int pos = r;
  if (pos_wrap > 0 && pos >= pos_wrap) pos = pos - pos_wrap + 1;     // probe rows: [MASK] at positions 1, 2, ...
  const int nseq = rows / rows_per_seq;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = i * 256 + lane * 4;
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < nseq; ++b) a += *(const f32x4*)(d + ((size_t)b * rows_per_seq + r) * D768 + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      atomicAdd(gpos + (size_t)pos * D768 + c + e, a[e]);
      atomicAdd(gtype + c + e, a[e]);
    }
  }
}

// label-smoothed KL (BertCaptioningLoss, modeling_bert.py:661-690): per row loss and d(loss)/d(logits) * (1/n_rows)
__global__ __launch_bounds__(1024) void ls_kl_kernel(const float* __restrict__ logits, int ldl, int V,
                                                     const int64_t* __restrict__ target, float eps, float inv_rows,
                                                     const float* __restrict__ row_weight, float* __restrict__ loss_sum,
                                                     bf16_t* __restrict__ dlogits, int ldd) {
  __shared__ float s_red[16];
  __shared__ float s_val;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* row = logits + (size_t)b * ldl;
  float mx = -INFINITY;
  for (int i = tid; i < V; i += 1024) mx = fmaxf(mx, row[i]);
  mx = wave_max(mx);
  if (lane == 0) s_red[w] = mx;
  __syncthreads();
  if (tid == 0) {
    float m = s_red[0];
    for (int k = 1; k < 16; ++k) m = fmaxf(m, s_red[k]);
    s_val = m;
  }
  __syncthreads();
  mx = s_val;
  float se = 0.f, sl = 0.f;
  for (int i = tid; i < V; i += 1024) {
    se += expf(row[i] - mx);
    sl += row[i];
  }
  se = wave_sum(se);
  sl = wave_sum(sl);
  __syncthreads();
  if (lane == 0) s_red[w] = se;
  __syncthreads();
  float tot = 0.f;
  for (int k = 0; k < 16; ++k) tot += s_red[k];
  __syncthreads();
  if (lane == 0) s_red[w] = sl;
  __syncthreads();
  float sumlogit = 0.f;
  for (int k = 0; k < 16; ++k) sumlogit += s_red[k];
  const float lse = mx + logf(tot);
  const int tgt = (int)target[b];
  if (row_weight) inv_rows = row_weight[b];        // per-row coefficient instead of the mean (policy-gradient weights)
  const float q_on = 1.0f - eps, q_off = eps / (float)(V - 1);
  if (tid == 0) {
    // sum_v q_v (log q_v - logp_v),  logp_v = x_v - lse
    const float ent = q_on * logf(q_on) + (eps > 0.f ? eps * logf(q_off) : 0.f);
    const float cross = q_on * (row[tgt] - lse) + q_off * ((sumlogit - row[tgt]) - (float)(V - 1) * lse);
    atomicAdd(loss_sum, (ent - cross) * inv_rows);
  }
  if (dlogits) {
    bf16_t* drow = dlogits + (size_t)b * ldd;
    for (int i = tid; i < ldd; i += 1024) {
      float gv = 0.f;
      if (i < V) gv = (expf(row[i] - lse) - (i == tgt ? q_on : q_off)) * inv_rows;
      drow[i] = f2bf(gv);
    }
  }
}

// focal loss with logits (loss.py:5-22), alpha 0.5 gamma 1, summed
__global__ __launch_bounds__(256) void focal_sum_kernel(const float* __restrict__ logits, int ldl, int V,
                                                        const float* __restrict__ label, float alpha, float* __restrict__ out,
                                                        int B) {
  __shared__ float s_red[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float acc = 0.f;
  const size_t total = (size_t)B * V;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int b = (int)(i / V), v = (int)(i - (size_t)b * V);
    const float x = logits[(size_t)b * ldl + v];
    const float y = label[i];
    const float sp = 1.0f / (1.0f + expf(-x));
    // logsigmoid(x) = min(x,0) - log1p(exp(-|x|))
    const float ls = fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
    const float lsn = fminf(-x, 0.f) - log1pf(expf(-fabsf(x)));
    float l = 0.f;
    if (y == 1.0f) l += alpha * (1.0f - sp) * ls;
    if (y == 0.0f) l += (1.0f - alpha) * sp * lsn;
    acc -= l;
  }
  acc = wave_sum(acc);
  if (lane == 0) s_red[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
}

// torch.nn.BCEWithLogitsLoss() (mean over B x V): the tag loss of a configuration whose `loss` is not 'focal'
// (modeling_bert.py:713-717); out += scale * sum of max(x, 0) - x y + log1p(exp(-|x|))
__global__ __launch_bounds__(256) void bce_sum_kernel(const float* __restrict__ logits, int ldl, int V, const float* __restrict__ label,
                                                      float scale, float* __restrict__ out, int B) {
  __shared__ float s_red[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float acc = 0.f;
  const size_t total = (size_t)B * V;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int b = (int)(i / V), v = (int)(i - (size_t)b * V);
    const float x = logits[(size_t)b * ldl + v];
    acc += fmaxf(x, 0.f) - x * label[i] + log1pf(expf(-fabsf(x)));
  }
  acc = wave_sum(acc);
  if (lane == 0) s_red[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, scale * ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])));
}

// sum of squares of a flat fp32 buffer -> out[0].  Two fixed-order passes (per-block partials, then one block adds
// them in index order): the result is bit-reproducible, which data-parallel ranks rely on -- they must derive the SAME
// clip coefficient from the same all-reduced gradient or their parameters drift apart.
constexpr int SUMSQ_BLOCKS = 2048;
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, size_t n4, float* __restrict__ part) {
  __shared__ float s_red[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = *(const f32x4*)(g + i * 4);
    acc += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  acc = wave_sum(acc);
  if (lane == 0) s_red[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, float* __restrict__ out) {
  __shared__ float s_red[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < SUMSQ_BLOCKS / 256; ++i) acc += part[i * 256 + threadIdx.x];
  acc = wave_sum(acc);
  if (lane == 0) s_red[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// fused clip + AdamW over a flat parameter buffer laid out in 1024-element chunks; chunk_lr/chunk_wd give each chunk's
// learning rate (0 = not owned by the optimizer) and weight decay.  Update order as solver.AdamW.step
// (optimization.py:187-208): moments, p -= step_size * m / (sqrt(v) + eps), then p -= lr * wd * p.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, const float* __restrict__ chunk_lr,
                                                    const float* __restrict__ chunk_wd, const float* __restrict__ gsumsq,
                                                    float clip, float lr_scale, float bc1, float bc2_sqrt, float b1, float b2,
                                                    float eps, size_t nchunks) {
  const size_t chunk = blockIdx.x;
  if (chunk >= nchunks) return;
  const float lr = chunk_lr[chunk] * lr_scale;
  if (chunk_lr[chunk] == 0.f) return;
  const float wd = chunk_wd[chunk];
  const float norm = sqrtf(gsumsq[0]);
  float coef = clip / (norm + 1e-6f);
  coef = coef < 1.0f ? coef : 1.0f;
  const float step_size = lr * bc2_sqrt / bc1;
  const size_t i = chunk * 1024 + threadIdx.x * 4;
  f32x4 pv = *(const f32x4*)(p + i), gv = *(const f32x4*)(g + i), mv = *(const f32x4*)(m + i), vv = *(const f32x4*)(v + i);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float ge = gv[e] * coef;
    mv[e] = mv[e] * b1 + (1.0f - b1) * ge;
    vv[e] = vv[e] * b2 + (1.0f - b2) * ge * ge;
    float pe = pv[e] - step_size * (mv[e] / (sqrtf(vv[e]) + eps));
    if (wd > 0.f) pe -= lr * wd * pe;
    pv[e] = pe;
  }
  *(f32x4*)(p + i) = pv;
  *(f32x4*)(m + i) = mv;
  *(f32x4*)(v + i) = vv;
}

// fp32 W[N][K] -> bf16 W[N][K] and bf16 WT[K][N] (both K and N multiples of 64)
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ w, bf16_t* __restrict__ wb,
                                                             bf16_t* __restrict__ wt, int N, int K, int ldt) {
  __shared__ bf16_t tile[64][66];
  const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64, tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = tid + i * 256;            // 1024 pieces of 4 elements
    const int r = e >> 4, cc = (e & 15) * 4;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (n0 + r < N) v = *(const f32x4*)(w + (size_t)(n0 + r) * K + k0 + cc);
    uint2 u;
    u.x = pack2bf(v[0], v[1]);
    u.y = pack2bf(v[2], v[3]);
    if (n0 + r < N && wb) *(uint2*)(wb + (size_t)(n0 + r) * K + k0 + cc) = u;
    const bf16_t* pu = (const bf16_t*)&u;
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[r][cc + j] = pu[j];
  }
  __syncthreads();
  if (!wt) return;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + i * 256;
    const int c = e >> 3, rr = (e & 7) * 8;
    bf16_t o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = tile[rr + j][c];
    if (n0 + rr + 7 < N) {
      *(uint4*)(wt + (size_t)(k0 + c) * ldt + n0 + rr) = *(const uint4*)o;
    } else {
      for (int j = 0; j < 8; ++j)
        if (n0 + rr + j < N) wt[(size_t)(k0 + c) * ldt + n0 + rr + j] = o[j];
    }
  }
}

// The same for a whole table of matrices in ONE launch (the ~110 GEMM weights refreshed after every optimizer step: as separate
// launches they are 110 x ~6.7 us of mostly launch floor).  Block b finds its matrix by bisection over the items' first tiles.
__global__ __launch_bounds__(256) void cast_transpose_multi_kernel(const vitcap_ct_item* __restrict__ items, int n_items) {
  __shared__ bf16_t tile[64][66];
  int lo = 0, hi = n_items - 1;
  const int b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].tile0 <= b) lo = mid; else hi = mid - 1;
  }
  const vitcap_ct_item it = items[lo];
  const int t = b - it.tile0, tn = (it.N + 63) / 64;
  const int n0 = (t % tn) * 64, k0 = (t / tn) * 64, tid = threadIdx.x;
  const float* __restrict__ w = it.w;
  bf16_t* __restrict__ wb = (bf16_t*)it.w_bf16;
  bf16_t* __restrict__ wt = (bf16_t*)it.wt_bf16;
  const int N = it.N, K = it.K, ldt = it.ldt;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = tid + i * 256;
    const int r = e >> 4, cc = (e & 15) * 4;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (n0 + r < N) v = *(const f32x4*)(w + (size_t)(n0 + r) * K + k0 + cc);
    uint2 u;
    u.x = pack2bf(v[0], v[1]);
    u.y = pack2bf(v[2], v[3]);
    if (n0 + r < N && wb) *(uint2*)(wb + (size_t)(n0 + r) * K + k0 + cc) = u;
    const bf16_t* pu = (const bf16_t*)&u;
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[r][cc + j] = pu[j];
  }
  __syncthreads();
  if (!wt) return;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + i * 256;
    const int c = e >> 3, rr = (e & 7) * 8;
    bf16_t o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = tile[rr + j][c];
    if (n0 + rr + 7 < N) {
      *(uint4*)(wt + (size_t)(k0 + c) * ldt + n0 + rr) = *(const uint4*)o;
    } else {
      for (int j = 0; j < 8; ++j)
        if (n0 + rr + j < N) wt[(size_t)(k0 + c) * ldt + n0 + rr + j] = o[j];
    }
  }
}

// dz = dg * f, f = the gelu'(z) factor the forward GEMM stored (vitcap_gemm_ex zout); backward of BertPredictionHeadTransform's
// activation -- elementwise, tiny
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ dg, const bf16_t* __restrict__ f,
                                                       bf16_t* __restrict__ dz, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dz[i] = f2bf(dg[i] * bf2f(f[i]));
}

// out[j] = sum_b x[b][j]   (pos_embed / cls_token gradients: the parameter is broadcast over the batch)
__global__ __launch_bounds__(256) void sum_over_batch_kernel(const float* __restrict__ x, size_t stride, int B,
                                                             float* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float a = 0.f;
  for (int b = 0; b < B; ++b) a += x[(size_t)b * stride + i];
  out[i] = a;
}

}  // namespace

extern "C" int vitcap_gelu_bwd(const float* dg, const void* z_bf16, void* dz_bf16, size_t n, void* stream) {
  VC_REQUIRE(dg && z_bf16 && dz_bf16 && n > 0, "gelu_bwd: bad arguments");
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dg,
                     (const bf16_t*)z_bf16, (bf16_t*)dz_bf16, n);
  VC_LAUNCH_CHECK("gelu_bwd");
  return VITCAP_OK;
}

extern "C" int vitcap_sum_over_batch(const float* x, size_t stride, int B, float* out, size_t n, void* stream) {
  VC_REQUIRE(x && out && B > 0 && n > 0, "sum_over_batch: bad arguments");
  hipLaunchKernelGGL(sum_over_batch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, stride, B,
                     out, n);
  VC_LAUNCH_CHECK("sum_over_batch");
  return VITCAP_OK;
}

extern "C" int vitcap_transpose_colsum(const void* x, int ldx, void* xt, int ldt, float* colsum, int R, int C, void* stream) {
  VC_REQUIRE(x && xt && R > 0 && C > 0 && C % 64 == 0 && ldt % 64 == 0 && ldt >= R && ldx % 8 == 0,
             "transpose: needs C %% 64 == 0, ldt %% 64 == 0 >= R (R=%d C=%d ldt=%d)", R, C, ldt);
  dim3 grid(ldt / 64, C / 64);
  hipLaunchKernelGGL(transpose_colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (bf16_t*)xt, ldt,
                     colsum, R, C);
  VC_LAUNCH_CHECK("transpose_colsum");
  return VITCAP_OK;
}

static int ln_bwd_grid(int M, int nw, int* rows_per_wave) {
  static const int wg_per_cu = [] { const char* e = getenv("VITCAP_LNBWD_WG_PER_CU"); const int v = e ? atoi(e) : 2; return v >= 1 && v <= 4 ? v : 2; }();
  static const int n_cu = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
               ? prop.multiProcessorCount : 256;
  }();
  int wgs = n_cu * wg_per_cu;
  if ((long long)wgs * nw > M) wgs = (M + nw - 1) / nw;
  *rows_per_wave = wgs * nw;                      // the row stride = total waves
  return wgs;
}

extern "C" int vitcap_cast_bf16_colsum(const float* x, void* y, float* colsum, int M, int D, void* stream) {
  VC_REQUIRE(x && y && colsum && M > 0 && D == D768, "cast_bf16_colsum: bad arguments (D must be 768)");
  constexpr int NW = 8;
  int rpw = 0;
  const int wgs = ln_bwd_grid(M, NW, &rpw);
  hipLaunchKernelGGL(cast_bf16_colsum_kernel<NW>, dim3(wgs), dim3(NW * 64), NW * D768 * sizeof(float), (hipStream_t)stream, x,
                     (bf16_t*)y, colsum, M, rpw);
  VC_LAUNCH_CHECK("cast_bf16_colsum");
  return VITCAP_OK;
}

extern "C" int vitcap_layernorm_bwd(const float* x, int ldx, const void* dy, int dy_is_f32, const float* gamma, float eps,
                                    const float* dres, float* dx_f32, void* dx_bf16, float* dgamma, float* dbeta,
                                    float* dx_bf16_colsum, int M, int D, void* stream) {
  VC_REQUIRE(x && dy && gamma && dgamma && dbeta && (dx_f32 || dx_bf16) && D == D768 && M > 0, "layernorm_bwd: bad arguments");
  VC_REQUIRE(!dx_bf16_colsum || dx_bf16, "layernorm_bwd: column sums are those of the bf16 output, which was not asked for");
  static const int nw = [] { const char* e = getenv("VITCAP_LNBWD_WAVES"); const int v = e ? atoi(e) : 8; return v == 4 || v == 16 ? v : 8; }();
  int rpw = 0;
  const int wgs = ln_bwd_grid(M, nw, &rpw);
  dim3 grid(wgs);
  const bool cs = dx_bf16_colsum != nullptr;
  const size_t lds = (size_t)(cs ? 3 : 2) * nw * D768 * sizeof(float);
#define LNB(F32_, NW_, CS_)                                                                                              \
  do {                                                                                                                  \
    auto kern = layernorm_bwd_kernel<F32_, NW_, CS_>;                                                                    \
    VC_FUNC_SMEM(kern, (int)lds);                                                                                       \
    hipLaunchKernelGGL(kern, grid, dim3(NW_ * 64), lds, (hipStream_t)stream, x, ldx, dy, gamma, eps, dres, dx_f32,       \
                       (bf16_t*)dx_bf16, dgamma, dbeta, dx_bf16_colsum, M, rpw);                                         \
  } while (0)
#define LNB2(F32_, CS_)                                                                                                  \
  do { if (nw == 4) LNB(F32_, 4, CS_); else if (nw == 8) LNB(F32_, 8, CS_); else LNB(F32_, 16, CS_); } while (0)
  if (dy_is_f32) { if (cs) LNB2(true, true); else LNB2(true, false); }
  else { if (cs) LNB2(false, true); else LNB2(false, false); }
#undef LNB2
#undef LNB
  VC_LAUNCH_CHECK("layernorm_bwd");
  return VITCAP_OK;
}

extern "C" int vitcap_reduce_slabs(const float* slabs, size_t slab_stride, int S, float* out, size_t n, int accumulate,
                                   void* stream) {
  VC_REQUIRE(slabs && out && S >= 1 && n % 4 == 0, "reduce_slabs: bad arguments");
  const size_t n4 = n / 4;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, slabs,
                     slab_stride, S, out, n4, accumulate);
  VC_LAUNCH_CHECK("reduce_slabs");
  return VITCAP_OK;
}

extern "C" int vitcap_cast_bf16(const float* x, void* y, size_t n, void* stream) {
  VC_REQUIRE(x && y && n % 4 == 0, "cast_bf16: bad arguments");
  const size_t n4 = n / 4;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y,
                     n4);
  VC_LAUNCH_CHECK("cast_bf16");
  return VITCAP_OK;
}

extern "C" int vitcap_embed_bwd(const float* d, const int64_t* ids, int rows_per_seq, float* gword, float* gpos, float* gtype,
                                int rows, int pos_wrap, void* stream) {
  VC_REQUIRE(d && ids && gword && gpos && gtype && rows > 0 && rows_per_seq > 0 && rows % rows_per_seq == 0,
             "embed_bwd: bad arguments (rows %d must be whole sequences of %d)", rows, rows_per_seq);
  hipLaunchKernelGGL(embed_bwd_kernel, dim3((rows + rows_per_seq + 3) / 4), dim3(256), 0, (hipStream_t)stream, d, ids, rows_per_seq, gword,
                     gpos, gtype, rows, pos_wrap);
  VC_LAUNCH_CHECK("embed_bwd");
  return VITCAP_OK;
}

extern "C" int vitcap_ls_kl_loss(const float* logits, int ldl, int V, const int64_t* target, float eps, int rows,
                                 const float* row_weight, float* loss_sum, void* dlogits_bf16, int ldd, void* stream) {
  VC_REQUIRE(logits && target && loss_sum && rows > 0 && V > 1, "ls_kl_loss: bad arguments");
  hipLaunchKernelGGL(ls_kl_kernel, dim3(rows), dim3(1024), 0, (hipStream_t)stream, logits, ldl, V, target, eps,
                     1.0f / (float)rows, row_weight, loss_sum, (bf16_t*)dlogits_bf16, ldd);
  VC_LAUNCH_CHECK("ls_kl_loss");
  return VITCAP_OK;
}

extern "C" int vitcap_focal_loss_sum(const float* logits, int ldl, int V, const float* label, float alpha, float* out, int B,
                                     void* stream) {
  VC_REQUIRE(logits && label && out && B > 0, "focal_loss: bad arguments");
  hipLaunchKernelGGL(focal_sum_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, logits, ldl, V, label, alpha, out, B);
  VC_LAUNCH_CHECK("focal_loss");
  return VITCAP_OK;
}

extern "C" int vitcap_bce_logits_mean(const float* logits, int ldl, int V, const float* label, float* out, int B, void* stream) {
  VC_REQUIRE(logits && label && out && B > 0 && V > 0, "bce_logits_mean: bad arguments");
  hipLaunchKernelGGL(bce_sum_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, logits, ldl, V, label,
                     1.0f / ((float)B * (float)V), out, B);
  VC_LAUNCH_CHECK("bce_logits_mean");
  return VITCAP_OK;
}

extern "C" int vitcap_sumsq(const float* g, size_t n, float* out, void* stream) {
  VC_REQUIRE(g && out && n % 4 == 0, "sumsq: bad arguments");
  static float* scratch[64] = {nullptr};          // per-device partials, allocated on first use
  int dev = 0;
  VC_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, "sumsq: bad device");
  if (!scratch[dev]) VC_REQUIRE(hipMalloc(&scratch[dev], SUMSQ_BLOCKS * sizeof(float)) == hipSuccess, "sumsq: scratch alloc failed");
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, (hipStream_t)stream, g, n / 4, scratch[dev]);
  VC_LAUNCH_CHECK("sumsq_partial");
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)scratch[dev], out);
  VC_LAUNCH_CHECK("sumsq_final");
  return VITCAP_OK;
}

extern "C" int vitcap_adamw_multi(float* p, const float* g, float* m, float* v, const float* chunk_lr, const float* chunk_wd,
                                  const float* gsumsq, float clip, float lr_scale, int step, float b1, float b2, float eps,
                                  size_t nchunks, void* stream) {
  VC_REQUIRE(p && g && m && v && chunk_lr && chunk_wd && gsumsq && step >= 1 && nchunks > 0, "adamw: bad arguments");
  const float bc1 = 1.0f - powf(b1, (float)step);
  const float bc2s = sqrtf(1.0f - powf(b2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, chunk_lr, chunk_wd,
                     gsumsq, clip, lr_scale, bc1, bc2s, b1, b2, eps, nchunks);
  VC_LAUNCH_CHECK("adamw");
  return VITCAP_OK;
}

extern "C" int vitcap_cast_transpose_multi(const vitcap_ct_item* items_dev, int n_items, int total_tiles, void* stream) {
  VC_REQUIRE(items_dev && n_items > 0 && total_tiles > 0, "cast_transpose_multi: bad arguments");
  hipLaunchKernelGGL(cast_transpose_multi_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, items_dev, n_items);
  VC_LAUNCH_CHECK("cast_transpose_multi");
  return VITCAP_OK;
}

extern "C" int vitcap_cast_transpose(const float* w, void* w_bf16, void* wt_bf16, int N, int K, int ldt, void* stream) {
  VC_REQUIRE(w && (w_bf16 || wt_bf16) && N > 0 && K > 0 && K % 64 == 0 && ldt >= N && ldt % 8 == 0,
             "cast_transpose: K must be a multiple of 64, ldt >= N");
  dim3 grid((N + 63) / 64, K / 64);
  hipLaunchKernelGGL(cast_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)w_bf16, (bf16_t*)wt_bf16, N, K,
                     ldt);
  VC_LAUNCH_CHECK("cast_transpose");
  return VITCAP_OK;
}


// ---- hidden-state dropout of the BERT parts in training (BertEmbeddings modeling_bert.py:236, BertSelfOutput :355, BertOutput :417;
// BertConfig.hidden_dropout_prob = the pipeline's `drop_out`, 0 in the shipped YAML, 0.1 by the pipeline's own default):
//   out[m][d] = x[m][d] * keep(b, r, d) / (1 - p) (+ residual[m][d]),   b = m / rows_per_seq, r = row0 + m % rows_per_seq
// keep is a pure function of (seed, b, r, d) (csrc/rng.h), so the backward pass recomputes it from the same seed -- no mask is stored
// -- and the CPU oracle replays it (oracle.hidden_keep).  The same entry point serves the backward pass: dx = dy * keep / (1 - p).
namespace {
__global__ __launch_bounds__(256) void hidden_dropout_kernel(const float* __restrict__ x, const float* __restrict__ res, float* __restrict__ out,
                                                           int M, int rows_per_seq, int row0, uint32_t seed, uint32_t thr, float scale,
                                                           const uint32_t* __restrict__ drop_salt) {
  const int gid = blockIdx.x * 256 + threadIdx.x;          // one float4 per thread: 192 per row
  const int m = gid / 192, c4 = gid - m * 192;
  if (m >= M) return;
  const int b = m / rows_per_seq, r = row0 + (m - b * rows_per_seq);
  const uint32_t stream = vc_drop_stream(vc_salted(seed, drop_salt), (uint32_t)b, 0x48u);
  const size_t off = (size_t)m * D768 + c4 * 4;
  f32x4 v = *(const f32x4*)(x + off);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = vc_drop_keep(stream, (uint32_t)r, (uint32_t)(c4 * 4 + e), thr) ? v[e] * scale : 0.f;
  if (res) v += *(const f32x4*)(res + off);
  *(f32x4*)(out + off) = v;
}
}  // namespace

extern "C" int vitcap_hidden_dropout(const float* x, const float* residual, float* out, int M, int D, int rows_per_seq, int row0,
                                     uint32_t seed, float p, void* stream) {
  VC_REQUIRE(x && out && M > 0 && D == D768 && rows_per_seq > 0 && row0 >= 0 && row0 + rows_per_seq <= 1024,
             "hidden_dropout: bad arguments (D must be 768, rows of a sequence < 1024)");
  VC_REQUIRE(p >= 0.f && p < 1.f, "hidden_dropout: p must be in [0, 1)");
  const uint32_t thr = (uint32_t)((double)p * 4294967296.0);
  const float scale = 1.0f / (1.0f - p);
  const long long n4 = (long long)M * 192;
  hipLaunchKernelGGL(hidden_dropout_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, residual, out, M,
                     rows_per_seq, row0, seed, thr, scale, vc_tls_drop_salt);
  VC_LAUNCH_CHECK("hidden_dropout");
  return VITCAP_OK;
}
