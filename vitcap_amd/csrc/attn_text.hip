// Caption-row attention of the decoder in TRAINING (teacher forcing): per image the joint sequence is laid out as
// [S_vis visual rows | T caption rows] (T = 20).  Caption row r attends every visual row and caption rows j <= r
// (the seq2seq mask of dataset.py:377-390 + ..._bertemb.py:57-85; visual rows are handled by the dense kernels).
// T is tiny, so this is a vector-ALU kernel: one workgroup per (image, head), scores for the T x (S_vis+T) block are
// kept in LDS.  The backward recomputes the probabilities, writes dQ for the caption rows, dK/dV for the caption
// rows directly into dqkv and the caption rows' contribution to the VISUAL keys' dK/dV into `extra`
// (bf16 [B*ld_rows][2][768]), which attn_bwd_dkv_kernel adds to its own result.
#include "common.h"
#include "rng.h"

namespace {

constexpr int HD = 64;
constexpr int NH = 12;
constexpr int QKV_LD = 2304;
constexpr int TMAX = 20;
constexpr int KMAX = 608;     // >= S_vis + T, multiple of 32

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// scores + softmax into sc[T][KMAX] (normalised probabilities); returns nothing.  All 256 threads participate.
// With dropout (drop_thr != 0) a dropped probability is stored with its sign bit set: |sc| is P, sign = dropped.
__device__ __forceinline__ bool dropped(float v) { return (__float_as_uint(v) >> 31) != 0; }

__device__ __forceinline__ void text_probs(const bf16_t* base, int S_vis, int T, int nkeys, float c_log2, float (*sc)[KMAX],
                                           const float (*qs)[HD], int tid, uint32_t drop_stream, uint32_t drop_thr) {
  const int lane = tid & 63, w = tid >> 6;
  for (int k = tid; k < nkeys; k += 256) {
    const bf16_t* kr = base + (size_t)k * QKV_LD + 768;
    float kf[HD];
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
      const bf16x8 v = *(const bf16x8*)(kr + c8 * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) kf[c8 * 8 + j] = (float)v[j];
    }
    for (int r = 0; r < T; ++r) {
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < HD; ++j) d += qs[r][j] * kf[j];
      const bool ok = k < S_vis || (k - S_vis) <= r;
      sc[r][k] = ok ? d : -INFINITY;
    }
  }
  __syncthreads();
  for (int r = w; r < T; r += 4) {
    float mx = -INFINITY;
    for (int k = lane; k < nkeys; k += 64) mx = fmaxf(mx, sc[r][k]);
    mx = wave_max(mx);
    const float m = ceilf(mx * c_log2);
    float se = 0.f;
    for (int k = lane; k < nkeys; k += 64) {
      const float p = fast_exp2(fmaf(sc[r][k], c_log2, -m));
      sc[r][k] = p;
      se += p;
    }
    se = wave_sum(se);
    const float inv = 1.0f / se;
    for (int k = lane; k < nkeys; k += 64) {
      float pn = sc[r][k] * inv;
      if (drop_thr != 0u && !vc_drop_keep(drop_stream, (uint32_t)(S_vis + r), (uint32_t)k, drop_thr))
        pn = __uint_as_float(__float_as_uint(pn) | 0x80000000u);
      sc[r][k] = pn;
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void attn_text_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int S_vis,
                                                            int T, int ld_rows, float c_log2, uint32_t drop_seed,
                                                            uint32_t drop_thr, float drop_scale) {
  __shared__ float sc[TMAX][KMAX];
  __shared__ float qs[TMAX][HD];
  const int tid = threadIdx.x;
  const int h = blockIdx.x, b = blockIdx.y;
  const bf16_t* base = qkv + (size_t)b * ld_rows * QKV_LD + h * HD;
  const int nkeys = S_vis + T;
  for (int i = tid; i < T * HD; i += 256) qs[i / HD][i % HD] = bf2f(base[(size_t)(S_vis + i / HD) * QKV_LD + i % HD]);
  __syncthreads();
  text_probs(base, S_vis, T, nkeys, c_log2, sc, qs, tid, vc_drop_stream(drop_seed, (uint32_t)b, (uint32_t)h), drop_thr);
  // O[r][d] = sum_k bf16(P[r][k]) V[k][d];  thread (d, rg): rows rg*5 .. rg*5+4
  const int d = tid & 63, rg = tid >> 6;
  float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < nkeys; ++k) {
    const float v = bf2f(base[(size_t)k * QKV_LD + 1536 + d]);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int r = rg * 5 + i;
      if (r < T) acc[i] += (dropped(sc[r][k]) ? 0.f : (float)(__bf16)sc[r][k]) * v;
    }
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int r = rg * 5 + i;
    if (r < T) out[((size_t)b * ld_rows + S_vis + r) * 768 + h * HD + d] = f2bf(acc[i] * drop_scale);
  }
}

__global__ __launch_bounds__(256) void attn_text_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                            bf16_t* __restrict__ dqkv, bf16_t* __restrict__ extra, int S_vis,
                                                            int T, int ld_rows, float c_log2, float scale, uint32_t drop_seed,
                                                            uint32_t drop_thr, float drop_scale) {
  __shared__ float sc[TMAX][KMAX];     // P
  __shared__ float ds[TMAX][KMAX];     // dP, then dS
  __shared__ float qs[TMAX][HD];
  __shared__ float dos[TMAX][HD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = blockIdx.x, b = blockIdx.y;
  const bf16_t* base = qkv + (size_t)b * ld_rows * QKV_LD + h * HD;
  const int nkeys = S_vis + T;
  for (int i = tid; i < T * HD; i += 256) {
    const int r = i / HD, c = i % HD;
    qs[r][c] = bf2f(base[(size_t)(S_vis + r) * QKV_LD + c]);
    dos[r][c] = bf2f(dout[((size_t)b * ld_rows + S_vis + r) * 768 + h * HD + c]);
  }
  __syncthreads();
  text_probs(base, S_vis, T, nkeys, c_log2, sc, qs, tid, vc_drop_stream(drop_seed, (uint32_t)b, (uint32_t)h), drop_thr);
  // dP[r][k] = (dO[r] . V[k]) o keep / (1-p)
  for (int k = tid; k < nkeys; k += 256) {
    const bf16_t* vr = base + (size_t)k * QKV_LD + 1536;
    float vf[HD];
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
      const bf16x8 v = *(const bf16x8*)(vr + c8 * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) vf[c8 * 8 + j] = (float)v[j];
    }
    for (int r = 0; r < T; ++r) {
      float dsum = 0.f;
#pragma unroll
      for (int j = 0; j < HD; ++j) dsum += dos[r][j] * vf[j];
      ds[r][k] = dropped(sc[r][k]) ? 0.f : dsum * drop_scale;
    }
  }
  __syncthreads();
  // dS = P o (dP - D), D[r] = sum_k P dP
  for (int r = w; r < T; r += 4) {
    float dd = 0.f;
    for (int k = lane; k < nkeys; k += 64) dd += fabsf(sc[r][k]) * ds[r][k];
    dd = wave_sum(dd);
    for (int k = lane; k < nkeys; k += 64) ds[r][k] = (float)(__bf16)(fabsf(sc[r][k]) * (ds[r][k] - dd));
  }
  __syncthreads();
  // dQ[r][d] = scale * sum_k dS[r][k] K[k][d]
  {
    const int d = tid & 63, rg = tid >> 6;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < nkeys; ++k) {
      const float kv = bf2f(base[(size_t)k * QKV_LD + 768 + d]);
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int r = rg * 5 + i;
        if (r < T) acc[i] += ds[r][k] * kv;
      }
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int r = rg * 5 + i;
      if (r < T) dqkv[((size_t)b * ld_rows + S_vis + r) * QKV_LD + h * HD + d] = f2bf(acc[i] * scale);
    }
  }
  // dK[k][d] = scale * sum_r dS[r][k] Q[r][d];  dV[k][d] = sum_r bf16(P[r][k]) dO[r][d]
  for (int k = tid; k < nkeys; k += 256) {
    float dk[HD], dv[HD];
#pragma unroll
    for (int j = 0; j < HD; ++j) { dk[j] = 0.f; dv[j] = 0.f; }
    for (int r = 0; r < T; ++r) {
      const float s_ = ds[r][k], p_ = dropped(sc[r][k]) ? 0.f : (float)(__bf16)sc[r][k];
#pragma unroll
      for (int j = 0; j < HD; ++j) {
        dk[j] += s_ * qs[r][j];
        dv[j] += p_ * dos[r][j];
      }
    }
    bf16_t* ok;
    bf16_t* ov;
    if (k < S_vis) {
      ok = extra + ((size_t)b * ld_rows + k) * 1536 + h * HD;
      ov = ok + 768;
    } else {
      ok = dqkv + ((size_t)b * ld_rows + k) * QKV_LD + 768 + h * HD;
      ov = ok + 768;
    }
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
      uint4 a, c;
      a.x = pack2bf(dk[c8 * 8 + 0] * scale, dk[c8 * 8 + 1] * scale);
      a.y = pack2bf(dk[c8 * 8 + 2] * scale, dk[c8 * 8 + 3] * scale);
      a.z = pack2bf(dk[c8 * 8 + 4] * scale, dk[c8 * 8 + 5] * scale);
      a.w = pack2bf(dk[c8 * 8 + 6] * scale, dk[c8 * 8 + 7] * scale);
      c.x = pack2bf(dv[c8 * 8 + 0] * drop_scale, dv[c8 * 8 + 1] * drop_scale);
      c.y = pack2bf(dv[c8 * 8 + 2] * drop_scale, dv[c8 * 8 + 3] * drop_scale);
      c.z = pack2bf(dv[c8 * 8 + 4] * drop_scale, dv[c8 * 8 + 5] * drop_scale);
      c.w = pack2bf(dv[c8 * 8 + 6] * drop_scale, dv[c8 * 8 + 7] * drop_scale);
      *(uint4*)(ok + c8 * 8) = a;
      *(uint4*)(ov + c8 * 8) = c;
    }
  }
}

}  // namespace

extern "C" int vitcap_attn_text_fwd(const void* qkv, void* out, int B, int S_vis, int T, int ld_rows, float scale,
                                    float p_drop, uint32_t drop_seed, void* stream) {
  VC_REQUIRE(qkv && out && B > 0 && T >= 1 && T <= TMAX && S_vis + T <= KMAX && ld_rows >= S_vis + T,
             "attn_text_fwd: bad arguments (T <= %d, S_vis+T <= %d)", TMAX, KMAX);
  VC_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "attn_text_fwd: p_drop %g out of range", (double)p_drop);
  hipLaunchKernelGGL(attn_text_fwd_kernel, dim3(NH, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (bf16_t*)out,
                     S_vis, T, ld_rows, scale * 1.4426950408889634f, drop_seed,
                     (uint32_t)((double)p_drop * 4294967296.0), 1.0f / (1.0f - p_drop));
  VC_LAUNCH_CHECK("attn_text_fwd");
  return VITCAP_OK;
}

extern "C" int vitcap_attn_text_bwd(const void* qkv, const void* dout, void* dqkv, void* extra_dkv, int B, int S_vis, int T,
                                    int ld_rows, float scale, float p_drop, uint32_t drop_seed, void* stream) {
  VC_REQUIRE(qkv && dout && dqkv && extra_dkv && B > 0 && T >= 1 && T <= TMAX && S_vis + T <= KMAX && ld_rows >= S_vis + T,
             "attn_text_bwd: bad arguments");
  VC_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "attn_text_bwd: p_drop %g out of range", (double)p_drop);
  hipLaunchKernelGGL(attn_text_bwd_kernel, dim3(NH, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv,
                     (const bf16_t*)dout, (bf16_t*)dqkv, (bf16_t*)extra_dkv, S_vis, T, ld_rows, scale * 1.4426950408889634f,
                     scale, drop_seed, (uint32_t)((double)p_drop * 4294967296.0), 1.0f / (1.0f - p_drop));
  VC_LAUNCH_CHECK("attn_text_bwd");
  return VITCAP_OK;
}
