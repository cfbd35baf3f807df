// Backward of the dense (unmasked) multi-head attention over packed qkv, gfx950.
//
// Given dO and the forward's log2-domain logsumexp L[q] (P[q][k] = exp2(c*s[q][k] - L[q]) is the NORMALISED
// probability), with D[q] = sum_d dO[q][d]*O[q][d]:
//     dV[k] = sum_q P[q][k] dO[q]          dP[q][k] = dO[q] . V[k]         dS = P o (dP - D[q])
//     dQ[q] = scale * sum_k dS[q][k] K[k]   dK[k] = scale * sum_q dS[q][k] Q[q]
// Two kernels, both built like the forward (32x32x16 MFMA, transposed score tiles so per-row quantities are
// lane-local, the MFMA C layout reused as the next B operand, no atomics):
//   attn_bwd_dq_dma_kernel  : a wave owns 32 queries and sweeps 64-key tiles:  S^T = K Q^T, dP^T = V dO^T,
//                             dQ^T += K^T dS^T.  Also writes D[q] for the second kernel.
//   attn_bwd_dkv_dma_kernel : a wave owns 32 keys and sweeps 64-query tiles:   S = Q K^T, dP = dO V^T,
//                             dV^T += dO^T P, dK^T += Q^T dS.
// P and dS are rounded to bf16 for the MFMAs (fp32 accumulation), like the forward's P.
// Left-over rows (S = 577/578 leaves 1/2) are handled on the vector ALU.
//
// Data path (round 5; rounds 1-4 staged the tiles through VGPRs and built transposed copies with v_perm: 15-19 % slower per
// launch, same bits -- profiles/r05_train_attn_bwd_ab.txt): tiles travel HBM/L2 -> LDS by LDS-DMA (global_load_lds, 16 B per lane,
// no VGPR round trip, no ds_write), ROW-major (64 rows x 128 B) with XOR-swizzled 16-byte chunks, in a ring of three stages on
// counted vmcnt waits; the row-major operands (S, dP) are read with ds_read_b128, the transposed ones (K^T, Q^T, dO^T) straight from
// the same tiles with ds_read_b64_tr_b16 -- the two runs of 4 consecutive rows a lane needs per 16-row MFMA step are exactly the C
// layout of the score tile (see attn.hip).
#include "common.h"
#include "rng.h"

namespace {

constexpr int HD = 64;
constexpr int NH = 12;
constexpr int QKV_LD = 2304;
constexpr int KT = 64;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// build knobs measured on one box against the shipped form (profiles/r05_train_attn_bwd_ab.txt): all within the run-to-run spread
#ifndef VC_BWD_PRIO          // 1: s_setprio(1) around the MFMA bursts (two waves of different workgroups share a SIMD)
#define VC_BWD_PRIO 0
#endif
#define VC_BWD_PRIO_ON() do { if (VC_BWD_PRIO) __builtin_amdgcn_s_setprio(1); } while (0)
#define VC_BWD_PRIO_OFF() do { if (VC_BWD_PRIO) __builtin_amdgcn_s_setprio(0); } while (0)
#ifndef VC_BWD_PREFETCH      // 1: the row-major fragments of a tile's second half are requested behind the first half's S / dP MFMAs
#define VC_BWD_PREFETCH 0
#endif
// waves per SIMD the dQ kernel is compiled for.  3 = 168 registers: reached with the S^T and the dP^T MFMA chains one after the other (one set
// of row-major fragments live at a time) and the tail mode as a template parameter (an instantiation holds either the left-over-key loop or
// the masked tail tile); what the allocator still spills (13 registers in the encoder's instantiation) is stored once in front of the loop
// and re-read once behind it -- checked in the ISA, no scratch access inside the loop.  Per launch -2.4 % (encoder) / -3.5 % (decoder)
// against two waves, the training step within its spread (profiles/r05_train_attn_bwd_ab.txt).
#ifndef VC_BWD_DQ_MINW
#define VC_BWD_DQ_MINW 3
#endif
constexpr int DT_B = KT * 128;               // one row-major tile: 64 rows x 128 B
constexpr int NSTG = 3;                      // ring: tile t+2 is in flight while tile t is multiplied
constexpr int DQ_STG = 2 * DT_B;             // dq stage:  [K | V]
constexpr int DKV_STG = 2 * DT_B + 1024;     // dkv stage: [Q | dO | L[64] | D[64] | 512 B the other two waves' copies land in]

__device__ __forceinline__ void glds16(const void* g, void* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const void* g, void* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 4, 0, 0);
}
__device__ __forceinline__ int kv_swz(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }   // attn.hip

// per-lane read offsets inside a row-major tile: rows (lane & 31) (+32) as ds_read_b128 fragments, and the transpose reads of a
// 16-row block (attn.hip: voff)
#define VC_BWD_READ_OFFSETS()                                                                     \
  int koff[4];                                                                                    \
  _Pragma("unroll") for (int ds = 0; ds < 4; ++ds) koff[ds] = (lane & 31) * 128 + (((2 * ds + half) ^ kv_swz(lane & 31)) * 16); \
  int toff[2][2];                                                                                 \
  {                                                                                               \
    const int j_ = (lane & 15) >> 2;                                                              \
    const int c2_ = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);                                    \
    _Pragma("unroll") for (int rd = 0; rd < 2; ++rd) {                                            \
      const int r_ = 8 * rd + 4 * half + j_;                                                      \
      _Pragma("unroll") for (int dt = 0; dt < 2; ++dt)                                            \
        toff[rd][dt] = r_ * 128 + (((c2_ + 4 * dt) ^ kv_swz(r_)) * 16) + (lane & 1) * 8;          \
    }                                                                                             \
  }
// joint_visible (common.h) without short-circuit control flow: in the peeled masked instances of the tile macros below the && / || form
// became a divergent branch per score and pushed the kernels into scratch (57 spilled registers); same truth table
__device__ __forceinline__ bool visible_nb(int q, int k, int S, int cf, int mf) {
  const bool plain = (cf <= 0) | (k < cf);
  const bool probe_q = (mf > 0) & (q >= mf), probe_k = (mf > 0) & (k >= mf);
  const bool text = probe_k ? (k == q) : (probe_q ? ((k - cf) <= (q - mf)) : (k <= q));
  return (k < S) & (plain | text);
}
#define VC_ZERO16 f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}

template <bool DROP, bool TAIL>
__global__ __launch_bounds__(256, VC_BWD_DQ_MINW) void attn_bwd_dq_dma_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                                 const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                                 float* __restrict__ dsum, bf16_t* __restrict__ dqkv, int S, int B,
                                                                 int ld_rows, float c_log2, float scale, uint32_t drop_seed,
                                                                 uint32_t drop_thr, float drop_scale, int causal_from, int mask_from,
                                                                 int q_lo, int q_hi, const uint32_t* __restrict__ drop_salt) {
  __shared__ __attribute__((aligned(1024))) char smem[NSTG * DQ_STG];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 31, half = lane >> 5;
  const int nqb = (q_hi - q_lo + 127) / 128;
  const int nwork = nqb * NH * B;
  int wid = blockIdx.x;
  {
    const int qd = nwork >> 3, rm = nwork & 7, xcd = wid & 7;
    wid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (wid >> 3);
  }
  const int qb = (q_lo >> 7) + wid % nqb;
  const int h = (wid / nqb) % NH, b = wid / (nqb * NH);
  const int q0 = qb * 128 + w * 32;
  const bool active = q0 < S;
  const bf16_t* base = qkv + (size_t)b * ld_rows * QKV_LD + h * HD;
  const int qr = (q0 + qi) < S ? (q0 + qi) : S - 1;
  const uint32_t hq = DROP ? (vc_drop_stream(vc_salted(drop_seed, drop_salt), (uint32_t)b, (uint32_t)h) ^ ((uint32_t)(q0 + qi) << 10)) : 0u;

  bf16x8 qf[4], dof[4];
  float dpart = 0.f;
  {
    const bf16_t* qp = base + (size_t)qr * QKV_LD + half * 8;
    const bf16_t* dp = dout + ((size_t)b * ld_rows + qr) * 768 + h * HD + half * 8;
    const bf16_t* op = o + ((size_t)b * ld_rows + qr) * 768 + h * HD + half * 8;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      qf[ds] = *(const bf16x8*)(qp + ds * 16);
      dof[ds] = *(const bf16x8*)(dp + ds * 16);
      const bf16x8 of = *(const bf16x8*)(op + ds * 16);
#pragma unroll
      for (int j = 0; j < 8; ++j) dpart += (float)dof[ds][j] * (float)of[j];
    }
  }
  const float Dq = dpart + __shfl_xor(dpart, 32, 64);          // D[q] = dO[q] . O[q]
  const float Lq = lse[((size_t)b * NH + h) * S + qr];
  if (active && half == 0 && q0 + qi < S) dsum[((size_t)b * NH + h) * S + q0 + qi] = Dq;

  const int nfull = S / KT, rem = S - nfull * KT;
  const bool tail_tile = rem > 8 || (causal_from > 0 && rem > 0);
  const int ntiles = nfull + (tail_tile ? 1 : 0);

  // ---- staging (attn.hip STAGE_TILE): wave w moves rows [8w, 8w+8) and [32+8w, 32+8w+8) of the K and of the V tile
  const int srow = w * 8 + (lane >> 3);
  const uint32_t s_chunk = (uint32_t)(((lane & 7) ^ kv_swz(srow)) * 16);
  const char* gbase = (const char*)base + 768 * 2;               // K columns of this head; V is 768 elements further
#define STAGE_TILE(t_, stg_)                                                                       \
  do {                                                                                             \
    char* sb_ = smem + (stg_) * DQ_STG + w * 1024;                                                 \
    int r0_ = (t_) * KT + srow, r1_ = r0_ + 32;                                                    \
    r0_ = r0_ < S ? r0_ : S - 1;                 /* tail tile: rows past the sequence re-read its last row (masked) */ \
    r1_ = r1_ < S ? r1_ : S - 1;                                                                   \
    const char* a0_ = gbase + (size_t)r0_ * (QKV_LD * 2) + s_chunk;                                \
    const char* a1_ = gbase + (size_t)r1_ * (QKV_LD * 2) + s_chunk;                                \
    glds16(a0_, sb_);                                                                              \
    glds16(a1_, sb_ + 4096);                                                                       \
    glds16(a0_ + 768 * 2, sb_ + DT_B);                                                             \
    glds16(a1_ + 768 * 2, sb_ + DT_B + 4096);                                                      \
  } while (0)
  VC_BWD_READ_OFFSETS();

  f32x16 dqt[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    dqt[0][r] = 0.f;
    dqt[1][r] = 0.f;
  }
  if (ntiles > 0) STAGE_TILE(0, 0);
  if (ntiles > 1) STAGE_TILE(1, 1);
  // every VGPR load so far provably complete on all paths into the loop (attn.hip: "waitcnt false dependency")
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)

#define DQ_ELEM(KT_, t_, MASKED_)                                                                                   \
  do {                                                                                                               \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                                 \
      float p = fast_exp2(fmaf(st[r], c_log2, -Lq));                                                                 \
      const int key = (t_) * KT + (KT_) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;                                    \
      if (MASKED_) p = visible_nb(q0 + qi, key, S, causal_from, mask_from) ? p : 0.f;                                \
      float dp = dpt[r];                                                                                             \
      if (DROP) dp = vc_lowbias32(hq ^ (uint32_t)key) >= drop_thr ? dp * drop_scale : 0.f;                           \
      st[r] = p * (dp - Dq);        /* dS^T */                                                                       \
    }                                                                                                                \
  } while (0)
  // one 32-key half of a tile: S^T = K Q^T, dP^T = V dO^T, dS^T, dQ^T += K^T dS^T (KT_ literal: the transpose reads take immediates)
#define DQ_HALF(KT_, t_)                                                                                    \
  do {                                                                                                               \
    f32x16 st, dpt;                                                                                                  \
    {                                                                                                                \
      bf16x8 kfr[4];                                                                                                 \
      _Pragma("unroll") for (int ds = 0; ds < 4; ++ds) kfr[ds] = *(const bf16x8*)(kl + (KT_) * 4096 + koff[ds]);     \
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[0], qf[0], VC_ZERO16, 0, 0, 0);                               \
      _Pragma("unroll") for (int ds = 1; ds < 4; ++ds) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[ds], qf[ds], st, 0, 0, 0); \
    }                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    {                                                                                                                \
      bf16x8 vfr[4];                                                                                                 \
      _Pragma("unroll") for (int ds = 0; ds < 4; ++ds) vfr[ds] = *(const bf16x8*)(kl + DT_B + (KT_) * 4096 + koff[ds]); \
      dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[0], dof[0], VC_ZERO16, 0, 0, 0);                             \
      _Pragma("unroll") for (int ds = 1; ds < 4; ++ds) dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[ds], dof[ds], dpt, 0, 0, 0); \
    }                                                                                                                \
    /* the transposed K fragments of this half go out now: their latency hides behind the elementwise part */       \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    s16x4 ktr[2][2][2];                                                                                              \
    _Pragma("unroll") for (int dt = 0; dt < 2; ++dt)                                                                 \
      _Pragma("unroll") for (int rd = 0; rd < 2; ++rd) {                                                             \
        ktr[0][dt][rd] = lds_tr_read<(KT_) * 4096>(kb_ + (uint32_t)toff[rd][dt]);                                    \
        ktr[1][dt][rd] = lds_tr_read<(KT_) * 4096 + 2048>(kb_ + (uint32_t)toff[rd][dt]);                             \
      }                                                                                                              \
    if (VC_BWD_PREFETCH && (KT_) == 0) {      /* the second half's row-major fragments: in flight during this half's elementwise part */ \
      _Pragma("unroll") for (int ds = 0; ds < 4; ++ds) {                                                             \
        knx[ds] = *(const bf16x8*)(kl + 4096 + koff[ds]);                                                            \
        vnx[ds] = *(const bf16x8*)(kl + DT_B + 4096 + koff[ds]);                                                     \
      }                                                                                                              \
      __builtin_amdgcn_sched_barrier(0);                                                                             \
    }                                                                                                                \
    if (TAIL && masked_) DQ_ELEM(KT_, t_, true);  /* wave-uniform: only the elementwise part exists twice */          \
    else DQ_ELEM(KT_, t_, false);                                                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                              \
                 : "+v"(ktr[0][0][0]), "+v"(ktr[0][0][1]), "+v"(ktr[0][1][0]), "+v"(ktr[0][1][1]),                   \
                   "+v"(ktr[1][0][0]), "+v"(ktr[1][0][1]), "+v"(ktr[1][1][0]), "+v"(ktr[1][1][1]));                  \
    VC_BWD_PRIO_ON();                                                                                                \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                               \
      bf16x8 sf;                                                                                                     \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) sf[j] = (__bf16)st[ks * 8 + j];                                  \
      _Pragma("unroll") for (int dt = 0; dt < 2; ++dt)                                                               \
        dqt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_pair(ktr[ks][dt][0], ktr[ks][dt][1]), sf, dqt[dt], 0, 0, 0); \
    }                                                                                                                \
    VC_BWD_PRIO_OFF();                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  } while (0)
#define DQ_TILE(stg_, t_)                                                                                   \
  do {                                                                                                               \
    const char* kl = smem + (stg_) * DQ_STG;                                                                         \
    const uint32_t kb_ = lds_addr(kl);                                                                               \
    bf16x8 knx[4], vnx[4];                                                                                           \
    DQ_HALF(0, t_);                                                                                                  \
    DQ_HALF(1, t_);                                                                                                  \
  } while (0)

  int stg = 0;
  for (int t = 0; t < ntiles; ++t) {
    // tile t landed (this wave's 4 pieces; tile t+1's 4 may stay in flight), then for every wave -- and every wave is done with
    // tile t-1, whose slot the next request overwrites
    if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + 2 < ntiles) STAGE_TILE(t + 2, stg == 0 ? 2 : stg - 1);
    if (active) {
      const bool masked_ = t >= nfull;
      DQ_TILE(stg, t);
    }
    stg = stg == 2 ? 0 : stg + 1;
  }
#undef DQ_TILE
#undef DQ_HALF
#undef DQ_ELEM
#undef STAGE_TILE
  // left-over keys on the vector ALU (TAIL = false: the host picks the instantiation by the same rule as tail_tile)
  if (!TAIL && active && !tail_tile) {
    for (int key = nfull * KT; key < S; ++key) {
      const bf16_t* kr = base + (size_t)key * QKV_LD + 768 + half * 8;
      const bf16_t* vr = base + (size_t)key * QKV_LD + 1536 + half * 8;
      float sp = 0.f, dp = 0.f;
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) {
        const bf16x8 kv = *(const bf16x8*)(kr + ds * 16);
        const bf16x8 vv = *(const bf16x8*)(vr + ds * 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          sp += (float)qf[ds][j] * (float)kv[j];
          dp += (float)dof[ds][j] * (float)vv[j];
        }
      }
      sp += __shfl_xor(sp, 32, 64);
      dp += __shfl_xor(dp, 32, 64);
      const float p = fast_exp2(fmaf(sp, c_log2, -Lq));
      if (DROP) dp = vc_lowbias32(hq ^ (uint32_t)key) >= drop_thr ? dp * drop_scale : 0.f;
      const float dsb = (float)(__bf16)(p * (dp - Dq));
      const bf16_t* ko = base + (size_t)key * QKV_LD + 768 + 4 * half;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const bf16x4 kk = *(const bf16x4*)(ko + dt * 32 + g * 8);
#pragma unroll
          for (int e = 0; e < 4; ++e) dqt[dt][g * 4 + e] = fmaf(dsb, (float)kk[e], dqt[dt][g * 4 + e]);
        }
    }
  }
  const int q = q0 + qi;
  if (q < S) {
    bf16_t* op = dqkv + ((size_t)b * ld_rows + q) * QKV_LD + h * HD + 4 * half;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 ov;
        ov.x = pack2bf(dqt[dt][g * 4 + 0] * scale, dqt[dt][g * 4 + 1] * scale);
        ov.y = pack2bf(dqt[dt][g * 4 + 2] * scale, dqt[dt][g * 4 + 3] * scale);
        *(uint2*)(op + dt * 32 + g * 8) = ov;
      }
  }
}

template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_dma_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                                  const float* __restrict__ lse, const float* __restrict__ dsum,
                                                                  const bf16_t* __restrict__ extra, bf16_t* __restrict__ dqkv, int S,
                                                                  int B, int ld_rows, float c_log2, float scale, uint32_t drop_seed,
                                                                  uint32_t drop_thr, float drop_scale, int causal_from, int mask_from,
                                                                  int q_lo, int q_hi, const uint32_t* __restrict__ drop_salt) {
  __shared__ __attribute__((aligned(1024))) char smem[NSTG * DKV_STG];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ki = lane & 31, half = lane >> 5;
  const int nkb = (S + 127) / 128;
  const int nwork = nkb * NH * B;
  int wid = blockIdx.x;
  {
    const int qd = nwork >> 3, rm = nwork & 7, xcd = wid & 7;
    wid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (wid >> 3);
  }
  const int kb_ = wid % nkb;
  const int h = (wid / nkb) % NH, b = wid / (nkb * NH);
  const int key0 = kb_ * 128 + w * 32;
  const bool active = key0 < S;
  const bf16_t* base = qkv + (size_t)b * ld_rows * QKV_LD + h * HD;
  const bf16_t* dob = dout + (size_t)b * ld_rows * 768 + h * HD;
  const float* Lb = lse + ((size_t)b * NH + h) * S;
  const float* Db = dsum + ((size_t)b * NH + h) * S;
  const int kr = (key0 + ki) < S ? (key0 + ki) : S - 1;
  const uint32_t hk = DROP ? (vc_drop_stream(vc_salted(drop_seed, drop_salt), (uint32_t)b, (uint32_t)h) ^ (uint32_t)(key0 + ki)) : 0u;

  bf16x8 kf[4], vf[4];     // B operands: K^T and V^T columns of this lane's key
  {
    const bf16_t* kp = base + (size_t)kr * QKV_LD + 768 + half * 8;
    const bf16_t* vp = base + (size_t)kr * QKV_LD + 1536 + half * 8;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      kf[ds] = *(const bf16x8*)(kp + ds * 16);
      vf[ds] = *(const bf16x8*)(vp + ds * 16);
    }
  }
  const int nfull = S / KT, rem = S - nfull * KT;
  const bool tail_tile = rem > 8 || (causal_from > 0 && rem > 0);
  const int ntiles = nfull + (tail_tile ? 1 : 0);
  const int t_lo = q_lo / KT;
  int t_hi = (q_hi + KT - 1) / KT;
  t_hi = t_hi < ntiles ? t_hi : ntiles;
  const bool wave_causal = causal_from > 0 && key0 + 31 >= causal_from;

  // ---- staging: wave w moves rows [8w, 8w+8) and [32+8w, 32+8w+8) of the Q and of the dO tile (4 pieces of 1 KiB) and 64 floats:
  // wave 0 the tile's L, wave 1 its D, waves 2 / 3 the same into the spare 512 B (every wave counts 5 requests per tile)
  const int srow = w * 8 + (lane >> 3);
  const uint32_t s_chunk = (uint32_t)(((lane & 7) ^ kv_swz(srow)) * 16);
  const char* qg = (const char*)base;
  const char* dg = (const char*)dob;
  const float* ldsrc = (w & 1) ? Db : Lb;
#define STAGE_TILE(t_, stg_)                                                                       \
  do {                                                                                             \
    char* st_ = smem + (stg_) * DKV_STG;                                                           \
    char* sb_ = st_ + w * 1024;                                                                    \
    int r0_ = (t_) * KT + srow, r1_ = r0_ + 32;                                                    \
    r0_ = r0_ < S ? r0_ : S - 1;                                                                   \
    r1_ = r1_ < S ? r1_ : S - 1;                                                                   \
    glds16(qg + (size_t)r0_ * (QKV_LD * 2) + s_chunk, sb_);                                        \
    glds16(qg + (size_t)r1_ * (QKV_LD * 2) + s_chunk, sb_ + 4096);                                 \
    glds16(dg + (size_t)r0_ * (768 * 2) + s_chunk, sb_ + DT_B);                                    \
    glds16(dg + (size_t)r1_ * (768 * 2) + s_chunk, sb_ + DT_B + 4096);                             \
    int i_ = (t_) * KT + lane;                                                                     \
    i_ = i_ < S ? i_ : S - 1;                                                                      \
    glds4(ldsrc + i_, st_ + 2 * DT_B + w * 256);                                                   \
  } while (0)
  VC_BWD_READ_OFFSETS();

  f32x16 dvt[2], dkt[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    dvt[0][r] = 0.f; dvt[1][r] = 0.f; dkt[0][r] = 0.f; dkt[1][r] = 0.f;
  }
  if (t_hi > t_lo) STAGE_TILE(t_lo, 0);
  if (t_lo + 1 < t_hi) STAGE_TILE(t_lo + 1, 1);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)

#define DKV_ELEM(QT_, t_, MASKED_)                                                                                  \
  do {                                                                                                               \
    /* lane holds, for its key, queries q = QT_*32 + 8g + 4*half + e  (r = 4g + e) */                                \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                  \
      const f32x4 L4 = *(const f32x4*)(Ll + (QT_) * 32 + g * 8 + 4 * half);                                          \
      const f32x4 D4 = *(const f32x4*)(Ll + 64 + (QT_) * 32 + g * 8 + 4 * half);                                     \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                                \
        const int r = g * 4 + e;                                                                                     \
        const int q = (t_) * KT + (QT_) * 32 + g * 8 + 4 * half + e;                                                 \
        float p = fast_exp2(fmaf(st[r], c_log2, -L4[e]));                                                            \
        if (MASKED_) p = ((q < S) & visible_nb(q, key0 + ki, S, causal_from, mask_from)) ? p : 0.f;                  \
        float dp = dpt[r], pd = p;                                                                                   \
        if (DROP) {                                                                                                  \
          const bool keep = vc_lowbias32(hk ^ ((uint32_t)q << 10)) >= drop_thr;                                      \
          dp = keep ? dp * drop_scale : 0.f;                                                                         \
          pd = keep ? p : 0.f;                                                                                       \
        }                                                                                                            \
        st[r] = pd;                          /* (dropped) P[q][key] */                                               \
        dpt[r] = p * (dp - D4[e]);           /* dS[q][key] */                                                        \
      }                                                                                                              \
    }                                                                                                                \
  } while (0)
  // one 32-query half of a tile: S = Q K^T, dP = dO V^T, P / dS, dV^T += dO^T P, dK^T += Q^T dS
#define DKV_HALF(QT_, t_)                                                                                   \
  do {                                                                                                               \
    bf16x8 qfr[4], dfr[4];                                                                                           \
    if (VC_BWD_PREFETCH && (QT_) == 1) {                                                                             \
      _Pragma("unroll") for (int ds = 0; ds < 4; ++ds) { qfr[ds] = qnx[ds]; dfr[ds] = dnx[ds]; }                     \
    } else {                                                                                                         \
      _Pragma("unroll") for (int ds = 0; ds < 4; ++ds) {                                                             \
        qfr[ds] = *(const bf16x8*)(ql + (QT_) * 4096 + koff[ds]);                                                    \
        dfr[ds] = *(const bf16x8*)(ql + DT_B + (QT_) * 4096 + koff[ds]);                                             \
      }                                                                                                              \
    }                                                                                                                \
    VC_BWD_PRIO_ON();                                                                                                \
    f32x16 st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[0], kf[0], VC_ZERO16, 0, 0, 0);                          \
    f32x16 dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dfr[0], vf[0], VC_ZERO16, 0, 0, 0);                         \
    _Pragma("unroll") for (int ds = 1; ds < 4; ++ds) {                                                               \
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[ds], kf[ds], st, 0, 0, 0);                                    \
      dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dfr[ds], vf[ds], dpt, 0, 0, 0);                                  \
    }                                                                                                                \
    VC_BWD_PRIO_OFF();                                                                                               \
    /* the transposed Q / dO fragments of this half go out now: their latency hides behind the elementwise part */  \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    s16x4 qtr[2][2], dtr[2][2];          /* [dt][rd] of one 16-query block at a time */                              \
    _Pragma("unroll") for (int dt = 0; dt < 2; ++dt)                                                                 \
      _Pragma("unroll") for (int rd = 0; rd < 2; ++rd) {                                                             \
        dtr[dt][rd] = lds_tr_read<DT_B + (QT_) * 4096>(qb_ + (uint32_t)toff[rd][dt]);                                \
        qtr[dt][rd] = lds_tr_read<(QT_) * 4096>(qb_ + (uint32_t)toff[rd][dt]);                                       \
      }                                                                                                              \
    if (VC_BWD_PREFETCH && (QT_) == 0) {      /* the second half's row-major fragments: in flight during this half's elementwise part */ \
      _Pragma("unroll") for (int ds = 0; ds < 4; ++ds) {                                                             \
        qnx[ds] = *(const bf16x8*)(ql + 4096 + koff[ds]);                                                            \
        dnx[ds] = *(const bf16x8*)(ql + DT_B + 4096 + koff[ds]);                                                     \
      }                                                                                                              \
      __builtin_amdgcn_sched_barrier(0);                                                                             \
    }                                                                                                                \
    if (masked_) DKV_ELEM(QT_, t_, true);         /* wave-uniform: only the elementwise part exists twice */          \
    else DKV_ELEM(QT_, t_, false);                                                                                   \
    VC_BWD_PRIO_ON();                                                                                                \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                               \
      bf16x8 pf, sf;                                                                                                 \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                \
        pf[j] = (__bf16)st[ks * 8 + j];                                                                              \
        sf[j] = (__bf16)dpt[ks * 8 + j];                                                                             \
      }                                                                                                              \
      asm volatile("s_waitcnt lgkmcnt(0)"                                                                            \
                   : "+v"(qtr[0][0]), "+v"(qtr[0][1]), "+v"(qtr[1][0]), "+v"(qtr[1][1]),                             \
                     "+v"(dtr[0][0]), "+v"(dtr[0][1]), "+v"(dtr[1][0]), "+v"(dtr[1][1]));                            \
      bf16x8 da[2], qa[2];                                                                                           \
      _Pragma("unroll") for (int dt = 0; dt < 2; ++dt) {                                                             \
        da[dt] = tr_pair(dtr[dt][0], dtr[dt][1]);                                                                    \
        qa[dt] = tr_pair(qtr[dt][0], qtr[dt][1]);                                                                    \
      }                                                                                                              \
      if (ks == 0) {       /* the second block's fragments fly behind the first block's MFMAs */                     \
        _Pragma("unroll") for (int dt = 0; dt < 2; ++dt)                                                             \
          _Pragma("unroll") for (int rd = 0; rd < 2; ++rd) {                                                         \
            dtr[dt][rd] = lds_tr_read<DT_B + (QT_) * 4096 + 2048>(qb_ + (uint32_t)toff[rd][dt]);                     \
            qtr[dt][rd] = lds_tr_read<(QT_) * 4096 + 2048>(qb_ + (uint32_t)toff[rd][dt]);                            \
          }                                                                                                          \
      }                                                                                                              \
      _Pragma("unroll") for (int dt = 0; dt < 2; ++dt) {                                                             \
        dvt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[dt], pf, dvt[dt], 0, 0, 0);                             \
        dkt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[dt], sf, dkt[dt], 0, 0, 0);                             \
      }                                                                                                              \
    }                                                                                                                \
    VC_BWD_PRIO_OFF();                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  } while (0)
#define DKV_TILE(stg_, t_)                                                                                  \
  do {                                                                                                               \
    const char* ql = smem + (stg_) * DKV_STG;                                                                        \
    const float* Ll = (const float*)(ql + 2 * DT_B);                                                                 \
    const uint32_t qb_ = lds_addr(ql);                                                                               \
    bf16x8 qnx[4], dnx[4];                                                                                           \
    DKV_HALF(0, t_);                                                                                                 \
    DKV_HALF(1, t_);                                                                                                 \
  } while (0)

  int stg = 0;
  for (int t = t_lo; t < t_hi; ++t) {
    if (t + 1 < t_hi) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + 2 < t_hi) STAGE_TILE(t + 2, stg == 0 ? 2 : stg - 1);
    if (active) {
      const bool masked_ = t >= nfull || wave_causal;
      DKV_TILE(stg, t);
    }
    stg = stg == 2 ? 0 : stg + 1;
  }
#undef DKV_TILE
#undef DKV_HALF
#undef DKV_ELEM
#undef STAGE_TILE
  // left-over queries on the vector ALU
  if (active && !tail_tile) {
    for (int q = (nfull * KT > q_lo ? nfull * KT : q_lo); q < (S < q_hi ? S : q_hi); ++q) {
      const bf16_t* qr = base + (size_t)q * QKV_LD + half * 8;
      const bf16_t* dr = dob + (size_t)q * 768 + half * 8;
      float sp = 0.f, dp = 0.f;
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) {
        const bf16x8 qv = *(const bf16x8*)(qr + ds * 16);
        const bf16x8 dv = *(const bf16x8*)(dr + ds * 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          sp += (float)qv[j] * (float)kf[ds][j];
          dp += (float)dv[j] * (float)vf[ds][j];
        }
      }
      sp += __shfl_xor(sp, 32, 64);
      dp += __shfl_xor(dp, 32, 64);
      const float p = fast_exp2(fmaf(sp, c_log2, -Lb[q]));
      float pb = (float)(__bf16)p;
      if (DROP) {
        const bool keep = vc_lowbias32(hk ^ ((uint32_t)q << 10)) >= drop_thr;
        dp = keep ? dp * drop_scale : 0.f;
        pb = keep ? pb : 0.f;
      }
      const float dsb = (float)(__bf16)(p * (dp - Db[q]));
      const bf16_t* qo = base + (size_t)q * QKV_LD + 4 * half;
      const bf16_t* dd = dob + (size_t)q * 768 + 4 * half;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const bf16x4 qq = *(const bf16x4*)(qo + dt * 32 + g * 8);
          const bf16x4 dq = *(const bf16x4*)(dd + dt * 32 + g * 8);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dvt[dt][g * 4 + e] = fmaf(pb, (float)dq[e], dvt[dt][g * 4 + e]);
            dkt[dt][g * 4 + e] = fmaf(dsb, (float)qq[e], dkt[dt][g * 4 + e]);
          }
        }
    }
  }
  const int key = key0 + ki;
  if (key < S) {
    bf16_t* ok = dqkv + ((size_t)b * ld_rows + key) * QKV_LD + 768 + h * HD + 4 * half;
    bf16_t* ov = ok + 768;
    const bf16_t* ex = extra ? extra + ((size_t)b * ld_rows + key) * 1536 + h * HD + 4 * half : nullptr;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float kx[4] = {0.f, 0.f, 0.f, 0.f}, vx[4] = {0.f, 0.f, 0.f, 0.f};
        if (ex) {
          const bf16x4 ek = *(const bf16x4*)(ex + dt * 32 + g * 8);
          const bf16x4 ev = *(const bf16x4*)(ex + 768 + dt * 32 + g * 8);
#pragma unroll
          for (int e = 0; e < 4; ++e) { kx[e] = (float)ek[e]; vx[e] = (float)ev[e]; }
        }
        uint2 o1, o2;
        o1.x = pack2bf(dkt[dt][g * 4 + 0] * scale + kx[0], dkt[dt][g * 4 + 1] * scale + kx[1]);
        o1.y = pack2bf(dkt[dt][g * 4 + 2] * scale + kx[2], dkt[dt][g * 4 + 3] * scale + kx[3]);
        const float vs = DROP ? drop_scale : 1.0f;
        o2.x = pack2bf(dvt[dt][g * 4 + 0] * vs + vx[0], dvt[dt][g * 4 + 1] * vs + vx[1]);
        o2.y = pack2bf(dvt[dt][g * 4 + 2] * vs + vx[2], dvt[dt][g * 4 + 3] * vs + vx[3]);
        *(uint2*)(ok + dt * 32 + g * 8) = o1;
        *(uint2*)(ov + dt * 32 + g * 8) = o2;
      }
  }
}


}  // namespace

extern "C" int vitcap_attn_dense_bwd_rows(const void* qkv, const void* out, const void* dout, const float* lse, float* dsum,
                                          const void* extra_dkv, void* dqkv, int B, int S, int ld_rows, float scale,
                                          float p_drop, uint32_t drop_seed, int causal_from, int mask_from, int q_lo, int q_hi,
                                          void* stream) {
  VC_REQUIRE(qkv && out && dout && lse && dsum && dqkv && B > 0 && S > 0 && ld_rows >= S, "attn_dense_bwd: bad arguments");
  VC_REQUIRE(p_drop >= 0.f && p_drop < 1.f && ld_rows < 1024, "attn_dense_bwd: p_drop %g / ld_rows %d out of range",
             (double)p_drop, ld_rows);
  VC_REQUIRE(causal_from == 0 || (causal_from >= (S / KT) * KT && causal_from <= S),
             "attn_dense_bwd: causal_from %d must lie in the last key tile of S=%d", causal_from, S);
  VC_REQUIRE(q_lo >= 0 && q_lo < q_hi && q_hi <= S && (q_lo & 127) == 0,
             "attn_dense_bwd: query range [%d, %d) must start on a multiple of 128 inside S=%d", q_lo, q_hi, S);
  const float c = scale * 1.4426950408889634f;
  dim3 grid(((S + 127) / 128) * NH * B);                 // dK/dV: every key block
  dim3 grid_q(((q_hi - q_lo + 127) / 128) * NH * B);     // dQ: the query blocks of the range
  const uint32_t thr = (uint32_t)((double)p_drop * 4294967296.0);
  const float rs = 1.0f / (1.0f - p_drop);
  const int rem_ = S % KT;
  const bool tail_tile_ = rem_ > 8 || (causal_from > 0 && rem_ > 0);      // the kernels' own rule: masked tail tile or left-over keys
#define VC_BWD_LAUNCH_DMA(DROP_)                                                                                          \
  do {                                                                                                                    \
    if (tail_tile_)                                                                                                       \
      hipLaunchKernelGGL((attn_bwd_dq_dma_kernel<DROP_, true>), grid_q, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, \
                         (const bf16_t*)out, (const bf16_t*)dout, lse, dsum, (bf16_t*)dqkv, S, B, ld_rows, c, scale,      \
                         drop_seed, thr, rs, causal_from, mask_from, q_lo, q_hi, vc_tls_drop_salt);                      \
    else                                                                                                                  \
      hipLaunchKernelGGL((attn_bwd_dq_dma_kernel<DROP_, false>), grid_q, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, \
                         (const bf16_t*)out, (const bf16_t*)dout, lse, dsum, (bf16_t*)dqkv, S, B, ld_rows, c, scale,      \
                         drop_seed, thr, rs, causal_from, mask_from, q_lo, q_hi, vc_tls_drop_salt);                      \
    VC_LAUNCH_CHECK("attn_bwd_dq_dma");                                                                                   \
    hipLaunchKernelGGL(attn_bwd_dkv_dma_kernel<DROP_>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv,       \
                       (const bf16_t*)dout, lse, (const float*)dsum, (const bf16_t*)extra_dkv, (bf16_t*)dqkv, S, B,       \
                       ld_rows, c, scale, drop_seed, thr, rs, causal_from, mask_from, q_lo, q_hi, vc_tls_drop_salt);     \
    VC_LAUNCH_CHECK("attn_bwd_dkv_dma");                                                                                  \
  } while (0)
  if (p_drop > 0.f) VC_BWD_LAUNCH_DMA(true);
  else VC_BWD_LAUNCH_DMA(false);
#undef VC_BWD_LAUNCH_DMA
  return VITCAP_OK;
}

extern "C" int vitcap_attn_dense_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* dsum,
                                     const void* extra_dkv, void* dqkv, int B, int S, int ld_rows, float scale,
                                     float p_drop, uint32_t drop_seed, int causal_from, int mask_from, void* stream) {
  return vitcap_attn_dense_bwd_rows(qkv, out, dout, lse, dsum, extra_dkv, dqkv, B, S, ld_rows, scale, p_drop, drop_seed,
                                    causal_from, mask_from, 0, S, stream);
}
