// Attention kernels for gfx950.
//
// (1) attn_dense: unmasked multi-head self-attention over packed qkv (encoder blocks S=577, decoder
//     visual prefill S=578), flash style, MFMA 32x32x16 bf16.
//     Workgroup = 4 waves = 128 query rows of one (image, head); each wave owns 32 query rows.
//     Per 64-key tile:  S^T = K . Q^T   (A operand = K rows from LDS, B operand = Q^T held in VGPRs)
//       -> every lane holds, for ONE query (lane & 31), 32 of the tile's 64 scores: softmax state
//          (running max, sum, the O^T accumulator columns) is lane-local, the two half-waves only
//          exchange their row max (one shuffle per tile).
//     O^T += V^T . P^T          (A operand = V^T from LDS, B operand = P^T straight from the score
//          registers: the MFMA C-layout of S^T *is* the B-operand layout of the next MFMA once the
//          keys of each 16-key block are stored in the order [0-3, 8-11, 4-7, 12-15] in V^T).
//     V is transposed on its way into LDS (4 keys x 4 d per thread, ds_write_b64).
//     Softmax runs in the log2 domain with the running max rounded UP to an integer: every rescale
//     factor is an exact power of two, so bf16(P) does not depend on the tile order (this is what
//     makes the bf16 path reproducible by the CPU oracle's rounding emulation).
// (2) attn_decode_step: 2 query rows per sequence against the cached visual K/V (read in place from
//     the prefill's packed qkv buffer) plus the text K/V cache; HBM-bound, 8 lanes per key row.
#include <stdlib.h>

#include "common.h"
#include "rng.h"

namespace {

constexpr int HD = 64;          // head dim
constexpr int NH = 12;          // heads
constexpr int QKV_LD = 2304;    // packed row: [q | k | v] x [head][64]
constexpr int KT = 64;          // keys per tile

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

typedef __attribute__((ext_vector_type(4))) int attn_i32x4;
// one LDS-DMA piece through a buffer descriptor: 64 lanes x 16 B -> 1 KiB at the wave-uniform LDS byte address `lds`; global address =
// descriptor base + per-lane voff + scalar soff.  Against glds16 (64-bit per-lane addresses) the per-tile address arithmetic moves from
// 8 v_lshl_add_u64 + 2 v_mad_i64_i32 per key tile to scalar adds (round 6: the loop is bound by vector-ALU issue and by the board's power).
__device__ __forceinline__ void bufdma16(uint32_t lds, uint32_t voff, const attn_i32x4& rsrc, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

__device__ __forceinline__ void glds16(const void* g, void* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
// 16-byte chunk c of LDS row r (128 B = one key's 64 head dims) is stored at chunk c ^ kv_swz(r): the 16-lane groups of the
// K fragment reads (ds_read_b128: 16 different rows, one chunk) and the 32-lane groups of the V transpose reads
// (ds_read_b64_tr_b16: 4 consecutive rows x 64 B) then touch every bank once (MI355X_MICROARCH.md, LDS).
__device__ __forceinline__ int kv_swz(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

// build-time knobs of the dense kernel (tools/attn_variants.sh measures them; the defaults are the shipped form)
#ifndef VC_ATTN_NSTG
#define VC_ATTN_NSTG 3
#endif
#ifndef VC_ATTN_MINW
#define VC_ATTN_MINW 3
#endif
#ifndef VC_ATTN_ROWSUM_MFMA
#define VC_ATTN_ROWSUM_MFMA 1
#endif
#ifndef VC_ATTN_VEARLY
#define VC_ATTN_VEARLY 1
#endif
#ifndef VC_ATTN_ABL           // timing ablations (tools/attn_variants.sh; results are WRONG when != 0): 1 no QK^T MFMAs, 2 no exp2 / convert
#define VC_ATTN_ABL 0         // arithmetic, 4 no P.V / row-sum MFMAs, 8 no per-tile barrier, 16 no LDS-DMA after the first two tiles, 32 no V reads
#endif
constexpr int NSTG = VC_ATTN_NSTG;       // K/V tile ring: tile t+NSTG-1 is in flight while tile t is multiplied
constexpr int TILE_B = KT * 128;         // one K (or V) tile: 64 keys x 128 B
constexpr int STG_B = 2 * TILE_B;        // [K | V]

// DROP: attention dropout of the decoder in training -- the probabilities that feed P.V are zeroed where
// vc_drop_keep() says so and the output is scaled by 1/(1-p); the softmax statistics (and lse) are those of the
// undropped row.
//
// Data path: K and V tiles go HBM/L2 -> LDS by LDS-DMA (global_load_lds, 16 B per lane, no VGPR round trip, no ds_write),
// both ROW-major (key x 64 dims) with XOR-swizzled 16-byte chunks; S^T = K . Q^T reads K rows with ds_read_b128, and the
// A operand of O^T += V^T . P^T is read TRANSPOSED straight from the row-major V tile (ds_read_b64_tr_b16: a 16-lane
// group fetches 4 keys x 16 dims and lane i receives dim i of the 4 keys).  The C layout of S^T holds, per lane, keys
// (r&3) + 8(r>>2) + 4 half of each 32-key block: exactly two runs of 4 consecutive keys per 16-key MFMA step, i.e. two
// transpose reads -- P^T goes from the score registers into the next MFMA without any cross-lane traffic.
// Row sums run through the matrix pipe as well: one more MFMA per 16-key block with an all-ones A operand accumulates
// sum_k bf16(P) for the lane's query in every register of `lacc` (the loop is bound by vector-ALU issue; the matrix pipe
// has room) -- so the normaliser is the sum of the ROUNDED probabilities the numerator uses.
template <bool DROP>
__global__ __launch_bounds__(256, VC_ATTN_MINW) void attn_dense_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                         float* __restrict__ lse, int S, int B, int ld_rows, float c_log2,
                                                         uint32_t drop_seed, uint32_t drop_thr, float drop_scale,
                                                         int causal_from, int mask_from, int q_lo, int q_rows, int rev,
                                                         const uint32_t* __restrict__ drop_salt) {
  __shared__ __attribute__((aligned(1024))) char smem[NSTG * STG_B];   // [stage][K | V]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 31, half = lane >> 5;
  // XCD-aware work mapping (1-D grid): workgroups are dealt round-robin to the 8 XCDs, so the q-blocks of one
  // (image, head) -- which share that head's K/V -- are made consecutive *within* an XCD and hit its L2 instead of
  // fetching K/V once per XCD (measured: 3.7x algorithmic fetch with the naive 3-D grid).
  // only the query blocks covering rows [q_lo, q_lo + q_rows) are computed (q_lo a multiple of 128; all S keys still count)
  const int nqb = (q_rows + 127) / 128;
  const int nwork = nqb * NH * B;
  int wid = blockIdx.x;
  {
    const int qd = nwork >> 3, rm = nwork & 7, xcd = wid & 7;
    wid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (wid >> 3);
  }
  if (rev) wid = nwork - 1 - wid;      // images last-to-first (common.h: vc_tls_walk_rev)
  const int qb = (q_lo >> 7) + wid % nqb;
  const int h = (wid / nqb) % NH, b = wid / (nqb * NH);
  const int q0 = qb * 128 + w * 32;
  const bool active = q0 < S;        // waves past the last query row only help with staging
  const uint32_t hq = DROP ? (vc_drop_stream(vc_salted(drop_seed, drop_salt), (uint32_t)b, (uint32_t)h) ^ ((uint32_t)(q0 + qi) << 10) ^ (4u * half)) : 0u;
  const bf16_t* base = qkv + (size_t)b * ld_rows * QKV_LD + h * HD;   // ld_rows >= S rows per image in the buffers

  // Q^T fragments: lane holds Q[q0+qi][ds*16 + half*8 .. +7]
  bf16x8 qf[4];
  {
    int qr = q0 + qi;
    qr = qr < S ? qr : S - 1;
    const bf16_t* qp = base + (size_t)qr * QKV_LD + half * 8;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) qf[ds] = *(const bf16x8*)(qp + ds * 16);
  }

  const int nfull = S / KT;                                     // tiles with 64 valid keys
  const int rem = S - nfull * KT;                               // 0..63 left-over keys
  // causal_from > 0 (decoder in training, rows [visual | caption]): keys >= causal_from are caption rows, visible only
  // to caption queries q >= key; they all sit in the masked tail tile (host checks causal_from >= nfull * KT)
  const bool tail_tile = rem > 8 || (causal_from > 0 && rem > 0);   // many left-overs: one masked MFMA tile
  const int ntiles = nfull + (tail_tile ? 1 : 0);

  // ---- staging: each wave moves rows [8w, 8w+8) and [32+8w, 32+8w+8) of the K and of the V tile (4 LDS-DMA pieces of 1 KiB);
  // lane l lands at LDS byte 16 l of the piece = row l>>3, physical chunk l&7, so it FETCHES logical chunk (l&7) ^ swz(row)
  const int srow = w * 8 + (lane >> 3);                          // (row + 32 has the same swizzle)
  const uint32_t s_off = (uint32_t)(srow * QKV_LD + 768) * 2u + (uint32_t)(((lane & 7) ^ kv_swz(srow)) * 16);
  const char* gbase = (const char*)base;
  // descriptor over this (image, head)'s rows (unbounded range: full tiles never leave the image's rows)
  const uint64_t gb64 = (uint64_t)gbase;
  const attn_i32x4 kv_rsrc = attn_i32x4{__builtin_amdgcn_readfirstlane((int)(uint32_t)gb64), __builtin_amdgcn_readfirstlane((int)(uint32_t)(gb64 >> 32)),
                                        -1, 0x00020000};
#define STAGE_TILE(t_, stg_)                                                                                    \
  do {                                                                                                          \
    char* sb_ = smem + (stg_) * STG_B + w * 1024;                                                               \
    if ((t_) < nfull) {                                                                                         \
      const uint32_t so_ = (uint32_t)(t_) * (uint32_t)(KT * QKV_LD * 2), sl_ = lds_addr(sb_);                   \
      bufdma16(sl_, s_off, kv_rsrc, so_);                                                                       \
      bufdma16(sl_ + 4096, s_off, kv_rsrc, so_ + 32 * QKV_LD * 2);                                              \
      bufdma16(sl_ + TILE_B, s_off, kv_rsrc, so_ + 768 * 2);                                                    \
      bufdma16(sl_ + TILE_B + 4096, s_off, kv_rsrc, so_ + 768 * 2 + 32 * QKV_LD * 2);                           \
    } else { /* tail tile: rows past the sequence re-read its last row (their scores are masked) */            \
      const int r0_ = (t_) * KT + srow, r1_ = r0_ + 32;                                                         \
      const uint32_t c_ = (uint32_t)(768 * 2 + ((lane & 7) ^ kv_swz(srow)) * 16);                               \
      const char* a0_ = gbase + (size_t)(r0_ < S ? r0_ : S - 1) * (QKV_LD * 2) + c_;                            \
      const char* a1_ = gbase + (size_t)(r1_ < S ? r1_ : S - 1) * (QKV_LD * 2) + c_;                            \
      glds16(a0_, sb_);                                                                                         \
      glds16(a1_, sb_ + 4096);                                                                                  \
      glds16(a0_ + 768 * 2, sb_ + TILE_B);                                                                      \
      glds16(a1_ + 768 * 2, sb_ + TILE_B + 4096);                                                               \
    }                                                                                                           \
  } while (0)

  // fragment read offsets inside a stage (per lane, fixed): K row qi (+32 kt), chunk (2 ds + half) ^ swz(qi);
  // V transpose read rd of a 16-key block: lane supplies row 8 rd + 4 half + j (j = (lane & 15) >> 2), 8 bytes at dims
  // 32 dt + 16 ((lane >> 4) & 1) + 4 (lane & 3)
  int koff[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) koff[ds] = qi * 128 + (((2 * ds + half) ^ kv_swz(qi)) * 16);
  int voff[2][2];
  {
    const int j = (lane & 15) >> 2;
    const int c2 = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int r = 8 * rd + 4 * half + j;                       // row within the 16-key block (block base is a multiple of 16)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        voff[rd][dt] = TILE_B + r * 128 + (((c2 + 4 * dt) ^ kv_swz(r)) * 16) + (lane & 1) * 8;
    }
  }

  f32x16 ot[2], lacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    ot[0][r] = 0.f;
    ot[1][r] = 0.f;
    lacc[r] = 0.f;
  }
  float m_i = -1e30f;   // running max, log2 domain, integer valued once set
  float l_i = 0.f;      // left-over keys (vector ALU path): this half-wave's partial row sum
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

  // ---- left-over keys (S = 577/578 leaves 1/2) FIRST: one online-softmax step per key on the vector ALU (the online softmax
  // does not care about key order).  The lane owns 32 of the 64 q dims (its partner lane^32 the rest) and 32 of the 64 output
  // dims.  The first left-over key's K/V rows are requested BEFORE the first two tiles' LDS-DMA (loads retire in order: behind
  // the tiles they would wait for 32 KB to land) and consumed after, so tiles 0 and 1 fly while the key is processed.
  const int key0 = nfull * KT;
  const bool left = active && !tail_tile && key0 < S;
  bf16x8 lk[4];
  bf16x4 lv[2][4];
  if (left) {
    const bf16_t* kr = base + (size_t)key0 * QKV_LD + 768 + half * 8;
    const bf16_t* vr = base + (size_t)key0 * QKV_LD + 1536 + 4 * half;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) lk[ds] = *(const bf16x8*)(kr + ds * 16);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) lv[dt][g] = *(const bf16x4*)(vr + dt * 32 + g * 8);
  }
  if (ntiles > 0) STAGE_TILE(0, 0);
  if (NSTG > 2 && ntiles > 1) STAGE_TILE(1, 1);
  if (left) {
    for (int key = key0; key < S; ++key) {
      if (key > key0) {
        const bf16_t* kr = base + (size_t)key * QKV_LD + 768 + half * 8;
        const bf16_t* vr = base + (size_t)key * QKV_LD + 1536 + 4 * half;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) lk[ds] = *(const bf16x8*)(kr + ds * 16);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int g = 0; g < 4; ++g) lv[dt][g] = *(const bf16x4*)(vr + dt * 32 + g * 8);
      }
      float sp = 0.f;
#pragma unroll
      for (int ds = 0; ds < 4; ++ds)
#pragma unroll
        for (int j = 0; j < 8; ++j) sp += (float)qf[ds][j] * (float)lk[ds][j];
      const float sc = sp + __shfl_xor(sp, 32, 64);
      const float m_new = fmaxf(m_i, ceilf(sc * c_log2));
      const float alpha = fast_exp2(m_i - m_new);       // 0 for the first key (m_i = -1e30)
      m_i = m_new;
      const float pv = fast_exp2(fmaf(sc, c_log2, -m_new));
      float pb = (float)(__bf16)pv;                      // the row sum counts the ROUNDED probability, as the MFMA path does
      l_i = l_i * alpha + (half == 0 ? pb : 0.f);
      if (DROP) pb = vc_lowbias32((hq ^ (4u * half)) ^ (uint32_t)key) >= drop_thr ? pb : 0.f;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) ot[dt][g * 4 + e] = fmaf(pb, (float)lv[dt][g][e], ot[dt][g * 4 + e] * alpha);
    }
  }
  // Every VGPR load so far (Q fragments, left-over rows) must be provably complete on ALL paths into the loop: otherwise the
  // compiler's waitcnt pass keeps those registers "possibly pending" across the loop and puts vmcnt(0) in front of the first
  // MFMAs of every iteration, i.e. it waits for the tile it has just requested (docs/LAB_r01_r04.md "waitcnt false dependency")
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)

  // One key-tile step; MASKED_ is a literal so the left-over masking exists only in the peeled tail instance
  // (inside the loop hipcc if-converted it into 97 extra VALU ops per tile).
#define TILE_COMPUTE(stg_, t_, MASKED_)                                                                         \
  do {                                                                                                          \
    const char* kl = smem + (stg_) * STG_B;                                                                     \
    f32x16 st[2];                                                                                               \
    if (VC_ATTN_ABL & 1) { _Pragma("unroll") for (int r = 0; r < 16; ++r) { st[0][r] = (float)(r + qi) * 1e-3f; st[1][r] = (float)(r - qi) * 1e-3f; } } else { \
    /* all eight K fragments are requested before the first MFMA (left to itself hipcc reads them two at a time, each pair     \
       behind its own lgkmcnt(0): four exposed LDS round trips per tile; measured anatomy in docs/LAB_r01_r04.md 4.2 iii) */          \
    bf16x8 kfr[2][4];                                                                                           \
    _Pragma("unroll") for (int kt = 0; kt < 2; ++kt)                                                            \
      _Pragma("unroll") for (int ds = 0; ds < 4; ++ds) kfr[kt][ds] = *(const bf16x8*)(kl + kt * 4096 + koff[ds]); \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
    _Pragma("unroll") for (int kt = 0; kt < 2; ++kt) {                                                          \
      st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[kt][0], qf[0],                                       \
                                                       f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0); \
      _Pragma("unroll") for (int ds = 1; ds < 4; ++ds)                                                          \
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[kt][ds], qf[ds], st[kt], 0, 0, 0);                 \
    }                                                                                                           \
    }                                                                                                           \
    /* the transpose reads of this tile's V fragments go out now: their latency hides behind the softmax       \
       (VC_ATTN_VEARLY; otherwise one 16-key block ahead of its MFMAs, 16 fewer live registers) */              \
    s16x4 vt[VC_ATTN_VEARLY ? 4 : 2][2][2];                                                                     \
    const uint32_t vb_ = lds_addr(kl);                                                                          \
    if (VC_ATTN_ABL & 32) { _Pragma("unroll") for (int i_ = 0; i_ < (VC_ATTN_VEARLY ? 4 : 2); ++i_) _Pragma("unroll") for (int dt = 0; dt < 2; ++dt) _Pragma("unroll") for (int rd = 0; rd < 2; ++rd) vt[i_][dt][rd] = s16x4{(short)(0x3c00 + i_), (short)0x3c00, (short)(0x3c00 + dt), (short)(0x3c00 + rd)}; } else \
    if (VC_ATTN_VEARLY) {                                                                                       \
      _Pragma("unroll") for (int dt = 0; dt < 2; ++dt)                                                          \
        _Pragma("unroll") for (int rd = 0; rd < 2; ++rd) {                                                      \
          const uint32_t a_ = vb_ + (uint32_t)voff[rd][dt];                                                     \
          vt[0][dt][rd] = lds_tr_read<0>(a_);                                                                   \
          vt[1][dt][rd] = lds_tr_read<2048>(a_);                                                                \
          vt[VC_ATTN_VEARLY ? 2 : 0][dt][rd] = lds_tr_read<4096>(a_);                                           \
          vt[VC_ATTN_VEARLY ? 3 : 1][dt][rd] = lds_tr_read<6144>(a_);                                           \
        }                                                                                                       \
    }                                                                                                           \
    if (MASKED_) {                                                                                              \
      const int kv0 = (t_) * KT;                                                                                \
      _Pragma("unroll") for (int kt = 0; kt < 2; ++kt)                                                          \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                        \
          const int key = kv0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;                                    \
          const bool vis_ = joint_visible(q0 + qi, key, S, causal_from, mask_from);                             \
          st[kt][r] = vis_ ? st[kt][r] : -INFINITY;                                                             \
        }                                                                                                       \
    }                                                                                                           \
    /* running max on the raw scores (scale > 0), integer ceiling in the log2 domain */                         \
    /* 32 scores -> 16 v_max3 / v_max in a tree (max is exact: any order gives the same value) */               \
    float ma_[5], mb_[5];                                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < 5; ++i_) {                                                          \
      ma_[i_] = max3f(st[0][3 * i_], st[0][3 * i_ + 1], st[0][3 * i_ + 2]);                                     \
      mb_[i_] = max3f(st[1][3 * i_], st[1][3 * i_ + 1], st[1][3 * i_ + 2]);                                     \
    }                                                                                                           \
    const float mc0_ = max3f(ma_[0], ma_[1], ma_[2]), mc1_ = max3f(ma_[3], ma_[4], st[0][15]);                   \
    const float mc2_ = max3f(mb_[0], mb_[1], mb_[2]), mc3_ = max3f(mb_[3], mb_[4], st[1][15]);                   \
    float mx = fmaxf(max3f(mc0_, mc1_, mc2_), mc3_);                                                            \
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));                                                                     \
    const float m_new = fmaxf(m_i, ceilf(mx * c_log2));                                                         \
    if (__builtin_amdgcn_ballot_w64(m_new != m_i) != 0) { /* rare after the first tiles: integer steps */       \
      const float alpha = fast_exp2(m_i - m_new);         /* exact power of two (0 on the first tile) */        \
      l_i *= alpha;                                                                                             \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                          \
        ot[0][r] *= alpha;                                                                                      \
        ot[1][r] *= alpha;                                                                                      \
        lacc[r] *= alpha;                                                                                       \
      }                                                                                                         \
      m_i = m_new;                                                                                              \
    }                                                                                                           \
    /* P = exp2(s*c - m) with one fma per score */                                                              \
    const float nm = -m_i;                                                                                      \
    _Pragma("unroll") for (int kt = 0; kt < 2; ++kt)                                                            \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) st[kt][r] = (VC_ATTN_ABL & 2) ? st[kt][r] + nm : fast_exp2(fmaf(st[kt][r], c_log2, nm)); \
    /* O^T += V^T . P^T over the four 16-key blocks; row sums: ones . P^T */                                    \
    if (VC_ATTN_VEARLY && !(VC_ATTN_ABL & 32)) {                                                                \
      asm volatile("s_waitcnt lgkmcnt(0)"                                                                       \
                   : "+v"(vt[0][0][0]), "+v"(vt[0][0][1]), "+v"(vt[0][1][0]), "+v"(vt[0][1][1]),                \
                     "+v"(vt[1][0][0]), "+v"(vt[1][0][1]), "+v"(vt[1][1][0]), "+v"(vt[1][1][1]),                \
                     "+v"(vt[VC_ATTN_VEARLY ? 2 : 0][0][0]), "+v"(vt[VC_ATTN_VEARLY ? 2 : 0][0][1]),            \
                     "+v"(vt[VC_ATTN_VEARLY ? 2 : 0][1][0]), "+v"(vt[VC_ATTN_VEARLY ? 2 : 0][1][1]),            \
                     "+v"(vt[VC_ATTN_VEARLY ? 3 : 1][0][0]), "+v"(vt[VC_ATTN_VEARLY ? 3 : 1][0][1]),            \
                     "+v"(vt[VC_ATTN_VEARLY ? 3 : 1][1][0]), "+v"(vt[VC_ATTN_VEARLY ? 3 : 1][1][1]));           \
    } else {                                                                                                    \
      _Pragma("unroll") for (int dt = 0; dt < 2; ++dt)                                                          \
        _Pragma("unroll") for (int rd = 0; rd < 2; ++rd) vt[0][dt][rd] = lds_tr_read<0>(vb_ + (uint32_t)voff[rd][dt]); \
    }                                                                                                           \
    const uint32_t hx = hq ^ (uint32_t)((t_) * KT);                                                             \
    float psum = 0.f;                                                                                           \
    _Pragma("unroll") for (int kb = 0; kb < 4; ++kb) {                                                          \
      const int kt = kb >> 1, ks = kb & 1;                                                                      \
      const int vs = VC_ATTN_VEARLY ? kb : (kb & 1);                                                            \
      if (!VC_ATTN_VEARLY) {                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vt[vs][0][0]), "+v"(vt[vs][0][1]), "+v"(vt[vs][1][0]), "+v"(vt[vs][1][1])); \
        if (kb < 3) {                                                                                           \
          _Pragma("unroll") for (int dt = 0; dt < 2; ++dt)                                                      \
            _Pragma("unroll") for (int rd = 0; rd < 2; ++rd) {                                                  \
              const uint32_t a_ = vb_ + (uint32_t)voff[rd][dt];                                                 \
              vt[vs ^ 1][dt][rd] = kb == 0 ? lds_tr_read<2048>(a_) : (kb == 1 ? lds_tr_read<4096>(a_) : lds_tr_read<6144>(a_)); \
            }                                                                                                   \
        }                                                                                                       \
      }                                                                                                         \
      bf16x8 pf;                                                                                                \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) pf[j] = (__bf16)st[kt][ks * 8 + j];                         \
      if (VC_ATTN_ABL & 4) { ot[0][kb] += (float)pf[0]; ot[1][kb] += (float)pf[5]; } else {                      \
      if (VC_ATTN_ROWSUM_MFMA == 1) lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, lacc, 0, 0, 0);    \
      else if (VC_ATTN_ROWSUM_MFMA == 2) { _Pragma("unroll") for (int j = 0; j < 8; ++j) psum += st[kt][ks * 8 + j]; } /* unrounded P (timing) */ \
      else { _Pragma("unroll") for (int j = 0; j < 8; ++j) psum += (float)pf[j]; }                              \
      if (DROP) {                                                                                               \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                         \
          const int r = ks * 8 + j;                                                                             \
          const uint32_t kbits = (uint32_t)(kt * 32 + (r & 3) + 8 * (r >> 2));                                  \
          pf[j] = vc_lowbias32(hx ^ kbits) >= drop_thr ? pf[j] : (__bf16)0.0f;                                  \
        }                                                                                                       \
      }                                                                                                         \
      _Pragma("unroll") for (int dt = 0; dt < 2; ++dt) {                                                        \
        const s16x8 v8 = __builtin_shufflevector(vt[vs][dt][0], vt[vs][dt][1], 0, 1, 2, 3, 4, 5, 6, 7);         \
        ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v8), pf, ot[dt], 0, 0, 0);  \
      }                                                                                                         \
      }                                                                                                         \
    }                                                                                                           \
    if (VC_ATTN_ROWSUM_MFMA != 1) l_i += psum;                                                                      \
  } while (0)

  int stg = 0;                       // ring slot of tile t (wave-uniform)
  for (int t = 0; t < nfull; ++t) {
    // tile t landed (this wave's 4 pieces; with three slots tile t+1's 4 may stay in flight), then for every wave -- and every
    // wave is done with tile t-1, whose slot the next request is about to overwrite
    if (NSTG > 2 && t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(VC_ATTN_ABL & 8)) __syncthreads();
    if (NSTG > 2) {
      const int stg2 = stg == 0 ? 2 : stg - 1;                     // (stg + 2) % 3
      if (t + 2 < ntiles && !(VC_ATTN_ABL & 16)) STAGE_TILE(t + 2, stg2);
    } else {
      if (t + 1 < ntiles) STAGE_TILE(t + 1, stg ^ 1);
    }
    if (active) TILE_COMPUTE(stg, t, false);
    stg = NSTG > 2 ? (stg == 2 ? 0 : stg + 1) : (stg ^ 1);
  }
  if (tail_tile) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (active) TILE_COMPUTE(stg, nfull, true);
  }
#undef TILE_COMPUTE
#undef STAGE_TILE

  // ---- normalise and store: lane holds O[q][dt*32 + 8*g + 4*half + 0..3]; every register of lacc holds the MFMA part of
  // the row sum, l_i the left-over keys' part (one half-wave)
  const float l_tot = (VC_ATTN_ROWSUM_MFMA == 1 ? lacc[0] : 0.f) + l_i + __shfl_xor(l_i, 32, 64);
  const float inv = DROP ? drop_scale / l_tot : 1.0f / l_tot;
  const int q = q0 + qi;
  if (lse && q < S && half == 0) lse[((size_t)b * NH + h) * S + q] = m_i + log2f(l_tot);   // log2-domain logsumexp (training)
  if (q < S) {
    bf16_t* op = out + ((size_t)b * ld_rows + q) * 768 + h * HD + 4 * half;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 o;
        o.x = pack2bf(ot[dt][g * 4 + 0] * inv, ot[dt][g * 4 + 1] * inv);
        o.y = pack2bf(ot[dt][g * 4 + 2] * inv, ot[dt][g * 4 + 3] * inv);
        *(uint2*)(op + dt * 32 + g * 8) = o;
      }
  }
}

// ------------------------------------------------------------------------------------------------
// Decode step.  One workgroup (256 threads) per (sequence, head).  Keys: S_vis visual rows (packed
// qkv buffer of the prefill), text rows 0..t-2 from the cache, text row t-1 (this step's row 0) and,
// for query row 1 only, this step's row 1 ([MASK]).  8 lanes share one key row (16 B each).
// Two passes over the scores: they are staged in LDS (<= 640 keys x 2 rows fp32) so the softmax uses
// the exact row max like the oracle.
// ------------------------------------------------------------------------------------------------
constexpr int MAXKEYS = 704;     // 578 visual + 50 tag + 41 text + pad

__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16_t* __restrict__ qkv_step,
                                                          const bf16_t* __restrict__ vis_qkv,
                                                          bf16_t* __restrict__ text_kv, bf16_t* __restrict__ out,
                                                          int S_vis, int t, int max_len, int seq_per_image,
                                                          float c_log2, const int32_t* __restrict__ live,
                                                          const bf16_t* __restrict__ tag_a, const bf16_t* __restrict__ tag_b,
                                                          int n_tag, const int64_t* __restrict__ tag_len) {
  VC_LIVE_EXIT(live);
#ifdef VC_ATTN_DECODE_FAT     // measurement build (tools/coresident_probe.py): 80 registers, cannot be resident next to two GEMM waves per SIMD
  asm volatile("" ::: "v72");
#endif
  __shared__ float sc[2][MAXKEYS];
  __shared__ float red[2][4];
  __shared__ float oacc[4][2][HD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = blockIdx.x, b = blockIdx.y;
  const int img = b / seq_per_image;
  const int sub = lane & 7;          // which 8-wide d slice
  const int kslot = tid >> 3;        // 0..31: key slot within a 32-key sweep
  // Optional tag keys (the predicted tag tokens visible to the caption): n_tag rows per image between the
  // visual and the text keys, packed like the visual rows ([image][n_tag][2304]).  Which of the two embedding branches of
  // modeling_bert.py:1435-1489 the step sees is the reference's own data-dependent test, `topk_len[0] + 20 <= L`, with
  // L = t + 1 + 50 text slots at step t.
  const bf16_t* tag_kv = nullptr;
  if (n_tag > 0) tag_kv = ((int)tag_len[0] + 20 <= t + 51) ? tag_a : tag_b;
  const int S_pre = S_vis + n_tag;   // keys before the text rows
  const int nkeys = S_pre + t + 1;   // visual | tags | text 0..t-1 | mask row

  const bf16_t* q0p = qkv_step + ((size_t)b * 2) * QKV_LD + h * HD + sub * 8;
  const bf16_t* q1p = q0p + QKV_LD;
  float q0[8], q1[8];
  {
    const bf16x8 a = *(const bf16x8*)q0p;
    const bf16x8 c = *(const bf16x8*)q1p;
#pragma unroll
    for (int j = 0; j < 8; ++j) { q0[j] = (float)a[j]; q1[j] = (float)c[j]; }
  }
  bf16_t* tkv = text_kv + (size_t)b * max_len * 2 * 768;
  // publish this step's real-token K/V (row 0) into the cache at position t-1
  if (tid < 16) {
    const int which = tid >> 3;   // 0 = K, 1 = V
    const uint4 v = *(const uint4*)(qkv_step + ((size_t)b * 2) * QKV_LD + 768 * (1 + which) + h * HD + sub * 8);
    *(uint4*)(tkv + ((size_t)(t - 1) * 2 + which) * 768 + h * HD + sub * 8) = v;
  }
  auto key_ptr = [&](int k, int which) -> const bf16_t* {   // which: 0 K, 1 V
    if (k < S_vis) return vis_qkv + ((size_t)img * S_vis + k) * QKV_LD + 768 * (1 + which) + h * HD + sub * 8;
    if (k < S_pre) return tag_kv + ((size_t)img * n_tag + (k - S_vis)) * QKV_LD + 768 * (1 + which) + h * HD + sub * 8;
    const int tp = k - S_pre;
    if (tp < t - 1) return tkv + ((size_t)tp * 2 + which) * 768 + h * HD + sub * 8;
    return qkv_step + ((size_t)b * 2 + (tp - (t - 1))) * QKV_LD + 768 * (1 + which) + h * HD + sub * 8;
  };

  // ---- pass 1: scores.  Four keys per thread per sweep (4 x 16 B loads in flight; the kernel is HBM-latency bound)
  float mx0 = -1e30f, mx1 = -1e30f;
  for (int kb = kslot; kb < nkeys; kb += 128) {
    bf16x8 kv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = kb + u * 32;
      kv[u] = *(const bf16x8*)key_ptr(k < nkeys ? k : nkeys - 1, 0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = kb + u * 32;
      float d0 = 0.f, d1 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float kf = (float)kv[u][j];
        d0 += q0[j] * kf;
        d1 += q1[j] * kf;
      }
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) {
        d0 += __shfl_xor(d0, o, 64);
        d1 += __shfl_xor(d1, o, 64);
      }
      if (k == nkeys - 1) d0 = -INFINITY;   // row 0 (position t-1) cannot see the [MASK] row
      if (k < nkeys) {
        if (sub == 0) { sc[0][k] = d0; sc[1][k] = d1; }
        mx0 = fmaxf(mx0, d0);
        mx1 = fmaxf(mx1, d1);
      }
    }
  }
  mx0 = wave_max(mx0);
  mx1 = wave_max(mx1);
  if (lane == 0) { red[0][w] = mx0; red[1][w] = mx1; }
  __syncthreads();
  // scores are kept raw; P = exp2(fma(s, c, -m)) with m = ceil(max(s) * c), as in the dense kernel
  const float m0 = ceilf(fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3])) * c_log2);
  const float m1 = ceilf(fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3])) * c_log2);

  // ---- pass 2: P.V   (P rounded to bf16 for the product, row sum from the unrounded fp32 P)
  float o0[8], o1[8], l0 = 0.f, l1 = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) { o0[j] = 0.f; o1[j] = 0.f; }
  for (int kb = kslot; kb < nkeys; kb += 128) {
    bf16x8 vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = kb + u * 32;
      vv[u] = *(const bf16x8*)key_ptr(k < nkeys ? k : nkeys - 1, 1);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = kb + u * 32;
      if (k < nkeys) {
        const float p0 = fast_exp2(fmaf(sc[0][k], c_log2, -m0)), p1 = fast_exp2(fmaf(sc[1][k], c_log2, -m1));
        l0 += p0;
        l1 += p1;
        const float p0b = (float)(__bf16)p0, p1b = (float)(__bf16)p1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float vf = (float)vv[u][j];
          o0[j] += p0b * vf;
          o1[j] += p1b * vf;
        }
      }
    }
  }
  // reduce over the 8 key slots of the wave (lanes with equal `sub`), then over waves through LDS
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    l0 += __shfl_xor(l0, o, 64);
    l1 += __shfl_xor(l1, o, 64);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o0[j] += __shfl_xor(o0[j], o, 64);
      o1[j] += __shfl_xor(o1[j], o, 64);
    }
  }
  __syncthreads();
  if (lane < 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      oacc[w][0][lane * 8 + j] = o0[j];
      oacc[w][1][lane * 8 + j] = o1[j];
    }
  }
  if (lane == 0) { red[0][w] = l0; red[1][w] = l1; }
  __syncthreads();
  if (tid < 128) {
    const int r = tid >> 6, d = tid & 63;
    const float l = (red[r][0] + red[r][1]) + (red[r][2] + red[r][3]);
    const float o = (oacc[0][r][d] + oacc[1][r][d]) + (oacc[2][r][d] + oacc[3][r][d]);
    out[((size_t)b * 2 + r) * 768 + h * HD + d] = f2bf(o / l);
  }
}

// ------------------------------------------------------------------------------------------------
// Decode step for SEVERAL sequences per image (beam search): one workgroup per (image, head).
// At 256 images x 5 beams the per-sequence kernel above is vector-ALU bound (~1500 VALU instructions per thread and
// (sequence, head): 207 us per launch, half of the decode phase).  The 2*beams <= 16 query rows of an image are the 16 columns
// of one mfma_f32_16x16x32_bf16, and every sequence of an image sees the SAME 578 visual key rows:
//   scores   S[q][key]  = Q . K_vis^T        2 MFMAs per 16 keys, the K rows loaded once for all beams;
//   context  O[q][d]    = P . V_vis          4 MFMAs per 32 keys; V is read k-contiguous from a per-(image, head) TRANSPOSED copy
//                                            V^T[d][key] written once per batch by vt_build_kernel (vitcap_attn_beam_vt);
//   the <= 41 text keys of each sequence (its own cache rows) on the vector ALU as before.
// Softmax exactly as above: raw scores staged in LDS, m = ceil(max * c), P = exp2(s*c - m) rounded to bf16 for the product,
// row sum from the unrounded P.  4 waves split the key blocks; partial O / row sums are added in wave order.
// ------------------------------------------------------------------------------------------------
constexpr int VT_KP = 608;            // 578 visual keys padded to a multiple of 32 (zeros)
constexpr int BEAM_SC_LD = 657;       // 608 visual + 41 text + pad, fp32 scores per query row (odd: rows start in different LDS banks)

__global__ __launch_bounds__(256) void vt_build_kernel(const bf16_t* __restrict__ vis_qkv, bf16_t* __restrict__ vt, int S_vis) {
  // grid (VT_KP / 32, NH, images): thread = (dim = tid & 63, key group = tid >> 6 -> 8 keys)
  const int img = blockIdx.z, h = blockIdx.y, key0 = blockIdx.x * 32 + (threadIdx.x >> 6) * 8, d = threadIdx.x & 63;
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = key0 + i;
    const bf16_t raw = k < S_vis ? vis_qkv[((size_t)img * S_vis + k) * QKV_LD + 1536 + h * HD + d] : (bf16_t)0;
    v[i] = __builtin_bit_cast(__bf16, raw);
  }
  *(bf16x8*)(vt + (((size_t)img * NH + h) * HD + d) * VT_KP + key0) = v;
}

// MAXP: (sequence, text key) pairs per 8-lane group in phase 1b (K * (t+1) <= 32 * MAXP); MAXT: text keys per group and sequence
// in phase 2b (t + 1 <= 8 * MAXT).  Two instantiations: <4, 3> (5 beams x 20 tokens: fits two workgroups per CU), <11, 6> (any).
template <int MAXP, int MAXT>
__global__ __launch_bounds__(256) void attn_decode_beam_kernel(const bf16_t* __restrict__ qkv_step,
                                                               const bf16_t* __restrict__ vis_qkv,
                                                               const bf16_t* __restrict__ vis_vt,
                                                               bf16_t* __restrict__ text_kv, bf16_t* __restrict__ out,
                                                               int S_vis, int t, int max_len, int K, float c_log2,
                                                               const int32_t* __restrict__ live, int groups_per_image) {
  VC_LIVE_EXIT(live);
  // dynamic LDS sized by the image's 2*K query rows (5 beams: 47 KB -> three workgroups per CU): scores [NQ][BEAM_SC_LD], the
  // four waves' partial contexts [4][NQ][64], the text part [NQ][64]
  extern __shared__ __attribute__((aligned(16))) char beam_smem[];
  const int NQ_ = 2 * K;
  float (*sc)[BEAM_SC_LD] = (float (*)[BEAM_SC_LD])beam_smem;
  float (*s_o)[HD] = (float (*)[HD])(beam_smem + (size_t)NQ_ * BEAM_SC_LD * 4);      // [4 * NQ][64]: wave w, query q -> row w * NQ + q
  float (*s_ot)[HD] = s_o + 4 * NQ_;
  __shared__ float s_max[16][4];
  __shared__ float s_l[16][4];
  __shared__ float s_lt[16];
  __shared__ __attribute__((aligned(16))) bf16_t s_q[16][HD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // blockIdx.y = a group of K <= 8 sequences that share one image's visual rows; an image may own several groups (constrained beam
  // search decodes states x beams sequences per image)
  const int h = blockIdx.x, img = blockIdx.y / groups_per_image;
  const int frow = lane & 15, fk = lane >> 4;
  const int NQ = 2 * K;                       // query rows of this group: row n = sequence n / 2, step row n % 2
  const int b0 = blockIdx.y * K;
  const int ntext = t + 1;                    // text keys per sequence: cache rows 0..t-2, this step's row 0, the [MASK] row

  // publish this step's real-token K/V (row 0) of every sequence into its cache at position t-1
  if (tid < 16 * K) {
    const int j = tid >> 4, which = (tid >> 3) & 1, sub8 = tid & 7;
    const uint4 v = *(const uint4*)(qkv_step + ((size_t)(b0 + j) * 2) * QKV_LD + 768 * (1 + which) + h * HD + sub8 * 8);
    *(uint4*)(text_kv + (size_t)(b0 + j) * max_len * 2 * 768 + ((size_t)(t - 1) * 2 + which) * 768 + h * HD + sub8 * 8) = v;
  }

  // every global load of the workgroup is requested up front (visual K, V^T, the sequences' text K and V rows): one memory
  // round trip per workgroup instead of one per loop iteration (the first version spent ~10 dependent round trips: 139 us)
  const int sub = lane & 7, grp = tid >> 3;   // vector-ALU part: 32 groups of 8 lanes, one (sequence, text key) pair each
  auto text_ptr = [&](int j, int tp, int which) -> const bf16_t* {
    if (tp < t - 1) return text_kv + (size_t)(b0 + j) * max_len * 2 * 768 + ((size_t)tp * 2 + which) * 768 + h * HD + sub * 8;
    return qkv_step + ((size_t)(b0 + j) * 2 + (tp - (t - 1))) * QKV_LD + 768 * (1 + which) + h * HD + sub * 8;
  };
  const int npair = K * ntext;
  bf16x8 tk[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int pr = grp + 32 * i;
    if (pr < npair) {
      const int j = pr / ntext;
      tk[i] = *(const bf16x8*)text_ptr(j, pr - j * ntext, 0);
    }
  }
  bf16x8 tv[2][MAXT];                         // wave w: sequences w and w + 4
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      const int j = w + 4 * jj, tp = (lane >> 3) + 8 * i;
      if (j < K && tp < ntext) tv[jj][i] = *(const bf16x8*)text_ptr(j, tp, 1);
    }
  if (tid < NQ * 8) {                         // the query rows for the vector-ALU part
    const int n = tid >> 3;
    *(bf16x8*)&s_q[n][(tid & 7) * 8] = *(const bf16x8*)(qkv_step + ((size_t)b0 * 2 + n) * QKV_LD + h * HD + (tid & 7) * 8);
  }

  // ---- phase 1a: visual scores on the matrix pipe.  Y operand = Q (row = query frow, 8 dims at fk*8 [+32]); X = K rows.
  bf16x8 qf[2];
  {
    const bf16_t* qp = qkv_step + ((size_t)b0 * 2 + (frow < NQ ? frow : 0)) * QKV_LD + h * HD + fk * 8;
    qf[0] = *(const bf16x8*)qp;
    qf[1] = *(const bf16x8*)(qp + 32);
    if (frow >= NQ) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { qf[0][j] = (__bf16)0.f; qf[1][j] = (__bf16)0.f; }
    }
  }
  const int nkb = (S_vis + 15) >> 4;          // 16-key blocks (37)
  constexpr int MAXB = 10;                    // blocks per wave: ceil(37 / 4)
  bf16x8 kf[MAXB][2];
#pragma unroll
  for (int i = 0; i < MAXB; ++i) {
    const int kb = w + 4 * i;
    if (kb < nkb) {
      int key = kb * 16 + frow;
      key = key < S_vis ? key : S_vis - 1;
      const bf16_t* kp = vis_qkv + ((size_t)img * S_vis + key) * QKV_LD + 768 + h * HD + fk * 8;
      kf[i][0] = *(const bf16x8*)kp;
      kf[i][1] = *(const bf16x8*)(kp + 32);
    }
  }
  float mx = -1e30f;                          // this lane's query (frow) over its keys
#pragma unroll
  for (int i = 0; i < MAXB; ++i) {
    const int kb = w + 4 * i;
    if (kb < nkb) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[i][0], qf[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[i][1], qf[1], acc, 0, 0, 0);
      // lane holds S[query frow][key kb*16 + fk*4 + e]
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int key = kb * 16 + fk * 4 + e;
        const float v = key < S_vis ? acc[e] : -INFINITY;
        if (frow < NQ) sc[frow][key] = v;
        mx = fmaxf(mx, v);
      }
    }
  }
  // the V^T fragments of phase 2a are requested here (the K fragments' registers are free again): their latency is covered by
  // the text scores and the maxima below
  constexpr int MAXV = 5;                     // 32-key blocks per wave: ceil(19 / 4)
  bf16x8 vf[MAXV][4];
  const bf16_t* vtb = vis_vt + ((size_t)img * NH + h) * HD * VT_KP;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int kb = w + 4 * i;
    if (kb < VT_KP / 32) {
#pragma unroll
      for (int db = 0; db < 4; ++db) vf[i][db] = *(const bf16x8*)(vtb + (size_t)(db * 16 + frow) * VT_KP + kb * 32 + fk * 8);
    }
  }
  if (w == 0 && fk == 0) {                    // keys S_vis .. VT_KP-1 of the padded blocks: no weight
    if (frow < NQ)
      for (int key = nkb * 16; key < VT_KP; ++key) sc[frow][key] = -INFINITY;
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  if (fk == 0) s_max[frow][w] = mx;

  // ---- phase 1b: text scores on the vector ALU: 8 lanes per (sequence, text key) pair, as attn_decode_kernel
  __syncthreads();                            // s_q
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int pr = grp + 32 * i;
    if (pr < npair) {
      const int j = pr / ntext, tp = pr - j * ntext;
      const bf16x8 a0 = __builtin_bit_cast(bf16x8, *(const uint4*)&s_q[2 * j][sub * 8]);
      const bf16x8 a1 = __builtin_bit_cast(bf16x8, *(const uint4*)&s_q[2 * j + 1][sub * 8]);
      float d0 = 0.f, d1 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float kx = (float)tk[i][e];
        d0 += (float)a0[e] * kx;
        d1 += (float)a1[e] * kx;
      }
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) {
        d0 += __shfl_xor(d0, o, 64);
        d1 += __shfl_xor(d1, o, 64);
      }
      if (tp == t) d0 = -INFINITY;            // row 0 (position t-1) cannot see the [MASK] row
      if (sub == 0) { sc[2 * j][VT_KP + tp] = d0; sc[2 * j + 1][VT_KP + tp] = d1; }
    }
  }
  __syncthreads();
  // row maxima: the 4 waves' visual maxima and the text scores
  float m_q;                                  // for query (tid >> 4) -- 16 threads per query
  {
    const int qn = tid >> 4, i16 = tid & 15;
    float m = fmaxf(fmaxf(s_max[qn][0], s_max[qn][1]), fmaxf(s_max[qn][2], s_max[qn][3]));
    if (qn < NQ)
      for (int tp = i16; tp < ntext; tp += 16) m = fmaxf(m, sc[qn][VT_KP + tp]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    m_q = ceilf(m * c_log2);
  }
  __syncthreads();
  if ((tid & 15) == 0) s_max[tid >> 4][0] = m_q;
  __syncthreads();

  // ---- phase 2a: P . V over the visual keys on the matrix pipe.  Y = P (row = query frow, 8 keys at fk*8 of a 32-key block)
  const float mrow = s_max[frow][0];
  f32x4 oacc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) oacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
  float lsum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int kb = w + 4 * i;
    if (kb < VT_KP / 32) {
      bf16x8 pf;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float p = frow < NQ ? fast_exp2(fmaf(sc[frow][kb * 32 + fk * 8 + e], c_log2, -mrow)) : 0.f;
        lsum += p;
        pf[e] = (__bf16)p;
      }
      // mfma(X = V^T rows (dims), Y = P rows (queries)): lane holds O[query frow][dim db*16 + fk*4 + e]
#pragma unroll
      for (int db = 0; db < 4; ++db) oacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[i][db], pf, oacc[db], 0, 0, 0);
    }
  }
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  if (fk == 0) s_l[frow][w] = lsum;
  if (frow < NQ) {
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int e = 0; e < 4; ++e) s_o[w * NQ + frow][db * 16 + fk * 4 + e] = oacc[db][e];
  }

  // ---- phase 2b: text keys on the vector ALU; wave w takes sequences w, w + 4 (8 key groups of 8 lanes each)
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int j = w + 4 * jj;
    if (j < K) {
      float o0[8], o1[8], l0 = 0.f, l1 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
      const float m0 = s_max[2 * j][0], m1 = s_max[2 * j + 1][0];
#pragma unroll
      for (int i = 0; i < MAXT; ++i) {
        const int tp = (lane >> 3) + 8 * i;
        if (tp < ntext) {
          const float p0 = fast_exp2(fmaf(sc[2 * j][VT_KP + tp], c_log2, -m0)), p1 = fast_exp2(fmaf(sc[2 * j + 1][VT_KP + tp], c_log2, -m1));
          l0 += p0;
          l1 += p1;
          const float p0b = (float)(__bf16)p0, p1b = (float)(__bf16)p1;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float vx = (float)tv[jj][i][e];
            o0[e] += p0b * vx;
            o1[e] += p1b * vx;
          }
        }
      }
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        l0 += __shfl_xor(l0, o, 64);
        l1 += __shfl_xor(l1, o, 64);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o0[e] += __shfl_xor(o0[e], o, 64);
          o1[e] += __shfl_xor(o1[e], o, 64);
        }
      }
      if (lane < 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s_ot[2 * j][lane * 8 + e] = o0[e];
          s_ot[2 * j + 1][lane * 8 + e] = o1[e];
        }
      }
      if (lane == 0) { s_lt[2 * j] = l0; s_lt[2 * j + 1] = l1; }
    }
  }
  __syncthreads();
  // ---- merge: out[q][d] = (sum over waves of the visual part + the text part) / (row sums)
  for (int i = tid; i < NQ * HD; i += 256) {
    const int qn = i >> 6, d = i & 63;
    const float l = ((s_l[qn][0] + s_l[qn][1]) + (s_l[qn][2] + s_l[qn][3])) + s_lt[qn];
    const float o = ((s_o[qn][d] + s_o[NQ + qn][d]) + (s_o[2 * NQ + qn][d] + s_o[3 * NQ + qn][d])) + s_ot[qn][d];
    out[((size_t)b0 * 2 + qn) * 768 + h * HD + d] = f2bf(o / l);
  }
}

}  // namespace

extern "C" int vitcap_attn_beam_vt(const void* vis_qkv, void* vis_vt, int n_images, int S_vis, void* stream) {
  VC_REQUIRE(vis_qkv && vis_vt && n_images > 0 && S_vis > 0 && S_vis <= VT_KP, "attn_beam_vt: bad arguments (S_vis <= %d)", VT_KP);
  hipLaunchKernelGGL(vt_build_kernel, dim3(VT_KP / 32, NH, n_images), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)vis_qkv,
                     (bf16_t*)vis_vt, S_vis);
  VC_LAUNCH_CHECK("attn_beam_vt");
  return VITCAP_OK;
}

static int attn_decode_groups(const void* qkv_step, const void* vis_qkv, const void* vis_vt, void* text_kv, void* out, int n_images,
                              int seq_per_image, int groups_per_image, int S_vis, int t, int max_len, float scale, void* stream);

extern "C" int vitcap_attn_decode_beams(const void* qkv_step, const void* vis_qkv, const void* vis_vt, void* text_kv, void* out,
                                        int n_images, int seq_per_image, int S_vis, int t, int max_len, float scale, void* stream) {
  return attn_decode_groups(qkv_step, vis_qkv, vis_vt, text_kv, out, n_images, seq_per_image, 1, S_vis, t, max_len, scale, stream);
}

extern "C" int vitcap_attn_decode_beam_groups(const void* qkv_step, const void* vis_qkv, const void* vis_vt, void* text_kv, void* out,
                                              int n_images, int seq_per_group, int groups_per_image, int S_vis, int t, int max_len,
                                              float scale, void* stream) {
  return attn_decode_groups(qkv_step, vis_qkv, vis_vt, text_kv, out, n_images, seq_per_group, groups_per_image, S_vis, t, max_len, scale,
                            stream);
}

static int attn_decode_groups(const void* qkv_step, const void* vis_qkv, const void* vis_vt, void* text_kv, void* out, int n_images,
                              int seq_per_image, int groups_per_image, int S_vis, int t, int max_len, float scale, void* stream) {
  VC_REQUIRE(qkv_step && vis_qkv && vis_vt && text_kv && out && n_images > 0 && groups_per_image >= 1, "attn_decode_beams: bad arguments");
  VC_REQUIRE(seq_per_image >= 1 && seq_per_image <= 8, "attn_decode_beams: 1..8 sequences per group (got %d)", seq_per_image);
  VC_REQUIRE(t >= 1 && t < max_len && max_len <= 41 && S_vis > 16 && S_vis <= VT_KP && (S_vis + 15) / 16 <= 40,
             "attn_decode_beams: t=%d max_len=%d S_vis=%d out of range", t, max_len, S_vis);
  const float c = scale * 1.4426950408889634f;
  const int nq = 2 * seq_per_image;
  const size_t smem = (size_t)nq * BEAM_SC_LD * 4 + (size_t)5 * nq * HD * 4;      // <= 62.5 KB at 8 sequences per image
  VC_FUNC_SMEM((attn_decode_beam_kernel<4, 3>), 64 * 1024);
  VC_FUNC_SMEM((attn_decode_beam_kernel<11, 6>), 64 * 1024);
  if (seq_per_image * (t + 1) <= 32 * 4 && t + 1 <= 8 * 3)
    hipLaunchKernelGGL((attn_decode_beam_kernel<4, 3>), dim3(NH, n_images * groups_per_image), dim3(256), smem, (hipStream_t)stream, (const bf16_t*)qkv_step,
                       (const bf16_t*)vis_qkv, (const bf16_t*)vis_vt, (bf16_t*)text_kv, (bf16_t*)out, S_vis, t, max_len,
                       seq_per_image, c, vc_tls_live, groups_per_image);
  else
    hipLaunchKernelGGL((attn_decode_beam_kernel<11, 6>), dim3(NH, n_images * groups_per_image), dim3(256), smem, (hipStream_t)stream, (const bf16_t*)qkv_step,
                       (const bf16_t*)vis_qkv, (const bf16_t*)vis_vt, (bf16_t*)text_kv, (bf16_t*)out, S_vis, t, max_len,
                       seq_per_image, c, vc_tls_live, groups_per_image);
  VC_LAUNCH_CHECK("attn_decode_beams");
  return VITCAP_OK;
}


#define VC_LAUNCH_DENSE(DROP_, grid_, stream_, ...) \
  hipLaunchKernelGGL(attn_dense_kernel<DROP_>, grid_, dim3(256), 0, (hipStream_t)(stream_), __VA_ARGS__, vc_tls_walk_rev ? 1 : 0, vc_tls_drop_salt)

extern "C" int vitcap_attn_dense_fwd(const void* qkv, void* out, int B, int S, float scale, void* stream) {
  VC_REQUIRE(qkv && out && B > 0 && S > 0, "attn_dense: bad arguments");
  VC_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0, "attn_dense: misaligned");
  const float c = scale * 1.4426950408889634f;
  dim3 grid(((S + 127) / 128) * NH * B);   // 1-D work list, remapped per XCD inside the kernel
  VC_LAUNCH_DENSE(false, grid, stream, (const bf16_t*)qkv, (bf16_t*)out, (float*)nullptr, S, B, S, c, 0u, 0u, 1.0f, 0, 0, 0, S);
  VC_LAUNCH_CHECK("attn_dense");
  return VITCAP_OK;
}

extern "C" int vitcap_attn_dense_fwd_rows(const void* qkv, void* out, int B, int S, int q_rows, float scale, void* stream) {
  VC_REQUIRE(qkv && out && B > 0 && S > 0 && q_rows >= 1 && q_rows <= S, "attn_dense_rows: bad arguments");
  VC_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0, "attn_dense_rows: misaligned");
  const float c = scale * 1.4426950408889634f;
  dim3 grid(((q_rows + 127) / 128) * NH * B);
  VC_LAUNCH_DENSE(false, grid, stream, (const bf16_t*)qkv, (bf16_t*)out, (float*)nullptr, S, B, S, c, 0u, 0u, 1.0f, 0, 0, 0, q_rows);
  VC_LAUNCH_CHECK("attn_dense_rows");
  return VITCAP_OK;
}

extern "C" int vitcap_attn_dense_fwd_train_rows(const void* qkv, void* out, float* lse, int B, int S, int ld_rows,
                                                float scale, float p_drop, uint32_t drop_seed, int causal_from, int mask_from,
                                                int q_lo, int q_hi, void* stream) {
  VC_REQUIRE(qkv && out && lse && B > 0 && S > 0 && ld_rows >= S, "attn_dense_train: bad arguments");
  VC_REQUIRE(q_lo >= 0 && q_lo < q_hi && q_hi <= S && (q_lo & 127) == 0, "attn_dense_train: query range [%d, %d) must start on a multiple of 128 inside S=%d", q_lo, q_hi, S);
  VC_REQUIRE(p_drop >= 0.f && p_drop < 1.f && ld_rows < 1024, "attn_dense_train: p_drop %g / ld_rows %d out of range",
             (double)p_drop, ld_rows);
  VC_REQUIRE(causal_from == 0 || (causal_from >= (S / KT) * KT && causal_from <= S),
             "attn_dense_train: causal_from %d must lie in the last key tile of S=%d", causal_from, S);
  VC_REQUIRE(mask_from == 0 || (causal_from > 0 && mask_from > causal_from && mask_from <= S),
             "attn_dense_train: mask_from %d must lie in (causal_from %d, S %d]", mask_from, causal_from, S);
  const float c = scale * 1.4426950408889634f;
  dim3 grid(((q_hi - q_lo + 127) / 128) * NH * B);
  if (p_drop > 0.f)
    VC_LAUNCH_DENSE(true, grid, stream, (const bf16_t*)qkv, (bf16_t*)out, lse, S, B, ld_rows, c, drop_seed,
                    (uint32_t)((double)p_drop * 4294967296.0), 1.0f / (1.0f - p_drop), causal_from, mask_from, q_lo, q_hi - q_lo);
  else
    VC_LAUNCH_DENSE(false, grid, stream, (const bf16_t*)qkv, (bf16_t*)out, lse, S, B, ld_rows, c, 0u, 0u, 1.0f, causal_from,
                    mask_from, q_lo, q_hi - q_lo);
  VC_LAUNCH_CHECK("attn_dense_train");
  return VITCAP_OK;
}

extern "C" int vitcap_attn_dense_fwd_train(const void* qkv, void* out, float* lse, int B, int S, int ld_rows,
                                           float scale, float p_drop, uint32_t drop_seed, int causal_from, int mask_from,
                                           void* stream) {
  return vitcap_attn_dense_fwd_train_rows(qkv, out, lse, B, S, ld_rows, scale, p_drop, drop_seed, causal_from, mask_from, 0, S,
                                          stream);
}

extern "C" int vitcap_attn_decode_step_tags(const void* qkv_step, const void* vis_qkv, void* text_kv, void* out, int B,
                                            int S_vis, int t, int max_len, int seq_per_image, float scale, const void* tag_qkv_a,
                                            const void* tag_qkv_b, int n_tag, const int64_t* tag_len, void* stream) {
  VC_REQUIRE(qkv_step && vis_qkv && text_kv && out && B > 0, "attn_decode: bad arguments");
  VC_REQUIRE(n_tag >= 0 && n_tag <= 50 && (n_tag == 0 || (tag_qkv_a && tag_qkv_b && tag_len)), "attn_decode: bad tag keys (n_tag=%d)", n_tag);
  VC_REQUIRE(t >= 1 && t < max_len && S_vis + n_tag + t + 1 <= MAXKEYS, "attn_decode: t=%d S_vis=%d out of range", t, S_vis);
  VC_REQUIRE(seq_per_image >= 1 && B % seq_per_image == 0, "attn_decode: bad seq_per_image");
  const float c = scale * 1.4426950408889634f;
  hipLaunchKernelGGL(attn_decode_kernel, dim3(NH, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv_step,
                     (const bf16_t*)vis_qkv, (bf16_t*)text_kv, (bf16_t*)out, S_vis, t, max_len, seq_per_image, c, vc_tls_live,
                     (const bf16_t*)tag_qkv_a, (const bf16_t*)tag_qkv_b, n_tag, tag_len);
  VC_LAUNCH_CHECK("attn_decode");
  return VITCAP_OK;
}

extern "C" int vitcap_attn_decode_step(const void* qkv_step, const void* vis_qkv, void* text_kv, void* out, int B,
                                       int S_vis, int t, int max_len, int seq_per_image, float scale, void* stream) {
  return vitcap_attn_decode_step_tags(qkv_step, vis_qkv, text_kv, out, B, S_vis, t, max_len, seq_per_image, scale, nullptr, nullptr,
                                      0, nullptr, stream);
}
