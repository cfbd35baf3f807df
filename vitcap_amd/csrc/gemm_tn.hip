// Weight-gradient GEMM for gfx950:  C[N][K] (fp32) = sum_m Y[m][n] * X[m][k]   ("TN": both operands are stored with the
// REDUCTION index m as the slow dimension, exactly as the backward pass produces them -- dY[M][N] and the saved
// activation X[M][K], row-major).  nn.Linear's dW = dY^T X (what autograd computes for modeling_bert.py / timm Linear
// layers) therefore needs no transposed copies of the activations.
//
// MFMA fragments want 8 consecutive reduction elements per lane for a fixed row; here those are strided by the row
// pitch.  The tiles are staged row-major ([64 m][256 cols], LDS-DMA, 16 B per lane) and read with the CDNA4 LDS
// transpose read `ds_read_b64_tr_b16`: a 16-lane group fetches a [4 m][16 col] block and lane i receives column i of
// the 4 rows (measured semantics: lane i, element j  <-  address of lane 4j + i/4, element i%4).
//
// Tile 256(n) x 256(k) per workgroup, 8 waves as 2(n) x 4(k), wave tile 128 x 64, mfma_f32_16x16x32_bf16, two
// 64-row stages in LDS (128 KiB).  The reduction dimension (M = batch x tokens, 36 928 at B=64) is long and the output
// small (768..3072 x 768..3072), so the grid is (output tiles) x (splits of M): every split writes its own fp32 slab
// (deterministic, no atomics), reduced by vitcap_reduce_slabs.
//
// LDS bank conflicts: a tile row is 512 B, so all rows alias to the same banks; 32-byte chunk C of row m is stored at
// chunk C ^ f(m), f(m) = 4*((m>>3)&3) + (m&3).  One transpose read touches rows {8g + j} (g = lane group, j = 0..3):
// 16 different f values, 8 per half-wave -> 8 distinct bank groups per LDS cycle.  The swizzle is applied by choosing
// which global 16-byte chunk each DMA lane fetches (the LDS side of the DMA is lane-linear).
#include <atomic>

#include "common.h"

namespace {

constexpr int TBN = 256, TBK = 256, TBM = 64;
constexpr int ROW_B = 512;                       // bytes per tile row (256 bf16)
constexpr int OP_BYTES = TBM * ROW_B;            // 32 KiB per operand per stage
#ifndef VC_TN_ALTERNATE     // 1: the two wave groups one segment apart (needs VC_TN_RING4)
#define VC_TN_ALTERNATE 1
#endif
#ifndef VC_TN_RING4
#define VC_TN_RING4 1
#endif
constexpr int STAGE_BYTES = 2 * OP_BYTES;

struct TnArgs {
  const bf16_t* Y;   // [M][ldy], columns n0.. used
  const bf16_t* X;   // [M][ldx]
  float* C;          // [split][N][K]
  const bf16_t* zeros;   // >= 512 B of zeros (rows past M)
  int M, N, K, ldy, ldx;
  int tiles_n, tiles_k, splits, stages_per_split;
  float* out;        // != nullptr: the workgroups of a tile add their slabs up themselves (below), out[N][K] (+)= sum over splits
  int* cnt;          // [tiles][2] arrive / depart tickets, zero between launches
  int accumulate;
};

__device__ __forceinline__ void glds16(const void* g, void* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
__device__ __forceinline__ int swz(int m) { return ((m >> 3) & 3) * 4 + (m & 3); }

__global__ __launch_bounds__(512) void gemm_tn_kernel(TnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = w >> 2, wk = w & 3;
  // XCD-aware work order: workgroup id b runs on XCD b % 8 (observed dispatch rule), so XCD x takes the x-th contiguous
  // chunk of the (split-major, tile-minor) work list -- the ~32 workgroups an XCD runs at once are then the tiles of ONE
  // split, which share their Y row-slab across tiles_k workgroups and their X row-slab across tiles_n in that XCD's L2
  // (dealt round-robin, every operand tile crossed the fabric once per consumer: 1.0 GB per qkv weight gradient against
  // 227 MB of unique data).  Worth 2-10 % (tools/wgrad_bench.py); the kernel is bound elsewhere, see below.
  const int tiles = p.tiles_n * p.tiles_k;
  const int nwg = tiles * p.splits;
  int pos = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = pos & 7;
    pos = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (pos >> 3);
  }
  const int split = pos / tiles, tile = pos - split * tiles;
  const int tn = tile / p.tiles_k, tk = tile - tn * p.tiles_k;
  const int n0 = tn * TBN, k0 = tk * TBK;
  const int s_begin = split * p.stages_per_split;
  const int total_stages = (p.M + TBM - 1) / TBM;
  int s_end = s_begin + p.stages_per_split;
  s_end = s_end < total_stages ? s_end : total_stages;
  const int nst = s_end > s_begin ? s_end - s_begin : 0;

  // ---- DMA: wave w fills rows 8w .. 8w+7 of each operand tile, 4 instructions of 2 rows each per operand
  const int drow = lane >> 5;                      // row within the instruction's pair
  const int dchunk = lane & 31;                    // physical 16-byte chunk within the row
#define STAGE(buf_, st_)                                                                            \
  do {                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                 \
      const int r_ = w * 8 + i * 2 + drow;           /* tile row 0..63 */                           \
      const int lc_ = dchunk ^ (swz(r_) << 1);       /* logical 16-byte chunk this lane fetches */  \
      const int m_ = (st_) * TBM + r_;                                                              \
      const bf16_t* ys_ = m_ < p.M ? p.Y + (size_t)m_ * p.ldy + n0 + lc_ * 8 : p.zeros + lc_ * 8;   \
      const bf16_t* xs_ = m_ < p.M ? p.X + (size_t)m_ * p.ldx + k0 + lc_ * 8 : p.zeros + lc_ * 8;   \
      char* dst_ = smem + (buf_) * STAGE_BYTES + (w * 8 + i * 2) * ROW_B;                           \
      glds16(ys_, dst_);                                                                            \
      glds16(xs_, dst_ + OP_BYTES);                                                                 \
    }                                                                                               \
  } while (0)

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- transpose-read addressing: lane group g = lane>>4 owns reduction rows 8g..8g+7 of each 32-row step; within the
  // group, lane i supplies the address of row j = i>>2, 8-byte piece q = i&3 of the 16-column block
  const int g = lane >> 4, gi = lane & 15;
  const int tj = gi >> 2, tq = gi & 3;
  const int fsw = g * 4 + tj;                      // swz(row) for both reads (rows +0..3 and +4..7 share it)
  const int row_lo = g * 8 + tj;                   // + ms*32 (+4 for the second read)

  // Measured anatomy (M = 36928, N = 2304, K = 768, 9 splits; us incl. the 12 us slab reduction): full 200 | no DMA 116 | no
  // transpose reads 136 | no MFMA 180 | DMA only 127 | reads only 79 | MFMA only 107 | barriers + slab stores only 41.
  // The MFMAs are already hidden; the critical path is LDS-DMA in (25 B/clk/CU, the same ceiling the NT kernel sees) PLUS
  // the transpose reads out, which do not overlap each other.  Tried and measured, both slower or equal, both reverted:
  // the two n-halves one phase apart as in gemm_nt_256_kernel (+3 %), and a ring of four 32-row stages with the request
  // issued three stages ahead (DMA-only 127 -> 104 us, but full 200 -> 242 -- with the waits of that time, see below: round 3
  // built it again on counted waits and it is the shipped form, VC_TN_RING4); register-staged tiles (global_load -> VGPR ->
  // ds_write_b128) instead of LDS-DMA: 195 -> 499 us.  SQ counters: no LDS bank conflicts, 67 % of wave cycles in
  // s_waitcnt/barriers, MFMA pipe 33 % busy.
  // Round 2, measured and reverted: the bias gradient (column sums of Y) as extra MFMAs against an all-ones fragment inside this
  // kernel (8 per 32-row step on the wk = 0 waves of the k-tile-0 workgroups, or 2 per wave) instead of the separate colsum pass:
  // the colsum launches disappear (-1.7 ms per training step) but this kernel slows from 13.15 to 14.6 / 14.9 ms per step -- the
  // MFMA issue slots are not free although the pipe is a third busy.  No net gain.
  // Transpose reads as inline asm (common.h lds_tr_read): with the intrinsic, hipcc put `s_waitcnt vmcnt(0)` in front of the
  // first read of every stage, i.e. the stage requested a few instructions earlier was awaited before this one was
  // multiplied -- LDS-DMA in and reads + MFMAs out ran back to back (the "DMA + reads do not overlap" of the anatomy above).
  // Per 32-row step: the X fragments and the first four Y fragments are requested, then -- behind their lgkmcnt(0) -- the
  // other four Y fragments, whose latency hides behind the first 16 MFMAs.
  uint32_t yaddr[8], xaddr[4];
  {
    const uint32_t lane_base = (uint32_t)(row_lo * ROW_B + tq * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) yaddr[i] = lane_base + (uint32_t)(((wn * 8 + i) ^ fsw) * 32);
#pragma unroll
    for (int j = 0; j < 4; ++j) xaddr[j] = lane_base + (uint32_t)(VC_TN_RING4 ? OP_BYTES / 2 : OP_BYTES) + (uint32_t)(((wk * 4 + j) ^ fsw) * 32);
  }
  const uint32_t lds0 = lds_addr(smem);
#define TN_STEP(MS_)                                                                                                  \
  do {                                                                                                                \
    s16x4 xl[4], xh[4], yl[8], yh[8];                                                                                 \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                   \
      xl[j] = lds_tr_read<(MS_) * 32 * ROW_B>(sb + xaddr[j]);                                                         \
      xh[j] = lds_tr_read<(MS_) * 32 * ROW_B + 4 * ROW_B>(sb + xaddr[j]);                                             \
    }                                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                   \
      yl[i] = lds_tr_read<(MS_) * 32 * ROW_B>(sb + yaddr[i]);                                                         \
      yh[i] = lds_tr_read<(MS_) * 32 * ROW_B + 4 * ROW_B>(sb + yaddr[i]);                                             \
    }                                                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                               \
                 : "+v"(xl[0]), "+v"(xh[0]), "+v"(xl[1]), "+v"(xh[1]), "+v"(xl[2]), "+v"(xh[2]), "+v"(xl[3]), "+v"(xh[3]), \
                   "+v"(yl[0]), "+v"(yh[0]), "+v"(yl[1]), "+v"(yh[1]), "+v"(yl[2]), "+v"(yh[2]), "+v"(yl[3]), "+v"(yh[3])); \
    _Pragma("unroll") for (int i = 4; i < 8; ++i) {                                                                   \
      yl[i] = lds_tr_read<(MS_) * 32 * ROW_B>(sb + yaddr[i]);                                                         \
      yh[i] = lds_tr_read<(MS_) * 32 * ROW_B + 4 * ROW_B>(sb + yaddr[i]);                                             \
    }                                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                     \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                   \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(xl[j], xh[j]), tr_pair(yl[i], yh[i]), acc[i][j], 0, 0, 0); \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                               \
                 : "+v"(yl[4]), "+v"(yh[4]), "+v"(yl[5]), "+v"(yh[5]), "+v"(yl[6]), "+v"(yh[6]), "+v"(yl[7]), "+v"(yh[7])); \
    _Pragma("unroll") for (int i = 4; i < 8; ++i)                                                                     \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                   \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(xl[j], xh[j]), tr_pair(yl[i], yh[i]), acc[i][j], 0, 0, 0); \
  } while (0)

#if VC_TN_RING4
  // ring of FOUR 32-row stages, requests three stages ahead: more bytes in flight per CU than the two 64-row buffers (the kernel
  // is bound by LDS-DMA throughput in), one barrier per 32-row step.  Round 2 measured this form slower -- with the transpose-read
  // intrinsic every step waited for vmcnt(0), which four short stages pay twice as often; with the asm reads the counted wait holds.
  const int s32_begin = 2 * s_begin, nst32 = 2 * nst;
#define STAGE32(buf_, st_)                                                                          \
  do {                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                 \
      const int r_ = w * 4 + i * 2 + drow;           /* stage row 0..31 */                          \
      const int lc_ = dchunk ^ (swz(r_) << 1);                                                      \
      const int m_ = (st_) * 32 + r_;                                                               \
      const bf16_t* ys_ = m_ < p.M ? p.Y + (size_t)m_ * p.ldy + n0 + lc_ * 8 : p.zeros + lc_ * 8;   \
      const bf16_t* xs_ = m_ < p.M ? p.X + (size_t)m_ * p.ldx + k0 + lc_ * 8 : p.zeros + lc_ * 8;   \
      char* dst_ = smem + (buf_) * (STAGE_BYTES / 2) + (w * 4 + i * 2) * ROW_B;                     \
      glds16(ys_, dst_);                                                                            \
      glds16(xs_, dst_ + OP_BYTES / 2);                                                             \
    }                                                                                               \
  } while (0)
  // stage T_ has landed for this wave when at most the requests of the two younger stages (4 per wave each) are outstanding
#define WAIT_STAGE(T_)                                                          \
  do {                                                                          \
    const int newer_ = nst32 - 1 - (T_);                                        \
    if (newer_ >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           \
    else if (newer_ == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       \
  } while (0)
#pragma unroll
  for (int q = 0; q < 3; ++q)
    if (q < nst32) STAGE32(q, s32_begin + q);
#if VC_TN_ALTERNATE
  // The two wave groups (wn = 0 / 1: waves w and w + 4 share a SIMD) run one segment apart, as in gemm_nt_256_kernel: a step is a
  // LOAD segment (24 transpose reads) and an MFMA segment (32 MFMAs), each closed by a barrier, and while one wave of a SIMD
  // multiplies, the other reads.  Barrier 2t opens L_A(t) (stage t landed for everyone: every wave waited for its own pieces
  // before it) and M_B(t-1); barrier 2t+1 opens M_A(t) and L_B(t).  The buffer of stage t-1 was last read in L_B(t-1), which
  // barrier 2t closes, so either group requests stage t+3 right behind that barrier.
  s16x4 xl[4], xh[4], yl[8], yh[8];
#define TN_LOAD()                                                                                                     \
  do {                                                                                                                \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                   \
      xl[j] = lds_tr_read<0>(sb + xaddr[j]);                                                                          \
      xh[j] = lds_tr_read<4 * ROW_B>(sb + xaddr[j]);                                                                  \
    }                                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                   \
      yl[i] = lds_tr_read<0>(sb + yaddr[i]);                                                                          \
      yh[i] = lds_tr_read<4 * ROW_B>(sb + yaddr[i]);                                                                  \
    }                                                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                               \
                 : "+v"(xl[0]), "+v"(xh[0]), "+v"(xl[1]), "+v"(xh[1]), "+v"(xl[2]), "+v"(xh[2]), "+v"(xl[3]), "+v"(xh[3]), \
                   "+v"(yl[0]), "+v"(yh[0]), "+v"(yl[1]), "+v"(yh[1]), "+v"(yl[2]), "+v"(yh[2]), "+v"(yl[3]), "+v"(yh[3]), \
                   "+v"(yl[4]), "+v"(yh[4]), "+v"(yl[5]), "+v"(yh[5]), "+v"(yl[6]), "+v"(yh[6]), "+v"(yl[7]), "+v"(yh[7])); \
  } while (0)
#define TN_MFMA()                                                                                                     \
  do {                                                                                                                \
    __builtin_amdgcn_s_setprio(1);                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                     \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                   \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(xl[j], xh[j]), tr_pair(yl[i], yh[i]), acc[i][j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                                    \
  } while (0)
  if (wn == 0) {
    for (int t = 0; t < nst32; ++t) {
      WAIT_STAGE(t);
      __builtin_amdgcn_s_barrier();                                           // 2t
      if (t + 3 < nst32) STAGE32((t + 3) & 3, s32_begin + t + 3);
      const uint32_t sb = lds0 + (uint32_t)((t & 3) * (STAGE_BYTES / 2));
      TN_LOAD();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();                                           // 2t + 1
      TN_MFMA();
    }
    __builtin_amdgcn_s_barrier();                                             // 2 nst32: group B's last
  } else {
    if (nst32 > 0) WAIT_STAGE(0);
    __builtin_amdgcn_s_barrier();                                             // 0
    if (3 < nst32) STAGE32(3, s32_begin + 3);
    for (int t = 0; t < nst32; ++t) {
      __builtin_amdgcn_s_barrier();                                           // 2t + 1
      const uint32_t sb = lds0 + (uint32_t)((t & 3) * (STAGE_BYTES / 2));
      TN_LOAD();
      if (t + 1 < nst32) WAIT_STAGE(t + 1);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();                                           // 2t + 2
      if (t + 4 < nst32) STAGE32((t + 4) & 3, s32_begin + t + 4);
      TN_MFMA();
    }
  }
#undef TN_LOAD
#undef TN_MFMA
#else
  for (int t = 0; t < nst32; ++t) {
    WAIT_STAGE(t);
    __builtin_amdgcn_s_barrier();            // stage t visible to all waves; every wave is done reading stage t-1 ...
    if (t + 3 < nst32) STAGE32((t + 3) & 3, s32_begin + t + 3);      // ... whose buffer the request for stage t+3 refills
    const uint32_t sb = lds0 + (uint32_t)((t & 3) * (STAGE_BYTES / 2));
    TN_STEP(0);
  }
#endif
#undef WAIT_STAGE
#undef STAGE32
#else
  if (nst > 0) STAGE(0, s_begin);
  for (int t = 0; t < nst; ++t) {
    const int buf = t & 1;
    if (t + 1 < nst) {
      STAGE(buf ^ 1, s_begin + t + 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // this stage's 8 pieces landed; the next 8 stay in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // raw s_barrier, not __syncthreads(): its workgroup-scope fence makes hipcc drain vmcnt(0) in front of the barrier, i.e.
    // wait for the stage requested three lines up.  Nothing here needs the fence: LDS is written by the DMA only (awaited by the
    // counted vmcnt above) and read by the asm transpose reads, each fenced by its own lgkmcnt(0) before the second barrier.
    __builtin_amdgcn_s_barrier();
    const uint32_t sb = lds0 + (uint32_t)(buf * STAGE_BYTES);
    TN_STEP(0);
    TN_STEP(1);
    __builtin_amdgcn_s_barrier();                          // everyone done reading `buf` before it is refilled
  }
#endif
#undef TN_STEP
#undef STAGE

  // ---- store the slab: with the operands swapped the lane holds 4 consecutive k of ONE n (16-byte stores)
  float* cs = p.C + (size_t)split * p.N * p.K;
  const int nl = lane & 15, kq = (lane >> 4) * 4;
  if (!p.out) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int n = n0 + wn * 128 + i * 16 + nl;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + wk * 64 + j * 16 + kq;
        if (n < p.N && k < p.K) *(f32x4*)(cs + (size_t)n * p.K + k) = acc[i][j];
      }
    }
    return;
  }
  // ---- the splits of a tile add themselves up (round 5; was one reduce_slabs launch behind every weight gradient).  The grid is at
  // most one workgroup per CU (the launcher checks), so the `splits` workgroups of a tile are resident together: each stores its slab
  // write-through (sc1), draws an arrive ticket, waits until all of them have, and then sums ITS share of the tile's rows over the
  // slabs in slab order -- the order reduce_slabs_kernel used: bit-identical results, no atomics on data.  The last to leave puts the
  // two tickets back to zero for the next launch.
  {
    const __amdgpu_buffer_rsrc_t rc = vc_rsrc(cs, (long long)p.N * p.K * 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int n = n0 + wn * 128 + i * 16 + nl;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + wk * 64 + j * 16 + kq;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vc_u32x4, acc[i][j]), rc, (unsigned)(n * p.K + k) * 4u, 0, 16);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* arrive = p.cnt + 2 * tile;
  if (tid == 0) {
    __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p.splits) __builtin_amdgcn_s_sleep(16);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // the other slabs sit in memory; drop what this CU / XCD may hold of their lines
  }
  __syncthreads();
  {
    const int rp = (TBN + p.splits - 1) / p.splits;        // rows of the tile per split
    const int r0 = split * rp;
    const int r1 = r0 + rp < TBN ? r0 + rp : TBN;
    const size_t slab = (size_t)p.N * p.K;
    for (int r = r0 + w; r < r1; r += 8) {                 // a wave per row: 64 lanes x 4 floats = the tile's 256 columns
      const size_t off = (size_t)(n0 + r) * p.K + k0 + lane * 4;
      f32x4 a = p.accumulate ? *(const f32x4*)(p.out + off) : f32x4{0.f, 0.f, 0.f, 0.f};
      for (int sp = 0; sp < p.splits; ++sp) a += *(const f32x4*)(p.C + sp * slab + off);
      *(f32x4*)(p.out + off) = a;
    }
  }
  if (tid == 0) {
    const int old = __hip_atomic_fetch_add(arrive + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == p.splits - 1) {
      __hip_atomic_store(arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(arrive + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// column sums of a bf16 matrix (bias gradients): out[n] += sum_m y[m][n].  A block covers 256 columns (16-byte loads, 32
// lanes per row = one 512-byte segment) x 128 rows, 8 rows in flight per step; HBM-bound.
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ y, int ldy, int M, int N, int rows_per_block,
                                                     float* __restrict__ out) {
  __shared__ float red[8][256];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = blockIdx.x * 256 + tx * 8;
  const int r0 = blockIdx.y * rows_per_block;
  int r1 = r0 + rows_per_block;
  r1 = r1 < M ? r1 : M;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < N) {
#pragma unroll 4
    for (int r = r0 + ty; r < r1; r += 8) {
      const uint4 v = *(const uint4*)(y + (size_t)r * ldy + c);
      a[0] += __uint_as_float(v.x << 16); a[1] += __uint_as_float(v.x & 0xffff0000u);
      a[2] += __uint_as_float(v.y << 16); a[3] += __uint_as_float(v.y & 0xffff0000u);
      a[4] += __uint_as_float(v.z << 16); a[5] += __uint_as_float(v.z & 0xffff0000u);
      a[6] += __uint_as_float(v.w << 16); a[7] += __uint_as_float(v.w & 0xffff0000u);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ty][tx * 8 + e] = a[e];
  __syncthreads();
  const int i = threadIdx.x;
  if (blockIdx.x * 256 + i < N) {
    float sacc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) sacc += red[k][i];
    atomicAdd(out + blockIdx.x * 256 + i, sacc);
  }
}

}  // namespace

static int launch_tn(const void* Y, int ldy, const void* X, int ldx, float* C_slabs, float* out, int accumulate, int M, int N, int K,
                     int splits, void* stream) {
  VC_REQUIRE(Y && X && C_slabs && M > 0 && N > 0 && K > 0 && splits >= 1, "gemm_tn: bad arguments");
  VC_REQUIRE(N % 256 == 0 && K % 256 == 0, "gemm_tn: N=%d and K=%d must be multiples of 256", N, K);
  VC_REQUIRE(ldy % 8 == 0 && ldx % 8 == 0 && ((uintptr_t)Y & 15) == 0 && ((uintptr_t)X & 15) == 0, "gemm_tn: misaligned operands");
  static bf16_t* zeros[64] = {nullptr};
  static int* tickets[64] = {nullptr};
  constexpr int SLOTS = 64, SLOT_INTS = 512;       // ticket blocks handed out round-robin: launches in flight on other streams get their own
  static std::atomic<unsigned> seq{0};
  int dev = 0;
  VC_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, "gemm_tn: bad device");
  if (!zeros[dev]) {
    VC_REQUIRE(hipMalloc(&zeros[dev], 1024) == hipSuccess && hipMemset(zeros[dev], 0, 1024) == hipSuccess,
               "gemm_tn: zero page allocation failed");
    VC_REQUIRE(hipMalloc(&tickets[dev], SLOTS * SLOT_INTS * sizeof(int)) == hipSuccess &&
                   hipMemset(tickets[dev], 0, SLOTS * SLOT_INTS * sizeof(int)) == hipSuccess, "gemm_tn: ticket allocation failed");
  }
  constexpr int smem = 2 * STAGE_BYTES;
  VC_FUNC_SMEM(gemm_tn_kernel, smem);
  TnArgs p;
  p.Y = (const bf16_t*)Y; p.X = (const bf16_t*)X; p.C = C_slabs; p.zeros = zeros[dev];
  p.M = M; p.N = N; p.K = K; p.ldy = ldy; p.ldx = ldx;
  p.tiles_n = N / TBN; p.tiles_k = K / TBK; p.splits = splits;
  const int total_stages = (M + TBM - 1) / TBM;
  p.stages_per_split = (total_stages + splits - 1) / splits;
  p.out = nullptr; p.cnt = nullptr; p.accumulate = accumulate;
  if (out) {
    // the in-kernel sum waits for the other splits of its tile: every workgroup of the launch must be resident at once (one per CU:
    // the kernel owns 128 KiB of LDS)
    int cus = 0;
    VC_REQUIRE(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess, "gemm_tn: device query failed");
    VC_REQUIRE(p.tiles_n * p.tiles_k * splits <= cus && p.tiles_n * p.tiles_k * 2 <= SLOT_INTS && (size_t)N * K * 4 < (1ull << 31),
               "gemm_tn(sum): %d tiles x %d splits must fit the %d CUs at once", p.tiles_n * p.tiles_k, splits, cus);
    p.out = out;
    p.cnt = tickets[dev] + (seq.fetch_add(1, std::memory_order_relaxed) % SLOTS) * SLOT_INTS;
  }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(p.tiles_n * p.tiles_k * splits), dim3(512), smem, (hipStream_t)stream, p);
  VC_LAUNCH_CHECK("gemm_tn");
  return VITCAP_OK;
}

extern "C" int vitcap_gemm_tn(const void* Y, int ldy, const void* X, int ldx, float* C_slabs, int M, int N, int K, int splits,
                              void* stream) {
  return launch_tn(Y, ldy, X, ldx, C_slabs, nullptr, 0, M, N, K, splits, stream);
}

extern "C" int vitcap_gemm_tn_sum(const void* Y, int ldy, const void* X, int ldx, float* C_slabs, float* out, int accumulate, int M,
                                  int N, int K, int splits, void* stream) {
  VC_REQUIRE(out, "gemm_tn_sum: out is null");
  return launch_tn(Y, ldy, X, ldx, C_slabs, out, accumulate, M, N, K, splits, stream);
}

extern "C" int vitcap_colsum_bf16(const void* y, int ldy, int M, int N, float* out, void* stream) {
  VC_REQUIRE(y && out && M > 0 && N > 0 && N % 8 == 0 && ldy % 8 == 0 && ((uintptr_t)y & 15) == 0, "colsum: bad arguments");
  const int rows_per_block = 128;
  dim3 grid((N + 255) / 256, (M + rows_per_block - 1) / rows_per_block);
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y, ldy, M, N, rows_per_block, out);
  VC_LAUNCH_CHECK("colsum");
  return VITCAP_OK;
}
