// bf16 NT GEMM, 256 x 256 x 64 tiles, FOUR waves of 128 x 128 each (one wave per SIMD, 512 registers per lane):
//   C = act(A . W^T + bias) (+ residual),  A[M][K], W[N][K] both K-contiguous.
//
// Why this form (round 4; docs/LAB_r01_r04.md section 4.3): the 8-wave kernel of gemm.hip (two waves per SIMD, 128 x 64 per wave) issues
// 24 ds_read_b128 per 64 MFMAs and pays two s_barriers per 16 MFMAs to alternate its wave pairs; its instruction stream is 29 %
// MFMA.  With ONE wave per SIMD the 128 x 128 wave tile keeps its 256 accumulator registers in the AGPR half of the unified
// 512-entry file and both k-halves' fragments (2 x 16 x 4 VGPRs) in the VGPR half:
//   * 32 ds_read_b128 + 16 LDS-DMA pieces per 128 MFMAs (0.375 memory instructions per MFMA instead of 0.5),
//   * ONE s_barrier per k-tile (at its middle: the k-half-1 fragments of tile t have been read, tile t+1 has landed),
//   * the wave's own MFMAs cover its own loads: each memory instruction sits behind an MFMA that is busy for 16 cycles.
// hipcc cannot keep 256 loop-carried accumulators in place (it rotates them through VGPRs: 250 v_accvgpr moves per k-tile,
// measured in the ISA) and schedules the loads in clumps, so the main loop is written as ordered `asm volatile` statements
// (cdna guide section 5.7): MFMAs with the accumulator tied ("+a"), fragment reads as asm ds_read_b128 fenced by explicit
// lgkmcnt waits, LDS-DMA as `s_mov m0` + `buffer_load_dwordx4 ... offen lds` with per-piece scalar offsets (no per-lane address
// arithmetic in the loop at all).  The compiler only allocates registers.
//
// Same tile walk (XCD-contiguous chunks, column groups), same LDS image (128-byte rows, 16-byte chunks XOR-swizzled by row & 7)
// and the SAME summation order per output element as gemm.hip's 256 x 256 kernel (k-tiles ascending, two 32-deep MFMAs per
// tile, same operand roles), so results are bit-identical to it (tests/test_hip_ops.py::test_gemm_4wave_bit_identical).
#include "common.h"
#include "gemm_args.h"

#include <atomic>
#include <utility>

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4;

constexpr int A_BYTES = 256 * 128;        // one operand's k-tile: 256 rows x 64 bf16
constexpr int BUF_BYTES = 2 * A_BYTES;    // A tile + W tile
constexpr int EP_ROWB = 272;              // epilogue staging row: 64 fp32 + 16 B pad (conflict-free ds_write_b128)
constexpr int EP_WAVE = 128 * EP_ROWB;    // one wave's 128 x 64 fp32 patch
constexpr int SMEM_4W = 4 * EP_WAVE > 2 * BUF_BYTES ? 4 * EP_WAVE : 2 * BUF_BYTES;
constexpr int PATCH_BYTES = 16 * 512;    // register epilogue, fp32 path: one wave's m-tile (16 rows x 128 fp32) behind the k-tile buffers
constexpr int SMEM_4WP = 2 * BUF_BYTES + 4 * PATCH_BYTES;   // = 160 KiB, the whole LDS of a CU

__device__ __forceinline__ void mfma_acc(f32x4& c, const bf16x8& w, const bf16x8& a) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(a));
}
__device__ __forceinline__ void mfma_zero(f32x4& c, const bf16x8& w, const bf16x8& a) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(w), "v"(a));
}
template <int OFF>
__device__ __forceinline__ void ds_read16(bf16x8& d, uint32_t addr) {
#if defined(VC_LOOP_ABL) && (VC_LOOP_ABL & 2)     // probe ablation: no fragment reads in the loop (wrong results)
  asm volatile("" : "+v"(d));
  return;
#endif
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
// one LDS-DMA piece: 64 lanes x 16 B -> 1 KiB at the wave-uniform LDS byte address `lds`; global address = rsrc base + voff + soff
__device__ __forceinline__ void dma16(uint32_t lds, uint32_t voff, const i32x4& rsrc, uint32_t soff) {
#if defined(VC_LOOP_ABL) && (VC_LOOP_ABL & 1)     // probe ablation: no LDS-DMA (wrong results)
  if (soff != 0xffffffffu) return;
#endif
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// A-panel prefetch (template parameter PF of the persistent kernel): one dword per lane into an LDS scratch line -- pulls 128-byte lines of the NEXT tile's A panel
// into this XCD's L2 (no VGPR destination: nothing for the compiler to reuse while the load is in flight)
__device__ __forceinline__ void pf64(uint32_t lds, uint32_t voff, const i32x4& rsrc, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
#ifndef VC_4W_PF_AHEAD
#define VC_4W_PF_AHEAD 2        // k-tiles between the slice a DMA requests and the slice prefetched next to it
#endif

// probe builds only (tools/probes/g4w_probe.hip defines VC_4W_STAMP): wave 0 of every workgroup records the shader clock at four
// points into p.rowstat (reused as a uint64 buffer): start, main loop entry, main loop exit, end
#ifdef VC_4W_STAMP
#define STAMP(i_)                                                                                        \
  if (w == 0 && lane == 0 && p.rowstat) ((unsigned long long*)p.rowstat)[blockIdx.x * 4 + (i_)] = __builtin_readcyclecounter()
#define STAMP_AT(slot_, i_)                                                                              \
  if (w == 0 && lane == 0 && p.rowstat) ((unsigned long long*)p.rowstat)[(slot_) * 4 + (i_)] = __builtin_readcyclecounter()
#else
#define STAMP(i_)
#define STAMP_AT(slot_, i_)
#endif

struct Frags {
  bf16x8 a[2][8];   // [k-half][m-tile]
  bf16x8 w[2][8];   // [k-half][n-tile]
};

// addresses the main loop needs (all wave-uniform values live in SGPRs)
struct Loop {
  uint32_t a_rd[2], w_rd[2];      // per-lane LDS read addresses of k-half 0 / 1 in buffer 0 (row base + swizzled chunk)
  uint32_t lds_a, lds_w;          // wave's first DMA destination (buffer 0)
  uint32_t voff_a, voff_w;        // per-lane global byte offset inside a piece
  uint32_t soff_a, soff_w;        // wave's first piece (scalar byte offset)
  uint32_t pstep_a, pstep_w;      // 8 rows
  uint32_t voff_pf, lds_pf;       // A-panel prefetch: this lane's row / line of the next tile's panel (or out of range), 256 B of LDS scratch per wave
};
// buffer descriptors of one output tile's operand panels (base = the tile's first row, range-checked to the matrix end: rows past
// M / N read as out-of-range -- no fault, and their outputs are never stored)
struct Src {
  i32x4 ra, rw;
};
__device__ __forceinline__ Src make_src(const GemmArgs& p, int m0, int n0) {
#ifdef VC_4W_A_RESIDENT      // probe (wrong results): every tile reads one of 8 A panels (3 MB: L2 / Infinity-Cache resident) -- same stream, no HBM
  m0 = (m0 >> 8 & 7) << 8;
#endif
#ifdef VC_4W_W_RESIDENT      // probe (wrong results): every tile reads one of 2 W panels
  n0 = (n0 >> 8 & 1) << 8;
#endif
  const bf16_t* abase = p.A + (size_t)m0 * p.lda;
  const bf16_t* wbase = p.W + (size_t)n0 * p.ldw;
  const long long arem = ((long long)(p.M - m0 - 1) * p.lda + p.K) * 2, wrem = ((long long)(p.N - n0 - 1) * p.ldw + p.K) * 2;
  const uint32_t arec = arem > 0xffffffffll ? 0xffffffffu : (uint32_t)arem, wrec = wrem > 0xffffffffll ? 0xffffffffu : (uint32_t)wrem;
  const uint64_t ab = (uint64_t)abase, wb = (uint64_t)wbase;
  Src r;
  r.ra = i32x4{(int)(uint32_t)ab, (int)(uint32_t)(ab >> 32), (int)arec, 0x00020000};
  r.rw = i32x4{(int)(uint32_t)wb, (int)(uint32_t)(wb >> 32), (int)wrec, 0x00020000};
  return r;
}
template <int MI>
__device__ __forceinline__ Loop make_loop(const GemmArgs& p, uint32_t smem_base, int lane, int w) {
  Loop L;
  // LDS-DMA piece: 8 rows x 128 B; lane -> row lane >> 3, LDS chunk position lane & 7, which holds source chunk (lane & 7) ^ (row & 7)
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  L.voff_a = (uint32_t)(srow * p.lda * 2 + schunk * 16);
  L.voff_w = (uint32_t)(srow * p.ldw * 2 + schunk * 16);
#ifdef VC_4W_DMA64
  L.voff_a = (uint32_t)((lane >> 2) * p.lda * 2 + (lane & 3) * 16);
  L.voff_w = (uint32_t)((lane >> 2) * p.ldw * 2 + (lane & 3) * 16);
#endif
  const int a_row0 = w * (8 * MI), w_row0 = w * 64;
  L.soff_a = (uint32_t)(a_row0 * p.lda * 2);
  L.soff_w = (uint32_t)(w_row0 * p.ldw * 2);
  L.pstep_a = (uint32_t)(8 * p.lda * 2);
  L.pstep_w = (uint32_t)(8 * p.ldw * 2);
  L.lds_a = smem_base + a_row0 * 128;
  L.lds_w = smem_base + A_BYTES + w_row0 * 128;
  const int wm = w >> 1, wn = w & 1, frow = lane & 15, fk = lane >> 4;
  const uint32_t arow = smem_base + (wm * 16 * MI + frow) * 128, wrow = smem_base + A_BYTES + (wn * 128 + frow) * 128;
  L.voff_pf = 0xfffffff0u;
  L.lds_pf = 0;
  L.a_rd[0] = arow + ((0 * 4 + fk) ^ (frow & 7)) * 16;
  L.a_rd[1] = arow + ((1 * 4 + fk) ^ (frow & 7)) * 16;
  L.w_rd[0] = wrow + ((0 * 4 + fk) ^ (frow & 7)) * 16;
  L.w_rd[1] = wrow + ((1 * 4 + fk) ^ (frow & 7)) * 16;
  return L;
}

template <int MI, int Q>
__device__ __forceinline__ void dma_piece(const Loop& L, const Src& src, uint32_t bufoff, uint32_t kb) {
  // piece Q of a k-tile: Q < MI -> 8 rows of A, else 8 rows of W
#ifdef VC_4W_DMA64     // probe (wrong results): every piece as 16 rows x 64 B (half lines) instead of 8 rows x 128 B -- same pieces, same bytes
  if constexpr (Q < MI) dma16(L.lds_a + bufoff + Q * 1024, L.voff_a, src.ra, L.soff_a + (Q / 2) * 2 * L.pstep_a + (Q & 1) * 64 + kb);
  else dma16(L.lds_w + bufoff + (Q - MI) * 1024, L.voff_w, src.rw, L.soff_w + ((Q - MI) / 2) * 2 * L.pstep_w + (Q & 1) * 64 + kb);
  return;
#endif
  if constexpr (Q < MI) dma16(L.lds_a + bufoff + Q * 1024, L.voff_a, src.ra, L.soff_a + Q * L.pstep_a + kb);
  else dma16(L.lds_w + bufoff + (Q - MI) * 1024, L.voff_w, src.rw, L.soff_w + (Q - MI) * L.pstep_w + kb);
}
template <int MI, int... Q>
__device__ __forceinline__ void dma_tile(const Loop& L, const Src& src, uint32_t bufoff, uint32_t kb, std::integer_sequence<int, Q...>) {
  (dma_piece<MI, Q>(L, src, bufoff, kb), ...);
}

template <int MI, int H, int R>
__device__ __forceinline__ void read_frag(Frags& f, uint32_t a_addr, uint32_t w_addr) {
  // fragment R of k-half H: W fragments first (the first MFMAs of a half need all eight), then the A fragments
  if constexpr (R < 8) ds_read16<R * 2048>(f.w[H][R], w_addr);
  else ds_read16<(R - 8) * 2048>(f.a[H][R - 8], a_addr);
}
template <int MI, int H, int... R>
__device__ __forceinline__ void read_frags(Frags& f, uint32_t a_addr, uint32_t w_addr, std::integer_sequence<int, R...>) {
  (read_frag<MI, H, R>(f, a_addr, w_addr), ...);
}

// One MFMA step S of a half (m-tile S / 8, n-tile S % 8) and the memory instructions scheduled behind it.
//   MODE 0: k-half 0 of a tile: reads the k-half-1 fragments of the same tile (buffer `cur`)
//   MODE 1: k-half 1: DMA of tile t+2 into `cur` (free since the mid-tile barrier) + k-half-0 fragments of tile t+1
//   MODE 2: k-half 1, no DMA (second-to-last tile)     MODE 3: k-half 1, nothing (last tile)
//   MODE 4: k-half 1 with the DMA but without fragment reads (persistent kernel, last k-tile of an output tile: the next tile's
//           fragments are read after the epilogue, so that no fragment register is live across it)
template <int MI, int MODE, bool FIRST, int S>
__device__ __forceinline__ void half_step(f32x4 (&acc)[MI][8], Frags& f, const Loop& L, const Src& src, uint32_t rd_a, uint32_t rd_w,
                                          uint32_t bufoff, uint32_t kb) {
  constexpr int H = MODE == 0 ? 0 : 1;
  constexpr int STEPS = MI * 8, NOPS = MI + 8;
  constexpr int STRIDE = (STEPS * 3 / 4) / NOPS > 0 ? (STEPS * 3 / 4) / NOPS : 1;
  if constexpr (FIRST) mfma_zero(acc[S / 8][S % 8], f.w[H][S % 8], f.a[H][S / 8]);
  else mfma_acc(acc[S / 8][S % 8], f.w[H][S % 8], f.a[H][S / 8]);
  if constexpr (MODE == 1 || MODE == 4) {
#ifdef VC_4W_DMA_DENSE          // probe: the tile's pieces behind the first MFMAs of the half (one per VC_4W_DMA_DENSE MFMAs) instead of spread over it
    if constexpr (S % VC_4W_DMA_DENSE == 0 && S / VC_4W_DMA_DENSE < NOPS) dma_piece<MI, S / VC_4W_DMA_DENSE>(L, src, bufoff, kb);
#else
    if constexpr (S % STRIDE == 0 && S / STRIDE < NOPS) dma_piece<MI, S / STRIDE>(L, src, bufoff, kb);
#endif
  }
  if constexpr (MODE != 3 && MODE != 4) {
    constexpr int PH = STRIDE > 1 ? 1 : 0;
    if constexpr (S % STRIDE == PH && S / STRIDE < NOPS) {
      read_frag<MI, 1 - H, S / STRIDE>(f, rd_a, rd_w);
    }
  }
}
template <int MI, int MODE, bool FIRST, int... S>
__device__ __forceinline__ void half_steps(f32x4 (&acc)[MI][8], Frags& f, const Loop& L, const Src& src, uint32_t rd_a, uint32_t rd_w, uint32_t bufoff,
                                           uint32_t kb, std::integer_sequence<int, S...>) {
  (half_step<MI, MODE, FIRST, S>(acc, f, L, src, rd_a, rd_w, bufoff, kb), ...);
}

// ---- schedule E ("early release", round 5).  With ONE rendezvous per k-tile the buffer of tile t is released at the tile's middle and
// tile t+2's pieces, requested behind k-half 1's MFMAs, must have landed one k-tile later: a lead of 1.25-2 k-halves (1.3-2.0k cycles),
// enough for operands that sit in the L2 / Infinity Cache (M = 36 928: the loop runs at 25.4k cycles per 12 k-tiles, the MFMA floor is
// 24.6k), NOT for an A panel that streams from HBM under load (M = 295 424: 35.9k, profiles/r04_g4w_probe.txt).  Here the k-half-1
// fragments of tile t are read behind the FIRST half of k-half 0's MFMAs, a first rendezvous (X) releases the buffer there, and the
// pieces of tile t+2 follow behind the second half of k-half 0: requested half a k-tile earlier, still awaited at the mid-tile
// rendezvous (Y) of tile t+1 with a COUNTED wait (the pieces requested in this tile stay in flight across it): lead 2.0-2.5 k-halves.
// Same MFMA order, same operands: bit-identical results.
constexpr int slot_of(int S, int N, int Q) {       // index r in [0, N) whose position (r * Q) / N is step S (Q >= N), else -1
  const int r = (S * N + Q - 1) / Q;
  return (S >= 0 && r < N && (r * Q) / N == S) ? r : -1;
}
//   H0: k-half 0 of tile t: reads of the tile's k-half-1 fragments in steps [0, QA), rendezvous X after step QA - 1 (DMA only), pieces
//       of tile t+2 in steps [QA, STEPS)
template <int MI, bool DMA, bool FIRST, int S>
__device__ __forceinline__ void half0_step_e(f32x4 (&acc)[MI][8], Frags& f, const Loop& L, const Src& src, uint32_t rd_a, uint32_t rd_w,
                                             uint32_t bufoff, uint32_t kb) {
  constexpr int STEPS = MI * 8, NOPS = MI + 8, QA = STEPS / 2, QB = STEPS - QA;
  if constexpr (FIRST) mfma_zero(acc[S / 8][S % 8], f.w[0][S % 8], f.a[0][S / 8]);
  else mfma_acc(acc[S / 8][S % 8], f.w[0][S % 8], f.a[0][S / 8]);
  if constexpr (S < QA) {
    constexpr int R = slot_of(S, NOPS, QA);
    if constexpr (R >= 0) read_frag<MI, 1, R>(f, rd_a, rd_w);
    if constexpr (S == QA - 1 && DMA) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  } else if constexpr (DMA) {
    constexpr int Q = slot_of(S - QA, NOPS, QB);
    if constexpr (Q >= 0) dma_piece<MI, Q>(L, src, bufoff, kb);
  }
}
template <int MI, bool DMA, bool FIRST, int... S>
__device__ __forceinline__ void half0_steps_e(f32x4 (&acc)[MI][8], Frags& f, const Loop& L, const Src& src, uint32_t rd_a, uint32_t rd_w,
                                              uint32_t bufoff, uint32_t kb, std::integer_sequence<int, S...>) {
  (half0_step_e<MI, DMA, FIRST, S>(acc, f, L, src, rd_a, rd_w, bufoff, kb), ...);
}
// MODE1 as in k_tile: 1 = pieces + next tile's k-half-0 fragments, 2 = fragments only, 3 = neither, 4 = pieces only
template <int MI, int MODE1, bool FIRST>
__device__ __forceinline__ void k_tile_e(f32x4 (&acc)[MI][8], Frags& f, const Loop& L, const Src& src, uint32_t cur, uint32_t kb2) {
  const uint32_t nxt = BUF_BYTES - cur;
  constexpr bool DMA = MODE1 == 1 || MODE1 == 4;
  half0_steps_e<MI, DMA, FIRST>(acc, f, L, src, L.a_rd[1] + cur, L.w_rd[1] + cur, cur, kb2, std::make_integer_sequence<int, MI * 8>{});
  // Y: tile t+1 has landed (this wave's pieces: everything older than the MI + 8 pieces just requested), then every wave's
  if constexpr (DMA) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(MI + 8) : "memory");
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  half_steps<MI, MODE1 == 3 || MODE1 == 4 ? 3 : 2, false>(acc, f, L, src, L.a_rd[0] + nxt, L.w_rd[0] + nxt, cur, kb2, std::make_integer_sequence<int, MI * 8>{});
  if constexpr (MODE1 != 3 && MODE1 != 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// One k-tile t (cur = its buffer's byte offset): k-half 0, the mid-tile rendezvous, k-half 1 (whose DMA, MODE1 == 1, requests the
// k-tile at byte offset kb2 of the panels `src` -- two k-tiles ahead in the stream, possibly the NEXT output tile's -- into `cur`).
template <int MI, int MODE1, bool FIRST, bool PF = false>
__device__ __forceinline__ void k_tile(f32x4 (&acc)[MI][8], Frags& f, const Loop& L, const Src& src, uint32_t cur, uint32_t kb2,
                                       const Src& pf, uint32_t pf_soff);
template <int MI, int MODE1, bool FIRST>
__device__ __forceinline__ void k_tile(f32x4 (&acc)[MI][8], Frags& f, const Loop& L, const Src& src, uint32_t cur, uint32_t kb2) {
  k_tile<MI, MODE1, FIRST, false>(acc, f, L, src, cur, kb2, src, 0xfffffff0u);
}
// PF (persistent bf16-output kernel at M >= 64k rows, round 6): every k-tile issues exactly ONE more memory instruction behind its DMA
// pieces -- whole rows of the NEXT output tile's A panel into a scratch line (lanes without a row are out of range: no fetch) -- and
// the mid-tile wait counts it (vmcnt(1): the prefetch stays in flight across the rendezvous).
template <int MI, int MODE1, bool FIRST, bool PF>
__device__ __forceinline__ void k_tile(f32x4 (&acc)[MI][8], Frags& f, const Loop& L, const Src& src, uint32_t cur, uint32_t kb2,
                                       const Src& pf, uint32_t pf_soff) {
#ifdef VC_4W_EARLY
  k_tile_e<MI, MODE1, FIRST>(acc, f, L, src, cur, kb2);
  return;
#endif
  const uint32_t nxt = BUF_BYTES - cur;
  half_steps<MI, 0, FIRST>(acc, f, L, src, L.a_rd[1] + cur, L.w_rd[1] + cur, 0, 0, std::make_integer_sequence<int, MI * 8>{});
  // every wave has read the whole of buffer `cur` (lgkmcnt) and its own pieces of tile t+1 have landed (vmcnt): after the barrier
  // `cur` may be overwritten and buffer `nxt` may be read
#if defined(VC_LOOP_ABL) && (VC_LOOP_ABL & 4)     // probe ablation: no mid-tile barrier (wrong results)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#else
  if constexpr (PF) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
  half_steps<MI, MODE1, false>(acc, f, L, src, L.a_rd[0] + nxt, L.w_rd[0] + nxt, cur, kb2, std::make_integer_sequence<int, MI * 8>{});
  if constexpr (PF) pf64(L.lds_pf, L.voff_pf, pf.ra, pf_soff);
  if constexpr (MODE1 != 3 && MODE1 != 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// The accumulators' last writers are asm MFMAs whose result latency hipcc does not know: a v_accvgpr_read it schedules right behind
// the asm statement that last wrote a register (long before the final s_nop) returns the rows of the MFMA's last pass stale -- seen
// as NaN / old sums in lanes 12-15 and 44-47 of two n-tiles.  After the final MFMA: 24 wait states, then every accumulator passes
// through an (empty) asm statement, so that each compiler read is ordered behind the wait.
template <int MI>
__device__ __forceinline__ void fence_accumulators(f32x4 (&acc)[MI][8]) {
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
  for (int i = 0; i < MI; ++i)
    asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]), "+a"(acc[i][6]), "+a"(acc[i][7]));
}

#define ROWS_OF(m_, orow_, rrow_)                                             \
  int orow_ = (m_), rrow_ = (m_);                                             \
  if (p.row_group > 0) {                                                      \
    const int g_ = (m_) / p.row_group, in_ = (m_) - g_ * p.row_group;         \
    orow_ = g_ * p.out_group_rows + p.out_row_off + in_;                      \
    rrow_ = p.res_periodic ? in_ : orow_;                                     \
  }

typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

// ---- epilogue straight from the accumulators (no LDS: in the persistent kernel the k-tile buffers already hold the next tile's
// first k-tiles).  After the operand-swapped MFMAs a lane (frow = lane & 15, fk = lane >> 4) holds, of m-tile i and n-tile j, the four
// columns j*16 + fk*4 .. +3 of row i*16 + frow:
//   fp32 output: that is one 16-byte store; the four lane groups cover 64 contiguous bytes of the row, the residual arrives by the
//                same pattern, requested one m-tile (8 loads) ahead;
//   bf16 output: two n-tiles (je, jo = je + 1) are packed and exchanged between neighbouring lane groups (v_permlane16_swap: group
//                fk = 1 / 3 gives its je values to group 0 / 2 and takes their jo values), after which group 0 / 2 holds columns
//                0..7 / 8..15 of tile je and group 1 / 3 those of tile jo: one 16-byte store per lane, 64 contiguous bytes per row.
// bias -> activation -> residual -> rounding in gemm.hip's order on the same values: bit-identical outputs.
template <int ACT, int OUT_F32, bool HAS_RES, int MI>
__device__ __forceinline__ void epilogue_regs(f32x4 (&acc)[MI][8], const GemmArgs& p, const int row_w, const int col_w, const int lane) {
  const int frow = lane & 15, fk = lane >> 4;
  f32x4 bias4[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int n = col_w + j * 16 + fk * 4;
    bias4[j] = (p.bias && n < p.N) ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 rres[2][8];
#define ISSUE_RES_R(i_)                                                                                            \
  if (HAS_RES) {                                                                                                   \
    const int m_ = row_w + (i_) * 16 + frow;                                                                       \
    ROWS_OF(m_, o_, r_);                                                                                           \
    (void)o_;                                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                \
      const int n_ = col_w + j * 16 + fk * 4;                                                                      \
      rres[(i_) & 1][j] = (m_ < p.M && n_ < p.N) ? *(const f32x4*)(p.res + (size_t)r_ * p.ldr + n_)                \
                                                 : f32x4{0.f, 0.f, 0.f, 0.f};                                      \
    }                                                                                                              \
  }
  ISSUE_RES_R(0);
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    if (i + 1 < MI) { ISSUE_RES_R(i + 1); }
    const int m = row_w + i * 16 + frow;
    ROWS_OF(m, orow, rrow);
    (void)rrow;
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = acc[i][j] + bias4[j];
      if (ACT == VITCAP_ACT_GELU_ERF) v[j] = gelu_erf4(v[j]);
      if (HAS_RES) v[j] += rres[i & 1][j];
    }
    if constexpr (OUT_F32) {
      float* dst = (float*)p.C + (size_t)orow * p.ldc + col_w + fk * 4;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (m < p.M && col_w + j * 16 + fk * 4 < p.N) *(f32x4*)(dst + j * 16) = v[j];
    } else {
      bf16_t* dst = (bf16_t*)p.C + (size_t)orow * p.ldc + col_w + (fk & 1) * 16 + (fk >> 1) * 8;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const u32x2_t s0 = __builtin_amdgcn_permlane16_swap(pack2bf(v[2 * jj][0], v[2 * jj][1]), pack2bf(v[2 * jj + 1][0], v[2 * jj + 1][1]), false, false);
        const u32x2_t s1 = __builtin_amdgcn_permlane16_swap(pack2bf(v[2 * jj][2], v[2 * jj][3]), pack2bf(v[2 * jj + 1][2], v[2 * jj + 1][3]), false, false);
        uint4 o;
        o.x = s0[0]; o.y = s1[0]; o.z = s0[1]; o.w = s1[1];
        if (m < p.M && col_w + jj * 32 + (fk & 1) * 16 + (fk >> 1) * 8 < p.N) *(uint4*)(dst + jj * 32) = o;
      }
    }
  }
#undef ISSUE_RES_R
}

// ---- the persistent kernel's epilogues (plain rows, N a multiple of 256).  With ONE wave per SIMD every instruction of the
// epilogue costs its full issue time (nothing else runs on the SIMD): the general forms above spend 9.8k cycles on a 128 x 128 bf16
// wave tile (stamped), two thirds of it on per-store predicates (compare, exec save / branch / restore) and 64-bit address
// arithmetic.  Here rows and columns are addressed through a buffer descriptor whose range ends behind row M - 1 (rows past M are
// dropped by the range check, no predicate), one 32-bit per-lane offset serves the whole tile, the m-tile's row offset rides in
// the scalar offset and the n-tile's in the immediate.  Same arithmetic per element.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
#ifndef VC_4W_STORE_AUX
#define VC_4W_STORE_AUX 0      // cache policy of the output stores (2 = nt)
#endif
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const void* base, long long bytes) {
  const unsigned rec = bytes <= 0 ? 0u : (bytes > 0xffffffffll ? 0xffffffffu : (unsigned)bytes);
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, rec, 0x00020000);
}

template <int ACT, int MI>
__device__ __forceinline__ void epilogue_regs_fast(f32x4 (&acc)[MI][8], const GemmArgs& p, const int row_w, const int col_w, const int lane) {
  const int frow = lane & 15, fk = lane >> 4;
  f32x4 bias4[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bias4[j] = p.bias ? *(const f32x4*)(p.bias + col_w + j * 16 + fk * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  const __amdgpu_buffer_rsrc_t rc = rows_rsrc((const bf16_t*)p.C + (size_t)row_w * p.ldc, (long long)(p.M - row_w) * p.ldc * 2);
  // The m-tile's row offset rides in the VECTOR offset, the scalar offset stays the constant 0: with a REGISTER soffset hipcc omits
  // the wait state between a 16-byte buffer store and a VALU write to its data registers (LLVM's hazard table says the hazard only
  // exists without one) -- on gfx950 it exists all the same: the next v_pk_add overwrote the store's last data dword in lanes
  // 12-15 / 44-47 (found as 32 wrong elements per wave tile, always the same ones).
  unsigned voff = (unsigned)(frow * p.ldc + col_w + (fk & 1) * 16 + (fk >> 1) * 8) * 2u;
  const unsigned istep = 16u * p.ldc * 2u;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      f32x4 ve = acc[i][2 * jj] + bias4[2 * jj], vo = acc[i][2 * jj + 1] + bias4[2 * jj + 1];
      if (ACT == VITCAP_ACT_GELU_ERF) {
        ve = gelu_erf4(ve);
        vo = gelu_erf4(vo);
      }
#if defined(VC_EPI_ABL) && (VC_EPI_ABL & 2)       // probe ablation: no lane exchange (wrong results)
      const u32x2_t s0 = u32x2_t{pack2bf(ve[0], ve[1]), pack2bf(vo[0], vo[1])};
      const u32x2_t s1 = u32x2_t{pack2bf(ve[2], ve[3]), pack2bf(vo[2], vo[3])};
#else
      const u32x2_t s0 = __builtin_amdgcn_permlane16_swap(pack2bf(ve[0], ve[1]), pack2bf(vo[0], vo[1]), false, false);
      const u32x2_t s1 = __builtin_amdgcn_permlane16_swap(pack2bf(ve[2], ve[3]), pack2bf(vo[2], vo[3]), false, false);
#endif
#if defined(VC_EPI_ABL) && (VC_EPI_ABL & 1)       // probe ablation: values kept alive, no store
      asm volatile("" ::"v"(s0), "v"(s1));
#else
      __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{s0[0], s1[0], s0[1], s1[1]}, rc, voff + jj * 64, 0, VC_4W_STORE_AUX);
#endif
    }
    voff += istep;
  }
}

// ---- the same epilogue with the TRAINING extras of vitcap_gemm_ex (round 5; gemm.hip's 8-wave kernel carried them alone until now:
// 16 % of a training step ran on its one-tile form only because of them):
//   aux    : out = (acc + bias) * aux[m][n]            (the stored gelu' factor: fc2's input gradient)
//   zout   : zout[m][n] = gelu'(acc + bias)            (next to out = gelu(acc + bias): fc1's forward; one erfc evaluation for both)
//   colsum : colsum[n] += sum over the tile's valid rows of the ROUNDED outputs (the bias gradient of the layer `out` is the output
//            gradient of; what vitcap_colsum_bf16 over the stored output would add)
// Same arithmetic per element and the same order as gemm.hip's epilogue: bit-identical `out` / `zout`.  aux arrives in the MFMA layout
// (8 bytes per lane and n-tile: the 4 columns the lane holds), requested one m-tile ahead.
template <int ACT, int MI>
__device__ __forceinline__ void epilogue_regs_fast_x(f32x4 (&acc)[MI][8], const GemmArgs& p, const int row_w, const int col_w, const int lane) {
  const int frow = lane & 15, fk = lane >> 4;
  f32x4 bias4[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bias4[j] = p.bias ? *(const f32x4*)(p.bias + col_w + j * 16 + fk * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  const __amdgpu_buffer_rsrc_t rc = rows_rsrc((const bf16_t*)p.C + (size_t)row_w * p.ldc, (long long)(p.M - row_w) * p.ldc * 2);
  const __amdgpu_buffer_rsrc_t rz = rows_rsrc(p.zout ? p.zout + (size_t)row_w * p.ldz : nullptr, p.zout ? (long long)(p.M - row_w) * p.ldz * 2 : 0);
  const __amdgpu_buffer_rsrc_t ra = rows_rsrc(p.aux ? p.aux + (size_t)row_w * p.ldaux : nullptr, p.aux ? (long long)(p.M - row_w) * p.ldaux * 2 : 0);
  unsigned voff = (unsigned)(frow * p.ldc + col_w + (fk & 1) * 16 + (fk >> 1) * 8) * 2u;
  unsigned voff_z = (unsigned)(frow * p.ldz + col_w + (fk & 1) * 16 + (fk >> 1) * 8) * 2u;
  unsigned voff_a = (unsigned)(frow * p.ldaux + col_w + fk * 4) * 2u;
  const unsigned istep = 16u * p.ldc * 2u, istep_z = 16u * p.ldz * 2u, istep_a = 16u * p.ldaux * 2u;
  const bool has_aux = p.aux != nullptr, has_z = p.zout != nullptr, has_cs = p.colsum != nullptr;
  u32x2_t ax[2][8];
#define ISSUE_AUX(i_)                                                                                              \
  if (has_aux) {                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                                  \
      ax[(i_) & 1][j] = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(ra, voff_a + j * 32, 0, 0)); \
    voff_a += istep_a;                                                                                             \
  }
  ISSUE_AUX(0);
  float cs[4][8];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj)
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[jj][e] = 0.f;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    if (i + 1 < MI) { ISSUE_AUX(i + 1); }
    const bool row_ok = row_w + i * 16 + frow < p.M;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      f32x4 ve = acc[i][2 * jj] + bias4[2 * jj], vo = acc[i][2 * jj + 1] + bias4[2 * jj + 1];
      if (has_aux) {
        const u32x2_t ae = ax[i & 1][2 * jj], ao = ax[i & 1][2 * jj + 1];
        ve *= f32x4{__uint_as_float(ae[0] << 16), __uint_as_float(ae[0] & 0xffff0000u), __uint_as_float(ae[1] << 16), __uint_as_float(ae[1] & 0xffff0000u)};
        vo *= f32x4{__uint_as_float(ao[0] << 16), __uint_as_float(ao[0] & 0xffff0000u), __uint_as_float(ao[1] << 16), __uint_as_float(ao[1] & 0xffff0000u)};
      }
      if (ACT == VITCAP_ACT_GELU_ERF) {
        if (has_z) {
          f32x4 de, dn;
          ve = gelu_erf4_grad(ve, de);
          vo = gelu_erf4_grad(vo, dn);
          const u32x2_t z0 = __builtin_amdgcn_permlane16_swap(pack2bf(de[0], de[1]), pack2bf(dn[0], dn[1]), false, false);
          const u32x2_t z1 = __builtin_amdgcn_permlane16_swap(pack2bf(de[2], de[3]), pack2bf(dn[2], dn[3]), false, false);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{z0[0], z1[0], z0[1], z1[1]}, rz, voff_z + jj * 64, 0, VC_4W_STORE_AUX);
        } else {
          ve = gelu_erf4(ve);
          vo = gelu_erf4(vo);
        }
      }
      const u32x2_t s0 = __builtin_amdgcn_permlane16_swap(pack2bf(ve[0], ve[1]), pack2bf(vo[0], vo[1]), false, false);
      const u32x2_t s1 = __builtin_amdgcn_permlane16_swap(pack2bf(ve[2], ve[3]), pack2bf(vo[2], vo[3]), false, false);
      const u32x4_t o = u32x4_t{s0[0], s1[0], s0[1], s1[1]};
      __builtin_amdgcn_raw_buffer_store_b128(o, rc, voff + jj * 64, 0, VC_4W_STORE_AUX);
      if (has_cs && row_ok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          cs[jj][2 * q] += __uint_as_float(o[q] << 16);
          cs[jj][2 * q + 1] += __uint_as_float(o[q] & 0xffff0000u);
        }
      }
    }
    voff += istep;
    voff_z += istep_z;
  }
#undef ISSUE_AUX
  if (has_cs) {
    // the 16 lanes frow = 0..15 of a lane group hold the same 32 columns: butterfly over lane bits 0..3, one atomic per column and group
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = cs[jj][e];
#pragma unroll
        for (int o = 1; o <= 8; o <<= 1) v += __shfl_xor(v, o, 64);
        if (frow == 0) atomicAdd(p.colsum + col_w + jj * 32 + (fk & 1) * 16 + (fk >> 1) * 8 + e, v);
      }
  }
}

// ---- deferred stores (round 6, DEFER forms of the persistent kernel).  The bf16 epilogue spends 6k of its 8.7k (qkv) / 15.5k (fc1)
// cycles at the ISSUE of its 32 stores per wave (without the stores the same epilogue takes 4.4k and the kernel 140 -> 116 us,
// docs/LAB_r01_r04.md 4.3).  Here the upper half of the wave tile's m-tiles is converted, packed and PARKED in VGPRs (4 x 16 B per
// m-tile and lane) instead of being stored, and its stores are issued four per k-tile in the NEXT tile's first k-tiles: same values,
// same addresses, same order per address -- bit-identical outputs (tests/test_hip_ops.py::test_gemm_4wave_deferred_stores_bit_identical).
// A workgroup's first tile finds a descriptor of zero records (stores dropped by the range check), its last tile flushes its own
// parked half.  What it buys is small (see the launcher's policy note): with one wave per SIMD a store's issue stalls the wave -- and
// the matrix pipe with it -- wherever in the stream it stands.
template <int MI>
struct Park {
  static constexpr int NP = MI / 2;          // parked m-tiles (the upper ones)
  u32x4_t v[NP * 4];
  unsigned voff[NP];
};
// (the buffer descriptor of the parked stores travels beside the struct, in a variable of its own: inside it hipcc hands the asm
// statement a VGPR quad for the "s" operand)
// descriptor of the parked tile's rows, rebuilt where it is used from ONE carried integer (the wave tile's first row, -1 = nothing
// parked: zero records, every store dropped): a descriptor carried across the tile loop ends up in VGPRs (the kernel runs out of
// SGPRs) and cannot be an "s" operand
__device__ __forceinline__ i32x4 park_rsrc(const GemmArgs& p, int park_row) {
  const int row = __builtin_amdgcn_readfirstlane(park_row);
  const uint64_t cb = (uint64_t)((const bf16_t*)p.C + (size_t)(row < 0 ? 0 : row) * p.ldc);
  const long long cbytes = row < 0 ? 0 : (long long)(p.M - row) * p.ldc * 2;
  const unsigned rec = cbytes <= 0 ? 0u : (cbytes > 0xffffffffll ? 0xffffffffu : (unsigned)cbytes);
  return i32x4{__builtin_amdgcn_readfirstlane((int)(uint32_t)cb), __builtin_amdgcn_readfirstlane((int)(uint32_t)(cb >> 32)),
               __builtin_amdgcn_readfirstlane((int)rec), 0x00020000};
}
template <int MI, int N>
__device__ __forceinline__ void store_parked(const Park<MI>& pk, const i32x4& prc) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen offset:%3" ::"v"(pk.v[N]), "v"(pk.voff[N / 4]), "s"(prc), "n"((N % 4) * 64) : "memory");
}
template <int MI, int... N>
__device__ __forceinline__ void store_parked_all(const Park<MI>& pk, const i32x4& prc, std::integer_sequence<int, N...>) {
  (store_parked<MI, N>(pk, prc), ...);
}

template <int ACT, int MI>
__device__ __forceinline__ void epilogue_regs_fast_defer(f32x4 (&acc)[MI][8], const GemmArgs& p, const int row_w, const int col_w, const int lane,
                                                         Park<MI>& pk) {
  constexpr int NP = Park<MI>::NP, NI = MI - NP;
  const int frow = lane & 15, fk = lane >> 4;
  f32x4 bias4[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bias4[j] = p.bias ? *(const f32x4*)(p.bias + col_w + j * 16 + fk * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16_t* cbase = (const bf16_t*)p.C + (size_t)row_w * p.ldc;
  const long long cbytes = (long long)(p.M - row_w) * p.ldc * 2;
  const __amdgpu_buffer_rsrc_t rc = rows_rsrc(cbase, cbytes);
  unsigned voff = (unsigned)(frow * p.ldc + col_w + (fk & 1) * 16 + (fk >> 1) * 8) * 2u;
  const unsigned istep = 16u * p.ldc * 2u;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    if (i >= NI) pk.voff[i - NI] = voff;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      f32x4 ve = acc[i][2 * jj] + bias4[2 * jj], vo = acc[i][2 * jj + 1] + bias4[2 * jj + 1];
      if (ACT == VITCAP_ACT_GELU_ERF) {
        ve = gelu_erf4(ve);
        vo = gelu_erf4(vo);
      }
      const u32x2_t s0 = __builtin_amdgcn_permlane16_swap(pack2bf(ve[0], ve[1]), pack2bf(vo[0], vo[1]), false, false);
      const u32x2_t s1 = __builtin_amdgcn_permlane16_swap(pack2bf(ve[2], ve[3]), pack2bf(vo[2], vo[3]), false, false);
      const u32x4_t o = u32x4_t{s0[0], s1[0], s0[1], s1[1]};
      if (i < NI) __builtin_amdgcn_raw_buffer_store_b128(o, rc, voff + jj * 64, 0, VC_4W_STORE_AUX);
      else pk.v[(i - NI) * 4 + jj] = o;
    }
    voff += istep;
  }
}

// k-tile of the DEFER kernel that issues the parked stores ST0 .. ST0 + 3 of the PREVIOUS output tile behind its first half's MFMAs;
// the mid-tile wait leaves exactly those four in flight (gfx9 retires VMEM operations in order: everything older -- the pieces of
// k-tile t + 1 that this rendezvous is about -- has completed when only the four youngest are outstanding)
template <int MI, int MODE1, bool FIRST, int ST0>
__device__ __forceinline__ void k_tile_st(f32x4 (&acc)[MI][8], Frags& f, const Loop& L, const Src& src, uint32_t cur, uint32_t kb2, const Park<MI>& pk,
                                          const GemmArgs& p, int park_row) {
  const uint32_t nxt = BUF_BYTES - cur;
  const i32x4 prc = park_rsrc(p, park_row);
  half_steps<MI, 0, FIRST>(acc, f, L, src, L.a_rd[1] + cur, L.w_rd[1] + cur, 0, 0, std::make_integer_sequence<int, MI * 8>{});
  store_parked<MI, ST0>(pk, prc);
  store_parked<MI, ST0 + 1>(pk, prc);
  store_parked<MI, ST0 + 2>(pk, prc);
  store_parked<MI, ST0 + 3>(pk, prc);
  asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  half_steps<MI, MODE1, false>(acc, f, L, src, L.a_rd[0] + nxt, L.w_rd[0] + nxt, cur, kb2, std::make_integer_sequence<int, MI * 8>{});
  if constexpr (MODE1 != 3 && MODE1 != 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// fp32 output and / or residual: each m-tile through the wave's 8 KiB LDS patch (see epilogue_patch), loads and stores by buffer
// instructions: lane -> columns (lane & 31) * 4 .. +3 of row 2 * it + (lane >> 5), two rows of 512 contiguous bytes per instruction
template <int ACT, int OUT_F32, bool HAS_RES, int MI>
__device__ __forceinline__ void epilogue_patch_fast(f32x4 (&acc)[MI][8], const GemmArgs& p, char* patch, const int row_w, const int col_w,
                                                    const int lane) {
  const int frow = lane & 15, fk = lane >> 4;
  const int rr = lane >> 5, rcx = lane & 31;
  const f32x4 bias4 = p.bias ? *(const f32x4*)(p.bias + col_w + rcx * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int CB = OUT_F32 ? 4 : 2;
  const __amdgpu_buffer_rsrc_t rc = rows_rsrc((const char*)p.C + (size_t)row_w * p.ldc * CB, (long long)(p.M - row_w) * p.ldc * CB);
  const __amdgpu_buffer_rsrc_t rs = rows_rsrc(HAS_RES ? p.res + (size_t)row_w * p.ldr : nullptr, HAS_RES ? (long long)(p.M - row_w) * p.ldr * 4 : 0);
  unsigned voff_c = (unsigned)(rr * p.ldc + col_w + rcx * 4) * CB;     // stores: row steps in the vector offset, soffset constant (see epilogue_regs_fast)
  const unsigned voff_r = (unsigned)(rr * p.ldr + col_w + rcx * 4) * 4u;
  const unsigned cstep = 2u * p.ldc * CB, rstep = 2u * p.ldr * 4u;     // two rows
  char* wr = patch + frow * 512;
  f32x4 rres[2][8];
  unsigned so_r = 0;                       // loads: running scalar offset (two rows per step), opaque: not hoisted across tiles
  asm volatile("" : "+s"(so_r));
#define ISSUE_RES_F(i_)                                                                                                           \
  if (HAS_RES) {                                                                                                                  \
    _Pragma("unroll") for (int it = 0; it < 8; ++it) {                                                                            \
      rres[(i_) & 1][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff_r, so_r, 0));                 \
      so_r += rstep;                                                                                                              \
    }                                                                                                                             \
  }
  ISSUE_RES_F(0);
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    if (i + 1 < MI) { ISSUE_RES_F(i + 1); }
#pragma unroll
    for (int j = 0; j < 8; ++j) *(f32x4*)(wr + (((j * 4 + fk) ^ (frow & 7)) * 16)) = acc[i][j];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int rl = it * 2 + rr;
      f32x4 v = *(const f32x4*)(patch + rl * 512 + ((rcx ^ (rl & 7)) * 16));
      v += bias4;
      if (ACT == VITCAP_ACT_GELU_ERF) v = gelu_erf4(v);
      if (HAS_RES) v += rres[i & 1][it];
      if (OUT_F32) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rc, voff_c, 0, VC_4W_STORE_AUX);
      else __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])}, rc, voff_c, 0, 0);
      voff_c += cstep;
    }
  }
#undef ISSUE_RES_F
}

// ---- fp32 outputs and residual adds: 16-byte pieces of 16 different rows per store instruction (what the register layout gives:
// 64 contiguous bytes per row) cost 50k cycles per tile against 30-35k for row-major stores (stamped, proj / fc2) -- the output and
// residual streams want whole rows.  Each wave therefore turns ONE m-tile (16 rows x 128 columns fp32 = 8 KiB) at a time through a
// private patch in the 32 KiB of LDS behind the two k-tile buffers (which, in the persistent kernel, already hold the next tile's
// first k-tiles): written in the MFMA layout (16-byte chunks XOR-swizzled by row & 7: conflict-free both ways), read back row-major
// -- a lane gets columns (lane & 31) * 4 .. +3 of row 2 * it + (lane >> 5) -- so every residual load and every store covers two rows
// of 512 contiguous bytes.  Wave-private: LDS operations of one wave execute in order, no barrier.  Same arithmetic per element.
template <int ACT, int OUT_F32, bool HAS_RES, int MI>
__device__ __forceinline__ void epilogue_patch(f32x4 (&acc)[MI][8], const GemmArgs& p, char* patch, const int row_w, const int col_w, const int lane) {
  const int frow = lane & 15, fk = lane >> 4;
  const int rr = lane >> 5, rc = lane & 31;           // read-back: row within a pair, 16-byte chunk of the row
  const int ncol = col_w + rc * 4;
  const bool col_ok = ncol < p.N;
  f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
  if (p.bias && col_ok) bias4 = *(const f32x4*)(p.bias + ncol);
  char* wr = patch + frow * 512;
  f32x4 rres[2][8];
#define ISSUE_RES_P(i_)                                                                                            \
  if (HAS_RES) {                                                                                                   \
    _Pragma("unroll") for (int it = 0; it < 8; ++it) {                                                             \
      const int m_ = row_w + (i_) * 16 + it * 2 + rr;                                                              \
      ROWS_OF(m_, o_, r_);                                                                                         \
      (void)o_;                                                                                                    \
      rres[(i_) & 1][it] = (m_ < p.M && col_ok) ? *(const f32x4*)(p.res + (size_t)r_ * p.ldr + ncol)               \
                                                : f32x4{0.f, 0.f, 0.f, 0.f};                                       \
    }                                                                                                              \
  }
  ISSUE_RES_P(0);
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    if (i + 1 < MI) { ISSUE_RES_P(i + 1); }
#pragma unroll
    for (int j = 0; j < 8; ++j) *(f32x4*)(wr + (((j * 4 + fk) ^ (frow & 7)) * 16)) = acc[i][j];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int rl = it * 2 + rr;
      f32x4 v = *(const f32x4*)(patch + rl * 512 + ((rc ^ (rl & 7)) * 16));
      const int m = row_w + i * 16 + rl;
      ROWS_OF(m, orow, rrow);
      (void)rrow;
      v += bias4;
      if (ACT == VITCAP_ACT_GELU_ERF) v = gelu_erf4(v);
      if (HAS_RES) v += rres[i & 1][it];
      if (m < p.M && col_ok) {
        if (OUT_F32) {
          *(f32x4*)((float*)p.C + (size_t)orow * p.ldc + ncol) = v;
        } else {
          uint2 o;
          o.x = pack2bf(v[0], v[1]);
          o.y = pack2bf(v[2], v[3]);
          *(uint2*)((bf16_t*)p.C + (size_t)orow * p.ldc + ncol) = o;
        }
      }
    }
  }
#undef ISSUE_RES_P
}

// ---- epilogue through LDS (one-tile-per-workgroup kernel only: nobody reads the k-tile buffers after the last tile's mid barrier):
// each wave parks one 64-column half of its tile (fp32) in a private 128 x 64 patch, reads it back row-major: 16 lanes cover one
// 64-column row segment (256 B fp32 / 128 B bf16).  Handles every column alignment vitcap_gemm_ex admits (N % 4 == 0).
template <int ACT, int OUT_F32, bool HAS_RES, int MI>
__device__ __forceinline__ void epilogue_lds(f32x4 (&acc)[MI][8], const GemmArgs& p, char* ep, const int row_w, const int col_w, const int lane) {
  const int frow = lane & 15, fk = lane >> 4;
  const int er = lane >> 4, ec = (lane & 15) * 4;    // read-back: row within a group of 4, first of 4 columns
  constexpr int CH = 2 * MI;                          // rows of a chunk = 4 * CH: the residual is requested one chunk ahead
  f32x4 rres[2][CH];
#define ISSUE_RES(hn_, c_, slot_)                                                                            \
  if (HAS_RES) {                                                                                             \
    const int nc_ = col_w + (hn_) * 64 + ec;                                                                 \
    _Pragma("unroll") for (int it = 0; it < CH; ++it) {                                                      \
      const int m_ = row_w + (c_) * (4 * CH) + it * 4 + er;                                                  \
      ROWS_OF(m_, o_, r_);                                                                                   \
      (void)o_;                                                                                              \
      rres[slot_][it] = (m_ < p.M && nc_ < p.N) ? *(const f32x4*)(p.res + (size_t)r_ * p.ldr + nc_)          \
                                                : f32x4{0.f, 0.f, 0.f, 0.f};                                 \
    }                                                                                                        \
  }
  ISSUE_RES(0, 0, 0);
#pragma unroll
  for (int hn = 0; hn < 2; ++hn) {
    const int ncol = col_w + hn * 64 + ec;
    const bool col_ok = ncol < p.N;
    f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias && col_ok) bias4 = *(const f32x4*)(p.bias + ncol);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) *(f32x4*)(ep + (i * 16 + frow) * EP_ROWB + (j * 16 + fk * 4) * 4) = acc[i][hn * 4 + j];
#pragma unroll
    for (int c = 0; c < 2; ++c) {           // two chunks of 8 * MI rows
      const int g = hn * 2 + c;
      if (g + 1 < 4) { ISSUE_RES((g + 1) >> 1, (g + 1) & 1, (g + 1) & 1); }
#pragma unroll
      for (int it = 0; it < CH; ++it) {
        const int rl = c * (4 * CH) + it * 4 + er;
        f32x4 v = *(const f32x4*)(ep + rl * EP_ROWB + ec * 4);
        const int m = row_w + rl;
        const bool ok = m < p.M && col_ok;
        ROWS_OF(m, orow, rrow);
        (void)rrow;
        v += bias4;
        if (ACT == VITCAP_ACT_GELU_ERF) v = gelu_erf4(v);
        if (HAS_RES) v += rres[g & 1][it];
        if (ok) {
          if (OUT_F32) {
            *(f32x4*)((float*)p.C + (size_t)orow * p.ldc + ncol) = v;
          } else {
            uint2 o;
            o.x = pack2bf(v[0], v[1]);
            o.y = pack2bf(v[2], v[3]);
            *(uint2*)((bf16_t*)p.C + (size_t)orow * p.ldc + ncol) = o;
          }
        }
      }
    }
  }
#undef ISSUE_RES
}
#undef ROWS_OF

// tile list position -> tile coordinates: column groups of group_n tiles (the tiles an XCD runs at once span few W tiles, which stay
// in its L2 for the walk down M, and each A tile is fetched once for group_n consumers)
__device__ __forceinline__ void tile_of(const GemmArgs& p, int pos, int& tm, int& tn) {
  const int per_g = p.tiles_m * p.group_n;
  const int gi = pos / per_g, rem = pos - gi * per_g;
  const int gleft = p.tiles_n - gi * p.group_n;
  const int gw = gleft < p.group_n ? gleft : p.group_n;
  tm = rem / gw;
  tn = gi * p.group_n + rem - tm * gw;
  if ((p.rev != 0) != ((gi & 1) != 0)) tm = p.tiles_m - 1 - tm;      // walk direction, see gemm.hip / common.h
}

// ---- one tile per workgroup (EPI: 0 = LDS epilogue, 1 = register epilogue)
template <int ACT, int OUT_F32, bool HAS_RES, int EPI, int MI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_nt_4w_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int bid = blockIdx.x;
  {
    // XCD-aware, bijective remap: the blocks of one XCD (block b runs on XCD b % 8) get a contiguous chunk of the tile list
    const int nwg = p.tiles_m * p.tiles_n;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int tm, tn;
  tile_of(p, bid, tm, tn);
  const int m0 = tm * (32 * MI), n0 = tn * 256;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Loop L = make_loop<MI>(p, lds_addr(smem), lane, w);
  const Src src = make_src(p, m0, n0);
  f32x4 acc[MI][8];
  Frags f;
  const int nk = p.K / 64;      // >= 2 (launcher)
  constexpr auto PIECES = std::make_integer_sequence<int, MI + 8>{};

  STAMP(0);
  // prologue: k-tiles 0 and 1 requested, tile 0 awaited, its k-half-0 fragments read
  dma_tile<MI>(L, src, 0, 0, PIECES);
  dma_tile<MI>(L, src, BUF_BYTES, 128, PIECES);
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(MI + 8) : "memory");
  read_frags<MI, 0>(f, L.a_rd[0], L.w_rd[0], PIECES);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  STAMP(1);
  uint32_t cur = BUF_BYTES;        // buffer of the tile about to run (after tile 0)
  if (nk == 2) {
    k_tile<MI, 2, true>(acc, f, L, src, 0, 0);
  } else {
    k_tile<MI, 1, true>(acc, f, L, src, 0, 2 * 128);
    for (int t = 1; t < nk - 2; ++t) {
      k_tile<MI, 1, false>(acc, f, L, src, cur, (uint32_t)(t + 2) * 128);
      cur = BUF_BYTES - cur;
    }
    k_tile<MI, 2, false>(acc, f, L, src, cur, 0);
    cur = BUF_BYTES - cur;
  }
  k_tile<MI, 3, false>(acc, f, L, src, cur, 0);
  fence_accumulators<MI>(acc);
  STAMP(2);
  const int row_w = m0 + (w >> 1) * (16 * MI), col_w = n0 + (w & 1) * 128;
  if constexpr (EPI == 1) {
    if constexpr (OUT_F32 || HAS_RES) epilogue_patch<ACT, OUT_F32, HAS_RES, MI>(acc, p, smem + 2 * BUF_BYTES + w * PATCH_BYTES, row_w, col_w, lane);
    else epilogue_regs<ACT, OUT_F32, HAS_RES, MI>(acc, p, row_w, col_w, lane);
  }
  else epilogue_lds<ACT, OUT_F32, HAS_RES, MI>(acc, p, smem + w * EP_WAVE, row_w, col_w, lane);
  STAMP(3);
}

// ---- persistent form: one workgroup per CU walks its share of the tile list as ONE continuous software pipeline.  The k-tile
// stream does not stop at an output tile's end: the last two k-tiles of a tile already request the first two k-tiles of the NEXT
// tile (buffers keep alternating with the stream position), the register epilogue sits between two k-tiles, and the next tile's
// k-half-0 fragments were read before it started -- no per-tile prologue (3-7k cycles of exposed DMA latency per tile in the
// one-tile kernel, stamped), and the epilogue's stores drain behind the next tile's first MFMAs.
template <int ACT, int OUT_F32, bool HAS_RES, int MI, bool EXTRAS = false, bool PF = false, bool DEFER = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_nt_4wp_kernel(GemmArgs p) {
  static_assert(!PF || (!OUT_F32 && !HAS_RES), "the A-panel prefetch needs a scratch line behind the k-tile buffers: bf16-output forms only");
  static_assert(!DEFER || (!OUT_F32 && !HAS_RES && !EXTRAS && !PF), "deferred stores: plain bf16-output form only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // this workgroup's positions: XCD x = blockIdx % 8 owns the contiguous chunk [c0, c1) of the tile list; its workgroups (slot =
  // blockIdx / 8 of nslot) take c0 + slot, c0 + slot + nslot, ...: at any time an XCD's CUs work on neighbouring tiles
  const int nwg = p.tiles_m * p.tiles_n;
  const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
  const int c0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int c1 = c0 + q + (xcd < r ? 1 : 0);
  const int nslot = ((int)gridDim.x - xcd + 7) >> 3;
  int pos = c0 + ((int)blockIdx.x >> 3);
  if (pos >= c1) return;
#ifdef VC_4W_STAMP
  // probe: start the XCDs a fraction of a tile period apart (p.direct_epilogue = units of 64 cycles per XCD step)
  for (int i = 0; i < ((int)blockIdx.x >> 3) * p.direct_epilogue; ++i) __builtin_amdgcn_s_sleep(1);
#endif
  Loop L = make_loop<MI>(p, lds_addr(smem), lane, w);
  int tm, tn;
  tile_of(p, pos, tm, tn);
  Src src = make_src(p, tm * (32 * MI), tn * 256);
  // (the fp32 / residual forms own all 160 KiB of LDS already: no scratch line, no prefetch)
  if constexpr (PF) L.lds_pf = lds_addr(smem) + 2 * BUF_BYTES + w * 256;
  uint32_t pf_step = 0;      // byte offset between the row chunks of consecutive k-tiles
  (void)pf_step;
  f32x4 acc[MI][8];
  Frags f;
  Park<MI> pk;                  // DEFER: the previous tile's parked stores (first tile: zero records -> every store is dropped)
  if constexpr (DEFER) {
#pragma unroll
    for (int n = 0; n < Park<MI>::NP * 4; ++n) pk.v[n] = u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
    for (int n = 0; n < Park<MI>::NP; ++n) pk.voff[n] = 0u;
  }
  int park_row = -1;            // first row of the wave tile whose upper half is parked (-1: nothing parked yet)
  (void)park_row;
  const int nk = p.K / 64;      // >= 2 (launcher; DEFER: >= Park::NP + 3)
  constexpr auto PIECES = std::make_integer_sequence<int, MI + 8>{};
  // the only prologue: k-tiles 0 and 1 of the first tile
  dma_tile<MI>(L, src, 0, 0, PIECES);
  dma_tile<MI>(L, src, BUF_BYTES, 128, PIECES);
  if constexpr (PF) {
    pf64(L.lds_pf, L.voff_pf, src.ra, 0);      // out of range: no fetch, one instruction for the counted wait
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(MI + 8 + 1) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(MI + 8) : "memory");
  }
  read_frags<MI, 0>(f, L.a_rd[0], L.w_rd[0], PIECES);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  uint32_t cur = 0;
  // ONE straight-line body per output tile (no branch around any accumulator update: hipcc merges accumulators that are defined on
  // two paths through VGPR copies and scratch -- measured: 750-1150 spilled registers in a version that branched on `more`).  A
  // workgroup's last tile requests its two look-ahead k-tiles through a descriptor of zero records: every lane is out of range,
  // nothing is fetched, and the (unused) LDS image is whatever the range check returns.
  while (true) {
    const int m0 = tm * (32 * MI), n0 = tn * 256;
    const int npos = pos + nslot;
    const bool more = npos < c1;
    tile_of(p, more ? npos : pos, tm, tn);
    Src nsrc = make_src(p, tm * (32 * MI), tn * 256);
    if (!more) {
      nsrc.ra[2] = 0;
      nsrc.rw[2] = 0;
    }
    if constexpr (PF) {
      // The column siblings of the next tile (same tm, the gw tiles of its column group) split its A panel by rows: sibling c takes
      // rows c, c + gw, c + 2 gw ...; k-tile t of THIS tile requests RPS of them, whole rows (nk lines of 128 B: one DRAM page).
      // lane -> (row within the chunk, line of the row)
      const int gw_ = p.group_n < p.tiles_n ? p.group_n : p.tiles_n;
#ifdef VC_4W_PF_ALL
      const int gwe = 1, c_ = 0;
#else
      const int gwe = gw_, c_ = tn % gw_;
#endif
      const int share = (32 * MI - c_ + gwe - 1) / gwe;
      const int rps = (share + nk - 1) / nk;
      const int Lq = w * 64 + lane;
      const int jr = (int)((float)Lq / (float)nk), line = Lq - jr * nk;       // exact for these magnitudes (Lq < 256, nk <= 48)
      L.voff_pf = (jr < rps) ? (uint32_t)((c_ + jr * gwe) * p.lda * 2 + line * 128) : 0xfffffff0u;
      pf_step = (uint32_t)(rps * gwe * p.lda * 2);
    }
#define PF_ARGS(t_) , nsrc, (uint32_t)(t_) * pf_step
    // k-tiles 0 .. nk-3 request k-tiles 2 .. nk-1 of this tile; k-tiles nk-2, nk-1 request k-tiles 0, 1 of the next tile (nk >= 3)
    STAMP_AT(pos, 0);
    int t_plain = 1;
    if constexpr (DEFER) {
      // k-tiles 0 .. NP-1 issue the previous tile's parked stores, one m-tile (4 stores) each
      constexpr int NP = Park<MI>::NP;
      k_tile_st<MI, 1, true, 0>(acc, f, L, src, cur, 2 * 128, pk, p, park_row);
      cur = BUF_BYTES - cur;
      STAMP_AT(pos, 1);
      if constexpr (NP > 1) { k_tile_st<MI, 1, false, 4>(acc, f, L, src, cur, 3 * 128, pk, p, park_row); cur = BUF_BYTES - cur; }
      if constexpr (NP > 2) { k_tile_st<MI, 1, false, 8>(acc, f, L, src, cur, 4 * 128, pk, p, park_row); cur = BUF_BYTES - cur; }
      if constexpr (NP > 3) { k_tile_st<MI, 1, false, 12>(acc, f, L, src, cur, 5 * 128, pk, p, park_row); cur = BUF_BYTES - cur; }
      t_plain = NP;
    } else {
      k_tile<MI, 1, true, PF>(acc, f, L, src, cur, 2 * 128 PF_ARGS(0));
      cur = BUF_BYTES - cur;
      STAMP_AT(pos, 1);
    }
    for (int t = t_plain; t < nk - 2; ++t) {
      k_tile<MI, 1, false, PF>(acc, f, L, src, cur, (uint32_t)(t + 2) * 128 PF_ARGS(t));
      cur = BUF_BYTES - cur;
    }
    k_tile<MI, 1, false, PF>(acc, f, L, nsrc, cur, 0 PF_ARGS(nk - 2));
    cur = BUF_BYTES - cur;
    k_tile<MI, 4, false, PF>(acc, f, L, nsrc, cur, 128 PF_ARGS(nk - 1));
#undef PF_ARGS
    cur = BUF_BYTES - cur;
    fence_accumulators<MI>(acc);
    STAMP_AT(pos, 2);
#ifdef VC_4W_GENERAL_EPI
    if constexpr (OUT_F32 || HAS_RES)
      epilogue_patch<ACT, OUT_F32, HAS_RES, MI>(acc, p, smem + 2 * BUF_BYTES + w * PATCH_BYTES, m0 + (w >> 1) * (16 * MI), n0 + (w & 1) * 128, lane);
    else
      epilogue_regs<ACT, OUT_F32, HAS_RES, MI>(acc, p, m0 + (w >> 1) * (16 * MI), n0 + (w & 1) * 128, lane);
#else
    if constexpr (OUT_F32 || HAS_RES)
      epilogue_patch_fast<ACT, OUT_F32, HAS_RES, MI>(acc, p, smem + 2 * BUF_BYTES + w * PATCH_BYTES, m0 + (w >> 1) * (16 * MI), n0 + (w & 1) * 128, lane);
    else
    {
      if constexpr (EXTRAS) epilogue_regs_fast_x<ACT, MI>(acc, p, m0 + (w >> 1) * (16 * MI), n0 + (w & 1) * 128, lane);
      else if constexpr (DEFER) {
        park_row = m0 + (w >> 1) * (16 * MI);
        epilogue_regs_fast_defer<ACT, MI>(acc, p, park_row, n0 + (w & 1) * 128, lane, pk);
      }
      else epilogue_regs_fast<ACT, MI>(acc, p, m0 + (w >> 1) * (16 * MI), n0 + (w & 1) * 128, lane);
    }
#endif
    STAMP_AT(pos, 3);
    if (!more) {
      if constexpr (DEFER) store_parked_all<MI>(pk, park_rsrc(p, park_row), std::make_integer_sequence<int, Park<MI>::NP * 4>{});      // the last tile flushes its own parked half
      break;
    }
    pos = npos;
    src = nsrc;
    // the next tile's k-tile 0 landed before the last mid-tile barrier: its k-half-0 fragments
    read_frags<MI, 0>(f, L.a_rd[0] + cur, L.w_rd[0] + cur, PIECES);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// Tile height by launch: a launch of T equal tiles on C CUs costs ceil(T / C) tile times, and a (32 * MI)-row tile costs about
// (fixed + MI) units -- 435 tiles of 256 rows (M = 36928, N = 768) pay 2 rounds for 1.7 rounds of work, 495 tiles of 224 rows pay
// 2 rounds of 7/8 the length; 1305 (N = 2304) pay 6, 1485 of 224 rows pay 6 x 7/8.  The kernel is the same code for every MI (the
// wave tile is MI x 8 MFMA tiles, k order per output element unchanged: bit-identical results whatever the height).
int device_cus() {
  static const int n = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      return prop.multiProcessorCount;
    return 256;
  }();
  return n;
}
// workgroups of the persistent grid = CUs it occupies (512 registers per lane: one workgroup owns a CU).  VITCAP_GEMM_4W_RESERVE = R
// leaves R CUs (a multiple of 8: the XCDs stay balanced) to whatever else runs on the GPU -- the batch pipeline's decode chain
std::atomic<int> g_reserve_cus{0};       // vitcap_gemm_reserve_cus: set by the caller while a collective is in flight
int persistent_cus() {
  static const int env_reserve = [] { const char* e = getenv("VITCAP_GEMM_4W_RESERVE"); return e ? atoi(e) : 0; }();
  const int run = g_reserve_cus.load(std::memory_order_relaxed);
  const int reserve = run > env_reserve ? run : env_reserve;
  const int n = device_cus() - (reserve > 0 ? reserve : 0);
  return n >= 8 ? n : 8;
}

int pick_mi(int M, int tiles_n, int form) {
  static const int env_mi = [] { const char* e = getenv("VITCAP_GEMM4W_MI"); return e ? atoi(e) : 0; }();
  if (env_mi >= 4 && env_mi <= 8 && env_mi != 5) return env_mi;
  const int n_cu = form == 2 ? persistent_cus() : device_cus();
  const float fixed = form == 2 ? 0.3f : 0.9f;       // per-tile cost that does not shrink with the tile: pipeline fill, barriers' skew, W traffic
  int best = 8;
  float best_c = 1e30f;
  for (int mi = 8; mi >= 6; --mi) {
    const long long tiles = (long long)((M + 32 * mi - 1) / (32 * mi)) * tiles_n;
    const float c = (float)((tiles + n_cu - 1) / n_cu) * (fixed + (float)mi);
    if (c < best_c * 0.985f) { best_c = c; best = mi; }      // a shorter tile must win 1.5 % to be taken
  }
  return best;
}

// form: 0 = one tile per workgroup + LDS epilogue, 1 = one tile per workgroup + register epilogue, 2 = persistent
template <int ACT, int OUT_F32, bool HAS_RES, int MI>
int launch_4w_mi(GemmArgs& p, hipStream_t s, int form) {
  p.tiles_m = (p.M + 32 * MI - 1) / (32 * MI);
  p.n_big = p.tiles_m * p.tiles_n;
  if (form == 2) {
    const int n_cu = persistent_cus();
    const bool extras = p.aux || p.zout || p.colsum;        // training extras: the bf16 register epilogue's second form (vc_4w_supports)
    constexpr int smem = (OUT_F32 || HAS_RES) ? SMEM_4WP : 2 * BUF_BYTES;
    // VITCAP_GEMM_4W_TIGHT=1 (experiments): the fewest workgroups (a multiple of 8: every XCD the same number) that finish in the same
    // number of rounds as all CUs would -- 1305 tiles take 6 rounds on 256 CUs and on 224 -- leaving the other CUs to whatever else
    // runs.  Measured (docs/LAB_r01_r04.md 4.3): no gain alone (19.55 vs 19.59 ms, B = 512 124.3 vs 123.3), and inside the batch pipeline only a
    // CONSTANT reservation helps the decode chain (VITCAP_GEMM_4W_RESERVE=32 ties the 8-wave kernel there), so the default is off
    static const int tight = [] { const char* e = getenv("VITCAP_GEMM_4W_TIGHT"); return e ? atoi(e) : 0; }();
    int grid = p.n_big < n_cu ? p.n_big : n_cu;
    if (tight && p.n_big > n_cu) {
      const int rounds = (p.n_big + n_cu - 1) / n_cu;
      const int need = ((p.n_big + rounds - 1) / rounds + 7) & ~7;
      if (need < grid) grid = need;
    }
    // one call site per kernel: VC_FUNC_SMEM remembers per call site that the attribute was set
    bool launched = false;
    if constexpr (!OUT_F32 && !HAS_RES) {
      if (extras) {
        auto kern = gemm_nt_4wp_kernel<ACT, OUT_F32, HAS_RES, MI, true>;
        VC_FUNC_SMEM(kern, smem);
        VC_LAUNCH_GEMM(kern, dim3(grid), dim3(256), smem, s, p);
        launched = true;
      }
      // A-panel prefetch one tile ahead (round 5's measured probe, shipped in round 6): from ~64k rows on the A panel streams from
      // HBM and the same loop takes 32-37k instead of 25.4k cycles per tile; whole rows of the NEXT tile's panel, split over the column
      // siblings, requested one per k-tile bring it to 29.7k (qkv +3.6 %, fc1 +2.1 % wall at M = 295 424, docs/LAB_r05.md section 1).
      // Below the threshold the operands are cache-resident and the extra instruction only costs (+0.8 % loop cycles): plain form.
      // VITCAP_GEMM_4W_PF: 0 = never, N > 1 = from N rows on (default 65536).
      // deferred stores (round 6): the upper half of every wave tile's stores ride behind the next tile's first k-tiles.  Measured
      // (profiles/r06_deferred_stores.txt): the epilogue shrinks as priced (qkv 8.7k -> 6.2k cycles per tile), but with ONE wave per
      // SIMD a store's issue (~150 cycles each) stalls the wave wherever it stands, so the next tile's loop pays what the epilogue
      // saved (25.3k -> 27.2k + 0.7k): a tie without an activation (qkv 138.9 -> 138.3 us, the training step 49.63 -> 49.23 ms),
      // a loss with GELU (fc1 211.8 -> 221.0 us: its stores were already hidden behind the activation's arithmetic) and from 64k rows
      // on, where the parked stores compete with the A panel's HBM stream (qkv 1 080 -> 1 153 us; B = 512 pipeline 4 160 -> 4 064
      // img/s).  Policy: plain bf16 outputs below 64k rows.  VITCAP_GEMM_4W_DEFER: 0 = never, 1 = that policy (default), 2 = always.
      {
        const char* de = getenv("VITCAP_GEMM_4W_DEFER");      // read per launch (not cached): the parity test switches it
        const int defer = de ? atoi(de) : 1;
        const bool want = defer == 2 || (defer == 1 && ACT == VITCAP_ACT_NONE && p.M < 65536);
        if (!launched && want && p.K / 64 >= Park<MI>::NP + 3) {
          auto kern = gemm_nt_4wp_kernel<ACT, OUT_F32, HAS_RES, MI, false, false, true>;
          VC_FUNC_SMEM(kern, smem);
          VC_LAUNCH_GEMM(kern, dim3(grid), dim3(256), smem, s, p);
          launched = true;
        }
      }
      if constexpr (MI == 8) {
        static const int pf_rows = [] { const char* e = getenv("VITCAP_GEMM_4W_PF"); return e ? atoi(e) : 65536; }();
        if (!launched && pf_rows > 0 && p.M >= pf_rows && p.tiles_m > 1) {
          auto kern = gemm_nt_4wp_kernel<ACT, OUT_F32, HAS_RES, MI, false, true>;
          constexpr int smem_pf = 2 * BUF_BYTES + 1024;
          VC_FUNC_SMEM(kern, smem_pf);
          VC_LAUNCH_GEMM(kern, dim3(grid), dim3(256), smem_pf, s, p);
          launched = true;
        }
      }
    }
    if (!launched) {
      auto kern = gemm_nt_4wp_kernel<ACT, OUT_F32, HAS_RES, MI>;
      VC_FUNC_SMEM(kern, smem);
      VC_LAUNCH_GEMM(kern, dim3(grid), dim3(256), smem, s, p);
    }
  } else {
    auto kern = gemm_nt_4w_kernel<ACT, OUT_F32, HAS_RES, 1, MI>;
    constexpr int smem = (OUT_F32 || HAS_RES) ? SMEM_4WP : 2 * BUF_BYTES;
    VC_FUNC_SMEM(kern, smem);
    VC_LAUNCH_GEMM(kern, dim3(p.n_big), dim3(256), smem, s, p);
  }
  VC_LAUNCH_CHECK("gemm_nt_4w");
  return VITCAP_OK;
}

template <int ACT, int OUT_F32, bool HAS_RES>
int launch_4w(const GemmArgs& a, hipStream_t s, int form) {
  GemmArgs p = a;
  p.tiles_n = (a.N + 255) / 256;
  p.group_n = vc_tile_group_n(p.tiles_n);
  p.tiles_m_small = 0;
  // the register epilogue's bf16 stores are 16 bytes wide
  const bool regs_ok = OUT_F32 || HAS_RES || ((a.N & 7) == 0 && (a.ldc & 7) == 0);
  if (form != 0 && !regs_ok) form = 0;
  // the persistent pipeline's tile body needs three k-tiles; its epilogues address plain rows of whole 256-column tiles
  if (form == 2 && (a.K < 192 || (a.N & 255) != 0 || a.row_group != 0 || (!OUT_F32 && (a.ldc & 7) != 0))) form = 1;
  if (form == 0) {
    p.tiles_m = (a.M + 255) / 256;
    p.n_big = p.tiles_m * p.tiles_n;
    auto kern = gemm_nt_4w_kernel<ACT, OUT_F32, HAS_RES, 0, 8>;
    VC_FUNC_SMEM(kern, SMEM_4W);
    VC_LAUNCH_GEMM(kern, dim3(p.n_big), dim3(256), SMEM_4W, s, p);
    VC_LAUNCH_CHECK("gemm_nt_4w");
    return VITCAP_OK;
  }
  switch (pick_mi(a.M, p.tiles_n, form)) {
    case 7: return launch_4w_mi<ACT, OUT_F32, HAS_RES, 7>(p, s, form);
    case 6: return launch_4w_mi<ACT, OUT_F32, HAS_RES, 6>(p, s, form);
    case 4: return launch_4w_mi<ACT, OUT_F32, HAS_RES, 4>(p, s, form);
    default: return launch_4w_mi<ACT, OUT_F32, HAS_RES, 8>(p, s, form);
  }
}

}  // namespace

extern "C" int vitcap_gemm_reserve_cus(int cus) {
  if (cus < 0) cus = 0;
  cus = (cus + 7) & ~7;
  return g_reserve_cus.exchange(cus, std::memory_order_relaxed);
}

int vc_4w_pick_mi(int M, int tiles_n, int form) { return form == 0 ? 8 : pick_mi(M, tiles_n, form); }

// Whether the AUTOMATIC choice hands a launch with training extras to the 4-wave kernel.  Measured inside the training step on one box
// (profiles/r05_train_extras_ab.txt): every such launch on the 8-wave kernel 52.16 / 52.28 ms per step; zout launches (fc1 forward) on
// the 4-wave kernel 52.47 / 52.44; aux + colsum launches (fc2 input gradient) as well 53.11 / 53.12 -- so the default is "none".
// VITCAP_GEMM_4W_EXTRAS: bit 0 = launches with zout, bit 1 = with aux, bit 2 = with colsum may go (tile_hint 42 always does).
bool vc_4w_extras_auto(const GemmArgs& a) {
  static const int mask = [] { const char* e = getenv("VITCAP_GEMM_4W_EXTRAS"); return e ? atoi(e) : 0; }();
  return !((a.zout && !(mask & 1)) || (a.aux && !(mask & 2)) || (a.colsum && !(mask & 4)));
}

bool vc_4w_supports(const GemmArgs& a, int act) {
#ifdef VC_4W_STAMP
  if (a.rowstat) return true;
#endif
  if (a.aux || a.zout || a.colsum) {
    // training extras: the PERSISTENT form's bf16 register epilogue only (no downgrade of the form inside launch_4w: the caller checks
    // that the persistent form was chosen) -- plain rows, whole 256-column tiles, 16-byte aligned rows of every operand
    return !a.res && !a.rowstat && !a.live && a.split_k <= 1 && a.K >= 192 && (a.N & 255) == 0 && a.row_group == 0 && (a.ldc & 7) == 0 &&
           (!a.aux || (a.ldaux & 7) == 0) && (!a.zout || (a.ldz & 7) == 0) && a.M >= 2048 &&
           (act == VITCAP_ACT_NONE || act == VITCAP_ACT_GELU_ERF);
  }
  // a.live (decode loops: early exit once every sequence has finished) is honoured by the 8-wave kernels only: such launches stay there
  return !a.aux && !a.zout && !a.colsum && !a.rowstat && !a.live && a.split_k <= 1 && a.K >= 128 &&
         (act == VITCAP_ACT_NONE || act == VITCAP_ACT_GELU_ERF);
}

int vc_dispatch_4w(const GemmArgs& a, int act, int out_f32, hipStream_t s, int form) {
  VC_REQUIRE(vc_4w_supports(a, act), "gemm(4-wave): unsupported options (training extras / split-K / activation %d / K < 128)", act);
  const bool res = a.res != nullptr;
#define CASE(ACT_, OUT_)              \
  if (act == ACT_ && out_f32 == OUT_) \
    return res ? launch_4w<ACT_, OUT_, true>(a, s, form) : launch_4w<ACT_, OUT_, false>(a, s, form);
  CASE(VITCAP_ACT_NONE, 0)
  CASE(VITCAP_ACT_NONE, 1)
  CASE(VITCAP_ACT_GELU_ERF, 0)
  CASE(VITCAP_ACT_GELU_ERF, 1)
#undef CASE
  vitcap_set_error("gemm(4-wave): unsupported act %d / out %d", act, out_f32);
  return VITCAP_EINVAL;
}
