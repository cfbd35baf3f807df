// bf16 NT GEMM with fused epilogue for gfx950 (MI355X):  C = act(A . W^T + bias) (+ residual)
//
// Both operands are K-contiguous (A[M][K], W[N][K] = nn.Linear layout), which is exactly the MFMA
// fragment layout (8 consecutive k per lane), so tiles go HBM -> LDS by `global_load_lds` (16 B per
// lane, no VGPR round trip) and LDS -> VGPR by ds_read_b128.
//
// Tile: (32*WM) x (32*WN) x 64 per 256-thread workgroup (2x2 waves, each wave WM x WN MFMA tiles of
// 16x16x32).  LDS rows are 128 B (one k-tile); 16-byte chunks are XOR-swizzled with (row & 7) so the
// 16-lane ds_read_b128 groups hit 16 distinct slots.  global_load_lds writes lane-linear, hence the
// swizzle is applied on the per-lane SOURCE address and again on the read (cdna guide rule 21).
// Two LDS buffers; the next k-tile's loads are issued before the current tile's MFMAs and drained
// once per k-tile (one barrier per tile).
//
// The accumulators are computed transposed (first MFMA operand = W fragment): each lane then holds
// 4 consecutive n for one m, so bias/residual loads and the C store are 8/16-byte vectors.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "gemm_args.h"
#include <string.h>


#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>

namespace {

// width (in 256-column tiles) of the column groups the 256x256 kernels walk; VITCAP_GEMM_GROUP_N overrides (experiments)
int tile_group_n(int tiles_n) {
  static const int env_gn = [] { const char* e = getenv("VITCAP_GEMM_GROUP_N"); return e ? atoi(e) : 0; }();
  // measured at M = 36928 (tools/group_sweep.sh): N = 3072 214 -> 199 us with groups of 2..6 tiles (W = 4.7 MB does not fit
  // one XCD's 4 MB L2 next to the A tiles; HBM fetch 169 -> 152 MiB per launch); N = 2304 (W = 3.5 MB fits) same time with
  // 7 % MORE fetch when grouped, so only wider outputs are grouped
  int g = env_gn > 0 ? env_gn : (tiles_n > 9 ? 3 : tiles_n);
  return g > tiles_n ? tiles_n : g;
}

}  // namespace
int vc_tile_group_n(int tiles_n) { return tile_group_n(tiles_n); }
namespace {

__device__ __forceinline__ void glds16(const void* g, void* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most `newer` (<= K) tiles of PIECES LDS-DMA instructions each are still outstanding
template <int K, int PIECES>
__device__ __forceinline__ void wait_newer_tiles(int newer) {
  if constexpr (K <= 0) {
    wait_vmcnt<0>();
  } else {
    if (newer >= K) wait_vmcnt<(K * PIECES < 63 ? K * PIECES : 63)>();
    else wait_newer_tiles<K - 1, PIECES>(newer);
  }
}

template <int WM, int WN, int ACT, int OUT_F32, bool HAS_RES, int NST, bool ROWSTAT = false>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs p) {
  VC_LIVE_EXIT(p.live);
  constexpr int BM = 32 * WM, BN = 32 * WN, BK = 64;
  constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BUF_BYTES = A_BYTES + W_BYTES;   // buffer b: A tile at b*BUF_BYTES, W tile behind it

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;

  // XCD-aware, bijective block remap: blocks that land on one XCD (id % 8) get a contiguous range of
  // tiles, n fastest, so A row-panels are shared through that XCD's L2.
  const int nwg = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid / p.tiles_n, tn = bid - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // per-lane source rows for the staging loads (row index fixed over the k loop)
  const int srow = lane >> 3;                        // 0..7 inside an 8-row glds piece
  const int schunk = (lane & 7) ^ (srow & 7);        // swizzled 16-B chunk this lane fetches
  const bf16_t* aptr[BM / 32];
  const bf16_t* wptr[BN / 32];
#pragma unroll
  for (int i = 0; i < BM / 32; ++i) {
    int r = m0 + w * (BM / 4) + i * 8 + srow;
    r = r < p.M ? r : p.M - 1;
    aptr[i] = p.A + (size_t)r * p.lda + schunk * 8;
  }
#pragma unroll
  for (int i = 0; i < BN / 32; ++i) {
    int r = n0 + w * (BN / 4) + i * 8 + srow;
    r = r < p.N ? r : p.N - 1;
    wptr[i] = p.W + (size_t)r * p.ldw + schunk * 8;
  }

  auto stage = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < BM / 32; ++i)
      glds16(aptr[i] + k0, smem + buf * BUF_BYTES + (w * (BM / 4) + i * 8) * 128);
#pragma unroll
    for (int i = 0; i < BN / 32; ++i)
      glds16(wptr[i] + k0, smem + buf * BUF_BYTES + A_BYTES + (w * (BN / 4) + i * 8) * 128);
  };

  f32x4 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;       // row inside a 16-row fragment
  const int fk = lane >> 4;         // which 8-wide k group inside a 32-wide MFMA k step
  const int nk = p.K / BK;

  // split-K (weight gradients): this workgroup owns k-tiles [t0, t1)
  int t0 = 0, t1 = nk;
  if (p.split_k > 1) {
    t0 = blockIdx.y * p.kt_per_split;
    t1 = t0 + p.kt_per_split < nk ? t0 + p.kt_per_split : nk;
  }
  // NST LDS stages, prefetch distance NST-1: the small-M (decode) shapes are LDS-DMA-latency-bound -- a 64x64 tile has
  // ~150 cycles of MFMA work per k-tile against ~800 cycles of DMA latency -- so three k-tiles are kept in flight and
  // tile t is awaited with a counted vmcnt (the newer tiles' pieces stay outstanding).
  constexpr int PIECES = BM / 32 + BN / 32;              // LDS-DMA instructions per wave per stage
#pragma unroll
  for (int i = 0; i < NST - 1; ++i)
    if (t0 + i < t1) stage(i, (t0 + i) * BK);

  for (int t = t0; t < t1; ++t) {
    const int buf = (t - t0) % NST;
    // tile t has landed once at most `newer` younger tiles' pieces are outstanding (loads retire in order)
    wait_newer_tiles<NST - 2, PIECES>(t1 - 1 - t < NST - 2 ? t1 - 1 - t : NST - 2);
    // Raw s_barrier, NOT __syncthreads(): the workgroup-scope fence in __syncthreads() makes hipcc emit `s_waitcnt vmcnt(0)` in front
    // of the barrier whenever LDS-DMA is in flight -- every k-tile then waited for the YOUNGEST request and the NST-deep ring
    // degenerated to one dependent memory round trip per k-tile (seen in the ISA; the counted wait above was dead code).
    // What the barrier must order needs no fence: tile t landed for every wave (each wave's own counted vmcnt) and every wave
    // finished tile t-1, whose fragments were consumed by MFMAs behind the compiler's own lgkmcnt waits.
    __builtin_amdgcn_s_barrier();
    if (t + NST - 1 < t1) stage((t - t0 + NST - 1) % NST, (t + NST - 1) * BK);
    const char* la = smem + buf * BUF_BYTES + (wm * WM * 16 + frow) * 128;
    const char* lw = smem + buf * BUF_BYTES + A_BYTES + (wn * WN * 16 + frow) * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + fk) ^ (frow & 7)) * 16;
      bf16x8 af[WM], wf[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) af[i] = *(const bf16x8*)(la + i * 16 * 128 + coff);
#pragma unroll
      for (int j = 0; j < WN; ++j) wf[j] = *(const bf16x8*)(lw + j * 16 * 128 + coff);
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
  }

  // epilogue: lane holds C[m][n..n+3], m = frow, n = fk*4 within each 16x16 tile
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int m = m0 + wm * WM * 16 + i * 16 + frow;
    if (m >= p.M) continue;
    int orow = m, rrow = m;
    if (p.row_group > 0) {
      const int g = m / p.row_group, in = m - g * p.row_group;
      orow = g * p.out_group_rows + p.out_row_off + in;
      rrow = p.res_periodic ? in : orow;
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n = n0 + wn * WN * 16 + j * 16 + fk * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[i][j];
      if (p.split_k > 1) {
        *(f32x4*)((float*)p.C + (size_t)blockIdx.y * p.slab + (size_t)orow * p.ldc + n) = v;
        continue;
      }
      if (p.bias) {
        const f32x4 b = *(const f32x4*)(p.bias + n);
        v += b;
      }
      if (p.aux) {
        const bf16x4 za = *(const bf16x4*)(p.aux + (size_t)orow * p.ldaux + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= (float)za[e];
      }
      if (ACT == VITCAP_ACT_GELU_ERF) {
        if (p.zout) {                  // the activation's derivative at the pre-activation, for the backward's `aux`
          f32x4 d;
          v = gelu_erf4_grad(v, d);
          uint2 zo;
          zo.x = pack2bf(d[0], d[1]);
          zo.y = pack2bf(d[2], d[3]);
          *(uint2*)(p.zout + (size_t)orow * p.ldz + n) = zo;
        } else {
          v = gelu_erf4(v);
        }
      } else if (ACT == VITCAP_ACT_TANH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
      }
      if (HAS_RES) {
        const f32x4 r = *(const f32x4*)(p.res + (size_t)rrow * p.ldr + n);
        v += r;
      }
      if (OUT_F32) {
        *(f32x4*)((float*)p.C + (size_t)orow * p.ldc + n) = v;
      } else {
        uint2 o;
        o.x = pack2bf(v[0], v[1]);
        o.y = pack2bf(v[2], v[3]);
        *(uint2*)((bf16_t*)p.C + (size_t)orow * p.ldc + n) = o;
      }
      if (ROWSTAT) acc[i][j] = v;          // keep the finished values (bias added) for the row statistics below
    }
  }
  if (ROWSTAT) {
    // Vocabulary GEMM of a decode step: next to the logits, every wave emits for each of its rows and each 32-column piece of
    // its WN*16 columns the maximum, the column of that maximum (lowest on ties, torch.argmax's rule) and sum exp(x - max) --
    // the pieces `argmax` and `log_softmax` (modeling_utils.py:846-851) are assembled from by vitcap_greedy_select_embed, and
    // the beam step's 2*beams best candidates by vitcap_row_topk_pieces, so the 30522-wide rows are never read back.
    // Piece index = first column / 32 (row stride 2*ceil(N/64) pieces whatever the tile); columns >= N never win (skipped
    // here, and the padded vocabulary columns carry a -1e30 bias).
    const int pieces = 2 * ((p.N + 63) / 64);
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int m = m0 + wm * WM * 16 + i * 16 + frow;
#pragma unroll
      for (int jp = 0; jp < WN / 2; ++jp) {
        float bm = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 2 * jp; j < 2 * jp + 2; ++j) {
          const int n = n0 + wn * WN * 16 + j * 16 + fk * 4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = acc[i][j][e];
            if (n + e < p.N && (x > bm || (x == bm && n + e < bi))) { bm = x; bi = n + e; }
          }
        }
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {             // the row's columns are spread over the 4 lane groups fk = lane >> 4
          const float om = __shfl_xor(bm, o, 64);
          const int oi = __shfl_xor(bi, o, 64);
          if (om > bm || (om == bm && oi < bi)) { bm = om; bi = oi; }
        }
        float se = 0.f;
#pragma unroll
        for (int j = 2 * jp; j < 2 * jp + 2; ++j) {
          const int n = n0 + wn * WN * 16 + j * 16 + fk * 4;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.N) se += expf(acc[i][j][e] - bm);
        }
        se += __shfl_xor(se, 16, 64);
        se += __shfl_xor(se, 32, 64);
        const int piece = (n0 + wn * WN * 16 + jp * 32) >> 5;
        if (fk == 0 && m < p.M && piece < pieces)
          *(f32x4*)(p.rowstat + ((size_t)m * pieces + piece) * 4) = f32x4{bm, __int_as_float(bi), se, 0.f};
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Large-M kernel: 256 x 128 x 64 tile, 8 waves (4 along M x 2 along N, 64x64 per wave), THREE LDS
// stages (3 x 48 KiB): the loads of k-tile t+2 are issued while tile t is multiplied, and tile t is
// awaited with a COUNTED `s_waitcnt vmcnt(6)` (this wave's 6 newer LDS-DMA pieces stay in flight)
// followed by a raw s_barrier -- no vmcnt(0) drain in the main loop (cdna guide T3/T4).
// ------------------------------------------------------------------------------------------------
template <int ACT, int OUT_F32, bool HAS_RES>
__global__ __launch_bounds__(512) void gemm_nt_big_kernel(GemmArgs p) {
  constexpr int BM = 256, BN = 128, BK = 64, WM = 4, WN = 4;
  constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, BUF_BYTES = A_BYTES + W_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;

  const int nwg = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid / p.tiles_n, tn = bid - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ (srow & 7);
  const bf16_t* aptr[4];
  const bf16_t* wptr[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = m0 + w * 32 + i * 8 + srow;
    r = r < p.M ? r : p.M - 1;
    aptr[i] = p.A + (size_t)r * p.lda + schunk * 8;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int r = n0 + w * 16 + i * 8 + srow;
    r = r < p.N ? r : p.N - 1;
    wptr[i] = p.W + (size_t)r * p.ldw + schunk * 8;
  }
#define STAGE(buf_, k0_)                                                                     \
  do {                                                                                       \
    char* sb_ = smem + (buf_) * BUF_BYTES;                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                            \
        glds16(aptr[i] + (k0_), sb_ + (w * 32 + i * 8) * 128);                               \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                            \
        glds16(wptr[i] + (k0_), sb_ + A_BYTES + (w * 16 + i * 8) * 128);                     \
  } while (0)

  f32x4 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int fk = lane >> 4;
  const int nk = p.K / BK;

  STAGE(0, 0);
  if (nk > 1) STAGE(1, BK);

  int buf = 0;
  for (int t = 0; t < nk; ++t) {
    if (t + 1 < nk)
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 2 < nk) {
      const int nb = buf >= 1 ? buf - 1 : 2;   // (buf + 2) % 3
      STAGE(nb, (t + 2) * BK);
    }
    const char* la = smem + buf * BUF_BYTES + (wm * 64 + frow) * 128;
    const char* lw = smem + buf * BUF_BYTES + A_BYTES + (wn * 64 + frow) * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + fk) ^ (frow & 7)) * 16;
      bf16x8 af[WM], wf[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) af[i] = *(const bf16x8*)(la + i * 16 * 128 + coff);
#pragma unroll
      for (int j = 0; j < WN; ++j) wf[j] = *(const bf16x8*)(lw + j * 16 * 128 + coff);
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    buf = buf == 2 ? 0 : buf + 1;
  }
#undef STAGE

#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int m = m0 + wm * 64 + i * 16 + frow;
    if (m >= p.M) continue;
    int orow = m, rrow = m;
    if (p.row_group > 0) {
      const int g = m / p.row_group, in = m - g * p.row_group;
      orow = g * p.out_group_rows + p.out_row_off + in;
      rrow = p.res_periodic ? in : orow;
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fk * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[i][j];
      if (p.bias) v += *(const f32x4*)(p.bias + n);
      if (ACT == VITCAP_ACT_GELU_ERF) {
        v = gelu_erf4(v);
      } else if (ACT == VITCAP_ACT_TANH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
      }
      if (HAS_RES) v += *(const f32x4*)(p.res + (size_t)rrow * p.ldr + n);
      if (OUT_F32) {
        *(f32x4*)((float*)p.C + (size_t)orow * p.ldc + n) = v;
      } else {
        uint2 o;
        o.x = pack2bf(v[0], v[1]);
        o.y = pack2bf(v[2], v[3]);
        *(uint2*)((bf16_t*)p.C + (size_t)orow * p.ldc + n) = o;
      }
    }
  }
}

template <int ACT, int OUT_F32, bool HAS_RES>
int launch_big(const GemmArgs& a, hipStream_t s) {
  constexpr int smem = 3 * (256 + 128) * 64 * 2;
  auto kern = gemm_nt_big_kernel<ACT, OUT_F32, HAS_RES>;
  VC_FUNC_SMEM(kern, smem);
  GemmArgs p = a;
  p.tiles_m = (a.M + 255) / 256;
  p.tiles_n = (a.N + 127) / 128;
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(512), smem, s, p);
  VC_LAUNCH_CHECK("gemm_nt_big");
  return VITCAP_OK;
}

int dispatch_big(const GemmArgs& a, int act, int out_f32, hipStream_t s) {
  const bool res = a.res != nullptr;
#define CASE(ACT_, OUT_)                                                  \
  if (act == ACT_ && out_f32 == OUT_)                                     \
    return res ? launch_big<ACT_, OUT_, true>(a, s) : launch_big<ACT_, OUT_, false>(a, s);
  CASE(VITCAP_ACT_NONE, 0)
  CASE(VITCAP_ACT_NONE, 1)
  CASE(VITCAP_ACT_GELU_ERF, 0)
  CASE(VITCAP_ACT_GELU_ERF, 1)
#undef CASE
  vitcap_set_error("gemm(big): unsupported act %d / out %d", act, out_f32);
  return VITCAP_EINVAL;
}



// ------------------------------------------------------------------------------------------------
// 256 x 256 x 64 kernel, 8 waves, two LDS buffers of 64 KiB, role-alternating schedule.
//
// Waves 0-3 (group A) and 4-7 (group B) sit pairwise on the four SIMDs.  Every k-tile is processed in four
// phases, one per 64x32 quadrant of the wave's 128x64 output; a phase is a LOAD segment (ds_read the
// quadrant's fragments, issue LDS-DMA for the next k-tile) followed by a COMPUTE segment (16 MFMAs under
// s_setprio 1), each closed by s_barrier.  Group B runs one barrier behind group A, so on every SIMD one
// wave is in its compute segment while its partner is in its load segment: the matrix pipe is fed by one
// wave while the other hides LDS/DMA latency (cdna guide T3/T4/T5, MI355X_MICROARCH "two waves per SIMD").
//   LDS reads complete (lgkmcnt(0)) before the barrier that ends a load segment;
//   the next k-tile's DMA is awaited (vmcnt(0)) right before the barrier after which group A reads it.
// ------------------------------------------------------------------------------------------------
// ---- LDS-free epilogue of the 256x256 kernels -------------------------------------------------------------------
// After the (operand-swapped) MFMAs a lane (frow = lane&15, fk = lane>>4) holds, for each of the wave's four 16-column
// tiles j, the 4 columns j*16 + fk*4.. of row frow.  A 4x4 transpose between the lane-group index fk and the tile index
// j -- two butterfly stages of v_permlane32_swap / v_permlane16_swap, pure VALU, no LDS round trip -- leaves the lane
// with the 16 CONSECUTIVE columns fk*16.. of its row: 4 lanes cover the wave's whole 64-column row segment (one 128-byte
// line in bf16, two in fp32) with 16-byte stores.  Replaces staging the accumulators through LDS (measured: the
// non-store part of that epilogue was ~200 us of a 1.09 ms 295424x2304x768 GEMM).
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
template <int NREG>
__device__ __forceinline__ void lane_tile_transpose(unsigned (&r)[4][NREG]) {
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
      const u32x2_t t = __builtin_amdgcn_permlane32_swap(r[c][k], r[c + 2][k], false, false);
      r[c][k] = t[0];
      r[c + 2][k] = t[1];
    }
#pragma unroll
  for (int c = 0; c < 4; c += 2)
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
      const u32x2_t t = __builtin_amdgcn_permlane16_swap(r[c][k], r[c + 1][k], false, false);
      r[c][k] = t[0];
      r[c + 1][k] = t[1];
    }
}

// acc[NI][4]: the wave's (16*NI) x 64 block (NI row tiles of 16, 4 column tiles of 16); row0/col0 = its origin in C
template <int ACT, int OUT_F32, bool HAS_RES, int NI>
__device__ __forceinline__ void epilogue_direct(f32x4 (&acc)[NI][4], const GemmArgs& p, int row0, int col0, int lane,
                                                bool no_store) {
  const int frow = lane & 15, fk = lane >> 4;
  f32x4 bias4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = col0 + j * 16 + fk * 4;
    bias4[j] = (p.bias && n < p.N) ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int ncol = col0 + fk * 16;                       // first of this lane's 16 consecutive columns after the transpose
  const bool col_ok = ncol < p.N && !no_store;           // N % 16 == 0 on this path
  f32x4 rres[2][4];
#define ISSUE_RES_D(i_)                                                                                    \
  if (HAS_RES) {                                                                                           \
    const int m_ = row0 + (i_) * 16 + frow;                                                                \
    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                          \
      rres[(i_) & 1][q] = (m_ < p.M && col_ok) ? *(const f32x4*)(p.res + (size_t)m_ * p.ldr + ncol + q * 4) \
                                               : f32x4{0.f, 0.f, 0.f, 0.f};                                \
  }
  ISSUE_RES_D(0);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    if (i + 1 < NI) { ISSUE_RES_D(i + 1); }
    const int m = row0 + i * 16 + frow;
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = acc[i][j] + bias4[j];
      if (ACT == VITCAP_ACT_GELU_ERF) {
        v[j] = gelu_erf4(v[j]);
      }
    }
    if (OUT_F32) {
      unsigned r[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) r[j][e] = __float_as_uint(v[j][e]);
      lane_tile_transpose<4>(r);
      if (m < p.M && col_ok) {
        float* dst = (float*)p.C + (size_t)m * p.ldc + ncol;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 o = f32x4{__uint_as_float(r[q][0]), __uint_as_float(r[q][1]), __uint_as_float(r[q][2]), __uint_as_float(r[q][3])};
          if (HAS_RES) o += rres[i & 1][q];
          *(f32x4*)(dst + q * 4) = o;
        }
      }
    } else {
      unsigned r[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r[j][0] = pack2bf(v[j][0], v[j][1]);
        r[j][1] = pack2bf(v[j][2], v[j][3]);
      }
      lane_tile_transpose<2>(r);
      if (m < p.M && col_ok) {
        bf16_t* dst = (bf16_t*)p.C + (size_t)m * p.ldc + ncol;
        uint4 o0, o1;
        o0.x = r[0][0]; o0.y = r[0][1]; o0.z = r[1][0]; o0.w = r[1][1];
        o1.x = r[2][0]; o1.y = r[2][1]; o1.z = r[3][0]; o1.w = r[3][1];
        *(uint4*)dst = o0;
        *(uint4*)(dst + 8) = o1;
      }
    }
  }
#undef ISSUE_RES_D
}

// One (64*MT) x 256 output tile at (m0, n0): MT = 4 is the 256 x 256 tile; MT = 3 / 2 are the 192- / 128-row tiles a mixed
// launch uses for the rows behind the last full round of 256-row tiles (launch_256).  Same k order per output element and the
// same epilogue arithmetic whatever MT, so the results do not depend on the tile height.
template <int ACT, int OUT_F32, bool HAS_RES, int PH, int MT>
__device__ __forceinline__ void gemm256_tile(const GemmArgs& p, char* smem, const int m0, const int n0, const int cidx = 0) {
  constexpr int BK = 64;
  constexpr int A_BYTES = 256 * BK * 2, BUF_BYTES = 2 * A_BYTES;   // the W tile sits behind a full-height A slot whatever MT
  constexpr int WR = 32 * MT;                                      // rows of one wave's block (2 waves along M, 4 along N)
  constexpr int NI = 2 * MT;                                       // ... in 16-row MFMA tiles
  constexpr int NP = MT + 4;                                       // LDS-DMA pieces (8 rows x 128 B per lane group) per wave and k-tile
  constexpr bool EXTRAS = (PH & 8) != 0;   // training extras (pre-activation copy `zout`, gelu' factor `aux`) compiled in: the
                                           // inference instantiations do not carry their 32 prefetch registers (222 instead of 226 VGPRs)
  constexpr int ABL = (PH >> 4) & 31;   // timing ablations (tools/gemm_bench.py only; results are wrong when != 0)
#ifndef LN_LOAD_AUX
#define LN_LOAD_AUX 0
#endif
  constexpr bool LN = (PH & 512) != 0;  // LayerNorm of the finished rows by the last of the row block's column tiles (GemmArgs.ln_*)
  constexpr bool NO_DMA = ABL & 1, NO_LDS = ABL & 2, NO_MFMA = ABL & 4, NO_STORE = ABL & 8, NO_EPI = ABL & 16;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = w >> 2;            // 0 = group A, 1 = group B
  const int wm = w >> 2, wn = w & 3;

  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ (srow & 7);
  const bf16_t* aptr[MT];
  const bf16_t* wptr[4];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int r = m0 + w * (8 * MT) + i * 8 + srow;
    r = r < p.M ? r : p.M - 1;
    aptr[i] = p.A + (size_t)r * p.lda + schunk * 8;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int c = n0 + w * 32 + i * 8 + srow;
    c = c < p.N ? c : p.N - 1;
    wptr[i] = p.W + (size_t)c * p.ldw + schunk * 8;
  }
  // piece q of a k-tile: q < MT -> 8 rows of A, else 8 rows of W
#define STAGE_PIECES(q0_, q1_, buf_, k0_)                                                                          \
  _Pragma("unroll") for (int q = (q0_); q < (q1_); ++q) {                                                          \
    if (q < MT)                                                                                                    \
      glds16(aptr[q < MT ? q : 0] + (k0_), smem + (buf_) * BUF_BYTES + (w * (8 * MT) + q * 8) * 128);              \
    else                                                                                                           \
      glds16(wptr[q < MT ? 0 : q - MT] + (k0_), smem + (buf_) * BUF_BYTES + A_BYTES + (w * 32 + (q - MT) * 8) * 128); \
  }
  // DMA schedule: the pieces of k-tile t+1 are spread two per load segment over L3(t-1), L0(t), L1(t), L2(t) (slot s takes
  // pieces 2s, 2s+1; MT = 3 leaves one piece for the last slot, MT = 2 none).  Buffer (t+1)&1 is free from L3(t-1) on (its
  // last reader was L2(t-1)); the pieces must have landed before the barrier after which group A starts L0(t+1): that wait
  // is vmcnt(2) -- the two pieces of tile t+2 issued in L3(t) may stay in flight -- or vmcnt(0) when nothing newer was issued.
#define STAGE_SLOT(s_, buf_, k0_) STAGE_PIECES(2 * (s_), (2 * (s_) + 2 < NP ? 2 * (s_) + 2 : NP), buf_, k0_)

  f32x4 acc[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int fk = lane >> 4;
  const int coff0 = ((0 * 4 + fk) ^ (frow & 7)) * 16;
  const int coff1 = ((1 * 4 + fk) ^ (frow & 7)) * 16;
  const int a_base = (wm * WR + frow) * 128;
  const int b_base = A_BYTES + (wn * 64 + frow) * 128;
  const int nk = p.K / BK;

  bf16x8 afr[MT][2];     // current m-half: MT m-tiles x 2 k-steps
  bf16x8 bfr[2][2][2];   // both n-halves: [nh][n-tile][k-step]

#define LOAD_A(buf_, mh_)                                                                        \
  _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                            \
    const char* ra_ = smem + (buf_) * BUF_BYTES + a_base + ((mh_) * MT + mt) * 16 * 128;         \
    afr[mt][0] = *(const bf16x8*)(ra_ + coff0);                                                  \
    afr[mt][1] = *(const bf16x8*)(ra_ + coff1);                                                  \
  }
#define LOAD_B(buf_, nh_)                                                                        \
  _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                             \
    const char* rb_ = smem + (buf_) * BUF_BYTES + b_base + ((nh_) * 2 + nt) * 16 * 128;          \
    bfr[nh_][nt][0] = *(const bf16x8*)(rb_ + coff0);                                             \
    bfr[nh_][nt][1] = *(const bf16x8*)(rb_ + coff1);                                             \
  }
#define END_LOAD()                                          \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
  __builtin_amdgcn_sched_barrier(0);                        \
  __builtin_amdgcn_s_barrier()
#define COMPUTE(mh_, nh_)                                                                               \
  __builtin_amdgcn_s_setprio(1);                                                                        \
  if (!NO_MFMA)                                                                                         \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                      \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                   \
      _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                  \
        acc[(mh_) * MT + mt][(nh_) * 2 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                 \
            bfr[nh_][nt][ks], afr[mt][ks], acc[(mh_) * MT + mt][(nh_) * 2 + nt], 0, 0, 0);              \
  __builtin_amdgcn_s_setprio(0)

  // prologue: k-tile 0 into buffer 0, first two pieces of k-tile 1 into buffer 1
  STAGE_PIECES(0, NP, 0, 0);
  if (nk > 1 && !NO_DMA) {
    STAGE_SLOT(0, 1, BK);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();   // group B runs one barrier behind

  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    const bool more = !NO_DMA && t + 1 < nk;
    const bool more2 = !NO_DMA && t + 2 < nk;
    const bool rd = !NO_LDS || t == 0;
    // ---- phase 0: quadrant (m-half 0, n-half 0)
    if (rd) { LOAD_A(buf, 0); LOAD_B(buf, 0); }
    if (more) { STAGE_SLOT(1, buf ^ 1, (t + 1) * BK); }
    END_LOAD();
    COMPUTE(0, 0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 1: (m-half 0, n-half 1)
    if (rd) { LOAD_B(buf, 1); }
    if (more) { STAGE_SLOT(2, buf ^ 1, (t + 1) * BK); }
    END_LOAD();
    COMPUTE(0, 1);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: (m-half 1, n-half 1)
    if (rd) { LOAD_A(buf, 1); }
    if (more) { STAGE_SLOT(3, buf ^ 1, (t + 1) * BK); }
    END_LOAD();
    COMPUTE(1, 1);
    __builtin_amdgcn_s_barrier();
    // ---- phase 3: (m-half 1, n-half 0); no LDS reads; first two pieces of k-tile t+2 (its buffer, `buf`, was last
    // read in L2 above by both groups before the barrier that precedes this segment for either group)
    if (more2) { STAGE_SLOT(0, buf, (t + 2) * BK); }
    if (grp == 1) {
      if (more2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    COMPUTE(1, 0);
    if (grp == 0) {
      if (more2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();   // match group B's extra barrier
#undef STAGE_PIECES
#undef STAGE_SLOT
#undef LOAD_A
#undef LOAD_B
#undef END_LOAD
#undef COMPUTE

  // ---- epilogue through LDS: the MFMA layout gives each lane 4 consecutive n of ONE row, i.e. 32-byte (bf16)
  // pieces of 16 different rows per store instruction; measured, that caps the output stream at ~2.2 TB/s and costs
  // 75 us per 36928x2304 GEMM.  Each wave therefore parks its accumulators (fp32) in a private (16*MT)x64 LDS patch
  // (two halves of its rows), reads them back row-major and issues full-line stores: 16 lanes cover one
  // 64-column row segment (256 B fp32 / 128 B bf16), bias / GELU / residual are applied in this coalesced pass.
  constexpr int EP_ROWB = 272;                       // 64 fp32 + 16 B pad: conflict-free ds_write_b128
  char* ep = smem + w * (64 * EP_ROWB);
  const int er = lane >> 4, ec = (lane & 15) * 4;    // read-back: row within a group of 4, first of 4 columns
  const int ncol = n0 + wn * 64 + ec;
  f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
  if (p.bias && ncol < p.N) bias4 = *(const f32x4*)(p.bias + ncol);
  if (NO_EPI) return;                 // ablation: no epilogue at all
  // residual rows are requested one chunk (2*MT iterations = 8*MT rows) ahead of their use
  const bool col_ok = ncol < p.N && !(NO_STORE && m0 >= 0);   // ablation: LDS staging and arithmetic but no global stores
  f32x4 rres[2][2 * MT];
#define ROWS_OF(m_, orow_, rrow_)                                             \
  int orow_ = (m_), rrow_ = (m_);                                             \
  if (p.row_group > 0) {                                                      \
    const int g_ = (m_) / p.row_group, in_ = (m_) - g_ * p.row_group;         \
    orow_ = g_ * p.out_group_rows + p.out_row_off + in_;                      \
    rrow_ = p.res_periodic ? in_ : orow_;                                     \
  }
#define ISSUE_RES(c_)                                                                                        \
  if (HAS_RES) {                                                                                             \
    _Pragma("unroll") for (int it = 0; it < 2 * MT; ++it) {                                                  \
      const int m_ = m0 + wm * WR + (c_) * (8 * MT) + it * 4 + er;                                           \
      ROWS_OF(m_, o_, r_);                                                                                   \
      (void)o_;                                                                                              \
      rres[(c_) & 1][it] = (m_ < p.M && col_ok) ? *(const f32x4*)(p.res + (size_t)r_ * p.ldr + ncol)        \
                                                : f32x4{0.f, 0.f, 0.f, 0.f};                                 \
    }                                                                                                        \
  }
  // plain rows, no aux / pre-activation copy, bf16 output without residual or fp32 output: straight from registers
  // measured (tools/gemm_res_bench.py, VITCAP_GEMM_DIRECT_EPILOGUE=0/1): +4-5 % for the GELU epilogue (fc1), neutral for
  // plain bf16 (qkv), SLOWER for the fp32 + residual outputs (proj 0.078 -> 0.106 ms) -- so it is used for GELU only
  if (p.direct_epilogue && ACT == VITCAP_ACT_GELU_ERF && !OUT_F32 && !HAS_RES && !(EXTRAS && p.zout) && !(EXTRAS && p.aux) && p.row_group == 0 &&
      (p.N & 15) == 0 && (p.ldc & 7) == 0) {
    epilogue_direct<ACT, OUT_F32, HAS_RES, NI>(acc, p, m0 + wm * WR, n0 + wn * 64, lane, NO_STORE);
    return;
  }
  // bf16 output, plain rows, no residual: 8 columns per lane, one 16-byte store (8 lanes = one 128-byte line of the output
  // row) -- half the store instructions of the general path below.  The training extras ride along in the same shape: the
  // pre-activation copy (zout, fc1 forward) leaves as a second 16-byte store, the gelu'(aux) factor (fc2's input gradient)
  // arrives as 16-byte loads issued for a whole half BEFORE the LDS staging, so their latency hides behind it
  // (before: 8-byte loads used immediately, scalar gelu': 337 us per call at M = 36928, N = 3072)
  if constexpr (!OUT_F32 && !HAS_RES) {
    if (p.row_group == 0 && (p.N & 7) == 0 && (p.ldc & 7) == 0 && (!(EXTRAS && p.zout) || (p.ldz & 7) == 0) && (!(EXTRAS && p.aux) || (p.ldaux & 7) == 0)) {
      const int wr = lane >> 3, wc = (lane & 7) * 8;
      const int ncw = n0 + wn * 64 + wc;
      const bool okc = ncw < p.N && !(NO_STORE && m0 >= 0);
      f32x4 b_lo = f32x4{0.f, 0.f, 0.f, 0.f}, b_hi = b_lo;
      f32x4 cs_lo = f32x4{0.f, 0.f, 0.f, 0.f}, cs_hi = cs_lo;      // EXTRAS && p.colsum: this lane's 8 columns summed over its 4*MT rows
      if (p.bias && ncw < p.N) {
        b_lo = *(const f32x4*)(p.bias + ncw);
        b_hi = *(const f32x4*)(p.bias + ncw + 4);
      }
#pragma unroll
      for (int hm = 0; hm < 2; ++hm) {
        uint4 axv[2 * MT];
        if (EXTRAS && p.aux) {
#pragma unroll
          for (int it = 0; it < 2 * MT; ++it) {
            const int m = m0 + wm * WR + hm * (16 * MT) + it * 8 + wr;
            axv[it] = (m < p.M && okc) ? *(const uint4*)(p.aux + (size_t)m * p.ldaux + ncw) : uint4{0u, 0u, 0u, 0u};
          }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            *(f32x4*)(ep + (i * 16 + frow) * EP_ROWB + (j * 16 + fk * 4) * 4) = acc[hm * MT + i][j];
#pragma unroll
        for (int it = 0; it < 2 * MT; ++it) {
          const int rl = it * 8 + wr;
          f32x4 v0 = *(const f32x4*)(ep + rl * EP_ROWB + wc * 4);
          f32x4 v1 = *(const f32x4*)(ep + rl * EP_ROWB + wc * 4 + 16);
          const int m = m0 + wm * WR + hm * (16 * MT) + rl;
          const bool ok = m < p.M && okc;
          v0 += b_lo;
          v1 += b_hi;
          if (EXTRAS && p.aux) {
            const uint4 a = axv[it];
            v0 *= f32x4{__uint_as_float(a.x << 16), __uint_as_float(a.x & 0xffff0000u), __uint_as_float(a.y << 16),
                        __uint_as_float(a.y & 0xffff0000u)};
            v1 *= f32x4{__uint_as_float(a.z << 16), __uint_as_float(a.z & 0xffff0000u), __uint_as_float(a.w << 16),
                        __uint_as_float(a.w & 0xffff0000u)};
          }
          if (ACT == VITCAP_ACT_GELU_ERF) {
            if (EXTRAS && p.zout) {      // gelu'(pre-activation) from the same erfc evaluation, for the backward's `aux`
              f32x4 d0, d1;
              v0 = gelu_erf4_grad(v0, d0);
              v1 = gelu_erf4_grad(v1, d1);
              if (ok) {
                uint4 zo;
                zo.x = pack2bf(d0[0], d0[1]);
                zo.y = pack2bf(d0[2], d0[3]);
                zo.z = pack2bf(d1[0], d1[1]);
                zo.w = pack2bf(d1[2], d1[3]);
                *(uint4*)(p.zout + (size_t)m * p.ldz + ncw) = zo;
              }
            } else {
              v0 = gelu_erf4(v0);
              v1 = gelu_erf4(v1);
            }
          }
          if (ok) {
            uint4 o;
            o.x = pack2bf(v0[0], v0[1]);
            o.y = pack2bf(v0[2], v0[3]);
            o.z = pack2bf(v1[0], v1[1]);
            o.w = pack2bf(v1[2], v1[3]);
            *(uint4*)((bf16_t*)p.C + (size_t)m * p.ldc + ncw) = o;
            if (EXTRAS && p.colsum) {         // the ROUNDED values, as vitcap_colsum_bf16 over the stored output would add them
              cs_lo += f32x4{__uint_as_float(o.x << 16), __uint_as_float(o.x & 0xffff0000u), __uint_as_float(o.y << 16),
                             __uint_as_float(o.y & 0xffff0000u)};
              cs_hi += f32x4{__uint_as_float(o.z << 16), __uint_as_float(o.z & 0xffff0000u), __uint_as_float(o.w << 16),
                             __uint_as_float(o.w & 0xffff0000u)};
            }
          }
        }
      }
      if (EXTRAS && p.colsum) {
        // the 8 lanes wr = 0..7 with the same (lane & 7) hold the same 8 columns: butterfly over lane bits 3..5, then one atomic
        // per column and wave (two waves of the workgroup share each column: 512 atomics per 256x256 tile)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int o = 8; o <= 32; o <<= 1) {
            cs_lo[e] += __shfl_xor(cs_lo[e], o, 64);
            cs_hi[e] += __shfl_xor(cs_hi[e], o, 64);
          }
        }
        if (wr == 0 && okc) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            atomicAdd(p.colsum + ncw + e, cs_lo[e]);
            atomicAdd(p.colsum + ncw + 4 + e, cs_hi[e]);
          }
        }
      }
      return;
    }
  }
  // fused LayerNorm launches address C through a buffer descriptor from the tile's first row (plain rows only)
  const __amdgpu_buffer_rsrc_t ln_rc = vc_rsrc(LN ? (const char*)p.C + (size_t)m0 * p.ldc * 4 : nullptr, LN ? (long long)(p.M - m0) * p.ldc * 4 : 0);
  ISSUE_RES(0);
#pragma unroll
  for (int hm = 0; hm < 2; ++hm) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *(f32x4*)(ep + (i * 16 + frow) * EP_ROWB + (j * 16 + fk * 4) * 4) = acc[hm * MT + i][j];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      const int c = hm * 2 + ch;
      if (c + 1 < 4) { ISSUE_RES(c + 1); }
#pragma unroll
      for (int it = 0; it < 2 * MT; ++it) {
        const int rl = ch * (8 * MT) + it * 4 + er;
        f32x4 v = *(const f32x4*)(ep + rl * EP_ROWB + ec * 4);
        const int m = m0 + wm * WR + hm * (16 * MT) + rl;
        const bool ok = m < p.M && col_ok;
        ROWS_OF(m, orow, rrow);
        (void)rrow;
        v += bias4;
        if (EXTRAS && p.aux && ok) {
          const bf16x4 za = *(const bf16x4*)(p.aux + (size_t)orow * p.ldaux + ncol);
          v *= f32x4{(float)za[0], (float)za[1], (float)za[2], (float)za[3]};
        }
        if (ACT == VITCAP_ACT_GELU_ERF) {
          if (EXTRAS && p.zout) {
            f32x4 d;
            v = gelu_erf4_grad(v, d);
            if (ok) {
              uint2 zo;
              zo.x = pack2bf(d[0], d[1]);
              zo.y = pack2bf(d[2], d[3]);
              *(uint2*)(p.zout + (size_t)orow * p.ldz + ncol) = zo;
            }
          } else {
            v = gelu_erf4(v);
          }
        }
        if (HAS_RES) v += rres[c & 1][it];
        if (ok) {
          if (OUT_F32) {
            if constexpr (LN) {
              // write-through (sc1): the row block's last-arriving workgroup reads these rows back, wherever on the chip it runs
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vc_u32x4, v), ln_rc, (unsigned)((orow - m0) * p.ldc + ncol) * 4u, 0, 16);
            } else {
              *(f32x4*)((float*)p.C + (size_t)orow * p.ldc + ncol) = v;
            }
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
#undef ROWS_OF
  if constexpr (LN) {
    // ---- LayerNorm of the row block by the LAST of its column tiles to get here (any placement of the tiles over XCDs / CUs):
    // every wave's write-through stores acknowledged -> barrier -> one relaxed agent-scope ticket; the workgroup that draws the last
    // ticket reads the block's rows back with sc1 loads (they bypass this CU's L1 and the XCD's L2 lines other XCDs wrote behind)
    // and normalises them with the LayerNorm kernel's own row function.  The ticket counter goes back to zero for the next launch.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* s_flag = (int*)smem;
    if (tid == 0) {
      const int old = __hip_atomic_fetch_add(p.ln_cnt + cidx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = old == p.tiles_n - 1;
      if (last) __hip_atomic_store(p.ln_cnt + cidx, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *s_flag = last;
    }
    __syncthreads();
    if (*s_flag == 0) return;
    if (!p.ln_out && !p.ln_out_f) return;      // (timing experiments: publish only)
    // the other tiles' rows sit in memory (write-through); one agent-scope acquire drops whatever this CU's L1 holds of them, then
    // plain loads fetch them through the L2 at the ordinary rate (sc1 loads measured 5x slower here: 85 row blocks finish together
    // and the uncached path serves them one after the other)
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    const int rows = p.M - m0 < 64 * MT ? p.M - m0 : 64 * MT;
    constexpr int RB = 8;                      // rows in flight per wave: 8 waves x 8 rows x 3 KB = 196 KB of requests per pass
    for (int r0 = w * RB; r0 < rows; r0 += 8 * RB) {
      f32x4 v[RB][3];
#pragma unroll
      for (int j = 0; j < RB; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i)           // rows past the block's end: the descriptor's range check returns zeros for the last block only
          v[j][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ln_rc, (unsigned)((r0 + j) * p.ldc + i * 256 + lane * 4) * 4u, 0, LN_LOAD_AUX));
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        const int row = m0 + r0 + j;
        if (r0 + j < rows)
          ln_row(v[j], p.ln_g, p.ln_b, p.ln_eps, lane, p.ln_out ? p.ln_out + (size_t)row * D768 : nullptr,
                 p.ln_out_f ? p.ln_out_f + (size_t)row * D768 : nullptr);
      }
    }
  }
}

// The launch: n_big workgroups own 256-row tiles of rows [0, 256 * tiles_m); with MTS != 0 the workgroups behind them own
// (64*MTS)-row tiles of the remaining rows.  Workgroups are dispatched in blockIdx order (block b on XCD b % 8), so every XCD
// works through its share of the big tiles first and fills the last, partial round with the short ones.
// VGPR cap 224: two waves of this kernel per SIMD then leave 64 of the 512 registers (and 24 KB of LDS) free, which is what one wave
// of the decode step's HBM-bound attention kernel needs (60 VGPRs, 8 KB): in the 2-slot batch pipeline that kernel can become
// RESIDENT NEXT TO the GEMM's workgroups instead of waiting for CUs to drain -- K/V streaming overlaps the MFMA work.
#ifndef VC_GEMM256_VGPRS
#define VC_GEMM256_VGPRS 224
#endif
template <int ACT, int OUT_F32, bool HAS_RES, int PH, int MTS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(VC_GEMM256_VGPRS))) void gemm_nt_256_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int bid = blockIdx.x;
  const bool small = MTS != 0 && bid >= p.n_big;
  if (small) bid -= p.n_big;       // the XCD of block b is b % 8: subtracting a constant rotates the XCD ids, chunks stay whole
  const int tiles_m = small ? p.tiles_m_small : p.tiles_m;
  {
    // XCD-aware, bijective remap: the blocks of one XCD get a contiguous chunk of the region's tile list
    const int nwg = tiles_m * p.tiles_n;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // column groups of group_n tiles: the tiles an XCD runs at once are consecutive positions of its chunk, so they span few W
  // tiles (which stay in its L2 for the walk down M) and each A tile is fetched once for group_n consumers
  const int per_g = tiles_m * p.group_n;
  const int gi = bid / per_g, rem = bid - gi * per_g;
  const int gleft = p.tiles_n - gi * p.group_n;
  const int gw = gleft < p.group_n ? gleft : p.group_n;
  int tm = rem / gw;
  const int tn = gi * p.group_n + rem - tm * gw;
  // walk direction (common.h: vc_tls_walk_rev): row tiles last-to-first, and every other column group the other way round (a
  // group's pass over A ends where the next one starts)
  if ((p.rev != 0) != ((gi & 1) != 0)) tm = tiles_m - 1 - tm;
  if (!small) {
    gemm256_tile<ACT, OUT_F32, HAS_RES, PH, 4>(p, smem, tm * 256, tn * 256, tm);
  } else {
    if constexpr (MTS != 0) gemm256_tile<ACT, OUT_F32, HAS_RES, PH, MTS>(p, smem, p.tiles_m * 256 + tm * (64 * MTS), tn * 256, p.tiles_m + tm);
  }
}

// ---- tile plan of a launch: how many rows go to 256-row tiles, and the height of the tiles behind them ----------------------
// A launch of T equal tiles on C CUs costs ceil(T / C) tile times: 435 tiles (M = 36928, N = 768) pay 2 rounds for 1.7 rounds of
// work, 1305 (N = 2304) pay 6 for 5.1.  The plan keeps 256-row tiles for the rows that fill whole rounds and cuts the rest into
// 192- or 128-row tiles, chosen by simulating the dispatch (earliest free CU takes the next workgroup) with tile costs relative
// to the 256-row tile (a shorter tile has the same W traffic and per-phase overheads for fewer flops: measured, tools/gemm_bench.py).
struct TilePlan { int tm_big, mts, tm_small; };
TilePlan plan_tiles(int M, int tiles_n, int K) {
  static const int mix_env = [] { const char* e = getenv("VITCAP_GEMM_MIX"); return e ? atoi(e) : -1; }();   // 0 = off; 3 / 2 = force height
  static const int c3_env = [] { const char* e = getenv("VITCAP_GEMM_MIX_C3"); return e ? atoi(e) : 0; }();  // tile costs in permille (tuning)
  static const int c2_env = [] { const char* e = getenv("VITCAP_GEMM_MIX_C2"); return e ? atoi(e) : 0; }();
  static const int big_env = [] { const char* e = getenv("VITCAP_GEMM_MIX_BIG"); return e ? atoi(e) : -1; }(); // force the number of 256-row m-tiles
  static const int n_cu = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      return prop.multiProcessorCount;
    return 256;
  }();
  const int tm_full = (M + 255) / 256;
  TilePlan best{tm_full, 0, 0};
  if (mix_env == 0) return best;
  struct Key { int M, tn, K; };
  static std::mutex mu;
  static std::vector<std::pair<Key, TilePlan>> cache;
  {
    std::lock_guard<std::mutex> g(mu);
    for (const auto& e : cache)
      if (e.first.M == M && e.first.tn == tiles_n && e.first.K == K) return e.second;
  }
  // relative tile costs: main loop ~ (64 MT + overhead) cycles per phase, epilogue ~ MT; K = 768 tiles are about half epilogue
  const float c3 = c3_env > 0 ? c3_env * 1e-3f : 0.80f, c2 = c2_env > 0 ? c2_env * 1e-3f : 0.60f;
  float best_t = ceilf((float)(tm_full * tiles_n) / n_cu);
  const float need = best_t * 0.97f;                         // a plan must win 3 % to replace the plain grid
  const int span = (int)(2.5f * n_cu / tiles_n) + 2;         // rows worth ~2.5 rounds can move to short tiles
  std::vector<float> heap;
  for (int mts = 3; mts >= 2; --mts) {
    if (mix_env > 0 && mix_env != mts) continue;
    const float cs = mts == 3 ? c3 : c2;
    for (int tb = tm_full - 1; tb >= 0 && tb >= tm_full - span; --tb) {
      if (big_env >= 0 && tb != big_env) continue;
      const int rows_left = M - tb * 256;
      const int ts = (rows_left + 64 * mts - 1) / (64 * mts);
      const int nb = tb * tiles_n, ns = ts * tiles_n;
      // big tiles: CU j is free at (nb / C + (j < nb % C)); the short tiles go to the earliest free CU
      heap.assign(n_cu, 0.f);
      for (int j = 0; j < n_cu; ++j) heap[j] = -(float)(nb / n_cu + (j < nb % n_cu ? 1 : 0));
      std::make_heap(heap.begin(), heap.end());              // max-heap of negated free times
      float end = nb ? (float)((nb + n_cu - 1) / n_cu) : 0.f;
      for (int i = 0; i < ns; ++i) {
        std::pop_heap(heap.begin(), heap.end());
        const float f = -heap.back() + cs;
        heap.back() = -f;
        std::push_heap(heap.begin(), heap.end());
        end = f > end ? f : end;
      }
      if (end < need && end < best_t - 1e-4f) { best_t = end; best = TilePlan{tb, mts, ts}; }
    }
  }
  std::lock_guard<std::mutex> g(mu);
  cache.push_back({Key{M, tiles_n, K}, best});
  return best;
}

template <int ACT, int OUT_F32, bool HAS_RES, int PH, int MTS>
int launch_256_t(const GemmArgs& p, int nwg, hipStream_t s) {
  constexpr int smem = 8 * 64 * 272;   // max(2 x 64 KiB k-tile buffers, 8 x 17 KiB epilogue patches)
  auto kern = gemm_nt_256_kernel<ACT, OUT_F32, HAS_RES, PH, MTS>;
  VC_FUNC_SMEM(kern, smem);
  VC_LAUNCH_GEMM(kern, dim3(nwg), dim3(512), smem, s, p);
  VC_LAUNCH_CHECK("gemm_nt_256");
  return VITCAP_OK;
}

// mix: -1 = plan_tiles decides, 0 = 256-row tiles only, 3 / 2 = every tile 192 / 128 rows (tile-cost measurements)
template <int ACT, int OUT_F32, bool HAS_RES, int PH>
int launch_256(const GemmArgs& a, hipStream_t s, int mix = 0) {
  if constexpr ((PH & 512) != 0) mix = 0;          // fused-LayerNorm launches: 256-row tiles only
  GemmArgs p = a;
  p.tiles_n = (a.N + 255) / 256;
  p.group_n = tile_group_n(p.tiles_n);
  TilePlan pl{(a.M + 255) / 256, 0, 0};
  if constexpr (((PH >> 4) & 31) == 0) {
    if (a.aux || a.zout || a.colsum) mix = 0;
    if (mix < 0) pl = plan_tiles(a.M, p.tiles_n, a.K);
    else if (mix > 0) pl = TilePlan{0, mix, (a.M + 64 * mix - 1) / (64 * mix)};
  }
  p.tiles_m = pl.tm_big;
  p.tiles_m_small = pl.tm_small;
  p.n_big = pl.tm_big * p.tiles_n;
  const int nwg = p.n_big + pl.tm_small * p.tiles_n;
  if constexpr (((PH >> 4) & 31) == 0 && (PH & 512) == 0) {
    if (a.aux || a.zout || a.colsum) return launch_256_t<ACT, OUT_F32, HAS_RES, PH | 8, 0>(p, nwg, s);   // training extras: 256-row tiles only
    if (pl.mts == 3) return launch_256_t<ACT, OUT_F32, HAS_RES, PH, 3>(p, nwg, s);
    if (pl.mts == 2) return launch_256_t<ACT, OUT_F32, HAS_RES, PH, 2>(p, nwg, s);
  }
  return launch_256_t<ACT, OUT_F32, HAS_RES, PH, 0>(p, nwg, s);
}

template <int PH>
int dispatch_256(const GemmArgs& a, int act, int out_f32, hipStream_t s, int mix = 0) {
  const bool res = a.res != nullptr;
#define CASE(ACT_, OUT_)                                                  \
  if (act == ACT_ && out_f32 == OUT_)                                     \
    return res ? launch_256<ACT_, OUT_, true, PH>(a, s, mix) : launch_256<ACT_, OUT_, false, PH>(a, s, mix);
  CASE(VITCAP_ACT_NONE, 0)
  CASE(VITCAP_ACT_NONE, 1)
  CASE(VITCAP_ACT_GELU_ERF, 0)
  CASE(VITCAP_ACT_GELU_ERF, 1)
#undef CASE
  vitcap_set_error("gemm(256): unsupported act %d / out %d", act, out_f32);
  return VITCAP_EINVAL;
}


// ------------------------------------------------------------------------------------------------
// Persistent variant of the 256x256x64 role-alternating kernel: one workgroup per CU walks the tile list.
// Before a tile's epilogue the first k-tile of the NEXT tile is already being fetched by LDS-DMA into the free
// k-tile buffer (the epilogue stages through the other one, 8 KiB per wave, XOR-swizzled instead of padded),
// the output stores are not waited for, and there is no workgroup relaunch between tiles: the prologue latency
// and most of the store drain disappear behind useful work.
// ------------------------------------------------------------------------------------------------
template <int ACT, int OUT_F32, bool HAS_RES>
__global__ __launch_bounds__(512) void gemm_nt_256p_kernel(GemmArgs p) {
  constexpr int BM = 256, BN = 256, BK = 64;
  constexpr int A_BYTES = BM * BK * 2, BUF_BYTES = 2 * A_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = w >> 2;
  const int wm = w >> 2, wn = w & 3;
  const int nwg = p.tiles_m * p.tiles_n;
  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ (srow & 7);
  const int frow = lane & 15;
  const int fk = lane >> 4;
  const int coff0 = ((0 * 4 + fk) ^ (frow & 7)) * 16;
  const int coff1 = ((1 * 4 + fk) ^ (frow & 7)) * 16;
  const int a_base = (wm * 128 + frow) * 128;
  const int b_base = A_BYTES + (wn * 64 + frow) * 128;
  const int nk = p.K / BK;

  const bf16_t* aptr[4];
  const bf16_t* wptr[4];
#define TILE_COORDS(tile_, m0_, n0_)                                                         \
  do {                                                                                       \
    int bid_ = (tile_);                                                                      \
    const int q_ = nwg >> 3, r_ = nwg & 7, xcd_ = bid_ & 7;                                  \
    bid_ = (xcd_ < r_ ? xcd_ * (q_ + 1) : r_ * (q_ + 1) + (xcd_ - r_) * q_) + (bid_ >> 3);   \
    /* column groups of group_n tiles: the 32 tiles an XCD works on at once span few W tiles (which then live in  */ \
    /* its L2 for the whole walk down M) and 32/group_n A tiles, each fetched once for group_n consumers          */ \
    const int per_g_ = p.tiles_m * p.group_n;                                                \
    const int g_ = bid_ / per_g_;                                                            \
    const int rem_ = bid_ - g_ * per_g_;                                                     \
    const int left_ = p.tiles_n - g_ * p.group_n;                                            \
    const int gw_ = left_ < p.group_n ? left_ : p.group_n;                                   \
    const int tm_ = rem_ / gw_;                                                              \
    m0_ = tm_ * BM;                                                                          \
    n0_ = (g_ * p.group_n + rem_ - tm_ * gw_) * BN;                                          \
  } while (0)
#define TILE_PTRS(m0_, n0_)                                                                  \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
    int r_ = (m0_) + w * 32 + i * 8 + srow;                                                  \
    r_ = r_ < p.M ? r_ : p.M - 1;                                                            \
    aptr[i] = p.A + (size_t)r_ * p.lda + schunk * 8;                                         \
    int c_ = (n0_) + w * 32 + i * 8 + srow;                                                  \
    c_ = c_ < p.N ? c_ : p.N - 1;                                                            \
    wptr[i] = p.W + (size_t)c_ * p.ldw + schunk * 8;                                         \
  }
#define STAGE_A(buf_, k0_)                                                                   \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                              \
      glds16(aptr[i] + (k0_), smem + (buf_) * BUF_BYTES + (w * 32 + i * 8) * 128)
#define STAGE_W(buf_, k0_)                                                                   \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                              \
      glds16(wptr[i] + (k0_), smem + (buf_) * BUF_BYTES + A_BYTES + (w * 32 + i * 8) * 128)
#define STAGE2(base_, row0_, p0_, buf_, k0_)                                                 \
  _Pragma("unroll") for (int i = (p0_); i < (p0_) + 2; ++i)                                  \
      glds16((base_)[i] + (k0_), smem + (buf_) * BUF_BYTES + (row0_) + (w * 32 + i * 8) * 128)

  bf16x8 afr[4][2];
  bf16x8 bfr[2][2][2];
#define LOAD_A(buf_, mh_)                                                                        \
  _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                             \
    const char* ra_ = smem + (buf_) * BUF_BYTES + a_base + ((mh_) * 4 + mt) * 16 * 128;          \
    afr[mt][0] = *(const bf16x8*)(ra_ + coff0);                                                  \
    afr[mt][1] = *(const bf16x8*)(ra_ + coff1);                                                  \
  }
#define LOAD_B(buf_, nh_)                                                                        \
  _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                             \
    const char* rb_ = smem + (buf_) * BUF_BYTES + b_base + ((nh_) * 2 + nt) * 16 * 128;          \
    bfr[nh_][nt][0] = *(const bf16x8*)(rb_ + coff0);                                             \
    bfr[nh_][nt][1] = *(const bf16x8*)(rb_ + coff1);                                             \
  }
#define END_LOAD()                                          \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
  __builtin_amdgcn_sched_barrier(0);                        \
  __builtin_amdgcn_s_barrier()
#define COMPUTE(mh_, nh_)                                                                               \
  __builtin_amdgcn_s_setprio(1);                                                                        \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                      \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                    \
      _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                  \
        acc[(mh_) * 4 + mt][(nh_) * 2 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                  \
            bfr[nh_][nt][ks], afr[mt][ks], acc[(mh_) * 4 + mt][(nh_) * 2 + nt], 0, 0, 0);               \
  __builtin_amdgcn_s_setprio(0)

  int pb = 0;                       // LDS buffer holding k-tile 0 of the current tile
  int tile = blockIdx.x;
  int m0, n0;
  TILE_COORDS(tile, m0, n0);
  TILE_PTRS(m0, n0);
  STAGE_A(0, 0);
  STAGE_W(0, 0);

  for (; tile < nwg; tile += gridDim.x) {
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // every wave has left the previous epilogue (its staging patch lives in buffer pb^1) before anyone DMAs into it
    __builtin_amdgcn_s_barrier();
    if (nk > 1) {
      STAGE2(aptr, 0, 0, pb ^ 1, BK);
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // k-tile 0 (and the previous tile's stores) done
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();

    for (int t = 0; t < nk; ++t) {
      const int buf = (pb + t) & 1;
      const bool more = t + 1 < nk;
      const bool more2 = t + 2 < nk;
      LOAD_A(buf, 0);
      LOAD_B(buf, 0);
      if (more) { STAGE2(aptr, 0, 2, buf ^ 1, (t + 1) * BK); }
      END_LOAD();
      COMPUTE(0, 0);
      __builtin_amdgcn_s_barrier();
      LOAD_B(buf, 1);
      if (more) { STAGE2(wptr, A_BYTES, 0, buf ^ 1, (t + 1) * BK); }
      END_LOAD();
      COMPUTE(0, 1);
      __builtin_amdgcn_s_barrier();
      LOAD_A(buf, 1);
      if (more) { STAGE2(wptr, A_BYTES, 2, buf ^ 1, (t + 1) * BK); }
      END_LOAD();
      COMPUTE(1, 1);
      __builtin_amdgcn_s_barrier();
      if (more2) { STAGE2(aptr, 0, 0, buf, (t + 2) * BK); }
      if (grp == 1) {
        if (more2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      COMPUTE(1, 0);
      if (grp == 0) {
        if (more2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();

    // ---- prefetch k-tile 0 of the next tile into the buffer the last k-tile did NOT use
    const int ep_buf = (pb + nk - 1) & 1;      // last k-tile's buffer: free now, used for epilogue staging
    const int cm0 = m0, cn0 = n0;
    pb = ep_buf ^ 1;
    // ---- epilogue: 4 passes of 32 rows through this wave's 8 KiB patch (XOR-swizzled 16-byte chunks).
    // The residual rows of pass g+1 are requested before pass g is processed, so their HBM latency hides behind a
    // whole pass instead of stalling every 4 rows (proj / fc2: the fp32 residual + fp32 output stream IS the cost).
    // bf16 output without residual: a lane converts 8 columns and issues ONE 16-byte store (8 lanes = one 128-byte
    // line of the row) -- the store tail is issue-bound, so half the store instructions is what counts.
    char* ep = smem + ep_buf * BUF_BYTES + w * 8192;
    constexpr bool WIDE = !OUT_F32 && !HAS_RES;
    if constexpr (WIDE) {
      const int er = lane >> 3, ec = lane & 7;            // row within a group of 8, 8-column piece
      const int ncol = cn0 + wn * 64 + ec * 8;
      const bool col_ok = ncol < p.N;                      // N % 8 == 0 on this path (checked by the dispatcher)
      f32x4 bias_lo = f32x4{0.f, 0.f, 0.f, 0.f}, bias_hi = bias_lo;
      if (p.bias && col_ok) {
        bias_lo = *(const f32x4*)(p.bias + ncol);
        bias_hi = *(const f32x4*)(p.bias + ncol + 4);
      }
      if (tile + (int)gridDim.x < nwg) {
        TILE_COORDS(tile + (int)gridDim.x, m0, n0);
        TILE_PTRS(m0, n0);
        STAGE_A(pb, 0);
        STAGE_W(pb, 0);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int row = i * 16 + frow;
            *(f32x4*)(ep + row * 256 + (((j * 4 + fk) ^ (row & 15)) * 16)) = acc[g * 2 + i][j];
          }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int rl = it * 8 + er;
          f32x4 v0 = *(const f32x4*)(ep + rl * 256 + (((ec * 2) ^ (rl & 15)) * 16));
          f32x4 v1 = *(const f32x4*)(ep + rl * 256 + (((ec * 2 + 1) ^ (rl & 15)) * 16));
          const int m = cm0 + wm * 128 + g * 32 + rl;
          v0 += bias_lo;
          v1 += bias_hi;
          if (ACT == VITCAP_ACT_GELU_ERF) {
            v0 = gelu_erf4(v0);
            v1 = gelu_erf4(v1);
          }
          if (m < p.M && col_ok) {
            uint4 o;
            o.x = pack2bf(v0[0], v0[1]);
            o.y = pack2bf(v0[2], v0[3]);
            o.z = pack2bf(v1[0], v1[1]);
            o.w = pack2bf(v1[2], v1[3]);
            *(uint4*)((bf16_t*)p.C + (size_t)m * p.ldc + ncol) = o;
          }
        }
      }
    } else {
    const int er = lane >> 4, ec = lane & 15;
    const int ncol = cn0 + wn * 64 + ec * 4;
    const bool col_ok = ncol < p.N;
    f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias && col_ok) bias4 = *(const f32x4*)(p.bias + ncol);
    f32x4 rres[2][8];
#define ISSUE_RES(g_)                                                                                     \
  if (HAS_RES) {                                                                                          \
    _Pragma("unroll") for (int it = 0; it < 8; ++it) {                                                    \
      const int m_ = cm0 + wm * 128 + (g_) * 32 + it * 4 + er;                                            \
      rres[(g_) & 1][it] = (m_ < p.M && col_ok) ? *(const f32x4*)(p.res + (size_t)m_ * p.ldr + ncol)    \
                                                : f32x4{0.f, 0.f, 0.f, 0.f};                              \
    }                                                                                                     \
  }
    ISSUE_RES(0);                      // ahead of the next tile's DMA in the memory pipeline
    if (tile + (int)gridDim.x < nwg) {
      TILE_COORDS(tile + (int)gridDim.x, m0, n0);
      TILE_PTRS(m0, n0);
      STAGE_A(pb, 0);
      STAGE_W(pb, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g + 1 < 4) { ISSUE_RES(g + 1); }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = i * 16 + frow;
          *(f32x4*)(ep + row * 256 + (((j * 4 + fk) ^ (row & 15)) * 16)) = acc[g * 2 + i][j];
        }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int rl = it * 4 + er;
        f32x4 v = *(const f32x4*)(ep + rl * 256 + ((ec ^ (rl & 15)) * 16));
        const int m = cm0 + wm * 128 + g * 32 + rl;
        v += bias4;
        if (ACT == VITCAP_ACT_GELU_ERF) {
          v = gelu_erf4(v);
        }
        if (HAS_RES) v += rres[g & 1][it];
        if (m < p.M && col_ok) {
          if (OUT_F32) {
            *(f32x4*)((float*)p.C + (size_t)m * p.ldc + ncol) = v;
          } else {
            uint2 o;
            o.x = pack2bf(v[0], v[1]);
            o.y = pack2bf(v[2], v[3]);
            *(uint2*)((bf16_t*)p.C + (size_t)m * p.ldc + ncol) = o;
          }
        }
      }
    }
#undef ISSUE_RES
    }
  }
#undef TILE_COORDS
#undef TILE_PTRS
#undef STAGE_A
#undef STAGE_W
#undef STAGE2
#undef LOAD_A
#undef LOAD_B
#undef END_LOAD
#undef COMPUTE
}

template <int ACT, int OUT_F32, bool HAS_RES>
int launch_256p(const GemmArgs& a, hipStream_t s) {
  constexpr int smem = 2 * 2 * 256 * 64 * 2;
  auto kern = gemm_nt_256p_kernel<ACT, OUT_F32, HAS_RES>;
  VC_FUNC_SMEM(kern, smem);
  static const int n_cu = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      return prop.multiProcessorCount;
    return 256;
  }();
  GemmArgs p = a;
  p.tiles_m = (a.M + 255) / 256;
  p.tiles_n = (a.N + 255) / 256;
  const int nwg = p.tiles_m * p.tiles_n;
  p.group_n = tile_group_n(p.tiles_n);
  VC_LAUNCH_GEMM(kern, dim3(nwg < n_cu ? nwg : n_cu), dim3(512), smem, s, p);
  VC_LAUNCH_CHECK("gemm_nt_256p");
  return VITCAP_OK;
}

int dispatch_256p(const GemmArgs& a, int act, int out_f32, hipStream_t s) {
  const bool res = a.res != nullptr;
#define CASE(ACT_, OUT_)                                                  \
  if (act == ACT_ && out_f32 == OUT_)                                     \
    return res ? launch_256p<ACT_, OUT_, true>(a, s) : launch_256p<ACT_, OUT_, false>(a, s);
  CASE(VITCAP_ACT_NONE, 0)
  CASE(VITCAP_ACT_NONE, 1)
  CASE(VITCAP_ACT_GELU_ERF, 0)
  CASE(VITCAP_ACT_GELU_ERF, 1)
#undef CASE
  vitcap_set_error("gemm(256p): unsupported act %d / out %d", act, out_f32);
  return VITCAP_EINVAL;
}

// ------------------------------------------------------------------------------------------------
// Skinny kernel for the decode-step GEMMs (M = 2B or B rows: 64..256).  These are weight-streaming,
// latency-bound problems: the whole weight matrix is read once and there is almost no reuse, so LDS
// staging buys nothing (cdna guide "GEMV / M <= 16 decode weights ... load straight to VGPRs, deep
// unroll, late vmcnt").  Each wave owns a 32x32 output patch and loads its MFMA fragments directly
// from global memory (16 B per lane, the fragment layout IS the memory layout for K-contiguous
// operands); two register sets of 4 k-steps (128 k) each keep 16 loads in flight while the other set
// is multiplied.  Parallelism comes from narrow 32-column tiles and split-K: with split_k > 1 the
// kernel writes fp32 partial slabs P[kz][M][N] and the consumer (vitcap_sum_layernorm) reduces them.
// ------------------------------------------------------------------------------------------------
struct SkinnyArgs {
  GemmArgs g;
  int split_k;     // >= 1
  int kc;          // k range per split (multiple of 128)
  size_t slab;     // elements between partial slabs
};

template <int WAVES_M, int ACT, int OUT_F32, bool HAS_RES, bool PARTIAL>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(SkinnyArgs q) {
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int BM = 32 * WAVES_M, BN = 32 * WAVES_N;
  const GemmArgs& p = q.g;
  VC_LIVE_EXIT(p.live);
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = w / WAVES_N, wn = w % WAVES_N;
  const int n0 = blockIdx.x * BN + wn * 32;
  const int m0 = blockIdx.y * BM + wm * 32;
  const int kz = blockIdx.z;
  const int kbeg = kz * q.kc;
  const int frow = lane & 15, fk = lane >> 4;

  const bf16_t* ap[2];
  const bf16_t* wp[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int r = m0 + i * 16 + frow;
    r = r < p.M ? r : p.M - 1;
    ap[i] = p.A + (size_t)r * p.lda + kbeg + fk * 8;
    int c = n0 + i * 16 + frow;
    c = c < p.N ? c : p.N - 1;
    wp[i] = p.W + (size_t)c * p.ldw + kbeg + fk * 8;
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 a0[4][2], w0[4][2], a1[4][2], w1[4][2];
#define LOADSET(A_, W_, k0_)                                                   \
  _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                           \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                         \
      A_[s_][i_] = *(const bf16x8*)(ap[i_] + (k0_) + s_ * 32);                 \
      W_[s_][i_] = *(const bf16x8*)(wp[i_] + (k0_) + s_ * 32);                 \
    }                                                                          \
  }
#define MMASET(A_, W_)                                                                          \
  _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                            \
      _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                          \
        acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W_[s_][j_], A_[s_][i_], acc[i_][j_], 0, 0, 0);

  const int nch = q.kc / 128;
  LOADSET(a0, w0, 0);
  for (int c = 0; c < nch; c += 2) {
    if (c + 1 < nch) { LOADSET(a1, w1, (c + 1) * 128); }
    MMASET(a0, w0);
    if (c + 1 < nch) {
      if (c + 2 < nch) { LOADSET(a0, w0, (c + 2) * 128); }
      MMASET(a1, w1);
    }
  }
#undef LOADSET
#undef MMASET

#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + i * 16 + frow;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + j * 16 + fk * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[i][j];
      if (PARTIAL) {
        *(f32x4*)((float*)p.C + (size_t)kz * q.slab + (size_t)m * p.ldc + n) = v;
        continue;
      }
      if (p.bias) v += *(const f32x4*)(p.bias + n);
      if (ACT == VITCAP_ACT_GELU_ERF) {
        v = gelu_erf4(v);
      } else if (ACT == VITCAP_ACT_TANH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
      }
      if (HAS_RES) v += *(const f32x4*)(p.res + (size_t)m * p.ldr + n);
      if (OUT_F32) {
        *(f32x4*)((float*)p.C + (size_t)m * p.ldc + n) = v;
      } else {
        uint2 o;
        o.x = pack2bf(v[0], v[1]);
        o.y = pack2bf(v[2], v[3]);
        *(uint2*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
      }
    }
  }
}

template <int WAVES_M>
int launch_skinny(const GemmArgs& a, int act, int out_f32, int split_k, hipStream_t s) {
  constexpr int BM = 32 * WAVES_M, BN = 32 * (4 / WAVES_M);
  SkinnyArgs q;
  q.g = a;
  q.split_k = split_k;
  q.kc = a.K / split_k;
  q.slab = (size_t)a.M * a.ldc;
  dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, split_k);
  const bool res = a.res != nullptr;
#define SK(ACT_, OUT_, RES_, PART_)                                                                       \
  hipLaunchKernelGGL((gemm_skinny_kernel<WAVES_M, ACT_, OUT_, RES_, PART_>), grid, dim3(256), 0, s, q)
  if (split_k > 1) SK(0, 1, false, true);
  else if (act == VITCAP_ACT_NONE && !out_f32 && !res) SK(0, 0, false, false);
  else if (act == VITCAP_ACT_NONE && out_f32 && !res) SK(0, 1, false, false);
  else if (act == VITCAP_ACT_NONE && out_f32 && res) SK(0, 1, true, false);
  else if (act == VITCAP_ACT_NONE && !out_f32 && res) SK(0, 0, true, false);
  else if (act == VITCAP_ACT_GELU_ERF && !out_f32 && !res) SK(1, 0, false, false);
  else if (act == VITCAP_ACT_GELU_ERF && out_f32 && !res) SK(1, 1, false, false);
  else if (act == VITCAP_ACT_TANH && !out_f32 && !res) SK(2, 0, false, false);
  else if (act == VITCAP_ACT_TANH && out_f32 && !res) SK(2, 1, false, false);
  else {
    vitcap_set_error("gemm(skinny): unsupported epilogue act=%d out=%d res=%d", act, out_f32, (int)res);
    return VITCAP_EINVAL;
  }
#undef SK
  VC_LAUNCH_CHECK("gemm_skinny");
  return VITCAP_OK;
}

template <int WM, int WN, int ACT, int OUT_F32, bool HAS_RES>
int launch(const GemmArgs& a, hipStream_t s) {
  constexpr int BM = 32 * WM, BN = 32 * WN;
  constexpr int NST = (WM <= 2 && WN <= 2) ? 4 : 2;     // small tiles (decode shapes): 4 stages
  constexpr int smem = NST * (BM + BN) * 64 * 2;
  auto kern = gemm_nt_kernel<WM, WN, ACT, OUT_F32, HAS_RES, NST>;
  if (smem > 48 * 1024) VC_FUNC_SMEM(kern, smem);
  GemmArgs p = a;
  p.tiles_m = (a.M + BM - 1) / BM;
  p.tiles_n = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n, a.split_k > 1 ? a.split_k : 1), dim3(256), smem, s, p);
  VC_LAUNCH_CHECK("gemm_nt");
  return VITCAP_OK;
}

// "Resident" form for the decode-step shapes (M <= 256 rows, K a multiple of 768): NST = 12 stages = the WHOLE 768-long k
// range of the workgroup is requested by LDS-DMA in the prologue (one HBM/L2 round trip instead of 12 / (NST-1) dependent
// ones), then multiplied.  K > 768 is cut into K / 768 splits (blockIdx.y) that write fp32 partial slabs, summed by
// vitcap_sum_layernorm.  LDS: 12 * (BM + BN) * 128 B <= 160 KiB, i.e. BM + BN <= 96.
template <int WM, int WN, int ACT, int OUT_F32, bool HAS_RES>
int launch_resident(const GemmArgs& a, hipStream_t s) {
  constexpr int BM = 32 * WM, BN = 32 * WN, NST = 12;
  constexpr int smem = NST * (BM + BN) * 64 * 2;
  static_assert(smem <= 160 * 1024, "resident tile does not fit the LDS");
  auto kern = gemm_nt_kernel<WM, WN, ACT, OUT_F32, HAS_RES, NST>;
  VC_FUNC_SMEM(kern, smem);
  GemmArgs p = a;
  p.tiles_m = (a.M + BM - 1) / BM;
  p.tiles_n = (a.N + BN - 1) / BN;
  const int splits = a.K / 768;
  if (splits > 1) {
    p.split_k = splits;
    p.kt_per_split = 12;
    p.slab = (size_t)a.M * a.ldc;
  }
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n, splits), dim3(256), smem, s, p);
  VC_LAUNCH_CHECK("gemm_nt(resident)");
  return VITCAP_OK;
}

template <int WM, int WN>
int dispatch_resident(const GemmArgs& a, int act, int out_f32, hipStream_t s) {
  const bool res = a.res != nullptr;
  if (a.K > 768) return launch_resident<WM, WN, VITCAP_ACT_NONE, 1, false>(a, s);      // partial slabs only
  if (act == VITCAP_ACT_NONE && !out_f32 && !res) return launch_resident<WM, WN, VITCAP_ACT_NONE, 0, false>(a, s);
  if (act == VITCAP_ACT_NONE && out_f32 && !res) return launch_resident<WM, WN, VITCAP_ACT_NONE, 1, false>(a, s);
  if (act == VITCAP_ACT_NONE && out_f32 && res) return launch_resident<WM, WN, VITCAP_ACT_NONE, 1, true>(a, s);
  if (act == VITCAP_ACT_GELU_ERF && !out_f32 && !res) return launch_resident<WM, WN, VITCAP_ACT_GELU_ERF, 0, false>(a, s);
  if (act == VITCAP_ACT_GELU_ERF && out_f32 && !res) return launch_resident<WM, WN, VITCAP_ACT_GELU_ERF, 1, false>(a, s);
  vitcap_set_error("gemm(resident): unsupported epilogue act=%d out=%d res=%d", act, out_f32, (int)res);
  return VITCAP_EINVAL;
}

// Vocabulary GEMM + row statistics (vitcap_gemm_desc.rowstat): 64 x 64 tiles for the greedy decode step (M <= 256 rows: weights
// stream once, most workgroups win), 128 x 128 tiles for beam batches (M = images x beams rows)
template <int WM, int WN, int NST>
int launch_rowstat_t(const GemmArgs& a, hipStream_t s) {
  constexpr int BM = 32 * WM, BN = 32 * WN, smem = NST * (BM + BN) * 64 * 2;
  auto kern = gemm_nt_kernel<WM, WN, VITCAP_ACT_NONE, 1, false, NST, true>;
  VC_FUNC_SMEM(kern, smem);
  GemmArgs p = a;
  p.tiles_m = (a.M + BM - 1) / BM;
  p.tiles_n = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n, 1), dim3(256), smem, s, p);
  VC_LAUNCH_CHECK("gemm_nt(rowstat)");
  return VITCAP_OK;
}

int launch_rowstat(const GemmArgs& a, hipStream_t s) {
  // re-measured with the ring's counted waits working (docs/LAB_r01_r04.md 4.2 x): 3 / 6 / 8 stages and 64x128 tiles are all slower than
  // 64x64 x 4 stages in the decode loop (decode phase 5.10-5.18 ms against 5.17-5.26)
  return a.M <= 256 ? launch_rowstat_t<2, 2, 4>(a, s) : launch_rowstat_t<4, 4, 2>(a, s);
}

template <int WM, int WN>
int dispatch(const GemmArgs& a, int act, int out_f32, hipStream_t s) {
  const bool res = a.res != nullptr;
#define CASE(ACT_, OUT_)                                                         \
  if (act == ACT_ && out_f32 == OUT_)                                            \
    return res ? launch<WM, WN, ACT_, OUT_, true>(a, s) : launch<WM, WN, ACT_, OUT_, false>(a, s);
  CASE(VITCAP_ACT_NONE, 0)
  CASE(VITCAP_ACT_NONE, 1)
  CASE(VITCAP_ACT_GELU_ERF, 0)
  CASE(VITCAP_ACT_GELU_ERF, 1)
  CASE(VITCAP_ACT_TANH, 0)
  CASE(VITCAP_ACT_TANH, 1)
#undef CASE
  vitcap_set_error("gemm: unsupported act %d / out %d", act, out_f32);
  return VITCAP_EINVAL;
}

}  // namespace

// which kernel family runs a large GEMM under tile_hint `hint` (0 = auto, 5 = one tile per workgroup): -1 = the 8-wave kernel of this
// file, 0..2 = a form of the 4-wave kernel (gemm4w.hip); see the measurements at the call site in vitcap_gemm_ex
static int large_gemm_form(int M, int N, int hint, bool f32_or_res = true) {
  static const int env_set = getenv("VITCAP_GEMM_4W") != nullptr;
  static const int env_tiles = [] { const char* e = getenv("VITCAP_GEMM_4W"); return e ? atoi(e) : -1; }();
  static const int env_auto = [] { const char* e = getenv("VITCAP_GEMM_4W"); const char* c = e ? strchr(e, ',') : nullptr; return c ? atoi(c + 1) : (e ? atoi(e) : -1); }();
  const long long tiles256 = (long long)((M + 255) / 256) * ((N + 255) / 256);
  if (M < 2048) return -1;
  // VITCAP_GEMM_4W_TILES_N=<N>[,<form>] (experiments): under tile_hint 5 the 4-wave kernel (form 1 unless given) for outputs N wide only
  static const int only_n = [] { const char* e = getenv("VITCAP_GEMM_4W_TILES_N"); return e ? atoi(e) : 0; }();
  static const int only_form = [] { const char* e = getenv("VITCAP_GEMM_4W_TILES_N"); const char* c = e ? strchr(e, ',') : nullptr; return c ? atoi(c + 1) : 1; }();
  if (hint == 5 && only_n > 0 && tiles256 < 8 * 256) return N == only_n ? only_form : -1;
  // tile_hint 5 (another stream's kernels must slip in between tiles): the 8-wave kernel at every size.  Round 4 switched to the
  // persistent 4-wave form from 8 rounds of tiles on (B = 512 greedy +1.3 %); with two encoder parts in flight (round 5 default) that
  // is a tie there (4 096 / 4 122 vs 4 109 / 4 116 img/s) and it cost beam 5 x 256 -3 % (3 575 vs 3 692 img/s: the persistent grids lock
  // the 1 280-sequence decode chain out) -- profiles/r05_split_beam_ab.txt
  // What DOES pay inside the pipeline (round 5, 2 encoder parts, same box, interleaved; profiles/r05_pipeline_gemm_forms.txt): the 4-wave
  // kernel in its one-tile-per-workgroup form for the bf16-output GEMMs only -- qkv alone 3 900, fc1 alone 3 907 against 3 856 / 3 866
  // img/s all 8-wave -- while the fp32 + residual GEMMs (proj / fc2: 3 748) and the LDS-epilogue form (3 580 / 3 599) lose.
  (void)tiles256;
  static const int env_mix = [] { const char* e = getenv("VITCAP_GEMM_4W_MIX"); return e ? atoi(e) : 1; }();
  // (from 64 k rows per launch on -- B = 512, beam 5 x 256 -- the mix is a tie or slightly behind: 8-wave there)
  if (hint == 5) return env_set ? env_tiles : ((env_mix && !f32_or_res && M < 65536) ? 1 : -1);
  if (hint == 0) return env_set ? env_auto : 2;
  return -1;
}

extern "C" int vitcap_gemm_large_form(int M, int N, int K, int tile_hint) {
  // tile_hint | 0x100: the query is about a bf16-output GEMM without residual (under tile_hint 5 those run the 4-wave one-tile form)
  const bool plain_bf16 = (tile_hint & 0x100) != 0;
  tile_hint &= 0xff;
  int form = large_gemm_form(M, N, tile_hint, !plain_bf16);
  if (form < 0 || form > 2) return -1;
  if (K < 128) return -1;                                   // vc_4w_supports
  // launch_4w's downgrade rules for plain rows (what the engine's large GEMMs use): the persistent pipeline needs three k-tiles and
  // whole 256-column tiles, the register epilogue 16-byte bf16 stores
  if (form == 2 && (K < 192 || (N & 255) != 0)) form = 1;
  if (form != 0 && (N & 7) != 0) form = 0;
  return form + 10 * vc_4w_pick_mi(M, (N + 255) / 256, form);
}

extern "C" int vitcap_gemm_ex(const void* A, const void* W, const float* bias, const float* residual, void* C,
                              const vitcap_gemm_desc* d, const void* aux_bf16, int ldaux, void* zout_bf16, int ldz,
                              void* stream);

// Persistent vs one-tile-per-workgroup form of the 256x256 GEMM is a PER-CALL choice (vitcap_gemm_desc.tile_hint 12 / 5;
// 0 = auto = persistent where it wins).  Alone on the GPU the persistent form wins (+2..11 %); when another stream's small
// kernels should slip in between (the batch pipeline of ImageCaptioning.generate_async) the non-persistent form wins,
// because a persistent grid owns every CU for the whole GEMM (3277 vs 3133 img/s at B=64) -- the engine asks for it through
// vitcap_gen_opts.gemm_mode.  There is no process-wide switch.
extern "C" int vitcap_gemm_tile_plan(int M, int N, int K, int* plan3) {
  VC_REQUIRE(M > 0 && N > 0 && K > 0 && plan3, "gemm_tile_plan: bad arguments");
  const TilePlan pl = plan_tiles(M, (N + 255) / 256, K);
  plan3[0] = pl.tm_big;
  plan3[1] = pl.mts;
  plan3[2] = pl.tm_small;
  return VITCAP_OK;
}

extern "C" int vitcap_gemm_bias_act(const void* A, const void* W, const float* bias, const float* residual,
                                    void* C, const vitcap_gemm_desc* d, void* stream) {
  return vitcap_gemm_ex(A, W, bias, residual, C, d, nullptr, 0, nullptr, 0, stream);
}

extern "C" int vitcap_gemm_ex(const void* A, const void* W, const float* bias, const float* residual, void* C,
                              const vitcap_gemm_desc* d, const void* aux_bf16, int ldaux, void* zout_bf16, int ldz,
                              void* stream) {
  VC_REQUIRE(A && W && C && d, "gemm: null pointer");
  VC_REQUIRE(d->abi == VITCAP_ABI_VERSION, "gemm: descriptor built against ABI %d, this library is ABI %d", d->abi, VITCAP_ABI_VERSION);
  VC_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gemm: empty problem M=%d N=%d K=%d", d->M, d->N, d->K);
  VC_REQUIRE(d->K % 64 == 0, "gemm: K=%d must be a multiple of 64", d->K);
  VC_REQUIRE(d->N % 4 == 0 && d->ldc % 4 == 0, "gemm: N=%d and ldc=%d must be multiples of 4", d->N, d->ldc);
  VC_REQUIRE(d->lda % 8 == 0 && d->ldw % 8 == 0, "gemm: lda=%d ldw=%d must be multiples of 8", d->lda, d->ldw);
  VC_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)C & 15) == 0,
             "gemm: A/W/C must be 16-byte aligned");
  VC_REQUIRE(!residual || (d->ldr % 4 == 0 && ((uintptr_t)residual & 15) == 0), "gemm: residual misaligned");
  VC_REQUIRE(!bias || ((uintptr_t)bias & 15) == 0, "gemm: bias misaligned");
  GemmArgs a;
  a.A = (const bf16_t*)A;
  a.W = (const bf16_t*)W;
  a.bias = bias;
  a.res = residual;
  a.C = C;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.lda = d->lda; a.ldw = d->ldw; a.ldc = d->ldc; a.ldr = d->ldr;
  a.row_group = d->row_group; a.out_group_rows = d->out_group_rows;
  a.out_row_off = d->out_row_off; a.res_periodic = d->res_periodic;
  a.tiles_m = a.tiles_n = 0;
  a.aux = (const bf16_t*)aux_bf16; a.ldaux = ldaux;
  a.colsum = d->colsum;
  a.zout = (bf16_t*)zout_bf16; a.ldz = ldz;
  a.split_k = 1; a.kt_per_split = 0; a.slab = 0;
  a.live = d->live;
  a.rev = vc_tls_walk_rev ? 1 : 0;
  a.rowstat = (float*)d->rowstat;
  a.ln_g = a.ln_b = nullptr; a.ln_eps = 0.f; a.ln_out = nullptr; a.ln_out_f = nullptr; a.ln_cnt = nullptr;
  {
    static const int direct = [] { const char* e = getenv("VITCAP_GEMM_DIRECT_EPILOGUE"); return e ? atoi(e) : 1; }();
    a.direct_epilogue = direct;
  }
  VC_REQUIRE(!(aux_bf16 && d->act != VITCAP_ACT_NONE), "gemm: aux (gelu') epilogue needs act == none");
  VC_REQUIRE(!zout_bf16 || d->act == VITCAP_ACT_GELU_ERF, "gemm: zout stores the GELU's derivative and needs act == gelu_erf");
  VC_REQUIRE(!(aux_bf16 || zout_bf16) || (d->tile_hint != 3 && d->tile_hint != 12 && !(d->tile_hint >= 7 && d->tile_hint <= 17 && d->tile_hint != 13 && d->tile_hint != 14 && d->tile_hint != 15)),
             "gemm: tile_hint %d selects a kernel without the training extras (aux / zout)", d->tile_hint);
  hipStream_t s = (hipStream_t)stream;
  // tile_hint: 0 auto, 1 = 64x64, 2 = 128x128, 3 = 256x128 (3-stage), 4 = skinny (register-fed, optional split-K),
  //            5 = 256x256 role-alternating (the auto choice for M >= 2048)
  const int hint = d->tile_hint;
  const int split_k = d->split_k > 1 ? d->split_k : 1;
  const bool plain_rows = d->row_group == 0;
  if (d->ln_out_bf16 || d->ln_out_f32) {
    // LayerNorm of the finished rows: inside the kernel where the launch form has the last-arriver pass, a LayerNorm launch behind
    // the GEMM otherwise -- the same row function either way, so the outputs do not depend on which
    VC_REQUIRE(d->ln_gamma && d->ln_beta, "gemm(ln): ln_gamma / ln_beta missing");
    VC_REQUIRE(d->N == 768 && d->ldc == 768 && d->out_dtype == VITCAP_OUT_F32 && d->act == VITCAP_ACT_NONE && plain_rows && split_k == 1 &&
                   !aux_bf16 && !zout_bf16 && !d->rowstat && !d->colsum,
               "gemm(ln): needs N == ldc == 768, fp32 output, no activation / row remap / split-K / training extras");
    // in-kernel only on request (ln_counters given) and under the 8-wave one-tile-per-workgroup form.  MEASURED A LOSS (docs/LAB_r01_r04.md
    // 4.3: a row block's last arriver pulls its 786 KB back at ~20 GB/s, 40 us per block on one CU, against 31 us for the LayerNorm
    // kernel over ALL rows on the whole chip): the engine does not ask for it unless VITCAP_GEMM_LN_FUSE=1
    const bool fused = d->ln_counters && d->M >= 2048 && hint == 5 && large_gemm_form(d->M, d->N, hint) < 0;
    if (fused) {
      a.ln_g = d->ln_gamma; a.ln_b = d->ln_beta; a.ln_eps = d->ln_eps;
      a.ln_out = (bf16_t*)d->ln_out_bf16; a.ln_out_f = d->ln_out_f32; a.ln_cnt = d->ln_counters;
      static const int dbg = [] { const char* e = getenv("VITCAP_GEMM_LN_DEBUG"); return e ? atoi(e) : 0; }();   // 1: no LayerNorm pass (timing; wrong results)
      if (dbg == 1) { a.ln_out = nullptr; a.ln_out_f = nullptr; }
      return residual ? launch_256<VITCAP_ACT_NONE, 1, true, 4 | 512>(a, s, 0) : launch_256<VITCAP_ACT_NONE, 1, false, 4 | 512>(a, s, 0);
    }
    vitcap_gemm_desc d2 = *d;
    d2.ln_out_bf16 = nullptr; d2.ln_out_f32 = nullptr;
    const int rc = vitcap_gemm_ex(A, W, bias, residual, C, &d2, aux_bf16, ldaux, zout_bf16, ldz, stream);
    if (rc != VITCAP_OK) return rc;
    // the LayerNorm reads the rows the GEMM has just written: it walks them the other way round (common.h: vc_tls_walk_rev; only the
    // engine ever sets the flag, and only there does this flip have an effect worth having)
    const bool dir = vc_tls_walk_rev;
    if (vc_tls_zigzag && d->M >= 2048) vc_tls_walk_rev = !dir;
    const int rc2 = vitcap_layernorm_fwd((const float*)C, d->ldc, d->ln_gamma, d->ln_beta, d->ln_eps, d->ln_out_bf16, d->ln_out_f32, d->M, 768, stream);
    vc_tls_walk_rev = dir;
    return rc2;
  }
  if (d->rowstat) {
    VC_REQUIRE(d->out_dtype == VITCAP_OUT_F32 && d->act == VITCAP_ACT_NONE && !residual && plain_rows && split_k == 1 && !aux_bf16 && !zout_bf16,
               "gemm(rowstat): fp32 output, no activation / residual / split-K");
    return launch_rowstat(a, s);
  }
  if (d->colsum) {
    // column sums of the finished bf16 output ride in the 256x256 kernel's 16-byte-store epilogue (training extras build)
    VC_REQUIRE(d->out_dtype == VITCAP_OUT_BF16 && !residual && plain_rows && split_k == 1 && d->M >= 2048 && d->N % 8 == 0 && d->ldc % 8 == 0 &&
                   (!aux_bf16 || ldaux % 8 == 0) && (!zout_bf16 || ldz % 8 == 0) && d->act != VITCAP_ACT_TANH,
               "gemm(colsum): needs bf16 output, no residual / row remap / split-K, M >= 2048, N, ldc (ldaux, ldz) multiples of 8");
    // the persistent 4-wave kernel's register epilogue carries the column sums too (round 5): on request (tile_hint 42), or where the
    // automatic choice allows it (vc_4w_extras_auto: off by default, slower inside the training step)
    if ((hint == 42 || (hint == 0 && vc_4w_extras_auto(a) && large_gemm_form(d->M, d->N, hint, false) == 2)) && vc_4w_supports(a, d->act))
      return vc_dispatch_4w(a, d->act, d->out_dtype, s, 2);
    return dispatch_256<4>(a, d->act, d->out_dtype, s, 0);
  }
  if (hint == 23 || hint == 24) {
    // the resident form's contract on the 4-stage ring: K > 768 -> K/768 raw fp32 partial slabs in C = [K/768][M][ldc], one per
    // 768-long k range (blockIdx.y); 64x32 (23) or 32x32 (24) tiles
    VC_REQUIRE(d->K % 768 == 0 && d->K > 768 && plain_rows && !aux_bf16 && !zout_bf16 && d->out_dtype == VITCAP_OUT_F32 && !bias && !residual &&
                   d->act == VITCAP_ACT_NONE,
               "gemm(ring slabs): needs K a multiple of 768 above it, plain rows, raw fp32 slabs (no bias / residual / activation)");
    a.split_k = d->K / 768;
    a.kt_per_split = 12;
    a.slab = (size_t)d->M * d->ldc;
    return hint == 23 ? dispatch<2, 1>(a, VITCAP_ACT_NONE, 1, s) : dispatch<1, 1>(a, VITCAP_ACT_NONE, 1, s);
  }
  if (hint == 20 || hint == 21 || hint == 22) {
    // resident whole-K form (decode-step shapes); K > 768 -> K/768 fp32 partial slabs in C = [K/768][M][ldc]
    VC_REQUIRE(d->K % 768 == 0 && plain_rows && !aux_bf16 && !zout_bf16, "gemm(resident): needs K %% 768 == 0, plain rows, no training extras");
    VC_REQUIRE(d->K == 768 || (d->out_dtype == VITCAP_OUT_F32 && !bias && !residual && d->act == VITCAP_ACT_NONE),
               "gemm(resident): K > 768 writes raw fp32 partial slabs");
    if (hint == 20) return dispatch_resident<2, 1>(a, d->act, d->out_dtype, s);     // 64 x 32 tiles
    if (hint == 21) return dispatch_resident<1, 1>(a, d->act, d->out_dtype, s);     // 32 x 32
    return dispatch_resident<1, 2>(a, d->act, d->out_dtype, s);                     // 32 x 64
  }
  if (split_k > 1 && d->M > 256) {
    // weight-gradient shape: few output tiles, very long K -> 128x128 tiles with ragged split-K into fp32 slabs
    VC_REQUIRE(d->out_dtype == VITCAP_OUT_F32 && plain_rows && !aux_bf16 && !zout_bf16, "gemm: split-K writes fp32 partial slabs only");
    const int nk = d->K / 64;
    a.split_k = split_k;
    a.kt_per_split = (nk + split_k - 1) / split_k;
    a.slab = (size_t)d->M * d->ldc;
    return dispatch<4, 4>(a, VITCAP_ACT_NONE, 1, s);
  }
  if (split_k > 1) {
    VC_REQUIRE(d->K % (128 * split_k) == 0, "gemm: K=%d not divisible into %d splits of multiples of 128", d->K, split_k);
    VC_REQUIRE(d->out_dtype == VITCAP_OUT_F32 && plain_rows, "gemm: split-K writes fp32 partial slabs only");
  }
  // measured at B=64 (tools/decode_bench.py): without split-K the LDS-staged 64x64 kernel is ~2x faster than the
  // register-fed one (7.4 vs 14.6 us for 128x2304x768), so the skinny kernel is used for split-K only.
  if (split_k > 1 || hint == 4) {
    VC_REQUIRE(d->K % 128 == 0 && plain_rows, "gemm(skinny): needs K %% 128 == 0 and no row remap");
    return d->M <= 64 ? launch_skinny<2>(a, d->act, d->out_dtype, split_k, s)
                      : launch_skinny<4>(a, d->act, d->out_dtype, split_k, s);
  }
  if (hint == 13) return dispatch<2, 1>(a, d->act, d->out_dtype, s);
  if (hint == 14) return dispatch<1, 1>(a, d->act, d->out_dtype, s);
  if (hint == 15) return dispatch<1, 2>(a, d->act, d->out_dtype, s);
  // decode shapes (M = 2 x sequences): weights stream from HBM once, the kernel is latency-bound, so the smallest
  // tiles (most workgroups, most bytes in flight) win: cold-weight 128x2304x768 takes 10.8 / 9.4 / 8.9 us with
  // 64x64 / 64x32 / 32x32 tiles (tools/cold_gemm.py), against a ~4.5 us launch floor
  if (hint == 0 && d->M <= 128) return dispatch<1, 1>(a, d->act, d->out_dtype, s);
  if (hint == 1 || (hint == 0 && d->M <= 256)) return dispatch<2, 2>(a, d->act, d->out_dtype, s);
  if (hint == 0 && d->act != VITCAP_ACT_TANH) {
    // Auto tile choice for 256 < M: the tile whose grid costs least under a linear model fitted to tools/gemm_bench.py on the
    // hot shapes (M = 577 .. 36928; N = 768 / 2304 / 3072; K = 768 / 3072; us).  64x64: 4 + W * c; 128x128: max(one tile, 9 + W * c);
    // 256x256: ceil(W / 256) * one tile.  A 256x256 grid of 30 workgroups (N = 768 at 1..8 images) leaves 7/8 of the chip idle
    // for a whole tile time (fc2 at 4 images: 65 us, against 28 us on 64x64 tiles); one image's encoder 3.3 -> 2.6 ms.
    const float kf = d->K > 768 ? (float)(d->K - 768) / 2304.f : 0.f;
    const auto grid = [&](int t) { return (float)(((d->M + t - 1) / t) * ((d->N + t - 1) / t)); };
    const float t64 = 4.f + grid(64) * (0.0165f + kf * 0.0385f);
    const float t128 = fmaxf(16.f + kf * 26.f, 9.f + grid(128) * (0.0375f + kf * 0.0745f));
    const float t256 = d->M >= 2048 ? ceilf(grid(256) / 256.f) * (23.f + kf * 57.f) : 1e30f;
    if (t64 < t128 && t64 < t256) return dispatch<2, 2>(a, d->act, d->out_dtype, s);
    if (t128 < t256) return dispatch<4, 4>(a, d->act, d->out_dtype, s);
  }
  if (hint == 2 || (hint == 0 && (d->M < 2048 || d->act == VITCAP_ACT_TANH)))
    return dispatch<4, 4>(a, d->act, d->out_dtype, s);
  if (hint == 3) return dispatch_big(a, d->act, d->out_dtype, s);
  // the persistent kernel's bf16 epilogue stores 8 columns (16 bytes) per lane
  const bool wide_ok = d->out_dtype == VITCAP_OUT_F32 || residual || (d->N % 8 == 0 && d->ldc % 8 == 0 && ((uintptr_t)C & 15) == 0);
  if (hint == 12 && wide_ok) return dispatch_256p(a, d->act, d->out_dtype, s);
  {
    // The 4-wave kernel (gemm4w.hip) behind the two production hints, measured end to end (docs/LAB_r01_r04.md section 4.3):
    //   auto (one stream): its persistent form, +1.8 % images/s at B = 64 and B = 512 against the 8-wave kernel + planned tile mix;
    //   tile_hint 5 (the 2-slot pipeline: another stream's small kernels must slip in between tiles): the 8-wave kernel stays
    //   (B = 64: 3802 vs 3703 img/s one-tile 4-wave, 3517 persistent -- a persistent grid owns every CU for the whole GEMM, and a
    //   512-register workgroup leaves no room for a co-resident decode wave), at every size since round 5 (large_gemm_form).
    // VITCAP_GEMM_4W = "<form for tile_hint 5>,<form for auto>" overrides (-1 = 8-wave kernel, 0..2 = form; experiments).
    const int form = large_gemm_form(d->M, d->N, hint, d->out_dtype == VITCAP_OUT_F32 || residual != nullptr);
    // training extras (aux / zout / colsum) ride in the persistent form's register epilogue only (round 5), and only where
    // vc_4w_extras_auto says so
    const bool extras = aux_bf16 || zout_bf16 || d->colsum;
    if (form >= 0 && form <= 2 && d->M >= 2048 && (!extras || (form == 2 && d->out_dtype == VITCAP_OUT_BF16 && vc_4w_extras_auto(a))) &&
        vc_4w_supports(a, d->act))
      return vc_dispatch_4w(a, d->act, d->out_dtype, s, form);
  }
  if (hint >= 40 && hint <= 42) {
    VC_REQUIRE(!(aux_bf16 || zout_bf16 || d->colsum) || (hint == 42 && d->out_dtype == VITCAP_OUT_BF16),
               "gemm(4-wave): aux / zout / colsum need the persistent form (tile_hint 42) and a bf16 output");
    return vc_dispatch_4w(a, d->act, d->out_dtype, s, hint - 40);
  }   // 4 waves x 128x128, one wave per SIMD (gemm4w.hip): 40 LDS epilogue, 41 register epilogue, 42 persistent
  if (hint == 30) return dispatch_256<4>(a, d->act, d->out_dtype, s, 3);    // every tile 192 x 256 (tile-cost measurement)
  if (hint == 31) return dispatch_256<4>(a, d->act, d->out_dtype, s, 2);    // every tile 128 x 256
  if (hint == 5 || hint == 32) return dispatch_256<4>(a, d->act, d->out_dtype, s, 0);    // 256 x 256 tiles only (no short tail tiles)
  if (hint == 33) return dispatch_256<4>(a, d->act, d->out_dtype, s, -1);   // 256-row tiles + the planned short tiles (what auto picks)
  if (hint == 7) return launch_256<0, 0, false, 4 + 16 * 1>(a, s);   // ablation: no DMA in the loop
  if (hint == 8) return launch_256<0, 0, false, 4 + 16 * 2>(a, s);   // ablation: no ds_read in the loop
  if (hint == 9) return launch_256<0, 0, false, 4 + 16 * 4>(a, s);   // ablation: no MFMA
  if (hint == 10) return launch_256<0, 0, false, 4 + 16 * 3>(a, s);  // ablation: MFMA + barriers only
  if (hint == 11) return launch_256<0, 0, false, 4 + 16 * 6>(a, s);  // ablation: DMA + barriers only
  if (hint == 16) return launch_256<0, 0, false, 4 + 16 * 8>(a, s);  // ablation: no global stores in the epilogue
  if (hint == 17) return launch_256<0, 0, false, 4 + 16 * 16>(a, s); // ablation: no epilogue
  // measured (tools/gemm_bench.py 5,12): the persistent variant wins without a residual operand (qkv +11 %, fc1 +6 %);
  // with one, its residual rows are requested a pass ahead and before the next tile's DMA.
  static const int env_persistent = [] { const char* e = getenv("VITCAP_GEMM_PERSISTENT"); return e ? atoi(e) : -1; }();
  // auto = one tile per workgroup: since the direct (LDS-free) epilogue and the column-group tile order the persistent form no
  // longer wins (tools/gemm_bench.py 0,5 at M = 36928 / 295424: qkv 148 / 1072 us persistent vs 145 / 1041, fc1 201 / 1645 vs
  // 192 / 1537, fc2 and proj equal; end to end +1.0 % at B = 64, +2.4 % at B = 512).  tile_hint 12 still selects it.
  const int use_persistent = env_persistent >= 0 ? env_persistent : 0;
  // with a residual: long-K shapes (fc2) gain from the persistent form, short-K ones (proj) do not (gemm_res_bench.py)
  if (hint == 0 && use_persistent && wide_ok && !aux_bf16 && !zout_bf16 && d->row_group == 0 && (!residual || d->K > 1024))
    return dispatch_256p(a, d->act, d->out_dtype, s);
  return dispatch_256<4>(a, d->act, d->out_dtype, s, -1);      // 256 x 256 tiles, short tiles in the last round where the plan wins
}
