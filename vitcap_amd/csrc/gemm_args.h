// Internal: the argument block shared by the NT GEMM kernels (gemm.hip, gemm4w.hip) and their launch macro.
#pragma once
#include "common.h"
#include <hip/hip_ext.h>

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* W;
  const float* bias;
  const float* res;
  void* C;
  int M, N, K;
  int lda, ldw, ldc, ldr;
  int row_group, out_group_rows, out_row_off, res_periodic;
  int tiles_m, tiles_n;
  int n_big, tiles_m_small;     // mixed launch of the 256-wide kernel: workgroups [0, n_big) own 256-row tiles, the rest short ones
  // training extras
  float* colsum;       // training: colsum[n] += sum over rows of the finished bf16 outputs (bias gradient of the layer whose
                       // output gradient this GEMM produces); fp32 [N], atomics; 256x256 kernel, bf16 output, plain rows
  const bf16_t* aux;   // epilogue multiplies by aux (the stored gelu' factor: backward of the MLP activation); bf16 [M][ldaux]
  int ldaux;
  bf16_t* zout;        // gelu'(pre-activation) for the backward (act == GELU only; the factor a later launch takes as `aux`); bf16 [M][ldz]
  int ldz;
  int direct_epilogue; // 256x256 kernels: register-transpose epilogue (1) or the LDS-staged one (0)
  int split_k;         // > 1: blockIdx.y = split, ragged k-tile ranges, fp32 partial slabs, no epilogue
  int kt_per_split;
  size_t slab;
  int group_n;                 // persistent kernel: width of a column group in tiles (tile walk order)
  const int32_t* live;         // decode loop: return at entry once *live == 0 (vitcap_gemm_desc.live)
  float* rowstat;              // ROWSTAT kernels: per (row, 32-column piece) {max, argmax column, sum exp(x - max), 0}
  // fused LayerNorm of the finished rows (N == 768, fp32 output; vitcap_gemm_desc.ln_*): the last of a row block's column tiles to
  // finish normalises the block
  const float* ln_g;
  const float* ln_b;
  float ln_eps;
  bf16_t* ln_out;              // [M][768] bf16 or null
  float* ln_out_f;             // [M][768] fp32 or null
  int32_t* ln_cnt;             // one counter per row block, zero on entry, zero again on exit
  int rev;                     // 256-row-tile kernels: walk the tile list last-to-first (vc_tls_walk_rev / tile_hint flag 0x1000)
};

// launch of a large-tile GEMM: with kernel-bound timing events when the engine's timing run asked for them (common.h)
#define VC_LAUNCH_GEMM(kern, grid, block, smem, s, p)                                                              \
  do {                                                                                                             \
    if (vc_tls_kev_start) {                                                                                        \
      hipExtLaunchKernelGGL(kern, grid, block, smem, s, vc_tls_kev_start, vc_tls_kev_stop, 0, p);                 \
      vc_tls_kev_used = true;                                                                                      \
    } else {                                                                                                       \
      hipLaunchKernelGGL(kern, grid, block, smem, s, p);                                                           \
    }                                                                                                              \
  } while (0)

// width (in 256-column tiles) of the column groups the 256x256 kernels walk (gemm.hip)
int vc_tile_group_n(int tiles_n);
// the 4-wave / 512-register 256x256 kernel (gemm4w.hip); mix = 0: 256-row tiles only
// form: 0 = one tile per workgroup + LDS epilogue, 1 = one tile per workgroup + register epilogue, 2 = persistent
int vc_dispatch_4w(const GemmArgs& a, int act, int out_f32, hipStream_t s, int form);
bool vc_4w_supports(const GemmArgs& a, int act);
bool vc_4w_extras_auto(const GemmArgs& a);     // policy: may the automatic choice give a launch with aux / zout / colsum to the 4-wave kernel
// m-tiles of 16 rows per wave (8 / 7 / 6 -> 256- / 224- / 192-row tiles) the 4-wave kernel picks for a launch
int vc_4w_pick_mi(int M, int tiles_n, int form);
