// Device back half of the JPEG decoder (include/vitcap_hip.h: vitcap_jpeg_backhalf): coefficient blocks from the host
// entropy decoder (jpeg_host.cpp) -> dequantisation + 8x8 inverse DCT -> component planes -> chroma upsampling + YCbCr -> RGB -> the
// uint8 HWC image that vitcap_image_preproc resizes.  It restates libjpeg(-turbo)'s DEFAULT decompression path, which is integer
// arithmetic end to end, so the pixels are bit-identical to Pillow's decoder (tests/test_hip_jpeg.py):
//   jidctint.c jpeg_idct_islow (CONST_BITS 13, PASS1_BITS 2, range-limit table), jdsample.c h2v1 / h2v2 fancy upsampling with
//   jdmainct.c's replicated context rows at the top / bottom edge, jdcolor.c ycc_rgb_convert (SCALEBITS 16).
// HBM-bound byte work: 3 B per pixel of coefficients in, 1.5 B of planes out and back in, 3 B of RGB out; a 640 x 480 image is 7 200
// independent blocks, a batch of 64 half a million threads -- no tiling to speak of, coalescing is what matters (adjacent threads take
// adjacent blocks of a block row: the 8-byte row segments they store are contiguous).
#include "common.h"
#include "../../include/vitcap_jpeg.h"

#include <string.h>

#include <vector>

namespace {

struct JpegPlan {
  const int16_t* coefs;
  uint8_t* rgb;
  long long plane_off[3];       // byte offsets of the component planes in the workspace (pitch = 8 * blocks_w)
  int pitch;                    // output row pitch in bytes
  int width, height, ncomp, hs0, vs0;
  int blocks_w[3], blocks_h[3], samp_w[3], samp_h[3], block0[3], nblocks;
  unsigned short qt[3][64];
};

constexpr int CONST_BITS = 13, PASS1_BITS = 2;
constexpr int F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373,
              F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069, F_2_053119869 = 16819,
              F_2_562915447 = 20995, F_3_072711026 = 25172;

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
// v_mul_i32_i24 runs at the full VALU rate (v_mul_lo_u32 at a quarter of it); exact while both operands fit 24 signed bits: constants
// are < 2^15, dequantised coefficients and first-pass results of a valid stream < 2^16 (libjpeg-turbo's SIMD path keeps them in int16)
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }

// one 1-D pass of jpeg_idct_islow: inputs i0..i7 -> o[0..7] descaled by SHIFT
template <int SHIFT>
__device__ __forceinline__ void idct_1d(int i0, int i1, int i2, int i3, int i4, int i5, int i6, int i7, int (&o)[8]) {
  int z2 = i2, z3 = i6;
  int z1 = mul24(z2 + z3, F_0_541196100);
  int tmp2 = z1 + mul24(z3, -F_1_847759065);
  int tmp3 = z1 + mul24(z2, F_0_765366865);
  int tmp0 = (i0 + i4) * (1 << CONST_BITS);
  int tmp1 = (i0 - i4) * (1 << CONST_BITS);
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = i7; tmp1 = i5; tmp2 = i3; tmp3 = i1;
  z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
  int z4 = tmp1 + tmp3;
  const int z5 = mul24(z3 + z4, F_1_175875602);
  tmp0 = mul24(tmp0, F_0_298631336); tmp1 = mul24(tmp1, F_2_053119869); tmp2 = mul24(tmp2, F_3_072711026); tmp3 = mul24(tmp3, F_1_501321110);
  z1 = mul24(z1, -F_0_899976223); z2 = mul24(z2, -F_2_562915447); z3 = mul24(z3, -F_1_961570560); z4 = mul24(z4, -F_0_390180644);
  z3 += z5; z4 += z5;
  tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
  o[0] = descale(tmp10 + tmp3, SHIFT); o[7] = descale(tmp10 - tmp3, SHIFT);
  o[1] = descale(tmp11 + tmp2, SHIFT); o[6] = descale(tmp11 - tmp2, SHIFT);
  o[2] = descale(tmp12 + tmp1, SHIFT); o[5] = descale(tmp12 - tmp1, SHIFT);
  o[3] = descale(tmp13 + tmp0, SHIFT); o[4] = descale(tmp13 - tmp0, SHIFT);
}

// sample_range_limit + CENTERJSAMPLE indexed by (x & RANGE_MASK): jdmaster.c prepare_range_limit_table
__device__ __forceinline__ unsigned range_limit(int x) {
  const int v = x & 1023;
  return (unsigned)(v < 128 ? v + 128 : (v < 512 ? 255 : (v < 896 ? 0 : v - 896)));
}

// one thread per 8x8 block: grid (ceil(max blocks / 256), images).  The 256 blocks of a workgroup are 32 KB of CONTIGUOUS coefficients:
// they are fetched with fully coalesced 16-byte loads (lane l of an instruction takes chunk l of a 4 KiB run) into LDS, block b at byte
// 144 b -- the 16 bytes of padding spread a 16-lane group's 16-byte reads of one row over all 64 banks -- and each thread then reads its
// own block back.  (First form: every lane loaded straight from its own 128-byte line, 64 lines per instruction: 145 us per batch of 64.)
constexpr int BLK_LDS = 144;
__global__ __launch_bounds__(256) void jpeg_idct_kernel(const JpegPlan* __restrict__ plans, uint8_t* __restrict__ ws) {
  __shared__ __attribute__((aligned(16))) char stage[256 * BLK_LDS];
  const JpegPlan& P = plans[blockIdx.y];
  const int blk0 = blockIdx.x * 256;
  if (blk0 >= P.nblocks) return;
  const int nb = P.nblocks - blk0 < 256 ? P.nblocks - blk0 : 256;
  {
    const uint4* src = (const uint4*)(P.coefs + (size_t)blk0 * 64);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int chunk = it * 256 + threadIdx.x;         // 16-byte chunk of the workgroup's run: block chunk / 8, row chunk % 8
      if (chunk < nb * 8) *(uint4*)(stage + (chunk >> 3) * BLK_LDS + (chunk & 7) * 16) = src[chunk];
    }
  }
  __syncthreads();
  const int blk = blk0 + threadIdx.x;
  if (blk >= P.nblocks) return;
  const int c = (P.ncomp == 3 && blk >= P.block0[2]) ? 2 : ((P.ncomp == 3 && blk >= P.block0[1]) ? 1 : 0);
  const int local = blk - P.block0[c];
  const int brow = local / P.blocks_w[c], bcol = local - brow * P.blocks_w[c];
  const uint4* src = (const uint4*)(stage + threadIdx.x * BLK_LDS);
  int ws_[8][8];       // workspace after pass 1: [row][col]
  {
    int in[8][8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const uint4 v = src[r];
      const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        in[r][2 * k] = mul24((int)(short)(w4[k] & 0xffffu), (int)P.qt[c][r * 8 + 2 * k]);
        in[r][2 * k + 1] = mul24((int)(short)(w4[k] >> 16), (int)P.qt[c][r * 8 + 2 * k + 1]);
      }
    }
#pragma unroll
    for (int col = 0; col < 8; ++col) {
      int o[8];
      idct_1d<CONST_BITS - PASS1_BITS>(in[0][col], in[1][col], in[2][col], in[3][col], in[4][col], in[5][col], in[6][col], in[7][col], o);
#pragma unroll
      for (int r = 0; r < 8; ++r) ws_[r][col] = o[r];
    }
  }
  const int pitch = P.blocks_w[c] * 8;
  uint8_t* dst = ws + P.plane_off[c] + (size_t)(brow * 8) * pitch + bcol * 8;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    int o[8];
    idct_1d<CONST_BITS + PASS1_BITS + 3>(ws_[r][0], ws_[r][1], ws_[r][2], ws_[r][3], ws_[r][4], ws_[r][5], ws_[r][6], ws_[r][7], o);
    uint2 px;
    px.x = range_limit(o[0]) | (range_limit(o[1]) << 8) | (range_limit(o[2]) << 16) | (range_limit(o[3]) << 24);
    px.y = range_limit(o[4]) | (range_limit(o[5]) << 8) | (range_limit(o[6]) << 16) | (range_limit(o[7]) << 24);
    *(uint2*)(dst + (size_t)r * pitch) = px;
  }
}

__device__ __forceinline__ unsigned clamp255(int v) { return (unsigned)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// Chroma samples of the FOUR output pixels (r, col0 .. col0 + 3), col0 a multiple of 4, from the downsampled plane p (sw x sh real
// samples).  jdsample.c's first / last column special cases are the general formulas with the missing neighbour replaced by the sample
// itself -- (3 v + v + 1) >> 2 = v, (3 s + s + 8) >> 4 = (4 s + 8) >> 4 -- so neighbour indices are simply clamped; the rows above the
// first and below the last real row are copies of that row (jdmainct.c context rows): the row index is clamped as well.
template <int HS, int VS>
__device__ __forceinline__ void chroma4(const uint8_t* __restrict__ p, int pitch, int sw, int sh, int r, int col0, int (&out)[4]) {
  if (HS == 1) {
    const unsigned v = *(const unsigned*)(p + (size_t)r * pitch + col0);       // plane pitch is a multiple of 8, col0 of 4
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = (int)((v >> (8 * k)) & 0xffu);
    return;
  }
  const int cc0 = col0 >> 1;                      // the pixels use chroma columns cc0, cc0 + 1 and their outer neighbours
  int idx[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = cc0 - 1 + j;
    idx[j] = c < 0 ? 0 : (c > sw - 1 ? sw - 1 : c);
  }
  // The (clamped) columns lie inside the 8 bytes from the aligned position at or below cc0 - 1: two dword loads per row instead of four
  // byte loads (the kernel is bound by the COUNT of its loads: 16 byte loads per thread ran at 1.9 TB/s of algorithmic traffic).  The
  // second dword may reach up to 7 bytes past the row's samples: inside the plane's 8-byte-padded pitch or, for its last row, the
  // workspace's 256-byte alignment padding.
  const int base4 = cc0 >= 1 ? ((cc0 - 1) & ~3) : 0;
  auto row8 = [&](const uint8_t* row) -> unsigned long long {
    const unsigned lo = *(const unsigned*)(row + base4), hi = *(const unsigned*)(row + base4 + 4);
    return ((unsigned long long)hi << 32) | lo;
  };
  auto pick = [&](unsigned long long q, int c) -> int { return (int)((q >> (8 * (c - base4))) & 0xffull); };
  if (VS == 1) {          // jdsample.c h2v1_fancy_upsample
    const unsigned long long q = row8(p + (size_t)r * pitch);
    int v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = pick(q, idx[j]);
    // a column past the last real one (odd widths: col0 + 2, + 3 beyond the image) is never stored; its index was clamped
    out[0] = (3 * v[1] + v[0] + 1) >> 2;
    out[1] = (3 * v[1] + v[2] + 2) >> 2;
    out[2] = (3 * v[2] + v[1] + 1) >> 2;
    out[3] = (3 * v[2] + v[3] + 2) >> 2;
    return;
  }
  // h2v2_fancy_upsample: the nearest row weighs 3, the next nearest (above for even output rows, below for odd ones) 1
  const int cr = r >> 1;
  int other = (r & 1) ? cr + 1 : cr - 1;
  other = other < 0 ? 0 : (other > sh - 1 ? sh - 1 : other);
  const unsigned long long q0 = row8(p + (size_t)cr * pitch), q1 = row8(p + (size_t)other * pitch);
  int cs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) cs[j] = 3 * pick(q0, idx[j]) + pick(q1, idx[j]);
  out[0] = (cs[1] * 3 + cs[0] + 8) >> 4;
  out[1] = (cs[1] * 3 + cs[2] + 7) >> 4;
  out[2] = (cs[2] * 3 + cs[1] + 8) >> 4;
  out[3] = (cs[2] * 3 + cs[3] + 7) >> 4;
}

// one thread per FOUR horizontally consecutive output pixels: grid (ceil(max groups / 256), images).  The luma bytes arrive as one aligned
// dword, the chroma neighbourhood is loaded once for the four pixels; with a row pitch that is a multiple of 4 (what vitcap_amd.imageio
// allocates) the 12 output bytes leave as three dword stores, otherwise -- and for the last, partial group of a row -- byte by byte.
__global__ __launch_bounds__(256) void jpeg_color_kernel(const JpegPlan* __restrict__ plans, const uint8_t* __restrict__ ws) {
  const JpegPlan& P = plans[blockIdx.y];
  const int w4 = (P.width + 3) >> 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= w4 * P.height) return;
  const int r = idx / w4, col0 = (idx - r * w4) * 4;
  const int npx = P.width - col0 < 4 ? P.width - col0 : 4;
  const unsigned y4 = *(const unsigned*)(ws + P.plane_off[0] + (size_t)r * (P.blocks_w[0] * 8) + col0);      // plane pitch is a multiple of 8
  uint8_t* out = P.rgb + (size_t)r * P.pitch + (size_t)col0 * 3;
  unsigned char px[12];
  if (P.ncomp == 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) px[3 * k] = px[3 * k + 1] = px[3 * k + 2] = (unsigned char)((y4 >> (8 * k)) & 0xffu);
  } else {
    const uint8_t* pb = ws + P.plane_off[1];
    const uint8_t* pr = ws + P.plane_off[2];
    const int pitch = P.blocks_w[1] * 8, sw = P.samp_w[1], sh = P.samp_h[1];
    int cb[4], cr[4];
    if (P.hs0 == 1) {
      chroma4<1, 1>(pb, pitch, sw, sh, r, col0, cb);
      chroma4<1, 1>(pr, pitch, sw, sh, r, col0, cr);
    } else if (P.vs0 == 1) {
      chroma4<2, 1>(pb, pitch, sw, sh, r, col0, cb);
      chroma4<2, 1>(pr, pitch, sw, sh, r, col0, cr);
    } else {
      chroma4<2, 2>(pb, pitch, sw, sh, r, col0, cb);
      chroma4<2, 2>(pr, pitch, sw, sh, r, col0, cr);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // jdcolor.c build_ycc_rgb_table / ycc_rgb_convert: FIX(x) = (int)(x * 65536 + 0.5)
      const int y = (int)((y4 >> (8 * k)) & 0xffu), xb = cb[k] - 128, xr = cr[k] - 128;
      px[3 * k] = (unsigned char)clamp255(y + ((91881 * xr + 32768) >> 16));
      px[3 * k + 1] = (unsigned char)clamp255(y + ((-22554 * xb + 32768 - 46802 * xr) >> 16));
      px[3 * k + 2] = (unsigned char)clamp255(y + ((116130 * xb + 32768) >> 16));
    }
  }
  if (npx == 4 && (P.pitch & 3) == 0 && ((uintptr_t)P.rgb & 3) == 0) {
    unsigned* o32 = (unsigned*)out;
#pragma unroll
    for (int q = 0; q < 3; ++q)
      o32[q] = (unsigned)px[4 * q] | ((unsigned)px[4 * q + 1] << 8) | ((unsigned)px[4 * q + 2] << 16) | ((unsigned)px[4 * q + 3] << 24);
  } else {
    for (int i = 0; i < 3 * npx; ++i) out[i] = px[i];
  }
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

bool info_ok(const vitcap_jpeg_info& I) {
  if (I.abi != VITCAP_JPEG_ABI || I.width <= 0 || I.height <= 0 || (I.ncomp != 1 && I.ncomp != 3) || I.nblocks <= 0) return false;
  int nb = 0;
  for (int c = 0; c < I.ncomp; ++c) {
    if (I.blocks_w[c] <= 0 || I.blocks_h[c] <= 0 || I.block0[c] != nb || I.samp_w[c] <= 0 || I.samp_h[c] <= 0 ||
        I.samp_w[c] > 8 * I.blocks_w[c] || I.samp_h[c] > 8 * I.blocks_h[c])
      return false;
    nb += I.blocks_w[c] * I.blocks_h[c];
  }
  if (nb != I.nblocks || I.width > 8 * I.blocks_w[0] || I.height > 8 * I.blocks_h[0]) return false;
  if (I.ncomp == 3) {
    const bool s444 = I.hs[0] == 1 && I.vs[0] == 1, s422 = I.hs[0] == 2 && I.vs[0] == 1, s420 = I.hs[0] == 2 && I.vs[0] == 2;
    if (!(s444 || s422 || s420) || I.hs[1] != 1 || I.vs[1] != 1 || I.hs[2] != 1 || I.vs[2] != 1) return false;
    if (I.hs[0] == 2 && (I.samp_w[1] <= 2 || I.samp_w[2] != I.samp_w[1] || 2 * I.samp_w[1] < I.width)) return false;
    if (I.vs[0] == 2 && 2 * I.samp_h[1] < I.height) return false;
  }
  return true;
}

}  // namespace

extern "C" size_t vitcap_jpeg_backhalf_workspace_bytes(const vitcap_jpeg_image* imgs, int B) {
  if (!imgs || B <= 0) return 0;
  size_t tot = align256((size_t)B * sizeof(JpegPlan));
  for (int i = 0; i < B; ++i) {
    const vitcap_jpeg_info& I = imgs[i].info;
    for (int c = 0; c < (I.ncomp == 3 ? 3 : 1); ++c) tot += align256((size_t)I.blocks_w[c] * 8 * I.blocks_h[c] * 8);
  }
  return tot + 256;
}

extern "C" int vitcap_jpeg_backhalf(const vitcap_jpeg_image* imgs, int B, void* workspace, size_t workspace_bytes, void* stream) {
  VC_REQUIRE(imgs && B > 0 && workspace, "jpeg_backhalf: bad arguments");
  VC_REQUIRE(workspace_bytes >= vitcap_jpeg_backhalf_workspace_bytes(imgs, B), "jpeg_backhalf: workspace too small");
  std::vector<JpegPlan> plans(B);
  size_t off = align256((size_t)B * sizeof(JpegPlan));
  int max_blocks = 0;
  long long max_groups = 0;
  for (int i = 0; i < B; ++i) {
    const vitcap_jpeg_info& I = imgs[i].info;
    VC_REQUIRE(info_ok(I), "jpeg_backhalf: image %d carries an inconsistent vitcap_jpeg_info (not from vitcap_jpeg_parse?)", i);
    VC_REQUIRE(imgs[i].coefs && imgs[i].rgb && imgs[i].pitch >= 3 * I.width, "jpeg_backhalf: image %d has a bad descriptor", i);
    JpegPlan& p = plans[i];
    memset(&p, 0, sizeof(p));
    p.coefs = imgs[i].coefs;
    p.rgb = imgs[i].rgb;
    p.pitch = imgs[i].pitch;
    p.width = I.width; p.height = I.height; p.ncomp = I.ncomp; p.hs0 = I.hs[0]; p.vs0 = I.vs[0]; p.nblocks = I.nblocks;
    for (int c = 0; c < I.ncomp; ++c) {
      p.blocks_w[c] = I.blocks_w[c]; p.blocks_h[c] = I.blocks_h[c]; p.samp_w[c] = I.samp_w[c]; p.samp_h[c] = I.samp_h[c];
      p.block0[c] = I.block0[c];
      p.plane_off[c] = (long long)off;
      off += align256((size_t)I.blocks_w[c] * 8 * I.blocks_h[c] * 8);
      memcpy(p.qt[c], I.qt[c], sizeof(p.qt[c]));
    }
    max_blocks = I.nblocks > max_blocks ? I.nblocks : max_blocks;
    const long long groups = (long long)((I.width + 3) / 4) * I.height;
    max_groups = groups > max_groups ? groups : max_groups;
  }
  VC_REQUIRE(off <= workspace_bytes, "jpeg_backhalf: workspace too small (%zu > %zu)", off, workspace_bytes);
  // pinned staging for the plan table (reused; an event guards reuse by the next call on this thread)
  static thread_local char* stage = nullptr;
  static thread_local size_t stage_cap = 0;
  static thread_local hipEvent_t stage_ev = nullptr;
  const size_t up = (size_t)B * sizeof(JpegPlan);
  if (stage_ev) (void)hipEventSynchronize(stage_ev);
  if (up > stage_cap) {
    if (stage) (void)hipHostFree(stage);
    VC_REQUIRE(hipHostMalloc((void**)&stage, up, hipHostMallocDefault) == hipSuccess, "jpeg_backhalf: pinned staging alloc failed");
    stage_cap = up;
  }
  if (!stage_ev) VC_REQUIRE(hipEventCreateWithFlags(&stage_ev, hipEventDisableTiming) == hipSuccess, "jpeg_backhalf: event");
  memcpy(stage, plans.data(), up);
  hipStream_t s = (hipStream_t)stream;
  VC_REQUIRE(hipMemcpyAsync(workspace, stage, up, hipMemcpyHostToDevice, s) == hipSuccess, "jpeg_backhalf: plan upload failed");
  (void)hipEventRecord(stage_ev, s);
  const JpegPlan* dplans = (const JpegPlan*)workspace;
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)(max_blocks + 255) / 256, B), dim3(256), 0, s, dplans, (uint8_t*)workspace);
  VC_LAUNCH_CHECK("jpeg_idct");
  hipLaunchKernelGGL(jpeg_color_kernel, dim3((unsigned)((max_groups + 255) / 256), B), dim3(256), 0, s, dplans, (const uint8_t*)workspace);
  VC_LAUNCH_CHECK("jpeg_color");
  return VITCAP_OK;
}
