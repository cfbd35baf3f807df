// Test-time image transform of the captioning pipeline on the device (uni_pipeline.py:1233-1265):
//   decoded RGB uint8 HWC image  ->  Resize(shorter side = 384 / crop_pct, PIL BICUBIC)  ->  CenterCrop(384)
//   ->  ToTensor (/255)  ->  Normalize(mean .5, std .5)  ->  [3][384][384] fp32 or bf16.
// torchvision's Resize on a PIL image is PIL.Image.resize, i.e. Pillow's two-pass separable convolution on 8-bit
// pixels (libImaging/Resample.c): double-precision bicubic weights (a = -0.5, support 2 x scale when downscaling),
// normalised, converted to 22-bit fixed point, a horizontal pass that ROUNDS TO uint8, then a vertical pass.
// That arithmetic is integer and is reproduced bit for bit: the weights are computed on the host exactly as Pillow
// does (same operation order, doubles), the passes run on the GPU in int32.
// Only what the centre crop needs is computed: crop columns in the horizontal pass, and of those only the source rows
// the crop rows' vertical taps touch.  HBM-bound byte work: ~1 MB in, 0.9 MB (fp32) out per 640x480 image.
#include <math.h>
#include <string.h>
#include <vector>

#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;   // Pillow: 8-bit pixels, 2 guard bits

double bicubic_filter(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

double bilinear_filter(double x) {
  if (x < 0.0) x = -x;
  if (x < 1.0) return 1.0 - x;
  return 0.0;
}

enum { FILTER_BICUBIC = 0, FILTER_BILINEAR = 1 };

// Pillow precompute_coeffs + normalize_coeffs_8bpc for the full-image box (in0 = 0, in1 = in_size)
int resample_coeffs(int in_size, int out_size, std::vector<int>& bounds, std::vector<int>& kk, int filter = FILTER_BICUBIC) {
  const double scale = (double)in_size / out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = (filter == FILTER_BILINEAR ? 1.0 : 2.0) * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  bounds.assign((size_t)out_size * 2, 0);
  kk.assign((size_t)out_size * ksize, 0);
  std::vector<double> pre(ksize);
  for (int xx = 0; xx < out_size; xx++) {
    const double center = (xx + 0.5) * scale;
    double ww = 0.0;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    int x;
    for (x = 0; x < xmax; x++) {
      const double arg = (x + xmin - center + 0.5) * ss;
      const double w = filter == FILTER_BILINEAR ? bilinear_filter(arg) : bicubic_filter(arg);
      pre[x] = w;
      ww += w;
    }
    for (x = 0; x < xmax; x++)
      if (ww != 0.0) pre[x] /= ww;
    for (; x < ksize; x++) pre[x] = 0;
    int* k = &kk[(size_t)xx * ksize];
    for (x = 0; x < ksize; x++) {
      if (pre[x] < 0) k[x] = (int)(-0.5 + pre[x] * (1 << PRECISION_BITS));
      else k[x] = (int)(0.5 + pre[x] * (1 << PRECISION_BITS));
    }
    bounds[xx * 2] = xmin;
    bounds[xx * 2 + 1] = xmax;
  }
  return ksize;
}

struct ImgPlan {          // one per image, lives in the device workspace
  const uint8_t* src;
  int height, width, pitch;
  int crop_x0, crop_y0;   // crop window inside the resized image
  int ksize_h, ksize_v;
  int row_first, rows;    // source rows the vertical taps of the crop rows touch
  long long hb_off, hk_off, vb_off, vk_off;   // int offsets of the bounds / weights tables (crop window only)
  long long tmp_off;      // byte offset of the horizontal-pass image [rows][crop][3]
  long long img_off;      // train path: byte offset of the resampled uint8 image [crop][crop][3]
};

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: tmp[r][x][c] for the needed rows and the crop columns
__global__ __launch_bounds__(256) void resample_h_kernel(const ImgPlan* __restrict__ plans, const int* __restrict__ tab,
                                                         uint8_t* __restrict__ ws, int crop) {
  const ImgPlan pl = plans[blockIdx.y];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int r = idx / crop, x = idx - r * crop;
  if (r >= pl.rows) return;
  const int xmin = tab[pl.hb_off + x * 2], xmax = tab[pl.hb_off + x * 2 + 1];
  const int* k = tab + pl.hk_off + (long long)x * pl.ksize_h;
  const uint8_t* s = pl.src + (size_t)(pl.row_first + r) * pl.pitch + (size_t)xmin * 3;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int i = 0; i < xmax; ++i) {
    const int w = k[i];
    s0 += s[i * 3 + 0] * w;
    s1 += s[i * 3 + 1] * w;
    s2 += s[i * 3 + 2] * w;
  }
  uint8_t* t = ws + pl.tmp_off + ((size_t)r * crop + x) * 3;
  t[0] = (uint8_t)clip8(s0);
  t[1] = (uint8_t)clip8(s1);
  t[2] = (uint8_t)clip8(s2);
}

// vertical pass on the crop rows + ToTensor + Normalize + CHW layout
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void resample_v_kernel(const ImgPlan* __restrict__ plans, const int* __restrict__ tab,
                                                         const uint8_t* __restrict__ ws, int crop, void* __restrict__ out,
                                                         uint8_t* __restrict__ out_u8) {
  const ImgPlan pl = plans[blockIdx.y];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int y = idx / crop, x = idx - y * crop;
  if (y >= crop) return;
  const int ymin = tab[pl.vb_off + y * 2], ymax = tab[pl.vb_off + y * 2 + 1];
  const int* k = tab + pl.vk_off + (long long)y * pl.ksize_v;
  const uint8_t* t = ws + pl.tmp_off + ((size_t)(ymin - pl.row_first) * crop + x) * 3;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int i = 0; i < ymax; ++i) {
    const int w = k[i];
    const uint8_t* p = t + (size_t)i * crop * 3;
    s0 += p[0] * w;
    s1 += p[1] * w;
    s2 += p[2] * w;
  }
  const int v[3] = {clip8(s0), clip8(s1), clip8(s2)};
  const size_t plane = (size_t)crop * crop;
  const size_t o = (size_t)blockIdx.y * 3 * plane + (size_t)y * crop + x;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    // torchvision ToTensor: byte -> float32 / 255 ; Normalize: (t - 0.5) / 0.5, each a correctly rounded fp32 operation
    const float tt = (float)v[c] / 255.0f;
    const float nv = (tt - 0.5f) / 0.5f;
    if (OUT_BF16) ((bf16_t*)out)[o + c * plane] = f2bf(nv);
    else ((float*)out)[o + c * plane] = nv;
    if (out_u8) out_u8[o + c * plane] = (uint8_t)v[c];
  }
}

struct HostPlan {
  ImgPlan p;
  std::vector<int> hb, hk, vb, vk;
};

// torchvision.transforms.functional.resize (v0.8-0.15, _compute_resized_output_size) with an int size: the shorter side
// becomes `size`, the longer int(size * long / short); CenterCrop: top = int(round((h - crop) / 2.0)) with Python's
// round-half-to-even.
void resized_dims(int h, int w, int size, int* oh, int* ow) {
  if (w <= h) { *ow = size; *oh = (int)((long long)size * h / w); }
  else { *oh = size; *ow = (int)((long long)size * w / h); }
}
int py_round_half(int diff) {   // round(diff / 2.0), half to even
  if (diff % 2 == 0) return diff / 2;
  const int fl = (diff - 1) / 2;            // diff odd and >= 0: x.5
  return (fl % 2 == 0) ? fl : fl + 1;
}

int build_plan(const vitcap_image& im, int resize_short, int crop, HostPlan& hp) {
  int oh, ow;
  resized_dims(im.height, im.width, resize_short, &oh, &ow);
  if (oh < crop || ow < crop) return -1;
  const int cx0 = py_round_half(ow - crop), cy0 = py_round_half(oh - crop);
  std::vector<int> b, k;
  const int ks_h = resample_coeffs(im.width, ow, b, k);
  hp.hb.assign(b.begin() + (size_t)cx0 * 2, b.begin() + (size_t)(cx0 + crop) * 2);
  hp.hk.assign(k.begin() + (size_t)cx0 * ks_h, k.begin() + (size_t)(cx0 + crop) * ks_h);
  const int ks_v = resample_coeffs(im.height, oh, b, k);
  hp.vb.assign(b.begin() + (size_t)cy0 * 2, b.begin() + (size_t)(cy0 + crop) * 2);
  hp.vk.assign(k.begin() + (size_t)cy0 * ks_v, k.begin() + (size_t)(cy0 + crop) * ks_v);
  int first = im.height, last = 0;
  for (int y = 0; y < crop; ++y) {
    const int lo = hp.vb[y * 2], hi = lo + hp.vb[y * 2 + 1];
    first = lo < first ? lo : first;
    last = hi > last ? hi : last;
  }
  hp.p.src = im.rgb;
  hp.p.height = im.height; hp.p.width = im.width; hp.p.pitch = im.pitch;
  hp.p.crop_x0 = cx0; hp.p.crop_y0 = cy0;
  hp.p.ksize_h = ks_h; hp.p.ksize_v = ks_v;
  hp.p.row_first = first; hp.p.rows = last - first;
  return 0;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// ---------------------------------------------------------------------------------------------------------------
// Train-time transform (get_inception_train_transform, src/data_layer/transform.py:52-81): the random parameters come
// from the host (vitcap_amd/augment.py); the image arithmetic is Pillow's and is reproduced bit for bit:
//   crop box + Image.resize(BILINEAR)            -> the two resample passes above with the triangle filter
//   ImageEnhance.Brightness/Contrast/Color       -> Image.blend(degenerate, image, factor), libImaging/Blend.c: float32
//                                                   in1 + alpha * (in2 - in1), truncated to uint8, clipped when alpha > 1
//   degenerate images                            -> black / the rounded mean of the L image / the L image; L by
//                                                   Convert.c rgb2l (19595 R + 38470 G + 7471 B + 0x8000) >> 16
//   transpose(FLIP_LEFT_RIGHT), ToTensor, Normalize
struct AugPlan {
  int op[3];          // application order; 0 brightness, 1 contrast, 2 saturation, -1 none
  float factor[3];
  int flip;
};

// vertical pass of the train path: uint8 HWC image into the workspace (the colour operations need whole-image statistics)
__global__ __launch_bounds__(256) void resample_v_u8_kernel(const ImgPlan* __restrict__ plans, const int* __restrict__ tab,
                                                            uint8_t* __restrict__ ws, int crop) {
  const ImgPlan pl = plans[blockIdx.y];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int y = idx / crop, x = idx - y * crop;
  if (y >= crop) return;
  const int ymin = tab[pl.vb_off + y * 2], ymax = tab[pl.vb_off + y * 2 + 1];
  const int* k = tab + pl.vk_off + (long long)y * pl.ksize_v;
  const uint8_t* t = ws + pl.tmp_off + ((size_t)(ymin - pl.row_first) * crop + x) * 3;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int i = 0; i < ymax; ++i) {
    const int w = k[i];
    const uint8_t* p = t + (size_t)i * crop * 3;
    s0 += p[0] * w;
    s1 += p[1] * w;
    s2 += p[2] * w;
  }
  uint8_t* o = ws + pl.img_off + ((size_t)y * crop + x) * 3;
  o[0] = (uint8_t)clip8(s0);
  o[1] = (uint8_t)clip8(s1);
  o[2] = (uint8_t)clip8(s2);
}

__device__ __forceinline__ int luma601(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// Blend.c: (UINT8)((int)in1 + alpha * ((int)in2 - (int)in1)) in float32, the product and the sum rounded separately
// (no fused multiply-add: the reference runs on a CPU build without contraction)
__device__ __forceinline__ int blend8(int in1, int in2, float alpha, bool extrapolate) {
#pragma clang fp contract(off)     // __fmul_rn + __fadd_rn still fused into one v_fma after inlining (seen: 121 bytes off by one at factor 0.6)
  const float prod = alpha * (float)(in2 - in1);
  const float t = (float)in1 + prod;
  if (extrapolate) {
    if (t <= 0.0f) return 0;
    if (t >= 255.0f) return 255;
  }
  return (int)t;
}

// one workgroup per image: the three colour operations in the drawn order, then flip + ToTensor + Normalize + CHW
template <bool OUT_BF16>
__global__ __launch_bounds__(1024) void color_jitter_kernel(const ImgPlan* __restrict__ plans, const AugPlan* __restrict__ augs,
                                                            uint8_t* __restrict__ ws, int crop, void* __restrict__ out,
                                                            uint8_t* __restrict__ out_u8) {
  const ImgPlan pl = plans[blockIdx.x];
  const AugPlan ap = augs[blockIdx.x];
  uint8_t* img = ws + pl.img_off;
  const int npix = crop * crop;
  const int tid = threadIdx.x;
  __shared__ long long s_part[16];
  __shared__ int s_mean;
  for (int o = 0; o < 3; ++o) {
    const int op = ap.op[o];
    if (op < 0) continue;
    const float a = ap.factor[o];
    const bool ext = !(a >= 0.0f && a <= 1.0f);
    int mean = 0;
    if (op == 1) {
      // ImageStat.Stat(image.convert("L")).mean[0]: sum / count in doubles, then int(mean + 0.5)
      long long part = 0;
      for (int i = tid; i < npix; i += 1024) part += luma601(img[i * 3], img[i * 3 + 1], img[i * 3 + 2]);
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
      if ((tid & 63) == 0) s_part[tid >> 6] = part;
      __syncthreads();
      if (tid == 0) {
        long long tot = 0;
        for (int q = 0; q < 16; ++q) tot += s_part[q];
        s_mean = (int)((double)tot / (double)npix + 0.5);
      }
      __syncthreads();
      mean = s_mean;
    }
    for (int i = tid; i < npix; i += 1024) {
      const int r = img[i * 3], g = img[i * 3 + 1], b = img[i * 3 + 2];
      int d0 = 0, d1 = 0, d2 = 0;                     // the degenerate image's pixel
      if (op == 1) d0 = d1 = d2 = mean;
      else if (op == 2) d0 = d1 = d2 = luma601(r, g, b);
      img[i * 3] = (uint8_t)blend8(d0, r, a, ext);
      img[i * 3 + 1] = (uint8_t)blend8(d1, g, a, ext);
      img[i * 3 + 2] = (uint8_t)blend8(d2, b, a, ext);
    }
    __syncthreads();       // each thread re-reads only its own pixels, but the contrast mean reads everyone's
  }
  const size_t plane = (size_t)npix;
  for (int i = tid; i < npix; i += 1024) {
    const int y = i / crop, x = i - y * crop;
    const int xs = ap.flip ? crop - 1 - x : x;
    const uint8_t* p = img + ((size_t)y * crop + xs) * 3;
    const size_t o = (size_t)blockIdx.x * 3 * plane + (size_t)i;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float tt = (float)p[c] / 255.0f;
      const float nv = (tt - 0.5f) / 0.5f;
      if (OUT_BF16) ((bf16_t*)out)[o + c * plane] = f2bf(nv);
      else ((float*)out)[o + c * plane] = nv;
      if (out_u8) out_u8[o + c * plane] = p[c];
    }
  }
}

int build_train_plan(const vitcap_image& im, const vitcap_train_aug& a, int size, HostPlan& hp) {
  if (a.top < 0 || a.left < 0 || a.height <= 0 || a.width <= 0 || a.top + a.height > im.height || a.left + a.width > im.width)
    return -1;
  const int ks_h = resample_coeffs(a.width, size, hp.hb, hp.hk, FILTER_BILINEAR);
  const int ks_v = resample_coeffs(a.height, size, hp.vb, hp.vk, FILTER_BILINEAR);
  int first = a.height, last = 0;
  for (int y = 0; y < size; ++y) {
    const int lo = hp.vb[y * 2], hi = lo + hp.vb[y * 2 + 1];
    first = lo < first ? lo : first;
    last = hi > last ? hi : last;
  }
  hp.p.src = im.rgb + (size_t)a.top * im.pitch + (size_t)a.left * 3;   // Image.crop, then resize of the cropped image
  hp.p.height = a.height; hp.p.width = a.width; hp.p.pitch = im.pitch;
  hp.p.crop_x0 = 0; hp.p.crop_y0 = 0;
  hp.p.ksize_h = ks_h; hp.p.ksize_v = ks_v;
  hp.p.row_first = first; hp.p.rows = last - first;
  return 0;
}

}  // namespace

extern "C" int vitcap_resample_coeffs(int in_size, int out_size, int* ksize_out, int* bounds, int* kk, int kk_capacity) {
  VC_REQUIRE(in_size > 0 && out_size > 0 && ksize_out && bounds && kk, "resample_coeffs: bad arguments");
  std::vector<int> b, k;
  const int ks = resample_coeffs(in_size, out_size, b, k);
  VC_REQUIRE((long long)out_size * ks <= kk_capacity, "resample_coeffs: kk_capacity %d < %lld", kk_capacity, (long long)out_size * ks);
  *ksize_out = ks;
  memcpy(bounds, b.data(), b.size() * sizeof(int));
  memcpy(kk, k.data(), k.size() * sizeof(int));
  return VITCAP_OK;
}

extern "C" int vitcap_resized_geometry(int height, int width, int resize_short, int crop, int* out_h, int* out_w, int* crop_y0,
                                       int* crop_x0) {
  VC_REQUIRE(height > 0 && width > 0 && resize_short > 0 && crop > 0 && out_h && out_w && crop_y0 && crop_x0,
             "resized_geometry: bad arguments");
  resized_dims(height, width, resize_short, out_h, out_w);
  VC_REQUIRE(*out_h >= crop && *out_w >= crop, "resized_geometry: %dx%d resized to %dx%d is smaller than the %d crop", height,
             width, *out_h, *out_w, crop);
  *crop_y0 = py_round_half(*out_h - crop);
  *crop_x0 = py_round_half(*out_w - crop);
  return VITCAP_OK;
}

extern "C" size_t vitcap_image_preproc_workspace_bytes(const vitcap_image* imgs, int B, int resize_short, int crop) {
  if (!imgs || B <= 0) return 0;
  size_t tot = align256((size_t)B * sizeof(ImgPlan));
  size_t tab = 0, tmp = 0;
  for (int i = 0; i < B; ++i) {
    int oh, ow;
    resized_dims(imgs[i].height, imgs[i].width, resize_short, &oh, &ow);
    const double sh = (double)imgs[i].width / ow, sv = (double)imgs[i].height / oh;
    const int ks_h = (int)ceil(2.0 * (sh < 1 ? 1 : sh)) * 2 + 1, ks_v = (int)ceil(2.0 * (sv < 1 ? 1 : sv)) * 2 + 1;
    tab += (size_t)crop * (4 + ks_h + ks_v) * sizeof(int);
    tmp += align256((size_t)imgs[i].height * crop * 3);          // upper bound: every source row needed
  }
  return tot + align256(tab) + tmp + 256;
}

extern "C" int vitcap_image_preproc(const vitcap_image* imgs, int B, int resize_short, int crop, int out_bf16, void* out,
                                    uint8_t* out_u8, void* workspace, size_t workspace_bytes, void* stream) {
  VC_REQUIRE(imgs && B > 0 && resize_short > 0 && crop > 0 && out && workspace, "image_preproc: bad arguments");
  VC_REQUIRE(workspace_bytes >= vitcap_image_preproc_workspace_bytes(imgs, B, resize_short, crop),
             "image_preproc: workspace too small");
  std::vector<HostPlan> plans(B);
  size_t tab_ints = 0;
  int max_rows = 0;
  for (int i = 0; i < B; ++i) {
    VC_REQUIRE(imgs[i].rgb && imgs[i].height > 0 && imgs[i].width > 0 && imgs[i].pitch >= imgs[i].width * 3,
               "image_preproc: image %d has a bad descriptor", i);
    VC_REQUIRE(build_plan(imgs[i], resize_short, crop, plans[i]) == 0,
               "image_preproc: image %d (%dx%d) is smaller than the crop after resizing", i, imgs[i].height, imgs[i].width);
    HostPlan& hp = plans[i];
    hp.p.hb_off = (long long)tab_ints; tab_ints += hp.hb.size();
    hp.p.hk_off = (long long)tab_ints; tab_ints += hp.hk.size();
    hp.p.vb_off = (long long)tab_ints; tab_ints += hp.vb.size();
    hp.p.vk_off = (long long)tab_ints; tab_ints += hp.vk.size();
    max_rows = hp.p.rows > max_rows ? hp.p.rows : max_rows;
  }
  const size_t plan_bytes = align256((size_t)B * sizeof(ImgPlan));
  const size_t tab_bytes = align256(tab_ints * sizeof(int));
  size_t tmp_off = plan_bytes + tab_bytes;
  for (int i = 0; i < B; ++i) {
    plans[i].p.tmp_off = (long long)tmp_off;
    tmp_off += align256((size_t)plans[i].p.rows * crop * 3);
  }
  VC_REQUIRE(tmp_off <= workspace_bytes, "image_preproc: workspace too small (%zu > %zu)", tmp_off, workspace_bytes);

  // host staging (pinned, reused): plans + tables in one upload; an event guards reuse by the next call
  static thread_local char* stage = nullptr;
  static thread_local size_t stage_cap = 0;
  static thread_local hipEvent_t stage_ev = nullptr;
  const size_t up = plan_bytes + tab_bytes;
  if (stage_ev) (void)hipEventSynchronize(stage_ev);
  if (up > stage_cap) {
    if (stage) (void)hipHostFree(stage);
    VC_REQUIRE(hipHostMalloc((void**)&stage, up, hipHostMallocDefault) == hipSuccess, "image_preproc: pinned staging alloc failed");
    stage_cap = up;
  }
  if (!stage_ev) VC_REQUIRE(hipEventCreateWithFlags(&stage_ev, hipEventDisableTiming) == hipSuccess, "image_preproc: event");
  for (int i = 0; i < B; ++i) memcpy(stage + (size_t)i * sizeof(ImgPlan), &plans[i].p, sizeof(ImgPlan));
  int* tab = (int*)(stage + plan_bytes);
  for (int i = 0; i < B; ++i) {
    const HostPlan& hp = plans[i];
    memcpy(tab + hp.p.hb_off, hp.hb.data(), hp.hb.size() * sizeof(int));
    memcpy(tab + hp.p.hk_off, hp.hk.data(), hp.hk.size() * sizeof(int));
    memcpy(tab + hp.p.vb_off, hp.vb.data(), hp.vb.size() * sizeof(int));
    memcpy(tab + hp.p.vk_off, hp.vk.data(), hp.vk.size() * sizeof(int));
  }
  hipStream_t s = (hipStream_t)stream;
  VC_REQUIRE(hipMemcpyAsync(workspace, stage, up, hipMemcpyHostToDevice, s) == hipSuccess, "image_preproc: table upload failed");
  (void)hipEventRecord(stage_ev, s);
  const ImgPlan* dplans = (const ImgPlan*)workspace;
  const int* dtab = (const int*)((char*)workspace + plan_bytes);
  dim3 gh(((unsigned)max_rows * crop + 255) / 256, B);
  hipLaunchKernelGGL(resample_h_kernel, gh, dim3(256), 0, s, dplans, dtab, (uint8_t*)workspace, crop);
  VC_LAUNCH_CHECK("resample_h");
  dim3 gv(((unsigned)crop * crop + 255) / 256, B);
  if (out_bf16)
    hipLaunchKernelGGL(resample_v_kernel<true>, gv, dim3(256), 0, s, dplans, dtab, (const uint8_t*)workspace, crop, out, out_u8);
  else
    hipLaunchKernelGGL(resample_v_kernel<false>, gv, dim3(256), 0, s, dplans, dtab, (const uint8_t*)workspace, crop, out, out_u8);
  VC_LAUNCH_CHECK("resample_v");
  return VITCAP_OK;
}

extern "C" size_t vitcap_image_train_preproc_workspace_bytes(const vitcap_image* imgs, const vitcap_train_aug* aug, int B, int size) {
  if (!imgs || !aug || B <= 0 || size <= 0) return 0;
  size_t tot = align256((size_t)B * sizeof(ImgPlan)) + align256((size_t)B * sizeof(AugPlan));
  size_t tab = 0, tmp = 0;
  for (int i = 0; i < B; ++i) {
    const int w = aug[i].width > 0 ? aug[i].width : 1, h = aug[i].height > 0 ? aug[i].height : 1;
    const double sh = (double)w / size, sv = (double)h / size;
    const int ks_h = (int)ceil(sh < 1 ? 1 : sh) * 2 + 1, ks_v = (int)ceil(sv < 1 ? 1 : sv) * 2 + 1;
    tab += (size_t)size * (4 + ks_h + ks_v) * sizeof(int);
    tmp += align256((size_t)h * size * 3) + align256((size_t)size * size * 3);
  }
  return tot + align256(tab) + tmp + 256;
}

extern "C" int vitcap_image_train_preproc(const vitcap_image* imgs, const vitcap_train_aug* aug, int B, int size, int out_bf16,
                                          void* out, uint8_t* out_u8, void* workspace, size_t workspace_bytes, void* stream) {
  VC_REQUIRE(imgs && aug && B > 0 && size > 0 && out && workspace, "image_train_preproc: bad arguments");
  VC_REQUIRE(workspace_bytes >= vitcap_image_train_preproc_workspace_bytes(imgs, aug, B, size),
             "image_train_preproc: workspace too small");
  std::vector<HostPlan> plans(B);
  std::vector<AugPlan> augs(B);
  size_t tab_ints = 0;
  int max_rows = 0;
  for (int i = 0; i < B; ++i) {
    VC_REQUIRE(imgs[i].rgb && imgs[i].height > 0 && imgs[i].width > 0 && imgs[i].pitch >= imgs[i].width * 3,
               "image_train_preproc: image %d has a bad descriptor", i);
    VC_REQUIRE(build_train_plan(imgs[i], aug[i], size, plans[i]) == 0,
               "image_train_preproc: crop box (%d,%d,%d,%d) of image %d lies outside its %dx%d pixels", aug[i].top, aug[i].left,
               aug[i].height, aug[i].width, i, imgs[i].height, imgs[i].width);
    HostPlan& hp = plans[i];
    hp.p.hb_off = (long long)tab_ints; tab_ints += hp.hb.size();
    hp.p.hk_off = (long long)tab_ints; tab_ints += hp.hk.size();
    hp.p.vb_off = (long long)tab_ints; tab_ints += hp.vb.size();
    hp.p.vk_off = (long long)tab_ints; tab_ints += hp.vk.size();
    max_rows = hp.p.rows > max_rows ? hp.p.rows : max_rows;
    int seen = 0;
    for (int o = 0; o < 3; ++o) {
      const int op = aug[i].op[o];
      VC_REQUIRE(op >= -1 && op <= 2, "image_train_preproc: image %d: colour operation %d is not 0/1/2 (or -1 = none)", i, op);
      if (op >= 0) {
        VC_REQUIRE(!(seen >> op & 1), "image_train_preproc: image %d applies colour operation %d twice", i, op);
        seen |= 1 << op;
        VC_REQUIRE(aug[i].factor[o] >= 0.f, "image_train_preproc: image %d: negative enhancement factor", i);
      }
      augs[i].op[o] = op;
      augs[i].factor[o] = aug[i].factor[o];
    }
    augs[i].flip = aug[i].flip ? 1 : 0;
  }
  const size_t plan_bytes = align256((size_t)B * sizeof(ImgPlan));
  const size_t aug_bytes = align256((size_t)B * sizeof(AugPlan));
  const size_t tab_bytes = align256(tab_ints * sizeof(int));
  size_t off = plan_bytes + aug_bytes + tab_bytes;
  for (int i = 0; i < B; ++i) {
    plans[i].p.tmp_off = (long long)off;
    off += align256((size_t)plans[i].p.rows * size * 3);
    plans[i].p.img_off = (long long)off;
    off += align256((size_t)size * size * 3);
  }
  VC_REQUIRE(off <= workspace_bytes, "image_train_preproc: workspace too small (%zu > %zu)", off, workspace_bytes);

  static thread_local char* stage = nullptr;
  static thread_local size_t stage_cap = 0;
  static thread_local hipEvent_t stage_ev = nullptr;
  const size_t up = plan_bytes + aug_bytes + tab_bytes;
  if (stage_ev) (void)hipEventSynchronize(stage_ev);
  if (up > stage_cap) {
    if (stage) (void)hipHostFree(stage);
    VC_REQUIRE(hipHostMalloc((void**)&stage, up, hipHostMallocDefault) == hipSuccess, "image_train_preproc: pinned staging alloc failed");
    stage_cap = up;
  }
  if (!stage_ev) VC_REQUIRE(hipEventCreateWithFlags(&stage_ev, hipEventDisableTiming) == hipSuccess, "image_train_preproc: event");
  for (int i = 0; i < B; ++i) {
    memcpy(stage + (size_t)i * sizeof(ImgPlan), &plans[i].p, sizeof(ImgPlan));
    memcpy(stage + plan_bytes + (size_t)i * sizeof(AugPlan), &augs[i], sizeof(AugPlan));
  }
  int* tab = (int*)(stage + plan_bytes + aug_bytes);
  for (int i = 0; i < B; ++i) {
    const HostPlan& hp = plans[i];
    memcpy(tab + hp.p.hb_off, hp.hb.data(), hp.hb.size() * sizeof(int));
    memcpy(tab + hp.p.hk_off, hp.hk.data(), hp.hk.size() * sizeof(int));
    memcpy(tab + hp.p.vb_off, hp.vb.data(), hp.vb.size() * sizeof(int));
    memcpy(tab + hp.p.vk_off, hp.vk.data(), hp.vk.size() * sizeof(int));
  }
  hipStream_t s = (hipStream_t)stream;
  VC_REQUIRE(hipMemcpyAsync(workspace, stage, up, hipMemcpyHostToDevice, s) == hipSuccess, "image_train_preproc: table upload failed");
  (void)hipEventRecord(stage_ev, s);
  const ImgPlan* dplans = (const ImgPlan*)workspace;
  const AugPlan* daugs = (const AugPlan*)((char*)workspace + plan_bytes);
  const int* dtab = (const int*)((char*)workspace + plan_bytes + aug_bytes);
  dim3 gh(((unsigned)max_rows * size + 255) / 256, B);
  hipLaunchKernelGGL(resample_h_kernel, gh, dim3(256), 0, s, dplans, dtab, (uint8_t*)workspace, size);
  VC_LAUNCH_CHECK("resample_h(train)");
  dim3 gv(((unsigned)size * size + 255) / 256, B);
  hipLaunchKernelGGL(resample_v_u8_kernel, gv, dim3(256), 0, s, dplans, dtab, (uint8_t*)workspace, size);
  VC_LAUNCH_CHECK("resample_v_u8");
  if (out_bf16)
    hipLaunchKernelGGL(color_jitter_kernel<true>, dim3(B), dim3(1024), 0, s, dplans, daugs, (uint8_t*)workspace, size, out, out_u8);
  else
    hipLaunchKernelGGL(color_jitter_kernel<false>, dim3(B), dim3(1024), 0, s, dplans, daugs, (uint8_t*)workspace, size, out, out_u8);
  VC_LAUNCH_CHECK("color_jitter");
  return VITCAP_OK;
}
