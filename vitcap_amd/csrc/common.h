// Shared device/host helpers for libvitcap_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>

#include "../../include/vitcap_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned short bf16_t;  // storage type

#define WAVE 64

void vitcap_set_error(const char* fmt, ...);

// Engine-internal: while vitcap_engine_decode enqueues its step loop, this thread-local pointer names the device
// counter of sequences (beam search: images) still unfinished; the decode-step launchers pass it to their kernels,
// which return at entry once it reads 0 -- the reference's `if cur_unfinished.max() == 0: break`
// (modeling_utils.py:866 / `if all(done): break`, :1072) without a host synchronisation.  NULL outside the engine.
extern thread_local const int32_t* vc_tls_live;
// Engine-internal: further EOS token ids of the call being enqueued (vitcap_gen_opts.eos_extra; -1 = unused).  The reference's
// greedy / sampling loop stops a sequence at ANY id of `eos_token_ids` (modeling_utils.py:862-865) and forces eos_token_ids[0]
// at the last position (:870-871); the step launchers read this next to their `eos` argument, which stays the first id.
// Engine-internal: walk direction of the streaming kernels of the call being enqueued (round 5).  Every kernel of the encoder /
// prefill chain is one pass over the batch's rows; a consumer that walks them in the SAME order as its producer finds, in the 256 MB
// Infinity Cache, the END of what the producer wrote while it asks for the BEGINNING (fc2's A operand at B = 64 is 227 MB: the probe
// with that operand cache-resident runs the kernel 21 % faster, profiles/r05_g4w_probe.txt).  With the flag set a kernel visits its
// row blocks last-to-first, so that what was written last is read first; the engine flips it after every streaming launch.
// Results do not depend on it (the order in which independent tiles run).  False outside the engine.
// Dropout salt (vitcap_set_dropout_salt): a device-resident 32-bit word XORed into every dropout seed by the training kernels.  The
// seeds themselves are launch arguments -- frozen when a training step is captured into a hipGraph -- so a captured step changes its
// keep decisions from replay to replay by rewriting this word (vitcap_amd/train.py, graph mode).  NULL = no salt (the default).
extern thread_local const uint32_t* vc_tls_drop_salt;
__device__ __forceinline__ uint32_t vc_salted(uint32_t seed, const uint32_t* salt) { return salt ? seed ^ *salt : seed; }
extern thread_local bool vc_tls_walk_rev;
extern thread_local bool vc_tls_zigzag;        // the engine call being enqueued alternates directions (GEMM + LayerNorm pairs flip in between)
struct VcEosExtra { int32_t id[3]; };
extern thread_local VcEosExtra vc_tls_eos_extra;
__device__ __forceinline__ bool vc_is_eos(int tok, int eos, const VcEosExtra& x) {
  return tok == eos || tok == x.id[0] || tok == x.id[1] || tok == x.id[2];
}
#define VC_LIVE_EXIT(live)                         \
  do {                                             \
    if ((live) != nullptr && *(live) == 0) return; \
  } while (0)

// Engine-internal, timing runs only (vitcap_engine_timing_begin): when set, the large-GEMM launchers hand these two events to
// hipExtLaunchKernelGGL, which binds them to THE KERNEL DISPATCH (start = the kernel begins executing, stop = it has completed:
// the timestamps rocprofv3 --kernel-trace reports), instead of bracketing the launch with stream markers whose interval also holds
// the time the dispatch waited for the chip behind another stream's kernels.  vc_tls_kev_used tells the engine that a launcher took them.
extern thread_local hipEvent_t vc_tls_kev_start, vc_tls_kev_stop;
extern thread_local bool vc_tls_kev_used;

#define VC_REQUIRE(cond, ...)                 \
  do {                                        \
    if (!(cond)) {                            \
      vitcap_set_error(__VA_ARGS__);          \
      return VITCAP_EINVAL;                   \
    }                                         \
  } while (0)

#define VC_LAUNCH_CHECK(name)                                                   \
  do {                                                                          \
    hipError_t e_ = hipGetLastError();                                          \
    if (e_ != hipSuccess) {                                                     \
      vitcap_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
      return VITCAP_ELAUNCH;                                                    \
    }                                                                           \
  } while (0)


// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: set it once per (kernel, device), from
// whichever thread gets there first (two racing threads both set it: idempotent), and report a failure instead of letting
// it surface later as a generic launch error.  One bit per device ordinal in a per-call-site word.
#define VC_FUNC_SMEM(kern, bytes)                                                                                   \
  do {                                                                                                              \
    static std::atomic<unsigned long long> vc_done_{0ull};                                                          \
    int vc_dev_ = 0;                                                                                                \
    (void)hipGetDevice(&vc_dev_);                                                                                   \
    const unsigned long long vc_bit_ = 1ull << (vc_dev_ & 63);                                                      \
    if (!(vc_done_.load(std::memory_order_acquire) & vc_bit_)) {                                                    \
      hipError_t vc_e_ = hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)); \
      if (vc_e_ != hipSuccess) {                                                                                    \
        vitcap_set_error("hipFuncSetAttribute(max dynamic LDS = %d) failed on device %d: %s", (int)(bytes), vc_dev_, \
                         hipGetErrorString(vc_e_));                                                                 \
        return VITCAP_ELAUNCH;                                                                                      \
      }                                                                                                             \
      vc_done_.fetch_or(vc_bit_, std::memory_order_release);                                                        \
    }                                                                                                               \
  } while (0)

// LDS transpose read (ds_read_b64_tr_b16: a 16-lane group fetches 4 rows x 16 columns of 16-bit elements, lane i receives
// column i of the 4 rows) as INLINE ASM.  hipcc's waitcnt pass puts `s_waitcnt vmcnt(0)` in front of the ds_read_tr16
// INTRINSIC whenever an LDS-DMA (global_load_lds) is in flight -- it cannot tell that the DMA fills another ring slot;
// plain ds_read_b128 loads are not affected -- so the tile just requested would be awaited before the current one is
// multiplied (found in the ISA of csrc/attn.hip and csrc/gemm_tn.hip: the prefetch distance silently became zero).  The asm
// form is invisible to that pass; its results must be fenced by an explicit `s_waitcnt lgkmcnt(..)` that names the
// destination registers as "+v" operands before the MFMAs that consume them.
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
template <int OFF>
__device__ __forceinline__ s16x4 lds_tr_read(uint32_t addr) {
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ bf16x8 tr_pair(s16x4 lo, s16x4 hi) {
  return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// round-to-nearest-even fp32 -> bf16 (matches torch .to(bfloat16) for finite values and NaN->qNaN)
__device__ __forceinline__ bf16_t f2bf(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  // hipcc lowers the __bf16 casts to one v_cvt_pk_bf16_f32 (round-to-nearest-even)
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, v);
}

// GELU (erf form) = x Phi(x), Phi(x) = 1 - erfc(x/sqrt2)/2 (x >= 0) or erfc(-x/sqrt2)/2 (x < 0): no cancellation on the
// negative tail.  log2(erfc(z)) is smooth on [0,4] (0 ... -26), so erfc(z)/2 = exp2(q(z)) with a degree-7 polynomial
// (Chebyshev fit, constant term carries the 1/2): 7 FMA + one exp2, branch-free.  |gelu - exact| <= 7e-7 absolute and
// <= 5e-6 relative for |x| < 5.6; beyond that z is clamped to 4 (erfc(4) = 1.5e-8: gelu -> x or -> 0 within 1e-7).
// ocml's erff is piecewise (divergent) and ~3x the instructions; the Abramowitz-Stegun rational form needs a second
// transcendental (rcp).
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fminf(fabsf(x) * 0.70710678118654752f, 4.0f);
  float q = fmaf(-2.177763781e-05f, z, 5.068330793e-04f);
  q = fmaf(q, z, -5.339398049e-03f);
  q = fmaf(q, z, 3.423144668e-02f);
  q = fmaf(q, z, -1.528908461e-01f);
  q = fmaf(q, z, -9.167589545e-01f);
  q = fmaf(q, z, -1.628154397e+00f);
  q = fmaf(q, z, 6.178960575e-06f - 1.0f);
  const float h = __builtin_amdgcn_exp2f(q);        // erfc(|x|/sqrt2) / 2
  return x * (x >= 0.f ? 1.0f - h : h);
}

// Four at a time on the packed fp32 pipe: the polynomial is 7 v_pk_fma_f32 per PAIR (the scalar form is 7 v_fmaak per
// element; the fc1 epilogue is VALU-bound: 128 activations per thread, two waves per SIMD).  Same coefficients, same
// operation order per element, so the results are bit-identical to gelu_erf().
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
  const f32x2 t = x * f32x2{0.70710678118654752f, 0.70710678118654752f};
  const f32x2 z = f32x2{fminf(fabsf(t[0]), 4.0f), fminf(fabsf(t[1]), 4.0f)};
#define VC_C2(c_) f32x2{c_, c_}
  f32x2 q = __builtin_elementwise_fma(VC_C2(-2.177763781e-05f), z, VC_C2(5.068330793e-04f));
  q = __builtin_elementwise_fma(q, z, VC_C2(-5.339398049e-03f));
  q = __builtin_elementwise_fma(q, z, VC_C2(3.423144668e-02f));
  q = __builtin_elementwise_fma(q, z, VC_C2(-1.528908461e-01f));
  q = __builtin_elementwise_fma(q, z, VC_C2(-9.167589545e-01f));
  q = __builtin_elementwise_fma(q, z, VC_C2(-1.628154397e+00f));
  q = __builtin_elementwise_fma(q, z, VC_C2(6.178960575e-06f - 1.0f));
#undef VC_C2
  const f32x2 h = f32x2{__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
  const f32x2 c = f32x2{1.0f, 1.0f} - h;
  const f32x2 sel = f32x2{x[0] >= 0.f ? c[0] : h[0], x[1] >= 0.f ? c[1] : h[1]};
  return x * sel;
}
__device__ __forceinline__ f32x4 gelu_erf4(f32x4 v) {
  const f32x2 a = gelu_erf2(f32x2{v[0], v[1]}), b = gelu_erf2(f32x2{v[2], v[3]});
  return f32x4{a[0], a[1], b[0], b[1]};
}

// GELU and its derivative gelu'(x) = Phi(x) + x phi(x) from ONE erfc evaluation: the training forward stores gelu'(z) (bf16) next
// to gelu(z), so the input-gradient GEMM's epilogue multiplies by a stored factor instead of evaluating gelu' from a stored z
// (rcp + exp2 + 9 packed ops per pair there, against one more exp2 + 3 packed ops per pair here).  The returned gelu is
// bit-identical to gelu_erf2().
__device__ __forceinline__ f32x2 gelu_erf2_grad(f32x2 x, f32x2& grad) {
  const f32x2 t = x * f32x2{0.70710678118654752f, 0.70710678118654752f};
  const f32x2 z = f32x2{fminf(fabsf(t[0]), 4.0f), fminf(fabsf(t[1]), 4.0f)};
#define VC_C2(c_) f32x2{c_, c_}
  f32x2 q = __builtin_elementwise_fma(VC_C2(-2.177763781e-05f), z, VC_C2(5.068330793e-04f));
  q = __builtin_elementwise_fma(q, z, VC_C2(-5.339398049e-03f));
  q = __builtin_elementwise_fma(q, z, VC_C2(3.423144668e-02f));
  q = __builtin_elementwise_fma(q, z, VC_C2(-1.528908461e-01f));
  q = __builtin_elementwise_fma(q, z, VC_C2(-9.167589545e-01f));
  q = __builtin_elementwise_fma(q, z, VC_C2(-1.628154397e+00f));
  q = __builtin_elementwise_fma(q, z, VC_C2(6.178960575e-06f - 1.0f));
  const f32x2 h = f32x2{__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
  const f32x2 c = f32x2{1.0f, 1.0f} - h;
  const f32x2 sel = f32x2{x[0] >= 0.f ? c[0] : h[0], x[1] >= 0.f ? c[1] : h[1]};      // Phi(x)
  const f32x2 arg = (t * VC_C2(-1.4426950408889634f)) * t;                              // -x^2/2 in log2 units
  const f32x2 e = f32x2{__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
  grad = __builtin_elementwise_fma(x * VC_C2(0.3989422804014327f), e, sel);            // Phi(x) + x phi(x)
#undef VC_C2
  return x * sel;
}
__device__ __forceinline__ f32x4 gelu_erf4_grad(f32x4 v, f32x4& grad) {
  f32x2 ga, gb;
  const f32x2 a = gelu_erf2_grad(f32x2{v[0], v[1]}, ga), b = gelu_erf2_grad(f32x2{v[2], v[3]}, gb);
  grad = f32x4{ga[0], ga[1], gb[0], gb[1]};
  return f32x4{a[0], a[1], b[0], b[1]};
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// LayerNorm of one 768-wide row held by one wave (lane l: columns i*256 + 4l .. +3, i = 0..2): two-pass statistics, the arithmetic of
// nn.LayerNorm in fp32.  Shared by the LayerNorm kernels (norm.hip) and the GEMM epilogues that normalise their own finished rows
// (gemm.hip / gemm4w.hip), so that a row's result does not depend on which of them produced it.
constexpr int D768 = 768;
__device__ __forceinline__ void ln_row(const f32x4 (&v)[3], const float* gamma, const float* beta, float eps,
                                       int lane, bf16_t* yb, float* yf) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  const float mean = wave_sum(s) * (1.0f / D768);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = v[i][e] - mean;
      q += d * d;
    }
  const float var = wave_sum(q) * (1.0f / D768);
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = i * 256 + lane * 4;
    const f32x4 g = *(const f32x4*)(gamma + c);
    const f32x4 b = *(const f32x4*)(beta + c);
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
    if (yf) *(f32x4*)(yf + c) = y;
    if (yb) {
      uint2 o;
      o.x = pack2bf(y[0], y[1]);
      o.y = pack2bf(y[2], y[3]);
      *(uint2*)(yb + c) = o;
    }
  }
}

typedef __attribute__((ext_vector_type(4))) unsigned vc_u32x4;
// raw buffer descriptor over `bytes` bytes from `base` (clipped to the 4 GiB a descriptor can span); out-of-range lanes read 0 / are dropped
__device__ __forceinline__ __amdgpu_buffer_rsrc_t vc_rsrc(const void* base, long long bytes) {
  const unsigned rec = bytes <= 0 ? 0u : (bytes > 0xffffffffll ? 0xffffffffu : (unsigned)bytes);
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, rec, 0x00020000);
}


// Visibility of key row k to query row q in the decoder's joint sequence under teacher forcing, rows laid out as
// [0, cf) visual | [cf, mf) caption tokens | [mf, S) [MASK] probes (mf = 0: no probe rows):
//   visual rows see visual rows only; token row i sees visual + tokens <= i (seq2seq mask, dataset.py:377-390);
//   probe row j (the [MASK] the generator appends at position j+1, modeling_bert.py:846-876) sees visual, tokens <= j
//   and itself -- exactly the rows one decode step's [MASK] query attends.  cf = 0: plain dense attention.
// (Keep it branch-free: an equivalent early-return formulation of the same predicate produced wrong masks when inlined
// into the MFMA tile macros by hipcc 7.2 -O3 -- caught by test_attn_joint_causal_forward_backward.)
__device__ __forceinline__ bool joint_visible(int q, int k, int S, int cf, int mf) {
  const bool plain = cf <= 0 || k < cf;                                   // visual key (or no mask at all)
  const bool causal = k <= q;                                             // token row / token key (q >= cf follows from k >= cf)
  const bool probe_q = mf > 0 && q >= mf, probe_k = mf > 0 && k >= mf;
  const bool text = probe_k ? (k == q) : (probe_q ? (k - cf) <= (q - mf) : causal);
  return k < S && (plain || text);
}
