// Probe (not product): the 4-wave GEMM kernel of vitcap_amd/csrc/gemm4w.hip alone, with shader-clock stamps per workgroup
// (prologue / main loop / epilogue) and hipEvent timing.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVC_4W_STAMP [-D...] tools/probes/g4w_probe.hip -o tools/probes/_bin/g4w_probe
//   tools/probes/_bin/g4w_probe [M]
#include "../../vitcap_amd/csrc/gemm4w.hip"

#include <string.h>
#include <algorithm>
#include <vector>

thread_local const int32_t* vc_tls_live = nullptr;
thread_local VcEosExtra vc_tls_eos_extra = {{-1, -1, -1}};
thread_local hipEvent_t vc_tls_kev_start = nullptr, vc_tls_kev_stop = nullptr;
thread_local bool vc_tls_kev_used = false;
void vitcap_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  fprintf(stderr, "\n");
  va_end(ap);
}
int vc_tile_group_n(int tiles_n) {
  const char* e = getenv("VITCAP_GEMM_GROUP_N");
  int g = e ? atoi(e) : (tiles_n > 9 ? 3 : tiles_n);
  return g > tiles_n ? tiles_n : g;
}

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));        \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

static unsigned rng = 12345;
static float frand() {
  rng = rng * 1664525u + 1013904223u;
  return ((rng >> 8) & 0xffff) / 32768.0f - 1.0f;
}
static bf16_t f2bf_h(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 36928;
  const int form = argc > 2 ? atoi(argv[2]) : 0;
  struct Shape { const char* name; int N, K, act, f32; bool res; };
  const Shape shapes[] = {{"qkv", 2304, 768, VITCAP_ACT_NONE, 0, false}, {"proj", 768, 768, VITCAP_ACT_NONE, 1, true},
                          {"fc1", 3072, 768, VITCAP_ACT_GELU_ERF, 0, false}, {"fc2", 768, 3072, VITCAP_ACT_NONE, 1, true}};
  const size_t maxA = (size_t)M * 3072, maxC = (size_t)M * 3072;
  std::vector<bf16_t> ha(maxA), hw((size_t)3072 * 3072);
  for (auto& v : ha) v = f2bf_h(frand());
  for (auto& v : hw) v = f2bf_h(frand() * 0.05f);
  bf16_t *dA, *dW;
  float *dBias, *dRes;
  void* dC;
  unsigned long long* dStamp;
  CK(hipMalloc(&dA, maxA * 2));
  CK(hipMalloc(&dW, hw.size() * 2));
  CK(hipMalloc(&dBias, 3072 * 4));
  CK(hipMalloc(&dRes, maxC * 4));
  CK(hipMalloc(&dC, maxC * 4));
  CK(hipMalloc(&dStamp, 65536 * 4 * 8));
  CK(hipMemcpy(dA, ha.data(), maxA * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(dBias, 0, 3072 * 4));
  CK(hipMemset(dRes, 0, maxC * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (const Shape& sh : shapes) {
    GemmArgs a{};
    a.A = dA; a.W = dW; a.bias = dBias; a.res = sh.res ? dRes : nullptr; a.C = dC;
    a.M = M; a.N = sh.N; a.K = sh.K; a.lda = sh.K; a.ldw = sh.K; a.ldc = sh.N; a.ldr = sh.N;
    a.split_k = 1;
    a.direct_epilogue = argc > 3 ? atoi(argv[3]) : 0;
    for (int i = 0; i < 3; ++i) vc_dispatch_4w(a, sh.act, sh.f32, 0, form);
    CK(hipDeviceSynchronize());
    const int iters = 20;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) vc_dispatch_4w(a, sh.act, sh.f32, 0, form);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, tf = 2.0 * M * sh.N * sh.K / us * 1e-6;
    // stamp run
    const int mi = pick_mi(M, (sh.N + 255) / 256, form == 0 ? 1 : form);
    const int nwg = form == 0 ? ((M + 255) / 256) * ((sh.N + 255) / 256) : ((M + 32 * mi - 1) / (32 * mi)) * ((sh.N + 255) / 256);
    a.rowstat = (float*)dStamp;
    CK(hipMemset(dStamp, 0, (size_t)nwg * 32));
    vc_dispatch_4w(a, sh.act, sh.f32, 0, form);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st((size_t)nwg * 4);
    CK(hipMemcpy(st.data(), dStamp, st.size() * 8, hipMemcpyDeviceToHost));
    // workgroups are dispatched in blockIdx order: the first 256 run in round 1 (all in phase), later ones de-phased
    auto avg = [&](int lo, int hi, int i0, int i1) {
      double s = 0;
      int n = 0;
      for (int b = lo; b < hi && b < nwg; ++b) { s += (double)(st[b * 4 + i1] - st[b * 4 + i0]); ++n; }
      return n ? s / n : 0.0;
    };
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < nwg; ++b) { tmin = std::min(tmin, st[b * 4]); tmax = std::max(tmax, st[b * 4 + 3]); }
    printf("M=%d %-4s N=%4d K=%4d form %d MI %d: %8.1f us %6.0f TF | tiles %d | clk ticks: first 256 tiles: first k-tile (form 0/1: prologue) %.0f loop %.0f epilogue %.0f | later tiles: %.0f loop %.0f "
           "epilogue %.0f | launch span %.0f ticks\n",
           M, sh.name, sh.N, sh.K, form, mi, us, tf, nwg, avg(0, 256, 0, 1), avg(0, 256, 1, 2), avg(0, 256, 2, 3), avg(256, nwg, 0, 1), avg(256, nwg, 1, 2),
           avg(256, nwg, 2, 3), (double)(tmax - tmin));
  }
  return 0;
}
