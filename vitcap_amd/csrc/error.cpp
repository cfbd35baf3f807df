// Thread-local error text + version for libvitcap_hip.so
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdint.h>
#include "../../include/vitcap_hip.h"
#include "common.h"

static thread_local char g_err[512] = "";
thread_local const int32_t* vc_tls_live = nullptr;   // see common.h
thread_local VcEosExtra vc_tls_eos_extra = {{-1, -1, -1}};
thread_local hipEvent_t vc_tls_kev_start = nullptr, vc_tls_kev_stop = nullptr;
thread_local bool vc_tls_kev_used = false;
thread_local bool vc_tls_walk_rev = false;         // see common.h
thread_local bool vc_tls_zigzag = false;
thread_local const uint32_t* vc_tls_drop_salt = nullptr;
extern "C" int vitcap_set_dropout_salt(const void* device_u32) {
  vc_tls_drop_salt = (const uint32_t*)device_u32;
  return 0;
}

void vitcap_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vitcap_last_error(void) { return g_err; }
extern "C" int vitcap_version(void) { return VITCAP_ABI_VERSION; }
