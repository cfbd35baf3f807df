// Thread-local error text + version for libvitcap_hip.so
#include <stdarg.h>
#include <stdio.h>
#include "../../include/vitcap_hip.h"

static thread_local char g_err[512] = "";

void vitcap_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vitcap_last_error(void) { return g_err; }
extern "C" int vitcap_version(void) { return 1; }
