// Error channel and version of libdmhomo_hip.so (no exceptions cross the C ABI).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/dmhomo_hip.h"

static thread_local char g_err[512] = "";

void dmh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dmh_last_error(void) { return g_err; }
extern "C" int dmh_version(void) { return DMH_ABI_VERSION; }
