// Error text plumbing for the C ABI (no exceptions cross the boundary).
#include <stdarg.h>

#include "common.h"

namespace {
thread_local char g_err[512] = {0};
}

void ldm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int ldm_version(void) { return 100; }

extern "C" int ldm_last_error(char* buf, int n) {
  if (!buf || n <= 0) return (int)strlen(g_err);
  strncpy(buf, g_err, (size_t)n - 1);
  buf[n - 1] = 0;
  return (int)strlen(buf);
}
