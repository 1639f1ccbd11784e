// Error reporting + version for the C-ABI library.
#include "common.h"

#include <stdarg.h>

static thread_local char g_err[512] = "no error";

void cmbpo_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *cmbpo_last_error(void) { return g_err; }

extern "C" int cmbpo_version(void) { return 1; }
