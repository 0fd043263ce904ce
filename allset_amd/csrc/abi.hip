// Versions + thread-local error text of the C ABI (include/allset_hip.h: core; include/allset_hip_ext.h: the rest).  No state besides the
// per-thread message buffer; nothing here touches the device.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace allset {

static thread_local char g_error[512] = {0};

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

void clear_error() { g_error[0] = '\0'; }

}  // namespace allset

extern "C" int allset_version(void) { return ALLSET_ABI_VERSION; }

extern "C" int allset_core_version(void) { return ALLSET_CORE_ABI_VERSION; }

extern "C" const char* allset_last_error(void) { return allset::g_error; }
