// capi.hip -- version / error plumbing of the C ABI (include/fpsg_hip.h).
#include <stdarg.h>
#include <string.h>

#include "fpsg_common.h"

namespace fpsg {
namespace {
thread_local char g_err[512] = "";
}
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace fpsg

extern "C" int fpsg_version(void) { return FPSG_ABI_VERSION; }
extern "C" const char* fpsg_last_error(void) { return fpsg::g_err; }
