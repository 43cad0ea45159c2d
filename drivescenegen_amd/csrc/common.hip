// Error reporting and version of libdsg.so (C ABI: include/dsg.h).
#include "dsg_common.h"

namespace dsg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace dsg

DSG_API int dsg_version(void) { return 100; }  // 0.1.0
DSG_API const char* dsg_last_error(void) { return dsg::g_err; }
