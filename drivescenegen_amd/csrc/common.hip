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

__global__ void zero_words_kernel(unsigned* __restrict__ p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}

hipError_t zero_words(void* p, size_t words, hipStream_t st) {
  if (words == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((words + 255) / 256 < 2048 ? (words + 255) / 256 : 2048);
  hipLaunchKernelGGL(zero_words_kernel, dim3(blocks), dim3(256), 0, st, static_cast<unsigned*>(p), words);
  return hipGetLastError();
}

}  // namespace dsg

DSG_API int dsg_version(void) { return 100; }  // 0.1.0
DSG_API const char* dsg_last_error(void) { return dsg::g_err; }
