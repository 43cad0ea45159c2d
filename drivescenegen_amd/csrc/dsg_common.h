// Internal helpers shared by the libdsg.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/dsg.h"

#define DSG_API extern "C" __attribute__((visibility("default")))

namespace dsg {

// thread-local error text behind dsg_last_error()
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

#define DSG_CHECK_ARG(cond, ...)                                      \
  do {                                                                \
    if (!(cond)) return ::dsg::fail(DSG_ERR_INVALID_ARG, __VA_ARGS__); \
  } while (0)

#define DSG_CHECK_SHAPE(cond, ...)                                          \
  do {                                                                      \
    if (!(cond)) return ::dsg::fail(DSG_ERR_UNSUPPORTED_SHAPE, __VA_ARGS__); \
  } while (0)

#define DSG_HIP(expr)                                                                            \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      return ::dsg::fail(DSG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),     \
                         __FILE__, __LINE__);                                                    \
  } while (0)

// kernel launches report asynchronous-launch errors immediately
#define DSG_LAUNCH_CHECK()                                                                   \
  do {                                                                                       \
    hipError_t e_ = hipGetLastError();                                                       \
    if (e_ != hipSuccess)                                                                    \
      return ::dsg::fail(DSG_ERR_HIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), \
                         __FILE__, __LINE__);                                                \
  } while (0)

// Zero-fill of `words` 32-bit words as a KERNEL launch.  Not hipMemsetAsync: captured into a hipGraph a memset becomes a memset
// node, and on this stack (ROCm 7.2, gfx950) replays 2.. of a captured dsg_unet_forward did not order that node against the
// kernel nodes behind it -- the range-guard slots were zeroed while their consumers ran (found by
// tests/test_gpu_unet.py::test_forward_and_scheduler_step_are_legal_under_stream_capture; a kernel node is ordered).
hipError_t zero_words(void* p, size_t words, hipStream_t st);

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

}  // namespace dsg
