// conv_h2_kernel for bf16 tensors / v_mfma_f32_32x32x16_bf16 (dsg_conv_args.compute_dtype == DSG_BF16): BASELINE.json
// configs[4] "mixed bf16".  Kernel: conv_h2_kernel.h; dispatch: conv_h2_launch.h.
#include "conv_h2_launch.h"

namespace dsg {
int conv_h2_launch_bf16(const dsg_conv_args* a, int hout, int wout, hipStream_t st) {
  return conv_h2_launch_t<1>(a, hout, wout, st);
}
}  // namespace dsg
