// conv_h2_kernel for fp16 tensors / v_mfma_f32_32x32x16_f16 with ONE product per MAC (dsg_conv_args.compute_dtype ==
// DSG_F16): the reference's own training precision, Accelerator(mixed_precision='fp16') at
// DriveSceneGen/scripts/train.py:24 / pipeline/training_pipeline.py:48-49.  Kernel: conv_h2_kernel.h.
#include "conv_h2_launch.h"

namespace dsg {
int conv_h2_launch_f16(const dsg_conv_args* a, int hout, int wout, hipStream_t st) {
  return conv_h2_launch_t<2>(a, hout, wout, st);
}
}  // namespace dsg
