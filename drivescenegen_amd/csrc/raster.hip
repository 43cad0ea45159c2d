// Direct scene rasteriser: what the reference draws with matplotlib-Agg -- lane way-points as
// scatter diamonds or 1.5-pt segments coloured (dx, dy, 0) on a 0.5-grey canvas
// (DriveSceneGen/utils/datasets/rasterization.py:57-126) and agent rectangles coloured (0, 0, speed/60 + 0.5) on black
// (DriveSceneGen/utils/datasets/visualization.py:283-296) -- as one pass over a list of antialiased oriented boxes.
//
// Every primitive is an oriented box (cx, cy, ux, uy, hx, hy, r, g, b) in pixel space (a diamond is a box turned by
// 45 degrees, a capped segment a box around it); boxes are composited in list order, out = out*(1-cov) + colour*cov,
// cov = product over the two box axes of clamp(0.5 - d/w, 0, 1), d = signed distance of the pixel centre to the edge
// pair, w = |ux| + |uy|.  One thread per pixel walks the list from LDS (order matters, so the loop is per pixel);
// a bounding-radius test rejects almost every box in 4 operations.  HBM traffic: the box list once per workgroup
// (L2-resident) and 12 B per pixel out -- the kernel is VALU-bound by the reject test.
#include "dsg_common.h"

namespace dsg {

constexpr int RB_CHUNK = 256;

__global__ __launch_bounds__(256) void rasterize_boxes_kernel(const float* __restrict__ boxes, int nbox,
                                                              float* __restrict__ out, int h, int w, float bg0,
                                                              float bg1, float bg2) {
  __shared__ float sb[RB_CHUNK][10];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int x = blockIdx.x * 16 + tx, y = blockIdx.y * 16 + ty;
  const float px = (float)x + 0.5f, py = (float)y + 0.5f;
  // tile centre / radius: boxes that cannot touch the 16x16 tile are dropped while the chunk is loaded
  const float tcx = (float)(blockIdx.x * 16) + 8.f, tcy = (float)(blockIdx.y * 16) + 8.f;
  float c0 = bg0, c1 = bg1, c2 = bg2;
  for (int base = 0; base < nbox; base += RB_CHUNK) {
    __syncthreads();
    const int i = base + (int)threadIdx.x;
    float rad = -1.f;
    if (i < nbox) {
      const float* b = boxes + (size_t)i * 9;
#pragma unroll
      for (int k = 0; k < 9; ++k) sb[threadIdx.x][k] = b[k];
      const float r = sqrtf(b[4] * b[4] + b[5] * b[5]) + 1.5f;
      rad = (fabsf(b[0] - tcx) > r + 8.f || fabsf(b[1] - tcy) > r + 8.f) ? -1.f : r;
    }
    sb[threadIdx.x][9] = rad;
    __syncthreads();
    const int cnt = min(RB_CHUNK, nbox - base);
    for (int j = 0; j < cnt; ++j) {
      const float rad_j = sb[j][9];
      if (rad_j < 0.f) continue;  // uniform
      const float dx = px - sb[j][0], dy = py - sb[j][1];
      if (fabsf(dx) > rad_j || fabsf(dy) > rad_j) continue;
      const float ux = sb[j][2], uy = sb[j][3];
      const float iw = 1.0f / (fabsf(ux) + fabsf(uy));
      const float ca = fminf(fmaxf(0.5f - (fabsf(dx * ux + dy * uy) - sb[j][4]) * iw, 0.f), 1.f);
      const float cb = fminf(fmaxf(0.5f - (fabsf(dy * ux - dx * uy) - sb[j][5]) * iw, 0.f), 1.f);
      const float cov = ca * cb;
      c0 = c0 * (1.f - cov) + sb[j][6] * cov;
      c1 = c1 * (1.f - cov) + sb[j][7] * cov;
      c2 = c2 * (1.f - cov) + sb[j][8] * cov;
    }
  }
  if (x < w && y < h) {
    const size_t plane = (size_t)h * w, o = (size_t)y * w + x;
    out[o] = c0;
    out[plane + o] = c1;
    out[2 * plane + o] = c2;
  }
}

}  // namespace dsg

DSG_API int dsg_rasterize_boxes(const float* boxes, int32_t nbox, float* out, int32_t h, int32_t w, float bg0, float bg1,
                                float bg2, void* stream) {
  DSG_CHECK_ARG(out != nullptr && (boxes != nullptr || nbox == 0), "dsg_rasterize_boxes: NULL pointer");
  DSG_CHECK_ARG(nbox >= 0 && h > 0 && w > 0, "dsg_rasterize_boxes: bad dims");
  hipLaunchKernelGGL(dsg::rasterize_boxes_kernel, dim3(dsg::cdiv(w, 16), dsg::cdiv(h, 16)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), boxes, nbox, out, h, w, bg0, bg1, bg2);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}
