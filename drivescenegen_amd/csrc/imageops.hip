// Image-side kernels next to the denoising path: HBM-bound byte/pixel streaming.
//
//  f1  input pipeline -- Image_Dataset.__getitem__ (DriveSceneGen/utils/datasets/dataset.py:21-24,43-45):
//      ToTensor (u8 HWC -> f32 CHW / 255), Resize((H, W), antialias=False) = bilinear, align_corners=False,
//      Normalize([0.5], [0.5]); one kernel, batched, straight from the decoded uint8 image.
//  f2  first vectorisation stage -- get_gray_image (DriveSceneGen/vectorization/utils/image_utils.py:13-43):
//      per-channel 256-bin histograms of the generated uint8 images (the peak is the background value) and
//      the +-0.1 background mask; extract_agents' threshold (vectorization/direct/extract_vehicles.py:136-148).
//      Both masks are byte look-ups: the host builds the 256-entry tables with the reference's own float
//      arithmetic, so results are bit-identical by construction.
#include "dsg_h16.h"
#include <algorithm>

namespace dsg {

// grid = (ceil(wo*ho/256), c, n).  SrcT = uint8_t: a decoded image, ToTensor's / 255 applied; float: the .pkl branch's
// `fig_tensor` [H][W][C] (dataset.py:37-41), used as it is.
template <typename SrcT>
__global__ __launch_bounds__(256) void resize_normalize_kernel(const SrcT* __restrict__ src, int hs, int ws, int c,
                                                               float* __restrict__ dst, int ho, int wo,
                                                               float scale_h, float scale_w, float mean,
                                                               float inv_std) {
  constexpr bool U8 = sizeof(SrcT) == 1;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= ho * wo) return;
  const int ci = blockIdx.y, n = blockIdx.z;
  const int oy = pix / wo, ox = pix - oy * wo;
  // torch area_pixel_compute_source_index(align_corners=False): max(scale*(dst+0.5)-0.5, 0)
  const float sy = fmaxf(scale_h * ((float)oy + 0.5f) - 0.5f, 0.f);
  const float sx = fmaxf(scale_w * ((float)ox + 0.5f) - 0.5f, 0.f);
  const int y0 = min((int)sy, hs - 1), x0 = min((int)sx, ws - 1);
  const int y1 = min(y0 + 1, hs - 1), x1 = min(x0 + 1, ws - 1);
  const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
  const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const SrcT* sp = src + (size_t)n * hs * ws * c + ci;
  auto at = [&](int y, int x) -> float {
    const float v = (float)sp[((size_t)y * ws + x) * c];
    return U8 ? v / 255.0f : v;
  };
  const float v00 = at(y0, x0), v01 = at(y0, x1), v10 = at(y1, x0), v11 = at(y1, x1);
  const float v = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
  dst[(((size_t)n * c + ci) * ho + oy) * wo + ox] = (v - mean) * inv_std;
}

// grid = (blocks, c, n); hist[n][c][256] must be zeroed by the caller
__global__ __launch_bounds__(256) void hist_u8_kernel(const uint8_t* __restrict__ img, int hw, int c,
                                                      unsigned* __restrict__ hist) {
  __shared__ unsigned h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int ci = blockIdx.y, n = blockIdx.z;
  const uint8_t* p = img + (size_t)n * hw * c + ci;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) atomicAdd(&h[p[(size_t)i * c]], 1u);
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[((size_t)n * c + ci) * 256 + threadIdx.x], h[threadIdx.x]);
}

// out[n][p] = (lut[n][0][img[n][p][ch0]] & (ch1 < 0 ? 1 : lut[n][1][img[n][p][ch1]])) ? on : off
__global__ __launch_bounds__(256) void mask_lut_kernel(const uint8_t* __restrict__ img, int hw, int c, int ch0, int ch1,
                                                       const uint8_t* __restrict__ lut, uint8_t on, uint8_t off,
                                                       uint8_t* __restrict__ out) {
  const int n = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= hw) return;
  const uint8_t* p = img + ((size_t)n * hw + i) * c;
  const uint8_t* l = lut + (size_t)n * 512;
  bool b = l[p[ch0]] != 0;
  if (ch1 >= 0) b = b && (l[256 + p[ch1]] != 0);
  out[(size_t)n * hw + i] = b ? on : off;
}


// [N][C][hw] fp32 <-> [N][C/8][hw][8] (fp32, or the 16-bit type dt: 1 bf16, 2 fp16): one thread per (n, c/8, pixel),
// eight channels each
__global__ __launch_bounds__(256) void layout_convert_kernel(const void* __restrict__ srcv, void* __restrict__ dstv, int c,
                                                             int hw, int64_t total, int to_blocked, int dt) {
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (i >= total) return;
  const int px = (int)(i % hw);
  const int64_t blk = i / hw;  // n * (c/8) + cb
  const size_t plain = (size_t)blk * 8 * hw + px, blocked = ((size_t)blk * hw + px) * 8;
  if (to_blocked) {
    const float* src = static_cast<const float*>(srcv);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[plain + (size_t)j * hw];
    if (dt) {
      *reinterpret_cast<uint4*>(static_cast<unsigned short*>(dstv) + blocked) =
          make_uint4(word_pack(v[0], v[1], dt), word_pack(v[2], v[3], dt), word_pack(v[4], v[5], dt), word_pack(v[6], v[7], dt));
    } else {
      float4* o = reinterpret_cast<float4*>(static_cast<float*>(dstv) + blocked);
      o[0] = make_float4(v[0], v[1], v[2], v[3]);
      o[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
  } else {
    float* dst = static_cast<float*>(dstv);
    float v[8];
    if (dt) {
      const uint4 q = *reinterpret_cast<const uint4*>(static_cast<const unsigned short*>(srcv) + blocked);
      v[0] = word_lo(q.x, dt); v[1] = word_hi(q.x, dt); v[2] = word_lo(q.y, dt); v[3] = word_hi(q.y, dt);
      v[4] = word_lo(q.z, dt); v[5] = word_hi(q.z, dt); v[6] = word_lo(q.w, dt); v[7] = word_hi(q.w, dt);
    } else {
      const float4* q = reinterpret_cast<const float4*>(static_cast<const float*>(srcv) + blocked);
      const float4 a = q[0], b = q[1];
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[plain + (size_t)j * hw] = v[j];
  }
}

}  // namespace dsg

DSG_API int dsg_resize_normalize_u8(const uint8_t* src, int32_t n, int32_t hs, int32_t ws, int32_t c, float* dst,
                                    int32_t ho, int32_t wo, float mean, float std, void* stream) {
  DSG_CHECK_ARG(src && dst, "dsg_resize_normalize_u8: NULL pointer");
  DSG_CHECK_ARG(n > 0 && hs > 0 && ws > 0 && c > 0 && ho > 0 && wo > 0 && std != 0.f && n <= 65535 && c <= 65535,
                "dsg_resize_normalize_u8: bad dims");
  hipLaunchKernelGGL(dsg::resize_normalize_kernel<uint8_t>, dim3(dsg::cdiv(ho * wo, 256), c, n), dim3(256), 0,
                     static_cast<hipStream_t>(stream), src, hs, ws, c, dst, ho, wo, (float)hs / (float)ho,
                     (float)ws / (float)wo, mean, 1.0f / std);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_resize_normalize_f32(const float* src, int32_t n, int32_t hs, int32_t ws, int32_t c, float* dst,
                                     int32_t ho, int32_t wo, float mean, float std, void* stream) {
  DSG_CHECK_ARG(src && dst, "dsg_resize_normalize_f32: NULL pointer");
  DSG_CHECK_ARG(n > 0 && hs > 0 && ws > 0 && c > 0 && ho > 0 && wo > 0 && std != 0.f && n <= 65535 && c <= 65535,
                "dsg_resize_normalize_f32: bad dims");
  hipLaunchKernelGGL(dsg::resize_normalize_kernel<float>, dim3(dsg::cdiv(ho * wo, 256), c, n), dim3(256), 0,
                     static_cast<hipStream_t>(stream), src, hs, ws, c, dst, ho, wo, (float)hs / (float)ho,
                     (float)ws / (float)wo, mean, 1.0f / std);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_hist_u8(const uint8_t* img, int32_t n, int32_t hw, int32_t c, uint32_t* hist, void* stream) {
  DSG_CHECK_ARG(img && hist, "dsg_hist_u8: NULL pointer");
  DSG_CHECK_ARG(n > 0 && hw > 0 && c > 0 && n <= 65535 && c <= 65535, "dsg_hist_u8: bad dims");
  hipStream_t st = static_cast<hipStream_t>(stream);
  DSG_HIP(dsg::zero_words(hist, (size_t)n * c * 256, st));
  const int blocks = std::max(1, std::min(64, dsg::cdiv(hw, 4096)));
  hipLaunchKernelGGL(dsg::hist_u8_kernel, dim3(blocks, c, n), dim3(256), 0, st, img, hw, c, hist);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_mask_lut_u8(const uint8_t* img, int32_t n, int32_t hw, int32_t c, int32_t ch0, int32_t ch1,
                            const uint8_t* lut, uint8_t on_value, uint8_t off_value, uint8_t* out, void* stream) {
  DSG_CHECK_ARG(img && lut && out, "dsg_mask_lut_u8: NULL pointer");
  DSG_CHECK_ARG(n > 0 && hw > 0 && c > 0 && ch0 >= 0 && ch0 < c && ch1 < c && n <= 65535, "dsg_mask_lut_u8: bad dims");
  hipLaunchKernelGGL(dsg::mask_lut_kernel, dim3(dsg::cdiv(hw, 256), n), dim3(256), 0, static_cast<hipStream_t>(stream),
                     img, hw, c, ch0, ch1, lut, on_value, off_value, out);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_layout_convert_dt(const void* src, void* dst, int32_t n, int32_t c, int32_t hw, int32_t to_blocked,
                                  int32_t blocked_dtype, void* stream) {
  DSG_CHECK_ARG(src && dst && src != dst, "dsg_layout_convert: NULL pointer or in-place");
  DSG_CHECK_ARG(n > 0 && c > 0 && c % 8 == 0 && hw > 0, "dsg_layout_convert: bad dims (C %% 8 != 0?)");
  DSG_CHECK_ARG(blocked_dtype >= DSG_F32 && blocked_dtype <= DSG_F16, "dsg_layout_convert: bad blocked_dtype %d", blocked_dtype);
  const int64_t total = (int64_t)n * (c / 8) * hw;
  hipLaunchKernelGGL(dsg::layout_convert_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), src, dst, c, hw, total, to_blocked, blocked_dtype);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_layout_convert(const float* src, float* dst, int32_t n, int32_t c, int32_t hw, int32_t to_blocked,
                               void* stream) {
  return dsg_layout_convert_dt(src, dst, n, c, hw, to_blocked, DSG_F32, stream);
}
