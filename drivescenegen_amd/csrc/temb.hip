// Timestep path for gfx950: sinusoid -> linear_1 -> SiLU -> linear_2 -> SiLU, and the batched
// time_emb_proj of all resnets as one small GEMV-class product.
//
// Replaces UNet2DModel.time_proj / time_embedding and every ResnetBlock2D.time_emb_proj(silu(temb))
// (diffusers 0.20.0 as used at DriveSceneGen/scripts/train.py:39-57; SURVEY.md App. A.2 lines 1-3).
// All resnets consume silu(temb), so the activation is applied once here.  Work is a few MFLOP per
// sample: latency-bound, one workgroup per sample / per 4 output rows.
#include "dsg_common.h"

namespace dsg {

// grid = n, block = 256; dynamic LDS = (ch + dim) floats
__global__ __launch_bounds__(256) void time_embed_kernel(const int64_t* __restrict__ timesteps,
                                                         const float* __restrict__ freqs, int ch, int dim,
                                                         const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2,
                                                         float* __restrict__ act, float* __restrict__ emb_out,
                                                         float* __restrict__ z1_out, float* __restrict__ z2_out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* e = sm;        // [ch] sinusoid, cos first (flip_sin_to_cos=True, freq_shift=0)
  float* h1 = sm + ch;  // [dim]
  const int n = blockIdx.x;
  const float t = (float)timesteps[n];
  const int half = ch / 2;
  for (int i = threadIdx.x; i < half; i += 256) {
    // freqs[i] = exp(-ln(10000) * i / half) comes from the host: at t ~ 1000 one ulp of the frequency
    // moves the angle by 6e-5 rad, so the table must be the reference's own fp32 values
    const float a = t * freqs[i];
    e[i] = cosf(a);
    e[half + i] = sinf(a);
  }
  __syncthreads();
  if (emb_out)
    for (int i = threadIdx.x; i < ch; i += 256) emb_out[(size_t)n * ch + i] = e[i];
  for (int j = threadIdx.x; j < dim; j += 256) {
    const float* wr = w1 + (size_t)j * ch;
    float s = 0.f;
    for (int k = 0; k < ch; ++k) s = fmaf(wr[k], e[k], s);
    if (z1_out) z1_out[(size_t)n * dim + j] = s + b1[j];
    h1[j] = silu_f(s + b1[j]);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < dim; j += 256) {
    const float* wr = w2 + (size_t)j * dim;
    float s = 0.f;
    for (int k = 0; k < dim; ++k) s = fmaf(wr[k], h1[k], s);
    if (z2_out) z2_out[(size_t)n * dim + j] = s + b2[j];
    act[(size_t)n * dim + j] = silu_f(s + b2[j]);
  }
}

// y[n][j] = x[n][:] . w[j][:] + b[j].  One wave per output row j, lanes stride the K axis
// (coalesced weight reads); grid = ceil(out_f / 4), block = 256.
// Eight images per pass: their dot products and shuffle trees are independent chains (the one-image-at-a-time loop was a
// single dependent chain of ~n x 100 cycles: 90 us at batch 64 for 0.2 GFLOP); per image the arithmetic -- lane-strided
// fmaf partials, then the shfl_down tree -- is unchanged, so results are bit-identical to the old kernel's.
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ y, int n,
                                                          int in_f, int out_f) {
  constexpr int NB = 8;
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= out_f) return;
  const float* wr = w + (size_t)j * in_f;
  const float bj = b ? b[j] : 0.f;
  for (int n0 = 0; n0 < n; n0 += NB) {
    float s[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) s[i] = 0.f;
    for (int k = lane; k < in_f; k += 64) {
      const float wk = wr[k];
#pragma unroll
      for (int i = 0; i < NB; ++i) s[i] = fmaf(wk, x[(size_t)min(n0 + i, n - 1) * in_f + k], s[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int i = 0; i < NB; ++i) s[i] += __shfl_down(s[i], o, 64);
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < NB; ++i)
        if (n0 + i < n) y[(size_t)(n0 + i) * out_f + j] = s[i] + bj;
    }
  }
}

}  // namespace dsg

static int time_embed_impl(const int64_t* timesteps, const float* freqs, int32_t n, int32_t ch, int32_t dim,
                           const float* w1, const float* b1, const float* w2, const float* b2, float* act,
                           float* emb, float* z1, float* z2, void* stream) {
  DSG_CHECK_ARG(timesteps && freqs && w1 && b1 && w2 && b2 && act, "dsg_time_embed_fwd: NULL pointer");
  DSG_CHECK_ARG(n > 0 && ch > 0 && (ch % 2) == 0 && dim > 0, "dsg_time_embed_fwd: bad dims");
  const size_t lds = (size_t)(ch + dim) * sizeof(float);
  DSG_CHECK_SHAPE(lds <= 64 * 1024, "dsg_time_embed_fwd: ch + dim too large (%d + %d)", ch, dim);
  hipLaunchKernelGGL(dsg::time_embed_kernel, dim3(n), dim3(256), lds, static_cast<hipStream_t>(stream), timesteps,
                     freqs, ch, dim, w1, b1, w2, b2, act, emb, z1, z2);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_time_embed_fwd(const int64_t* timesteps, const float* freqs, int32_t n, int32_t ch, int32_t dim,
                               const float* w1, const float* b1, const float* w2, const float* b2, float* act,
                               void* stream) {
  return time_embed_impl(timesteps, freqs, n, ch, dim, w1, b1, w2, b2, act, nullptr, nullptr, nullptr, stream);
}

// Training variant: also returns the sinusoid [N][ch] and the two pre-activations [N][dim] for the backward.
DSG_API int dsg_time_embed_fwd_train(const int64_t* timesteps, const float* freqs, int32_t n, int32_t ch, int32_t dim,
                                     const float* w1, const float* b1, const float* w2, const float* b2, float* act,
                                     float* emb, float* z1, float* z2, void* stream) {
  DSG_CHECK_ARG(emb && z1 && z2, "dsg_time_embed_fwd_train: NULL pointer");
  return time_embed_impl(timesteps, freqs, n, ch, dim, w1, b1, w2, b2, act, emb, z1, z2, stream);
}

DSG_API int dsg_linear_fwd(const float* x, const float* w, const float* b, float* y, int32_t n, int32_t in_f,
                           int32_t out_f, void* stream) {
  DSG_CHECK_ARG(x && w && y, "dsg_linear_fwd: NULL pointer");
  DSG_CHECK_ARG(n > 0 && in_f > 0 && out_f > 0, "dsg_linear_fwd: bad dims");
  hipLaunchKernelGGL(dsg::linear_rows_kernel, dim3(dsg::cdiv(out_f, 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, w, b, y, n, in_f, out_f);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}
