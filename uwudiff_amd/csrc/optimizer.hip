// Flat-buffer optimizer kernels: global grad norm (+clip coefficient), fused AdamW with bf16 shadow
// refresh, dtype casts.  HBM-bound: AdamW moves 28 B/param (+2 B/param for the bf16 shadow).
// Reference: src/duwu/trainer/trainer.py:52-74 (torch.optim.AdamW + Lightning gradient_clip_val).
#include "common.h"

// stage 1: per-block partial sums (deterministic order), stage 2: one block folds the partials
__global__ void __launch_bounds__(256) sqnorm_partial_kernel(const float* __restrict__ g, int64_t n,
                                                             float* __restrict__ partial) {
  __shared__ float red[4];
  const int64_t n4 = n >> 2;
  float a = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 v = load4(g + 4 * i);
    a += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0) {  // tail
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) a += g[i] * g[i];
  }
  float t = block_sum<4>(a, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ void __launch_bounds__(256) sqnorm_final_kernel(const float* __restrict__ partial, int np,
                                                           float pre_scale, float max_norm,
                                                           float* __restrict__ out) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < np; i += 256) a += partial[i];
  float t = block_sum<4>(a, red);
  if (threadIdx.x == 0) {
    float sq = t * pre_scale * pre_scale;
    out[0] = sq;
    float coef = 1.f;
    if (max_norm > 0.f) {
      // torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
      coef = fminf(max_norm / (sqrtf(sq) + 1e-6f), 1.f);
    }
    out[1] = coef;
  }
}

// torch.optim.AdamW (single tensor, no amsgrad):
//   p *= 1 - lr*wd ; m = lerp(m, g, 1-b1) ; v = b2*v + (1-b2) g^2 ;
//   p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// zero_grad: the consumed gradient is overwritten with zeros in the same pass (the next backward accumulates into the flat
// buffer: 4 more bytes written per parameter instead of a separate fill kernel that writes 4 and a launch)
__global__ void __launch_bounds__(256) adamw_kernel(float* __restrict__ p, float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    bf16_t* __restrict__ pbf, int64_t n, float lr, float b1,
                                                    float b2, float eps, float wd, float step_size,
                                                    float inv_bc2_sqrt, float pre_scale,
                                                    const float* __restrict__ clip, int zero_grad) {
  const float gscale = pre_scale * (clip ? clip[1] : 1.f);
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const float decay = 1.f - lr * wd;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 pv = load4(p + 4 * i), gv = load4(g + 4 * i), mv = load4(m + 4 * i), vv = load4(v + 4 * i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float gg = gv[j] * gscale;
      float pp = pv[j] * decay;
      float mm = mv[j] + (gg - mv[j]) * (1.f - b1);
      float vn = vv[j] * b2 + (1.f - b2) * gg * gg;
      float denom = sqrtf(vn) * inv_bc2_sqrt + eps;
      pp = pp - step_size * (mm / denom);
      pv[j] = pp;
      mv[j] = mm;
      vv[j] = vn;
    }
    store4(p + 4 * i, pv);
    store4(m + 4 * i, mv);
    store4(v + 4 * i, vv);
    if (pbf) store4(pbf + 4 * i, pv);
    if (zero_grad) store4(g + 4 * i, f32x4{0.f, 0.f, 0.f, 0.f});
  }
  if (blockIdx.x == 0) {
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
      float gg = g[i] * gscale;
      float pp = p[i] * decay;
      float mm = m[i] + (gg - m[i]) * (1.f - b1);
      float vn = v[i] * b2 + (1.f - b2) * gg * gg;
      pp = pp - step_size * (mm / (sqrtf(vn) * inv_bc2_sqrt + eps));
      p[i] = pp;
      m[i] = mm;
      v[i] = vn;
      if (pbf) pbf[i] = (bf16_t)pp;
      if (zero_grad) g[i] = 0.f;
    }
  }
}

template <typename TS, typename TD>
__global__ void __launch_bounds__(256) cast_kernel(const TS* __restrict__ s, TD* __restrict__ d, int64_t n) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) store4(d + 4 * i, load4(s + 4 * i));
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) d[i] = from_f32<TD>(to_f32(s[i]));
}


extern "C" int uwu_grad_sqnorm_clip(const float* g, int64_t n, float pre_scale, float max_norm, float* partial,
                                    float* out, void* stream) {
  UWU_CHECK_ARG(g && partial && out && n > 0, "grad_sqnorm_clip: bad args");
  UWU_CHECK_ARG(((uintptr_t)g & 15) == 0, "grad_sqnorm_clip: g must be 16-byte aligned");
  int grid = ew_grid(n / 4, 256);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, n, partial);
  UWU_LAUNCH_CHECK("sqnorm_partial");
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, grid, pre_scale,
                     max_norm, out);
  UWU_LAUNCH_CHECK("sqnorm_final");
  return UWU_OK;
}

extern "C" int uwu_adamw_step(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int step, float pre_scale,
                              const float* clip, int zero_grad, void* stream) {
  UWU_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adamw_step: bad args");
  UWU_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
                "adamw_step: buffers must be 16-byte aligned");
  UWU_CHECK_ARG(p_bf16 == nullptr || ((uintptr_t)p_bf16 & 7) == 0, "adamw_step: bf16 shadow must be 8-byte aligned");
  // bias corrections in double on the host, as torch does with python floats
  double bc1 = 1.0 - pow((double)beta1, (double)step);
  double bc2 = 1.0 - pow((double)beta2, (double)step);
  float step_size = (float)((double)lr / bc1);
  float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     (bf16_t*)p_bf16, n, lr, beta1, beta2, eps, weight_decay, step_size, inv_bc2_sqrt, pre_scale,
                     clip, zero_grad);
  UWU_LAUNCH_CHECK("adamw_step");
  return UWU_OK;
}

extern "C" int uwu_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
  UWU_CHECK_ARG(src && dst && n > 0, "cast_f32_to_bf16: bad args");
  hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(ew_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (bf16_t*)dst, n);
  UWU_LAUNCH_CHECK("cast_f32_to_bf16");
  return UWU_OK;
}

extern "C" int uwu_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream) {
  UWU_CHECK_ARG(src && dst && n > 0, "cast_bf16_to_f32: bad args");
  hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(ew_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src, dst, n);
  UWU_LAUNCH_CHECK("cast_bf16_to_f32");
  return UWU_OK;
}
