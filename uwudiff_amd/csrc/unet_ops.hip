// Kernels specific to the UNet2DConditionModel-shape denoiser (reference src/duwu/modules/unet_patch.py:13-57 ==
// diffusers.UNet2DConditionModel; restated block semantics in SURVEY.md section 8 row a11).  Activations are
// channels-last / token-major: x[b, p, c] with p = y*W + x, so Linear / attention consume them without transposes
// and a 3x3 convolution is im2col + the MFMA GEMM (K = 9*C).  Everything here is HBM-bound.
#include "common.h"

namespace {

// (GroupNorm lives in groupnorm.hip)
// ---- im2col / col2im for 3x3, padding 1, stride 1 or 2 (channels-last) ----------------------------------
// col[(b, oy, ox), (ky, kx, c)] = x[b, oy*s + ky - 1, ox*s + kx - 1, c]   (zero outside)
template <typename T>
__global__ void __launch_bounds__(256) im2col3x3_kernel(const T* __restrict__ x, T* __restrict__ col, int B, int H,
                                                        int W, int C, int Ho, int Wo, int stride) {
  const int c4n = C / 4;
  const int64_t total = (int64_t)B * Ho * Wo * 9 * c4n;
  const int64_t step = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
    const int c4 = (int)(i % c4n);
    int64_t r = i / c4n;
    const int tap = (int)(r % 9);
    r /= 9;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const int iy = oy * stride + tap / 3 - 1, ix = ox * stride + tap % 3 - 1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = load4(x + (((int64_t)b * H + iy) * W + ix) * C + 4 * c4);
    store4(col + i * 4, v);
  }
}
// dx[b, iy, ix, c] = sum over taps of dcol[(b, oy, ox), tap, c] with oy*s + ky - 1 == iy (gather form, no atomics)
template <typename T>
__global__ void __launch_bounds__(256) col2im3x3_kernel(const T* __restrict__ dcol, T* __restrict__ dx, int B, int H,
                                                        int W, int C, int Ho, int Wo, int stride) {
  const int c4n = C / 4;
  const int64_t total = (int64_t)B * H * W * c4n;
  const int64_t step = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
    const int c4 = (int)(i % c4n);
    int64_t r = i / c4n;
    const int ix = (int)(r % W);
    r /= W;
    const int iy = (int)(r % H);
    const int b = (int)(r / H);
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ty = iy + 1 - tap / 3, tx = ix + 1 - tap % 3;  // = oy*stride, ox*stride
      if (ty < 0 || tx < 0 || ty % stride || tx % stride) continue;
      const int oy = ty / stride, ox = tx / stride;
      if (oy >= Ho || ox >= Wo) continue;
      a = a + load4(dcol + ((((int64_t)b * Ho + oy) * Wo + ox) * 9 + tap) * C + 4 * c4);
    }
    store4(dx + i * 4, a);
  }
}

// ---- GEGLU: h[m, :F] , gate[m, F:2F] -> out = h * gelu_erf(gate)   (diffusers GEGLU uses exact GELU) ------
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_erf_f(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
template <typename T>
__global__ void __launch_bounds__(256) geglu_fwd_kernel(const T* __restrict__ hg, T* __restrict__ out, int64_t M,
                                                        int F) {
  const int f4n = F / 4;
  const int64_t total = M * f4n, step = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
    const int64_t m = i / f4n;
    const int f = (int)(i - m * f4n) * 4;
    f32x4 h = load4(hg + m * 2 * F + f), gt = load4(hg + m * 2 * F + F + f), o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = h[e] * gelu_erf_f(gt[e]);
    store4(out + m * F + f, o);
  }
}
template <typename T>
__global__ void __launch_bounds__(256) geglu_bwd_kernel(const T* __restrict__ hg, const T* __restrict__ dout,
                                                        T* __restrict__ dhg, int64_t M, int F) {
  const int f4n = F / 4;
  const int64_t total = M * f4n, step = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
    const int64_t m = i / f4n;
    const int f = (int)(i - m * f4n) * 4;
    f32x4 h = load4(hg + m * 2 * F + f), gt = load4(hg + m * 2 * F + F + f), d = load4(dout + m * F + f), dh, dg;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dh[e] = d[e] * gelu_erf_f(gt[e]);
      dg[e] = d[e] * h[e] * dgelu_erf_f(gt[e]);
    }
    store4(dhg + m * 2 * F + f, dh);
    store4(dhg + m * 2 * F + F + f, dg);
  }
}

// ---- nearest 2x upsample (channels-last) and its adjoint (sum of the 4 children) ---------------------------
template <typename T, bool FWD>
__global__ void __launch_bounds__(256) upsample2x_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int H,
                                                         int W, int C) {
  const int c4n = C / 4;
  if (FWD) {  // src [B,H,W,C] -> dst [B,2H,2W,C]
    const int64_t total = (int64_t)B * 2 * H * 2 * W * c4n, step = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
      const int c4 = (int)(i % c4n);
      int64_t r = i / c4n;
      const int ox = (int)(r % (2 * W));
      r /= 2 * W;
      const int oy = (int)(r % (2 * H));
      const int b = (int)(r / (2 * H));
      store4(dst + i * 4, load4(src + (((int64_t)b * H + oy / 2) * W + ox / 2) * C + 4 * c4));
    }
  } else {  // src = d(out) [B,2H,2W,C] -> dst = d(in) [B,H,W,C]
    const int64_t total = (int64_t)B * H * W * c4n, step = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
      const int c4 = (int)(i % c4n);
      int64_t r = i / c4n;
      const int x = (int)(r % W);
      r /= W;
      const int y = (int)(r % H);
      const int b = (int)(r / H);
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dxx = 0; dxx < 2; ++dxx)
          a = a + load4(src + (((int64_t)b * 2 * H + 2 * y + dy) * 2 * W + 2 * x + dxx) * C + 4 * c4);
      store4(dst + i * 4, a);
    }
  }
}

// x[b, p, :] += v[b, :]  (time-embedding injection of a ResnetBlock2D)
template <typename T>
__global__ void __launch_bounds__(256) add_rowvec_kernel(T* __restrict__ x, const T* __restrict__ v, int64_t total4,
                                                         int64_t per_sample4, int c4n) {
  const int64_t step = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += step) {
    const int64_t b = i / per_sample4;
    const int c4 = (int)(i % c4n);
    store4(x + 4 * i, load4(x + 4 * i) + load4(v + (b * c4n + c4) * 4));
  }
}

// NCHW fp32 <-> channels-last T
template <typename T, bool TO_CL>
__global__ void __launch_bounds__(256) layout_kernel(float* __restrict__ nchw, T* __restrict__ cl, int B, int C, int HW) {
  const int64_t total = (int64_t)B * C * HW, step = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int p = (int)(r % HW);
    const int b = (int)(r / HW);
    const int64_t j = ((int64_t)b * C + c) * HW + p;
    if (TO_CL) cl[i] = from_f32<T>(nchw[j]);
    else nchw[j] = to_f32(cl[i]);
  }
}

}  // namespace

#define UNET_DISPATCH(dtype, EXPR_F32, EXPR_BF16)                   \
  if ((dtype) == UWU_F32) { EXPR_F32; }                              \
  else if ((dtype) == UWU_BF16) { EXPR_BF16; }                       \
  else { uwu_set_error("bad dtype %d", (int)(dtype)); return UWU_EINVAL; }

extern "C" int uwu_im2col3x3(const void* x, void* col, int B, int H, int W, int C, int stride, int dtype, void* stream) {
  UWU_CHECK_ARG(x && col && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && (stride == 1 || stride == 2),
                "im2col3x3: bad args (C=%d stride=%d)", C, stride);
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  const int64_t total = (int64_t)B * Ho * Wo * 9 * (C / 4);
  hipStream_t st = (hipStream_t)stream;
  UNET_DISPATCH(dtype,
                hipLaunchKernelGGL((im2col3x3_kernel<float>), dim3(ew_grid(total, 256)), dim3(256), 0, st,
                                   (const float*)x, (float*)col, B, H, W, C, Ho, Wo, stride),
                hipLaunchKernelGGL((im2col3x3_kernel<bf16_t>), dim3(ew_grid(total, 256)), dim3(256), 0, st,
                                   (const bf16_t*)x, (bf16_t*)col, B, H, W, C, Ho, Wo, stride))
  UWU_LAUNCH_CHECK("im2col3x3");
  return UWU_OK;
}

extern "C" int uwu_col2im3x3(const void* dcol, void* dx, int B, int H, int W, int C, int stride, int dtype, void* stream) {
  UWU_CHECK_ARG(dcol && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && (stride == 1 || stride == 2),
                "col2im3x3: bad args");
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  const int64_t total = (int64_t)B * H * W * (C / 4);
  hipStream_t st = (hipStream_t)stream;
  UNET_DISPATCH(dtype,
                hipLaunchKernelGGL((col2im3x3_kernel<float>), dim3(ew_grid(total, 256)), dim3(256), 0, st,
                                   (const float*)dcol, (float*)dx, B, H, W, C, Ho, Wo, stride),
                hipLaunchKernelGGL((col2im3x3_kernel<bf16_t>), dim3(ew_grid(total, 256)), dim3(256), 0, st,
                                   (const bf16_t*)dcol, (bf16_t*)dx, B, H, W, C, Ho, Wo, stride))
  UWU_LAUNCH_CHECK("col2im3x3");
  return UWU_OK;
}

extern "C" int uwu_geglu_fwd(const void* hg, void* out, int64_t M, int F, int dtype, void* stream) {
  UWU_CHECK_ARG(hg && out && M > 0 && F > 0 && F % 4 == 0, "geglu_fwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  UNET_DISPATCH(dtype,
                hipLaunchKernelGGL((geglu_fwd_kernel<float>), dim3(ew_grid(M * F / 4, 256)), dim3(256), 0, st,
                                   (const float*)hg, (float*)out, M, F),
                hipLaunchKernelGGL((geglu_fwd_kernel<bf16_t>), dim3(ew_grid(M * F / 4, 256)), dim3(256), 0, st,
                                   (const bf16_t*)hg, (bf16_t*)out, M, F))
  UWU_LAUNCH_CHECK("geglu_fwd");
  return UWU_OK;
}

extern "C" int uwu_geglu_bwd(const void* hg, const void* dout, void* dhg, int64_t M, int F, int dtype, void* stream) {
  UWU_CHECK_ARG(hg && dout && dhg && M > 0 && F > 0 && F % 4 == 0, "geglu_bwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  UNET_DISPATCH(dtype,
                hipLaunchKernelGGL((geglu_bwd_kernel<float>), dim3(ew_grid(M * F / 4, 256)), dim3(256), 0, st,
                                   (const float*)hg, (const float*)dout, (float*)dhg, M, F),
                hipLaunchKernelGGL((geglu_bwd_kernel<bf16_t>), dim3(ew_grid(M * F / 4, 256)), dim3(256), 0, st,
                                   (const bf16_t*)hg, (const bf16_t*)dout, (bf16_t*)dhg, M, F))
  UWU_LAUNCH_CHECK("geglu_bwd");
  return UWU_OK;
}

extern "C" int uwu_upsample2x(const void* src, void* dst, int B, int H, int W, int C, int backward, int dtype,
                              void* stream) {
  UWU_CHECK_ARG(src && dst && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "upsample2x: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)B * H * W * (C / 4) * (backward ? 1 : 4);
  const int grid = ew_grid(total, 256);
  if (!backward) {
    UNET_DISPATCH(dtype,
                  hipLaunchKernelGGL((upsample2x_kernel<float, true>), dim3(grid), dim3(256), 0, st, (const float*)src,
                                     (float*)dst, B, H, W, C),
                  hipLaunchKernelGGL((upsample2x_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, st,
                                     (const bf16_t*)src, (bf16_t*)dst, B, H, W, C))
  } else {
    UNET_DISPATCH(dtype,
                  hipLaunchKernelGGL((upsample2x_kernel<float, false>), dim3(grid), dim3(256), 0, st, (const float*)src,
                                     (float*)dst, B, H, W, C),
                  hipLaunchKernelGGL((upsample2x_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, st,
                                     (const bf16_t*)src, (bf16_t*)dst, B, H, W, C))
  }
  UWU_LAUNCH_CHECK("upsample2x");
  return UWU_OK;
}

extern "C" int uwu_add_rowvec(void* x, const void* v, int B, int HW, int C, int dtype, void* stream) {
  UWU_CHECK_ARG(x && v && B > 0 && HW > 0 && C > 0 && C % 4 == 0, "add_rowvec: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int64_t per4 = (int64_t)HW * C / 4, total4 = per4 * B;
  UNET_DISPATCH(dtype,
                hipLaunchKernelGGL((add_rowvec_kernel<float>), dim3(ew_grid(total4, 256)), dim3(256), 0, st, (float*)x,
                                   (const float*)v, total4, per4, C / 4),
                hipLaunchKernelGGL((add_rowvec_kernel<bf16_t>), dim3(ew_grid(total4, 256)), dim3(256), 0, st,
                                   (bf16_t*)x, (const bf16_t*)v, total4, per4, C / 4))
  UWU_LAUNCH_CHECK("add_rowvec");
  return UWU_OK;
}

extern "C" int uwu_nchw_to_cl(const float* nchw, void* cl, int B, int C, int HW, int dtype, void* stream) {
  UWU_CHECK_ARG(nchw && cl && B > 0 && C > 0 && HW > 0, "nchw_to_cl: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)B * C * HW;
  UNET_DISPATCH(dtype,
                hipLaunchKernelGGL((layout_kernel<float, true>), dim3(ew_grid(total, 256)), dim3(256), 0, st,
                                   const_cast<float*>(nchw), (float*)cl, B, C, HW),
                hipLaunchKernelGGL((layout_kernel<bf16_t, true>), dim3(ew_grid(total, 256)), dim3(256), 0, st,
                                   const_cast<float*>(nchw), (bf16_t*)cl, B, C, HW))
  UWU_LAUNCH_CHECK("nchw_to_cl");
  return UWU_OK;
}

extern "C" int uwu_cl_to_nchw(const void* cl, float* nchw, int B, int C, int HW, int dtype, void* stream) {
  UWU_CHECK_ARG(nchw && cl && B > 0 && C > 0 && HW > 0, "cl_to_nchw: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)B * C * HW;
  UNET_DISPATCH(dtype,
                hipLaunchKernelGGL((layout_kernel<float, false>), dim3(ew_grid(total, 256)), dim3(256), 0, st, nchw,
                                   (float*)const_cast<void*>(cl), B, C, HW),
                hipLaunchKernelGGL((layout_kernel<bf16_t, false>), dim3(ew_grid(total, 256)), dim3(256), 0, st, nchw,
                                   (bf16_t*)const_cast<void*>(cl), B, C, HW))
  UWU_LAUNCH_CHECK("cl_to_nchw");
  return UWU_OK;
}
