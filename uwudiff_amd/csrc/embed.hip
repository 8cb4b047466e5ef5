// Embedding / layout kernels: sinusoidal timestep features, SiLU, patchify / unpatchify, positional add.
// All HBM-bound elementwise; vector width 4 elements per lane.
#include "common.h"

namespace {

// [cos(t f_i) | sin(t f_i)], f_i = exp(-ln(max_period) * i / half)  (DiT TimestepEmbedder; diffusers
// Timesteps(flip_sin_to_cos=True, downscale_freq_shift=0))
template <typename T>
__global__ void timestep_embedding_kernel(const float* __restrict__ t, int B, int dim, float max_period,
                                          T* __restrict__ out) {
  const int half = dim >> 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, j = i - b * half;
  const float f = expf(-logf(max_period) * (float)j / (float)half);
  const float a = t[b] * f;
  out[(int64_t)b * dim + j] = from_f32<T>(cosf(a));
  out[(int64_t)b * dim + half + j] = from_f32<T>(sinf(a));
}

template <typename T>
__global__ void silu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 v = load4(x + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
    store4(y + 4 * i, v);
  }
}
template <typename T>
__global__ void silu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 v = load4(x + 4 * i), g = load4(dy + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] *= dsilu_f(v[e]);
    store4(dx + 4 * i, g);
  }
}

// token m = (b, th, tw); feature k = (c, ph, pw);  img[b][c][th*p+ph][tw*p+pw]
template <typename T, bool TO_TOKENS>
__global__ void patch_kernel(float* __restrict__ img, T* __restrict__ tok, int B, int C, int H, int W, int p) {
  const int gh = H / p, gw = W / p, kp = C * p * p;
  const int64_t total = (int64_t)B * gh * gw * kp;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    int64_t m = i / kp;
    int k = (int)(i - m * kp);
    int tw = (int)(m % gw);
    int64_t r = m / gw;
    int th = (int)(r % gh);
    int b = (int)(r / gh);
    int pw = k % p, ph = (k / p) % p, c = k / (p * p);
    int64_t src = (((int64_t)b * C + c) * H + (th * p + ph)) * W + (tw * p + pw);
    if (TO_TOKENS)
      tok[i] = from_f32<T>(img[src]);
    else
      img[src] = to_f32(tok[i]);
  }
}

template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    store4(o + 4 * i, load4(a + 4 * i) + load4(b + 4 * i));
}

template <typename T>
__global__ void add_pos_kernel(T* __restrict__ x, const float* __restrict__ pos, int64_t total4, int64_t td4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
    f32x4 v = load4(x + 4 * i), pv = load4(pos + 4 * (i % td4));
    store4(x + 4 * i, v + pv);
  }
}

}  // namespace

#define DISPATCH_T(dtype, NAME, ...)                                                    \
  if ((dtype) == UWU_F32) { NAME(float, __VA_ARGS__); }                                  \
  else if ((dtype) == UWU_BF16) { NAME(bf16_t, __VA_ARGS__); }                           \
  else { uwu_set_error("bad dtype %d", (int)(dtype)); return UWU_EINVAL; }

extern "C" int uwu_timestep_embedding(const float* t, int B, int dim, float max_period, void* out, int dtype,
                                      void* stream) {
  UWU_CHECK_ARG(t && out && B > 0 && dim > 0 && dim % 2 == 0, "timestep_embedding: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int n = B * dim / 2;
#define TE(T, ...) hipLaunchKernelGGL((timestep_embedding_kernel<T>), dim3(cdiv(n, 256)), dim3(256), 0, st, t, B, dim, max_period, (T*)out)
  DISPATCH_T(dtype, TE, 0)
#undef TE
  UWU_LAUNCH_CHECK("timestep_embedding");
  return UWU_OK;
}

extern "C" int uwu_silu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream) {
  UWU_CHECK_ARG(x && y && n > 0 && n % 4 == 0, "silu_fwd: n must be a positive multiple of 4");
  hipStream_t st = (hipStream_t)stream;
#define SF(T, ...) hipLaunchKernelGGL((silu_fwd_kernel<T>), dim3(ew_grid(n / 4, 256)), dim3(256), 0, st, (const T*)x, (T*)y, n / 4)
  DISPATCH_T(dtype, SF, 0)
#undef SF
  UWU_LAUNCH_CHECK("silu_fwd");
  return UWU_OK;
}

extern "C" int uwu_silu_bwd(const void* x, const void* dy, void* dx, int64_t n, int dtype, void* stream) {
  UWU_CHECK_ARG(x && dy && dx && n > 0 && n % 4 == 0, "silu_bwd: n must be a positive multiple of 4");
  hipStream_t st = (hipStream_t)stream;
#define SB(T, ...) hipLaunchKernelGGL((silu_bwd_kernel<T>), dim3(ew_grid(n / 4, 256)), dim3(256), 0, st, (const T*)x, (const T*)dy, (T*)dx, n / 4)
  DISPATCH_T(dtype, SB, 0)
#undef SB
  UWU_LAUNCH_CHECK("silu_bwd");
  return UWU_OK;
}

extern "C" int uwu_patchify(const float* img, void* tok, int B, int C, int H, int W, int p, int dtype, void* stream) {
  UWU_CHECK_ARG(img && tok && B > 0 && C > 0 && p > 0 && H % p == 0 && W % p == 0, "patchify: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)B * C * H * W;
#define PF(T, ...) hipLaunchKernelGGL((patch_kernel<T, true>), dim3(ew_grid(total, 256)), dim3(256), 0, st, const_cast<float*>(img), (T*)tok, B, C, H, W, p)
  DISPATCH_T(dtype, PF, 0)
#undef PF
  UWU_LAUNCH_CHECK("patchify");
  return UWU_OK;
}

extern "C" int uwu_unpatchify(const void* tok, int dtype, float* img, int B, int C, int H, int W, int p, void* stream) {
  UWU_CHECK_ARG(img && tok && B > 0 && C > 0 && p > 0 && H % p == 0 && W % p == 0, "unpatchify: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)B * C * H * W;
#define UF(T, ...) hipLaunchKernelGGL((patch_kernel<T, false>), dim3(ew_grid(total, 256)), dim3(256), 0, st, img, (T*)const_cast<void*>(tok), B, C, H, W, p)
  DISPATCH_T(dtype, UF, 0)
#undef UF
  UWU_LAUNCH_CHECK("unpatchify");
  return UWU_OK;
}

extern "C" int uwu_add_pos(void* x, const float* pos, int B, int T, int D, int dtype, void* stream) {
  UWU_CHECK_ARG(x && pos && B > 0 && T > 0 && D > 0 && D % 4 == 0, "add_pos: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int64_t td4 = (int64_t)T * D / 4, total4 = td4 * B;
#define AP(T_, ...) hipLaunchKernelGGL((add_pos_kernel<T_>), dim3(ew_grid(total4, 256)), dim3(256), 0, st, (T_*)x, pos, total4, td4)
  DISPATCH_T(dtype, AP, 0)
#undef AP
  UWU_LAUNCH_CHECK("add_pos");
  return UWU_OK;
}

extern "C" int uwu_add(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream) {
  UWU_CHECK_ARG(a && b && out && n > 0 && n % 4 == 0, "add: n must be a positive multiple of 4");
  hipStream_t st = (hipStream_t)stream;
#define AD(T, ...) hipLaunchKernelGGL((add_kernel<T>), dim3(ew_grid(n / 4, 256)), dim3(256), 0, st, (const T*)a, (const T*)b, (T*)out, n / 4)
  DISPATCH_T(dtype, AD, 0)
#undef AD
  UWU_LAUNCH_CHECK("add");
  return UWU_OK;
}

// dst[c][r] = src[r][c] for a bf16 matrix (weights only: the input-gradient GEMM of a store-heavy Linear wants its weight
// contraction-contiguous, csrc/gemm.hip gemm_as_kernel).  32 x 32 tiles through LDS, coalesced on both sides.
namespace {
__global__ void __launch_bounds__(256) transpose_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int rows,
                                                             int cols, int lds, int ldd) {
  __shared__ bf16_t tile[32][34];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (r < rows && c < cols) ? src[(int64_t)r * lds + c] : (bf16_t)0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (c < cols && r < rows) dst[(int64_t)c * ldd + r] = tile[tx][ty + 8 * k];
  }
}
}  // namespace

extern "C" int uwu_transpose_bf16(const void* src, void* dst, int rows, int cols, int ld_src, int ld_dst, void* stream) {
  UWU_CHECK_ARG(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows, "transpose_bf16: bad argument");
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src, (bf16_t*)dst, rows, cols, ld_src, ld_dst);
  UWU_LAUNCH_CHECK("transpose_bf16");
  return UWU_OK;
}
