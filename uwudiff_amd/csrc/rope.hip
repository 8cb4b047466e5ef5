// Axial rotary position embedding with learnable per-head log-frequencies.
// Reference: src/duwu/modules/rope.py:56-71 (rotate_half / apply_rotary_emb), :83-108 (AxialRoPE.get_freqs/forward),
// used on q and k by RoPEAttnProcessor2_0 (rope_unet.py:143-147).  Restated exactly as written there:
//   theta[b,t,h,:] = repeat_interleave( cat(pos_h * exp(fh[h,:]), pos_w * exp(fw[h,:])), 2 )        (d values)
//   rotate_half(x) = (-x[0], x[1], -x[2], x[3], ...)      <- the reference negates the EVEN element of each pair
//   y = x * cos(theta) + rotate_half(x) * sin(theta)   =>  y[2i] = x[2i](cos - sin), y[2i+1] = x[2i+1](cos + sin)
// x: token-major [B*T, H*d]; pos: fp32 [B*T, 2] (h, w); fh, fw: fp32 [H, d/4].  HBM-bound elementwise.
#include "common.h"

namespace {

template <typename T, bool BWD>
__global__ void __launch_bounds__(256) axial_rope_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                         const float* __restrict__ pos, const float* __restrict__ fh,
                                                         const float* __restrict__ fw, T* __restrict__ out,
                                                         float* __restrict__ dfh, float* __restrict__ dfw,
                                                         int64_t rows, int H, int d, int ldx, int pos_rows) {
  // one thread = one (row, head, pair); pairs i in [0, d/2): i < d/4 -> h-axis frequency i, else w-axis i - d/4
  const int hp = d / 2, q4 = d / 4;
  const int64_t total = rows * H * hp, step = (int64_t)gridDim.x * 256;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += step) {
    const int i = (int)(idx % hp);
    const int64_t r2 = idx / hp;
    const int h = (int)(r2 % H);
    const int64_t row = r2 / H;
    const bool wax = i >= q4;
    const int fi = wax ? i - q4 : i;
    const float p = pos[2 * (pos_rows > 0 ? row % pos_rows : row) + (wax ? 1 : 0)];  // pos_rows: positions shared by the batch
    const float ef = __expf((wax ? fw : fh)[h * q4 + fi]);
    const float th = p * ef;
    float sn, cs;
    __sincosf(th, &sn, &cs);
    const int64_t o = row * ldx + (int64_t)h * d + 2 * i;
    if (!BWD) {
      const float a = to_f32(x[o]), b = to_f32(x[o + 1]);
      out[o] = from_f32<T>(a * (cs - sn));
      out[o + 1] = from_f32<T>(b * (cs + sn));
    } else {
      const float ga = to_f32(dy[o]), gb = to_f32(dy[o + 1]);
      out[o] = from_f32<T>(ga * (cs - sn));
      out[o + 1] = from_f32<T>(gb * (cs + sn));
      if (dfh) {  // d/d log-freq = d/dtheta * theta
        const float a = to_f32(x[o]), b = to_f32(x[o + 1]);
        const float dth = ga * a * (-sn - cs) + gb * b * (cs - sn);
        atomicAdd((wax ? dfw : dfh) + h * q4 + fi, dth * th);
      }
    }
  }
}

// m[t, h*d + e]: the factor y = x * m of the forward above, for shared positions pos[T, 2] -- the table the attention
// kernels multiply q / k by while staging them (uwu_attention_rope_fwd / _bwd)
__global__ void __launch_bounds__(256) axial_rope_table_kernel(const float* __restrict__ pos, const float* __restrict__ fh,
                                                               const float* __restrict__ fw, float* __restrict__ tab, int T,
                                                               int H, int d, int ldt) {
  const int hp = d / 2, q4 = d / 4;
  const int total = T * H * hp;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int i = idx % hp, r2 = idx / hp, h = r2 % H, t = r2 / H;
    const bool wax = i >= q4;
    const int fi = wax ? i - q4 : i;
    const float th = pos[2 * t + (wax ? 1 : 0)] * __expf((wax ? fw : fh)[h * q4 + fi]);
    float sn, cs;
    __sincosf(th, &sn, &cs);
    float* o = tab + (int64_t)t * ldt + h * d + 2 * i;
    o[0] = cs - sn;
    o[1] = cs + sn;
  }
}

}  // namespace

extern "C" int uwu_axial_rope_table(const float* pos, const float* fh, const float* fw, float* tab, int T, int H, int d,
                                    int ldt, void* stream) {
  UWU_CHECK_ARG(pos && fh && fw && tab && T > 0 && H > 0 && d > 0 && d % 4 == 0 && ldt >= H * d, "axial_rope_table: bad args");
  hipLaunchKernelGGL(axial_rope_table_kernel, dim3(ew_grid((int64_t)T * H * (d / 2), 256)), dim3(256), 0, (hipStream_t)stream,
                     pos, fh, fw, tab, T, H, d, ldt);
  UWU_LAUNCH_CHECK("axial_rope_table");
  return UWU_OK;
}

extern "C" int uwu_axial_rope_fwd(const void* x, const float* pos, const float* fh, const float* fw, void* y,
                                  int64_t rows, int H, int d, int ldx, int dtype, void* stream) {
  UWU_CHECK_ARG(x && pos && fh && fw && y && rows > 0 && H > 0 && d > 0 && d % 4 == 0 && ldx >= H * d,
                "axial_rope_fwd: bad args (d=%d)", d);
  hipStream_t st = (hipStream_t)stream;
  const int grid = ew_grid(rows * H * (d / 2), 256);
  if (dtype == UWU_F32)
    hipLaunchKernelGGL((axial_rope_kernel<float, false>), dim3(grid), dim3(256), 0, st, (const float*)x, nullptr, pos, fh,
                       fw, (float*)y, nullptr, nullptr, rows, H, d, ldx, 0);
  else if (dtype == UWU_BF16)
    hipLaunchKernelGGL((axial_rope_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, nullptr, pos,
                       fh, fw, (bf16_t*)y, nullptr, nullptr, rows, H, d, ldx, 0);
  else
    UWU_CHECK_ARG(false, "axial_rope_fwd: bad dtype");
  UWU_LAUNCH_CHECK("axial_rope_fwd");
  return UWU_OK;
}

static int rope_bwd_impl(const void* x, const void* dy, const float* pos, const float* fh, const float* fw, void* dx,
                         float* dfh, float* dfw, int64_t rows, int H, int d, int ldx, int dtype, int pos_rows, void* stream);
extern "C" int uwu_axial_rope_bwd(const void* x, const void* dy, const float* pos, const float* fh, const float* fw,
                                  void* dx, float* dfh, float* dfw, int64_t rows, int H, int d, int ldx, int dtype,
                                  void* stream) {
  return rope_bwd_impl(x, dy, pos, fh, fw, dx, dfh, dfw, rows, H, d, ldx, dtype, 0, stream);
}
// the same with positions shared by the batch: pos [pos_rows, 2], row r uses pos[r % pos_rows]
extern "C" int uwu_axial_rope_bwd_shared(const void* x, const void* dy, const float* pos, int pos_rows, const float* fh,
                                         const float* fw, void* dx, float* dfh, float* dfw, int64_t rows, int H, int d,
                                         int ldx, int dtype, void* stream) {
  UWU_CHECK_ARG(pos_rows > 0, "axial_rope_bwd_shared: pos_rows must be positive");
  return rope_bwd_impl(x, dy, pos, fh, fw, dx, dfh, dfw, rows, H, d, ldx, dtype, pos_rows, stream);
}
static int rope_bwd_impl(const void* x, const void* dy, const float* pos, const float* fh, const float* fw, void* dx,
                         float* dfh, float* dfw, int64_t rows, int H, int d, int ldx, int dtype, int pos_rows, void* stream) {
  UWU_CHECK_ARG(x && dy && pos && fh && fw && dx && rows > 0 && H > 0 && d > 0 && d % 4 == 0 && ldx >= H * d,
                "axial_rope_bwd: bad args");
  UWU_CHECK_ARG((dfh == nullptr) == (dfw == nullptr), "axial_rope_bwd: dfh/dfw go together");
  hipStream_t st = (hipStream_t)stream;
  const int grid = ew_grid(rows * H * (d / 2), 256);
  if (dtype == UWU_F32)
    hipLaunchKernelGGL((axial_rope_kernel<float, true>), dim3(grid), dim3(256), 0, st, (const float*)x, (const float*)dy,
                       pos, fh, fw, (float*)dx, dfh, dfw, rows, H, d, ldx, pos_rows);
  else if (dtype == UWU_BF16)
    hipLaunchKernelGGL((axial_rope_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x,
                       (const bf16_t*)dy, pos, fh, fw, (bf16_t*)dx, dfh, dfw, rows, H, d, ldx, pos_rows);
  else
    UWU_CHECK_ARG(false, "axial_rope_bwd: bad dtype");
  UWU_LAUNCH_CHECK("axial_rope_bwd");
  return UWU_OK;
}
