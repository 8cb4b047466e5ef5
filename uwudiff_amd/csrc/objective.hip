// Objective kernels: schedule gather, RF time sampling, q-sample, fused loss fwd+bwd.
// HBM-bound elementwise + wavefront-shuffle reductions.  Reference: src/duwu/loss/diffusion.py,
// src/duwu/loss/rectified_flow.py (line cites at each kernel).
#include "common.h"

// ---------------------------------------------------------------------------------------
// diffusion.py:53-62 (sigma lookup), :141-153 (min-SNR), :155-167 (debias)
__global__ void schedule_gather_kernel(const int64_t* __restrict__ t, const float* __restrict__ sigmas_desc,
                                       const float* __restrict__ all_snr, const float* __restrict__ abar,
                                       int n_train, int B, int snr_mode, float gamma, int debias,
                                       float* __restrict__ coef) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int64_t ti = t[b];
  ti = ti < 0 ? 0 : (ti >= n_train ? n_train - 1 : ti);
  // position of t in the descending scheduler.timesteps is N-1-t
  float sigma = sigmas_desc[n_train - 1 - ti];
  float w = 1.f;
  float snr = all_snr[ti];
  if (snr_mode != 0) {
    float m = fminf(snr, gamma);
    w *= (snr_mode == 2) ? m / (snr + 1.f) : m / snr;
  }
  if (debias) {
    float s = fminf(snr, 1000.f);
    w *= 1.f / sqrtf(s);
  }
  float a = abar[ti];
  coef[4 * b + 0] = sigma;
  coef[4 * b + 1] = w;
  coef[4 * b + 2] = sqrtf(a);
  coef[4 * b + 3] = sqrtf(1.f - a);
}

// rectified_flow.py:29-42 and :98-129
__global__ void rf_time_kernel(const float* __restrict__ u01, float smax, const float* __restrict__ tbl,
                               int n, int B, float* __restrict__ coef, float* __restrict__ tout) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float max_time = smax / (1.f + smax);
  float time = u01[b] * max_time;
  float sigma = time / (1.f - time);
  float ls = logf(fmaxf(sigma, 1e-10f));
  // low_idx = argmax(cumsum(ls >= tbl)) = (#entries with tbl <= ls) - 1 for an ascending table,
  // (0 when none), clamped to n-2.  Table is ascending so count by binary search.
  int lo = 0, hi = n;  // first index with tbl[idx] > ls
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (tbl[mid] <= ls) lo = mid + 1; else hi = mid;
  }
  int low_idx = lo - 1;
  if (low_idx < 0) low_idx = 0;
  if (low_idx > n - 2) low_idx = n - 2;
  int high_idx = low_idx + 1;
  float low = tbl[low_idx], high = tbl[high_idx];
  float w = (low - ls) / (low - high);
  w = fminf(fmaxf(w, 0.f), 1.f);
  tout[b] = (1.f - w) * (float)low_idx + w * (float)high_idx;
  coef[4 * b + 0] = sigma;
  coef[4 * b + 1] = 1.f;
  coef[4 * b + 2] = 0.f;
  coef[4 * b + 3] = 0.f;
}

// ---- in-kernel draws (diffusion.py:68-76: noise = randn_like(x), then t = randint(0, N, [B]); rectified_flow.py:37: rand(B)) ------
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11): counter-based, so a draw needs
// no state -- element group i of a tensor is philox(counter = offset + i, key = seed).  seed / offset come from the caller (the
// host mirror takes them from torch's CUDA generator and advances its offset, so torch.manual_seed() reproduces a run); the
// reference's ORDER of draws (noise before timesteps) is the order in which the host reserves offset ranges.  oracle/philox.py
// restates the generator and the transforms; tests/test_objective_gpu.py compares bit for bit / to fp32 rounding.
struct PhiloxOut { unsigned v[4]; };
__device__ __forceinline__ PhiloxOut philox4x32_10(unsigned long long counter, unsigned long long seed) {
  unsigned c0 = (unsigned)counter, c1 = (unsigned)(counter >> 32), c2 = 0u, c3 = 0u;
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return PhiloxOut{{c0, c1, c2, c3}};
}
// 24 random bits -> (0, 1): (r >> 8 + 1/2) / 2^24, exact in fp32
__device__ __forceinline__ float philox_u01(unsigned r) { return ((float)(r >> 8) + 0.5f) * 5.9604644775390625e-08f; }
// four N(0, 1) values from one counter: Box-Muller on the pairs (v0, v1) and (v2, v3)
__device__ __forceinline__ f32x4 philox_normal4(unsigned long long counter, unsigned long long seed) {
  const PhiloxOut o = philox4x32_10(counter, seed);
  f32x4 z;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float rad = sqrtf(-2.f * __logf(philox_u01(o.v[2 * h])));
    float sn, cs;
    __sincosf(6.283185307179586f * philox_u01(o.v[2 * h + 1]), &sn, &cs);
    z[2 * h] = rad * cs;
    z[2 * h + 1] = rad * sn;
  }
  return z;
}
__global__ void philox_raw_kernel(unsigned* __restrict__ out, int64_t n, unsigned long long seed, unsigned long long offset) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const PhiloxOut o = philox4x32_10(offset + (unsigned long long)i, seed);
#pragma unroll
  for (int j = 0; j < 4; ++j) out[4 * i + j] = o.v[j];
}
__global__ void philox_normal_kernel(float* __restrict__ out, int64_t total4, unsigned long long seed, unsigned long long offset) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < total4; i += stride) store4(out + 4 * i, philox_normal4(offset + (unsigned long long)i, seed));
}
// t[b] ~ U{0 .. n_train-1} (value j of counter offset + b / 4: (r * n_train) >> 32), then the gather of schedule_gather_kernel
__global__ void schedule_draw_kernel(int64_t* __restrict__ t, int n_train, int B, unsigned long long seed,
                                     unsigned long long offset) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const PhiloxOut o = philox4x32_10(offset + (unsigned long long)(b >> 2), seed);
  t[b] = (int64_t)(((unsigned long long)o.v[b & 3] * (unsigned long long)n_train) >> 32);
}
__global__ void u01_draw_kernel(float* __restrict__ u, int B, unsigned long long seed, unsigned long long offset) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const PhiloxOut o = philox4x32_10(offset + (unsigned long long)(b >> 2), seed);
  u[b] = philox_u01(o.v[b & 3]);
}
// q-sample with the noise drawn in the kernel: reads x, writes noise (the loss needs it again), noisy and -- with the VAE latent
// normalisation -- x_norm: one pass instead of randn_like + q-sample
__global__ void qsample_draw_kernel(const float* __restrict__ x, const float* __restrict__ coef, int64_t n4, int64_t total4,
                                    int use_norm, float mean, float inv_std, float* __restrict__ x_norm,
                                    float* __restrict__ noise, float* __restrict__ noisy, unsigned long long seed,
                                    unsigned long long offset) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < total4; i += stride) {
    const int b = (int)(i / n4);
    const float sigma = coef[4 * b];
    const float scale = 1.f / sqrtf(sigma * sigma + 1.f);
    f32x4 xv = load4(x + 4 * i), o;
    const f32x4 nv = philox_normal4(offset + (unsigned long long)i, seed);
    if (use_norm) {
#pragma unroll
      for (int j = 0; j < 4; ++j) xv[j] = (xv[j] - mean) * inv_std;
      store4(x_norm + 4 * i, xv);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (xv[j] + nv[j] * sigma) * scale;
    store4(noise + 4 * i, nv);
    store4(noisy + 4 * i, o);
  }
}

// diffusion.py:77-82 ; rectified_flow.py:67-71
__global__ void qsample_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                               const float* __restrict__ coef, int64_t n4, int64_t total4,
                               float* __restrict__ noisy, bf16_t* __restrict__ noisy_bf) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < total4; i += stride) {
    int b = (int)(i / n4);
    float sigma = coef[4 * b];
    float scale = 1.f / sqrtf(sigma * sigma + 1.f);
    f32x4 xv = load4(x + 4 * i), nv = load4(noise + 4 * i), o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (xv[j] + nv[j] * sigma) * scale;
    store4(noisy + 4 * i, o);
    if (noisy_bf) store4(noisy_bf + 4 * i, o);
  }
}

// trainer.py:241-244 fused into the forward process: x_n = (x - vae_mean) / vae_std is what the loss sees as the clean
// latent, so it is written once here (the loss kernel reads it back) together with noisy = (x_n + noise sigma) scale.
__global__ void qsample_norm_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                                    const float* __restrict__ coef, int64_t n4, int64_t total4, float mean, float inv_std,
                                    float* __restrict__ x_norm, float* __restrict__ noisy) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < total4; i += stride) {
    int b = (int)(i / n4);
    float sigma = coef[4 * b];
    float scale = 1.f / sqrtf(sigma * sigma + 1.f);
    f32x4 xv = load4(x + 4 * i), nv = load4(noise + 4 * i), xn, o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xn[j] = (xv[j] - mean) * inv_std;
      o[j] = (xn[j] + nv[j] * sigma) * scale;
    }
    store4(x_norm + 4 * i, xn);
    store4(noisy + 4 * i, o);
  }
}

// text_encoders.py:196-262: one encoder's [B, S, F] hidden states land in the [B, S_total, F_total] context at
// (sequence offset of its bucket, feature offset inside the bucket), times its attention mask (zero_for_padding).
// The destination is zero-filled beforehand, which is the reference's F.pad of the narrower buckets.
template <typename T>
__global__ void ctx_place_kernel(const T* __restrict__ src, const long long* __restrict__ mask, float* __restrict__ out,
                                 int B, int S, int F, int S_total, int F_total, int s_off, int f_off) {
  const int64_t total = (int64_t)B * S * F, step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
    const int f = (int)(i % F);
    const int64_t bs = i / F;
    const int sidx = (int)(bs % S), b = (int)(bs / S);
    float v = to_f32(src[i]);
    if (mask) v *= (float)mask[bs];
    out[((int64_t)b * S_total + s_off + sidx) * F_total + f_off + f] = v;
  }
}

// ---------------------------------------------------------------------------------------
// diffusion.py:100-125 elementwise, in the reference's own operation order
__device__ __forceinline__ void x0_eps(int ptype, float xt, float out, float s, float scales, float& x0,
                                       float& eps) {
  switch (ptype) {
    case UWU_PT_SAMPLE:
      x0 = out;
      eps = (xt / scales - x0) / s;
      break;
    case UWU_PT_EPSILON:
      eps = out;
      x0 = xt / scales - s * eps;
      break;
    case UWU_PT_V:
      x0 = scales * (xt - s * out);
      eps = (xt / scales - x0) / s;
      break;
    default:  // UWU_PT_RF
      x0 = (xt / scales - s * out) / (1.f + s);
      eps = (xt / scales + out) / (1.f + s);
      break;
  }
}
// diffusion.py:84-98
__device__ __forceinline__ float target_of(int ttype, float x0, float eps, float sa, float sb) {
  switch (ttype) {
    case UWU_PT_EPSILON: return eps;
    case UWU_PT_V: return sa * eps - sb * x0;
    case UWU_PT_SAMPLE: return x0;
    default: return eps - x0;
  }
}
// d pred / d model_output (per-sample scalar; every conversion is affine in `out`)
__device__ __forceinline__ float pred_coeff(int ptype, int ttype, bool convert, float s, float scales,
                                            float sa, float sb) {
  if (!convert) return 1.f;
  float dx0, deps;
  switch (ptype) {
    case UWU_PT_SAMPLE: dx0 = 1.f; deps = -1.f / s; break;
    case UWU_PT_EPSILON: dx0 = -s; deps = 1.f; break;
    case UWU_PT_V: dx0 = -scales * s; deps = scales; break;
    default: dx0 = -s / (1.f + s); deps = 1.f / (1.f + s); break;
  }
  switch (ttype) {
    case UWU_PT_EPSILON: return deps;
    case UWU_PT_V: return sa * deps - sb * dx0;
    case UWU_PT_SAMPLE: return dx0;
    default: return deps - dx0;
  }
}

// One workgroup per sample: pass over the sample computes pred/target, the squared-error sum
// (wave shuffle + LDS reduce) and writes d loss / d out in the same pass -- the gradient only
// needs per-sample scalars known up front (w_b, c_b, 1/(n*B)).  diffusion.py:177-193.
template <typename TO, int NT>
__global__ void __launch_bounds__(NT) loss_fwd_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ noise, const float* __restrict__ xt,
    const TO* __restrict__ mo, const float* __restrict__ coef, int ptype, int ttype, int convert, int B,
    int64_t n, float* __restrict__ losses, TO* __restrict__ grad, float* __restrict__ pred_o,
    float* __restrict__ target_o) {
  __shared__ float red[NT / 64];
  const int b = blockIdx.x;
  const float sigma = coef[4 * b], w = coef[4 * b + 1], sa = coef[4 * b + 2], sb = coef[4 * b + 3];
  const float scales = 1.f / sqrtf(sigma * sigma + 1.f);
  const float c = pred_coeff(ptype, ttype, convert != 0, sigma, scales, sa, sb);
  const float gs = 2.f * w * c / ((float)n * (float)B);
  const int64_t base = (int64_t)b * n;
  float acc = 0.f;
  for (int64_t i = (int64_t)threadIdx.x * 4; i < n; i += (int64_t)NT * 4) {
    f32x4 xv = load4(x + base + i), nv = load4(noise + base + i), ov = load4(mo + base + i);
    f32x4 xtv = convert ? load4(xt + base + i) : xv;
    f32x4 pv, tv, gv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float p;
      if (convert) {
        float x0, eps;
        x0_eps(ptype, xtv[j], ov[j], sigma, scales, x0, eps);
        p = target_of(ttype, x0, eps, sa, sb);
      } else {
        p = ov[j];
      }
      float t = target_of(ttype, xv[j], nv[j], sa, sb);
      float d = p - t;
      acc += d * d;
      pv[j] = p;
      tv[j] = t;
      gv[j] = gs * d;
    }
    store4(grad + base + i, gv);
    if (pred_o) store4(pred_o + base + i, pv);
    if (target_o) store4(target_o + base + i, tv);
  }
  float tot = block_sum<NT / 64>(acc, red);
  if (threadIdx.x == 0) losses[b] = w * (tot / (float)n);
}

__global__ void mean_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += v[i];
  float t = block_sum<4>(a, red);
  if (threadIdx.x == 0) out[0] = t / (float)n;
}

template <typename T>
__global__ void scale_inplace_kernel(T* __restrict__ y, int64_t n4, const float* __restrict__ scale) {
  const float s = scale[0];
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    f32x4 v = load4(y + 4 * i);
    v = v * s;
    store4(y + 4 * i, v);
  }
}

// ---------------------------------------------------------------------------------------
// Sampler step: classifier-free guidance combine (reference sampling/cfg.py:113-125),
// eps-denoiser (k_diffusion_wrapper.py:98-108: denoised = x - sigma*eps) and the Euler-ancestral update
// (k_diffusion_euler.py:42-47 with k-diffusion's to_d / get_ancestral_step) fused into one pass:
//   eps = uncond + (cond - uncond)*cfg ;  d = (x - denoised)/sigma = eps ;
//   x' = x + d*(sigma_down - sigma) + noise * s_noise * sigma_up
__global__ void sampler_step_kernel(const float* __restrict__ x, const float* __restrict__ eps_c,
                                    const float* __restrict__ eps_u, const float* __restrict__ noise,
                                    float* __restrict__ out, float* __restrict__ denoised, int64_t n4, float cfg,
                                    float sigma, float sigma_down, float sigma_up_noise) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 xv = load4(x + 4 * i), c = load4(eps_c + 4 * i);
    f32x4 u = eps_u ? load4(eps_u + 4 * i) : c;
    f32x4 nz = noise ? load4(noise + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 o, dn;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float eps = u[e] + (c[e] - u[e]) * cfg;
      dn[e] = xv[e] - sigma * eps;
      const float d = (xv[e] - dn[e]) / sigma;
      o[e] = xv[e] + d * (sigma_down - sigma) + nz[e] * sigma_up_noise;
    }
    store4(out + 4 * i, o);
    if (denoised) store4(denoised + 4 * i, dn);
  }
}

__global__ void scale_copy_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n4, float s) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) store4(y + 4 * i, load4(x + 4 * i) * s);
}

// ------------------------------------------------------------------------------- C ABI

extern "C" int uwu_schedule_gather(const int64_t* timesteps, const float* sigmas_desc, const float* all_snr,
                                   const float* alphas_cumprod, int n_train, int B, int snr_mode, float gamma,
                                   int debias, float* coef, void* stream) {
  UWU_CHECK_ARG(timesteps && sigmas_desc && all_snr && alphas_cumprod && coef, "schedule_gather: null pointer");
  UWU_CHECK_ARG(B > 0 && n_train > 1, "schedule_gather: B=%d n_train=%d", B, n_train);
  UWU_CHECK_ARG(snr_mode >= 0 && snr_mode <= 2, "schedule_gather: snr_mode=%d", snr_mode);
  hipLaunchKernelGGL(schedule_gather_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, timesteps,
                     sigmas_desc, all_snr, alphas_cumprod, n_train, B, snr_mode, gamma, debias, coef);
  UWU_LAUNCH_CHECK("schedule_gather");
  return UWU_OK;
}

extern "C" int uwu_rf_time_to_sigma(const float* u01, float sigma_max, const float* log_sigmas_asc, int n_train,
                                    int B, float* coef, float* timesteps, void* stream) {
  UWU_CHECK_ARG(u01 && log_sigmas_asc && coef && timesteps, "rf_time_to_sigma: null pointer");
  UWU_CHECK_ARG(B > 0 && n_train > 2, "rf_time_to_sigma: B=%d n_train=%d", B, n_train);
  hipLaunchKernelGGL(rf_time_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, u01, sigma_max,
                     log_sigmas_asc, n_train, B, coef, timesteps);
  UWU_LAUNCH_CHECK("rf_time_to_sigma");
  return UWU_OK;
}

extern "C" int uwu_philox_raw(uint32_t* out, int64_t n_counters, uint64_t seed, uint64_t offset, void* stream) {
  UWU_CHECK_ARG(out && n_counters > 0, "philox_raw: bad args");
  hipLaunchKernelGGL(philox_raw_kernel, dim3(cdiv(n_counters, 256)), dim3(256), 0, (hipStream_t)stream, out, n_counters,
                     (unsigned long long)seed, (unsigned long long)offset);
  UWU_LAUNCH_CHECK("philox_raw");
  return UWU_OK;
}
extern "C" int uwu_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
  UWU_CHECK_ARG(out && n > 0 && n % 4 == 0 && ((uintptr_t)out & 15) == 0, "philox_normal: n=%lld must be a positive multiple of 4", (long long)n);
  hipLaunchKernelGGL(philox_normal_kernel, dim3(ew_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, out, n / 4,
                     (unsigned long long)seed, (unsigned long long)offset);
  UWU_LAUNCH_CHECK("philox_normal");
  return UWU_OK;
}
extern "C" int uwu_draw_timesteps(int64_t* timesteps, int n_train, int B, uint64_t seed, uint64_t offset, void* stream) {
  UWU_CHECK_ARG(timesteps && B > 0 && n_train > 0, "draw_timesteps: bad args");
  hipLaunchKernelGGL(schedule_draw_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, timesteps, n_train, B,
                     (unsigned long long)seed, (unsigned long long)offset);
  UWU_LAUNCH_CHECK("draw_timesteps");
  return UWU_OK;
}
extern "C" int uwu_draw_u01(float* u01, int B, uint64_t seed, uint64_t offset, void* stream) {
  UWU_CHECK_ARG(u01 && B > 0, "draw_u01: bad args");
  hipLaunchKernelGGL(u01_draw_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, u01, B, (unsigned long long)seed,
                     (unsigned long long)offset);
  UWU_LAUNCH_CHECK("draw_u01");
  return UWU_OK;
}
extern "C" int uwu_qsample_draw(const float* x, const float* coef, int B, int64_t n, int use_norm, float vae_mean, float vae_std,
                                float* x_norm, float* noise, float* noisy, uint64_t seed, uint64_t offset, void* stream) {
  UWU_CHECK_ARG(x && coef && noise && noisy && (!use_norm || x_norm), "qsample_draw: null pointer");
  UWU_CHECK_ARG(B > 0 && n > 0 && n % 4 == 0, "qsample_draw: n=%lld must be a positive multiple of 4", (long long)n);
  UWU_CHECK_ARG(!use_norm || vae_std != 0.f, "qsample_draw: vae_std must be non-zero");
  const int64_t total4 = (int64_t)B * n / 4;
  hipLaunchKernelGGL(qsample_draw_kernel, dim3(ew_grid(total4, 256)), dim3(256), 0, (hipStream_t)stream, x, coef, n / 4, total4,
                     use_norm, vae_mean, use_norm ? 1.f / vae_std : 1.f, x_norm, noise, noisy, (unsigned long long)seed,
                     (unsigned long long)offset);
  UWU_LAUNCH_CHECK("qsample_draw");
  return UWU_OK;
}

extern "C" int uwu_qsample(const float* x, const float* noise, const float* coef, int B, int64_t n, float* noisy,
                           void* noisy_bf16, void* stream) {
  UWU_CHECK_ARG(x && noise && coef && noisy, "qsample: null pointer");
  UWU_CHECK_ARG(B > 0 && n > 0 && n % 4 == 0, "qsample: n=%lld must be a positive multiple of 4", (long long)n);
  int64_t total4 = (int64_t)B * n / 4;
  hipLaunchKernelGGL(qsample_kernel, dim3(ew_grid(total4, 256)), dim3(256), 0, (hipStream_t)stream, x, noise, coef,
                     n / 4, total4, noisy, (bf16_t*)noisy_bf16);
  UWU_LAUNCH_CHECK("qsample");
  return UWU_OK;
}

extern "C" int uwu_qsample_norm(const float* x, const float* noise, const float* coef, int B, int64_t n, float vae_mean,
                                float vae_std, float* x_norm, float* noisy, void* stream) {
  UWU_CHECK_ARG(x && noise && coef && noisy && x_norm, "qsample_norm: null pointer");
  UWU_CHECK_ARG(B > 0 && n > 0 && n % 4 == 0, "qsample_norm: n=%lld must be a positive multiple of 4", (long long)n);
  UWU_CHECK_ARG(vae_std != 0.f, "qsample_norm: vae_std must be non-zero");
  int64_t total4 = (int64_t)B * n / 4;
  hipLaunchKernelGGL(qsample_norm_kernel, dim3(ew_grid(total4, 256)), dim3(256), 0, (hipStream_t)stream, x, noise, coef,
                     n / 4, total4, vae_mean, 1.f / vae_std, x_norm, noisy);
  UWU_LAUNCH_CHECK("qsample_norm");
  return UWU_OK;
}

extern "C" int uwu_ctx_place(const void* src, int dtype, const int64_t* mask, float* out, int B, int S, int F, int S_total,
                             int F_total, int s_off, int f_off, void* stream) {
  UWU_CHECK_ARG(src && out && B > 0 && S > 0 && F > 0, "ctx_place: bad argument");
  UWU_CHECK_ARG(s_off >= 0 && f_off >= 0 && s_off + S <= S_total && f_off + F <= F_total, "ctx_place: block outside the context");
  const int grid = ew_grid((int64_t)B * S * F, 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UWU_F32)
    hipLaunchKernelGGL((ctx_place_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)src, (const long long*)mask, out, B,
                       S, F, S_total, F_total, s_off, f_off);
  else if (dtype == UWU_BF16)
    hipLaunchKernelGGL((ctx_place_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)src, (const long long*)mask, out,
                       B, S, F, S_total, F_total, s_off, f_off);
  else
    UWU_CHECK_ARG(false, "ctx_place: bad dtype");
  UWU_LAUNCH_CHECK("ctx_place");
  return UWU_OK;
}

extern "C" int uwu_loss_fwd_bwd(const float* x, const float* noise, const float* xt, const void* model_output,
                                int out_dtype, const float* coef, int pred_type, int target_type, int force_convert,
                                int B, int64_t n, float* losses, float* loss_mean, void* grad_out, float* pred,
                                float* target, void* stream) {
  UWU_CHECK_ARG(x && noise && model_output && coef && losses && grad_out, "loss_fwd_bwd: null pointer");
  UWU_CHECK_ARG(B > 0 && n > 0 && n % 4 == 0, "loss_fwd_bwd: n=%lld must be a positive multiple of 4", (long long)n);
  UWU_CHECK_ARG(pred_type >= 0 && pred_type <= 3 && target_type >= 0 && target_type <= 3,
                "loss_fwd_bwd: Unsupported prediction/target type %d/%d", pred_type, target_type);
  int convert = (force_convert || pred_type != target_type) ? 1 : 0;
  UWU_CHECK_ARG(!convert || xt, "loss_fwd_bwd: xt required when prediction_type != target_type");
  constexpr int NT = 256;
  if (out_dtype == UWU_F32) {
    hipLaunchKernelGGL((loss_fwd_bwd_kernel<float, NT>), dim3(B), dim3(NT), 0, (hipStream_t)stream, x, noise, xt,
                       (const float*)model_output, coef, pred_type, target_type, convert, B, n, losses,
                       (float*)grad_out, pred, target);
  } else if (out_dtype == UWU_BF16) {
    hipLaunchKernelGGL((loss_fwd_bwd_kernel<bf16_t, NT>), dim3(B), dim3(NT), 0, (hipStream_t)stream, x, noise, xt,
                       (const bf16_t*)model_output, coef, pred_type, target_type, convert, B, n, losses,
                       (bf16_t*)grad_out, pred, target);
  } else {
    UWU_CHECK_ARG(false, "loss_fwd_bwd: bad dtype %d", out_dtype);
  }
  UWU_LAUNCH_CHECK("loss_fwd_bwd");
  if (loss_mean) {
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, losses, B, loss_mean);
    UWU_LAUNCH_CHECK("loss_mean");
  }
  return UWU_OK;
}

template <typename T>
__global__ void scale_into_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n4, const float* __restrict__ scale) {
  const float s = scale[0];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) store4(y + 4 * i, load4(x + 4 * i) * s);
}

__global__ void cast_i64_f32_kernel(const int64_t* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = (float)x[i];
}

extern "C" int uwu_cast_i64_to_f32(const int64_t* x, float* y, int64_t n, void* stream) {
  UWU_CHECK_ARG(x && y && n > 0, "cast_i64_to_f32: bad args (n=%lld)", (long long)n);
  hipLaunchKernelGGL(cast_i64_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
  UWU_LAUNCH_CHECK("cast_i64_to_f32");
  return UWU_OK;
}

extern "C" int uwu_scale_into(const void* x, void* y, int dtype, int64_t n, const float* scale, void* stream) {
  UWU_CHECK_ARG(x && y && scale && n > 0 && n % 4 == 0, "scale_into: bad args (n=%lld)", (long long)n);
  UWU_CHECK_ARG(dtype == UWU_F32 || dtype == UWU_BF16, "scale_into: bad dtype %d", dtype);
  if (dtype == UWU_F32)
    hipLaunchKernelGGL((scale_into_kernel<float>), dim3(ew_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                       (float*)y, n / 4, scale);
  else
    hipLaunchKernelGGL((scale_into_kernel<bf16_t>), dim3(ew_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (bf16_t*)y, n / 4, scale);
  UWU_LAUNCH_CHECK("scale_into");
  return UWU_OK;
}

extern "C" int uwu_scale_inplace(void* y, int dtype, int64_t n, const float* scale, void* stream) {
  UWU_CHECK_ARG(y && scale && n > 0 && n % 4 == 0, "scale_inplace: bad args (n=%lld)", (long long)n);
  if (dtype == UWU_F32)
    hipLaunchKernelGGL((scale_inplace_kernel<float>), dim3(ew_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       (float*)y, n / 4, scale);
  else
    hipLaunchKernelGGL((scale_inplace_kernel<bf16_t>), dim3(ew_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       (bf16_t*)y, n / 4, scale);
  UWU_LAUNCH_CHECK("scale_inplace");
  return UWU_OK;
}

extern "C" int uwu_sampler_step(const float* x, const float* eps_cond, const float* eps_uncond, const float* noise,
                                float* out, float* denoised, int64_t n, float cfg, float sigma, float sigma_down,
                                float sigma_up, float s_noise, void* stream) {
  UWU_CHECK_ARG(x && eps_cond && out && n > 0 && n % 4 == 0 && sigma > 0.f, "sampler_step: bad args");
  hipLaunchKernelGGL(sampler_step_kernel, dim3(ew_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, eps_cond,
                     eps_uncond, noise, out, denoised, n / 4, cfg, sigma, sigma_down, sigma_up * s_noise);
  UWU_LAUNCH_CHECK("sampler_step");
  return UWU_OK;
}

// out = base + a * (eu + cfg (ec - eu)) + b * eu + c * noise ; eu == NULL: unguided (eu := ec); noise == NULL: c unused
__global__ void __launch_bounds__(256) sampler_combine_kernel(const float* __restrict__ base, const float* __restrict__ ec,
                                                              const float* __restrict__ eu, const float* __restrict__ nz,
                                                              float* __restrict__ out, int64_t n4, float cfg, float a,
                                                              float b, float c) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 xv = load4(base + 4 * i), cv = load4(ec + 4 * i);
    const f32x4 uv = eu ? load4(eu + 4 * i) : cv;
    f32x4 o = xv + (uv + (cv - uv) * cfg) * a + uv * b;
    if (nz) o = o + load4(nz + 4 * i) * c;
    store4(out + 4 * i, o);
  }
}

extern "C" int uwu_sampler_combine(const float* base, const float* eps_cond, const float* eps_uncond, const float* noise,
                                   float* out, int64_t n, float cfg, float a, float b, float c, void* stream) {
  UWU_CHECK_ARG(base && eps_cond && out && n > 0 && n % 4 == 0, "sampler_combine: bad args");
  hipLaunchKernelGGL(sampler_combine_kernel, dim3(ew_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, base,
                     eps_cond, eps_uncond, noise, out, n / 4, cfg, a, b, c);
  UWU_LAUNCH_CHECK("sampler_combine");
  return UWU_OK;
}

extern "C" int uwu_scale_copy(const float* x, float* y, int64_t n, float scale, void* stream) {
  UWU_CHECK_ARG(x && y && n > 0 && n % 4 == 0, "scale_copy: bad args");
  hipLaunchKernelGGL(scale_copy_kernel, dim3(ew_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n / 4, scale);
  UWU_LAUNCH_CHECK("scale_copy");
  return UWU_OK;
}
