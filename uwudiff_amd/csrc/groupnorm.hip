// GroupNorm(G) (+ SiLU) forward / backward on channels-last activations x[B, HW, C]  (diffusers ResnetBlock2D /
// Transformer2DModel norms of the reference's denoiser, unet_patch.py:13-57).
// HBM-bound.  Every pass streams whole rows with 16-byte (bf16) / 32-byte (fp32) accesses per lane: a thread owns a
// fixed 8-channel slot and walks rows, so the per-channel constants live in registers and a 256-thread workgroup
// reads RY consecutive rows (RY * C * 2 contiguous bytes) per iteration.  Grid = (row chunks, B): thousands of
// workgroups instead of the B*G of the first version (one workgroup per group walking 2-byte strided elements:
// 500 / 895 us per launch on [6, 16384, 320]; now 3 / 5 coalesced passes over the tensor).
//   forward : sums kernel (sum x, sum x^2 per (b, group): registers -> LDS -> one global atomic per group and
//             workgroup) -> finalize (mean, rstd) -> apply (y = silu?(x a_c + b_c))
//   backward: sums kernel (go = dy * silu'(z); per channel sum go, sum go*xhat -> dgamma / dbeta atomics, per group
//             s1 = sum gamma go, s2 = sum gamma go xhat) -> dx = rstd (go gamma - s1/n - xhat s2/n)
#include "common.h"

namespace {

struct GnMap {
  int VPR, SX, RY, NS;  // 8-channel slots per row; slots covered per pass; rows per iteration; passes over the slots
};
__host__ __device__ inline GnMap gn_map(int C) {
  GnMap m;
  m.VPR = C / 8;
  m.SX = m.VPR < 256 ? m.VPR : 256;
  m.RY = 256 / m.SX;
  m.NS = (m.VPR + m.SX - 1) / m.SX;
  return m;
}
constexpr int GN_MAXNS = 2;       // C <= 4096
constexpr int GN_RED = 2 * 4096;  // floats: RY * C <= 4096 channels-rows, two sums

// MODE 0: v1 = x, v2 = x^2.  MODE 1: go = dy * silu'(xhat gamma + beta): v1 = go, v2 = go * xhat.
template <typename T, int MODE>
__global__ void __launch_bounds__(256) gn_sums_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ acc1, float* __restrict__ acc2,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta, int HW, int C,
                                                      int G, int rows_per_block, int silu) {
  __shared__ float red[GN_RED];
  const GnMap m = gn_map(C);
  const int tid = threadIdx.x, sx = tid % m.SX, ry = tid / m.SX;
  const int b = blockIdx.y, cpg = C / G;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > HW) r1 = HW;
  f32x8 a1[GN_MAXNS], a2[GN_MAXNS], cm[GN_MAXNS], cr[GN_MAXNS], cg[GN_MAXNS], cb[GN_MAXNS];
#pragma unroll
  for (int s = 0; s < GN_MAXNS; ++s) {
    a1[s] = a2[s] = f32x8{};
    const int slot = sx + s * m.SX;
    if (MODE == 1 && slot < m.VPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = 8 * slot + e, g = b * G + c / cpg;
        cm[s][e] = mean[g];
        cr[s][e] = rstd[g];
        cg[s][e] = gamma[c];
        cb[s][e] = beta[c];
      }
    }
  }
  if (ry < m.RY) {
    for (int row = r0 + ry; row < r1; row += m.RY) {
      const int64_t base = ((int64_t)b * HW + row) * C;
#pragma unroll
      for (int s = 0; s < GN_MAXNS; ++s) {
        const int slot = sx + s * m.SX;
        if (slot >= m.VPR) continue;
        const f32x8 xv = load8(x + base + 8 * slot);
        if constexpr (MODE == 0) {
          a1[s] = a1[s] + xv;
          a2[s] = a2[s] + xv * xv;
        } else {
          f32x8 go = load8(dy + base + 8 * slot);
          const f32x8 xh = (xv - cm[s]) * cr[s];
          if (silu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) go[e] *= dsilu_f(xh[e] * cg[s][e] + cb[s][e]);
          }
          a1[s] = a1[s] + go;
          a2[s] = a2[s] + go * xh;
        }
      }
    }
  }
  // block reduction over the RY row-lanes of every channel
  float* r1s = red;
  float* r2s = red + GN_RED / 2;
  if (ry < m.RY) {
#pragma unroll
    for (int s = 0; s < GN_MAXNS; ++s) {
      const int slot = sx + s * m.SX;
      if (slot >= m.VPR) continue;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        r1s[ry * C + 8 * slot + e] = a1[s][e];
        r2s[ry * C + 8 * slot + e] = a2[s][e];
      }
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float t1 = 0.f, t2 = 0.f;
    for (int y = 0; y < m.RY; ++y) {
      t1 += r1s[y * C + c];
      t2 += r2s[y * C + c];
    }
    if constexpr (MODE == 1) {
      atomicAdd(dbeta + c, t1);
      atomicAdd(dgamma + c, t2);
      const float gm = gamma[c];
      t1 *= gm;
      t2 *= gm;
    }
    r1s[c] = t1;  // row 0 of the buffer: only this thread touches column c
    r2s[c] = t2;
  }
  __syncthreads();
  if (tid < G) {
    float t1 = 0.f, t2 = 0.f;
    for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
      t1 += r1s[c];
      t2 += r2s[c];
    }
    atomicAdd(acc1 + b * G + tid, t1);
    atomicAdd(acc2 + b * G + tid, t2);
  }
}

__global__ void gn_finalize_kernel(float* __restrict__ mean, float* __restrict__ rstd, int BG, float inv_n, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BG) return;
  const float mu = mean[i] * inv_n;
  float var = rstd[i] * inv_n - mu * mu;
  if (var < 0.f) var = 0.f;
  mean[i] = mu;
  rstd[i] = 1.f / sqrtf(var + eps);
}

// MODE 0: y = silu?(x a + b), a = gamma rstd, b = beta - mean a.
// MODE 1: dx = rstd (go gamma - s1/n - xhat s2/n), go = dy * silu'(xhat gamma + beta).
template <typename T, int MODE>
__global__ void __launch_bounds__(256) gn_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ s1, const float* __restrict__ s2,
                                                       T* __restrict__ out, int HW, int C, int G, int rows_per_block,
                                                       int silu, float inv_n) {
  const GnMap m = gn_map(C);
  const int tid = threadIdx.x, sx = tid % m.SX, ry = tid / m.SX;
  if (ry >= m.RY) return;
  const int b = blockIdx.y, cpg = C / G;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > HW) r1 = HW;
  f32x8 ca[GN_MAXNS], cb[GN_MAXNS], cm[GN_MAXNS], cr[GN_MAXNS], c1[GN_MAXNS], c2[GN_MAXNS];
#pragma unroll
  for (int s = 0; s < GN_MAXNS; ++s) {
    const int slot = sx + s * m.SX;
    if (slot >= m.VPR) continue;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = 8 * slot + e, g = b * G + c / cpg;
      if constexpr (MODE == 0) {
        ca[s][e] = gamma[c] * rstd[g];
        cb[s][e] = beta[c] - mean[g] * ca[s][e];
      } else {
        ca[s][e] = gamma[c];
        cb[s][e] = beta[c];
        cm[s][e] = mean[g];
        cr[s][e] = rstd[g];
        c1[s][e] = s1[g] * inv_n;
        c2[s][e] = s2[g] * inv_n;
      }
    }
  }
  for (int row = r0 + ry; row < r1; row += m.RY) {
    const int64_t base = ((int64_t)b * HW + row) * C;
#pragma unroll
    for (int s = 0; s < GN_MAXNS; ++s) {
      const int slot = sx + s * m.SX;
      if (slot >= m.VPR) continue;
      const f32x8 xv = load8(x + base + 8 * slot);
      f32x8 o;
      if constexpr (MODE == 0) {
        o = xv * ca[s] + cb[s];
        if (silu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = silu_f(o[e]);
        }
      } else {
        f32x8 go = load8(dy + base + 8 * slot);
        const f32x8 xh = (xv - cm[s]) * cr[s];
        if (silu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) go[e] *= dsilu_f(xh[e] * ca[s][e] + cb[s][e]);
        }
        o = cr[s] * (go * ca[s] - c1[s] - xh * c2[s]);
      }
      store8(out + base + 8 * slot, o);
    }
  }
}

int gn_rows_per_block(int HW, int B, int RY) {
  // ~2048 workgroups in all, at least 4 iterations of RY rows each
  int chunks = 2048 / (B > 0 ? B : 1);
  if (chunks < 1) chunks = 1;
  int rows = (HW + chunks - 1) / chunks;
  if (rows < 4 * RY) rows = 4 * RY;
  return rows;
}

template <typename T>
int gn_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int B, int HW, int C,
           int G, float eps, int silu, hipStream_t st) {
  const GnMap m = gn_map(C);
  const int rows = gn_rows_per_block(HW, B, m.RY);
  const dim3 grid((HW + rows - 1) / rows, B);
  (void)hipMemsetAsync(mean, 0, sizeof(float) * B * G, st);
  (void)hipMemsetAsync(rstd, 0, sizeof(float) * B * G, st);
  hipLaunchKernelGGL((gn_sums_kernel<T, 0>), grid, dim3(256), 0, st, (const T*)x, (const T*)nullptr, nullptr, nullptr,
                     gamma, beta, mean, rstd, nullptr, nullptr, HW, C, G, rows, 0);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((B * G + 255) / 256), dim3(256), 0, st, mean, rstd, B * G,
                     1.f / ((float)HW * (float)(C / G)), eps);
  hipLaunchKernelGGL((gn_apply_kernel<T, 0>), grid, dim3(256), 0, st, (const T*)x, (const T*)nullptr, mean, rstd, gamma,
                     beta, nullptr, nullptr, (T*)y, HW, C, G, rows, silu, 0.f);
  return UWU_OK;
}

template <typename T>
int gn_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
           void* dx, float* dgamma, float* dbeta, float* ws, int B, int HW, int C, int G, int silu, hipStream_t st) {
  const GnMap m = gn_map(C);
  const int rows = gn_rows_per_block(HW, B, m.RY);
  const dim3 grid((HW + rows - 1) / rows, B);
  float* s1 = ws;
  float* s2 = ws + B * G;
  (void)hipMemsetAsync(ws, 0, sizeof(float) * 2 * B * G, st);
  hipLaunchKernelGGL((gn_sums_kernel<T, 1>), grid, dim3(256), 0, st, (const T*)x, (const T*)dy, mean, rstd, gamma, beta,
                     s1, s2, dgamma, dbeta, HW, C, G, rows, silu);
  hipLaunchKernelGGL((gn_apply_kernel<T, 1>), grid, dim3(256), 0, st, (const T*)x, (const T*)dy, mean, rstd, gamma, beta,
                     s1, s2, (T*)dx, HW, C, G, rows, silu, 1.f / ((float)HW * (float)(C / G)));
  return UWU_OK;
}

}  // namespace

extern "C" int uwu_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                 float* rstd, int B, int HW, int C, int G, float eps, int silu, int dtype,
                                 void* stream) {
  UWU_CHECK_ARG(x && gamma && beta && y && mean && rstd, "groupnorm_fwd: null pointer");
  UWU_CHECK_ARG(B > 0 && HW > 0 && C > 0 && G > 0 && G <= 256 && C % G == 0 && C % 8 == 0 && C <= 4096,
                "groupnorm_fwd: bad shape C=%d G=%d (C %% 8 == 0, C <= 4096)", C, G);
  UWU_CHECK_ARG((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "groupnorm_fwd: x / y must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (dtype == UWU_F32) rc = gn_fwd<float>(x, gamma, beta, y, mean, rstd, B, HW, C, G, eps, silu, st);
  else if (dtype == UWU_BF16) rc = gn_fwd<bf16_t>(x, gamma, beta, y, mean, rstd, B, HW, C, G, eps, silu, st);
  else { uwu_set_error("groupnorm_fwd: bad dtype %d", dtype); return UWU_EINVAL; }
  UWU_LAUNCH_CHECK("groupnorm_fwd");
  return rc;
}

extern "C" int uwu_groupnorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd,
                                 const float* gamma, const float* beta, void* dx, float* dgamma, float* dbeta,
                                 float* ws, int B, int HW, int C, int G, int silu, int dtype, void* stream) {
  UWU_CHECK_ARG(dy && x && mean && rstd && gamma && beta && dx && dgamma && dbeta && ws, "groupnorm_bwd: null pointer");
  UWU_CHECK_ARG(B > 0 && HW > 0 && C > 0 && G > 0 && G <= 256 && C % G == 0 && C % 8 == 0 && C <= 4096,
                "groupnorm_bwd: bad shape C=%d G=%d", C, G);
  UWU_CHECK_ARG((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0, "groupnorm_bwd: tensors must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (dtype == UWU_F32) rc = gn_bwd<float>(dy, x, mean, rstd, gamma, beta, dx, dgamma, dbeta, ws, B, HW, C, G, silu, st);
  else if (dtype == UWU_BF16) rc = gn_bwd<bf16_t>(dy, x, mean, rstd, gamma, beta, dx, dgamma, dbeta, ws, B, HW, C, G, silu, st);
  else { uwu_set_error("groupnorm_bwd: bad dtype %d", dtype); return UWU_EINVAL; }
  UWU_LAUNCH_CHECK("groupnorm_bwd");
  return rc;
}
