// fp32 Linears with a handful of rows: the conditioning path of the denoisers (timestep MLP, pooled-text projection, the
// adaLN modulation Linear for all blocks at once: reference src/duwu/modules/rope_unet.py:306-309, 393-411 restated in
// oracle/dit.py) sees M = per-GPU batch rows.  Through the 128x128-tile GEMM these were grids of 3 workgroups (K = 1280:
// 87 us) or atomics over a 42 MB weight gradient (77 us); at the reference YAML's batch 16 the eleven launches were 430 us
// of a 3.7 ms step.  Here the work is laid out as what it is -- matrix-vector products over a weight matrix that is read
// (or updated) exactly once, HBM-bound:
//   forward  Y[M,N]  = X[M,K] . W[N,K]^T + b   (optionally Y2 = silu(Y)):  wave <-> output column, lane <-> k
//   dgrad    dX[M,K] = dY[M,N] . W[N,K]:  workgroup <-> slab of n, lane <-> k, wave <-> rows m = w, w + 4, ..
//   wgrad    dW[N,K] += dY[M,N]^T . X[M,K],  db[N] += column sums of dY:  wave <-> weight row, lane <-> k
// X (and the dY slab) sit in LDS; every weight element crosses HBM once.
#include "common.h"

namespace {

constexpr int SK_MAX_LDS = 128 * 1024;

__device__ __forceinline__ float silu_f32(float v) { return v / (1.f + __expf(-v)); }

// ---- forward: 4 waves x NPW columns per workgroup; MT = padded row count (accumulators per column)
template <int MT, int NPW, int EPI>
__global__ void __launch_bounds__(256) skinny_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ Y,
                                                         float* __restrict__ Y2, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [M][K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < (M * K) >> 2; i += 256) reinterpret_cast<f32x4*>(xs)[i] = reinterpret_cast<const f32x4*>(X)[i];
  __syncthreads();
  const int n0 = (blockIdx.x * 4 + wave) * NPW;
  if (n0 >= N) return;  // wave-uniform
  float acc[NPW][MT];
#pragma unroll
  for (int j = 0; j < NPW; ++j)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[j][m] = 0.f;
  for (int k = lane; k < K; k += 64) {
    float w[NPW];
#pragma unroll
    for (int j = 0; j < NPW; ++j) w[j] = (n0 + j < N) ? W[(int64_t)(n0 + j) * K + k] : 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float x = m < M ? xs[m * K + k] : 0.f;
#pragma unroll
      for (int j = 0; j < NPW; ++j) acc[j][m] += w[j] * x;
    }
  }
#pragma unroll
  for (int j = 0; j < NPW; ++j) {
    const int n = n0 + j;
    if (n >= N) break;
    const float b = EPI != UWU_EPI_NONE ? bias[n] : 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float v = wave_sum(acc[j][m]) + b;
      if (lane == 0 && m < M) {
        Y[(int64_t)m * N + n] = v;
        if (EPI == UWU_EPI_BIAS_SILU) Y2[(int64_t)m * N + n] = silu_f32(v);
      }
    }
  }
}

// ---- input gradient: workgroup = slab of NS reduction indices n; wave w owns rows m = w + 4 r (r < R); K <= 64 KC
template <int R, int KC>
__global__ void __launch_bounds__(256) skinny_dgrad_kernel(const float* __restrict__ dY, const float* __restrict__ W,
                                                           float* __restrict__ dX, int M, int N, int K, int NS) {
  extern __shared__ __attribute__((aligned(16))) float dys[];  // [M][NS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * NS;
  const int ns = min(NS, N - n0);
  for (int i = tid; i < M * NS; i += 256) {
    const int m = i / NS, n = i - m * NS;
    dys[i] = n < ns ? dY[(int64_t)m * N + n0 + n] : 0.f;
  }
  __syncthreads();
  float acc[R][KC];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < KC; ++c) acc[r][c] = 0.f;
#pragma unroll 2
  for (int n = 0; n < ns; ++n) {
    float w[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) w[c] = (c * 64 + lane < K) ? W[(int64_t)(n0 + n) * K + c * 64 + lane] : 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int m = wave + 4 * r;
      const float d = m < M ? dys[m * NS + n] : 0.f;
#pragma unroll
      for (int c = 0; c < KC; ++c) acc[r][c] += d * w[c];
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int m = wave + 4 * r;
    if (m >= M) break;
#pragma unroll
    for (int c = 0; c < KC; ++c)
      if (c * 64 + lane < K) atomicAdd(dX + (int64_t)m * K + c * 64 + lane, acc[r][c]);
  }
}

// ---- weight (and bias) gradient: workgroup = 16 weight rows, wave = 4 of them
__global__ void __launch_bounds__(256) skinny_wgrad_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                           float* __restrict__ dW, float* __restrict__ db, int M, int N,
                                                           int K) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // xs [M][K], then dys [M][16]
  float* xs = sm;
  float* dys = sm + M * K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * 16;
  for (int i = tid; i < (M * K) >> 2; i += 256) reinterpret_cast<f32x4*>(xs)[i] = reinterpret_cast<const f32x4*>(X)[i];
  for (int i = tid; i < M * 16; i += 256) {
    const int m = i >> 4, j = i & 15;
    dys[i] = n0 + j < N ? dY[(int64_t)m * N + n0 + j] : 0.f;
  }
  __syncthreads();
  const int nb = n0 + 4 * wave;
  for (int k = lane; k < K; k += 64) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int m = 0; m < M; ++m) {
      const float x = xs[m * K + k];
      const f32x4 d = *reinterpret_cast<const f32x4*>(dys + m * 16 + 4 * wave);
      a0 += d[0] * x;
      a1 += d[1] * x;
      a2 += d[2] * x;
      a3 += d[3] * x;
    }
    if (nb + 0 < N) dW[(int64_t)(nb + 0) * K + k] += a0;
    if (nb + 1 < N) dW[(int64_t)(nb + 1) * K + k] += a1;
    if (nb + 2 < N) dW[(int64_t)(nb + 2) * K + k] += a2;
    if (nb + 3 < N) dW[(int64_t)(nb + 3) * K + k] += a3;
  }
  if (db && lane < 4 && nb + lane < N) {
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += dys[m * 16 + 4 * wave + lane];
    db[nb + lane] += s;
  }
}

// (one call per kernel instantiation: every use site carries its own flag)
#define SK_ALLOW_LDS(kern)                                                                                                  \
  do {                                                                                                                      \
    static bool done_ = false;                                                                                              \
    if (!done_) {                                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SK_MAX_LDS); \
      done_ = true;                                                                                                         \
    }                                                                                                                       \
  } while (0)

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// shapes the three kernels take: up to 64 rows, X [M, K] within 128 KB of LDS, K a multiple of 4
extern "C" int uwu_skinny_linear_ok(int M, int N, int K) {
  return M >= 1 && M <= 64 && N >= 1 && K >= 4 && K % 4 == 0 && (size_t)M * K * sizeof(float) + (size_t)M * 64 <= SK_MAX_LDS;
}

extern "C" int uwu_skinny_linear_fwd(const float* X, const float* W, const float* bias, float* Y, float* Y2, int M, int N,
                                     int K, int epilogue, void* stream) {
  UWU_CHECK_ARG(X && W && Y, "skinny_linear_fwd: null pointer");
  UWU_CHECK_ARG(uwu_skinny_linear_ok(M, N, K), "skinny_linear_fwd: M=%d N=%d K=%d not covered", M, N, K);
  UWU_CHECK_ARG(epilogue == UWU_EPI_NONE || epilogue == UWU_EPI_BIAS || epilogue == UWU_EPI_BIAS_SILU,
                "skinny_linear_fwd: epilogue %d", epilogue);
  UWU_CHECK_ARG(epilogue == UWU_EPI_NONE || bias, "skinny_linear_fwd: bias missing");
  UWU_CHECK_ARG(epilogue != UWU_EPI_BIAS_SILU || Y2, "skinny_linear_fwd: second output missing");
  UWU_CHECK_ARG(aligned16(X), "skinny_linear_fwd: X must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)M * K * sizeof(float);
  UwuProfScope prof(stream);
#define SK_FWD(MT, NPW, EPI)                                                                                          \
  do {                                                                                                                \
    auto kern = skinny_fwd_kernel<MT, NPW, EPI>;                                                                      \
    SK_ALLOW_LDS(kern);                                                                                               \
    hipLaunchKernelGGL(kern, dim3((N + 4 * NPW - 1) / (4 * NPW)), dim3(256), lds, st, X, W, bias, Y, Y2, M, N, K);    \
  } while (0)
#define SK_FWD_E(MT, NPW)                                                   \
  do {                                                                      \
    if (epilogue == UWU_EPI_NONE) SK_FWD(MT, NPW, UWU_EPI_NONE);            \
    else if (epilogue == UWU_EPI_BIAS) SK_FWD(MT, NPW, UWU_EPI_BIAS);       \
    else SK_FWD(MT, NPW, UWU_EPI_BIAS_SILU);                                \
  } while (0)
  if (M <= 16) SK_FWD_E(16, 4);
  else if (M <= 32) SK_FWD_E(32, 2);
  else SK_FWD_E(64, 1);
#undef SK_FWD_E
#undef SK_FWD
  prof.done(UWU_PROF_OTHER, 1, 2.0 * M * N * K, ((double)N * K + (double)M * K + (double)M * N) * 4);
  UWU_LAUNCH_CHECK("skinny_linear_fwd");
  return UWU_OK;
}

// dX is OVERWRITTEN (zeroed here, then accumulated by the slabs with fp32 atomics)
extern "C" int uwu_skinny_linear_dgrad(const float* dY, const float* W, float* dX, int M, int N, int K, void* stream) {
  UWU_CHECK_ARG(dY && W && dX, "skinny_linear_dgrad: null pointer");
  UWU_CHECK_ARG(M >= 1 && M <= 64 && N >= 1 && K >= 1 && K <= 512, "skinny_linear_dgrad: M=%d N=%d K=%d not covered", M, N, K);
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(dX, 0, (size_t)M * K * sizeof(float), st) != hipSuccess) {
    uwu_set_error("skinny_linear_dgrad: memset failed");
    return UWU_ELAUNCH;
  }
  // slab length: ~96 workgroups, at most 16 M reduction indices per slab (atomics = workgroups x M x K), at least 16
  int NS = (N / 96 + 15) & ~15;
  if (NS > 16 * M) NS = 16 * M;
  if (NS > ((SK_MAX_LDS / 4 / M) & ~15)) NS = (SK_MAX_LDS / 4 / M) & ~15;  // the dY slab [M][NS] lives in LDS
  if (NS < 16) NS = 16;
  const int grid = (N + NS - 1) / NS;
  const size_t lds = (size_t)M * NS * sizeof(float);
  UwuProfScope prof(stream);
#define SK_DG(R, KC)                                                                                  \
  do {                                                                                                \
    auto kern = skinny_dgrad_kernel<R, KC>;                                                           \
    SK_ALLOW_LDS(kern);                                                                               \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, dY, W, dX, M, N, K, NS);                 \
  } while (0)
#define SK_DG_K(R)                    \
  do {                                \
    if (K <= 256) SK_DG(R, 4);        \
    else if (K <= 384) SK_DG(R, 6);   \
    else SK_DG(R, 8);                 \
  } while (0)
  if (M <= 16) SK_DG_K(4);
  else if (M <= 32) SK_DG_K(8);
  else SK_DG_K(16);
#undef SK_DG_K
#undef SK_DG
  prof.done(UWU_PROF_OTHER, 1, 2.0 * M * N * K, ((double)N * K + (double)M * K + (double)M * N) * 4);
  UWU_LAUNCH_CHECK("skinny_linear_dgrad");
  return UWU_OK;
}

// dW[N, K] += dY^T X; db[N] += column sums of dY (db may be NULL)
extern "C" int uwu_skinny_linear_wgrad(const float* dY, const float* X, float* dW, float* db, int M, int N, int K,
                                       void* stream) {
  UWU_CHECK_ARG(dY && X && dW, "skinny_linear_wgrad: null pointer");
  UWU_CHECK_ARG(uwu_skinny_linear_ok(M, N, K), "skinny_linear_wgrad: M=%d N=%d K=%d not covered", M, N, K);
  UWU_CHECK_ARG(aligned16(X), "skinny_linear_wgrad: X must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)M * K * sizeof(float) + (size_t)M * 16 * sizeof(float);
  SK_ALLOW_LDS(skinny_wgrad_kernel);
  UwuProfScope prof(stream);
  hipLaunchKernelGGL(skinny_wgrad_kernel, dim3((N + 15) / 16), dim3(256), lds, st, dY, X, dW, db, M, N, K);
  prof.done(UWU_PROF_OTHER, 1, 2.0 * M * N * K, ((double)N * K * 2 + (double)M * K + (double)M * N) * 4);
  UWU_LAUNCH_CHECK("skinny_linear_wgrad");
  return UWU_OK;
}
