// fp8 quantisation for the block-scaled-MFMA GEMM path (gemm.hip: gemm_f8_kernel; BASELINE config 5).
//
// Per-tensor scaling: x_fp8 = sat(x * scale), scale = FMT_MAX / amax.  Two policies share the kernels:
//   * just-in-time: uwu_fp8_amax -> uwu_fp8_update_scales -> uwu_fp8_quantize   (two passes over x; parity tests, step 0)
//   * delayed:      uwu_fp8_quantize with the scale of the previous step while it records this step's amax
//                   (one pass; the scale is refreshed by uwu_fp8_update_scales between steps).
// Everything lives on the device (scales are read through pointers), so neither policy needs a host round trip.
//
// uwu_fp8_quantize reads a row-major [M, K] bf16 / fp32 tensor ONCE and writes any of
//   out   [M, K]  (the operand of a contraction over K),
//   out_t [K, M]  (the operand of a contraction over M: X^T / dY^T of the weight gradient, W^T of the input gradient),
//   colsum[K] +=  column sums of x in fp32 (the bias gradient of the Linear whose dY is being quantised),
//   amax          max |x| (atomic max on the float bits).
// HBM-bound: 128 x 64 tiles, 16-byte loads; the transposed copy is packed four rows to a dword in registers (v_perm_b32),
// crosses an LDS dword image and leaves as 32 contiguous bytes per lane along M (whole 128-byte lines per k row).
#include "common.h"

namespace {

constexpr float F8_MAX[2] = {448.f, 57344.f};  // OCP e4m3fn / e5m2

template <int FMT>
__device__ __forceinline__ unsigned cvt2(float a, float b) {  // two fp8 bytes in the low 16 bits
  const float mx = FMT == 0 ? 448.f : 57344.f;
  a = fminf(fmaxf(a, -mx), mx);
  b = fminf(fmaxf(b, -mx), mx);
  if constexpr (FMT == 0) return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xFFFFu;
  else return (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false) & 0xFFFFu;
}

// v >= 0: the bit pattern is monotone.  Atomics on ONE address serialise at ~12 ns each (MI355X_MICROARCH.md "fanin"):
// with one per workgroup a 27 000-workgroup launch spent 0.33 ms on them alone, so a workgroup first looks at the value
// (an L2-coherent load; a stale smaller value only costs an unnecessary atomic) and almost all of them skip it.
__device__ __forceinline__ void atomic_max_pos(float* dst, float v) {
  const unsigned cur = __hip_atomic_load(reinterpret_cast<unsigned*>(dst), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (__float_as_uint(v) > cur) atomicMax(reinterpret_cast<unsigned*>(dst), __float_as_uint(v));
}

// One atomic per workgroup on ONE address: behind an update_scales (amax = 0) every workgroup of a launch sees 0 and takes it, so
// the launch costs ~12 ns x gridDim.x whatever the tensor size -- 2048 workgroups on a 10 MB weight were 39 us per launch and
// 4.1 ms per DiT-XL/2 fp8 step (112 weights scale just in time).  The grid is two workgroups per CU now, four 16-byte loads per
// thread in flight.
template <typename T>
__global__ void __launch_bounds__(256) amax_kernel(const T* __restrict__ x, int64_t n, float* __restrict__ amax) {
  __shared__ float red[4];
  float m = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 2048;
  int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
  for (; i + 3 * stride + 8 <= n; i += 4 * stride) {
    const f32x8 v0 = load8(x + i), v1 = load8(x + i + stride), v2 = load8(x + i + 2 * stride), v3 = load8(x + i + 3 * stride);
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(fmaxf(m, fmaxf(fabsf(v0[e]), fabsf(v1[e]))), fmaxf(fabsf(v2[e]), fabsf(v3[e])));
  }
  for (; i < n; i += stride) {
    if (i + 8 <= n) {
      const f32x8 v = load8(x + i);
#pragma unroll
      for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[e]));
    } else {
      for (int64_t j = i; j < n; ++j) m = fmaxf(m, fabsf(to_f32(x[j])));
    }
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomic_max_pos(amax, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

__global__ void update_scales_kernel(float* __restrict__ amax, float* __restrict__ scale, const int* __restrict__ fmt,
                                     int n, float margin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = amax[i];
  if (a > 0.f && isfinite(a)) scale[i] = F8_MAX[fmt[i] & 1] / (a * margin);
  else if (!(scale[i] > 0.f)) scale[i] = 1.f;  // nothing seen yet: keep a previous scale, else 1
  amax[i] = 0.f;
}

// one 128 (m) x 64 (k) tile per workgroup; thread = (8-element k chunk c, group of 4 rows rg)
template <typename T, int FMT>
__global__ void __launch_bounds__(256) quantize_kernel(const T* __restrict__ x, int M, int K, int ldx,
                                                       const float* __restrict__ scale, unsigned char* __restrict__ out,
                                                       int ldo, unsigned char* __restrict__ out_t, int ldt,
                                                       float* __restrict__ amax, float* __restrict__ colsum) {
  constexpr int P = 33;                 // dword pitch of the transposed image (odd: the b32 accesses spread over the banks)
  __shared__ unsigned img[64 * P];      // [k][m / 4]: one dword = 4 consecutive rows of one column
  __shared__ float cs[4][64];
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_k = (K + 63) / 64;
  const int tm = blockIdx.x / tiles_k, tk = blockIdx.x - tm * tiles_k;
  const int m0 = tm * 128, k0 = tk * 64;
  const float s = scale[0];
  const int c = tid & 7, rg = tid >> 3;
  const int k = k0 + 8 * c;
  float mx = 0.f;
  f32x8 csum = {0, 0, 0, 0, 0, 0, 0, 0};
  uint2 pk[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + 4 * rg + r;
    f32x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (m < M && k + 8 <= K) v = load8(x + (int64_t)m * ldx + k);
    else if (m < M)
      for (int e = 0; e < 8; ++e)
        if (k + e < K) v[e] = to_f32(x[(int64_t)m * ldx + k + e]);
#pragma unroll
    for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(v[e]));
    csum = csum + v;
    const unsigned b01 = cvt2<FMT>(v[0] * s, v[1] * s), b23 = cvt2<FMT>(v[2] * s, v[3] * s);
    const unsigned b45 = cvt2<FMT>(v[4] * s, v[5] * s), b67 = cvt2<FMT>(v[6] * s, v[7] * s);
    pk[r] = uint2{b01 | (b23 << 16), b45 | (b67 << 16)};
    if (out && m < M) {
      if (k + 8 <= K) *reinterpret_cast<uint2*>(out + (int64_t)m * ldo + k) = pk[r];
      else
        for (int e = 0; e < 8; ++e)
          if (k + e < K) out[(int64_t)m * ldo + k + e] = (unsigned char)((e < 4 ? pk[r].x : pk[r].y) >> (8 * (e & 3)));
    }
  }
  if (out_t) {  // byte e of the four rows -> one dword of column 8c + e (v_perm_b32: two selects per dword)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned w0 = e < 4 ? pk[0].x : pk[0].y, w1 = e < 4 ? pk[1].x : pk[1].y;
      const unsigned w2 = e < 4 ? pk[2].x : pk[2].y, w3 = e < 4 ? pk[3].x : pk[3].y;
      const unsigned sel = 0x0c0c0400u + (unsigned)(e & 3) * 0x0101u;                // {0, 0, w1[e], w0[e]}
      const unsigned lo = __builtin_amdgcn_perm(w1, w0, sel), hi = __builtin_amdgcn_perm(w3, w2, sel);
      img[(8 * c + e) * P + rg] = lo | (hi << 16);
    }
  }
  if (amax) {
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
  }
  if (colsum) {  // lanes with equal c (stride 8) hold the same columns: fold the 8 row groups of the wave, then the waves
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = csum[e];
      t += __shfl_xor(t, 8, 64);
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      if (lane < 8) cs[wave][8 * c + e] = t;
    }
  }
  __syncthreads();
  if (amax && tid == 0) atomic_max_pos(amax, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
  if (colsum && tid < 64 && k0 + tid < K) atomicAdd(colsum + k0 + tid, cs[0][tid] + cs[1][tid] + cs[2][tid] + cs[3][tid]);
  if (out_t) {  // thread -> (k row, 32-row segment of m): 32 contiguous bytes, four lanes cover a 128-byte line
    const int kr = tid >> 2, seg = tid & 3;
    const int kk = k0 + kr, mm = m0 + 32 * seg;
    if (kk < K && mm < M) {
      unsigned w[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) w[i] = img[kr * P + 8 * seg + i];
      unsigned char* dst = out_t + (int64_t)kk * ldt + mm;
      if (mm + 32 <= M) {
        *reinterpret_cast<uint4*>(dst) = uint4{w[0], w[1], w[2], w[3]};
        *reinterpret_cast<uint4*>(dst + 16) = uint4{w[4], w[5], w[6], w[7]};
      } else {
        for (int e = 0; e < 32 && mm + e < M; ++e) dst[e] = (unsigned char)(w[e >> 2] >> (8 * (e & 3)));
      }
    }
  }
}

}  // namespace

extern "C" int uwu_fp8_amax(const void* x, int dtype, int64_t n, float* amax, void* stream) {
  UWU_CHECK_ARG(x && amax && n > 0, "fp8_amax: bad argument");
  UWU_CHECK_ARG(dtype == UWU_F32 || dtype == UWU_BF16, "fp8_amax: bad dtype");
  UWU_CHECK_ARG(((uintptr_t)x & 15) == 0, "fp8_amax: x must be 16-byte aligned");
  int grid = ew_grid((n + 7) / 8, 256);
  static UwuEnv wide("UWU_FP8_AMAX_WIDE");  // "1": the uncapped grid (A/B)
  const int cap = 2 * (uwu_dev_cus() > 0 ? uwu_dev_cus() : 256);
  if (grid > cap && !wide.get().is('1')) grid = cap;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UWU_F32) hipLaunchKernelGGL((amax_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)x, n, amax);
  else hipLaunchKernelGGL((amax_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, n, amax);
  UWU_LAUNCH_CHECK("fp8_amax");
  return UWU_OK;
}

extern "C" int uwu_fp8_update_scales(float* amax, float* scale, const int* fmt, int n, float margin, void* stream) {
  UWU_CHECK_ARG(amax && scale && fmt && n > 0 && margin > 0.f, "fp8_update_scales: bad argument");
  hipLaunchKernelGGL(update_scales_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, amax, scale, fmt, n,
                     margin);
  UWU_LAUNCH_CHECK("fp8_update_scales");
  return UWU_OK;
}

extern "C" int uwu_fp8_quantize(const void* x, int dtype, int M, int K, int ldx, const float* scale, int fmt, void* out,
                                int ldo, void* out_t, int ldt, float* amax, float* colsum, void* stream) {
  UWU_CHECK_ARG(x && scale && M > 0 && K > 0 && ldx >= K, "fp8_quantize: bad argument");
  UWU_CHECK_ARG(out || out_t, "fp8_quantize: no output");
  UWU_CHECK_ARG(dtype == UWU_F32 || dtype == UWU_BF16, "fp8_quantize: bad dtype");
  UWU_CHECK_ARG(fmt == UWU_FP8_E4M3 || fmt == UWU_FP8_E5M2, "fp8_quantize: bad format %d", fmt);
  UWU_CHECK_ARG(((uintptr_t)x & 15) == 0 && ldx % 8 == 0, "fp8_quantize: x must be 16-byte aligned, ldx % 8 == 0");
  UWU_CHECK_ARG(!out || (ldo >= K && ldo % 8 == 0 && ((uintptr_t)out & 7) == 0), "fp8_quantize: out / ldo misaligned");
  UWU_CHECK_ARG(!out_t || (ldt >= M && ldt % 16 == 0 && ((uintptr_t)out_t & 15) == 0), "fp8_quantize: out_t / ldt misaligned");
  const int64_t tiles = (int64_t)((M + 127) / 128) * ((K + 63) / 64);
  UWU_CHECK_ARG(tiles < (1ll << 31), "fp8_quantize: too many tiles");
  hipStream_t st = (hipStream_t)stream;
  UwuProfScope prof(stream);
#define Q_LAUNCH(T, F)                                                                                              \
  hipLaunchKernelGGL((quantize_kernel<T, F>), dim3((unsigned)tiles), dim3(256), 0, st, (const T*)x, M, K, ldx, scale, \
                     (unsigned char*)out, ldo, (unsigned char*)out_t, ldt, amax, colsum)
  if (dtype == UWU_F32) {
    if (fmt == UWU_FP8_E4M3) Q_LAUNCH(float, 0); else Q_LAUNCH(float, 1);
  } else {
    if (fmt == UWU_FP8_E4M3) Q_LAUNCH(bf16_t, 0); else Q_LAUNCH(bf16_t, 1);
  }
#undef Q_LAUNCH
  const double mk = (double)M * K;
  prof.done(UWU_PROF_OTHER, 0, 0.0, mk * (dtype == UWU_BF16 ? 2 : 4) + mk * ((out ? 1 : 0) + (out_t ? 1 : 0)));
  UWU_LAUNCH_CHECK("fp8_quantize");
  return UWU_OK;
}
