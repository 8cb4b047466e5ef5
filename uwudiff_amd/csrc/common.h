// Shared device/host helpers for libuwu_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/uwu_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define UWU_WAVE 64

// ---- error plumbing (host) -------------------------------------------------------------
void uwu_set_error(const char* fmt, ...);

#define UWU_CHECK_ARG(cond, ...)       \
  do {                                 \
    if (!(cond)) {                     \
      uwu_set_error(__VA_ARGS__);      \
      return UWU_EINVAL;               \
    }                                  \
  } while (0)

#define UWU_LAUNCH_CHECK(name)                                                  \
  do {                                                                          \
    hipError_t e_ = hipGetLastError();                                          \
    if (e_ != hipSuccess) {                                                     \
      uwu_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
      return UWU_ELAUNCH;                                                       \
    }                                                                           \
  } while (0)

#include "env.h"

// ---- per-device facts (api.cpp): CU count, opt-in LDS, per-device hipFuncSetAttribute -------------------------------------
int uwu_dev_index();
int uwu_dev_cus();
bool uwu_dev_lds_fits(size_t bytes);
constexpr int UWU_MAX_DEV = 64;
bool uwu_func_lds(const void* fn, size_t bytes, unsigned char* done /* UWU_MAX_DEV flags, zero-initialised */);

// ---- live profiler hooks (prof.cpp) ------------------------------------------------------
int uwu_prof_begin(void* stream);
void uwu_prof_end(int slot, int tag, int kind, double flops, double bytes, void* stream);
struct UwuProfScope {
  int slot;
  void* st;
  explicit UwuProfScope(void* s) : slot(uwu_prof_begin(s)), st(s) {}
  void done(int tag, int kind, double flops, double bytes) {
    if (slot >= 0) uwu_prof_end(slot, tag, kind, flops, bytes, st);
  }
};

// ---- scalar conversions -----------------------------------------------------------------
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }  // RNE, NaN-safe (v_cvt_pk_bf16_f32)

// ---- wave / block reductions ------------------------------------------------------------
// Whole-wave reductions as six DPP steps + one v_readlane (result uniform in every lane).  __shfl_xor compiles to
// ds_bpermute_b32 -- a trip through the LDS crossbar and an lgkmcnt wait per step; the LayerNorm kernels ran two such
// dependent six-step chains per token row.  Steps: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror
// (every lane of a 16-lane row then holds the row's value), row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2
// and 3 (gfx9 DPP controls): lane 63 holds the wave's value.
#define UWU_DPP_F(v, ctrl, rmask, ident) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(ident)), __builtin_bit_cast(int, v), ctrl, rmask, 0xF, false))
__device__ __forceinline__ float wave_sum(float v) {
  v += UWU_DPP_F(v, 0xB1, 0xF, 0.f);
  v += UWU_DPP_F(v, 0x4E, 0xF, 0.f);
  v += UWU_DPP_F(v, 0x141, 0xF, 0.f);
  v += UWU_DPP_F(v, 0x140, 0xF, 0.f);
  v += UWU_DPP_F(v, 0x142, 0xA, 0.f);
  v += UWU_DPP_F(v, 0x143, 0xC, 0.f);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, UWU_DPP_F(v, 0xB1, 0xF, v));
  v = fmaxf(v, UWU_DPP_F(v, 0x4E, 0xF, v));
  v = fmaxf(v, UWU_DPP_F(v, 0x141, 0xF, v));
  v = fmaxf(v, UWU_DPP_F(v, 0x140, 0xF, v));
  v = fmaxf(v, UWU_DPP_F(v, 0x142, 0xA, v));
  v = fmaxf(v, UWU_DPP_F(v, 0x143, 0xC, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// Sum over each aligned group of 16 lanes (a DPP row), result in all 16: four v_add_f32 with DPP operands
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror) -- no trip through the LDS crossbar that
// __shfl_xor (ds_bpermute) takes.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}
// sum over a block of NW waves; result valid in every thread. `red` = NW floats of LDS.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += red[i];
  return t;
}

// ---- vector load/store of 4 consecutive elements as fp32 ---------------------------------
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4(const bf16_t* p) {
  bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store4(bf16_t* p, f32x4 v) {
  bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
  *reinterpret_cast<bf16x4*>(p) = o;
}

// 8 consecutive elements as fp32 (one 16-byte access for bf16)
typedef float f32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x8 load8(const float* p) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  return f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
__device__ __forceinline__ f32x8 load8(const bf16_t* p) {
  bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
  f32x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (float)v[i];
  return r;
}
__device__ __forceinline__ void store8(float* p, f32x8 v) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ void store8(bf16_t* p, f32x8 v) {
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
  *reinterpret_cast<bf16x8*>(p) = o;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float dsilu_f(float x) {
  float s = 1.f / (1.f + __expf(-x));
  return s * (1.f + x * (1.f - s));
}
// tanh-approximate GELU (DiT / diffusers "gelu-approximate"):  0.5 x (1 + tanh(u)) = x * s,  s = 1 / (1 + exp(-2u)),
// u = k0 (x + k1 x^3).  exp(-2u) = exp2(x * (CA + CB x^2)); v_exp_f32 / v_rcp_f32 (1 ulp) instead of an IEEE divide:
// the epilogue of the MLP GEMMs is VALU-bound, 7 instructions per element here against ~22 before.
__device__ __forceinline__ float gelu_sig_f(float x, float x2) {
  const float CA = -2.f * 1.4426950408889634f * 0.7978845608028654f, CB = CA * 0.044715f;
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * (CA + CB * x2)));
}
__device__ __forceinline__ float gelu_tanh_f(float x) { return x * gelu_sig_f(x, x * x); }
// d/dx [x s(x)] = s + x s (1 - s) * 2 k0 (1 + 3 k1 x^2)
__device__ __forceinline__ float dgelu_tanh_f(float x) {
  const float D0 = 2.f * 0.7978845608028654f, D1 = 3.f * 0.044715f * D0;
  const float x2 = x * x;
  const float s = gelu_sig_f(x, x2);
  return s + x * s * (1.f - s) * (D0 + D1 * x2);
}
// Two elements at a time: the multiplies / adds / fmas become v_pk_*_f32 (two fp32 per lane per instruction; only
// v_exp_f32 / v_rcp_f32 stay scalar).  The GEMM epilogues that apply these are VALU-bound.
__device__ __forceinline__ f32x2 gelu_sig_f2(f32x2 x, f32x2 x2) {
  const float CA = -2.f * 1.4426950408889634f * 0.7978845608028654f, CB = CA * 0.044715f;
  const f32x2 z = x * (f32x2{CA, CA} + f32x2{CB, CB} * x2);
  const f32x2 d = f32x2{1.f, 1.f} + f32x2{__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])};
  return f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
__device__ __forceinline__ f32x2 gelu_tanh_f2(f32x2 x) { return x * gelu_sig_f2(x, x * x); }
__device__ __forceinline__ f32x2 dgelu_tanh_f2(f32x2 x) {
  const float D0 = 2.f * 0.7978845608028654f, D1 = 3.f * 0.044715f * D0;
  const f32x2 x2 = x * x;
  const f32x2 s = gelu_sig_f2(x, x2);
  return s + (x * s) * ((f32x2{1.f, 1.f} - s) * (f32x2{D0, D0} + f32x2{D1, D1} * x2));
}
__device__ __forceinline__ f32x4 gelu_tanh_f4(f32x4 x) {
  const f32x2 a = gelu_tanh_f2(f32x2{x[0], x[1]}), b = gelu_tanh_f2(f32x2{x[2], x[3]});
  return f32x4{a[0], a[1], b[0], b[1]};
}
__device__ __forceinline__ f32x4 dgelu_tanh_f4(f32x4 x) {
  const f32x2 a = dgelu_tanh_f2(f32x2{x[0], x[1]}), b = dgelu_tanh_f2(f32x2{x[2], x[3]});
  return f32x4{a[0], a[1], b[0], b[1]};
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
// elementwise launch: cap at 256 CUs x 8 blocks and grid-stride the rest
static inline int ew_grid(int64_t work, int block) {
  int64_t g = (work + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}
