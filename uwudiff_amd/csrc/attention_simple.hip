// Generic-dtype scaled-dot-product attention (forward + backward), VALU only.
//
// This is the exact-fp32 parity path (and the fallback for shapes the MFMA kernel in attention_mfma.hip does
// not cover: query counts that are not multiples of 64, head dim 32, fp32 operands).  Two lanes share one
// row (each owns half of the head dimension, partial dot products are exchanged with one DPP/shuffle), K/V (or
// Q/dO) tiles of 32 rows are staged in LDS as fp32 and read as wave-broadcasts.
// Semantics: F.scaled_dot_product_attention(q,k,v, dropout_p=0, is_causal=False) -- reference
// src/duwu/modules/rope_unet.py:151-153; flash-style backward recomputes P from the saved log-sum-exp.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int ROWS = 64;   // rows per workgroup (128 threads, 2 lanes per row)
constexpr int TILE = 32;   // staged rows per step

struct AttnArgs {
  const void *q, *k, *v, *o, *dO;
  void *out, *dq, *dk, *dv;
  float* lse;
  float* delta;
  const float* kbias;  // optional additive score bias per key, fp32 [B, Tk]
  int B, Tq, Tk, H, d, ldq, ldk, ldv, ldo;
  float scale;
};

template <typename T>
__device__ __forceinline__ void stage_tile(const T* __restrict__ base, int64_t row_stride, int row0, int nrows,
                                           int DH, float* __restrict__ dst) {
  // TILE x DH elements -> fp32 LDS [TILE][DH]; rows >= nrows are zero-filled
  const int chunks = TILE * DH / 4;
  for (int c = threadIdx.x; c < chunks; c += 128) {
    const int r = (c * 4) / DH, col = (c * 4) - r * DH;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row0 + r < nrows) v = load4(base + (int64_t)(row0 + r) * row_stride + col);
    store4(dst + r * DH + col, v);
  }
}

template <typename T, int DH>
__global__ void __launch_bounds__(128) attn_fwd_simple(const AttnArgs a) {
  constexpr int HALF = DH / 2;
  __shared__ __attribute__((aligned(16))) float Ks[TILE * DH];
  __shared__ __attribute__((aligned(16))) float Vs[TILE * DH];
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
  const int r = threadIdx.x >> 1, half = threadIdx.x & 1;
  const int t = blockIdx.x * ROWS + r;
  const bool valid = t < a.Tq;
  const T* q = static_cast<const T*>(a.q) + (int64_t)b * a.Tq * a.ldq + h * DH;
  const T* k = static_cast<const T*>(a.k) + (int64_t)b * a.Tk * a.ldk + h * DH;
  const T* v = static_cast<const T*>(a.v) + (int64_t)b * a.Tk * a.ldv + h * DH;
  float qr[HALF], oa[HALF];
#pragma unroll
  for (int i = 0; i < HALF; i += 4) {
    f32x4 qv = valid ? load4(q + (int64_t)t * a.ldq + half * HALF + i) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      qr[i + e] = qv[e] * a.scale;
      oa[i + e] = 0.f;
    }
  }
  float m = -INFINITY, l = 0.f;
  const float* kbp = a.kbias ? a.kbias + (int64_t)b * a.Tk : nullptr;
  for (int k0 = 0; k0 < a.Tk; k0 += TILE) {
    __syncthreads();
    stage_tile(k, a.ldk, k0, a.Tk, DH, Ks);
    stage_tile(v, a.ldv, k0, a.Tk, DH, Vs);
    __syncthreads();
    float s[TILE];
    float tmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < TILE; ++j) {
      float p = 0.f;
      const float* kr = Ks + j * DH + half * HALF;
#pragma unroll
      for (int i = 0; i < HALF; i += 4) {
        f32x4 kv = load4(kr + i);
        p += qr[i] * kv[0] + qr[i + 1] * kv[1] + qr[i + 2] * kv[2] + qr[i + 3] * kv[3];
      }
      p += __shfl_xor(p, 1, 64);
      s[j] = (k0 + j < a.Tk) ? (kbp ? p + kbp[k0 + j] : p) : -INFINITY;
      tmax = fmaxf(tmax, s[j]);
    }
    const float mn = fmaxf(m, tmax);
    const float alpha = __expf(m - mn);  // m = -inf on the first tile -> 0
    l *= alpha;
#pragma unroll
    for (int i = 0; i < HALF; ++i) oa[i] *= alpha;
#pragma unroll
    for (int j = 0; j < TILE; ++j) {
      const float p = __expf(s[j] - mn);
      l += p;
      const float* vr = Vs + j * DH + half * HALF;
#pragma unroll
      for (int i = 0; i < HALF; i += 4) {
        f32x4 vv = load4(vr + i);
        oa[i] += p * vv[0];
        oa[i + 1] += p * vv[1];
        oa[i + 2] += p * vv[2];
        oa[i + 3] += p * vv[3];
      }
    }
    m = mn;
  }
  if (valid) {
    const float inv = 1.f / l;
    T* o = static_cast<T*>(a.out) + (int64_t)b * a.Tq * a.ldo + h * DH + (int64_t)t * a.ldo + half * HALF;
#pragma unroll
    for (int i = 0; i < HALF; i += 4)
      store4(o + i, f32x4{oa[i] * inv, oa[i + 1] * inv, oa[i + 2] * inv, oa[i + 3] * inv});
    if (half == 0) a.lse[((int64_t)b * a.H + h) * a.Tq + t] = m + __logf(l);
  }
}

// dq (+ delta = rowsum(dO*O)); one lane pair per query row
template <typename T, int DH>
__global__ void __launch_bounds__(128) attn_bwd_dq_simple(const AttnArgs a) {
  constexpr int HALF = DH / 2;
  __shared__ __attribute__((aligned(16))) float Ks[TILE * DH];
  __shared__ __attribute__((aligned(16))) float Vs[TILE * DH];
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
  const int r = threadIdx.x >> 1, half = threadIdx.x & 1;
  const int t = blockIdx.x * ROWS + r;
  const bool valid = t < a.Tq;
  const T* q = static_cast<const T*>(a.q) + (int64_t)b * a.Tq * a.ldq + h * DH;
  const T* k = static_cast<const T*>(a.k) + (int64_t)b * a.Tk * a.ldk + h * DH;
  const T* v = static_cast<const T*>(a.v) + (int64_t)b * a.Tk * a.ldv + h * DH;
  const T* o = static_cast<const T*>(a.o) + (int64_t)b * a.Tq * a.ldo + h * DH;
  const T* dO = static_cast<const T*>(a.dO) + (int64_t)b * a.Tq * a.ldo + h * DH;
  float qr[HALF], dor[HALF], dq[HALF];
  float dl = 0.f;
#pragma unroll
  for (int i = 0; i < HALF; i += 4) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 qv = valid ? load4(q + (int64_t)t * a.ldq + half * HALF + i) : z;
    f32x4 gv = valid ? load4(dO + (int64_t)t * a.ldo + half * HALF + i) : z;
    f32x4 ov = valid ? load4(o + (int64_t)t * a.ldo + half * HALF + i) : z;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      qr[i + e] = qv[e];
      dor[i + e] = gv[e];
      dq[i + e] = 0.f;
      dl += gv[e] * ov[e];
    }
  }
  dl += __shfl_xor(dl, 1, 64);
  const int64_t sidx = ((int64_t)b * a.H + h) * a.Tq + t;
  const float lse = valid ? a.lse[sidx] : 0.f;
  if (valid && half == 0) a.delta[sidx] = dl;
  const float* kbp = a.kbias ? a.kbias + (int64_t)b * a.Tk : nullptr;
  for (int k0 = 0; k0 < a.Tk; k0 += TILE) {
    __syncthreads();
    stage_tile(k, a.ldk, k0, a.Tk, DH, Ks);
    stage_tile(v, a.ldv, k0, a.Tk, DH, Vs);
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < TILE; ++j) {
      const float* kr = Ks + j * DH + half * HALF;
      const float* vr = Vs + j * DH + half * HALF;
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int i = 0; i < HALF; i += 4) {
        f32x4 kv = load4(kr + i), vv = load4(vr + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s += qr[i + e] * kv[e];
          dp += dor[i + e] * vv[e];
        }
      }
      s += __shfl_xor(s, 1, 64);
      dp += __shfl_xor(dp, 1, 64);
      const float p = (k0 + j < a.Tk) ? __expf(s * a.scale + (kbp ? kbp[k0 + j] : 0.f) - lse) : 0.f;
      const float ds = p * (dp - dl);
#pragma unroll
      for (int i = 0; i < HALF; i += 4) {
        f32x4 kv = load4(kr + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) dq[i + e] += ds * kv[e];
      }
    }
  }
  if (valid) {
    T* out = static_cast<T*>(a.dq) + (int64_t)b * a.Tq * a.ldq + h * DH + (int64_t)t * a.ldq + half * HALF;
#pragma unroll
    for (int i = 0; i < HALF; i += 4)
      store4(out + i, f32x4{dq[i] * a.scale, dq[i + 1] * a.scale, dq[i + 2] * a.scale, dq[i + 3] * a.scale});
  }
}

// dk, dv; one lane pair per key row, Q/dO tiles staged
template <typename T, int DH>
__global__ void __launch_bounds__(128) attn_bwd_dkv_simple(const AttnArgs a) {
  constexpr int HALF = DH / 2;
  __shared__ __attribute__((aligned(16))) float Qs[TILE * DH];
  __shared__ __attribute__((aligned(16))) float Gs[TILE * DH];
  __shared__ float lse_s[TILE], del_s[TILE];
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
  const int r = threadIdx.x >> 1, half = threadIdx.x & 1;
  const int t = blockIdx.x * ROWS + r;
  const bool valid = t < a.Tk;
  const T* q = static_cast<const T*>(a.q) + (int64_t)b * a.Tq * a.ldq + h * DH;
  const T* k = static_cast<const T*>(a.k) + (int64_t)b * a.Tk * a.ldk + h * DH;
  const T* v = static_cast<const T*>(a.v) + (int64_t)b * a.Tk * a.ldv + h * DH;
  const T* dO = static_cast<const T*>(a.dO) + (int64_t)b * a.Tq * a.ldo + h * DH;
  float kr[HALF], vr[HALF], dk[HALF], dv[HALF];
#pragma unroll
  for (int i = 0; i < HALF; i += 4) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 kv = valid ? load4(k + (int64_t)t * a.ldk + half * HALF + i) : z;
    f32x4 vv = valid ? load4(v + (int64_t)t * a.ldv + half * HALF + i) : z;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      kr[i + e] = kv[e];
      vr[i + e] = vv[e];
      dk[i + e] = 0.f;
      dv[i + e] = 0.f;
    }
  }
  const int64_t sbase = ((int64_t)b * a.H + h) * a.Tq;
  const float kbv = (a.kbias && valid) ? a.kbias[(int64_t)b * a.Tk + t] : 0.f;
  for (int q0 = 0; q0 < a.Tq; q0 += TILE) {
    __syncthreads();
    stage_tile(q, a.ldq, q0, a.Tq, DH, Qs);
    stage_tile(dO, a.ldo, q0, a.Tq, DH, Gs);
    if (threadIdx.x < TILE) {
      const bool ok = q0 + threadIdx.x < a.Tq;
      lse_s[threadIdx.x] = ok ? a.lse[sbase + q0 + threadIdx.x] : INFINITY;  // exp(s - inf) = 0
      del_s[threadIdx.x] = ok ? a.delta[sbase + q0 + threadIdx.x] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < TILE; ++j) {
      const float* qr = Qs + j * DH + half * HALF;
      const float* gr = Gs + j * DH + half * HALF;
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int i = 0; i < HALF; i += 4) {
        f32x4 qv = load4(qr + i), gv = load4(gr + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s += kr[i + e] * qv[e];
          dp += vr[i + e] * gv[e];
        }
      }
      s += __shfl_xor(s, 1, 64);
      dp += __shfl_xor(dp, 1, 64);
      const float p = __expf(s * a.scale + kbv - lse_s[j]);
      const float ds = p * (dp - del_s[j]);
#pragma unroll
      for (int i = 0; i < HALF; i += 4) {
        f32x4 qv = load4(qr + i), gv = load4(gr + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dv[i + e] += p * gv[e];
          dk[i + e] += ds * qv[e];
        }
      }
    }
  }
  if (valid) {
    T* odk = static_cast<T*>(a.dk) + (int64_t)b * a.Tk * a.ldk + h * DH + (int64_t)t * a.ldk + half * HALF;
    T* odv = static_cast<T*>(a.dv) + (int64_t)b * a.Tk * a.ldv + h * DH + (int64_t)t * a.ldv + half * HALF;
#pragma unroll
    for (int i = 0; i < HALF; i += 4) {
      store4(odk + i, f32x4{dk[i] * a.scale, dk[i + 1] * a.scale, dk[i + 2] * a.scale, dk[i + 3] * a.scale});
      store4(odv + i, f32x4{dv[i], dv[i + 1], dv[i + 2], dv[i + 3]});
    }
  }
}

template <typename T, int DH>
int run_fwd(const AttnArgs& a, hipStream_t st) {
  hipLaunchKernelGGL((attn_fwd_simple<T, DH>), dim3(cdiv(a.Tq, ROWS), a.B * a.H), dim3(128), 0, st, a);
  UWU_LAUNCH_CHECK("attention_fwd(simple)");
  return UWU_OK;
}
template <typename T, int DH>
int run_bwd(const AttnArgs& a, hipStream_t st) {
  hipLaunchKernelGGL((attn_bwd_dq_simple<T, DH>), dim3(cdiv(a.Tq, ROWS), a.B * a.H), dim3(128), 0, st, a);
  UWU_LAUNCH_CHECK("attention_bwd_dq(simple)");
  hipLaunchKernelGGL((attn_bwd_dkv_simple<T, DH>), dim3(cdiv(a.Tk, ROWS), a.B * a.H), dim3(128), 0, st, a);
  UWU_LAUNCH_CHECK("attention_bwd_dkv(simple)");
  return UWU_OK;
}

template <typename T>
int by_head(const AttnArgs& a, bool bwd, hipStream_t st) {
  switch (a.d) {
    case 32: return bwd ? run_bwd<T, 32>(a, st) : run_fwd<T, 32>(a, st);
    case 64: return bwd ? run_bwd<T, 64>(a, st) : run_fwd<T, 64>(a, st);
    case 72: return bwd ? run_bwd<T, 72>(a, st) : run_fwd<T, 72>(a, st);
    default:
      uwu_set_error("attention: head dim %d not instantiated in the generic kernels (32, 64, 72)", a.d);
      return UWU_EINVAL;
  }
}

}  // namespace

static int uwu_attention_simple(const AttnArgs& a, int dtype, bool bwd, hipStream_t st) {
  if (dtype == UWU_F32) return by_head<float>(a, bwd, st);
  if (dtype == UWU_BF16) return by_head<bf16_t>(a, bwd, st);
  uwu_set_error("attention: bad dtype %d", dtype);
  return UWU_EINVAL;
}

// MFMA kernels (attention_mfma.hip)
bool uwu_attn_mfma_fwd_ok(int Tq, int Tk, int d, int ldq, int ldk, int ldv, int ldo);
bool uwu_attn_mfma_bwd_ok(int Tq, int Tk, int d, int ldq, int ldk, int ldv, int ldo);
int uwu_attn_mfma_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const float* kbias, int B,
                      int T, int Tk, int H, int d, int ldq, int ldk, int ldv, int ldo, float scale, hipStream_t st);
int uwu_attn_mfma_bwd(const void* q, const void* k, const void* v, const void* o, const void* dO, const float* lse,
                      float* delta, const float* kbias, void* dq, void* dk, void* dv, int B, int T, int Tk, int H,
                      int d, int ldq, int ldk, int ldv, int ldo, float scale, hipStream_t st);
static bool force_simple() {
  static UwuEnv on("UWU_ATTN_SIMPLE");
  return on.get().is('1');
}

static int check_common(const void* q, const void* k, const void* v, int B, int Tq, int Tk, int H, int d, int ldq,
                        int ldk, int ldv, int ldo, int dtype) {
  UWU_CHECK_ARG(q && k && v, "attention: null pointer");
  UWU_CHECK_ARG(B > 0 && Tq > 0 && Tk > 0 && H > 0 && d > 0, "attention: bad shape");
  UWU_CHECK_ARG(ldq >= H * d && ldk >= H * d && ldv >= H * d && ldo >= H * d, "attention: row stride < H*d");
  const int al = dtype == UWU_BF16 ? 4 : 4;  // 4-element vectors
  UWU_CHECK_ARG(ldq % al == 0 && ldk % al == 0 && ldv % al == 0 && ldo % al == 0 && d % 8 == 0,
                "attention: strides must be multiples of 4 and d a multiple of 8");
  const uintptr_t m = dtype == UWU_BF16 ? 7 : 15;
  UWU_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & m) == 0, "attention: misaligned q/k/v");
  return UWU_OK;
}

static int attention_fwd_impl(const void* q, const void* k, const void* v, void* o, float* lse, const float* kbias,
                              int B, int Tq, int Tk, int H, int d, int ldq, int ldk, int ldv, int ldo, float scale,
                              int dtype, void* stream) {
  int rc = check_common(q, k, v, B, Tq, Tk, H, d, ldq, ldk, ldv, ldo, dtype);
  if (rc) return rc;
  UWU_CHECK_ARG(o && lse, "attention_fwd: null output");
  UWU_CHECK_ARG(scale > 0.f, "attention: scale must be positive");
  if (dtype == UWU_BF16 && !force_simple() && uwu_attn_mfma_fwd_ok(Tq, Tk, d, ldq, ldk, ldv, ldo) &&
      (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) == 0) {
    UwuProfScope prof(stream);
    rc = uwu_attn_mfma_fwd(q, k, v, o, lse, kbias, B, Tq, Tk, H, d, ldq, ldk, ldv, ldo, scale, (hipStream_t)stream);
    // algorithmic work: QK^T + PV = 4 Tq Tk d per head; q, o and k, v once (bf16)
    prof.done(UWU_PROF_ATTN_FWD, 0, 4.0 * B * H * Tq * Tk * d, 2.0 * B * H * d * (2.0 * Tq + 2.0 * Tk));
    return rc;
  }
  AttnArgs a{};
  a.q = q; a.k = k; a.v = v; a.out = o; a.lse = lse; a.kbias = kbias;
  a.B = B; a.Tq = Tq; a.Tk = Tk; a.H = H; a.d = d; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.scale = scale;
  return uwu_attention_simple(a, dtype, false, (hipStream_t)stream);
}

static int attention_bwd_impl(const void* q, const void* k, const void* v, const void* o, const void* dO,
                              const float* lse, float* delta, const float* kbias, void* dq, void* dk, void* dv, int B,
                              int Tq, int Tk, int H, int d, int ldq, int ldk, int ldv, int ldo, float scale, int dtype,
                              void* stream) {
  int rc = check_common(q, k, v, B, Tq, Tk, H, d, ldq, ldk, ldv, ldo, dtype);
  if (rc) return rc;
  UWU_CHECK_ARG(o && dO && lse && delta && dq && dk && dv, "attention_bwd: null pointer");
  UWU_CHECK_ARG(scale > 0.f, "attention: scale must be positive");
  if (dtype == UWU_BF16 && !force_simple() && uwu_attn_mfma_bwd_ok(Tq, Tk, d, ldq, ldk, ldv, ldo) &&
      (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)dO | (uintptr_t)dq | (uintptr_t)dk |
        (uintptr_t)dv) & 15) == 0) {
    UwuProfScope prof(stream);
    rc = uwu_attn_mfma_bwd(q, k, v, o, dO, lse, delta, kbias, dq, dk, dv, B, Tq, Tk, H, d, ldq, ldk, ldv, ldo, scale,
                           (hipStream_t)stream);
    // five products (S, dP, dV, dK, dQ) = 10 Tq Tk d per head; q, o, dO, dq and k, v, dk, dv once (bf16)
    prof.done(UWU_PROF_ATTN_BWD, 0, 10.0 * B * H * Tq * Tk * d, 2.0 * B * H * d * (4.0 * Tq + 4.0 * Tk));
    return rc;
  }
  AttnArgs a{};
  a.q = q; a.k = k; a.v = v; a.o = o; a.dO = dO; a.lse = const_cast<float*>(lse); a.delta = delta; a.dq = dq; a.dk = dk; a.dv = dv;
  a.kbias = kbias;
  a.B = B; a.Tq = Tq; a.Tk = Tk; a.H = H; a.d = d; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.scale = scale;
  return uwu_attention_simple(a, dtype, true, (hipStream_t)stream);
}

extern "C" int uwu_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int Tq,
                                 int Tk, int H, int d, int ldq, int ldk, int ldv, int ldo, float scale, int dtype,
                                 void* stream) {
  return attention_fwd_impl(q, k, v, o, lse, nullptr, B, Tq, Tk, H, d, ldq, ldk, ldv, ldo, scale, dtype, stream);
}

extern "C" int uwu_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* dO,
                                 const float* lse, float* delta, void* dq, void* dk, void* dv, int B, int Tq, int Tk,
                                 int H, int d, int ldq, int ldk, int ldv, int ldo, float scale, int dtype,
                                 void* stream) {
  return attention_bwd_impl(q, k, v, o, dO, lse, delta, nullptr, dq, dk, dv, B, Tq, Tk, H, d, ldq, ldk, ldv, ldo,
                            scale, dtype, stream);
}

extern "C" int uwu_attention_bias_fwd(const void* q, const void* k, const void* v, const float* key_bias, void* o,
                                      float* lse, int B, int Tq, int Tk, int H, int d, int ldq, int ldk, int ldv,
                                      int ldo, float scale, int dtype, void* stream) {
  UWU_CHECK_ARG(key_bias, "attention_bias_fwd: null key_bias");
  return attention_fwd_impl(q, k, v, o, lse, key_bias, B, Tq, Tk, H, d, ldq, ldk, ldv, ldo, scale, dtype, stream);
}

extern "C" int uwu_attention_bias_bwd(const void* q, const void* k, const void* v, const float* key_bias,
                                      const void* o, const void* dO, const float* lse, float* delta, void* dq,
                                      void* dk, void* dv, int B, int Tq, int Tk, int H, int d, int ldq, int ldk,
                                      int ldv, int ldo, float scale, int dtype, void* stream) {
  UWU_CHECK_ARG(key_bias, "attention_bias_bwd: null key_bias");
  return attention_bwd_impl(q, k, v, o, dO, lse, delta, key_bias, dq, dk, dv, B, Tq, Tk, H, d, ldq, ldk, ldv, ldo,
                            scale, dtype, stream);
}
