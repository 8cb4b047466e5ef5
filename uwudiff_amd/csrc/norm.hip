// adaLN-Zero pre-norm kernels (HBM-bound; one wavefront per token row, shuffle reductions).
//   fwd:  x_out = x_in + gate_b*y ; h = LN(x_out)*(1+scale_b)+shift_b          (rope_unet.py:306-309,344-349,393-411)
//   bwd:  dx_out = dx_in + LN_bwd(dh*(1+scale_b)) ; dy = gate_b*dx_out ; per-sample dshift/dscale/dgate
// plus column sums for bias gradients.
#include "common.h"

namespace {
// streaming (nontemporal) 16-byte accesses for tensors a kernel touches once
__device__ __forceinline__ f32x8 load8_nt(const float* p) { return load8(p); }
__device__ __forceinline__ f32x8 load8_nt(const bf16_t* p) {
  typedef unsigned nu32x4 __attribute__((ext_vector_type(4)));
  nu32x4 raw = __builtin_nontemporal_load(reinterpret_cast<const nu32x4*>(p));
  bf16x8 v = *reinterpret_cast<bf16x8*>(&raw);
  f32x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (float)v[i];
  return r;
}
__device__ __forceinline__ void store8_nt(float* p, f32x8 v) { store8(p, v); }
__device__ __forceinline__ void store8_nt(bf16_t* p, f32x8 v) {
  typedef unsigned nu32x4 __attribute__((ext_vector_type(4)));
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
  __builtin_nontemporal_store(*reinterpret_cast<nu32x4*>(&o), reinterpret_cast<nu32x4*>(p));
}

// the same 8 elements kept in their storage type (bf16: 4 registers instead of 8) until they are needed as fp32
template <typename T> struct Raw8;
template <> struct Raw8<float> { typedef f32x8 type; };
template <> struct Raw8<bf16_t> { typedef bf16x8 type; };
__device__ __forceinline__ f32x8 raw8_nt(const float* p) { return load8(p); }
__device__ __forceinline__ bf16x8 raw8_nt(const bf16_t* p) {
  typedef unsigned nu32x4 __attribute__((ext_vector_type(4)));
  nu32x4 raw = __builtin_nontemporal_load(reinterpret_cast<const nu32x4*>(p));
  return *reinterpret_cast<bf16x8*>(&raw);
}
__device__ __forceinline__ f32x8 raw8(const float* p) { return load8(p); }
__device__ __forceinline__ bf16x8 raw8(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ f32x8 unpack8(f32x8 v) { return v; }
__device__ __forceinline__ f32x8 unpack8(bf16x8 v) {
  f32x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (float)v[i];
  return r;
}

constexpr int MAX_D = 4096;  // 16-byte vectors: 8 elements per lane, NIT = ceil(D/512) (template parameter)

template <typename T, int MAX_IT>
__global__ void __launch_bounds__(256) add_ln_mod_fwd_kernel(const T* __restrict__ x_in, const T* __restrict__ y,
                                                             const float* __restrict__ gate,
                                                             const float* __restrict__ shift,
                                                             const float* __restrict__ scale, int mod_ld,
                                                             T* __restrict__ x_out, T* __restrict__ h,
                                                             float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                             int M, int T_tok, int D, float eps, int affine) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nit = (D + 511) >> 9;
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    const int b = row / T_tok;
    const int64_t off = (int64_t)row * D;
    f32x8 v[MAX_IT];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
      const int d = it * 512 + lane * 8;
      if (it < nit && d < D) {
        // x_in / y are read once and x_out is next read by the following LayerNorm, two GEMMs later: streaming
        // (nontemporal) accesses for the three, so that they do not displace h, which the next GEMM reads right away
        // (M = 196608: 150 -> 139 us)
        f32x8 xv = load8_nt(x_in + off + d);
        if (y) {
          f32x8 yv = load8_nt(y + off + d), gv = load8(gate + (int64_t)b * mod_ld + d);
          xv = xv + gv * yv;
          // keep the stored residual stream and the normalised value consistent in reduced precision
          if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[e] = (float)(bf16_t)xv[e];
          }
          store8_nt(x_out + off + d, xv);
        }
        v[it] = xv;
        _Pragma("unroll") for (int e = 0; e < 8; ++e) s += xv[e];
      } else {
        v[it] = f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
      const int d = it * 512 + lane * 8;
      if (it < nit && d < D) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float c = v[it][e] - mean;
          q += c * c;
        }
      }
    }
    const float var = wave_sum(q) / (float)D;
    const float rstd = 1.f / sqrtf(var + eps);
    if (lane == 0) {
      mean_o[row] = mean;
      rstd_o[row] = rstd;
    }
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
      const int d = it * 512 + lane * 8;
      if (it < nit && d < D) {
        f32x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[it][e] - mean) * rstd;
        if (scale) {
          f32x8 sc = load8(scale + (int64_t)b * mod_ld + d), sh = load8(shift + (int64_t)b * mod_ld + d);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = o[e] * (affine ? sc[e] : 1.f + sc[e]) + sh[e];
        }
        store8(h + off + d, o);
      }
    }
  }
}

// D a multiple of 128 (every DiT width): FOUR rows per wave -- a row belongs to 16 lanes, NCH = D / 128 chunks of 8 elements per
// lane (chunk c of lane l16 = elements 128 c + 8 l16 ..: one load instruction moves 256 contiguous bytes of each of the 4 rows).
// The kernel above keeps one 768-byte row per wave in flight (D = 384: 48 of 64 lanes busy, 1.5 KB of loads per wave and two
// dependent whole-wave reductions per row): 604 MB in 149 us = 4.0 TB/s at M = 196608 while the backward streams at 5.6.
// Here a wave has 6 KB of loads in flight, the statistics are four-step DPP sums inside a 16-lane row (row16_sum), and the
// instruction stream per row shrinks 4x.
template <typename T, int NCH>
__global__ void __launch_bounds__(256) add_ln_mod_fwd16_kernel(const T* __restrict__ x_in, const T* __restrict__ y,
                                                               const float* __restrict__ gate,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ scale, int mod_ld,
                                                               T* __restrict__ x_out, T* __restrict__ h,
                                                               float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                               int M, int T_tok, float eps, int affine) {
  constexpr int D = 128 * NCH;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane >> 4, l16 = lane & 15;
  for (int row0 = (blockIdx.x * 4 + wave) * 4; row0 < M; row0 += gridDim.x * 16) {
    const int row = row0 + sub;
    const bool ok = row < M;
    const int rc = ok ? row : M - 1;  // (every lane takes part in the DPP sums: out-of-range rows recompute the last one)
    const int b = rc / T_tok;
    const int64_t off = (int64_t)rc * D + 8 * l16;
    const float* gb = gate + (int64_t)b * mod_ld + 8 * l16;
    f32x8 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int cidx = 0; cidx < NCH; ++cidx) {
      f32x8 xv = load8_nt(x_in + off + 128 * cidx);
      if (y) {
        const f32x8 yv = load8_nt(y + off + 128 * cidx), gv = load8(gb + 128 * cidx);
        xv = xv + gv * yv;
        if constexpr (sizeof(T) == 2) {  // keep the stored residual stream and the normalised value consistent
#pragma unroll
          for (int e = 0; e < 8; ++e) xv[e] = (float)(bf16_t)xv[e];
        }
        if (ok) store8_nt(x_out + off + 128 * cidx, xv);
      }
      v[cidx] = xv;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += xv[e];
    }
    const float mean = row16_sum(s) * (1.f / (float)D);
    float q = 0.f;
#pragma unroll
    for (int cidx = 0; cidx < NCH; ++cidx)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float c = v[cidx][e] - mean;
        q += c * c;
      }
    const float var = row16_sum(q) * (1.f / (float)D);
    const float rstd = 1.f / sqrtf(var + eps);
    if (l16 == 0 && ok) {
      mean_o[row] = mean;
      rstd_o[row] = rstd;
    }
#pragma unroll
    for (int cidx = 0; cidx < NCH; ++cidx) {
      f32x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[cidx][e] - mean) * rstd;
      if (scale) {
        const f32x8 sc = load8(scale + (int64_t)b * mod_ld + 8 * l16 + 128 * cidx), sh = load8(shift + (int64_t)b * mod_ld + 8 * l16 + 128 * cidx);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = o[e] * (affine ? sc[e] : 1.f + sc[e]) + sh[e];
      }
      if (ok) store8(h + off + 128 * cidx, o);
    }
  }
}

// ---- fp8 mode (BASELINE config 5), delayed scaling: the LayerNorm output leaves as the fp8 operand of the next GEMMs -------
// h is consumed only by a Linear (row-major e4m3, contraction over D) and by that Linear's weight gradient (transposed,
// contraction over the tokens), so instead of bf16 h + a quantising pass (quant.hip: 2 + 2 bytes per element) this kernel writes
// both fp8 images itself.  One workgroup = 64 consecutive rows, 8 waves x 8 rows, the loads of the next row in flight while a
// row is reduced.  A lane owns 8 columns per 512-column pass: its 8 bytes go to global memory (row-major image) and into an
// LDS copy of the tile [64 rows][D] (8-byte writes, conflict-free).  After the last row a thread takes 4 columns x 16 rows of
// that copy -- 16 dwords, consecutive lanes consecutive dwords -- transposes the four 4 x 4 byte blocks with v_perm_b32 and
// stores 16 contiguous bytes of each of its 4 columns of the transposed image (64-byte runs along the tokens per tile).
__device__ __forceinline__ unsigned q8_pack4(float a, float b, float c, float d, float s) {
  a = fminf(fmaxf(a * s, -448.f), 448.f);
  b = fminf(fmaxf(b * s, -448.f), 448.f);
  c = fminf(fmaxf(c * s, -448.f), 448.f);
  d = fminf(fmaxf(d * s, -448.f), 448.f);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
}

template <int MAX_IT, int NW>  // NW waves x 8 rows = one tile
__global__ void __launch_bounds__(64 * NW, 4) add_ln_mod_fwd_q8_kernel(
    const bf16_t* __restrict__ x_in, const bf16_t* __restrict__ y, const float* __restrict__ gate,
    const float* __restrict__ shift, const float* __restrict__ scale, int mod_ld, bf16_t* __restrict__ x_out,
    unsigned char* __restrict__ q8, int ldq, unsigned char* __restrict__ q8t, int ldqt, const float* __restrict__ q_scale,
    float* __restrict__ q_amax, float* __restrict__ mean_o, float* __restrict__ rstd_o, int M, int T_tok, int D, float eps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char img[];  // [8 NW][D + 8] fp8 bytes
  __shared__ float red[NW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nit = (D + 511) >> 9;
  const int row0 = blockIdx.x * 8 * NW;
  const int pitch = D + 8;  // (rows 16 apart: banks 32 apart)
  const float qs = q_scale[0];
  float mx = 0.f;
  // the loads of row r + 1 are issued before row r is reduced (two rows in flight per wave: with one, the dependent chain
  // load -> two wave reductions -> store of each row left the 16 waves of a CU short of the bytes in flight HBM needs)
  // (y has one buffer: it is consumed by the first pass over a row, and the next row is fetched right after that pass)
  bf16x8 xr[2][MAX_IT], yr[MAX_IT];
  auto fetch = [&](int r, int buf) {
    const int64_t off = (int64_t)(row0 + 8 * wave + r) * D;
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
      const int d = it * 512 + lane * 8;
      if (it < nit && d < D) {
        xr[buf][it] = raw8_nt(x_in + off + d);
        if (y) yr[it] = raw8_nt(y + off + d);
      }
    }
  };
  fetch(0, 0);
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int buf = r & 1;
    const int row = row0 + 8 * wave + r;
    const int b = row / T_tok;
    const int64_t off = (int64_t)row * D;
    bf16x8 v[MAX_IT];  // the residual-stream values (exactly bf16) stay packed between the passes
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
      const int d = it * 512 + lane * 8;
      if (it < nit && d < D) {
        f32x8 xv = unpack8(xr[buf][it]);
        v[it] = xr[buf][it];
        if (y) {
          const f32x8 yv = unpack8(yr[it]), gv = load8(gate + (int64_t)b * mod_ld + d);
          xv = xv + gv * yv;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[it][e] = (bf16_t)xv[e];
          typedef unsigned nu32x4 __attribute__((ext_vector_type(4)));
          __builtin_nontemporal_store(*reinterpret_cast<nu32x4*>(&v[it]), reinterpret_cast<nu32x4*>(x_out + off + d));
          xv = unpack8(v[it]);
        }
        _Pragma("unroll") for (int e = 0; e < 8; ++e) s += xv[e];
      }
    }
    if (r + 1 < 8) fetch(r + 1, buf ^ 1);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
      const int d = it * 512 + lane * 8;
      if (it < nit && d < D) {
        const f32x8 xv = unpack8(v[it]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float c = xv[e] - mean;
          q += c * c;
        }
      }
    }
    const float var = wave_sum(q) / (float)D;
    const float rstd = 1.f / sqrtf(var + eps);
    if (lane == 0) {
      mean_o[row] = mean;
      rstd_o[row] = rstd;
    }
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
      const int d = it * 512 + lane * 8;
      if (it < nit && d < D) {
        f32x8 o;
        const f32x8 sc = load8(scale + (int64_t)b * mod_ld + d), sh = load8(shift + (int64_t)b * mod_ld + d);
        const f32x8 xv = unpack8(v[it]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o[e] = (xv[e] - mean) * rstd * (1.f + sc[e]) + sh[e];
          mx = fmaxf(mx, fabsf(o[e]));
        }
        const unsigned lo = q8_pack4(o[0], o[1], o[2], o[3], qs), hi = q8_pack4(o[4], o[5], o[6], o[7], qs);
        *reinterpret_cast<uint2*>(q8 + (int64_t)row * ldq + d) = uint2{lo, hi};
        *reinterpret_cast<uint2*>(img + (8 * wave + r) * pitch + d) = uint2{lo, hi};
      }
    }
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  if (q_amax && threadIdx.x == 0) {
    float a = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) a = fmaxf(a, red[w]);
    const unsigned cur = __hip_atomic_load(reinterpret_cast<unsigned*>(q_amax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__float_as_uint(a) > cur) atomicMax(reinterpret_cast<unsigned*>(q_amax), __float_as_uint(a));
  }
  // item = (column quad cq, row block rb): rows 16 rb .. + 15, columns 4 cq .. + 3; neighbouring lanes = the row blocks of one
  // column quad, so that their 16-byte stores form one contiguous run of each column
  constexpr int RB = NW / 2;
  const int nq = D >> 2;
  for (int idx = threadIdx.x; idx < RB * nq; idx += 64 * NW) {
    const int cq = idx / RB, rb = idx - cq * RB;
    const unsigned* src = reinterpret_cast<const unsigned*>(img + (16 * rb) * pitch) + cq;
    unsigned w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = src[i * (pitch >> 2)];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned sel = 0x0c0c0400u + (unsigned)e * 0x0101u;
      uint4 o;
      o.x = __builtin_amdgcn_perm(w[1], w[0], sel) | (__builtin_amdgcn_perm(w[3], w[2], sel) << 16);
      o.y = __builtin_amdgcn_perm(w[5], w[4], sel) | (__builtin_amdgcn_perm(w[7], w[6], sel) << 16);
      o.z = __builtin_amdgcn_perm(w[9], w[8], sel) | (__builtin_amdgcn_perm(w[11], w[10], sel) << 16);
      o.w = __builtin_amdgcn_perm(w[13], w[12], sel) | (__builtin_amdgcn_perm(w[15], w[14], sel) << 16);
      *reinterpret_cast<uint4*>(q8t + (int64_t)(4 * cq + e) * ldqt + row0 + 16 * rb) = o;
    }
  }
}

// One workgroup = ROWS consecutive rows of ONE sample (T_tok % ROWS == 0); the 4 waves split the rows, keep the
// per-column partial sums for dshift/dscale/dgate in registers, fold them through LDS and issue one fp32 atomic
// per column per workgroup.
template <typename T, int MAX_IT, int NW = 4>
__global__ void __launch_bounds__(64 * NW) add_ln_mod_bwd_kernel(
    const T* __restrict__ dh, const T* __restrict__ x, const float* __restrict__ mean_i,
    const float* __restrict__ rstd_i, const float* __restrict__ scale, const T* __restrict__ dx_in,
    const T* __restrict__ y, const float* __restrict__ gate, int mod_ld, T* __restrict__ dx_out, T* __restrict__ dy,
    float* __restrict__ dshift, float* __restrict__ dscale, float* __restrict__ dgate, int M, int T_tok, int D,
    int rows_per_block, int affine) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [NW waves][3][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nit = (D + 511) >> 9;
  const int row0 = blockIdx.x * rows_per_block;
  const int b = row0 / T_tok;
  f32x8 a_sh[MAX_IT], a_sc[MAX_IT], a_g[MAX_IT];
#pragma unroll
  for (int it = 0; it < MAX_IT; ++it) a_sh[it] = a_sc[it] = a_g[it] = f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  for (int r = wave; r < rows_per_block; r += NW) {
    const int row = row0 + r;
    if (row >= M) break;
    const int64_t off = (int64_t)row * D;
    const float mean = mean_i[row], rstd = rstd_i[row];
    if constexpr (MAX_IT == 1) {  // D <= 512: everything fits in registers as fp32 (88-92 VGPRs)
      f32x8 xh[MAX_IT], g[MAX_IT];
      float s1 = 0.f, s2 = 0.f;
  #pragma unroll
      for (int it = 0; it < MAX_IT; ++it) {
        const int d = it * 512 + lane * 8;
        if (it < nit && d < D) {
          // streaming accesses for what is read once here (x, y, dx_in) and for dx_out (next read two GEMMs later); dh was
          // just written by the previous GEMM and dy feeds the next one (M = 196608: 189 -> 177 us)
          f32x8 xv = load8_nt(x + off + d), dv = load8(dh + off + d);
          f32x8 sc = scale ? load8(scale + (int64_t)b * mod_ld + d)
                           : (affine ? f32x8{1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f} : f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
  #pragma unroll
          for (int e = 0; e < 8; ++e) {
            float xhat = (xv[e] - mean) * rstd;
            float gg = dv[e] * (affine ? sc[e] : 1.f + sc[e]);
            xh[it][e] = xhat;
            g[it][e] = gg;
            s1 += gg;
            s2 += gg * xhat;
            a_sh[it][e] += dv[e];
            a_sc[it][e] += dv[e] * xhat;
          }
        }
      }
      s1 = wave_sum(s1) / (float)D;
      s2 = wave_sum(s2) / (float)D;
  #pragma unroll
      for (int it = 0; it < MAX_IT; ++it) {
        const int d = it * 512 + lane * 8;
        if (it < nit && d < D) {
          f32x8 dx;
  #pragma unroll
          for (int e = 0; e < 8; ++e) dx[e] = rstd * (g[it][e] - s1 - xh[it][e] * s2);
          if (dx_in) dx = dx + load8_nt(dx_in + off + d);
          store8_nt(dx_out + off + d, dx);
          if (y) {
            f32x8 yv = load8_nt(y + off + d), gv = load8(gate + (int64_t)b * mod_ld + d);
            store8(dy + off + d, gv * dx);
  #pragma unroll
            for (int e = 0; e < 8; ++e) a_g[it][e] += dx[e] * yv[e];
          }
        }
      }
    } else {
      // Wide rows (MAX_IT >= 2): x and dh stay in their storage type between the two passes and the modulation vector is read
      // again (L1 / L2) -- with fp32 copies of xhat, g and the scale the D = 1152 instantiation needed 194 registers (2 waves
      // per SIMD: too few loads in flight for an HBM-bound kernel)
      typename Raw8<T>::type xr[MAX_IT], dr[MAX_IT];
      float s1 = 0.f, s2 = 0.f;
      auto scale_of = [&](int d) {
        return scale ? load8(scale + (int64_t)b * mod_ld + d)
                     : (affine ? f32x8{1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f} : f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
      };
  #pragma unroll
      for (int it = 0; it < MAX_IT; ++it) {
        const int d = it * 512 + lane * 8;
        if (it < nit && d < D) {
          // streaming accesses for what is read once here (x, y, dx_in) and for dx_out (next read two GEMMs later); dh was
          // just written by the previous GEMM and dy feeds the next one (M = 196608: 189 -> 177 us)
          xr[it] = raw8_nt(x + off + d);
          dr[it] = raw8(dh + off + d);
          const f32x8 xv = unpack8(xr[it]), dv = unpack8(dr[it]), sc = scale_of(d);
  #pragma unroll
          for (int e = 0; e < 8; ++e) {
            float xhat = (xv[e] - mean) * rstd;
            float gg = dv[e] * (affine ? sc[e] : 1.f + sc[e]);
            s1 += gg;
            s2 += gg * xhat;
            a_sh[it][e] += dv[e];
            a_sc[it][e] += dv[e] * xhat;
          }
        }
      }
      s1 = wave_sum(s1) / (float)D;
      s2 = wave_sum(s2) / (float)D;
  #pragma unroll
      for (int it = 0; it < MAX_IT; ++it) {
        const int d = it * 512 + lane * 8;
        if (it < nit && d < D) {
          const f32x8 xv = unpack8(xr[it]), dv = unpack8(dr[it]), sc = scale_of(d);
          f32x8 dx;
  #pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xhat = (xv[e] - mean) * rstd, gg = dv[e] * (affine ? sc[e] : 1.f + sc[e]);
            dx[e] = rstd * (gg - s1 - xhat * s2);
          }
          if (dx_in) dx = dx + load8_nt(dx_in + off + d);
          store8_nt(dx_out + off + d, dx);
          if (y) {
            f32x8 yv = load8_nt(y + off + d), gv = load8(gate + (int64_t)b * mod_ld + d);
            store8(dy + off + d, gv * dx);
  #pragma unroll
            for (int e = 0; e < 8; ++e) a_g[it][e] += dx[e] * yv[e];
          }
        }
      }
    }
  }
  // fold the waves' column partials.  D <= 1024: one LDS slab per wave, summed by the flushing threads.  Wider rows: ONE slab,
  // the waves add to it in turn (a slab per wave = 55 KB at D = 1152: two workgroups per CU, 2 waves per SIMD, 3.4 TB/s)
  constexpr bool ONE_SLAB = MAX_IT >= 3;
  float* mine = red + (ONE_SLAB ? 0 : (int64_t)wave * 3 * D);
  if constexpr (ONE_SLAB) {
    for (int w = 0; w < NW; ++w) {
      if (wave == w) {
#pragma unroll
        for (int it = 0; it < MAX_IT; ++it) {
          const int d = it * 512 + lane * 8;
          if (it < nit && d < D) {
            if (w == 0) {
              store8(mine + d, a_sh[it]);
              store8(mine + D + d, a_sc[it]);
              store8(mine + 2 * D + d, a_g[it]);
            } else {
              store8(mine + d, load8(mine + d) + a_sh[it]);
              store8(mine + D + d, load8(mine + D + d) + a_sc[it]);
              store8(mine + 2 * D + d, load8(mine + 2 * D + d) + a_g[it]);
            }
          }
        }
      }
      __syncthreads();
    }
  } else {
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
      const int d = it * 512 + lane * 8;
      if (it < nit && d < D) {
        store8(mine + d, a_sh[it]);
        store8(mine + D + d, a_sc[it]);
        store8(mine + 2 * D + d, a_g[it]);
      }
    }
    __syncthreads();
  }
  const int ncol = (y ? 3 : 2) * D;
  for (int c = threadIdx.x; c < ncol; c += 64 * NW) {
    float t = 0.f;
    if constexpr (ONE_SLAB) {
      t = red[c];
    } else {
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red[w * 3 * D + c];
    }
    const int which = c / D, d = c - which * D;
    float* dst = which == 0 ? dshift : (which == 1 ? dscale : dgate);
    if (dst) atomicAdd(dst + (int64_t)b * mod_ld + d, t);
  }
}

// LayerNorm backward with ONE weight / bias vector for all rows (the UNet's transformer blocks; the adaLN form above has a
// shift / scale per sample).  With per-sample vectors the column sums of a workgroup go to B x D different addresses;
// here every workgroup adds to the same 2 D floats, and the kernel above -- few rows per workgroup to fill the chip --
// issued 1536 x 2560 global atomics at M = 6144, D = 1280 (50 us for 63 MB of traffic; 16 rows per workgroup: 31 us).
// This one: 8 waves x 4..8 rows per workgroup, one global atomic per column per workgroup.
template <typename T, int MAX_IT>
__global__ void __launch_bounds__(512) ln_affine_bwd_kernel(
    const T* __restrict__ dh, const T* __restrict__ x, const float* __restrict__ mean_i,
    const float* __restrict__ rstd_i, const float* __restrict__ weight, const T* __restrict__ dx_in,
    T* __restrict__ dx_out, float* __restrict__ dbias, float* __restrict__ dweight, int M, int D, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [8 waves][2][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nit = (D + 511) >> 9;
  const int row0 = blockIdx.x * rows_per_block;
  f32x8 a_sh[MAX_IT], a_sc[MAX_IT], wv[MAX_IT];
#pragma unroll
  for (int it = 0; it < MAX_IT; ++it) {
    a_sh[it] = a_sc[it] = f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int d = it * 512 + lane * 8;
    wv[it] = (it < nit && d < D) ? load8(weight + d) : a_sh[it];
  }
  for (int r = wave; r < rows_per_block; r += 8) {
    const int row = row0 + r;
    if (row >= M) break;
    const int64_t off = (int64_t)row * D;
    const float mean = mean_i[row], rstd = rstd_i[row];
    f32x8 xh[MAX_IT], g[MAX_IT], din[MAX_IT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
      const int d = it * 512 + lane * 8;
      if (it < nit && d < D) {
        const f32x8 xv = load8_nt(x + off + d), dv = load8(dh + off + d);
        if (dx_in) din[it] = load8_nt(dx_in + off + d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xhat = (xv[e] - mean) * rstd, gg = dv[e] * wv[it][e];
          xh[it][e] = xhat;
          g[it][e] = gg;
          s1 += gg;
          s2 += gg * xhat;
          a_sh[it][e] += dv[e];
          a_sc[it][e] += dv[e] * xhat;
        }
      }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
      const int d = it * 512 + lane * 8;
      if (it < nit && d < D) {
        f32x8 dx;
#pragma unroll
        for (int e = 0; e < 8; ++e) dx[e] = rstd * (g[it][e] - s1 - xh[it][e] * s2);
        if (dx_in) dx = dx + din[it];
        store8_nt(dx_out + off + d, dx);
      }
    }
  }
  // fold the 8 waves' column sums (private [2][D] slabs, then 512 threads add the slabs: consecutive lanes, consecutive
  // columns -- coalesced global atomics; LDS float atomics into one slab measured 3x slower than the whole old kernel)
  float* mine = red + (int64_t)wave * 2 * D;
#pragma unroll
  for (int it = 0; it < MAX_IT; ++it) {
    const int d = it * 512 + lane * 8;
    if (it < nit && d < D) {
      store8(mine + d, a_sh[it]);
      store8(mine + D + d, a_sc[it]);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D; c += 512) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) t += red[w * 2 * D + c];
    atomicAdd(c < D ? dbias + c : dweight + (c - D), t);
  }
}

// out[n] += sum_m X[m,n]: 16 row-lanes x 16 column-groups(4 cols) per workgroup, rows split over gridDim.y
template <typename T>
__global__ void __launch_bounds__(256) colsum_kernel(const T* __restrict__ X, int M, int N, int ldx,
                                                     float* __restrict__ out, int rows_per_block) {
  __shared__ f32x4 red[16][17];
  const int cg = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int n = blockIdx.x * 64 + cg * 4;
  X += (int64_t)blockIdx.z * M * ldx;  // batched form: gridDim.z independent [M, N] slabs -> out[z, :]
  out += (int64_t)blockIdx.z * N;
  const int r0 = blockIdx.y * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  if (n < N)
    for (int r = r0 + rg; r < r1; r += 16) a = a + load4(X + (int64_t)r * ldx + n);
  red[rg][cg] = a;
  __syncthreads();
  if (rg == 0 && n < N) {
    f32x4 t = red[0][cg];
#pragma unroll
    for (int i = 1; i < 16; ++i) t = t + red[i][cg];
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(out + n + e, t[e]);
  }
}

}  // namespace

extern "C" int uwu_add_ln_modulate_fwd_q8(const void* x_in, const void* y, const float* gate, const float* shift,
                                          const float* scale, int mod_ld, void* x_out, void* q8, int ldq, void* q8t, int ldqt,
                                          const float* q_scale, float* q_amax, float* mean, float* rstd, int B, int T, int D,
                                          float eps, void* stream) {
  UWU_CHECK_ARG(x_in && q8 && q8t && q_scale && mean && rstd && shift && scale, "add_ln_modulate_fwd_q8: null pointer");
  UWU_CHECK_ARG(B > 0 && T > 0 && D > 0 && D % 8 == 0 && D <= 1536, "add_ln_modulate_fwd_q8: D=%d unsupported", D);
  UWU_CHECK_ARG(((int64_t)B * T) % 64 == 0, "add_ln_modulate_fwd_q8: B*T must be a multiple of 64");
  UWU_CHECK_ARG((y == nullptr) || (gate && x_out), "add_ln_modulate_fwd_q8: y needs gate and x_out");
  UWU_CHECK_ARG(mod_ld % 4 == 0, "add_ln_modulate_fwd_q8: mod_ld must be a multiple of 4");
  UWU_CHECK_ARG((((uintptr_t)x_in | (uintptr_t)y | (uintptr_t)x_out | (uintptr_t)q8t) & 15) == 0 && ((uintptr_t)q8 & 7) == 0,
                "add_ln_modulate_fwd_q8: misaligned tensor");
  const int M = B * T;
  UWU_CHECK_ARG(ldq >= D && ldq % 8 == 0 && ldqt >= M && ldqt % 16 == 0, "add_ln_modulate_fwd_q8: bad leading dimension");
  hipStream_t st = (hipStream_t)stream;
  // tile = 8 rows per wave: 8 waves (64-byte runs of the transposed image) while two such workgroups fit a CU's LDS, else 4
  static UwuEnv nw_e("UWU_LN_Q8_NW");
  int nw = (size_t)64 * (D + 8) * 2 <= 150 * 1024 ? 8 : 4;
  if (nw_e.get().set && (nw_e.get().ival == 4 || nw_e.get().ival == 8)) nw = nw_e.get().ival;
  UWU_CHECK_ARG(M % (8 * nw) == 0, "add_ln_modulate_fwd_q8: B*T must be a multiple of %d", 8 * nw);
  const size_t lds = (size_t)8 * nw * (D + 8);
  UwuProfScope prof(stream);
#define Q8_LAUNCH(NIT, NW)                                                                                                 \
  {                                                                                                                        \
    static bool done = false;                                                                                              \
    if (!done) {                                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(add_ln_mod_fwd_q8_kernel<NIT, NW>),                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 8 * NW * (1536 + 8));                          \
      done = true;                                                                                                         \
    }                                                                                                                      \
    hipLaunchKernelGGL((add_ln_mod_fwd_q8_kernel<NIT, NW>), dim3(M / (8 * NW)), dim3(64 * NW), lds, st,                   \
                       (const bf16_t*)x_in, (const bf16_t*)y, gate, shift, scale, mod_ld, (bf16_t*)x_out,                 \
                       (unsigned char*)q8, ldq, (unsigned char*)q8t, ldqt, q_scale, q_amax, mean, rstd, M, T, D, eps);     \
  }
#define Q8_CASE(NIT)                \
  case NIT:                         \
    if (nw == 8) Q8_LAUNCH(NIT, 8)  \
    else Q8_LAUNCH(NIT, 4)          \
    break;
  switch ((D + 511) / 512) { Q8_CASE(1) Q8_CASE(2) Q8_CASE(3) }
#undef Q8_CASE
#undef Q8_LAUNCH
  const double md = (double)M * D;
  prof.done(UWU_PROF_LN_FWD, 0, 8.0 * md, md * (2.0 * (1 + (y ? 2 : 0)) + 2.0));
  UWU_LAUNCH_CHECK("add_ln_modulate_fwd_q8");
  return UWU_OK;
}

extern "C" int uwu_add_ln_modulate_fwd(const void* x_in, const void* y, const float* gate, const float* shift,
                                       const float* scale, int mod_ld, void* x_out, void* h, float* mean, float* rstd,
                                       int B, int T, int D, float eps, int affine, int dtype, void* stream) {
  UWU_CHECK_ARG(x_in && h && mean && rstd, "add_ln_modulate_fwd: null pointer");
  UWU_CHECK_ARG(B > 0 && T > 0 && D > 0 && D % 8 == 0 && D <= MAX_D, "add_ln_modulate_fwd: D=%d unsupported", D);
  UWU_CHECK_ARG((y == nullptr) || (gate && x_out), "add_ln_modulate_fwd: y needs gate and x_out");
  UWU_CHECK_ARG((scale == nullptr) == (shift == nullptr), "add_ln_modulate_fwd: scale/shift go together");
  UWU_CHECK_ARG(!scale || mod_ld % 4 == 0, "add_ln_modulate_fwd: mod_ld must be a multiple of 4");
  UWU_CHECK_ARG((((uintptr_t)x_in | (uintptr_t)h) & 15) == 0, "add_ln_modulate_fwd: tensors must be 16-byte aligned");
  const int M = B * T;
  int grid = (M + 3) / 4;
  if (grid > 8192) grid = 8192;
  hipStream_t st = (hipStream_t)stream;
  UWU_CHECK_ARG(dtype == UWU_F32 || dtype == UWU_BF16, "add_ln_modulate_fwd: bad dtype");
  UwuProfScope prof(stream);
  static UwuEnv r16("UWU_LN_ROW16");  // "0": the one-row-per-wave kernel at every width (A/B comparisons)
  const bool al16 = (((uintptr_t)x_in | (uintptr_t)h | (uintptr_t)y | (uintptr_t)x_out) & 15) == 0 && (mod_ld % 4 == 0);
  // (up to D = 768: 126 registers, 4 waves per SIMD; at D = 1152 its 174 registers left 2 waves per SIMD and it ran at 3.6 TB/s
  //  against the one-row-per-wave kernel's 4.7 -- UWU_LN_ROW16=1 forces it up to D = 1536)
  if (!r16.get().is('0') && D % 128 == 0 && D <= (r16.get().is('1') ? 1536 : 768) && al16 && M >= 4096) {
    int g16 = (M + 15) / 16;
    if (g16 > 4096) g16 = 4096;
#define F16_CASE(NCH)                                                                                                          \
  case NCH:                                                                                                                    \
    if (dtype == UWU_F32)                                                                                                      \
      hipLaunchKernelGGL((add_ln_mod_fwd16_kernel<float, NCH>), dim3(g16), dim3(256), 0, st, (const float*)x_in, (const float*)y, \
                         gate, shift, scale, mod_ld, (float*)x_out, (float*)h, mean, rstd, M, T, eps, affine);                   \
    else                                                                                                                       \
      hipLaunchKernelGGL((add_ln_mod_fwd16_kernel<bf16_t, NCH>), dim3(g16), dim3(256), 0, st, (const bf16_t*)x_in,            \
                         (const bf16_t*)y, gate, shift, scale, mod_ld, (bf16_t*)x_out, (bf16_t*)h, mean, rstd, M, T, eps,       \
                         affine);                                                                                               \
    break;
    switch (D / 128) {
      F16_CASE(1) F16_CASE(2) F16_CASE(3) F16_CASE(4) F16_CASE(5) F16_CASE(6) F16_CASE(7) F16_CASE(8) F16_CASE(9) F16_CASE(10)
      F16_CASE(11) F16_CASE(12)
    }
#undef F16_CASE
    const double e = dtype == UWU_BF16 ? 2.0 : 4.0, md = (double)M * D;
    prof.done(UWU_PROF_LN_FWD, dtype == UWU_BF16 ? 0 : 1, 8.0 * md, md * e * (2 + (y ? 1 : 0) + (x_out && x_out != x_in ? 1 : 0)));
    UWU_LAUNCH_CHECK("add_ln_modulate_fwd");
    return UWU_OK;
  }
#define FWD_CASE(NIT)                                                                                              \
  case NIT:                                                                                                        \
    if (dtype == UWU_F32)                                                                                          \
      hipLaunchKernelGGL((add_ln_mod_fwd_kernel<float, NIT>), dim3(grid), dim3(256), 0, st, (const float*)x_in,    \
                         (const float*)y, gate, shift, scale, mod_ld, (float*)x_out, (float*)h, mean, rstd, M, T, D, \
                         eps, affine);                                                                                     \
    else                                                                                                           \
      hipLaunchKernelGGL((add_ln_mod_fwd_kernel<bf16_t, NIT>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x_in,  \
                         (const bf16_t*)y, gate, shift, scale, mod_ld, (bf16_t*)x_out, (bf16_t*)h, mean, rstd, M,  \
                         T, D, eps, affine);                                                                               \
    break;
  switch ((D + 511) / 512) {
    FWD_CASE(1) FWD_CASE(2) FWD_CASE(3) FWD_CASE(4) FWD_CASE(5) FWD_CASE(6) FWD_CASE(7) FWD_CASE(8)
  }
#undef FWD_CASE
  {  // algorithmic bytes: x (+ y) read, h (+ x_out) written
    const double e = dtype == UWU_BF16 ? 2.0 : 4.0, md = (double)M * D;
    prof.done(UWU_PROF_LN_FWD, dtype == UWU_BF16 ? 0 : 1, 8.0 * md, md * e * (2 + (y ? 1 : 0) + (x_out && x_out != x_in ? 1 : 0)));
  }
  UWU_LAUNCH_CHECK("add_ln_modulate_fwd");
  return UWU_OK;
}

static bool small_ln_on() {  // UWU_LN_SMALL=0: the 4-wave kernel at every size (A/B comparisons)
  static UwuEnv on("UWU_LN_SMALL");
  return !on.get().is('0');
}

extern "C" int uwu_add_ln_modulate_bwd(const void* dh, const void* x, const float* mean, const float* rstd,
                                       const float* scale, const void* dx_in, const void* y, const float* gate,
                                       int mod_ld, void* dx_out, void* dy, float* dshift, float* dscale, float* dgate,
                                       int B, int T, int D, int affine, int dtype, void* stream) {
  UWU_CHECK_ARG(dh && x && mean && rstd && dx_out, "add_ln_modulate_bwd: null pointer");
  UWU_CHECK_ARG(B > 0 && T > 0 && D > 0 && D % 8 == 0 && D <= MAX_D, "add_ln_modulate_bwd: D=%d unsupported", D);
  UWU_CHECK_ARG((((uintptr_t)dh | (uintptr_t)x | (uintptr_t)dx_out) & 15) == 0, "add_ln_modulate_bwd: tensors must be 16-byte aligned");
  UWU_CHECK_ARG((y == nullptr) || (gate && dy), "add_ln_modulate_bwd: y needs gate and dy");
  UWU_CHECK_ARG(!(scale || y) || mod_ld % 4 == 0, "add_ln_modulate_bwd: mod_ld must be a multiple of 4");
  const int M = B * T;
  hipStream_t st = (hipStream_t)stream;
  UWU_CHECK_ARG(dtype == UWU_F32 || dtype == UWU_BF16, "add_ln_modulate_bwd: bad dtype");
  static UwuEnv aff_e("UWU_LN_AFFINE");  // UWU_LN_AFFINE=0: the per-sample kernel for every case (A/B comparisons)
  const bool aff_on = !aff_e.get().is('0');
  if (aff_on && affine && mod_ld == 0 && scale && !y && dshift && dscale && D <= 2048 && (dx_in == nullptr || ((uintptr_t)dx_in & 15) == 0)) {
    // one weight / bias vector for every row: 8-wave workgroups, LDS-folded column sums
    int rpb = M >= 16384 ? 64 : (M >= 4096 ? 32 : 16);
    {
      static UwuEnv rows_e("UWU_LN_ROWS");
      const int forced = rows_e.get().set ? rows_e.ival : 0;
      if (forced > 0) rpb = forced;
    }
    const int grid = (M + rpb - 1) / rpb;
    const size_t lds2 = (size_t)8 * 2 * D * sizeof(float);
    if (lds2 > 64 * 1024) {
      static bool done_f = false, done_b = false;
      bool& done = dtype == UWU_F32 ? done_f : done_b;
      if (!done) {
        for (int nit = 1; nit <= 4; ++nit) {
          const void* kf = nullptr;
          if (dtype == UWU_F32)
            kf = nit == 1 ? (const void*)ln_affine_bwd_kernel<float, 1> : nit == 2 ? (const void*)ln_affine_bwd_kernel<float, 2>
                 : nit == 3 ? (const void*)ln_affine_bwd_kernel<float, 3> : (const void*)ln_affine_bwd_kernel<float, 4>;
          else
            kf = nit == 1 ? (const void*)ln_affine_bwd_kernel<bf16_t, 1> : nit == 2 ? (const void*)ln_affine_bwd_kernel<bf16_t, 2>
                 : nit == 3 ? (const void*)ln_affine_bwd_kernel<bf16_t, 3> : (const void*)ln_affine_bwd_kernel<bf16_t, 4>;
          (void)hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
        done = true;
      }
    }
    UwuProfScope prof(stream);
#define AFF_CASE(NIT)                                                                                                   \
  case NIT:                                                                                                             \
    if (dtype == UWU_F32)                                                                                               \
      hipLaunchKernelGGL((ln_affine_bwd_kernel<float, NIT>), dim3(grid), dim3(512), lds2, st, (const float*)dh,         \
                         (const float*)x, mean, rstd, scale, (const float*)dx_in, (float*)dx_out, dshift, dscale, M, D, \
                         rpb);                                                                                          \
    else                                                                                                                \
      hipLaunchKernelGGL((ln_affine_bwd_kernel<bf16_t, NIT>), dim3(grid), dim3(512), lds2, st, (const bf16_t*)dh,       \
                         (const bf16_t*)x, mean, rstd, scale, (const bf16_t*)dx_in, (bf16_t*)dx_out, dshift, dscale, M, \
                         D, rpb);                                                                                       \
    break;
    switch ((D + 511) / 512) {
      AFF_CASE(1) AFF_CASE(2) AFF_CASE(3) AFF_CASE(4)
    }
#undef AFF_CASE
    const double e = dtype == UWU_BF16 ? 2.0 : 4.0, md = (double)M * D;
    prof.done(UWU_PROF_LN_BWD, dtype == UWU_BF16 ? 0 : 1, 16.0 * md, md * e * (3 + (dx_in ? 1 : 0)));
    UWU_LAUNCH_CHECK("ln_affine_bwd");
    return UWU_OK;
  }
  int rows = 32;
  while (rows > 1 && T % rows) rows >>= 1;
  // small batches: fewer rows per workgroup (down to one per wave) until the grid has ~4 workgroups per CU -- at
  // M = 4096 the 128 workgroups of 32 rows took 18.5 us, most of it eight dependent row passes per wave
  // ... but every workgroup adds its column sums to the sample's 3 D floats: at M = 4096 the 1024 workgroups of 4 rows issued
  // 1.2 M global atomics (13 us for 19 MB).  Small batches: 8 waves x 2 rows per workgroup, a quarter of the atomics.
  const bool small = M / rows < 1024 && T % 16 == 0 && D <= 640 && small_ln_on();  // (8 x 3 x D floats of LDS <= 64 KB)
  if (small) rows = 16;
  while (!small && rows > 4 && M / rows < 1024) rows >>= 1;
  {
    static UwuEnv rows_e("UWU_LN_ROWS");
    const int forced = rows_e.get().set ? rows_e.ival : 0;
    if (forced > 0 && T % forced == 0) rows = forced;
  }
  const size_t lds = (size_t)(small ? 8 : (D > 1024 ? 1 : 4)) * 3 * D * sizeof(float);  // (D > 1024: MAX_IT >= 3, one slab)
  if (small) {
    UwuProfScope prof(stream);
#define SMALL_CASE(NIT)                                                                                                \
  case NIT:                                                                                                            \
    hipLaunchKernelGGL((add_ln_mod_bwd_kernel<bf16_t, NIT, 8>), dim3(M / rows), dim3(512), lds, st, (const bf16_t*)dh, \
                       (const bf16_t*)x, mean, rstd, scale, (const bf16_t*)dx_in, (const bf16_t*)y, gate, mod_ld,      \
                       (bf16_t*)dx_out, (bf16_t*)dy, dshift, dscale, dgate, M, T, D, rows, affine);                    \
    break;
    if (dtype == UWU_BF16) {
      switch ((D + 511) / 512) { SMALL_CASE(1) SMALL_CASE(2) }
#undef SMALL_CASE
      const double md = (double)M * D;
      prof.done(UWU_PROF_LN_BWD, 0, 16.0 * md, md * 2.0 * (3 + (dx_in ? 1 : 0) + (y ? 2 : 0)));
      UWU_LAUNCH_CHECK("add_ln_modulate_bwd");
      return UWU_OK;
    }
  }
  UWU_CHECK_ARG(lds <= 160 * 1024, "add_ln_modulate_bwd: D=%d needs %zu B of LDS (> 160 KB)", D, lds);
  UwuProfScope prof(stream);
#define BWD_CASE(NIT)                                                                                               \
  case NIT:                                                                                                         \
    if (lds > 64 * 1024) { /* above the default dynamic-LDS limit: raise it once per instantiation */              \
      static bool done_f = false, done_b = false;                                                                   \
      bool& done = dtype == UWU_F32 ? done_f : done_b;                                                              \
      if (!done) {                                                                                                  \
        const void* kf = dtype == UWU_F32 ? reinterpret_cast<const void*>(add_ln_mod_bwd_kernel<float, NIT>)        \
                                          : reinterpret_cast<const void*>(add_ln_mod_bwd_kernel<bf16_t, NIT>);      \
        (void)hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                      \
        done = true;                                                                                                \
      }                                                                                                             \
    }                                                                                                               \
    if (dtype == UWU_F32)                                                                                           \
      hipLaunchKernelGGL((add_ln_mod_bwd_kernel<float, NIT>), dim3(M / rows), dim3(256), lds, st, (const float*)dh, \
                         (const float*)x, mean, rstd, scale, (const float*)dx_in, (const float*)y, gate, mod_ld,    \
                         (float*)dx_out, (float*)dy, dshift, dscale, dgate, M, T, D, rows, affine);                         \
    else                                                                                                            \
      hipLaunchKernelGGL((add_ln_mod_bwd_kernel<bf16_t, NIT>), dim3(M / rows), dim3(256), lds, st,                  \
                         (const bf16_t*)dh, (const bf16_t*)x, mean, rstd, scale, (const bf16_t*)dx_in,              \
                         (const bf16_t*)y, gate, mod_ld, (bf16_t*)dx_out, (bf16_t*)dy, dshift, dscale, dgate, M, T,   \
                         D, rows, affine);                                                                                  \
    break;
  switch ((D + 511) / 512) {
    BWD_CASE(1) BWD_CASE(2) BWD_CASE(3) BWD_CASE(4) BWD_CASE(5) BWD_CASE(6)
    default:
      UWU_CHECK_ARG(false, "add_ln_modulate_bwd: D=%d > 3072 not instantiated", D);
  }
#undef BWD_CASE
  {  // algorithmic bytes: dh, x (+ dx_in, y) read, dx_out (+ dy) written
    const double e = dtype == UWU_BF16 ? 2.0 : 4.0, md = (double)M * D;
    prof.done(UWU_PROF_LN_BWD, dtype == UWU_BF16 ? 0 : 1, 16.0 * md, md * e * (3 + (dx_in ? 1 : 0) + (y ? 2 : 0)));
  }
  UWU_LAUNCH_CHECK("add_ln_modulate_bwd");
  return UWU_OK;
}

static int colsum_impl(const void* X, int dtype, int batch, int M, int N, int ldx, float* out, int accumulate,
                       void* stream);
extern "C" int uwu_colsum(const void* X, int dtype, int M, int N, int ldx, float* out, int accumulate, void* stream) {
  return colsum_impl(X, dtype, 1, M, N, ldx, out, accumulate, stream);
}
// out[b, n] (+)= sum_m X[b, m, n] for `batch` contiguous [M, ldx] slabs (per-sample reductions)
extern "C" int uwu_colsum_batched(const void* X, int dtype, int batch, int M, int N, int ldx, float* out,
                                  int accumulate, void* stream) {
  return colsum_impl(X, dtype, batch, M, N, ldx, out, accumulate, stream);
}
static int colsum_impl(const void* X, int dtype, int batch, int M, int N, int ldx, float* out, int accumulate,
                       void* stream) {
  UWU_CHECK_ARG(X && out && batch > 0 && M > 0 && N > 0 && N % 4 == 0 && ldx % 4 == 0 && ldx >= N,
                "colsum: bad args (N=%d ldx=%d)", N, ldx);
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) {
    if (hipMemsetAsync(out, 0, (size_t)batch * N * sizeof(float), st) != hipSuccess) {
      uwu_set_error("colsum: memset failed");
      return UWU_ELAUNCH;
    }
  }
  int splits = (M + 255) / 256;
  const int strips = (N + 63) / 64;
  int want = 1024 / (strips * batch);
  if (want < 1) want = 1;
  if (splits > want) splits = want;
  const int rows = (M + splits - 1) / splits;
  splits = (M + rows - 1) / rows;
  if (dtype == UWU_F32)
    hipLaunchKernelGGL((colsum_kernel<float>), dim3(strips, splits, batch), dim3(256), 0, st, (const float*)X, M, N, ldx,
                       out, rows);
  else if (dtype == UWU_BF16)
    hipLaunchKernelGGL((colsum_kernel<bf16_t>), dim3(strips, splits, batch), dim3(256), 0, st, (const bf16_t*)X, M, N,
                       ldx, out, rows);
  else
    UWU_CHECK_ARG(false, "colsum: bad dtype");
  UWU_LAUNCH_CHECK("colsum");
  return UWU_OK;
}
