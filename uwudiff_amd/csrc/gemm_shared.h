// Shared pieces of the MFMA GEMM kernels (gemm.hip, gemm_p8.hip): argument block, LDS images + LDS-DMA staging,
// fragment reads and the epilogue of one wave tile.  Everything but GemmArgs lives in an anonymous namespace:
// each translation unit gets its own copy.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#define RETURN_IF(expr)            \
  do {                             \
    const int rc_ = (expr);        \
    if (rc_ != UWU_OK) return rc_; \
  } while (0)

struct GemmArgs {
  const void* A;
  const void* B;
  void* C;
  void* C2;
  const float* bias;
  const void* aux;
  int M, N, K, lda, ldb, ldc, ldaux, epi, tiles_m, tiles_n, k_tiles_per_split, wide;
  int xs, nloc, part_m;  // gemm_tr_kernel: XCD partition (slice lanes, tiles per tile lane, tile lanes along m)
  int aux16;  // dGELU: aux rows are 16-byte addressable in the paired-lane layout of the wide stores (bf16, ldaux % 8 == 0)
  // implicit-GEMM 3x3 convolution (padding 1): geometry of the gathered operand (CONV template parameters)
  int cH, cW, cC, cHo, cWo, cS;  // input H x W x C (channels-last), output Ho x Wo, stride
  const void* zero;              // >= 16 zero bytes in device memory: the source of padded / out-of-range pixels
  // gemm_f8_kernel<EMIT>: the epilogue's result also leaves as fp8 bytes, row-major [M, N] and transposed [N, M]
  void* q8;
  void* q8t;
  const float* q_scale;  // device float: q = sat(value * q_scale)
  float* q_amax;         // optional: atomic max |value|
  int ldq, ldqt;
  int nt_c;  // bit 0 / 1: bf16 wide stores of C / C2 are streaming (nontemporal) stores: UWU_GEMM_NT_C (experiments)
  int p8_cont;  // gemm_p8_kernel: continuous mode allowed (UWU_P8_CONT=0: every tile drains; A/B comparisons)
};

namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROW_BYTES = 128;
constexpr int TILE_BYTES = 128 * ROW_BYTES;  // 16 KB per operand tile

template <typename T> struct GT;
template <> struct GT<bf16_t> { static constexpr int EPC = 8; static constexpr int BK = 64; };
template <> struct GT<float> { static constexpr int EPC = 4; static constexpr int BK = 32; };

__device__ __forceinline__ int swz(int row, int chunk) {
  return row * ROW_BYTES + (((chunk ^ (row >> 1) ^ (row >> 4)) & 7) << 4);
}

// Fragment read as inline asm for the LDS-DMA path: hipcc cannot prove that a compiler-visible ds_read does not
// alias the in-flight DMA destination (the OTHER stage) and drains s_waitcnt vmcnt(0) in front of it, so the next
// stage's load would never overlap this stage's MFMAs (PMC: waves parked 46 % of their cycles).  The dependency
// on the DMA is carried by the explicit vmcnt(0) + barrier at the end of each K-step instead.
typedef unsigned gu32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 lds_read128_asm(const char* p) {
  gu32x4 v;
  const unsigned a = (unsigned)(size_t)((const __attribute__((address_space(3))) char*)p);
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory");
  return uint4{v[0], v[1], v[2], v[3]};
}

// ---- staging: one 128-row x 128-byte operand tile per call, 256 threads ------------------------------
template <typename T, bool TRANS>
struct Stager {
  uint4 r[4];

  // base: operand base pointer; ld: leading dimension (elements); row0: first output-row (m or n) of the tile;
  // k0: first reduction index; nrows: operand extent in the output dim; K: reduction extent.
  __device__ __forceinline__ void load(const T* __restrict__ base, int ld, int row0, int k0, int nrows, int K,
                                       int tid) {
    constexpr int EPC = GT<T>::EPC;
    if constexpr (!TRANS) {
      const int c = tid & 7, rr = tid >> 3;
      const int kk = k0 + c * EPC;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int grow = row0 + rr + 32 * p;
        if (grow < nrows && kk < K)
          r[p] = *reinterpret_cast<const uint4*>(base + (int64_t)grow * ld + kk);
        else
          r[p] = uint4{0, 0, 0, 0};
      }
    } else if constexpr (sizeof(T) == 2) {
      const int xg = tid & 15, kg = tid >> 4;
      const int x = row0 + 8 * xg;
#pragma unroll
      for (int kr = 0; kr < 4; ++kr) {
        const int k = k0 + 4 * kg + kr;
        if (k < K && x < nrows)
          r[kr] = *reinterpret_cast<const uint4*>(base + (int64_t)k * ld + x);
        else
          r[kr] = uint4{0, 0, 0, 0};
      }
    } else {
      const int xg = tid & 31, kg = tid >> 5;
      const int x = row0 + 4 * xg;
#pragma unroll
      for (int kr = 0; kr < 4; ++kr) {
        const int k = k0 + 4 * kg + kr;
        if (k < K && x < nrows)
          r[kr] = *reinterpret_cast<const uint4*>(base + (int64_t)k * ld + x);
        else
          r[kr] = uint4{0, 0, 0, 0};
      }
    }
  }

  __device__ __forceinline__ void store(char* __restrict__ lds, int tid) const {
    if constexpr (!TRANS) {
      const int c = tid & 7, rr = tid >> 3;
#pragma unroll
      for (int p = 0; p < 4; ++p) *reinterpret_cast<uint4*>(lds + swz(rr + 32 * p, c)) = r[p];
    } else if constexpr (sizeof(T) == 2) {
      const int xg = tid & 15, kg = tid >> 4;
      const unsigned* w0 = reinterpret_cast<const unsigned*>(&r[0]);
      const unsigned* w1 = reinterpret_cast<const unsigned*>(&r[1]);
      const unsigned* w2 = reinterpret_cast<const unsigned*>(&r[2]);
      const unsigned* w3 = reinterpret_cast<const unsigned*>(&r[3]);
#pragma unroll
      for (int xi = 0; xi < 8; ++xi) {
        const int row = 8 * xg + xi;
        const unsigned sel = (xi & 1) ? 0x07060302u : 0x05040100u;
        uint2 o;
        o.x = __builtin_amdgcn_perm(w1[xi >> 1], w0[xi >> 1], sel);  // {k+1, k}
        o.y = __builtin_amdgcn_perm(w3[xi >> 1], w2[xi >> 1], sel);  // {k+3, k+2}
        *reinterpret_cast<uint2*>(lds + swz(row, kg >> 1) + 8 * (kg & 1)) = o;
      }
    } else {
      const int xg = tid & 31, kg = tid >> 5;
      const unsigned* w0 = reinterpret_cast<const unsigned*>(&r[0]);
      const unsigned* w1 = reinterpret_cast<const unsigned*>(&r[1]);
      const unsigned* w2 = reinterpret_cast<const unsigned*>(&r[2]);
      const unsigned* w3 = reinterpret_cast<const unsigned*>(&r[3]);
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        const int row = 4 * xg + xi;
        *reinterpret_cast<uint4*>(lds + swz(row, kg)) = uint4{w0[xi], w1[xi], w2[xi], w3[xi]};
      }
    }
  }
};

// LDS-DMA staging of a K-contiguous 128-row tile (global_load_lds_dwordx4: no VGPR round trip, no ds_write).
// One wave-instruction fills 8 rows x 128 B; the LDS image must be lane-linear, so the chunk swizzle is applied
// to the per-lane SOURCE address (lane l lands at chunk position l&7 of row l>>3, hence fetches logical chunk
// (l&7) ^ f(row)); rows past the operand's extent are clamped (their products are never stored).
// Requires full K tiles (K % BK == 0).
template <typename T>
__device__ __forceinline__ void glds_tile(const T* __restrict__ base, int ld, int row0, int k0, int nrows,
                                          char* __restrict__ lds_tile, int tid) {
  constexpr int EPC = GT<T>::EPC;
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int grp = wave + 4 * g;
    const int row = 8 * grp + (lane >> 3);
    const int c = ((lane & 7) ^ (row >> 1) ^ (row >> 4)) & 7;
    int grow = row0 + row;
    if (grow >= nrows) grow = nrows - 1;
    const T* src = base + (int64_t)grow * ld + k0 + c * EPC;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds_tile + grp * 1024), 16, 0, 0);
  }
}

template <typename T>
__device__ __forceinline__ void mma_frag(const uint4& a, const uint4& b, f32x4& c) {
  if constexpr (sizeof(T) == 2) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a),
                                                *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
  } else {
    const float* af = reinterpret_cast<const float*>(&a);
    const float* bf = reinterpret_cast<const float*>(&b);
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], bf[e], c, 0, 0, 0);
  }
}



// n / d and n % d for 0 <= n < 2^24 (exact in fp32): one multiply + a correction instead of an integer division
__device__ __forceinline__ void divmod24(int n, int d, float rcp, int& q, int& r) {
  q = (int)((float)n * rcp);
  r = n - q * d;
  if (r < 0) { r += d; --q; }
  if (r >= d) { r -= d; ++q; }
}

// Epilogue of one wave tile of FI x FJ 16x16 accumulators in the swapped-operand orientation (each lane owns 4
// consecutive columns of one row): bias / GELU / dGELU(+column sums) math and 8- or 16-byte stores.
// (m_w, n_w) = first row / column of the wave tile.
// EPI >= 0 fixes the epilogue at compile time (the hot bf16 instantiations: a third of the code and no branches in
// the 16..32-fold unrolled store loops); EPI = -1 reads g.epi at run time.
// What the epilogue READS (bias, dGELU's aux), fetched by epi_prefetch() at KERNEL START: C / C2 / aux / bias carry
// no restrict, so the compiler keeps a load that follows a store in program order behind it, and loads placed inside
// the per-fragment loop exposed one full memory latency per fragment; loaded at the top of the epilogue they still
// cost one exposed HBM latency per tile while the workgroup holds its LDS and registers.  Issued before the K loop
// they are simply there.  Bias: one float4 per column fragment.  aux: a ring of PD rows of fragments (row i + PD is
// loaded when row i has been consumed).
template <typename T, int FI, int FJ>
struct EpiPre {
  static constexpr int PD = FI < 4 ? FI : 4;
  typedef typename std::conditional<sizeof(T) == 2, uint2, f32x4>::type AuxRaw;
  f32x4 bias[FJ];
  AuxRaw aux[PD][FJ];
};
template <typename T, int FI, int FJ>
__device__ __forceinline__ void epi_load_aux_row(const GemmArgs& g, int m_w, int n_w, int fr, int fq, int i,
                                                 typename EpiPre<T, FI, FJ>::AuxRaw (&dst)[FJ]) {
  typedef typename EpiPre<T, FI, FJ>::AuxRaw AuxRaw;
  const T* aux = static_cast<const T*>(g.aux);
  const int m = m_w + 16 * i + fr;
  if constexpr (sizeof(T) == 2 && FJ % 2 == 0) {
    if (g.aux16) {
      // one 16-byte load per PAIR of column fragments, in the lane layout of the wide stores (8 consecutive columns per
      // lane: 64 contiguous bytes per row and instruction instead of 32, half the load instructions); the pair is
      // un-swapped with v_permlane16_swap when the row is consumed (epi_unswap_aux), so the load itself stays asynchronous
      const bool odd = fq & 1;
#pragma unroll
      for (int jp = 0; jp < FJ / 2; ++jp) {
        const int nb = n_w + 32 * jp;
        const int n = odd ? nb + 16 + 4 * (fq - 1) : nb + 4 * fq;
        uint4 v = {0u, 0u, 0u, 0u};
        if (m < g.M && n < g.N) v = *reinterpret_cast<const uint4*>(aux + (int64_t)m * g.ldaux + n);
        dst[2 * jp] = AuxRaw{v.x, v.y};
        dst[2 * jp + 1] = AuxRaw{v.z, v.w};
      }
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < FJ; ++j) {
    const int n = n_w + 16 * j + 4 * fq;
    if (m < g.M && n < g.N) dst[j] = *reinterpret_cast<const AuxRaw*>(aux + (int64_t)m * g.ldaux + n);
    else dst[j] = AuxRaw{};
  }
}
template <typename T, int FI, int FJ, int EPI>
__device__ __forceinline__ void epi_prefetch(EpiPre<T, FI, FJ>& pre, const GemmArgs& g, int m_w, int n_w, int fr,
                                             int fq) {
  const int epi = EPI >= 0 ? EPI : g.epi;
  const bool has_bias = epi == UWU_EPI_BIAS || epi == UWU_EPI_BIAS_GELU || epi == UWU_EPI_BIAS_SILU;
#pragma unroll
  for (int j = 0; j < FJ; ++j) {
    const int n = n_w + 16 * j + 4 * fq;
    pre.bias[j] = (has_bias && n < g.N) ? load4(g.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (epi == UWU_EPI_DGELU) {
#pragma unroll
    for (int i = 0; i < EpiPre<T, FI, FJ>::PD; ++i) epi_load_aux_row<T, FI, FJ>(g, m_w, n_w, fr, fq, i, pre.aux[i]);
  }
}

// cs_lds: >= 4 * 16 * FJ floats of LDS, free once every wave has left the K loop (dGELU column sums only).
template <typename T, typename TC, int FI, int FJ, int EPI = -1>
__device__ __forceinline__ void epilogue_tile(f32x4 (&acc)[FI][FJ], EpiPre<T, FI, FJ>& pre, const GemmArgs& g, int m_w,
                                              int n_w, int fr, int fq, float* cs_lds, int wm, int wn, int cs_t = -1,
                                              f32x4* cs_out = nullptr) {
  // cs_t: this thread's index among the flushers of its 2 x 2-wave column group (default: threadIdx.x, one group)
  // cs_out != NULL: the dGELU column sums of this call are ADDED to cs_out[FJ] (this lane's 4 columns per fragment, rows
  // fr still apart) and nothing is flushed -- the caller folds several calls and flushes itself (gemm_p8.hip)
  TC* C = static_cast<TC*>(g.C);
  TC* C2 = static_cast<TC*>(g.C2);
  const int epi = EPI >= 0 ? EPI : g.epi;
  const bool has_bias = epi == UWU_EPI_BIAS || epi == UWU_EPI_BIAS_GELU || epi == UWU_EPI_BIAS_SILU;
  const bool two = epi == UWU_EPI_BIAS_GELU || epi == UWU_EPI_BIAS_SILU;
  const bool dgelu = epi == UWU_EPI_DGELU;
  constexpr int PD = EpiPre<T, FI, FJ>::PD;
  typedef typename EpiPre<T, FI, FJ>::AuxRaw AuxRaw;
  auto aux_f32 = [](const AuxRaw& r) {
    if constexpr (sizeof(T) == 2) {
      const bf16x4 b = *reinterpret_cast<const bf16x4*>(&r);
      return f32x4{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
    } else {
      return r;
    }
  };

  // per-fragment math on this lane's 4 consecutive columns
  auto finish = [&](f32x4 v, const f32x4& b, const AuxRaw& a, f32x4& second) {
    if (has_bias) v = v + b;
    if (dgelu) {
      v = v * dgelu_tanh_f4(aux_f32(a));
    }
    if (epi == UWU_EPI_BIAS_GELU) {
      second = gelu_tanh_f4(v);
    } else if (epi == UWU_EPI_BIAS_SILU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) second[e] = silu_f(v[e]);
    }
    return v;
  };
  // UWU_EPI_DGELU with C2 != NULL: C2 is a float[N] that receives += the column sums of C (the bias gradient of
  // the Linear whose pre-activation is `aux`), summed over this tile's rows in registers / lanes, then atomics
  float* colsum = dgelu ? reinterpret_cast<float*>(g.C2) : nullptr;
  f32x4 csum[FJ];
#pragma unroll
  for (int j = 0; j < FJ; ++j) csum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // (the fp32 values before rounding to the output type: one add per element, and closer to the fp32 reference)
  auto add_cs = [&](f32x4& acc_, const f32x4& v) { acc_ = acc_ + v; };
  // Global fp32 atomics are paid per wave-INSTRUCTION (~50 ns each per CU), not per lane: one atomic per (wave,
  // fragment, element) with 4 live lanes each cost 77 us per launch on the fc2 input-gradient.  So the four waves
  // first meet in LDS (the two wm halves cover the same columns) and 32 FJ lanes issue ONE atomic each.
  auto flush_cs = [&]() {
    if (!colsum) return;  // uniform
    constexpr int BNT = 2 * 16 * FJ;  // columns of the workgroup tile
    __syncthreads();                  // every wave has left the K loop: its LDS is free
#pragma unroll
    for (int j = 0; j < FJ; ++j) {
      f32x4 v = csum[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = row16_sum(v[e]);  // over the 16 rows fr of this column group
      if (fr == 0) store4(cs_lds + wm * BNT + wn * 16 * FJ + 16 * j + 4 * fq, v);
    }
    __syncthreads();
    const int t = cs_t >= 0 ? cs_t : (int)threadIdx.x, n = n_w - wn * 16 * FJ + t;
    if (t < BNT && n < g.N) atomicAdd(colsum + n, cs_lds[t] + cs_lds[BNT + t]);
  };
  // bf16 output: stores are instruction-bound (fp32 output of the same tile costs the same per store), so
  // pair lanes l and l^16 (column groups fq, fq^1) and swap half of two neighbouring 16-column subtiles:
  // even fq keeps subtile 2jp (own 4 columns + partner's next 4), odd fq keeps subtile 2jp+1 -> one 16-B store
  // per lane covering 8 consecutive columns, 64 contiguous bytes per row per instruction, half the stores.
  const bool wide = sizeof(TC) == 2 && g.wide;
  const bool odd = fq & 1;
  auto pack = [](const f32x4& v) {
    bf16x4 b = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    return *reinterpret_cast<uint2*>(&b);
  };
#pragma unroll
  for (int i = 0; i < FI; ++i) {
    const int m = m_w + 16 * i + fr;
    const bool mok = m < g.M;
    AuxRaw(&arow)[FJ] = pre.aux[i % PD];
    if constexpr (sizeof(T) == 2 && FJ % 2 == 0) {
      if (dgelu && g.aux16) {  // paired 16-byte aux loads -> this lane's own 4 columns of both fragments (the swap is an involution)
#pragma unroll
        for (int jp = 0; jp < FJ / 2; ++jp) {
          typedef unsigned su32x2 __attribute__((ext_vector_type(2)));
          const su32x2 sx = __builtin_amdgcn_permlane16_swap(arow[2 * jp].x, arow[2 * jp + 1].x, false, false);
          const su32x2 sy = __builtin_amdgcn_permlane16_swap(arow[2 * jp].y, arow[2 * jp + 1].y, false, false);
          arow[2 * jp] = AuxRaw{sx[0], sy[0]};
          arow[2 * jp + 1] = AuxRaw{sx[1], sy[1]};
        }
      }
    }
    f32x4 v[FJ], sec[FJ];
#pragma unroll
    for (int j = 0; j < FJ; ++j) {
      const int n = n_w + 16 * j + 4 * fq;
      sec[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      v[j] = (mok && n < g.N) ? finish(acc[i][j], pre.bias[j], arow[j], sec[j]) : acc[i][j];
      if (colsum && mok && n < g.N) add_cs(csum[j], v[j]);
    }
    // this row's aux fragments are consumed: fetch row i + PD into the slot (before this row's stores)
    if (dgelu && i + PD < FI) epi_load_aux_row<T, FI, FJ>(g, m_w, n_w, fr, fq, i + PD, arow);
    if (wide) {
      if constexpr (sizeof(TC) == 2) {
#pragma unroll
        for (int jp = 0; jp < FJ / 2; ++jp) {
          const int nb = n_w + 32 * jp;  // first column of subtile 2jp
          // v_permlane16_swap exchanges odd 16-lane rows of its first operand with even rows of its second: with
          // (p0, p1) = (this lane's subtile 2jp, subtile 2jp+1) an even-fq lane ends up with {own p0, partner's p0} and
          // an odd-fq lane with {partner's p1, own p1} -- in both cases the 8 consecutive columns it stores, with no
          // select and no trip through the LDS crossbar (ds_bpermute).
          auto exchange_store = [&](TC* dst, const f32x4& x0, const f32x4& x1, bool stream_out = false) {
            const uint2 p0 = pack(x0), p1 = pack(x1);
            typedef unsigned su32x2 __attribute__((ext_vector_type(2)));
            const su32x2 sx = __builtin_amdgcn_permlane16_swap(p0.x, p1.x, false, false);
            const su32x2 sy = __builtin_amdgcn_permlane16_swap(p0.y, p1.y, false, false);
            const int n = odd ? nb + 16 + 4 * (fq - 1) : nb + 4 * fq;
            if (mok && n < g.N) {
              typedef unsigned su32x4 __attribute__((ext_vector_type(4)));
              const su32x4 o = su32x4{sx[0], sy[0], sx[1], sy[1]};
              su32x4* ptr = reinterpret_cast<su32x4*>(dst + (int64_t)m * g.ldc + n);
              if (stream_out) __builtin_nontemporal_store(o, ptr);
              else *ptr = o;
            }
          };
          // the pre-activation of a GELU Linear is only read again in the backward pass: streaming store, so that
          // it does not push the activation (read next by fc2) out of the caches
          exchange_store(C, v[2 * jp], v[2 * jp + 1], epi == UWU_EPI_BIAS_GELU || (g.nt_c & 1));
          if (two) exchange_store(C2, sec[2 * jp], sec[2 * jp + 1], (g.nt_c & 2) != 0);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < FJ; ++j) {
        const int n = n_w + 16 * j + 4 * fq;
        if (!mok || n >= g.N) continue;
        store4(C + (int64_t)m * g.ldc + n, v[j]);
        if (two) store4(C2 + (int64_t)m * g.ldc + n, sec[j]);
      }
    }
  }
  if (cs_out) {
#pragma unroll
    for (int j = 0; j < FJ; ++j) cs_out[j] = cs_out[j] + csum[j];
  } else {
    flush_cs();
  }
}

constexpr int R_ROWB = 64;
constexpr int R_BSUB = 32 * 256;  // B part of a stage: 128 rows x 64 B (TB = 0) or 32 k-rows x 256 B (TB = 1)

__device__ __forceinline__ int r_gsw(int row) { return (0 - (row >> 2)) & 3; }
__device__ __forceinline__ int r_swz(int row, int chunk) { return row * R_ROWB + (((chunk ^ r_gsw(row)) & 3) << 4); }
template <int OFF>
__device__ __forceinline__ uint4 r_read128(unsigned addr) {
  gu32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return uint4{v[0], v[1], v[2], v[3]};
}
template <int N>
__device__ __forceinline__ void r_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// transposing LDS read (4 x 16 block of b16 per 16 lanes, delivered column-major); EXEC must be all ones
template <int OFF>
__device__ __forceinline__ uint2 t_read_tr(unsigned addr) {
  typedef unsigned tu32x2 __attribute__((ext_vector_type(2)));
  tu32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return uint2{v[0], v[1]};
}
// Byte offset inside a [32 k][128 x] sub-image that lane = 16 g + 4 q + p supplies to transposed read t (0 / 1) of
// the fragment whose first column is 8 * xb8: row 8 g + 4 t + q, columns 4 p .. 4 p + 3.  Chunk ch of k-row r sits
// at 256 r + 16 (ch ^ (((r & 3) << 2) | ((r >> 2) & 3)))  (cdna_hip_programming.md T10, image (b)); fragment f of
// the wave only flips chunk bits: address ^ (f << 5).
__device__ __forceinline__ unsigned tr_lane_base(int lane, int t, int xb8) {
  const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int krow = 8 * tg + 4 * t + tq;
  const int f = (tq << 2) | ((2 * tg + t) & 3);
  return (unsigned)(256 * krow + 16 * ((xb8 + (tp >> 1)) ^ f) + 8 * (tp & 1));
}

// ---- live profiler (prof.cpp): HIP event pairs around launches on the launch stream (bench.py `roofline`) -------
// tag of a GEMM launch: forward / input gradient / weight gradient, the two GELU Linears apart
static int gemm_tag(const GemmArgs& g, bool tb, bool acc) {
  if (acc) return UWU_PROF_GEMM_WGRAD;
  if (g.epi == UWU_EPI_BIAS_GELU) return UWU_PROF_GEMM_FC1_GELU;
  if (g.epi == UWU_EPI_DGELU) return UWU_PROF_GEMM_FC2_DGELU;
  return tb ? UWU_PROF_GEMM_DGRAD : UWU_PROF_GEMM_FWD;
}
// algorithmic bytes: both operands once, the output once (+ the second output / the aux operand of the GELU epilogues)
static double gemm_bytes(const GemmArgs& g, int es, int esc) {
  double b = ((double)g.M * g.K + (double)g.N * g.K) * es + (double)g.M * g.N * esc;
  if (g.epi == UWU_EPI_BIAS_GELU || g.epi == UWU_EPI_BIAS_SILU) b += (double)g.M * g.N * esc;
  if (g.epi == UWU_EPI_DGELU) b += (double)g.M * g.N * es;
  return b;
}


}  // namespace
// the 8-phase 256 x 256 kernel (gemm_p8.hip): returns UWU_OK / error; `tb` = B is [K][N] (input gradients)
int uwu_launch_gemm_p8(const GemmArgs& g, bool tb, hipStream_t st);
bool uwu_gemm_p8_ok(const GemmArgs& g, bool tb);
// its 128 x 384 sibling (gemm_p8n.hip): N a multiple of 384
int uwu_launch_gemm_p8n(const GemmArgs& g, bool tb, hipStream_t st);
bool uwu_gemm_p8n_ok(const GemmArgs& g, bool tb);
// the 8-phase kernel on fp8 operands (gemm_p8f.hip): forward / input gradient, and the weight gradient's K slices
bool uwu_gemm_p8f_ok(const GemmArgs& g);
int uwu_launch_gemm_p8f(GemmArgs g, int fmt_a, const float* sa, const float* sb, hipStream_t st);
int uwu_gemm_p8f_split(int tiles, int steps);
bool uwu_gemm_p8f_part_ok(const GemmArgs& g);
int uwu_launch_gemm_p8f_part(GemmArgs& g, int fmt_a, const float* sa, const float* sb, void* scratch, hipStream_t st);
