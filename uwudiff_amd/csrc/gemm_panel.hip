// Panel GEMM for the denoiser's token-parallel Linears (bf16, K-contiguous operands):
//     C[M,N] = A[M,K] . B[N,K]^T (+ bias, + GELU / dGELU epilogue),   M = tokens (65536 at B=256), K,N = 384..1536
// Reference op sequence replaced: nn.Linear fwd / input-grad inside the transformer blocks
// (reference src/duwu/modules/rope_unet.py:122-166, 404).
//
// Why a second kernel next to gemm.hip: with K this short a 128x128 tile pulls (128+128)*K*2 bytes through the
// CU's L2 port for 128*128*K*2 flop = 64 flop/B, and the chip's L2 delivers ~11-12 TB/s to LDS, so the 128x128
// kernel is L2-bound at half the speed the MFMAs and HBM would allow (measured phases: loads 45-70 us,
// MFMA 26 us, stores 40 us, serialised).  Here one workgroup owns a 256 x 384 tile = 154 flop/B:
//   * the A panel (activations) is read from HBM exactly once per 384 output columns and the n-tiles of one
//     panel run back to back on the same CU (N = 384 / 1152 / 1536 -> 1 / 3 / 4 tiles, no padding),
//   * 4 waves (one per SIMD, 512 registers each), wave tile 128 x 192 = 4 x 6 accumulators of
//     v_mfma_f32_32x32x16_bf16 -> 20 ds_read_b128 per 48 MFMAs,
//   * K-step = 32 bf16 = 64 B per row; 3-stage LDS ring (3 x 40 KB) filled by LDS-DMA
//     (global_load_lds_dwordx4), two K-steps in flight while the third is consumed,
//   * persistent: (tile, K-step) pairs are ONE stream, so the ring keeps running across tile boundaries, and
//     the epilogue's stores are never drained: the counted s_waitcnt vmcnt(N) at the next K-steps simply allows
//     for them (stores complete in issue order behind the DMA they follow).
// LDS image per stage: [A rows 0..255 | B rows 0..383] x 64 B, 16-byte chunk c of row r at position
// c ^ ((r >> 2) & 3): the DMA writes lane-linear (16 rows x 64 B per wave-instruction), so the swizzle is applied
// to the per-lane SOURCE address; the fragment reads (lane -> row l&31, chunk 2s + (l>>5)) are conflict-free in
// the four ds_read_b128 lane groups of MI355X_MICROARCH.md (LDS table).
// Operands are swapped in the MFMA (D rows = n in registers, columns = m on the lanes), so a lane owns 4
// consecutive output columns; v_permlane32_swap pairs lanes l / l+32 to 8 columns = one 16-byte store.
// The bias is not added in the epilogue at all: the accumulators START at the bias (same 384 v_mov as zeroing).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int ROWB = 64;                         // bytes per tile row per K-step
constexpr int PBM = 256, PBN = 384;              // workgroup tile
constexpr int PA_BYTES = PBM * ROWB;             // 16 KB
constexpr int PSTAGE = (PBM + PBN) * ROWB;       // 40 KB
constexpr int PNST = 3;                          // ring stages
constexpr int POFF_TAB = PNST * PSTAGE;          // float table [N]: bias (forward) or column sums (dGELU)
constexpr int PMAX_N = 6144;                     // table capacity (24 KB) -> 144 KB LDS in all
constexpr int PPIECES = 10;                      // DMA instructions per wave per K-step (4 A + 6 B)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Fragment / table reads as inline asm: hipcc cannot prove that a ds_read does not alias an in-flight LDS-DMA
// destination and would drain vmcnt(0) in front of every compiler-visible LDS read (gemm_ring.hip has the story).
template <int OFF>
__device__ __forceinline__ u32x4 lds_read128(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)((const __attribute__((address_space(3))) char*)p);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// The 24 accumulators of a wave (384 registers) do not fit one register file, and hipcc selects ONE form per
// function for the MFMA builtin (all accumulators in AGPRs -> 128 of them spilled to scratch).  So the MFMAs are
// inline asm with the file chosen per accumulator: column fragments j = 0..3 live in AGPRs (256), j = 4..5 in VGPRs
// (128), leaving 128 VGPRs for fragments and addresses.  asm volatile keeps the hand-written issue order.  The
// compiler does not know these are MFMAs: PANEL_MFMA_SETTLE() before compiler-generated reads of the accumulators.
__device__ __forceinline__ void mfma_ag(f32x16& acc, const u32x4& a, const u32x4& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_vg(f32x16& acc, const u32x4& a, const u32x4& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
#define PANEL_MFMA_SETTLE() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")

struct PanelArgs {
  const bf16_t* A;
  const bf16_t* B;
  bf16_t* C;
  bf16_t* C2;
  const float* bias;
  const bf16_t* aux;
  float* colsum;
  int M, N, K, lda, ldb, ldc, ldaux, tiles_n, ntiles;
};

enum { PEPI_NONE = 0, PEPI_BIAS = 1, PEPI_BIAS_GELU = 2, PEPI_DGELU = 3 };

// Exchange between lanes l and l+32: on return a lane with h = 0 holds {own a, partner's a}, a lane with h = 1
// holds {partner's b, own b} -- i.e. (x, y) are 2 x 4-column groups that are adjacent in the output row.
__device__ __forceinline__ void pair_swap(float& a, float& b) {
  u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  bf16x2 p = {(bf16_t)lo, (bf16_t)hi};
  return *reinterpret_cast<unsigned*>(&p);
}

// Stores (and dGELU's aux loads) of one wave tile.  Loop order j, p outer / i inner: everything that depends on the
// column only (table reads) is done once per 8-column group.
template <int EPI>
__device__ __forceinline__ void panel_epilogue(f32x16 (&acc)[4][6], const PanelArgs& g, int m_w, int n_w,
                                               unsigned tab_lds, int r, int h) {
  // (uniform base) + (32-bit lane byte offset) addressing as in the DMA
  const unsigned lane_c = (unsigned)((r * g.ldc + 8 * h) * 2);
  const unsigned lane_x = (unsigned)((r * g.ldaux + 8 * h) * 2);
  char* cbase = reinterpret_cast<char*>(g.C) + ((int64_t)m_w * g.ldc + n_w) * 2;
  char* c2base = reinterpret_cast<char*>(g.C2) + ((int64_t)m_w * g.ldc + n_w) * 2;
  const char* xbase = reinterpret_cast<const char*>(g.aux) + ((int64_t)m_w * g.ldaux + n_w) * 2;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int n = n_w + 32 * j + 16 * p + 8 * h;  // first of this lane's 8 output columns (after the exchange)
      const int ncol = 32 * j + 16 * p;             // uniform part of it inside the wave tile
      float cs[8];
      if constexpr (EPI == PEPI_DGELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] = 0.f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[i][j][8 * p + e];
          v[4 + e] = acc[i][j][8 * p + 4 + e];
          pair_swap(v[e], v[4 + e]);
        }
        if constexpr (EPI == PEPI_DGELU) {
          const f32x8 u = load8(reinterpret_cast<const bf16_t*>(xbase + ((int64_t)32 * i * g.ldaux + ncol) * 2 + lane_x));
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v[e] *= dgelu_tanh_f(u[e]);
            cs[e] += (float)(bf16_t)v[e];  // sum what the consumers read: the bf16-rounded values
          }
        }
        uint4 o = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
        *reinterpret_cast<uint4*>(cbase + ((int64_t)32 * i * g.ldc + ncol) * 2 + lane_c) = o;
        if constexpr (EPI == PEPI_BIAS_GELU) {
          float f[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = gelu_tanh_f(v[e]);
          uint4 o2 = {pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7])};
          *reinterpret_cast<uint4*>(c2base + ((int64_t)32 * i * g.ldc + ncol) * 2 + lane_c) = o2;
        }
      }
      if constexpr (EPI == PEPI_DGELU) {
        // column sums of this wave's 128 rows: lanes of equal h hold the same 8 columns for 32 different rows
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float s = cs[e];
#pragma unroll
          for (int o = 1; o < 32; o <<= 1) s += __shfl_xor(s, o, 64);
          cs[e] = s;
        }
        if (r == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            asm volatile("ds_add_f32 %0, %1" ::"v"(tab_lds + 4u * (unsigned)(n + e)), "v"(cs[e]) : "memory");
        }
      }
    }
  }
}

// accumulators start at the bias of their output column (register i of a 32x32 tile is row n = (i&3)+8(i>>2)+4h)
template <int EPI>
__device__ __forceinline__ void panel_init_acc(f32x16 (&acc)[4][6], unsigned tab_lds, int n_w, int h) {
  if constexpr (EPI == PEPI_BIAS || EPI == PEPI_BIAS_GELU) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      u32x4 b[4];
      const unsigned a0 = tab_lds + 4u * (unsigned)(n_w + 32 * j + 4 * h);
      b[0] = lds_read128<0>(a0);
      b[1] = lds_read128<32>(a0);
      b[2] = lds_read128<64>(a0);
      b[3] = lds_read128<96>(a0);
      wait_lgkm0();
      f32x16 t;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) t[4 * q + e] = __uint_as_float(b[q][e]);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = t;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[i][j] = f32x16{};
  }
}

template <int EPI>
__global__ void __launch_bounds__(256, 1) gemm_panel_kernel(const PanelArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NSTORE = (EPI == PEPI_BIAS_GELU) ? 96 : 48;            // store instructions per wave per tile
  constexpr int WAIT_EPI = (PPIECES + NSTORE > 63) ? 63 : PPIECES + NSTORE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int nk = g.K >> 5;
  const int t0 = (int)((int64_t)g.ntiles * blockIdx.x / gridDim.x);
  const int t1 = (int)((int64_t)g.ntiles * (blockIdx.x + 1) / gridDim.x);
  const int total = (t1 - t0) * nk;
  if (total == 0) return;  // uniform

  float* tab = reinterpret_cast<float*>(smem + POFF_TAB);
  if constexpr (EPI == PEPI_BIAS || EPI == PEPI_BIAS_GELU) {
    for (int i = tid; i < g.N; i += 256) tab[i] = g.bias[i];
  } else if constexpr (EPI == PEPI_DGELU) {
    for (int i = tid; i < g.N; i += 256) tab[i] = 0.f;
  }
  __syncthreads();

  const unsigned smem_base = lds_addr(smem);
  const unsigned tab_lds = smem_base + POFF_TAB;
  const unsigned sw = (unsigned)((r >> 2) & 3);
  // fragment byte offsets inside a stage for k-halves s = 0 / 1 (frag i / j adds 2048 i / 2048 j)
  const unsigned fa0 = (unsigned)((wm * 128 + r) * ROWB) + ((((unsigned)h) ^ sw) << 4);
  const unsigned fb0 = (unsigned)(PA_BYTES + (wn * 192 + r) * ROWB) + ((((unsigned)h) ^ sw) << 4);
#define fa1 (fa0 ^ 32u)  /* chunk 2 + h sits at position (h ^ sw) ^ 2 */
#define fb1 (fb0 ^ 32u)

  // ---- DMA issue cursor: this wave's pieces are p = wave + 4q; q < 4 -> A rows 16p.., q >= 4 -> B rows 16(p-16)..
  const int prow = lane >> 2;
  const int csrc = (lane & 3) ^ ((lane >> 4) & 3);  // logical chunk that must land at position lane & 3
  // addresses are (wave-uniform 64-bit base) + (32-bit per-lane byte offset): the per-piece / per-step / per-tile
  // arithmetic stays on the scalar unit and one VGPR per operand holds the lane part (saddr addressing)
  const unsigned laneA = (unsigned)(((wave * 16 + prow) * g.lda + 8 * csrc) * 2);
  const unsigned laneB = (unsigned)(((wave * 16 + prow) * g.ldb + 8 * csrc) * 2);
  const int64_t strideA = (int64_t)64 * g.lda * 2, strideB = (int64_t)64 * g.ldb * 2;  // 64 rows, bytes
  int ist = 0, iks = 0, itile = t0;
  const char* ia = reinterpret_cast<const char*>(g.A) + (int64_t)(itile / g.tiles_n) * PBM * g.lda * 2;
  const char* ib = reinterpret_cast<const char*>(g.B) + (int64_t)(itile % g.tiles_n) * PBN * g.ldb * 2;
  auto issue = [&]() {
    char* sb = smem + ist * PSTAGE + wave * 1024;
    const char* pa = ia + iks * 64;
    const char* pb = ib + iks * 64;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa + q * strideA + laneA),
                                       (__attribute__((address_space(3))) void*)(sb + q * 4096), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < 6; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb + q * strideB + laneB),
                                       (__attribute__((address_space(3))) void*)(sb + PA_BYTES + q * 4096), 16, 0, 0);
    ist = (ist + 1 == PNST) ? 0 : ist + 1;
    if (++iks == nk) {
      if (itile + 1 < t1) {
        ++itile;
        iks = 0;
        ia = reinterpret_cast<const char*>(g.A) + (int64_t)(itile / g.tiles_n) * PBM * g.lda * 2;
        ib = reinterpret_cast<const char*>(g.B) + (int64_t)(itile % g.tiles_n) * PBN * g.ldb * 2;
      } else {
        iks = nk - 1;  // past the end of the stream: re-fetch the last K-step (keeps the vmcnt arithmetic uniform)
      }
    }
  };

  f32x16 acc[4][6];
  // Fragment registers (compile-time indices only): A of the even / odd k-half, B of column fragments 0-2 / 3-5 in
  // ping-pong -> 56 registers next to the 384 accumulators.  A K-step is four passes of 12 MFMAs
  // (half 0 | j 0-2, half 0 | j 3-5, half 1 | j 0-2, half 1 | j 3-5); the operands of pass k+1 are read under the
  // MFMAs of pass k.
  u32x4 ae[4], ao[4], bp[3], bq[3];
#define PANEL_RD_A(AF, STG, FA)                                        \
  {                                                                    \
    const unsigned a_ = smem_base + (unsigned)(STG) * PSTAGE + (FA);   \
    AF[0] = lds_read128<0>(a_);                                        \
    AF[1] = lds_read128<2048>(a_);                                     \
    AF[2] = lds_read128<4096>(a_);                                     \
    AF[3] = lds_read128<6144>(a_);                                     \
  }
#define PANEL_RD_B(BF, STG, FB, J0)                                    \
  {                                                                    \
    const unsigned b_ = smem_base + (unsigned)(STG) * PSTAGE + (FB);   \
    BF[0] = lds_read128<(J0) * 2048>(b_);                              \
    BF[1] = lds_read128<(J0) * 2048 + 2048>(b_);                       \
    BF[2] = lds_read128<(J0) * 2048 + 4096>(b_);                       \
  }
  // pass over column fragments 0-2 (all AGPR accumulators) / 3-5 (3 in AGPRs, 4 and 5 in VGPRs)
#define PANEL_MMA_LO(AF, BF)                              \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {         \
    mfma_ag(acc[i][0], BF[0], AF[i]);                     \
    mfma_ag(acc[i][1], BF[1], AF[i]);                     \
    mfma_ag(acc[i][2], BF[2], AF[i]);                     \
  }
#define PANEL_MMA_HI(AF, BF)                              \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {         \
    mfma_ag(acc[i][3], BF[0], AF[i]);                     \
    mfma_vg(acc[i][4], BF[1], AF[i]);                     \
    mfma_vg(acc[i][5], BF[2], AF[i]);                     \
  }
#define PANEL_SYNC_LDS() \
  wait_lgkm0();          \
  __builtin_amdgcn_sched_barrier(0);

  // ---- prologue: three K-steps in flight, operands of the first pass
  int tile = t0, ks = 0;
  int n_w = (tile % g.tiles_n) * PBN + wn * 192;
  panel_init_acc<EPI>(acc, tab_lds, n_w, h);
  issue();
  issue();
  issue();
  wait_vm<2 * PPIECES>();
  __builtin_amdgcn_s_barrier();
  PANEL_RD_A(ae, 0, fa0)
  PANEL_RD_B(bp, 0, fb0, 0)
  PANEL_SYNC_LDS()

  int cst = 0;        // stage of the step being consumed
  int after_epi = 0;  // K-steps since an epilogue whose stores may still be in flight (2, 1, 0)
#pragma clang loop unroll(disable)
  for (; tile < t1; ++tile) {
#pragma clang loop unroll(disable)
    for (ks = 0; ks < nk; ++ks) {
      const int nst = (cst + 1 == PNST) ? 0 : cst + 1;
      PANEL_RD_B(bq, cst, fb0, 3)
      PANEL_MMA_LO(ae, bp)
      PANEL_SYNC_LDS()
      PANEL_RD_A(ao, cst, fa1)
      PANEL_RD_B(bp, cst, fb1, 0)
      PANEL_MMA_HI(ae, bq)
      PANEL_SYNC_LDS()
      PANEL_RD_B(bq, cst, fb1, 3)
      PANEL_MMA_LO(ao, bp)
      PANEL_SYNC_LDS()  // this wave has read everything it needs from stage cst
      // step s+1 has landed (for this wave): the younger operations are step s+2's pieces and, right after a
      // tile end, that tile's stores
      if (after_epi) {
        wait_vm<WAIT_EPI>();
        --after_epi;
      } else {
        wait_vm<PPIECES>();
      }
      __builtin_amdgcn_s_barrier();  // ... for every wave; and every wave is done with stage cst
      issue();                       // step s+3 -> stage cst
      PANEL_RD_A(ae, nst, fa0)       // first operands of the next step (possibly of the next tile)
      PANEL_RD_B(bp, nst, fb0, 0)
      PANEL_MMA_HI(ao, bq)
      PANEL_SYNC_LDS()
      cst = nst;
      // The settle belongs INSIDE the loop: register-allocator spill code for the accumulators lands on the loop's
      // exit edge, i.e. before anything written after the loop, and would read a tile still in the MFMA pipe.
      if (ks == nk - 1) PANEL_MFMA_SETTLE();
    }
    panel_epilogue<EPI>(acc, g, (tile / g.tiles_n) * PBM + wm * 128, n_w, tab_lds, r, h);
    n_w = ((tile + 1) % g.tiles_n) * PBN + wn * 192;
    panel_init_acc<EPI>(acc, tab_lds, n_w, h);
    after_epi = 2;
  }
  wait_vm<0>();  // the re-fetched tail pieces must land before this workgroup's LDS is handed on
#undef PANEL_RD_A
#undef PANEL_RD_B
#undef PANEL_MMA_LO
#undef PANEL_MMA_HI
#undef PANEL_SYNC_LDS
#undef fa1
#undef fb1
  if constexpr (EPI == PEPI_DGELU) {
    __syncthreads();
    // this workgroup's tiles cover columns [n_lo, n_hi) of possibly several row panels: flush the LDS sums
    for (int i = tid; i < g.N; i += 256) {
      const float v = tab[i];
      if (v != 0.f) atomicAdd(g.colsum + i, v);
    }
  }
}

template <int EPI>
int launch_panel(const PanelArgs& g, hipStream_t st) {
  auto kern = gemm_panel_kernel<EPI>;
  constexpr int LDS = POFF_TAB + PMAX_N * 4;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  int grid = g.ntiles < 256 ? g.ntiles : 256;  // persistent: one workgroup per CU walks its run of tiles
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, st, g);
  UWU_LAUNCH_CHECK("gemm_panel");
  return UWU_OK;
}

}  // namespace

// Host gate (called from uwu_gemm): bf16 in/out, K-contiguous operands, whole tiles only.
bool uwu_gemm_panel_ok(int M, int N, int K, int lda, int ldb, int ldc, int ldaux, int dtype, int c_dtype, int epilogue,
                       const void* A, const void* B, const void* C, const void* C2, const void* aux) {
  // Opt-in (UWU_GEMM_PANEL=1, read per call so a test can switch it): on the denoiser's shapes this kernel measured
  // SLOWER than gemm.hip's 128x128 kernel (65536x1152x384: 125 us against 108 us) although it moves 2.4x fewer
  // bytes through L2 -- DESIGN.md section 4.1 has the phase probes (tools/panel_probe.sh) and the reasons.
  const char* e = getenv("UWU_GEMM_PANEL");
  if (!(e && e[0] == '1')) return false;
  if (dtype != UWU_BF16 || c_dtype != UWU_BF16) return false;
  if (epilogue != UWU_EPI_NONE && epilogue != UWU_EPI_BIAS && epilogue != UWU_EPI_BIAS_GELU &&
      epilogue != UWU_EPI_DGELU)
    return false;
  if (epilogue == UWU_EPI_DGELU && !C2) return false;  // this kernel's dGELU always produces the column sums
  if (M % PBM || N % PBN || K % 32 || K < 128 || N > PMAX_N) return false;
  if ((M / PBM) * (N / PBN) < 128) return false;  // too few tiles to fill the chip: gemm.hip's 128x128 tiles
  if (lda % 8 || ldb % 8 || ldc % 8 || (epilogue == UWU_EPI_DGELU && ldaux % 8)) return false;
  if ((int64_t)M * lda >= (1ll << 31) || (int64_t)N * ldb >= (1ll << 31)) return false;  // 32-bit lane offsets
  uintptr_t al = (uintptr_t)A | (uintptr_t)B | (uintptr_t)C;
  if (epilogue == UWU_EPI_BIAS_GELU) al |= (uintptr_t)C2;
  if (epilogue == UWU_EPI_DGELU) al |= (uintptr_t)aux;
  return (al & 15) == 0;
}

int uwu_gemm_panel(const void* A, const void* B, void* C, void* C2, const float* bias, const void* aux, int M, int N,
                   int K, int lda, int ldb, int ldc, int ldaux, int epilogue, hipStream_t st) {
  PanelArgs g;
  g.A = static_cast<const bf16_t*>(A);
  g.B = static_cast<const bf16_t*>(B);
  g.C = static_cast<bf16_t*>(C);
  g.C2 = static_cast<bf16_t*>(C2);
  g.bias = bias;
  g.aux = static_cast<const bf16_t*>(aux);
  g.colsum = nullptr;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = ldaux;
  g.tiles_n = N / PBN;
  g.ntiles = (M / PBM) * g.tiles_n;
  switch (epilogue) {
    case UWU_EPI_NONE: return launch_panel<PEPI_NONE>(g, st);
    case UWU_EPI_BIAS: return launch_panel<PEPI_BIAS>(g, st);
    case UWU_EPI_BIAS_GELU: return launch_panel<PEPI_BIAS_GELU>(g, st);
    case UWU_EPI_DGELU:
      g.colsum = reinterpret_cast<float*>(C2);  // optional float[N] += column sums (gemm.hip's convention)
      g.C2 = nullptr;
      if (g.colsum) return launch_panel<PEPI_DGELU>(g, st);
      break;
  }
  uwu_set_error("gemm_panel: unsupported epilogue %d", epilogue);
  return UWU_EINVAL;
}
