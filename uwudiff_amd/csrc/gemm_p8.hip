// 256 x 256 bf16 MFMA GEMM for long contractions (K >= 512): the Linears of DiT-B/2, DiT-XL/2 and the SDXL-shape UNet
// (BASELINE configs 3-5; reference: nn.Linear fwd / bwd inside the transformer blocks, src/duwu/modules/rope_unet.py:122-166,
// 393-411).  gemm_big_kernel (gemm.hip) runs the same tile as a two-stage loop -- one vmcnt(0) + barrier per 64-deep K step, so
// every step exposes what is left of a load latency and all eight waves read LDS, then all eight issue MFMAs.  This kernel is
// the "8-phase" schedule of cdna_hip_programming.md section 5 rebuilt for these operand layouts:
//
//   * the operands of a K step (BK = 64: 128-byte rows) are four HALF-TILES of 128 rows x 128 B = 16 KB: B0 A0 B1 A1; LDS is a
//     ring of eight half-tile slots (two K steps, 128 KB).  The stream of half-tiles is filled by LDS-DMA SEVEN elements ahead of
//     the K step being multiplied, one element (two DMA instructions per thread) per phase; the only wait is a counted vmcnt
//     once per K step that leaves the three youngest half-tiles in flight across the barriers.
//   * a wave owns 64 rows of EACH A half and 32 columns of EACH B half (2 x 2 quadrants of 64 x 32 = 4 x 2 fragments): phase p of
//     a K step multiplies one quadrant over the whole K step (16 MFMAs 16x16x32): (a0,b0) (a0,b1) (a1,b1) (a1,b0).  So every
//     half-tile is read in ONE phase (B0 + A0 in phase 0, B1 in 1, A1 in 2), its slot is free right after and is refilled one
//     or two phases later with the element eight positions on.
//   * waves 0-3 and waves 4-7 (SIMD partners) run ONE BARRIER APART: each phase is { LDS reads + DMA issue | barrier | MFMAs |
//     barrier }, so while one wave of a SIMD issues its 16 MFMAs the other one reads its fragments and issues its DMA -- the
//     matrix pipe of a SIMD is handed back and forth and never waits for LDS.
//
// Ordering argument (what makes the reads and refills safe; B(k) = k-th workgroup barrier, group 0 = waves 0-3 runs phase q
// between B(2q-1) and B(2q+1), group 1 one barrier later):
//   RAW  the wait of K step t (its phase 3, in front of that phase's first barrier) retires this wave's share of every element
//        up to (t+1, A1).  Group 0 reads K step t+1 after B(8t+7), which group 1 only reaches after ITS wait; group 1 reads after
//        B(8t+8); group 0's wait came before B(8t+6).
//   WAR  element n + 8 is issued in the phase after n - 1's ... see the table in issue_for(): B0's slot is refilled ONE phase
//        after its reads, so phase 0 retires the four B0 reads (issued first, lgkmcnt(8)) in front of its first barrier; the
//        other three slots are refilled two phases after their reads, which a later barrier of the reading group separates.
//
// TB = 1 (input gradients, W stored [K][N]): the B half-tiles are two [32 k][128 n] sub-images filled untouched and read by
// ds_read_b64_tr_b16, exactly as in gemm_big_kernel / gemm_r3_kernel.
#include "gemm_shared.h"

namespace {

constexpr int P8_HT = 128 * ROW_BYTES;  // half-tile: 128 rows x 128 B
constexpr int P8_LDS = 8 * P8_HT;       // ring of eight half-tile slots (128 KB)

template <int H>
using IC = std::integral_constant<int, H>;

template <typename TC, int EPI, bool TB>
__global__ void __launch_bounds__(512, 2) gemm_p8_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int nblk = g.tiles_m * g.tiles_n;
  int tile;
  {  // XCD-aware tile order as in gemm_kernel
    const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3;
    const int q = nblk >> 3, rm = nblk & 7;
    tile = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + loc;
  }
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  const int nk = g.K >> 6;

  // ---- LDS-DMA sources: a half-tile = 16 pieces of 8 rows x 128 B; this thread moves pieces wave and wave + 8 ----------------
  const T* pa[2][2];
  const T* pb[2][2];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = 8 * (wave + 8 * q) + (lane >> 3);
      const int c = ((lane & 7) ^ (row >> 1) ^ (row >> 4)) & 7;  // logical chunk that must land at position lane & 7
      int ga = m0 + 128 * half + row;
      if (ga >= g.M) ga = g.M - 1;  // (clamped rows / columns: their products are never stored)
      pa[half][q] = static_cast<const T*>(g.A) + (int64_t)ga * g.lda + 8 * c;
      if constexpr (!TB) {
        int gb = n0 + 128 * half + row;
        if (gb >= g.N) gb = g.N - 1;
        pb[half][q] = static_cast<const T*>(g.B) + (int64_t)gb * g.ldb + 8 * c;
      } else {  // q = k half of the K step: sub-image q, piece wave = k-rows 4 wave .. + 3, 256 B each
        const int drow = lane >> 4;
        const int dchunk = (lane & 15) ^ (((drow & 3) << 2) | (wave & 3));
        int x = n0 + 128 * half + 8 * dchunk;
        if (x > g.N - 8) x = g.N - 8;
        pb[half][q] = static_cast<const T*>(g.B) + (int64_t)(32 * q + 4 * wave + drow) * g.ldb + x;
      }
    }
  }
  const int64_t bstep = TB ? (int64_t)64 * g.ldb : 64;
  // element h of K step t into slot (par, h); h: 0 = B0, 1 = A0, 2 = B1, 3 = A1
  auto issue = [&](auto hc, auto pc, int t) __attribute__((always_inline)) {
    constexpr int h = decltype(hc)::value, par = decltype(pc)::value, half = h >> 1;
    char* slot = smem + (par * 4 + h) * P8_HT;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const T* src = (h & 1) ? pa[half][q] + (int64_t)t * 64 : pb[half][q] + t * bstep;
      char* dst = (!(h & 1) && TB) ? slot + q * R_BSUB + wave * 1024 : slot + (wave + 8 * q) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  // ---- fragment read addresses (parity 0; parity 1 = + 4 slots) ---------------------------------------------------------------
  const unsigned smem_base = (unsigned)(size_t)((const __attribute__((address_space(3))) char*)smem);
  unsigned a_ad[4][2], b_ad[2][2];  // [fragment][kk] inside a half-tile: A rows 64 grp + 16 i + fr, B rows 32 wc + 16 j + fr
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a_ad[i][kk] = smem_base + (unsigned)swz(64 * grp + 16 * i + fr, 4 * kk + fq);
#pragma unroll
    for (int j = 0; j < 2; ++j) b_ad[j][kk] = smem_base + (unsigned)swz(32 * wc + 16 * j + fr, 4 * kk + fq);
  }
  unsigned bt_ad[2][2];  // TB: [fragment j][transposed read t] inside a [32 k][128 n] sub-image
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int t = 0; t < 2; ++t) bt_ad[j][t] = smem_base + tr_lane_base(lane, t, 4 * wc + 2 * j);

  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[x][y][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 af[4][2], bf0[2][2], bf1[2][2];

  auto read_a = [&](auto slotc) __attribute__((always_inline)) {  // 8 reads
    constexpr unsigned hi = decltype(slotc)::value >= 4 ? 65536u : 0u, off = decltype(slotc)::value * P8_HT - hi;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i][kk] = r_read128<off>(a_ad[i][kk] + hi);
  };
  auto read_b = [&](auto slotc, uint4 (&bf)[2][2]) __attribute__((always_inline)) {  // 4 reads (TB: 8 transposing reads)
    constexpr unsigned up = decltype(slotc)::value >= 4 ? 65536u : 0u, off = decltype(slotc)::value * P8_HT - up;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (!TB) {
          bf[j][kk] = r_read128<off>(b_ad[j][kk] + up);
        } else if (kk == 0) {
          const uint2 lo = t_read_tr<off>(bt_ad[j][0] + up), hi = t_read_tr<off>(bt_ad[j][1] + up);
          bf[j][kk] = uint4{lo.x, lo.y, hi.x, hi.y};
        } else {
          const uint2 lo = t_read_tr<off + R_BSUB>(bt_ad[j][0] + up), hi = t_read_tr<off + R_BSUB>(bt_ad[j][1] + up);
          bf[j][kk] = uint4{lo.x, lo.y, hi.x, hi.y};
        }
      }
  };
  auto mma_quadrant = [&](f32x4 (&c)[4][2], const uint4 (&bf)[2][2]) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma_frag<T>(bf[j][kk], af[i][kk], c[i][j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // One K step (parity PAR static).  Phase p issues element (t, p) + 7:  p = 0 -> (t + 1, A1) into the OTHER parity (read last
  // in phase 2 of step t - 1), p = 1 -> (t + 2, B0) over (t, B0) (read in phase 0: retired there by lgkmcnt(8)), p = 2 ->
  // (t + 2, A0) over (t, A0) (phase 0), p = 3 -> (t + 2, B1) over (t, B1) (phase 1).
  auto kstep = [&](auto pc, int t) __attribute__((always_inline)) {
    constexpr int par = decltype(pc)::value;
    const bool more1 = t + 1 < nk, more2 = t + 2 < nk;
    // phase 0: (a0, b0)
    read_b(IC<par * 4 + 0>{}, bf0);
    read_a(IC<par * 4 + 1>{});
    if (more1) issue(IC<3>{}, IC<par ^ 1>{}, t + 1);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");  // the B0 reads are done: its slot is refilled in the next phase
    bar();
    mma_quadrant(acc[0][0], bf0);
    bar();
    // phase 1: (a0, b1)
    read_b(IC<par * 4 + 2>{}, bf1);
    if (more2) issue(IC<0>{}, IC<par>{}, t + 2);
    bar();
    mma_quadrant(acc[0][1], bf1);
    bar();
    // phase 2: (a1, b1)
    read_a(IC<par * 4 + 3>{});
    if (more2) issue(IC<1>{}, IC<par>{}, t + 2);
    bar();
    mma_quadrant(acc[1][1], bf1);
    bar();
    // phase 3: (a1, b0) -- no LDS reads; the wait for K step t + 1
    if (more2) {
      issue(IC<2>{}, IC<par>{}, t + 2);
      r_wait_vm<6>();
    } else {
      r_wait_vm<0>();
    }
    bar();
    mma_quadrant(acc[1][0], bf0);
    bar();
  };

  // ---- prologue: elements 0 .. 6 (K steps 0 and, but for its A1, 1); K step 0 has landed when all but three have ------------
  issue(IC<0>{}, IC<0>{}, 0);
  issue(IC<1>{}, IC<0>{}, 0);
  issue(IC<2>{}, IC<0>{}, 0);
  issue(IC<3>{}, IC<0>{}, 0);
  if (nk > 1) {
    issue(IC<0>{}, IC<1>{}, 1);
    issue(IC<1>{}, IC<1>{}, 1);
    issue(IC<2>{}, IC<1>{}, 1);
    r_wait_vm<6>();
  } else {
    r_wait_vm<0>();
  }
  bar();
  if (grp == 1) bar();  // waves 4-7 run one barrier behind their SIMD partners
  for (int t = 0; t < nk; t += 2) {
    kstep(IC<0>{}, t);
    if (t + 1 < nk) kstep(IC<1>{}, t + 1);
  }
  if (grp == 0) bar();  // every wave has passed the same number of barriers; nobody reads LDS any more

  // ---- epilogue: four 64 x 32 quadrants per wave ------------------------------------------------------------------------
  f32x4 cs[2][2];
#pragma unroll
  for (int y = 0; y < 2; ++y)
#pragma unroll
    for (int j = 0; j < 2; ++j) cs[y][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int x = 0; x < 2; ++x) {
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int m_q = m0 + 128 * x + 64 * grp, n_q = n0 + 128 * y + 32 * wc;
      EpiPre<T, 4, 2> pre;
      epi_prefetch<T, 4, 2, EPI>(pre, g, m_q, n_q, fr, fq);
      epilogue_tile<T, TC, 4, 2, EPI>(acc[x][y], pre, g, m_q, n_q, fr, fq, nullptr, 0, 0, -1, cs[y]);
    }
  }
  if constexpr (EPI == UWU_EPI_DGELU) {
    float* colsum = reinterpret_cast<float*>(g.C2);
    if (colsum) {  // uniform.  The two wave groups cover the same columns: they meet in LDS, 256 threads issue one atomic each
      float* cs_lds = reinterpret_cast<float*>(smem);
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x4 v = cs[y][j];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = row16_sum(v[e]);
          if (fr == 0) store4(cs_lds + 256 * grp + 128 * y + 32 * wc + 16 * j + 4 * fq, v);
        }
      __syncthreads();
      if (tid < 256 && n0 + tid < g.N) atomicAdd(colsum + n0 + tid, cs_lds[tid] + cs_lds[256 + tid]);
    }
  }
}

template <typename TC, int EPI, bool TB>
int launch_p8(GemmArgs g, hipStream_t st) {
  auto kern = gemm_p8_kernel<TC, EPI, TB>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
    attr_done = true;
  }
  g.tiles_m = (g.M + 255) / 256;
  g.tiles_n = (g.N + 255) / 256;
  UwuProfScope prof(st);
  hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n), dim3(512), P8_LDS, st, g);
  prof.done(gemm_tag(g, TB, false), 0, 2.0 * g.M * g.N * g.K, gemm_bytes(g, 2, sizeof(TC)));
  UWU_LAUNCH_CHECK("gemm_p8");
  return UWU_OK;
}

}  // namespace

// bf16 in / bf16 out, K a multiple of 64, 16-byte addressable operands.  UWU_GEMM_P8=0: off, =1: every shape it can run
// (tests, A/B comparisons); default: K >= 512 and at least one tile per CU.
bool uwu_gemm_p8_ok(const GemmArgs& g, bool tb) {
  static UwuEnv on("UWU_GEMM_P8"), kmin_e("UWU_P8_KMIN");
  if (on.get().is('0')) return false;
  if (g.K % 64 || g.K < 128) return false;
  if ((((uintptr_t)g.A | (uintptr_t)g.B) & 15) || g.lda % 8 || g.ldb % 8) return false;
  if (tb && (g.N % 8 || g.N < 8)) return false;
  if (g.epi != UWU_EPI_NONE && g.epi != UWU_EPI_BIAS && g.epi != UWU_EPI_BIAS_GELU && g.epi != UWU_EPI_DGELU) return false;
  if (tb ? (g.epi != UWU_EPI_NONE && g.epi != UWU_EPI_DGELU) : g.epi == UWU_EPI_DGELU) return false;
  if (on.is('1')) return true;
  const int kmin = kmin_e.get().set ? kmin_e.ival : 512;
  const int64_t tiles = (int64_t)((g.M + 255) / 256) * ((g.N + 255) / 256);
  // padded column tiles: at most 1/8 of the columns may be padding (N = 1152 -> 5 tiles of 256: 10 %)
  const int64_t npad = (int64_t)((g.N + 255) / 256) * 256;
  return g.K >= kmin && tiles >= 256 && (npad - g.N) * 8 <= npad;
}

int uwu_launch_gemm_p8(const GemmArgs& g, bool tb, hipStream_t st) {
  if (!tb) {
    if (g.epi == UWU_EPI_NONE) return launch_p8<bf16_t, UWU_EPI_NONE, false>(g, st);
    if (g.epi == UWU_EPI_BIAS) return launch_p8<bf16_t, UWU_EPI_BIAS, false>(g, st);
    if (g.epi == UWU_EPI_BIAS_GELU) return launch_p8<bf16_t, UWU_EPI_BIAS_GELU, false>(g, st);
  } else {
    if (g.epi == UWU_EPI_NONE) return launch_p8<bf16_t, UWU_EPI_NONE, true>(g, st);
    if (g.epi == UWU_EPI_DGELU) return launch_p8<bf16_t, UWU_EPI_DGELU, true>(g, st);
  }
  uwu_set_error("gemm_p8: epilogue %d not instantiated (tb=%d)", g.epi, (int)tb);
  return UWU_EINVAL;
}
