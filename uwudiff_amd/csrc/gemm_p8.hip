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
constexpr int P8_RING = 8 * P8_HT;     // ring of eight half-tile slots (128 KB)
constexpr int P8_BIAS = P8_RING + 2048;  // behind the dGELU column sums of a tile (2 x 256 floats): 8 waves x 64 bias floats
constexpr int P8_LDS = P8_BIAS + 2048;

template <int H>
using IC = std::integral_constant<int, H>;

// LDS image of a K-contiguous half-tile: [128 rows][128 B], 16-byte chunk c of row r at position c ^ ((r >> 1) & 7).  The
// ds_read_b128 lane groups (16 lanes: rows fr of one parity pair set, chunk 4 kk + fq) then cover all 64 banks once, and -- unlike
// swz() of gemm_shared.h, whose (r >> 4) term serves register-staged transposed writes -- the address of fragment i is the
// address of fragment 0 plus 2048 i: one address register per operand and k half instead of one per fragment.
__device__ __forceinline__ int p8_swz(int row, int chunk) { return row * ROW_BYTES + (((chunk ^ (row >> 1)) & 7) << 4); }

// ABL: timing-only builds (UWU_P8_ABL, plain forward only): 1 = no output stores, 2 = two K steps, 3 = wave groups in step,
// 4 = no MFMAs, 5 = no LDS-DMA inside the K loop, 6 = no fragment reads
//
// PERSISTENT: grid = one workgroup per CU; a workgroup walks tiles L = blockIdx.x, + gridDim.x, ... (same XCD-chunked order as
// the one-tile-per-workgroup kernels).  After the K loop of a tile the first seven half-tiles of the NEXT tile are requested
// BEFORE the epilogue of this one: the first fetch of a tile and the drain of the previous tile's stores -- a third of the
// launch at K = 768 when every workgroup pays them one after the other (ablations: 345 us = 261 without stores; 146 with two
// K steps) -- run under each other and under the first K steps.  Vector-memory operations retire in issue order, so the wait
// for K step 0 of the next tile is vmcnt(6 + S) with S = the stores this wave is known to have issued after the request
// (16, 32 with two outputs, 0 for a ragged tile: waiting for more than needed is always safe).
template <typename TC, int EPI, bool TB, int ABL = 0>
__global__ void __launch_bounds__(512, 2) gemm_p8_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int nblk = g.tiles_m * g.tiles_n;
  const int nk = ABL == 2 ? 2 : g.K >> 6;
  auto tile_of = [&](int L) __attribute__((always_inline)) {  // XCD-aware tile order as in gemm_kernel
    const int xcd = L & 7, loc = L >> 3;
    const int q = nblk >> 3, rm = nblk & 7;
    return (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + loc;
  };

  // ---- LDS-DMA sources: a half-tile = 16 pieces of 8 rows x 128 B; this thread moves pieces wave and wave + 8 ----------------
  // (32-bit byte offsets from wave-uniform bases: global_load_lds in its SGPR-base form, 8 address registers instead of 16)
  unsigned oa[2][2], ob[2][2];
  const char* abase;
  const char* bbase;
  auto setup = [&](int tile, int& m0, int& n0) __attribute__((always_inline)) {
    const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
    m0 = tm * 256;
    n0 = tn * 256;
    abase = reinterpret_cast<const char*>(static_cast<const T*>(g.A) + (int64_t)m0 * g.lda);
    bbase = reinterpret_cast<const char*>(static_cast<const T*>(g.B) + (TB ? (int64_t)n0 : (int64_t)n0 * g.ldb));
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int row = 8 * (wave + 8 * q) + (lane >> 3);
        const int c = ((lane & 7) ^ (row >> 1)) & 7;  // logical chunk that must land at position lane & 7 (p8_swz)
        int ga = 128 * half + row;
        if (ga >= g.M - m0) ga = g.M - m0 - 1;  // (clamped rows / columns: their products are never stored)
        oa[half][q] = (unsigned)(ga * g.lda + 8 * c) * 2u;
        if constexpr (!TB) {
          int gb = 128 * half + row;
          if (gb >= g.N - n0) gb = g.N - n0 - 1;
          ob[half][q] = (unsigned)(gb * g.ldb + 8 * c) * 2u;
        } else {  // q = k half of the K step: sub-image q, piece wave = k-rows 4 wave .. + 3, 256 B each
          const int drow = lane >> 4;
          const int dchunk = (lane & 15) ^ (((drow & 3) << 2) | (wave & 3));
          int x = 128 * half + 8 * dchunk;
          if (x > g.N - n0 - 8) x = g.N - n0 - 8;
          ob[half][q] = (unsigned)((32 * q + 4 * wave + drow) * g.ldb + x) * 2u;
        }
      }
    }
  };
  const int64_t bstep = TB ? (int64_t)128 * g.ldb : 128;  // bytes per K step
  // element h of K step t into slot (par, h); h: 0 = B0, 1 = A0, 2 = B1, 3 = A1
  auto issue = [&](auto hc, auto pc, int t) __attribute__((always_inline)) {
    constexpr int h = decltype(hc)::value, par = decltype(pc)::value, half = h >> 1;
    char* slot = smem + (par * 4 + h) * P8_HT;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const char* src = (h & 1) ? abase + (int64_t)t * 128 + oa[half][q] : bbase + t * bstep + ob[half][q];
      char* dst = (!(h & 1) && TB) ? slot + q * R_BSUB + wave * 1024 : slot + (wave + 8 * q) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  // ---- fragment read addresses (slots 0-3; slots 4-7 sit 64 KB up, beyond the 16-bit offset field: second base) ---------------
  const unsigned smem_base = (unsigned)(size_t)((const __attribute__((address_space(3))) char*)smem);
  unsigned a_ad[2], b_ad[2];  // [kk]: fragment 0 inside a half-tile (A rows 64 grp + fr, B rows 32 wc + fr); fragment i: + 2048 i
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    a_ad[kk] = smem_base + (unsigned)p8_swz(64 * grp + fr, 4 * kk + fq);
    b_ad[kk] = smem_base + (unsigned)p8_swz(32 * wc + fr, 4 * kk + fq);
  }
  unsigned bt_ad[2][2];  // TB: [fragment j][transposed read t] inside a [32 k][128 n] sub-image
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int t = 0; t < 2; ++t) bt_ad[j][t] = smem_base + tr_lane_base(lane, t, 4 * wc + 2 * j);

  f32x4 acc[2][2][4][2];
  uint4 af[4][2], bf0[2][2], bf1[2][2];

  auto read_a = [&](auto slotc) __attribute__((always_inline)) {  // 8 reads
    if constexpr (ABL == 6) return;
    constexpr unsigned hi = decltype(slotc)::value >= 4 ? 65536u : 0u, off = decltype(slotc)::value * P8_HT - hi;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i == 0) af[i][kk] = r_read128<off>(a_ad[kk] + hi);
        if (i == 1) af[i][kk] = r_read128<off + 2048>(a_ad[kk] + hi);
        if (i == 2) af[i][kk] = r_read128<off + 4096>(a_ad[kk] + hi);
        if (i == 3) af[i][kk] = r_read128<off + 6144>(a_ad[kk] + hi);
      }
  };
  auto read_b = [&](auto slotc, uint4 (&bf)[2][2]) __attribute__((always_inline)) {  // 4 reads (TB: 8 transposing reads)
    if constexpr (ABL == 6) return;
    constexpr unsigned up = decltype(slotc)::value >= 4 ? 65536u : 0u, off = decltype(slotc)::value * P8_HT - up;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (!TB) {
          bf[j][kk] = j == 0 ? r_read128<off>(b_ad[kk] + up) : r_read128<off + 2048>(b_ad[kk] + up);
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
        for (int j = 0; j < 2; ++j) {
          if constexpr (ABL == 4) asm volatile("" ::"v"(*reinterpret_cast<const gu32x4*>(&bf[j][kk])), "v"(*reinterpret_cast<const gu32x4*>(&af[i][kk])));
          else mma_frag<T>(bf[j][kk], af[i][kk], c[i][j]);
        }
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
  constexpr bool has_bias = EPI == UWU_EPI_BIAS || EPI == UWU_EPI_BIAS_GELU || EPI == UWU_EPI_BIAS_SILU;
  constexpr int NST = (EPI == UWU_EPI_BIAS_GELU || EPI == UWU_EPI_BIAS_SILU) ? 32 : 16;  // stores of a full tile per wave
  // CONTINUOUS mode (K steps even, no epilogue loads): the element stream runs on into the NEXT tile -- the last two K steps of a
  // tile request the first seven half-tiles of the next one -- and the finished accumulators leave quadrant by quadrant in the
  // load intervals of the next tile's K step 0 (quadrant q is final after phase q of the last K step and needed again in phase
  // q of the next K step 0), so a tile boundary has no drain, no refill and no extra barrier.  Otherwise (dGELU: its epilogue
  // loads would drain the queue; odd K step counts) a tile ends with the two wave groups back in step, requests the next tile
  // ahead of its epilogue and starts over.
  constexpr bool can_cont = EPI != UWU_EPI_DGELU && ABL != 3;
  const bool cont = can_cont && g.p8_cont && !(nk & 1) && nk >= 4;
  int s_prev = 0;       // stores this wave issued behind the request for K step 1 of the tile (0 unless all are known to exist)
  bool pending = false; // the accumulators still hold the PREVIOUS tile (continuous mode)
  int m0, n0, em0 = 0, en0 = 0;  // tile being loaded / tile whose results are in the accumulators
  int L = blockIdx.x, Ln = 0;
  bool stream = false;
  const unsigned bias_ad = smem_base + P8_BIAS + wave * 256 + 16 * fq;

  // bias of a tile: 64 floats per wave (its 2 x 32 columns) by LDS-DMA into the wave's own 256 bytes -- an ordinary load's wait
  // becomes vmcnt(0) while LDS-DMA is in flight (cdna_hip_programming.md section 5, trap (b)), this one is just one more
  // element of the stream (the counted waits ignore it: one operation more than needed may be waited for)
  auto bias_dma = [&](int n_tile) __attribute__((always_inline)) {
    if constexpr (has_bias) {
      int n = n_tile + 128 * (lane >> 5) + 32 * wc + (lane & 31);
      if (n >= g.N) n = g.N - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.bias + n),
                                       (__attribute__((address_space(3))) void*)(smem + P8_BIAS + wave * 256), 4, 0, 0);
    }
  };
  f32x4 cs[2][2];  // dGELU column sums
  // epilogue of quadrant (x, y) of tile (em0, en0); the accumulators are zero afterwards
  auto epi_quadrant = [&](auto xc, auto yc) __attribute__((always_inline)) {
    constexpr int x = decltype(xc)::value, y = decltype(yc)::value;
    if constexpr (ABL == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[x][y][i][j]));
    } else {
      const int m_q = em0 + 128 * x + 64 * grp, n_q = en0 + 128 * y + 32 * wc;
      EpiPre<T, 4, 2> pre;
      if constexpr (has_bias) {
        const uint4 b0 = r_read128<128 * y>(bias_ad), b1 = r_read128<128 * y + 64>(bias_ad);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        pre.bias[0] = *reinterpret_cast<const f32x4*>(&b0);
        pre.bias[1] = *reinterpret_cast<const f32x4*>(&b1);
      } else {
        pre.bias[0] = pre.bias[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if constexpr (EPI == UWU_EPI_DGELU) {
#pragma unroll
        for (int i = 0; i < EpiPre<T, 4, 2>::PD; ++i) epi_load_aux_row<T, 4, 2>(g, m_q, n_q, fr, fq, i, pre.aux[i]);
      }
      epilogue_tile<T, TC, 4, 2, EPI>(acc[x][y], pre, g, m_q, n_q, fr, fq, nullptr, 0, 0, -1, EPI == UWU_EPI_DGELU ? cs[y] : nullptr);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[x][y][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // One K step (parity PAR static).  Phase p requests element (t, p) + 7:  p = 0 -> (t + 1, A1) into the OTHER parity (read last
  // in phase 2 of step t - 1), p = 1 -> (t + 2, B0) over (t, B0) (read in phase 0: retired there by lgkmcnt(8)), p = 2 ->
  // (t + 2, A0) over (t, A0) (phase 0), p = 3 -> (t + 2, B1) over (t, B1) (phase 1).  Past the end of the tile the stream goes
  // on with the next tile (`stream`): its sources replace this tile's right after the tile's last element has been requested.
  // (1, A1) of every tile is requested at the tile boundary, ahead of the previous tile's stores, not in phase 0 of K step 0.
  auto kstep = [&](auto pc, int t) __attribute__((always_inline)) {
    constexpr int par = decltype(pc)::value;
    const int t1 = t + 1, t2 = t + 2;
    const bool strm = can_cont && stream;
    const bool iss1 = ABL != 5 && (par == 1 || t != 0) && (t1 < nk || strm), iss2 = ABL != 5 && (t2 < nk || strm);
    const int k1 = t1 < nk ? t1 : t1 - nk, k2 = t2 < nk ? t2 : t2 - nk;
    const bool ep = can_cont && par == 0 && t == 0 && pending;
    // phase 0: (a0, b0)
    if (ep) epi_quadrant(IC<0>{}, IC<0>{});
    read_b(IC<par * 4 + 0>{}, bf0);
    read_a(IC<par * 4 + 1>{});
    if (iss1) issue(IC<3>{}, IC<par ^ 1>{}, k1);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");  // the B0 reads are done: its slot is refilled in the next phase
    bar();
    mma_quadrant(acc[0][0], bf0);
    bar();
    if (par == 0 && strm && t == nk - 2) {  // this tile's last element is on its way: the sources become the next tile's
      em0 = m0;
      en0 = n0;
      setup(tile_of(Ln), m0, n0);
    }
    // phase 1: (a0, b1)
    if (ep) epi_quadrant(IC<0>{}, IC<1>{});
    read_b(IC<par * 4 + 2>{}, bf1);
    if (iss2) issue(IC<0>{}, IC<par>{}, k2);
    bar();
    mma_quadrant(acc[0][1], bf1);
    bar();
    // phase 2: (a1, b1)
    if (ep) epi_quadrant(IC<1>{}, IC<1>{});
    read_a(IC<par * 4 + 3>{});
    if (iss2) issue(IC<1>{}, IC<par>{}, k2);
    bar();
    mma_quadrant(acc[1][1], bf1);
    bar();
    // phase 3: (a1, b0) -- no LDS reads; the wait for K step t + 1
    if (ep) {
      epi_quadrant(IC<1>{}, IC<0>{});
      bias_dma(n0);  // this tile's bias, behind the last read of the previous tile's
    }
    if (iss2) {
      issue(IC<2>{}, IC<par>{}, k2);
      // K step 0: elements (1, *) are OLDER than the previous tile's stores, which may stay in flight with (2, B0 .. B1)
      if (par == 0 && t == 0 && s_prev == NST) r_wait_vm<6 + NST>();
      else r_wait_vm<6>();
    } else {
      r_wait_vm<0>();
    }
    bar();
    mma_quadrant(acc[1][0], bf0);
    bar();
  };
  // elements 0 .. 7 of a tile (K steps 0 and 1: the whole ring; nk >= 2)
  auto prologue = [&]() __attribute__((always_inline)) {
    issue(IC<0>{}, IC<0>{}, 0);
    issue(IC<1>{}, IC<0>{}, 0);
    issue(IC<2>{}, IC<0>{}, 0);
    issue(IC<3>{}, IC<0>{}, 0);
    issue(IC<0>{}, IC<1>{}, 1);
    issue(IC<1>{}, IC<1>{}, 1);
    issue(IC<2>{}, IC<1>{}, 1);
    issue(IC<3>{}, IC<1>{}, 1);
  };
  // ABL 9: wave 0 / wave 4 stamp the 100 MHz clock into g.q8 ([workgroup][tile][16] uint64; UWU_P8_STAMPS=<address>)
  int iter = 0;
  auto stamp = [&](int slot) __attribute__((always_inline)) {
    if constexpr (ABL == 9) {
      if ((wave & 3) == 0 && lane == 0 && iter < 16)
        reinterpret_cast<unsigned long long*>(g.q8)[((size_t)blockIdx.x * 16 + iter) * 16 + 8 * grp + slot] = __builtin_amdgcn_s_memrealtime();
    }
  };
  auto full_tile = [&]() __attribute__((always_inline)) {
    return em0 + 256 <= g.M && en0 + 256 <= g.N && g.wide && sizeof(TC) == 2 && ABL != 1;
  };

#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[x][y][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  setup(tile_of(L), m0, n0);
  bias_dma(n0);
  prologue();
  r_wait_vm<8>();  // K step 0 has landed when all but elements 4-7 have
  bar();
  if (grp == 1 && ABL != 3) bar();  // waves 4-7 run one barrier behind their SIMD partners
  for (;;) {
    stamp(0);
    Ln = L + gridDim.x;
    const bool has_next = Ln < nblk;
    stream = cont && has_next;
    for (int t = 0; t < nk; t += 2) {
      kstep(IC<0>{}, t);
      if (t == 0) stamp(1);
      if (t + 1 < nk) kstep(IC<1>{}, t + 1);
      if (t == 0) stamp(2);
      if (t == 2) stamp(3);
    }
    stamp(4);
    if (can_cont && stream) {
      // (the sources, m0 / n0 and em0 / en0 were switched in K step nk - 2)
      issue(IC<3>{}, IC<1>{}, 1);  // next tile's (1, A1): its slot was last read in phase 2 of the last K step
      pending = true;
      s_prev = full_tile() ? NST : 0;
      L = Ln;
      stamp(6);
      ++iter;
      continue;
    }
    if (grp == 0 && ABL != 3) bar();  // every wave has passed the same number of barriers; nobody reads LDS any more
    em0 = m0;
    en0 = n0;
    // dGELU: the epilogue's aux loads would each drain the queue behind the request (trap (b)): request after the epilogue
    constexpr bool early = EPI != UWU_EPI_DGELU;
    if (early && has_next) {
      setup(tile_of(Ln), m0, n0);
      prologue();
      __builtin_amdgcn_sched_barrier(0);
    }
    stamp(5);
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int j = 0; j < 2; ++j) cs[y][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    epi_quadrant(IC<0>{}, IC<0>{});
    epi_quadrant(IC<0>{}, IC<1>{});
    epi_quadrant(IC<1>{}, IC<1>{});
    epi_quadrant(IC<1>{}, IC<0>{});
    if constexpr (EPI == UWU_EPI_DGELU) {
      float* colsum = reinterpret_cast<float*>(g.C2);
      if (colsum) {  // uniform.  The two wave groups cover the same columns: they meet in LDS (behind the ring: the next
                     // tile may already be landing in it), 256 threads issue one atomic each
        float* cs_lds = reinterpret_cast<float*>(smem + P8_RING);
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
        if (tid < 256 && en0 + tid < g.N) atomicAdd(colsum + en0 + tid, cs_lds[tid] + cs_lds[256 + tid]);
        __syncthreads();  // (the next tile's epilogue writes cs_lds again)
      }
    }
    stamp(6);
    ++iter;
    if (!has_next) break;
    if (!early) {
      setup(tile_of(Ln), m0, n0);
      prologue();
    }
    bias_dma(n0);
    L = Ln;
    pending = false;
    // K step 0 has landed when all but elements 4-7 and the younger stores have
    if (early && full_tile()) {
      s_prev = NST;
      r_wait_vm<8 + NST>();
    } else {
      s_prev = 0;
      r_wait_vm<8>();
    }
    bar();
    if (grp == 1 && ABL != 3) bar();
  }
}

// workgroups of the persistent grid: one per CU of the current device (a multiple of 8, so that a workgroup's tiles stay on
// one XCD chunk); UWU_P8_GRID=n overrides (sweeps)
int p8_cus_impl() {
  static UwuEnv ge("UWU_P8_GRID");
  if (ge.get().set && ge.ival >= 8) return ge.ival & ~7;
  const int cus = uwu_dev_cus();
  return cus >= 8 ? cus & ~7 : 256;
}

template <typename TC, int EPI, bool TB, int ABL = 0>
int launch_p8(GemmArgs g, hipStream_t st) {
  auto kern = gemm_p8_kernel<TC, EPI, TB, ABL>;
  static unsigned char done[UWU_MAX_DEV];
  if (!uwu_func_lds(reinterpret_cast<const void*>(kern), P8_LDS, done)) {
    uwu_set_error("gemm_p8: the device cannot give a workgroup %d bytes of LDS", P8_LDS);
    return UWU_ELAUNCH;
  }
  g.tiles_m = (g.M + 255) / 256;
  g.tiles_n = (g.N + 255) / 256;
  const int nblk = g.tiles_m * g.tiles_n, ncu = p8_cus_impl();
  {
    static UwuEnv ce("UWU_P8_CONT");
    g.p8_cont = ce.get().is('0') ? 0 : 1;
  }
  UwuProfScope prof(st);
  hipLaunchKernelGGL(kern, dim3(nblk < ncu ? nblk : ncu), dim3(512), P8_LDS, st, g);
  prof.done(gemm_tag(g, TB, false), 0, 2.0 * g.M * g.N * g.K, gemm_bytes(g, 2, sizeof(TC)));
  UWU_LAUNCH_CHECK("gemm_p8");
  return UWU_OK;
}

}  // namespace

int uwu_p8_cus() { return p8_cus_impl(); }

// bf16 in / bf16 out, K a multiple of 64, 16-byte addressable operands.  UWU_GEMM_P8=0: off, =1: every shape it can run
// (tests, A/B comparisons); default: K >= 512 and at least one tile per CU.
bool uwu_gemm_p8_ok(const GemmArgs& g, bool tb) {
  static UwuEnv on("UWU_GEMM_P8"), kmin_e("UWU_P8_KMIN"), tmin_e("UWU_P8_MINTILES");
  if (on.get().is('0') || !uwu_dev_lds_fits(P8_LDS)) return false;
  if (g.K % 64 || g.K < 128) return false;
  if ((((uintptr_t)g.A | (uintptr_t)g.B) & 15) || g.lda % 8 || g.ldb % 8) return false;
  if (tb && (g.N % 8 || g.N < 8)) return false;
  if (g.epi != UWU_EPI_NONE && g.epi != UWU_EPI_BIAS && g.epi != UWU_EPI_BIAS_GELU && g.epi != UWU_EPI_DGELU) return false;
  if (tb ? (g.epi != UWU_EPI_NONE && g.epi != UWU_EPI_DGELU) : g.epi == UWU_EPI_DGELU) return false;
  if (on.is('1')) return true;
  // dGELU: its epilogue loads keep a tile from streaming into the next (every tile drains): in the step the 256 x 256 two-stage
  // kernel is faster (DiT-B/2 fc2 input gradient: 697 vs 668 TFLOP/s)
  static UwuEnv dg("UWU_P8_DGELU");  // "1": try it anyway (A/B)
  if (g.epi == UWU_EPI_DGELU && !dg.get().is('1')) return false;
  const int kmin = kmin_e.get().set ? kmin_e.ival : 512;
  const int64_t tiles = (int64_t)((g.M + 255) / 256) * ((g.N + 255) / 256);
  // padded column tiles: at most 1/8 of the columns may be padding (N = 1152 -> 5 tiles of 256: 10 %)
  const int64_t npad = (int64_t)((g.N + 255) / 256) * 256;
  const int tmin = tmin_e.get().set ? tmin_e.ival : 160;  // (SDXL-shape UNet, 4x128x128 x 12: its 240-tile Linears on this kernel 32.3 -> 32.8 images/s)
  return g.K >= kmin && tiles >= tmin && (npad - g.N) * 8 <= npad;
}

int uwu_launch_gemm_p8(const GemmArgs& g, bool tb, hipStream_t st) {
  if (!tb) {
    if (g.epi == UWU_EPI_NONE) {
      static UwuEnv abl("UWU_P8_ABL");
      if (abl.get().set) {
        if (abl.ival == 1) return launch_p8<bf16_t, UWU_EPI_NONE, false, 1>(g, st);
        if (abl.ival == 2) return launch_p8<bf16_t, UWU_EPI_NONE, false, 2>(g, st);
        if (abl.ival == 3) return launch_p8<bf16_t, UWU_EPI_NONE, false, 3>(g, st);
        if (abl.ival == 4) return launch_p8<bf16_t, UWU_EPI_NONE, false, 4>(g, st);
        if (abl.ival == 5) return launch_p8<bf16_t, UWU_EPI_NONE, false, 5>(g, st);
        if (abl.ival == 6) return launch_p8<bf16_t, UWU_EPI_NONE, false, 6>(g, st);
        if (abl.ival == 9 && getenv("UWU_P8_STAMPS")) {
          GemmArgs g9 = g;
          g9.q8 = (void*)strtoull(getenv("UWU_P8_STAMPS"), nullptr, 0);
          return launch_p8<bf16_t, UWU_EPI_NONE, false, 9>(g9, st);
        }
      }
      return launch_p8<bf16_t, UWU_EPI_NONE, false>(g, st);
    }
    if (g.epi == UWU_EPI_BIAS) return launch_p8<bf16_t, UWU_EPI_BIAS, false>(g, st);
    if (g.epi == UWU_EPI_BIAS_GELU) return launch_p8<bf16_t, UWU_EPI_BIAS_GELU, false>(g, st);
  } else {
    if (g.epi == UWU_EPI_NONE) return launch_p8<bf16_t, UWU_EPI_NONE, true>(g, st);
    if (g.epi == UWU_EPI_DGELU) return launch_p8<bf16_t, UWU_EPI_DGELU, true>(g, st);
  }
  uwu_set_error("gemm_p8: epilogue %d not instantiated (tb=%d)", g.epi, (int)tb);
  return UWU_EINVAL;
}
