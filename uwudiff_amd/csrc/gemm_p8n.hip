// 128 x 384 sibling of gemm_p8.hip for the Linears whose N is a multiple of 384 and not of 256 -- every forward / input-gradient
// Linear of DiT-S/2 but its two GELU ones (N = 384 / 1152: the headline config, BASELINE configs[1]) and the N = 1152 / 3456 ones of
// DiT-XL/2 (which the 256 x 256 kernel pads by 10 % / 4 %).  Reference op: nn.Linear fwd / bwd inside the transformer blocks,
// src/duwu/modules/rope_unet.py:122-166, 393-411.  gemm_wide_kernel (gemm.hip) covers these shapes with 192 x 384 tiles in a
// two-stage loop, eight waves in step; here:
//
//   * the SAME half-tile ring as gemm_p8.hip (eight slots of 128 rows x 128 B, filled by LDS-DMA, the two wave groups one barrier
//     apart), but a K step is the elements B0 A B1 B2: one 128-row A half-tile and the three 128-column thirds of W.  With N = 384
//     a tile spans all of N, so an A row crosses L2 -> LDS once (the property gemm_wide_kernel was built for).
//   * a wave owns 64 rows (grp) x 32 columns of EVERY third (wc): phase y of a K step multiplies its A rows (read once, in phase 0,
//     and kept) with third y: 16 MFMAs, three phases per K step.  Every slot is read in exactly one phase (B0 + A: 0, B1: 1, B2: 2).
//   * requests: phase 0 -> (t+1, B2), phase 1 -> (t+2, B0), phase 2 -> (t+2, A), (t+2, B1).  Each of them lands in a slot whose
//     reads were one phase earlier and retired in front of that phase's first barrier (lgkmcnt(8) behind the B0 reads of phase
//     0, lgkmcnt(0) in phases 1 and 2), or two phases earlier (A).  Every phase waits vmcnt(10) in front of its first barrier: the
//     element the NEXT phase reads is always the sixth-youngest request (B0, A | B1 | B2 of the coming phases were requested five
//     elements back), so five half-tiles stay in flight across the barriers -- a request has four phases to land.
//   * persistent and CONTINUOUS like gemm_p8.hip: the stream runs on into the workgroup's next tile (sources switched right after
//     the tile's last request), the finished thirds leave in the load intervals of the next tile's K step 0 (third y is final after
//     phase y of the last K step and needed again in phase y of the next K step 0); the waits of the next tile's first two K steps
//     let those stores stay in flight (vmcnt is in issue order).  Needs an even number of K steps >= 4 and an epilogue without
//     loads (plain / bias / bias + GELU; K-major W: plain) -- other cases stay on the older kernels.
#include "gemm_shared.h"

namespace {

constexpr int N8_HT = 128 * ROW_BYTES;  // a third of W: 128 rows x 128 B = 16 KB
// FI = A fragments per wave: 4 -> 128-row tiles, 6 -> 192-row tiles (131 flop per L2 -> LDS byte instead of 96: the tile of
// gemm_wide_kernel).  LDS per K-step parity: B0 | B1 | B2 | A (32 FI rows)
template <int FI> struct N8 {
  static constexpr int AROWS = 32 * FI, AB = AROWS * ROW_BYTES, PAR = 3 * N8_HT + AB, RING = 2 * PAR;
  static constexpr int BIAS = RING, LDS = BIAS + 8 * 512;  // + 8 waves x 128 bias floats (96 used)
  static constexpr int APIECES = AROWS / 64;               // LDS-DMA instructions per thread for an A element (2 or 3)
  static constexpr int VM = 2 * 4 + APIECES;               // DMA instructions of the five youngest requests (always one A among them)
};

template <int H>
using IC = std::integral_constant<int, H>;

__device__ __forceinline__ int n8_swz(int row, int chunk) { return row * ROW_BYTES + (((chunk ^ (row >> 1)) & 7) << 4); }  // = p8_swz

template <int N>
__device__ __forceinline__ void n8_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ABL (UWU_P8_ABL, plain forward): 1 = no output stores (timing only)
template <typename TC, int EPI, bool TB, int FI, int ABL = 0>
__global__ void __launch_bounds__(512, 2) gemm_p8n_kernel(const GemmArgs g) {
  typedef N8<FI> G;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int nblk = g.tiles_m * g.tiles_n;
  const int nk = g.K >> 6;  // even, >= 4 (checked on the host)
  auto tile_of = [&](int L) __attribute__((always_inline)) {  // XCD-aware tile order as in gemm_kernel
    const int xcd = L & 7, loc = L >> 3;
    const int q = nblk >> 3, rm = nblk & 7;
    return (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + loc;
  };

  // ---- LDS-DMA sources: a half-tile = 16 pieces of 8 rows x 128 B (TB: 2 x 8 pieces of 4 k-rows x 256 B); this thread moves
  // pieces wave and wave + 8.  32-bit byte offsets from wave-uniform bases.
  // (N is a multiple of 384: no W row is ever clamped, so thirds / pieces are wave-uniform offsets on ONE lane offset; the 192-row form
  // takes only M % 192 == 0 -- no clamped A row either -- and keeps one lane offset for A as well: it lives at the register limit)
  constexpr int NOA = FI == 6 ? 1 : G::APIECES;
  unsigned oa[NOA], ob;
  const char* abase;
  const char* bbase;
  auto setup = [&](int tile, int& m0, int& n0) __attribute__((always_inline)) {
    const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
    m0 = tm * G::AROWS;
    n0 = tn * 384;
    abase = reinterpret_cast<const char*>(static_cast<const T*>(g.A) + (int64_t)m0 * g.lda);
    bbase = reinterpret_cast<const char*>(static_cast<const T*>(g.B) + (TB ? (int64_t)n0 : (int64_t)n0 * g.ldb));
    // (the lane's row / chunk are recomputed from the lane id at every call: kept across the K loop they were spilled, and a reload
    // of a spilled register waits vmcnt(0) -- the whole ring -- at every tile)
    int lane_ = lane;
    asm volatile("" : "+v"(lane_));
#pragma unroll
    for (int q = 0; q < NOA; ++q) {
      const int row = 8 * (wave + 8 * q) + (lane_ >> 3);
      const int c = ((lane_ & 7) ^ (row >> 1)) & 7;  // logical chunk that must land at position lane & 7 (n8_swz)
      int ga = row;
      if (ga >= g.M - m0) ga = g.M - m0 - 1;  // (clamped rows / columns: their products are never stored)
      oa[q] = (unsigned)(ga * g.lda + 8 * c) * 2u;
    }
    if constexpr (!TB) {  // piece wave of third 0 (rows 8 wave + lane / 8; piece wave + 8: 64 rows on -- same chunk swizzle)
      const int row = 8 * wave + (lane >> 3);
      ob = (unsigned)(row * g.ldb + 8 * (((lane & 7) ^ (row >> 1)) & 7)) * 2u;
    } else {  // sub-image 0 (k half 0) of third 0: piece wave = k-rows 4 wave .. + 3, 256 B each
      const int drow = lane >> 4;
      const int dchunk = (lane & 15) ^ (((drow & 3) << 2) | (wave & 3));
      ob = (unsigned)((4 * wave + drow) * g.ldb + 8 * dchunk) * 2u;
    }
  };
  // wave-uniform byte offset of third y, piece / k half q of a W element
  auto boff = [&](int y, int q) __attribute__((always_inline)) {
    return TB ? (int64_t)64 * q * g.ldb + 256 * y : ((int64_t)128 * y + 64 * q) * g.ldb * 2;
  };
  const int64_t bstep = TB ? (int64_t)128 * g.ldb : 128;  // bytes per K step
  // element e of K step kt into slot (par, e); e: 0 = B0, 1 = A, 2 = B1, 3 = B2
  auto issue = [&](auto ec, auto pc, int kt) __attribute__((always_inline)) {
    constexpr int e = decltype(ec)::value, par = decltype(pc)::value;
    constexpr int y = e == 0 ? 0 : e - 1;  // third of a B element
    char* slot = smem + par * G::PAR + (e == 1 ? 3 : y) * N8_HT;
    if constexpr (e == 1) {
#pragma unroll
      for (int q = 0; q < G::APIECES; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(abase + (int64_t)kt * 128 +
                                                                                      (NOA == 1 ? (int64_t)128 * q * g.lda + oa[0] : (int64_t)oa[q % NOA])),
                                         (__attribute__((address_space(3))) void*)(slot + (wave + 8 * q) * 1024), 16, 0, 0);
    } else {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        char* dst = TB ? slot + q * R_BSUB + wave * 1024 : slot + (wave + 8 * q) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bbase + kt * bstep + boff(y, q) + ob),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
  };

  // ---- fragment read addresses of parity 0 (parity 1: + G::PAR, beyond the 16-bit offset field: added to the base) -------------
  const unsigned smem_base = (unsigned)(size_t)((const __attribute__((address_space(3))) char*)smem);
  unsigned a_ad[2], b_ad[2];  // [kk]: fragment 0 (A rows 16 FI grp + fr of the A element, B rows 32 wc + fr of third 0); fragment i: + 2048 i
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    a_ad[kk] = smem_base + 3 * N8_HT + (unsigned)n8_swz(16 * FI * grp + fr, 4 * kk + fq);
    b_ad[kk] = smem_base + (unsigned)n8_swz(32 * wc + fr, 4 * kk + fq);
  }
  unsigned bt_ad[2][2];  // TB: [fragment j][transposed read t] inside a [32 k][128 n] sub-image
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int t = 0; t < 2; ++t) bt_ad[j][t] = smem_base + tr_lane_base(lane, t, 4 * wc + 2 * j);

  f32x4 acc[3][FI][2];
  uint4 af[FI][2], bf[2][2];

  auto read_a = [&](auto pc) __attribute__((always_inline)) {  // 2 FI reads
    constexpr unsigned hi = decltype(pc)::value * G::PAR;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < FI; ++i) {
        if (i == 0) af[i][kk] = r_read128<0>(a_ad[kk] + hi);
        if (i == 1) af[i][kk] = r_read128<2048>(a_ad[kk] + hi);
        if (i == 2) af[i][kk] = r_read128<4096>(a_ad[kk] + hi);
        if (i == 3) af[i][kk] = r_read128<6144>(a_ad[kk] + hi);
        if (i == 4) af[i][kk] = r_read128<8192>(a_ad[kk] + hi);
        if (i == 5) af[i][kk] = r_read128<10240>(a_ad[kk] + hi);
      }
  };
  auto read_b = [&](auto pc, auto yc) __attribute__((always_inline)) {  // third y: 4 reads (TB: 8 transposing reads)
    constexpr unsigned up = decltype(pc)::value * G::PAR, off = decltype(yc)::value * N8_HT;
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
  auto mma_third = [&](f32x4 (&c)[FI][2]) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma_frag<T>(bf[j][kk], af[i][kk], c[i][j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  constexpr bool has_bias = EPI == UWU_EPI_BIAS || EPI == UWU_EPI_BIAS_GELU || EPI == UWU_EPI_BIAS_SILU;
  constexpr int SPT = (EPI == UWU_EPI_BIAS_GELU || EPI == UWU_EPI_BIAS_SILU) ? 2 * FI : FI;  // stores per third and wave (full tile)
  bool pending = false, pfull = false;  // the accumulators still hold the PREVIOUS tile; all of its stores are known to exist
  int m0, n0, em0 = 0, en0 = 0;         // tile being loaded / tile whose results are in the accumulators
  int L = blockIdx.x, Ln = 0;
  bool stream = false;
  const unsigned bias_ad = smem_base + G::BIAS + wave * 512 + 16 * fq;
  // bias of a tile: this wave's 3 x 32 columns by LDS-DMA into its own 512 bytes (see gemm_p8.hip): thirds 0 / 1, then third 2
  auto bias_dma = [&](int n_tile) __attribute__((always_inline)) {
    if constexpr (has_bias) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int n = n_tile + 128 * (2 * h + (lane >> 5)) + 32 * wc + (lane & 31);
        if (n >= g.N) n = g.N - 1;  // (third "3" of the second request does not exist: clamped, never read)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.bias + n),
                                         (__attribute__((address_space(3))) void*)(smem + G::BIAS + wave * 512 + 256 * h), 4, 0, 0);
      }
    }
  };
  // epilogue of third y of tile (em0, en0); the accumulators are zero afterwards
  auto epi_third = [&](auto yc) __attribute__((always_inline)) {
    constexpr int y = decltype(yc)::value;
    if constexpr (ABL == 1) {
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[y][i][j]));
    } else {
      const int m_q = em0 + 16 * FI * grp, n_q = en0 + 128 * y + 32 * wc;
      EpiPre<T, FI, 2> pre;
      if constexpr (has_bias) {
        const uint4 b0 = r_read128<128 * y>(bias_ad), b1 = r_read128<128 * y + 64>(bias_ad);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        pre.bias[0] = *reinterpret_cast<const f32x4*>(&b0);
        pre.bias[1] = *reinterpret_cast<const f32x4*>(&b1);
      } else {
        pre.bias[0] = pre.bias[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      epilogue_tile<T, TC, FI, 2, EPI>(acc[y], pre, g, m_q, n_q, fr, fq, nullptr, 0, 0, -1, nullptr);
    }
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[y][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // the wait in front of a phase's first barrier: the element the NEXT phase reads is the sixth-youngest request, so ten DMA
  // instructions (+ `st` stores of the previous tile that were issued behind it) may stay in flight; `all` = every one of the five
  // younger requests exists (else: the tail of the workgroup's last tile -- drain)
  auto wait_next = [&](bool all, int st) __attribute__((always_inline)) {
    if (!all) n8_wait_vm<0>();
    else if (st == 0) n8_wait_vm<G::VM>();
    else if (st == 1) n8_wait_vm<G::VM + SPT>();
    else if (st == 2) n8_wait_vm<G::VM + 2 * SPT>();
    else n8_wait_vm<G::VM + 3 * SPT>();
  };

  // One K step (parity PAR static); kt: K step inside the tile.
  auto kstep = [&](auto pc, int t) __attribute__((always_inline)) {
    constexpr int par = decltype(pc)::value;
    const int t1 = t + 1, t2 = t + 2;
    const bool ex1 = t1 < nk || stream, ex2 = t2 < nk || stream;  // the requests of this K step exist
    const int k1 = t1 < nk ? t1 : t1 - nk, k2 = t2 < nk ? t2 : t2 - nk;
    const bool ep = par == 0 && t == 0 && pending;
    // stores of the previous tile that may stay in flight at the waits of this K step: all of them leave ahead of phase 0 of K step 0,
    // i.e. behind the requests the three waits of K step 0 and the first wait of K step 1 are for ((1, B2) is the first younger one)
    const int sf = (pending && pfull && ABL != 1) ? 1 : 0;
    const int st0 = sf * (t <= 1 ? 3 : 0), st1 = sf * (t == 0 ? 3 : 0), st2 = st1;
    // phase 0: third 0
    if (ep) {  // every fragment register is free here (ahead of phases 1 and 2 the A fragments are live: the 192-row form spilled)
      epi_third(IC<0>{});
      epi_third(IC<1>{});
      epi_third(IC<2>{});
      bias_dma(n0);  // this tile's bias (m0 / n0 name the tile being multiplied until K step nk - 2), behind the last read of the previous one's
    }
    read_b(IC<par>{}, IC<0>{});
    read_a(IC<par>{});
    if (ex1) issue(IC<3>{}, IC<par ^ 1>{}, k1);          // (t+1, B2): over (t-1, B2), read and retired one phase ago
    if constexpr (FI == 4) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");  // the B0 reads (issued first) are done: its slot is
    else asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");                    // refilled in the next phase
    wait_next(ex1, st0);                                 // (t, B1) has landed
    bar();
    mma_third(acc[0]);
    bar();
    if (par == 0 && stream && t == nk - 2) {  // this tile's last request (nk-1, B2) is on its way: the sources become the next tile's
      em0 = m0;
      en0 = n0;
      setup(tile_of(Ln), m0, n0);
    }
    // phase 1: third 1
    read_b(IC<par>{}, IC<1>{});
    if (ex2) issue(IC<0>{}, IC<par>{}, k2);              // (t+2, B0): over (t, B0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the B1 reads are done: its slot is refilled in the next phase
    wait_next(ex2, st1);                                 // (t, B2) has landed
    bar();
    mma_third(acc[1]);
    bar();
    // phase 2: third 2
    read_b(IC<par>{}, IC<2>{});
    if (ex2) {
      issue(IC<1>{}, IC<par>{}, k2);                     // (t+2, A): over (t, A), read two phases ago
      issue(IC<2>{}, IC<par>{}, k2);                     // (t+2, B1): over (t, B1)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the B2 reads are done: its slot is refilled in the next phase
    wait_next(ex2, st2);                                 // (t+1, B0) and (t+1, A) have landed
    bar();
    mma_third(acc[2]);
    bar();
  };

#pragma unroll
  for (int y = 0; y < 3; ++y)
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[y][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  setup(tile_of(L), m0, n0);
  bias_dma(n0);
  // K step 0 and, but for its B2 (phase 0 of K step 0 requests it), K step 1
  issue(IC<0>{}, IC<0>{}, 0);
  issue(IC<1>{}, IC<0>{}, 0);
  issue(IC<2>{}, IC<0>{}, 0);
  issue(IC<3>{}, IC<0>{}, 0);
  issue(IC<0>{}, IC<1>{}, 1);
  issue(IC<1>{}, IC<1>{}, 1);
  issue(IC<2>{}, IC<1>{}, 1);
  n8_wait_vm<G::VM>();  // (0, B0) and (0, A) have landed: the five requests behind them may fly
  bar();
  if (grp == 1) bar();  // waves 4-7 run one barrier behind their SIMD partners
  for (;;) {
    Ln = L + gridDim.x;
    stream = Ln < nblk;
    for (int t = 0; t < nk; t += 2) {
      kstep(IC<0>{}, t);
      kstep(IC<1>{}, t + 1);
    }
    if (!stream) break;
    // (the sources, m0 / n0 and em0 / en0 were switched in K step nk - 2)
    pending = true;
    pfull = em0 + G::AROWS <= g.M && en0 + 384 <= g.N && g.wide && sizeof(TC) == 2;
    L = Ln;
  }
  if (grp == 0) bar();  // every wave has passed the same number of barriers
  em0 = m0;
  en0 = n0;
  epi_third(IC<0>{});
  epi_third(IC<1>{});
  epi_third(IC<2>{});
}

int p8n_cus() {
  static UwuEnv ge("UWU_P8_GRID");
  if (ge.get().set && ge.ival >= 8) return ge.ival & ~7;
  const int cus = uwu_dev_cus();
  return cus >= 8 ? cus & ~7 : 256;
}

template <typename TC, int EPI, bool TB, int FI, int ABL = 0>
int launch_p8n_fi(GemmArgs g, hipStream_t st) {
  auto kern = gemm_p8n_kernel<TC, EPI, TB, FI, ABL>;
  static unsigned char done[UWU_MAX_DEV];
  if (!uwu_func_lds(reinterpret_cast<const void*>(kern), N8<FI>::LDS, done)) {
    uwu_set_error("gemm_p8n: the device cannot give a workgroup %d bytes of LDS", N8<FI>::LDS);
    return UWU_ELAUNCH;
  }
  g.tiles_m = (g.M + 32 * FI - 1) / (32 * FI);
  g.tiles_n = g.N / 384;
  const int nblk = g.tiles_m * g.tiles_n, ncu = p8n_cus();
  UwuProfScope prof(st);
  hipLaunchKernelGGL(kern, dim3(nblk < ncu ? nblk : ncu), dim3(512), N8<FI>::LDS, st, g);
  prof.done(gemm_tag(g, TB, false), 0, 2.0 * g.M * g.N * g.K, gemm_bytes(g, 2, sizeof(TC)));
  UWU_LAUNCH_CHECK("gemm_p8n");
  return UWU_OK;
}

// 192-row tiles unless UWU_P8N_ROWS=128 (A/B comparisons) or the 128-row grid fills the chip and the 192-row one does not
template <typename TC, int EPI, bool TB, int ABL = 0>
int launch_p8n(const GemmArgs& g, hipStream_t st) {
  static UwuEnv rows("UWU_P8N_ROWS");
  const int64_t t192 = (int64_t)((g.M + 191) / 192) * (g.N / 384);
  // (the two-output GELU form of the 192-row kernel spills inside the K loop: 128 rows)
  const bool small = g.M % 192 != 0 || EPI == UWU_EPI_BIAS_GELU || (rows.get().set ? rows.ival == 128 : t192 < p8n_cus());
  return small ? launch_p8n_fi<TC, EPI, TB, 4, ABL>(g, st) : launch_p8n_fi<TC, EPI, TB, 6, ABL>(g, st);
}

}  // namespace

// bf16 in / bf16 out, N a multiple of 384, an even number >= 4 of 64-deep K steps, 16-byte addressable operands.
// UWU_GEMM_P8N=0: off, =1: every shape it can run (tests, A/B comparisons); default: at least one tile per CU.
bool uwu_gemm_p8n_ok(const GemmArgs& g, bool tb) {
  static UwuEnv on("UWU_GEMM_P8N");
  if (on.get().is('0') || !uwu_dev_lds_fits(N8<6>::LDS)) return false;
  if (g.N % 384 || g.K % 128 || g.K < 256) return false;
  if ((((uintptr_t)g.A | (uintptr_t)g.B) & 15) || g.lda % 8 || g.ldb % 8) return false;
  if (tb ? g.epi != UWU_EPI_NONE : (g.epi != UWU_EPI_NONE && g.epi != UWU_EPI_BIAS && g.epi != UWU_EPI_BIAS_GELU)) return false;
  if (on.is('1')) return true;
  // In the DiT-S/2 step (B = 768, same box, alternating): the K-major-weight form takes the input gradients from 154 to 148 us per
  // launch (976 -> 1016 TFLOP/s), the forward forms lose to gemm_as_kernel (qkv + bias) / gemm_wide_kernel (162 -> 171 us): only
  // input gradients by default.
  const int64_t tiles = (int64_t)((g.M + 127) / 128) * (g.N / 384);
  return tb && tiles >= 256;
}

int uwu_launch_gemm_p8n(const GemmArgs& g, bool tb, hipStream_t st) {
  if (!tb) {
    if (g.epi == UWU_EPI_NONE) {
      static UwuEnv abl("UWU_P8_ABL");
      if (abl.get().set && abl.ival == 1) return launch_p8n<bf16_t, UWU_EPI_NONE, false, 1>(g, st);
      return launch_p8n<bf16_t, UWU_EPI_NONE, false>(g, st);
    }
    if (g.epi == UWU_EPI_BIAS) return launch_p8n<bf16_t, UWU_EPI_BIAS, false>(g, st);
    if (g.epi == UWU_EPI_BIAS_GELU) return launch_p8n<bf16_t, UWU_EPI_BIAS_GELU, false>(g, st);
  } else if (g.epi == UWU_EPI_NONE) {
    return launch_p8n<bf16_t, UWU_EPI_NONE, true>(g, st);
  }
  uwu_set_error("gemm_p8n: epilogue %d not instantiated (tb=%d)", g.epi, (int)tb);
  return UWU_EINVAL;
}
