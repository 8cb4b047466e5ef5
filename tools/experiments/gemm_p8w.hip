// Weight gradients on the 8-phase schedule of gemm_p8.hip:  dW[M, N] (fp32) += A[K, M]^T . B[K, N],  A = dY, B = X, both
// K-major bf16 (reference: the weight gradient autograd forms for every nn.Linear of the blocks, src/duwu/modules/rope_unet.py:
// 122-166, 393-411).  gemm_trw_kernel (gemm.hip) runs 192 x 384 tiles with one barrier per 32-deep K step and all eight waves in
// step (0.8 - 1.0 PFLOP/s on the DiT-B/2 and DiT-XL/2 shapes); this kernel:
//
//   * 256 x 256 tiles, K step 64, the half-tile ring of gemm_p8.hip with BOTH operands as [32 k][128 x] sub-images filled untouched
//     by LDS-DMA and read by ds_read_b64_tr_b16 (the layout of gemm_p8's K-major weight operand), the two wave groups one
//     barrier apart.
//   * STREAM-K inside an XCD.  The token range is cut into 8 parts, one per XCD (all tiles of a part read the same operand rows:
//     they share an L2); inside an XCD the (tile, K step pair) space -- tile-major -- is cut into equal runs, one per workgroup,
//     whatever the tile count: no partially filled round of workgroups (36 tiles of DiT-B/2's fc1 on 32 CUs would be 1.125
//     rounds).  A run crosses tile boundaries: the element stream simply goes on (continuous mode of gemm_p8.hip: the sources are
//     switched two K steps ahead, the finished accumulators leave quadrant by quadrant in the next K step's load intervals).
//   * a tile's sum over an XCD's token part arrives in up to `planes` pieces (the runs that touch it); piece number o of XCD x
//     goes to scratch plane x * planes + o, planes a tile has no piece for are zero-filled by the run that ends the tile, and
//     splitk_reduce adds the 8 * planes planes to dW.
//   * fused bias gradient (column sums of dY) as in gemm_trw_kernel: one more MFMA against an all-ones operand for the A
//     fragment i == wave column, on the tiles of the first column of tiles.
#include "gemm_shared.h"

namespace {

constexpr int W8_HT = 128 * ROW_BYTES;  // half-tile: 128 columns x 64 k = two [32 k][128 x] sub-images of 8 KB
constexpr int W8_LDS = 8 * W8_HT;

template <int H>
using IC = std::integral_constant<int, H>;

// LDS-DMA with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset: one address register per source instead of
// a pair (this kernel lives at the 256-register limit).  Inline asm: m0 is saved / set / restored inside the statement; hipcc does
// not count the operation (the kernel counts vmcnt by hand anyway).
// (the LDS destination = one wave-uniform base register + a compile-time offset: sixteen hoisted slot addresses do not fit the SGPRs)
template <unsigned OFF>
__device__ __forceinline__ void glds16_saddr(unsigned voff, const void* sbase /* wave-uniform */, unsigned lds_base /* wave-uniform */) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %3, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_base), "i"(OFF)
               : "memory", "scc");
}
// a ^ C as an instruction the optimiser does not hoist out of the K loop (24 hoisted fragment addresses do not fit the registers)
template <unsigned C>
__device__ __forceinline__ unsigned xor_keep(unsigned a) {
  if constexpr (C == 0) return a;
  unsigned r;
  asm volatile("v_xor_b32 %0, %2, %1" : "=v"(r) : "v"(a), "i"(C));
  return r;
}

// g.A = dY [K][M] (lda), g.B = X [K][N] (ldb), g.C2 = scratch planes, g.bias = bias gradient (or NULL)
// g.k_tiles_per_split = K-step PAIRS per XCD part (kp), g.wide = pairs per run (U), g.part_m = planes per XCD
template <int ABL = 0>
__global__ void __launch_bounds__(512, 2) gemm_p8w_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(128))) char smem[];  // (fragment addresses are formed by XOR on address bits 5-6)
  typedef bf16_t T;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int xcd = blockIdx.x & 7, run = blockIdx.x >> 3;
  const int kp = g.k_tiles_per_split, U = g.wide, planes = g.part_m;
  const int total = g.tiles_m * g.tiles_n * kp;
  int pos = run * U;
  const int pos_end = pos + U < total ? pos + U : total;
  if (pos >= pos_end) return;  // uniform

  const unsigned smem_base = (unsigned)(size_t)((const __attribute__((address_space(3))) char*)smem);
  // ---- LDS-DMA sources: piece `wave` = k-rows 4 wave .. + 3 (256 B each) of sub-image q (k half) of a half-tile ---------------
  unsigned oa[2], ob[2];  // [half]; the k half q of a K step adds 32 rows to the wave-uniform base
  const char* abase;
  const char* bbase;
  // piece of the (tile, pair) space that starts at `p`: tile, first / one-past-last pair inside the XCD's part
  int m0, n0, nk;  // the piece being LOADED: tile origin, K steps
  int l_tile;
  auto setup = [&](int p) __attribute__((always_inline)) {
    const int tile = p / kp, k0 = p - tile * kp;
    int k1 = k0 + (pos_end - p);
    if (k1 > kp) k1 = kp;
    nk = 2 * (k1 - k0);
    l_tile = tile;
    const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
    m0 = tm * 256;
    n0 = tn * 256;
    const int64_t krow = ((int64_t)xcd * kp + k0) * 128;  // first token row of the piece
    abase = reinterpret_cast<const char*>(static_cast<const T*>(g.A) + krow * g.lda + m0);
    bbase = reinterpret_cast<const char*>(static_cast<const T*>(g.B) + krow * g.ldb + n0);
    const int drow = lane >> 4;
    const int dchunk = (lane & 15) ^ (((drow & 3) << 2) | (wave & 3));
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      int xa = 128 * half + 8 * dchunk, xb = xa;
      if (xa > g.M - m0 - 8) xa = g.M - m0 - 8;  // (clamped columns: their products are never stored)
      if (xb > g.N - n0 - 8) xb = g.N - n0 - 8;
      oa[half] = (unsigned)((4 * wave + drow) * g.lda + xa) * 2u;
      ob[half] = (unsigned)((4 * wave + drow) * g.ldb + xb) * 2u;
    }
  };
  const int64_t astep = (int64_t)128 * g.lda, bstep = (int64_t)128 * g.ldb;  // bytes per K step
  const unsigned dma_base = smem_base + wave * 1024;
  // element h of K step t into slot (par, h); h: 0 = B0, 1 = A0, 2 = B1, 3 = A1
  auto issue = [&](auto hc, auto pc, int t) __attribute__((always_inline)) {
    constexpr int h = decltype(hc)::value, par = decltype(pc)::value, half = h >> 1;
    constexpr unsigned slot = (par * 4 + h) * W8_HT;
    const char* base = (h & 1) ? abase + t * astep : bbase + t * bstep;
    glds16_saddr<slot>((h & 1) ? oa[half] : ob[half], base, dma_base);
    glds16_saddr<slot + R_BSUB>((h & 1) ? oa[half] : ob[half], base + ((h & 1) ? astep : bstep) / 2, dma_base);
  };

  // ---- fragment reads: fragment f of an operand = columns 16 f .. of its half-tile (A: 4 grp + i, B: 2 wc + j) ----------------
  // [transposed read t]: addresses of this wave's fragment 0 in slot 0; fragment i: ^ (i << 5); slots 4-7: ^ 0x10000 (the kernel
  // has no static LDS: the dynamic array starts at address 0, so XOR of address bits 5, 6 and 16 is addition)
  unsigned a_t[2], b_t[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    a_t[t] = smem_base + tr_lane_base(lane, t, 8 * grp);
    b_t[t] = smem_base + tr_lane_base(lane, t, 4 * wc);
  }
  f32x4 acc[2][2][4][2];
  f32x4 sacc[2];  // bias gradient: rows of A half x, fragment i == wc
  uint4 af[4][2], bf0[2][2], bf1[2][2];
  const uint4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};

  auto read_a = [&](auto slotc) __attribute__((always_inline)) {  // 16 transposing reads
    constexpr int up = decltype(slotc)::value >= 4 ? 1 : 0;
    constexpr unsigned off = decltype(slotc)::value * W8_HT - up * 65536u;
    auto frag = [&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr unsigned x = (unsigned)(i << 5) | ((unsigned)up << 16);
      const unsigned a0 = xor_keep<x>(a_t[0]), a1 = xor_keep<x>(a_t[1]);
      const uint2 l0 = t_read_tr<off>(a0), h0 = t_read_tr<off>(a1);
      const uint2 l1 = t_read_tr<off + R_BSUB>(a0), h1 = t_read_tr<off + R_BSUB>(a1);
      af[i][0] = uint4{l0.x, l0.y, h0.x, h0.y};
      af[i][1] = uint4{l1.x, l1.y, h1.x, h1.y};
    };
    frag(IC<0>{});
    frag(IC<1>{});
    frag(IC<2>{});
    frag(IC<3>{});
  };
  auto read_b = [&](auto slotc, uint4 (&bf)[2][2]) __attribute__((always_inline)) {  // 8 transposing reads
    constexpr int up = decltype(slotc)::value >= 4 ? 1 : 0;
    constexpr unsigned off = decltype(slotc)::value * W8_HT - up * 65536u;
    auto frag = [&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      constexpr unsigned x = (unsigned)(j << 5) | ((unsigned)up << 16);
      const unsigned a0 = xor_keep<x>(b_t[0]), a1 = xor_keep<x>(b_t[1]);
      const uint2 l0 = t_read_tr<off>(a0), h0 = t_read_tr<off>(a1);
      const uint2 l1 = t_read_tr<off + R_BSUB>(a0), h1 = t_read_tr<off + R_BSUB>(a1);
      bf[j][0] = uint4{l0.x, l0.y, h0.x, h0.y};
      bf[j][1] = uint4{l1.x, l1.y, h1.x, h1.y};
    };
    frag(IC<0>{});
    frag(IC<1>{});
  };
  bool do_sum = false;  // the piece being multiplied belongs to the first column of tiles and a bias gradient is wanted
  auto mma_quadrant = [&](f32x4 (&c)[4][2], const uint4 (&bf)[2][2], auto sumc) __attribute__((always_inline)) {
    constexpr int sx = decltype(sumc)::value;  // >= 0: also the bias-gradient MFMAs of A half sx
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma_frag<T>(bf[j][kk], af[i][kk], c[i][j]);
    if constexpr (sx >= 0) {
      if (do_sum) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (wc == i) mma_frag<T>(ones, af[i][kk], sacc[sx]);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- the piece whose sums are in the accumulators (e*) and the finished piece that is leaving them (p*) -----------------------
  int e_m0 = 0, e_n0 = 0, e_tile = 0, e_first = 0;  // e_first: run index of the first run that touches e_tile
  bool e_last = false;                               // the piece ends its tile inside this XCD's part
  float* p_out = nullptr;                            // plane of the leaving piece
  int p_m0 = 0, p_n0 = 0;
  bool pending = false, stream = false;
  int s_prev = 0;
  auto plane_ptr = [&](int ord) __attribute__((always_inline)) {
    return static_cast<float*>(g.C2) + (int64_t)(xcd * planes + ord) * g.M * g.N;
  };
  // quadrant (x, y) of the leaving piece: 8 float4 stores per lane; the accumulators are zero afterwards
  bool p_full = false;  // the leaving piece's tile lies inside M x N: no bounds checks, all 32 stores exist
  auto epi_quadrant = [&](auto xc, auto yc) __attribute__((always_inline)) {
    constexpr int x = decltype(xc)::value, y = decltype(yc)::value;
    if constexpr (ABL == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[x][y][i][j]));
    } else {
      const int mq = p_m0 + 128 * x + 64 * grp + fr, nq = p_n0 + 128 * y + 32 * wc + 4 * fq;
      float* base = p_out + (int64_t)mq * g.N + nq;
      const int64_t rstep = (int64_t)16 * g.N;
      if (p_full) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          store4(base, acc[x][y][i][0]);
          store4(base + 16, acc[x][y][i][1]);
          base += rstep;
          asm volatile("" : "+v"(base));  // (one row pointer at a time: four precomputed ones spilled, and a reload waits vmcnt(0))
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (mq + 16 * i < g.M && nq + 16 * j < g.N) store4(base + 16 * j, acc[x][y][i][j]);
          base += rstep;
          asm volatile("" : "+v"(base));
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[x][y][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // what a finished piece does at once: its bias-gradient sums (atomics) and the zero planes of a tile it ends early
  auto finish_piece = [&]() __attribute__((always_inline)) {
    const int ord = run - e_first;
    p_out = plane_ptr(ord);
    p_m0 = e_m0;
    p_n0 = e_n0;
    p_full = e_m0 + 256 <= g.M && e_n0 + 256 <= g.N;
    if (do_sum) {  // D[n][m]: column m = fr on the lane, every row equal
      float* bg = const_cast<float*>(g.bias);
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int m = e_m0 + 128 * x + 64 * grp + 16 * wc + fr;
        if (fq == 0 && m < g.M) atomicAdd(bg + m, sacc[x][0]);
        sacc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (e_last) {
      for (int o = ord + 1; o < planes; ++o) {  // (rare: a tile that fewer runs touch than the scratch has planes)
        float* z = plane_ptr(o);
#pragma unroll 1
        for (int r = 0; r < 32; ++r) {  // this wave's 32 float4 slots: quadrant r >> 3, fragment row (r >> 1) & 3, column r & 1
          const int m = e_m0 + 128 * (r >> 4) + 64 * grp + 16 * ((r >> 1) & 3) + fr;
          const int n = e_n0 + 128 * ((r >> 3) & 1) + 32 * wc + 16 * (r & 1) + 4 * fq;
          if (m < g.M && n < g.N) store4(z + (int64_t)m * g.N + n, f32x4{0.f, 0.f, 0.f, 0.f});
        }
      }
    }
  };
  // the piece at `p` starts to be multiplied: remember what its epilogue needs (called with the sources already its own)
  auto begin_piece = [&](int p) __attribute__((always_inline)) {
    e_m0 = m0;
    e_n0 = n0;
    e_tile = l_tile;
    e_first = (e_tile * kp) / U;
    e_last = (p - e_tile * kp) + nk / 2 == kp;
    do_sum = g.bias != nullptr && n0 == 0;
  };

  // One K step of the piece being multiplied (cnk K steps; parity PAR static), see gemm_p8.hip.  `strm`: the stream goes on into
  // the next piece of this run, whose sources replace this piece's right after its last element has been requested.
  int cnk = 0, next_pos = 0;
  auto kstep = [&](auto pc, int t) __attribute__((always_inline)) {
    constexpr int par = decltype(pc)::value;
    const int t1 = t + 1, t2 = t + 2;
    const bool iss1 = (par == 1 || t != 0) && (t1 < cnk || stream), iss2 = t2 < cnk || stream;
    const int k1 = t1 < cnk ? t1 : t1 - cnk, k2 = t2 < cnk ? t2 : t2 - cnk;
    const bool ep = par == 0 && t == 0 && pending;
    // phase 0: (a0, b0)
    if (ep) {  // (all fragment registers are free here; ahead of phase 1 only 16 are)
      epi_quadrant(IC<0>{}, IC<0>{});
      epi_quadrant(IC<0>{}, IC<1>{});
    }
    read_b(IC<par * 4 + 0>{}, bf0);
    read_a(IC<par * 4 + 1>{});
    if (iss1) issue(IC<3>{}, IC<par ^ 1>{}, k1);
    asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");  // 24 reads issued: the 8 of B0 (first) are done, its slot is refilled next
    bar();
    mma_quadrant(acc[0][0], bf0, IC<0>{});
    bar();
    if (par == 0 && stream && t == cnk - 2) setup(next_pos);  // this piece's last element is on its way
    // phase 1: (a0, b1)
    read_b(IC<par * 4 + 2>{}, bf1);
    if (iss2) issue(IC<0>{}, IC<par>{}, k2);
    bar();
    mma_quadrant(acc[0][1], bf1, IC<-1>{});
    bar();
    // phase 2: (a1, b1)
    if (ep) {  // (the A fragment registers are free)
      epi_quadrant(IC<1>{}, IC<1>{});
      epi_quadrant(IC<1>{}, IC<0>{});
    }
    read_a(IC<par * 4 + 3>{});
    if (iss2) issue(IC<1>{}, IC<par>{}, k2);
    bar();
    mma_quadrant(acc[1][1], bf1, IC<1>{});
    bar();
    // phase 3: (a1, b0) -- no LDS reads; the wait for K step t + 1
    if (iss2) {
      issue(IC<2>{}, IC<par>{}, k2);
      if (par == 0 && t == 0 && s_prev == 32) r_wait_vm<6 + 32>();  // (gemm_p8.hip: the leaving piece's stores may stay in flight)
      else r_wait_vm<6>();
    } else {
      r_wait_vm<0>();
    }
    bar();
    mma_quadrant(acc[1][0], bf0, IC<-1>{});
    bar();
  };

#pragma unroll
  for (int x = 0; x < 2; ++x) {
    sacc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[x][y][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  setup(pos);
  issue(IC<0>{}, IC<0>{}, 0);
  issue(IC<1>{}, IC<0>{}, 0);
  issue(IC<2>{}, IC<0>{}, 0);
  issue(IC<3>{}, IC<0>{}, 0);
  issue(IC<0>{}, IC<1>{}, 1);
  issue(IC<1>{}, IC<1>{}, 1);
  issue(IC<2>{}, IC<1>{}, 1);
  issue(IC<3>{}, IC<1>{}, 1);
  r_wait_vm<8>();  // K step 0 has landed when all but elements 4-7 have
  bar();
  if (grp == 1) bar();  // waves 4-7 run one barrier behind their SIMD partners
  for (;;) {
    begin_piece(pos);
    cnk = nk;
    next_pos = pos + cnk / 2;
    stream = next_pos < pos_end;
    for (int t = 0; t < cnk; t += 2) {
      kstep(IC<0>{}, t);
      kstep(IC<1>{}, t + 1);
    }
    if (!stream) break;
    // (the sources were switched in K step cnk - 2: m0 / n0 / nk describe the next piece)
    issue(IC<3>{}, IC<1>{}, 1);  // next piece's (1, A1): its slot was last read in phase 2 of the last K step
    finish_piece();
    pending = true;
    s_prev = (p_full && ABL != 1) ? 32 : 0;
    pos = next_pos;
  }
  if (grp == 0) bar();  // every wave has passed the same number of barriers
  finish_piece();
  epi_quadrant(IC<0>{}, IC<0>{});
  epi_quadrant(IC<0>{}, IC<1>{});
  epi_quadrant(IC<1>{}, IC<1>{});
  epi_quadrant(IC<1>{}, IC<0>{});
}

int p8w_cus() {
  static int cus[16] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16) dev = 0;
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8 ? prop.multiProcessorCount & ~7 : 256;
  }
  static UwuEnv ge("UWU_P8_GRID");
  return ge.get().set && ge.ival >= 8 ? ge.ival & ~7 : cus[dev];
}

struct P8wPlan {
  int kp, U, planes, runs;  // K-step pairs per XCD part, pairs per run, planes per XCD, runs per XCD
};
P8wPlan p8w_plan(int M, int N, int K) {
  P8wPlan p;
  const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
  p.kp = K / 1024;
  const int per_xcd = p8w_cus() / 8;
  const int64_t total = (int64_t)tiles * p.kp;
  p.U = (int)((total + per_xcd - 1) / per_xcd);
  if (p.U < 1) p.U = 1;
  p.runs = (int)((total + p.U - 1) / p.U);
  // a tile of kp pairs is touched by at most ceil((kp - 1) / U) + 1 runs
  p.planes = (p.kp - 1 + p.U - 1) / p.U + 1;
  if (p.planes > p.kp) p.planes = p.kp;
  return p;
}

}  // namespace

// bf16, K a multiple of 1024 tokens (8 XCD parts of whole K-step pairs), 16-byte addressable operands.
// UWU_GEMM_P8W=0: off, =1: every shape it can run (tests); default: long reductions over at least four tiles.
bool uwu_gemm_p8w_ok(const GemmArgs& g) {
  static UwuEnv on("UWU_GEMM_P8W");
  if (on.get().is('0')) return false;
  if (g.K % 1024 || g.K < 2048 || g.M % 8 || g.N % 8 || g.M < 8 || g.N < 8) return false;
  if ((((uintptr_t)g.A | (uintptr_t)g.B) & 15) || g.lda % 8 || g.ldb % 8) return false;
  if (on.is('1')) return true;
  const int tiles = ((g.M + 255) / 256) * ((g.N + 255) / 256);
  const int64_t mpad = (int64_t)((g.M + 255) / 256) * 256, npad = (int64_t)((g.N + 255) / 256) * 256;
  return g.K >= 16384 && tiles >= 4 && (mpad * npad - (int64_t)g.M * g.N) * 8 <= mpad * npad;
}
size_t uwu_gemm_p8w_scratch_bytes(int M, int N, int K) {
  if (K % 1024 || K < 2048) return 0;
  const P8wPlan p = p8w_plan(M, N, K);
  return (size_t)8 * p.planes * M * N * sizeof(float);
}
// planes of M x N floats the launch wrote into `scratch` (to be summed into C by the caller), or < 0 on error
int uwu_launch_gemm_p8w(GemmArgs g, void* scratch, hipStream_t st) {
  auto kern = gemm_p8w_kernel<0>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
    attr_done = true;
  }
  const P8wPlan p = p8w_plan(g.M, g.N, g.K);
  g.tiles_m = (g.M + 255) / 256;
  g.tiles_n = (g.N + 255) / 256;
  g.k_tiles_per_split = p.kp;
  g.wide = p.U;
  g.part_m = p.planes;
  g.C2 = scratch;
  hipLaunchKernelGGL(kern, dim3(8 * p.runs), dim3(512), W8_LDS, st, g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    uwu_set_error("gemm_p8w: launch failed: %s", hipGetErrorString(e));
    return UWU_ELAUNCH;
  }
  return 8 * p.planes;
}
