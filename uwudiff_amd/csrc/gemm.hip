// MFMA GEMM for gfx950: C[M,N] = opA(A) . opB(B) (+ epilogue), fp32 accumulate.
//
// One kernel template serves the three contractions of every Linear in the denoiser:
//   forward  Y  = X  . W^T          (transA=0, transB=0)   K-contiguous operands
//   dgrad    dX = dY . W            (transA=0, transB=1)   W is [red][out]   -> transposed staging of B
//   wgrad    dW = dY^T . X          (transA=1, transB=1)   both [red][out]   -> transposed staging of A and B
// and both operand types: bf16 (v_mfma_f32_16x16x32_bf16) and exact fp32 (v_mfma_f32_16x16x4_f32, the
// parity mode).  Reference op sequence replaced: nn.Linear fwd/bwd inside diffusers blocks
// (reference src/duwu/modules/rope_unet.py:122-166, 404).
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves, 2x2, 64x64 per wave = 4x4 MFMA 16x16 tiles),
// K step = 128 bytes per row (64 bf16 / 32 fp32), LDS double-buffered (2 x 32 KB -> 2 workgroups per CU).
// LDS image: [row][128 B], 16-B chunk c of row r stored at chunk (c ^ (r>>1) ^ (r>>4)) & 7:
//   * fragment reads (ds_read_b128, lane -> row, fixed chunk) are bank-conflict free,
//   * row-major staging writes (8 lanes x 16 B per row) are conflict free,
//   * transposed staging writes (ds_write_b64 bf16 / ds_write_b128 fp32) are <= 2-way.
// The fp32 path reuses the same image: lane group g=lane>>4 reads chunk 4*kk+g and feeds element e of it to
// the e-th 16x16x4 MFMA, i.e. a k-permutation applied identically to A and B.
// Global->LDS goes through registers (prefetch of tile t+1 issued before the MFMAs of tile t); K-strided
// operands are transposed in registers (4x8 bf16 / 4x4 fp32 blocks) so HBM reads stay 16 B/lane coalesced.
#include "gemm_shared.h"

namespace {


// ACC = atomic-accumulate epilogue (standard accumulator orientation: registers walk rows, lanes walk
// 16 consecutive columns -> 64-B atomic segments).  Otherwise the MFMA operands are swapped so that each lane
// owns 4 consecutive columns of one row and can apply the epilogue on / store 8-16 B vectors directly.
// GL = LDS-DMA staging (only with TA = TB = false and full K tiles).
template <typename T, typename TC, bool TA, bool TB, bool ACC, bool GL = false, int EPI = -1>
__global__ void __launch_bounds__(256, 2) gemm_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BK = GT<T>::BK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give every XCD a
  // contiguous run of tile ids (n fastest) -> the n-tiles of one A row-panel hit the same L2.
  const int nblk = g.tiles_m * g.tiles_n;
  int tile;
  {
    const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3;
    const int q = nblk >> 3, rm = nblk & 7;
    tile = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + loc;
  }
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int ktiles = (g.K + BK - 1) / BK;
  const int kt_begin = blockIdx.z * g.k_tiles_per_split;
  int kt_end = kt_begin + g.k_tiles_per_split;
  if (kt_end > ktiles) kt_end = ktiles;
  if (kt_begin >= kt_end) return;  // uniform per block

  const T* A = static_cast<const T*>(g.A);
  const T* B = static_cast<const T*>(g.B);

  Stager<T, TA> sa;
  Stager<T, TB> sb;
  EpiPre<T, 4, 4> pre;
  // (compile-time epilogues only: the run-time variants would hold bias AND aux registers through the K loop)
  if constexpr (!ACC && EPI >= 0) epi_prefetch<T, 4, 4, EPI>(pre, g, m0 + wm * 64, n0 + wn * 64, lane & 15, lane >> 4);
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if constexpr (GL) {
    glds_tile<T>(A, g.lda, m0, kt_begin * BK, g.M, smem, tid);
    glds_tile<T>(B, g.ldb, n0, kt_begin * BK, g.N, smem + TILE_BYTES, tid);
  } else {
    sa.load(A, g.lda, m0, kt_begin * BK, g.M, g.K, tid);
    sb.load(B, g.ldb, n0, kt_begin * BK, g.N, g.K, tid);
    sa.store(smem, tid);
    sb.store(smem + TILE_BYTES, tid);
  }
  __syncthreads();

  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int cur = (kt - kt_begin) & 1;
    const char* la = smem + cur * 2 * TILE_BYTES;
    const char* lb = la + TILE_BYTES;
    const bool more = (kt + 1 < kt_end);
    if (more) {
      if constexpr (GL) {  // DMA the next tile straight into the other stage (free since the last barrier)
        char* na = smem + (cur ^ 1) * 2 * TILE_BYTES;
        glds_tile<T>(A, g.lda, m0, (kt + 1) * BK, g.M, na, tid);
        glds_tile<T>(B, g.ldb, n0, (kt + 1) * BK, g.N, na + TILE_BYTES, tid);
      } else {  // prefetch next tile into registers; latency hides under the MFMAs below
        sa.load(A, g.lda, m0, (kt + 1) * BK, g.M, g.K, tid);
        sb.load(B, g.ldb, n0, (kt + 1) * BK, g.N, g.K, tid);
      }
    }
    if constexpr (GL) {
      // both k-halves' fragments up front (two register sets); the MFMAs of half 0 run under the reads of half 1
      uint4 af[2][4], bf[2][4];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[kk][i] = lds_read128_asm(la + swz(wm * 64 + 16 * i + fr, 4 * kk + fq));
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[kk][j] = lds_read128_asm(lb + swz(wn * 64 + 16 * j + fr, 4 * kk + fq));
        if (kk == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma_frag<T>(bf[0][j], af[0][i], acc[i][j]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma_frag<T>(bf[1][j], af[1][i], acc[i][j]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next stage has landed (this wave's pieces)
      __builtin_amdgcn_s_barrier();                     // ... everybody's; and everybody is done reading `cur`
      continue;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[i] = *reinterpret_cast<const uint4*>(la + swz(wm * 64 + 16 * i + fr, 4 * kk + fq));
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bf[j] = *reinterpret_cast<const uint4*>(lb + swz(wn * 64 + 16 * j + fr, 4 * kk + fq));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (ACC)
            mma_frag<T>(af[i], bf[j], acc[i][j]);
          else
            mma_frag<T>(bf[j], af[i], acc[i][j]);
        }
    }
    if constexpr (!GL) {
      if (more) {
        char* na = smem + (cur ^ 1) * 2 * TILE_BYTES;
        sa.store(na, tid);
        sb.store(na + TILE_BYTES, tid);
      }
    }
    __syncthreads();  // with LDS-DMA outstanding hipcc drains vmcnt(0) here: the next stage is complete
  }

  // ---------------------------------------------------------------- epilogue
  if constexpr (ACC) {
    float* C = static_cast<float*>(g.C);
    const bool alone = gridDim.z == 1;  // one K slice: this workgroup is the tile's only writer -> plain read-add-write
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + 16 * j + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + wm * 64 + 16 * i + 4 * fq + r;
          if (m < g.M && n < g.N) {
            float* c = C + (int64_t)m * g.ldc + n;
            if (alone) *c += acc[i][j][r];
            else atomicAdd(c, acc[i][j][r]);
          }
        }
      }
  } else {
    if constexpr (EPI < 0) epi_prefetch<T, 4, 4, EPI>(pre, g, m0 + wm * 64, n0 + wn * 64, fr, fq);
    epilogue_tile<T, TC, 4, 4, EPI>(acc, pre, g, m0 + wm * 64, n0 + wn * 64, fr, fq, reinterpret_cast<float*>(smem), wm, wn);
  }
}

// ---- ring kernel for the token-parallel Linears: forward (B = W [N,K]) and input gradient (B = W [K,N]) ---------
// PMC on the 128x128 kernel (65536x1152x384): the L2 -> LDS intake (903 MB per launch at ~12 TB/s chip-wide) is the
// longest phase and the waves are parked 46 % of their cycles on the 2-stage pipeline; its input-gradient path stages
// the K-major weight through registers (v_perm + ds_write_b64: 20 % LDS conflicts, 5x more VALU than MFMA
// instructions).  Same two-workgroups-per-CU structure (independent barriers: one workgroup's MFMAs run under the
// other's waits, and tile ends / store bursts stagger by themselves), but:
//   * K-step 32, everything staged by LDS-DMA into a ring (counted vmcnt waits, one barrier per K-step, fragment
//     reads as inline asm, see lds_read128_asm): FI = 8 -> 256x128 tile (85 flop per L2 byte instead of 64), 3 stages
//     of 24 KB; FI = 4 -> 128x128 tile, 4 stages of 16 KB;
//   * A (activations / output gradients, K-contiguous): image [rows][64 B], 16-byte chunk c of row r at position
//     c ^ G[(r>>2)&3], G = {0,3,2,1} (conflict-free in the four ds_read_b128 lane groups); the DMA writes lane-linear,
//     so the swizzle is applied to the per-lane source address;
//   * B: TB = 0 the same image (weight rows are K-contiguous); TB = 1 the K-major weight goes to LDS untouched as a
//     [32 k][128 n] sub-image and the fragments are gathered by ds_read_b64_tr_b16 (layout: gemm_tr_kernel below).

// CONV (implicit-GEMM 3x3 convolution, channels-last, padding 1; no im2col matrix in HBM): the A rows are gathered --
// LDS-DMA takes a per-lane source address, so a K-step of a tap reads the tile's pixels shifted by that tap and the
// zero page where the tap falls outside the image.  K runs channel-chunk-major, tap-minor: the nine taps of a 32-channel
// chunk re-read the same few KB of activations (L1 / L2 hits) before the next chunk is touched.
//   CONV = 1 forward:  Y[(b,oy,ox), co] = sum_{tap,c} X[b, oy s + ky - 1, ox s + kx - 1, c] W[co][tap][c]   (TB = 0)
//   CONV = 2 dgrad:    dX[(b,iy,ix), c] = sum_{tap,co} dY[b, (iy + 1 - ky) / s, (ix + 1 - kx) / s, co] W[co][tap][c]
//                      (TB = 1: for a fixed tap the weight is a [co][c] matrix with row stride 9 C)
template <typename TC, int EPI, bool TB, int FI, int CONV = 0>
__global__ void __launch_bounds__(256, 2) gemm_r3_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;
  static_assert(FI == 8 || FI == 4, "256x128 or 128x128");
  constexpr int RBM = 32 * FI, RBN = 128;
  constexpr int A_BYTES = RBM * R_ROWB, STAGE = A_BYTES + R_BSUB;
  constexpr int NST = FI == 8 ? 3 : 4;
  constexpr int QA = FI / 2, PS = QA + 2;  // DMA instructions per wave per K-step: A pieces + 2 B pieces
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int nblk = g.tiles_m * g.tiles_n;
  int tile;
  {  // XCD-aware tile order as in gemm_kernel
    const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3;
    const int q = nblk >> 3, rm = nblk & 7;
    tile = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + loc;
  }
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * RBM, n0 = tn * RBN;
  const int nk = g.K >> 5;

  // per-lane DMA sources (rows / columns past the operand are clamped: their products are never stored)
  const int prow = lane >> 2;
  const int csrc = ((lane & 3) ^ r_gsw(prow)) & 3;  // logical chunk that must land at position lane & 3
  const T* pa[QA];
  const T* pb[2];
  int cy[QA], cx[QA];  // CONV: row -> (image base folded into pa, y, x) of the output (1) / input (2) pixel
  const T* const zsrc = static_cast<const T*>(g.zero) + 8 * 0;
#pragma unroll
  for (int q = 0; q < QA; ++q) {
    int row = m0 + 16 * (wave + 4 * q) + prow;
    if constexpr (CONV == 0) {
      if (row >= g.M) row = g.M - 1;
      pa[q] = static_cast<const T*>(g.A) + (int64_t)row * g.lda + 8 * csrc;
    } else {
      // rows of this GEMM = pixels of the (CONV 1: output, CONV 2: input) image; the gathered tensor is the other one
      const int RH = CONV == 1 ? g.cHo : g.cH, RW = CONV == 1 ? g.cWo : g.cW;  // row image
      const int GH = CONV == 1 ? g.cH : g.cHo, GW = CONV == 1 ? g.cW : g.cWo;  // gathered image
      int b, rem, y, x;
      divmod24(row < g.M ? row : 0, RH * RW, 1.f / (float)(RH * RW), b, rem);
      divmod24(rem, RW, 1.f / (float)RW, y, x);
      if (row >= g.M) y = -100000;  // every tap out of range -> zero rows
      cy[q] = y;
      cx[q] = x;
      pa[q] = static_cast<const T*>(g.A) + (int64_t)b * GH * GW * g.lda + 8 * csrc;
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if constexpr (!TB) {
      int row = n0 + 16 * (wave + 4 * q) + prow;
      if (row >= g.N) row = g.N - 1;
      pb[q] = static_cast<const T*>(g.B) + (int64_t)row * g.ldb + 8 * csrc;
    } else {  // piece P = wave + 4 q: k-rows 4 P .. 4 P + 3 of the sub-image, 256 B each
      const int drow = lane >> 4;
      const int dchunk = (lane & 15) ^ (((drow & 3) << 2) | (wave & 3));  // (P & 3) == (wave & 3)
      int x = n0 + 8 * dchunk;
      if (x > g.N - 8) x = g.N - 8;
      pb[q] = static_cast<const T*>(g.B) + (int64_t)(4 * (wave + 4 * q) + drow) * g.ldb + x;
    }
  }
  const int64_t bstep = TB ? (int64_t)32 * g.ldb : 32;
  auto issue = [&](int s) {
    char* st = smem + (s % NST) * STAGE + wave * 1024;
    int64_t boff = s * bstep;
    if constexpr (CONV != 0) {
      const int ch = s / 9, tap = s - 9 * ch, ky = tap / 3, kx = tap - 3 * ky;  // wave-uniform
      // weight: [co][tap][c].  forward (TB = 0): row co, columns tap C + 32 ch;  dgrad (TB = 1): rows 32 ch .. of the
      // [co][c] matrix of this tap (row stride ldb = 9 C)
      boff = TB ? (int64_t)32 * ch * g.ldb + tap * g.cC : (int64_t)tap * g.cC + 32 * ch;
#pragma unroll
      for (int q = 0; q < QA; ++q) {
        int gy, gx;
        bool ok;
        if constexpr (CONV == 1) {
          gy = cy[q] * g.cS + ky - 1;
          gx = cx[q] * g.cS + kx - 1;
          ok = gy >= 0 && gy < g.cH && gx >= 0 && gx < g.cW;
        } else {
          const int ty = cy[q] + 1 - ky, tx = cx[q] + 1 - kx;
          ok = ty >= 0 && tx >= 0;
          if (g.cS == 2) {
            ok = ok && !((ty | tx) & 1);
            gy = ty >> 1;
            gx = tx >> 1;
          } else {
            gy = ty;
            gx = tx;
          }
          ok = ok && gy < g.cHo && gx < g.cWo;
        }
        const int GW = CONV == 1 ? g.cW : g.cWo;
        const T* src = ok ? pa[q] + (int64_t)(gy * GW + gx) * g.lda + 32 * ch : zsrc;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(st + q * 4096), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int q = 0; q < QA; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa[q] + s * 32),
                                         (__attribute__((address_space(3))) void*)(st + q * 4096), 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb[q] + boff),
                                       (__attribute__((address_space(3))) void*)(st + A_BYTES + q * 4096), 16, 0, 0);
  };

  f32x4 acc[FI][4];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  EpiPre<T, FI, 4> pre;
  if constexpr (EPI >= 0) epi_prefetch<T, FI, 4, EPI>(pre, g, m0 + wm * 16 * FI, n0 + wn * 64, fr, fq);

  const unsigned smem_base = (unsigned)(size_t)((const __attribute__((address_space(3))) char*)smem);
  const unsigned a_off = (unsigned)r_swz(wm * 16 * FI + fr, fq);
  const unsigned b_off = (unsigned)(A_BYTES + r_swz(wn * 64 + fr, fq));                     // TB = 0
  const unsigned b_t0 = A_BYTES + tr_lane_base(lane, 0, 8 * wn), b_t1 = A_BYTES + tr_lane_base(lane, 1, 8 * wn);  // TB = 1

  issue(0);
  if (nk > 1) issue(1);
  if (NST > 3 && nk > 2) issue(2);
  for (int s = 0; s < nk; ++s) {
    // K-step s has landed (this wave's pieces); the younger operations are the pieces of the steps issued after it
    const int ahead = nk - 1 - s < NST - 2 ? nk - 1 - s : NST - 2;
    if (ahead >= 2) r_wait_vm<2 * PS>();
    else if (ahead == 1) r_wait_vm<PS>();
    else r_wait_vm<0>();
    __builtin_amdgcn_s_barrier();              // ... everybody's; and everybody is done reading stage (s-1) % NST
    if (s + NST - 1 < nk) issue(s + NST - 1);  // -> stage (s-1) % NST
    const unsigned sb0 = smem_base + (unsigned)((s % NST) * STAGE);
    uint4 bf[4], af[FI];
    if constexpr (!TB) {
      bf[0] = r_read128<0>(sb0 + b_off);
      bf[1] = r_read128<1024>(sb0 + b_off);
      bf[2] = r_read128<2048>(sb0 + b_off);
      bf[3] = r_read128<3072>(sb0 + b_off);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint2 lo = t_read_tr<0>(sb0 + (b_t0 ^ (unsigned)(j << 5)));
        const uint2 hi = t_read_tr<0>(sb0 + (b_t1 ^ (unsigned)(j << 5)));
        bf[j] = uint4{lo.x, lo.y, hi.x, hi.y};
      }
    }
    const unsigned sa = sb0 + a_off;
    af[0] = r_read128<0>(sa);
    af[1] = r_read128<1024>(sa);
    af[2] = r_read128<2048>(sa);
    af[3] = r_read128<3072>(sa);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (FI == 8) {
      af[4] = r_read128<4096>(sa);
      af[5] = r_read128<5120>(sa);
      af[6] = r_read128<6144>(sa);
      af[7] = r_read128<7168>(sa);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) mma_frag<T>(bf[j], af[i], acc[i][j]);
    if constexpr (FI == 8) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 4; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma_frag<T>(bf[j], af[i], acc[i][j]);
    }
  }
  if constexpr (EPI < 0) epi_prefetch<T, FI, 4, EPI>(pre, g, m0 + wm * 16 * FI, n0 + wn * 64, fr, fq);
  epilogue_tile<T, TC, FI, 4, EPI>(acc, pre, g, m0 + wm * 16 * FI, n0 + wn * 64, fr, fq, reinterpret_cast<float*>(smem),
                                   wm, wn);
}

// ---- 256x256 tile, one 8-wave workgroup per CU: 128 flop per L2 -> LDS byte (the 256x128 ring: 85) -----------------
// Not persistent on purpose: a persistent variant with the next tile's first stage in flight under the epilogue was
// 20 % SLOWER (vmcnt also counts the epilogue's stores, and the CUs' store bursts line up); as separate workgroups
// the tiles drift apart by themselves.
// A [M,K] K-contiguous, K-step 64 (128-byte rows: whole cache lines per DMA row), two stages of 64 KB.
// Waves 2 (M) x 4 (N): each 128 x 64 (FI = 8, FJ = 4).  A and (TB = 0) W [N,K]: the 128-row sub-tiles of gemm_kernel's
// LDS-DMA path (glds_tile / swz), A sub-tile = wm, W sub-tile = wn >> 1.  TB = 1 (input gradients, W [K,N]): four
// [32 k][128 n] sub-images per stage (k-half, n-half) read by ds_read_b64_tr_b16 exactly as in gemm_r3_kernel.
template <typename TC, int EPI, bool TB>
__global__ void __launch_bounds__(512, 2) gemm_big_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;
  constexpr int STAGE = 4 * TILE_BYTES;  // A0 | A1 | W0 | W1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
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
  const T* A = static_cast<const T*>(g.A);
  const T* B = static_cast<const T*>(g.B);
  const int half = tid >> 8, t256 = tid & 255;
  // TB = 1: this wave moves piece P = wave (k-rows 4P .. 4P+3, 256 B each) of each of the four sub-images
  const T* pbt[2] = {nullptr, nullptr};
  if constexpr (TB) {
    const int drow = lane >> 4;
    const int dchunk = (lane & 15) ^ (((drow & 3) << 2) | (wave & 3));
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      int x = n0 + 128 * nh + 8 * dchunk;
      if (x > g.N - 8) x = g.N - 8;
      pbt[nh] = B + (int64_t)(4 * wave + drow) * g.ldb + x;
    }
  }
  auto issue = [&](int s) {
    char* st = smem + (s & 1) * STAGE;
    glds_tile<T>(A, g.lda, m0 + 128 * half, s * 64, g.M, st + half * TILE_BYTES, t256);
    if constexpr (!TB) {
      glds_tile<T>(B, g.ldb, n0 + 128 * half, s * 64, g.N, st + (2 + half) * TILE_BYTES, t256);
    } else {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)(pbt[nh] + (int64_t)(64 * s + 32 * kh) * g.ldb),
              (__attribute__((address_space(3))) void*)(st + 2 * TILE_BYTES + (2 * kh + nh) * R_BSUB + wave * 1024), 16,
              0, 0);
    }
  };
  const unsigned smem_base = (unsigned)(size_t)((const __attribute__((address_space(3))) char*)smem);
  const unsigned b_t0 = tr_lane_base(lane, 0, 8 * (wn & 1)), b_t1 = tr_lane_base(lane, 1, 8 * (wn & 1));
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(0);
  for (int s = 0; s < nk; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stage s landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();                     // ... everybody's; everybody is done reading stage s - 1
    if (s + 1 < nk) issue(s + 1);
    const char* la = smem + (s & 1) * STAGE + wm * TILE_BYTES;
    const char* lb = smem + (s & 1) * STAGE + (2 + (wn >> 1)) * TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 bf[4], af[8];
      if constexpr (!TB) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = lds_read128_asm(lb + swz((wn & 1) * 64 + 16 * j + fr, 4 * kk + fq));
      } else {
        const unsigned sub = smem_base + (unsigned)((s & 1) * STAGE + 2 * TILE_BYTES + (2 * kk + (wn >> 1)) * R_BSUB);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint2 lo = t_read_tr<0>(sub + (b_t0 ^ (unsigned)(j << 5)));
          const uint2 hi = t_read_tr<0>(sub + (b_t1 ^ (unsigned)(j << 5)));
          bf[j] = uint4{lo.x, lo.y, hi.x, hi.y};
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = lds_read128_asm(la + swz(16 * i + fr, 4 * kk + fq));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 4; i < 8; ++i) af[i] = lds_read128_asm(la + swz(16 * i + fr, 4 * kk + fq));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma_frag<T>(bf[j], af[i], acc[i][j]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 4; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma_frag<T>(bf[j], af[i], acc[i][j]);
    }
  }
  // (epilogue inputs are fetched here, not before the K loop: the accumulators leave no registers to hold them)
  EpiPre<T, 8, 4> pre;
  epi_prefetch<T, 8, 4, EPI>(pre, g, m0 + wm * 128, n0 + wn * 64, fr, fq);
  // dGELU column sums: two column groups of 128 (wn >> 1), each folded by its 2 x 2 waves; waves 0-3 flush
  epilogue_tile<T, TC, 8, 4, EPI>(acc, pre, g, m0 + wm * 128, n0 + wn * 64, fr, fq,
                                  reinterpret_cast<float*>(smem) + (wn >> 1) * 256, wm, wn & 1, wm == 0 ? (tid & 127) : 128);
}

// ---- 192x384 tile, 8 waves (2 x 4 of 96 x 96: FI = FJ = 6): the N = 384 / 1152 Linears ---------------------------
// With N = 384 the 256x128 ring reads every A row-panel three times (once per column tile); this tile covers the whole
// width, so A crosses L2 -> LDS once (128 flop per byte, as the 256x256 kernel, without its column padding).
// Same structure as gemm_big_kernel: K-step 64 (128-byte rows), two stages of 72 KB, one workgroup per CU.
// Images: A 192 rows | W 384 rows (TB = 0, swz) or 2 x 3 sub-images [32 k][128 n] (TB = 1, transposing reads).
template <typename TC, int EPI, bool TB>
__global__ void __launch_bounds__(512, 2) gemm_wide_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;
  constexpr int FI = 6, FJ = 6, BMR = 192, BNC = 384;
  constexpr int A_BYTES = BMR * ROW_BYTES, STAGE = (BMR + BNC) * ROW_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int nblk = g.tiles_m * g.tiles_n;
  int tile;
  {  // XCD-aware tile order as in gemm_kernel
    const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3;
    const int q = nblk >> 3, rm = nblk & 7;
    tile = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + loc;
  }
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * BMR, n0 = tn * BNC;
  const int nk = g.K >> 6;
  const T* A = static_cast<const T*>(g.A);
  const T* B = static_cast<const T*>(g.B);

  // per-lane DMA sources; piece p = wave + 8 q is rows 8p .. 8p+7 of an image (lane: row lane>>3, 16-byte slot lane&7
  // receives the logical chunk the swizzle assigns to that slot).  Rows past the operand are clamped.
  const T* pa[3];
  const T* pb[6];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int row = 8 * (wave + 8 * q) + (lane >> 3);
    const int c = ((lane & 7) ^ (row >> 1) ^ (row >> 4)) & 7;
    int grow = m0 + row;
    if (grow >= g.M) grow = g.M - 1;
    pa[q] = A + (int64_t)grow * g.lda + 8 * c;
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    if constexpr (!TB) {
      const int row = 8 * (wave + 8 * q) + (lane >> 3);
      const int c = ((lane & 7) ^ (row >> 1) ^ (row >> 4)) & 7;
      int grow = n0 + row;
      if (grow >= g.N) grow = g.N - 1;
      pb[q] = B + (int64_t)grow * g.ldb + 8 * c;
    } else {  // q = 3 kh + nh: piece P = wave (k-rows 4P .. 4P+3) of sub-image (kh, nh)
      const int kh = q / 3, nh = q - 3 * kh;
      const int drow = lane >> 4;
      const int dchunk = (lane & 15) ^ (((drow & 3) << 2) | (wave & 3));
      int x = n0 + 128 * nh + 8 * dchunk;
      if (x > g.N - 8) x = g.N - 8;
      pb[q] = B + (int64_t)(32 * kh + 4 * wave + drow) * g.ldb + x;
    }
  }
  // N = 384: this workgroup is the only reader of its A rows -> streaming (nt) DMA, so that the once-read activation does
  // not displace what the neighbouring launches re-read (per-kernel time unchanged, whole step +1.4 % on the same box)
  const bool a_once = g.tiles_n == 1;
  auto issue = [&](int s) {
    char* st = smem + (s & 1) * STAGE;
    if (a_once) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa[q] + 64 * s),
                                         (__attribute__((address_space(3))) void*)(st + (wave + 8 * q) * 1024), 16, 0, 2);
    } else {
#pragma unroll
      for (int q = 0; q < 3; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa[q] + 64 * s),
                                         (__attribute__((address_space(3))) void*)(st + (wave + 8 * q) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      if constexpr (!TB)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb[q] + 64 * s),
                                         (__attribute__((address_space(3))) void*)(st + A_BYTES + (wave + 8 * q) * 1024),
                                         16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb[q] + (int64_t)(64 * s) * g.ldb),
                                         (__attribute__((address_space(3))) void*)(st + A_BYTES + q * R_BSUB + wave * 1024),
                                         16, 0, 0);
    }
  };
  const unsigned smem_base = (unsigned)(size_t)((const __attribute__((address_space(3))) char*)smem);
  // TB = 1: fragment j starts at column 96 wn + 16 j: sub-image (col >> 7), first 8-column chunk (col & 127) >> 3
  unsigned tb0[FJ], tb1[FJ];
  if constexpr (TB) {
#pragma unroll
    for (int j = 0; j < FJ; ++j) {
      const int col = 96 * wn + 16 * j;
      tb0[j] = A_BYTES + (col >> 7) * R_BSUB + tr_lane_base(lane, 0, (col & 127) >> 3);
      tb1[j] = A_BYTES + (col >> 7) * R_BSUB + tr_lane_base(lane, 1, (col & 127) >> 3);
    }
  }
  f32x4 acc[FI][FJ];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(0);
  for (int s = 0; s < nk; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stage s landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();                     // ... everybody's; everybody is done reading stage s - 1
    if (s + 1 < nk) issue(s + 1);
    const char* la = smem + (s & 1) * STAGE;
    const char* lb = la + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 bf[FJ], af[FI];
      if constexpr (!TB) {
#pragma unroll
        for (int j = 0; j < FJ; ++j) bf[j] = lds_read128_asm(lb + swz(96 * wn + 16 * j + fr, 4 * kk + fq));
      } else {
        const unsigned sb = smem_base + (unsigned)((s & 1) * STAGE + kk * 3 * R_BSUB);
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
          const uint2 lo = t_read_tr<0>(sb + tb0[j]);
          const uint2 hi = t_read_tr<0>(sb + tb1[j]);
          bf[j] = uint4{lo.x, lo.y, hi.x, hi.y};
        }
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) af[i] = lds_read128_asm(la + swz(96 * wm + 16 * i + fr, 4 * kk + fq));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 3; i < FI; ++i) af[i] = lds_read128_asm(la + swz(96 * wm + 16 * i + fr, 4 * kk + fq));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) mma_frag<T>(bf[j], af[i], acc[i][j]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 3; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) mma_frag<T>(bf[j], af[i], acc[i][j]);
    }
  }
  EpiPre<T, FI, FJ> pre;
  epi_prefetch<T, FI, FJ, EPI>(pre, g, m0 + wm * 96, n0 + wn * 96, fr, fq);
  epilogue_tile<T, TC, FI, FJ, EPI>(acc, pre, g, m0 + wm * 96, n0 + wn * 96, fr, fq, nullptr, wm, wn & 1);
}

// ---- 64x128 tile for small token counts (a 128x128 grid that would leave most CUs idle) ---------------------------
// Per-GPU batch 16 (the reference yaml) is M = 4096 rows: 32 x 3 = 96 tiles of 128x128 for an N = 384 Linear on 256
// CUs.  Half-height tiles double the workgroups; 4 waves of 32 x 64 (FI = 2, FJ = 4), K-step 64, three LDS-DMA stages of
// 24 KB.  Operand images as in gemm_kernel's LDS-DMA path (TB = 0) / gemm_r3_kernel's transposing reads (TB = 1).
constexpr int M64_NST = 3;
template <typename TC, int EPI, bool TB>
__global__ void __launch_bounds__(256, 2) gemm_m64_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;
  constexpr int A_BYTES = 64 * ROW_BYTES, STAGE = A_BYTES + TILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int nblk = g.tiles_m * g.tiles_n;
  int tile;
  {  // XCD-aware tile order as in gemm_kernel
    const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3;
    const int q = nblk >> 3, rm = nblk & 7;
    tile = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + loc;
  }
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * 64, n0 = tn * 128;
  const int nk = g.K >> 6;
  const T* A = static_cast<const T*>(g.A);
  const T* B = static_cast<const T*>(g.B);
  // A: pieces p = wave + 4 q (rows 8p .. 8p+7); TB = 1: piece P = wave + 4 q of the k-half sub-images
  const T* pa[2];
  const T* pbt[2] = {nullptr, nullptr};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = 8 * (wave + 4 * q) + (lane >> 3);
    const int c = ((lane & 7) ^ (row >> 1) ^ (row >> 4)) & 7;
    int grow = m0 + row;
    if (grow >= g.M) grow = g.M - 1;
    pa[q] = A + (int64_t)grow * g.lda + 8 * c;
    if constexpr (TB) {
      const int drow = lane >> 4;
      const int dchunk = (lane & 15) ^ (((drow & 3) << 2) | (wave & 3));
      int x = n0 + 8 * dchunk;
      if (x > g.N - 8) x = g.N - 8;
      pbt[q] = B + (int64_t)(4 * (wave + 4 * q) + drow) * g.ldb + x;
    }
  }
  auto issue = [&](int s) {
    char* st = smem + (s % M64_NST) * STAGE;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa[q] + 64 * s),
                                       (__attribute__((address_space(3))) void*)(st + (wave + 4 * q) * 1024), 16, 0, 0);
    if constexpr (!TB) {
      glds_tile<T>(B, g.ldb, n0, s * 64, g.N, st + A_BYTES, tid);
    } else {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)(pbt[q] + (int64_t)(64 * s + 32 * kh) * g.ldb),
              (__attribute__((address_space(3))) void*)(st + A_BYTES + kh * R_BSUB + (wave + 4 * q) * 1024), 16, 0, 0);
    }
  };
  const unsigned smem_base = (unsigned)(size_t)((const __attribute__((address_space(3))) char*)smem);
  const unsigned b_t0 = A_BYTES + tr_lane_base(lane, 0, 8 * wn), b_t1 = A_BYTES + tr_lane_base(lane, 1, 8 * wn);
  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  EpiPre<T, 2, 4> pre;
  epi_prefetch<T, 2, 4, EPI>(pre, g, m0 + wm * 32, n0 + wn * 64, fr, fq);

  // 3-stage ring, two K-steps in flight (6 DMA instructions per wave and stage, counted waits).  With one step in flight
  // a K-step cost a whole load latency (~0.9 us: the fc1 input gradient at batch 16, K = 1536, took 22 us).
  issue(0);
  if (nk > 1) issue(1);
  for (int s = 0; s < nk; ++s) {
    if (s + 1 < nk) r_wait_vm<6>(); else r_wait_vm<0>();  // stage s landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();                         // ... everybody's; everybody is done reading stage s - 1
    if (s + 2 < nk) issue(s + 2);
    const char* la = smem + (s % M64_NST) * STAGE;
    const char* lb = la + A_BYTES;
    uint4 af[2][2], bf[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < 2; ++i) af[kk][i] = lds_read128_asm(la + swz(wm * 32 + 16 * i + fr, 4 * kk + fq));
      if constexpr (!TB) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[kk][j] = lds_read128_asm(lb + swz(wn * 64 + 16 * j + fr, 4 * kk + fq));
      } else {
        const unsigned sb = smem_base + (unsigned)((s % M64_NST) * STAGE + kk * R_BSUB);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint2 lo = t_read_tr<0>(sb + (b_t0 ^ (unsigned)(j << 5)));
          const uint2 hi = t_read_tr<0>(sb + (b_t1 ^ (unsigned)(j << 5)));
          bf[kk][j] = uint4{lo.x, lo.y, hi.x, hi.y};
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma_frag<T>(bf[kk][j], af[kk][i], acc[i][j]);
  }
  epilogue_tile<T, TC, 2, 4, EPI>(acc, pre, g, m0 + wm * 32, n0 + wn * 64, fr, fq, reinterpret_cast<float*>(smem), wm, wn);
}

// ---- weight-gradient kernel: both operands K-major (dW[M,N] += A[K,M]^T . B[K,N]), bf16, split-K + fp32 atomics ----
// PMC on the register-transposing path (1536x384x65536): MFMA busy 20 %, a third of the LDS cycles are the 2-way
// conflicts of the transposing ds_write_b64, and with 128x128 tiles the launch pulls 1.2 GB through L2.  Here the
// K-major tiles go to LDS untouched by LDS-DMA ([k][128 x] sub-images of 32 rows x 256 B) and the MFMA fragments
// are gathered by the CDNA4 transposing read ds_read_b64_tr_b16 (16 lanes read a 4 x 16 block and receive it
// column-major: lane i gets column i of 4 consecutive k) -- no VGPR round trip, no ds_write, no permutes.
// Tile 256x128 or 128x256 (FI x FJ = 8x4 / 4x8 accumulators per wave, 2x2 waves), K-step 32, 3-stage ring of 24 KB
// -> two workgroups per CU as gemm_r3_kernel.  Sub-image layout (cdna_hip_programming.md T10, image (b)):
// 16-byte chunk ch of k-row r at  256 r + 16 (ch ^ (((r & 3) << 2) | ((r >> 2) & 3))); the DMA writes lane-linear
// (4 rows per wave-instruction), so the XOR is applied to the per-lane source column.
constexpr int T_SUB = 32 * 256;        // one sub-image: 32 k-rows x 128 elements
constexpr int T_STAGE = 3 * T_SUB;     // A sub-images then B sub-images (2 + 1 or 1 + 2)
constexpr int T_NST = 3;
constexpr int T_PS = 6;                // DMA instructions per wave per K-step (24 pieces of 4 rows / 4 waves)


// PART: the split-K partial goes to a dense scratch [split][M][N] with plain 16-byte stores (swapped MFMA operands:
// a lane owns 4 consecutive columns) and splitk_reduce_kernel adds the slices to C -- global fp32 atomics move only
// ~1.3 TB/s chip-wide, and with ~500 workgroups x 128 KB of accumulators they cost as much as the whole K loop.
// CONVW (weight gradient of the implicit-GEMM 3x3 convolution): dW[co][(tap, c)] += sum_m dY[m][co] X[pixel(m) + tap][c].
// The B rows (K index m = output pixel) are gathered per lane: column x -> (tap, c) is fixed per lane, the pixel of
// row m is recomputed every K-step (two divmod24), padded positions read the zero page.
template <int FI, int FJ, bool PART, bool CONVW = false>
__global__ void __launch_bounds__(256, 2) gemm_tr_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;
  static_assert((FI == 8 && FJ == 4) || (FI == 4 && FJ == 8), "256x128 or 128x256");
  constexpr int TBM = 32 * FI, TBN = 32 * FJ;
  constexpr int NA = TBM / 128;  // A sub-images per stage (B: 3 - NA)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  // Workgroup -> (output tile, K slice z).  All tiles of one K slice read the same A / B rows, so they must share an
  // L2: workgroups b and b+8 land on the same XCD (round-robin dispatch), hence XCD x = b & 7 takes the slices
  // z = x, x+8, ... and walks the tiles of one slice before the next.  (With the plain (tile, z) grid the tiles of
  // a slice were spread over all 8 XCDs: PMC showed 73 % L2 misses and ~700 MB of fabric reads per launch for
  // 250 MB of operands.)  g.wide carries the number of slices.
  // Weights with MANY tiles and a short reduction (the UNet's Linears: 400 tiles, 192 K-steps) need no 8-fold split for
  // parallelism, and 8 fp32 slices of such an output are far more traffic than the operands.  There the 8 XCDs form
  // xs slice lanes x 8 / xs tile lanes: XCD x takes the slices z = (x mod xs) + xs j of the tiles whose row (part_m) or
  // column index is congruent to x / xs -- an XCD still reads only its own share of one operand.  xs = 8 is the case above.
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int sl = xcd & (g.xs - 1), tl = xcd / g.xs, TL = 8 / g.xs;
  const int jz = loc / g.nloc, rr = loc - jz * g.nloc;
  const int zsl = sl + g.xs * jz;
  if (zsl >= g.wide) return;  // uniform per block
  int tm, tn;
  if (g.part_m) {
    const int u = rr / g.tiles_n;
    tm = tl + TL * u;
    tn = rr - u * g.tiles_n;
  } else {
    const int u = rr / g.tiles_m;
    tn = tl + TL * u;
    tm = rr - u * g.tiles_m;
  }
  if (tm >= g.tiles_m || tn >= g.tiles_n) return;  // uniform per block
  const int m0 = tm * TBM, n0 = tn * TBN;
  const int s_begin = zsl * g.k_tiles_per_split;  // K-steps of 32 rows
  int s_end = s_begin + g.k_tiles_per_split;
  if (s_end > (g.K >> 5)) s_end = g.K >> 5;
  const int ns = s_end - s_begin;
  if (ns <= 0) return;  // uniform per block

  // ---- DMA: this wave's pieces P = wave + 4q (q < 6); sub-image P >> 3, rows 4 (P & 7) .. +3 of it
  const int drow = lane >> 4;
  const int dsw = ((drow & 3) << 2) | (wave & 3);  // f(row) of the destination row: (P & 7) & 3 == wave & 3
  const int dchunk = (lane & 15) ^ dsw;            // logical chunk that must land at position lane & 15
  const T* src[T_PS];
  int ctap[T_PS];  // CONVW: ky * 4 + kx of this lane's column in piece q
#pragma unroll
  for (int q = 0; q < T_PS; ++q) {
    const int P = wave + 4 * q, S = P >> 3, lp = P & 7;
    const bool isA = S < NA;
    int x = (isA ? m0 + 128 * S : n0 + 128 * (S - NA)) + 8 * dchunk;
    const int X = isA ? g.M : g.N;
    if (x > X - 8) x = X - 8;  // columns past the operand: clamped (their products are never accumulated)
    const T* base = static_cast<const T*>(isA ? g.A : g.B);
    ctap[q] = 0;
    if (CONVW && !isA) {
      int tap, c;
      divmod24(x, g.cC, 1.f / (float)g.cC, tap, c);
      const int ky = tap / 3;
      ctap[q] = ky * 4 + (tap - 3 * ky);
      src[q] = base + c;
    } else {
      src[q] = base + (int64_t)(s_begin * 32 + 4 * lp + drow) * (isA ? g.lda : g.ldb) + x;
    }
  }
  const int64_t stepA = (int64_t)32 * g.lda, stepB = (int64_t)32 * g.ldb;
  const float rcp_img = CONVW ? 1.f / (float)(g.cHo * g.cWo) : 0.f, rcp_w = CONVW ? 1.f / (float)g.cWo : 0.f;
  auto issue = [&](int s) {  // s = step index relative to s_begin
    char* st = smem + (s % T_NST) * T_STAGE;
#pragma unroll
    for (int q = 0; q < T_PS; ++q) {
      const int P = wave + 4 * q, S = P >> 3, lp = P & 7;
      const T* p;
      if (CONVW && S >= NA) {
        const int m = (s_begin + s) * 32 + 4 * lp + drow;  // output pixel (b, oy, ox); g.K = B Ho Wo is a multiple of 32
        int b, rem, oy, ox;
        divmod24(m, g.cHo * g.cWo, rcp_img, b, rem);
        divmod24(rem, g.cWo, rcp_w, oy, ox);
        const int gy = oy * g.cS + (ctap[q] >> 2) - 1, gx = ox * g.cS + (ctap[q] & 3) - 1;
        const bool ok = gy >= 0 && gy < g.cH && gx >= 0 && gx < g.cW;
        p = ok ? src[q] + ((int64_t)(b * g.cH + gy) * g.cW + gx) * g.cC : static_cast<const T*>(g.zero);
      } else {
        p = src[q] + s * (S < NA ? stepA : stepB);
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(st + S * T_SUB + lp * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[FI][FJ];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Fused bias gradient: g.bias != NULL asks for bias[m] += sum_k A[k][m] (the column sums of dY).  That is one more
  // output column with B = 1: extra MFMAs against a constant all-ones fragment (the MFMA pipe is 20 % busy in this
  // kernel), and the separate colsum pass over dY (28 us per Linear) disappears.
  // The tile's first column block does it (tn == 0); its two waves of equal wm split the FI row-fragments in halves
  // (two code copies, so the FI / 2 extra accumulators keep compile-time indices).
  const bool do_sum = g.bias != nullptr && tn == 0;  // wave-uniform
  constexpr int FH = FI / 2;
  f32x4 sacc[FH];
#pragma unroll
  for (int i = 0; i < FH; ++i) sacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};  // 8 x bf16(1.0)

  // ---- transposed fragment reads: lane = 16 g + 4 q + p supplies row 8 g + 4 t + q, columns 4 p .. 4 p + 3 of the
  // fragment's 16-column block; fragment fi only flips chunk bits: address ^ (fi << 5)
  const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  auto tr_base = [&](int t, int xb8) {  // xb8 = (first column of the wave's share inside the sub-image) / 8
    const int krow = 8 * tg + 4 * t + tq;
    const int f = (tq << 2) | ((2 * tg + t) & 3);
    return (unsigned)(256 * krow + 16 * ((xb8 + (tp >> 1)) ^ f) + 8 * (tp & 1));
  };
  const unsigned smem_base = (unsigned)(size_t)((const __attribute__((address_space(3))) char*)smem);
  // A: FI == 8 -> the wave owns sub-image wm entirely; FI == 4 -> columns 64 wm .. of the single sub-image
  const unsigned a_sub = (FI == 8) ? wm * T_SUB : 0, a_xb8 = (FI == 8) ? 0 : 8 * wm;
  const unsigned b_sub = NA * T_SUB + ((FJ == 8) ? wn * T_SUB : 0), b_xb8 = (FJ == 8) ? 0 : 8 * wn;
  const unsigned a_t0 = a_sub + tr_base(0, a_xb8), a_t1 = a_sub + tr_base(1, a_xb8);
  const unsigned b_t0 = b_sub + tr_base(0, b_xb8), b_t1 = b_sub + tr_base(1, b_xb8);

  issue(0);
  if (ns > 1) issue(1);
  for (int s = 0; s < ns; ++s) {
    if (s + 1 < ns) r_wait_vm<T_PS>(); else r_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (s + 2 < ns) issue(s + 2);
    const unsigned sb = smem_base + (unsigned)((s % T_NST) * T_STAGE);
    uint4 af[FI], bf[FJ];
#pragma unroll
    for (int j = 0; j < FJ; ++j) {
      const uint2 lo = t_read_tr<0>(sb + (b_t0 ^ (unsigned)(j << 5)));
      const uint2 hi = t_read_tr<0>(sb + (b_t1 ^ (unsigned)(j << 5)));
      bf[j] = uint4{lo.x, lo.y, hi.x, hi.y};
    }
#pragma unroll
    for (int i = 0; i < FI; ++i) {
      const uint2 lo = t_read_tr<0>(sb + (a_t0 ^ (unsigned)(i << 5)));
      const uint2 hi = t_read_tr<0>(sb + (a_t1 ^ (unsigned)(i << 5)));
      af[i] = uint4{lo.x, lo.y, hi.x, hi.y};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FJ; ++j) {
        if constexpr (PART)
          mma_frag<T>(bf[j], af[i], acc[i][j]);
        else
          mma_frag<T>(af[i], bf[j], acc[i][j]);
      }
    if (do_sum) {
      if (wn == 0) {
#pragma unroll
        for (int i = 0; i < FH; ++i) {
          if constexpr (PART) mma_frag<T>(ones, af[i], sacc[i]);
          else mma_frag<T>(af[i], ones, sacc[i]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < FH; ++i) {
          if constexpr (PART) mma_frag<T>(ones, af[FH + i], sacc[i]);
          else mma_frag<T>(af[FH + i], ones, sacc[i]);
        }
      }
    }
  }
  if (do_sum) {
    float* bg = const_cast<float*>(g.bias);
#pragma unroll
    for (int i = 0; i < FH; ++i) {
      const int mb = m0 + wm * 16 * FI + 16 * (wn * FH + i);
      if constexpr (PART) {  // D[n][m]: column m = fr on the lane, every row equal
        const int m = mb + fr;
        if (fq == 0 && m < g.M) atomicAdd(bg + m, sacc[i][0]);
      } else {  // D[m][n]: rows 4 fq + r in the registers, every column equal
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = mb + 4 * fq + r;
          if (fr == 0 && m < g.M) atomicAdd(bg + m, sacc[i][r]);
        }
      }
    }
  }
  if constexpr (PART) {
    if (g.wide == 1) {  // one slice: the tile is this workgroup's alone -> plain read-modify-write of C
      float* C = static_cast<float*>(g.C);
#pragma unroll
      for (int i = 0; i < FI; ++i) {
        const int m = m0 + wm * 16 * FI + 16 * i + fr;
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
          const int n = n0 + wn * 16 * FJ + 16 * j + 4 * fq;
          if (m < g.M && n < g.N) {
            float* c = C + (int64_t)m * g.ldc + n;
            store4(c, load4(c) + acc[i][j]);
          }
        }
      }
      return;
    }
    float* P = static_cast<float*>(g.C2) + (int64_t)zsl * g.M * g.N;
#pragma unroll
    for (int i = 0; i < FI; ++i) {
      const int m = m0 + wm * 16 * FI + 16 * i + fr;
#pragma unroll
      for (int j = 0; j < FJ; ++j) {
        const int n = n0 + wn * 16 * FJ + 16 * j + 4 * fq;
        if (m < g.M && n < g.N) store4(P + (int64_t)m * g.N + n, acc[i][j]);
      }
    }
    return;
  }
  // atomic accumulate (registers walk rows, lanes walk 16 consecutive columns -> 64-byte atomic segments)
  float* C = static_cast<float*>(g.C);
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FJ; ++j) {
      const int n = n0 + wn * 16 * FJ + 16 * j + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 16 * FI + 16 * i + 4 * fq + r;
        if (m < g.M && n < g.N) atomicAdd(C + (int64_t)m * g.ldc + n, acc[i][j][r]);
      }
    }
}

// ---- wide streaming weight gradient: 192 x 384 or 384 x 192 tile, ONE 8-wave workgroup per CU ------------------------------
// Why: all tiles of a K slice read the same rows, so the UNIQUE bytes a launch has in flight are only
// (slices running at once) x (stages in flight) x (32 rows x (M + N) x 2 B).  With 256x128 tiles the 18 tiles of an fc1 slice
// leave room for ~3.5 slices per XCD: ~6 MB in flight chip-wide, which at ~2.6 us of loaded HBM latency is the 2.3-2.6 TB/s
// the 256x128 kernel measures (a third of the HBM rate, although it only READS).  A tile that spans the whole 384-wide
// operand needs 8 tiles per slice: 4 slices per XCD at one workgroup per CU, a 4-stage ring (3 in flight), every operand row
// crosses L2 -> LDS once per 192 (384) output rows instead of once per 128 -- ~2.5x the unique bytes in flight.
// Structure as gemm_tr_kernel (K-major operands untouched in LDS, ds_read_b64_tr_b16 fragments, split-K partials to a
// scratch + splitk_reduce_kernel, fused bias gradient); waves WM x WN, wave tile 96 x 96 (FI = FJ = 6: 144 accumulator
// registers; a 256-row tile needs 192 and spilled).  The 192-wide operand fills one and a half [32 k][128 x] sub-images: the
// DMA lanes of the unused half are masked off.
constexpr int W_NSUB = 5;                 // sub-images [32 k][128 x] per stage: NA for A, 5 - NA for B
constexpr int W_STAGE = W_NSUB * T_SUB;   // 40 KB
// Who waits for what in this kernel (at ~250 VGPRs the compiler copies registers around, and it believes an inline-asm
// ds_read has delivered at its #ASMEND -- a copy it placed between such a read and the hand-written lgkmcnt wait carried
// the PREVIOUS K-step's fragment into the bias MFMA; cdna_hip_programming.md section 5.7 item 1):
//   * fragment reads are the BUILTIN transposing read, so hipcc counts lgkmcnt itself and may interleave them with MFMAs;
//   * the LDS-DMA is inline asm (m0 set and restored inside the statement): invisible to hipcc, so it neither waits
//     vmcnt(0) before the visible reads nor drains the ring at the barrier; its completion is the hand-counted vmcnt wait.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16_asm(const void* gsrc, unsigned lds_dst /* wave-uniform */) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
template <int WM, int WN, int FI, int FJ, int NST>
__global__ void __launch_bounds__(512, 2) gemm_trw_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;
  static_assert(WM * WN == 8, "8 waves");
  constexpr int TBM = 16 * FI * WM, TBN = 16 * FJ * WN;
  constexpr int NA = (TBM + 127) / 128, NB = (TBN + 127) / 128;
  static_assert(NA + NB == W_NSUB, "five sub-images per stage");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int fr = lane & 15, fq = lane >> 4;
  const int nblk = g.tiles_m * g.tiles_n;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int zsl = xcd + 8 * (loc / nblk), tile = loc % nblk;
  if (zsl >= g.wide) return;  // uniform per block
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * TBM, n0 = tn * TBN;
  const int s_begin = zsl * g.k_tiles_per_split;
  int s_end = s_begin + g.k_tiles_per_split;
  if (s_end > (g.K >> 5)) s_end = g.K >> 5;
  const int ns = s_end - s_begin;
  if (ns <= 0) return;  // uniform per block

  // ---- DMA: wave w moves rows 4w .. 4w+3 of every sub-image (piece q = sub-image q)
  const int drow = lane >> 4;
  const int dchunk = (lane & 15) ^ (((drow & 3) << 2) | (wave & 3));
  const T* src[W_NSUB];
  bool live[W_NSUB];  // lanes whose 8 columns lie inside the tile (the last sub-image of a 192-wide operand is half used)
#pragma unroll
  for (int q = 0; q < W_NSUB; ++q) {
    const bool isA = q < NA;
    const int xl = 128 * (isA ? q : q - NA) + 8 * dchunk;  // column inside the tile
    live[q] = xl < (isA ? TBM : TBN);
    int x = (isA ? m0 : n0) + xl;
    const int X = isA ? g.M : g.N;
    if (x > X - 8) x = X - 8;  // columns past the operand: clamped (their products are never stored)
    src[q] = static_cast<const T*>(isA ? g.A : g.B) + (int64_t)(s_begin * 32 + 4 * wave + drow) * (isA ? g.lda : g.ldb) + x;
  }
  const int64_t stepA = (int64_t)32 * g.lda, stepB = (int64_t)32 * g.ldb;
  const unsigned smem_base = (unsigned)(size_t)((const __attribute__((address_space(3))) char*)smem);
  auto issue = [&](int s) {
    const unsigned st = smem_base + (unsigned)((s % NST) * W_STAGE + wave * 1024);
#pragma unroll
    for (int q = 0; q < W_NSUB; ++q)
      if (live[q])  // (EXEC-masked DMA: the other lanes' LDS slots keep stale bytes no fragment reads; every wave has live lanes)
        glds16_asm(src[q] + s * (q < NA ? stepA : stepB), st + q * T_SUB);
  };

  f32x4 acc[FI][FJ];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fused bias gradient (column sums of A = dY): row fragment fi is summed by wave column fi % WN, in its slot fi / WN
  const bool do_sum = g.bias != nullptr && tn == 0;  // wave-uniform
  constexpr int FS = (FI + WN - 1) / WN;
  f32x4 sacc[FS];
#pragma unroll
  for (int i = 0; i < FS; ++i) sacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};

  // fragment gidx of an operand (16 columns each, 8 per sub-image): sub-image gidx >> 3, chunk bits (gidx & 7) << 5
  const unsigned t0 = tr_lane_base(lane, 0, 0), t1 = tr_lane_base(lane, 1, 0);
  auto rd = [&](unsigned off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) char*)smem + off));
  };
  auto frag = [&](unsigned stage_off, int gidx) {
    const unsigned sub = stage_off + (unsigned)((gidx >> 3) * T_SUB), fl = (unsigned)((gidx & 7) << 5);
    const s16x4 lo = rd(sub + (t0 ^ fl)), hi = rd(sub + (t1 ^ fl));
    uint4 r;
    r.x = ((unsigned)(unsigned short)lo[0]) | ((unsigned)(unsigned short)lo[1] << 16);
    r.y = ((unsigned)(unsigned short)lo[2]) | ((unsigned)(unsigned short)lo[3] << 16);
    r.z = ((unsigned)(unsigned short)hi[0]) | ((unsigned)(unsigned short)hi[1] << 16);
    r.w = ((unsigned)(unsigned short)hi[2]) | ((unsigned)(unsigned short)hi[3] << 16);
    return r;
  };
  const int ga0 = FI * wm, gb0 = 8 * NA + FJ * wn;  // (16-column fragment index counted over the operand's sub-images)

  constexpr int AHEAD = NST - 1;
  static_assert(NST == 4, "the vmcnt ladder below is written for three K-steps ahead");
#pragma unroll
  for (int s = 0; s < AHEAD; ++s)
    if (s < ns) issue(s);
  constexpr int GI = FI / 2;  // A fragments per half
  for (int s = 0; s < ns; ++s) {
    // K-step s has landed once at most the pieces of the (up to AHEAD - 1) younger steps are outstanding
    const int younger = ns - 1 - s < AHEAD - 1 ? ns - 1 - s : AHEAD - 1;
    if (younger >= 2) r_wait_vm<2 * W_NSUB>();
    else if (younger == 1) r_wait_vm<W_NSUB>();
    else r_wait_vm<0>();
    __syncthreads();  // everybody's pieces of step s; everybody is done reading stage (s - 1) % NST
    if (s + AHEAD < ns) issue(s + AHEAD);
    const unsigned so = (unsigned)((s % NST) * W_STAGE);
    uint4 bf[FJ], af[GI];
#pragma unroll
    for (int j = 0; j < FJ; ++j) bf[j] = frag(so, gb0 + j);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
      for (int i = 0; i < GI; ++i) af[i] = frag(so, ga0 + GI * hh + i);
#pragma unroll
      for (int i = 0; i < GI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) mma_frag<T>(bf[j], af[i], acc[GI * hh + i][j]);
      if (do_sum) {
#pragma unroll
        for (int i = 0; i < GI; ++i) {
          const int fi = GI * hh + i;
          if (fi % WN == wn) mma_frag<T>(ones, af[i], sacc[fi / WN]);
        }
      }
    }
  }
  if (do_sum) {
    float* bg = const_cast<float*>(g.bias);
#pragma unroll
    for (int i = 0; i < FS; ++i) {  // D[n][m]: column m = fr on the lane, every row equal
      const int fi = i * WN + wn;
      const int m = m0 + wm * 16 * FI + 16 * fi + fr;
      if (fi < FI && fq == 0 && m < g.M) atomicAdd(bg + m, sacc[i][0]);
    }
  }
  float* P = static_cast<float*>(g.C2) + (int64_t)zsl * g.M * g.N;
#pragma unroll
  for (int i = 0; i < FI; ++i) {
    const int m = m0 + wm * 16 * FI + 16 * i + fr;
#pragma unroll
    for (int j = 0; j < FJ; ++j) {
      const int n = n0 + wn * 16 * FJ + 16 * j + 4 * fq;
      if (m < g.M && n < g.N) store4(P + (int64_t)m * g.N + n, acc[i][j]);
    }
  }
}

// ---- A-stationary Linear for K = 384 with a store-heavy epilogue (fc1 + bias + GELU: 1208 MB out for 151 MB in) -------------
// In gemm_big_kernel a tile's K loop and its store phase follow each other (one workgroup per CU, whose LDS is not released
// before the stores are acknowledged): 438 us = ~190 us of loop + ~230 us of drain at the chip's write rate.  Here a workgroup
// owns 256 rows for its whole life and walks the N columns in chunks of 64:
//   * A (32 rows x 384 per wave) is loaded ONCE, straight into the MFMA fragment registers (96 VGPRs) -- it never touches LDS;
//   * W chunks [64 n][384 k] (48 KB) stream through three LDS stages by LDS-DMA; a chunk's 12 K-steps run without a barrier;
//   * the chunk's epilogue (bias from an LDS copy, GELU, two paired 16-byte stores per fragment pair) issues its 8 store
//     instructions and moves on: vector-memory operations retire in issue order, the next chunk's DMA was issued BEFORE these
//     stores, so `s_waitcnt vmcnt(8)` waits for the DMA alone and the stores drain under the next chunk's MFMAs.
// Every wave reads the whole W chunk from LDS (8-fold): LDS and MFMA time are equal (3072 clocks per chunk), the stores need
// 64 KB per chunk per CU -- the kernel is bound by the chip's write rate, not by the sum of the phases.
constexpr int AS_K = 384, AS_BN = 64, AS_STAGE = AS_BN * AS_K * 2;  // 48 KB per W chunk
constexpr int AS_NST = 3;
constexpr int AS_LDS = AS_NST * AS_STAGE + 8192;                    // + the bias vector (<= 2048 columns) as fp32
template <typename TC, int EPI>
__global__ void __launch_bounds__(512, 2) gemm_as_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;
  constexpr int FI = 2, FJ = 4, KS = AS_K / 32;  // wave tile 32 rows x 64 columns; 12 K-steps of 32
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int m0 = blockIdx.x * 256 + wave * 32;
  const T* A = static_cast<const T*>(g.A);
  const T* B = static_cast<const T*>(g.B);
  float* bias_lds = reinterpret_cast<float*>(smem + AS_NST * AS_STAGE);
  if (EPI == UWU_EPI_BIAS || EPI == UWU_EPI_BIAS_GELU)
    for (int n = tid; n < g.N; n += 512) bias_lds[n] = g.bias[n];

  // A fragments: lane (fr, fq) holds A[m0 + 16 i + fr][32 s + 8 fq .. + 7]
  uint4 af[FI][KS];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int s = 0; s < KS; ++s)
      af[i][s] = *reinterpret_cast<const uint4*>(A + (int64_t)(m0 + 16 * i + fr) * g.lda + 32 * s + 8 * fq);

  // W chunk DMA: piece q of wave w = 64-column K block q (8 KB sub-image, swizzled rows of 128 B), rows 8 w .. 8 w + 7
  const int drow = 8 * wave + (lane >> 3);
  const int dc = ((lane & 7) ^ (drow >> 1) ^ (drow >> 4)) & 7;
  const T* bsrc = B + (int64_t)drow * g.ldb + 8 * dc;
  const unsigned smem_base = (unsigned)(size_t)((const __attribute__((address_space(3))) char*)smem);
  auto issue = [&](int c) {
    const T* p = bsrc + (int64_t)c * AS_BN * g.ldb;
#pragma unroll
    for (int q = 0; q < AS_K / 64; ++q)  // (asm DMA: hipcc must not see it, or it drains vmcnt(0) in front of every LDS read)
      glds16_asm(p + 64 * q, smem_base + (unsigned)((c % AS_NST) * AS_STAGE + q * 8192 + wave * 1024));
  };
  const int nchunks = g.N / AS_BN;
  // Three stages, the DMA runs two chunks ahead.  The epilogue of chunk c - 1 (bias, GELU, rounding, lane exchange, stores:
  // VALU + vector-memory work) is cut into four units and issued BETWEEN the K-steps of chunk c, whose MFMAs run in the matrix
  // pipe meanwhile: with a whole-chunk epilogue after the K loop all eight waves sat in the same phase between the per-chunk
  // barriers (231 us of K loops + 237 us of epilogues, nothing overlapped).  In issue order a wave has, at the top of chunk c:
  //   .. DMA(c) | stores(c-3) | DMA(c+1) | stores(c-2)      (NS = 8 store instructions and 6 DMA instructions per chunk)
  // and needs DMA(c): everything younger may stay in flight -> vmcnt(2 NS + 6).
  static_assert(EPI == UWU_EPI_BIAS_GELU || EPI == UWU_EPI_BIAS || EPI == UWU_EPI_DGELU, "the interleaved epilogues");
  // vector-memory instructions of one chunk's epilogue: 8 stores (two outputs), 4 stores, or 4 stores + 4 aux loads (dGELU)
  constexpr int NS = EPI == UWU_EPI_BIAS ? 4 : 8;
  TC* const C = static_cast<TC*>(g.C);
  TC* const C2 = static_cast<TC*>(g.C2);
  const bool odd = fq & 1;
  auto pack = [](const f32x4& v) {
    bf16x4 b = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    return *reinterpret_cast<uint2*>(&b);
  };
  // one unit = row block i, fragment pair jp of the chunk at columns n0: 8 consecutive columns per lane after the exchange
  // dGELU: the pre-activation tile of unit u (8 consecutive columns per lane, the layout of the paired stores) is loaded right
  // after the unit of the previous chunk has consumed its registers -- a whole chunk period before it is needed
  uint4 auxr[4];
  const T* const auxp = static_cast<const T*>(g.aux);
  auto aux_load = [&](int n0, int u) {
    const int i = u >> 1, jp = u & 1;
    const int m = m0 + 16 * i + fr, n = n0 + 32 * jp + (odd ? 16 + 4 * (fq - 1) : 4 * fq);
    auxr[u] = *reinterpret_cast<const uint4*>(auxp + (int64_t)m * g.ldaux + n);
  };
  auto epi_unit = [&](const f32x4 (&pa)[FI][FJ], int n0, int i, int jp) {
    f32x4 v0 = pa[i][2 * jp], v1 = pa[i][2 * jp + 1];
    typedef unsigned su32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned su32x4 __attribute__((ext_vector_type(4)));
    if constexpr (EPI == UWU_EPI_DGELU) {
      const uint4 a = auxr[2 * i + jp];  // un-swap: this lane's own 4 columns of both fragments
      const su32x2 sx = __builtin_amdgcn_permlane16_swap(a.x, a.z, false, false);
      const su32x2 sy = __builtin_amdgcn_permlane16_swap(a.y, a.w, false, false);
      const uint2 r0 = {sx[0], sy[0]}, r1 = {sx[1], sy[1]};
      const bf16x4 u0 = *reinterpret_cast<const bf16x4*>(&r0), u1 = *reinterpret_cast<const bf16x4*>(&r1);
      v0 = v0 * dgelu_tanh_f4(f32x4{(float)u0[0], (float)u0[1], (float)u0[2], (float)u0[3]});
      v1 = v1 * dgelu_tanh_f4(f32x4{(float)u1[0], (float)u1[1], (float)u1[2], (float)u1[3]});
    } else {
      const float* bl = bias_lds + n0 + 32 * jp + 4 * fq;
      v0 = v0 + *reinterpret_cast<const f32x4*>(bl);
      v1 = v1 + *reinterpret_cast<const f32x4*>(bl + 16);
    }
    const int m = m0 + 16 * i + fr;
    const int n = n0 + 32 * jp + (odd ? 16 + 4 * (fq - 1) : 4 * fq);
    auto exchange_store = [&](TC* dst, const f32x4& x0, const f32x4& x1, bool stream_out) {
      const uint2 p0 = pack(x0), p1 = pack(x1);
      const su32x2 sx = __builtin_amdgcn_permlane16_swap(p0.x, p1.x, false, false);
      const su32x2 sy = __builtin_amdgcn_permlane16_swap(p0.y, p1.y, false, false);
      const su32x4 o = su32x4{sx[0], sy[0], sx[1], sy[1]};
      su32x4* ptr = reinterpret_cast<su32x4*>(dst + (int64_t)m * g.ldc + n);
      if (stream_out) __builtin_nontemporal_store(o, ptr);
      else *ptr = o;
    };
    if constexpr (EPI == UWU_EPI_BIAS_GELU) {
      exchange_store(C, v0, v1, true);  // pre-activation: only read again in the backward pass
      exchange_store(C2, gelu_tanh_f4(v0), gelu_tanh_f4(v1), false);
    } else {
      exchange_store(C, v0, v1, false);
    }
  };
  f32x4 prev[FI][FJ];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FJ; ++j) prev[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  issue(0);
  issue(1);
  for (int c = 0; c < nchunks; ++c) {
    if (c == 0) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");  // DMA(1) may fly; this thread's bias words are in LDS
    else if (c <= 2 || c + 1 >= nchunks) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // head / tail of the sequence
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NS + 6) : "memory");
    __builtin_amdgcn_s_barrier();  // chunk c landed for everybody; everybody is done reading the stage of chunk c - 1
    if (c + 2 < nchunks) issue(c + 2);
    const char* lb = smem + (c % AS_NST) * AS_STAGE;
    f32x4 acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragments of K-step s + 1 are requested before the MFMAs of step s (two register sets): with one wave in the matrix
    // pipe per SIMD at a time nothing else hides the LDS latency
    // (the dGELU variant has no registers for the second set: one spilled register would put scratch accesses into the counted
    // vmcnt sequence)
    constexpr int NBF = EPI == UWU_EPI_DGELU ? 1 : 2;
    uint4 bf[NBF][FJ];
    auto frags = [&](uint4 (&dst)[FJ], int s) {
#pragma unroll
      for (int j = 0; j < FJ; ++j)
        dst[j] = *reinterpret_cast<const uint4*>(lb + (s >> 1) * 8192 + swz(16 * j + fr, 4 * (s & 1) + fq));
    };
    if constexpr (NBF == 2) frags(bf[0], 0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if constexpr (NBF == 2) {
        if (s + 1 < KS) frags(bf[(s + 1) & 1], s + 1);
      } else {
        frags(bf[0], s);
      }
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) mma_frag<T>(bf[s & (NBF - 1)][j], af[i][s], acc[i][j]);
      if (s % 3 == 1) {  // units after K-steps 1, 4, 7, 10
        if (c > 0) epi_unit(prev, (c - 1) * AS_BN, (s / 3) >> 1, (s / 3) & 1);
        if constexpr (EPI == UWU_EPI_DGELU) aux_load(c * AS_BN, s / 3);
      }
    }
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FJ; ++j) prev[i][j] = acc[i][j];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) epi_unit(prev, (nchunks - 1) * AS_BN, u >> 1, u & 1);
}

// ---- fp8 (OCP e4m3 / e5m2) operands on the block-scaled MFMA: BASELINE config 5 ("fp8 MFMA GEMMs") -----------------------
// v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (E8M0 = 127) runs at twice the bf16 rate (MI355X_MICROARCH.md,
// Matrix cores) -- the non-scaled fp8 MFMAs only reach the bf16 rate.  ONE kernel shape serves forward, input gradient and
// weight gradient because every operand is handed over contraction-contiguous ("NT"): the quantising kernels of
// quant.hip write the transposed fp8 copies (W^T for dgrad, dY^T / X^T for wgrad) while they convert.
//   C[M,N] = alpha * A[M,K] . B[N,K]^T,  alpha = 1 / (scale_a * scale_b)  (per-tensor quantisation scales, read from
//   device memory so that delayed scaling needs no host round trip), fp32 accumulate, bf16 (or fp32 partial) out.
// Structure = gemm_big_kernel: 256x256 tile, 8 waves (2 x 4 of 128 x 64), K-step = 128-byte rows = 128 fp8 = ONE MFMA
// per 16x16 output fragment and step (the bf16 kernel: two MFMAs of K = 32), two LDS-DMA stages of 64 KB, same swizzled
// image.  A lane's 32 operand bytes are 16-byte chunks fq and 4 + fq of the row -- a permutation of k applied to both
// operands alike, chosen because those are exactly the two conflict-free reads of the bf16 kernel.
// FA: element format of the A operand (0 = e4m3, 1 = e5m2: output gradients); B (weights / activations) is e4m3.
// PART: split-K partial sums (fp32) into a dense scratch [split][M][N]; splitk_reduce_kernel adds them to C.
struct f8_t { unsigned char v; };
template <> struct GT<f8_t> { static constexpr int EPC = 16; static constexpr int BK = 128; };
typedef int i32x8 __attribute__((ext_vector_type(8)));

// EMIT (UWU_EPI_BIAS_GELU / UWU_EPI_DGELU): the operand the NEXT fp8 GEMMs contract over is produced here instead of by a
// quantising pass over the bf16 result (quant.hip: 4 bytes of HBM traffic per element, 0.25 ms per [49152, 4608] tensor):
//   BIAS_GELU: C = bf16 pre-activation (kept for the backward pass), q8 / q8t = e4m3(gelu(.) * q_scale)
//   DGELU:     C2 = float[N] column sums if non-null, q8 / q8t = e5m2(result * q_scale); no bf16 copy (nothing reads it)
// The 256 x 256 result tile is staged in LDS twice -- as it is and transposed (a 4 x 4 byte block sits in the dwords of four
// neighbouring lanes: four quad broadcasts + two v_perm_b32 give each lane four consecutive ROWS of one column) -- and leaves
// as whole 256-byte rows of both images.
constexpr int F8Q_PITCH = 272;  // bytes per staged row (68 dwords: the dword writes of a wave spread over all 32 banks)
constexpr int F8_EMIT_LDS = 2 * 256 * F8Q_PITCH + 2 * 256 * 4 + 64;

template <int FMT>
__device__ __forceinline__ unsigned f8_pack4(const f32x4& v, float s) {
  const float mx = FMT == 0 ? 448.f : 57344.f;
  float a = fminf(fmaxf(v[0] * s, -mx), mx), b = fminf(fmaxf(v[1] * s, -mx), mx);
  float c = fminf(fmaxf(v[2] * s, -mx), mx), d = fminf(fmaxf(v[3] * s, -mx), mx);
  int r;
  if constexpr (FMT == 0) {
    r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  } else {
    r = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false);
    r = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, r, true);
  }
  return (unsigned)r;
}

// FULL: the tile has no row / column past M / N (a wave-uniform fact) -- the per-element masks of the ragged form (a v_cndmask per
// element and output, ~8 % of this epilogue's VALU) are compiled out.
template <int EPI, bool FULL = false>
__device__ __forceinline__ void f8_emit_epilogue(f32x4 (&acc)[8][4], const GemmArgs& g, char* smem, int m0, int n0, int tid) {
  constexpr int FMT = EPI == UWU_EPI_DGELU ? 1 : 0;
  constexpr int QP = F8Q_PITCH;
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3, fr = lane & 15, fq = lane >> 4;
  unsigned char* t_rm = reinterpret_cast<unsigned char*>(smem);
  unsigned char* t_tr = t_rm + 256 * QP;
  float* cs = reinterpret_cast<float*>(smem + 2 * 256 * QP);  // [2][256] column sums of the two wave rows
  float* red = cs + 512;                                      // [8] per-wave |max|
  bf16_t* C = static_cast<bf16_t*>(g.C);
  const bf16_t* aux = static_cast<const bf16_t*>(g.aux);
  float* colsum = EPI == UWU_EPI_DGELU ? reinterpret_cast<float*>(g.C2) : nullptr;
  const float qs = g.q_scale[0];
  const int m_w = m0 + wm * 128, n_w = n0 + wn * 64;
  f32x4 bias[4], csum[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n_w + 16 * j + 4 * fq;
    bias[j] = (EPI == UWU_EPI_BIAS_GELU && n < g.N) ? load4(g.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    csum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto aux_row = [&](int i, uint2 (&dst)[4]) {
    const int m = m_w + 16 * i + fr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n_w + 16 * j + 4 * fq;
      // (predicated in the FULL form too: unconditional, hipcc hoisted the aux loads of all eight rows and spilled 42 registers)
      dst[j] = (m < g.M && n < g.N) ? *reinterpret_cast<const uint2*>(aux + (int64_t)m * g.ldaux + n) : uint2{0u, 0u};
    }
  };
  uint2 ar[2][4];
  if constexpr (EPI == UWU_EPI_DGELU) aux_row(0, ar[0]);
  __syncthreads();  // every wave has left the K loop: the stages are free
  const bool odd = fq & 1;
  const int kq = lane & 3;
  const unsigned sel = 0x0c0c0400u + (unsigned)kq * 0x0101u;
  float mx = 0.f;
  auto pack = [](const f32x4& v) {
    bf16x4 b = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    return *reinterpret_cast<uint2*>(&b);
  };
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ml = wm * 128 + 16 * i + fr, m = m0 + ml;
    const bool mok = m < g.M;
    __builtin_amdgcn_sched_barrier(0);  // (keeps the unrolled rows apart: hoisted aux loads of later rows spilled registers)
    if constexpr (EPI == UWU_EPI_DGELU)
      if (i + 1 < 8) aux_row(i + 1, ar[(i + 1) & 1]);
    f32x4 v[4], o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n_w + 16 * j + 4 * fq;
      v[j] = acc[i][j];
      if constexpr (EPI == UWU_EPI_BIAS_GELU) {
        v[j] = v[j] + bias[j];
        o[j] = gelu_tanh_f4(v[j]);
      } else {
        const bf16x4 u = *reinterpret_cast<const bf16x4*>(&ar[i & 1][j]);
        v[j] = v[j] * dgelu_tanh_f4(f32x4{(float)u[0], (float)u[1], (float)u[2], (float)u[3]});
        o[j] = v[j];
        if (FULL || (mok && n < g.N)) csum[j] = csum[j] + v[j];
      }
      if constexpr (!FULL)
        if (!(mok && n < g.N)) o[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fabsf(o[j][e]));
      asm volatile("" : "+v"(mx));  // (taken now: left to the optimiser the max chain sank to the end of the tile and o[] was spilled)
      const unsigned pk = f8_pack4<FMT>(o[j], qs);
      const int nl = wn * 64 + 16 * j + 4 * fq;
      *reinterpret_cast<unsigned*>(t_rm + ml * QP + nl) = pk;
      // 4 x 4 byte transpose inside the quad of lanes that holds rows 4 (fr / 4) .. + 3 of these four columns
      // (quad broadcasts: every lane is written, so there is no "old" value to set up -- update_dpp(0, ..) cost a v_mov per DPP)
      const int pi = (int)pk;
      const unsigned d0 = (unsigned)__builtin_amdgcn_mov_dpp(pi, 0x00, 0xF, 0xF, true);
      const unsigned d1 = (unsigned)__builtin_amdgcn_mov_dpp(pi, 0x55, 0xF, 0xF, true);
      const unsigned d2 = (unsigned)__builtin_amdgcn_mov_dpp(pi, 0xAA, 0xF, 0xF, true);
      const unsigned d3 = (unsigned)__builtin_amdgcn_mov_dpp(pi, 0xFF, 0xF, 0xF, true);
      const unsigned lo = __builtin_amdgcn_perm(d1, d0, sel), hi = __builtin_amdgcn_perm(d3, d2, sel);
      *reinterpret_cast<unsigned*>(t_tr + (nl + kq) * QP + (ml & ~3)) = lo | (hi << 16);
      __builtin_amdgcn_sched_barrier(0);  // (fragment by fragment: the scheduler otherwise kept every o[] alive for the max chain and spilled)
    }
    if constexpr (EPI == UWU_EPI_BIAS_GELU) {  // bf16 pre-activation: paired 16-byte stores as epilogue_tile (8 consecutive columns per lane)
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        typedef unsigned su32x2 __attribute__((ext_vector_type(2)));
        typedef unsigned su32x4 __attribute__((ext_vector_type(4)));
        const uint2 p0 = pack(v[2 * jp]), p1 = pack(v[2 * jp + 1]);
        const su32x2 sx = __builtin_amdgcn_permlane16_swap(p0.x, p1.x, false, false);
        const su32x2 sy = __builtin_amdgcn_permlane16_swap(p0.y, p1.y, false, false);
        const int nb = n_w + 32 * jp;
        const int n = odd ? nb + 16 + 4 * (fq - 1) : nb + 4 * fq;
        if (FULL || (mok && n < g.N)) {
          const su32x4 ov = su32x4{sx[0], sy[0], sx[1], sy[1]};
          su32x4* ptr = reinterpret_cast<su32x4*>(C + (int64_t)m * g.ldc + n);
          __builtin_nontemporal_store(ov, ptr);  // read again in the backward pass only
        }
      }
    }
  }
  if (colsum) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 t = csum[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = row16_sum(t[e]);
      if (fr == 0) store4(cs + wm * 256 + wn * 64 + 16 * j + 4 * fq, t);
    }
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  if (colsum && tid < 256 && n0 + tid < g.N) atomicAdd(colsum + n0 + tid, cs[tid] + cs[256 + tid]);
  if (g.q_amax && tid == 0) {
    float a = red[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) a = fmaxf(a, red[w]);
    // (look first: atomics on one address serialise; almost every workgroup can skip it -- quant.hip)
    const unsigned cur = __hip_atomic_load(reinterpret_cast<unsigned*>(g.q_amax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__float_as_uint(a) > cur) atomicMax(reinterpret_cast<unsigned*>(g.q_amax), __float_as_uint(a));
  }
  // both images leave as whole rows: 16 lanes x 16 bytes = one 256-byte row per 16 threads, 32 rows per pass
  unsigned char* q8 = static_cast<unsigned char*>(g.q8);
  unsigned char* q8t = static_cast<unsigned char*>(g.q8t);
  const int r0 = tid >> 4, c16 = 16 * (tid & 15);
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int r = 32 * p + r0;
    if (q8 && m0 + r < g.M && n0 + c16 < g.N)
      *reinterpret_cast<uint4*>(q8 + (int64_t)(m0 + r) * g.ldq + n0 + c16) = *reinterpret_cast<const uint4*>(t_rm + r * QP + c16);
    if (q8t && n0 + r < g.N && m0 + c16 < g.M)
      *reinterpret_cast<uint4*>(q8t + (int64_t)(n0 + r) * g.ldqt + m0 + c16) = *reinterpret_cast<const uint4*>(t_tr + r * QP + c16);
  }
}

template <typename TC, int EPI, int FA, bool PART, bool EMIT = false>
__global__ void __launch_bounds__(512, 2) gemm_f8_kernel(const GemmArgs g, const float* __restrict__ scale_a,
                                                         const float* __restrict__ scale_b) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 4 * TILE_BYTES;  // A0 | A1 | W0 | W1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int nblk = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x, zsl = 0;
  if constexpr (PART) {  // K slices: XCD x takes slices x, x + 8, ... (all tiles of a slice share one L2, as gemm_tr_kernel)
    const int xcd = bid & 7, loc = bid >> 3;
    zsl = xcd + 8 * (loc / nblk);
    bid = loc % nblk;
    if (zsl >= g.wide) return;  // uniform per block
  }
  int tile = bid;
  if constexpr (!PART) {  // XCD-aware tile order as in gemm_kernel
    const int xcd = bid & 7, loc = bid >> 3;
    const int q = nblk >> 3, rm = nblk & 7;
    tile = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + loc;
  }
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  int s_begin = 0, s_end = g.K >> 7;
  if constexpr (PART) {
    s_begin = zsl * g.k_tiles_per_split;
    if (s_begin + g.k_tiles_per_split < s_end) s_end = s_begin + g.k_tiles_per_split;
    if (s_end <= s_begin) return;  // uniform per block
  }
  const f8_t* A = static_cast<const f8_t*>(g.A);
  const f8_t* B = static_cast<const f8_t*>(g.B);
  const int half = tid >> 8, t256 = tid & 255;
  auto issue = [&](int s) {
    char* st = smem + (s & 1) * STAGE;
    glds_tile<f8_t>(A, g.lda, m0 + 128 * half, s * 128, g.M, st + half * TILE_BYTES, t256);
    glds_tile<f8_t>(B, g.ldb, n0 + 128 * half, s * 128, g.N, st + (2 + half) * TILE_BYTES, t256);
  };
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto frag = [&](const char* base, int row) {
    const uint4 lo = lds_read128_asm(base + swz(row, fq)), hi = lds_read128_asm(base + swz(row, 4 + fq));
    return i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
  };

  issue(s_begin);
  for (int s = s_begin; s < s_end; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stage s landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();                     // ... everybody's; everybody is done reading stage s - 1
    if (s + 1 < s_end) issue(s + 1);
    const char* la = smem + (s & 1) * STAGE + wm * TILE_BYTES;
    const char* lb = smem + (s & 1) * STAGE + (2 + (wn >> 1)) * TILE_BYTES;
    i32x8 bf[4], af[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[j] = frag(lb, (wn & 1) * 64 + 16 * j + fr);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = frag(la, 64 * h + 16 * i + fr);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)  // swapped operands (a lane ends up with 4 consecutive columns): MFMA-A = weight fragment
          acc[4 * h + i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf[j], af[i], acc[4 * h + i][j], 0, FA, 0,
                                                                               0x7F7F7F7F, 0, 0x7F7F7F7F);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const float alpha = 1.f / (scale_a[0] * scale_b[0]);
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = acc[i][j] * alpha;
  if constexpr (PART) {
    float* P = static_cast<float*>(g.C2) + (int64_t)zsl * g.M * g.N;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + wm * 128 + 16 * i + fr;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + 16 * j + 4 * fq;
        if (m < g.M && n < g.N) store4(P + (int64_t)m * g.N + n, acc[i][j]);
      }
    }
  } else if constexpr (EMIT) {
    // (the dGELU form keeps the masked epilogue only: a second copy of it cost that kernel 32 spilled registers)
    if (EPI == UWU_EPI_BIAS_GELU && g.p8_cont && m0 + 256 <= g.M && n0 + 256 <= g.N) f8_emit_epilogue<EPI, EPI == UWU_EPI_BIAS_GELU>(acc, g, smem, m0, n0, tid);
    else f8_emit_epilogue<EPI, false>(acc, g, smem, m0, n0, tid);
  } else {
    EpiPre<bf16_t, 8, 4> pre;
    epi_prefetch<bf16_t, 8, 4, EPI>(pre, g, m0 + wm * 128, n0 + wn * 64, fr, fq);
    epilogue_tile<bf16_t, TC, 8, 4, EPI>(acc, pre, g, m0 + wm * 128, n0 + wn * 64, fr, fq,
                                         reinterpret_cast<float*>(smem) + (wn >> 1) * 256, wm, wn & 1,
                                         wm == 0 ? (tid & 127) : 128);
  }
}

template <typename T, typename TC, bool TA, bool TB, bool ACC, bool GL = false, int EPI = -1>
int launch(const GemmArgs& g, int split, hipStream_t st) {
  auto kern = gemm_kernel<T, TC, TA, TB, ACC, GL, EPI>;
  static bool attr_done = false;  // per instantiation
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                        4 * TILE_BYTES);
    attr_done = true;
  }
  dim3 grid(g.tiles_m * g.tiles_n, 1, split);
  UwuProfScope prof(st);
  hipLaunchKernelGGL(kern, grid, dim3(256), 4 * TILE_BYTES, st, g);
  prof.done(gemm_tag(g, TB, ACC), sizeof(T) == 2 ? 0 : 1, 2.0 * g.M * g.N * g.K, gemm_bytes(g, sizeof(T), sizeof(TC)));
  UWU_LAUNCH_CHECK("gemm");
  return UWU_OK;
}

template <typename TC, int EPI, bool TB, int FI, int CONV = 0>
int launch_r3(GemmArgs g, hipStream_t st) {
  auto kern = gemm_r3_kernel<TC, EPI, TB, FI, CONV>;
  constexpr int LDS = (FI == 8 ? 3 : 4) * (32 * FI * R_ROWB + R_BSUB);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  g.tiles_m = (g.M + 32 * FI - 1) / (32 * FI);
  g.tiles_n = (g.N + 127) / 128;
  UwuProfScope prof(st);
  hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n), dim3(256), LDS, st, g);
  if (CONV) prof.done(UWU_PROF_CONV, 0, 2.0 * g.M * g.N * g.K, ((double)g.M * g.K / 9 + (double)g.N * g.K + (double)g.M * g.N) * 2);
  else prof.done(gemm_tag(g, TB, false), 0, 2.0 * g.M * g.N * g.K, gemm_bytes(g, 2, sizeof(TC)));
  UWU_LAUNCH_CHECK("gemm_r3");
  return UWU_OK;
}
template <typename TC, int EPI, bool TB>
int launch_big(GemmArgs g, hipStream_t st) {
  auto kern = gemm_big_kernel<TC, EPI, TB>;
  constexpr int LDS = 2 * 4 * TILE_BYTES;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  g.tiles_m = (g.M + 255) / 256;
  g.tiles_n = (g.N + 255) / 256;
  UwuProfScope prof(st);
  hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n), dim3(512), LDS, st, g);
  prof.done(gemm_tag(g, TB, false), 0, 2.0 * g.M * g.N * g.K, gemm_bytes(g, 2, sizeof(TC)));
  UWU_LAUNCH_CHECK("gemm_big");
  return UWU_OK;
}
// A-stationary kernel: K = 384, whole 256-row panels, 64-column chunks, bf16 output with the paired 16-byte stores.
// UWU_GEMM_AS=0 turns it off (A/B comparisons).
static bool use_as(const GemmArgs& g, int out_bytes) {
  static UwuEnv on("UWU_GEMM_AS"), nmin_e("UWU_AS_NMIN");  // UWU_AS_NMIN=n: sweeps
  if (on.get().is('0')) return false;
  const int nmin = nmin_e.get().set ? nmin_e.ival : 1024;
  // one workgroup per 256 rows: below one per CU the chip is under-filled (per-GPU batch 64: 9.8k -> 8.2k images/s with it)
  return g.K == AS_K && g.M % 256 == 0 && g.M >= 256 * 256 && g.N % AS_BN == 0 && g.N <= 2048 && g.N >= nmin && out_bytes == 2 &&
         g.lda % 8 == 0 &&
         g.ldb % 8 == 0 && g.ldc % 8 == 0 && (((uintptr_t)g.A | (uintptr_t)g.B | (uintptr_t)g.C | (uintptr_t)g.C2) & 15) == 0;
}
static bool use_as_bias() {  // the plain bias Linears with N >= 1024 (qkv forward: 285 -> 231 us in the step); UWU_GEMM_AS_BIAS=0: off
  static UwuEnv on("UWU_GEMM_AS_BIAS");
  return !on.get().is('0');
}
template <typename TC, int EPI>
int launch_as(GemmArgs g, hipStream_t st) {
  auto kern = gemm_as_kernel<TC, EPI>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, AS_LDS);
    attr_done = true;
  }
  g.wide = 1;  // paired 16-byte stores
  UwuProfScope prof(st);
  hipLaunchKernelGGL(kern, dim3(g.M / 256), dim3(512), AS_LDS, st, g);
  prof.done(gemm_tag(g, false, false), 0, 2.0 * g.M * g.N * g.K, gemm_bytes(g, 2, sizeof(TC)));
  UWU_LAUNCH_CHECK("gemm_as");
  return UWU_OK;
}
template <typename TC, int EPI, bool TB>
int launch_m64(GemmArgs g, hipStream_t st) {
  auto kern = gemm_m64_kernel<TC, EPI, TB>;
  constexpr int LDS = M64_NST * (64 * ROW_BYTES + TILE_BYTES);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  g.tiles_m = (g.M + 63) / 64;
  g.tiles_n = (g.N + 127) / 128;
  UwuProfScope prof(st);
  hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n), dim3(256), LDS, st, g);
  prof.done(gemm_tag(g, TB, false), 0, 2.0 * g.M * g.N * g.K, gemm_bytes(g, 2, sizeof(TC)));
  UWU_LAUNCH_CHECK("gemm_m64");
  return UWU_OK;
}
// 64x128 kernel: when the 128x128 grid has fewer tiles than the chip has CUs.  UWU_GEMM_M64=0 turns it off.
static bool use_m64(const GemmArgs& g, bool tb) {
  static UwuEnv on("UWU_GEMM_M64"), thr_e("UWU_M64_TILES");  // UWU_M64_TILES=n: sweeps
  if (on.get().is('0')) return false;
  if (g.K % 64 || (((uintptr_t)g.A | (uintptr_t)g.B) & 15) || g.lda % 8 || g.ldb % 8) return false;
  if (tb && (g.N % 8 || g.N < 8)) return false;
  const int64_t tiles = (int64_t)((g.M + 127) / 128) * ((g.N + 127) / 128);
  const int thr = thr_e.get().set ? thr_e.ival : 256;
  return tiles < thr && g.M > 64;
}
template <typename TC, int EPI, bool TB>
int launch_wide(GemmArgs g, hipStream_t st) {
  auto kern = gemm_wide_kernel<TC, EPI, TB>;
  constexpr int LDS = 2 * (192 + 384) * ROW_BYTES;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  g.tiles_m = (g.M + 191) / 192;
  g.tiles_n = g.N / 384;
  UwuProfScope prof(st);
  hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n), dim3(512), LDS, st, g);
  prof.done(gemm_tag(g, TB, false), 0, 2.0 * g.M * g.N * g.K, gemm_bytes(g, 2, sizeof(TC)));
  UWU_LAUNCH_CHECK("gemm_wide");
  return UWU_OK;
}
// 192x384 kernel: N a multiple of 384 (and not of 256), and enough tiles that the last round of one-workgroup-per-CU
// tiles is not mostly empty.  Same box, M = 131072: qkv fwd 193 -> 183 us, fc2 fwd 207 -> 185, qkv / fc1 input gradients
// 149 -> 138 / 191 -> 172; at M = 65536 (342 tiles of 192 rows = 1.3 rounds of 256 CUs) it loses 5-10 %, hence the
// fill rule.  UWU_GEMM_WIDE=0 turns it off, =1 forces it (tests, A/B comparisons).
static bool use_wide(const GemmArgs& g) {
  if (g.K % 64 || g.N % 384 || g.N % 256 == 0) return false;
  static UwuEnv on("UWU_GEMM_WIDE");
  if (on.get().is('0')) return false;
  if (on.is('1')) return true;
  const int64_t tiles = (int64_t)((g.M + 191) / 192) * (g.N / 384);
  const int64_t rounds = (tiles + 255) / 256;
  return tiles * 100 >= rounds * 256 * 85;
}
// 256x256 kernel: taken where the 256x128 ring would be and N is a multiple of 256 (no padded column tiles).
// Same-box A/B of the whole step: DiT-S/2 +1.8 % (only its two GELU Linears qualify: fc1 + GELU 187 -> 167 us at B = 256;
// the dGELU input gradient is a wash there),
// DiT-B/2 +5.5 %, DiT-L/2 +1.9 %, SDXL UNet +-0.  UWU_GEMM_BIG=0 turns it off (A/B comparisons).
// (A masked ragged last column tile was tried on DiT-XL/2's N = 3456 / 1152 Linears: +0.4 % at 4 % padding, -1.8 % at
// 11 % -- not taken.)
static bool use_big(const GemmArgs& g) {
  static UwuEnv on("UWU_GEMM_BIG");
  return !on.get().is('0') && g.K % 64 == 0 && g.N % 256 == 0;
}
// C[m][n] += sum over the split-K slices of the scratch [split][M][N]; one float4 per thread
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ C,
                                                            int M, int N, int ldc, int split) {
  const int64_t slice = (int64_t)M * N;
  for (int64_t idx = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; idx < slice; idx += (int64_t)gridDim.x * 1024) {
    f32x4 v = load4(part + idx);
    for (int z = 1; z < split; ++z) v = v + load4(part + z * slice + idx);
    const int m = (int)(idx / N), n = (int)(idx - (int64_t)m * N);
    float* c = C + (int64_t)m * ldc + n;
    store4(c, load4(c) + v);
  }
}

// The same sum with the slices divided among four lanes of threads: a workgroup takes 64 float4 of the output per pass, thread
// (zl, cl) adds slices zl, zl + 4, .. of column group cl on two accumulators (loads of 8 slices in flight), the four partial sums
// meet in LDS.  With one thread per output float4 a [384 x 384] gradient in 128 slices was 36 864 threads walking 128 dependent
// adds each on 144 of the 256 CUs.
__global__ void __launch_bounds__(256) splitk_reduce4_kernel(const float* __restrict__ part, float* __restrict__ C,
                                                             int M, int N, int ldc, int split) {
  __shared__ f32x4 red[4][64];
  const int zl = threadIdx.x >> 6, cl = threadIdx.x & 63;
  const int64_t slice = (int64_t)M * N;
  for (int64_t base = (int64_t)blockIdx.x * 256; base < slice; base += (int64_t)gridDim.x * 256) {
    const int64_t idx = base + 4 * cl;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (idx < slice) {
      int z = zl;
      for (; z + 4 < split; z += 8) {
        v0 = v0 + load4(part + (int64_t)z * slice + idx);
        v1 = v1 + load4(part + (int64_t)(z + 4) * slice + idx);
      }
      if (z < split) v0 = v0 + load4(part + (int64_t)z * slice + idx);
    }
    red[zl][cl] = v0 + v1;
    __syncthreads();
    if (zl == 0 && idx < slice) {
      const f32x4 v = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
      const int m = (int)(idx / N), n = (int)(idx - (int64_t)m * N);
      float* c = C + (int64_t)m * ldc + n;
      store4(c, load4(c) + v);
    }
    __syncthreads();
  }
}
// launches the reduce (UWU_SPLITK_REDUCE4=0: the one-thread-per-float4 form)
static void launch_splitk_reduce(const float* part, float* C, int M, int N, int ldc, int split, hipStream_t st) {
  static UwuEnv r4("UWU_SPLITK_REDUCE4");
  const int64_t quads = (int64_t)M * N / 4;
  if (!r4.get().is('0') && split >= 8) {
    int rg = (int)((quads + 63) / 64);
    if (rg > 8192) rg = 8192;
    hipLaunchKernelGGL(splitk_reduce4_kernel, dim3(rg), dim3(256), 0, st, part, C, M, N, ldc, split);
    return;
  }
  int rg = (int)((quads + 255) / 256);
  if (rg > 4096) rg = 4096;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rg), dim3(256), 0, st, part, C, M, N, ldc, split);
}

// Number of K slices for the streaming weight-gradient kernel: a multiple of 8 (one group of slices per XCD), as
// many groups as fit the XCD's 64 workgroup slots (32 CUs x 2) in one round.
// Outputs with >= 64 tiles of a reduction of a few thousand rows: fewer slices, the XCDs divided between slices and tiles
// (gemm_tr_kernel's xs): every halving of the slice count halves the fp32 slice traffic.
int tr_split(int tiles, int steps) {
  static UwuEnv forced_e("UWU_TR_SPLIT"), min_e("UWU_TR_MINSTEPS");  // sweeps
  const int forced = forced_e.get().ival;
  int split;
  if (forced > 0) {
    split = forced;
  } else if (tiles >= 320 && steps <= 1024) {  // (sweep at 6144 / 24576 tokens: 400 tiles 361 -> 193 us, 200 tiles 166 -> 115,
    split = 1;                                 //  150 tiles 122 -> 107, 100 tiles 223 -> 217; 50 tiles stay at 8 slices)
  } else if (tiles >= 140 && steps <= 1024) {
    split = 2;
  } else if (tiles >= 80 && steps <= 1024) {
    split = 4;
  } else {
    int per_xcd = 64 / tiles;
    if (per_xcd < 1) per_xcd = 1;
    split = 8 * per_xcd;
    const int min_steps = min_e.get().set ? min_e.ival : 32;
    while (split > 8 && split * min_steps > steps) split -= 8;  // keep >= 32 K-steps per slice (batch 16: 3.82k -> 4.13k img/s, batch 64: 9.05k -> 9.79k with the four side streams)
  }
  if (split > steps) split = steps;
  return split < 1 ? 1 : split;
}
// slice lanes among the 8 XCDs: the largest power of two <= 8 that divides the slice count
int tr_xs(int split) { return split % 8 == 0 ? 8 : (split % 4 == 0 ? 4 : (split % 2 == 0 ? 2 : 1)); }

template <int FI, int FJ, bool CONVW = false>
int launch_tr(GemmArgs g, void* scratch, size_t scratch_bytes, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tr_kernel<FI, FJ, false, CONVW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, T_NST * T_STAGE);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tr_kernel<FI, FJ, true, CONVW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, T_NST * T_STAGE);
    attr_done = true;
  }
  g.tiles_m = (g.M + 32 * FI - 1) / (32 * FI);
  g.tiles_n = (g.N + 32 * FJ - 1) / (32 * FJ);
  const int tiles = g.tiles_m * g.tiles_n, steps = g.K / 32;
  int split = tr_split(tiles, steps);
  g.k_tiles_per_split = (steps + split - 1) / split;
  split = (steps + g.k_tiles_per_split - 1) / g.k_tiles_per_split;
  // 16-byte rows in the scratch and in C: slices go to the scratch (one slice: straight into C); otherwise 8 slice lanes + atomics
  const bool vec_ok = g.N % 4 == 0 && g.ldc % 4 == 0 && (((uintptr_t)g.C | (uintptr_t)scratch) & 15) == 0;
  const bool part = vec_ok && (split == 1 || (scratch && scratch_bytes >= (size_t)split * g.M * g.N * sizeof(float)));
  if (!part && split < 8 && steps >= 8) {  // the atomic path wants all XCDs through the slice lanes
    split = 8;
    g.k_tiles_per_split = (steps + split - 1) / split;
    split = (steps + g.k_tiles_per_split - 1) / g.k_tiles_per_split;
  }
  g.wide = split;
  g.xs = tr_xs(split);
  const int TL = 8 / g.xs;
  g.part_m = g.tiles_m >= g.tiles_n;
  g.nloc = g.part_m ? ((g.tiles_m + TL - 1) / TL) * g.tiles_n : ((g.tiles_n + TL - 1) / TL) * g.tiles_m;
  const int grid = 8 * g.nloc * ((split + g.xs - 1) / g.xs);
  UwuProfScope prof(st);
  if (part && split == 1) {
    hipLaunchKernelGGL((gemm_tr_kernel<FI, FJ, true, CONVW>), dim3(grid), dim3(256), T_NST * T_STAGE, st, g);
  } else if (part) {
    g.C2 = scratch;
    hipLaunchKernelGGL((gemm_tr_kernel<FI, FJ, true, CONVW>), dim3(grid), dim3(256), T_NST * T_STAGE, st, g);
    launch_splitk_reduce(static_cast<const float*>(scratch), static_cast<float*>(g.C), g.M, g.N, g.ldc, split, st);
  } else {
    hipLaunchKernelGGL((gemm_tr_kernel<FI, FJ, false, CONVW>), dim3(grid), dim3(256), T_NST * T_STAGE, st, g);
  }
  prof.done(CONVW ? UWU_PROF_CONV : UWU_PROF_GEMM_WGRAD, 0, 2.0 * g.M * g.N * g.K, ((double)g.M * g.K + (double)g.N * g.K) * 2 + (double)g.M * g.N * 4);
  UWU_LAUNCH_CHECK("gemm_tr");
  return UWU_OK;
}

// wide weight-gradient kernel: K slices in groups of 8 (one group per XCD), one workgroup per CU
int trw_split(int tiles, int steps) {
  int per_xcd = 32 / tiles;
  if (per_xcd < 1) per_xcd = 1;
  int split = 8 * per_xcd;
  while (split > 8 && split * 8 > steps) split -= 8;  // keep >= 8 K-steps per slice
  if (split > steps) split = steps;
  return split < 1 ? 1 : split;
}
// 0 = not taken, 1 = 192 x 384 tiles, 2 = 384 x 192 tiles.  UWU_GEMM_TRW=0 turns it off (A/B comparisons).
int pick_trw(const GemmArgs& g) {
  static UwuEnv on("UWU_GEMM_TRW");
  if (on.get().is('0')) return 0;
  // short reductions (per-GPU batch < 128 images): the 4-stage ring of a whole-LDS workgroup barely fills and nothing else fits
  // on its CU; the 256x128 kernel (two workgroups per CU) measured 1-2 % faster there.  UWU_GEMM_TRW=1 forces it (tests).
  const bool force = on.is('1');
  if (g.K % 32 || g.K < (force ? 4096 : 32768) || g.M % 8 || g.N % 8) return 0;
  if ((((uintptr_t)g.A | (uintptr_t)g.B) & 15) || g.lda % 8 || g.ldb % 8) return 0;
  if (g.N % 384 == 0 && g.M >= 192) return 1;
  if (g.M % 384 == 0 && g.N >= 192) return 2;
  return 0;
}
size_t trw_scratch_bytes(int M, int N, int K, int kind) {
  const int tiles = kind == 1 ? ((M + 191) / 192) * (N / 384) : (M / 384) * ((N + 191) / 192);
  return (size_t)trw_split(tiles, K / 32) * M * N * sizeof(float);
}
template <int WM, int WN, int FI, int FJ>
int launch_trw(GemmArgs g, void* scratch, hipStream_t st) {
  constexpr int NST = 4;
  auto kern = gemm_trw_kernel<WM, WN, FI, FJ, NST>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, NST * W_STAGE);
    attr_done = true;
  }
  g.tiles_m = (g.M + 16 * FI * WM - 1) / (16 * FI * WM);
  g.tiles_n = (g.N + 16 * FJ * WN - 1) / (16 * FJ * WN);
  const int tiles = g.tiles_m * g.tiles_n, steps = g.K / 32;
  int split = trw_split(tiles, steps);
  g.k_tiles_per_split = (steps + split - 1) / split;
  split = (steps + g.k_tiles_per_split - 1) / g.k_tiles_per_split;
  g.wide = split;
  g.C2 = scratch;
  const int grid = 8 * tiles * ((split + 7) / 8);
  UwuProfScope prof(st);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), NST * W_STAGE, st, g);
  launch_splitk_reduce(static_cast<const float*>(scratch), static_cast<float*>(g.C), g.M, g.N, g.ldc, split, st);
  prof.done(UWU_PROF_GEMM_WGRAD, 0, 2.0 * g.M * g.N * g.K, ((double)g.M * g.K + (double)g.N * g.K) * 2 + (double)g.M * g.N * 4);
  UWU_LAUNCH_CHECK("gemm_trw");
  return UWU_OK;
}
// K-major x K-major accumulate (the weight gradients): 0 = keep the 128x128 kernel, 1 = 256x128, 2 = 128x256
int pick_tr(const GemmArgs& g) {
  static UwuEnv on("UWU_GEMM_TR");  // "0": off (A/B comparisons)
  if (on.get().is('0')) return 0;
  if (g.K % 32 || g.K < 96 || g.M % 8 || g.N % 8 || g.M < 8 || g.N < 8) return 0;
  if ((((uintptr_t)g.A | (uintptr_t)g.B) & 15) || g.lda % 8 || g.ldb % 8) return 0;
  auto padded = [](int x, int b) { return (double)(((x + b - 1) / b) * b) / x; };
  const double w1 = padded(g.M, 256) * padded(g.N, 128), w2 = padded(g.M, 128) * padded(g.N, 256);
  const double w0 = padded(g.M, 128) * padded(g.N, 128);
  if ((w1 < w2 ? w1 : w2) > 1.35 * w0) return 0;  // too much padding: the small tile wastes less
  if (g.K < 2048) return 0;                       // short reductions: nothing to stream
  return w1 <= w2 ? 1 : 2;
}

// Ring kernel choice for bf16 operands with A K-contiguous: 0 = gemm_kernel, 8 = 256x128, 4 = 128x128.
// 256x128 pays off on the wide-N Linears (65536x1152x384: 84 us against 97; x1536: 108 against 132); with 768 tiles
// (N = 384) the second round of 512 workgroup slots would be half empty.  The 128x128 ring (4 stages) replaces
// gemm_kernel's register-staged input-gradient path (K-major weight: qkv dgrad 96 -> 70 us, fc1 dgrad 111 -> 90).
int pick_r3(const GemmArgs& g, bool tb) {
  static UwuEnv on("UWU_GEMM_R3"), t8_e("UWU_R3_T8"), t4_e("UWU_R3_T4");  // "0": off (A/B comparisons); thresholds: sweeps
  if (on.get().is('0')) return 0;
  if (g.K % 32 || g.K < 96) return 0;
  if ((((uintptr_t)g.A | (uintptr_t)g.B) & 15) || g.lda % 8 || g.ldb % 8) return 0;
  if (tb && (g.N % 8 || g.N < 8)) return 0;
  const int thr8 = t8_e.get().set ? t8_e.ival : 512;  // sweep at per-GPU batches 16..256: 512 / 128 never lose
  const int thr4 = t4_e.get().set ? t4_e.ival : 128;
  const int64_t t8 = (int64_t)((g.M + 255) / 256) * ((g.N + 127) / 128);
  // long contractions (the UNet's K = 640 .. 5120): the larger tile's operand reuse pays from one workgroup per CU on
  // (SDXL shape, 12 x 4x128x128: 480 tiles of 256x128 per 1280-wide Linear; 30.6 -> 31.8 images/s)
  const bool t8_env = t8_e.set;
  if (t8 >= ((g.K >= 640 && !t8_env) ? 256 : thr8)) return 8;
  const int64_t t4 = (int64_t)((g.M + 127) / 128) * ((g.N + 127) / 128);
  return (tb && t4 >= thr4) ? 4 : 0;  // K-contiguous B at N = 384: gemm_kernel's 128-byte rows measured faster (proj 35 vs 43 us)
}

bool no_glds() {
  static UwuEnv on("UWU_GEMM_NO_GLDS");
  return on.get().is('1');
}

template <typename T, typename TC>
int dispatch_trans(const GemmArgs& g, int ta, int tb, bool acc, int split, hipStream_t st) {
  if (acc) {
    if constexpr (sizeof(TC) == 4) {
      if constexpr (sizeof(T) == 2) {
        if (ta == 1 && tb == 1) {
          const int tr = split > 1 ? pick_tr(g) : 0;  // the streaming kernel chooses its own number of K slices
          GemmArgs gt = g;
          gt.bias = nullptr;  // (the fused bias gradient is uwu_gemm_wgrad's)
          if (tr == 1) return launch_tr<8, 4>(gt, nullptr, 0, st);
          if (tr == 2) return launch_tr<4, 8>(gt, nullptr, 0, st);
        }
      }
      if (ta == 1 && tb == 1) return launch<T, float, true, true, true>(g, split, st);
      if (ta == 0 && tb == 0) return launch<T, float, false, false, true>(g, split, st);
      if (ta == 0 && tb == 1) return launch<T, float, false, true, true>(g, split, st);
    }
    uwu_set_error("gemm: ACCUM epilogue needs fp32 C and (transA,transB) in {(0,0),(0,1),(1,1)}");
    return UWU_EINVAL;
  }
  constexpr bool hot = sizeof(T) == 2 && sizeof(TC) == 2;  // bf16 in / bf16 out: compile-time epilogues
  if (ta == 0 && tb == 0) {
    if constexpr (hot) {
      if (uwu_gemm_p8_ok(g, false)) return uwu_launch_gemm_p8(g, false, st);
      if (g.N % 256 && uwu_gemm_p8n_ok(g, false)) return uwu_launch_gemm_p8n(g, false, st);
    }
    if constexpr (sizeof(T) == 2) {
      const int r3 = pick_r3(g, false);
      if (r3 == 8) {
        if constexpr (hot) {
          if (g.epi == UWU_EPI_BIAS && use_as_bias() && use_as(g, sizeof(TC))) return launch_as<TC, UWU_EPI_BIAS>(g, st);
          if (use_wide(g)) {
            if (g.epi == UWU_EPI_NONE) return launch_wide<TC, UWU_EPI_NONE, false>(g, st);
            if (g.epi == UWU_EPI_BIAS) return launch_wide<TC, UWU_EPI_BIAS, false>(g, st);
          }
          if (g.epi == UWU_EPI_BIAS_GELU && use_as(g, sizeof(TC))) return launch_as<TC, UWU_EPI_BIAS_GELU>(g, st);
          if (g.epi == UWU_EPI_DGELU && g.C2 == nullptr && g.aux && g.ldaux % 8 == 0 && ((uintptr_t)g.aux & 15) == 0 &&
              use_as(g, sizeof(TC)))
            return launch_as<TC, UWU_EPI_DGELU>(g, st);
          if (use_big(g)) {
            if (g.epi == UWU_EPI_BIAS_GELU) return launch_big<TC, UWU_EPI_BIAS_GELU, false>(g, st);
            if (g.epi == UWU_EPI_NONE) return launch_big<TC, UWU_EPI_NONE, false>(g, st);
            if (g.epi == UWU_EPI_BIAS) return launch_big<TC, UWU_EPI_BIAS, false>(g, st);
          }
          if (g.epi == UWU_EPI_NONE) return launch_r3<TC, UWU_EPI_NONE, false, 8>(g, st);
          if (g.epi == UWU_EPI_BIAS) return launch_r3<TC, UWU_EPI_BIAS, false, 8>(g, st);
          if (g.epi == UWU_EPI_BIAS_GELU) return launch_r3<TC, UWU_EPI_BIAS_GELU, false, 8>(g, st);
        }
        return launch_r3<TC, -1, false, 8>(g, st);
      }
    }
    if constexpr (hot) {
      if (use_m64(g, false)) {
        if (g.epi == UWU_EPI_NONE) return launch_m64<TC, UWU_EPI_NONE, false>(g, st);
        if (g.epi == UWU_EPI_BIAS) return launch_m64<TC, UWU_EPI_BIAS, false>(g, st);
        if (g.epi == UWU_EPI_BIAS_GELU) return launch_m64<TC, UWU_EPI_BIAS_GELU, false>(g, st);
      }
    }
    if (g.K % GT<T>::BK == 0 && !no_glds()) {
      if constexpr (hot) {
        if (g.epi == UWU_EPI_NONE) return launch<T, TC, false, false, false, true, UWU_EPI_NONE>(g, split, st);
        if (g.epi == UWU_EPI_BIAS) return launch<T, TC, false, false, false, true, UWU_EPI_BIAS>(g, split, st);
        if (g.epi == UWU_EPI_BIAS_GELU) return launch<T, TC, false, false, false, true, UWU_EPI_BIAS_GELU>(g, split, st);
      }
      return launch<T, TC, false, false, false, true>(g, split, st);
    }
    return launch<T, TC, false, false, false>(g, split, st);
  }
  if (ta == 0 && tb == 1) {
    if constexpr (hot) {
      if (uwu_gemm_p8_ok(g, true)) return uwu_launch_gemm_p8(g, true, st);
      if (g.N % 256 && uwu_gemm_p8n_ok(g, true)) return uwu_launch_gemm_p8n(g, true, st);
      const int r3 = pick_r3(g, true);
      if (r3 == 8 && use_wide(g) && g.epi == UWU_EPI_NONE) return launch_wide<TC, UWU_EPI_NONE, true>(g, st);
      if (r3 == 8 && use_big(g)) {
        if (g.epi == UWU_EPI_DGELU) return launch_big<TC, UWU_EPI_DGELU, true>(g, st);
        if (g.epi == UWU_EPI_NONE) return launch_big<TC, UWU_EPI_NONE, true>(g, st);
      }
      if (r3 == 8 && g.epi == UWU_EPI_NONE) return launch_r3<TC, UWU_EPI_NONE, true, 8>(g, st);
      if (r3 == 8 && g.epi == UWU_EPI_DGELU) return launch_r3<TC, UWU_EPI_DGELU, true, 8>(g, st);
      if (r3 != 8 && use_m64(g, true)) {
        if (g.epi == UWU_EPI_NONE) return launch_m64<TC, UWU_EPI_NONE, true>(g, st);
        if (g.epi == UWU_EPI_DGELU) return launch_m64<TC, UWU_EPI_DGELU, true>(g, st);
      }
      if (r3 == 4 && g.epi == UWU_EPI_NONE) return launch_r3<TC, UWU_EPI_NONE, true, 4>(g, st);
      if (r3 == 4 && g.epi == UWU_EPI_DGELU) return launch_r3<TC, UWU_EPI_DGELU, true, 4>(g, st);
      if (g.epi == UWU_EPI_NONE) return launch<T, TC, false, true, false, false, UWU_EPI_NONE>(g, split, st);
      if (g.epi == UWU_EPI_DGELU) return launch<T, TC, false, true, false, false, UWU_EPI_DGELU>(g, split, st);
    }
    return launch<T, TC, false, true, false>(g, split, st);
  }
  if (ta == 1 && tb == 1) return launch<T, TC, true, true, false>(g, split, st);
  uwu_set_error("gemm: (transA=1, transB=0) is not instantiated");
  return UWU_EINVAL;
}

template <int EPI, int FA>
int launch_f8(GemmArgs g, const float* sa, const float* sb, hipStream_t st) {
  auto kern = gemm_f8_kernel<bf16_t, EPI, FA, false>;
  constexpr int LDS = 2 * 4 * TILE_BYTES;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  g.tiles_m = (g.M + 255) / 256;
  g.tiles_n = (g.N + 255) / 256;
  UwuProfScope prof(st);
  hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n), dim3(512), LDS, st, g, sa, sb);
  prof.done(gemm_tag(g, false, false), 0, 2.0 * g.M * g.N * g.K, gemm_bytes(g, 1, 2));
  UWU_LAUNCH_CHECK("gemm_f8");
  return UWU_OK;
}
template <int EPI, int FA>
int launch_f8_emit(GemmArgs g, const float* sa, const float* sb, hipStream_t st) {
  auto kern = gemm_f8_kernel<bf16_t, EPI, FA, false, true>;
  constexpr int LDS = F8_EMIT_LDS > 2 * 4 * TILE_BYTES ? F8_EMIT_LDS : 2 * 4 * TILE_BYTES;
  static unsigned char done[UWU_MAX_DEV];  // per device (141 KB of LDS: not every part has it)
  if (!uwu_func_lds(reinterpret_cast<const void*>(kern), LDS, done)) {
    uwu_set_error("gemm_f8(emit): the device cannot give a workgroup %d bytes of LDS", LDS);
    return UWU_ELAUNCH;
  }
  g.tiles_m = (g.M + 255) / 256;
  g.tiles_n = (g.N + 255) / 256;
  {
    static UwuEnv full("UWU_F8_EMIT_FULL");  // "0": the masked epilogue for full tiles too (A/B)
    g.p8_cont = full.get().is('0') ? 0 : 1;
  }
  UwuProfScope prof(st);
  hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n), dim3(512), LDS, st, g, sa, sb);
  // bytes: operands once, the bf16 output (if any), the dGELU aux, both fp8 images
  double by = (double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N * ((g.C ? 2 : 0) + (g.aux ? 2 : 0) + (g.q8 ? 1 : 0) + (g.q8t ? 1 : 0));
  prof.done(gemm_tag(g, false, false), 0, 2.0 * g.M * g.N * g.K, by);
  UWU_LAUNCH_CHECK("gemm_f8(emit)");
  return UWU_OK;
}
// number of K slices for an fp8 weight gradient: enough workgroups for ~2 rounds of the chip, >= 4 K-steps per slice
int f8_split(int tiles, int steps) {
  int split = (512 + tiles - 1) / tiles;
  split = (split + 7) / 8 * 8;
  while (split > 8 && split * 4 > steps) split -= 8;
  if (split > steps) split = steps;
  return split < 1 ? 1 : split;
}
template <int FA>
int launch_f8_part(GemmArgs g, const float* sa, const float* sb, void* scratch, hipStream_t st) {
  auto kern = gemm_f8_kernel<float, UWU_EPI_NONE, FA, true>;
  constexpr int LDS = 2 * 4 * TILE_BYTES;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  g.tiles_m = (g.M + 255) / 256;
  g.tiles_n = (g.N + 255) / 256;
  const int tiles = g.tiles_m * g.tiles_n, steps = g.K / 128;
  int split = f8_split(tiles, steps);
  g.k_tiles_per_split = (steps + split - 1) / split;
  split = (steps + g.k_tiles_per_split - 1) / g.k_tiles_per_split;
  g.wide = split;
  g.C2 = scratch;
  const int grid = 8 * tiles * ((split + 7) / 8);
  UwuProfScope prof(st);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, st, g, sa, sb);
  launch_splitk_reduce(static_cast<const float*>(scratch), static_cast<float*>(g.C), g.M, g.N, g.ldc, split, st);
  prof.done(UWU_PROF_GEMM_WGRAD, 0, 2.0 * g.M * g.N * g.K, ((double)g.M * g.K + (double)g.N * g.K) + (double)g.M * g.N * 4);
  UWU_LAUNCH_CHECK("gemm_f8(split-K)");
  return UWU_OK;
}

}  // namespace

// ---- implicit-GEMM 3x3 convolution (padding 1, stride 1 / 2, channels-last bf16): no im2col matrix ------------------------
__device__ uint4 g_conv_zero[4];  // zero page for padded pixels (zero-initialised device memory)

static const void* conv_zero_page() {
  static void* p = nullptr;
  if (!p && hipGetSymbolAddress(&p, HIP_SYMBOL(g_conv_zero)) != hipSuccess) p = nullptr;
  return p;
}

extern "C" int uwu_conv3x3_implicit_ok(int B, int H, int W, int C, int Cout, int stride, int dtype) {
  if (dtype != UWU_BF16 || C % 32 || Cout % 32 || C < 32 || Cout < 32 || (stride != 1 && stride != 2)) return 0;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const int64_t Mi = (int64_t)B * H * W, Mo = (int64_t)B * Ho * Wo;
  if (Mi >= (1 << 24) || Mo >= (1 << 24) || Mo % 32 || 9 * (int64_t)C >= (1 << 24)) return 0;
  return 1;
}

static int conv_args(GemmArgs& g, int B, int H, int W, int C, int stride) {
  g.cH = H; g.cW = W; g.cC = C; g.cS = stride;
  g.cHo = (H - 1) / stride + 1;
  g.cWo = (W - 1) / stride + 1;
  g.zero = conv_zero_page();
  if (!g.zero) { uwu_set_error("conv3x3: zero page unavailable"); return UWU_ELAUNCH; }
  return UWU_OK;
}

// y[(b,oy,ox), co] = sum x[b, oy s + ky - 1, ox s + kx - 1, c] w[co][ky][kx][c] + bias[co]
extern "C" int uwu_conv3x3_fwd(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int C, int Cout,
                               int stride, int dtype, void* stream) {
  UWU_CHECK_ARG(x && w && y && B > 0 && H > 0 && W > 0, "conv3x3_fwd: bad argument");
  UWU_CHECK_ARG(uwu_conv3x3_implicit_ok(B, H, W, C, Cout, stride, dtype), "conv3x3_fwd: shape not covered by the implicit-GEMM kernel (C=%d Cout=%d)", C, Cout);
  UWU_CHECK_ARG((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0), "conv3x3_fwd: misaligned tensor");
  GemmArgs g{};
  RETURN_IF(conv_args(g, B, H, W, C, stride));
  g.A = x; g.B = w; g.C = y; g.bias = bias;
  g.M = B * g.cHo * g.cWo; g.N = Cout; g.K = 9 * C; g.lda = C; g.ldb = 9 * C; g.ldc = Cout;
  g.epi = bias ? UWU_EPI_BIAS : UWU_EPI_NONE;
  g.wide = (Cout % 8 == 0) ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  if (bias) return launch_r3<bf16_t, UWU_EPI_BIAS, false, 8, 1>(g, st);
  return launch_r3<bf16_t, UWU_EPI_NONE, false, 8, 1>(g, st);
}

// dx[(b,iy,ix), c] = sum dy[b, (iy + 1 - ky) / s, (ix + 1 - kx) / s, co] w[co][ky][kx][c]   (positions that divide evenly)
extern "C" int uwu_conv3x3_dgrad(const void* dy, const void* w, void* dx, int B, int H, int W, int C, int Cout, int stride,
                                 int dtype, void* stream) {
  UWU_CHECK_ARG(dy && w && dx && B > 0 && H > 0 && W > 0, "conv3x3_dgrad: bad argument");
  UWU_CHECK_ARG(uwu_conv3x3_implicit_ok(B, H, W, C, Cout, stride, dtype), "conv3x3_dgrad: shape not covered by the implicit-GEMM kernel");
  UWU_CHECK_ARG((((uintptr_t)dy | (uintptr_t)w | (uintptr_t)dx) & 15) == 0, "conv3x3_dgrad: misaligned tensor");
  GemmArgs g{};
  RETURN_IF(conv_args(g, B, H, W, C, stride));
  g.A = dy; g.B = w; g.C = dx;
  g.M = B * H * W; g.N = C; g.K = 9 * Cout; g.lda = Cout; g.ldb = 9 * C; g.ldc = C;
  g.epi = UWU_EPI_NONE;
  g.wide = (C % 8 == 0) ? 1 : 0;
  return launch_r3<bf16_t, UWU_EPI_NONE, true, 8, 2>(g, (hipStream_t)stream);
}

extern "C" size_t uwu_conv3x3_wgrad_scratch_bytes(int C, int Cout, int64_t Mo) {
  return uwu_gemm_wgrad_scratch_bytes(Cout, 9 * C, (int)Mo);
}
// dw[co][ky][kx][c] += sum dy[(b,oy,ox), co] x[b, oy s + ky - 1, ox s + kx - 1, c];  db[co] += sum dy
extern "C" int uwu_conv3x3_wgrad(const void* dy, const void* x, float* dw, float* db, int B, int H, int W, int C, int Cout,
                                 int stride, int dtype, void* scratch, size_t scratch_bytes, void* stream) {
  UWU_CHECK_ARG(dy && x && dw && B > 0 && H > 0 && W > 0, "conv3x3_wgrad: bad argument");
  UWU_CHECK_ARG(uwu_conv3x3_implicit_ok(B, H, W, C, Cout, stride, dtype), "conv3x3_wgrad: shape not covered by the implicit-GEMM kernel");
  UWU_CHECK_ARG((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dw) & 15) == 0, "conv3x3_wgrad: misaligned tensor");
  GemmArgs g{};
  RETURN_IF(conv_args(g, B, H, W, C, stride));
  g.A = dy; g.B = x; g.C = dw; g.bias = db;
  g.M = Cout; g.N = 9 * C; g.K = B * g.cHo * g.cWo; g.lda = Cout; g.ldb = C; g.ldc = 9 * C;
  g.epi = UWU_EPI_ACCUM;
  auto padded = [](int v, int b) { return (double)(((v + b - 1) / b) * b) / v; };
  const bool tall = padded(g.M, 256) * padded(g.N, 128) <= padded(g.M, 128) * padded(g.N, 256);
  hipStream_t st = (hipStream_t)stream;
  if (tall) return launch_tr<8, 4, true>(g, scratch, scratch_bytes, st);
  return launch_tr<4, 8, true>(g, scratch, scratch_bytes, st);
}

extern "C" size_t uwu_gemm_fp8_scratch_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K < 128) return 0;
  const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
  const int a = f8_split(tiles, K / 128), b = K % 128 ? 0 : uwu_gemm_p8f_split(tiles, K / 128);  // (either kernel may take it)
  return (size_t)(a > b ? a : b) * M * N * sizeof(float);
}

extern "C" int uwu_gemm_fp8(const void* A, const void* B, void* C, void* C2, const float* bias, const void* aux, int M,
                            int N, int K, int lda, int ldb, int ldc, int ldaux, int fmt_a, int epilogue,
                            const float* scale_a, const float* scale_b, void* scratch, size_t scratch_bytes,
                            void* stream) {
  UWU_CHECK_ARG(A && B && C && scale_a && scale_b, "gemm_fp8: null operand");
  UWU_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 128 == 0, "gemm_fp8: K=%d must be a positive multiple of 128", K);
  UWU_CHECK_ARG(fmt_a == UWU_FP8_E4M3 || fmt_a == UWU_FP8_E5M2, "gemm_fp8: bad operand format %d", fmt_a);
  UWU_CHECK_ARG((((uintptr_t)A | (uintptr_t)B) & 15) == 0 && lda % 16 == 0 && ldb % 16 == 0 && lda >= K && ldb >= K,
                "gemm_fp8: operands must be 16-byte aligned with leading dimensions that are multiples of 16");
  GemmArgs g{};
  g.A = A; g.B = B; g.C = C; g.C2 = C2; g.bias = bias; g.aux = aux;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = ldaux; g.epi = epilogue;
  hipStream_t st = (hipStream_t)stream;
  if (epilogue == UWU_EPI_ACCUM) {  // C fp32 += (weight gradient): split-K partial sums in `scratch`
    UWU_CHECK_ARG(N % 4 == 0 && ldc % 4 == 0 && ldc >= N && ((uintptr_t)C & 15) == 0, "gemm_fp8: ACCUM needs 16-byte rows in C");
    UWU_CHECK_ARG(scratch && ((uintptr_t)scratch & 15) == 0 && scratch_bytes >= uwu_gemm_fp8_scratch_bytes(M, N, K),
                  "gemm_fp8: ACCUM needs uwu_gemm_fp8_scratch_bytes(M, N, K) of scratch");
    if (uwu_gemm_p8f_part_ok(g)) {  // the 8-phase kernel over (K slice, tile) units, then the same reduce
      UwuProfScope prof(stream);
      RETURN_IF(uwu_launch_gemm_p8f_part(g, fmt_a == UWU_FP8_E5M2, scale_a, scale_b, scratch, st));
      launch_splitk_reduce(static_cast<const float*>(scratch), static_cast<float*>(C), M, N, ldc, g.wide, st);
      prof.done(UWU_PROF_GEMM_WGRAD, 0, 2.0 * M * N * K, ((double)M * K + (double)N * K) + (double)M * N * 4);
      UWU_LAUNCH_CHECK("gemm_p8f(split-K)");
      return UWU_OK;
    }
    return fmt_a == UWU_FP8_E5M2 ? launch_f8_part<1>(g, scale_a, scale_b, scratch, st)
                                 : launch_f8_part<0>(g, scale_a, scale_b, scratch, st);
  }
  UWU_CHECK_ARG(N % 8 == 0 && ldc % 8 == 0 && ldc >= N && ((uintptr_t)C & 15) == 0, "gemm_fp8: N and ldc must be multiples of 8");
  g.wide = 1;
  if (epilogue == UWU_EPI_BIAS || epilogue == UWU_EPI_BIAS_GELU)
    UWU_CHECK_ARG(bias && ((uintptr_t)bias & 15) == 0, "gemm_fp8: bias missing/misaligned");
  if (epilogue == UWU_EPI_BIAS_GELU) UWU_CHECK_ARG(C2 && ((uintptr_t)C2 & 15) == 0, "gemm_fp8: C2 missing/misaligned");
  if (epilogue == UWU_EPI_DGELU)
    UWU_CHECK_ARG(aux && ldaux % 4 == 0 && ldaux >= N && ((uintptr_t)aux & 7) == 0, "gemm_fp8: aux missing/misaligned");
  if (uwu_gemm_p8f_ok(g)) {
    UwuProfScope prof(stream);
    RETURN_IF(uwu_launch_gemm_p8f(g, fmt_a == UWU_FP8_E5M2, scale_a, scale_b, st));
    prof.done(gemm_tag(g, false, false), 0, 2.0 * M * N * K, gemm_bytes(g, 1, 2));
    return UWU_OK;
  }
#define F8_CASE(E)                                                                   \
  case E:                                                                            \
    return fmt_a == UWU_FP8_E5M2 ? launch_f8<E, 1>(g, scale_a, scale_b, st) : launch_f8<E, 0>(g, scale_a, scale_b, st);
  switch (epilogue) {
    F8_CASE(UWU_EPI_NONE) F8_CASE(UWU_EPI_BIAS) F8_CASE(UWU_EPI_BIAS_GELU) F8_CASE(UWU_EPI_DGELU)
  }
#undef F8_CASE
  uwu_set_error("gemm_fp8: epilogue %d not available", epilogue);
  return UWU_EINVAL;
}

extern "C" int uwu_gemm_fp8_emit(const void* A, const void* B, void* C, float* colsum, const float* bias, const void* aux,
                                 int M, int N, int K, int lda, int ldb, int ldc, int ldaux, int fmt_a, int epilogue,
                                 const float* scale_a, const float* scale_b, void* q8, int ldq, void* q8t, int ldqt,
                                 const float* q_scale, float* q_amax, void* stream) {
  UWU_CHECK_ARG(A && B && scale_a && scale_b && q_scale && (q8 || q8t), "gemm_fp8_emit: null operand");
  UWU_CHECK_ARG(epilogue == UWU_EPI_BIAS_GELU || epilogue == UWU_EPI_DGELU, "gemm_fp8_emit: epilogue %d not available", epilogue);
  UWU_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 128 == 0, "gemm_fp8_emit: K=%d must be a positive multiple of 128", K);
  UWU_CHECK_ARG(M % 16 == 0 && N % 16 == 0, "gemm_fp8_emit: M=%d and N=%d must be multiples of 16", M, N);
  UWU_CHECK_ARG(fmt_a == UWU_FP8_E4M3 || fmt_a == UWU_FP8_E5M2, "gemm_fp8_emit: bad operand format %d", fmt_a);
  UWU_CHECK_ARG((((uintptr_t)A | (uintptr_t)B) & 15) == 0 && lda % 16 == 0 && ldb % 16 == 0 && lda >= K && ldb >= K,
                "gemm_fp8_emit: operands must be 16-byte aligned with leading dimensions that are multiples of 16");
  UWU_CHECK_ARG(!C || (ldc % 8 == 0 && ldc >= N && ((uintptr_t)C & 15) == 0), "gemm_fp8_emit: C / ldc misaligned");
  UWU_CHECK_ARG(!q8 || (ldq % 16 == 0 && ldq >= N && ((uintptr_t)q8 & 15) == 0), "gemm_fp8_emit: q8 / ldq misaligned");
  UWU_CHECK_ARG(!q8t || (ldqt % 16 == 0 && ldqt >= M && ((uintptr_t)q8t & 15) == 0), "gemm_fp8_emit: q8t / ldqt misaligned");
  GemmArgs g{};
  g.A = A; g.B = B; g.C = C; g.C2 = colsum; g.bias = bias; g.aux = aux;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = ldaux; g.epi = epilogue;
  g.q8 = q8; g.q8t = q8t; g.q_scale = q_scale; g.q_amax = q_amax; g.ldq = ldq; g.ldqt = ldqt;
  g.wide = 1;
  hipStream_t st = (hipStream_t)stream;
  if (epilogue == UWU_EPI_BIAS_GELU) {
    UWU_CHECK_ARG(C && bias && ((uintptr_t)bias & 15) == 0 && !colsum, "gemm_fp8_emit: BIAS_GELU needs C (the pre-activation) and bias");
    return fmt_a == UWU_FP8_E5M2 ? launch_f8_emit<UWU_EPI_BIAS_GELU, 1>(g, scale_a, scale_b, st)
                                 : launch_f8_emit<UWU_EPI_BIAS_GELU, 0>(g, scale_a, scale_b, st);
  }
  UWU_CHECK_ARG(aux && ldaux % 4 == 0 && ldaux >= N && ((uintptr_t)aux & 7) == 0, "gemm_fp8_emit: aux missing/misaligned");
  UWU_CHECK_ARG(!C, "gemm_fp8_emit: DGELU emits fp8 only (C must be NULL)");
  return fmt_a == UWU_FP8_E5M2 ? launch_f8_emit<UWU_EPI_DGELU, 1>(g, scale_a, scale_b, st)
                               : launch_f8_emit<UWU_EPI_DGELU, 0>(g, scale_a, scale_b, st);
}

extern "C" int uwu_gemm(const void* A, const void* B, void* C, void* C2, const float* bias, const void* aux, int M,
                        int N, int K, int lda, int ldb, int ldc, int ldaux, int transA, int transB, int dtype,
                        int c_dtype, int epilogue, int split_k, void* stream) {
  UWU_CHECK_ARG(A && B && C, "gemm: null operand");
  UWU_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
  UWU_CHECK_ARG(dtype == UWU_F32 || dtype == UWU_BF16, "gemm: bad dtype %d", dtype);
  UWU_CHECK_ARG(c_dtype == UWU_F32 || c_dtype == dtype, "gemm: c_dtype must be fp32 or the operand dtype");
  const int epc = dtype == UWU_BF16 ? 8 : 4;
  const int bk = dtype == UWU_BF16 ? 64 : 32;
  // 16-byte vector access rules (checked on the host so a bad shape can never fault on the device)
  UWU_CHECK_ARG((((uintptr_t)A | (uintptr_t)B) & 15) == 0, "gemm: A/B must be 16-byte aligned");
  UWU_CHECK_ARG(lda % epc == 0 && ldb % epc == 0, "gemm: lda/ldb must be multiples of %d", epc);
  UWU_CHECK_ARG(transA ? (M % epc == 0) : (K % epc == 0), "gemm: A contiguous extent must be a multiple of %d", epc);
  UWU_CHECK_ARG(transB ? (N % epc == 0) : (K % epc == 0), "gemm: B contiguous extent must be a multiple of %d", epc);
  UWU_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? N : K), "gemm: leading dimension too small");
  UWU_CHECK_ARG(epilogue >= UWU_EPI_NONE && epilogue <= UWU_EPI_ACCUM, "gemm: bad epilogue %d", epilogue);
  const bool acc = epilogue == UWU_EPI_ACCUM;
  if (!acc) {
    UWU_CHECK_ARG(split_k <= 1, "gemm: split_k needs UWU_EPI_ACCUM");
    UWU_CHECK_ARG(N % 4 == 0 && ldc % 4 == 0 && ldc >= N, "gemm: N and ldc must be multiples of 4");
    const int cal = c_dtype == UWU_BF16 ? 7 : 15;
    UWU_CHECK_ARG(((uintptr_t)C & cal) == 0, "gemm: C misaligned");
    if (epilogue == UWU_EPI_BIAS || epilogue == UWU_EPI_BIAS_GELU || epilogue == UWU_EPI_BIAS_SILU)
      UWU_CHECK_ARG(bias && ((uintptr_t)bias & 15) == 0, "gemm: bias missing/misaligned");
    if (epilogue == UWU_EPI_BIAS_GELU || epilogue == UWU_EPI_BIAS_SILU)
      UWU_CHECK_ARG(C2 && ((uintptr_t)C2 & cal) == 0, "gemm: C2 missing/misaligned");
    if (epilogue == UWU_EPI_DGELU)
      UWU_CHECK_ARG(aux && ldaux % 4 == 0 && ldaux >= N && ((uintptr_t)aux & (dtype == UWU_BF16 ? 7 : 15)) == 0,
                    "gemm: aux missing/misaligned");
  } else {
    UWU_CHECK_ARG(c_dtype == UWU_F32 && ldc >= N, "gemm: ACCUM needs fp32 C");
  }
  GemmArgs g{};
  g.A = A; g.B = B; g.C = C; g.C2 = C2; g.bias = bias; g.aux = aux;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = ldaux; g.epi = epilogue;
  g.tiles_m = (M + BM - 1) / BM;
  g.tiles_n = (N + BN - 1) / BN;
  // 16-byte epilogue stores need 8-column granularity and 16-byte aligned rows
  g.wide = (!acc && c_dtype == UWU_BF16 && N % 8 == 0 && ldc % 8 == 0 && ((uintptr_t)C & 15) == 0 &&
            (C2 == nullptr || epilogue == UWU_EPI_DGELU || ((uintptr_t)C2 & 15) == 0)) ? 1 : 0;
  {
    static UwuEnv ntc("UWU_GEMM_NT_C");
    g.nt_c = ntc.get().set ? ntc.ival : 0;
  }
  g.aux16 = (epilogue == UWU_EPI_DGELU && dtype == UWU_BF16 && N % 8 == 0 && ldaux % 8 == 0 && ((uintptr_t)aux & 15) == 0) ? 1 : 0;
  {
    static UwuEnv a16("UWU_GEMM_AUX16");  // "0": the 8-byte aux loads (A/B comparisons)
    if (a16.get().is('0')) g.aux16 = 0;
  }

  const int ktiles = (K + bk - 1) / bk;
  int split = split_k < 1 ? 1 : split_k;
  if (split > ktiles) split = ktiles;
  g.k_tiles_per_split = (ktiles + split - 1) / split;
  split = (ktiles + g.k_tiles_per_split - 1) / g.k_tiles_per_split;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UWU_BF16) {
    if (c_dtype == UWU_BF16) return dispatch_trans<bf16_t, bf16_t>(g, transA, transB, acc, split, st);
    return dispatch_trans<bf16_t, float>(g, transA, transB, acc, split, st);
  }
  return dispatch_trans<float, float>(g, transA, transB, acc, split, st);
}

// Weight gradient with caller-provided split-K scratch: C[M,N] (fp32) += A[K,M]^T . B[K,N], operands K-major.
// When `scratch` holds uwu_gemm_wgrad_scratch_bytes(M, N, K) the split-K slices of the streaming kernel are written
// there and reduced by a second kernel; otherwise its slices are accumulated with atomics.  Shapes the streaming
// kernel does not take go to uwu_gemm(..., UWU_EPI_ACCUM) with `blocks` workgroups as the split-K target.
extern "C" size_t uwu_gemm_wgrad_scratch_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K < 32) return 0;
  // the larger of the two tile orientations' slice counts (pick_tr chooses by padding)
  const int t1 = ((M + 255) / 256) * ((N + 127) / 128), t2 = ((M + 127) / 128) * ((N + 255) / 256);
  const int s1 = tr_split(t1, K / 32), s2 = tr_split(t2, K / 32);
  size_t b = (size_t)(s1 > s2 ? s1 : s2) * M * N * sizeof(float);
  if (N % 384 == 0 && trw_scratch_bytes(M, N, K, 1) > b) b = trw_scratch_bytes(M, N, K, 1);
  if (M % 384 == 0 && trw_scratch_bytes(M, N, K, 2) > b) b = trw_scratch_bytes(M, N, K, 2);
  return b;
}

extern "C" int uwu_gemm_wgrad(const void* A, const void* B, float* C, float* bias_grad, int M, int N, int K, int lda,
                              int ldb, int ldc, int dtype, int blocks, void* scratch, size_t scratch_bytes,
                              void* stream) {
  UWU_CHECK_ARG(A && B && C, "gemm_wgrad: null operand");
  UWU_CHECK_ARG(M > 0 && N > 0 && K > 0 && blocks > 0, "gemm_wgrad: bad shape M=%d N=%d K=%d blocks=%d", M, N, K, blocks);
  UWU_CHECK_ARG(dtype == UWU_F32 || dtype == UWU_BF16, "gemm_wgrad: bad dtype %d", dtype);
  UWU_CHECK_ARG(lda >= M && ldb >= N && ldc >= N, "gemm_wgrad: leading dimension too small");
  if (dtype == UWU_BF16) {
    GemmArgs g{};
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.epi = UWU_EPI_ACCUM;
    g.bias = bias_grad;
    const int trw = pick_trw(g);
    if (trw && scratch && (((uintptr_t)C | (uintptr_t)scratch) & 15) == 0 && ldc % 4 == 0 &&
        scratch_bytes >= trw_scratch_bytes(M, N, K, trw)) {
      if (trw == 1) return launch_trw<2, 4, 6, 6>(g, scratch, (hipStream_t)stream);
      return launch_trw<4, 2, 6, 6>(g, scratch, (hipStream_t)stream);
    }
    const int tr = pick_tr(g);
    if (tr == 1) return launch_tr<8, 4>(g, scratch, scratch_bytes, (hipStream_t)stream);
    if (tr == 2) return launch_tr<4, 8>(g, scratch, scratch_bytes, (hipStream_t)stream);
  }
  if (bias_grad) {
    const int rc = uwu_colsum(A, dtype, K, M, lda, bias_grad, 1, stream);
    if (rc != UWU_OK) return rc;
  }
  const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  const int bk = dtype == UWU_BF16 ? 64 : 32;
  int split = (blocks + tiles - 1) / tiles;
  // a slice that adds a whole fp32 tile with atomics has to amortise them over >= 8 K steps (the cross-attention
  // key / value weights see K = B x 77 tokens: 5 slices of 1-2 steps each took 119 us, one slice of 8 takes 15)
  const int ksteps = (K + bk - 1) / bk;
  if (split > ksteps / 8) split = ksteps / 8;
  if (split < 1) split = 1;
  return uwu_gemm(A, B, C, nullptr, nullptr, nullptr, M, N, K, lda, ldb, ldc, 0, 1, 1, dtype, UWU_F32, UWU_EPI_ACCUM,
                  split, stream);
}

// (the profiler entry points live in prof.cpp: uwu_prof_enable / uwu_prof_collect; uwu_gemm_prof_* wrap them)
