// Self-attention backward for T = 256 tokens, head dim 64 (every DiT shape at 256^2 latents): PERSISTENT workgroups that
// stream heads through LDS-DMA rings.  Replaces F.scaled_dot_product_attention's backward (reference
// src/duwu/modules/rope_unet.py:151-153) for that shape; the general kernels stay in attention_mfma.hip.
//
// Why a second kernel (measured on the one in attention_mfma.hip, B x H = 4608 heads: 452 us = 24.5 us per head and CU for
// 4.3 us of MFMA work and 10.9 us of HBM time at the CU's share of 6 TB/s): one 8-wave workgroup per CU and head, operands
// staged through registers one tile ahead -- every head pays its prologue (K^T image, K / V fragments, first tile: two dependent
// HBM latencies), every tile a load latency the short compute phase cannot cover, and the workgroup's exit waits for the
// dK / dV store drain.  249 VGPRs leave no room to prefetch further through registers, and two workgroups per CU do not fit
// (128 KB of fp32 dK / dV accumulators per head).
// Here a workgroup owns its CU for the whole launch and walks heads h = blockIdx.x, + gridDim.x, ...:
//   * Q / dO tiles (64 query rows, 16 KB) arrive by LDS-DMA (global_load_lds_dwordx4: no VGPRs) into a ring of three slots,
//     two tiles ahead and ACROSS head boundaries; the next head's K image (for dQ) and lse row arrive the same way, its K / V
//     fragments are loaded straight into the fragment registers as soon as the last phase 1 of the current head has read them;
//   * ONE row-major image per tile serves both uses of Q and dO: row reads (ds_read_b128) for S = Q.K^T, dP = dO.V^T and
//     transposing reads (ds_read_b64_tr_b16) for dV^T += dO^T.P, dK^T += Q^T.dS -- no transposed copies, no staging VALU;
//   * dS crosses LDS once, as a [key][q] image written with 8-byte stores (the accumulator holds 4 consecutive q per key) and
//     read back with transposing reads as the B operand of dQ^T = K^T.dS^T; K^T comes from the row-major K image the same way;
//   * dK / dV of a finished head are written while the next head's first tile is already in LDS; dq leaves as one 16-byte
//     store per lane (v_permlane16_swap pairs two 4-column groups).
// Image swizzles (16-byte chunks of 128-byte rows, 8-byte slots for dS^T) make every one of these accesses bank-conflict
// free; tools/model_attn_lds.py checks the lane maps and the bank model on the host.
// Vector-memory operations retire in issue order (loads, stores and LDS-DMA alike), so the only hand-counted wait is
// `s_waitcnt vmcnt(1)` at the top of a tile: everything but the previous tile's dq store has landed.
#include "common.h"

namespace {

constexpr int P_RING = 3;
// LDS map of the backward for head dim DH (64, or 72 = 64 + an 8-column TAIL kept in compact 16-byte-row images):
//   ring of P_RING slots: Q tile (8 KB) | dO tile (8 KB) [| tails [8 waves][Q 8 rows x 16 B | dO 8 rows x 16 B] | lse[64] of the tile]
//   dS^T image [256 keys][64 q] bf16;  2 x K image [256 keys][64 d] [+ K tail [256 keys][8 d]] (head parity);
//   (DH = 64) 2 x lse[256] (head parity);  2 x -delta[64] (tile parity)
template <int DH>
struct PGeom {
  static constexpr bool TAIL = DH > 64;
  static constexpr int SLOT = 16384 + (TAIL ? 2048 + 256 : 0);
  static constexpr int OFF_TT = 16384, OFF_TL = 16384 + 2048;   // inside a slot: tails, lse piece
  static constexpr int OFF_DS = P_RING * SLOT;
  static constexpr int KIMG = 32768 + (TAIL ? 4096 : 0);
  static constexpr int OFF_K = OFF_DS + 32768;
  static constexpr int OFF_LSE = OFF_K + 2 * KIMG;
  static constexpr int OFF_DELTA = OFF_LSE + (TAIL ? 0 : 2 * 1024);
  static constexpr int OFF_ZERO = OFF_DELTA + 2 * 256;          // TAIL: 16 bytes of zeros
  static constexpr int LDS = OFF_ZERO + (TAIL ? 16 : 0);        // 150016 B (DH = 64), 163088 B (DH = 72)
};
static_assert(PGeom<64>::LDS == 150016 && PGeom<72>::LDS <= 163840, "LDS budget");
constexpr int P_LDS = PGeom<64>::LDS;

struct PArgs {
  const bf16_t *q, *k, *v, *o, *dO;
  bf16_t *dq, *dk, *dv;
  const float* lse;
  int B, H, ldq, ldk, ldv, ldo, nheads;
  float scale;
  int prio;  // UWU_P256_PRIO (A/B): 1 = waves 4-7 run at s_setprio 1, 2 = waves 0-3
};

// 16-byte chunk swizzle of the 128-byte-row images (Q / dO tiles, K image): bit0 = r2 ^ r4, bit1 = r3 ^ r4, bit2 = r1
__device__ __forceinline__ int fsw(int row) {
  return (((row >> 2) ^ (row >> 4)) & 1) | ((((row >> 3) ^ (row >> 4)) & 1) << 1) | (((row >> 1) & 1) << 2);
}
// 8-byte slot swizzle of the dS^T image: bit0 = r0, bit1 = r2, bit2 = r3, bit3 = r1
__device__ __forceinline__ int Fsw(int row) {
  return (row & 1) | (((row >> 2) & 1) << 1) | (((row >> 3) & 1) << 2) | (((row >> 1) & 1) << 3);
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned su32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst /* wave-uniform */) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ void glds4(const void* gsrc, unsigned lds_dst /* wave-uniform */) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
// the same with the address split into a wave-uniform base (SGPR pair) and a 32-bit per-lane byte offset: the per-tile address
// arithmetic shrinks to one v_add_u32 (the 64-bit per-lane pointers cost ~25 VALU + ~80 SALU instructions per tile)
__device__ __forceinline__ void glds16_s(const void* sbase /* wave-uniform */, unsigned voff, unsigned lds_dst /* wave-uniform */) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ f32x16 mfma32(const uint4& a, const uint4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0,
                                                 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0,
                                                 0, 0);
}
__device__ __forceinline__ uint4 pack8(const f32x16& x, int s) {
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (bf16_t)x[8 * s + j];
  return *reinterpret_cast<uint4*>(&f);
}

extern __shared__ __attribute__((aligned(16))) char p_smem[];

// lo = x of lane (l & 31), hi = x of lane (l & 31) + 32, in every lane: one v_permlane32_swap (lanes 32-63 of its first operand
// trade places with lanes 0-31 of its second) instead of a trip through the LDS crossbar (__shfl_xor).  Inline asm: handed the
// same value twice, the BUILTIN's two results came back folded into one register (hipcc, ROCm 7.2: max(x', x') of one half
// only -- caught by the bit-identity test).  The two v_nop are the wait states between a VALU write and the permlane read.
__device__ __forceinline__ void lane_halves(float x, float& lo, float& hi) {
  float a = x, b = x;
  asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  lo = a;
  hi = b;
}

// two transposing reads (rows +0..3 of a block and the four rows behind a1) -> one 8-element MFMA operand fragment
__device__ __forceinline__ uint4 tr_pair(unsigned a0, unsigned a1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) char*)p_smem + a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) char*)p_smem + a1));
  uint4 r;
  r.x = ((unsigned)(unsigned short)lo[0]) | ((unsigned)(unsigned short)lo[1] << 16);
  r.y = ((unsigned)(unsigned short)lo[2]) | ((unsigned)(unsigned short)lo[3] << 16);
  r.z = ((unsigned)(unsigned short)hi[0]) | ((unsigned)(unsigned short)hi[1] << 16);
  r.w = ((unsigned)(unsigned short)hi[2]) | ((unsigned)(unsigned short)hi[3] << 16);
  return r;
}

// ABL (tools only, UWU_P256_ABL): timing-only builds with wrong results -- 1: no dQ product, 2: no global stores, 4: no phase 1
// (dS^T image left stale)
// DH = 72 (DiT-XL/2's 16 heads of 1152): columns 64..71 of every operand travel as TAILS -- compact images with 16-byte rows next
// to the 128-byte-row images of columns 0..63.  As a contraction index (S = Q.K^T, dP = dO.V^T) the tail is a fifth k step whose
// upper half is zero in the K / V fragment registers; as an output index (dV^T, dK^T, dQ^T rows 64..71) it is a third / fifth row
// tile whose rows past 71 hold garbage nobody stores (an MFMA row only ever sees its own row of A).
template <int ABL, int DH = 64>
__global__ void __launch_bounds__(512, 2) attn_bwd_p256(const PArgs a) {
  using G = PGeom<DH>;
  constexpr bool TAIL = G::TAIL;
  constexpr int T = 256, NKS = TAIL ? 5 : 4, NDT = TAIL ? 3 : 2;
  constexpr int P_SLOT = G::SLOT, P_OFF_DS = G::OFF_DS, P_OFF_K = G::OFF_K, P_OFF_LSE = G::OFF_LSE, P_OFF_DELTA = G::OFF_DELTA;
  constexpr int KIMG = G::KIMG;
  static_assert(!TAIL || ABL == 0, "the timing-only ablations exist for head dim 64");
  char* const smem = p_smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int fr = lane & 15, fq = lane >> 4;
  const unsigned smem_base = (unsigned)(size_t)((const __attribute__((address_space(3))) char*)p_smem);
  const float c = a.scale * 1.4426950408889634f;
  const float ninv_scale = -1.f / a.scale;
  const int nmy = (a.nheads - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;  // heads of this workgroup
  const int total = 4 * nmy;                                                             // its tiles
  const int k0 = wave * 32;  // this wave's keys

  // per-lane pieces of the LDS addresses (tools/model_attn_lds.py derives and checks them)
  const int drow = tid >> 3;                             // DMA / delta: row of the 64-row tile this thread moves
  const int dchunk = (tid & 7) ^ fsw(drow);              // ... and the logical 16-byte chunk that lands at LDS byte 16 * tid
  // (TAIL: the head-dim-72 instance has no registers to park these in -- each phase rebuilds its own from a lane id the
  // compiler cannot see through, a dozen VALU instructions per tile)
  unsigned B0, A0, K0, D0, TQ0, TA0, KT0;
  auto phase1_consts = [&](int ln) {
    const int r_ = ln & 31, h_ = ln >> 5;
    B0 = (unsigned)(r_ * 128 + ((h_ ^ fsw(r_)) << 4));  // row read: chunk 2 s + h of row r
    const int rowL = 4 * h_ + ((ln & 15) >> 2), chunkL = 2 * ((ln >> 4) & 1) + ((ln & 3) >> 1);
    A0 = (unsigned)(rowL * 128 + ((chunkL ^ fsw(rowL)) << 4) + 8 * (ln & 1));  // transposing read of a Q / dO tile
    // tails (16-byte rows; a tile's tails sit as [row >> 3][Q 8 rows | dO 8 rows]): row read of row r, transposing read (rows
    // 4 h + q, 8-byte half lane & 1; lanes 16-31 and columns past 8 repeat lanes 0-15 / columns 0-7: garbage rows)
    TQ0 = (unsigned)((r_ >> 3) * 256 + (r_ & 7) * 16);
    TA0 = (unsigned)(rowL * 16 + 8 * (ln & 1));
  };
  auto phase2_consts = [&](int ln) {
    const int fr_ = ln & 15, fq_ = ln >> 4;
    const int rowK = 8 * fq_ + (fr_ >> 2);
    K0 = (unsigned)(rowK * 128 + ((((fr_ & 3) >> 1) ^ fsw(rowK)) << 4) + 8 * (fr_ & 1));  // transposing read of the K image
    D0 = (unsigned)(rowK * 128 + (((fr_ & 3) ^ Fsw(rowK)) << 3));                         // ... of the dS^T image
    KT0 = (unsigned)(rowK * 16 + 8 * (fr_ & 1));                                          // ... of the K tail (rows 8 fq + (fr >> 2))
  };
  auto opaque_lane = [&]() {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    return ln;
  };
  // (same idea for the per-lane parts of the global addresses: the TAIL instance rebuilds them where they are used instead of
  // keeping a dozen hoisted 64-bit offsets alive -- those were the registers it spilled, and a spill reload inside the tile loop
  // waits on the vector-memory counter, i.e. on the DMA ring)
  auto my_tid = [&]() {
    int x = tid;
    if constexpr (TAIL) asm volatile("" : "+v"(x));
    return x;
  };
  if constexpr (!TAIL) {
    phase1_consts(lane);
    phase2_consts(lane);
  }

  auto head_ptrs = [&](int j, int& b, int& hd) {
    const int bh = (int)blockIdx.x + j * (int)gridDim.x;
    b = bh / a.H;
    hd = bh - b * a.H;
  };
  uint4 oreg;
  unsigned otail = 0;  // TAIL: lanes (tid & 7) < 4 hold columns 64 + 2 (tid & 7), + 1 of their row's O
  // tile g = 4 j + t: rows 64 t .. 64 t + 63 of head j's Q and dO into ring slot g % 3; this thread's O chunk into oreg
  auto issue_tile = [&](int g) {
    int b, hd;
    head_ptrs(g >> 2, b, hd);
    const int tid = my_tid(), lane = tid & 63, drow = tid >> 3, dchunk = (tid & 7) ^ fsw(drow);
    const int row = 64 * (g & 3) + drow;
    const bf16_t* qs = a.q + ((int64_t)b * T + row) * a.ldq + hd * DH + 8 * dchunk;
    const bf16_t* gs = a.dO + ((int64_t)b * T + row) * a.ldo + hd * DH + 8 * dchunk;
    const unsigned dst = smem_base + (unsigned)((g % P_RING) * P_SLOT + wave * 1024);
    glds16(qs, dst);
    glds16(gs, dst + 8192);
    oreg = *reinterpret_cast<const uint4*>(a.o + ((int64_t)b * T + row) * a.ldo + hd * DH + 8 * dchunk);
    if constexpr (TAIL) {
      // one more operation moves both tails (lanes 0-7: Q rows 8 wave + lane, lanes 8-15: dO rows 8 wave + lane - 8), one the
      // lse piece of the tile (every wave writes the same 256 bytes: equal instruction counts for every wave)
      const unsigned slot = smem_base + (unsigned)((g % P_RING) * P_SLOT);
      if (lane < 16) {
        const int64_t trow = (int64_t)b * T + 64 * (g & 3) + 8 * wave + (lane & 7);
        const bf16_t* src = lane < 8 ? a.q + trow * a.ldq + hd * DH + 64 : a.dO + trow * a.ldo + hd * DH + 64;
        glds16(src, slot + (unsigned)(G::OFF_TT + wave * 256));
      }
      glds4(a.lse + ((int64_t)b * a.H + hd) * T + 64 * (g & 3) + lane, slot + (unsigned)G::OFF_TL);
      // (no load, store or LDS-DMA the compiler counts sits behind a divergent branch in this kernel: a skippable operation makes
      // its wait-count pass distrust the hand-counted wait at the top of the tile and add its own -- lanes 4-7 of a row repeat
      // lanes 0-3 here and are masked in delta_of)
      otail = *reinterpret_cast<const unsigned*>(a.o + ((int64_t)b * T + row) * a.ldo + hd * DH + 64 + 2 * (tid & 3));
    }
  };
  // K image (4 pieces of 64 keys) and the lse row of head j
  auto issue_head = [&](int j) {
    int b, hd;
    head_ptrs(j, b, hd);
    const int tid = my_tid(), lane = tid & 63, drow = tid >> 3, dchunk = (tid & 7) ^ fsw(drow);
    const unsigned dst = smem_base + (unsigned)(P_OFF_K + (j & 1) * KIMG + wave * 1024);
#pragma unroll
    for (int p = 0; p < 4; ++p)  // (fsw only looks at row bits 1-4: the swizzle of row 64 p + drow is that of drow)
      glds16(a.k + ((int64_t)b * T + 64 * p + drow) * a.ldk + hd * DH + 8 * dchunk, dst + p * 8192);
    if constexpr (TAIL) {  // K tail: wave w moves rows 32 w .. 32 w + 31 (the lse travels with the tiles)
      if (lane < 32)
        glds16(a.k + ((int64_t)b * T + 32 * wave + lane) * a.ldk + hd * DH + 64,
               smem_base + (unsigned)(P_OFF_K + (j & 1) * KIMG + 32768 + wave * 512));
    } else {
      // 256 floats: waves w and w + 4 move the same 64 (equal instruction counts for every wave)
      glds4(a.lse + ((int64_t)b * a.H + hd) * T + 64 * (wave & 3) + lane,
            smem_base + (unsigned)(P_OFF_LSE + (j & 1) * 1024 + (wave & 3) * 256));
    }
  };
  uint4 kf[NKS], vf[NKS];
  auto load_kv = [&](int j) {
    int b, hd;
    head_ptrs(j, b, hd);
    const int tid = my_tid(), r = tid & 31, h = (tid >> 5) & 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[s] = *reinterpret_cast<const uint4*>(a.k + ((int64_t)b * T + k0 + r) * a.ldk + hd * DH + 16 * s + 8 * h);
      vf[s] = *reinterpret_cast<const uint4*>(a.v + ((int64_t)b * T + k0 + r) * a.ldv + hd * DH + 16 * s + 8 * h);
    }
    if constexpr (TAIL) {  // k slots 0-7 of the fifth step are columns 64..71, slots 8-15 (the upper lane half) meet zeros
      // (both lane halves load the eight real columns: the zeros of slots 8-15 are on the Q / dO side -- masking here would wait
      // for these loads, and with them for the DMA ring, in the middle of a tile)
      kf[4] = *reinterpret_cast<const uint4*>(a.k + ((int64_t)b * T + k0 + r) * a.ldk + hd * DH + 64);
      vf[4] = *reinterpret_cast<const uint4*>(a.v + ((int64_t)b * T + k0 + r) * a.ldv + hd * DH + 64);
    }
  };
  // -delta[q] = -sum_d dO[q][d] O[q][d] of tile g (its dO is in LDS, its O chunk in oreg): 8 lanes per row
  auto delta_of = [&](int g) {
    const uint4 gch = *reinterpret_cast<const uint4*>(smem + (g % P_RING) * P_SLOT + 8192 + 16 * tid);
    const bf16x8 gv = *reinterpret_cast<const bf16x8*>(&gch), ov = *reinterpret_cast<const bf16x8*>(&oreg);
    float ds = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ds += (float)gv[e] * (float)ov[e];
    if constexpr (TAIL) {
      const unsigned gtc = *reinterpret_cast<const unsigned*>(smem + (g % P_RING) * P_SLOT + G::OFF_TT + (drow >> 3) * 256 + 128 +
                                                              (drow & 7) * 16 + 4 * (tid & 3));
      const bf16x2 gt2 = *reinterpret_cast<const bf16x2*>(&gtc), ot2 = *reinterpret_cast<const bf16x2*>(&otail);
      const float dt2 = (float)gt2[0] * (float)ot2[0] + (float)gt2[1] * (float)ot2[1];
      ds += (tid & 4) ? 0.f : dt2;
    }
    ds += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ds), 0xB1, 0xF, 0xF, true));   // lane ^ 1
    ds += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ds), 0x4E, 0xF, 0xF, true));   // lane ^ 2
    ds += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ds), 0x141, 0xF, 0xF, true));  // half-row mirror
    if ((tid & 7) == 0) reinterpret_cast<float*>(smem + P_OFF_DELTA + (g & 1) * 256)[drow] = -ds;
  };

  // (TAIL: rows 64..71 of dK^T / dV^T are registers 0..3 of a third row tile -- only those four are kept across tiles, the tile's
  // own product lives in short-lived registers: 8 instead of 32 permanent ones)
  f32x16 dkT[2], dvT[2];
  f32x4 dkt = {0.f, 0.f, 0.f, 0.f}, dvt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) dkT[dt] = dvT[dt] = f32x16{};

  // dK / dV of head j: every wave passes its [32 keys][64 d] tiles through 4 KB of the K image buffer of that head (no longer
  // read: the caller has passed a barrier behind the head's last dQ phase) and stores whole 128-byte rows
  auto store_dkdv = [&](int j) {
    int b, hd;
    head_ptrs(j, b, hd);
    const int tid = my_tid(), lane = tid & 63, r = tid & 31, h = (tid >> 5) & 1;
    char* mine = smem + P_OFF_K + (j & 1) * KIMG + wave * 4096;
    char* mine_t = smem + P_OFF_K + (j & 1) * KIMG + 32768 + wave * 512;  // TAIL: [32 keys][8 d]
#pragma unroll
    for (int which = 0; which < 2; ++which) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int d0 = 32 * dt + 8 * g4 + 4 * h;
          const f32x16& x = which ? dvT[dt] : dkT[dt];
          const float sc = which ? 1.f : a.scale;
          store4(reinterpret_cast<bf16_t*>(mine + r * 128 + (((d0 >> 3) ^ (r & 7)) << 4) + 2 * (d0 & 7)),
                 f32x4{x[4 * g4] * sc, x[4 * g4 + 1] * sc, x[4 * g4 + 2] * sc, x[4 * g4 + 3] * sc});
        }
      bf16_t* dst = (which ? a.dv : a.dk) + (int64_t)b * T * (which ? a.ldv : a.ldk) + hd * DH;
      const int ld = which ? a.ldv : a.ldk;
#pragma unroll
      for (int p = 0; p < 4; ++p) {  // (same wave wrote and reads: no barrier; the compiler orders the LDS accesses)
        const int key = 8 * p + (lane >> 3), ch = lane & 7;
        const uint4 x = *reinterpret_cast<const uint4*>(mine + key * 128 + ((ch ^ (key & 7)) << 4));
        if constexpr (!(ABL & 2)) *reinterpret_cast<uint4*>(dst + (int64_t)(k0 + key) * ld + 8 * ch) = x;
        else asm volatile("" ::"v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w));
      }
      if constexpr (TAIL) {  // rows 64 + 4 h + i of the third tile are registers 0..3; the rest of it is garbage
        const f32x4 x = which ? dvt : dkt;
        const float sc = which ? 1.f : a.scale;
        store4(reinterpret_cast<bf16_t*>(mine_t + r * 16 + 8 * h), f32x4{x[0] * sc, x[1] * sc, x[2] * sc, x[3] * sc});
        // (lanes 32-63 repeat lanes 0-31: the same bytes to the same addresses)
        const uint4 xt = *reinterpret_cast<const uint4*>(mine_t + r * 16);
        *reinterpret_cast<uint4*>(dst + (int64_t)(k0 + r) * ld + 64) = xt;
      }
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) dkT[dt] = dvT[dt] = f32x16{};
    dkt = dvt = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // ---- prologue: head 0's K image / lse / fragments and tile 0, then tile 1 behind the first delta
  if (total == 0) return;
  if ((a.prio == 1 && wave >= 4) || (a.prio == 2 && wave < 4)) __builtin_amdgcn_s_setprio(1);
  if constexpr (TAIL)
    if (tid < 4) reinterpret_cast<unsigned*>(smem + G::OFF_ZERO)[tid] = 0u;  // (the prologue's barrier publishes it)
  issue_head(0);
  issue_tile(0);
  load_kv(0);
  // The waits are the BUILTIN (s_waitcnt immediates: vmcnt in bits 3:0, expcnt 6:4 = 7 "no wait", lgkmcnt 11:8): hipcc's own
  // wait-count pass sees them and learns that the K / V fragment loads and the O chunk -- ordinary loads it tracks, all older
  // than the one operation left in flight -- have landed.  With inline-asm waits it did not know, and put vmcnt(7) .. vmcnt(0)
  // in front of the first S / dP MFMAs of EVERY tile: a full drain of the DMA ring just issued (the LDS-DMA itself stays inline
  // asm: hipcc must not count it, or it would drain it before every LDS read).
  __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  delta_of(0);
  if (1 < total) issue_tile(1);
  // (hipcc puts no wait in front of a bare s_barrier: LDS writes that other waves read behind a barrier are drained by hand)
  __builtin_amdgcn_s_waitcnt(0x0070);  // tile 1: nothing younger to leave in flight yet (and the loop's wait is unconditional:
                                       // behind a branch the wait-count pass assumes the path around it)
  for (int g = 0; g < total; ++g) {
    const int j = g >> 2, t = g & 3;
    // [A] tile g + 1 (issued one iteration ago, with the K image / lse of its head if it opens one) has landed: every
    // vector-memory operation of this wave except the youngest one -- the dq store of tile g - 1 -- is complete
    // (DH = 72: the dq tail is a second store -- two operations stay in flight)
    __builtin_amdgcn_s_waitcnt(TAIL ? 0x0072 : (ABL & 3) ? 0x0070 : 0x0071);  // vmcnt(1) lgkmcnt(0)  (timing builds without that store: vmcnt(0))
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();  // [B] ... everybody's pieces; everybody has finished phase 2 of tile g - 1
    if (t == 0 && j > 0) store_dkdv(j - 1);
    if (g + 1 < total) delta_of(g + 1);                    // [D] consumes oreg (O chunk of tile g + 1)
    if (g + 2 < total) {                                   // [C] ring slot (g + 2) % 3 was tile g - 1's
      issue_tile(g + 2);
      if (((g + 2) & 3) == 0) issue_head((g + 2) >> 2);
    }
    // ---- [E] phase 1: this wave's 32 keys against the 64 query rows of tile g
    if constexpr (!(ABL & 4)) {
      if constexpr (TAIL) phase1_consts(opaque_lane());
      const unsigned Qs = (unsigned)((g % P_RING) * P_SLOT), Gs = Qs + 8192;
      const float* ls = TAIL ? reinterpret_cast<const float*>(smem + (g % P_RING) * P_SLOT + G::OFF_TL)
                             : reinterpret_cast<const float*>(smem + P_OFF_LSE + (j & 1) * 1024) + 64 * t;
      const float* dl = reinterpret_cast<const float*>(smem + P_OFF_DELTA + (g & 1) * 256);
      char* const dsimg = smem + P_OFF_DS;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        // row constants as the INITIAL accumulators: S' = Q.K^T - lse / scale, dP' = dO.V^T - delta
        f32x16 S, dP;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 l4 = load4(ls + 32 * sub + 8 * g4 + 4 * h);
          const f32x4 d4 = load4(dl + 32 * sub + 8 * g4 + 4 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            S[4 * g4 + e] = l4[e] * ninv_scale;
            dP[4 * g4 + e] = d4[e];
          }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const unsigned off = (B0 ^ (unsigned)(s << 5)) + 4096u * sub;
          const uint4 qa = *reinterpret_cast<const uint4*>(smem + Qs + off);
          const uint4 ga = *reinterpret_cast<const uint4*>(smem + Gs + off);
          S = mfma32(qa, kf[s], S);
          dP = mfma32(ga, vf[s], dP);
        }
        if constexpr (TAIL) {  // the upper lane half (k slots 8-15) reads the 16 zero bytes
          const unsigned off = Qs + (unsigned)G::OFF_TT + TQ0 + 1024u * sub;
          const uint4 qa = *reinterpret_cast<const uint4*>(smem + (h ? (unsigned)G::OFF_ZERO : off));
          const uint4 ga = *reinterpret_cast<const uint4*>(smem + (h ? (unsigned)G::OFF_ZERO : off + 128));
          S = mfma32(qa, kf[4], S);
          dP = mfma32(ga, vf[4], dP);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float p = __builtin_amdgcn_exp2f(S[i] * c);
          S[i] = p;
          dP[i] *= p;
        }
        // dS rounded to bf16 once: the pairs feed the [key][q] image (4 consecutive q per 8-byte store) and the dK operand
        bf16x2 dsp[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) dsp[jj] = bf16x2{(bf16_t)dP[2 * jj], (bf16_t)dP[2 * jj + 1]};
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          uint2 w;
          w.x = *reinterpret_cast<const unsigned*>(&dsp[2 * g4]);
          w.y = *reinterpret_cast<const unsigned*>(&dsp[2 * g4 + 1]);
          const int key = k0 + r;
          *reinterpret_cast<uint2*>(dsimg + key * 128 + (((8 * sub + 2 * g4 + h) ^ Fsw(key)) << 3)) = w;
        }
        f32x16 tv = {}, tk = {};  // TAIL: this sub-tile's rows 64..95 of dV^T / dK^T
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const uint4 pf = pack8(S, s2);
          uint4 dsf;
          dsf.x = *reinterpret_cast<const unsigned*>(&dsp[4 * s2]);
          dsf.y = *reinterpret_cast<const unsigned*>(&dsp[4 * s2 + 1]);
          dsf.z = *reinterpret_cast<const unsigned*>(&dsp[4 * s2 + 2]);
          dsf.w = *reinterpret_cast<const unsigned*>(&dsp[4 * s2 + 3]);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const unsigned x0 = (A0 ^ (unsigned)(0x30 * s2) ^ (unsigned)(0x40 * dt)) + 2048u * s2 + 4096u * sub;
            const unsigned x1 = (x0 ^ 0x20u) + 1024u;
            const uint4 gt = tr_pair(Gs + x0, Gs + x1);
            const uint4 qt = tr_pair(Qs + x0, Qs + x1);
            dvT[dt] = mfma32(gt, pf, dvT[dt]);
            dkT[dt] = mfma32(qt, dsf, dkT[dt]);
          }
          if constexpr (TAIL) {
            const unsigned x0 = Qs + (unsigned)G::OFF_TT + TA0 + 512u * s2 + 1024u * sub;
            const uint4 gt = tr_pair(x0 + 128u, x0 + 128u + 256u);
            const uint4 qt = tr_pair(x0, x0 + 256u);
            tv = mfma32(gt, pf, tv);
            tk = mfma32(qt, dsf, tk);
          }
        }
        if constexpr (TAIL) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dvt[e] += tv[e];
            dkt[e] += tk[e];
          }
        }
      }
    }
    // the K / V fragments are dead until the next head's first phase 1: load its rows now, under phase 2 and the stores
    if (t == 3 && j + 1 < nmy) load_kv(j + 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's dS^T (and delta) writes have reached LDS
    __builtin_amdgcn_s_barrier();                       // [F] dS^T image complete
    // ---- [G] phase 2: dQ^T[d][q] = K^T[d][key] . dS^T[key][q]; wave w owns q-block w & 3 and d-blocks 2 (w >> 2), + 1
    if constexpr (!(ABL & 1)) {
      if constexpr (TAIL) phase2_consts(opaque_lane());
      const int qblk = wave & 3, db0 = 2 * (wave >> 2);
      const unsigned Ki = (unsigned)(P_OFF_K + (j & 1) * KIMG), Di = (unsigned)P_OFF_DS;
      const unsigned ka = K0 ^ (unsigned)(db0 << 5), kb = K0 ^ (unsigned)((db0 + 1) << 5), da = D0 ^ (unsigned)(qblk << 5);
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
      // (TAIL: waves w and w + 4 BOTH form rows 64..71 of query block w & 3 and store the same eight bytes per lane -- one
      // instruction stream for every wave, so the hand-counted wait at the top of the tile needs no branch: behind one, hipcc's
      // wait-count pass no longer trusted it and drained the DMA ring in front of the S / dP MFMAs of every tile)
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const unsigned ko = 4096u * kk;
        const uint4 fa = tr_pair(Ki + ka + ko, Ki + (ka ^ 0x10u) + 512u + ko);
        const uint4 fb = tr_pair(Ki + kb + ko, Ki + (kb ^ 0x10u) + 512u + ko);
        const uint4 fd = tr_pair(Di + da + ko, Di + (da ^ 0x10u) + 512u + ko);
        acc0 = mfma16(fa, fd, acc0);
        acc1 = mfma16(fb, fd, acc1);
        if constexpr (TAIL) {
          const unsigned kt = Ki + 32768u + KT0 + 512u * kk;
          acc2 = mfma16(tr_pair(kt, kt + 64u), fd, acc2);
        }
      }
      // D[row = d = 16 db + 4 fq + reg][col = q = fr]: the lane holds 4 consecutive d of both d-blocks of one query row.
      // v_permlane16_swap pairs fq with fq ^ 1: even fq keeps d-block db0 (8 consecutive d), odd fq takes d-block db0 + 1
      int b, hd;
      head_ptrs(j, b, hd);
      const f32x4 v0 = acc0 * a.scale, v1 = acc1 * a.scale;
      const bf16x4 p0b = {(bf16_t)v0[0], (bf16_t)v0[1], (bf16_t)v0[2], (bf16_t)v0[3]};
      const bf16x4 p1b = {(bf16_t)v1[0], (bf16_t)v1[1], (bf16_t)v1[2], (bf16_t)v1[3]};
      const uint2 p0 = *reinterpret_cast<const uint2*>(&p0b), p1 = *reinterpret_cast<const uint2*>(&p1b);
      const su32x2 sx = __builtin_amdgcn_permlane16_swap(p0.x, p1.x, false, false);
      const su32x2 sy = __builtin_amdgcn_permlane16_swap(p0.y, p1.y, false, false);
      const bool odd = fq & 1;
      const int d = odd ? 16 * (db0 + 1) + 4 * (fq - 1) : 16 * db0 + 4 * fq;
      bf16_t* dst = a.dq + ((int64_t)b * T + 64 * t + 16 * qblk + fr) * a.ldq + hd * DH + d;
      if constexpr (!(ABL & 2)) *reinterpret_cast<uint4*>(dst) = uint4{sx[0], sy[0], sx[1], sy[1]};
      else asm volatile("" ::"v"(sx[0]), "v"(sy[0]), "v"(sx[1]), "v"(sy[1]), "v"(dst));
      if constexpr (TAIL) {
        // D[row = 64 + 4 fq + reg][col = q = fr]: lanes fq < 2 hold the eight real rows; the K tail's transposing reads fetch
        // columns 0..7 twice, so rows 72..79 -- lanes fq >= 2 -- are copies of rows 64..71 and store the same bytes again
        const f32x4 v2 = acc2 * a.scale;
        store4(a.dq + ((int64_t)b * T + 64 * t + 16 * qblk + fr) * a.ldq + hd * DH + 64 + 4 * (fq & 1), v2);
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // every wave has finished the last dQ phase: the K image buffer is free
  store_dkdv(nmy - 1);
}

// ------------------------------------------------------------------------------------------- forward
// Same machinery: a workgroup = 8 waves x 32 query rows = one whole head; K / V tiles of 64 keys (8 KB + 8 KB, row-major,
// swizzled by the DMA's source addresses) stream through a ring of F_RING slots, F_RING - 1 tiles ahead and across head
// boundaries; S^T = K.Q^T from row reads, O^T += V^T.P^T with V^T gathered by transposing reads (the k-slot permutation of the
// accumulator -> operand reuse is the one the reads' row order already has).  One barrier per tile.  Q arrives by LDS-DMA too
// (the next head's 32 KB image three tiles before it is needed; a wave reads its fragments from it once per head): with
// ordinary global loads for Q, hipcc's wait-count pass put vmcnt(3) .. vmcnt(0) in front of the S MFMAs of every tile and
// drained the ring.  The kernel has NO load the compiler tracks; every wait is hand-counted (younger()).  O leaves through a
// wave-private 4 KB LDS stage as whole 128-byte rows.  The arithmetic per output element is the sequence of the kernel in
// attention_mfma.hip (bit-identical results).
// Head dim 72: columns 64..71 travel as tails exactly as in the backward (compact 16-byte-row images: a tile's K / V tails as
// [row >> 3][K 8 rows | V 8 rows] behind its tiles, the Q tail behind the Q image, a fifth k step of S^T against Q fragments whose
// upper lane half is read from 16 zero bytes, a third row tile of O^T of which rows 64..71 = registers 0..3 are stored); the ring
// is four slots there (three tiles ahead) to stay inside the 160 KB.
template <int DH>
struct FGeom {
  static constexpr bool TAIL = DH > 64;
  static constexpr int RING = TAIL ? 4 : 5;
  static constexpr int SLOT = 16384 + (TAIL ? 2048 : 0);
  static constexpr int OFF_TT = 16384;
  static constexpr int OST = 4096 + (TAIL ? 512 : 0);           // per-wave O stage: [32 q][128 B] [+ [32 q][16 B]]
  static constexpr int OFF_O = RING * SLOT;
  static constexpr int OFF_Q = OFF_O + 8 * OST;                  // Q image [256 q][64 d] [+ tail [256 q][16 B]] of the head in flight
  static constexpr int OFF_ZERO = OFF_Q + 32768 + (TAIL ? 4096 : 0);
  static constexpr int LDS = OFF_ZERO + (TAIL ? 16 : 0);        // 147456 B (DH = 64), 147472 B (DH = 72)
};
static_assert(FGeom<64>::LDS == 147456 && FGeom<72>::LDS <= 163840, "LDS budget");
constexpr int F_LDS = FGeom<64>::LDS;

struct FArgs {
  const bf16_t *q, *k, *v;
  bf16_t* out;
  float* lse;
  int B, H, ldq, ldk, ldv, ldo, nheads;
  float scale;
  int prio;
};

template <int N>
__device__ __forceinline__ void wait_vm_n() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most n (wave-uniform, 0 .. 20+) of this wave's vector-memory operations are in flight
__device__ __forceinline__ void wait_vm(int n) {
  switch (n < 20 ? n : 20) {
    case 20: wait_vm_n<20>(); break;
    case 19: wait_vm_n<19>(); break;
    case 18: wait_vm_n<18>(); break;
    case 17: wait_vm_n<17>(); break;
    case 16: wait_vm_n<16>(); break;
    case 15: wait_vm_n<15>(); break;
    case 14: wait_vm_n<14>(); break;
    case 13: wait_vm_n<13>(); break;
    case 12: wait_vm_n<12>(); break;
    case 11: wait_vm_n<11>(); break;
    case 10: wait_vm_n<10>(); break;
    case 9: wait_vm_n<9>(); break;
    case 8: wait_vm_n<8>(); break;
    case 7: wait_vm_n<7>(); break;
    case 6: wait_vm_n<6>(); break;
    case 5: wait_vm_n<5>(); break;
    case 4: wait_vm_n<4>(); break;
    case 3: wait_vm_n<3>(); break;
    case 2: wait_vm_n<2>(); break;
    case 1: wait_vm_n<1>(); break;
    default: wait_vm_n<0>();
  }
}

template <int DH>
__global__ void __launch_bounds__(512, 2) attn_fwd_p256(const FArgs a) {
  using G = FGeom<DH>;
  constexpr bool TAIL = G::TAIL;
  constexpr int T = 256, NKS = TAIL ? 5 : 4, NDT = TAIL ? 3 : 2;
  constexpr int F_RING = G::RING, AHEAD = F_RING - 1, F_SLOT = G::SLOT, F_OFF_O = G::OFF_O, F_OFF_Q = G::OFF_Q;
  // vector-memory operations per wave: a tile (K, V [, tails]), a Q image (4 pieces [, tail]), a head's stores (4 O rows [, tail], lse)
  constexpr int TILE_OPS = TAIL ? 3 : 2, Q_OPS = TAIL ? 5 : 4, ST_OPS = TAIL ? 6 : 5;
  char* const smem = p_smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const unsigned smem_base = (unsigned)(size_t)((const __attribute__((address_space(3))) char*)p_smem);
  const float c = a.scale * 1.4426950408889634f;
  const int nmy = (a.nheads - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = 4 * nmy;
  const int q0 = wave * 32;
  const int drow = tid >> 3;
  const int dchunk = (tid & 7) ^ fsw(drow);
  const unsigned B0 = (unsigned)(r * 128 + ((h ^ fsw(r)) << 4));
  const unsigned QR0 = (unsigned)(F_OFF_Q + (q0 + r) * 128 + ((h ^ fsw(q0 + r)) << 4));  // chunk 2 s + h of query row q0 + r
  const int rowL = 4 * h + ((lane & 15) >> 2), chunkL = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
  const unsigned A0 = (unsigned)(rowL * 128 + ((chunkL ^ fsw(rowL)) << 4) + 8 * (lane & 1));
  const unsigned TQ0 = (unsigned)((r >> 3) * 256 + (r & 7) * 16);  // tails: row read of row r, transposing read (rows 4 h + q)
  const unsigned TA0 = (unsigned)(rowL * 16 + 8 * (lane & 1));
  auto head_ptrs = [&](int j, int& b, int& hd) {
    const int bh = (int)blockIdx.x + j * (int)gridDim.x;
    b = bh / a.H;
    hd = bh - b * a.H;
  };
  // per-lane byte offsets of this thread's DMA pieces inside a head's [256][ld] slab (row drow, 16-byte chunk dchunk) and
  // wave-uniform slab bases: every workgroup walks heads blockIdx.x, + gridDim.x, ...
  const unsigned kvoff = (unsigned)((drow * a.ldk + 8 * dchunk) * 2), vvoff = (unsigned)((drow * a.ldv + 8 * dchunk) * 2);
  const unsigned qvoff = (unsigned)((drow * a.ldq + 8 * dchunk) * 2);
  const bf16_t *kb_n = nullptr, *vb_n = nullptr, *qb_n = nullptr;  // slabs of the head whose tiles are being issued
  int b_c = 0, hd_c = 0;                                             // (batch, head) of the head being computed
  auto set_issue_head = [&](int j) {
    int b, hd;
    head_ptrs(j, b, hd);
    kb_n = a.k + (int64_t)b * T * a.ldk + hd * DH;
    vb_n = a.v + (int64_t)b * T * a.ldv + hd * DH;
    qb_n = a.q + (int64_t)b * T * a.ldq + hd * DH;
  };
  auto issue_tile = [&](int g) {  // keys 64 t .. 64 t + 63 of the issue head: K tile | V tile into slot g % F_RING  (2 operations)
    const int t64 = 64 * (g & 3);
    const unsigned dst = smem_base + (unsigned)((g % F_RING) * F_SLOT + wave * 1024);
    glds16_s(kb_n, kvoff + (unsigned)(t64 * a.ldk * 2), dst);
    glds16_s(vb_n, vvoff + (unsigned)(t64 * a.ldv * 2), dst + 8192);
    if constexpr (TAIL) {  // lanes 0-7: the K tail of rows 8 wave + lane, lanes 8-15: the V tail  (one more operation)
      if (lane < 16) {
        const int trow = t64 + 8 * wave + (lane & 7);
        const bf16_t* src = lane < 8 ? kb_n + (int64_t)trow * a.ldk + 64 : vb_n + (int64_t)trow * a.ldv + 64;
        glds16(src, smem_base + (unsigned)((g % F_RING) * F_SLOT + G::OFF_TT + wave * 256));
      }
    }
  };
  auto issue_q = [&]() {  // Q image of the issue head  (4 operations)
#pragma unroll
    for (int p = 0; p < 4; ++p)
      glds16_s(qb_n, qvoff + (unsigned)(64 * p * a.ldq * 2), smem_base + (unsigned)(F_OFF_Q + p * 8192 + wave * 1024));
    if constexpr (TAIL) {  // Q tail: wave w moves rows 32 w .. 32 w + 31  (one more operation)
      if (lane < 32) glds16(qb_n + (int64_t)(32 * wave + lane) * a.ldq + 64, smem_base + (unsigned)(F_OFF_Q + 32768 + wave * 512));
    }
  };
  // What a wave issues behind the DMA of tile i, inside iteration i's body: the next head's Q image (t == 1: 4 operations),
  // the head's O rows and lse (t == 3: 4 + 1 stores)
  auto extras = [&](int i) { return ((i & 3) == 1 && (i >> 2) + 1 < nmy ? Q_OPS : 0) + ((i & 3) == 3 ? ST_OPS : 0); };
  // vector-memory operations of this wave that are YOUNGER than the DMA of tile g when iteration g begins: tile g's DMA was
  // issued in iteration g - (F_RING - 1) (or in the prologue, with the tiles behind it), everything since is in issue order
  auto younger = [&](int g) {
    int y = 0;
    const int gi = g - (F_RING - 1);  // issuing iteration
    if (gi < 0) {
      for (int x = g + 1; x < F_RING - 1; ++x) y += x < total ? TILE_OPS : 0;  // the prologue's tiles behind it
    } else {
      y += extras(gi);
    }
    for (int i = gi < 0 ? 0 : gi + 1; i < g; ++i) y += (i + F_RING - 1 < total ? TILE_OPS : 0) + extras(i);
    return y;
  };
  if (total == 0) return;
  if ((a.prio == 1 && wave >= 4) || (a.prio == 2 && wave < 4)) __builtin_amdgcn_s_setprio(1);
  static_assert(AHEAD == 4 || AHEAD == 3, "the tile issued in iteration g belongs to the next head from t = 4 - AHEAD on");
  if constexpr (TAIL)
    if (tid < 4) reinterpret_cast<unsigned*>(smem + G::OFF_ZERO)[tid] = 0u;  // (published by the first tile's barrier)
  set_issue_head(0);
  head_ptrs(0, b_c, hd_c);
  issue_q();
#pragma unroll
  for (int g = 0; g < F_RING - 1; ++g) issue_tile(g);  // (total >= 4)

  uint4 qf[NKS];
  f32x16 o[NDT];
  float m = -INFINITY, l = 0.f;
  for (int g = 0; g < total; ++g) {
    const int j = g >> 2, t = g & 3;
    int n_fly = younger(g);  // tile g has landed when at most this many operations are in flight
    if (t == 0 && j > 0) {   // ... and the head's Q image, issued one iteration AFTER its first tile (in iteration g - 3, behind
      // that iteration's tile): younger than it are the tiles of iterations g - 2 and g - 1 and the previous head's 5 stores
      const int yq = (g - 2 + F_RING - 1 < total ? TILE_OPS : 0) + (g - 1 + F_RING - 1 < total ? TILE_OPS : 0) + ST_OPS;
      n_fly = yq < n_fly ? yq : n_fly;
    }
    wait_vm(n_fly);
    __builtin_amdgcn_s_barrier();  // ... everybody's pieces; everybody has finished reading tile g - 1
    if (t == 0) head_ptrs(j, b_c, hd_c);
    // the tiles issued from t = 4 - AHEAD on and the Q image issued at t == 1 are the next head's
    if (t == 4 - AHEAD && j + 1 < nmy) set_issue_head(j + 1);
    if (g + F_RING - 1 < total) issue_tile(g + F_RING - 1);  // into tile g - 1's slot
    if (t == 0) {
#pragma unroll
      for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const uint4*>(smem + (QR0 ^ (unsigned)(s << 5)));
      if constexpr (TAIL)  // k slots 8-15 of the fifth step (the upper lane half): zeros
        qf[4] = *reinterpret_cast<const uint4*>(smem + (h ? (unsigned)G::OFF_ZERO : (unsigned)(F_OFF_Q + 32768 + (q0 + r) * 16)));
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) o[dt] = f32x16{};
      m = -INFINITY;
      l = 0.f;
    }
    if (t == 1 && j + 1 < nmy) issue_q();  // every wave has its fragments (read before this iteration's barrier)
    const unsigned Ks = (unsigned)((g % F_RING) * F_SLOT), Vs = Ks + 8192;
    f32x16 s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      s[kt] = f32x16{};
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) {
        const uint4 kfr = *reinterpret_cast<const uint4*>(smem + Ks + (B0 ^ (unsigned)(ss << 5)) + 4096u * kt);
        s[kt] = mfma32(kfr, qf[ss], s[kt]);
      }
      if constexpr (TAIL) {
        const uint4 kfr = *reinterpret_cast<const uint4*>(smem + Ks + (unsigned)G::OFF_TT + TQ0 + 1024u * kt);
        s[kt] = mfma32(kfr, qf[4], s[kt]);
      }
    }
    float mx = s[0][0];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[kt][i]);
    {  // the other lane half holds the other 32 keys of the row
      float lo, hi;
      lane_halves(mx, lo, hi);
      mx = fmaxf(lo, hi);
    }
    const float mn = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f((m - mn) * c);
    const float mc = mn * c;
    float ls = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float p = __builtin_amdgcn_exp2f(s[kt][i] * c - mc);
        s[kt][i] = p;
        ls += p;
      }
    l = l * alpha + ls;
    m = mn;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
    if constexpr (TAIL) {  // rows 64..71 are registers 0..3 of the third tile; the rest of it is never stored
#pragma unroll
      for (int i = 0; i < 4; ++i) o[2][i] *= alpha;
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const uint4 pf = pack8(s[kt], s2);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const unsigned x0 = (A0 ^ (unsigned)(0x30 * s2) ^ (unsigned)(0x40 * dt)) + 2048u * s2 + 4096u * kt;
          const uint4 vt = tr_pair(Vs + x0, Vs + (x0 ^ 0x20u) + 1024u);
          o[dt] = mfma32(vt, pf, o[dt]);
        }
        if constexpr (TAIL) {
          const unsigned x0 = Ks + (unsigned)G::OFF_TT + 128u + TA0 + 512u * s2 + 1024u * kt;
          o[2] = mfma32(tr_pair(x0, x0 + 256u), pf, o[2]);
        }
      }
    if (t == 3) {  // the head is complete: O / l through the wave's LDS stage as whole rows, lse  (4 + 1 stores: extras())
      const int b = b_c, hd = hd_c;
      float lt;
      {
        float lo, hi;
        lane_halves(l, lo, hi);
        lt = lo + hi;
      }
      const float inv = 1.f / lt;
      char* mine = smem + F_OFF_O + wave * G::OST;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int d0 = 32 * dt + 8 * g4 + 4 * h;
          store4(reinterpret_cast<bf16_t*>(mine + r * 128 + (((d0 >> 3) ^ (r & 7)) << 4) + 2 * (d0 & 7)),
                 f32x4{o[dt][4 * g4] * inv, o[dt][4 * g4 + 1] * inv, o[dt][4 * g4 + 2] * inv, o[dt][4 * g4 + 3] * inv});
        }
      bf16_t* ob = a.out + ((int64_t)b * T + q0) * a.ldo + hd * DH;
#pragma unroll
      for (int p = 0; p < 4; ++p) {  // (same wave wrote and reads: no barrier)
        const int row = 8 * p + (lane >> 3), ch = lane & 7;
        const uint4 x = *reinterpret_cast<const uint4*>(mine + row * 128 + ((ch ^ (row & 7)) << 4));
        *reinterpret_cast<uint4*>(ob + (int64_t)row * a.ldo + 8 * ch) = x;
      }
      if constexpr (TAIL) {  // (lanes 32-63 repeat lanes 0-31: one store instruction, the same bytes to the same addresses)
        store4(reinterpret_cast<bf16_t*>(mine + 4096 + r * 16 + 8 * h), f32x4{o[2][0] * inv, o[2][1] * inv, o[2][2] * inv, o[2][3] * inv});
        const uint4 xt = *reinterpret_cast<const uint4*>(mine + 4096 + r * 16);
        *reinterpret_cast<uint4*>(ob + (int64_t)r * a.ldo + 64) = xt;
      }
      // (every lane stores: lanes r and r + 32 write the same value to the same address -- one store INSTRUCTION either way,
      // which is what the hand-counted waits count)
      a.lse[((int64_t)b * a.H + hd) * T + q0 + r] = m * a.scale + __logf(lt);
    }
  }
}

}  // namespace

static int p256_prio() {
  static UwuEnv e("UWU_P256_PRIO");
  return e.get().set ? e.ival : 0;
}

bool uwu_attn_p256_ok(int T, int Tk, int d, int ldq, int ldk, int ldv, int ldo) {
  static UwuEnv on("UWU_ATTN_P256");  // "0": the one-workgroup-per-head kernel of attention_mfma.hip (A/B comparisons)
  // (a device that cannot give one workgroup P_LDS bytes falls through to the per-head kernels of attention_mfma.hip)
  if (on.get().is('0') || T != 256 || Tk != 256 || ldq % 8 || ldk % 8 || ldv % 8 || ldo % 8 || uwu_dev_cus() <= 0) return false;
  if (d == 64) return uwu_dev_lds_fits(P_LDS);
  static UwuEnv on72("UWU_ATTN_P256_D72");  // "0": head dim 72 stays on the key-block + dq kernels of attention_mfma.hip
  return d == 72 && !on72.get().is('0') && uwu_dev_lds_fits(PGeom<72>::LDS);
}

int uwu_attn_p256_bwd(const void* q, const void* k, const void* v, const void* o, const void* dO, const float* lse, void* dq,
                      void* dk, void* dv, int B, int H, int d, int ldq, int ldk, int ldv, int ldo, float scale, hipStream_t st) {
  static unsigned char done[5][UWU_MAX_DEV];  // per kernel instance, per device
  const int n_cu = uwu_dev_cus();
  if (d == 72) {
    static unsigned char done72[UWU_MAX_DEV];
    if (n_cu <= 0 || !uwu_func_lds(reinterpret_cast<const void*>(attn_bwd_p256<0, 72>), PGeom<72>::LDS, done72)) {
      uwu_set_error("attention_bwd(p256, d = 72): the device cannot give a workgroup %d bytes of LDS", (int)PGeom<72>::LDS);
      return UWU_ELAUNCH;
    }
    PArgs a{};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (const bf16_t*)o; a.dO = (const bf16_t*)dO;
    a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.lse = lse;
    a.B = B; a.H = H; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.nheads = B * H; a.scale = scale;
    a.prio = p256_prio();
    const int grid = a.nheads < n_cu ? a.nheads : n_cu;
    hipLaunchKernelGGL((attn_bwd_p256<0, 72>), dim3(grid), dim3(512), PGeom<72>::LDS, st, a);
    UWU_LAUNCH_CHECK("attention_bwd(p256, d = 72)");
    return UWU_OK;
  }
  if (n_cu <= 0 || !uwu_func_lds(reinterpret_cast<const void*>(attn_bwd_p256<0>), P_LDS, done[0]) ||
      !uwu_func_lds(reinterpret_cast<const void*>(attn_bwd_p256<1>), P_LDS, done[1]) ||
      !uwu_func_lds(reinterpret_cast<const void*>(attn_bwd_p256<2>), P_LDS, done[2]) ||
      !uwu_func_lds(reinterpret_cast<const void*>(attn_bwd_p256<3>), P_LDS, done[3]) ||
      !uwu_func_lds(reinterpret_cast<const void*>(attn_bwd_p256<4>), P_LDS, done[4])) {
    uwu_set_error("attention_bwd(p256): the device cannot give a workgroup %d bytes of LDS", (int)P_LDS);
    return UWU_ELAUNCH;
  }
  PArgs a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (const bf16_t*)o; a.dO = (const bf16_t*)dO;
  a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.lse = lse;
  a.B = B; a.H = H; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.nheads = B * H; a.scale = scale;
  a.prio = p256_prio();
  // one workgroup per CU (150 KB of LDS each); every workgroup exits after its last head -- no inter-workgroup waits
  const int grid = a.nheads < n_cu ? a.nheads : n_cu;
  static UwuEnv abl("UWU_P256_ABL");  // timing-only ablations / stamps (tools/bench_attn.py); results are wrong with them
  switch (abl.get().ival) {
    case 1: hipLaunchKernelGGL(attn_bwd_p256<1>, dim3(grid), dim3(512), P_LDS, st, a); break;
    case 2: hipLaunchKernelGGL(attn_bwd_p256<2>, dim3(grid), dim3(512), P_LDS, st, a); break;
    case 3: hipLaunchKernelGGL(attn_bwd_p256<3>, dim3(grid), dim3(512), P_LDS, st, a); break;
    case 4: hipLaunchKernelGGL(attn_bwd_p256<4>, dim3(grid), dim3(512), P_LDS, st, a); break;
    default: hipLaunchKernelGGL(attn_bwd_p256<0>, dim3(grid), dim3(512), P_LDS, st, a);
  }
  UWU_LAUNCH_CHECK("attention_bwd(p256)");
  return UWU_OK;
}

bool uwu_attn_p256_fwd_ok(int nheads, int T, int Tk, int d, int ldq, int ldk, int ldv, int ldo) {
  static UwuEnv on("UWU_ATTN_P256F");  // "0": the kernel of attention_mfma.hip (A/B comparisons); "1": at every head count
  if (on.get().is('0')) return false;
  // One workgroup per CU walking nheads / 256 heads: with few heads per workgroup the uneven shares (384 heads: 19.6 us
  // against 16.0 for the two-workgroups-per-head kernel) and the lack of a second workgroup per CU cost more than the
  // streaming gains (768 heads: 28.6 vs 27.0 us; 4608: 155 vs 160).  The backward wins at every size (96 heads: 19 vs 23 us).
  if (!on.is('1') && nheads < 1024) return false;
  if (T != 256 || Tk != 256 || ldq % 8 || ldk % 8 || ldv % 8 || ldo % 8 || uwu_dev_cus() <= 0) return false;
  if (d == 64) return uwu_dev_lds_fits(F_LDS);
  static UwuEnv on72("UWU_ATTN_P256F_D72");  // "0": head dim 72 stays on the kernel of attention_mfma.hip
  return d == 72 && !on72.get().is('0') && uwu_dev_lds_fits(FGeom<72>::LDS);
}

int uwu_attn_p256_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int d, int ldq, int ldk, int ldv,
                      int ldo, float scale, hipStream_t st) {
  static unsigned char done[2][UWU_MAX_DEV];
  const int n_cu = uwu_dev_cus();
  const int lds = d == 72 ? FGeom<72>::LDS : F_LDS;
  const void* fn = d == 72 ? reinterpret_cast<const void*>(attn_fwd_p256<72>) : reinterpret_cast<const void*>(attn_fwd_p256<64>);
  if (n_cu <= 0 || !uwu_func_lds(fn, lds, done[d == 72])) {
    uwu_set_error("attention_fwd(p256): the device cannot give a workgroup %d bytes of LDS", lds);
    return UWU_ELAUNCH;
  }
  FArgs a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.out = (bf16_t*)o; a.lse = lse;
  a.B = B; a.H = H; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.nheads = B * H; a.scale = scale;
  a.prio = p256_prio();
  const int grid = a.nheads < n_cu ? a.nheads : n_cu;
  if (d == 72) hipLaunchKernelGGL(attn_fwd_p256<72>, dim3(grid), dim3(512), lds, st, a);
  else hipLaunchKernelGGL(attn_fwd_p256<64>, dim3(grid), dim3(512), lds, st, a);
  UWU_LAUNCH_CHECK("attention_fwd(p256)");
  return UWU_OK;
}
