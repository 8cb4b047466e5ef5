// 256 x 256 fp8 (block-scaled MFMA, unit block scales) GEMM for the fp8 Linears of DiT-XL/2 (BASELINE config 5; reference: the
// nn.Linear fwd / bwd inside the transformer blocks, src/duwu/modules/rope_unet.py:122-166, 393-411, under the reference's fp8
// autocast): the 8-phase schedule of gemm_p8.hip on fp8 operands.  gemm_f8_kernel (gemm.hip) runs the same tile as a two-stage
// loop with one vmcnt(0) + barrier per K step -- at twice the MFMA rate of bf16 a K step is half as long and the exposed latency
// per step weighs twice as much (0.22 - 0.40 of the 5 PFLOP/s fp8 peak at the DiT-XL/2 shapes).
//
// Everything that moves bytes is the bf16 kernel unchanged: a K step is still 128-byte rows (128 fp8 here, 64 bf16 there), the
// operands of a K step are the four half-tiles B0 A0 B1 A1 of 128 rows x 128 B, the ring has eight slots filled by LDS-DMA seven
// elements ahead, one counted vmcnt per K step, two wave groups one barrier apart, the persistent grid with the element stream
// running on into the next tile.  A lane's operand of v_mfma_scale_f32_16x16x128_f8f6f4 is 32 bytes = the 16-byte chunks fq and
// 4 + fq of its row -- exactly the two ds_read_b128 per fragment that the bf16 kernel issues for its k halves --, so a quadrant
// is 4 x 2 MFMAs of 32 cycles instead of 2 x 4 x 2 of 16.  Every operand is contraction-contiguous ("NT": forward x . W^T,
// input gradient dY . (W^T)^T, weight gradient dY^T . X^T^T -- the transposed fp8 copies exist, quant.hip), so there is no
// transposing-read form.  alpha = 1 / (scale_a scale_b) multiplies the accumulators in front of the shared epilogue.
//
// PART (weight gradients): a unit of work is (K slice z, tile); slice z of a [M, N] fp32 partial-sum slab in the scratch takes
// the tile's sum over its K range, splitk_reduce adds the slabs into C.  Units are numbered slice-major, so the contiguous chunk
// of units an XCD walks stays inside one or two slices (their operand ranges share that XCD's L2).
#include "gemm_shared.h"

int uwu_p8_cus();  // gemm_p8.hip

namespace {

constexpr int P8_HT = 128 * ROW_BYTES;  // half-tile: 128 rows x 128 B
constexpr int P8_RING = 8 * P8_HT;     // ring of eight half-tile slots (128 KB)
constexpr int P8_BIAS = P8_RING;       // 8 waves x 64 bias floats
constexpr int P8_LDS = P8_BIAS + 2048;
typedef int fi32x8 __attribute__((ext_vector_type(8)));

template <int H>
using IC = std::integral_constant<int, H>;

// LDS image of a K-contiguous half-tile: [128 rows][128 B], 16-byte chunk c of row r at position c ^ ((r >> 1) & 7).  The
// ds_read_b128 lane groups (16 lanes: rows fr of one parity pair set, chunk 4 kk + fq) then cover all 64 banks once, and -- unlike
// swz() of gemm_shared.h, whose (r >> 4) term serves register-staged transposed writes -- the address of fragment i is the
// address of fragment 0 plus 2048 i: one address register per operand and k half instead of one per fragment.
__device__ __forceinline__ int p8_swz(int row, int chunk) { return row * ROW_BYTES + (((chunk ^ (row >> 1)) & 7) << 4); }

// PERSISTENT: grid = one workgroup per CU; a workgroup walks units L = blockIdx.x, + gridDim.x, ... exactly as gemm_p8_kernel
// (see there for the tile-boundary argument: the next tile's first seven half-tiles are requested before this tile's results
// leave, the wait of its K step 0 lets the stores stay in flight).
template <typename TC, int EPI, int FA, bool PART>
__global__ void __launch_bounds__(512, 2) gemm_p8f_kernel(const GemmArgs g, const float* __restrict__ scale_a,
                                                          const float* __restrict__ scale_b) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;  // (the epilogue's bias / output conversions; the operands are bytes)
  constexpr bool TB = false;
  constexpr int ABL = 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int ntile = g.tiles_m * g.tiles_n;
  const int nblk = PART ? ntile * g.wide : ntile;   // units of work (PART: g.wide K slices per tile, slice-major)
  const int nk = PART ? g.k_tiles_per_split : g.K >> 7;  // K steps of 128 fp8 per unit
  const float alpha = 1.f / (scale_a[0] * scale_b[0]);
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
  int zs = 0, ezs = 0;  // K slice of the unit being loaded / of the unit whose results are in the accumulators
  auto setup = [&](int unit, int& m0, int& n0) __attribute__((always_inline)) {
    int tile = unit;
    if constexpr (PART) {
      zs = unit / ntile;
      tile = unit - zs * ntile;
    }
    const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
    m0 = tm * 256;
    n0 = tn * 256;
    const int64_t k0 = PART ? (int64_t)zs * nk * 128 : 0;
    abase = static_cast<const char*>(g.A) + (int64_t)m0 * g.lda + k0;
    bbase = static_cast<const char*>(g.B) + (int64_t)n0 * g.ldb + k0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int row = 8 * (wave + 8 * q) + (lane >> 3);
        const int c = ((lane & 7) ^ (row >> 1)) & 7;  // logical chunk that must land at position lane & 7 (p8_swz)
        int ga = 128 * half + row;
        if (ga >= g.M - m0) ga = g.M - m0 - 1;  // (clamped rows / columns: their products are never stored)
        oa[half][q] = (unsigned)(ga * g.lda + 16 * c);
        int gb = 128 * half + row;
        if (gb >= g.N - n0) gb = g.N - n0 - 1;
        ob[half][q] = (unsigned)(gb * g.ldb + 16 * c);
      }
    }
  };
  const int64_t bstep = 128;  // bytes per K step
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
  f32x4 acc[2][2][4][2];
  // a fragment = 8 consecutive registers (the MFMA's 32-byte operand): chunks fq | 4 + fq of its row.  Plain LDS loads, not the
  // inline-asm reads of gemm_p8.hip: the register allocator can then place the two 16-byte halves straight into the tuple (with
  // asm outputs it copied every fragment -- 188 v_mov_b64 -- and spilled inside the K loop)
  fi32x8 af[4], bf0[2], bf1[2];
  typedef const __attribute__((address_space(3))) gu32x4* lds_v4;
  auto frag8 = [&](unsigned lo_addr, unsigned hi_addr) __attribute__((always_inline)) {
    const gu32x4 lo = *reinterpret_cast<lds_v4>((size_t)lo_addr), hi = *reinterpret_cast<lds_v4>((size_t)hi_addr);
    return fi32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };

  auto read_a = [&](auto slotc) __attribute__((always_inline)) {  // 8 reads
    constexpr unsigned off = decltype(slotc)::value * P8_HT;
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = frag8(a_ad[0] + off + 2048u * i, a_ad[1] + off + 2048u * i);
  };
  auto read_b = [&](auto slotc, fi32x8 (&bf)[2]) __attribute__((always_inline)) {  // 4 reads
    constexpr unsigned off = decltype(slotc)::value * P8_HT;
#pragma unroll
    for (int j = 0; j < 2; ++j) bf[j] = frag8(b_ad[0] + off + 2048u * j, b_ad[1] + off + 2048u * j);
  };
  const int unit_scale = 0x7F7F7F7F;  // E8M0 block scales of 1.0 for both operands
  auto mma_quadrant = [&](f32x4 (&c)[4][2], const fi32x8 (&bf)[2]) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    // swapped operands as in gemm_f8_kernel (a lane ends up with 4 consecutive columns): MFMA-A = the weight fragment (e4m3),
    // MFMA-B = the activation / gradient fragment (format FA); unit E8M0 block scales
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // (inline asm with the accumulator tied in place: with the builtin, hipcc's register allocation treated the result of this
        // instruction as a fresh early-clobber tuple -- 38 registers more than the same loop on bf16 MFMAs, spilled inside the K loop)
        if constexpr (FA == 0)
          asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
                       : "+v"(c[i][j]) : "v"(bf[j]), "v"(af[i]), "v"(unit_scale));
        else
          asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] blgp:1"
                       : "+v"(c[i][j]) : "v"(bf[j]), "v"(af[i]), "v"(unit_scale));
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
  constexpr int NST = 16;  // stores of a full bf16 tile per wave (PART: full_tile() is false -- nothing is assumed about its stores)
  // CONTINUOUS mode (K steps even, no epilogue loads): the element stream runs on into the NEXT tile -- the last two K steps of a
  // tile request the first seven half-tiles of the next one -- and the finished accumulators leave quadrant by quadrant in the
  // load intervals of the next tile's K step 0 (quadrant q is final after phase q of the last K step and needed again in phase
  // q of the next K step 0), so a tile boundary has no drain, no refill and no extra barrier.  Otherwise (dGELU: its epilogue
  // loads would drain the queue; odd K step counts) a tile ends with the two wave groups back in step, requests the next tile
  // ahead of its epilogue and starts over.
  constexpr bool can_cont = true;
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
      // (the lane's share of the output addresses is rebuilt here from a lane id hipcc cannot see through: hoisted out of the
      // tile loop, eight copies of this epilogue kept a dozen 64-bit addresses alive across the K loop and spilled them)
      // (the MFMAs are inline asm: hipcc's hazard recogniser does not know that acc[x][y] is a matrix-pipe result.  Its last
      // write is at least three phases -- barriers -- back; the wait states make the distance explicit anyway)
      int ln = lane;
      asm volatile("s_nop 15\n\ts_nop 15" : "+v"(ln));
      const int fr = ln & 15, fq = ln >> 4;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[x][y][i][j] = acc[x][y][i][j] * alpha;
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
      if constexpr (PART) {  // fp32 partial sums of K slice ezs: dense [M][N] slab ezs of the scratch (g.C2)
        float* P = static_cast<float*>(g.C2) + (int64_t)ezs * g.M * g.N;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = m_q + 16 * i + fr;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int n = n_q + 16 * j + 4 * fq;
            if (m < g.M && n < g.N) store4(P + (int64_t)m * g.N + n, acc[x][y][i][j]);
          }
        }
      } else {
        epilogue_tile<T, TC, 4, 2, EPI>(acc[x][y], pre, g, m_q, n_q, fr, fq, nullptr, 0, 0, -1, nullptr);
      }
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
      ezs = zs;
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
  auto stamp = [&](int) __attribute__((always_inline)) {};
  auto full_tile = [&]() __attribute__((always_inline)) {
    return !PART && em0 + 256 <= g.M && en0 + 256 <= g.N && g.wide && sizeof(TC) == 2;
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
      continue;
    }
    if (grp == 0 && ABL != 3) bar();  // every wave has passed the same number of barriers; nobody reads LDS any more
    em0 = m0;
    en0 = n0;
    ezs = zs;
    constexpr bool early = true;
    if (early && has_next) {
      setup(tile_of(Ln), m0, n0);
      prologue();
      __builtin_amdgcn_sched_barrier(0);
    }
    stamp(5);
    epi_quadrant(IC<0>{}, IC<0>{});
    epi_quadrant(IC<0>{}, IC<1>{});
    epi_quadrant(IC<1>{}, IC<1>{});
    epi_quadrant(IC<1>{}, IC<0>{});
    stamp(6);
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


template <typename TC, int EPI, int FA, bool PART>
int launch_p8f(GemmArgs g, const float* sa, const float* sb, hipStream_t st) {
  auto kern = gemm_p8f_kernel<TC, EPI, FA, PART>;
  static unsigned char done[UWU_MAX_DEV];
  if (!uwu_func_lds(reinterpret_cast<const void*>(kern), P8_LDS, done)) {
    uwu_set_error("gemm_p8f: the device cannot give a workgroup %d bytes of LDS", P8_LDS);
    return UWU_ELAUNCH;
  }
  const int units = g.tiles_m * g.tiles_n * (PART ? g.wide : 1), ncu = uwu_p8_cus();
  g.p8_cont = 1;
  hipLaunchKernelGGL(kern, dim3(units < ncu ? units : ncu), dim3(512), P8_LDS, st, g, sa, sb);
  UWU_LAUNCH_CHECK("gemm_p8f");
  return UWU_OK;
}

}  // namespace

// fp8 in / bf16 out (forward, input gradient): K a multiple of 128, at least 4 K steps, enough tiles (an even number of K steps
// runs as one continuous element stream across tiles, an odd one drains at every tile end).  UWU_GEMM_P8F=0: off, =1: every shape it can run (tests, A/B comparisons).
bool uwu_gemm_p8f_ok(const GemmArgs& g) {
  static UwuEnv on("UWU_GEMM_P8F"), tmin_e("UWU_P8F_MINTILES");
  if (on.get().is('0') || !uwu_dev_lds_fits(P8_LDS)) return false;
  static UwuEnv odd("UWU_P8F_ODD");  // "0": only even K step counts (the continuous stream); odd ones drain at every tile end
  if (g.K % 128 || g.K < 512 || (odd.get().is('0') && g.K % 256)) return false;
  if (g.epi != UWU_EPI_NONE && g.epi != UWU_EPI_BIAS) return false;
  if (on.is('1')) return true;
  const int64_t tiles = (int64_t)((g.M + 255) / 256) * ((g.N + 255) / 256);
  const int tmin = tmin_e.get().set ? tmin_e.ival : 160;
  return tiles >= tmin;
}

int uwu_launch_gemm_p8f(GemmArgs g, int fmt_a, const float* sa, const float* sb, hipStream_t st) {
  g.tiles_m = (g.M + 255) / 256;
  g.tiles_n = (g.N + 255) / 256;
  if (g.epi == UWU_EPI_NONE)
    return fmt_a ? launch_p8f<bf16_t, UWU_EPI_NONE, 1, false>(g, sa, sb, st) : launch_p8f<bf16_t, UWU_EPI_NONE, 0, false>(g, sa, sb, st);
  if (g.epi == UWU_EPI_BIAS)
    return fmt_a ? launch_p8f<bf16_t, UWU_EPI_BIAS, 1, false>(g, sa, sb, st) : launch_p8f<bf16_t, UWU_EPI_BIAS, 0, false>(g, sa, sb, st);
  uwu_set_error("gemm_p8f: epilogue %d not instantiated", g.epi);
  return UWU_EINVAL;
}

// Weight gradients: the number of K slices for `tiles` output tiles and `steps` K steps -- a divisor of steps with an even
// quotient >= 4 that fills the rounds of one-unit-per-CU best (ties: fewer slices = fewer partial sums); 0 = none qualifies.
int uwu_gemm_p8f_split(int tiles, int steps) {
  const int ncu = uwu_p8_cus();
  int best = 0;
  double best_fill = 0.0;
  for (int s = 1; s <= 64 && s <= steps; ++s) {
    if (steps % s) continue;
    const int q = steps / s;
    if (q < 4 || (q & 1)) continue;
    const int64_t units = (int64_t)tiles * s;
    const int64_t rounds = (units + ncu - 1) / ncu;
    // a slice more = one more fp32 slab written and read back by the reduce: ~2.5 % of a launch at the DiT-XL/2 sizes
    const double fill = (double)units / (double)(rounds * ncu) - 0.025 * s;
    if (fill > best_fill) {
      best_fill = fill;
      best = s;
    }
  }
  return best;
}

bool uwu_gemm_p8f_part_ok(const GemmArgs& g) {
  static UwuEnv on("UWU_GEMM_P8F_PART");
  if (on.get().is('0') || !uwu_dev_lds_fits(P8_LDS) || g.K % 128) return false;
  const int tiles = ((g.M + 255) / 256) * ((g.N + 255) / 256);
  return uwu_gemm_p8f_split(tiles, g.K / 128) > 0;
}

// partial sums into `scratch` ([split][M][N] fp32); the caller adds them into C (launch_splitk_reduce).  Returns the slice count
// through g.wide.
int uwu_launch_gemm_p8f_part(GemmArgs& g, int fmt_a, const float* sa, const float* sb, void* scratch, hipStream_t st) {
  g.tiles_m = (g.M + 255) / 256;
  g.tiles_n = (g.N + 255) / 256;
  const int steps = g.K / 128;
  const int split = uwu_gemm_p8f_split(g.tiles_m * g.tiles_n, steps);
  g.wide = split;
  g.k_tiles_per_split = steps / split;
  g.C2 = scratch;
  return fmt_a ? launch_p8f<float, UWU_EPI_NONE, 1, true>(g, sa, sb, st) : launch_p8f<float, UWU_EPI_NONE, 0, true>(g, sa, sb, st);
}
