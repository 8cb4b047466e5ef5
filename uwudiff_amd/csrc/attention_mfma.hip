// MFMA flash attention for gfx950 (bf16 operands, fp32 softmax/accumulate); head dim 64 (the tuned case), 72 (DiT-XL)
// and 128; self- and cross-attention.
// Head dims other than 64 reuse the 64-wide machinery: every LDS image is NB = ceil(DH/64) blocks of [64][64] bf16 with
// the same swizzle, columns past DH are zero-filled while staging, and the product loops run over NKS = ceil(DH/16)
// contraction steps and NDT = ceil(DH/32) output tiles (d = 72: 5 and 3 instead of 4.5 and 2.25 -- the MFMAs are not
// the bound here, the softmax VALU work is, and that does not depend on DH).
// Replaces F.scaled_dot_product_attention fwd/bwd (reference src/duwu/modules/rope_unet.py:151-153).
//
// All products use v_mfma_f32_32x32x16_bf16.  Lane maps (cdna_hip_programming.md section 3):
//   A: lane (r=l&31, h=l>>5) holds A[row r][k=8h+j];  B: holds B[k=8h+j][col r];
//   C/D: col = l&31, row = (i&3) + 8*(i>>2) + 4*h for register i in [0,16).
// An accumulator tile X[row][col] is reused as the B operand of the next MFMA (contraction over X's ROW index)
// by converting registers 8s..8s+7 to bf16: logical k-slot (h,j) then means row 16s + 8(j>>2) + 4h + (j&3),
// so the OTHER operand's LDS image is stored with that row permutation (done by the transposing stager).
//
// forward  (4 waves x 32 query rows, KV tiles of 64 keys, online softmax):
//   S^T[key][q] = K.Q^T  (K tile in LDS, Q fragments in registers)  -> softmax over registers + one lane-half
//   exchange -> P^T stays in registers as the B operand of  O^T[d][q] += V^T[d][key].P^T[key][q].
// backward (8 waves x 32 keys, whole key range <= 256 in one workgroup, query tiles of 64 rows):
//   S[q][key] = Q.K^T, dP[q][key] = dO.V^T   (key on the lane; K/V fragments in registers)
//   P = exp2(c*S'), dS = P*dP' with S' = S - lse/scale and dP' = dP - delta formed by the accumulators' initial values
//   dV^T[d][key] += dO^T[d][q].P[q][key],  dK^T[d][key] += Q^T[d][q].dS[q][key]   (accumulators as B operands)
//   dQ contracts over the key = the LANE index of dS, so dS takes the one trip through LDS: every wave writes
//   its 32-key slice into a shared [q][key] bf16 image, then each wave computes two 16x16 blocks of
//   dQ^T[d][q] = K^T[d][key].dS^T[key][q] over ALL keys with v_mfma_f32_16x16x32_bf16 against a resident K^T
//   image -- no cross-wave reduction (LDS float atomics measured 5x slower than the whole rest of the kernel).
#include "common.h"

namespace {

__device__ __forceinline__ int swz(int row, int chunk) {
  return row * 128 + (((chunk ^ (row >> 1) ^ (row >> 4)) & 7) << 4);
}
__device__ __forceinline__ f32x16 mfma32(const uint4& a, const uint4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a),
                                                 *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}
__device__ __forceinline__ uint4 pack8(const f32x16& x, int s) {
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (bf16_t)x[8 * s + j];
  return *reinterpret_cast<uint4*>(&f);
}
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }

struct MArgs {
  const bf16_t *q, *k, *v, *o, *dO;
  bf16_t *out, *dq, *dk, *dv;
  float* lse;
  float* delta;
  const float* kbias;  // optional additive score bias per key, fp32 [B, Tk] (same for every head and query)
  int B, T, Tk, H, ldq, ldk, ldv, ldo;  // T = queries, Tk = keys (== T for self-attention)
  float scale;
  // ROPE variants: per-(token, head, element) factors of the reference's axial RoPE (y = x * m, rope.hip), fp32 [T, ldt]
  // with the head's DH columns at hd * DH; q and k are multiplied while they are staged (both stay un-rotated in HBM)
  const float* rope;
  int ldt;
};

__device__ __forceinline__ uint4 rope8(uint4 v, const float* __restrict__ t) {
  const bf16x8 b = *reinterpret_cast<const bf16x8*>(&v);
  const f32x4 t0 = load4(t), t1 = load4(t + 4);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    o[j] = (bf16_t)((float)b[j] * t0[j]);
    o[4 + j] = (bf16_t)((float)b[4 + j] * t1[j]);
  }
  return *reinterpret_cast<const uint4*>(&o);
}
__device__ __forceinline__ uint2 rope4(uint2 v, const float* __restrict__ t) {
  const bf16x4 b = *reinterpret_cast<const bf16x4*>(&v);
  const f32x4 t0 = load4(t);
  bf16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (bf16_t)((float)b[j] * t0[j]);
  return *reinterpret_cast<const uint2*>(&o);
}

// Row-major staging of a [64 rows][64 bf16] tile: `nthr` threads, 16 B per lane, swizzled image.
// Transposed staging of the same tile into a [64 cols][64 rows-permuted] image: 256 threads, each 4 rows x 4 cols.
struct TStage {
  uint2 r[4];
  // rows past `nrows` are clamped to the last one (cross-attention key tiles: their scores are masked)
  // columns at or past `ncols` (head dims that do not fill the 64-wide block) read as zero
  __device__ __forceinline__ void load(const bf16_t* __restrict__ base, int ld, int row0, int t256,
                                       int nrows = 0x7fffffff, int ncols = 64, const float* __restrict__ tab = nullptr,
                                       int tld = 0) {
    const int cg = t256 & 15, rq = t256 >> 4;  // cols 4cg..4cg+3, rows 4rq..4rq+3
#pragma unroll
    for (int kr = 0; kr < 4; ++kr) {
      int row = row0 + 4 * rq + kr;
      if (row >= nrows) row = nrows - 1;
      r[kr] = (4 * cg < ncols) ? *reinterpret_cast<const uint2*>(base + (int64_t)row * ld + 4 * cg) : uint2{0u, 0u};
      if (tab) r[kr] = rope4(r[kr], tab + (int64_t)row * tld + 4 * cg);
    }
  }
  __device__ __forceinline__ void store(char* __restrict__ lds, int t256) const {
    const int cg = t256 & 15, rq = t256 >> 4;
    const int chunk = 2 * (rq >> 2) + (rq & 1), sub = 8 * ((rq >> 1) & 1);
    const unsigned* w0 = reinterpret_cast<const unsigned*>(&r[0]);
    const unsigned* w1 = reinterpret_cast<const unsigned*>(&r[1]);
    const unsigned* w2 = reinterpret_cast<const unsigned*>(&r[2]);
    const unsigned* w3 = reinterpret_cast<const unsigned*>(&r[3]);
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      const int row = 4 * cg + ci;  // image row = source column
      const unsigned sel = (ci & 1) ? 0x07060302u : 0x05040100u;
      uint2 o;
      o.x = __builtin_amdgcn_perm(w1[ci >> 1], w0[ci >> 1], sel);
      o.y = __builtin_amdgcn_perm(w3[ci >> 1], w2[ci >> 1], sel);
      *reinterpret_cast<uint2*>(lds + swz(row, chunk) + sub) = o;
    }
  }
};

// ------------------------------------------------------------------------------------------- forward
// BIAS: the per-key bias (in units of the raw dot product, i.e. divided by `scale`) is staged next to the key tile
// and becomes the INITIAL value of the score accumulator, so the softmax code is the same with and without it.
template <int DH>
struct HeadGeom {
  static constexpr int NB = (DH + 63) / 64;   // 64-wide LDS blocks per image
  static constexpr int NKS = (DH + 15) / 16;  // contraction steps over the head dim (k = 16 per 32x32x16 MFMA)
  static constexpr int NDT = (DH + 31) / 32;  // 32-row output tiles over the head dim
};
__device__ __forceinline__ uint4 load16_or_zero(const bf16_t* p, bool ok) {
  return ok ? *reinterpret_cast<const uint4*>(p) : uint4{0u, 0u, 0u, 0u};
}
extern __shared__ __attribute__((aligned(16))) char dyn_smem[];

template <bool BIAS, int DH, bool ROPE = false>
__global__ void __launch_bounds__(256, 2) attn_fwd_mfma(const MArgs a) {
  using G = HeadGeom<DH>;
  constexpr int NB = G::NB, NKS = G::NKS, NDT = G::NDT;
  constexpr int STAGE = 2 * NB * 8192;  // K blocks | V^T blocks
  // 2 stages; one 64-wide block fits the static limit, wider heads take dynamic LDS
  __shared__ __attribute__((aligned(16))) char st_smem[NB == 1 ? 2 * STAGE : 16];
  __shared__ __attribute__((aligned(16))) float kbs[BIAS ? 128 : 4];
  char* const smem = NB == 1 ? st_smem : dyn_smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  // 1-D grid, XCD-aware: workgroups w and w+8 share an XCD (round-robin dispatch), so every XCD takes a contiguous
  // run of logical ids (query tile fastest) and the query tiles of one head read its K / V through the same L2.
  const int ntq = (a.T + 127) / 128, total = ntq * a.B * a.H;
  int lid;
  {
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int q = total >> 3, rm = total & 7;
    lid = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + loc;
  }
  const int bh = lid / ntq, b = bh / a.H, hd = bh - b * a.H;
  const int q0 = (lid - bh * ntq) * 128 + wave * 32;
  const bool active = q0 < a.T;  // wave-uniform
  const bf16_t* qb = a.q + (int64_t)b * a.T * a.ldq + hd * DH;
  const bf16_t* kb = a.k + (int64_t)b * a.Tk * a.ldk + hd * DH;
  const bf16_t* vb = a.v + (int64_t)b * a.Tk * a.ldv + hd * DH;
  const float c = a.scale * 1.4426950408889634f;

  uint4 qf[NKS];
  if (active) {
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      qf[s] = load16_or_zero(qb + (int64_t)(q0 + r) * a.ldq + 16 * s + 8 * h, 16 * s + 8 * h < DH);
      if constexpr (ROPE) qf[s] = rope8(qf[s], a.rope + (int64_t)(q0 + r) * a.ldt + hd * DH + 16 * s + 8 * h);
    }
  }
  f32x16 o[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) o[dt] = f32x16{};
  float m = -INFINITY, l = 0.f;

  uint4 kreg[NB][2];
  TStage vreg[NB];
  float kbreg = 0.f;
  const float inv_scale = 1.f / a.scale;
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        int row = k0 + (tid >> 3) + 32 * p;
        if (row >= a.Tk) row = a.Tk - 1;  // keys past Tk: clamped, masked below
        const int col = 64 * blk + 8 * (tid & 7);
        kreg[blk][p] = load16_or_zero(kb + (int64_t)row * a.ldk + col, col < DH);
        if constexpr (ROPE) kreg[blk][p] = rope8(kreg[blk][p], a.rope + (int64_t)row * a.ldt + hd * DH + col);
      }
      vreg[blk].load(vb + 64 * blk, a.ldv, k0, tid, a.Tk, DH - 64 * blk);
    }
    if constexpr (BIAS)
      if (tid < 64) kbreg = a.kbias[(int64_t)b * a.Tk + min(k0 + tid, a.Tk - 1)] * inv_scale;
  };
  auto store_tile = [&](int stage) {
    char* st = smem + stage * STAGE;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
      for (int p = 0; p < 2; ++p)
        *reinterpret_cast<uint4*>(st + blk * 8192 + swz((tid >> 3) + 32 * p, tid & 7)) = kreg[blk][p];
      vreg[blk].store(st + (NB + blk) * 8192, tid);
    }
    if constexpr (BIAS)
      if (tid < 64) kbs[stage * 64 + tid] = kbreg;
  };

  const int nt = (a.Tk + 63) / 64;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const char* Ks = smem + (t & 1) * STAGE;
    const char* Vs = Ks + NB * 8192;
    if (t + 1 < nt) load_tile((t + 1) * 64);
    if (active) {
      f32x16 s[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        s[kt] = f32x16{};
        if constexpr (BIAS) {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 b4 = load4(kbs + (t & 1) * 64 + 32 * kt + 8 * g4 + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[kt][4 * g4 + e] = b4[e];
          }
        }
#pragma unroll
        for (int ss = 0; ss < NKS; ++ss) {
          uint4 kf = *reinterpret_cast<const uint4*>(Ks + (ss >> 2) * 8192 + swz(32 * kt + r, 2 * (ss & 3) + h));
          s[kt] = mfma32(kf, qf[ss], s[kt]);
        }
      }
      if ((t + 1) * 64 > a.Tk) {  // ragged last key tile (cross-attention, Tk = 77): keys past Tk score -inf
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (t * 64 + 32 * kt + (i & 3) + 8 * (i >> 2) + 4 * h >= a.Tk) s[kt][i] = -INFINITY;
      }
      float mx = s[0][0];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[kt][i]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m, mx);
      const float alpha = fexp2((m - mn) * c);
      const float mc = mn * c;
      float ls = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float p = fexp2(s[kt][i] * c - mc);
          s[kt][i] = p;
          ls += p;
        }
      l = l * alpha + ls;
      m = mn;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const uint4 pf = pack8(s[kt], s2);
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) {
            uint4 vf = *reinterpret_cast<const uint4*>(Vs + (dt >> 1) * 8192 +
                                                       swz(32 * (dt & 1) + r, 2 * (2 * kt + s2) + h));
            o[dt] = mfma32(vf, pf, o[dt]);
          }
        }
    }
    if (t + 1 < nt) store_tile((t + 1) & 1);
    __syncthreads();
  }
  if (active) {
    const float lt = l + __shfl_xor(l, 32, 64);
    const float inv = 1.f / lt;
    bf16_t* ob = a.out + (int64_t)b * a.T * a.ldo + hd * DH + (int64_t)(q0 + r) * a.ldo;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int d0 = 32 * dt + 8 * g4 + 4 * h;
        if (DH % 32 == 0 || d0 < DH)
          store4(ob + d0, f32x4{o[dt][4 * g4] * inv, o[dt][4 * g4 + 1] * inv, o[dt][4 * g4 + 2] * inv,
                                o[dt][4 * g4 + 3] * inv});
      }
    if (h == 0) a.lse[((int64_t)b * a.H + hd) * a.T + q0 + r] = m * a.scale + __logf(lt);
  }
}

// ------------------------------------------------------------------------------------------- backward
// LDS map, NB = blocks of 64 head-dim columns: 2 stages x (Qs | dOs | QTs | dOTs, NB x 8 KB each), then
// [2][64] lse*log2e and [2][64] delta, then (DQ variant, NB = 1) the dS and K^T images
constexpr int bwd_stage(int nb) { return 32768 * nb; }
constexpr int bwd_off_lse(int nb) { return 2 * bwd_stage(nb); }
constexpr int BWD_OFF_DS = bwd_off_lse(1) + 1024;   // dS image  [64 q][256 keys] bf16 (512-B rows)
constexpr int BWD_OFF_KT = BWD_OFF_DS + 32768;      // K^T image [64 d][256 keys] bf16 (512-B rows)
constexpr int BWD_LDS = BWD_OFF_KT + 32768;
constexpr int bwd_lds_kv(int nb) { return bwd_off_lse(nb) + 1024; }

// 512-byte rows (256 bf16): 16-B chunk c of row r lives at chunk c ^ (r & 15) -> the 16x16x32 fragment reads
// (16 lanes = 16 consecutive rows, same chunk) hit 16 distinct slots of the 256-B bank row.
__device__ __forceinline__ int off512(int row, int chunk) { return row * 512 + ((chunk ^ (row & 15)) << 4); }

// DQ = true (T <= 256): one workgroup per head owns all keys and also forms dQ (phase 2).
// DQ = false (longer sequences): a workgroup owns one block of 256 keys of a head, walks all query tiles and produces
// only dK / dV for its keys (no dS image, no K^T image: 65 KB of LDS, two workgroups per CU); dQ comes from
// attn_bwd_dq_mfma.  The key blocks of a head sit next to each other in an XCD-aware 1-D grid (they read the same Q / dO).
// BIAS (key-block variant only): the lane's key bias / scale is the initial value of the S accumulator.
template <bool DQ, bool BIAS, int DH, bool ROPE = false>
__global__ void __launch_bounds__(512, 2) attn_bwd_mfma(const MArgs a) {
  static_assert(!(DQ && BIAS), "a key bias runs through the key-block variant");
  static_assert(!ROPE || (DQ && DH == 64), "the RoPE variant is the one-kernel head-dim-64 case");
  static_assert(!DQ || DH == 64, "the one-kernel variant is the head-dim-64 case");
  using G = HeadGeom<DH>;
  constexpr int NB = G::NB, NKS = G::NKS, NDT = G::NDT;
  constexpr int BWD_STAGE = bwd_stage(NB), BWD_OFF_LSE = bwd_off_lse(NB);
  char* const smem = dyn_smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  int bh, kblk = 0;
  if constexpr (DQ) {
    bh = blockIdx.x;
  } else {
    const int nkb = (a.Tk + 255) / 256, total = nkb * a.B * a.H;
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int q = total >> 3, rm = total & 7;
    const int lid = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + loc;
    bh = lid / nkb;
    kblk = lid - bh * nkb;
  }
  const int b = bh / a.H, hd = bh - b * a.H;
  const int k0 = kblk * 256 + wave * 32;
  const bool active = k0 < a.Tk;  // wave-uniform
  const bool kvalid = k0 + r < a.Tk;  // this lane's key exists (ragged last block of cross-attention)
  const bf16_t* qb = a.q + (int64_t)b * a.T * a.ldq + hd * DH;
  const bf16_t* kb = a.k + (int64_t)b * a.Tk * a.ldk + hd * DH;
  const bf16_t* vb = a.v + (int64_t)b * a.Tk * a.ldv + hd * DH;
  const bf16_t* gb = a.dO + (int64_t)b * a.T * a.ldo + hd * DH;
  const bf16_t* ob = a.o + (int64_t)b * a.T * a.ldo + hd * DH;
  const float* lseb = a.lse + ((int64_t)b * a.H + hd) * a.T;
  const float c = a.scale * 1.4426950408889634f;
  char* dsimg = smem + BWD_OFF_DS;
  char* ktimg = smem + BWD_OFF_KT;

  // K^T image for the dQ product: [d][key], natural key order, built once (4 keys x 4 d per thread-step)
  if constexpr (DQ)
  for (int kt64 = tid >> 8; kt64 * 64 < a.T; kt64 += 2) {
    TStage ks;
    const int t256 = tid & 255;
    ks.load(kb, a.ldk, kt64 * 64, t256, 0x7fffffff, 64, ROPE ? a.rope + hd * DH : nullptr, a.ldt);
    const int cg = t256 & 15, rq = t256 >> 4;
    const int key = kt64 * 64 + 4 * rq;
    const unsigned* w0 = reinterpret_cast<const unsigned*>(&ks.r[0]);
    const unsigned* w1 = reinterpret_cast<const unsigned*>(&ks.r[1]);
    const unsigned* w2 = reinterpret_cast<const unsigned*>(&ks.r[2]);
    const unsigned* w3 = reinterpret_cast<const unsigned*>(&ks.r[3]);
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      const int row = 4 * cg + ci;
      const unsigned sel = (ci & 1) ? 0x07060302u : 0x05040100u;
      uint2 o;
      o.x = __builtin_amdgcn_perm(w1[ci >> 1], w0[ci >> 1], sel);
      o.y = __builtin_amdgcn_perm(w3[ci >> 1], w2[ci >> 1], sel);
      *reinterpret_cast<uint2*>(ktimg + off512(row, key >> 3) + 8 * ((key >> 2) & 1)) = o;
    }
  }

  // per-wave static operands: K and V rows of this wave's 32 keys (B operands of S and dP)
  uint4 kf[NKS], vf[NKS];
  if (active) {
    const int krow = kvalid ? k0 + r : a.Tk - 1;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      kf[s] = load16_or_zero(kb + (int64_t)krow * a.ldk + 16 * s + 8 * h, 16 * s + 8 * h < DH);
      if constexpr (ROPE) kf[s] = rope8(kf[s], a.rope + (int64_t)krow * a.ldt + hd * DH + 16 * s + 8 * h);
      vf[s] = load16_or_zero(vb + (int64_t)krow * a.ldv + 16 * s + 8 * h, 16 * s + 8 * h < DH);
    }
  }
  f32x16 dkT[NDT], dvT[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) dkT[dt] = dvT[dt] = f32x16{};
  float kbr = 0.f;
  if constexpr (BIAS)
    if (active) kbr = a.kbias[(int64_t)b * a.Tk + (kvalid ? k0 + r : a.Tk - 1)] / a.scale;
  (void)kbr;

  // staging registers: row-major Q and dO (one 16-B chunk each), transposed Q (threads 0-255) or dO (256-511)
  // delta[q] = sum_d dO[q][d] * O[q][d] is computed here from the staged dO chunk and the matching O chunk (8 lanes
  // per row, three shuffles) instead of by a separate pass over O and dO
  uint4 qreg[NB], greg[NB], oreg[NB];
  TStage treg[NB];
  float lreg = 0.f;
  const float inv_scale = 1.f / a.scale;
  auto load_tile = [&](int q0) {
    const int row = tid >> 3, ch = tid & 7;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      const int col = 64 * blk + 8 * ch;
      qreg[blk] = load16_or_zero(qb + (int64_t)(q0 + row) * a.ldq + col, col < DH);
      if constexpr (ROPE) qreg[blk] = rope8(qreg[blk], a.rope + (int64_t)(q0 + row) * a.ldt + hd * DH + col);
      greg[blk] = load16_or_zero(gb + (int64_t)(q0 + row) * a.ldo + col, col < DH);
      oreg[blk] = load16_or_zero(ob + (int64_t)(q0 + row) * a.ldo + col, col < DH);
      if (tid < 256) treg[blk].load(qb + 64 * blk, a.ldq, q0, tid, 0x7fffffff, DH - 64 * blk,
                                    ROPE ? a.rope + hd * DH + 64 * blk : nullptr, a.ldt);
      else treg[blk].load(gb + 64 * blk, a.ldo, q0, tid - 256, 0x7fffffff, DH - 64 * blk);
    }
    if (tid < 64) lreg = -lseb[q0 + tid] * inv_scale;  // row constant of S, in units of the raw dot product
  };
  auto store_tile = [&](int stage) {
    char* st = smem + stage * BWD_STAGE;
    const int row = tid >> 3, ch = tid & 7;
    float ds = 0.f;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      *reinterpret_cast<uint4*>(st + blk * 8192 + swz(row, ch)) = qreg[blk];
      *reinterpret_cast<uint4*>(st + (NB + blk) * 8192 + swz(row, ch)) = greg[blk];
      if (tid < 256) treg[blk].store(st + (2 * NB + blk) * 8192, tid);
      else treg[blk].store(st + (3 * NB + blk) * 8192, tid - 256);
      const bf16x8 gv = *reinterpret_cast<const bf16x8*>(&greg[blk]), ov = *reinterpret_cast<const bf16x8*>(&oreg[blk]);
#pragma unroll
      for (int j = 0; j < 8; ++j) ds += (float)gv[j] * (float)ov[j];
    }
    float* ls = reinterpret_cast<float*>(smem + BWD_OFF_LSE) + stage * 64;
    float* dl = reinterpret_cast<float*>(smem + BWD_OFF_LSE + 512) + stage * 64;
    if (tid < 64) ls[tid] = lreg;
    ds += __shfl_xor(ds, 1, 64);
    ds += __shfl_xor(ds, 2, 64);
    ds += __shfl_xor(ds, 4, 64);
    if (ch == 0) dl[row] = -ds;
  };

  const int nt = a.T / 64;
  const int nk32 = a.T / 32;
  const int fr = lane & 15, fq = lane >> 4;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int stage = t & 1;
    const char* Qs = smem + stage * BWD_STAGE;
    const char* Gs = Qs + NB * 8192;
    const char* QTs = Qs + 2 * NB * 8192;
    const char* GTs = Qs + 3 * NB * 8192;
    const float* ls = reinterpret_cast<const float*>(smem + BWD_OFF_LSE) + stage * 64;
    const float* dl = reinterpret_cast<const float*>(smem + BWD_OFF_LSE + 512) + stage * 64;
    if (t + 1 < nt) load_tile((t + 1) * 64);
    // ---- phase 1: this wave's 32 keys against the 64 staged query rows
    if (active) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        // the row constants -lse / scale and -delta are the INITIAL accumulators of S and dP (one LDS read each, no
        // subtraction after the chains): p = exp2(c * S'), dS = p * dP'
        f32x16 S, dP;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 l4 = load4(ls + 32 * sub + 8 * g4 + 4 * h);
          const f32x4 d4 = load4(dl + 32 * sub + 8 * g4 + 4 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            S[4 * g4 + e] = BIAS ? l4[e] + kbr : l4[e];
            dP[4 * g4 + e] = d4[e];
          }
        }
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
          uint4 qa = *reinterpret_cast<const uint4*>(Qs + (s >> 2) * 8192 + swz(32 * sub + r, 2 * (s & 3) + h));
          uint4 ga = *reinterpret_cast<const uint4*>(Gs + (s >> 2) * 8192 + swz(32 * sub + r, 2 * (s & 3) + h));
          S = mfma32(qa, kf[s], S);
          dP = mfma32(ga, vf[s], dP);
        }
        // P and dS in place (rows = q in registers, cols = key on lanes)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float p = fexp2(S[i] * c);
          if constexpr (!DQ) p = kvalid ? p : 0.f;  // (self-attention with T <= 256: every key of an active wave exists)
          S[i] = p;
          dP[i] *= p;
        }
        // dS is rounded to bf16 ONCE (8 packed pairs): the pairs feed both the shared image and the dK operand
        bf16x2 dsp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) dsp[j] = bf16x2{(bf16_t)dP[2 * j], (bf16_t)dP[2 * j + 1]};
        // dS -> shared image [q][key] (the one transpose: dQ contracts over the lane index)
        if constexpr (DQ) {
          const int key = k0 + r;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int qrow = 32 * sub + (i & 3) + 8 * (i >> 2) + 4 * h;
            *reinterpret_cast<bf16_t*>(dsimg + off512(qrow, key >> 3) + (key & 7) * 2) = dsp[i >> 1][i & 1];
          }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const uint4 pf = pack8(S, s2);
          uint4 dsf;
          dsf.x = *reinterpret_cast<const unsigned*>(&dsp[4 * s2]);
          dsf.y = *reinterpret_cast<const unsigned*>(&dsp[4 * s2 + 1]);
          dsf.z = *reinterpret_cast<const unsigned*>(&dsp[4 * s2 + 2]);
          dsf.w = *reinterpret_cast<const unsigned*>(&dsp[4 * s2 + 3]);
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) {
            const int off = (dt >> 1) * 8192 + swz(32 * (dt & 1) + r, 2 * (2 * sub + s2) + h);
            uint4 gt = *reinterpret_cast<const uint4*>(GTs + off);
            uint4 qt = *reinterpret_cast<const uint4*>(QTs + off);
            dvT[dt] = mfma32(gt, pf, dvT[dt]);
            dkT[dt] = mfma32(qt, dsf, dkT[dt]);
          }
        }
      }
    }
    if constexpr (DQ) __syncthreads();  // dS image complete for all keys
    // ---- phase 2: dQ^T[d][q] = K^T[d][key] . dS^T[key][q]; wave w owns q-block (w&3) and d-blocks 2(w>>2), +1
    // Both d-blocks advance together (two independent MFMA chains sharing the dS fragment), two key steps per
    // iteration with all six fragment reads issued first; the swizzled address of key step kk is base ^ (kk << 6)
    // (chunk = 4 kk + fq only flips address bits 6-8), so the loop carries no address arithmetic.  The first version
    // ran one dependent read -> MFMA chain per d-block with the full swizzle per step.
    if constexpr (DQ) {
      const int qblk = wave & 3, db0 = 2 * (wave >> 2);
      bf16_t* dqrow = a.dq + (int64_t)b * a.T * a.ldq + hd * 64 + (int64_t)(t * 64 + 16 * qblk + fr) * a.ldq;
      const int k0o = off512(16 * db0 + fr, fq), k1o = off512(16 * db0 + 16 + fr, fq), dso = off512(16 * qblk + fr, fq);
      auto frag = [](const char* img, int off, int kk) { return *reinterpret_cast<const uint4*>(img + (off ^ (kk << 6))); };
      auto mma16 = [](const uint4& x, const uint4& y, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&x),
                                                       *reinterpret_cast<const bf16x8*>(&y), c, 0, 0, 0);
      };
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      for (int kk = 0; kk < nk32; kk += 2) {  // T is a multiple of 64: nk32 is even
        const uint4 ka0 = frag(ktimg, k0o, kk), kb0 = frag(ktimg, k1o, kk), d0 = frag(dsimg, dso, kk);
        const uint4 ka1 = frag(ktimg, k0o, kk + 1), kb1 = frag(ktimg, k1o, kk + 1), d1 = frag(dsimg, dso, kk + 1);
        acc0 = mma16(ka0, d0, acc0);
        acc1 = mma16(kb0, d0, acc1);
        acc0 = mma16(ka1, d1, acc0);
        acc1 = mma16(kb1, d1, acc1);
      }
      // D[row = d = 16db + 4fq + reg][col = q = fr]: 4 consecutive d of one query row per lane
      store4(dqrow + 16 * db0 + 4 * fq, acc0 * a.scale);
      store4(dqrow + 16 * db0 + 16 + 4 * fq, acc1 * a.scale);
    }
    if (t + 1 < nt) store_tile((t + 1) & 1);
    __syncthreads();  // dS image free again; next stage visible
  }
  if constexpr (DQ && DH == 64) {
    // One-kernel variant (T <= 256): the accumulators hold 4 consecutive d of one key per lane -- stored directly that is 16
    // contiguous bytes per row and instruction, eight partial writes per 128-byte dK / dV row.  Every wave passes its
    // [32 keys][64 d] tiles through its own 8 KB of the (now idle) LDS and stores whole rows: 8 lanes x 16 B per key.
    if (ROPE || a.ldt != -99) {  // (non-RoPE launches: ldt = -99 switches back to the direct stores, UWU_ATTN_ROWSTORE=0)
      char* mine = smem + wave * 8192;  // [2 tensors][32 keys][128 B]
      if (active) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int d0 = 32 * dt + 8 * g4 + 4 * h;
            store4(reinterpret_cast<bf16_t*>(mine + r * 128) + d0,
                   f32x4{dkT[dt][4 * g4] * a.scale, dkT[dt][4 * g4 + 1] * a.scale, dkT[dt][4 * g4 + 2] * a.scale,
                         dkT[dt][4 * g4 + 3] * a.scale});
            store4(reinterpret_cast<bf16_t*>(mine + 4096 + r * 128) + d0,
                   f32x4{dvT[dt][4 * g4], dvT[dt][4 * g4 + 1], dvT[dt][4 * g4 + 2], dvT[dt][4 * g4 + 3]});
          }
        // (same wave wrote and reads: no barrier; the compiler orders the LDS accesses)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int key = 8 * p + (lane >> 3), ch = lane & 7;
          const uint4 kx = *reinterpret_cast<const uint4*>(mine + key * 128 + ch * 16);
          const uint4 vx = *reinterpret_cast<const uint4*>(mine + 4096 + key * 128 + ch * 16);
          *reinterpret_cast<uint4*>(a.dk + (int64_t)b * a.Tk * a.ldk + hd * DH + (int64_t)(k0 + key) * a.ldk + 8 * ch) = kx;
          *reinterpret_cast<uint4*>(a.dv + (int64_t)b * a.Tk * a.ldv + hd * DH + (int64_t)(k0 + key) * a.ldv + 8 * ch) = vx;
        }
      }
      return;
    }
  }
  if (active && kvalid) {
    bf16_t* dkb = a.dk + (int64_t)b * a.Tk * a.ldk + hd * DH + (int64_t)(k0 + r) * a.ldk;
    bf16_t* dvb = a.dv + (int64_t)b * a.Tk * a.ldv + hd * DH + (int64_t)(k0 + r) * a.ldv;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int d0 = 32 * dt + 8 * g4 + 4 * h;
        if (DH % 32 != 0 && d0 >= DH) continue;
        store4(dkb + d0, f32x4{dkT[dt][4 * g4] * a.scale, dkT[dt][4 * g4 + 1] * a.scale,
                               dkT[dt][4 * g4 + 2] * a.scale, dkT[dt][4 * g4 + 3] * a.scale});
        store4(dvb + d0, f32x4{dvT[dt][4 * g4], dvT[dt][4 * g4 + 1], dvT[dt][4 * g4 + 2], dvT[dt][4 * g4 + 3]});
      }
  }
}

// ------------------------------------------------------------------------------------------- dQ, long sequences
// Query-owner pass (structure of the forward kernel): 4 waves x 32 query rows, 64-key tiles of K (row-major), V
// (row-major) and K^T (transposing stager) through LDS.  With the query on the lane,
//   S^T[key][q] = K . Q^T,  dP^T[key][q] = V . dO^T   (Q / dO fragments of the wave's rows live in registers)
//   dS^T = P^T (dP^T - delta[q]),  P^T = exp2(c S^T - lse[q])      (lse / delta are per-lane scalars)
//   dQ^T[d][q] += K^T[d][key] . dS^T[key][q]              (dS^T accumulator reused as the B operand, as P^T in forward)
// so dQ accumulates over all key tiles in the wave's own registers: no atomics, no cross-workgroup reduction.
// delta[q] = sum_d dO O is formed from the wave's own dO fragments and the matching O chunks.
template <bool BIAS, int DH>
__global__ void __launch_bounds__(256, 2) attn_bwd_dq_mfma(const MArgs a) {
  using G = HeadGeom<DH>;
  constexpr int NB = G::NB, NKS = G::NKS, NDT = G::NDT;
  constexpr int STAGE = 3 * NB * 8192;  // K blocks | V blocks | K^T blocks
  __shared__ __attribute__((aligned(16))) char st_smem[NB == 1 ? 2 * STAGE : 16];
  __shared__ __attribute__((aligned(16))) float kbs[BIAS ? 128 : 4];
  char* const smem = NB == 1 ? st_smem : dyn_smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int ntq = (a.T + 127) / 128, total = ntq * a.B * a.H;
  int lid;
  {
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int q = total >> 3, rm = total & 7;
    lid = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + loc;
  }
  const int bh = lid / ntq, b = bh / a.H, hd = bh - b * a.H;
  const int q0 = (lid - bh * ntq) * 128 + wave * 32;
  const bool active = q0 < a.T;  // wave-uniform
  const bf16_t* qb = a.q + (int64_t)b * a.T * a.ldq + hd * DH;
  const bf16_t* kb = a.k + (int64_t)b * a.Tk * a.ldk + hd * DH;
  const bf16_t* vb = a.v + (int64_t)b * a.Tk * a.ldv + hd * DH;
  const bf16_t* gb = a.dO + (int64_t)b * a.T * a.ldo + hd * DH;
  const bf16_t* ob = a.o + (int64_t)b * a.T * a.ldo + hd * DH;
  const float c = a.scale * 1.4426950408889634f;

  uint4 qf[NKS], gf[NKS];
  float lq = 0.f, dl = 0.f;
  if (active) {
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      const bool ok = 16 * s + 8 * h < DH;
      qf[s] = load16_or_zero(qb + (int64_t)(q0 + r) * a.ldq + 16 * s + 8 * h, ok);
      gf[s] = load16_or_zero(gb + (int64_t)(q0 + r) * a.ldo + 16 * s + 8 * h, ok);
      const uint4 of = load16_or_zero(ob + (int64_t)(q0 + r) * a.ldo + 16 * s + 8 * h, ok);
      const bf16x8 gv = *reinterpret_cast<const bf16x8*>(&gf[s]), ov = *reinterpret_cast<const bf16x8*>(&of);
#pragma unroll
      for (int j = 0; j < 8; ++j) dl += (float)gv[j] * (float)ov[j];
    }
    dl += __shfl_xor(dl, 32, 64);  // the two lane halves hold the two halves of the row
    lq = a.lse[((int64_t)b * a.H + hd) * a.T + q0 + r] * 1.4426950408889634f;
  }
  f32x16 dq[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) dq[dt] = f32x16{};

  uint4 kreg[NB][2], vreg[NB][2];
  TStage ktreg[NB];
  float kbreg = 0.f;
  const float inv_scale = 1.f / a.scale;
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        int row = k0 + (tid >> 3) + 32 * p;
        if (row >= a.Tk) row = a.Tk - 1;  // keys past Tk: clamped, their dS is zeroed below
        const int col = 64 * blk + 8 * (tid & 7);
        kreg[blk][p] = load16_or_zero(kb + (int64_t)row * a.ldk + col, col < DH);
        vreg[blk][p] = load16_or_zero(vb + (int64_t)row * a.ldv + col, col < DH);
      }
      ktreg[blk].load(kb + 64 * blk, a.ldk, k0, tid, a.Tk, DH - 64 * blk);
    }
    if constexpr (BIAS)
      if (tid < 64) kbreg = a.kbias[(int64_t)b * a.Tk + min(k0 + tid, a.Tk - 1)] * inv_scale;
  };
  auto store_tile = [&](int stage) {
    char* st = smem + stage * STAGE;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        *reinterpret_cast<uint4*>(st + blk * 8192 + swz((tid >> 3) + 32 * p, tid & 7)) = kreg[blk][p];
        *reinterpret_cast<uint4*>(st + (NB + blk) * 8192 + swz((tid >> 3) + 32 * p, tid & 7)) = vreg[blk][p];
      }
      ktreg[blk].store(st + (2 * NB + blk) * 8192, tid);
    }
    if constexpr (BIAS)
      if (tid < 64) kbs[stage * 64 + tid] = kbreg;
  };

  const int nt = (a.Tk + 63) / 64;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const char* Ks = smem + (t & 1) * STAGE;
    const char* Vs = Ks + NB * 8192;
    const char* KTs = Ks + 2 * NB * 8192;
    if (t + 1 < nt) load_tile((t + 1) * 64);
    if (active) {
      f32x16 s[2], dp[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        s[kt] = f32x16{};
        dp[kt] = f32x16{};
        if constexpr (BIAS) {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 b4 = load4(kbs + (t & 1) * 64 + 32 * kt + 8 * g4 + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[kt][4 * g4 + e] = b4[e];
          }
        }
#pragma unroll
        for (int ss = 0; ss < NKS; ++ss) {
          const int off = (ss >> 2) * 8192 + swz(32 * kt + r, 2 * (ss & 3) + h);
          const uint4 kf = *reinterpret_cast<const uint4*>(Ks + off);
          const uint4 vf = *reinterpret_cast<const uint4*>(Vs + off);
          s[kt] = mfma32(kf, qf[ss], s[kt]);
          dp[kt] = mfma32(vf, gf[ss], dp[kt]);
        }
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float p = fexp2(s[kt][i] * c - lq);
          s[kt][i] = p * (dp[kt][i] - dl);
        }
      if ((t + 1) * 64 > a.Tk) {  // ragged last key tile
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (t * 64 + 32 * kt + (i & 3) + 8 * (i >> 2) + 4 * h >= a.Tk) s[kt][i] = 0.f;
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const uint4 dsf = pack8(s[kt], s2);
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) {
            const uint4 ktf = *reinterpret_cast<const uint4*>(KTs + (dt >> 1) * 8192 +
                                                              swz(32 * (dt & 1) + r, 2 * (2 * kt + s2) + h));
            dq[dt] = mfma32(ktf, dsf, dq[dt]);
          }
        }
    }
    if (t + 1 < nt) store_tile((t + 1) & 1);
    __syncthreads();
  }
  if (active) {
    bf16_t* dqb = a.dq + (int64_t)b * a.T * a.ldq + hd * DH + (int64_t)(q0 + r) * a.ldq;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int d0 = 32 * dt + 8 * g4 + 4 * h;
        if (DH % 32 != 0 && d0 >= DH) continue;
        store4(dqb + d0, f32x4{dq[dt][4 * g4] * a.scale, dq[dt][4 * g4 + 1] * a.scale, dq[dt][4 * g4 + 2] * a.scale,
                               dq[dt][4 * g4 + 3] * a.scale});
      }
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------- host side
// head dim 64 / 72 / 128, whole 64-row query tiles; any number of keys (ragged key tiles are masked: cross-attention,
// Tk = 77)
bool uwu_attn_mfma_fwd_ok(int Tq, int Tk, int d, int ldq, int ldk, int ldv, int ldo) {
  return (d == 64 || d == 72 || d == 128) && Tk >= 1 && Tq % 64 == 0 && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 &&
         ldo % 8 == 0;
}
bool uwu_attn_mfma_bwd_ok(int Tq, int Tk, int d, int ldq, int ldk, int ldv, int ldo) {
  return uwu_attn_mfma_fwd_ok(Tq, Tk, d, ldq, ldk, ldv, ldo);
}

namespace {

template <typename K>
void allow_lds(K kernel, int bytes) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <int DH>
void launch_fwd(const MArgs& a, hipStream_t st) {
  constexpr int NB = HeadGeom<DH>::NB;
  constexpr int lds = NB == 1 ? 0 : 2 * 2 * NB * 8192;  // wider heads: the two stages live in dynamic LDS
  static bool once = false;
  if (!once && lds) {
    allow_lds(attn_fwd_mfma<true, DH>, lds);
    allow_lds(attn_fwd_mfma<false, DH>, lds);
  }
  once = true;
  const dim3 grid(((a.T + 127) / 128) * a.B * a.H);
  if (a.kbias) hipLaunchKernelGGL((attn_fwd_mfma<true, DH>), grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL((attn_fwd_mfma<false, DH>), grid, dim3(256), lds, st, a);
}

template <int DH>
void launch_bwd(const MArgs& a, hipStream_t st) {
  constexpr int NB = HeadGeom<DH>::NB;
  constexpr int lds_kv = bwd_lds_kv(NB);                  // key-block variant: the two staging stages + lse / delta
  constexpr int lds_dq = NB == 1 ? 0 : 2 * 3 * NB * 8192;
  static bool once = false;
  if (!once) {
    if constexpr (DH == 64) allow_lds(attn_bwd_mfma<true, false, 64>, BWD_LDS);
    allow_lds(attn_bwd_mfma<false, false, DH>, lds_kv);
    allow_lds(attn_bwd_mfma<false, true, DH>, lds_kv);
    if (lds_dq) {
      allow_lds(attn_bwd_dq_mfma<true, DH>, lds_dq);
      allow_lds(attn_bwd_dq_mfma<false, DH>, lds_dq);
    }
  }
  once = true;
  const dim3 gkv(a.B * a.H * ((a.Tk + 255) / 256)), gq(a.B * a.H * ((a.T + 127) / 128));
  if (a.kbias) {  // biased scores: always the two-kernel form
    hipLaunchKernelGGL((attn_bwd_mfma<false, true, DH>), gkv, dim3(512), lds_kv, st, a);
    hipLaunchKernelGGL((attn_bwd_dq_mfma<true, DH>), gq, dim3(256), lds_dq, st, a);
    return;
  }
  if constexpr (DH == 64) {
    if (a.T == a.Tk && a.T <= 256) {
      hipLaunchKernelGGL((attn_bwd_mfma<true, false, 64>), dim3(a.B * a.H), dim3(512), BWD_LDS, st, a);
      return;
    }
  }
  // dK / dV per block of 256 keys, dQ per tile of 128 queries
  hipLaunchKernelGGL((attn_bwd_mfma<false, false, DH>), gkv, dim3(512), lds_kv, st, a);
  hipLaunchKernelGGL((attn_bwd_dq_mfma<false, DH>), gq, dim3(256), lds_dq, st, a);
}

}  // namespace

// Self-attention with the axial-RoPE factors folded into the q / k staging: head dim 64, bf16,
// T == Tk <= 256 (a multiple of 64) -- the DiT shapes.  The backward returns the gradients wrt the ROTATED q', k';
// uwu_axial_rope_bwd turns them into dq, dk and the log-frequency gradients.
extern "C" int uwu_attention_rope_fwd(const void* q, const void* k, const void* v, const float* rope_tab, void* o, float* lse,
                                      int B, int T, int H, int d, int ldq, int ldk, int ldv, int ldo, int ldt, float scale,
                                      int dtype, void* stream) {
  UWU_CHECK_ARG(q && k && v && rope_tab && o && lse, "attention_rope_fwd: null pointer");
  UWU_CHECK_ARG(dtype == UWU_BF16 && d == 64 && T % 64 == 0 && T >= 64 && T <= 256 && B > 0 && H > 0 && scale > 0.f,
                "attention_rope_fwd: bf16, head dim 64, T a multiple of 64 up to 256 (got d=%d T=%d)", d, T);
  UWU_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && ldt % 4 == 0 && ldt >= H * d &&
                    (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)rope_tab) & 15) == 0,
                "attention_rope_fwd: misaligned tensor / leading dimension");
  MArgs a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.out = (bf16_t*)o; a.lse = lse;
  a.B = B; a.T = T; a.Tk = T; a.H = H; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.scale = scale;
  a.rope = rope_tab; a.ldt = ldt;
  UwuProfScope prof(stream);
  hipLaunchKernelGGL((attn_fwd_mfma<false, 64, true>), dim3(((T + 127) / 128) * B * H), dim3(256), 0, (hipStream_t)stream, a);
  prof.done(UWU_PROF_ATTN_FWD, 0, 4.0 * B * H * T * T * d, 2.0 * B * H * d * 4.0 * T);
  UWU_LAUNCH_CHECK("attention_rope_fwd");
  return UWU_OK;
}

extern "C" int uwu_attention_rope_bwd(const void* q, const void* k, const void* v, const float* rope_tab, const void* o,
                                      const void* dO, const float* lse, void* dq, void* dk, void* dv, int B, int T, int H,
                                      int d, int ldq, int ldk, int ldv, int ldo, int ldt, float scale, int dtype, void* stream) {
  UWU_CHECK_ARG(q && k && v && rope_tab && o && dO && lse && dq && dk && dv, "attention_rope_bwd: null pointer");
  UWU_CHECK_ARG(dtype == UWU_BF16 && d == 64 && T % 64 == 0 && T >= 64 && T <= 256 && B > 0 && H > 0 && scale > 0.f,
                "attention_rope_bwd: bf16, head dim 64, T a multiple of 64 up to 256 (got d=%d T=%d)", d, T);
  UWU_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && ldt % 4 == 0 && ldt >= H * d &&
                    (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)dO | (uintptr_t)dq | (uintptr_t)dk |
                      (uintptr_t)dv | (uintptr_t)rope_tab) & 15) == 0,
                "attention_rope_bwd: misaligned tensor / leading dimension");
  MArgs a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (const bf16_t*)o;
  a.dO = (const bf16_t*)dO; a.lse = const_cast<float*>(lse);
  a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv;
  a.B = B; a.T = T; a.Tk = T; a.H = H; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.scale = scale;
  a.rope = rope_tab; a.ldt = ldt;
  static bool once = false;
  if (!once) allow_lds(attn_bwd_mfma<true, false, 64, true>, BWD_LDS);
  once = true;
  UwuProfScope prof(stream);
  hipLaunchKernelGGL((attn_bwd_mfma<true, false, 64, true>), dim3(B * H), dim3(512), BWD_LDS, (hipStream_t)stream, a);
  prof.done(UWU_PROF_ATTN_BWD, 0, 10.0 * B * H * T * T * d, 2.0 * B * H * d * 8.0 * T);
  UWU_LAUNCH_CHECK("attention_rope_bwd");
  return UWU_OK;
}

bool uwu_attn_p256_fwd_ok(int nheads, int T, int Tk, int d, int ldq, int ldk, int ldv, int ldo);
int uwu_attn_p256_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int d, int ldq, int ldk, int ldv,
                      int ldo, float scale, hipStream_t st);

int uwu_attn_mfma_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const float* kbias, int B,
                      int T, int Tk, int H, int d, int ldq, int ldk, int ldv, int ldo, float scale, hipStream_t st) {
  UWU_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) == 0,
                "attention(mfma): q/k/v/o must be 16-byte aligned");
  if (!kbias && uwu_attn_p256_fwd_ok(B * H, T, Tk, d, ldq, ldk, ldv, ldo))
    return uwu_attn_p256_fwd(q, k, v, o, lse, B, H, d, ldq, ldk, ldv, ldo, scale, st);
  MArgs a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.out = (bf16_t*)o; a.lse = lse;
  a.kbias = kbias;
  a.B = B; a.T = T; a.Tk = Tk; a.H = H; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.scale = scale;
  if (d == 64) launch_fwd<64>(a, st);
  else if (d == 72) launch_fwd<72>(a, st);
  else launch_fwd<128>(a, st);
  UWU_LAUNCH_CHECK("attention_fwd(mfma)");
  return UWU_OK;
}

// T = 256, head dim 64, no key bias: the persistent LDS-DMA kernel (attention_p256.hip)
bool uwu_attn_p256_ok(int T, int Tk, int d, int ldq, int ldk, int ldv, int ldo);
int uwu_attn_p256_bwd(const void* q, const void* k, const void* v, const void* o, const void* dO, const float* lse, void* dq,
                      void* dk, void* dv, int B, int H, int d, int ldq, int ldk, int ldv, int ldo, float scale, hipStream_t st);

int uwu_attn_mfma_bwd(const void* q, const void* k, const void* v, const void* o, const void* dO, const float* lse,
                      float* delta, const float* kbias, void* dq, void* dk, void* dv, int B, int T, int Tk, int H,
                      int d, int ldq, int ldk, int ldv, int ldo, float scale, hipStream_t st) {
  UWU_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)dO | (uintptr_t)dq |
                  (uintptr_t)dk | (uintptr_t)dv) & 15) == 0,
                "attention_bwd(mfma): tensors must be 16-byte aligned");
  (void)delta;  // the row sums of dO * O are formed inside the kernels
  if (!kbias && uwu_attn_p256_ok(T, Tk, d, ldq, ldk, ldv, ldo))
    return uwu_attn_p256_bwd(q, k, v, o, dO, lse, dq, dk, dv, B, H, d, ldq, ldk, ldv, ldo, scale, st);
  MArgs a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (const bf16_t*)o;
  a.dO = (const bf16_t*)dO; a.lse = const_cast<float*>(lse); a.delta = delta;
  a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv;
  a.kbias = kbias;
  a.B = B; a.T = T; a.Tk = Tk; a.H = H; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.scale = scale;
  {
    static UwuEnv rs("UWU_ATTN_ROWSTORE");  // UWU_ATTN_ROWSTORE=0: direct 8-byte dK / dV stores (A/B comparisons)
    a.ldt = rs.get().is('0') ? -99 : 0;
  }
  if (d == 64) launch_bwd<64>(a, st);
  else if (d == 72) launch_bwd<72>(a, st);
  else launch_bwd<128>(a, st);
  UWU_LAUNCH_CHECK("attention_bwd(mfma)");
  return UWU_OK;
}
