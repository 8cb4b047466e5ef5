// DiT (adaLN-Zero) forward / backward driver: every kernel launch of one network pass is issued from this
// C++ function, so the host pays one C call per pass instead of ~25 Python round trips per layer.
// Block semantics follow the reference's ada_norm_zero branch (src/duwu/modules/rope_unet.py:306-309,
// 344-349, 393-411) and attention processor (:122-166); the call contract is diffusion.py:172-176.
//
// Precision policy: activations and GEMM operands in `dtype` (bf16 or fp32), fp32 accumulation, fp32 LN
// statistics / softmax / modulation.  The conditioning path (timestep MLP, pooled-text projection: M = B
// rows only) runs in fp32 from the fp32 master weights; the batched adaLN Linear does too in fp32 mode and at
// small batches, and takes bf16 operands with fp32 accumulation / outputs in bf16 mode at B >= 64 (mod_bf16).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/uwu_hip.h"
#include "env.h"

void uwu_set_error(const char* fmt, ...);

namespace {

// workgroups a weight-gradient GEMM aims for: two per CU (the streaming kernel runs two per CU)
constexpr int WGRAD_BLOCKS = 512;

struct Layout {
  size_t es;  // activation element size
  int64_t M, D, D3, D4, Kp, Ko;
  // fp32 conditioning path
  size_t feat, t_pre, t_h, temb, yemb, c, sc, mod;
  size_t sc16, dmod16;  // bf16 mode, B >= 64: bf16 copies of silu(c) and of d(mod) -- operands of the adaLN Linear on the bf16 MFMA path
  // token path
  size_t tok, xe;
  size_t layer0, layer_stride;
  // block recomputation (uwu_dit_desc.checkpoint): only the first `keep` bytes of a block's slab (x0 and the row statistics of
  // its first LayerNorm) are per layer; every other sub-buffer lives ONCE at scratch0 + sub and holds the block in flight
  bool ckpt;
  size_t keep, scratch0;
  // per-layer sub-offsets
  size_t o_x0, o_m1, o_r1, o_h1, o_qkv, o_lse, o_ao, o_y1, o_x1, o_m2, o_r2, o_h2, o_u, o_f, o_y2;
  size_t o_rope;                     // rope mode: this layer's factor table [T, D] fp32
  size_t o_h1t, o_aot, o_h2t, o_ft;  // fp8 mode: transposed fp8 copies of the Linear inputs (operands of the weight gradients)
  size_t xF, mF, rF, hF, otok;
  // backward scratch
  size_t dx, dy, dh, dqkv, dao, du, delta, dotok, dmod, dsc, dc, dth, dtp;
  size_t w2t;  // fc2 weight of the block in flight, transposed to [4D, D] (input gradient through the A-stationary kernel)
  size_t wsc, wsc_bytes;  // split-K scratch of the weight-gradient GEMMs
  // fp8 mode: quantised operand copies (reused by every Linear) and the per-layer fp8 weights (W and W^T)
  size_t x8, x8t, dy8, dy8t, w8, w8_layer;
  size_t total;
};

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

Layout make_layout(const uwu_dit_desc& d) {
  Layout L{};
  L.es = d.dtype == UWU_BF16 ? 2 : 4;
  L.M = (int64_t)d.B * d.T;
  L.D = d.D;
  L.D3 = 3 * (int64_t)d.D;
  L.D4 = (int64_t)d.mlp_ratio * d.D;
  L.Kp = (int64_t)d.in_ch * d.patch * d.patch;
  L.Ko = (int64_t)d.out_ch * d.patch * d.patch;
  size_t p = 0;
  auto take = [&](size_t bytes) { size_t o = p; p = al(p + bytes); return o; };
  const size_t f4 = 4;
  L.feat = take((size_t)d.B * d.freq_dim * f4);
  L.t_pre = take((size_t)d.B * d.D * f4);
  L.t_h = take((size_t)d.B * d.D * f4);
  L.temb = take((size_t)d.B * d.D * f4);
  L.yemb = take((size_t)d.B * d.D * f4);
  L.c = take((size_t)d.B * d.D * f4);
  L.sc = take((size_t)d.B * d.D * f4);
  L.mod = take((size_t)d.B * d.mod_total * f4);
  L.sc16 = take((size_t)d.B * d.D * 2);
  L.dmod16 = take((size_t)d.B * d.mod_total * 2);
  L.tok = take(L.M * L.Kp * L.es);
  L.xe = take(L.M * L.D * L.es);
  // per-layer block
  size_t q = 0;
  auto sub = [&](size_t bytes) { size_t o = q; q = al(q + bytes); return o; };
  L.o_x0 = sub(L.M * L.D * L.es);
  L.o_m1 = sub(L.M * f4);
  L.o_r1 = sub(L.M * f4);
  L.o_h1 = sub(L.M * L.D * L.es);
  L.o_qkv = sub(L.M * L.D3 * L.es);
  L.o_lse = sub((size_t)d.B * d.H * d.T * f4);
  L.o_ao = sub(L.M * L.D * L.es);
  L.o_y1 = sub(L.M * L.D * L.es);
  L.o_x1 = sub(L.M * L.D * L.es);
  L.o_m2 = sub(L.M * f4);
  L.o_r2 = sub(L.M * f4);
  L.o_h2 = sub(L.M * L.D * L.es);
  L.o_u = sub(L.M * L.D4 * L.es);
  L.o_f = sub(L.M * L.D4 * L.es);
  L.o_y2 = sub(L.M * L.D * L.es);
  if (d.rope) L.o_rope = sub((size_t)d.T * d.D * f4);
  if (d.fp8) {
    L.o_h1t = sub(L.M * L.D);
    L.o_aot = sub(L.M * L.D);
    L.o_h2t = sub(L.M * L.D);
    L.o_ft = sub(L.M * L.D4);
  }
  L.keep = L.o_h1;  // o_x0, o_m1, o_r1 come first
  L.ckpt = d.checkpoint != 0;
  L.layer_stride = L.ckpt ? L.keep : q;
  L.layer0 = p;
  p += (size_t)d.L * L.layer_stride;
  if (L.ckpt) L.scratch0 = take(q);
  L.xF = take(L.M * L.D * L.es);
  L.mF = take(L.M * f4);
  L.rF = take(L.M * f4);
  L.hF = take(L.M * L.D * L.es);
  L.otok = take(L.M * L.Ko * L.es);
  L.dx = take(L.M * L.D * L.es);
  L.dy = take(L.M * L.D * L.es);
  L.dh = take(L.M * L.D * L.es);
  L.dqkv = take(L.M * L.D3 * L.es);
  L.dao = take(L.M * L.D * L.es);
  L.du = take(L.M * L.D4 * L.es);
  L.w2t = take((size_t)L.D4 * d.D * 2);
  L.delta = take((size_t)d.B * d.H * d.T * f4);
  L.dotok = take(L.M * L.Ko * L.es);
  L.dmod = take((size_t)d.B * d.mod_total * f4);
  L.dsc = take((size_t)d.B * d.D * f4);
  L.dc = take((size_t)d.B * d.D * f4);
  L.dth = take((size_t)d.B * d.D * f4);
  L.dtp = take((size_t)d.B * d.D * f4);
  // the largest weight gradient is fc1/fc2 (D4 x D); qkv (D3 x D) needs less
  L.wsc_bytes = uwu_gemm_wgrad_scratch_bytes((int)L.D4, d.D, (int)L.M);
  for (const size_t t : {uwu_gemm_wgrad_scratch_bytes(d.D, (int)L.D4, (int)L.M),
                         uwu_gemm_wgrad_scratch_bytes((int)L.D3, d.D, (int)L.M),
                         uwu_gemm_wgrad_scratch_bytes(d.D, d.D, (int)L.M)})
    if (t > L.wsc_bytes) L.wsc_bytes = t;
  if (d.fp8) {
    for (const size_t t : {uwu_gemm_fp8_scratch_bytes((int)L.D4, d.D, (int)L.M), uwu_gemm_fp8_scratch_bytes(d.D, (int)L.D4, (int)L.M),
                           uwu_gemm_fp8_scratch_bytes((int)L.D3, d.D, (int)L.M), uwu_gemm_fp8_scratch_bytes(d.D, d.D, (int)L.M)})
      if (t > L.wsc_bytes) L.wsc_bytes = t;
  }
  L.wsc_bytes = (L.wsc_bytes + 255) & ~(size_t)255;
  // forked backward (small batches): the four weight gradients of a block may run on up to four streams, each with its own slices
  L.wsc = take(L.wsc_bytes * (L.M <= 16384 ? 4 : 1));
  if (d.fp8) {
    L.x8 = take(L.M * L.D4);
    L.x8t = 0;
    L.dy8 = take(L.M * L.D4);
    L.dy8t = take(L.M * L.D4);
    L.w8_layer = al((size_t)24 * d.D * d.D);  // qkv | qkv^T | o | o^T | fc1 | fc1^T | fc2 | fc2^T
    L.w8 = take((size_t)d.L * L.w8_layer);
  }
  L.total = p;
  return L;
}

int check_desc(const uwu_dit_desc* d) {
  if (!d) { uwu_set_error("dit: null descriptor"); return UWU_EINVAL; }
  if (d->B <= 0 || d->T <= 0 || d->D <= 0 || d->H <= 0 || d->L <= 0 || d->D % d->H || d->D % 8 ||
      d->img % d->patch || (d->img / d->patch) * (d->img / d->patch) != d->T || d->mlp_ratio <= 0 ||
      d->freq_dim <= 0 || d->freq_dim % 8) {
    uwu_set_error("dit: inconsistent shape B=%d T=%d D=%d H=%d L=%d img=%d patch=%d", d->B, d->T, d->D, d->H, d->L,
                  d->img, d->patch);
    return UWU_EINVAL;
  }
  if (d->dtype != UWU_F32 && d->dtype != UWU_BF16) { uwu_set_error("dit: bad dtype"); return UWU_EINVAL; }
  if (d->mod_total != d->L * 6 * d->D + 2 * d->D) { uwu_set_error("dit: mod_total mismatch"); return UWU_EINVAL; }
  if ((d->in_ch * d->patch * d->patch) % 8 || (d->out_ch * d->patch * d->patch) % 8 || (d->cond_dim % 8)) {
    uwu_set_error("dit: C*p*p and cond_dim must be multiples of 8");
    return UWU_EINVAL;
  }
  if (!d->w || !d->w32 || !d->pos || !d->ws) { uwu_set_error("dit: null buffer"); return UWU_EINVAL; }
  if (d->rope && (d->dtype != UWU_BF16 || d->D / d->H != 64 || d->T % 64 || d->T > 256 || !d->pos_xy)) {
    uwu_set_error("dit: the fused axial-RoPE attention needs bf16, head dim 64, T a multiple of 64 up to 256 and pos_xy");
    return UWU_EINVAL;
  }
  if (d->fp8) {
    if (d->fp8 < 0 || d->fp8 > 2 || d->dtype != UWU_BF16 || d->D % 128 || d->mlp_ratio != 4 || ((int64_t)d->B * d->T) % 128 ||
        !d->f8_scale || !d->f8_amax || !d->f8_fmt) {
      uwu_set_error("dit: fp8 mode needs bf16 activations, D %% 128 == 0, B*T %% 128 == 0, mlp_ratio 4 and the scale / amax / fmt arrays");
      return UWU_EINVAL;
    }
  }
  return UWU_OK;
}

#define RUN(expr)              \
  do {                         \
    int rc_ = (expr);          \
    if (rc_ != UWU_OK) return rc_; \
  } while (0)

// Y[M,N] = X[M,K] . W[N,K]^T (+ epilogue)
int lin_fwd(const void* X, const void* W, const float* bias, void* Y, void* Y2, int M, int N, int K, int dt, int cdt,
            int epi, void* st) {
  return uwu_gemm(X, W, Y, Y2, bias, nullptr, M, N, K, K, K, N, 0, 0, 0, dt, cdt, epi, 1, st);
}
static bool fc2_dgrad_as() {  // UWU_DIT_FC2DG_AS=0: the 256x256 kernel on the K-major weight (A/B comparisons)
  static UwuEnv on("UWU_DIT_FC2DG_AS");  // (UwuEnv: re-read after uwu_env_refresh(), so in-process A/B runs compare two paths)
  return !on.get().is('0');
}
// the fp32 conditioning Linears ([B, *] rows): matrix-vector kernels when the shape is covered (csrc/skinny.hip)
int lin_fwd32(const float* X, const float* W, const float* bias, float* Y, float* Y2, int M, int N, int K, int epi, void* st) {
  // measured at batch 16 (us, matrix-vector kernel vs 128x128-tile GEMM): 256 -> 384: 19 vs 25, 1280 -> 384: 22 vs 87; the
  // adaLN Linear (N = 28416) 53 vs 33 -- its 64 wave reductions per 4 columns cost more than the tiles; 64 rows: slower throughout
  if (M <= 16 && N <= 4096 && uwu_skinny_linear_ok(M, N, K)) return uwu_skinny_linear_fwd(X, W, bias, Y, Y2, M, N, K, epi, st);
  return lin_fwd(X, W, bias, Y, Y2, M, N, K, UWU_F32, UWU_F32, epi, st);
}
// dX[M,K] = dY[M,N] . W[N,K]   (optionally * gelu'(aux[M,K]), then colsum[K] += column sums of dX)
int lin_dgrad(const void* dY, const void* W, void* dX, const void* aux, int M, int N, int K, int dt, void* st,
              float* colsum = nullptr) {
  return uwu_gemm(dY, W, dX, colsum, nullptr, aux, M, K, N, N, K, K, K, 0, 1, dt, dt,
                  aux ? UWU_EPI_DGELU : UWU_EPI_NONE, 1, st);
}
// The batched adaLN Linear (D -> L*6*D + 2*D outputs per sample) on the bf16 MFMA path: bf16 operands (the bf16 shadow of the
// weight; silu(c) and d(mod) cast once), fp32 accumulate and fp32 outputs.  In the exact-fp32 MFMA mode (1/16 of the bf16 rate) its
// three GEMMs were 0.85 ms of the 45 ms step at B = 768 (16.5 GFLOP each).  The reference runs this Linear under bf16 autocast
// like every other one (Lightning precision "bf16-mixed", configs/demo_training_latent.yaml); the fp32 parity mode and small
// batches (where the matrix-vector kernels take the conditioning Linears) are unchanged.  UWU_DIT_MOD_BF16=0: off (A/B).
static bool mod_bf16(const uwu_dit_desc& d) {
  static UwuEnv on("UWU_DIT_MOD_BF16");
  return !on.get().is('0') && d.dtype == UWU_BF16 && d.B >= 64 && d.mod_total % 8 == 0;
}
// dX[M,K] = dY[M,N] . W[N,K] for a LONG reduction N and few rows M (the batched adaLN linear: N = L*6*D+2*D):
// split the reduction over workgroups and accumulate with fp32 atomics into a zeroed dX.
int lin_dgrad_splitk(const void* dY, const void* W, float* dX, int M, int N, int K, void* st) {
  if (hipMemsetAsync(dX, 0, (size_t)M * K * sizeof(float), (hipStream_t)st) != hipSuccess) {
    uwu_set_error("dgrad_splitk: memset failed");
    return UWU_ELAUNCH;
  }
  const int tiles = ((M + 127) / 128) * ((K + 127) / 128);
  const int ktiles = (N + 31) / 32;
  int split = (512 + tiles - 1) / tiles;
  if (split > ktiles) split = ktiles;
  return uwu_gemm(dY, W, dX, nullptr, nullptr, nullptr, M, K, N, N, K, K, 0, 0, 1, UWU_F32, UWU_F32, UWU_EPI_ACCUM, split,
                  st);
}
// dW[N,K] += dY[M,N]^T . X[M,K]   (fp32 atomics, split over the token dimension)
// dW[N(out), K(in)] += dY[M, N]^T . X[M, K] and, with db, db[N] += column sums of dY (the bias gradient)
int lin_wgrad(const void* dY, const void* X, float* dW, int M, int N, int K, int dt, void* st, void* scratch = nullptr,
              size_t scratch_bytes = 0, float* db = nullptr) {
  // bf16 goes to the streaming kernel (split-K scratch, bias gradient fused as an extra all-ones column)
  if (dt == UWU_BF16 && scratch && uwu_gemm_wgrad_scratch_bytes(N, K, M) <= scratch_bytes)
    return uwu_gemm_wgrad(dY, X, dW, db, N, K, M, N, K, K, dt, WGRAD_BLOCKS, scratch, scratch_bytes, st);
  if (db) {
    const int rc = uwu_colsum(dY, dt, M, N, N, db, 1, st);
    if (rc != UWU_OK) return rc;
  }
  const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
  const int bk = dt == UWU_BF16 ? 64 : 32;
  const int ktiles = (M + bk - 1) / bk;
  // measured sweep (tools/bench_kernels.py, UWU_WGRAD_BLOCKS): ~3 workgroups per CU for the large weight
  // matrices, ~1 per CU when only a few output tiles exist (the atomic traffic tiles*split*64 KB dominates there)
  const int target = tiles >= 16 ? 768 : 256;
  int split = (target + tiles - 1) / tiles;
  if (split > ktiles) split = ktiles;
  if (split < 1) split = 1;
  return uwu_gemm(dY, X, dW, nullptr, nullptr, nullptr, N, K, M, N, K, K, 0, 1, 1, dt, UWU_F32, UWU_EPI_ACCUM, split, st);
}


// dW += dY^T X and db += colsum(dY) of an fp32 conditioning Linear
int lin_wgrad32(const float* dY, const float* X, float* dW, float* db, int M, int N, int K, void* st) {
  // batch 16: 13 us vs 28 + 5 (GEMM + column sums) for the 384-wide Linears, 57 vs 77 + 5 for the adaLN Linear
  if (M <= 16 && uwu_skinny_linear_ok(M, N, K)) return uwu_skinny_linear_wgrad(dY, X, dW, db, M, N, K, st);
  RUN(lin_wgrad(dY, X, dW, M, N, K, UWU_F32, st));
  return uwu_colsum(dY, UWU_F32, M, N, N, db, 1, st);
}

// ---- fp8 Linears (BASELINE config 5) ----------------------------------------------------------------------------
// Every operand of the block-scaled-MFMA GEMM is contraction-contiguous fp8, so each Linear quantises its input
// (forward), and in the backward its output gradient (row-major for the input gradient, transposed for the weight
// gradient; the bias gradient = its column sums comes out of the same pass) and the saved input (transposed).
struct F8 {
  int mode;  // 0 off, 1 just-in-time scaling, 2 delayed scaling
  float* scale;
  float* amax;
  const int32_t* fmt;
  char *x8, *x8t, *dy8, *dy8t, *wsc, *w8;
  size_t wsc_bytes, w8_layer;
  int D;
  void* st;
  int role(int l, int r) const { return 12 * l + r; }
  // W (row-major [N,K]) and W^T ([K,N]) of Linear i (0 qkv, 1 proj, 2 fc1, 3 fc2) of block l
  char* w(int l, int i, bool t) const {
    static const int off[4] = {0, 6, 8, 16}, sz[4] = {3, 1, 4, 4};
    return w8 + (size_t)l * w8_layer + ((size_t)off[i] + (t ? sz[i] : 0)) * D * D;
  }
  // quantise x [M,K] with the scale of `role`; jit: derive that scale from x first (amax pass + update)
  int quant(const void* x, int dtype, int M, int K, int role, void* out, void* out_t, float* colsum, bool have_scale) const {
    const int fm = role % 12 >= 4 && role % 12 < 8 ? UWU_FP8_E5M2 : UWU_FP8_E4M3;
    if (!have_scale && mode == 1) {
      RUN(uwu_fp8_amax(x, dtype, (int64_t)M * K, amax + role, st));
      RUN(uwu_fp8_update_scales(amax + role, scale + role, fmt + role, 1, 1.f, st));
    }
    return uwu_fp8_quantize(x, dtype, M, K, K, scale + role, fm, out, K, out_t, M,
                            (!have_scale && mode == 2) ? amax + role : nullptr, colsum, st);
  }
};

// Y = X . W^T (+ epilogue) with fp8 operands: X is quantised here (role rx), the weight was quantised at the start of
// the forward (role rw)
int f8_fwd(const F8& f, const void* X, void* Xt_save, const void* W8, const float* bias, void* Y, void* Y2, int M, int N, int K,
           int rx, int rw, int epi) {
  // one pass over X: the row-major copy for this GEMM and the transposed copy the weight gradient will contract over
  // (X == NULL: the producer of X already left both, ln_fwd_q8 below)
  if (X) RUN(f.quant(X, UWU_BF16, M, K, rx, f.x8, Xt_save, nullptr, false));
  return uwu_gemm_fp8(f.x8, W8, Y, Y2, bias, nullptr, M, N, K, K, K, N, 0, UWU_FP8_E4M3, epi, f.scale + rx, f.scale + rw,
                      nullptr, 0, f.st);
}
// backward of Y[M,N] = X[M,K] . W[N,K]^T: dW[N,K] += dY^T X, db[N] += colsum(dY), dX[M,K] = dY . W (optionally x gelu'(aux),
// colsum of dX into dcol).  X8t = the transposed fp8 copy of the input saved by the forward (scale of role rx).
// dY == NULL: the producer of dY already left its fp8 copies in dY8 / dY8t (f8_emit below) and its column sums where they belong.
// rnext >= 0 (with aux): dX leaves only as fp8 (e5m2, scale / amax of role rnext) in q8 [M,K] / q8t [K,M] for the next f8_bwd.
int f8_bwd(const F8& f, const void* dY, const void* X8t, const void* W8t, float* dW, float* db, void* dX, const void* aux,
           float* dcol, int M, int N, int K, int rdy, int rx, int rw, const void* dY8 = nullptr, const void* dY8t = nullptr,
           int rnext = -1, void* q8 = nullptr, void* q8t = nullptr) {
  if (dY) {
    RUN(f.quant(dY, UWU_BF16, M, N, rdy, f.dy8, f.dy8t, db, false));
    dY8 = f.dy8;
    dY8t = f.dy8t;
  }
  RUN(uwu_gemm_fp8(dY8t, X8t, dW, nullptr, nullptr, nullptr, N, K, M, M, M, K, 0, UWU_FP8_E5M2, UWU_EPI_ACCUM,
                   f.scale + rdy, f.scale + rx, f.wsc, f.wsc_bytes, f.st));
  if (rnext >= 0)
    return uwu_gemm_fp8_emit(dY8, W8t, nullptr, dcol, nullptr, aux, M, K, N, N, N, K, K, UWU_FP8_E5M2, UWU_EPI_DGELU, f.scale + rdy,
                             f.scale + rw, q8, K, q8t, M, f.scale + rnext, f.amax + rnext, f.st);
  if (!dX) return UWU_OK;
  return uwu_gemm_fp8(dY8, W8t, dX, dcol, nullptr, aux, M, K, N, N, N, K, K, UWU_FP8_E5M2, aux ? UWU_EPI_DGELU : UWU_EPI_NONE,
                      f.scale + rdy, f.scale + rw, nullptr, 0, f.st);
}
// Delayed scaling knows the scale of a tensor before the tensor exists, so the two widest ones of a block -- gelu(u) (fc2's
// input) and du (fc1's output gradient), [M, 4 D] each -- leave the GEMM that produces them as fp8 (row-major and transposed)
// and the quantising pass over their bf16 copy (4 bytes of HBM traffic per element) is gone.  The first step (just-in-time
// scaling: the scale comes from the tensor) keeps the two-pass form.  UWU_F8_EMIT=0: off (A/B).
static bool f8_emit(const uwu_dit_desc& d);
// h = LN(x_in + gate * y) * (1 + scale) + shift feeding ONLY an fp8 Linear: with delayed scaling the LayerNorm kernel writes the
// e4m3 images itself (row-major into the shared x8, transposed into the layer's saved copy) and no bf16 h exists.
// Returns 1 when it did (the caller then passes X = NULL to f8_fwd), 0 when the bf16 form ran, < 0 on error.
static int ln_fwd_f8(const uwu_dit_desc& d, const F8& f, const void* x_in, const void* y, const float* gate, const float* shift,
                     const float* scale, int ML, void* x_out, void* h, void* h8t, int role, float* mean, float* rstd, void* st) {
  const int64_t M = (int64_t)d.B * d.T;
  if (f8_emit(d) && d.D <= 1536 && M % 64 == 0) {
    const int rc = uwu_add_ln_modulate_fwd_q8(x_in, y, gate, shift, scale, ML, x_out, f.x8, d.D, h8t, (int)M, f.scale + role,
                                              f.amax + role, mean, rstd, d.B, d.T, d.D, d.ln_eps, st);
    return rc == UWU_OK ? 1 : rc;
  }
  const int rc = uwu_add_ln_modulate_fwd(x_in, y, gate, shift, scale, ML, x_out, h, mean, rstd, d.B, d.T, d.D, d.ln_eps, 0, d.dtype, st);
  return rc == UWU_OK ? 0 : rc;
}
static bool f8_emit(const uwu_dit_desc& d) {
  static UwuEnv on("UWU_F8_EMIT");
  return !on.get().is('0') && d.fp8 == 2;
}

// ---- small batches: weight gradients on a second stream --------------------------------------------------------------
// At the reference YAML's batch (16 images, M = 4096 token rows) every kernel of the backward under-fills the chip (10-25 us
// launches on a fraction of the CUs, back to back).  The four weight gradients of a block do not feed the chain of input
// gradients, so with a side stream they run BESIDE it.  Ordering by events only: the side stream waits for the producer of
// its dY; the main stream waits for the side stream's consumer before a kernel overwrites that dY buffer (dy is rewritten
// by the next LayerNorm backward, du / dqkv by the next block).  At large batches the kernels fill the chip by themselves
// and concurrent launches only time-slice (measured round 1: -7 % .. +7 %), so the fork is taken for M <= 16384 only.
struct Fork {
  hipStream_t main;
  hipStream_t side[4];  // one per weight gradient of a block (qkv's = the caller's side stream, which carries layer_done)
  hipEvent_t ev[8];     // 0-3: producer done (recorded on main); 4-7: consumer done (recorded on side[slot])
  bool on = false;
};
unsigned fork_event_flags() {
  const char* e = getenv("UWU_DIT_EVENT_SYSTEM");  // =1: default (system-scope) events, for A/B comparisons
  return (e && e[0] == '1') ? hipEventDisableTiming : (hipEventDisableTiming | hipEventReleaseToDevice);
}
bool fork_resources(Fork& f, hipStream_t caller_side) {
  static hipEvent_t pool[8];
  static hipStream_t extra[3];
  static bool made = false;
  if (!made) {
    for (auto& e : pool)
      // device-scope release: these events only order streams of this GPU.  The default system-scope fence (cache writeback +
      // invalidate so that the HOST could read the results) sat in front of every kernel that followed a record on the main stream
      if (hipEventCreateWithFlags(&e, fork_event_flags()) != hipSuccess) return false;
    for (auto& x : extra)
      if (hipStreamCreateWithFlags(&x, hipStreamNonBlocking) != hipSuccess) return false;
    made = true;
  }
  for (int i = 0; i < 8; ++i) f.ev[i] = pool[i];
  static int n_side = -1;  // UWU_DIT_SIDE_STREAMS=n (1..4): how many streams the four weight gradients of a block share
  if (n_side < 0) {
    const char* e = getenv("UWU_DIT_SIDE_STREAMS");
    n_side = e ? atoi(e) : 1;  // measured at batch 16 / 64 (img/s): 1: 4736 / 9606, 2: 4594 / 9548, 4: 4543 / 9512 -- with the
    if (n_side < 1 || n_side > 4) n_side = 1;  // runtime's 4 hardware queues more streams only share queues (8 queues: 3x slower)
  }
  for (int i = 0; i < 3; ++i) f.side[i] = n_side == 1 ? caller_side : extra[i % (n_side - 1)];
  f.side[3] = caller_side;
  return true;
}
// side stream `slot`: wait for `produced` (recorded on main now), run fn there, record `consumed`
template <typename F>
int on_side(const Fork& f, int slot, F&& fn) {
  if (!f.on) return fn(static_cast<void*>(f.main), 0);
  if (hipEventRecord(f.ev[slot], f.main) != hipSuccess || hipStreamWaitEvent(f.side[slot], f.ev[slot], 0) != hipSuccess) {
    uwu_set_error("dit_backward: stream fork failed");
    return UWU_ELAUNCH;
  }
  const int rc = fn(static_cast<void*>(f.side[slot]), f.side[slot] == f.side[3] ? 0 : slot + 1);
  if (rc != UWU_OK) return rc;
  if (hipEventRecord(f.ev[4 + slot], f.side[slot]) != hipSuccess) { uwu_set_error("dit_backward: stream fork failed"); return UWU_ELAUNCH; }
  return UWU_OK;
}
// main stream: do not overwrite a buffer before the side stream's consumer `slot` of it has finished
int join_side(const Fork& f, int slot) {
  if (f.on && hipStreamWaitEvent(f.main, f.ev[4 + slot], 0) != hipSuccess) {
    uwu_set_error("dit_backward: stream join failed");
    return UWU_ELAUNCH;
  }
  return UWU_OK;
}

struct Ptrs {
  char* ws;
  const Layout& L;
  template <typename T = void>
  T* at(size_t off) const { return reinterpret_cast<T*>(ws + off); }
  template <typename T = void>
  T* lay(int l, size_t sub) const {
    if (L.ckpt && sub >= L.keep) return reinterpret_cast<T*>(ws + L.scratch0 + sub);
    return reinterpret_cast<T*>(ws + L.layer0 + (size_t)l * L.layer_stride + sub);
  }
};

struct LayerW {
  const void *qkv_w, *o_w, *fc1_w, *fc2_w;
  const float *qkv_b, *o_b, *fc1_b, *fc2_b;
  int64_t off_qkv_w, off_qkv_b, off_o_w, off_o_b, off_fc1_w, off_fc1_b, off_fc2_w, off_fc2_b;
};

inline int64_t pad64(int64_t n) { return (n + 63) & ~(int64_t)63; }

LayerW layer_weights(const uwu_dit_desc& d, int l) {
  LayerW w{};
  const int64_t D = d.D, D4 = (int64_t)d.mlp_ratio * d.D;
  int64_t o = d.off_layer0 + (int64_t)l * d.layer_stride;
  w.off_qkv_w = o; o += pad64(3 * D * D);
  w.off_qkv_b = o; o += pad64(3 * D);
  w.off_o_w = o; o += pad64(D * D);
  w.off_o_b = o; o += pad64(D);
  w.off_fc1_w = o; o += pad64(D4 * D);
  w.off_fc1_b = o; o += pad64(D4);
  w.off_fc2_w = o; o += pad64(D * D4);
  w.off_fc2_b = o; o += pad64(D);
  const size_t es = d.dtype == UWU_BF16 ? 2 : 4;
  const char* wb = static_cast<const char*>(d.w);
  w.qkv_w = wb + w.off_qkv_w * es;
  w.o_w = wb + w.off_o_w * es;
  w.fc1_w = wb + w.off_fc1_w * es;
  w.fc2_w = wb + w.off_fc2_w * es;
  w.qkv_b = d.w32 + w.off_qkv_b;
  w.o_b = d.w32 + w.off_o_b;
  w.fc1_b = d.w32 + w.off_fc1_b;
  w.fc2_b = d.w32 + w.off_fc2_b;
  return w;
}


// One transformer block of the forward pass.  recompute = true (block recomputation inside the backward): the block's
// input x0 is the copy the forward kept, so its first LayerNorm takes it as is instead of forming it from the previous
// block's pending MLP branch.
int block_forward(const uwu_dit_desc& d, const Layout& L, const Ptrs& P, const F8& f8, int l, bool recompute, void* st) {
  const int dt = d.dtype, B = d.B, T = d.T, D = d.D, M = (int)L.M, D3 = (int)L.D3, D4 = (int)L.D4;
  const size_t es = L.es;
  const float* w32 = d.w32;
  const int ML = d.mod_total;
  const float* mod = P.at<float>(L.mod);
  const float scale = 1.f / sqrtf((float)(D / d.H));
  const LayerW w = layer_weights(d, l);
  const float* m = mod + (int64_t)l * 6 * D;  // shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
  void* x0 = P.lay(l, L.o_x0);
  // LN1 (+ pending MLP branch of the previous layer: x0 = x1_prev + gate_mlp_prev * y2_prev)
  int h1q = 0, h2q = 0;  // fp8 mode: 1 = the LayerNorm left its output as fp8 (no bf16 h1 / h2)
  if (d.fp8) {
    if (l == 0 || recompute) {
      h1q = ln_fwd_f8(d, f8, x0, nullptr, nullptr, m + 0, m + D, ML, x0, P.lay(l, L.o_h1), P.lay(l, L.o_h1t), f8.role(l, 0),
                      P.lay<float>(l, L.o_m1), P.lay<float>(l, L.o_r1), st);
    } else {
      const float* mp = mod + (int64_t)(l - 1) * 6 * D;
      h1q = ln_fwd_f8(d, f8, P.lay(l - 1, L.o_x1), P.lay(l - 1, L.o_y2), mp + 5 * D, m + 0, m + D, ML, x0, P.lay(l, L.o_h1),
                      P.lay(l, L.o_h1t), f8.role(l, 0), P.lay<float>(l, L.o_m1), P.lay<float>(l, L.o_r1), st);
    }
    if (h1q < 0) return h1q;
  } else
  if (l == 0 || recompute) {  // (recompute: x0 of this block was kept; its statistics come out the same)
    RUN(uwu_add_ln_modulate_fwd(x0, nullptr, nullptr, m + 0, m + D, ML, x0, P.lay(l, L.o_h1),
                                P.lay<float>(l, L.o_m1), P.lay<float>(l, L.o_r1), B, T, D, d.ln_eps, 0, dt, st));
  } else {
    const float* mp = mod + (int64_t)(l - 1) * 6 * D;
    RUN(uwu_add_ln_modulate_fwd(P.lay(l - 1, L.o_x1), P.lay(l - 1, L.o_y2), mp + 5 * D, m + 0, m + D, ML, x0,
                                P.lay(l, L.o_h1), P.lay<float>(l, L.o_m1), P.lay<float>(l, L.o_r1), B, T, D,
                                d.ln_eps, 0, dt, st));
  }
  if (d.fp8) RUN(f8_fwd(f8, h1q ? nullptr : P.lay(l, L.o_h1), P.lay(l, L.o_h1t), f8.w(l, 0, false), w.qkv_b, P.lay(l, L.o_qkv), nullptr, M, D3, D, f8.role(l, 0), f8.role(l, 8), UWU_EPI_BIAS));
  else RUN(lin_fwd(P.lay(l, L.o_h1), w.qkv_w, w.qkv_b, P.lay(l, L.o_qkv), nullptr, M, D3, D, dt, dt, UWU_EPI_BIAS, st));
  char* qkv = P.lay<char>(l, L.o_qkv);
  if (d.rope) {  // this layer's factor table, then attention with q / k rotated while they are staged
    const int64_t fo = (int64_t)l * d.H * (D / d.H / 4);
    RUN(uwu_axial_rope_table(d.pos_xy, w32 + d.off_rope_h + fo, w32 + d.off_rope_w + fo, P.lay<float>(l, L.o_rope), T, d.H,
                             D / d.H, D, st));
    RUN(uwu_attention_rope_fwd(qkv, qkv + (size_t)D * es, qkv + (size_t)2 * D * es, P.lay<float>(l, L.o_rope), P.lay(l, L.o_ao),
                               P.lay<float>(l, L.o_lse), B, T, d.H, D / d.H, D3, D3, D3, D, D, scale, dt, st));
  } else
  RUN(uwu_attention_fwd(qkv, qkv + (size_t)D * es, qkv + (size_t)2 * D * es, P.lay(l, L.o_ao), P.lay<float>(l, L.o_lse),
                        B, T, T, d.H, D / d.H, D3, D3, D3, D, scale, dt, st));
  if (d.fp8) RUN(f8_fwd(f8, P.lay(l, L.o_ao), P.lay(l, L.o_aot), f8.w(l, 1, false), w.o_b, P.lay(l, L.o_y1), nullptr, M, D, D, f8.role(l, 1), f8.role(l, 9), UWU_EPI_BIAS));
  else RUN(lin_fwd(P.lay(l, L.o_ao), w.o_w, w.o_b, P.lay(l, L.o_y1), nullptr, M, D, D, dt, dt, UWU_EPI_BIAS, st));
  // x1 = x0 + gate_msa * y1 ; h2 = LN(x1)*(1+scale_mlp)+shift_mlp
  if (d.fp8) {
    h2q = ln_fwd_f8(d, f8, x0, P.lay(l, L.o_y1), m + 2 * D, m + 3 * D, m + 4 * D, ML, P.lay(l, L.o_x1), P.lay(l, L.o_h2),
                    P.lay(l, L.o_h2t), f8.role(l, 2), P.lay<float>(l, L.o_m2), P.lay<float>(l, L.o_r2), st);
    if (h2q < 0) return h2q;
  } else
  RUN(uwu_add_ln_modulate_fwd(x0, P.lay(l, L.o_y1), m + 2 * D, m + 3 * D, m + 4 * D, ML, P.lay(l, L.o_x1),
                              P.lay(l, L.o_h2), P.lay<float>(l, L.o_m2), P.lay<float>(l, L.o_r2), B, T, D, d.ln_eps,
                              0, dt, st));
  if (d.fp8) {
    if (f8_emit(d)) {  // fc1 writes u (bf16, for the backward pass) and gelu(u) as e4m3: row-major into dy8 (free in the forward), transposed into ft
      const int r2 = f8.role(l, 2), r3 = f8.role(l, 3);
      if (!h2q) RUN(f8.quant(P.lay(l, L.o_h2), UWU_BF16, M, D, r2, f8.x8, P.lay(l, L.o_h2t), nullptr, false));
      RUN(uwu_gemm_fp8_emit(f8.x8, f8.w(l, 2, false), P.lay(l, L.o_u), nullptr, w.fc1_b, nullptr, M, D4, D, D, D, D4, 0, UWU_FP8_E4M3,
                            UWU_EPI_BIAS_GELU, f8.scale + r2, f8.scale + f8.role(l, 10), f8.dy8, D4, P.lay(l, L.o_ft), M, f8.scale + r3,
                            f8.amax + r3, st));
      RUN(uwu_gemm_fp8(f8.dy8, f8.w(l, 3, false), P.lay(l, L.o_y2), nullptr, w.fc2_b, nullptr, M, D, D4, D4, D4, D, 0, UWU_FP8_E4M3,
                       UWU_EPI_BIAS, f8.scale + r3, f8.scale + f8.role(l, 11), nullptr, 0, st));
      return UWU_OK;
    }
    RUN(f8_fwd(f8, h2q ? nullptr : P.lay(l, L.o_h2), P.lay(l, L.o_h2t), f8.w(l, 2, false), w.fc1_b, P.lay(l, L.o_u), P.lay(l, L.o_f), M, D4, D, f8.role(l, 2), f8.role(l, 10), UWU_EPI_BIAS_GELU));
    RUN(f8_fwd(f8, P.lay(l, L.o_f), P.lay(l, L.o_ft), f8.w(l, 3, false), w.fc2_b, P.lay(l, L.o_y2), nullptr, M, D, D4, f8.role(l, 3), f8.role(l, 11), UWU_EPI_BIAS));
  } else {
    RUN(lin_fwd(P.lay(l, L.o_h2), w.fc1_w, w.fc1_b, P.lay(l, L.o_u), P.lay(l, L.o_f), M, D4, D, dt, dt,
                UWU_EPI_BIAS_GELU, st));
    RUN(lin_fwd(P.lay(l, L.o_f), w.fc2_w, w.fc2_b, P.lay(l, L.o_y2), nullptr, M, D, D4, dt, dt, UWU_EPI_BIAS, st));
  }
  return UWU_OK;
}

}  // namespace

extern "C" size_t uwu_dit_workspace_bytes(const uwu_dit_desc* d) {
  if (!d) return 0;
  return make_layout(*d).total;
}

extern "C" int64_t uwu_dit_layer_param_stride(int D, int mlp_ratio) {
  const int64_t d = D, d4 = (int64_t)mlp_ratio * D;
  return pad64(3 * d * d) + pad64(3 * d) + pad64(d * d) + pad64(d) + pad64(d4 * d) + pad64(d4) + pad64(d * d4) + pad64(d);
}

extern "C" int uwu_dit_forward(const uwu_dit_desc* dp, const float* noisy, const float* t, const float* cond,
                               float* out, void* st) {
  RUN(check_desc(dp));
  const uwu_dit_desc& d = *dp;
  if (!noisy || !t || !out) { uwu_set_error("dit_forward: null tensor"); return UWU_EINVAL; }
  if ((d.cond_dim > 0) != (cond != nullptr)) { uwu_set_error("dit_forward: cond / cond_dim mismatch"); return UWU_EINVAL; }
  const Layout L = make_layout(d);
  if (d.ws_bytes < L.total) { uwu_set_error("dit_forward: workspace too small (%zu < %zu)", d.ws_bytes, L.total); return UWU_EINVAL; }
  if (d.layer_stride != uwu_dit_layer_param_stride(d.D, d.mlp_ratio)) { uwu_set_error("dit: layer_stride mismatch"); return UWU_EINVAL; }
  Ptrs P{static_cast<char*>(d.ws), L};
  const int dt = d.dtype, B = d.B, T = d.T, D = d.D, M = (int)L.M, D3 = (int)L.D3, D4 = (int)L.D4;
  const size_t es = L.es;
  const char* wb = static_cast<const char*>(d.w);
  const float* w32 = d.w32;
  const int ML = d.mod_total;

  // ---- conditioning path (fp32): c = MLP(sinusoid(t)) + proj(cond); mod = Linear(silu(c)) for ALL layers at once
  RUN(uwu_timestep_embedding(t, B, d.freq_dim, 10000.f, P.at(L.feat), UWU_F32, st));
  RUN(lin_fwd32(P.at<float>(L.feat), w32 + d.off_t_w1, w32 + d.off_t_b1, P.at<float>(L.t_pre), P.at<float>(L.t_h), B, D,
                d.freq_dim, UWU_EPI_BIAS_SILU, st));
  RUN(lin_fwd32(P.at<float>(L.t_h), w32 + d.off_t_w2, w32 + d.off_t_b2, P.at<float>(L.temb), nullptr, B, D, D, UWU_EPI_BIAS, st));
  const void* cptr = P.at(L.temb);
  if (cond) {
    RUN(lin_fwd32(cond, w32 + d.off_y_w, w32 + d.off_y_b, P.at<float>(L.yemb), nullptr, B, D, d.cond_dim, UWU_EPI_BIAS, st));
    RUN(uwu_add(P.at(L.temb), P.at(L.yemb), P.at(L.c), (int64_t)B * D, UWU_F32, st));
    cptr = P.at(L.c);
  }
  RUN(uwu_silu_fwd(cptr, P.at(L.sc), (int64_t)B * D, UWU_F32, st));
  if (mod_bf16(d)) {
    RUN(uwu_cast_f32_to_bf16(P.at<float>(L.sc), P.at(L.sc16), (int64_t)B * D, st));
    RUN(lin_fwd(P.at(L.sc16), wb + d.off_mod_w * es, w32 + d.off_mod_b, P.at(L.mod), nullptr, B, ML, D, UWU_BF16, UWU_F32, UWU_EPI_BIAS, st));
  } else {
    RUN(lin_fwd32(P.at<float>(L.sc), w32 + d.off_mod_w, w32 + d.off_mod_b, P.at<float>(L.mod), nullptr, B, ML, D, UWU_EPI_BIAS, st));
  }
  const float* mod = P.at<float>(L.mod);

  // ---- fp8 mode: this step's weights as fp8 (W for the forward, W^T for the input gradients), scaled per tensor
  const F8 f8{d.fp8, d.f8_scale, d.f8_amax, d.f8_fmt, P.at<char>(L.x8), nullptr, P.at<char>(L.dy8),
              P.at<char>(L.dy8t), P.at<char>(L.wsc), P.at<char>(L.w8), L.wsc_bytes, L.w8_layer, d.D, st};
  if (d.fp8) {
    // weights always scale just in time: every weight's amax pass first, then ONE scale update for all 12 L roles -- under
    // delayed scaling the same launch turns the activations' amax values of the last step into this step's scales (just-in-time
    // mode: their amax slots are zero here and their scales stay) --, then the quantising passes with those scales
    const F8 fw{2, d.f8_scale, d.f8_amax, d.f8_fmt, nullptr, nullptr, nullptr, nullptr, nullptr, f8.w8, 0, L.w8_layer, d.D, st};
    const int rows[4] = {D3, D, D4, D}, cols[4] = {D, D, D, D4};
    for (int l = 0; l < d.L; ++l) {
      const LayerW w = layer_weights(d, l);
      const void* ws4[4] = {w.qkv_w, w.o_w, w.fc1_w, w.fc2_w};
      for (int i = 0; i < 4; ++i)
        RUN(uwu_fp8_amax(ws4[i], UWU_BF16, (int64_t)rows[i] * cols[i], d.f8_amax + fw.role(l, 8 + i), st));
    }
    RUN(uwu_fp8_update_scales(d.f8_amax, d.f8_scale, d.f8_fmt, 12 * d.L, 1.f, st));
    for (int l = 0; l < d.L; ++l) {
      const LayerW w = layer_weights(d, l);
      const void* ws4[4] = {w.qkv_w, w.o_w, w.fc1_w, w.fc2_w};
      for (int i = 0; i < 4; ++i)
        RUN(fw.quant(ws4[i], UWU_BF16, rows[i], cols[i], fw.role(l, 8 + i), fw.w(l, i, false), fw.w(l, i, true), nullptr, true));
    }
  }

  // ---- patch embedding (Conv2d k=p,s=p == patchify + GEMM) + fixed 2-D sin-cos positions
  RUN(uwu_patchify(noisy, P.at(L.tok), B, d.in_ch, d.img, d.img, d.patch, dt, st));
  RUN(lin_fwd(P.at(L.tok), wb + d.off_patch_w * es, w32 + d.off_patch_b, P.lay(0, L.o_x0), nullptr, M, D, (int)L.Kp, dt,
              dt, UWU_EPI_BIAS, st));
  RUN(uwu_add_pos(P.lay(0, L.o_x0), d.pos, B, T, D, dt, st));

  for (int l = 0; l < d.L; ++l) RUN(block_forward(d, L, P, f8, l, false, st));
  // ---- final adaLN + linear + unpatchify
  {
    const int l = d.L - 1;
    const float* mp = mod + (int64_t)l * 6 * D;
    const float* mf = mod + (int64_t)d.L * 6 * D;  // shift, scale
    RUN(uwu_add_ln_modulate_fwd(P.lay(l, L.o_x1), P.lay(l, L.o_y2), mp + 5 * D, mf + 0, mf + D, ML, P.at(L.xF),
                                P.at(L.hF), P.at<float>(L.mF), P.at<float>(L.rF), B, T, D, d.ln_eps, 0, dt, st));
    RUN(lin_fwd(P.at(L.hF), wb + d.off_final_w * es, w32 + d.off_final_b, P.at(L.otok), nullptr, M, (int)L.Ko, D, dt, dt,
                UWU_EPI_BIAS, st));
    RUN(uwu_unpatchify(P.at(L.otok), dt, out, B, d.out_ch, d.img, d.img, d.patch, st));
  }
  return UWU_OK;
}

extern "C" int uwu_dit_backward(const uwu_dit_desc* dp, const float* dout, void* st) {
  RUN(check_desc(dp));
  const uwu_dit_desc& d = *dp;
  if (!dout || !d.g32) { uwu_set_error("dit_backward: null tensor"); return UWU_EINVAL; }
  const Layout L = make_layout(d);
  if (d.ws_bytes < L.total) { uwu_set_error("dit_backward: workspace too small"); return UWU_EINVAL; }
  Ptrs P{static_cast<char*>(d.ws), L};
  const int dt = d.dtype, B = d.B, T = d.T, D = d.D, M = (int)L.M, D3 = (int)L.D3, D4 = (int)L.D4;
  const size_t es = L.es;
  const char* wb = static_cast<const char*>(d.w);
  const float* w32 = d.w32;
  float* g = d.g32;
  const int ML = d.mod_total;
  const float* mod = P.at<float>(L.mod);
  float* dmod = P.at<float>(L.dmod);
  if (hipMemsetAsync(dmod, 0, (size_t)B * ML * sizeof(float), (hipStream_t)st) != hipSuccess) {
    uwu_set_error("dit_backward: memset failed");
    return UWU_ELAUNCH;
  }
  const float scale = 1.f / sqrtf((float)(D / d.H));
  const F8 f8{d.fp8, d.f8_scale, d.f8_amax, d.f8_fmt, P.at<char>(L.x8), nullptr, P.at<char>(L.dy8),
              P.at<char>(L.dy8t), P.at<char>(L.wsc), P.at<char>(L.w8), L.wsc_bytes, L.w8_layer, d.D, st};
  Fork fk{static_cast<hipStream_t>(st), {}, {}, false};
  // (block recomputation reuses ONE slab of saved activations: the recomputation of block l - 1 must not overtake a weight
  // gradient of block l on another stream, so that mode keeps everything on the call's stream)
  fk.on = d.side_stream && d.side_stream != st && !d.fp8 && dt == UWU_BF16 && M <= 16384 && !L.ckpt &&
          fork_resources(fk, static_cast<hipStream_t>(d.side_stream));
  char* const wsc = P.at<char>(L.wsc);  // region k of the split-K scratch belongs to the stream that fn(.., k) runs on
  bool side_used[4] = {false, false, false, false};  // a consumer-done event of this slot has been recorded in this call

  // ---- output head
  RUN(uwu_patchify(dout, P.at(L.dotok), B, d.out_ch, d.img, d.img, d.patch, dt, st));
  RUN(lin_wgrad(P.at(L.dotok), P.at(L.hF), g + d.off_final_w, M, (int)L.Ko, D, dt, st));
  RUN(uwu_colsum(P.at(L.dotok), dt, M, (int)L.Ko, (int)L.Ko, g + d.off_final_b, 1, st));
  RUN(lin_dgrad(P.at(L.dotok), wb + d.off_final_w * es, P.at(L.dh), nullptr, M, (int)L.Ko, D, dt, st));
  {
    const int l = d.L - 1;
    const float* mp = mod + (int64_t)l * 6 * D;
    const float* mf = mod + (int64_t)d.L * 6 * D;
    float* dmp = dmod + (int64_t)l * 6 * D;
    float* dmf = dmod + (int64_t)d.L * 6 * D;
    // final LN bwd + gate bwd of the last MLP branch: dx = d/d x1_{L-1}; dy = gate_mlp * dx
    RUN(uwu_add_ln_modulate_bwd(P.at(L.dh), P.at(L.xF), P.at<float>(L.mF), P.at<float>(L.rF), mf + D, nullptr,
                                P.lay(l, L.o_y2), mp + 5 * D, ML, P.at(L.dx), P.at(L.dy), dmf + 0, dmf + D, dmp + 5 * D,
                                B, T, D, 0, dt, st));
  }
  for (int l = d.L - 1; l >= 0; --l) {
    const LayerW w = layer_weights(d, l);
    const float* m = mod + (int64_t)l * 6 * D;
    float* dm = dmod + (int64_t)l * 6 * D;
    // ---- MLP branch: y2 = fc2(gelu(fc1(h2)))
    if (d.fp8) {
      // fc2: dW2, db2, du = (dy.W2) * gelu'(u) with colsum(du) = fc1.bias gradient;  fc1: dW1, dh = du.W1
      if (f8_emit(d)) {  // du leaves fc2's input-gradient GEMM as e5m2: row-major into x8 (free in the backward), transposed into dy8t
        RUN(f8_bwd(f8, P.at(L.dy), P.lay(l, L.o_ft), f8.w(l, 3, true), g + w.off_fc2_w, g + w.off_fc2_b, nullptr, P.lay(l, L.o_u),
                   g + w.off_fc1_b, M, D, D4, f8.role(l, 7), f8.role(l, 3), f8.role(l, 11), nullptr, nullptr, f8.role(l, 6), f8.x8, f8.dy8t));
        RUN(f8_bwd(f8, nullptr, P.lay(l, L.o_h2t), f8.w(l, 2, true), g + w.off_fc1_w, nullptr, P.at(L.dh), nullptr, nullptr, M, D4, D,
                   f8.role(l, 6), f8.role(l, 2), f8.role(l, 10), f8.x8, f8.dy8t));
      } else {
      RUN(f8_bwd(f8, P.at(L.dy), P.lay(l, L.o_ft), f8.w(l, 3, true), g + w.off_fc2_w, g + w.off_fc2_b, P.at(L.du), P.lay(l, L.o_u),
                 g + w.off_fc1_b, M, D, D4, f8.role(l, 7), f8.role(l, 3), f8.role(l, 11)));
      RUN(f8_bwd(f8, P.at(L.du), P.lay(l, L.o_h2t), f8.w(l, 2, true), g + w.off_fc1_w, nullptr, P.at(L.dh), nullptr, nullptr, M, D4, D,
                 f8.role(l, 6), f8.role(l, 2), f8.role(l, 10)));
      }
    } else {
    RUN(on_side(fk, 0, [&](void* s2, int k) {
      return lin_wgrad(P.at(L.dy), P.lay(l, L.o_f), g + w.off_fc2_w, M, D, D4, dt, s2, wsc + (size_t)k * L.wsc_bytes, L.wsc_bytes, g + w.off_fc2_b);
    }));
    side_used[0] = true;
    if (side_used[1]) RUN(join_side(fk, 1));  // the previous block's fc1 weight gradient still reads du
    // du = (dy.W2) * gelu'(u).  The fc1 bias gradient = colsum(du) comes out of the fc1 weight-gradient kernel (extra MFMAs
    // against an all-ones fragment, free there) -- as fp32 atomics in this epilogue it cost 79 us per launch at B = 768
    if (dt == UWU_BF16 && D == 384 && M % 256 == 0 && M >= 256 * 256 && D4 % 64 == 0 && D4 >= 1024 && D4 <= 2048 && fc2_dgrad_as()) {
      // the store-heavy input gradient (604 MB in, 604 MB out for 151 MB of dy): W2 transposed once (1.2 MB), then the
      // A-stationary kernel with dy held in fragment registers and the dGELU epilogue between the K-steps
      RUN(uwu_transpose_bf16(w.fc2_w, P.at(L.w2t), D, D4, D4, D, st));
      RUN(uwu_gemm(P.at(L.dy), P.at(L.w2t), P.at(L.du), nullptr, nullptr, P.lay(l, L.o_u), M, D4, D, D, D, D4, D4, 0, 0, dt, dt,
                   UWU_EPI_DGELU, 1, st));
    } else {
      RUN(lin_dgrad(P.at(L.dy), w.fc2_w, P.at(L.du), P.lay(l, L.o_u), M, D, D4, dt, st, nullptr));
    }
    RUN(on_side(fk, 1, [&](void* s2, int k) {
      return lin_wgrad(P.at(L.du), P.lay(l, L.o_h2), g + w.off_fc1_w, M, D4, D, dt, s2, wsc + (size_t)k * L.wsc_bytes, L.wsc_bytes, g + w.off_fc1_b);
    }));
    side_used[1] = true;
    RUN(lin_dgrad(P.at(L.du), w.fc1_w, P.at(L.dh), nullptr, M, D4, D, dt, st));
    RUN(join_side(fk, 0));  // LN2 backward rewrites dy: the fc2 weight gradient must have read it
    }
    // LN2 bwd (+ residual) and gate bwd of the attention branch
    RUN(uwu_add_ln_modulate_bwd(P.at(L.dh), P.lay(l, L.o_x1), P.lay<float>(l, L.o_m2), P.lay<float>(l, L.o_r2), m + 4 * D,
                                P.at(L.dx), P.lay(l, L.o_y1), m + 2 * D, ML, P.at(L.dx), P.at(L.dy), dm + 3 * D,
                                dm + 4 * D, dm + 2 * D, B, T, D, 0, dt, st));
    // ---- attention branch: y1 = proj(attn(qkv(h1)))
    if (d.fp8) {
      RUN(f8_bwd(f8, P.at(L.dy), P.lay(l, L.o_aot), f8.w(l, 1, true), g + w.off_o_w, g + w.off_o_b, P.at(L.dao), nullptr, nullptr, M, D, D,
                 f8.role(l, 5), f8.role(l, 1), f8.role(l, 9)));
    } else {
    RUN(on_side(fk, 2, [&](void* s2, int k) {
      return lin_wgrad(P.at(L.dy), P.lay(l, L.o_ao), g + w.off_o_w, M, D, D, dt, s2, wsc + (size_t)k * L.wsc_bytes, L.wsc_bytes, g + w.off_o_b);
    }));
    side_used[2] = true;
    RUN(lin_dgrad(P.at(L.dy), w.o_w, P.at(L.dao), nullptr, M, D, D, dt, st));
    if (side_used[3]) RUN(join_side(fk, 3));  // the previous block's qkv weight gradient still reads dqkv
    }
    char* qkv = P.lay<char>(l, L.o_qkv);
    char* dqkv = P.at<char>(L.dqkv);
    if (d.rope) {
      // gradients wrt the rotated q' / k' come out of the attention kernel; the elementwise pass turns them into dq / dk in
      // place and accumulates the log-frequency gradients of this layer
      const int64_t fo = (int64_t)l * d.H * (D / d.H / 4);
      RUN(uwu_attention_rope_bwd(qkv, qkv + (size_t)D * es, qkv + (size_t)2 * D * es, P.lay<float>(l, L.o_rope), P.lay(l, L.o_ao),
                                 P.at(L.dao), P.lay<float>(l, L.o_lse), dqkv, dqkv + (size_t)D * es, dqkv + (size_t)2 * D * es,
                                 B, T, d.H, D / d.H, D3, D3, D3, D, D, scale, dt, st));
      for (int qk = 0; qk < 2; ++qk)
        RUN(uwu_axial_rope_bwd_shared(qkv + (size_t)qk * D * es, dqkv + (size_t)qk * D * es, d.pos_xy, T,
                                      w32 + d.off_rope_h + fo, w32 + d.off_rope_w + fo, dqkv + (size_t)qk * D * es,
                                      g + d.off_rope_h + fo, g + d.off_rope_w + fo, (int64_t)M, d.H, D / d.H, D3, dt, st));
    } else
    RUN(uwu_attention_bwd(qkv, qkv + (size_t)D * es, qkv + (size_t)2 * D * es, P.lay(l, L.o_ao), P.at(L.dao),
                          P.lay<float>(l, L.o_lse), P.at<float>(L.delta), dqkv, dqkv + (size_t)D * es,
                          dqkv + (size_t)2 * D * es, B, T, T, d.H, D / d.H, D3, D3, D3, D, scale, dt, st));
    if (d.fp8) {  // (the input gradient follows the weight gradient inside f8_bwd; the block's event is recorded after both)
      RUN(f8_bwd(f8, dqkv, P.lay(l, L.o_h1t), f8.w(l, 0, true), g + w.off_qkv_w, g + w.off_qkv_b, P.at(L.dh), nullptr, nullptr, M, D3, D,
                 f8.role(l, 4), f8.role(l, 0), f8.role(l, 8)));
    } else {
    RUN(on_side(fk, 3, [&](void* s2, int k) {
      return lin_wgrad(dqkv, P.lay(l, L.o_h1), g + w.off_qkv_w, M, D3, D, dt, s2, wsc + (size_t)k * L.wsc_bytes, L.wsc_bytes, g + w.off_qkv_b);
    }));
    side_used[3] = true;
    }
    // every parameter gradient of block l is now in flight (on the side stream when the backward is forked: its launches
    // are in order, so the last weight gradient's stream carries the event)
    if (fk.on && d.layer_done && d.layer_done[l])
      for (int k = 0; k < 3; ++k)
        if (fk.side[k] != fk.side[3] && hipStreamWaitEvent(fk.side[3], fk.ev[4 + k], 0) != hipSuccess) {
          uwu_set_error("dit_backward: stream join failed");
          return UWU_ELAUNCH;
        }
    if (d.layer_done && d.layer_done[l] &&
        hipEventRecord(static_cast<hipEvent_t>(d.layer_done[l]), fk.on ? fk.side[3] : static_cast<hipStream_t>(st)) != hipSuccess) {
      uwu_set_error("dit_backward: hipEventRecord(layer_done[%d]) failed", l);
      return UWU_ELAUNCH;
    }
    if (!d.fp8) RUN(lin_dgrad(dqkv, w.qkv_w, P.at(L.dh), nullptr, M, D3, D, dt, st));
    if (!d.fp8) RUN(join_side(fk, 2));  // LN1 backward rewrites dy: the proj weight gradient must have read it
    // block recomputation: every saved tensor of block l except x0 / mean / rstd of its first LayerNorm is dead now -- rerun
    // block l - 1 from its kept input into the shared slab (its y2 is needed right below, the rest by the next iteration)
    if (L.ckpt && l > 0) RUN(block_forward(d, L, P, f8, l - 1, true, st));
    // LN1 bwd (+ residual) and gate bwd of the previous layer's MLP branch
    if (l > 0) {
      const float* mp = mod + (int64_t)(l - 1) * 6 * D;
      float* dmp = dmod + (int64_t)(l - 1) * 6 * D;
      RUN(uwu_add_ln_modulate_bwd(P.at(L.dh), P.lay(l, L.o_x0), P.lay<float>(l, L.o_m1), P.lay<float>(l, L.o_r1), m + D,
                                  P.at(L.dx), P.lay(l - 1, L.o_y2), mp + 5 * D, ML, P.at(L.dx), P.at(L.dy), dm + 0,
                                  dm + D, dmp + 5 * D, B, T, D, 0, dt, st));
    } else {
      RUN(uwu_add_ln_modulate_bwd(P.at(L.dh), P.lay(l, L.o_x0), P.lay<float>(l, L.o_m1), P.lay<float>(l, L.o_r1), m + D,
                                  P.at(L.dx), nullptr, nullptr, ML, P.at(L.dx), nullptr, dm + 0, dm + D, nullptr, B, T, D, 0, dt, st));
    }
  }
  if (fk.on)
    for (int k = 0; k < 4; ++k) RUN(join_side(fk, k));  // each side stream is in order: its last weight gradient ends it
  // ---- patch embedding (input latents need no gradient; positions are fixed)
  RUN(lin_wgrad(P.at(L.dx), P.at(L.tok), g + d.off_patch_w, M, D, (int)L.Kp, dt, st));
  RUN(uwu_colsum(P.at(L.dx), dt, M, D, D, g + d.off_patch_b, 1, st));

  // ---- conditioning path (fp32)
  if (mod_bf16(d)) {
    // dW += d(mod)^T . silu(c), db += colsum(d(mod)) (fp32), d(silu c) = d(mod) . W: bf16 operands, fp32 accumulation / outputs
    RUN(uwu_cast_f32_to_bf16(dmod, P.at(L.dmod16), (int64_t)B * ML, st));
    RUN(lin_wgrad(P.at(L.dmod16), P.at(L.sc16), g + d.off_mod_w, B, ML, D, UWU_BF16, st));
    RUN(uwu_colsum(dmod, UWU_F32, B, ML, ML, g + d.off_mod_b, 1, st));
    if (hipMemsetAsync(P.at(L.dsc), 0, (size_t)B * D * sizeof(float), (hipStream_t)st) != hipSuccess) {
      uwu_set_error("dit_backward: memset failed");
      return UWU_ELAUNCH;
    }
    {
      const int tiles = ((B + 127) / 128) * ((D + 127) / 128), ktiles = (ML + 63) / 64;
      int split = (512 + tiles - 1) / tiles;
      if (split > ktiles) split = ktiles;
      RUN(uwu_gemm(P.at(L.dmod16), wb + d.off_mod_w * es, P.at(L.dsc), nullptr, nullptr, nullptr, B, D, ML, ML, D, D, 0, 0, 1, UWU_BF16,
                   UWU_F32, UWU_EPI_ACCUM, split, st));
    }
  } else {
    RUN(lin_wgrad32(dmod, P.at<float>(L.sc), g + d.off_mod_w, g + d.off_mod_b, B, ML, D, st));
    RUN(lin_dgrad_splitk(dmod, w32 + d.off_mod_w, P.at<float>(L.dsc), B, ML, D, st));  // (matrix-vector form: 99 us vs 40)
  }
  const void* cptr = d.cond_dim > 0 ? P.at(L.c) : P.at(L.temb);
  RUN(uwu_silu_bwd(cptr, P.at(L.dsc), P.at(L.dc), (int64_t)B * D, UWU_F32, st));
  if (d.cond_dim > 0) {
    // cond is an input; its projection weights get dW += dc^T cond.  `cond` itself is re-read from the caller.
  }
  RUN(lin_wgrad32(P.at<float>(L.dc), P.at<float>(L.t_h), g + d.off_t_w2, g + d.off_t_b2, B, D, D, st));
  if (B <= 16 && uwu_skinny_linear_ok(B, D, D) && D <= 512) RUN(uwu_skinny_linear_dgrad(P.at<float>(L.dc), w32 + d.off_t_w2, P.at<float>(L.dth), B, D, D, st));
  else RUN(lin_dgrad(P.at(L.dc), w32 + d.off_t_w2, P.at(L.dth), nullptr, B, D, D, UWU_F32, st));
  RUN(uwu_silu_bwd(P.at(L.t_pre), P.at(L.dth), P.at(L.dtp), (int64_t)B * D, UWU_F32, st));
  RUN(lin_wgrad32(P.at<float>(L.dtp), P.at<float>(L.feat), g + d.off_t_w1, g + d.off_t_b1, B, D, d.freq_dim, st));
  return UWU_OK;
}

// gradient of the pooled-conditioning projection (needs the caller's `cond` tensor again)
extern "C" int uwu_dit_backward_cond(const uwu_dit_desc* dp, const float* cond, void* st) {
  RUN(check_desc(dp));
  const uwu_dit_desc& d = *dp;
  if (d.cond_dim <= 0) return UWU_OK;
  if (!cond || !d.g32) { uwu_set_error("dit_backward_cond: null tensor"); return UWU_EINVAL; }
  const Layout L = make_layout(d);
  Ptrs P{static_cast<char*>(d.ws), L};
  return lin_wgrad32(P.at<float>(L.dc), cond, d.g32 + d.off_y_w, d.g32 + d.off_y_b, d.B, d.D, d.cond_dim, st);
}
