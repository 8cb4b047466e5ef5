/*
 * uwu_hip.h -- C ABI of libuwu_hip.so: the MI355X (gfx950) kernels behind the `duwu`
 * training inner loop.
 *
 * The reference (KohakuBlueleaf/UwUDiff) has NO FFI on this path: its boundary is a Python
 * plugin slot (`instantiate_any`, reference src/duwu/utils/__init__.py:41-50; `load_any`,
 * src/duwu/loader.py:58-67).  Each entry below therefore cites the reference *Python* lines
 * whose torch/diffusers op sequence it replaces; INTEGRATION.md shows the ctypes binding a
 * maintainer adds on the reference side.
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch tensors' data_ptr());
 *   - no entry allocates, frees or synchronises; work is enqueued on `stream` (a hipStream_t
 *     passed as void*);
 *   - return 0 on success, a negative UWU_E* code otherwise; uwu_last_error() gives the text;
 *   - `dtype` arguments: UWU_F32 = 0, UWU_BF16 = 1 (storage type of the activation operands;
 *     accumulation, statistics, loss and optimizer state are always fp32).
 */
#ifndef UWU_HIP_H
#define UWU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UWU_F32 0
#define UWU_BF16 1

#define UWU_OK 0
#define UWU_EINVAL (-1)   /* bad argument / unsupported shape */
#define UWU_ELAUNCH (-2)  /* hipLaunch error */
#define UWU_ENOTIMPL (-3)

/* prediction / target parameterisations (reference src/duwu/loss/diffusion.py:84-125) */
#define UWU_PT_EPSILON 0
#define UWU_PT_V 1
#define UWU_PT_SAMPLE 2
#define UWU_PT_RF 3

/* GEMM epilogues */
#define UWU_EPI_NONE 0
#define UWU_EPI_BIAS 1       /* C = A.B + bias[n]                                        */
#define UWU_EPI_BIAS_GELU 2  /* C = A.B + bias ; C2 = gelu_tanh(C)   (fc1 forward)       */
#define UWU_EPI_DGELU 3      /* C = (A.B) * gelu_tanh'(aux[m,n])     (fc2 dgrad -> dU); if C2 != NULL it is a
                                float[N] receiving += column sums of C (fc1 bias gradient)              */
#define UWU_EPI_BIAS_SILU 4  /* C = silu(A.B + bias)                 (timestep MLP)      */
#define UWU_EPI_ACCUM 5      /* C += A.B  (fp32 C only; wgrad accumulation, split-K)     */

const char* uwu_last_error(void);
int uwu_version(void);
/* The library's environment switches (UWU_GEMM_*, ...: kernel A/B comparisons and sweeps, tools/README.md) are read once
 * and cached; call this after changing one inside a running process.  Returns the new generation number. */
int uwu_env_refresh(void);
/* Zero `bytes` bytes of device memory on `stream` (hipMemsetAsync): the flat gradient buffer and the optimizer moments at their
 * creation (reference: torch.zeros_like in torch.optim.AdamW's state initialisation / autograd's first accumulation). */
int uwu_memset_zero(void* p, uint64_t bytes, void* stream);

/* ------------------------------------------------------------------ objective (a1-a10) */

/* Per-sample schedule gather: replaces the Python loops at diffusion.py:58 (B x nonzero().item()),
 * :146 and :159-161.  coef[b] = {sigma_b, weight_b, sqrt(abar_t), sqrt(1-abar_t)}.
 *   sigma_b  = sigmas_desc[N-1-t_b]                         (diffusion.py:53-62)
 *   weight_b = [min(snr,gamma)/snr | /(snr+1) for v]  x  [1/sqrt(min(snr,1000))]   (:141-167)
 * snr_mode: 0 none, 1 epsilon form, 2 v form; debias: 0/1. */
int uwu_schedule_gather(const int64_t* timesteps, const float* sigmas_desc, const float* all_snr,
                        const float* alphas_cumprod, int n_train, int B, int snr_mode, float gamma,
                        int debias, float* coef, void* stream);

/* Rectified-flow time sampling: rectified_flow.py:29-42 and sigma_to_timestep :98-129.
 *   time = u01*smax/(1+smax); sigma = time/(1-time); timesteps = log-sigma piecewise-linear inverse
 * over log_sigmas_asc[n_train] (ascending log of sigmas[:-1] flipped).  coef[b] = {sigma,1,0,0}. */
int uwu_rf_time_to_sigma(const float* u01, float sigma_max, const float* log_sigmas_asc, int n_train,
                         int B, float* coef, float* timesteps, void* stream);

/* Forward process, diffusion.py:77-82 / rectified_flow.py:67-71:
 *   noisy = (x + noise*sigma_b) * (sigma_b^2+1)^-1/2 ; fp32 out and (optional, may be NULL) bf16 copy.
 * n = elements per sample. */
int uwu_qsample(const float* x, const float* noise, const float* coef, int B, int64_t n, float* noisy,
                void* noisy_bf16, void* stream);

/* The same with the VAE latent normalisation of trainer.py:241-244 folded in: x_norm = (x - vae_mean) / vae_std
 * (written once: it is the clean latent the loss compares against), noisy = (x_norm + noise*sigma_b)*(sigma_b^2+1)^-1/2. */
int uwu_qsample_norm(const float* x, const float* noise, const float* coef, int B, int64_t n, float vae_mean,
                     float vae_std, float* x_norm, float* noisy, void* stream);

/* In-kernel draws of the training objective (diffusion.py:68-70 `torch.randint(0, N, (B,))`, :75 `torch.randn_like(x)`;
 * rectified_flow.py:37 `torch.rand(B)`): Philox4x32-10, counter = offset + element group, key = seed (no device state).  The
 * host mirror reserves the offset ranges in the reference's order of draws (noise, then timesteps / u01) from torch's CUDA
 * generator.  uwu_philox_raw: n_counters x 4 raw words (parity with oracle/philox.py); uwu_philox_normal: n N(0,1) floats, 4 per
 * counter (Box-Muller); uwu_draw_timesteps: t[b] = (word (b % 4) of counter b / 4) * n_train >> 32; uwu_draw_u01: the same word
 * as ((r >> 8) + 1/2) / 2^24.  uwu_qsample_draw = uwu_qsample / uwu_qsample_norm with the noise drawn inside (element group i of
 * the [B, n] tensor = counter offset + i): writes noise, noisy and (use_norm) x_norm in one pass. */
int uwu_philox_raw(uint32_t* out, int64_t n_counters, uint64_t seed, uint64_t offset, void* stream);
int uwu_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);
int uwu_draw_timesteps(int64_t* timesteps, int n_train, int B, uint64_t seed, uint64_t offset, void* stream);
int uwu_draw_u01(float* u01, int B, uint64_t seed, uint64_t offset, void* stream);
int uwu_qsample_draw(const float* x, const float* coef, int B, int64_t n, int use_norm, float vae_mean, float vae_std,
                     float* x_norm, float* noise, float* noisy, uint64_t seed, uint64_t offset, void* stream);

/* Conditioning front-end, text_encoders.py:196-262: place one text encoder's hidden states src [B,S,F] (fp32 / bf16),
 * times its attention mask [B,S] (int64, may be NULL: zero_for_padding off or no mask), into the zero-filled context
 * out [B,S_total,F_total] (fp32) at sequence offset s_off (its concat bucket) and feature offset f_off (its place
 * inside the bucket; narrower buckets stay zero-padded on the feature axis). */
int uwu_ctx_place(const void* src, int dtype, const int64_t* mask, float* out, int B, int S, int F, int S_total,
                  int F_total, int s_off, int f_off, void* stream);

/* Fused prediction conversion + target + per-sample weighted MSE + d loss/d model_output,
 * diffusion.py:177-193 (+ :100-139) and rectified_flow.py:79-96.
 *   xt: the tensor the reference passes as `xt` (clean x for DiffusionLoss :177, noisy for RF :80).
 *   out_dtype: dtype of model_output and of grad_out.
 *   pred/target may be NULL (aux outputs).  losses[b] = w_b * mean((pred-target)^2);
 *   grad_out = d mean_b(losses) / d model_output.  force_convert=1 is the RF path (always converts). */
int uwu_loss_fwd_bwd(const float* x, const float* noise, const float* xt, const void* model_output,
                     int out_dtype, const float* coef, int pred_type, int target_type, int force_convert,
                     int B, int64_t n, float* losses, float* loss_mean, void* grad_out, float* pred,
                     float* target, void* stream);

/* y[i] *= *scale (device scalar); used when autograd hands a non-unit upstream gradient. */
int uwu_scale_inplace(void* y, int dtype, int64_t n, const float* scale, void* stream);
/* y[i] = x[i] * *scale: the same for a SAVED gradient that must stay as it is (the loss's d/d(model output), produced in its
 * forward pass, reference diffusion.py:170-193 via autograd) -- one pass instead of a clone + the in-place form. */
int uwu_scale_into(const void* x, void* y, int dtype, int64_t n, const float* scale, void* stream);
/* y[i] = (float)x[i] for the int64 timesteps the loss hands the denoiser (reference diffusion.py:68-70 draws them as int64,
 * rope_unet.py / the DiT's timestep embedding reads them as floats): the conversion in front of the forward, on the library's stream. */
int uwu_cast_i64_to_f32(const int64_t* x, float* y, int64_t n, void* stream);

/* Conditioning front-end (section 8f rank 4): ragged -> padded aggregation of per-caption embeddings.
 * Reference src/duwu/utils/aggregation.py:6-171.  `starts` = device int32 [B+1] prefix sums of n_elements; one "unit"
 * is one caption's [seq, ...] block of unit_bytes contiguous bytes (any dtype: pure byte movement).
 *   concat: out[b, 0 : n_b * unit] = emb[starts[b] : starts[b+1]], the rest of the max_n * unit row = pad pattern
 *           (pad_bits = the pad value's bit pattern in the low elem_size bytes)       aggregation.py:15-39,64-108
 *   split:  inverse of concat (gathers the valid prefix of every padded row)           aggregation.py:111-171
 *   first:  out[b] = emb[starts[b]]                                                    aggregation.py:174-185 */
int uwu_aggregate_concat(const void* emb, const int* starts, void* out, int B, int max_n, int64_t unit_bytes,
                         int elem_size, uint64_t pad_bits, void* stream);
int uwu_aggregate_split(const void* cat, const int* starts, void* out, int B, int max_n, int64_t unit_bytes,
                        void* stream);
int uwu_aggregate_first(const void* emb, const int* starts, void* out, int B, int64_t unit_bytes, void* stream);

/* Sampler (section 8f rank 1).  One Euler-ancestral step with classifier-free guidance, fused:
 *   eps = uncond + (cond-uncond)*cfg  (reference sampling/cfg.py:113-125; eps_uncond NULL -> no guidance)
 *   denoised = x - sigma*eps          (sampling/k_diffusion_wrapper.py:98-108)
 *   x' = x + eps*(sigma_down - sigma) + noise*s_noise*sigma_up   (sampling/k_diffusion_euler.py:42-47)
 * denoised may be NULL.  uwu_scale_copy: y = x*scale (the c_in = 1/sqrt(sigma^2+1) input scaling). */
int uwu_sampler_step(const float* x, const float* eps_cond, const float* eps_uncond, const float* noise, float* out,
                     float* denoised, int64_t n, float cfg, float sigma, float sigma_down, float sigma_up,
                     float s_noise, void* stream);
/* General sampler update for the DPM-Solver-2 and CFG++ variants (sampling/k_diffusion_dpm2.py:8-111,
 * k_diffusion_euler.py:51-106): every step of those samplers is
 *   out = base + a * eps_cfg + b * eps_uncond + c * noise,   eps_cfg = eps_uncond + (eps_cond - eps_uncond) * cfg
 * for scalars (a, b, c) derived from the sigmas on the host (denoised = x - sigma eps; to_d(x, sigma, denoised) = eps).
 * eps_uncond NULL: unguided (eps_uncond := eps_cond).  noise NULL: no noise term. */
int uwu_sampler_combine(const float* base, const float* eps_cond, const float* eps_uncond, const float* noise,
                        float* out, int64_t n, float cfg, float a, float b, float c, void* stream);
int uwu_scale_copy(const float* x, float* y, int64_t n, float scale, void* stream);

/* ------------------------------------------------------------------ optimizer (a15) */

/* Global L2 norm of a flat fp32 buffer (Lightning gradient_clip_val, demo_training.yaml:12):
 * out[0] = sum(g^2) * pre_scale^2 ; out[1] = clip coefficient min(1, max_norm/(sqrt(out[0])+1e-6))
 * (max_norm <= 0 -> 1).  partial: workspace of >= 1024 floats. */
int uwu_grad_sqnorm_clip(const float* g, int64_t n, float pre_scale, float max_norm, float* partial,
                         float* out, void* stream);

/* torch.optim.AdamW single-tensor semantics (trainer.py:52-74, demo_training_latent.yaml:30-39) on flat
 * buffers; g_eff = g * pre_scale * clip[1] (clip may be NULL).  Also refreshes the bf16 shadow
 * (may be NULL).  step is the 1-based step count. */
int uwu_adamw_step(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, float pre_scale,
                   const float* clip, int zero_grad, void* stream);

int uwu_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
int uwu_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream);

/* ------------------------------------------------------------------ gradient exchange (section 8e)
 * One RCCL communicator per rank (librccl resolved at run time), created once; uwu_allreduce_flat sums a slice of the
 * flat fp32 gradient buffer over the ranks in place on the caller's stream (reference: the bucketed all-reduce of
 * Lightning DDP, configs/demo_training.yaml:5-7).  uwu_comm_unique_id: 128 bytes made by rank 0, to be handed to every
 * rank through any side channel; uwu_comm_init is collective.  The 1/world average stays folded into the optimizer
 * kernels (pre_scale).  uwudiff_amd/gradsync.py uses it when UWU_RCCL_DIRECT=1 (default: torch.distributed's RCCL
 * process group, which needs no second communicator). */
int uwu_comm_unique_id(void* id128);
int uwu_comm_init(const void* id128, int rank, int world, void** comm);
int uwu_allreduce_flat(void* comm, float* buf, int64_t n, void* stream);
int uwu_comm_destroy(void* comm);

/* ------------------------------------------------------------------ dense contractions (a11-a13) */

/* C[M,N] = opA(A) . opB(B)  (+ epilogue), fp32 accumulate on MFMA.
 *   transA=0: A is [M,K] row-major (lda);  transA=1: A is [K,M] row-major (lda).
 *   transB=0: B is [N,K] row-major (ldb) -- torch Linear weight layout; transB=1: B is [K,N] (ldb).
 *   dtype: operand storage (UWU_F32 -> v_mfma_f32_16x16x4_f32, UWU_BF16 -> v_mfma_f32_16x16x32_bf16).
 *   c_dtype: storage of C / C2 (bf16 or fp32).  bias fp32[N].  aux: [M,N] operand-dtype tensor (ldaux).
 *   split_k > 1 requires UWU_EPI_ACCUM with fp32 C (atomic accumulation).
 * Replaces nn.Linear forward / dgrad / wgrad inside the denoiser (rope_unet.py:122-166). */
int uwu_gemm(const void* A, const void* B, void* C, void* C2, const float* bias, const void* aux,
             int M, int N, int K, int lda, int ldb, int ldc, int ldaux, int transA, int transB, int dtype,
             int c_dtype, int epilogue, int split_k, void* stream);

/* fp8 operands (OCP e4m3 / e5m2, BASELINE.json config 5 "fp8 MFMA GEMMs"; the reference has no fp8 path -- its
 * Linear is nn.Linear under bf16 autocast, rope_unet.py:122-166, demo_training_latent.yaml:6):
 *   C[M,N] = (A[M,K] . B[N,K]^T) / (scale_a * scale_b)  (+ epilogue), both operands contraction-contiguous fp8 bytes,
 * v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (2x the bf16 MFMA rate), fp32 accumulate.
 * A has element format fmt_a (UWU_FP8_E4M3 or UWU_FP8_E5M2: output gradients), B is e4m3.  scale_a / scale_b: device
 * floats, the per-tensor quantisation scales (x_fp8 = x * scale).  Epilogues NONE / BIAS / BIAS_GELU / DGELU write
 * bf16 C (C2, aux bf16 as in uwu_gemm); UWU_EPI_ACCUM adds into fp32 C through split-K partial sums in `scratch`
 * (uwu_gemm_fp8_scratch_bytes).  K % 128 == 0, leading dimensions multiples of 16 bytes. */
#define UWU_FP8_E4M3 0
#define UWU_FP8_E5M2 1
size_t uwu_gemm_fp8_scratch_bytes(int M, int N, int K);
int uwu_gemm_fp8(const void* A, const void* B, void* C, void* C2, const float* bias, const void* aux, int M, int N,
                 int K, int lda, int ldb, int ldc, int ldaux, int fmt_a, int epilogue, const float* scale_a,
                 const float* scale_b, void* scratch, size_t scratch_bytes, void* stream);

/* The same GEMM whose epilogue also EMITS the fp8 operand of the next GEMMs (delayed scaling: the scale of a tensor is
 * known before the tensor exists), so that no quantising pass reads the bf16 result again:
 *   UWU_EPI_BIAS_GELU: C = bf16 pre-activation A.B^T + bias (required: the backward pass reads it), q8 [M,N] / q8t [N,M] =
 *                      e4m3(gelu(pre-activation) * q_scale[0])
 *   UWU_EPI_DGELU:     v = (A.B^T) * gelu'(aux); C must be NULL (no bf16 copy); colsum[N] += column sums of v if non-NULL;
 *                      q8 / q8t = e5m2(v * q_scale[0])
 * q_amax (optional) receives max |value| by atomic max (the next step's scale).  Either of q8 / q8t may be NULL.
 * M, N multiples of 16; ldq >= N, ldqt >= M, multiples of 16.  Reference: the MLP of the transformer block,
 * rope_unet.py:399-411 (nn.Linear -> GELU(tanh) -> nn.Linear under autocast); the fp8 path is this build's (config 5). */
int uwu_gemm_fp8_emit(const void* A, const void* B, void* C, float* colsum, const float* bias, const void* aux, int M, int N,
                      int K, int lda, int ldb, int ldc, int ldaux, int fmt_a, int epilogue, const float* scale_a,
                      const float* scale_b, void* q8, int ldq, void* q8t, int ldqt, const float* q_scale, float* q_amax,
                      void* stream);

/* 3x3 convolution, padding 1, stride 1 or 2, channels-last bf16, as an implicit GEMM: the MFMA ring kernels gather
 * their activation operand pixel by pixel (per-lane LDS-DMA source addresses, a zero page for the padding) so no
 * im2col matrix exists in HBM (reference: the resblock / down / up-sample Conv2d of diffusers' UNet2DConditionModel,
 * src/duwu/modules/unet_patch.py:13-57).  x [B,H,W,C], w [Cout][3][3][C] (the layout of this build's flat parameter
 * blob), y / dy [B,Ho,Wo,Cout].  uwu_conv3x3_implicit_ok tells whether a shape is covered (bf16, C and Cout multiples
 * of 32, B*Ho*Wo a multiple of 32); other shapes use uwu_im2col3x3 + uwu_gemm.  wgrad accumulates into fp32 dw / db. */
int uwu_conv3x3_implicit_ok(int B, int H, int W, int C, int Cout, int stride, int dtype);
int uwu_conv3x3_fwd(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int C, int Cout,
                    int stride, int dtype, void* stream);
int uwu_conv3x3_dgrad(const void* dy, const void* w, void* dx, int B, int H, int W, int C, int Cout, int stride, int dtype,
                      void* stream);
size_t uwu_conv3x3_wgrad_scratch_bytes(int C, int Cout, int64_t Mo);
int uwu_conv3x3_wgrad(const void* dy, const void* x, float* dw, float* db, int B, int H, int W, int C, int Cout, int stride,
                      int dtype, void* scratch, size_t scratch_bytes, void* stream);

/* Per-tensor fp8 quantisation (quant.hip).  amax: device float, max |x| is folded in with an atomic max (caller zeroes
 * it, or uwu_fp8_update_scales does).  update_scales: scale[i] = FMT_MAX(fmt[i]) / (amax[i] * margin) where amax[i] > 0
 * (otherwise the previous scale, or 1), then amax[i] = 0 -- the "delayed scaling" step between two training steps, or
 * the second half of just-in-time scaling.  quantize: one pass over x [M,K] (bf16 / fp32, row-major, ldx) writing any
 * of out [M,K] (ldo), out_t [K,M] (ldt; the operand layout of a contraction over M), colsum[K] += column sums of x
 * (fp32 atomics; a bias gradient), amax.  Values are saturated to the format's largest finite number. */
int uwu_fp8_amax(const void* x, int dtype, int64_t n, float* amax, void* stream);
int uwu_fp8_update_scales(float* amax, float* scale, const int* fmt, int n, float margin, void* stream);
int uwu_fp8_quantize(const void* x, int dtype, int M, int K, int ldx, const float* scale, int fmt, void* out, int ldo,
                     void* out_t, int ldt, float* amax, float* colsum, void* stream);

/* Weight gradient of a Linear with caller-provided split-K scratch:  C[M,N] (fp32) += A[K,M]^T . B[K,N], both
 * operands K-major (A = dY [tokens, out], B = X [tokens, in]; reference: autograd of nn.Linear inside the blocks,
 * rope_unet.py:122-166).  bf16 operands with K % 32 == 0 run the streaming kernel (LDS-DMA + transposing LDS
 * reads, K split into a multiple of 8 slices, one group of slices per XCD); with `scratch` of at least
 * uwu_gemm_wgrad_scratch_bytes(M, N, K) bytes the slices are written there and summed by a second kernel, with
 * scratch == NULL they are accumulated with fp32 atomics.  Other shapes / fp32 operands are
 * uwu_gemm(transA=1, transB=1, UWU_EPI_ACCUM) with `blocks` workgroups as the split-K target.
 * bias_grad (optional, fp32[M]) += column sums of A = the Linear's bias gradient; the streaming kernel computes it
 * with extra MFMAs against an all-ones fragment instead of a separate pass over dY.
 * Results differ only in summation order. */
size_t uwu_gemm_wgrad_scratch_bytes(int M, int N, int K);
int uwu_gemm_wgrad(const void* A, const void* B, float* C, float* bias_grad, int M, int N, int K, int lda, int ldb,
                   int ldc, int dtype, int blocks, void* scratch, size_t scratch_bytes, void* stream);

/* Live measurement for bench.py's `roofline` object: when enabled, every uwu_gemm launch is bracketed by a HIP
 * event pair recorded on that launch's stream; collect() returns the summed launch duration, the summed
 * algorithmic FLOPs (2*M*N*K) and the launch count for operand kind 0 (bf16) or 1 (fp32). */
int uwu_gemm_prof_enable(int on);
int uwu_gemm_prof_collect(int kind, double* ms, double* flops, int* launches);

/* The same, per kernel family (measurement harness): tag = one of UWU_PROF_* (or < 0: all GEMM
 * tags), kind 0 = bf16 / fp8 operands, 1 = fp32, < 0 = any.  flops / bytes are the ALGORITHMIC work of the recorded
 * launches (GEMM 2*M*N*K and operands + outputs once; attention 4*T*Tk*d per head forward, x2.5 backward, and
 * q/k/v/o (+ gradients) once; LayerNorm: every tensor it reads or writes once). */
enum {
  UWU_PROF_GEMM_FWD = 0, UWU_PROF_GEMM_DGRAD = 1, UWU_PROF_GEMM_WGRAD = 2, UWU_PROF_GEMM_FC1_GELU = 3,
  UWU_PROF_GEMM_FC2_DGELU = 4, UWU_PROF_ATTN_FWD = 5, UWU_PROF_ATTN_BWD = 6, UWU_PROF_LN_FWD = 7, UWU_PROF_LN_BWD = 8,
  UWU_PROF_CONV = 9, UWU_PROF_OTHER = 10
};
int uwu_prof_enable(int on);
int uwu_prof_collect(int tag, int kind, double* ms, double* flops, double* bytes, int* launches);

/* out[n] (+)= sum_m X[m,n]   (bias gradients). accumulate: 0 overwrite, 1 add. */
int uwu_colsum(const void* X, int dtype, int M, int N, int ldx, float* out, int accumulate, void* stream);

/* fp32 Linears with at most 64 rows (the conditioning path: timestep MLP, pooled-text projection, the adaLN modulation
 * Linear of all blocks -- reference src/duwu/modules/rope_unet.py:306-309,393-411 run these as nn.Linear on [B, *]).
 * Matrix-vector layouts that read / update the weight matrix once (csrc/skinny.hip).  uwu_skinny_linear_ok() says whether
 * a shape is covered (M <= 64, X [M, K] fits LDS); dgrad additionally needs K <= 512.
 *   fwd:   Y[M,N] = X[M,K] . W[N,K]^T (+ bias) ; epilogue UWU_EPI_NONE / _BIAS / _BIAS_SILU (Y2 = silu(Y))
 *   dgrad: dX[M,K] = dY[M,N] . W[N,K]           (dX overwritten)
 *   wgrad: dW[N,K] += dY^T . X ; db[N] += column sums of dY (db may be NULL) */
int uwu_skinny_linear_ok(int M, int N, int K);
int uwu_skinny_linear_fwd(const float* X, const float* W, const float* bias, float* Y, float* Y2, int M, int N, int K,
                          int epilogue, void* stream);
int uwu_skinny_linear_dgrad(const float* dY, const float* W, float* dX, int M, int N, int K, void* stream);
int uwu_skinny_linear_wgrad(const float* dY, const float* X, float* dW, float* db, int M, int N, int K, void* stream);
/* batched: out[b, n] (+)= sum_m X[b, m, n] for `batch` contiguous [M, ldx] slabs. */
int uwu_colsum_batched(const void* X, int dtype, int batch, int M, int N, int ldx, float* out, int accumulate,
                       void* stream);

/* ------------------------------------------------------------------ norms / modulation (a13) */

/* adaLN-Zero pre-norm with fused residual update (rope_unet.py:306-309, 344-349, 393-411):
 *   x_out = x_in + gate_b * y            (y/gate may be NULL: x_out = x_in, not written if x_out==x_in)
 *   h     = LayerNorm(x_out; eps, no affine) * (1 + scale_b) + shift_b          (affine = 0, adaLN-Zero)
 *   h     = LayerNorm(x_out; eps) * scale + shift   with mod_ld = 0: one shared (gamma, beta) row  (affine = 1,
 *           the plain pre-LN of the SDXL transformer blocks; dscale/dshift then are dgamma/dbeta)
 * x_*, y, h: [B*T, D] in `dtype`; shift/scale/gate: fp32 rows of a [B, mod_ld] buffer; mean/rstd fp32[B*T]. */
int uwu_add_ln_modulate_fwd(const void* x_in, const void* y, const float* gate, const float* shift,
                            const float* scale, int mod_ld, void* x_out, void* h, float* mean, float* rstd,
                            int B, int T, int D, float eps, int affine, int dtype, void* stream);
/* fp8 mode, delayed scaling: the same LayerNorm whose output leaves ONLY as the e4m3 operand of the next GEMMs -- q8 [B*T, D]
 * (contraction over D) and q8t [D, B*T] (the weight gradient's contraction over the tokens) = sat(h * q_scale[0]); q_amax
 * (optional) receives max |h|.  bf16 tensors, per-sample shift / scale required, B*T a multiple of 64, D <= 1536. */
int uwu_add_ln_modulate_fwd_q8(const void* x_in, const void* y, const float* gate, const float* shift, const float* scale,
                               int mod_ld, void* x_out, void* q8, int ldq, void* q8t, int ldqt, const float* q_scale,
                               float* q_amax, float* mean, float* rstd, int B, int T, int D, float eps, void* stream);


/* Backward of the above, fused with the residual/gate backward of the branch that feeds x:
 *   dx_out = dx_in + LN_bwd(dh * (1+scale_b))        (dx_in may be NULL for the last norm)
 *   dy     = gate_b * dx_out                          (if y given)
 *   dshift_b += sum_t dh ; dscale_b += sum_t dh*xhat ; dgate_b += sum_t dx_out*y   (fp32 atomics)
 * dmod rows live in a zero-initialised [B, mod_ld] fp32 buffer. */
int uwu_add_ln_modulate_bwd(const void* dh, const void* x, const float* mean, const float* rstd,
                            const float* scale, const void* dx_in, const void* y, const float* gate,
                            int mod_ld, void* dx_out, void* dy, float* dshift, float* dscale, float* dgate,
                            int B, int T, int D, int affine, int dtype, void* stream);

/* ------------------------------------------------------------------ attention (a12) */

/* o = softmax(scale * Q K^T) V, non-causal, no mask (rope_unet.py:151-153 F.scaled_dot_product_attention;
 * diffusers AttnProcessor2_0 for the stock UNet).  Row (b,t) of q lives at q + (b*Tq+t)*ldq + h*d, likewise
 * k/v with Tk, ldk/ldv and o with ldo -- so a packed [B*T, 3*H*d] projection is addressed in place with
 * q=base, k=base+H*d, v=base+2*H*d, ld=3*H*d, and cross-attention (Tk=77) uses separate tensors.
 * lse: fp32 [B,H,Tq] log-sum-exp of the scaled scores (saved for backward). */
int uwu_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int Tq, int Tk,
                      int H, int d, int ldq, int ldk, int ldv, int ldo, float scale, int dtype, void* stream);
/* dq/dk/dv use the strides of q/k/v; dO uses ldo.  delta: fp32 workspace [B,H,Tq] (rowsum(dO*O)). */
int uwu_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* dO,
                      const float* lse, float* delta, void* dq, void* dk, void* dv, int B, int Tq, int Tk, int H,
                      int d, int ldq, int ldk, int ldv, int ldo, float scale, int dtype, void* stream);
/* The same with an additive score bias per key: o = softmax(scale * Q K^T + key_bias[b, :]) V.  key_bias: fp32
 * [B, Tk], shared by every head and query row and not differentiated -- this is how the reference applies
 * `encoder_attention_mask` to cross-attention: `(1 - mask) * -10000.0`, unsqueezed over the query axis
 * (rope_unet.py:440-453) and broadcast over heads by `prepare_attention_mask` (rope_unet.py:106-114) before
 * F.scaled_dot_product_attention(attn_mask=...) (rope_unet.py:151-153).  Finite values only. */
int uwu_attention_bias_fwd(const void* q, const void* k, const void* v, const float* key_bias, void* o, float* lse,
                           int B, int Tq, int Tk, int H, int d, int ldq, int ldk, int ldv, int ldo, float scale,
                           int dtype, void* stream);
int uwu_attention_bias_bwd(const void* q, const void* k, const void* v, const float* key_bias, const void* o,
                           const void* dO, const float* lse, float* delta, void* dq, void* dk, void* dv, int B,
                           int Tq, int Tk, int H, int d, int ldq, int ldk, int ldv, int ldo, float scale, int dtype,
                           void* stream);

/* Axial RoPE with learnable per-head log-frequencies, exactly as the reference writes it (src/duwu/modules/rope.py:
 * 56-71, 83-108; applied to q and k at rope_unet.py:143-147).  x/y: [rows, ldx] token-major with H heads of width d;
 * pos: fp32 [rows, 2] (h, w); fh/fw: fp32 [H, d/4].  bwd: dx, and (optional) dfh/dfw += gradient wrt log-freqs. */
int uwu_axial_rope_fwd(const void* x, const float* pos, const float* fh, const float* fw, void* y, int64_t rows,
                       int H, int d, int ldx, int dtype, void* stream);
int uwu_axial_rope_bwd(const void* x, const void* dy, const float* pos, const float* fh, const float* fw, void* dx,
                       float* dfh, float* dfw, int64_t rows, int H, int d, int ldx, int dtype, void* stream);

/* Axial RoPE folded into self-attention (rope_unet.py:143-153: RoPE on q and k, then SDPA): the attention kernels multiply
 * q and k by the factor table tab [T, ldt] (fp32; head h's d columns at h*d; shared positions pos [T,2]) while staging
 * them, so no rotated copy of q / k exists.  bf16, head dim 64, T == Tk a multiple of 64 up to 256 (the DiT shapes).
 * uwu_attention_rope_bwd returns dq / dk wrt the ROTATED q' / k'; uwu_axial_rope_bwd (x = the raw q / k, dy = those, dx in
 * place) turns them into the gradients of q / k and of the log-frequencies. */
int uwu_axial_rope_bwd_shared(const void* x, const void* dy, const float* pos, int pos_rows, const float* fh, const float* fw,
                              void* dx, float* dfh, float* dfw, int64_t rows, int H, int d, int ldx, int dtype, void* stream);
int uwu_axial_rope_table(const float* pos, const float* fh, const float* fw, float* tab, int T, int H, int d, int ldt,
                         void* stream);
int uwu_attention_rope_fwd(const void* q, const void* k, const void* v, const float* rope_tab, void* o, float* lse, int B,
                           int T, int H, int d, int ldq, int ldk, int ldv, int ldo, int ldt, float scale, int dtype,
                           void* stream);
int uwu_attention_rope_bwd(const void* q, const void* k, const void* v, const float* rope_tab, const void* o, const void* dO,
                           const float* lse, void* dq, void* dk, void* dv, int B, int T, int H, int d, int ldq, int ldk,
                           int ldv, int ldo, int ldt, float scale, int dtype, void* stream);

/* ------------------------------------------------------------------ embeddings / layout (a11, a13) */

/* Sinusoidal timestep features [B, dim]: [cos(t*f_i) | sin(t*f_i)], f_i = exp(-ln(max_period)*i/half). */
int uwu_timestep_embedding(const float* t, int B, int dim, float max_period, void* out, int dtype,
                           void* stream);
/* y = silu(x) elementwise (dtype in/out); dx = dy * silu'(x). */
int uwu_silu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream);
int uwu_silu_bwd(const void* x, const void* dy, void* dx, int64_t n, int dtype, void* stream);

/* out = a + b elementwise (same dtype). */
int uwu_add(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream);
/* dst[c][r] = src[r][c], bf16 (a weight matrix laid out contraction-contiguous for the input-gradient GEMM) */
int uwu_transpose_bf16(const void* src, void* dst, int rows, int cols, int ld_src, int ld_dst, void* stream);

/* patchify: latent [B,C,H,W] (fp32) -> tokens [B*(H/p)*(W/p), C*p*p] (dtype), feature order (c,ph,pw).
 * unpatchify is the inverse (tokens -> fp32 image).  Conv2d(k=p,s=p) patch embedding == patchify + GEMM. */
int uwu_patchify(const float* img, void* tok, int B, int C, int H, int W, int p, int dtype, void* stream);
int uwu_unpatchify(const void* tok, int dtype, float* img, int B, int C, int H, int W, int p, void* stream);
/* x[b,t,:] += pos[t,:]  (fixed 2-D sin-cos table, fp32 [T,D]) */
int uwu_add_pos(void* x, const float* pos, int B, int T, int D, int dtype, void* stream);

/* ------------------------------------------------------------------ UNet2DConditionModel-shape denoiser (a11) */
/* Channels-last activations x[b, p, c] (p = y*W + x).  Replace the torch/diffusers op sequences of
 * ResnetBlock2D / Transformer2DModel / Down-/Upsample2D (reference src/duwu/modules/unet_patch.py:13-57). */

/* y = GroupNorm(x; G groups, eps, gamma, beta) [then SiLU if silu]; mean/rstd: fp32 [B*G]. */
int uwu_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                      int B, int HW, int C, int G, float eps, int silu, int dtype, void* stream);
/* dgamma/dbeta are accumulated (fp32 atomics).  ws: fp32 scratch of 2*B*G floats (per-group sums).
 * Both directions: C % 8 == 0, C <= 4096, 16-byte aligned tensors (whole-row vector accesses). */
int uwu_groupnorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                      const float* beta, void* dx, float* dgamma, float* dbeta, float* ws, int B, int HW, int C, int G,
                      int silu, int dtype, void* stream);
/* 3x3 / padding 1 / stride 1|2 convolution = im2col + uwu_gemm: col[(b,oy,ox), (ky,kx,c)]; col2im is the adjoint
 * in gather form (no atomics). */
int uwu_im2col3x3(const void* x, void* col, int B, int H, int W, int C, int stride, int dtype, void* stream);
int uwu_col2im3x3(const void* dcol, void* dx, int B, int H, int W, int C, int stride, int dtype, void* stream);
/* GEGLU feed-forward gate: hg [M, 2F] = (h | gate) -> out [M, F] = h * gelu_erf(gate), and its backward. */
int uwu_geglu_fwd(const void* hg, void* out, int64_t M, int F, int dtype, void* stream);
int uwu_geglu_bwd(const void* hg, const void* dout, void* dhg, int64_t M, int F, int dtype, void* stream);
/* nearest-neighbour 2x upsample [B,H,W,C] -> [B,2H,2W,C]; backward=1: src is d(out) [B,2H,2W,C], dst d(in). */
int uwu_upsample2x(const void* src, void* dst, int B, int H, int W, int C, int backward, int dtype, void* stream);
/* x[b, p, :] += v[b, :] (time-embedding injection). */
int uwu_add_rowvec(void* x, const void* v, int B, int HW, int C, int dtype, void* stream);
/* NCHW fp32 <-> channels-last (dtype). */
int uwu_nchw_to_cl(const float* nchw, void* cl, int B, int C, int HW, int dtype, void* stream);
int uwu_cl_to_nchw(const void* cl, float* nchw, int B, int C, int HW, int dtype, void* stream);

/* ------------------------------------------------------------------ whole-network drivers */

/* DiT forward/backward: all kernel launches of one network pass issued from C++ (no per-op
 * host round trip).  The argument block is a plain C struct of device pointers and sizes. */
typedef struct uwu_dit_desc {
  int32_t B, T, D, H, L, mlp_ratio, in_ch, out_ch, patch, img, dtype, cond_dim, freq_dim;
  float ln_eps;
  int32_t mod_total;   /* = L*6*D + 2*D : all adaLN linears batched in one GEMM */
  /* parameter blob (operand dtype copy for GEMM operands, fp32 master for biases) */
  const void* w;        /* operand-dtype flat parameters  */
  const float* w32;     /* fp32 master flat parameters    */
  float* g32;           /* fp32 flat gradient buffer (accumulated into) */
  /* offsets (in elements) into the flat blobs */
  int64_t off_patch_w, off_patch_b, off_t_w1, off_t_b1, off_t_w2, off_t_b2, off_y_w, off_y_b,
      off_mod_w, off_mod_b, off_final_w, off_final_b;
  int64_t off_layer0, layer_stride; /* per layer: qkv_w, qkv_b, o_w, o_b, fc1_w, fc1_b, fc2_w, fc2_b */
  const float* pos;     /* [T, D] fp32 */
  void* ws;             /* activation workspace (saved for backward) */
  size_t ws_bytes;
  void* const* layer_done; /* optional: L hipEvent_t handles (or NULL).  uwu_dit_backward records layer_done[l] on its
                              stream once every parameter-gradient launch of block l has been issued, so a data-parallel
                              host can start reducing that block's slice of g32 while the backward continues
                              (blocks finish in the order L-1 .. 0; the non-block parameters finish last). */
  /* fp8 Linears (BASELINE config 5): 0 = off; 1 = just-in-time per-tensor scaling (amax pass, then quantise); 2 = delayed
   * scaling (quantise with the scale derived from the previous step's amax; call with 1 for the first step).  Needs
   * dtype == UWU_BF16, D % 128 == 0.  Roles (12 per block, index 12*l + r): r 0-3 = inputs of qkv / proj / fc1 / fc2
   * (e4m3), 4-7 = their output gradients (e5m2), 8-11 = their weights (e4m3). */
  /* optional second stream (hipStream_t) for the weight gradients of small batches (B*T <= 16384 token rows, bf16): they
   * run beside the chain of input gradients, ordered by events; NULL = everything on the call's stream. */
  void* side_stream;
  /* axial-RoPE self-attention (rope_unet.py:143-147): rope != 0 rotates q and k inside the
   * attention kernels with learnable per-layer, per-head log-frequencies at w32 + off_rope_h / off_rope_w ([L, H, d/4]
   * each; gradients into g32 at the same offsets) and the shared token positions pos_xy [T, 2] (fp32). */
  int32_t rope;
  int64_t off_rope_h, off_rope_w;
  const float* pos_xy;
  int32_t fp8;
  float* f8_scale;      /* [12 L] per-tensor quantisation scales */
  float* f8_amax;       /* [12 L] running max |x| of the current step */
  const int32_t* f8_fmt; /* [12 L] UWU_FP8_E4M3 / UWU_FP8_E5M2 per role */
  /* block recomputation (the reference's enable_gradient_checkpointing, test_scripts/test_train.py:38-39; diffusers
   * rope_unet.py:484-507 checkpoints per transformer block): != 0 keeps per block only its input x0 and the row statistics
   * of its first LayerNorm; every other saved activation lives in ONE slab shared by all blocks, and uwu_dit_backward reruns
   * block l - 1 from its kept input before it needs that block's tensors.  Workspace: L * (M*D + 8 M) + one full block slab
   * instead of L full slabs.  Same kernels on the same inputs: the recomputed tensors are bit-identical. */
  int32_t checkpoint;
} uwu_dit_desc;

size_t uwu_dit_workspace_bytes(const uwu_dit_desc* d);
/* noisy: fp32 [B,C,H,W]; t: fp32 [B]; cond: fp32 [B, cond_dim] or NULL; out: fp32 [B,out_ch,H,W] */
int uwu_dit_forward(const uwu_dit_desc* d, const float* noisy, const float* t, const float* cond,
                    float* out, void* stream);
/* dout: fp32 [B,out_ch,H,W] gradient of the loss wrt the network output; parameter gradients are ACCUMULATED
 * into d->g32 (zero it at the start of a step).  uwu_dit_backward_cond adds the gradient of the pooled-
 * conditioning projection (it needs the caller's `cond` tensor again). */
int uwu_dit_backward(const uwu_dit_desc* d, const float* dout, void* stream);
int uwu_dit_backward_cond(const uwu_dit_desc* d, const float* cond, void* stream);
/* elements per transformer block in the flat parameter blob (every tensor padded to 64 elements):
 * qkv_w[3D,D] qkv_b[3D] o_w[D,D] o_b[D] fc1_w[rD,D] fc1_b[rD] fc2_w[D,rD] fc2_b[D] */
int64_t uwu_dit_layer_param_stride(int D, int mlp_ratio);

#ifdef __cplusplus
}
#endif
#endif /* UWU_HIP_H */
