/* dm_hip.h - C ABI of libdm_hip.so, the MI355X (gfx950) kernel library under
 * diffusion_model_amd.
 *
 * The reference (barebucan/Diffusion_model) has NO native / FFI boundary: its hot path is
 * torch.nn modules composed in unet.py plus ~60 lines of schedule / sampler / loss arithmetic.
 * Every entry point below therefore cites the reference *Python* statement whose ATen work it
 * replaces (file:line into the reference checkout).  The Python host side
 * (the diffusion_model_amd Python package) binds these with ctypes; INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless noted.
 *   - activations: NHWC bf16, addressed as [pixel][channel] with an explicit pixel stride
 *     (ld*, in elements) so slices of a concat buffer can be produced / consumed in place.
 *   - weights: fp32 master in the reference layout (OIHW / [out,in]); kernels consume a bf16
 *     shadow packed by dm_pack_weight.
 *   - return 0 (DM_OK) or a negative error code; never throws; no allocation; no implicit
 *     synchronisation; stream-ordered on `stream` (a hipStream_t passed as void*).
 */
#ifndef DM_HIP_H
#define DM_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dm_stream_t;

#define DM_OK 0
#define DM_ERR_ARG (-1)
#define DM_ERR_LAUNCH (-2)
#define DM_ERR_TOO_LARGE (-3)   /* a size beyond an entry point's documented fixed maximum */

/* ABI version / arch string ("gfx950").  Version 2 (round 2): dm_conv_args / dm_wgrad_args grew the trailing
 * norm_scale / norm_shift fields.  Version 3: dm_attn_args grew the trailing workspace / ws_bytes fields and
 * dm_attn_bwd_workspace_bytes, dm_film_fwd / dm_film_bwd were added.  Version 4 (round 3): the fp32 compute mode (the *_f32
 * entry points at the end of this header); nothing older changed.  Callers built against an older version must be rebuilt. */
int dm_abi_version(void);
const char* dm_target_arch(void);
/* Text of the HIP error behind the calling thread's last DM_ERR_LAUNCH ("" if none). */
const char* dm_last_hip_error(void);
void dm_note_hip_error(int code);
/* Name of the kernel instantiation the calling thread's last dm_conv3x3 / dm_gemm / dm_wgrad / dm_attn_* call
 * launched, e.g. "conv_mfma<TH=16,TW=16,S=1,NT=8,MODE=2>" (NT = 16-cout tiles per workgroup; MODE 0 s1, 1 s2,
 * 2 nearest-x2, 3 zero-insert, 4 implicit-GEMM 1x1, 5 four 8x8 images per tile), "gemm_stream<MT=2,...>",
 * "wgrad<...>".  The dispatch is size-dependent (unet.py:74-75,225-229,234,317-320 all map onto one entry point),
 * so parity tests assert through this hook that they verified the instantiation the B=64 training step runs.
 * Test/diagnostic hook: no reference counterpart. */
const char* dm_last_variant(void);

/* Dropout under HIP-graph replay.  Every dropout site (nn.Dropout of the ResBlock, unet.py:76,105; the attention
 * dropout_p of F.scaled_dot_product_attention, unet.py:136,214) takes its seed as a kernel ARGUMENT, which a captured
 * graph replays unchanged.  When a device counter is registered here, every such kernel adds it to its (hashed) seed
 * at run time, so the caller bumps it once per micro-step (dm_bump_dropout_counter: one tiny launch, capturable) and
 * every replay draws fresh masks -- while the forward and backward of one micro-step still agree on them.
 * dev_counter = NULL unregisters.  Library-global state: one process drives one GPU (one rank per GPU). */
int dm_set_dropout_counter(const unsigned int* dev_counter);
int dm_bump_dropout_counter(unsigned int* dev_counter, dm_stream_t stream);

/* ------------------------------------------------------------------ weights */
/* Bytes of the packed bf16 shadow of an [N][K][taps] weight (N padded to 16, K to 64). */
long dm_packed_weight_bytes(int N, int K, int taps);
/* Pack fp32 weights into MFMA-fragment order.  taps = 9 (3x3) or 1 (1x1 / Linear).
 * transpose = 0: forward operand from W[N][K][kh][kw];
 * transpose = 1: data-gradient operand: W is [K][N][kh][kw], taps are flipped (2-kh, 2-kw).
 * Replaces: the implicit weight reads of nn.Conv2d / nn.Linear (unet.py:74-75,79,113-114,
 * 160-163,225,234,261,317-320,333-335). */
int dm_pack_weight(const float* w, void* dst, int N, int K, int taps, int transpose, dm_stream_t stream);
/* The same for a whole table of weights in ONE launch (the repack after every optimizer step,
 * main_stl10.py:509).  The table lives in DEVICE memory; entry i owns blocks [block_begin, block_begin+nblocks)
 * of the launch (256 threads each, ascending, contiguous from 0), total_blocks = their sum. */
typedef struct dm_pack_desc {
  const float* w; void* dst;
  int N, K, taps, transpose;      /* as in dm_pack_weight */
  int block_begin, nblocks;
} dm_pack_desc;
int dm_pack_weights_batch(const dm_pack_desc* table_dev, int n, int total_blocks, dm_stream_t stream);

/* ------------------------------------------------------------------ convolution / GEMM */
typedef struct dm_conv_args {
  const void* x; long ldx;        /* input NHWC bf16 [B,H,W,Cin], pixel stride ldx */
  const void* w;                  /* packed weights (dm_pack_weight, taps=9) */
  const float* bias;              /* [Cout] fp32 or NULL */
  const float* chan_add;          /* [B,Cout] fp32 added per (sample, channel) (time-embedding) or NULL */
  const void* res; long ldr;      /* bf16 residual added to the output or NULL */
  void* y; long ldy;              /* bf16 output or NULL */
  float* yf; long ldyf;           /* fp32 output or NULL */
  int B, H, W, Cin, Cout;         /* H,W = INPUT spatial size */
  int mode;                       /* 0: 3x3 s1 p1; 1: 3x3 s2 p1; 2: nearest-x2 then 3x3 s1 p1;
                                     3: zero-insert-x2 then 3x3 s1 p1 (data-gradient of mode 1) */
  int n_valid;                    /* store only the first n_valid output channels (0 = all) */
  int pool2;                      /* sum 2x2 output pixels into one (data-gradient of mode 2) */
  int act_silu_in;                /* apply SiLU to the input while staging */
  float drop_p;                   /* dropout on (conv+bias+chan_add) before the residual add */
  unsigned long long drop_seed;
  const float* norm_scale;        /* [B,Cin] fp32 or NULL: the conv reads silu(x * norm_scale + norm_shift) -- GroupNorm + */
  const float* norm_shift;        /* SiLU folded into the consumer (dm_gn_prepare); zero padding stays zero */
} dm_conv_args;
/* Implicit-GEMM 3x3 convolution on bf16 MFMA, LDS-staged input halo.
 * Replaces: nn.Conv2d 3x3 (unet.py:74-75,234,317-319), F.interpolate(nearest x2)+conv
 * (unet.py:227-229), the `h + time_emb_proj` add (unet.py:101-102), nn.Dropout
 * (unet.py:105), the ResBlock / UnetLayer residual adds (unet.py:108,274) and, with
 * transposed+flipped weights, the autograd data-gradient of the same convolutions. */
int dm_conv3x3(const dm_conv_args* args, dm_stream_t stream);

typedef struct dm_gemm_args {
  const void* x; long ldx;        /* [M,K] bf16 */
  const void* w;                  /* packed weights (taps=1) */
  const float* bias; const float* chan_add;
  const void* res; long ldr;
  void* y; long ldy;
  float* yf; long ldyf;
  long M; int K, N;
  int rows_per_batch;             /* chan_add row = m / rows_per_batch */
  int n_valid; int act_silu_in;
} dm_gemm_args;
/* Y[M,N] = X[M,K] W[N,K]^T (+bias)(+chan_add)(+res).  Replaces nn.Conv2d 1x1 and nn.Linear
 * (unet.py:79,113-114,160-163,261,320,87-90,333-335) and their data-gradients. */
int dm_gemm(const dm_gemm_args* args, dm_stream_t stream);

typedef struct dm_wgrad_args {
  const void* x; long ldx;        /* forward input NHWC bf16 [B,H,W,Cin] ([M,K] for taps=1) */
  const void* dy; long lddy;      /* output gradient NHWC bf16 [B,Ho,Wo,Cout] ([M,N]) */
  float* dw;                      /* fp32 gradient in the REFERENCE layout [Cout][Cin][kh][kw]; accumulated (+=) */
  float* dbias;                   /* [Cout] fp32 accumulated (+=) or NULL */
  int B, H, W, Cin, Cout;         /* H,W = forward INPUT size; for taps=1: B*H*W = M rows */
  int taps;                       /* 9 or 1 */
  int mode;                       /* as dm_conv_args.mode (0,1,2) */
  int act_silu_in;                /* forward applied SiLU to x while staging */
  int n_valid, k_valid;           /* real Cout / Cin to write into dw (0 = all) */
  void* workspace; long ws_bytes; /* >= dm_wgrad_workspace_bytes */
  const float* norm_scale;        /* as dm_conv_args.norm_scale / norm_shift: the forward conv read */
  const float* norm_shift;        /* silu(x * norm_scale + norm_shift) (taps = 9, mode 0 only) */
} dm_wgrad_args;
long dm_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int taps, int mode);
/* Weight (and bias) gradient of the convolutions / linears above: split-K MFMA with
 * transposed LDS reads, deterministic two-stage reduction (partial slabs summed in a fixed order, no atomics:
 * the same inputs give the same bits; dw / dbias are read-modify-written, so two calls that target the same dw
 * must be stream-ordered).  Replaces autograd's
 * conv/linear weight-gradient for scaled_loss.backward() (main_stl10.py:502-503). */
int dm_wgrad(const dm_wgrad_args* args, dm_stream_t stream);

/* ------------------------------------------------------------------ GroupNorm + SiLU */
long dm_gn_workspace_bytes(int B, int HW, int C, int G);
/* y = act(GroupNorm(x + chan_add)) ; stats[b][g] = {mean, rstd} saved for the backward.  C <= 1024, a power-of-two
 * multiple of 8 (C/8 divides 256), G <= 64, C/G a multiple of 8.
 * Replaces nn.GroupNorm(8,C) + nn.SiLU (+ the time-embedding add) (unet.py:77-78,94,101-102,106). */
int dm_gn_silu_fwd(const void* x, long ldx, const float* chan_add, const float* gamma, const float* beta,
                   void* y, long ldy, float* stats, int B, int HW, int C, int G, float eps, int act_silu,
                   void* workspace, long ws_bytes, dm_stream_t stream);
/* GroupNorm folded into its consumer: statistics of x (+ chan_add) per (sample, group) in ONE pass over x, then
 * stats[b][g] = {mean, rstd} and the per-(sample, channel) affine  scale = gamma * rstd,
 * shift = beta + (chan_add - mean) * scale  ([B][C] fp32 each), so that a convolution given norm_scale / norm_shift
 * reads silu(x * scale + shift) = SiLU(GroupNorm(x)) while staging its input patch -- the normalised activation is
 * never written to memory (4 -> 2 bytes per element for the forward GroupNorm; unet.py:94,106 feeding :95,107).
 * Same constraints and workspace as dm_gn_silu_fwd. */
int dm_gn_prepare(const void* x, long ldx, const float* chan_add, const float* gamma, const float* beta,
                  float* stats, float* scale, float* shift, int B, int HW, int C, int G, float eps,
                  void* workspace, long ws_bytes, dm_stream_t stream);
/* dx = d/dx of the above (+ add_in), optional dropout mask (the mask dm_conv3x3 applied to x);
 * dgamma/dbeta accumulated (+=).  All three reductions (dgamma, dbeta, dx_colsum) are summed in a fixed order through
 * workspace partials (no atomics): the same inputs give the same bits.  Groups of 16 channels and more whose (sample, group)
 * slab is at most 72 KiB take a one-pass kernel (x and dy read once: 6 instead of 10 bytes per element); the result differs
 * from the two-pass kernels' only through the summation order of the group means.
 * dx_colsum (optional, [B][C] fp32, accumulated +=): per-sample channel sums
 * of the written dx = the gradient of the time-embedding projection / conv1 bias of the ResBlock
 * (unet.py:101-102), produced in the same pass instead of a separate reduction over dx. */
int dm_gn_silu_bwd(const void* x, long ldx, const float* chan_add, const void* dy, long lddy,
                   const float* gamma, const float* beta, const float* stats, const void* add_in, long ldadd,
                   void* dx, long lddx, float* dgamma, float* dbeta, int B, int HW, int C, int G, int act_silu,
                   float drop_p, unsigned long long drop_seed, float* dx_colsum, void* workspace, long ws_bytes,
                   dm_stream_t stream);

/* ------------------------------------------------------------------ attention */
typedef struct dm_attn_args {
  const void* q; long ldq;        /* [B, Nq, heads*hd] bf16, token stride ldq */
  const void* k; long ldk;        /* [B, Nk, heads*hd] */
  const void* v; long ldv;
  void* o; long ldo;              /* [B, Nq, heads*hd] */
  float* lse;                     /* [B, heads, Nq] fp32 log-sum-exp (natural log of sum exp(scale*s)) */
  const void* d_o; long lddo;     /* backward only */
  void* dq; long lddq; void* dk; long lddk; void* dv; long lddv;
  float* delta;                   /* fp32 scratch (backward): max(B*heads*Nq, 2*B*Nk*heads*hd) floats;
                                     with Nk == 1: 32*B*heads*hd floats (per-block partial sums of dV) */
  int B, heads, Nq, Nk, hd;       /* hd in {8,16,32} (MFMA flash / few-token kernels) or {64,128} (latent-model widths,
                                     utils.py:145-156: MFMA kernels with LDS-staged keys) */
  float scale;                    /* 1/sqrt(hd) */
  float drop_p; unsigned long long drop_seed;
  void* workspace; long ws_bytes; /* backward, optional: dm_attn_bwd_workspace_bytes() bytes of scratch.  With it, the shapes
                                     that function returns > 0 for take the single-pass backward (probabilities computed
                                     once for dq, dk and dv); without it (NULL) every shape runs the two-kernel backward */
} dm_attn_args;
/* Flash-style fused softmax(QK^T*scale)V on bf16 MFMA, online softmax, no N x N matrix in
 * memory.  Replaces F.scaled_dot_product_attention for self- and cross-attention
 * (unet.py:119-126,136,165-172,214).
 * One key (Nk == 1, the CLIP context of unet.py:205-214): the softmax is identically 1, so o = dropout(v), dq = dk = 0 and
 * lse = scale * q.k exactly; a caller with no use for lse may pass q = k = lse = NULL (forward) and additionally
 * dq = dk = NULL (backward: only dv is produced); a caller that wants ONLY lse (the output formed elsewhere:
 * dm_cross1_fwd) may pass o = NULL (forward).  Any other NULL is DM_ERR_ARG. */
int dm_attn_fwd(const dm_attn_args* args, dm_stream_t stream);
int dm_attn_bwd(const dm_attn_args* args, dm_stream_t stream);
/* Scratch the single-pass backward wants for this shape -- one fp32 partial copy of dq per key split (currently 4; the
 * count is an implementation detail: ALWAYS size the buffer with this function) -- or 0 when the shape has none. */
long dm_attn_bwd_workspace_bytes(int B, int heads, int Nq, int Nk, int hd);

/* One-token cross-attention, output side, as one streaming pass.  Replaces, for a single context token (the CLIP-table
 * configuration: context [B, 512] -> one key, unet.py:197-198), F.scaled_dot_product_attention + to_out + the residual
 * add of CrossAttention.forward / UnetLayer.forward (unet.py:205-220, 274) and their autograd backward.
 * With one key the softmax is exactly 1, so  y[b, n, :] = res[b, n, :] + sum_h Z[b, n, h] * U[b, h, :]  with
 * U[b, h, :] = w_out[:, head h] . v[b, head h] / keep  and Z the dropout mask of the one probability per (pixel, head)
 * -- the same words dm_attn_fwd draws with Nk == 1, so the two formulations are interchangeable under one seed.
 * dm_cross1_fwd: writes U into `u` (fp32 [B, heads, C]) and streams res -> y (res NULL: no residual).
 * dm_cross1_bwd: streams dy once into dU (written to `u`), then dw_out += sum_b dU (x) v (dw_out NULL: skipped) and
 *   dv[b, ci] = sum_co dU[b, h(ci), co] * w_out[co, ci].  Fixed summation order (bit-reproducible).  The gradient with
 *   respect to res is dy itself; the query / key inputs of the attention get exactly zero (dm_attn_bwd, Nk == 1).
 * heads must be 8 and C one of 64, 128, 256, 512, 1024 (DM_ERR_ARG otherwise: use dm_attn_* + dm_gemm). */
typedef struct dm_cross1_args {
  const void* v; long ldv;          /* bf16 [B, C]: to_v(context) of the one token */
  const float* w_out;               /* fp32 [C, C] (out, in): the to_out weight (master copy) */
  const void* res; long ldr;        /* bf16 [B*N, C], row stride ldr (forward; NULL = none) */
  void* y; long ldy;                /* bf16 [B*N, C] (forward) */
  const void* dy; long lddy;        /* bf16 [B*N, C] (backward) */
  float* u;                         /* fp32 [B, heads, C]: U (forward) / dU (backward), caller's buffer */
  float* dw_out;                    /* fp32 [C, C], accumulated += (backward; NULL = skip) */
  void* dv; long lddv;              /* bf16 [B, C] (backward) */
  void* workspace; long ws_bytes;   /* backward: dm_cross1_bwd_workspace_bytes() bytes */
  int B, N, heads, C;
  float drop_p; unsigned long long drop_seed;
} dm_cross1_args;
int dm_cross1_fwd(const dm_cross1_args* args, dm_stream_t stream);
int dm_cross1_bwd(const dm_cross1_args* args, dm_stream_t stream);
long dm_cross1_bwd_workspace_bytes(int B, int heads, int C);

/* ------------------------------------------------------------------ layout / elementwise */
/* NCHW fp32 -> NHWC bf16, channels zero-padded to Cpad (UNET.forward entry, unet.py:344). */
int dm_nchw_f32_to_nhwc_bf16(const float* x, void* y, int B, int C, int HW, long ldy, int Cpad,
                             dm_stream_t stream);
/* NHWC (bf16 or fp32) -> NCHW fp32 (UNET.forward exit, unet.py:356). */
int dm_nhwc_to_nchw_f32(const void* x, int x_is_f32, long ldx, float* y, int B, int C, int HW,
                        dm_stream_t stream);
/* dtype casts of small contiguous buffers (time-embedding rows, per-(b,c) sums). */
int dm_cast_f32_to_bf16(const float* x, void* y, long n, dm_stream_t stream);
int dm_cast_bf16_to_f32(const void* x, float* y, long n, dm_stream_t stream);
/* out = a (+ b): residual add / strided copy into a concat slice (unet.py:274,355). */
int dm_add_bf16(const void* a, long lda, const void* b, long ldb, void* out, long ldo, long P, int C,
                dm_stream_t stream);
/* dx = dy * silu'(x): backward of the head's nn.SiLU (unet.py:356). */
int dm_silu_bwd_bf16(const void* dy, long lddy, const void* x, long ldx, void* dx, long lddx, long P, int C,
                     dm_stream_t stream);
/* FiLM conditioning of a ResBlock (FiLMLayer.forward, unet.py:23-42, then nn.Dropout, unet.py:105):
 * out[b][p][c] = dropout(gamma[b][c] * h[b][p][c] + beta[b][c]).  gamma_beta: fp32 [B, 2C], one row [gamma | beta] per
 * sample = the output of film_mlp (unet.py:32-35).  The dropout mask is the one dm_gn_silu_bwd regenerates from
 * (drop_seed, pixel, channel). */
int dm_film_fwd(const void* h, long ldh, const float* gamma_beta, void* out, long ldo, int B, int HW, int C,
                float drop_p, unsigned long long drop_seed, dm_stream_t stream);
/* Backward of the same: dh0 = gamma * dh, and dgamma_partials[chunk][b][c] = the chunk's sum over pixels of dh * h0
 * (fp32 [dm_film_bwd_chunks(HW), B, C]; the caller sums the chunks in order -- no atomics).  dbeta is the per-(sample,
 * channel) sum of dh, which dm_gn_silu_bwd already returns (dx_colsum). */
int dm_film_bwd_chunks(int HW);
int dm_film_bwd(const void* dh, long lddh, const void* h0, long ldh, const float* gamma_beta, void* dh0, long ldo,
                float* dgamma_partials, int B, int HW, int C, dm_stream_t stream);
/* out[b][c] (=|+=) sum over the batch's rows of x[row][c]  (bias / time-projection gradients). */
int dm_colsum_bf16(const void* x, long ldx, float* out, int nbatch, int rows_per_batch, int C, int accumulate,
                   dm_stream_t stream);
/* out[c] += sum_b in[b][c]. */
int dm_batchsum_f32(const float* in, float* out, int B, int C, dm_stream_t stream);
/* The same sum added to out1 and (optional) out2, plus an optional bf16 copy of `in` ([B][C]): the bias gradients of
 * conv1 and of the time projection and the bf16 operand of the projection's weight gradient (unet.py:87-90,101-102)
 * from a ResBlock's per-(sample, channel) sums in one pass. */
int dm_batchsum2_cast_f32(const float* in, float* out1, float* out2, void* cast_bf16, int B, int C, dm_stream_t stream);

/* ------------------------------------------------------------------ diffusion math (fp32) */
/* x_t = sqrt(a) x + sqrt(1-a) e ; v = sqrt(a) e - sqrt(1-a) x   (main_stl10.py:456-465). */
int dm_qsample_vtarget(const float* x, const float* e, const float* abar_t, float* x_t, float* v, int B,
                       long chw, dm_stream_t stream);
long dm_mse_workspace_bytes(int B);
/* loss = sum_b w[b] * sum_chw (out-v)^2 ; dout = 2 w[b] grad_scale (out-v)   (main_stl10.py:477-482,502). */
int dm_mse_loss_fwd_bwd(const float* out, const float* v, const float* w, float grad_scale, float* dout,
                        float* loss, int B, long chw, void* workspace, long ws_bytes, dm_stream_t stream);
/* Allocation-free variant of dm_mse_loss_fwd_bwd.  NO counterpart in the reference: it exists for callers that
 * cannot allocate device memory or bring a workspace -- code under HIP graph capture, tight serving / monitoring
 * loops, harnesses that forbid hipMalloc.  Same kernels and results as dm_mse_loss_fwd_bwd; the per-(sample, block)
 * partial sums live in a static device array of the library (512 KiB, held on every GPU the library runs on until
 * the process ends) instead of a caller workspace.  Calls no hipMalloc / hipFree / allocating routine.
 *   - B <= dm_mse_noalloc_max_batch() (4096); larger B returns DM_ERR_TOO_LARGE and launches nothing.
 *   - every call on one GPU shares that GPU's array: calls that overlap on one GPU (different streams) race.
 *   - dm_init_device() resolves the array's address for the CURRENT device (hipGetSymbolAddress, once per device);
 *     the entry point does the same on its first call on a device -- call dm_init_device() up front when the first
 *     call would otherwise fall inside a stream capture. */
int dm_init_device(void);
int dm_mse_noalloc_max_batch(void);
int dm_mse_loss_fwd_bwd_noalloc(const float* out, const float* v, const float* w, float grad_scale, float* dout,
                                float* loss, int B, long chw, dm_stream_t stream);
/* One fused sampler step: CFG combine on v, v->eps, DDPM mean step, eta-scaled noise, optional
 * clamp.  coef[b][8] = {sqrt(1-abar_t), sqrt(abar_t), 1/sqrt(alpha_t), beta_t/sqrt(1-abar_t),
 * eta*sqrt(posterior_var) (0 = no noise), guidance scale, clamp flag, unused}.
 * Replaces main_stl10.py:130-136 + visualization.py:92-109,144-152. */
int dm_cfg_ddpm_step(const float* z, const float* v_uncond, const float* v_cond, const float* noise,
                     const float* coef, float* z_out, int B, long chw, dm_stream_t stream);
/* One fused DPM-Solver++(2M, midpoint) step for the v-prediction model with CFG: what
 * visualization.py:509-530 does per step through diffusers' DPMSolverMultistepScheduler.step
 * (prediction_type "v_prediction", algorithm_type "dpmsolver++", solver_order 2; main_stl10.py:676-684).
 * coef[b][8] = {guidance (1.0 = conditional branch only, visualization.py:524-526), alpha_s, sigma_s,
 * sigma_t/sigma_s, alpha_t*(exp(-h)-1), 1/r0, second-order flag, unused};
 * x0 = alpha_s z - sigma_s v_hat -> x0_out;  z_out = ratio z - c x0 [- c/2 (1/r0)(x0 - x0_prev)].
 * x0_prev is read only when the flag is set (pass any valid pointer otherwise). */
int dm_cfg_dpmpp2m_step(const float* z, const float* v_uncond, const float* v_cond, const float* x0_prev,
                        const float* coef, float* z_out, float* x0_out, int B, long chw, dm_stream_t stream);

/* ------------------------------------------------------------------ optimizer */
long dm_sqnorm_workspace_bytes(void);
/* norm_out[0] = pre_scale*||g||_2, norm_out[1] = min(1, max_norm/(norm+1e-6))
 * (torch.nn.utils.clip_grad_norm_, main_stl10.py:508). */
int dm_grad_norm_clipcoef(const float* g, long n, float max_norm, float pre_scale, float* norm_out,
                          void* workspace, long ws_bytes, dm_stream_t stream);
/* Fused clip-scale + AdamW (or Adam) + EMA over a flat fp32 buffer
 * (main_stl10.py:396,509-510; main.py:152-153,181-182). clip = norm_out of the call above or NULL. */
int dm_adamw_ema_step(float* p, const float* g, float* m, float* v, float* ema, const float* clip, long n,
                      float lr, float beta1, float beta2, float eps, float wd, int step, double ema_decay,
                      float pre_scale, int adamw, dm_stream_t stream);

/* Optimizer schedule resident on the DEVICE, for replaying a whole optimizer step as one HIP graph (the launch-bound
 * small-shape configuration, reference main.py:159-188): the scalars that change every step -- Adam's step count, the
 * LambdaLR learning rate (main_stl10.py:405-414; total_steps = 0: constant rate, main.py:153) and the bias corrections --
 * live in a dm_opt_state_bytes()-byte device struct.  dm_opt_state_advance (ONE thread; capturable) moves it to the next
 * step: lr = base_lr * lambda(sched_step) with `warmup_steps` linear warm-up steps then cosine over total_steps;
 * dm_adamw_ema_step_dev is dm_adamw_ema_step reading lr / bias corrections from the struct instead of its arguments. */
int dm_opt_state_bytes(void);
int dm_opt_state_init(void* state_dev, int step, int sched_step, dm_stream_t stream);
int dm_opt_state_advance(void* state_dev, double base_lr, int warmup_steps, int total_steps, double beta1, double beta2,
                         dm_stream_t stream);
int dm_adamw_ema_step_dev(float* p, const float* g, float* m, float* v, float* ema, const float* clip, long n,
                          const void* state_dev, float beta1, float beta2, float eps, float wd, double ema_decay,
                          float pre_scale, int adamw, dm_stream_t stream);
/* dst <- slot (counter_dev[0] % nslots) of a device-resident ring of equally sized slots (slot_bytes, a multiple of 16;
 * 16-byte aligned pointers): the inputs of a captured step (x, t, e, context of main_stl10.py:449-469) selected by a
 * device counter, so a graph replay needs no host-side copy.  No reference counterpart (the reference feeds its step
 * from a DataLoader on the host). */
int dm_ring_fetch(const void* ring, void* dst, long slot_bytes, int nslots, const unsigned int* counter_dev,
                  dm_stream_t stream);

/* ------------------------------------------------------------------ fp32 COMPUTE MODE
 * The reference computes in fp32 end to end (every nn.Conv2d / nn.Linear / nn.GroupNorm / SDPA of unet.py:70-356 on
 * fp32 tensors; main_stl10.py:474-482) and the parity bar for that precision is 1e-4.  These entry points are the
 * operators above on fp32 NHWC activations ([pixel][channel] float, pixel stride ld* in elements) with fp32 weights:
 * contractions on v_mfma_f32_16x16x4_f32 (fp32 inputs, exact fp32 FMA chains), attention as fp32 flash-style kernels,
 * GroupNorm statistics accumulated in double.  Same argument structs and meanings as their bf16 counterparts unless noted;
 * every activation / gradient pointer is float*.  No channel-count or head-dim multiples are required (any Cin / Cout;
 * hd in {8, 16, 32, 64, 128}; any Nq, Nk >= 1).  The mode exists for parity and reference-precision runs: it is general and
 * exact, not tuned like the bf16 kernels. */
/* fp32 weight repack [taps][round-up(K, 64)][round-up(N, 64)] (zero padded); transpose as dm_pack_weight.
 * Replaces the weight reads of nn.Conv2d / nn.Linear (unet.py:74-75,79,113-114,160-163,225,234,261,317-320,333-335). */
long dm_packed_weight_f32_bytes(int N, int K, int taps);
int dm_pack_weight_f32(const float* w, float* dst, int N, int K, int taps, int transpose, dm_stream_t stream);
/* dm_conv3x3 / dm_gemm in fp32 (unet.py:74-75,79,225-229,234,261,317-320 and their data-gradients).  Output to y (float*),
 * or to yf when y is NULL.  pool2 and norm_scale / norm_shift are not taken (DM_ERR_ARG): the data-gradient of the
 * nearest-x2 convolution is the full-size convolution followed by dm_pool2_f32. */
int dm_conv3x3_f32(const dm_conv_args* args, dm_stream_t stream);
int dm_gemm_f32(const dm_gemm_args* args, dm_stream_t stream);
/* y[b][h][w][c] = sum of the 2x2 block of x [B, Ho, Wo, C] (contiguous): the backward of F.interpolate(nearest x2), unet.py:227. */
int dm_pool2_f32(const float* x, float* y, long ldy, int B, int Ho, int Wo, int C, dm_stream_t stream);
/* dm_wgrad in fp32 (the weight gradients of scaled_loss.backward(), main_stl10.py:502-503): ordered split-K, bit-reproducible. */
long dm_wgrad_f32_workspace_bytes(int B, int H, int W, int Cin, int Cout, int taps, int mode);
int dm_wgrad_f32(const dm_wgrad_args* args, dm_stream_t stream);
/* dm_gn_silu_fwd / dm_gn_silu_bwd in fp32 (unet.py:77-78,94,101-102,106).  Forward: workspace unused (may be NULL).
 * Backward: workspace >= dm_gn_f32_workspace_bytes(B, C); C / G must divide 256 and be <= 128; dgamma / dbeta / dx_colsum
 * are summed in a fixed order (bit-reproducible). */
int dm_gn_silu_fwd_f32(const void* x, long ldx, const float* chan_add, const float* gamma, const float* beta,
                       void* y, long ldy, float* stats, int B, int HW, int C, int G, float eps, int act_silu,
                       void* workspace, long ws_bytes, dm_stream_t stream);
long dm_gn_f32_workspace_bytes(int B, int C);
int dm_gn_silu_bwd_f32(const void* x, long ldx, const float* chan_add, const void* dy, long lddy,
                       const float* gamma, const float* beta, const float* stats, const void* add_in, long ldadd,
                       void* dx, long lddx, float* dgamma, float* dbeta, int B, int HW, int C, int G, int act_silu,
                       float drop_p, unsigned long long drop_seed, float* dx_colsum, void* workspace, long ws_bytes,
                       dm_stream_t stream);
/* F.scaled_dot_product_attention in fp32 (unet.py:119-126,136,165-172,214), forward and backward; every pointer of
 * dm_attn_args required by the direction must be set (no NULL shortcuts); delta: B*heads*Nq floats; workspace unused. */
int dm_attn_fwd_f32(const dm_attn_args* args, dm_stream_t stream);
int dm_attn_bwd_f32(const dm_attn_args* args, dm_stream_t stream);
/* Layout / elementwise helpers of the mode (unet.py:274,344,355,356; FiLM unet.py:23-42,105). */
int dm_nchw_to_nhwc_f32(const float* x, float* y, int B, int C, int HW, long ldy, int Cpad, dm_stream_t stream);
int dm_add_f32(const float* a, long lda, const float* b, long ldb, float* out, long ldo, long P, int C, dm_stream_t stream);
int dm_silu_bwd_f32(const float* dy, long lddy, const float* x, long ldx, float* dx, long lddx, long P, int C,
                    dm_stream_t stream);
int dm_colsum_f32(const float* x, long ldx, float* out, int nbatch, int rows_per_batch, int C, int accumulate,
                  dm_stream_t stream);
int dm_film_fwd_f32(const float* h, long ldh, const float* gamma_beta, float* out, long ldo, int B, int HW, int C,
                    float drop_p, unsigned long long drop_seed, dm_stream_t stream);
/* dgamma: fp32 [B, C], written (one ordered sum per (sample, channel)). */
int dm_film_bwd_f32(const float* dh, long lddh, const float* h0, long ldh, const float* gamma_beta, float* dh0, long ldo,
                    float* dgamma, int B, int HW, int C, dm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DM_HIP_H */
