/* eegclip.h -- C ABI of libeegclip_hip.so: the MI355X (gfx950) kernels behind the EEG<->CLIP hot path.
 *
 * The reference (dongyangli-del/EEG_Image_decode) has NO FFI / plugin ABI: its boundary is Python
 * (SURVEY.md section 8b).  This header is the native boundary a maintainer binds instead of the stock
 * PyTorch op sequences; every entry point names the reference lines whose arithmetic it replaces
 * (paths relative to the reference root).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (fp32 unless stated); the caller owns all buffers, including
 *     workspaces; the library never allocates, frees or synchronises and keeps no global state.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it.
 *   - return value: 0 on success, a positive hipError_t from the launch, or a negative EEGCLIP_E* code
 *     for argument errors (nothing is enqueued in that case).
 *   - one calling thread per process/rank; re-entrant per stream.
 */
#ifndef EEGCLIP_H
#define EEGCLIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define EEGCLIP_ABI_VERSION 11
#define EEGCLIP_EINVAL (-1)   /* bad shape / null pointer / unsupported combination */
#define EEGCLIP_EALIGN (-2)   /* pointer or stride violates an alignment requirement */

int eegclip_abi_version(void);

/* Two-level index map: element offset of logical index i is (i / div) * so + (i % div) * si.
 * A plain strided dimension is {div = 2^62, so = 0, si = stride}.  Lets one GEMM address the
 * (B,64,250) / (B,40,63,36) / (B,36,40) views of the encoder without copies. */
typedef struct {
    long long div, so, si;
} eegclip_dim;

#define EEGCLIP_ACT_NONE 0
#define EEGCLIP_ACT_GELU 1 /* exact erf GELU (F.gelu default) */
#define EEGCLIP_ACT_SILU 2 /* x * sigmoid(x) (nn.SiLU: diffusion prior) */
#define EEGCLIP_ACT_GELU_GRAD 3 /* backward of GELU: v *= gelu'(R[m,n]) -- R holds the forward PRE-activation and is not added */

/* Arithmetic of the contraction (eegclip_gemm_desc.precision).  Inputs, accumulation and outputs are fp32 in both modes.
 *   F32     v_mfma_f32_16x16x4_f32: every product and sum rounded once in fp32 (bit-equal to an fmaf chain); 157 TF peak.
 *   BF16X3  split precision on the bf16 matrix cores (2.5 PF peak): each operand is split while it is staged into LDS as
 *           a = a_hi + a_lo (a_hi = bf16(a), a_lo = bf16(a - a_hi)) and a*b is formed as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with fp32
 *           accumulation; the dropped a_lo*b_lo term and the rounding of a_lo bound the error of a product by ~2^-16 |a||b| (fp32: 2^-24).
 *           Measured on the encoder: embeddings move by <= 3e-5 (parity budget 1e-3).  Needs the plain-stride operand class (every
 *           Linear of the path, forward and backward); other index maps run as F32. */
#define EEGCLIP_PREC_F32 0
#define EEGCLIP_PREC_BF16X3 1

/* C[m,n] (+)= epilogue( alpha * sum_k A[m,k] * B[k,n] )          fp32 in, fp32 accumulate, fp32 out; products per `precision`.
 * epilogue order: +bias_n[n] +bias_m[m] -> (store Cpre) -> act -> dropout(p, Philox(seed, site, m*N+n)) -> +R[m,n]
 * (act = EEGCLIP_ACT_GELU_GRAD: no forward activation; after the dropout stage v *= gelu'(R[m,n]) instead of v += R[m,n] -- the
 *  gradient of dropout(gelu(pre)) w.r.t. pre fused into the GEMM that produces the upstream gradient, Transformer_EncDec.py:48).
 * rowsum_a (optional, [M]): rowsum_a[m] += sum_k A[m,k] (atomic; every K slice adds its share).  With A = dY^T this is the bias
 * gradient of the Linear whose weight gradient dW = dY^T X the same launch computes (one pass over dY instead of two).
 * Replaces every nn.Linear / 1x1-conv / (63x1)-conv GEMM and its backward on the path:
 *   models/subject_layers/Embed.py:146-149 (value embedding + PE), SelfAttention_Family.py:199-201,213 (Q/K/V/out),
 *   Transformer_EncDec.py:48-51 (FFN), Retrieval/ATMS_retrieval.py:106,113 (spatial / 1x1 conv), :157-167 (head),
 *   models/loss.py:122-123 (logits), Generation/diffusion_prior.py:167-203 (prior MLP).
 * split_k > 1: K is cut into split_k slices whose partial products are atomically added into C (C must be
 * zero or hold the value to accumulate onto); bias is added by slice 0; act/dropout/R/Cpre are not allowed.
 * With `workspace` (BF16X3 launches; at least eegclip_gemm_workspace_bytes(d) bytes, contents irrelevant, not shared by launches that may run
 * concurrently) slice s writes its partial product into slab s of the workspace instead and a second kernel of the same call adds the slabs
 * IN SLICE ORDER (+ bias, + C when accumulating) and stores C once: no atomics on C (measured: the 2.1 M device-scope atomics of a
 * 256 x 250 x 16384 / 32-slice weight gradient cost ~15 of its 31 us) and a result that does not depend on the arrival order. */
typedef struct {
    int M, N, K;
    const float* A;
    eegclip_dim Am, Ak;
    const float* B;
    eegclip_dim Bk, Bn;
    float* C;
    eegclip_dim Cm, Cn;
    float* Cpre;          /* optional pre-activation copy, indexed like C (NULL = off) */
    const float* bias_n;  /* [N] or NULL */
    const float* bias_m;  /* [M] or NULL */
    const float* R;       /* residual or NULL */
    eegclip_dim Rm, Rn;
    float alpha;
    int accumulate;       /* 1: C += result (after the epilogue chain).  2: the old C is added FIRST and goes through the chain with the product:
                             C = epilogue(alpha * A B + bias + C_old) -- dropout of a sum of two gradient contributions in one pass */
    int act;              /* EEGCLIP_ACT_* */
    float drop_p;         /* 0 = no dropout */
    unsigned long long seed;
    unsigned int drop_site;
    int split_k;          /* >= 1 */
    float* rowsum_a;      /* [M] or NULL: += sum_k A[m,k] */
    int precision;        /* EEGCLIP_PREC_* (0 = exact fp32 products); bits 8..15: BF16X3 tile configuration, 0 = chosen by the library,
                             1..6 = 64x64x32 | same, two LDS images | 64x64x64 | same, two images | 128x128x32 | same, two images (tuning aid) */
    float* workspace;     /* optional split-K scratch (see above); NULL = atomics */
    long long workspace_bytes;
} eegclip_gemm_desc;

/* fp32 matrices -> bf16 planes hi = bf16(x), lo = bf16(x - hi), [rows][ld_out] with zeros beyond the source columns; transpose != 0: the planes
 * of the transposed matrix ([cols][ld_out]).  transpose = 2: the transposed planes WITHOUT the zero fill beyond the source rows (rows % 4 == 0, ld_out % 4 == 0,
 * 8-byte aligned planes: the output may be a column block of a wider matrix) by a kernel path that needs no LDS and ~20 registers, so that it can run beside
 * a kernel that owns the CUs' LDS (the projection head's weights and the loss targets at the start of a step).  Up to 24 matrices per launch: every Linear
 * weight of the encoder, both orientations (W for Y = X W^T, W^T for dX = dY W), once per step. */
typedef struct {
    const float* src;
    void* hi;
    void* lo;
    int rows, cols;       /* of the source */
    long long ld_src, ld_out;
    int transpose;
    float* copy;          /* NULL, or (transpose == 0 only) where to leave an fp32 copy of the source as well, rows ld_copy >= cols elements apart: several
                             matrices stacked into ONE contraction operand (the image and text targets of the loss gradient, models/loss.py:122-140) */
    long long ld_copy;
} eegclip_split_item;
int eegclip_split_rows(const eegclip_split_item* items, int n, void* stream);
int eegclip_gemm_f32(const eegclip_gemm_desc* d, void* stream);
/* bytes of split-K workspace this launch can use (0: it would not use one -- split_k == 1, exact-fp32 products or an operand class outside the
 * BF16X3 kernels) */
long long eegclip_gemm_workspace_bytes(const eegclip_gemm_desc* d);
/* n independent problems (outputs must not overlap) with the result of n eegclip_gemm_f32 calls.  Members that differ only in M, K, split_k,
 * A, B, C, bias_n and rowsum_a (at most 16 of them) run as ONE grid: the joint-subject model's per-subject value embeddings -- the
 * reference's Python loop of B Linear calls, models/subject_layers/Embed.py:144 -- and their weight gradients.  Anything else is launched
 * member by member. */
int eegclip_gemm_f32_grouped(const eegclip_gemm_desc* descs, int n, void* stream);

/* ---- LayerNorm (rows of <= 1024 floats).  Transformer_EncDec.py:47,51,77-78 ; ATMS_retrieval.py:166 ; diffusion_prior.py:120,140,155
 * fwd: y = (x-mean)*rstd*gamma+beta, mean/rstd[rows] saved (may be NULL).  bwd: dx (+)= ..., dgamma/dbeta += (atomic);
 * dx_drop (optional): second output dx * mask/(1-p), mask = Philox(seed, site, row*cols+c) -- the gradient entering the
 * dropout(sublayer(.)) branch of the residual that fed this LayerNorm (Transformer_EncDec.py:45,51), saving a copy + a mask pass.
 * bwd runs two independent kernels; either may be left out: dx = NULL (and gamma may then be NULL) -> parameter gradients only,
 * dgamma = dbeta = NULL -> input gradient only. */
int eegclip_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                          int rows, int cols, float eps, void* stream);
/* the post-LN sublayer tail in one pass (Transformer_EncDec.py:45-51,77-78): v = resid + dropout(x) (resid NULL: v = x), x_out = v (optional, may
 * alias x), y = LN(v; gamma, beta), and optionally a second LayerNorm of y straight after (gamma2 != NULL: y2 = LN(y; gamma2, beta2) -- the
 * layer's norm2 followed by the encoder's final norm).  Mask = Philox(seed, site, row*cols + c): identical to the GEMM dropout epilogue. */
int eegclip_residual_layernorm_fwd(const float* x, const float* resid, float* x_out, float drop_p, unsigned long long seed, unsigned int site,
                                   const float* gamma, const float* beta, float* y, float* mean, float* rstd, const float* gamma2,
                                   const float* beta2, float* y2, float* mean2, float* rstd2, int rows, int cols, float eps, void* stream);
/* the same pass, y additionally as bf16 hi | lo planes (value = hi + lo, [rows][cols], cols % 4 == 0, 8-byte aligned): the operand form of the fused InfoNCE
 * kernels / plane GEMMs without a separate eegclip_split_rows launch (the projection head's output, Retrieval/ATMS_retrieval.py:157-167 -> models/loss.py:122) */
int eegclip_residual_layernorm_fwd_planes(const float* x, const float* resid, float* x_out, float drop_p, unsigned long long seed, unsigned int site,
                                          const float* gamma, const float* beta, float* y, float* mean, float* rstd, const float* gamma2,
                                          const float* beta2, float* y2, float* mean2, float* rstd2, int rows, int cols, float eps, void* y_hi, void* y_lo,
                                          void* stream);
int eegclip_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                          float* dx, float* dgamma, float* dbeta, int rows, int cols, int accumulate_dx, float* dx_drop, float drop_p,
                          unsigned long long seed, unsigned int site, void* stream);
/* the parameter half of eegclip_layernorm_bwd (dgamma += sum_rows dy*xhat, dbeta += sum_rows dy) through a caller-owned workspace of
 * eegclip_layernorm_bwd_params_workspace_floats(rows, cols) floats (contents irrelevant): per-workgroup partial rows + a column reduction
 * instead of hundreds of contended atomics per column (the atomic form is bound by them: 19 us for 16384 x 250; this one streams). */
/* eegclip_layernorm_bwd with both halves in ONE pass over dy / x: the input-gradient kernel leaves the parameter-gradient partial rows in
 * `workspace` (eegclip_layernorm_bwd_full_workspace_floats(rows, cols) floats), a column reduction adds them into dgamma / dbeta. */
long long eegclip_layernorm_bwd_full_workspace_floats(int rows, int cols);
int eegclip_layernorm_bwd_full(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx, float* dgamma,
                               float* dbeta, int rows, int cols, int accumulate_dx, float* dx_drop, float drop_p, unsigned long long seed,
                               unsigned int site, float* workspace, void* stream);
long long eegclip_layernorm_bwd_params_workspace_floats(int rows, int cols);
int eegclip_layernorm_bwd_params(const float* dy, const float* x, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                                 int rows, int cols, float* workspace, void* stream);

/* LayerNorm -> SiLU -> dropout in one pass (prior stage, Generation/diffusion_prior.py:117-121,137-143): y_ln = LN(x) is kept for
 * the backward, y_act = dropout(silu(y_ln)).  silu_bwd: dx (+)= dy*mask/(1-p)*silu'(pre). */
int eegclip_layernorm_silu_fwd(const float* x, const float* gamma, const float* beta, float* y_ln, float* y_act, float* mean, float* rstd,
                               int rows, int cols, float eps, float drop_p, unsigned long long seed, unsigned int site, void* stream);
int eegclip_silu_bwd(const float* dy, const float* pre, float* dx, long long n, int accumulate, float drop_p, unsigned long long seed,
                     unsigned int site, void* stream);

/* ---- BatchNorm2d (+ELU, +dropout) over an (outer, C, inner) view.  ATMS_retrieval.py:104-105,107-109
 * sums: double[2C] (sum, sum of squares) accumulated atomically -- zero it first.
 * finalize: train=1 -> mean/rstd from the batch sums (biased var) and running stats updated in place with the unbiased
 * var (momentum) and *num_batches_tracked += 1 (int64 on the device, may be NULL); train=0 -> mean/rstd from the running stats.
 * bn_elu_fwd: y = dropout(ELU(gamma*(x-mean)*rstd+beta)).   bn_elu_bwd: dx, dgamma +=, dbeta += (sums = double[2C] scratch, zeroed). */
int eegclip_bn_stats(const float* x, int outer, int C, int inner, double* sums, void* stream);
int eegclip_bn_finalize(const double* sums, double count, float eps, float momentum, int C, float* mean, float* rstd,
                        float* running_mean, float* running_var, int train, long long* num_batches_tracked, void* stream);
/* finalize (train) from `nrows` per-workgroup partial rows [sum(C) | sumsq(C)] (fp64, 8-byte aligned), summed in a fixed order: one launch, no atomics,
 * nothing to clear.  sums_out (optional): the 2C column sums (what a data-parallel job all-reduces); mean = rstd = NULL: only those.  2C <= 512. */
int eegclip_bn_finalize_rows(const double* rows, int nrows, double count, float eps, float momentum, int C, float* mean, float* rstd,
                             float* running_mean, float* running_var, long long* num_batches_tracked, double* sums_out, void* stream);
int eegclip_bn_elu_fwd(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, float* y,
                       int outer, int C, int inner, float drop_p, unsigned long long seed, unsigned int site, void* stream);
int eegclip_bn_elu_bwd(const float* dz, const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                       double* sums, float* dx, float* dgamma, float* dbeta, int outer, int C, int inner, float drop_p,
                       unsigned long long seed, unsigned int site, void* stream);

/* split form of bn_elu_bwd for data-parallel SyncBN: stats -> copy sums to sums_local -> all-reduce sums[2C] -> apply with the GLOBAL
 * count.  dx uses the global sums; dgamma/dbeta += this rank's own sums (sums_local; NULL = sums) so that the later mean-all-reduce of
 * the parameter gradients reproduces the single-process value. */
int eegclip_bn_elu_bwd_stats(const float* dz, const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                             double* sums, int outer, int C, int inner, float drop_p, unsigned long long seed, unsigned int site, void* stream);
int eegclip_bn_elu_bwd_apply(const float* dz, const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                             const double* sums, const double* sums_local, double count, float* dx, float* dgamma, float* dbeta, int outer, int C, int inner, float drop_p,
                             unsigned long long seed, unsigned int site, void* stream);

/* ---- token insertion + embedding dropout on h (B,L,D) in place.  Embed.py:116-121,158-162
 * row 0 of sample b <- tokens[ids[b]] (ids == NULL: tokens[0], the shared token).  bwd: dh *= mask/(1-p); dtokens[id] += dh[b,0,:] */
int eegclip_embed_finish(float* h, const float* tokens, const long long* ids, int B, int L, int D, float drop_p,
                         unsigned long long seed, unsigned int site, void* stream);
int eegclip_embed_finish_bwd(float* dh, float* dtokens, const long long* ids, int B, int L, int D, float drop_p,
                             unsigned long long seed, unsigned int site, void* stream);

/* ---- elementwise helpers (backward of the fused GEMM epilogues) */
int eegclip_dropout_scale(float* x, long long n, float drop_p, unsigned long long seed, unsigned int site, void* stream);
int eegclip_gelu_bwd(const float* dy, const float* pre, float* dx, long long n, int accumulate, float drop_p,
                     unsigned long long seed, unsigned int site, void* stream);        /* dx (+)= dy*mask/(1-p)*gelu'(pre) */
/* epilogue for a split-K GEMM product `acc` (M,N): out = dropout(act(acc + bias_n))[+ resid], pre = acc + bias_n (optional) -- the
 * same stage order and the same Philox element index (m*N+n) as the fused GEMM epilogue, so the two forms are interchangeable */
int eegclip_bias_act(const float* acc, const float* bias_n, float* pre, const float* resid, float* out, int M, int N, int act, float drop_p,
                     unsigned long long seed, unsigned int site, void* stream);
int eegclip_axpby(const float* x, float* y, long long n, float a, float b, void* stream); /* y = a*x + b*y */
int eegclip_reduce_mid(const float* x, int outer, int mid, int inner, float* out, void* stream); /* out[m] += sum_{o,i} x[o][m][i] */
/* out[c] += sum over blocks and rows r in [row0, blk_rows) of x[blk*blk_stride + r*cols + c] */
int eegclip_colsum_blocks(const float* x, int nblk, int blk_rows, int row0, int cols, long long blk_stride, float* out, void* stream);
int eegclip_sumsq(const float* x, long long n, double* out, void* stream);              /* *out += sum x^2 */
/* dataset staging: a chunk of the reference's on-disk trials -- float64 (n_items, reps, channels, T), Retrieval/eegdatasets_leaveone.py:151-157,
 * 199-203 -- to the float32 layout the loops consume, time window applied (tidx: the Tw selected sample indices, :293-306;
 * NULL = all T of them, Tw == T).
 * mean_reps = 0: every repetition is a sample, dst (n_items*reps, channels, Tw) (training split, :255); mean_reps = 1: the mean over
 * repetitions taken after the float32 cast, dst (n_items, channels, Tw) (test split, :220). */
int eegclip_stage_eeg(const double* src, float* dst, long long n_items, int reps, int channels, int T, const int* tidx, int Tw, int mean_reps,
                      void* stream);
/* sample-block gather (scatter = 0: dst[j] = src[idx[j]]) / scatter (dst[idx[j]] = src[j]); a block = row_floats contiguous floats at
 * j*stride.  Used by the joint-subject model (Retrieval/ATMS_retrieval_joint_train.py:172-192, models/subject_layers/Embed.py:142-144) to
 * bring a mixed-subject batch into subject order, so that each subject's value-embedding Linear is one GEMM over a contiguous block, and by the
 * HBM-resident dataset loader to assemble a shuffled batch (EEG windows, labels as 8-byte rows, feature rows). */
int eegclip_gather_rows(float* dst, long long dst_stride, const float* src, long long src_stride, const int* idx, int n, int row_floats,
                        int scatter, void* stream);

/* ---- fused AdamW / Adam step on a flat fp32 segment (torch.optim.AdamW math; ATMS_retrieval.py:548, diffusion_prior.py:286)
 * `step` is the 1-based step count of this update.  g is multiplied by grad_scale and, if grad_scale_dev != NULL, by
 * *grad_scale_dev (device scalar: gradient clipping without a host sync). */
int eegclip_adamw_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                       float weight_decay, long long step, float grad_scale, const float* grad_scale_dev, void* stream);
/* the same update, and g := 0 behind the read: optimizer.step() followed by optimizer.zero_grad() (ATMS_retrieval.py:209,231) in one pass */
int eegclip_adamw_step_zero_grad(float* p, float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                                 float weight_decay, long long step, float grad_scale, const float* grad_scale_dev, void* stream);
int eegclip_clip_scale(const double* sumsq, float max_norm, float* scale_out, void* stream); /* min(1, max_norm/(sqrt(sumsq)+1e-6)) */

/* ---- diffusion prior / DDPM pieces (diffusers==0.30.0 semantics restated; Generation/diffusion_prior.py:29,314,370-376)
 * timestep_embedding: out[n] = [cos(t_n f_i) | sin(t_n f_i)], f_i = exp(-ln(1e4) i/(dim/2))      (Timesteps(dim, True, 0))
 * ddpm_add_noise:     out = sqrt_acp[t_n]*h + sqrt_1macp[t_n]*noise        (tables: float[1000] on the device)
 * ddpm_step:          eps = eps_u + g*(eps_c-eps_u) (eps_u NULL: eps = eps_c); x0 = clamp((x - sb*eps)/sa, -1, 1);
 *                     out = c0*x0 + ct*x + sigma*noise (noise NULL or sigma 0: none).  out may alias x; out_dup (optional) gets a second copy
 *                     (the conditional / unconditional halves of the next step's classifier-free-guidance input).
 * prior_stage_infer:  the tail of a prior stage in the SAMPLING chain (diffusion_prior.py:186-199 inside :358-377), where the time and
 *                     condition embeddings are chain invariants computed up front:  y = SiLU(LayerNorm(x; gamma, beta, eps)) (+ skip);
 *                     act_out = y (optional); xin_out = y + te[col] + (row < ce_rows ? ce[row][col] : 0) (optional, the next stage's input).
 * mse_loss_grad:      *loss += mean((pred-target)^2); dpred = 2*(pred-target)/n */
int eegclip_timestep_embedding(const float* t, int n, int dim, float* out, void* stream);
int eegclip_ddpm_add_noise(const float* h, const float* noise, const long long* t, const float* sqrt_acp, const float* sqrt_1macp,
                           float* out, int n, int d, void* stream);
int eegclip_ddpm_step(const float* x, const float* eps_c, const float* eps_u, float guidance, float sa, float sb, float c0, float ct,
                      float sigma, const float* noise, float* out, float* out_dup, long long n, void* stream);
int eegclip_prior_stage_infer(const float* x, const float* gamma, const float* beta, const float* skip, float* act_out, const float* te,
                              const float* ce, int ce_rows, float* xin_out, int rows, int cols, float eps, void* stream);
int eegclip_mse_loss_grad(const float* pred, const float* target, long long n, float* loss, float* dpred, void* stream);
/* the same with both outputs scaled by `weight` (a term of a mixed objective: Generation/ATMS_reconstruction.py:227) */
int eegclip_mse_loss_grad_scaled(const float* pred, const float* target, long long n, float weight, float* loss, float* dpred, void* stream);

/* ---- 64-token multi-head self-attention.  SelfAttention_Family.py:56-75
 * qkv: (B*L, ld) rows with q | k | v column blocks of H*E each; ctx/dctx: (B*L, H*E); dqkv like qkv.  L must be 64, E <= 64.
 * dropout acts on the softmax probabilities; the mask is Philox(seed, site, ((b*H+h)*L+i)*L+j) and is regenerated in bwd. */
int eegclip_attention_fwd(const float* qkv, float* ctx, int B, int L, int H, int E, int ld, float scale, float drop_p,
                          unsigned long long seed, unsigned int site, void* stream);
int eegclip_attention_bwd(const float* qkv, const float* dctx, float* dqkv, int B, int L, int H, int E, int ld, float scale,
                          float drop_p, unsigned long long seed, unsigned int site, void* stream);
/* the same contract with split-bf16 products (hi*lo + lo*hi + hi*hi on v_mfma_f32_16x16x32_bf16, fp32 accumulate: ~2^-16 relative per term, as
 * EEGCLIP_PREC_BF16X3): E and ld even, qkv / dctx / dqkv 8-byte aligned.  csrc/attention_x3.hip
 * dqkv_planes = 1: dq | dk | dv leave as token planes instead (see eegclip_wgrad_tok: 3 groups of B blocks of 64 KB at `dqkv`, channel 64 head + e,
 * H <= 4, 16-byte aligned) -- what eegclip_token_block_bwd part 1 and the q | k | v weight gradient read. */
int eegclip_attention_bwd_x3(const float* qkv, const float* dctx, void* dqkv, int dqkv_planes, int B, int L, int H, int E, int ld, float scale,
                          float drop_p, unsigned long long seed, unsigned int site, void* stream);

/* ---- tsconv front: Conv2d(1,40,(1,25)) -> AvgPool2d((1,51),(1,5)).  ATMS_retrieval.py:102-103
 * Computed as a 51-sample box filter of every token row (one wave-level prefix sum, shared by the 40 filters) followed by the 25-tap
 * convolution at stride 5 -- the two commute, and K = 25 on the matrix cores instead of 75 for the folded filter; the (B,40,H,226)
 * intermediate of the reference never exists.
 * x: token rows of T=250 floats at x + b*xs_b + h*xs_h (h < H); w25: the (40,25) conv taps; y/dy: (B,40,H,36).
 * fwd optionally accumulates the BatchNorm batch sums of y into sums (double[80], zeroed by the caller).
 * bwd_w: dw25 += sum over workgroup partials (two-stage reduction through the caller's `workspace`).
 * bwd_x overwrites dx rows h < H. */
/* workspace (optional, 8-byte aligned, eegclip_tsconv_fwd_workspace_floats(B, H) floats): the BatchNorm batch sums leave every workgroup as a
 * partial row and are column-summed by a second kernel instead of contended fp64 atomics */
long long eegclip_tsconv_fwd_workspace_floats(int B, int H);
int eegclip_tsconv_fwd(const float* x, long long xs_b, long long xs_h, const float* w25, const float* bias, float* y, int B, int H,
                       int T, int C, double* sums, float* workspace, void* stream);
long long eegclip_tsconv_bwd_w_workspace_floats(int B, int H);   /* size of `workspace` below (per-workgroup partial tap gradients) */
int eegclip_tsconv_bwd_w(const float* x, long long xs_b, long long xs_h, const float* dy, float* dw25, float* workspace, int B, int H,
                         int T, int C, void* stream);
int eegclip_tsconv_bwd_x(const float* dy, const float* w25, float* dx, long long xs_b, long long xs_h, int B, int H, int T, int C,
                         void* stream);

/* ---- fused spatial stage of tsconv: BatchNorm2d(40) -> ELU -> Conv2d(40,40,(H,1)) and its backward (ATMS_retrieval.py:104-106).
 * y1/dy1: (B,40,H,36) conv+pool output and its gradient; mean/rstd/gamma/beta: BatchNorm1 statistics and affine (float[40]);
 * Ws: (40,40,H) spatial weights; y2/dy2: (B,40,36).  z1 = ELU(BN(y1)) is never materialised: every kernel re-evaluates it while
 * staging y1, and bwd_x recomputes dz1 = Ws^T dy2 on the matrix cores.  H <= 64.
 *   fwd         y2 = bs + Ws * z1 ; optional BatchNorm2 batch sums of y2 into sums2 (double[80], zeroed by the caller)
 *   bwd_w       dWs += sum_{b,w} dy2 (x) z1       (two-stage reduction through `workspace`, size from ..._workspace_floats)
 *   bwd_x_stats sums[c] += sum da, sums[40+c] += sum da*xhat,  da = (Ws^T dy2) * ELU'(BN(y1))     (double[80], zeroed by the caller)
 *   bwd_x_apply dy1 = gamma*rstd*(da - sums/count - xhat*sums'/count); dgamma/dbeta += sums_local (NULL = sums) -- SyncBN: all-reduce
 *               sums between the two calls and pass the global count.  sconv_fwd: with `workspace` (eegclip_sconv_fwd_workspace_floats(B) floats)
 * the K-slice partial tiles go to slabs as plain stores and the BatchNorm2-statistics kernel of the same call sums them into y2; workspace = NULL:
 * they are added into y2 with atomics (y2_is_zero = 0 lets the call clear y2 itself, != 0 says the caller already did). */
int eegclip_sconv_fwd(const float* y1, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* Ws,
                      const float* bs, float* y2, double* sums2, int B, int H, int y2_is_zero, float* workspace, void* stream);
long long eegclip_sconv_fwd_workspace_floats(int B);
long long eegclip_sconv_bwd_w_workspace_floats(int B, int H);
/* precision: EEGCLIP_PREC_F32 (exact fp32 products) | EEGCLIP_PREC_BF16X3 (split-bf16 products; dy2 16-byte aligned) */
int eegclip_sconv_bwd_w(const float* y1, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* dy2,
                        float* dWs, float* workspace, int B, int H, int precision, void* stream);
/* WsT_hi / WsT_lo (both or neither): bf16 planes of Ws^T, [(c,h)][64 o] -- eegclip_split_rows{src = Ws, rows = 40, cols = 40 H, ld_src = 40 H,
 * ld_out = 64, transpose = 1}, 16-byte aligned.  Given: the K = 40 contraction runs as split-bf16 products (hi*lo + lo*hi + hi*hi on the bf16
 * matrix cores, fp32 accumulate: ~2^-16 relative per term, as EEGCLIP_PREC_BF16X3); NULL: exact fp32 products. */
/* workspace (optional, 8-byte aligned, eegclip_sconv_bwd_x_stats_workspace_floats(B) floats): the 80 sums leave every workgroup as a partial row and
 * are column-summed by a second kernel instead of 1280-way contended fp64 atomics. */
long long eegclip_sconv_bwd_x_stats_workspace_floats(int B);
int eegclip_sconv_bwd_x_stats(const float* dy2, const float* Ws, const void* WsT_hi, const void* WsT_lo, const float* y1, const float* mean,
                              const float* rstd, const float* gamma, const float* beta, double* sums, float* workspace, int B, int H, void* stream);
int eegclip_sconv_bwd_x_apply(const float* dy2, const float* Ws, const void* WsT_hi, const void* WsT_lo, const float* y1, const float* mean,
                              const float* rstd, const float* gamma, const float* beta, const double* sums, const double* sums_local,
                              double count, float* dy1, float* dgamma, float* dbeta, int B, int H, void* stream);

/* ---- InfoNCE around the logits GEMM.  models/loss.py:122-140  (scale = pointer to the RAW logit_scale on the device)
 * lse_rows/cols: log-sum-exp of scale*X along rows / columns.  infonce_grad: X (rows x cols block of raw logits, positives at
 * column i+col0) <- scale * G in place, *loss += weighted loss contribution, *dscale += sum G.*raw.  lse_r or lse_c may be NULL
 * (row-only / column-only term of the row-sharded loss).  infonce_loss: loss only (eval). */
int eegclip_lse_rows(const float* X, int rows, int cols, long long ld, const float* scale, float* lse, void* stream);
int eegclip_lse_cols(const float* X, int rows, int cols, long long ld, const float* scale, float* lse, void* stream);
int eegclip_infonce_grad(float* X, int rows, int cols, long long ld, int col0, int n_total, const float* scale, const float* lse_r,
                         const float* lse_c, float weight, float* loss, float* dscale, void* stream);
int eegclip_infonce_loss(const float* X, int n, long long ld, const float* scale, const float* lse_r, const float* lse_c, float weight,
                         float* loss, void* stream);

#define EEGCLIP_DT_BF16 0
#define EEGCLIP_DT_F16 1

/* ---- 16-bit Linear layers of the SDXL sampling path (row F2: the UNet's to_q / to_k / to_v / to_out and to_k_ip / to_v_ip projections around
 * the cross-attention of Generation/custom_pipeline.py:365-373, which diffusers runs as library GEMMs):
 *     C[m, n] = sum_k A[m, k] W[n, k] + bias[n] + R[r_div ? m / r_div : m, n]        A (M, K), W (N, K) = nn.Linear weight layout, C (M, N)
 * fp16 / bf16 (dtype = EEGCLIP_DT_*) in and out, fp32 accumulate on the 16-bit matrix cores; bias, R optional (NULL).  r_div > 0: R has one row per
 * block of r_div consecutive rows (a per-sample embedding added to every token).  N % 128 == 0, K % 64 == 0, lda / ldw multiples of 8, ldc / ldr
 * multiples of 4, A / W 16-byte and C / bias / R 8-byte aligned; any M.
 * sampler_step: one step of the denoising loop over the latents (custom_pipeline.py:376-385), fused: eps = eps_u + guidance * (eps_c - eps_u)
 * (eps_c NULL: eps = eps_u), x' = cx * x + ce * eps + cn * noise (noise NULL: no noise term), out = x' and optionally scaled = x' * in_scale (the
 * next model input: scheduler.scale_model_input).  cx / ce / cn come from the scheduler (DDIM eta = 0, Euler ancestral).  n % 4 == 0. */
int eegclip_gemm16(const void* A, long long lda, const void* W, long long ldw, void* C, long long ldc, const void* bias, const void* R, long long ldr,
                   int r_div, int M, int N, int K, int dtype, void* stream);
int eegclip_sampler_step(const void* x, const void* eps_u, const void* eps_c, const void* noise, void* out, void* scaled, float guidance, float cx,
                         float ce, float cn, float in_scale, long long n, int dtype, void* stream);

/* ---- fused InfoNCE (models/loss.py:100-141) on the bf16 matrix cores: the N x N logits are never written on the forward.
 * A "block" is S = s * Q K^T (Q: the n rows that are scored, K: the N rows they are scored against, both (., D) row-major) with the positive of
 * row i at column col0 + i (loss.py:129-130).  The symmetric loss of the reference is two blocks, (A, B) and (B, A) -- it computes both logit
 * matrices, loss.py:122-123; the row-sharded data-parallel form (loss.py:113-115) is the same two blocks against the gathered features.
 *   split_bf16   x -> bf16 planes hi = bf16(x), lo = bf16(x - hi) (lo may be NULL); n % 8 == 0, 16-byte aligned pointers
 *   fused_fwd    per block: row log-sum-exp -> lse[n]; *loss += sum_blocks weight / n_total * sum_rows (lse - S[i, col0 + i])   (loss = NULL: the per-tile
 *                partials only, see eegclip_infonce_fused_grad_finalize)
 *                (`part` = workspace of eegclip_infonce_fused_workspace_floats(n, N) floats, `diag` = n floats; both written here)
 *   fused_grad   per block: G[i, j] = s * weight / n_total * (exp(S_ij - lse[i]) + (lse_k ? exp(S_ij - lse_k[j]) : 0) - (lse_k ? 2 : 1) [j == col0 + i])
 *                in fp32 (n x N, leading dimension ldg, ldg % 4 == 0) and *dscale += d loss / d s: then dQ = G K and dK = G^T Q are plain GEMMs.
 *                lse_k = the lse of the SWAPPED block (the column normaliser of the square single-process case: one G for both CE terms).
 *                lse == NULL with lse_k given: the key-normalised term alone, G[i, j] = s * weight / n_total * (exp(S_ij - lse_k[j]) - [j == col0 + i]) -- the
 *                swapped block's gradient matrix produced TRANSPOSED (rows = gathered queries, columns = the rank's targets; col0 may be negative).
 * planes = 1: one bf16 product (features rounded to bf16: logit error ~2^-9 |q||k|, the throughput mode); planes = 2: q k = q_hi k_hi + q_hi k_lo +
 * q_lo k_hi, fp32 accumulate (logits within ~5e-5 of exact fp32 products: the parity mode).  All blocks of one call share n, N, D; at most 8.
 * Supported shapes: n, N, D multiples of 64 (eegclip_infonce_fused_supported); anything else takes the GEMM + lse_rows/lse_cols route.
 * Tile form (the library's choice when the upper bits of `planes` are 0): 64 x 64 logits per workgroup for small blocks, 128 x 128 once 256 such tiles exist,
 * 256 x 256 (one product only; forward, and the gradient pass only when it finalises the forward's partials) once 256 of THOSE exist -- N >= 4096 square.
 * Benches / tests may force it: bits 8..15 = 64 | 128 | 255 (= 256 x 256), bits 16..17 = 1 (4 waves) | 2 (8 waves) | 3 (4 MFMA + 4 producer waves), bit 18 = the
 * 8-wave forms without the cross-barrier fragment prefetch.  The forward and a FINALISING gradient pass of one block must use the same tile form (partial layout). */
typedef struct {
    const void* q_hi;     /* (n, D) bf16 */
    const void* q_lo;     /* (n, D) bf16 or NULL (planes = 1) */
    const void* k_hi;     /* (N, D) bf16 */
    const void* k_lo;
    int col0;             /* positive of row i = column col0 + i */
    float weight;         /* loss weight of this block (the 1/2 of the symmetric loss and the target mix included) */
    float* part;          /* workspace (fwd) */
    float* diag;          /* [n] positives (fwd out) */
    float* lse;           /* [n] (fwd out, grad in) */
    const float* lse_k;   /* [N] or NULL (grad in) */
    float* G;             /* (n, ldg) fp32 (grad out) */
    long long ldg;
    const float* part_k;  /* eegclip_infonce_fused_grad_finalize only: `part` / `diag` of the SWAPPED block (its rows are this block's keys) */
    const float* diag_k;
    void *G_hi, *G_lo;    /* grad out, optional: G again as bf16 hi | lo planes (leading dimension ldg; 8-byte aligned) -- the A operand of the query-gradient
                             plane GEMM dQ = G K (eegclip_head_gemm); G itself may then be NULL */
} eegclip_infonce_problem;
int eegclip_split_bf16(const float* x, void* hi, void* lo, long long n, void* stream);
int eegclip_infonce_fused_supported(int n, int N, int D);
long long eegclip_infonce_fused_workspace_floats(int n, int N);
int eegclip_infonce_fused_fwd(const eegclip_infonce_problem* blocks, int n_blocks, int n, int N, int D, int planes, int n_total,
                              const float* scale, float* loss, void* stream);
int eegclip_infonce_fused_grad(const eegclip_infonce_problem* blocks, int n_blocks, int n, int N, int D, int planes, int n_total,
                               const float* scale, float* dscale, void* stream);
/* training: the finalize folded into the gradient pass (no log-sum-exp launch between the two tile launches).  eegclip_infonce_fused_fwd with loss = NULL
 * leaves only the per-tile partials; this call forms, per workgroup, the log-sum-exp of its rows (from `part`) and of its keys (from `part_k`), writes G as
 * eegclip_infonce_fused_grad with both normalisers does, and adds the loss terms of BOTH blocks of every entry (the block and its swapped block, same
 * weight) to *loss.  Square blocks only (n == N: the single-process symmetric loss, models/loss.py:122-140). */
int eegclip_infonce_fused_grad_finalize(const eegclip_infonce_problem* blocks, int n_blocks, int n, int N, int D, int planes, int n_total,
                                        const float* scale, float* loss, float* dscale, void* stream);

/* ---- tail of tsconv + Enc_eeg projection, one workgroup per sample (ATMS_retrieval.py:107-109,113-114,145):
 *   fwd: z2 = dropout(ELU(BatchNorm2(y2)))  (B,40,36);  feat[b, w*40+e] = bias[e] + sum_c W[e,c] z2[b,c,w]   (B,1440)
 *   bwd: dW += dfeat^T z2 ; dbias += sum dfeat ; dz2 = W^T dfeat (written) ; sums[2*40] += BatchNorm-backward statistics of
 *        da = dz2 * mask/(1-p) * ELU'(BN(y2)) -- followed (after the SyncBN all-reduce of sums, if any) by eegclip_bn_elu_bwd_apply(dz2, y2, ...). */
int eegclip_proj1x1_fwd(const float* y2, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* W,
                        const float* bias, float* z2, float* feat, int B, float drop_p, unsigned long long seed, unsigned int site, void* stream);
/* the same with the BatchNorm2 finalize folded in (training): the batch statistics come as `nrows` partial rows [sum(40) | sumsq(40)] (fp64, what
 * eegclip_cstack_fwd leaves per sample) summed in a fixed order by every workgroup; workgroup 0 stores mean / rstd and updates the running statistics and
 * the step counter (eegclip_bn_finalize's work without its launch) */
int eegclip_proj1x1_fwd_rows(const float* y2, const double* rows, int nrows, double count, float eps, float momentum, float* mean, float* rstd,
                             float* running_mean, float* running_var, long long* num_batches_tracked, const float* gamma, const float* beta, const float* W,
                             const float* bias, float* z2, float* feat, int B, float drop_p, unsigned long long seed, unsigned int site, void* stream);
/* workspace (optional, 8-byte aligned, eegclip_proj1x1_bwd_workspace_floats(B) floats, contents irrelevant): dW / dbias / sums are accumulated through
 * per-workgroup partial rows + a column reduction instead of 1720 contended atomics per workgroup */
long long eegclip_proj1x1_bwd_workspace_floats(int B);
int eegclip_proj1x1_bwd(const float* dfeat, const float* z2, const float* W, const float* y2, const float* mean, const float* rstd,
                        const float* gamma, const float* beta, float* dz2, float* dW, float* dbias, double* sums, float* workspace, int B,
                        float drop_p, unsigned long long seed, unsigned int site, void* stream);

/* ---- large-batch InfoNCE logits on the bf16 matrix cores (models/loss.py:122-123 at global batch 2048, D = 1024):
 *   c[m][n] = (*scale) * sum_k a[m][k] * b[n][k]     a (M,K), b (N,K) bf16 row-major (eegclip_cast_bf16 of the fp32 features), c fp32.
 * fp32 accumulation and fp32 logits; M, N multiples of 128, K multiple of 64, ldc multiple of 4, 16-byte aligned pointers
 * (other sizes: eegclip_gemm_f32).  cast_bf16: n multiple of 8. */
int eegclip_cast_bf16(const float* x, void* y_bf16, long long n, void* stream);
int eegclip_logits_bf16(const void* a_bf16, const void* b_bf16, float* c, int M, int N, int K, long long ldc, const float* scale, void* stream);

/* ---- SDXL UNet cross-attention with the IP-Adapter image branch fused (call site Generation/custom_pipeline.py:365-373; arithmetic =
 * diffusers 0.30.0 AttnProcessor2_0 / IPAdapterAttnProcessor2_0, restated -- parity unpinned):
 *   out = softmax(q k^T / sqrt(64)) v + ip_scale * softmax(q k_ip^T / 8) v_ip
 * q/out (B, HW, heads*64); k/v (B, S, heads*64); k_ip/v_ip (B, S_ip, heads*64) or NULL with S_ip = 0.  16-bit I/O (dtype), fp32 accumulate.
 * head_dim must be 64, S and S_ip <= 128, pointers 16-byte aligned. */
int eegclip_cross_attn_fwd(const void* q, const void* k, const void* v, const void* k_ip, const void* v_ip, void* out, int B, int HW, int heads,
                           int head_dim, int S, int S_ip, float ip_scale, int dtype, void* stream);

/* ---- retrieval readouts.  ATMS_retrieval.py:246 (argmax), :320 (top-5).  ties -> lowest index; out_idx: int64 (rows, k), k <= 8 */
int eegclip_topk_rows(const float* X, int rows, int cols, long long ld, int k, const float* scale /* device scalar or NULL: rank by scale*x */,
                      long long* out_idx, void* stream);
int eegclip_count_equal(const long long* pred, int stride, const long long* labels, int n, int* count, void* stream);
/* *count += #{rows whose arg-max column of scale * X (ties -> lowest index) equals labels[row]}: top-1 + comparison of the running train accuracy
 * (ATMS_retrieval.py:241-250) in one launch */
int eegclip_top1_count(const float* X, int rows, int cols, long long ld, const float* scale, const long long* labels, int* count, void* stream);

/* ---- launch-plan executor: the forward / backward of the encoder are fixed sequences of the entry points above (~50 per direction).  Issuing
 * them one foreign call at a time from Python costs ~5 us each -- more than the GPU needs for most of them -- so a sequence is described ONCE as an
 * array of (function id, argument slots) and replayed by one call that loops in C.  (The reference's step is paced by the Python interpreter in the
 * same way: Retrieval/ATMS_retrieval.py:209-250 issues every torch op of the step from its batch loop.)
 *   fn      id from eegclip_plan_fn_id("eegclip_...") (every entry point whose last parameter is the stream), or a pseudo op:
 *           EEGCLIP_PLAN_MEMSET (a[0].p = device pointer, a[1].i = bytes: hipMemsetAsync to 0), EEGCLIP_PLAN_JOIN (main stream waits for the side stream)
 *   flags   EEGCLIP_PLAN_SIDE: launch on `side_stream`, ordered behind everything enqueued on the main stream so far (fork event `events[index]`);
 *           EEGCLIP_PLAN_SKIP: leave the op out of this run
 *   a[k]    the k-th argument in its natural C type (pointers .p, int .i32, unsigned .u32, long long .i, unsigned long long .u, float .f, double .d);
 *           the trailing stream argument is supplied by the executor.
 * plan_run executes ops [begin, end); *dirty carries "the side stream holds work the main stream has not waited for" between calls (a plan with
 * host callbacks is run in segments); a trailing join is issued when end == n_total.  Returns 0 or the first failing op's code (*failed = its index).
 * plan_events fills `out[n]` with hipEvent_t handles (timing disabled) for the fork / join events of one plan. */
typedef union {
    void* p;
    long long i;
    unsigned long long u;
    double d;
    float f;
    int i32;
    unsigned int u32;
} eegclip_plan_arg;
#define EEGCLIP_PLAN_MAX_ARGS 28
typedef struct {
    int fn;
    int flags;
    eegclip_plan_arg a[EEGCLIP_PLAN_MAX_ARGS];
} eegclip_plan_op;
#define EEGCLIP_PLAN_MEMSET (-2)
#define EEGCLIP_PLAN_JOIN (-3)
#define EEGCLIP_PLAN_SIDE 1
#define EEGCLIP_PLAN_SKIP 2
#define EEGCLIP_PLAN_SIDE2 4 /* the op runs on the SECOND side stream (independent weight-gradient GEMMs side by side) */
int eegclip_plan_fn_id(const char* name);
int eegclip_plan_events(int n, void** out);
/* destroys events made by eegclip_plan_events (null entries are skipped): call when a plan is dropped -- plans are rebuilt per batch size /
 * mode / world size, so a long-lived process would otherwise leak their fork / join events */
int eegclip_plan_events_destroy(int n, void* const* events);
int eegclip_plan_run(const eegclip_plan_op* ops, int begin, int end, int n_total, void* main_stream, void* side_stream, void* side_stream2,
                     void* const* events, void* join_event, void* join_event2, int* dirty, int* failed);

/* ---- the transformer block of the encoder as ONE launch, one workgroup per sample (csrc/token_block.hip): value embedding + positional
 * embedding + subject token + dropout (models/subject_layers/Embed.py:141-162), fused q | k | v projection, 4-head attention with probability
 * dropout, output projection (SelfAttention_Family.py:56-75,194-213), dropout + residual + LayerNorm, FFN 250 -> 256 (GELU, dropout) -> 250,
 * dropout + residual + LayerNorm, final LayerNorm (Transformer_EncDec.py:39-51,61-80).  Replaces ten launches of the forward plan; writes every
 * tensor the backward reads -- fp32 in the layouts of the unfused kernels, or token planes where only a weight-gradient GEMM reads it -- with the same
 * Philox masks (seed, site, flat element index).  Arithmetic of
 * the Linears: split-bf16 products, fp32 accumulate (EEGCLIP_PREC_BF16X3).  Specialised for 63 channels x 250 samples, d_model 250, 4 heads x 62,
 * d_ff 256 (the reference's only configuration, Retrieval/ATMS_retrieval.py:52-66).
 * eegclip_token_block_pack: the five weight matrices (nn.Linear layout (out, in), row-major; wqkv = q | k | v rows stacked) -> bf16 hi | lo planes
 * in MFMA-fragment order, eegclip_token_block_packed_bytes() bytes at `packed` (16-byte aligned): once per optimizer step. */
typedef struct {
    int B;                                     /* samples = workgroups */
    const float* x;                            /* (B, 63, 250) */
    const void* packed;                        /* eegclip_token_block_pack output */
    const float *bv, *pe, *tokens;             /* value-embedding bias (250); positional table rows 0..62 (row stride 250); token table (rows of 250) */
    const long long* ids;                      /* (B) row of `tokens` per sample, NULL: row 0 (the shared token) */
    const float *bqkv, *bo, *ln1_g, *ln1_b, *b1, *b2, *ln2_g, *ln2_b, *ln3_g, *ln3_b;
    float *h, *qkv, *r1, *mu1, *rs1, *f1, *r2, *n2, *mu2, *rs2, *n3, *mu3, *rs3;      /* outputs, rows = B * 64; n2 may be NULL (not stored) */
    void *xp, *hp, *ctxp, *n1p, *g1p;          /* outputs as token planes (B blocks of 64 KB, see eegclip_wgrad_tok): the X operands of the weight gradients --
                                                  EEG sample (token row 1 + channel; NULL: not written), h, ctx (channel 64 head + d), n1, g1; ones column in all but g1 */
    float drop_p, eps, scale;                  /* dropout probability of all five sites (0: evaluation), LayerNorm eps, softmax scale */
    unsigned long long seed;
    unsigned int site_embed, site_attn, site_attn_out, site_ffn_act, site_ffn_out;
    /* joint-subject model (Retrieval/ATMS_retrieval_joint_train.py:172-192, models/subject_layers/Embed.py:127-131,142-144: one value embedding per
       subject, chosen per sample): embed_subject non-NULL = (B) subject of each sample, its matrix is number embed_subject[b] of `packed_embed`
       (eegclip_token_block_pack_embed) and its bias bv + embed_subject[b] * bv_stride (even, 8-byte aligned rows).  NULL: the matrix in `packed`, bias bv. */
    const void* packed_embed;
    const int* embed_subject;
    long long bv_stride;
    /* (round 6) cs_rows non-NULL: the workgroup goes on with the BatchNorm1 batch sums of the conv stack for its sample -- eegclip_cstack_stats1's work
       (Retrieval/ATMS_retrieval.py:102-104: y1 = pool(conv(n3 rows 0 .. cs_H - 1)), per-channel [sum | sumsq] as row b of cs_rows (B, 80) fp64) from the n3
       rows it has just written (still in L2): that launch and its 16 MB read go.  cs_w25 (40, 25) / cs_bias (40): the temporal conv; cs_H <= 64. */
    const float *cs_w25, *cs_bias;
    double* cs_rows;
    int cs_H;
} eegclip_token_block_desc;
long long eegclip_token_block_packed_bytes(void);
int eegclip_token_block_pack(const float* wv, const float* wqkv, const float* wo, const float* w1, const float* w2, void* packed, void* stream);
/* the encoder's whole per-step weight preparation as ONE launch: eegclip_token_block_pack + eegclip_cstack_pack_all (below) */
int eegclip_weight_prep(const float* wv, const float* wqkv, const float* wo, const float* w1, const float* w2, void* packed, const float* Ws, void* cs_packed,
                        void* cs_packed_t, int H, void* stream);
/* n_subjects value-embedding matrices ((250, 250) row-major, w_stride floats apart) -> n_subjects packed operands at `out`
 * (eegclip_token_block_packed_embed_bytes(n_subjects) bytes, 16-byte aligned) */
long long eegclip_token_block_packed_embed_bytes(int n_subjects);
int eegclip_token_block_pack_embed(const float* w0, long long w_stride, int n_subjects, void* out, void* stream);
int eegclip_token_block_fwd(const eegclip_token_block_desc* d, void* stream);

/* backward of the block's dX chain, one workgroup per sample (csrc/token_block.hip), `part`:
 *   0  final LayerNorm', LayerNorm2' + FFN-output dropout', dg1 = df2 W2 with dropout' gelu' (-> dg1 = df1), dn1 = dr2 + df1 W1, LayerNorm1' +
 *      attention-output dropout', dctx = da1 Wo: writes df2, dg1, da1 (token planes), dr1 (residual path), dctx and one partial row of the six LayerNorm
 *      parameter gradients per sample into `partials` (eegclip_token_block_bwd_workspace_floats(B) floats)     (Transformer_EncDec.py:45-51,77-78)
 *   1  dr1 <- dropout'_embed(dr1 + dq Wq + dk Wk + dv Wv), d{q,k,v} read as token planes                     (SelfAttention_Family.py:199-207, Embed.py:162)
 *   2  dln*_g / dln*_b += column sums of `partials`
 * between 0 and 1 runs eegclip_attention_bwd; the weight gradients stay batch-wide GEMMs over what these parts leave in HBM. */
typedef struct {
    int B;
    const void* packed;
    const float *dn3, *n2, *r2, *r1, *f1, *mu1, *rs1, *mu2, *rs2, *mu3, *rs3, *ln1_g, *ln2_g, *ln2_b, *ln3_g;      /* n2 NULL: re-evaluated from r2 (needs ln2_b) */
    float *dr1, *dctx, *partials;
    void *df2p, *dg1p, *da1p, *dr1p;           /* token planes (eegclip_wgrad_tok dY operands): part 0 writes df2p, dg1p, da1p; part 1 writes dr1p unless NULL */
    const void* dqkvp;                         /* part 1 input: dq | dk | dv token planes of eegclip_attention_bwd_x3 (groups B * 64 KB apart) */
    float *dln3_g, *dln3_b, *dln2_g, *dln2_b, *dln1_g, *dln1_b;
    float drop_p;
    unsigned long long seed;
    unsigned int site_embed, site_attn_out, site_ffn_act, site_ffn_out;
} eegclip_token_block_bwd_desc;
long long eegclip_token_block_bwd_workspace_floats(int B);
int eegclip_token_block_bwd(const eegclip_token_block_bwd_desc* d, int part, void* stream);

/* ---- C = A B^T from bf16 hi | lo planes (csrc/gemm_planes.hip): the Linear layers of the diffusion prior (Generation/diffusion_prior.py:167-203) in
 * EEGCLIP_PREC_BF16X3 arithmetic WITHOUT the per-launch fp32 -> plane conversion of eegclip_gemm_f32: A (M, K) and B (N, K) are each two k-contiguous
 * bf16 planes (value = hi + lo; eegclip_split_bf16, eegclip_split_rows, or the plane output of the producing launch), rows lda / ldb ELEMENTS apart
 * (multiples of 8, 16-byte aligned bases).  M, N multiples of 64, K of 32.  Epilogue, in this order: v = acc + bias[n]; Cpre[m][n] = v (if given);
 * v = act(v) (EEGCLIP_ACT_NONE | EEGCLIP_ACT_SILU); v += R[m][n]; v += C[m][n] (accumulate); C[m][n] = v (if given); planes_of = 1: v again as hi | lo
 * planes at p_hi / p_lo (row stride ldp elements), 2: the value stored to Cpre instead.  All fp32 row strides in elements, multiples of 4. */
typedef struct {
    const void *a_hi, *a_lo, *b_hi, *b_lo;
    long long lda, ldb;
    int M, N, K;
    float* C;
    long long ldc;
    float* Cpre;
    long long ldcpre;
    const float* bias;
    const float* R;
    long long ldr;
    void *p_hi, *p_lo;
    long long ldp;
    int act, accumulate, planes_of;
} eegclip_gemm_planes_desc;
int eegclip_gemm_planes(const eegclip_gemm_planes_desc* d, void* stream);

/* matrices (rows a multiple of 64, cols a multiple of 4) -> the bf16 hi | lo planes of their transposes ([cols][ld_out]; the `transpose` field is ignored):
 * the weights a dX plane GEMM contracts over the output index, up to 24 per launch (tiled through LDS -- eegclip_split_rows' transposing path is for small
 * matrices) */
int eegclip_split_transpose(const eegclip_split_item* items, int n, void* stream);

/* ---- InfoNCE of one process at the training batch size (csrc/infonce_small.hip; models/loss.py:122-140 with world_size = 1): the raw logits of all T targets
 * are ONE K-parallel eegclip_head_gemm launch (A = the query planes, B = the targets' planes stacked, M = n, N = T n: partial slabs), then
 *   eegclip_infonce_small_fwd   adds the slabs, keeps the raw logits, row log-sum-exps, positives and per-row-block column partials in `workspace`
 *                               (eegclip_infonce_small_workspace_floats(n, T) floats)
 *   eegclip_infonce_small_grad  finishes the column log-sum-exps, adds sum_t w_t * ClipLoss_t to *loss and d loss / d scale to *dscale, and writes
 *                               G = [G_1 | .. | G_T] (n, ldg), G_t = s dL/dS_t, as bf16 hi | lo planes -- the A operand of dQ = G [B_1; ..; B_T]
 * n a multiple of 8 and <= 1024, T <= 4 (eegclip_infonce_small_supported). */
int eegclip_infonce_small_supported(int n, int T);
long long eegclip_infonce_small_workspace_floats(int n, int T);
int eegclip_infonce_small_fwd(const float* slabs, int nslabs, long long slab_stride, int n, int T, const float* scale, float* workspace, void* stream);
int eegclip_infonce_small_grad(int n, int T, const float* scale, const float* workspace, float w0, float w1, float w2, float w3, void* g_hi, void* g_lo,
                               long long ldg, float* loss, float* dscale, void* stream);

/* ---- the SDXL VAE's layers (csrc/vae.hip; Generation/custom_pipeline_low_level.py:8-31 `vae.encode`, Generation/custom_pipeline.py:421 `vae.decode`; the module is
 * diffusers 0.30.0's AutoencoderKL, not vendored by the reference) on 16-bit PADDED NHWC activations: a tensor is [image][H + 2 pad][W + 2 pad][C] with a zero
 * border that no launch writes (pad = 1; 0 for tensors that only 1 x 1 layers / the attention read).
 *   eegclip_conv16       out[n][y][x][co] = bias[co] + residual[n][y][x][co] + sum_{ky,kx,ci} in[n][y * stride + ky - pad_top][x * stride + kx - pad_left][ci]
 *                        W[co][ky][kx][ci]   (W stored [Cout][KS * KS][Cin], 16-bit; KS 1 | 3, stride 1 | 2; pad_top / pad_left <= in_pad, the bottom / right
 *                        taps must stay inside the padded frame: stride 2 with pad (0, 1, 0, 1) is the encoder's downsampler).  upsample = 1: `in` is (Hi, Wi)
 *                        and the convolution runs over its nearest-2x upsampling (Ho = 2 Hi; KS 3, stride 1, pad 1) -- Upsample2D + conv without the
 *                        upsampled tensor.  Cin % 64 == 0 and Cout % 128 == 0: implicit GEMM on v_mfma_f32_32x32x16 (the tile loop of eegclip_gemm16, row
 *                        addresses recomputed per tap); otherwise a direct kernel for the 3 / 4 / 8-channel layers (weights <= 150 KB).
 *   eegclip_groupnorm16  y = GroupNorm(x) (* SiLU if silu): statistics over the interior pixels, fp64 sums in `sums` (N * groups * 2 doubles, cleared by the
 *                        call); C / groups a multiple of 4
 *   eegclip_softmax_rows16   rows of a 16-bit matrix <- softmax(scale * row), in place (the mid-block attention's score matrix)
 *   eegclip_vae_sample16     z = mean + exp(0.5 clamp(logvar, -30, 20)) * noise from moments (pixels, 2 L) = [mean | logvar] (noise NULL: the mode) */
typedef struct {
    const void* in;
    const void* W;
    void* out;
    const void* bias;
    const void* residual;        /* in the output's layout, or NULL */
    int N, Hi, Wi, Cin, in_pad;
    int Ho, Wo, Cout, out_pad;
    int KS, stride, pad_top, pad_left, upsample, dtype;
} eegclip_conv16_desc;
int eegclip_conv16(const eegclip_conv16_desc* d, void* stream);
int eegclip_groupnorm16(const void* x, int N, int H, int W, int C, int pad, int groups, const void* gamma, const void* beta, float eps, int silu, void* y,
                        int out_pad, double* sums, int dtype, void* stream);
int eegclip_softmax_rows16(void* s, int rows, int cols, long long ld, float scale, int dtype, void* stream);
int eegclip_vae_sample16(const void* moments, const void* noise, void* z, long long pixels, int latent_channels, int dtype, void* stream);

/* ---- the projection head's GEMMs at M = the batch (csrc/head_gemm.hip; Retrieval/ATMS_retrieval.py:157-167 forward, its input gradients, and the query
 * gradient of the loss, models/loss.py:122-140): C[m][n] = sum_k A[m][k] B[n][k] from k-contiguous bf16 hi | lo planes like eegclip_gemm_planes, but
 * K-PARALLEL across workgroups without atomics or in-launch hand-offs: with slices > 1 workgroup (64 x 64 tile, slice s) stores its partial tile into
 * C + s * slab_stride (row stride ldc) and the launch that CONSUMES the result adds the slabs in slice order (eegclip_head_act, eegclip_head_act_bwd,
 * eegclip_residual_layernorm_fwd_slabs, eegclip_layernorm_bwd_slabs, eegclip_proj1x1_bwd_slabs) -- no epilogue field may be set then.  slices = 1: the
 * launch runs the epilogue itself, in this order: v += bias[n]; Cpre[m][n] = v; act: EEGCLIP_ACT_GELU v = gelu(v) | EEGCLIP_ACT_GELU_GRAD
 * v *= gelu'(aux[m][n]); v += R[m][n] (R may be C); C[m][n] = v; p_hi / p_lo: v again as planes.  Any M >= 1, N a multiple of 4, K a multiple of 32 with at
 * least one 32-k tile per slice (<= 16 slices); lda / ldb multiples of 8, fp32 strides multiples of 4, 16-byte aligned bases.
 * eegclip_head_gemm_slices = the slice count that fills the chip for a shape. */
typedef struct {
    const void *a_hi, *a_lo, *b_hi, *b_lo;
    long long lda, ldb;
    int M, N, K, slices;
    long long slab_stride;
    const float* bias;
    float* Cpre;
    long long ldcpre;
    int act;
    const float* aux;
    long long ldaux;
    const float* R;
    long long ldr;
    float* C;
    long long ldc;
    void *p_hi, *p_lo;
    long long ldp;
    int b_kmajor;          /* 0: B (N, K) planes, k contiguous, rows ldb >= K apart.  1: B given as B[k][n] planes (K rows of N, n contiguous, rows ldb >= N apart,
                              N % 8 == 0): the operand of an input-gradient GEMM dX = dY W is the forward's weight planes as they are, no transposed copy */
} eegclip_head_gemm_desc;
int eegclip_head_gemm_slices(int M, int N, int K);
int eegclip_head_gemm(const eegclip_head_gemm_desc* d, void* stream);
/* the launches that CONSUME a K-parallel eegclip_head_gemm result add its partial slabs while they load it (value = sum_{s < nslabs} p[s * slab_stride + i],
 * slice order; nslabs = 1: a plain operand), so the K split costs no launch of its own:
 *   eegclip_head_act            u = slabs + bias; pre = u; out = gelu(u); out_hi | out_lo = out again as bf16 planes (M x N contiguous, N % 4 == 0; any output
 *                               may be NULL)                                                     Proj_eeg's Linear -> GELU, ATMS_retrieval.py:160-162
 *   eegclip_head_act_bwd        dx = base + slabs * gelu'(pre) (base may be NULL or dx itself), dx_hi | dx_lo = dx as planes; n % 4 == 0
 *   eegclip_residual_layernorm_fwd_slabs    eegclip_residual_layernorm_fwd_planes (planes optional) with x = slabs + x_bias[col]
 *   eegclip_layernorm_bwd_slabs             input-gradient half of eegclip_layernorm_bwd with dy = slabs; dx_drop also as planes dd_hi | dd_lo (optional);
 *                                           dy_sum (optional): the summed dy for the parameter-gradient half, a launch of its own
 *   eegclip_proj1x1_bwd_slabs               eegclip_proj1x1_bwd with dfeat = slabs
 *   eegclip_proj1x1_fwd_rows_planes         eegclip_proj1x1_fwd_rows that also leaves feat as bf16 planes (the head's first A operand); rows == NULL: mean / rstd
 *                                           are inputs (eval mode, or statistics finalised by an earlier launch) */
int eegclip_head_act(const float* slabs, int nslabs, long long slab_stride, const float* bias, float* pre, float* out, void* out_hi, void* out_lo, int M, int N,
                     void* stream);
int eegclip_head_act_bwd(const float* slabs, int nslabs, long long slab_stride, const float* pre, const float* base, float* dx, void* dx_hi, void* dx_lo,
                         long long n, void* stream);
int eegclip_residual_layernorm_fwd_slabs(const float* x, const float* resid, float* x_out, float drop_p, unsigned long long seed, unsigned int site,
                                         const float* gamma, const float* beta, float* y, float* mean, float* rstd, const float* gamma2, const float* beta2,
                                         float* y2, float* mean2, float* rstd2, int rows, int cols, float eps, void* y_hi, void* y_lo, int nslabs,
                                         long long slab_stride, const float* x_bias, void* stream);
int eegclip_layernorm_bwd_slabs(const float* dy, int nslabs, long long slab_stride, const float* x, const float* gamma, const float* mean, const float* rstd,
                                float* dx, int rows, int cols, float* dx_drop, void* dd_hi, void* dd_lo, float* dy_sum, float drop_p, unsigned long long seed,
                                unsigned int site, void* stream);
int eegclip_proj1x1_bwd_slabs(const float* dfeat, int nslabs, long long slab_stride, const float* z2, const float* W, const float* y2, const float* mean,
                              const float* rstd, const float* gamma, const float* beta, float* dz2, float* dW, float* dbias, double* sums, float* workspace,
                              int B, float drop_p, unsigned long long seed, unsigned int site, void* stream);
int eegclip_proj1x1_fwd_rows_planes(const float* y2, const double* rows, int nrows, double count, float eps, float momentum, float* mean, float* rstd,
                                    float* running_mean, float* running_var, long long* num_batches_tracked, const float* gamma, const float* beta,
                                    const float* W, const float* bias, float* z2, float* feat, int B, float drop_p, unsigned long long seed,
                                    unsigned int site, void* feat_hi, void* feat_lo, const eegclip_split_item* riders, int n_riders, void* stream);
/* riders (<= 4, or NULL / 0): dense eegclip_split_item entries (transpose = 0, no copy, ld_src = ld_out = cols, rows * cols % 4 == 0) split into planes by EXTRA
 * workgroups of that launch (csrc/split_rider.h): this step's projection-head weights and the loss targets, which the plane GEMMs after it read -- the launch is
 * one small workgroup per sample and leaves the memory system idle, the splits cost no launch, no second stream and no join.
 * The backward's two halves as launches of their own: eegclip_proj1x1_bwd_rows leaves dz2, one fp64 partial row per sample ([dW 1600 | dbias 40 | BatchNorm2-
 * backward sums 80]) in `workspace` (eegclip_proj1x1_bwd_workspace_floats(B) floats) and the 80 BatchNorm sums again as a compact (B, 80) fp64 table bn_rows;
 * eegclip_bn_elu_bwd_apply_rows = eegclip_bn_elu_bwd_apply with the batch sums taken from such a table (`nrows` rows of 2 C doubles, `ld` doubles apart, first
 * column col0; added in a fixed order by every workgroup: no cleared accumulator, no reduction on the dX chain); eegclip_proj1x1_bwd_reduce adds the partial
 * rows' dW / dbias column sums to the gradients (and the BatchNorm sums to `sums` unless NULL) -- read by the optimizer only. */
int eegclip_proj1x1_bwd_rows(const float* dfeat, int nslabs, long long slab_stride, const float* z2, const float* W, const float* y2, const float* mean,
                             const float* rstd, const float* gamma, const float* beta, float* dz2, float* workspace, double* bn_rows, int B, float drop_p,
                             unsigned long long seed, unsigned int site, void* stream);
int eegclip_proj1x1_bwd_reduce(const float* workspace, int B, float* dW, float* dbias, double* sums, void* stream);
int eegclip_bn_elu_bwd_apply_rows(const float* dz, const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                  const double* rows, int nrows, long long ld, int col0, double count, float* dx, float* dgamma, float* dbeta, int outer, int C,
                                  int inner, float drop_p, unsigned long long seed, unsigned int site, void* stream);
/* stage tail of the diffusion prior with plane outputs (csrc/prior.hip; Generation/diffusion_prior.py:173-175,186-199):
 *   forward   y_ln = LayerNorm(x), y_act = dropout(SiLU(y_ln)) (+ skip), mean / rstd saved; act_hi / act_lo: y_act again as bf16 planes (or NULL)
 *   backward  d = dropout'(dact) * silu'(y_ln); dx = LayerNorm'(d) as fp32 (dx, or NULL) and / or bf16 planes (dx_hi / dx_lo, or NULL);
 *             dgamma += sum_rows d * xhat, dbeta += sum_rows d  -- one pass instead of eegclip_silu_bwd + eegclip_layernorm_bwd
 *   eegclip_silu_bwd_planes: dy * silu'(pre) as planes only (n a multiple of 4, 16-byte aligned inputs) */
int eegclip_prior_stage_fwd(const float* x, const float* gamma, const float* beta, const float* skip, float* y_ln, float* y_act, float* mean, float* rstd,
                            void* act_hi, void* act_lo, int rows, int cols, float eps, float drop_p, unsigned long long seed, unsigned int site, void* stream);
int eegclip_prior_stage_bwd(const float* dact, const float* y_ln, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                            void* dx_hi, void* dx_lo, float* dgamma, float* dbeta, int rows, int cols, float drop_p, unsigned long long seed,
                            unsigned int site, float* workspace, void* stream);
/* workspace == NULL: dgamma / dbeta are added by that launch (atomics).  Else (cols % 4 == 0, 16-byte aligned) it takes
 * eegclip_prior_stage_bwd_workspace_floats(rows, cols) floats of per-workgroup partial rows and eegclip_prior_stage_bwd_params adds their column sums to
 * dgamma / dbeta -- a launch of its own, so it can run off the dX chain */
long long eegclip_prior_stage_bwd_workspace_floats(int rows, int cols);
int eegclip_prior_stage_bwd_params(const float* workspace, int rows, int cols, float* dgamma, float* dbeta, void* stream);
int eegclip_silu_bwd_planes(const float* dy, const float* pre, void* dx_hi, void* dx_lo, long long n, void* stream);

/* ---- weight gradients of the transformer block from TOKEN-MAJOR bf16 planes (csrc/wgrad_tok.hip): dW[o][i] += sum_t dY[t][o] X[t][i], t = the
 * 64 B token rows of a batch (models/subject_layers/Transformer_EncDec.py:48-49, SelfAttention_Family.py:199-213, Embed.py:146 w.r.t. the weights).
 * Operand layout ("token planes"): per sample one 64 KB block [hi | lo][64 tokens][256 channels] bf16 -- what eegclip_token_block_fwd / _bwd and
 * eegclip_attention_bwd_x3 write instead of fp32 for the tensors only these GEMMs read (eegclip_tok_planes_from_f32 makes the same from fp32).
 * Channels past the tensor's width are zero, EXCEPT channel 255 of an X operand's hi plane = 1.0: column 255 of the product is then the bias
 * gradient sum_t dY[t][o] (bias_mfma = 1 when X has 256 real channels: the kernel forms it against an all-ones fragment instead).
 * heads_m / heads_n: the operand's channel c = 64 head + d (d < 62) is row / column 62 head + d of `out`; m_groups: dY is m_groups channel groups
 * of 256 (dq | dk | dv), a_group_stride bytes apart, group g = rows 248 g .. of `out` (heads_m) or 256 g ...
 * One launch takes up to 12 problems (gradients that become ready together; the subjects of a joint-subject batch); `slices` K slices per output tile (eegclip_wgrad_tok_slices picks
 * ~one 128 x 128 tile workgroup per CU), partial tiles in `workspace` (eegclip_wgrad_tok_workspace_floats floats) summed in slice order by
 * eegclip_wgrad_tok_reduce: bit-reproducible.  variant 0: 512-thread workgroups (2 waves per SIMD), 1: 256-thread.  Split-bf16 products, fp32 accumulate. */
typedef struct {
    const void* a;                             /* dY token planes (B blocks of 64 KB per channel group) */
    const void* b;                             /* X token planes */
    long long a_group_stride;                  /* bytes between channel groups of a (m_groups > 1) */
    int m_groups, heads_m, heads_n;
    int M, N;                                  /* rows / columns of out */
    float* out;                                /* (M, N) row stride ldo: accumulated into */
    long long ldo;
    float* bias_out;                           /* (M) accumulated into, or NULL */
    int bias_mfma;
    int sample0, samples;                      /* contract over samples sample0 .. sample0 + samples - 1 only (samples = 0: the whole batch) ... */
    const int* sample_index;                   /* ... of this list of B sample numbers (device memory; NULL: identity; one list per launch).  The joint-subject
                                                  value embedding (Embed.py:142-144): list = the batch ordered by subject, one problem per subject present */
} eegclip_wgrad_tok_problem;
int eegclip_wgrad_tok_slices(int total_m_groups, int B);
long long eegclip_wgrad_tok_workspace_floats(const eegclip_wgrad_tok_problem* p, int n_prob, int B, int slices);
int eegclip_wgrad_tok(const eegclip_wgrad_tok_problem* p, int n_prob, int B, int slices, float* workspace, int variant, void* stream);
/* the second half of the operation (its own entry point = its own kernel: per-launch timing): out += the slices of `workspace`, same arguments */
int eegclip_wgrad_tok_reduce(const eegclip_wgrad_tok_problem* p, int n_prob, int B, int slices, float* workspace, void* stream);
/* ... or, at the end of a training step (Retrieval/ATMS_retrieval.py:230-231: loss.backward(); optimizer.step()), the sums STEP THE OPTIMIZER instead of
 * landing in the gradient: [G, G + n) is one run of the flat gradient buffer, P / M / V the parameters and AdamW moments at the same offsets; every
 * problem's out (dense: ldo == N) and bias_out must lie inside the run, disjoint.  Element x: g = G[x] + slices (what the plain form would store),
 * AdamW(step) on P / M / V [x] exactly as eegclip_adamw_step_zero_grad, G[x] = 0.  Elements of the run that no problem covers are stepped from G as it
 * is.  One launch for reduction + optimizer.step() + zero_grad(). */
int eegclip_wgrad_tok_reduce_adamw(const eegclip_wgrad_tok_problem* p, int n_prob, int B, int slices, float* workspace, float* P, float* G, float* M, float* V,
                                   long long n, float lr, float beta1, float beta2, float eps, float weight_decay, long long step, void* stream);
/* the same kernel over PLAIN 2-D planes: out[m][n] += sum_r dY[r][m] X[r][n] with dY (rows, M) and X (rows, N) each a hi and a lo bf16 plane, channel
 * index contiguous, rows lda / ldb ELEMENTS apart (multiples of 8; 16-byte aligned).  rows: multiple of 32.  The kernel reads whole 128-channel tiles:
 * the planes must be readable up to the next multiple of 128 channels past M / N in every row (a column block of a wider buffer, or 256 bytes of
 * padding behind the last row) -- what it reads there does not reach `out`.  bias_out[m] += sum_r dY[r][m].  slices: K slices of this problem (1: plain
 * read-modify-write of out; > 1: fp32 atomics -- small outputs that would not fill the chip otherwise).  Up to 24 problems per launch.
 * The weight gradients of the diffusion prior's Linear layers (Generation/diffusion_prior.py:167-203, rows = the batch). */
typedef struct {
    const void *a_hi, *a_lo;                   /* dY planes */
    long long lda;
    const void *b_hi, *b_lo;                   /* X planes */
    long long ldb;
    int rows, M, N;
    float* out;                                /* (M, N), row stride ldo (multiple of 4), N a multiple of 4 */
    long long ldo;
    float* bias_out;                           /* (M) or NULL */
    int slices;
} eegclip_wgrad_planes_problem;
int eegclip_wgrad_planes(const eegclip_wgrad_planes_problem* p, int n_prob, void* stream);
/* fp32 [rows = 64 B][cols] (row stride ld) -> token planes at dst (rows / 64 blocks of 64 KB, 16-byte aligned); heads: column 62 head + d -> channel
 * 64 head + d; ones: hi[.][255] = 1.0 */
int eegclip_tok_planes_from_f32(const float* src, long long ld, int rows, int cols, int heads, int ones, void* dst, void* stream);

/* ---- the conv stack of Enc_eeg recomputed from the token rows (csrc/cstack*.hip, round 5).  Retrieval/ATMS_retrieval.py:102-106:
 *   Conv2d(1,40,(1,25)) -> AvgPool2d((1,51),(1,5)) -> BatchNorm2d(40) -> ELU -> Conv2d(40,40,(H,1))
 * y1 = conv + pool output (B,40,H,36), its activation z1 and their gradients never exist in HBM: every kernel re-derives the y1 tile it needs on the
 * bf16 matrix cores (split-bf16 products, EEGCLIP_PREC_BF16X3 arithmetic) from the H token rows of a sample staged once per workgroup in LDS, and
 * chains the next contraction from the accumulator registers.  x: token rows of 250 floats at x + b*xs_b + h*xs_h (h < H <= 64); w25 (40,25) taps.
 *   eegclip_cstack_pack    Ws (40,40,H) fp32 -> bf16 hi | lo planes in MFMA-fragment order (eegclip_cstack_packed_bytes(H) bytes, 16-byte aligned);
 *                          once per optimizer step
 *   eegclip_cstack_stats1  BatchNorm1 batch sums of y1: rows[b] = [sum_c(40) | sumsq_c(40)] (fp64) of sample b -- no atomics, nothing to clear
 *   eegclip_cstack_fwd     y2[b,o,w] = bias2[o] + sum_{c,h} Ws[o,c,h] ELU(BN1(y1))[b,c,h,w]; BatchNorm1 statistics from `nstat1` partial rows (summed
 *                          in row order by every workgroup; nstat1 = 1: the all-reduced sums of a data-parallel job) with element count `count1`;
 *                          workgroup 0 stores mean1 / rstd1 (for the backward) and updates the running statistics + step counter (what
 *                          eegclip_bn_finalize did).  stat1 = NULL: eval mode, mean1 / rstd1 are INPUTS.  stat2 (optional): BatchNorm2 partial rows
 *                          [sum_o | sumsq_o] of y2 per sample. */
typedef struct {
    int B, H;
    const float* x;
    long long xs_b, xs_h;
    const float *w25, *bias1;
    const double* stat1;
    int nstat1;
    double count1;
    float eps, momentum;
    const float *gamma1, *beta1;
    float *mean1, *rstd1;
    float *run_mean1, *run_var1;
    long long* nbt1;
    const void* packed;
    const float* bias2;
    float* y2;
    double* stat2;
} eegclip_cstack_fwd_desc;
long long eegclip_cstack_packed_bytes(int H);
int eegclip_cstack_pack(const float* Ws, void* packed, int H, void* stream);
int eegclip_cstack_stats1(const float* x, long long xs_b, long long xs_h, const float* w25, const float* bias, double* rows, int B, int H, void* stream);
int eegclip_cstack_fwd(const eegclip_cstack_fwd_desc* d, void* stream);

/* backward of the same stack, given dy2 (B,40,36) = the gradient entering the spatial conv's output (16-byte aligned):
 *   eegclip_cstack_pack_t     Ws -> Ws^T fragments for dz1 = Ws^T dy2 (eegclip_cstack_packed_t_bytes(H) bytes, 16-byte aligned); once per optimizer step
 *   eegclip_cstack_bwd_stats  rows_out[b] = [sum da (40) | sum da * xhat (40)] (fp64) of sample b, da = dz1 * ELU'(BN1(y1))
 *   eegclip_cstack_bwd_apply  dy1 = BatchNorm1-backward(da) with the sums of `nstat` partial rows `stat` (fixed summation order; nstat = 1: all-reduced
 *                             sums; a zero row: eval-mode BatchNorm, no batch terms) and element count `count`; token-row gradients dx[b][h][0..249]
 *                             (rows h < H overwritten, same strides as x), taps gradient dw25 += (per-sample partials in `dw_partials`,
 *                             eegclip_cstack_bwd_workspace_floats(B) floats, summed in sample order), dgamma / dbeta += this rank's own sums
 *                             (stat_local / nstat_local; NULL: stat)
 *   eegclip_cstack_bwd_w2     dWs[o][c][h] += sum_{b,w} dy2[b][o][w] ELU(BN1(y1))[b][c][h][w]   (workspace: eegclip_cstack_bwd_w2_workspace_floats(B, H) floats;
 *                             per-group slabs summed in a fixed order: bit-reproducible)
 * mean1 / rstd1: the batch statistics eegclip_cstack_fwd stored. */
typedef struct {
    int B, H;
    const float* x;
    long long xs_b, xs_h;
    const float *w25, *bias1;
    const float *mean1, *rstd1, *gamma1, *beta1;
    const void* packed_t;
    const float* dy2;
    double* rows_out;
    const double* stat;
    int nstat;
    double count;
    const double* stat_local;
    int nstat_local;
    float *dgamma, *dbeta;
    float* dx;
    float* dw_partials;
    float* dw25;
} eegclip_cstack_bwd_desc;
long long eegclip_cstack_packed_t_bytes(int H);
int eegclip_cstack_pack_t(const float* Ws, void* packed_t, int H, void* stream);
/* both fragment sets of a step in one launch (packed_t may be NULL: forward only) */
int eegclip_cstack_pack_all(const float* Ws, void* packed, void* packed_t, int H, void* stream);
int eegclip_cstack_bwd_stats(const eegclip_cstack_bwd_desc* d, void* stream);
long long eegclip_cstack_bwd_workspace_floats(int B);
int eegclip_cstack_bwd_apply(const eegclip_cstack_bwd_desc* d, void* stream);
/* dw25 == NULL in the descriptor: eegclip_cstack_bwd_apply leaves the tap gradient as its B partial rows and this call sums them (dw25 += ...; only the
 * optimizer reads the result, so the launch can run beside the rest of the backward) */
int eegclip_cstack_bwd_taps_reduce(const float* dw_partials, int B, float* dw25, void* stream);
long long eegclip_cstack_bwd_w2_workspace_floats(int B, int H);
int eegclip_cstack_bwd_w2(const float* x, long long xs_b, long long xs_h, const float* w25, const float* bias1, const float* mean1, const float* rstd1,
                          const float* gamma1, const float* beta1, const float* dy2, float* dWs, float* workspace, int B, int H, void* stream);

/* ---- per-kernel timing by the kernel's own GPU timestamps (bench.py roofline): eegclip_time_next_launch(start, stop) arms a pair of
 * library-owned events for the FIRST kernel the calling thread's next entry point launches (hipExtLaunchKernel start / stop events: what
 * rocprofv3 reports, without the marker packets of an event bracket).  Read with eegclip_timing_elapsed_ms after synchronising. */
void* eegclip_timing_event_create(void);
int eegclip_timing_event_destroy(void* event);
int eegclip_time_next_launch(void* start, void* stop);          /* (NULL, NULL) disarms a pair that no launch has consumed */
float eegclip_timing_elapsed_ms(void* start, void* stop);

#ifdef __cplusplus
}
#endif
#endif /* EEGCLIP_H */
