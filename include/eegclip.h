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

#define EEGCLIP_ABI_VERSION 1
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

/* C[m,n] (+)= epilogue( alpha * sum_k A[m,k] * B[k,n] )          fp32 in, fp32 MFMA (exact f32), fp32 out.
 * epilogue order: +bias_n[n] +bias_m[m] -> (store Cpre) -> act -> dropout(p, Philox(seed, site, m*N+n)) -> +R[m,n]
 * Replaces every nn.Linear / 1x1-conv / (63x1)-conv GEMM and its backward on the path:
 *   models/subject_layers/Embed.py:146-149 (value embedding + PE), SelfAttention_Family.py:199-201,213 (Q/K/V/out),
 *   Transformer_EncDec.py:48-51 (FFN), Retrieval/ATMS_retrieval.py:106,113 (spatial / 1x1 conv), :157-167 (head),
 *   models/loss.py:122-123 (logits), Generation/diffusion_prior.py:167-203 (prior MLP).
 * split_k > 1: K is cut into split_k slices whose partial products are atomically added into C (C must be
 * zero or hold the value to accumulate onto); bias is added by slice 0; act/dropout/R/Cpre are not allowed. */
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
    int accumulate;       /* 1: C += result */
    int act;              /* EEGCLIP_ACT_* */
    float drop_p;         /* 0 = no dropout */
    unsigned long long seed;
    unsigned int drop_site;
    int split_k;          /* >= 1 */
} eegclip_gemm_desc;

int eegclip_gemm_f32(const eegclip_gemm_desc* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EEGCLIP_H */
