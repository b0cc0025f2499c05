// 64-token multi-head self-attention of the ATM-S encoder (models/subject_layers/SelfAttention_Family.py:56-75):
//   A = dropout(softmax(Q K^T / sqrt(E)));  O = A V        L = 64 tokens, H = 4 heads, E = 62.
//
// One wavefront per (sample, head): lane i owns query row i, so the whole softmax row (max, sum, probabilities) is
// lane-local -- no cross-lane reduction at all.  K and V rows are read from LDS as wave-wide broadcasts
// (ds_read_b128, all lanes same address => conflict free); the 64x64 score matrix lives in LDS with a 65-float row
// stride so "lane i walks row i" (forward) and "lane j walks column j" (backward, dK/dV) are both bank-conflict free.
// Nothing but Q/K/V in and the context out touches HBM: the (B,4,64,64) probability tensor the reference
// materialises (and its dropout mask) never exists; backward recomputes P and regenerates the Philox mask.
#include "eeg_common.h"

namespace eeg {

constexpr int AT_L = 64;      // tokens (= lanes)
constexpr int AT_EP = 64;     // padded head width in LDS
constexpr int AT_SP = 65;     // padded score-row stride

struct attn_args {
    const float* qkv;   // (B*L, ld): q at col h*E+e, k at HE + h*E+e, v at 2HE + h*E+e
    int B, H, E, ld;
    float scale;
    float drop_p;
    unsigned long long seed;
    unsigned site;
};

// cooperative coalesced load of one (64 x E) head slice into LDS rows of stride `stride`, zero padded to 64 columns
__device__ __forceinline__ void load_head(float* dst, int stride, const float* src, int ld, int E, int lane) {
#pragma unroll 4
    for (int r = 0; r < AT_L; ++r) dst[r * stride + lane] = lane < E ? src[(long long)r * ld + lane] : 0.f;
}
__device__ __forceinline__ void store_head(float* dst, int ld, const float* src, int stride, int E, int lane) {
#pragma unroll 4
    for (int r = 0; r < AT_L; ++r)
        if (lane < E) dst[(long long)r * ld + lane] = src[r * stride + lane];
}

// acc[e] += w * row[e]  (row broadcast from LDS, 16 x float4)
__device__ __forceinline__ void axpy_row(float (&acc)[AT_EP], float w, const float* row) {
#pragma unroll
    for (int e4 = 0; e4 < AT_EP / 4; ++e4) {
        const float4 r = *reinterpret_cast<const float4*>(row + 4 * e4);
        acc[4 * e4 + 0] += w * r.x;
        acc[4 * e4 + 1] += w * r.y;
        acc[4 * e4 + 2] += w * r.z;
        acc[4 * e4 + 3] += w * r.w;
    }
}
__device__ __forceinline__ float dot_row(const float (&a)[AT_EP], const float* row) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int e4 = 0; e4 < AT_EP / 4; ++e4) {
        const float4 r = *reinterpret_cast<const float4*>(row + 4 * e4);
        s0 += a[4 * e4 + 0] * r.x;
        s1 += a[4 * e4 + 1] * r.y;
        s2 += a[4 * e4 + 2] * r.z;
        s3 += a[4 * e4 + 3] * r.w;
    }
    return (s0 + s1) + (s2 + s3);
}

__global__ __launch_bounds__(64) void attention_fwd_kernel(const attn_args a, float* __restrict__ ctx /* (B*L, H*E) */) {
    EEG_LDS_BASE(float, lds);
    float* Ss = lds;                        // [64][65]  Q rows first, then scores / probabilities
    float* Ks = lds + AT_L * AT_SP;         // [64][64]
    float* Vs = Ks + AT_L * AT_EP;          // [64][64]
    const int lane = threadIdx.x;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int HE = a.H * a.E;
    const float* base = a.qkv + (long long)b * AT_L * a.ld + h * a.E;
    load_head(Ss, AT_SP, base, a.ld, a.E, lane);
    load_head(Ks, AT_EP, base + HE, a.ld, a.E, lane);
    load_head(Vs, AT_EP, base + 2 * HE, a.ld, a.E, lane);
    __syncthreads();

    float* srow = Ss + lane * AT_SP;        // this lane's private row: q, then s, then p
    float q[AT_EP];
#pragma unroll
    for (int e = 0; e < AT_EP; ++e) q[e] = srow[e];
    float mx = -INFINITY;
    for (int j = 0; j < AT_L; ++j) {
        const float s = dot_row(q, Ks + j * AT_EP) * a.scale;
        srow[j] = s;
        mx = fmaxf(mx, s);
    }
    float sum = 0.f;
    for (int j = 0; j < AT_L; ++j) {
        const float p = expf(srow[j] - mx);
        srow[j] = p;
        sum += p;
    }
    const float inv = 1.0f / sum;
    const float ks = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const unsigned long long rbase = ((unsigned long long)blockIdx.x * AT_L + lane) * AT_L;
    float o[AT_EP];
#pragma unroll
    for (int e = 0; e < AT_EP; ++e) o[e] = 0.f;
    for (int j = 0; j < AT_L; ++j) {
        float p = srow[j] * inv;
        if (a.drop_p > 0.f) p = dropout_keep(a.seed, a.site, rbase + j, a.drop_p) ? p * ks : 0.f;
        axpy_row(o, p, Vs + j * AT_EP);
    }
    __syncthreads();                         // every lane is done reading K before it is reused as the output stage
#pragma unroll
    for (int e = 0; e < AT_EP; ++e) Ks[lane * AT_EP + e] = o[e];
    __syncthreads();
    store_head(ctx + (long long)b * AT_L * HE + h * a.E, HE, Ks, AT_EP, a.E, lane);
}

__global__ __launch_bounds__(64) void attention_bwd_kernel(const attn_args a, const float* __restrict__ dctx /* (B*L, H*E) */,
                                                            float* __restrict__ dqkv /* (B*L, ld) */) {
    EEG_LDS_BASE(float, lds);
    float* P = lds;                          // [64][65]  Q rows -> scores -> P -> dropped P (for dV)
    float* D = lds + AT_L * AT_SP;           // [64][65]  dO rows -> dP -> dS (for dQ, dK)
    float* X = D + AT_L * AT_SP;             // [64][64]  K, later Q
    float* Y = X + AT_L * AT_EP;             // [64][64]  V, later dO
    const int lane = threadIdx.x;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int HE = a.H * a.E;
    const float* qbase = a.qkv + (long long)b * AT_L * a.ld + h * a.E;
    const float* dobase = dctx + (long long)b * AT_L * HE + h * a.E;
    float* dqbase = dqkv + (long long)b * AT_L * a.ld + h * a.E;
    load_head(P, AT_SP, qbase, a.ld, a.E, lane);
    load_head(D, AT_SP, dobase, HE, a.E, lane);
    load_head(X, AT_EP, qbase + HE, a.ld, a.E, lane);
    load_head(Y, AT_EP, qbase + 2 * HE, a.ld, a.E, lane);
    __syncthreads();

    float* prow = P + lane * AT_SP;
    float* drow = D + lane * AT_SP;
    const float ks = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const unsigned long long rbase = ((unsigned long long)blockIdx.x * AT_L + lane) * AT_L;
    float r[AT_EP];
    // ---- recompute P (row `lane`) -------------------------------------------------------------
#pragma unroll
    for (int e = 0; e < AT_EP; ++e) r[e] = prow[e];          // q_i
    float mx = -INFINITY;
    for (int j = 0; j < AT_L; ++j) {
        const float s = dot_row(r, X + j * AT_EP) * a.scale;
        prow[j] = s;
        mx = fmaxf(mx, s);
    }
    float sum = 0.f;
    for (int j = 0; j < AT_L; ++j) {
        const float p = expf(prow[j] - mx);
        prow[j] = p;
        sum += p;
    }
    const float inv = 1.0f / sum;
    // ---- dP = dO V^T through the dropout mask; delta = sum_j P * dP ------------------------------
#pragma unroll
    for (int e = 0; e < AT_EP; ++e) r[e] = drow[e];          // dO_i
    unsigned long long keep_bits = ~0ull;
    float delta = 0.f;
    for (int j = 0; j < AT_L; ++j) {
        float dp = dot_row(r, Y + j * AT_EP);
        if (a.drop_p > 0.f) {
            const bool k = dropout_keep(a.seed, a.site, rbase + j, a.drop_p);
            if (!k) keep_bits &= ~(1ull << j);
            dp = k ? dp * ks : 0.f;
        }
        const float p = prow[j] * inv;
        prow[j] = p;
        drow[j] = dp;
        delta += p * dp;
    }
    // ---- dS = P * (dP - delta) * scale  -> D ;  dropped P -> P -------------------------------------
    for (int j = 0; j < AT_L; ++j) {
        const float p = prow[j];
        drow[j] = p * (drow[j] - delta) * a.scale;
        prow[j] = ((keep_bits >> j) & 1ull) ? p * ks : 0.f;
    }
    // ---- dQ_i = sum_j dS_ij K_j -------------------------------------------------------------------
#pragma unroll
    for (int e = 0; e < AT_EP; ++e) r[e] = 0.f;
    for (int j = 0; j < AT_L; ++j) axpy_row(r, drow[j], X + j * AT_EP);
    __syncthreads();                          // all lanes done with K (X) and V (Y)
#pragma unroll
    for (int e = 0; e < AT_EP; ++e) Y[lane * AT_EP + e] = r[e];
    __syncthreads();
    store_head(dqbase, a.ld, Y, AT_EP, a.E, lane);
    __syncthreads();
    // ---- phase 2: lane j owns key/value row j; needs every Q_i and dO_i as broadcasts --------------
    load_head(X, AT_EP, qbase, a.ld, a.E, lane);
    load_head(Y, AT_EP, dobase, HE, a.E, lane);
    __syncthreads();
    float dv[AT_EP];
#pragma unroll
    for (int e = 0; e < AT_EP; ++e) { r[e] = 0.f; dv[e] = 0.f; }
    for (int i = 0; i < AT_L; ++i) {
        axpy_row(r, D[i * AT_SP + lane], X + i * AT_EP);      // dK_j += dS_ij Q_i
        axpy_row(dv, P[i * AT_SP + lane], Y + i * AT_EP);     // dV_j += Pdrop_ij dO_i
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < AT_EP; ++e) { X[lane * AT_EP + e] = r[e]; Y[lane * AT_EP + e] = dv[e]; }
    __syncthreads();
    store_head(dqbase + HE, a.ld, X, AT_EP, a.E, lane);
    store_head(dqbase + 2 * HE, a.ld, Y, AT_EP, a.E, lane);
}

}  // namespace eeg

using namespace eeg;

static int attn_check(const float* qkv, int B, int L, int H, int E, int ld, float drop_p) {
    if (!qkv || B < 1 || L != AT_L || H < 1 || E < 1 || E > AT_EP || ld < 3 * H * E || drop_p < 0.f || drop_p >= 1.f)
        return EEGCLIP_EINVAL;
    return 0;
}

extern "C" int eegclip_attention_fwd(const float* qkv, float* ctx, int B, int L, int H, int E, int ld, float scale, float drop_p,
                                     unsigned long long seed, unsigned site, void* stream) {
    if (int rc = attn_check(qkv, B, L, H, E, ld, drop_p)) return rc;
    if (!ctx) return EEGCLIP_EINVAL;
    attn_args a{qkv, B, H, E, ld, scale, drop_p, seed, site};
    const size_t lds = (AT_L * AT_SP + 2 * AT_L * AT_EP) * sizeof(float);
    EEG_LAUNCH(attention_fwd_kernel, dim3(B * H), dim3(64), lds, stream, a, ctx);
    return (int)hipGetLastError();
}

extern "C" int eegclip_attention_bwd(const float* qkv, const float* dctx, float* dqkv, int B, int L, int H, int E, int ld, float scale,
                                     float drop_p, unsigned long long seed, unsigned site, void* stream) {
    if (int rc = attn_check(qkv, B, L, H, E, ld, drop_p)) return rc;
    if (!dctx || !dqkv) return EEGCLIP_EINVAL;
    attn_args a{qkv, B, H, E, ld, scale, drop_p, seed, site};
    const size_t lds = (2 * AT_L * AT_SP + 2 * AT_L * AT_EP) * sizeof(float);
    EEG_LAUNCH(attention_bwd_kernel, dim3(B * H), dim3(64), lds, stream, a, dctx, dqkv);
    return (int)hipGetLastError();
}
