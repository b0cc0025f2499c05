// 64-token multi-head self-attention of the ATM-S encoder (models/subject_layers/SelfAttention_Family.py:56-75):
//   A = dropout(softmax(Q K^T / sqrt(E)));  O = A V        L = 64 tokens, H = 4 heads, E = 62 (padded to 64 in LDS only).
//
// One 256-thread workgroup per (sample, head); wave w owns query rows 16w..16w+15.  Every contraction runs on the f32 matrix
// cores (v_mfma_f32_16x16x4_f32: exact f32): S = Q K^T, O = P V and, in backward, dP = dO V^T, dQ = dS K, dK = dS^T Q,
// dV = P^T dO -- 128 MFMAs per wave forward, 320 backward.  Q/K/V(/dO) tiles live in LDS with a 68-float row stride:
//   * k-contiguous operands (Q, K, dO, V^T-as-rows, P, dS rows) are fetched as one ds_read_b128 per lane and the four
//     floats feed four consecutive MFMA k-steps (the k-slot -> lane assignment is free as long as A and B agree);
//   * k-strided operands (V, K, Q, dO, P^T, dS^T columns) are ds_read_b32 of 16 consecutive floats per 16-lane group.
// The softmax works on the MFMA accumulator layout (row = 4*(lane>>4)+r, col = lane&15): row max / sum are 4 xor-shuffles.
// Nothing but Q/K/V in and the context out touches HBM: the (B,4,64,64) probability tensor the reference materialises (and its
// dropout mask) never exists; backward recomputes P and regenerates the Philox mask.
#include "eeg_common.h"

namespace eeg {

constexpr int AT_L = 64;      // tokens
constexpr int AT_LD = 68;     // LDS row stride in floats (272 B: 16-B aligned rows, 4-bank skew per row)
constexpr int AT_T = AT_L * AT_LD;

struct attn_args {
    const float* qkv;   // (B*L, ld): q at col h*E+e, k at HE + h*E+e, v at 2HE + h*E+e
    int B, H, E, ld;
    float scale;
    float drop_p;
    unsigned long long seed;
    unsigned site;
};

// cooperative coalesced load of one (64 x E) head slice into LDS [64][AT_LD], zero padded to 64 columns
__device__ __forceinline__ void load_head(float* dst, const float* src, int ld, int E) {
    for (int i = threadIdx.x; i < AT_L * AT_L; i += blockDim.x) {
        const int r = i >> 6, c = i & 63;
        dst[r * AT_LD + c] = c < E ? src[(long long)r * ld + c] : 0.f;
    }
}

// C[16 x 64] (4 n-tiles) = A[16 x 64] * B^T   where both A (rows a_row0..+15) and B (rows 16t..16t+15 of `bm`) are k-contiguous in LDS
__device__ __forceinline__ void mma_rows_x_rowsT(const float* am, int a_row0, const float* bm, f32x4 (&c)[4], int lane) {
    const int fr = lane & 15, g = lane >> 4;
#pragma unroll
    for (int KK = 0; KK < 4; ++KK) {
        const float4 a = *reinterpret_cast<const float4*>(am + (a_row0 + fr) * AT_LD + 16 * KK + 4 * g);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 b = *reinterpret_cast<const float4*>(bm + (16 * t + fr) * AT_LD + 16 * KK + 4 * g);
            c[t] = mfma_f32_16x16x4(a.x, b.x, c[t]);
            c[t] = mfma_f32_16x16x4(a.y, b.y, c[t]);
            c[t] = mfma_f32_16x16x4(a.z, b.z, c[t]);
            c[t] = mfma_f32_16x16x4(a.w, b.w, c[t]);
        }
    }
}

// C[16 x 64] = A[16 x 64] * B   A rows k-contiguous (rows a_row0..), B = bm[k][n] row-major (k-strided operand)
__device__ __forceinline__ void mma_rows_x_mat(const float* am, int a_row0, const float* bm, f32x4 (&c)[4], int lane) {
    const int fr = lane & 15, g = lane >> 4;
#pragma unroll
    for (int KK = 0; KK < 4; ++KK) {
        const float4 a = *reinterpret_cast<const float4*>(am + (a_row0 + fr) * AT_LD + 16 * KK + 4 * g);
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float* brow = bm + (16 * KK + 4 * g + s) * AT_LD + fr;
#pragma unroll
            for (int t = 0; t < 4; ++t) c[t] = mfma_f32_16x16x4(av[s], brow[16 * t], c[t]);
        }
    }
}

// C[16 x 64] = A^T[16 x 64] * B   with A^T[m][k] = am[k][a_col0 + m] (column block of a row-major matrix), B = bm[k][n] row-major
__device__ __forceinline__ void mma_colsT_x_mat(const float* am, int a_col0, const float* bm, f32x4 (&c)[4], int lane) {
    const int fr = lane & 15, g = lane >> 4;
#pragma unroll
    for (int KK = 0; KK < 4; ++KK) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = 16 * KK + 4 * g + s;
            const float a = am[k * AT_LD + a_col0 + fr];
            const float* brow = bm + k * AT_LD + fr;
#pragma unroll
            for (int t = 0; t < 4; ++t) c[t] = mfma_f32_16x16x4(a, brow[16 * t], c[t]);
        }
    }
}

__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// scores (accumulator layout) -> probabilities in place; returns nothing (rows are normalised)
__device__ __forceinline__ void softmax_rows(f32x4 (&s)[4], float scale) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t) { s[t][r] *= scale; mx = fmaxf(mx, s[t][r]); }
        mx = group16_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) { s[t][r] = expf(s[t][r] - mx); sum += s[t][r]; }
        const float inv = 1.0f / group16_sum(sum);
#pragma unroll
        for (int t = 0; t < 4; ++t) s[t][r] *= inv;
    }
}

__device__ __forceinline__ void zero4(f32x4 (&c)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) c[t] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// accumulator tile set -> global rows (row-major, `ld`), columns < E
__device__ __forceinline__ void store_rows(float* dst, int ld, int row0, const f32x4 (&c)[4], int E, int lane) {
    const int fr = lane & 15, g = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float* p = dst + (long long)(row0 + 4 * g + r) * ld;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (16 * t + fr < E) p[16 * t + fr] = c[t][r];
    }
}

__global__ __launch_bounds__(256) void attention_fwd_kernel(const attn_args a, float* __restrict__ ctx /* (B*L, H*E) */) {
    EEG_LDS_BASE(float, lds);
    float *Qs = lds, *Ks = lds + AT_T, *Vs = lds + 2 * AT_T, *Ps = lds + 3 * AT_T;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int HE = a.H * a.E;
    const float* base = a.qkv + (long long)b * AT_L * a.ld + h * a.E;
    load_head(Qs, base, a.ld, a.E);
    load_head(Ks, base + HE, a.ld, a.E);
    load_head(Vs, base + 2 * HE, a.ld, a.E);
    __syncthreads();
    f32x4 s[4];
    zero4(s);
    mma_rows_x_rowsT(Qs, 16 * w, Ks, s, lane);               // S[16w + 4g + r][16t + fr]
    softmax_rows(s, a.scale);
    const float ks = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * w + 4 * g + r;
        const unsigned long long rbase = ((unsigned long long)blockIdx.x * AT_L + row) * AT_L;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float p = s[t][r];
            if (a.drop_p > 0.f) p = dropout_keep(a.seed, a.site, rbase + 16 * t + fr, a.drop_p) ? p * ks : 0.f;
            Ps[row * AT_LD + 16 * t + fr] = p;
        }
    }
    __syncthreads();                                          // P rows of this wave are complete in LDS (A-operand order on re-read)
    f32x4 o[4];
    zero4(o);
    mma_rows_x_mat(Ps, 16 * w, Vs, o, lane);                  // O[row][e] = sum_j P[row][j] V[j][e]
    store_rows(ctx + (long long)b * AT_L * HE + h * a.E, HE, 16 * w, o, a.E, lane);
}

__global__ __launch_bounds__(256) void attention_bwd_kernel(const attn_args a, const float* __restrict__ dctx /* (B*L, H*E) */,
                                                             float* __restrict__ dqkv /* (B*L, ld) */) {
    EEG_LDS_BASE(float, lds);
    float *Qs = lds, *Ks = lds + AT_T, *Vs = lds + 2 * AT_T, *Ds = lds + 3 * AT_T, *Ps = lds + 4 * AT_T, *Ss = lds + 5 * AT_T;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int HE = a.H * a.E;
    const float* qbase = a.qkv + (long long)b * AT_L * a.ld + h * a.E;
    float* dqbase = dqkv + (long long)b * AT_L * a.ld + h * a.E;
    load_head(Qs, qbase, a.ld, a.E);
    load_head(Ks, qbase + HE, a.ld, a.E);
    load_head(Vs, qbase + 2 * HE, a.ld, a.E);
    load_head(Ds, dctx + (long long)b * AT_L * HE + h * a.E, HE, a.E);
    __syncthreads();
    f32x4 p[4], dp[4];
    zero4(p);
    zero4(dp);
    mma_rows_x_rowsT(Qs, 16 * w, Ks, p, lane);                // recompute S, then P
    softmax_rows(p, a.scale);
    mma_rows_x_rowsT(Ds, 16 * w, Vs, dp, lane);               // dP[row][j] = sum_e dO[row][e] V[j][e]
    const float ks = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * w + 4 * g + r;
        const unsigned long long rbase = ((unsigned long long)blockIdx.x * AT_L + row) * AT_L;
        float pd[4], dd[4];
        float delta = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            bool keep = true;
            if (a.drop_p > 0.f) keep = dropout_keep(a.seed, a.site, rbase + 16 * t + fr, a.drop_p);
            pd[t] = keep ? p[t][r] * ks : 0.f;                // dropped probabilities (feed dV)
            dd[t] = keep ? dp[t][r] * ks : 0.f;               // gradient w.r.t. the un-dropped probabilities
            delta += p[t][r] * dd[t];
        }
        delta = group16_sum(delta);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            Ps[row * AT_LD + 16 * t + fr] = pd[t];
            Ss[row * AT_LD + 16 * t + fr] = p[t][r] * (dd[t] - delta) * a.scale;      // dS (scale folded in)
        }
    }
    __syncthreads();                                          // all 64 rows of Pdrop and dS are in LDS
    f32x4 acc[4];
    zero4(acc);
    mma_rows_x_mat(Ss, 16 * w, Ks, acc, lane);                // dQ[row][e] = sum_j dS[row][j] K[j][e]
    store_rows(dqbase, a.ld, 16 * w, acc, a.E, lane);
    zero4(acc);
    mma_colsT_x_mat(Ss, 16 * w, Qs, acc, lane);               // dK[j][e] = sum_i dS[i][j] Q[i][e]      (rows j = 16w..)
    store_rows(dqbase + HE, a.ld, 16 * w, acc, a.E, lane);
    zero4(acc);
    mma_colsT_x_mat(Ps, 16 * w, Ds, acc, lane);               // dV[j][e] = sum_i Pdrop[i][j] dO[i][e]
    store_rows(dqbase + 2 * HE, a.ld, 16 * w, acc, a.E, lane);
}

}  // namespace eeg

using namespace eeg;

static int attn_check(const float* qkv, int B, int L, int H, int E, int ld, float drop_p) {
    if (!qkv || B < 1 || L != AT_L || H < 1 || E < 1 || E > 64 || ld < 3 * H * E || drop_p < 0.f || drop_p >= 1.f) return EEGCLIP_EINVAL;
    return 0;
}

extern "C" int eegclip_attention_fwd(const float* qkv, float* ctx, int B, int L, int H, int E, int ld, float scale, float drop_p,
                                     unsigned long long seed, unsigned site, void* stream) {
    if (int rc = attn_check(qkv, B, L, H, E, ld, drop_p)) return rc;
    if (!ctx) return EEGCLIP_EINVAL;
    attn_args a{qkv, B, H, E, ld, scale, drop_p, seed, site};
    EEG_LAUNCH(attention_fwd_kernel, dim3(B * H), dim3(256), 4 * AT_T * sizeof(float), stream, a, ctx);
    return (int)hipGetLastError();
}

extern "C" int eegclip_attention_bwd(const float* qkv, const float* dctx, float* dqkv, int B, int L, int H, int E, int ld, float scale,
                                     float drop_p, unsigned long long seed, unsigned site, void* stream) {
    if (int rc = attn_check(qkv, B, L, H, E, ld, drop_p)) return rc;
    if (!dctx || !dqkv) return EEGCLIP_EINVAL;
    attn_args a{qkv, B, H, E, ld, scale, drop_p, seed, site};
    EEG_LAUNCH(attention_bwd_kernel, dim3(B * H), dim3(256), 6 * AT_T * sizeof(float), stream, a, dctx, dqkv);
    return (int)hipGetLastError();
}
