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

typedef float f32x2 __attribute__((ext_vector_type(2)));

// cooperative coalesced load of (64 x E) head slices into LDS [64][AT_LD], zero padded to 64 columns.  All N slices' global loads are
// issued before the first LDS store (one exposed memory latency per kernel instead of one per slice); VEC2: 8-byte loads when E, the
// row strides and the base pointers are even (the encoder: E = 62, ld = 744 / 248).
template <int N, bool VEC2>
__device__ __forceinline__ void load_heads(float* const (&dst)[N], const float* const (&src)[N], const int (&ld)[N], int E) {
    const int t = threadIdx.x;
    if (VEC2) {
        f32x2 v[N][8];
#pragma unroll
        for (int n = 0; n < N; ++n)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = t + 256 * j, r = i >> 5, c = 2 * (i & 31);
                v[n][j] = *reinterpret_cast<const f32x2*>(src[n] + (long long)r * ld[n] + (c < E ? c : 0));
            }
#pragma unroll
        for (int n = 0; n < N; ++n)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = t + 256 * j, r = i >> 5, c = 2 * (i & 31);
                *reinterpret_cast<f32x2*>(dst[n] + r * AT_LD + c) = c < E ? v[n][j] : f32x2{0.f, 0.f};
            }
    } else {
#pragma unroll
        for (int n = 0; n < N; ++n)
            for (int i = t; i < AT_L * AT_L; i += 256) {
                const int r = i >> 6, c = i & 63;
                dst[n][r * AT_LD + c] = c < E ? src[n][(long long)r * ld[n] + c] : 0.f;
            }
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

template <bool VEC2>
__global__ __launch_bounds__(256) void attention_fwd_kernel(const attn_args a, float* __restrict__ ctx /* (B*L, H*E) */) {
    EEG_LDS_BASE(float, lds);
    float *Qs = lds, *Ks = lds + AT_T, *Vs = lds + 2 * AT_T, *Ps = lds + 3 * AT_T;
    const int lane = threadIdx.x & 63, w = wave_uniform(threadIdx.x >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int HE = a.H * a.E;
    const float* base = a.qkv + (long long)b * AT_L * a.ld + h * a.E;
    {
        float* const dst[3] = {Qs, Ks, Vs};
        const float* const src[3] = {base, base + HE, base + 2 * HE};
        const int lds_[3] = {a.ld, a.ld, a.ld};
        load_heads<3, VEC2>(dst, src, lds_, a.E);
    }
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
        bool keep[4] = {true, true, true, true};
        if (a.drop_p > 0.f) dropout_keep_quad(a.seed, a.site, rbase, fr, a.drop_p, keep);
#pragma unroll
        for (int t = 0; t < 4; ++t) Ps[row * AT_LD + 16 * t + fr] = keep[t] ? s[t][r] * ks : 0.f;
    }
    __syncthreads();                                          // P rows of this wave are complete in LDS (A-operand order on re-read)
    f32x4 o[4];
    zero4(o);
    mma_rows_x_mat(Ps, 16 * w, Vs, o, lane);                  // O[row][e] = sum_j P[row][j] V[j][e]
    store_rows(ctx + (long long)b * AT_L * HE + h * a.E, HE, 16 * w, o, a.E, lane);
}

// Backward keeps FOUR 17 KB tiles (Q, K, V, dO) = 70 KB so two workgroups share a CU: the dropped probabilities replace V once every
// wave has its dP, and dS replaces dO once every wave has its dV (the first version held six tiles, 104 KB, one workgroup per CU).
template <bool VEC2>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const attn_args a, const float* __restrict__ dctx /* (B*L, H*E) */,
                                                             float* __restrict__ dqkv /* (B*L, ld) */) {
    EEG_LDS_BASE(float, lds);
    float *Qs = lds, *Ks = lds + AT_T, *Vs = lds + 2 * AT_T, *Ds = lds + 3 * AT_T;
    float *Ps = Vs, *Ss = Ds;
    const int lane = threadIdx.x & 63, w = wave_uniform(threadIdx.x >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int HE = a.H * a.E;
    const float* qbase = a.qkv + (long long)b * AT_L * a.ld + h * a.E;
    float* dqbase = dqkv + (long long)b * AT_L * a.ld + h * a.E;
    {
        float* const dst[4] = {Qs, Ks, Vs, Ds};
        const float* const src[4] = {qbase, qbase + HE, qbase + 2 * HE, dctx + (long long)b * AT_L * HE + h * a.E};
        const int lds_[4] = {a.ld, a.ld, a.ld, HE};
        load_heads<4, VEC2>(dst, src, lds_, a.E);
    }
    __syncthreads();
    f32x4 p[4], dp[4];
    zero4(p);
    zero4(dp);
    mma_rows_x_rowsT(Qs, 16 * w, Ks, p, lane);                // recompute S, then P
    softmax_rows(p, a.scale);
    mma_rows_x_rowsT(Ds, 16 * w, Vs, dp, lane);               // dP[row][j] = sum_e dO[row][e] V[j][e]
    const float ks = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    float dsv[4][4];                                          // dS rows of this wave, parked in registers until dO is dead
    __syncthreads();                                          // every wave is done with V: its tile becomes Pdrop
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * w + 4 * g + r;
        const unsigned long long rbase = ((unsigned long long)blockIdx.x * AT_L + row) * AT_L;
        bool keep[4] = {true, true, true, true};
        if (a.drop_p > 0.f) dropout_keep_quad(a.seed, a.site, rbase, fr, a.drop_p, keep);
        float dd[4];
        float delta = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            dd[t] = keep[t] ? dp[t][r] * ks : 0.f;            // gradient w.r.t. the un-dropped probabilities
            delta += p[t][r] * dd[t];
        }
        delta = group16_sum(delta);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            Ps[row * AT_LD + 16 * t + fr] = keep[t] ? p[t][r] * ks : 0.f;             // dropped probabilities (feed dV)
            dsv[r][t] = p[t][r] * (dd[t] - delta) * a.scale;                          // dS (scale folded in)
        }
    }
    __syncthreads();                                          // all 64 rows of Pdrop are in LDS
    f32x4 acc[4];
    zero4(acc);
    mma_colsT_x_mat(Ps, 16 * w, Ds, acc, lane);               // dV[j][e] = sum_i Pdrop[i][j] dO[i][e]
    store_rows(dqbase + 2 * HE, a.ld, 16 * w, acc, a.E, lane);
    __syncthreads();                                          // every wave is done with dO: its tile becomes dS
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t) Ss[(16 * w + 4 * g + r) * AT_LD + 16 * t + fr] = dsv[r][t];
    __syncthreads();
    zero4(acc);
    mma_rows_x_mat(Ss, 16 * w, Ks, acc, lane);                // dQ[row][e] = sum_j dS[row][j] K[j][e]
    store_rows(dqbase, a.ld, 16 * w, acc, a.E, lane);
    zero4(acc);
    mma_colsT_x_mat(Ss, 16 * w, Qs, acc, lane);               // dK[j][e] = sum_i dS[i][j] Q[i][e]      (rows j = 16w..)
    store_rows(dqbase + HE, a.ld, 16 * w, acc, a.E, lane);
}

}  // namespace eeg

using namespace eeg;

static bool attn_aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }
static int attn_check(const float* qkv, int B, int L, int H, int E, int ld, float drop_p) {
    if (!qkv || B < 1 || L != AT_L || H < 1 || E < 1 || E > 64 || ld < 3 * H * E || drop_p < 0.f || drop_p >= 1.f) return EEGCLIP_EINVAL;
    return 0;
}

extern "C" int eegclip_attention_fwd(const float* qkv, float* ctx, int B, int L, int H, int E, int ld, float scale, float drop_p,
                                     unsigned long long seed, unsigned site, void* stream) {
    if (int rc = attn_check(qkv, B, L, H, E, ld, drop_p)) return rc;
    if (!ctx) return EEGCLIP_EINVAL;
    attn_args a{qkv, B, H, E, ld, scale, drop_p, seed, site};
    const bool vec2 = (E % 2 == 0) && (ld % 2 == 0) && attn_aligned8(qkv);
    if (vec2) EEG_LAUNCH(attention_fwd_kernel<true>, dim3(B * H), dim3(256), 4 * AT_T * sizeof(float), stream, a, ctx);
    else      EEG_LAUNCH(attention_fwd_kernel<false>, dim3(B * H), dim3(256), 4 * AT_T * sizeof(float), stream, a, ctx);
    return (int)hipGetLastError();
}

extern "C" int eegclip_attention_bwd(const float* qkv, const float* dctx, float* dqkv, int B, int L, int H, int E, int ld, float scale,
                                     float drop_p, unsigned long long seed, unsigned site, void* stream) {
    if (int rc = attn_check(qkv, B, L, H, E, ld, drop_p)) return rc;
    if (!dctx || !dqkv) return EEGCLIP_EINVAL;
    attn_args a{qkv, B, H, E, ld, scale, drop_p, seed, site};
    const bool vec2 = (E % 2 == 0) && (ld % 2 == 0) && attn_aligned8(qkv) && attn_aligned8(dctx);
    if (vec2) EEG_LAUNCH(attention_bwd_kernel<true>, dim3(B * H), dim3(256), 4 * AT_T * sizeof(float), stream, a, dctx, dqkv);
    else      EEG_LAUNCH(attention_bwd_kernel<false>, dim3(B * H), dim3(256), 4 * AT_T * sizeof(float), stream, a, dctx, dqkv);
    return (int)hipGetLastError();
}
