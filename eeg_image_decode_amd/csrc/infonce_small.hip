// CLIP-symmetric InfoNCE (models/loss.py:122-140) of ONE process at the training batch size (n = 256 per GPU): the n x n logits of every target as a
// K-parallel plane GEMM + two small row-block kernels.
//
// Why: at n = 256 the fused tile kernels (csrc/infonce_fused.hip) are 64 workgroups that each walk D = 1024 alone -- 12 + 17 us of the step's critical
// path for 0.5 GFLOP (round 6 trace), and they form every product twice (the block and its swap).  Here the raw logits S_t = A B_t^T of all T targets are
// ONE launch of csrc/head_gemm.hip (M = n, N = T n, K = D split over 8 slices: 256 workgroups of 4 k-tiles, partial slabs), and
//   infonce_small_fwd   adds the slabs (slice order): row workgroups (4 query rows each) keep the raw logits (n x T n fp32 = 0.5 MB), the row
//                       log-sum-exps and the positives; column workgroups of the SAME launch (8 key columns over all rows each) the column log-sum-exps --
//                       both kinds add the slabs themselves (the 4 MB are L2-resident), nothing is exchanged inside the launch;
//   infonce_small_grad  elementwise: adds the loss and d loss / d scale and writes G_t = s dL/dS_t as bf16 hi | lo planes side by side -- the A operand of
//                       the query-gradient GEMM.
// (First version: 32 row workgroups + column partials finished redundantly by every gradient workgroup, runtime-length slab loops: 23 + 33 us -- dependent
// load chains; every element's slab loads are now in flight together.)
// The arithmetic is the reference's: L = sum_t w_t / (2 n) [sum_i (lse_j S_ij - S_ii) + sum_j (lse_i S_ij - S_jj)], S = s A B^T with the RAW scale s;
// G = w_t / (2 n) (softmax_rows + softmax_cols - 2 I) s.
#include "eeg_common.h"

namespace eeg {

constexpr int IS_R = 4;                   // query rows per workgroup of the gradient kernel
constexpr int IS_RF = 2;                  // query rows per row workgroup of the forward kernel (one pass of slab loads, one (row, target) pair per wave at T = 2)
constexpr int IS_C = 8;                   // key columns per column workgroup
constexpr int IS_MAXT = 4;
constexpr int IS_MAXS = 8;                // partial slabs (all their loads of an element are in flight together: no runtime-length dependent-load loop)

struct is_args {
    const float* slabs;                   // [nslabs][n][T n] raw partial logits
    int nslabs;
    long long stride;
    int n, T;
    const float* scale;
    float* S;                             // [n][T n] raw logits (fwd out, grad in)
    float* lse_r;                         // [T][n]  row log-sum-exps
    float* diag;                          // [T][n]  scaled positives
    float* lse_c;                         // [T n]   column log-sum-exps
    float w[IS_MAXT];                     // loss weight of target t (the 1/2 of the symmetric loss NOT included)
    unsigned short *g_hi, *g_lo;          // [n][ldg] planes (grad out)
    long long ldg;
    float* loss;
    float* dscale;
};

// sum of the slabs at float4 offset `off`, slice order; every load issued before the first add
__device__ __forceinline__ f32x4 is_slab_sum(const float* __restrict__ slabs, int nslabs, long long stride, long long off) {
    f32x4 p[IS_MAXS];
#pragma unroll
    for (int sl = 0; sl < IS_MAXS; ++sl) p[sl] = *reinterpret_cast<const f32x4*>(slabs + (long long)(sl < nslabs ? sl : 0) * stride + off);
    f32x4 x = p[0];
#pragma unroll
    for (int sl = 1; sl < IS_MAXS; ++sl)
        if (sl < nslabs) {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] += p[sl][e];
        }
    return x;
}

// forward: workgroups [0, n / IS_RF) own IS_RF query rows each (raw logits out, row log-sum-exps, positives); workgroups behind them own IS_C key columns each
// over ALL rows (column log-sum-exps) -- both kinds add the slabs themselves, nothing is exchanged inside the launch
__global__ __launch_bounds__(256) void infonce_small_fwd_kernel(const is_args a) {
    EEG_LDS_BASE(float, v);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, NC = a.T * a.n, nrb = a.n / IS_RF;
    const float s = *a.scale;
    if ((int)blockIdx.x < nrb) {
        const int r0 = blockIdx.x * IS_RF;                    // v: [IS_RF][T n] scaled logits of this row block
        for (int q = t; q < IS_RF * NC / 4; q += 256) {
            const long long off = (long long)r0 * NC + 4 * q;
            const f32x4 x = is_slab_sum(a.slabs, a.nslabs, a.stride, off);
            *reinterpret_cast<f32x4*>(a.S + off) = x;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * q + e] = s * x[e];
        }
        __syncthreads();
        for (int pr = wave; pr < IS_RF * a.T; pr += 4) {      // one wave per (row, target)
            const int i = pr / a.T, tg = pr - i * a.T;
            const float* row = v + i * NC + tg * a.n;
            float m = -3.0e38f;
            for (int j = lane; j < a.n; j += 64) m = fmaxf(m, row[j]);
            m = wave_max(m);
            float l = 0.f;
            for (int j = lane; j < a.n; j += 64) l += expf(row[j] - m);
            l = wave_sum(l);
            if (lane == 0) {
                a.lse_r[tg * a.n + r0 + i] = m + logf(l);
                a.diag[tg * a.n + r0 + i] = row[r0 + i];
            }
        }
        return;
    }
    // ---- column workgroup: columns c0 .. c0 + 7, thread = row (rows t, t + 256, ..): two passes over registers, cross-wave combine through LDS
    const int c0 = ((int)blockIdx.x - nrb) * IS_C;
    float* red = v;                                          // [2][4][IS_C]
    float x[4][IS_C];                                        // up to 1024 rows
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = t + 256 * k;
        if (r < a.n) {
#pragma unroll
            for (int h = 0; h < IS_C / 4; ++h) {
                const f32x4 y = is_slab_sum(a.slabs, a.nslabs, a.stride, (long long)r * NC + c0 + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[k][4 * h + e] = s * y[e];
            }
        } else {
#pragma unroll
            for (int c = 0; c < IS_C; ++c) x[k][c] = -3.0e38f;
        }
    }
    float m[IS_C], l[IS_C];
#pragma unroll
    for (int c = 0; c < IS_C; ++c) {
        m[c] = wave_max(fmaxf(fmaxf(x[0][c], x[1][c]), fmaxf(x[2][c], x[3][c])));
        if (lane == 0) red[wave * IS_C + c] = m[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < IS_C; ++c) {
        m[c] = fmaxf(fmaxf(red[c], red[IS_C + c]), fmaxf(red[2 * IS_C + c], red[3 * IS_C + c]));
        float e_ = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) e_ += expf(x[k][c] - m[c]);      // (absent rows: exp(-3e38 - m) = 0)
        l[c] = wave_sum(e_);
        if (lane == 0) red[4 * IS_C + wave * IS_C + c] = l[c];
    }
    __syncthreads();
    if (t < IS_C) a.lse_c[c0 + t] = m[t] + logf((red[4 * IS_C + t] + red[5 * IS_C + t]) + (red[6 * IS_C + t] + red[7 * IS_C + t]));
}

// gradient: workgroup = IS_R query rows; elementwise over their T n columns
__global__ __launch_bounds__(256) void infonce_small_grad_kernel(const is_args a) {
    EEG_LDS_BASE(float, red);                                // [8]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, NC = a.T * a.n, r0 = blockIdx.x * IS_R;
    const float s = *a.scale;
    float ds = 0.f, ls = 0.f;
    const float inv2n = 0.5f / (float)a.n;
    // two column quads per thread and pass, their loads in flight together (n = 256, T = 2: the whole row block in one pass)
    constexpr int NQ = 2;
    for (int qb = t; qb < IS_R * NC / 4; qb += 256 * NQ) {
      f32x4 raws[NQ], lcvs[NQ];
      float lrs[NQ];
#pragma unroll
      for (int u = 0; u < NQ; ++u) {
        const int q = qb + 256 * u;
        const bool ok = q < IS_R * NC / 4;
        const int i = ok ? (4 * q) / NC : 0, j0 = ok ? 4 * q - i * NC : 0, tg = j0 / a.n;
        lrs[u] = a.lse_r[tg * a.n + r0 + i];
        raws[u] = *reinterpret_cast<const f32x4*>(a.S + (long long)(r0 + i) * NC + j0);
        lcvs[u] = *reinterpret_cast<const f32x4*>(a.lse_c + j0);
      }
#pragma unroll
      for (int u = 0; u < NQ; ++u) {
        const int q = qb + 256 * u;
        if (q >= IS_R * NC / 4) continue;
        const int i = (4 * q) / NC, j0 = 4 * q - i * NC, tg = j0 / a.n;          // (n % 4 == 0: the 4 columns share a target)
        const float c = a.w[tg] * inv2n, lr = lrs[u];
        const f32x4 raw = raws[u];
        const f32x4 lcv = lcvs[u];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = s * raw[e];
            float g = expf(x - lr) + expf(x - lcv[e]);
            if (j0 + e == tg * a.n + r0 + i) g -= 2.f;
            g *= c;
            ds += g * raw[e];
            o[e] = g * s;
        }
        u32x2_t hi, lo;
        x3_split4(o[0], o[1], o[2], o[3], hi, lo);
        *reinterpret_cast<u32x2_t*>(a.g_hi + (long long)(r0 + i) * a.ldg + j0) = hi;
        *reinterpret_cast<u32x2_t*>(a.g_lo + (long long)(r0 + i) * a.ldg + j0) = lo;
      }
    }
    // loss terms of this block's rows and of the columns that are its rows' positives
    if (t < IS_R * a.T) {
        const int i = t / a.T, tg = t - i * a.T, idx = tg * a.n + r0 + i;
        ls = a.w[tg] * inv2n * ((a.lse_r[idx] - a.diag[idx]) + (a.lse_c[idx] - a.diag[idx]));
    }
    ds = wave_sum(ds);
    ls = wave_sum(ls);
    if (lane == 0) { red[wave] = ds; red[4 + wave] = ls; }
    __syncthreads();
    if (t == 0) {
        atomicAdd(a.dscale, (red[0] + red[1]) + (red[2] + red[3]));
        atomicAdd(a.loss, (red[4] + red[5]) + (red[6] + red[7]));
    }
}

}  // namespace eeg

using namespace eeg;

static int is_check(int n, int T, int nslabs, long long stride) {
    if (n < 8 || n % 8 || n > 1024 || T < 1 || T > IS_MAXT || nslabs < 1 || nslabs > IS_MAXS || (nslabs > 1 && (stride < (long long)n * T * n || (stride & 3)))) return EEGCLIP_EINVAL;
    return 0;
}

extern "C" int eegclip_infonce_small_supported(int n, int T) { return is_check(n, T, 1, 0) == 0 ? 1 : 0; }
extern "C" long long eegclip_infonce_small_workspace_floats(int n, int T) {      // S | lse_r | diag | lse_c
    if (is_check(n, T, 1, 0)) return 0;
    return (long long)n * T * n + 3LL * T * n;
}

static void is_fill(is_args& a, const float* slabs, int nslabs, long long stride, int n, int T, const float* scale, float* ws) {
    a.slabs = slabs; a.nslabs = nslabs; a.stride = stride; a.n = n; a.T = T; a.scale = scale;
    a.S = ws;
    a.lse_r = ws + (long long)n * T * n;
    a.diag = a.lse_r + (long long)T * n;
    a.lse_c = a.diag + (long long)T * n;
}

extern "C" int eegclip_infonce_small_fwd(const float* slabs, int nslabs, long long slab_stride, int n, int T, const float* scale, float* workspace, void* stream) {
    if (!slabs || !scale || !workspace) return EEGCLIP_EINVAL;
    if (const int rc = is_check(n, T, nslabs, slab_stride)) return rc;
    if ((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(workspace)) & 15u) return EEGCLIP_EALIGN;
    is_args a{};
    is_fill(a, slabs, nslabs, slab_stride, n, T, scale, workspace);
    EEG_LAUNCH(infonce_small_fwd_kernel, dim3(n / IS_RF + T * n / IS_C), dim3(256), (size_t)IS_RF * T * n * sizeof(float) + 256, stream, a);
    return (int)hipGetLastError();
}

extern "C" int eegclip_infonce_small_grad(int n, int T, const float* scale, const float* workspace, float w0, float w1, float w2, float w3, void* g_hi, void* g_lo,
                                          long long ldg, float* loss, float* dscale, void* stream) {
    if (!scale || !workspace || !g_hi || !g_lo || !loss || !dscale || ldg < (long long)T * n || (ldg & 3)) return EEGCLIP_EINVAL;
    if (const int rc = is_check(n, T, 1, 0)) return rc;
    if ((reinterpret_cast<uintptr_t>(workspace) & 15u) || ((reinterpret_cast<uintptr_t>(g_hi) | reinterpret_cast<uintptr_t>(g_lo)) & 7u)) return EEGCLIP_EALIGN;
    is_args a{};
    is_fill(a, nullptr, 1, 0, n, T, scale, const_cast<float*>(workspace));
    a.w[0] = w0; a.w[1] = w1; a.w[2] = w2; a.w[3] = w3;
    a.g_hi = static_cast<unsigned short*>(g_hi); a.g_lo = static_cast<unsigned short*>(g_lo); a.ldg = ldg;
    a.loss = loss; a.dscale = dscale;
    EEG_LAUNCH(infonce_small_grad_kernel, dim3(n / IS_R), dim3(256), 8 * sizeof(float), stream, a);
    return (int)hipGetLastError();
}
