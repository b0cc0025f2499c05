// LayerNorm forward/backward (rows of 250 or 1024 floats) and the BatchNorm2d(+ELU, +dropout) pieces of tsconv.
// All of these are HBM-bound streaming kernels: one wavefront per row / coalesced lane-strided access,
// 64-lane xor-butterfly reductions, statistics kept in fp32 (LN) or fp64 atomics (BN batch sums).
#include "eeg_common.h"

#include <stdlib.h>

namespace eeg {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int LN_MAXC = 16;   // columns per lane: cols <= 1024 (NC = 4 instantiation for cols <= 256: the encoder's 250-wide rows
                              // would otherwise drag 12 dead guarded iterations through every loop)

// y = (x - mean) * rstd * gamma + beta ; mean/rstd saved for backward.  One wave per row, grid-stride over rows.
template <int NC>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ y,
                                                             float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                             int rows, int cols, float eps) {
    const int lane = threadIdx.x & 63, wave = wave_uniform(threadIdx.x >> 6);
    const float inv = 1.0f / (float)cols;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float* xr = x + (long long)row * cols;
        float v[NC];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < cols ? xr[c] : 0.f;
            s += v[i];
        }
        const float mean = wave_sum(s) * inv;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            const float dlt = c < cols ? v[i] - mean : 0.f;
            q += dlt * dlt;
        }
        const float rstd = rsqrtf(wave_sum(q) * inv + eps);
        float* yr = y + (long long)row * cols;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            if (c < cols) yr[c] = (v[i] - mean) * rstd * gamma[c] + beta[c];
        }
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
    }
}

// Residual + LayerNorm, the post-LN sublayer tail of the encoder (Transformer_EncDec.py:45-51):  v = resid + dropout(x);  y = LN(v)
// [;  y2 = LN2(y): the layer's norm2 followed directly by the encoder's final norm, :77-78].  The dropout and the residual used to be
// the epilogue of the GEMM that produces x; there every output element costs a whole Philox block (a lane's 16 accumulator values
// sit in 16 different blocks: +12 us on a 32 us GEMM), here a lane owns 4 consecutive columns and one block serves all four.
// Same mask (Philox(seed, site, row * cols + c)) and the same arithmetic order, so results are bit-identical to the epilogue form.
typedef float ln_f32x4u __attribute__((ext_vector_type(4), aligned(4)));       // 16-byte global access from a dword-aligned address
// the row operand of a LayerNorm launch as the partial slabs of a K-parallel GEMM (csrc/head_gemm.hip): value = sum_{s < n} p[s * stride + i] (+ bias[col]),
// added in slice order while the row is loaded -- the GEMM's split costs no launch of its own.  n <= 1: the plain operand.
struct ln_slabs {
    int n;
    long long stride;
    const float* bias;
};
template <int NG, bool VEC2, bool DOUBLE>
__global__ __launch_bounds__(256) void residual_layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ resid,
                                                                      float* __restrict__ x_out, float drop_p, unsigned long long seed, unsigned site,
                                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                      float* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                                      const float* __restrict__ gamma2, const float* __restrict__ beta2,
                                                                      float* __restrict__ y2, float* __restrict__ mean2_out, float* __restrict__ rstd2_out,
                                                                      int rows, int cols, float eps, unsigned short* __restrict__ y_hi,
                                                                      unsigned short* __restrict__ y_lo, const ln_slabs xs) {
    const int lane = threadIdx.x & 63, wave = wave_uniform(threadIdx.x >> 6);
    const float inv = 1.0f / (float)cols;
    const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    auto ld4 = [&](const float* p, int c, float (&v)[4]) {       // 4 consecutive columns starting at c (c % 4 == 0), zero past `cols`
        if (VEC2) {
            if (c + 3 < cols) {                                   // one 16-byte access (dword alignment is enough for global_load_dwordx4)
                const ln_f32x4u t = *reinterpret_cast<const ln_f32x4u*>(p + c);
                v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const bool ok = c + 2 * h < cols;
                    const f32x2 t = ok ? *reinterpret_cast<const f32x2*>(p + c + 2 * h) : f32x2{0.f, 0.f};
                    v[2 * h] = t[0];
                    v[2 * h + 1] = t[1];
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = c + e < cols ? p[c + e] : 0.f;
        }
    };
    auto st4 = [&](float* p, int c, const float (&v)[4]) {
        if (VEC2) {
            if (c + 3 < cols) *reinterpret_cast<ln_f32x4u*>(p + c) = ln_f32x4u{v[0], v[1], v[2], v[3]};
            else {
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    if (c + 2 * h < cols) *reinterpret_cast<f32x2*>(p + c + 2 * h) = f32x2{v[2 * h], v[2 * h + 1]};
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < cols) p[c + e] = v[e];
        }
    };
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const unsigned long long rbase = (unsigned long long)row * cols;
        const unsigned off = (unsigned)(rbase & 3ull);            // wave-uniform misalignment of this row against the Philox blocks
        float v[NG][4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int c = 256 * i + 4 * lane;
            ld4(x + rbase, c, v[i]);
            for (int sl = 1; sl < xs.n; ++sl) {
                float pv[4];
                ld4(x + (long long)sl * xs.stride + rbase, c, pv);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i][e] += pv[e];
            }
            if (xs.bias) {
                float pv[4];
                ld4(xs.bias, c, pv);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i][e] += pv[e];
            }
            if (resid) {
                float rv[4];
                ld4(resid + rbase, c, rv);
                bool keep[4] = {true, true, true, true};
                if (drop_p > 0.f && c < cols) {
                    bool k0[4], k1[4];
                    dropout_keep4(seed, site, rbase + c - off, drop_p, k0);
                    if (off) dropout_keep4(seed, site, rbase + c - off + 4, drop_p, k1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) keep[e] = (off + e < 4) ? k0[(off + e) & 3] : k1[(off + e) & 3];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float d = v[i][e];
                    if (drop_p > 0.f) d = keep[e] ? d * keep_scale : 0.f;
                    v[i][e] = c + e < cols ? d + rv[e] : 0.f;
                }
                if (x_out) st4(x_out + rbase, c, v[i]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) s += v[i][e];
        }
        const float mean = wave_sum(s) * inv;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NG; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dlt = 256 * i + 4 * lane + e < cols ? v[i][e] - mean : 0.f;
                q += dlt * dlt;
            }
        const float rstd = rsqrtf(wave_sum(q) * inv + eps);
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int c = 256 * i + 4 * lane;
            float gv[4], bv[4];
            ld4(gamma, c, gv);
            ld4(beta, c, bv);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[i][e] = c + e < cols ? (v[i][e] - mean) * rstd * gv[e] + bv[e] : 0.f;
                s2 += v[i][e];
            }
            st4(y + rbase, c, v[i]);
            if (y_hi && c < cols) {                               // y again as bf16 hi | lo planes (cols % 4 == 0: checked by the host): the next contraction's operand
                u32x2_t hi, lo;
                x3_split4(v[i][0], v[i][1], v[i][2], v[i][3], hi, lo);
                *reinterpret_cast<u32x2_t*>(y_hi + rbase + c) = hi;
                *reinterpret_cast<u32x2_t*>(y_lo + rbase + c) = lo;
            }
        }
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
        if (DOUBLE) {
            const float m2 = wave_sum(s2) * inv;
            float q2 = 0.f;
#pragma unroll
            for (int i = 0; i < NG; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dlt = 256 * i + 4 * lane + e < cols ? v[i][e] - m2 : 0.f;
                    q2 += dlt * dlt;
                }
            const float r2 = rsqrtf(wave_sum(q2) * inv + eps);
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const int c = 256 * i + 4 * lane;
                float gv[4], bv[4], o[4];
                ld4(gamma2, c, gv);
                ld4(beta2, c, bv);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - m2) * r2 * gv[e] + bv[e];
                st4(y2 + rbase, c, o);
            }
            if (lane == 0) {
                if (mean2_out) mean2_out[row] = m2;
                if (rstd2_out) rstd2_out[row] = r2;
            }
        }
    }
}

// ---- rows of exactly 1024 columns, FEW of them (the projection head's LayerNorm at M = the batch, ATMS_retrieval.py:166): ONE 256-thread workgroup per row,
// lane t owns columns 4 t .. 4 t + 3 (one 16-byte access per operand, one Philox block), row sums through one LDS exchange.  The wave-per-row kernels above put
// 256 rows on 64 workgroups and walk each row in four dependent 256-column groups: 15 us for 1 MB at B = 256 (round 6 trace); this form is 256 workgroups.
__device__ __forceinline__ void ln_block_sum2(float& a, float& b, float* red) {       // sums of a and b over the 256 threads (4 waves); red: 8 floats of LDS
    a = wave_sum(a);
    b = wave_sum(b);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();                                             // (red may still be read from the previous exchange)
    if (lane == 0) { red[w] = a; red[4 + w] = b; }
    __syncthreads();
    a = (red[0] + red[1]) + (red[2] + red[3]);
    b = (red[4] + red[5]) + (red[6] + red[7]);
}
__global__ __launch_bounds__(256) void head_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ resid, float* __restrict__ x_out, float drop_p,
                                                           unsigned long long seed, unsigned site, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out, float eps,
                                                           unsigned short* __restrict__ y_hi, unsigned short* __restrict__ y_lo, const ln_slabs xs) {
    EEG_LDS_BASE(float, red);
    constexpr int COLS = 1024;
    const int row = blockIdx.x, c = 4 * threadIdx.x;
    const long long i0 = (long long)row * COLS + c;
    // every load of the row is issued here (the affine parameters were fetched behind the two block reductions: one more L2 round trip each)
    const f32x4 zero4{0.f, 0.f, 0.f, 0.f};
    const f32x4 gv = *reinterpret_cast<const f32x4*>(gamma + c), bv = *reinterpret_cast<const f32x4*>(beta + c);
    const f32x4 rv = resid ? *reinterpret_cast<const f32x4*>(resid + i0) : zero4;
    const f32x4 pb = xs.bias ? *reinterpret_cast<const f32x4*>(xs.bias + c) : zero4;
    f32x4 v = slab_sum4_inflight(x, xs.n < 1 ? 1 : xs.n, xs.stride, i0);
    if (xs.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += pb[e];
    }
    if (resid) {
        if (drop_p > 0.f) {
            bool keep[4];
            dropout_keep4(seed, site, (unsigned long long)i0, drop_p, keep);
            const float ks = 1.0f / (1.0f - drop_p);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = keep[e] ? v[e] * ks : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += rv[e];
        if (x_out) *reinterpret_cast<f32x4*>(x_out + i0) = v;
    }
    float s = (v[0] + v[1]) + (v[2] + v[3]), dummy = 0.f;
    ln_block_sum2(s, dummy, red);
    const float mean = s * (1.0f / COLS);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) q += (v[e] - mean) * (v[e] - mean);
    ln_block_sum2(q, dummy, red);
    const float rstd = rsqrtf(q * (1.0f / COLS) + eps);
    
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (v[e] - mean) * rstd * gv[e] + bv[e];
    *reinterpret_cast<f32x4*>(y + i0) = v;
    if (y_hi) {
        u32x2_t hi, lo;
        x3_split4(v[0], v[1], v[2], v[3], hi, lo);
        *reinterpret_cast<u32x2_t*>(y_hi + i0) = hi;
        *reinterpret_cast<u32x2_t*>(y_lo + i0) = lo;
    }
    if (threadIdx.x == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
}
__global__ __launch_bounds__(256) void head_ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx,
                                                           int accumulate_dx, float* __restrict__ dx_drop, float drop_p, unsigned long long seed, unsigned site,
                                                           const ln_slabs dys, unsigned short* __restrict__ dd_hi, unsigned short* __restrict__ dd_lo,
                                                           float* __restrict__ dy_sum) {
    EEG_LDS_BASE(float, red);
    constexpr int COLS = 1024;
    const int row = blockIdx.x, c = 4 * threadIdx.x;
    const long long i0 = (long long)row * COLS + c;
    // every load of the row before its first store (vmcnt retires in order and counts stores: a load behind a store waits for the store's round trip)
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + i0), gv = *reinterpret_cast<const f32x4*>(gamma + c);
    const float mu = mean[row], rs = rstd[row];
    const f32x4 dx_old = accumulate_dx ? *reinterpret_cast<const f32x4*>(dx + i0) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 dv = slab_sum4_inflight(dy, dys.n < 1 ? 1 : dys.n, dys.stride, i0);
    if (dy_sum) *reinterpret_cast<f32x4*>(dy_sum + i0) = dv;
    float xh[4], g[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        xh[e] = (xv[e] - mu) * rs;
        g[e] = dv[e] * gv[e];
        s1 += g[e];
        s2 += g[e] * xh[e];
    }
    ln_block_sum2(s1, s2, red);
    const float c1 = s1 * (1.0f / COLS), c2 = s2 * (1.0f / COLS);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = rs * (g[e] - c1 - xh[e] * c2);
    if (accumulate_dx) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += dx_old[e];
    }
    *reinterpret_cast<f32x4*>(dx + i0) = v;
    if (dx_drop) {
        f32x4 dd = v;
        if (drop_p > 0.f) {
            bool keep[4];
            dropout_keep4(seed, site, (unsigned long long)i0, drop_p, keep);
            const float ks = 1.0f / (1.0f - drop_p);
#pragma unroll
            for (int e = 0; e < 4; ++e) dd[e] = keep[e] ? v[e] * ks : 0.f;
        }
        *reinterpret_cast<f32x4*>(dx_drop + i0) = dd;
        if (dd_hi) {
            u32x2_t hi, lo;
            x3_split4(dd[0], dd[1], dd[2], dd[3], hi, lo);
            *reinterpret_cast<u32x2_t*>(dd_hi + i0) = hi;
            *reinterpret_cast<u32x2_t*>(dd_lo + i0) = lo;
        }
    }
}
constexpr int HEAD_LN_MAX_ROWS = 4096;          // beyond that the wave-per-row kernels fill the chip on their own

// backward, part 1:  dx (+)= rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma   [+ dx_drop = dropout'(dx)]
// One wave per row, one row per wave.  Lane l owns the 4 CONSECUTIVE columns 256 i + 4 l .. + 3 of every 256-column group: 8-byte
// accesses, and the dropout mask of a lane's 4 elements comes from one Philox block (two when row * cols is not a multiple of 4 --
// a wave-uniform case) instead of four: with a lane-strided column map the 4.1 M Philox evaluations of a (16384, 250) tensor, not
// HBM, set this kernel's time.  A fused variant that also accumulated dgamma / dbeta (LDS combine + one atomic per column and
// workgroup) measured slower than this + the column-parallel kernel below: 1024 same-address atomics per parameter.
// PART: the kernel also leaves this workgroup's share of the parameter gradients (sum_rows dy * xhat | sum_rows dy, per column) as one partial row
// in `partials` -- the operands are in registers anyway; layernorm_bwd_param_stage2_kernel sums the rows.  (With atomics instead of partial rows
// this fusion lost; see the note above.)
template <int NG, bool VEC2, bool PART = false>
__global__ __launch_bounds__(256) void layernorm_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, float* __restrict__ dx, int rows, int cols,
                                                                int accumulate_dx, float* __restrict__ dx_drop, float drop_p,
                                                                unsigned long long seed, unsigned site, float* __restrict__ partials,
                                                                const ln_slabs dys, unsigned short* __restrict__ dd_hi, unsigned short* __restrict__ dd_lo,
                                                                float* __restrict__ dy_sum) {
    float pgam[PART ? NG : 1][4], pbet[PART ? NG : 1][4];
#pragma unroll
    for (int i = 0; i < (PART ? NG : 1); ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { pgam[i][e] = 0.f; pbet[i][e] = 0.f; }
    const int lane = threadIdx.x & 63, wave = wave_uniform(threadIdx.x >> 6);
    const float inv = 1.0f / (float)cols;
    const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    auto ld4 = [&](const float* p, int c, float (&v)[4]) {       // 4 consecutive columns starting at c (c % 4 == 0), zero past `cols`
        if (VEC2) {
            if (c + 3 < cols) {                                   // one 16-byte access (dword alignment is enough for global_load_dwordx4)
                const ln_f32x4u t = *reinterpret_cast<const ln_f32x4u*>(p + c);
                v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const bool ok = c + 2 * h < cols;             // cols even: a pair is all in or all out
                    const f32x2 t = ok ? *reinterpret_cast<const f32x2*>(p + c + 2 * h) : f32x2{0.f, 0.f};
                    v[2 * h] = t[0];
                    v[2 * h + 1] = t[1];
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = c + e < cols ? p[c + e] : 0.f;
        }
    };
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float* xr = x + (long long)row * cols;
        const float* dr = dy + (long long)row * cols;
        const float mu = mean[row], rs = rstd[row];
        float xh[NG][4], g[NG][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int c = 256 * i + 4 * lane;
            float xv[4], dv[4], gv[4];
            ld4(xr, c, xv);
            ld4(dr, c, dv);
            for (int sl = 1; sl < dys.n; ++sl) {
                float pv[4];
                ld4(dr + (long long)sl * dys.stride, c, pv);
#pragma unroll
                for (int e = 0; e < 4; ++e) dv[e] += pv[e];
            }
            if (dy_sum) {                                         // the summed upstream gradient, for the parameter-gradient launch (cols % 4 == 0 with VEC2)
                if (VEC2 && c + 3 < cols) *reinterpret_cast<ln_f32x4u*>(dy_sum + (long long)row * cols + c) = ln_f32x4u{dv[0], dv[1], dv[2], dv[3]};
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e < cols) dy_sum[(long long)row * cols + c + e] = dv[e];
                }
            }
            ld4(gamma, c, gv);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[i][e] = c + e < cols ? (xv[e] - mu) * rs : 0.f;
                g[i][e] = dv[e] * gv[e];
                s1 += g[i][e];
                s2 += g[i][e] * xh[i][e];
                if (PART) { pgam[i][e] += dv[e] * xh[i][e]; pbet[i][e] += dv[e]; }
            }
        }
        const float c1 = wave_sum(s1) * inv, c2 = wave_sum(s2) * inv;
        float* dxr = dx + (long long)row * cols;
        const unsigned long long rbase = (unsigned long long)row * cols;
        const unsigned off = (unsigned)(rbase & 3ull);            // wave-uniform misalignment of this row against the Philox blocks
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int c = 256 * i + 4 * lane;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = rs * (g[i][e] - c1 - xh[i][e] * c2);
                if (accumulate_dx && c + e < cols) v[e] += dxr[c + e];
            }
            bool keep[4] = {true, true, true, true};
            if (dx_drop && drop_p > 0.f && c < cols) {
                const unsigned long long i0 = rbase + c;          // element index of v[0]; block-aligned start is i0 - off
                bool k0[4], k1[4];
                dropout_keep4(seed, site, i0 - off, drop_p, k0);
                if (off) dropout_keep4(seed, site, i0 - off + 4, drop_p, k1);
#pragma unroll
                for (int e = 0; e < 4; ++e) keep[e] = (off + e < 4) ? k0[(off + e) & 3] : k1[(off + e) & 3];
            }
            if (VEC2 && c + 3 < cols) {
                *reinterpret_cast<ln_f32x4u*>(dxr + c) = ln_f32x4u{v[0], v[1], v[2], v[3]};
                if (dx_drop) {
                    const ln_f32x4u dd = ln_f32x4u{keep[0] ? v[0] * keep_scale : 0.f, keep[1] ? v[1] * keep_scale : 0.f, keep[2] ? v[2] * keep_scale : 0.f,
                                                   keep[3] ? v[3] * keep_scale : 0.f};
                    *reinterpret_cast<ln_f32x4u*>(dx_drop + rbase + c) = dd;
                    if (dd_hi) {                                  // ... again as bf16 hi | lo planes (cols % 4 == 0: checked by the host): the next GEMM's operand
                        u32x2_t hi, lo;
                        x3_split4(dd[0], dd[1], dd[2], dd[3], hi, lo);
                        *reinterpret_cast<u32x2_t*>(dd_hi + rbase + c) = hi;
                        *reinterpret_cast<u32x2_t*>(dd_lo + rbase + c) = lo;
                    }
                }
            } else if (VEC2) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (c + 2 * h < cols) {
                        *reinterpret_cast<f32x2*>(dxr + c + 2 * h) = f32x2{v[2 * h], v[2 * h + 1]};
                        if (dx_drop)
                            *reinterpret_cast<f32x2*>(dx_drop + rbase + c + 2 * h) =
                                f32x2{keep[2 * h] ? v[2 * h] * keep_scale : 0.f, keep[2 * h + 1] ? v[2 * h + 1] * keep_scale : 0.f};
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (c + e < cols) {
                        dxr[c + e] = v[e];
                        if (dx_drop) dx_drop[rbase + c + e] = keep[e] ? v[e] * keep_scale : 0.f;
                    }
                }
            }
        }
    }
    if (PART) {
        // cross-wave combine through LDS (dynamic: [3][2][NG * 256] floats), wave 0 writes the partial row [dgamma | dbeta]
        EEG_LDS_BASE(float, red);
        constexpr int W = NG * 256;
        if (wave > 0) {
#pragma unroll
            for (int i = 0; i < NG; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    red[((wave - 1) * 2 + 0) * W + 256 * i + 4 * lane + e] = pgam[i][e];
                    red[((wave - 1) * 2 + 1) * W + 256 * i + 4 * lane + e] = pbet[i][e];
                }
        }
        __syncthreads();
        if (wave == 0) {
            float* out = partials + (long long)blockIdx.x * 2 * cols;
#pragma unroll
            for (int i = 0; i < NG; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 256 * i + 4 * lane + e;
                    if (c < cols) {
                        float gsum = pgam[i][e], bsum = pbet[i][e];
#pragma unroll
                        for (int w = 0; w < 3; ++w) { gsum += red[(w * 2 + 0) * W + c]; bsum += red[(w * 2 + 1) * W + c]; }
                        out[c] = gsum;
                        out[cols + c] = bsum;
                    }
                }
        }
    }
}

// backward, part 2:  dgamma[c] += sum_rows dy*xhat ; dbeta[c] += sum_rows dy.   Column-parallel: block = 64 columns x 4 row groups,
// lanes walk columns (coalesced), each thread strides over its share of the rows; LDS combine, one atomic per column per block.
// The kernel is bound by its ATOMICS, not by the 33.6 MB it reads: time is proportional to the number of workgroups at any access width
// (16-byte row-wise variant, 16384 x 250: 128 workgroups 27.6 us, 256 -> 33, 512 -> 55, 1024 -> 103 us = ~5 float atomics per ns device-wide;
// this version issues 131 K atomics: 19 us).  It runs on the backward's second stream, off the dX chain.
__global__ __launch_bounds__(256) void layernorm_bwd_param_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int cols) {
    EEG_LDS_BASE(float, red);   // [2][4][64]
    const int lane = threadIdx.x & 63, g = wave_uniform(threadIdx.x >> 6);
    const int c = blockIdx.y * 64 + lane;
    float pg = 0.f, pb = 0.f;
    if (c < cols) {
#pragma unroll 8
        for (int r = blockIdx.x * 4 + g; r < rows; r += gridDim.x * 4) {
            const float d = dy[(long long)r * cols + c];
            pg += d * (x[(long long)r * cols + c] - mean[r]) * rstd[r];
            pb += d;
        }
    }
    red[g * 64 + lane] = pg;
    red[256 + g * 64 + lane] = pb;
    __syncthreads();
    if (g == 0 && c < cols) {
        atomicAdd(dgamma + c, (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]));
        atomicAdd(dbeta + c, (red[256 + lane] + red[320 + lane]) + (red[384 + lane] + red[448 + lane]));
    }
}

// Parameter gradients WITHOUT the atomics: stage 1 streams dy and x once (a wave per row, 16-byte accesses, lane owns 4 consecutive columns of
// each 256-column group; LNP_ROWS rows per wave: 16384 rows -> 1024 workgroups, every row's loads independent) and writes one partial
// [dgamma | dbeta] row per workgroup with plain stores; stage 2 sums the partial rows column-wise (a thread per column and slice of partials,
// LDS combine) and adds into dgamma / dbeta -- 16 partial sums per address instead of 256..1024 contended atomics.
constexpr int LNP_ROWS = 4;
template <int NG>
__global__ __launch_bounds__(256) void layernorm_bwd_param_stage1_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                          float* __restrict__ partials, int rows, int cols) {
    EEG_LDS_BASE(float, red);   // [3 waves][2][NG * 256]
    constexpr int W = NG * 256;
    const int lane = threadIdx.x & 63, wave = wave_uniform(threadIdx.x >> 6);
    float pg[NG][4], pb[NG][4];
#pragma unroll
    for (int i = 0; i < NG; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { pg[i][e] = 0.f; pb[i][e] = 0.f; }
    const int r0 = ((int)blockIdx.x * 4 + wave) * LNP_ROWS;
#pragma unroll
    for (int k = 0; k < LNP_ROWS; ++k) {
        const int r = r0 + k;
        if (r >= rows) break;
        const float mu = mean[r], rs = rstd[r];
        const float* xr = x + (long long)r * cols;
        const float* dr = dy + (long long)r * cols;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int c = 256 * i + 4 * lane;
            if (c >= cols) continue;
            float xv[4], dv[4];
            if (c + 3 < cols) {
                const ln_f32x4u a = *reinterpret_cast<const ln_f32x4u*>(xr + c), b = *reinterpret_cast<const ln_f32x4u*>(dr + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) { xv[e] = a[e]; dv[e] = b[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { xv[e] = c + e < cols ? xr[c + e] : 0.f; dv[e] = c + e < cols ? dr[c + e] : 0.f; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pg[i][e] += dv[e] * (xv[e] - mu) * rs;
                pb[i][e] += dv[e];
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < NG; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                red[((wave - 1) * 2 + 0) * W + 256 * i + 4 * lane + e] = pg[i][e];
                red[((wave - 1) * 2 + 1) * W + 256 * i + 4 * lane + e] = pb[i][e];
            }
    }
    __syncthreads();
    if (wave == 0) {
        float* out = partials + (long long)blockIdx.x * 2 * cols;
#pragma unroll
        for (int i = 0; i < NG; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 256 * i + 4 * lane + e;
                if (c < cols) {
                    float g = pg[i][e], bsum = pb[i][e];
#pragma unroll
                    for (int w = 0; w < 3; ++w) { g += red[(w * 2 + 0) * W + c]; bsum += red[(w * 2 + 1) * W + c]; }
                    out[c] = g;
                    out[cols + c] = bsum;
                }
            }
    }
}

// stage 2: grid (column blocks of 64 over the 2*cols partial columns, LNP_SLICES slices of the partial rows); 4 waves of a workgroup take every
// 4th row of the slice; one atomic per column and slice (LNP_SLICES per address)
constexpr int LNP_SLICES = 16;
__global__ __launch_bounds__(256) void layernorm_bwd_param_stage2_kernel(const float* __restrict__ partials, int nparts, int cols,
                                                                          float* __restrict__ dgamma, float* __restrict__ dbeta) {
    EEG_LDS_BASE(float, red);   // [4][64]
    const int lane = threadIdx.x & 63, g = wave_uniform(threadIdx.x >> 6);
    const int c = blockIdx.x * 64 + lane;                    // column of the [dgamma | dbeta] partial row
    float s = 0.f;
    if (c < 2 * cols) {
        const int per = (nparts + LNP_SLICES - 1) / LNP_SLICES;
        const int p0 = blockIdx.y * per, p1 = p0 + per < nparts ? p0 + per : nparts;
#pragma unroll 8
        for (int p = p0 + g; p < p1; p += 4) s += partials[(long long)p * 2 * cols + c];
    }
    red[g * 64 + lane] = s;
    __syncthreads();
    if (g == 0 && c < 2 * cols) {
        const float t = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
        if (c < cols) atomicAdd(dgamma + c, t);
        else          atomicAdd(dbeta + c - cols, t);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// BatchNorm2d over a (outer, C, inner) view: channel c owns elements x[o][c][i].
// sums[c] = sum x, sums[C + c] = sum x^2   (fp64 atomics: 5.8e5 terms per channel at B=256)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int outer, int C, int inner,
                                                        double* __restrict__ sums) {
    // grid: (chunks, C).  each block reduces a strided set of (o) slabs of channel blockIdx.y
    const int c = blockIdx.y;
    double s = 0.0, q = 0.0;
    const long long per = (long long)inner;
    for (int o = blockIdx.x; o < outer; o += gridDim.x) {
        const float* p = x + ((long long)o * C + c) * per;
        for (int i = threadIdx.x; i < inner; i += blockDim.x) {
            const float v = p[i];
            s += v;
            q += (double)v * v;
        }
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(sums + c, s);
        atomicAdd(sums + C + c, q);
    }
}

// train: mean/rstd from batch sums (biased variance), running stats updated with the unbiased variance
// (torch BatchNorm semantics, momentum 0.1).  eval: mean/rstd from the running stats.
__global__ void bn_finalize_kernel(const double* __restrict__ sums, double count, float eps, float momentum, int C,
                                   float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, int train, long long* __restrict__ num_batches_tracked) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (train && c == 0 && num_batches_tracked) *num_batches_tracked += 1;     // nn.BatchNorm2d's step counter, kept on the device
    if (train) {
        const double m = sums[c] / count;
        double var = sums[C + c] / count - m * m;
        if (var < 0.0) var = 0.0;
        mean[c] = (float)m;
        rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
        }
    } else {
        mean[c] = running_mean[c];
        rstd[c] = 1.0f / sqrtf(running_var[c] + eps);
    }
}

// BatchNorm finalize from per-workgroup partial rows [sum(C) | sumsq(C)] (fp64) in ONE launch: the rows are summed in a fixed order (thread (slice, col)
// sums rows slice, slice + S, ...; the S slices are added in slice order) -- no atomics, nothing to clear -- then mean / rstd / running statistics as
// bn_finalize_kernel.  sums_out (optional): the 2C column sums (the operand of a data-parallel all-reduce); mean == NULL: only those.
__global__ __launch_bounds__(512) void bn_finalize_rows_kernel(const double* __restrict__ rows, int nrows, double count, float eps, float momentum, int C,
                                                               float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ running_mean,
                                                               float* __restrict__ running_var, long long* __restrict__ num_batches_tracked,
                                                               double* __restrict__ sums_out) {
    EEG_LDS_BASE(double, scr);                 // [slices][2C]
    const int t = threadIdx.x, nsl = 512 / (2 * C);
    if (t < nsl * 2 * C) {
        const int sl = t / (2 * C), col = t % (2 * C);
        double s = 0.0;
#pragma unroll 8
        for (int r = sl; r < nrows; r += nsl) s += rows[(long long)r * 2 * C + col];
        scr[t] = s;
    }
    __syncthreads();
    if (t < 2 * C) {
        double s = 0.0;
        for (int sl = 0; sl < nsl; ++sl) s += scr[sl * 2 * C + t];
        scr[t] = s;                            // (slice 0's slot: read only by this thread above)
        if (sums_out) sums_out[t] = s;
    }
    __syncthreads();
    if (mean && t < C) {
        if (t == 0 && num_batches_tracked) *num_batches_tracked += 1;
        const double m = scr[t] / count;
        double var = scr[C + t] / count - m * m;
        if (var < 0.0) var = 0.0;
        mean[t] = (float)m;
        rstd[t] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
            running_mean[t] = (1.f - momentum) * running_mean[t] + momentum * (float)m;
            running_var[t] = (1.f - momentum) * running_var[t] + momentum * (float)unb;
        }
    }
}

// y = dropout( ELU( gamma * (x - mean) * rstd + beta ) )   over the (outer, C, inner) view, flat grid-stride
__global__ __launch_bounds__(256) void bn_elu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ y, long long n,
                                                          int C, int inner, float drop_p, unsigned long long seed,
                                                          unsigned site) {
    const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)((i / inner) % C);
        float v = elu1(gamma[c] * (x[i] - mean[c]) * rstd[c] + beta[c]);
        if (drop_p > 0.f) v = dropout_keep(seed, site, (unsigned long long)i, drop_p) ? v * ks : 0.f;
        y[i] = v;
    }
}

// backward, pass 1:  da = dz * mask/(1-p) * ELU'(bn(x)) ; sums[c] += da ; sums[C+c] += da * xhat
__global__ __launch_bounds__(256) void bn_elu_bwd_stats_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                int outer, int C, int inner, float drop_p,
                                                                unsigned long long seed, unsigned site,
                                                                double* __restrict__ sums) {
    const int c = blockIdx.y;
    const float mu = mean[c], rs = rstd[c], g = gamma[c], b = beta[c];
    const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    double s = 0.0, q = 0.0;
    // the channel's outer * inner elements as one flat index space: every thread busy also when inner is tiny (BatchNorm #2: 36)
    const long long per = (long long)outer * inner;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < per; j += (long long)gridDim.x * blockDim.x) {
        const long long idx = ((j / inner) * C + c) * inner + j % inner;
        const float xh = (x[idx] - mu) * rs;
        const float u = g * xh + b;
        float d = dz[idx];
        if (drop_p > 0.f) d = dropout_keep(seed, site, (unsigned long long)idx, drop_p) ? d * ks : 0.f;
        const float da = u > 0.f ? d : d * expf(u);
        s += da;
        q += (double)da * xh;
    }
    EEG_LDS_BASE(double, red);   // [2][4]: one fp64 atomic pair per workgroup, not per wave
    s = wave_sum(s);
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; red[4 + (threadIdx.x >> 6)] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        double ts = 0.0, tq = 0.0;
        for (int k = 0; k < nw; ++k) { ts += red[k]; tq += red[4 + k]; }
        atomicAdd(sums + c, ts);
        atomicAdd(sums + C + c, tq);
    }
}

// backward, pass 2:  dx = gamma * rstd * (da - sum_da/n - xhat * sum_da_xhat/n) ; dgamma += sum_da_xhat ; dbeta += sum_da
__global__ __launch_bounds__(256) void bn_elu_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const double* __restrict__ sums, const double* __restrict__ sums_param,
                                                                double count, float* __restrict__ dx,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta, long long n,
                                                                int C, int inner, float drop_p, unsigned long long seed,
                                                                unsigned site) {
    const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gtid < C) {      // parameter gradients come from `sums_param` (this rank's own sums under SyncBN; == sums otherwise)
        atomicAdd(dgamma + gtid, (float)sums_param[C + gtid]);
        atomicAdd(dbeta + gtid, (float)sums_param[gtid]);
    }
    for (long long i = gtid; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)((i / inner) % C);
        const float rs = rstd[c], g = gamma[c];
        const float xh = (x[i] - mean[c]) * rs;
        const float u = g * xh + beta[c];
        float d = dz[i];
        if (drop_p > 0.f) d = dropout_keep(seed, site, (unsigned long long)i, drop_p) ? d * ks : 0.f;
        const float da = u > 0.f ? d : d * expf(u);
        const float m1 = (float)(sums[c] / count), m2 = (float)(sums[C + c] / count);
        dx[i] = g * rs * (da - m1 - xh * m2);
    }
}

// bn_elu_bwd_apply_kernel whose batch sums come as `nrows` fp64 partial rows (`ld` doubles apart, the 2 C sums at column `col0`: what eegclip_proj1x1_bwd_rows
// leaves per sample).  A workgroup = (channel c, block of 32 samples): it needs only TWO columns of the table -- sum_da[c] and sum_da_xhat[c], one pair of
// loads per thread and one exchange, the same fixed order in every workgroup of a channel: no atomics, no cleared accumulator, no reduction launch on the dX
// chain.  (First version: flat elementwise grid, every workgroup summed all 2 C columns -- 1440 workgroups x 164 KB of table reads = 20 us; 128 fat workgroups
// serialised their element loop instead: 19 us.)  Workgroup (c, 0) adds dgamma[c] / dbeta[c].
__global__ __launch_bounds__(256) void bn_elu_bwd_apply_rows_kernel(const float* __restrict__ dz, const float* __restrict__ x, const float* __restrict__ mean,
                                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, const double* __restrict__ rows, int nrows, long long ld,
                                                                     int col0, double count, float* __restrict__ dx, float* __restrict__ dgamma,
                                                                     float* __restrict__ dbeta, int outer, int C, int inner, float drop_p,
                                                                     unsigned long long seed, unsigned site) {
    EEG_LDS_BASE(double, scr);                               // [4][2] wave sums
    const int t = threadIdx.x, c = blockIdx.x, lane = t & 63, w = t >> 6;
    const int b0 = blockIdx.y * 32, nb = outer - b0 < 32 ? outer - b0 : 32;
    // the elements of this workgroup in batches of NB per thread, all loads of a batch in flight (a plain runtime-length loop waits for one L2 round trip per
    // element); the FIRST batch -- all of it at inner = 36 -- is issued before the table is summed: it does not depend on the sums
    constexpr int NB = 6;
    float xv[NB], dv[NB];
    long long idx[NB];
    auto load = [&](int q0) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int q = q0 + 256 * j;
            const bool ok = q < nb * inner;
            idx[j] = ok ? ((long long)(b0 + q / inner) * C + c) * inner + q % inner : -1;
            xv[j] = ok ? x[idx[j]] : 0.f;
            dv[j] = ok ? dz[idx[j]] : 0.f;
        }
    };
    load(t);
    double s0 = 0.0, s1 = 0.0;
    for (int r = t; r < nrows; r += 256) {
        s0 += rows[(long long)r * ld + col0 + c];
        s1 += rows[(long long)r * ld + col0 + C + c];
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    if (lane == 0) { scr[2 * w] = s0; scr[2 * w + 1] = s1; }
    __syncthreads();
    const double sum_da = (scr[0] + scr[2]) + (scr[4] + scr[6]), sum_dax = (scr[1] + scr[3]) + (scr[5] + scr[7]);
    if (blockIdx.y == 0 && t == 0) {
        dgamma[c] += (float)sum_dax;
        dbeta[c] += (float)sum_da;
    }
    const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const float rs = rstd[c], g = gamma[c], mu = mean[c], be = beta[c];
    const float m1 = (float)(sum_da / count), m2 = (float)(sum_dax / count);
    for (int q0 = t; q0 < nb * inner; q0 += 256 * NB) {
        if (q0 != t) load(q0);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (idx[j] < 0) continue;
            const long long i = idx[j];
            const float xh = (xv[j] - mu) * rs;
            const float u = g * xh + be;
            float d = dv[j];
            if (drop_p > 0.f) d = dropout_keep(seed, site, (unsigned long long)i, drop_p) ? d * ks : 0.f;
            const float da = u > 0.f ? d : d * expf(u);
            dx[i] = g * rs * (da - m1 - xh * m2);
        }
    }
}

static inline int grid_for(long long n, int block, int cap) {
    long long g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace eeg

using namespace eeg;

extern "C" int eegclip_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                     int rows, int cols, float eps, void* stream) {
    if (!x || !gamma || !beta || !y || rows < 0 || cols < 1 || cols > 64 * LN_MAXC) return EEGCLIP_EINVAL;
    if (rows == 0) return 0;
    const int grid = grid_for(rows, 4, 8192);
    if (cols <= 256) EEG_LAUNCH(layernorm_fwd_kernel<4>, dim3(grid), dim3(256), 0, stream, x, gamma, beta, y, mean, rstd, rows, cols, eps);
    else             EEG_LAUNCH(layernorm_fwd_kernel<LN_MAXC>, dim3(grid), dim3(256), 0, stream, x, gamma, beta, y, mean, rstd, rows, cols, eps);
    return (int)hipGetLastError();
}

static int residual_layernorm_fwd_go(const float* x, const float* resid, float* x_out, float drop_p, unsigned long long seed,
                                              unsigned int site, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                              const float* gamma2, const float* beta2, float* y2, float* mean2, float* rstd2, int rows, int cols,
                                              float eps, unsigned short* y_hi, unsigned short* y_lo, void* stream, ln_slabs xs = ln_slabs{1, 0, nullptr}) {
    const bool dbl = gamma2 != nullptr;
    if (!x || !gamma || !beta || !y || rows < 0 || cols < 1 || cols > 64 * LN_MAXC || drop_p < 0.f || drop_p >= 1.f) return EEGCLIP_EINVAL;
    if ((x_out || drop_p > 0.f) && !resid) return EEGCLIP_EINVAL;
    if (dbl && (!beta2 || !y2)) return EEGCLIP_EINVAL;
    if (rows == 0) return 0;
    if (cols == 1024 && rows <= HEAD_LN_MAX_ROWS && !dbl && (xs.n <= 1 || !(xs.stride & 3)) &&
        !((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(resid) | reinterpret_cast<uintptr_t>(x_out) | reinterpret_cast<uintptr_t>(gamma) |
           reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(xs.bias)) & 15u)) {
        EEG_LAUNCH(head_ln_fwd_kernel, dim3(rows), dim3(256), 8 * sizeof(float), stream, x, resid, x_out, drop_p, seed, site, gamma, beta, y, mean, rstd, eps, y_hi,
                   y_lo, xs);
        return (int)hipGetLastError();
    }
    const int grid = grid_for(rows, 4, 8192);
    const bool vec2 = (cols % 2 == 0) &&
                      !((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(resid) | reinterpret_cast<uintptr_t>(x_out) |
                         reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(y) |
                         reinterpret_cast<uintptr_t>(gamma2) | reinterpret_cast<uintptr_t>(beta2) | reinterpret_cast<uintptr_t>(y2) |
                         reinterpret_cast<uintptr_t>(xs.bias) | (uintptr_t)((xs.stride & 1) << 2)) & 7u);
#define EEG_RLN_GO(NG, V2, DB)                                                                                                              \
    EEG_LAUNCH((residual_layernorm_fwd_kernel<NG, V2, DB>), dim3(grid), dim3(256), 0, stream, x, resid, x_out, drop_p, seed, site, gamma, beta, y, \
               mean, rstd, gamma2, beta2, y2, mean2, rstd2, rows, cols, eps, y_hi, y_lo, xs)
    if (cols <= 256) {
        if (vec2) { if (dbl) EEG_RLN_GO(1, true, true); else EEG_RLN_GO(1, true, false); }
        else      { if (dbl) EEG_RLN_GO(1, false, true); else EEG_RLN_GO(1, false, false); }
    } else {
        if (vec2) { if (dbl) EEG_RLN_GO(4, true, true); else EEG_RLN_GO(4, true, false); }
        else      { if (dbl) EEG_RLN_GO(4, false, true); else EEG_RLN_GO(4, false, false); }
    }
#undef EEG_RLN_GO
    return (int)hipGetLastError();
}

extern "C" int eegclip_residual_layernorm_fwd(const float* x, const float* resid, float* x_out, float drop_p, unsigned long long seed,
                                              unsigned int site, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                              const float* gamma2, const float* beta2, float* y2, float* mean2, float* rstd2, int rows, int cols,
                                              float eps, void* stream) {
    return residual_layernorm_fwd_go(x, resid, x_out, drop_p, seed, site, gamma, beta, y, mean, rstd, gamma2, beta2, y2, mean2, rstd2, rows, cols, eps, nullptr,
                                     nullptr, stream);
}

extern "C" int eegclip_residual_layernorm_fwd_planes(const float* x, const float* resid, float* x_out, float drop_p, unsigned long long seed,
                                                     unsigned int site, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                                     const float* gamma2, const float* beta2, float* y2, float* mean2, float* rstd2, int rows, int cols,
                                                     float eps, void* y_hi, void* y_lo, void* stream) {
    if (!y_hi || !y_lo || (cols & 3) || ((reinterpret_cast<uintptr_t>(y_hi) | reinterpret_cast<uintptr_t>(y_lo)) & 7u)) return EEGCLIP_EINVAL;
    return residual_layernorm_fwd_go(x, resid, x_out, drop_p, seed, site, gamma, beta, y, mean, rstd, gamma2, beta2, y2, mean2, rstd2, rows, cols, eps,
                                     static_cast<unsigned short*>(y_hi), static_cast<unsigned short*>(y_lo), stream);
}

// eegclip_residual_layernorm_fwd_planes whose row operand x is the partial slabs of a K-parallel GEMM: x = sum_{s < nslabs} x[s * slab_stride + .] + x_bias[col]
extern "C" int eegclip_residual_layernorm_fwd_slabs(const float* x, const float* resid, float* x_out, float drop_p, unsigned long long seed,
                                                    unsigned int site, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                                    const float* gamma2, const float* beta2, float* y2, float* mean2, float* rstd2, int rows, int cols,
                                                    float eps, void* y_hi, void* y_lo, int nslabs, long long slab_stride, const float* x_bias, void* stream) {
    if ((y_hi != nullptr) != (y_lo != nullptr) || (y_hi && ((cols & 3) || ((reinterpret_cast<uintptr_t>(y_hi) | reinterpret_cast<uintptr_t>(y_lo)) & 7u)))) return EEGCLIP_EINVAL;
    if (nslabs < 1 || nslabs > 16 || (nslabs > 1 && slab_stride < (long long)rows * cols)) return EEGCLIP_EINVAL;
    return residual_layernorm_fwd_go(x, resid, x_out, drop_p, seed, site, gamma, beta, y, mean, rstd, gamma2, beta2, y2, mean2, rstd2, rows, cols, eps,
                                     static_cast<unsigned short*>(y_hi), static_cast<unsigned short*>(y_lo), stream, ln_slabs{nslabs, slab_stride, x_bias});
}

static bool head_ln_bwd_applies(const float* dy, const float* x, const float* gamma, const float* dx, const float* dx_drop, const float* dy_sum, int rows, int cols,
                                int nslabs, long long slab_stride) {
    return cols == 1024 && rows <= HEAD_LN_MAX_ROWS && (nslabs <= 1 || !(slab_stride & 3)) &&
           !((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(dx) |
              reinterpret_cast<uintptr_t>(dx_drop) | reinterpret_cast<uintptr_t>(dy_sum)) & 15u);
}

// the input-gradient half of eegclip_layernorm_bwd whose upstream gradient dy is the partial slabs of a K-parallel GEMM (dy = sum_{s < nslabs} dy[s * slab_stride + .])
// and whose second output dx_drop = dropout'(dx) leaves again as bf16 hi | lo planes (dd_hi / dd_lo, or NULL): the A operand of the next plane GEMM; dy_sum
// (or NULL): the summed dy, for the parameter-gradient half (eegclip_layernorm_bwd with dx == NULL), which runs as a launch of its own off the dX chain
extern "C" int eegclip_layernorm_bwd_slabs(const float* dy, int nslabs, long long slab_stride, const float* x, const float* gamma, const float* mean, const float* rstd,
                                           float* dx, int rows, int cols, float* dx_drop, void* dd_hi, void* dd_lo, float* dy_sum, float drop_p,
                                           unsigned long long seed, unsigned int site, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || rows < 0 || cols < 1 || cols > 64 * LN_MAXC || drop_p < 0.f || drop_p >= 1.f) return EEGCLIP_EINVAL;
    if (nslabs < 1 || nslabs > 16 || (nslabs > 1 && slab_stride < (long long)rows * cols)) return EEGCLIP_EINVAL;
    if ((dd_hi != nullptr) != (dd_lo != nullptr) || (dd_hi && (!dx_drop || (cols & 3) || ((reinterpret_cast<uintptr_t>(dd_hi) | reinterpret_cast<uintptr_t>(dd_lo)) & 7u))))
        return EEGCLIP_EINVAL;
    if (rows == 0) return 0;
    if (head_ln_bwd_applies(dy, x, gamma, dx, dx_drop, dy_sum, rows, cols, nslabs, slab_stride)) {
        EEG_LAUNCH(head_ln_bwd_kernel, dim3(rows), dim3(256), 8 * sizeof(float), stream, dy, x, gamma, mean, rstd, dx, 0, dx_drop, drop_p, seed, site,
                   ln_slabs{nslabs, slab_stride, nullptr}, static_cast<unsigned short*>(dd_hi), static_cast<unsigned short*>(dd_lo), dy_sum);
        return (int)hipGetLastError();
    }
    const dim3 grid(grid_for(rows, 4, 8192));
    const bool vec2 = (cols % 2 == 0) && !((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx) |
                                            reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(dx_drop) | (uintptr_t)((slab_stride & 1) << 2)) & 7u);
    if (dd_hi && !vec2) return EEGCLIP_EALIGN;
    const ln_slabs dys{nslabs, slab_stride, nullptr};
#define EEG_LN_BWD_GO(NG, V2)                                                                                                          \
    EEG_LAUNCH((layernorm_bwd_dx_kernel<NG, V2>), grid, dim3(256), 0, stream, dy, x, gamma, mean, rstd, dx, rows, cols, 0, dx_drop, drop_p, seed, site, \
               (float*)nullptr, dys, static_cast<unsigned short*>(dd_hi), static_cast<unsigned short*>(dd_lo), dy_sum)
    if (cols <= 256) { if (vec2) EEG_LN_BWD_GO(1, true); else EEG_LN_BWD_GO(1, false); }
    else             { if (vec2) EEG_LN_BWD_GO(4, true); else EEG_LN_BWD_GO(4, false); }
#undef EEG_LN_BWD_GO
    return (int)hipGetLastError();
}

extern "C" int eegclip_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                     float* dx, float* dgamma, float* dbeta, int rows, int cols, int accumulate_dx, float* dx_drop,
                                     float drop_p, unsigned long long seed, unsigned int site, void* stream) {
    // either half may be left out: dx == NULL -> parameter gradients only, dgamma == dbeta == NULL -> input gradient only (the two are
    // independent kernels; the encoder's backward runs the parameter half on its second stream, off the dX chain)
    const bool want_dx = dx != nullptr, want_par = dgamma != nullptr || dbeta != nullptr;
    if (!dy || !x || !mean || !rstd || (!want_dx && !want_par) || (want_par && (!dgamma || !dbeta)) || (want_dx && !gamma) || rows < 0 || cols < 1 ||
        cols > 64 * LN_MAXC)
        return EEGCLIP_EINVAL;
    if (drop_p < 0.f || drop_p >= 1.f || (!want_dx && dx_drop)) return EEGCLIP_EINVAL;
    if (rows == 0) return 0;
    const dim3 grid(grid_for(rows, 4, 8192));
    const bool vec2 = (cols % 2 == 0) && !((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx) |
                                            reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(dx_drop)) & 7u);
#define EEG_LN_BWD_GO(NG, V2)                                                                                                          \
    EEG_LAUNCH((layernorm_bwd_dx_kernel<NG, V2>), grid, dim3(256), 0, stream, dy, x, gamma, mean, rstd, dx, rows, cols, accumulate_dx, dx_drop, \
               drop_p, seed, site, (float*)nullptr, ln_slabs{1, 0, nullptr}, (unsigned short*)nullptr, (unsigned short*)nullptr, (float*)nullptr)
    if (want_dx && head_ln_bwd_applies(dy, x, gamma, dx, dx_drop, nullptr, rows, cols, 1, 0)) {
        EEG_LAUNCH(head_ln_bwd_kernel, dim3(rows), dim3(256), 8 * sizeof(float), stream, dy, x, gamma, mean, rstd, dx, accumulate_dx, dx_drop, drop_p, seed, site,
                   ln_slabs{1, 0, nullptr}, (unsigned short*)nullptr, (unsigned short*)nullptr, (float*)nullptr);
    } else if (want_dx) {
        if (cols <= 256) { if (vec2) EEG_LN_BWD_GO(1, true); else EEG_LN_BWD_GO(1, false); }
        else             { if (vec2) EEG_LN_BWD_GO(4, true); else EEG_LN_BWD_GO(4, false); }
    }
#undef EEG_LN_BWD_GO
    if (want_par) {
        int chunks = (rows + 63) / 64;              // >= 16 rows per thread before the atomics
        if (chunks > 256) chunks = 256;
        EEG_LAUNCH(layernorm_bwd_param_kernel, dim3(chunks, (cols + 63) / 64), dim3(256), 512 * sizeof(float), stream, dy, x, mean, rstd, dgamma,
                   dbeta, rows, cols);
    }
    return (int)hipGetLastError();
}

// dx (+ dropout'(dx)) AND the parameter gradients from ONE pass over dy / x: the input-gradient kernel leaves per-workgroup partial rows,
// stage 2 sums them.  workspace: eegclip_layernorm_bwd_full_workspace_floats(rows, cols) floats.
static int lnf_grid(int rows) { const int g = (rows + 15) / 16; return g < 1 ? 1 : (g > 1024 ? 1024 : g); }
extern "C" long long eegclip_layernorm_bwd_full_workspace_floats(int rows, int cols) { return rows < 1 || cols < 1 ? 0 : (long long)lnf_grid(rows) * 2 * cols; }

extern "C" int eegclip_layernorm_bwd_full(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                                          float* dgamma, float* dbeta, int rows, int cols, int accumulate_dx, float* dx_drop, float drop_p,
                                          unsigned long long seed, unsigned int site, float* workspace, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace || rows < 0 || cols < 1 || cols > 64 * LN_MAXC) return EEGCLIP_EINVAL;
    if (drop_p < 0.f || drop_p >= 1.f) return EEGCLIP_EINVAL;
    if (rows == 0) return 0;
    const int parts = lnf_grid(rows);
    const dim3 grid(parts);
    const bool vec2 = (cols % 2 == 0) && !((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx) |
                                            reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(dx_drop)) & 7u);
#define EEG_LNF_GO(NG, V2)                                                                                                                         \
    EEG_LAUNCH((layernorm_bwd_dx_kernel<NG, V2, true>), grid, dim3(256), 3 * 2 * NG * 256 * sizeof(float), stream, dy, x, gamma, mean, rstd, dx, rows, \
               cols, accumulate_dx, dx_drop, drop_p, seed, site, workspace, ln_slabs{1, 0, nullptr}, (unsigned short*)nullptr, (unsigned short*)nullptr, (float*)nullptr)
    if (cols <= 256) { if (vec2) EEG_LNF_GO(1, true); else EEG_LNF_GO(1, false); }
    else             { if (vec2) EEG_LNF_GO(4, true); else EEG_LNF_GO(4, false); }
#undef EEG_LNF_GO
    const int slices = parts < LNP_SLICES ? parts : LNP_SLICES;
    EEG_LAUNCH(layernorm_bwd_param_stage2_kernel, dim3((2 * cols + 63) / 64, slices), dim3(256), 256 * sizeof(float), stream, workspace, parts, cols, dgamma, dbeta);
    return (int)hipGetLastError();
}

static long long lnp_parts(int rows) { return ((long long)rows + 4 * LNP_ROWS - 1) / (4 * LNP_ROWS); }
extern "C" long long eegclip_layernorm_bwd_params_workspace_floats(int rows, int cols) {
    return rows < 1 || cols < 1 ? 0 : lnp_parts(rows) * 2 * cols;
}

extern "C" int eegclip_layernorm_bwd_params(const float* dy, const float* x, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                                            int rows, int cols, float* workspace, void* stream) {
    if (!dy || !x || !mean || !rstd || !dgamma || !dbeta || !workspace || rows < 0 || cols < 1 || cols > 64 * LN_MAXC) return EEGCLIP_EINVAL;
    if (rows == 0) return 0;
    const int parts = (int)lnp_parts(rows);
    if (cols <= 256) EEG_LAUNCH(layernorm_bwd_param_stage1_kernel<1>, dim3(parts), dim3(256), 3 * 2 * 256 * sizeof(float), stream, dy, x, mean, rstd, workspace, rows, cols);
    else             EEG_LAUNCH(layernorm_bwd_param_stage1_kernel<4>, dim3(parts), dim3(256), 3 * 2 * 1024 * sizeof(float), stream, dy, x, mean, rstd, workspace, rows, cols);
    const int slices = parts < LNP_SLICES ? parts : LNP_SLICES;
    EEG_LAUNCH(layernorm_bwd_param_stage2_kernel, dim3((2 * cols + 63) / 64, slices), dim3(256), 256 * sizeof(float), stream, workspace, parts, cols, dgamma, dbeta);
    return (int)hipGetLastError();
}

extern "C" int eegclip_bn_stats(const float* x, int outer, int C, int inner, double* sums, void* stream) {
    if (!x || !sums || outer < 1 || C < 1 || inner < 1) return EEGCLIP_EINVAL;
    int chunks = outer < 64 ? outer : 64;
    EEG_LAUNCH(bn_stats_kernel, dim3(chunks, C), dim3(inner >= 256 ? 256 : 64), 0, stream, x, outer, C, inner, sums);
    return (int)hipGetLastError();
}

extern "C" int eegclip_bn_finalize(const double* sums, double count, float eps, float momentum, int C, float* mean, float* rstd,
                                   float* running_mean, float* running_var, int train, long long* num_batches_tracked, void* stream) {
    if (!mean || !rstd || C < 1) return EEGCLIP_EINVAL;
    if (train && (!sums || count < 1.0)) return EEGCLIP_EINVAL;
    if (!train && (!running_mean || !running_var)) return EEGCLIP_EINVAL;
    EEG_LAUNCH(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, sums, count, eps, momentum, C, mean, rstd,
               running_mean, running_var, train, num_batches_tracked);
    return (int)hipGetLastError();
}

extern "C" int eegclip_bn_finalize_rows(const double* rows, int nrows, double count, float eps, float momentum, int C, float* mean, float* rstd,
                                        float* running_mean, float* running_var, long long* num_batches_tracked, double* sums_out, void* stream) {
    if (!rows || nrows < 1 || C < 1 || 2 * C > 512 || count < 1.0) return EEGCLIP_EINVAL;
    if ((mean == nullptr) != (rstd == nullptr) || (!mean && !sums_out) || (running_mean == nullptr) != (running_var == nullptr)) return EEGCLIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(rows) | reinterpret_cast<uintptr_t>(sums_out)) & 7u) return EEGCLIP_EALIGN;
    EEG_LAUNCH(bn_finalize_rows_kernel, dim3(1), dim3(512), (512 / (2 * C)) * 2 * C * sizeof(double), stream, rows, nrows, count, eps, momentum, C, mean,
               rstd, running_mean, running_var, num_batches_tracked, sums_out);
    return (int)hipGetLastError();
}

extern "C" int eegclip_bn_elu_fwd(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                  float* y, int outer, int C, int inner, float drop_p, unsigned long long seed, unsigned site,
                                  void* stream) {
    if (!x || !mean || !rstd || !gamma || !beta || !y || outer < 1 || C < 1 || inner < 1 || drop_p < 0.f || drop_p >= 1.f)
        return EEGCLIP_EINVAL;
    const long long n = (long long)outer * C * inner;
    EEG_LAUNCH(bn_elu_fwd_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, stream, x, mean, rstd, gamma, beta, y, n, C, inner,
               drop_p, seed, site);
    return (int)hipGetLastError();
}

// split form for data-parallel SyncBN: all-reduce `sums` across ranks between the two calls and pass the GLOBAL element count
extern "C" int eegclip_bn_elu_bwd_stats(const float* dz, const float* x, const float* mean, const float* rstd, const float* gamma,
                                        const float* beta, double* sums, int outer, int C, int inner, float drop_p, unsigned long long seed,
                                        unsigned site, void* stream) {
    if (!dz || !x || !mean || !rstd || !gamma || !beta || !sums || outer < 1 || C < 1 || inner < 1 || drop_p < 0.f || drop_p >= 1.f)
        return EEGCLIP_EINVAL;
    const int chunks = grid_for((long long)outer * inner, 1024, 64);        // >= 4 elements per thread
    EEG_LAUNCH(bn_elu_bwd_stats_kernel, dim3(chunks, C), dim3(256), 8 * sizeof(double), stream, dz, x, mean, rstd, gamma, beta, outer, C,
               inner, drop_p, seed, site, sums);
    return (int)hipGetLastError();
}
extern "C" int eegclip_bn_elu_bwd_apply_rows(const float* dz, const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                             const double* rows, int nrows, long long ld, int col0, double count, float* dx, float* dgamma, float* dbeta,
                                             int outer, int C, int inner, float drop_p, unsigned long long seed, unsigned site, void* stream) {
    if (!dz || !x || !mean || !rstd || !gamma || !beta || !rows || nrows < 1 || ld < col0 + 2 * C || col0 < 0 || !dx || !dgamma || !dbeta || outer < 1 || C < 1 ||
        inner < 1 || count < 1.0 || drop_p < 0.f || drop_p >= 1.f)
        return EEGCLIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(rows) & 7u) return EEGCLIP_EALIGN;
    EEG_LAUNCH(bn_elu_bwd_apply_rows_kernel, dim3(C, (outer + 31) / 32), dim3(256), 8 * sizeof(double), stream, dz, x, mean, rstd, gamma, beta, rows, nrows,
               ld, col0, count, dx, dgamma, dbeta, outer, C, inner, drop_p, seed, site);
    return (int)hipGetLastError();
}

extern "C" int eegclip_bn_elu_bwd_apply(const float* dz, const float* x, const float* mean, const float* rstd, const float* gamma,
                                        const float* beta, const double* sums, const double* sums_local, double count, float* dx, float* dgamma,
                                        float* dbeta, int outer, int C, int inner, float drop_p, unsigned long long seed, unsigned site,
                                        void* stream) {
    if (!dz || !x || !mean || !rstd || !gamma || !beta || !sums || !dx || !dgamma || !dbeta || outer < 1 || C < 1 || inner < 1 || count < 1.0 ||
        drop_p < 0.f || drop_p >= 1.f)
        return EEGCLIP_EINVAL;
    const long long n = (long long)outer * C * inner;
    EEG_LAUNCH(bn_elu_bwd_apply_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, stream, dz, x, mean, rstd, gamma, beta, sums,
               sums_local ? sums_local : sums, count, dx, dgamma, dbeta, n, C, inner, drop_p, seed, site);
    return (int)hipGetLastError();
}

extern "C" int eegclip_bn_elu_bwd(const float* dz, const float* x, const float* mean, const float* rstd, const float* gamma,
                                  const float* beta, double* sums /* [2C], zeroed by the caller */, float* dx, float* dgamma,
                                  float* dbeta, int outer, int C, int inner, float drop_p, unsigned long long seed, unsigned site,
                                  void* stream) {
    if (!dz || !x || !mean || !rstd || !gamma || !beta || !sums || !dx || !dgamma || !dbeta || outer < 1 || C < 1 || inner < 1 ||
        drop_p < 0.f || drop_p >= 1.f)
        return EEGCLIP_EINVAL;
    const long long n = (long long)outer * C * inner;
    const int chunks = grid_for((long long)outer * inner, 1024, 64);
    EEG_LAUNCH(bn_elu_bwd_stats_kernel, dim3(chunks, C), dim3(256), 8 * sizeof(double), stream, dz, x, mean, rstd, gamma, beta,
               outer, C, inner, drop_p, seed, site, sums);
    EEG_LAUNCH(bn_elu_bwd_apply_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, stream, dz, x, mean, rstd, gamma, beta, sums, sums,
               (double)outer * inner, dx, dgamma, dbeta, n, C, inner, drop_p, seed, site);
    return (int)hipGetLastError();
}
