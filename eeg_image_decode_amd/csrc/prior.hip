// Diffusion-prior pieces around the GEMMs (Generation/diffusion_prior.py): LayerNorm->SiLU->dropout stage tail, SiLU backward,
// sinusoidal timestep embedding, DDPM add_noise / ancestral step (+ classifier-free-guidance mix), MSE loss + gradient.
// All streaming / HBM-bound; the DDPM step fuses what the reference does in ~12 separate elementwise launches per step.
#include "eeg_common.h"

namespace eeg {

constexpr int LNS_MAXC = 16;

__global__ __launch_bounds__(256) void layernorm_silu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float* __restrict__ y_ln,
                                                                  float* __restrict__ y_act, float* __restrict__ mean_out,
                                                                  float* __restrict__ rstd_out, int rows, int cols, float eps,
                                                                  float drop_p, unsigned long long seed, unsigned site,
                                                                  const float* __restrict__ skip, unsigned short* __restrict__ act_hi,
                                                                  unsigned short* __restrict__ act_lo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv = 1.0f / (float)cols;
    const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float* xr = x + (long long)row * cols;
        float v[LNS_MAXC];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LNS_MAXC; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < cols ? xr[c] : 0.f;
            s += v[i];
        }
        const float mean = wave_sum(s) * inv;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LNS_MAXC; ++i) {
            const int c = lane + 64 * i;
            const float dlt = c < cols ? v[i] - mean : 0.f;
            q += dlt * dlt;
        }
        const float rstd = rsqrtf(wave_sum(q) * inv + eps);
#pragma unroll
        for (int i = 0; i < LNS_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < cols) {
                const long long idx = (long long)row * cols + c;
                const float y = (v[i] - mean) * rstd * gamma[c] + beta[c];
                y_ln[idx] = y;
                float a = silu(y);
                if (drop_p > 0.f) a = dropout_keep(seed, site, (unsigned long long)idx, drop_p) ? a * ks : 0.f;
                if (skip) a += skip[idx];                    // decoder stages: x += hidden_activations[-1 - j]   (diffusion_prior.py:199)
                y_act[idx] = a;
                if (act_hi) {                                // the activation again as bf16 planes: operand of the next plane GEMM
                    const unsigned short hb = f32_to_bf16_bits(a);
                    act_hi[idx] = hb;
                    act_lo[idx] = f32_to_bf16_bits(a - bf16_bits_to_f32(hb));
                }
            }
        }
        if (lane == 0) {
            mean_out[row] = mean;
            rstd_out[row] = rstd;
        }
    }
}

// The same stage tail with a lane owning 4 CONSECUTIVE columns (cols % 4 == 0): 16-byte loads / stores, 8-byte plane stores and ONE Philox block per
// lane and column group (dropout_keep4: the mask of element idx is word idx & 3 of block idx >> 2 -- identical to the per-element kernel above).
// A wave owns a row; NG column groups of 256 cover <= 1024 columns.
template <int NG>
__global__ __launch_bounds__(256) void prior_stage_fwd_v4_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  const float* __restrict__ skip, float* __restrict__ y_ln, float* __restrict__ y_act,
                                                                  float* __restrict__ mean_out, float* __restrict__ rstd_out, unsigned short* __restrict__ act_hi,
                                                                  unsigned short* __restrict__ act_lo, int rows, int cols, float eps, float drop_p,
                                                                  unsigned long long seed, unsigned site) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv = 1.0f / (float)cols;
    const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const long long base = (long long)row * cols;
        f32x4 v[NG], sk[NG];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int c = 256 * i + 4 * lane;
            const bool ok = c < cols;
            v[i] = ok ? *reinterpret_cast<const f32x4*>(x + base + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            sk[i] = ok && skip ? *reinterpret_cast<const f32x4*>(skip + base + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
        const float mean = wave_sum(s) * inv;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NG; ++i)
            if (256 * i + 4 * lane < cols) {
#pragma unroll
                for (int e = 0; e < 4; ++e) q += (v[i][e] - mean) * (v[i][e] - mean);
            }
        const float rstd = rsqrtf(wave_sum(q) * inv + eps);
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int c = 256 * i + 4 * lane;
            if (c < cols) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), b = *reinterpret_cast<const f32x4*>(beta + c);
                bool keep[4] = {true, true, true, true};
                if (drop_p > 0.f) dropout_keep4(seed, site, (unsigned long long)(base + c), drop_p, keep);
                f32x4 y, a;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
                    float t = silu(y[e]);
                    if (drop_p > 0.f) t = keep[e] ? t * ks : 0.f;
                    a[e] = t + sk[i][e];
                }
                *reinterpret_cast<f32x4*>(y_ln + base + c) = y;
                *reinterpret_cast<f32x4*>(y_act + base + c) = a;
                if (act_hi) {
                    u32x2_t h, l;
                    x3_split4(a[0], a[1], a[2], a[3], h, l);
                    *reinterpret_cast<u32x2_t*>(act_hi + base + c) = h;
                    *reinterpret_cast<u32x2_t*>(act_lo + base + c) = l;
                }
            }
        }
        if (lane == 0) {
            mean_out[row] = mean;
            rstd_out[row] = rstd;
        }
    }
}

// Backward of a stage tail, lane = 4 consecutive columns (see prior_stage_bwd_kernel below for the arithmetic): 8 waves, a row per wave; the parameter
// gradients of the workgroup's 8 rows meet in LDS -- one atomic per column and workgroup.
template <int NG>
__global__ __launch_bounds__(512) void prior_stage_bwd_v4_kernel(const float* __restrict__ dact, const float* __restrict__ y_ln, const float* __restrict__ x,
                                                                  const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  float* __restrict__ dx, unsigned short* __restrict__ dx_hi, unsigned short* __restrict__ dx_lo,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int cols, float drop_p,
                                                                  unsigned long long seed, unsigned site, float* __restrict__ partials) {
    EEG_LDS_BASE(f32x4, red);                                // [2][8 waves][NG][64 lanes]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv = 1.0f / (float)cols;
    const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    f32x4 pg[NG], pb[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) pg[i] = pb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int row = blockIdx.x * 8 + wave; row < rows; row += gridDim.x * 8) {
        const long long base = (long long)row * cols;
        const float mu = mean[row], rs = rstd[row];
        f32x4 g[NG], xh[NG];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int c = 256 * i + 4 * lane;
            g[i] = xh[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < cols) {
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(dact + base + c), y4 = *reinterpret_cast<const f32x4*>(y_ln + base + c);
                const f32x4 x4 = *reinterpret_cast<const f32x4*>(x + base + c), gm = *reinterpret_cast<const f32x4*>(gamma + c);
                bool keep[4] = {true, true, true, true};
                if (drop_p > 0.f) dropout_keep4(seed, site, (unsigned long long)(base + c), drop_p, keep);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float d = d4[e];
                    if (drop_p > 0.f) d = keep[e] ? d * ks : 0.f;
                    d *= silu_grad(y4[e]);
                    xh[i][e] = (x4[e] - mu) * rs;
                    pg[i][e] += d * xh[i][e];
                    pb[i][e] += d;
                    g[i][e] = d * gm[e];
                    s1 += g[i][e];
                    s2 += g[i][e] * xh[i][e];
                }
            }
        }
        const float m1 = wave_sum(s1) * inv, m2 = wave_sum(s2) * inv;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int c = 256 * i + 4 * lane;
            if (c < cols) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = rs * (g[i][e] - m1 - xh[i][e] * m2);
                if (dx) *reinterpret_cast<f32x4*>(dx + base + c) = v;
                if (dx_hi) {
                    u32x2_t h, l;
                    x3_split4(v[0], v[1], v[2], v[3], h, l);
                    *reinterpret_cast<u32x2_t*>(dx_hi + base + c) = h;
                    *reinterpret_cast<u32x2_t*>(dx_lo + base + c) = l;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        red[(wave * NG + i) * 64 + lane] = pg[i];
        red[((8 + wave) * NG + i) * 64 + lane] = pb[i];
    }
    __syncthreads();
    for (int q = threadIdx.x; q < NG * 64; q += 512) {         // q = i * 64 + lane  <->  columns 256 i + 4 lane .. + 3
        const int i = q >> 6, l = q & 63, c = 256 * i + 4 * l;
        if (c < cols) {
            f32x4 sg = f32x4{0.f, 0.f, 0.f, 0.f}, sb = sg;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const f32x4 a = red[(w * NG + i) * 64 + l], b = red[((8 + w) * NG + i) * 64 + l];
#pragma unroll
                for (int e = 0; e < 4; ++e) { sg[e] += a[e]; sb[e] += b[e]; }
            }
            if (partials) {                                    // one partial row per workgroup: [gridDim.x][dgamma | dbeta], summed by prior_param_sum_kernel
                float* pr = partials + (long long)blockIdx.x * 2 * cols;
                *reinterpret_cast<f32x4*>(pr + c) = sg;
                *reinterpret_cast<f32x4*>(pr + cols + c) = sb;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    atomicAdd(dgamma + c + e, sg[e]);
                    atomicAdd(dbeta + c + e, sb[e]);
                }
            }
        }
    }
}

// out[c] += sum over the partial rows, c over [dgamma | dbeta] (2 cols values): a thread per column, 4 independent accumulators.  (Every workgroup of the
// stage-backward kernel adding its partials with atomics put 128 .. 256 workgroups x 2 cols atomics onto 64 cache lines: same-line atomics retire one after
// the other -- 12 of that kernel's 18 us; with 32 workgroups the kernel was latency-bound instead, 16 us.)
__global__ __launch_bounds__(256) void prior_param_sum_kernel(const float* __restrict__ partials, int parts, int cols, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta) {
    // blockIdx.y = one of gridDim.y groups of partial rows (a thread walking all 128 rows of a batch of 1024 alone took 11 us: 32 dependent-latency
    // rounds on 8 workgroups); the groups meet with one atomic per column -- gridDim.y-way contention only
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= 2 * cols) return;
    const int p0 = (int)((long long)blockIdx.y * parts / gridDim.y), p1 = (int)((long long)(blockIdx.y + 1) * parts / gridDim.y);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int p = p0;
    for (; p + 3 < p1; p += 4) {
        a0 += partials[(long long)p * 2 * cols + c];
        a1 += partials[(long long)(p + 1) * 2 * cols + c];
        a2 += partials[(long long)(p + 2) * 2 * cols + c];
        a3 += partials[(long long)(p + 3) * 2 * cols + c];
    }
    for (; p < p1; ++p) a0 += partials[(long long)p * 2 * cols + c];
    const float s = (a0 + a1) + (a2 + a3);
    if (p1 > p0) atomicAdd(c < cols ? dgamma + c : dbeta + (c - cols), s);
}

// Inference form of a prior stage's tail for the sampling chain (Generation/diffusion_prior.py:186-199 inside :358-377).  Every row of a
// sampling batch shares the timestep and the condition never changes along the chain, so the time / condition embeddings of all stages are
// computed once per chain; what is left per stage and DDPM step is
//     y = SiLU(LayerNorm(x)) (+ skip)                 -> act_out   (kept for the decoder's skip additions and the output layer)
//     y + te[c] + (row < ce_rows ? ce[row][c] : 0)    -> xin_out   (the next stage's input: x + time embedding + condition embedding)
// in one launch instead of LayerNorm-SiLU, two time-embedding GEMMs, a condition GEMM and the skip axpby.
__global__ __launch_bounds__(256) void prior_stage_infer_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const float* __restrict__ skip,
                                                                 float* __restrict__ act_out, const float* __restrict__ te,
                                                                 const float* __restrict__ ce, int ce_rows, float* __restrict__ xin_out, int rows,
                                                                 int cols, float eps) {
    // one workgroup per row, <= 4 columns per thread: a sampling batch has 16 rows, so a wave-per-row kernel is 16 waves walking 16 dependent
    // loads each (6.5 us, rocprofv3); here every operand of a row is requested in one round trip before the first reduction
    EEG_LDS_BASE(float, red);                                   // [0..3] sums, [4..7] squared deviations, one per wave
    constexpr int PC = LNS_MAXC / 4;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int row = blockIdx.x;
    const long long base = (long long)row * cols;
    const bool has_ce = xin_out && row < ce_rows;
    float v[PC], gm[PC], bt[PC], sk[PC], tv[PC], cv[PC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PC; ++i) {
        const int c = t + 256 * i;
        const bool ok = c < cols;
        v[i] = ok ? x[base + c] : 0.f;
        gm[i] = ok ? gamma[c] : 0.f;
        bt[i] = ok ? beta[c] : 0.f;
        sk[i] = ok && skip ? skip[base + c] : 0.f;
        tv[i] = ok && xin_out ? te[c] : 0.f;
        cv[i] = ok && has_ce ? ce[base + c] : 0.f;
        s += v[i];
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PC; ++i) {
        const float dlt = t + 256 * i < cols ? v[i] - mean : 0.f;
        q += dlt * dlt;
    }
    q = wave_sum(q);
    if (lane == 0) red[4 + wave] = q;
    __syncthreads();
    const float rstd = rsqrtf(((red[4] + red[5]) + (red[6] + red[7])) / (float)cols + eps);
#pragma unroll
    for (int i = 0; i < PC; ++i) {
        const int c = t + 256 * i;
        if (c < cols) {
            const float a = silu((v[i] - mean) * rstd * gm[i] + bt[i]) + sk[i];
            if (act_out) act_out[base + c] = a;
            if (xin_out) xin_out[base + c] = a + tv[i] + cv[i];
        }
    }
}

__global__ __launch_bounds__(256) void silu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pre, float* __restrict__ dx,
                                                        long long n, int accumulate, float drop_p, unsigned long long seed, unsigned site) {
    const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float d = dy[i];
        if (drop_p > 0.f) d = dropout_keep(seed, site, (unsigned long long)i, drop_p) ? d * ks : 0.f;
        const float v = d * silu_grad(pre[i]);
        dx[i] = accumulate ? dx[i] + v : v;
    }
}

// dx = dy * silu'(pre) as bf16 hi | lo planes only (the gradient of the time embeddings' hidden layer: its one reader is a weight-gradient plane GEMM)
__global__ __launch_bounds__(256) void silu_bwd_planes_kernel(const float* __restrict__ dy, const float* __restrict__ pre, unsigned short* __restrict__ hi,
                                                               unsigned short* __restrict__ lo, long long n4) {
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (long long)gridDim.x * blockDim.x) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(dy + 4 * q), p = *reinterpret_cast<const f32x4*>(pre + 4 * q);
        u32x2_t h, l;
        x3_split4(d[0] * silu_grad(p[0]), d[1] * silu_grad(p[1]), d[2] * silu_grad(p[2]), d[3] * silu_grad(p[3]), h, l);
        *reinterpret_cast<u32x2_t*>(hi + 4 * q) = h;
        *reinterpret_cast<u32x2_t*>(lo + 4 * q) = l;
    }
}

// Backward of a stage tail in ONE pass (diffusion_prior.py:173-175,186-199 differentiated): act = dropout(SiLU(LayerNorm(x))) -> d LN-output =
// dropout'(dact) * silu'(y_ln); dx = rstd (g - mean(g) - xhat mean(g xhat)), g = d * gamma; dgamma += sum_rows d * xhat, dbeta += sum_rows d.
// Replaces silu_bwd + layernorm_bwd (two kernels, three passes) per stage; dx leaves as bf16 hi | lo planes (operand of the dX and weight-gradient
// plane GEMMs) and / or fp32.  A wave owns a row (<= 1024 columns: 16 per lane); a lane's columns are the same for every row, so the parameter
// gradients accumulate in registers over the workgroup's rows and meet in LDS: one atomic per column and workgroup.
__global__ __launch_bounds__(256) void prior_stage_bwd_kernel(const float* __restrict__ dact, const float* __restrict__ y_ln, const float* __restrict__ x,
                                                               const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               float* __restrict__ dx, unsigned short* __restrict__ dx_hi, unsigned short* __restrict__ dx_lo,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int cols, float drop_p,
                                                               unsigned long long seed, unsigned site) {
    EEG_LDS_BASE(float, red);                                // [2][4 waves][64 LNS_MAXC]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv = 1.0f / (float)cols;
    const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    float gm[LNS_MAXC], pg[LNS_MAXC], pb[LNS_MAXC];
#pragma unroll
    for (int i = 0; i < LNS_MAXC; ++i) {
        const int c = lane + 64 * i;
        gm[i] = c < cols ? gamma[c] : 0.f;
        pg[i] = 0.f;
        pb[i] = 0.f;
    }
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const long long base = (long long)row * cols;
        const float mu = mean[row], rs = rstd[row];
        float g[LNS_MAXC], xh[LNS_MAXC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LNS_MAXC; ++i) {
            const int c = lane + 64 * i;
            g[i] = 0.f;
            xh[i] = 0.f;
            if (c < cols) {
                float d = dact[base + c];
                if (drop_p > 0.f) d = dropout_keep(seed, site, (unsigned long long)(base + c), drop_p) ? d * ks : 0.f;
                d *= silu_grad(y_ln[base + c]);
                xh[i] = (x[base + c] - mu) * rs;
                pg[i] += d * xh[i];
                pb[i] += d;
                g[i] = d * gm[i];
                s1 += g[i];
                s2 += g[i] * xh[i];
            }
        }
        const float m1 = wave_sum(s1) * inv, m2 = wave_sum(s2) * inv;
#pragma unroll
        for (int i = 0; i < LNS_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < cols) {
                const float v = rs * (g[i] - m1 - xh[i] * m2);
                if (dx) dx[base + c] = v;
                if (dx_hi) {
                    const unsigned short hb = f32_to_bf16_bits(v);
                    dx_hi[base + c] = hb;
                    dx_lo[base + c] = f32_to_bf16_bits(v - bf16_bits_to_f32(hb));
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < LNS_MAXC; ++i) {
        red[(wave * LNS_MAXC + i) * 64 + lane] = pg[i];
        red[((4 + wave) * LNS_MAXC + i) * 64 + lane] = pb[i];
    }
    __syncthreads();
    for (int q = threadIdx.x; q < 64 * LNS_MAXC; q += 256) {   // q = i * 64 + lane  <->  column lane + 64 i
        const int i = q >> 6, l = q & 63, c = l + 64 * i;
        if (c < cols) {
            float sg = 0.f, sb = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                sg += red[(w * LNS_MAXC + i) * 64 + l];
                sb += red[((4 + w) * LNS_MAXC + i) * 64 + l];
            }
            atomicAdd(dgamma + c, sg);
            atomicAdd(dbeta + c, sb);
        }
    }
}

__global__ void timestep_embedding_kernel(const float* __restrict__ t, int n, int dim, float* __restrict__ out) {
    const int half = dim / 2;
    const int total = n * half;
    const float k = -9.210340371976184f / (float)half;      // -ln(10000) / half
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int r = i / half, j = i % half;
        const float ang = t[r] * expf(k * (float)j);
        out[(long long)r * dim + j] = cosf(ang);
        out[(long long)r * dim + half + j] = sinf(ang);
    }
}

__global__ __launch_bounds__(256) void ddpm_add_noise_kernel(const float* __restrict__ h, const float* __restrict__ noise,
                                                              const long long* __restrict__ t, const float* __restrict__ sa,
                                                              const float* __restrict__ sb, float* __restrict__ out, int n, int d) {
    const long long total = (long long)n * d;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long tt = t[i / d];
        out[i] = sa[tt] * h[i] + sb[tt] * noise[i];
    }
}

__global__ __launch_bounds__(256) void ddpm_step_kernel(const float* __restrict__ x, const float* __restrict__ eps_c,
                                                         const float* __restrict__ eps_u, float g, float sa, float sb, float c0, float ct,
                                                         float sigma, const float* __restrict__ noise, float* __restrict__ out, float* __restrict__ out_dup,
                                                         long long n) {
    const float inv_sa = 1.0f / sa;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float e = eps_c[i];
        if (eps_u) { const float u = eps_u[i]; e = u + g * (e - u); }
        const float xi = x[i];
        float x0 = (xi - sb * e) * inv_sa;
        x0 = fminf(1.0f, fmaxf(-1.0f, x0));
        float v = c0 * x0 + ct * xi;
        if (noise && sigma != 0.f) v += sigma * noise[i];
        out[i] = v;
        if (out_dup) out_dup[i] = v;                    // second copy: the conditional / unconditional halves of the next step's 2N-row input
    }
}

__global__ __launch_bounds__(256) void mse_loss_grad_kernel(const float* __restrict__ pred, const float* __restrict__ target, long long n,
                                                             float* __restrict__ loss, float* __restrict__ dpred, float weight) {
    const float inv = weight / (float)n;
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float d = pred[i] - target[i];
        s += d * d;
        if (dpred) dpred[i] = 2.0f * d * inv;
    }
    // one atomic per workgroup (2048 same-address atomics at ~12 ns each were 25 of this kernel's 31 us at 1024 x 1024)
    EEG_LDS_BASE(float, red);                                // [4 waves]
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0 && loss) atomicAdd(loss, ((red[0] + red[1]) + (red[2] + red[3])) * inv);
}

static inline int pgrid(long long n, int cap = 2048) {
    long long g = (n + 255) / 256;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace eeg

using namespace eeg;

extern "C" int eegclip_layernorm_silu_fwd(const float* x, const float* gamma, const float* beta, float* y_ln, float* y_act, float* mean,
                                          float* rstd, int rows, int cols, float eps, float drop_p, unsigned long long seed, unsigned site,
                                          void* stream) {
    if (!x || !gamma || !beta || !y_ln || !y_act || !mean || !rstd || rows < 0 || cols < 1 || cols > 64 * LNS_MAXC || drop_p < 0.f || drop_p >= 1.f)
        return EEGCLIP_EINVAL;
    if (rows == 0) return 0;
    int grid = (rows + 3) / 4;
    if (grid > 2048) grid = 2048;
    EEG_LAUNCH(layernorm_silu_fwd_kernel, dim3(grid), dim3(256), 0, stream, x, gamma, beta, y_ln, y_act, mean, rstd, rows, cols, eps, drop_p,
               seed, site, (const float*)nullptr, (unsigned short*)nullptr, (unsigned short*)nullptr);
    return (int)hipGetLastError();
}

extern "C" int eegclip_prior_stage_fwd(const float* x, const float* gamma, const float* beta, const float* skip, float* y_ln, float* y_act, float* mean,
                                       float* rstd, void* act_hi, void* act_lo, int rows, int cols, float eps, float drop_p, unsigned long long seed,
                                       unsigned site, void* stream) {
    if (!x || !gamma || !beta || !y_ln || !y_act || !mean || !rstd || rows < 0 || cols < 1 || cols > 64 * LNS_MAXC || drop_p < 0.f || drop_p >= 1.f ||
        (!act_hi) != (!act_lo))
        return EEGCLIP_EINVAL;
    if (rows == 0) return 0;
    int grid = (rows + 3) / 4;
    if (grid > 2048) grid = 2048;
    const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(skip) |
                         reinterpret_cast<uintptr_t>(y_ln) | reinterpret_cast<uintptr_t>(y_act);
    const bool v4 = cols % 4 == 0 && !(al & 15u) && !((reinterpret_cast<uintptr_t>(act_hi) | reinterpret_cast<uintptr_t>(act_lo)) & 7u);
    unsigned short *ph = static_cast<unsigned short*>(act_hi), *plo = static_cast<unsigned short*>(act_lo);
    if (v4 && cols <= 256) EEG_LAUNCH(prior_stage_fwd_v4_kernel<1>, dim3(grid), dim3(256), 0, stream, x, gamma, beta, skip, y_ln, y_act, mean, rstd, ph, plo, rows, cols, eps, drop_p, seed, site);
    else if (v4) EEG_LAUNCH(prior_stage_fwd_v4_kernel<4>, dim3(grid), dim3(256), 0, stream, x, gamma, beta, skip, y_ln, y_act, mean, rstd, ph, plo, rows, cols, eps, drop_p, seed, site);
    else EEG_LAUNCH(layernorm_silu_fwd_kernel, dim3(grid), dim3(256), 0, stream, x, gamma, beta, y_ln, y_act, mean, rstd, rows, cols, eps, drop_p, seed, site, skip, ph, plo);
    return (int)hipGetLastError();
}

static int psb_grid(int rows) { const int g = (rows + 7) / 8; return g < 1 ? 1 : (g > 1024 ? 1024 : g); }
extern "C" long long eegclip_prior_stage_bwd_workspace_floats(int rows, int cols) { return rows < 1 || cols < 1 ? 0 : (long long)psb_grid(rows) * 2 * cols; }

// workspace: NULL -> the parameter gradients are added with atomics by this launch; else eegclip_prior_stage_bwd_workspace_floats(rows, cols) floats that
// receive one partial row per workgroup, to be summed into dgamma / dbeta by eegclip_prior_stage_bwd_params (its own launch: it can run off the dX chain)
extern "C" int eegclip_prior_stage_bwd(const float* dact, const float* y_ln, const float* x, const float* gamma, const float* mean, const float* rstd,
                                       float* dx, void* dx_hi, void* dx_lo, float* dgamma, float* dbeta, int rows, int cols, float drop_p,
                                       unsigned long long seed, unsigned site, float* workspace, void* stream) {
    if (!dact || !y_ln || !x || !gamma || !mean || !rstd || (!workspace && (!dgamma || !dbeta)) || (!dx && !dx_hi) || (!dx_hi) != (!dx_lo) || rows < 0 || cols < 1 ||
        cols > 64 * LNS_MAXC || drop_p < 0.f || drop_p >= 1.f)
        return EEGCLIP_EINVAL;
    if (rows == 0) return 0;
    unsigned short *ph = static_cast<unsigned short*>(dx_hi), *plo = static_cast<unsigned short*>(dx_lo);
    const uintptr_t al = reinterpret_cast<uintptr_t>(dact) | reinterpret_cast<uintptr_t>(y_ln) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gamma) |
                         reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(workspace);
    const bool v4 = cols % 4 == 0 && !(al & 15u) && !((reinterpret_cast<uintptr_t>(dx_hi) | reinterpret_cast<uintptr_t>(dx_lo)) & 7u);
    if (workspace && !v4) return EEGCLIP_EALIGN;             // (the partial-row form is the vectorised kernel's)
    // with a workspace: a row per wave, 8 rows per workgroup, nothing contended; without: 32 rows per workgroup, so that few workgroups end in atomics
    int grid = workspace ? psb_grid(rows) : (rows + 31) / 32;
    if (grid > 1024) grid = 1024;
    if (v4 && cols <= 256) EEG_LAUNCH(prior_stage_bwd_v4_kernel<1>, dim3(grid), dim3(512), 2 * 8 * 1 * 64 * sizeof(f32x4), stream, dact, y_ln, x, gamma, mean, rstd, dx, ph, plo, dgamma, dbeta, rows, cols, drop_p, seed, site, workspace);
    else if (v4) EEG_LAUNCH(prior_stage_bwd_v4_kernel<4>, dim3(grid), dim3(512), 2 * 8 * 4 * 64 * sizeof(f32x4), stream, dact, y_ln, x, gamma, mean, rstd, dx, ph, plo, dgamma, dbeta, rows, cols, drop_p, seed, site, workspace);
    else EEG_LAUNCH(prior_stage_bwd_kernel, dim3((rows + 7) / 8 > 1024 ? 1024 : (rows + 7) / 8), dim3(256), 2 * 4 * 64 * LNS_MAXC * sizeof(float), stream, dact, y_ln, x, gamma, mean, rstd, dx, ph, plo, dgamma, dbeta, rows, cols, drop_p, seed, site);
    return (int)hipGetLastError();
}

extern "C" int eegclip_prior_stage_bwd_params(const float* workspace, int rows, int cols, float* dgamma, float* dbeta, void* stream) {
    if (!workspace || !dgamma || !dbeta || rows < 1 || cols < 1) return EEGCLIP_EINVAL;
    const int parts = psb_grid(rows);
    EEG_LAUNCH(prior_param_sum_kernel, dim3((2 * cols + 255) / 256, parts >= 16 ? 16 : 1), dim3(256), 0, stream, workspace, parts, cols, dgamma, dbeta);
    return (int)hipGetLastError();
}

extern "C" int eegclip_silu_bwd_planes(const float* dy, const float* pre, void* dx_hi, void* dx_lo, long long n, void* stream) {
    if (!dy || !pre || !dx_hi || !dx_lo || n < 0 || (n & 3)) return EEGCLIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(pre)) & 15u) return EEGCLIP_EALIGN;
    if ((reinterpret_cast<uintptr_t>(dx_hi) | reinterpret_cast<uintptr_t>(dx_lo)) & 7u) return EEGCLIP_EALIGN;
    if (n == 0) return 0;
    EEG_LAUNCH(silu_bwd_planes_kernel, dim3(pgrid(n / 4, 4096)), dim3(256), 0, stream, dy, pre, static_cast<unsigned short*>(dx_hi),
               static_cast<unsigned short*>(dx_lo), n / 4);
    return (int)hipGetLastError();
}

extern "C" int eegclip_prior_stage_infer(const float* x, const float* gamma, const float* beta, const float* skip, float* act_out, const float* te,
                                         const float* ce, int ce_rows, float* xin_out, int rows, int cols, float eps, void* stream) {
    if (!x || !gamma || !beta || rows < 0 || cols < 1 || cols > 64 * LNS_MAXC || (!act_out && !xin_out)) return EEGCLIP_EINVAL;
    if (xin_out && (!te || ce_rows < 0 || (ce_rows > 0 && !ce))) return EEGCLIP_EINVAL;
    if (rows == 0) return 0;
    EEG_LAUNCH(prior_stage_infer_kernel, dim3(rows), dim3(256), 8 * sizeof(float), stream, x, gamma, beta, skip, act_out, te, ce, ce_rows, xin_out, rows, cols, eps);
    return (int)hipGetLastError();
}

extern "C" int eegclip_silu_bwd(const float* dy, const float* pre, float* dx, long long n, int accumulate, float drop_p,
                                unsigned long long seed, unsigned site, void* stream) {
    if (!dy || !pre || !dx || n < 0 || drop_p < 0.f || drop_p >= 1.f) return EEGCLIP_EINVAL;
    if (n == 0) return 0;
    EEG_LAUNCH(silu_bwd_kernel, dim3(pgrid(n, 4096)), dim3(256), 0, stream, dy, pre, dx, n, accumulate, drop_p, seed, site);
    return (int)hipGetLastError();
}

extern "C" int eegclip_timestep_embedding(const float* t, int n, int dim, float* out, void* stream) {
    if (!t || !out || n < 1 || dim < 2 || (dim & 1)) return EEGCLIP_EINVAL;
    EEG_LAUNCH(timestep_embedding_kernel, dim3(pgrid((long long)n * dim / 2, 256)), dim3(256), 0, stream, t, n, dim, out);
    return (int)hipGetLastError();
}

extern "C" int eegclip_ddpm_add_noise(const float* h, const float* noise, const long long* t, const float* sqrt_acp, const float* sqrt_1macp,
                                      float* out, int n, int d, void* stream) {
    if (!h || !noise || !t || !sqrt_acp || !sqrt_1macp || !out || n < 1 || d < 1) return EEGCLIP_EINVAL;
    EEG_LAUNCH(ddpm_add_noise_kernel, dim3(pgrid((long long)n * d)), dim3(256), 0, stream, h, noise, t, sqrt_acp, sqrt_1macp, out, n, d);
    return (int)hipGetLastError();
}

extern "C" int eegclip_ddpm_step(const float* x, const float* eps_c, const float* eps_u, float guidance, float sa, float sb, float c0, float ct,
                                 float sigma, const float* noise, float* out, float* out_dup, long long n, void* stream) {
    if (!x || !eps_c || !out || n < 1 || sa == 0.f) return EEGCLIP_EINVAL;
    EEG_LAUNCH(ddpm_step_kernel, dim3(pgrid(n)), dim3(256), 0, stream, x, eps_c, eps_u, guidance, sa, sb, c0, ct, sigma, noise, out, out_dup, n);
    return (int)hipGetLastError();
}

extern "C" int eegclip_mse_loss_grad(const float* pred, const float* target, long long n, float* loss, float* dpred, void* stream) {
    if (!pred || !target || n < 1 || (!loss && !dpred)) return EEGCLIP_EINVAL;
    EEG_LAUNCH(mse_loss_grad_kernel, dim3(pgrid(n / 8 + 1, 256)), dim3(256), 4 * sizeof(float), stream, pred, target, n, loss, dpred, 1.0f);
    return (int)hipGetLastError();
}
// *loss += weight * mean((pred - target)^2), dpred = weight * 2 (pred - target) / n: the MSE term of the reconstruction objective inside the step plan
// (Generation/ATMS_reconstruction.py:227: 10 * alpha * MSE), its gradient written as one more slab of the encoder backward's upstream gradient
extern "C" int eegclip_mse_loss_grad_scaled(const float* pred, const float* target, long long n, float weight, float* loss, float* dpred, void* stream) {
    if (!pred || !target || n < 1 || (!loss && !dpred)) return EEGCLIP_EINVAL;
    EEG_LAUNCH(mse_loss_grad_kernel, dim3(pgrid(n / 8 + 1, 256)), dim3(256), 4 * sizeof(float), stream, pred, target, n, loss, dpred, weight);
    return (int)hipGetLastError();
}
