// Tail of tsconv + the Enc_eeg projection (Retrieval/ATMS_retrieval.py:107-109,113-114,145), one workgroup per sample:
//       z2 = dropout(ELU(BatchNorm2(y2)))            y2, z2 (B,40,36)
//       feat[b, w*40 + e] = bias[e] + sum_c W[e,c] * z2[b,c,w]          1x1 conv + 'b e h w -> b (h w) e' + flatten -> (B,1440)
// and its backward up to the BatchNorm statistics.  The work is tiny (58 k MACs per sample); as three launches (BN+ELU elementwise, a
// 9216 x 40 x 40 GEMM through two-level index maps, split-K weight gradient) it cost 29 + 7 us forward and 22 + 23 + 8 us backward,
// almost all of it launch latency and index arithmetic.  Here a sample's 40 x 36 tile, its gradient and the 40 x 40 weights sit in LDS.
#include "eeg_common.h"
#include "split_rider.h"

#include <stdlib.h>

namespace eeg {

constexpr int PJ_C = 40;       // channels in and out
constexpr int PJ_W = 36;       // positions
constexpr int PJ_N = PJ_C * PJ_W;
constexpr int PJ_LW = 41;      // padded row stride of the [e][c] / [w][e] images
constexpr int PJ_PART = PJ_C * PJ_C + PJ_C + 2 * PJ_C;      // partial row of the backward: dW | dbias | BatchNorm-backward sums

// rows != NULL (training, round 5): the BatchNorm2 batch statistics come as `nrows` partial rows [sum(40) | sumsq(40)] (fp64; what eegclip_cstack_fwd leaves
// per sample) -- every workgroup sums them in a fixed order (3 slices x 80 columns, slices added in order: the same value everywhere, no atomics), workgroup
// 0 stores mean / rstd for the backward and updates the running statistics + step counter: eegclip_bn_finalize's work without its launch.
__global__ __launch_bounds__(256) void proj1x1_fwd_kernel(const float* __restrict__ y2, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ Wt, const float* __restrict__ bias, float* __restrict__ z2,
                                                           float* __restrict__ feat, int B, float drop_p, unsigned long long seed, unsigned site,
                                                           const double* __restrict__ rows, int nrows, double count, float eps, float momentum,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out, float* __restrict__ run_mean,
                                                           float* __restrict__ run_var, long long* __restrict__ nbt,
                                                           unsigned short* __restrict__ feat_hi, unsigned short* __restrict__ feat_lo, const rider_table riders) {
    if ((int)blockIdx.x >= B) {           // extra workgroups: plane splits riding in this launch (csrc/split_rider.h); nothing here depends on them
        split_rider(riders, (int)blockIdx.x - B, (int)gridDim.x - B);
        return;
    }
    EEG_LDS_BASE(float, lds);
    float* zs = lds;                      // [40][36]  z2 of this sample
    float* ws = zs + PJ_N;                // [40][41]  W[e][c]
    float* aff = ws + PJ_C * PJ_LW;       // [40] scale | [40] shift of BatchNorm2
    double* scr = reinterpret_cast<double*>(aff + 2 * PJ_C);      // [3][80] (rows != NULL)
    const int t = threadIdx.x, b = blockIdx.x;
    const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    // every global load of the sample is issued HERE, before the statistics prologue: as `for (i = t; ..) lds[..] = global[..]` loops (runtime trip count, not
    // unrolled) each iteration waited for its own L2 round trip -- 6 + 7 serialised latencies per workgroup of a kernel that is nothing but latency
    constexpr int NY = (PJ_N + 255) / 256, NWL = (PJ_C * PJ_C + 255) / 256;
    float yv[NY], wv[NWL];
#pragma unroll
    for (int j = 0; j < NY; ++j) yv[j] = t + 256 * j < PJ_N ? y2[(long long)b * PJ_N + t + 256 * j] : 0.f;
#pragma unroll
    for (int j = 0; j < NWL; ++j) wv[j] = t + 256 * j < PJ_C * PJ_C ? Wt[t + 256 * j] : 0.f;
    if (rows) {
        if (t < 3 * 2 * PJ_C) {
            const int sl = t / (2 * PJ_C), col = t % (2 * PJ_C);
            scr[t] = ordered_column_sum<3, 29>(rows, nrows, sl, 2 * PJ_C, col);      // (29 loads in flight: 3 L2 round trips for the 86 rows of a slice)
        }
        __syncthreads();
        if (t < PJ_C) {
            const double s = (scr[t] + scr[2 * PJ_C + t]) + scr[4 * PJ_C + t], q = (scr[PJ_C + t] + scr[3 * PJ_C + t]) + scr[5 * PJ_C + t];
            const double m = s / count;
            double var = q / count - m * m;
            if (var < 0.0) var = 0.0;
            const float mf = (float)m, rs = (float)(1.0 / sqrt(var + (double)eps));
            if (b == 0) {
                mean_out[t] = mf;
                rstd_out[t] = rs;
                if (run_mean) {
                    const double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
                    run_mean[t] = (1.f - momentum) * run_mean[t] + momentum * mf;
                    run_var[t] = (1.f - momentum) * run_var[t] + momentum * (float)unb;
                }
                if (t == 0 && nbt) *nbt += 1;
            }
            aff[t] = gamma[t] * rs;
            aff[PJ_C + t] = beta[t] - mf * gamma[t] * rs;
        }
    } else if (t < PJ_C) {
        aff[t] = gamma[t] * rstd[t];
        aff[PJ_C + t] = beta[t] - mean[t] * gamma[t] * rstd[t];
    }
#pragma unroll
    for (int j = 0; j < NWL; ++j) {
        const int i = t + 256 * j;
        if (i < PJ_C * PJ_C) ws[(i / PJ_C) * PJ_LW + i % PJ_C] = wv[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NY; ++j) {
        const int i = t + 256 * j;
        if (i < PJ_N) {
            const int c = i / PJ_W;
            const long long idx = (long long)b * PJ_N + i;
            float v = elu1(yv[j] * aff[c] + aff[PJ_C + c]);
            if (drop_p > 0.f) v = dropout_keep(seed, site, (unsigned long long)idx, drop_p) ? v * ks : 0.f;
            z2[idx] = v;
            zs[i] = v;
        }
    }
    __syncthreads();
    for (int o = t; o < PJ_N; o += 256) {
        const int w = o / PJ_C, e = o % PJ_C;
        float acc = bias[e];
#pragma unroll 8
        for (int c = 0; c < PJ_C; ++c) acc += ws[e * PJ_LW + c] * zs[c * PJ_W + w];
        feat[(long long)b * PJ_N + o] = acc;
        if (feat_hi) {                    // ... again as bf16 hi | lo planes: the A operand of the projection head's first Linear (csrc/head_gemm.hip)
            const unsigned short hi = f32_to_bf16_bits(acc);
            feat_hi[(long long)b * PJ_N + o] = hi;
            feat_lo[(long long)b * PJ_N + o] = f32_to_bf16_bits(acc - bf16_bits_to_f32(hi));
        }
    }
}

// backward, first half (everything before the BatchNorm batch sums are known); one workgroup walks `spw` samples:
//   dW[e,c] += sum_{b,w} dfeat[b,w,e] z2[b,c,w] ;  dbias[e] += sum_{b,w} dfeat[b,w,e]
//   dz2[b,c,w] = sum_e W[e,c] dfeat[b,w,e]                                     (written: the apply pass re-derives da from it)
//   da = dz2 * mask/(1-p) * ELU'(BN(y2)) ;  sums[c] += da ;  sums[40 + c] += da * xhat        (fp64 atomics, 80 per workgroup)
__global__ __launch_bounds__(256) void proj1x1_bwd_kernel(const float* __restrict__ dfeat, const float* __restrict__ z2, const float* __restrict__ Wt,
                                                           const float* __restrict__ y2, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ dz2,
                                                           float* __restrict__ dW, float* __restrict__ dbias, double* __restrict__ sums, double* __restrict__ partials, int B, int spw,
                                                           float drop_p, unsigned long long seed, unsigned site, int nslabs, long long slab_stride,
                                                           double* __restrict__ bn_rows) {
    EEG_LDS_BASE(float, lds);
    float* zs = lds;                      // [40][36]  z2[c][w]
    float* ds = zs + PJ_N;                // [36][41]  dfeat[w][e]
    float* ws = ds + PJ_W * PJ_LW;        // [40][41]  W[e][c]
    float* das = ws + PJ_C * PJ_LW;       // [2][40][36]  da and da * xhat of this sample (reduced per channel by 80 threads: no LDS atomics)
    const int t = threadIdx.x;
    const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    {
        constexpr int NWL = (PJ_C * PJ_C + 255) / 256;
        float wv[NWL];
#pragma unroll
        for (int j = 0; j < NWL; ++j) wv[j] = t + 256 * j < PJ_C * PJ_C ? Wt[t + 256 * j] : 0.f;
#pragma unroll
        for (int j = 0; j < NWL; ++j) {
            const int i = t + 256 * j;
            if (i < PJ_C * PJ_C) ws[(i / PJ_C) * PJ_LW + i % PJ_C] = wv[j];
        }
    }
    double csum = 0.0;                                  // threads < 80: running channel sum (t < 40: da, else da * xhat)
    constexpr int NWO = (PJ_C * PJ_C + 255) / 256;      // weight-gradient entries per thread
    float dwp[NWO];
#pragma unroll
    for (int j = 0; j < NWO; ++j) dwp[j] = 0.f;
    float dbp = 0.f;                                    // threads < 40: bias gradient of channel t
    const int b0 = blockIdx.x * spw;
    constexpr int NY = (PJ_N + 255) / 256, SLB = 4;
    for (int b = b0; b < b0 + spw && b < B; ++b) {
        // all global loads of the sample in flight together (z2, y2, the partial slabs of dfeat in groups of SLB): as runtime-length loops each element waited
        // for its own L2 round trip, and each slab for the previous one
        float zv[NY], yv[NY], gv[NY], mv[NY], rv[NY], gav[NY], bev[NY];
#pragma unroll
        for (int j = 0; j < NY; ++j) {
            const bool ok = t + 256 * j < PJ_N;
            const int i = ok ? t + 256 * j : 0;
            const long long idx = (long long)b * PJ_N + i;
            zv[j] = z2[idx];
            yv[j] = y2[idx];
            gv[j] = 0.f;
            mv[j] = mean[i / PJ_W];
            rv[j] = rstd[i / PJ_W];
            gav[j] = gamma[i / PJ_W];
            bev[j] = beta[i / PJ_W];
        }
        for (int s0 = 0; s0 < nslabs; s0 += SLB) {      // (partial slabs of the K-parallel GEMM, added in slice order)
            float pv[SLB][NY];
#pragma unroll
            for (int k = 0; k < SLB; ++k)
#pragma unroll
                for (int j = 0; j < NY; ++j) {
                    const bool ok = t + 256 * j < PJ_N && s0 + k < nslabs;
                    pv[k][j] = dfeat[(ok ? (long long)(s0 + k) * slab_stride : 0) + (long long)b * PJ_N + (ok ? t + 256 * j : 0)];
                }
#pragma unroll
            for (int k = 0; k < SLB; ++k)
                if (s0 + k < nslabs) {
#pragma unroll
                    for (int j = 0; j < NY; ++j) gv[j] = (s0 + k == 0) ? pv[k][j] : gv[j] + pv[k][j];
                }
        }
        __syncthreads();                                // previous sample consumed (first pass: orders the weight staging)
#pragma unroll
        for (int j = 0; j < NY; ++j) {
            const int i = t + 256 * j;
            if (i < PJ_N) {
                zs[i] = zv[j];
                ds[(i / PJ_C) * PJ_LW + i % PJ_C] = gv[j];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NY; ++j) {                  // input gradient of the 1x1 conv + BatchNorm-backward statistics
            const int i = t + 256 * j;
            if (i >= PJ_N) break;
            const int c = i / PJ_W, w = i % PJ_W;
            float acc = 0.f;
#pragma unroll 8
            for (int e = 0; e < PJ_C; ++e) acc += ws[e * PJ_LW + c] * ds[w * PJ_LW + e];
            const long long idx = (long long)b * PJ_N + i;
            dz2[idx] = acc;
            const float xh = (yv[j] - mv[j]) * rv[j];
            const float u = gav[j] * xh + bev[j];
            float d = acc;
            if (drop_p > 0.f) d = dropout_keep(seed, site, (unsigned long long)idx, drop_p) ? d * ks : 0.f;
            const float da = u > 0.f ? d : d * expf(u);
            das[i] = da;
            das[PJ_N + i] = da * xh;
        }
#pragma unroll
        for (int j = 0; j < NWO; ++j) {                 // weight gradient
            const int o = t + 256 * j;
            if (o < PJ_C * PJ_C) {
                const int e = o / PJ_C, c = o % PJ_C;
                float acc = 0.f;
#pragma unroll 6
                for (int w = 0; w < PJ_W; ++w) acc += ds[w * PJ_LW + e] * zs[c * PJ_W + w];
                dwp[j] += acc;
            }
        }
        if (t < PJ_C) {
            float acc = 0.f;
            for (int w = 0; w < PJ_W; ++w) acc += ds[w * PJ_LW + t];
            dbp += acc;
        }
        __syncthreads();                                // da tiles complete
        if (t < 2 * PJ_C) {
            float acc = 0.f;
            for (int w = 0; w < PJ_W; ++w) acc += das[t * PJ_W + w];
            csum += acc;
        }
    }
    if (partials) {
        // one partial row [dW 1600 | dbias 40 | sums 80] per workgroup, reduced column-wise by proj1x1_bwd_reduce_kernel: with one workgroup per
        // sample the 1720 atomics per workgroup (256-way contention per address) were half of the kernel's 30 us
        double* row = partials + (long long)blockIdx.x * PJ_PART;
#pragma unroll
        for (int j = 0; j < NWO; ++j) {
            const int o = t + 256 * j;
            if (o < PJ_C * PJ_C) row[o] = (double)dwp[j];
        }
        if (t < PJ_C) row[PJ_C * PJ_C + t] = (double)dbp;
        if (t < 2 * PJ_C) row[PJ_C * PJ_C + PJ_C + t] = csum;
        if (bn_rows && t < 2 * PJ_C) bn_rows[(long long)blockIdx.x * 2 * PJ_C + t] = csum;      // ... again as a compact [workgroup][80] table for the apply pass's prologue
        return;
    }
#pragma unroll
    for (int j = 0; j < NWO; ++j) {
        const int o = t + 256 * j;
        if (o < PJ_C * PJ_C) atomicAdd(dW + o, dwp[j]);
    }
    if (t < PJ_C) atomicAdd(dbias + t, dbp);
    if (t < 2 * PJ_C) atomicAdd(sums + t, csum);
}

// column sums of the partial rows: grid (column blocks of 64, PJ_SLICES slices of the rows); one atomic per column and slice
constexpr int PJ_SLICES = 8;
__global__ __launch_bounds__(256) void proj1x1_bwd_reduce_kernel(const double* __restrict__ partials, int nparts, float* __restrict__ dW,
                                                                  float* __restrict__ dbias, double* __restrict__ sums) {
    EEG_LDS_BASE(double, red);   // [4][64]
    const int lane = threadIdx.x & 63, g = wave_uniform(threadIdx.x >> 6);
    const int c = blockIdx.x * 64 + lane;
    double s = 0.0;
    if (c < PJ_PART) {
        const int per = (nparts + PJ_SLICES - 1) / PJ_SLICES;
        const int p0 = blockIdx.y * per, p1 = p0 + per < nparts ? p0 + per : nparts;
#pragma unroll 8
        for (int p = p0 + g; p < p1; p += 4) s += partials[(long long)p * PJ_PART + c];
    }
    red[g * 64 + lane] = s;
    __syncthreads();
    if (g == 0 && c < PJ_PART) {
        const double v = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
        if (c < PJ_C * PJ_C)             atomicAdd(dW + c, (float)v);
        else if (c < PJ_C * PJ_C + PJ_C) atomicAdd(dbias + c - PJ_C * PJ_C, (float)v);
        else if (sums)                   atomicAdd(sums + c - PJ_C * PJ_C - PJ_C, v);
    }
}

}  // namespace eeg

using namespace eeg;

static size_t pj_fwd_lds() { return (PJ_N + PJ_C * PJ_LW + 2 * PJ_C) * sizeof(float) + 6 * PJ_C * sizeof(double); }

extern "C" int eegclip_proj1x1_fwd(const float* y2, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* W,
                                   const float* bias, float* z2, float* feat, int B, float drop_p, unsigned long long seed, unsigned int site,
                                   void* stream) {
    if (!y2 || !mean || !rstd || !gamma || !beta || !W || !bias || !z2 || !feat || B < 1 || drop_p < 0.f || drop_p >= 1.f) return EEGCLIP_EINVAL;
    EEG_LAUNCH(proj1x1_fwd_kernel, dim3(B), dim3(256), pj_fwd_lds(), stream, y2, mean, rstd, gamma, beta, W, bias, z2, feat, B, drop_p, seed, site,
               (const double*)nullptr, 0, 1.0, 0.f, 0.f, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (long long*)nullptr,
               (unsigned short*)nullptr, (unsigned short*)nullptr, rider_table{});
    return (int)hipGetLastError();
}

extern "C" int eegclip_proj1x1_fwd_rows(const float* y2, const double* rows, int nrows, double count, float eps, float momentum, float* mean, float* rstd,
                                        float* running_mean, float* running_var, long long* num_batches_tracked, const float* gamma, const float* beta,
                                        const float* W, const float* bias, float* z2, float* feat, int B, float drop_p, unsigned long long seed,
                                        unsigned int site, void* stream) {
    if (!y2 || !rows || nrows < 1 || count < 1.0 || !mean || !rstd || !gamma || !beta || !W || !bias || !z2 || !feat || B < 1 || drop_p < 0.f || drop_p >= 1.f)
        return EEGCLIP_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return EEGCLIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(rows) & 7u) return EEGCLIP_EALIGN;
    EEG_LAUNCH(proj1x1_fwd_kernel, dim3(B), dim3(256), pj_fwd_lds(), stream, y2, (const float*)nullptr, (const float*)nullptr, gamma, beta, W, bias, z2, feat, B,
               drop_p, seed, site, rows, nrows, count, eps, momentum, mean, rstd, running_mean, running_var, num_batches_tracked, (unsigned short*)nullptr,
               (unsigned short*)nullptr, rider_table{});
    return (int)hipGetLastError();
}

// eegclip_proj1x1_fwd_rows that also leaves `feat` as bf16 hi | lo planes (feat_hi / feat_lo, (B, 1440) each); riders: dense plane splits performed by extra
// workgroups of this launch (csrc/split_rider.h)
extern "C" int eegclip_proj1x1_fwd_rows_planes(const float* y2, const double* rows, int nrows, double count, float eps, float momentum, float* mean, float* rstd,
                                               float* running_mean, float* running_var, long long* num_batches_tracked, const float* gamma, const float* beta,
                                               const float* W, const float* bias, float* z2, float* feat, int B, float drop_p, unsigned long long seed,
                                               unsigned int site, void* feat_hi, void* feat_lo, const eegclip_split_item* riders, int n_riders, void* stream) {
    rider_table rt;
    if (const int rc = rider_table_from(riders, n_riders, rt)) return rc;
    long long r4 = 0;
    for (int i = 0; i < rt.n; ++i) r4 += rt.it[i].n4;
    int rblocks = (int)((r4 + 2047) / 2048);                 // ~8 float4 per rider thread
    if (rblocks > 512) rblocks = 512;
    if (!y2 || !gamma || !beta || !W || !bias || !z2 || !feat || !feat_hi || !feat_lo || !mean || !rstd || B < 1 || drop_p < 0.f || drop_p >= 1.f) return EEGCLIP_EINVAL;
    if (rows && (nrows < 1 || count < 1.0)) return EEGCLIP_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return EEGCLIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(rows) & 7u) return EEGCLIP_EALIGN;
    // rows == NULL: mean / rstd are INPUTS (eval mode, or statistics finalised by an earlier launch: data parallelism)
    EEG_LAUNCH(proj1x1_fwd_kernel, dim3(B + rblocks), dim3(256), pj_fwd_lds(), stream, y2, rows ? (const float*)nullptr : mean, rows ? (const float*)nullptr : rstd, gamma, beta,
               W, bias, z2, feat, B, drop_p, seed, site, rows, nrows, count, eps, momentum, mean, rstd, running_mean, running_var, num_batches_tracked,
               static_cast<unsigned short*>(feat_hi), static_cast<unsigned short*>(feat_lo), rt);
    return (int)hipGetLastError();
}

extern "C" long long eegclip_proj1x1_bwd_workspace_floats(int B) { return B < 1 ? 0 : 2LL * B * PJ_PART; }      // (doubles, counted in floats)

static int proj1x1_bwd_go(const float* dfeat, int nslabs, long long slab_stride, const float* z2, const float* W, const float* y2, const float* mean,
                          const float* rstd, const float* gamma, const float* beta, float* dz2, float* dW, float* dbias, double* sums, float* workspace, int B,
                          float drop_p, unsigned long long seed, unsigned int site, void* stream);
extern "C" int eegclip_proj1x1_bwd(const float* dfeat, const float* z2, const float* W, const float* y2, const float* mean, const float* rstd,
                                   const float* gamma, const float* beta, float* dz2, float* dW, float* dbias, double* sums, float* workspace, int B,
                                   float drop_p, unsigned long long seed, unsigned int site, void* stream) {
    return proj1x1_bwd_go(dfeat, 1, 0, z2, W, y2, mean, rstd, gamma, beta, dz2, dW, dbias, sums, workspace, B, drop_p, seed, site, stream);
}
// eegclip_proj1x1_bwd whose upstream gradient is the partial slabs of a K-parallel GEMM: dfeat = sum_{s < nslabs} dfeat[s * slab_stride + .]
extern "C" int eegclip_proj1x1_bwd_slabs(const float* dfeat, int nslabs, long long slab_stride, const float* z2, const float* W, const float* y2, const float* mean,
                                         const float* rstd, const float* gamma, const float* beta, float* dz2, float* dW, float* dbias, double* sums,
                                         float* workspace, int B, float drop_p, unsigned long long seed, unsigned int site, void* stream) {
    if (nslabs < 1 || nslabs > 16 || (nslabs > 1 && slab_stride < (long long)B * PJ_N)) return EEGCLIP_EINVAL;
    return proj1x1_bwd_go(dfeat, nslabs, slab_stride, z2, W, y2, mean, rstd, gamma, beta, dz2, dW, dbias, sums, workspace, B, drop_p, seed, site, stream);
}
// the two halves of eegclip_proj1x1_bwd_slabs as launches of their own: _rows leaves one partial row [dW 1600 | dbias 40 | BatchNorm-backward sums 80] (fp64) per
// sample in `workspace` and dz2; eegclip_bn_elu_bwd_apply_rows takes the BatchNorm sums straight from those rows (fixed-order sum in its prologue), so that
// eegclip_proj1x1_bwd_reduce -- dW, dbias (and `sums` unless NULL) -- is read by the optimizer only and can leave the dX chain
extern "C" int eegclip_proj1x1_bwd_rows(const float* dfeat, int nslabs, long long slab_stride, const float* z2, const float* W, const float* y2, const float* mean,
                                        const float* rstd, const float* gamma, const float* beta, float* dz2, float* workspace, double* bn_rows, int B,
                                        float drop_p, unsigned long long seed, unsigned int site, void* stream) {
    if (reinterpret_cast<uintptr_t>(bn_rows) & 7u) return EEGCLIP_EALIGN;
    if (!dfeat || !z2 || !W || !y2 || !mean || !rstd || !gamma || !beta || !dz2 || !workspace || B < 1 || drop_p < 0.f || drop_p >= 1.f) return EEGCLIP_EINVAL;
    if (nslabs < 1 || nslabs > 16 || (nslabs > 1 && slab_stride < (long long)B * PJ_N)) return EEGCLIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(workspace) & 7u) return EEGCLIP_EALIGN;
    const size_t lds = (PJ_N + PJ_W * PJ_LW + PJ_C * PJ_LW + 2 * PJ_N) * sizeof(float);
    EEG_LAUNCH(proj1x1_bwd_kernel, dim3(B), dim3(256), lds, stream, dfeat, z2, W, y2, mean, rstd, gamma, beta, dz2, (float*)nullptr, (float*)nullptr,
               (double*)nullptr, reinterpret_cast<double*>(workspace), B, 1, drop_p, seed, site, nslabs, slab_stride, bn_rows);
    return (int)hipGetLastError();
}
extern "C" int eegclip_proj1x1_bwd_reduce(const float* workspace, int B, float* dW, float* dbias, double* sums, void* stream) {
    if (!workspace || !dW || !dbias || B < 1) return EEGCLIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(workspace) & 7u) return EEGCLIP_EALIGN;
    const int slices = B < PJ_SLICES ? B : PJ_SLICES;
    EEG_LAUNCH(proj1x1_bwd_reduce_kernel, dim3((PJ_PART + 63) / 64, slices), dim3(256), 256 * sizeof(double), stream, reinterpret_cast<const double*>(workspace), B,
               dW, dbias, sums);
    return (int)hipGetLastError();
}
static int proj1x1_bwd_go(const float* dfeat, int nslabs, long long slab_stride, const float* z2, const float* W, const float* y2, const float* mean,
                          const float* rstd, const float* gamma, const float* beta, float* dz2, float* dW, float* dbias, double* sums, float* workspace, int B,
                          float drop_p, unsigned long long seed, unsigned int site, void* stream) {
    if (!dfeat || !z2 || !W || !y2 || !mean || !rstd || !gamma || !beta || !dz2 || !dW || !dbias || !sums || B < 1 || drop_p < 0.f || drop_p >= 1.f)
        return EEGCLIP_EINVAL;
    if (workspace && (reinterpret_cast<uintptr_t>(workspace) & 7u)) return EEGCLIP_EALIGN;
    // samples per workgroup: with a workspace one (the per-sample work is ~11 us of LDS-bound arithmetic: as parallel as possible); with atomics
    // fewer, fatter ones once the grid still fills the chip
    const int spw = workspace ? 1 : (B >= 512 ? 4 : (B >= 128 ? 2 : 1));
    const size_t lds = (PJ_N + PJ_W * PJ_LW + PJ_C * PJ_LW + 2 * PJ_N) * sizeof(float);
    const int nwg = (B + spw - 1) / spw;
    double* parts = reinterpret_cast<double*>(workspace);
    EEG_LAUNCH(proj1x1_bwd_kernel, dim3(nwg), dim3(256), lds, stream, dfeat, z2, W, y2, mean, rstd, gamma, beta, dz2, dW, dbias, sums, parts, B, spw,
               drop_p, seed, site, nslabs, slab_stride, (double*)nullptr);
    if (parts) {
        const int slices = nwg < PJ_SLICES ? nwg : PJ_SLICES;
        EEG_LAUNCH(proj1x1_bwd_reduce_kernel, dim3((PJ_PART + 63) / 64, slices), dim3(256), 256 * sizeof(double), stream, parts, nwg, dW, dbias, sums);
    }
    return (int)hipGetLastError();
}
