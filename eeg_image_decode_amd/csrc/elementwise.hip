// Small streaming kernels around the GEMMs: subject-token insertion + embedding dropout, dropout/GELU backward,
// bias-gradient reductions, fused AdamW.  All are flat grid-stride, coalesced, HBM-bound.
#include "eeg_common.h"

namespace eeg {

static inline int ew_grid(long long n, int block = 256, int cap = 4096) {
    long long g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// h: (B, L, D).  Row 0 of every sample <- subject token (table[ids[b]] or the shared token), then inverted dropout
// over the whole (B,L,D) block in place.            (models/subject_layers/Embed.py:116-121,158-162)
// Both embedding kernels walk the (B, L, D) tensor in groups of 4 consecutive elements (n % 4 == 0 required by the launcher): one
// 16-byte access and ONE Philox block per group instead of four of each.
__global__ __launch_bounds__(256) void embed_finish_kernel(float* __restrict__ h, const float* __restrict__ tokens,
                                                            const long long* __restrict__ ids, int B, int L, int D, float drop_p,
                                                            unsigned long long seed, unsigned site) {
    const long long n4 = (long long)B * L * D / 4;
    const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (long long)gridDim.x * blockDim.x) {
        const long long i0 = 4 * q;
        f32x4 v = *reinterpret_cast<const f32x4*>(h + i0);
        bool keep[4] = {true, true, true, true};
        if (drop_p > 0.f) dropout_keep4(seed, site, (unsigned long long)i0, drop_p, keep);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long long i = i0 + e;
            const int l = (int)((i / D) % L);
            if (l == 0) {
                const int b = (int)(i / ((long long)D * L));
                v[e] = tokens[(ids ? ids[b] : 0) * D + (int)(i % D)];
            }
            v[e] = keep[e] ? v[e] * ks : 0.f;
        }
        *reinterpret_cast<f32x4*>(h + i0) = v;
    }
}

// backward of the above: dh *= mask/(1-p) in place; dtokens[id] += sum_b dh[b,0,:]
// The first `tok_blocks` workgroups own the token rows (l = 0) exclusively -- scale them, and sum them over the batch in registers with ONE
// atomic per (workgroup, column) on the shared-token path (ids == NULL); the streaming workgroups skip those rows.  (Per-element atomics
// from the streaming loop put 256 same-address atomics on each of the 250 columns in the middle of a bandwidth kernel: 36 us for 32 MB.)
__global__ __launch_bounds__(256) void embed_finish_bwd_kernel(float* __restrict__ dh, float* __restrict__ dtokens,
                                                                const long long* __restrict__ ids, int B, int L, int D, float drop_p,
                                                                unsigned long long seed, unsigned site, int tok_blocks) {
    const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    if ((int)blockIdx.x < tok_blocks) {
        // thread -> column d (and a slice of the batch): rows b = blockIdx.x, blockIdx.x + tok_blocks, ...
        for (int d = threadIdx.x; d < D; d += blockDim.x) {
            float acc = 0.f;
            long long row = 0;                                // token row the running sum belongs to (ids == NULL: the shared token, row 0)
            // 8 batch rows at a time: all their loads are issued before the first use (the rows are L*D floats apart: one memory latency per
            // element when loaded one by one -- this serial chain, not the streaming part, was the kernel's 31 us)
            for (int b0 = blockIdx.x; b0 < B; b0 += 8 * tok_blocks) {
                float v8[8];
                long long r8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int b = b0 + u * tok_blocks;
                    v8[u] = b < B ? dh[(long long)b * L * D + d] : 0.f;
                    r8[u] = (b < B && ids) ? ids[b] : 0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int b = b0 + u * tok_blocks;
                    if (b >= B) break;
                    const long long i = (long long)b * L * D + d;
                    float v = v8[u];
                    if (drop_p > 0.f) {
                        v = dropout_keep(seed, site, (unsigned long long)i, drop_p) ? v * ks : 0.f;
                        dh[i] = v;
                    }
                    if (dtokens) {
                        // runs of equal ids are summed in a register and flushed with ONE atomic (a single-subject batch -- the reference's
                        // loops -- is one run: 32 x 250 atomics instead of 256-way contention on each of 250 addresses)
                        const long long r = r8[u];
                        if (r != row) {
                            if (acc != 0.f) atomicAdd(dtokens + row * D + d, acc);
                            acc = 0.f;
                            row = r;
                        }
                        acc += v;
                    }
                }
            }
            if (dtokens && acc != 0.f) atomicAdd(dtokens + row * D + d, acc);
        }
        return;
    }
    if (!(drop_p > 0.f)) return;                              // nothing to scale: the token blocks did the reduction
    const long long n4 = (long long)B * L * D / 4;
    const long long stride = (long long)(gridDim.x - tok_blocks) * blockDim.x;
    for (long long q = (long long)(blockIdx.x - tok_blocks) * blockDim.x + threadIdx.x; q < n4; q += stride) {
        const long long i0 = 4 * q;
        f32x4 v = *reinterpret_cast<const f32x4*>(dh + i0);
        bool keep[4];
        dropout_keep4(seed, site, (unsigned long long)i0, drop_p, keep);
        bool tok[4];
        bool any_tok = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            tok[e] = (int)(((i0 + e) / D) % L) == 0;          // token-row elements belong to the token blocks
            any_tok |= tok[e];
            v[e] = keep[e] ? v[e] * ks : 0.f;
        }
        if (!any_tok) {
            *reinterpret_cast<f32x4*>(dh + i0) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (!tok[e]) dh[i0 + e] = v[e];
        }
    }
}

__global__ __launch_bounds__(256) void dropout_scale_kernel(float* __restrict__ x, long long n, float drop_p,
                                                             unsigned long long seed, unsigned site) {
    const float ks = 1.f / (1.f - drop_p);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        x[i] = dropout_keep(seed, site, (unsigned long long)i, drop_p) ? x[i] * ks : 0.f;
}

// dx (+)= dy * mask/(1-p) * gelu'(pre)      (backward of y = dropout(gelu(pre)); p = 0 -> plain GELU backward)
// ---- the projection head's elementwise stages behind its K-parallel GEMMs (csrc/head_gemm.hip): they ADD the GEMM's partial slabs (slice order), so the
// split costs no launch of its own.  One lane = 4 consecutive columns (16-byte accesses; N % 4 == 0).
//   head_act      u = sum_s slab[s] + bias;  pre = u;  out = gelu(u);  (out_hi | out_lo) = out again as bf16 planes     (ATMS_retrieval.py:160-162)
//   head_act_bwd  dx = base + (sum_s slab[s]) * gelu'(pre)  (base may be dx: the residual branch's gradient), + planes   (its backward)
__device__ __forceinline__ f32x4 slab_sum4(const float* __restrict__ slabs, int nslabs, long long stride, long long i) {
    return slab_sum4_inflight(slabs, nslabs < 1 ? 1 : nslabs, stride, i);
}
__global__ __launch_bounds__(256) void head_act_kernel(const float* __restrict__ slabs, int nslabs, long long stride, const float* __restrict__ bias,
                                                        float* __restrict__ pre, float* __restrict__ out, unsigned short* __restrict__ out_hi,
                                                        unsigned short* __restrict__ out_lo, long long n4, int N) {
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (long long)gridDim.x * blockDim.x) {
        const long long i = 4 * q;
        f32x4 v = slab_sum4(slabs, nslabs, stride, i);
        if (bias) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bias + (int)(i % N));
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += b[e];
        }
        if (pre) *reinterpret_cast<f32x4*>(pre + i) = v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        if (out) *reinterpret_cast<f32x4*>(out + i) = v;
        if (out_hi) {
            u32x2_t hi, lo;
            x3_split4(v[0], v[1], v[2], v[3], hi, lo);
            *reinterpret_cast<u32x2_t*>(out_hi + i) = hi;
            *reinterpret_cast<u32x2_t*>(out_lo + i) = lo;
        }
    }
}
__global__ __launch_bounds__(256) void head_act_bwd_kernel(const float* __restrict__ slabs, int nslabs, long long stride, const float* __restrict__ pre,
                                                            const float* __restrict__ base, float* __restrict__ dx, unsigned short* __restrict__ dx_hi,
                                                            unsigned short* __restrict__ dx_lo, long long n4) {
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (long long)gridDim.x * blockDim.x) {
        const long long i = 4 * q;
        f32x4 v = slab_sum4(slabs, nslabs, stride, i);
        const f32x4 u = *reinterpret_cast<const f32x4*>(pre + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= gelu_erf_grad(u[e]);
        if (base) {
            const f32x4 r = *reinterpret_cast<const f32x4*>(base + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = r[e] + v[e];
        }
        if (dx) *reinterpret_cast<f32x4*>(dx + i) = v;
        if (dx_hi) {
            u32x2_t hi, lo;
            x3_split4(v[0], v[1], v[2], v[3], hi, lo);
            *reinterpret_cast<u32x2_t*>(dx_hi + i) = hi;
            *reinterpret_cast<u32x2_t*>(dx_lo + i) = lo;
        }
    }
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pre,
                                                        float* __restrict__ dx, long long n, int accumulate, float drop_p,
                                                        unsigned long long seed, unsigned site) {
    const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float d = dy[i];
        if (drop_p > 0.f) d = dropout_keep(seed, site, (unsigned long long)i, drop_p) ? d * ks : 0.f;
        const float v = d * gelu_erf_grad(pre[i]);
        dx[i] = accumulate ? dx[i] + v : v;
    }
}

// epilogue of a split-K GEMM whose raw product sits in `acc` (M, N) row-major:
//   v = acc + bias_n ; pre = v (optional) ; v = act(v) ; v = dropout(v) ; v += resid ; out = v       (same order as the fused GEMM epilogue)
__global__ __launch_bounds__(256) void bias_act_kernel(const float* __restrict__ acc, const float* __restrict__ bias_n, float* __restrict__ pre,
                                                        const float* __restrict__ resid, float* __restrict__ out, long long n, int N, int act,
                                                        float drop_p, unsigned long long seed, unsigned site) {
    const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = acc[i];
        if (bias_n) v += bias_n[i % N];
        if (pre) pre[i] = v;
        if (act == EEGCLIP_ACT_GELU) v = gelu_erf(v);
        else if (act == EEGCLIP_ACT_SILU) v = silu(v);
        if (drop_p > 0.f) v = dropout_keep(seed, site, (unsigned long long)i, drop_p) ? v * ks : 0.f;
        if (resid) v += resid[i];
        out[i] = v;
    }
}

// y = a*x + b*y
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, float a,
                                                     float b) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = a * x[i] + (b != 0.f ? b * y[i] : 0.f);
}

// out[m] += sum_{o,i} x[o][m][i]   over an (outer, mid, inner) view   (bias gradients)
// grid (chunks, ceil(mid/64)); block 256 = 4 row-groups x 64 consecutive `mid` (inner == 1) or flat (inner > 1)
__global__ __launch_bounds__(256) void reduce_mid_kernel(const float* __restrict__ x, int outer, int mid, int inner,
                                                          float* __restrict__ out) {
    EEG_LDS_BASE(float, red);   // [4][64]
    if (inner == 1) {
        const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
        const int m = blockIdx.y * 64 + lane;
        float s = 0.f;
        if (m < mid)
            for (int o = blockIdx.x * 4 + g; o < outer; o += gridDim.x * 4) s += x[(long long)o * mid + m];
        red[g * 64 + lane] = s;
        __syncthreads();
        if (g == 0 && m < mid) atomicAdd(out + m, red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane]);
    } else {
        // one `mid` index per blockIdx.y*64 + wave-slot is wasteful for tiny tensors, but these are (B,40,36)-sized
        for (int mm = 0; mm < 64; ++mm) {
            const int m = blockIdx.y * 64 + mm;
            if (m >= mid) break;
            float s = 0.f;
            for (int o = blockIdx.x; o < outer; o += gridDim.x) {
                const float* p = x + ((long long)o * mid + m) * inner;
                for (int i = threadIdx.x; i < inner; i += blockDim.x) s += p[i];
            }
            s = wave_sum(s);
            if ((threadIdx.x & 63) == 0 && s != 0.f) atomicAdd(out + m, s);
        }
    }
}

// out[c] += sum_{blk} sum_{r=row0}^{blk_rows-1} x[blk*blk_stride + r*cols + c]     (value-embedding bias gradient: every
// row of each (64,250) token block except the subject-token row 0).  grid (chunks, ceil(cols/64)), block 4 x 64.
__global__ __launch_bounds__(256) void colsum_blocks_kernel(const float* __restrict__ x, int nblk, int blk_rows, int row0, int cols,
                                                             long long blk_stride, float* __restrict__ out) {
    EEG_LDS_BASE(float, red);   // [4][64]
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    const int per = blk_rows - row0;
    const long long total = (long long)nblk * per;
    float s = 0.f;
    if (c < cols)
        for (long long o = blockIdx.x * 4 + g; o < total; o += (long long)gridDim.x * 4) {
            const long long blk = o / per;
            const int r = row0 + (int)(o % per);
            s += x[blk * blk_stride + (long long)r * cols + c];
        }
    red[g * 64 + lane] = s;
    __syncthreads();
    if (g == 0 && c < cols) atomicAdd(out + c, red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane]);
}

// torch.optim.AdamW / Adam single-step math on a flat fp32 segment (decoupled weight decay, bias correction).
// bc1 = 1 - beta1^t, bc2s = sqrt(1 - beta2^t) are computed on the host in double.
template <bool ZERO>       // ZERO: the gradient is cleared behind the read (optimizer.step() + zero_grad() in one pass over g)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                     float wd, float bc1, float bc2s, float grad_scale,
                                                     const float* __restrict__ grad_scale_dev) {
    const float step = lr / bc1;
    if (grad_scale_dev) grad_scale *= *grad_scale_dev;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * grad_scale;
        if (ZERO) g[i] = 0.f;
        adamw_element(p + i, m + i, v + i, gi, lr, b1, b2, eps, wd, step, bc2s);
    }
}

// sum of squares of a flat fp32 buffer into a double accumulator (clip_grad_norm_).  ONE atomic per workgroup: same-address atomics retire one after
// the other at ~12 ns each -- one per wave on 1024 workgroups was 4096 of them, 49 of this kernel's 60 us on the prior's 13.6 M gradients (round 4).
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long long n, double* __restrict__ out) {
    EEG_LDS_BASE(double, red);                               // [4 waves]
    double s = 0.0;
    const long long n4 = ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) ? n / 4 : 0;      // 16-byte loads over the aligned bulk
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (long long)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * q);
        s += ((double)v[0] * v[0] + (double)v[1] * v[1]) + ((double)v[2] * v[2] + (double)v[3] * v[3]);
    }
    for (long long i = 4 * n4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) s += (double)x[i] * x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}

__global__ void clip_scale_kernel(const double* __restrict__ sumsq, float max_norm, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float nrm = (float)sqrt(*sumsq);
        const float c = max_norm / (nrm + 1e-6f);
        *out = c < 1.f ? c : 1.f;
    }
}

// sample-block gather / scatter for the joint-subject value embedding (Embed.py:142-144 picks a Linear per sample; the product sorts the
// batch by subject so that every subject is one GEMM over a contiguous block).  One "row" is a sample's `row_floats` contiguous floats:
//   scatter == 0:  dst[j*dst_stride + e] = src[idx[j]*src_stride + e]        scatter != 0:  dst[idx[j]*dst_stride + e] = src[j*src_stride + e]
__global__ __launch_bounds__(256) void gather_rows_kernel(float* __restrict__ dst, long long dst_stride, const float* __restrict__ src,
                                                          long long src_stride, const int* __restrict__ idx, int n, int row2, int scatter) {
    const long long total = (long long)n * row2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int j = (int)(i / row2), e = (int)(i % row2);       // 8-byte pieces: a 63 x 250 sample is not a multiple of 16 bytes
        const int k = idx[j];
        const long long so = (scatter ? (long long)j : (long long)k) * src_stride, dd = (scatter ? (long long)k : (long long)j) * dst_stride;
        reinterpret_cast<float2*>(dst + dd)[e] = reinterpret_cast<const float2*>(src + so)[e];
    }
}

// Dataset staging (Retrieval/eegdatasets_leaveone.py:157,203 `.float()`, :220 mean over repetitions, :293-306 time-window mask): the
// reference's on-disk trials are float64 (images, repetitions, channels, T); one pass turns a chunk of them into the float32 training
// layout in HBM.  row = (image, repetition, channel) [or (image, channel) when averaging]; a thread owns one output sample, so a wave
// reads Tw consecutive doubles of a row (the mask of a contiguous window is a contiguous index list).
//   mean_reps == 0:  dst[(i*reps + r)*C + c][j] = (float) src[i][r][c][tidx[j]]
//   mean_reps != 0:  dst[i*C + c][j] = (sum_r (float) src[i][r][c][tidx[j]]) / reps         float32 accumulation in repetition order
__global__ __launch_bounds__(256) void stage_eeg_kernel(const double* __restrict__ src, float* __restrict__ dst, long long n_items, int reps,
                                                        int C, int T, const int* __restrict__ tidx, int Tw, int mean_reps) {
    const long long rows = mean_reps ? n_items * C : n_items * reps * C;
    const long long total = rows * Tw;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long row = e / Tw;
        const int t = tidx ? tidx[(int)(e % Tw)] : (int)(e % Tw);
        if (!mean_reps) {
            dst[e] = (float)src[row * T + t];
        } else {
            const long long i = row / C;
            const int c = (int)(row % C);
            float acc = 0.f;
            for (int r = 0; r < reps; ++r) acc += (float)src[((i * reps + r) * C + c) * T + t];
            dst[e] = acc / (float)reps;
        }
    }
}

// whole-window cast (tidx == NULL: every stored sample is kept -- the real data's [0, 1] s window): a flat float64 -> float32 stream, four values
// per thread (two 16-byte loads, one 16-byte store)
__global__ __launch_bounds__(256) void stage_eeg_flat_kernel(const double* __restrict__ src, float* __restrict__ dst, long long n4, long long n) {
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) {
        const f64x2 a = reinterpret_cast<const f64x2*>(src)[2 * q], b = reinterpret_cast<const f64x2*>(src)[2 * q + 1];
        reinterpret_cast<float4*>(dst)[q] = make_float4((float)a[0], (float)a[1], (float)b[0], (float)b[1]);
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) dst[4 * n4 + threadIdx.x] = (float)src[4 * n4 + threadIdx.x];
}

}  // namespace eeg

using namespace eeg;

extern "C" int eegclip_stage_eeg(const double* src, float* dst, long long n_items, int reps, int channels, int T, const int* tidx, int Tw,
                                 int mean_reps, void* stream) {
    if (!src || !dst || n_items < 1 || reps < 1 || channels < 1 || T < 1 || Tw < 1 || Tw > T || (!tidx && Tw != T)) return EEGCLIP_EINVAL;
    const long long total = (mean_reps ? n_items * channels : n_items * reps * channels) * Tw;
    if (!tidx && !mean_reps && !((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u)) {
        EEG_LAUNCH(stage_eeg_flat_kernel, dim3(ew_grid(total / 4, 256, 16384)), dim3(256), 0, stream, src, dst, total / 4, total);
        return (int)hipGetLastError();
    }
    EEG_LAUNCH(stage_eeg_kernel, dim3(ew_grid(total, 256, 16384)), dim3(256), 0, stream, src, dst, n_items, reps, channels, T, tidx, Tw, mean_reps);
    return (int)hipGetLastError();
}

extern "C" int eegclip_gather_rows(float* dst, long long dst_stride, const float* src, long long src_stride, const int* idx, int n,
                                   int row_floats, int scatter, void* stream) {
    if (!dst || !src || !idx || n < 1 || row_floats < 2 || (row_floats & 1) || (dst_stride & 1) || (src_stride & 1)) return EEGCLIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 7u) return EEGCLIP_EALIGN;
    EEG_LAUNCH(gather_rows_kernel, dim3(ew_grid((long long)n * (row_floats / 2))), dim3(256), 0, stream, dst, dst_stride, src, src_stride, idx, n,
               row_floats / 2, scatter);
    return (int)hipGetLastError();
}

extern "C" int eegclip_clip_scale(const double* sumsq, float max_norm, float* scale_out, void* stream) {
    if (!sumsq || !scale_out || max_norm <= 0.f) return EEGCLIP_EINVAL;
    EEG_LAUNCH(clip_scale_kernel, dim3(1), dim3(64), 0, stream, sumsq, max_norm, scale_out);
    return (int)hipGetLastError();
}


extern "C" int eegclip_embed_finish(float* h, const float* tokens, const long long* ids, int B, int L, int D, float drop_p,
                                    unsigned long long seed, unsigned site, void* stream) {
    if (!h || !tokens || B < 1 || L < 1 || D < 1 || drop_p < 0.f || drop_p >= 1.f) return EEGCLIP_EINVAL;
    if (((long long)L * D) % 4 != 0) return EEGCLIP_EINVAL;                 // 4-element groups never straddle the end of the tensor
    if (reinterpret_cast<uintptr_t>(h) & 15u) return EEGCLIP_EALIGN;
    EEG_LAUNCH(embed_finish_kernel, dim3(ew_grid((long long)B * L * D / 4)), dim3(256), 0, stream, h, tokens, ids, B, L, D, drop_p, seed,
               site);
    return (int)hipGetLastError();
}

extern "C" int eegclip_embed_finish_bwd(float* dh, float* dtokens, const long long* ids, int B, int L, int D, float drop_p,
                                        unsigned long long seed, unsigned site, void* stream) {
    if (!dh || B < 1 || L < 1 || D < 1 || drop_p < 0.f || drop_p >= 1.f) return EEGCLIP_EINVAL;
    if (((long long)L * D) % 4 != 0) return EEGCLIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(dh) & 15u) return EEGCLIP_EALIGN;
    const int tok_blocks = B < 32 ? B : 32;
    EEG_LAUNCH(embed_finish_bwd_kernel, dim3(tok_blocks + ew_grid((long long)B * L * D / 4)), dim3(256), 0, stream, dh, dtokens, ids, B, L, D,
               drop_p, seed, site, tok_blocks);
    return (int)hipGetLastError();
}

extern "C" int eegclip_dropout_scale(float* x, long long n, float drop_p, unsigned long long seed, unsigned site, void* stream) {
    if (!x || n < 0 || drop_p < 0.f || drop_p >= 1.f) return EEGCLIP_EINVAL;
    if (n == 0 || drop_p == 0.f) return 0;
    EEG_LAUNCH(dropout_scale_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, x, n, drop_p, seed, site);
    return (int)hipGetLastError();
}

extern "C" int eegclip_head_act(const float* slabs, int nslabs, long long slab_stride, const float* bias, float* pre, float* out, void* out_hi, void* out_lo,
                                int M, int N, void* stream) {
    if (!slabs || nslabs < 1 || nslabs > 16 || M < 1 || N < 4 || (N & 3) || (nslabs > 1 && (slab_stride < (long long)M * N || (slab_stride & 3)))) return EEGCLIP_EINVAL;
    if ((!pre && !out && !out_hi) || ((out_hi != nullptr) != (out_lo != nullptr))) return EEGCLIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(pre) | reinterpret_cast<uintptr_t>(out)) & 15u) return EEGCLIP_EALIGN;
    if ((reinterpret_cast<uintptr_t>(out_hi) | reinterpret_cast<uintptr_t>(out_lo)) & 7u) return EEGCLIP_EALIGN;
    const long long n4 = (long long)M * N / 4;
    EEG_LAUNCH(head_act_kernel, dim3(ew_grid(n4)), dim3(256), 0, stream, slabs, nslabs, slab_stride, bias, pre, out, static_cast<unsigned short*>(out_hi),
               static_cast<unsigned short*>(out_lo), n4, N);
    return (int)hipGetLastError();
}
extern "C" int eegclip_head_act_bwd(const float* slabs, int nslabs, long long slab_stride, const float* pre, const float* base, float* dx, void* dx_hi, void* dx_lo,
                                    long long n, void* stream) {
    if (!slabs || !pre || nslabs < 1 || nslabs > 16 || n < 4 || (n & 3) || (nslabs > 1 && (slab_stride < n || (slab_stride & 3)))) return EEGCLIP_EINVAL;
    if ((!dx && !dx_hi) || ((dx_hi != nullptr) != (dx_lo != nullptr))) return EEGCLIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(pre) | reinterpret_cast<uintptr_t>(base) | reinterpret_cast<uintptr_t>(dx)) & 15u) return EEGCLIP_EALIGN;
    if ((reinterpret_cast<uintptr_t>(dx_hi) | reinterpret_cast<uintptr_t>(dx_lo)) & 7u) return EEGCLIP_EALIGN;
    EEG_LAUNCH(head_act_bwd_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, stream, slabs, nslabs, slab_stride, pre, base, dx, static_cast<unsigned short*>(dx_hi),
               static_cast<unsigned short*>(dx_lo), n / 4);
    return (int)hipGetLastError();
}
extern "C" int eegclip_gelu_bwd(const float* dy, const float* pre, float* dx, long long n, int accumulate, float drop_p,
                                unsigned long long seed, unsigned site, void* stream) {
    if (!dy || !pre || !dx || n < 0 || drop_p < 0.f || drop_p >= 1.f) return EEGCLIP_EINVAL;
    if (n == 0) return 0;
    EEG_LAUNCH(gelu_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, dy, pre, dx, n, accumulate, drop_p, seed, site);
    return (int)hipGetLastError();
}

extern "C" int eegclip_bias_act(const float* acc, const float* bias_n, float* pre, const float* resid, float* out, int M, int N, int act,
                                float drop_p, unsigned long long seed, unsigned site, void* stream) {
    if (!acc || !out || M < 1 || N < 1 || act < 0 || act > EEGCLIP_ACT_SILU || drop_p < 0.f || drop_p >= 1.f) return EEGCLIP_EINVAL;
    const long long n = (long long)M * N;
    EEG_LAUNCH(bias_act_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, acc, bias_n, pre, resid, out, n, N, act, drop_p, seed, site);
    return (int)hipGetLastError();
}

extern "C" int eegclip_axpby(const float* x, float* y, long long n, float a, float b, void* stream) {
    if (!x || !y || n < 0) return EEGCLIP_EINVAL;
    if (n == 0) return 0;
    EEG_LAUNCH(axpby_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, x, y, n, a, b);
    return (int)hipGetLastError();
}

extern "C" int eegclip_reduce_mid(const float* x, int outer, int mid, int inner, float* out, void* stream) {
    if (!x || !out || outer < 1 || mid < 1 || inner < 1) return EEGCLIP_EINVAL;
    int chunks = inner == 1 ? (outer + 3) / 4 : outer;
    if (chunks > 128) chunks = 128;
    EEG_LAUNCH(reduce_mid_kernel, dim3(chunks, (mid + 63) / 64), dim3(256), 4 * 64 * sizeof(float), stream, x, outer, mid, inner, out);
    return (int)hipGetLastError();
}

extern "C" int eegclip_colsum_blocks(const float* x, int nblk, int blk_rows, int row0, int cols, long long blk_stride, float* out,
                                     void* stream) {
    if (!x || !out || nblk < 1 || blk_rows < 1 || row0 < 0 || row0 >= blk_rows || cols < 1) return EEGCLIP_EINVAL;
    long long total = (long long)nblk * (blk_rows - row0);
    long long chunks = (total + 3) / 4;
    if (chunks > 128) chunks = 128;
    EEG_LAUNCH(colsum_blocks_kernel, dim3((int)chunks, (cols + 63) / 64), dim3(256), 4 * 64 * sizeof(float), stream, x, nblk, blk_rows,
               row0, cols, blk_stride, out);
    return (int)hipGetLastError();
}

extern "C" int eegclip_adamw_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                                  float eps, float weight_decay, long long step, float grad_scale, const float* grad_scale_dev,
                                  void* stream) {
    if (!p || !g || !m || !v || n < 0 || step < 1) return EEGCLIP_EINVAL;
    if (n == 0) return 0;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    EEG_LAUNCH(adamw_kernel<false>, dim3(ew_grid(n, 256, 2048)), dim3(256), 0, stream, p, const_cast<float*>(g), m, v, n, lr, beta1, beta2, eps,
               weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale, grad_scale_dev);
    return (int)hipGetLastError();
}

extern "C" int eegclip_adamw_step_zero_grad(float* p, float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                                            float weight_decay, long long step, float grad_scale, const float* grad_scale_dev, void* stream) {
    if (!p || !g || !m || !v || n < 0 || step < 1) return EEGCLIP_EINVAL;
    if (n == 0) return 0;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    EEG_LAUNCH(adamw_kernel<true>, dim3(ew_grid(n, 256, 2048)), dim3(256), 0, stream, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
               (float)bc1, (float)sqrt(bc2), grad_scale, grad_scale_dev);
    return (int)hipGetLastError();
}

extern "C" int eegclip_sumsq(const float* x, long long n, double* out, void* stream) {
    if (!x || !out || n < 0) return EEGCLIP_EINVAL;
    if (n == 0) return 0;
    EEG_LAUNCH(sumsq_kernel, dim3(ew_grid(n, 1024, 256)), dim3(256), 4 * sizeof(double), stream, x, n, out);
    return (int)hipGetLastError();
}
