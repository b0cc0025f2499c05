// CLIP-symmetric InfoNCE (models/loss.py:100-141) around the logits GEMM, plus the retrieval readouts
// (Retrieval/ATMS_retrieval.py:241-250 argmax, :320 top-5).
//
//   raw = A B^T (GEMM) ; S = s*raw (s = RAW logit_scale, read from device memory -- no host sync)
//   L = w/(2N) * sum_i [ LSE_j S_ij + LSE_j S_ji - 2 S_ii ]
//   G = w/(2N) * (softmax_rows(S) + softmax_cols(S) - 2I) ;  dA = (s G) B ; ds = sum G .* raw
// The grad kernel overwrites raw with s*G in place so the dA GEMM needs no host-side scalar.
// Row-sharded form (local_loss): the row block of rank r is n x N with the positives at column i + col0.
#include "eeg_common.h"

namespace eeg {

// one wave per row: lse[i] = log sum_j exp(s * X[i][j])
__global__ __launch_bounds__(256) void lse_rows_kernel(const float* __restrict__ X, int rows, int cols, long long ld,
                                                        const float* __restrict__ scale, float* __restrict__ lse) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float s = scale ? *scale : 1.f;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float* xr = X + row * ld;
        float mx = -INFINITY;
        for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, s * xr[c]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int c = lane; c < cols; c += 64) sum += expf(s * xr[c] - mx);
        sum = wave_sum(sum);
        if (lane == 0) lse[row] = mx + logf(sum);
    }
}

// lse[j] = log sum_i exp(s * X[i][j]);  block = 16 columns x 16 row groups (64-byte row segments), online (max,sum) per thread, LDS
// merge.  (64 columns x 4 row groups left 4 workgroups walking 64 dependent exp steps at N = 256: 24 us for a 256 KB matrix.)
__global__ __launch_bounds__(256) void lse_cols_kernel(const float* __restrict__ X, int rows, int cols, long long ld,
                                                        const float* __restrict__ scale, float* __restrict__ lse) {
    EEG_LDS_BASE(float, red);   // [2][16][16]
    const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const float s = scale ? *scale : 1.f;
    float mx = -INFINITY, sum = 0.f;
    if (c < cols) {
#pragma unroll 4
        for (int r = g; r < rows; r += 16) {
            const float v = s * X[r * ld + c];
            if (v > mx) { sum = sum * expf(mx - v) + 1.f; mx = v; }
            else sum += expf(v - mx);
        }
    }
    red[g * 16 + cl] = mx;
    red[256 + g * 16 + cl] = sum;
    __syncthreads();
    if (g == 0 && c < cols) {
        float M = red[cl];
        for (int k = 1; k < 16; ++k) M = fmaxf(M, red[k * 16 + cl]);
        float S = 0.f;
        for (int k = 0; k < 16; ++k) {
            const float mk = red[k * 16 + cl];
            if (mk > -INFINITY) S += red[256 + k * 16 + cl] * expf(mk - M);
        }
        lse[c] = M + logf(S);
    }
}

// in place: X[i][j] <- s * G_ij ;  *loss += w/(2N)*sum_i(lse_r[i] + lse_c[i+col0] - 2 S_i,i+col0) ; *dscale += sum G .* raw
// rows x cols block; `n_total` is N in the 1/(2N) factor; lse_c may be NULL (row term only) and lse_r may be NULL (col term only)
__global__ __launch_bounds__(256) void infonce_grad_kernel(float* __restrict__ X, int rows, int cols, long long ld, int col0,
                                                            int n_total, const float* __restrict__ scale,
                                                            const float* __restrict__ lse_r, const float* __restrict__ lse_c, float w,
                                                            float* __restrict__ loss, float* __restrict__ dscale) {
    const float s = *scale;
    const float k = w / (2.0f * (float)n_total);
    float lsum = 0.f, dsum = 0.f;
    const long long n = (long long)rows * cols;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(idx / cols), j = (int)(idx % cols);
        const float raw = X[i * ld + j];
        const float v = s * raw;
        float g = 0.f;
        if (lse_r) g += expf(v - lse_r[i]);
        if (lse_c) g += expf(v - lse_c[j]);
        if (j == i + col0) {
            const float nterm = (lse_r ? 1.f : 0.f) + (lse_c ? 1.f : 0.f);
            g -= nterm;
            lsum += (lse_r ? lse_r[i] : 0.f) + (lse_c ? lse_c[j] : 0.f) - nterm * v;
        }
        g *= k;
        dsum += g * raw;
        X[i * ld + j] = s * g;
    }
    // one atomic pair per WORKGROUP: a pair per wave (1024 waves at N = 256) serialised 2048 same-address atomics in L2
    EEG_LDS_BASE(float, red);   // [2][4]
    lsum = wave_sum(lsum);
    dsum = wave_sum(dsum);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = lsum; red[4 + (threadIdx.x >> 6)] = dsum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float l4 = (red[0] + red[1]) + (red[2] + red[3]), d4 = (red[4] + red[5]) + (red[6] + red[7]);
        if (loss && l4 != 0.f) atomicAdd(loss, k * l4);
        if (dscale) atomicAdd(dscale, d4);
    }
}

// loss only (eval): *loss += w/(2N) * sum_i (lse_r[i] + lse_c[i] - 2 s raw_ii)
__global__ void infonce_loss_kernel(const float* __restrict__ X, int n, long long ld, const float* __restrict__ scale,
                                    const float* __restrict__ lse_r, const float* __restrict__ lse_c, float w, float* __restrict__ loss) {
    const float s = *scale;
    float lsum = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        lsum += lse_r[i] + lse_c[i] - 2.f * s * X[i * ld + i];
    lsum = wave_sum(lsum);
    if ((threadIdx.x & 63) == 0) atomicAdd(loss, lsum * w / (2.0f * (float)n));
}

// top-k (k <= 8) per row, one wave per row, ties -> lowest index; k == 1 is argmax.  Optional row-gather `sel`
// (candidate subset, ATMS_retrieval.py:299-305): logical column j reads X[row][sel[j]].
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ X, int rows, int cols, long long ld, int k,
                                                         const float* __restrict__ scale, long long* __restrict__ out_idx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float sgn = (scale && *scale < 0.f) ? -1.f : 1.f;     // ranking by scale*x without a host-side sign check
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float* xr = X + row * ld;
        int taken[8];
        for (int t = 0; t < k; ++t) {
            float best = -INFINITY;
            int bi = 0x7fffffff;
            for (int c = lane; c < cols; c += 64) {
                bool skip = false;
                for (int u = 0; u < t; ++u) skip |= (taken[u] == c);
                const float v = sgn * xr[c];
                if (!skip && (v > best || (v == best && c < bi) || bi == 0x7fffffff)) { best = v; bi = c; }
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const float ob = __shfl_xor(best, m, 64);
                const int oi = __shfl_xor(bi, m, 64);
                if (oi != 0x7fffffff && (bi == 0x7fffffff || ob > best || (ob == best && oi < bi))) { best = ob; bi = oi; }
            }
            taken[t] = bi;
            if (lane == 0) out_idx[(long long)row * k + t] = bi;
        }
    }
}

// *count += #{i : pred[i*stride] == labels[i]}
__global__ void count_equal_kernel(const long long* __restrict__ pred, int stride, const long long* __restrict__ labels, int n,
                                   int* __restrict__ count) {
    int c = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) c += (pred[(long long)i * stride] == labels[i]);
    c = (int)wave_sum((float)c);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

// *count += #{rows whose arg-max column (ties -> lowest index, ranking by scale * x) equals labels[row]}: the running train accuracy of the batch loop
// (ATMS_retrieval.py:241-250) as ONE launch -- one wave per row, eight independent loads in flight per lane -- instead of a top-k kernel (14 us at
// 256 x 1654) + a comparison kernel
__global__ __launch_bounds__(256) void top1_count_kernel(const float* __restrict__ X, int rows, int cols, long long ld, const float* __restrict__ scale,
                                                          const long long* __restrict__ labels, int* __restrict__ count) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float sgn = (scale && *scale < 0.f) ? -1.f : 1.f;
    int hits = 0;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float* xr = X + row * ld;
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int c0 = lane; c0 < cols; c0 += 8 * 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = c0 + 64 * u < cols ? sgn * xr[c0 + 64 * u] : -INFINITY;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + 64 * u;
                if (c < cols && (v[u] > best || bi == 0x7fffffff)) { best = v[u]; bi = c; }        // (a lane walks its columns in increasing order: ties keep the lowest)
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float ob = __shfl_xor(best, m, 64);
            const int oi = __shfl_xor(bi, m, 64);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ob > best || (ob == best && oi < bi))) { best = ob; bi = oi; }
        }
        hits += (long long)bi == labels[row];
    }
    if (lane == 0 && hits) atomicAdd(count, hits);
}

}  // namespace eeg

using namespace eeg;

extern "C" int eegclip_lse_rows(const float* X, int rows, int cols, long long ld, const float* scale, float* lse, void* stream) {
    if (!X || !lse || rows < 1 || cols < 1 || ld < cols) return EEGCLIP_EINVAL;
    int grid = (rows + 3) / 4;
    if (grid > 2048) grid = 2048;
    EEG_LAUNCH(lse_rows_kernel, dim3(grid), dim3(256), 0, stream, X, rows, cols, ld, scale, lse);
    return (int)hipGetLastError();
}
extern "C" int eegclip_lse_cols(const float* X, int rows, int cols, long long ld, const float* scale, float* lse, void* stream) {
    if (!X || !lse || rows < 1 || cols < 1 || ld < cols) return EEGCLIP_EINVAL;
    EEG_LAUNCH(lse_cols_kernel, dim3((cols + 15) / 16), dim3(256), 512 * sizeof(float), stream, X, rows, cols, ld, scale, lse);
    return (int)hipGetLastError();
}
extern "C" int eegclip_infonce_grad(float* X, int rows, int cols, long long ld, int col0, int n_total, const float* scale,
                                    const float* lse_r, const float* lse_c, float weight, float* loss, float* dscale, void* stream) {
    if (!X || !scale || (!lse_r && !lse_c) || rows < 1 || cols < 1 || ld < cols || n_total < 1 || col0 < 0) return EEGCLIP_EINVAL;
    long long n = (long long)rows * cols;
    long long g = (n + 1023) / 1024;          // 4 elements per thread
    if (g > 2048) g = 2048;
    EEG_LAUNCH(infonce_grad_kernel, dim3((int)g), dim3(256), 8 * sizeof(float), stream, X, rows, cols, ld, col0, n_total, scale, lse_r, lse_c,
               weight, loss, dscale);
    return (int)hipGetLastError();
}
extern "C" int eegclip_infonce_loss(const float* X, int n, long long ld, const float* scale, const float* lse_r, const float* lse_c,
                                    float weight, float* loss, void* stream) {
    if (!X || !scale || !lse_r || !lse_c || !loss || n < 1 || ld < n) return EEGCLIP_EINVAL;
    EEG_LAUNCH(infonce_loss_kernel, dim3(1), dim3(256), 0, stream, X, n, ld, scale, lse_r, lse_c, weight, loss);
    return (int)hipGetLastError();
}
extern "C" int eegclip_topk_rows(const float* X, int rows, int cols, long long ld, int k, const float* scale, long long* out_idx,
                                 void* stream) {
    if (!X || !out_idx || rows < 1 || cols < 1 || ld < cols || k < 1 || k > 8 || k > cols) return EEGCLIP_EINVAL;
    int grid = (rows + 3) / 4;
    if (grid > 2048) grid = 2048;
    EEG_LAUNCH(topk_rows_kernel, dim3(grid), dim3(256), 0, stream, X, rows, cols, ld, k, scale, out_idx);
    return (int)hipGetLastError();
}
extern "C" int eegclip_top1_count(const float* X, int rows, int cols, long long ld, const float* scale, const long long* labels, int* count, void* stream) {
    if (!X || !labels || !count || rows < 1 || cols < 1 || ld < cols) return EEGCLIP_EINVAL;
    int grid = (rows + 3) / 4;
    if (grid > 1024) grid = 1024;
    EEG_LAUNCH(top1_count_kernel, dim3(grid), dim3(256), 0, stream, X, rows, cols, ld, scale, labels, count);
    return (int)hipGetLastError();
}
extern "C" int eegclip_count_equal(const long long* pred, int stride, const long long* labels, int n, int* count, void* stream) {
    if (!pred || !labels || !count || n < 1 || stride < 1) return EEGCLIP_EINVAL;
    EEG_LAUNCH(count_equal_kernel, dim3(1), dim3(256), 0, stream, pred, stride, labels, n, count);
    return (int)hipGetLastError();
}
