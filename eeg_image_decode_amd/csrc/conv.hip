// tsconv front: Conv2d(1,40,(1,25)) -> AvgPool2d((1,51),(1,5))  (Retrieval/ATMS_retrieval.py:102-103, the reference's
// CPU hotspot: 61 % of its step) folded into ONE 75-tap stride-5 temporal filter
//       y[b,c,h,w] = bias[c] + sum_{u<75} weff[c,u] * x[b,h,5w+u],   weff[c,u] = 1/51 * sum_{t<=24, 0<=u-t<=50} w[c,t]
// so the (B,40,63,226) intermediate (583 MB at B=256) never exists.  x is the encoder output (B,64,250) read in
// place (rows 0..62 of every sample: subject token + channels 0..61, ATMS_retrieval.py:91) -- no slice copy.
//
// The stride-5 filter is evaluated in polyphase form (u = 5g + r, X[q][r] = x[5q+r]) so that each lane keeps a
// register window and every LDS operand feeds >= 4 FMAs:
//   fwd   : lane (c, 4 outputs)      : 90-float x window in registers, 1 weight read per 4 FMAs
//   bwd_w : lane (c, r)              : dy[c][0..35] in registers, 1 x read per ~11 FMAs, 15 accumulators
//   bwd_x : lane (r, 10 outputs)     : per channel a 24-float dy window + 15 taps in registers, 150 FMAs
#include "eeg_common.h"

namespace eeg {

constexpr int TS_C = 40;     // temporal filters
constexpr int TS_T = 250;    // samples per token row
constexpr int TS_U = 75;     // folded taps
constexpr int TS_W = 36;     // outputs per row
constexpr int TS_K1 = 25;    // raw conv taps
constexpr int TS_POOL = 51;

// weff[c][u] from the raw (40,25) taps
__global__ void tsconv_fold_kernel(const float* __restrict__ w25, float* __restrict__ weff) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= TS_C * TS_U) return;
    const int c = i / TS_U, u = i % TS_U;
    const int t0 = u - (TS_POOL - 1) > 0 ? u - (TS_POOL - 1) : 0, t1 = u < TS_K1 - 1 ? u : TS_K1 - 1;
    float s = 0.f;
    for (int t = t0; t <= t1; ++t) s += w25[c * TS_K1 + t];
    weff[i] = s * (1.0f / TS_POOL);
}
// dw25[c][t] += 1/51 * sum_{u=t}^{t+50} dweff[c][u]
__global__ void tsconv_unfold_grad_kernel(const float* __restrict__ dweff, float* __restrict__ dw25) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= TS_C * TS_K1) return;
    const int c = i / TS_K1, t = i % TS_K1;
    float s = 0.f;
    for (int u = t; u < t + TS_POOL; ++u) s += dweff[c * TS_U + u];
    dw25[i] += s * (1.0f / TS_POOL);
}

// ---------------------------------------------------------------------------------------------------------------
// forward.  block = 384 threads (360 active): t -> (wg = t % 9, c = t / 9); outputs w = 4wg .. 4wg+3 of channel c.
// LDS: weff transposed [u][c] (3000 f) | x row (250 f, padded to 256) | stats scratch 2*360 f
__global__ __launch_bounds__(384) void tsconv_fwd_kernel(const float* __restrict__ x, long long xs_b, long long xs_h,
                                                          const float* __restrict__ weff, const float* __restrict__ bias,
                                                          float* __restrict__ y, int B, int H, double* __restrict__ sums) {
    EEG_LDS_BASE(float, lds);
    float* wl = lds;                    // [75][40]
    float* xl = lds + TS_U * TS_C;      // [256]
    float* sc = xl + 256;               // [2][360]
    const int t = threadIdx.x;
    for (int i = t; i < TS_U * TS_C; i += blockDim.x) {
        const int u = i / TS_C, c = i % TS_C;
        wl[i] = weff[c * TS_U + u];
    }
    const bool active = t < 360;
    const int wg = active ? t % 9 : 0, c = active ? t / 9 : 0;
    const float bc = bias[c];
    float ssum = 0.f, ssq = 0.f;
    const int rows = B * H;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int b = row / H, h = row % H;
        __syncthreads();                 // previous row's window reads are done (also covers the weight staging)
        if (t < TS_T) xl[t] = x[b * xs_b + h * xs_h + t];
        __syncthreads();
        if (active) {
            float win[90];
#pragma unroll
            for (int i = 0; i < 90; ++i) win[i] = xl[20 * wg + i];
            float a0 = bc, a1 = bc, a2 = bc, a3 = bc;
#pragma unroll
            for (int u = 0; u < TS_U; ++u) {
                const float wv = wl[u * TS_C + c];
                a0 += wv * win[u];
                a1 += wv * win[5 + u];
                a2 += wv * win[10 + u];
                a3 += wv * win[15 + u];
            }
            float* yp = y + (((long long)b * TS_C + c) * H + h) * TS_W + 4 * wg;
            *reinterpret_cast<float4*>(yp) = make_float4(a0, a1, a2, a3);
            ssum += (a0 + a1) + (a2 + a3);
            ssq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
    if (sums) {                          // BatchNorm batch statistics fused into the producer
        __syncthreads();
        if (active) { sc[t] = ssum; sc[360 + t] = ssq; }
        __syncthreads();
        if (t < TS_C) {
            double s = 0.0, q = 0.0;
            for (int k = 0; k < 9; ++k) { s += sc[t * 9 + k]; q += sc[360 + t * 9 + k]; }
            atomicAdd(sums + t, s);
            atomicAdd(sums + TS_C + t, q);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// dweff[c][5g+r] += sum_rows sum_w dy[row][c][w] * x[row][5(w+g)+r].   block = 256 (200 active): t -> (c = t % 40, r = t / 40)
// LDS: dy slab [40][37] | x row [256]
__global__ __launch_bounds__(256) void tsconv_bwd_w_kernel(const float* __restrict__ x, long long xs_b, long long xs_h,
                                                            const float* __restrict__ dy, float* __restrict__ dweff, int B, int H) {
    EEG_LDS_BASE(float, lds);
    float* dl = lds;                     // [40][37]
    float* xl = lds + TS_C * 37;         // [256]
    const int t = threadIdx.x;
    const bool active = t < 200;
    const int c = active ? t % TS_C : 0, r = active ? t / TS_C : 0;
    float acc[15];
#pragma unroll
    for (int g = 0; g < 15; ++g) acc[g] = 0.f;
    const int rows = B * H;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int b = row / H, h = row % H;
        __syncthreads();
        if (t < TS_T) xl[t] = x[b * xs_b + h * xs_h + t];
        for (int i = t; i < TS_C * TS_W; i += blockDim.x) {
            const int cc = i / TS_W, w = i % TS_W;
            dl[cc * 37 + w] = dy[(((long long)b * TS_C + cc) * H + h) * TS_W + w];
        }
        __syncthreads();
        if (active) {
            float d[TS_W];
#pragma unroll
            for (int w = 0; w < TS_W; ++w) d[w] = dl[c * 37 + w];
#pragma unroll
            for (int q = 0; q < 50; ++q) {
                const float xv = xl[5 * q + r];
#pragma unroll
                for (int g = 0; g < 15; ++g) {
                    const int w = q - g;
                    if (w >= 0 && w < TS_W) acc[g] += d[w] * xv;
                }
            }
        }
    }
    if (active) {
#pragma unroll
        for (int g = 0; g < 15; ++g) atomicAdd(dweff + c * TS_U + 5 * g + r, acc[g]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// dx[row][5q+r] = sum_c sum_g dy[row][c][q-g] * weff[c][5g+r].   block = 256 (250 active) = 10 rows x (5 q-blocks x 5 r)
// LDS: weff [40][75] | dy slabs [10][40*36]
constexpr int TSX_ROWS = 10;
__global__ __launch_bounds__(256) void tsconv_bwd_x_kernel(const float* __restrict__ dy, const float* __restrict__ weff,
                                                            float* __restrict__ dx, long long xs_b, long long xs_h, int B, int H) {
    EEG_LDS_BASE(float, lds);
    float* wl = lds;                         // [40][75]
    float* dl = lds + TS_C * TS_U;           // [10][1440]
    const int t = threadIdx.x;
    for (int i = t; i < TS_C * TS_U; i += blockDim.x) wl[i] = weff[i];
    const bool active = t < 250;
    const int rl = active ? t / 25 : 0, qb = active ? (t % 25) / 5 : 0, r = active ? t % 5 : 0;
    const int rows = B * H;
    for (int row0 = blockIdx.x * TSX_ROWS; row0 < rows; row0 += gridDim.x * TSX_ROWS) {
        __syncthreads();
        for (int i = t; i < TSX_ROWS * TS_C * TS_W; i += blockDim.x) {
            const int rr = i / (TS_C * TS_W), rem = i % (TS_C * TS_W);
            const int cc = rem / TS_W, w = rem % TS_W;
            const int row = row0 + rr;
            float v = 0.f;
            if (row < rows) {
                const int b = row / H, h = row % H;
                v = dy[(((long long)b * TS_C + cc) * H + h) * TS_W + w];
            }
            dl[i] = v;
        }
        __syncthreads();
        const int row = row0 + rl;
        if (active && row < rows) {
            float acc[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) acc[k] = 0.f;
            const int q0 = qb * 10;
            for (int c = 0; c < TS_C; ++c) {
                float dw[24], wv[15];
                const float* dp = dl + rl * (TS_C * TS_W) + c * TS_W;
#pragma unroll
                for (int i = 0; i < 24; ++i) {
                    const int w = q0 - 14 + i;
                    dw[i] = (w >= 0 && w < TS_W) ? dp[w] : 0.f;
                }
#pragma unroll
                for (int g = 0; g < 15; ++g) wv[g] = wl[c * TS_U + 5 * g + r];
#pragma unroll
                for (int k = 0; k < 10; ++k)
#pragma unroll
                    for (int g = 0; g < 15; ++g) acc[k] += dw[k - g + 14] * wv[g];
            }
            const int b = row / H, h = row % H;
            float* xp = dx + b * xs_b + h * xs_h;
#pragma unroll
            for (int k = 0; k < 10; ++k) xp[5 * (q0 + k) + r] = acc[k];
        }
    }
}

}  // namespace eeg

using namespace eeg;

extern "C" int eegclip_tsconv_fold(const float* w25, float* weff, void* stream) {
    if (!w25 || !weff) return EEGCLIP_EINVAL;
    EEG_LAUNCH(tsconv_fold_kernel, dim3((TS_C * TS_U + 255) / 256), dim3(256), 0, stream, w25, weff);
    return (int)hipGetLastError();
}
extern "C" int eegclip_tsconv_unfold_grad(const float* dweff, float* dw25, void* stream) {
    if (!dweff || !dw25) return EEGCLIP_EINVAL;
    EEG_LAUNCH(tsconv_unfold_grad_kernel, dim3((TS_C * TS_K1 + 255) / 256), dim3(256), 0, stream, dweff, dw25);
    return (int)hipGetLastError();
}

static int ts_check(int B, int H, int T, int C) {
    return (B < 1 || H < 1 || T != TS_T || C != TS_C) ? EEGCLIP_EINVAL : 0;
}

extern "C" int eegclip_tsconv_fwd(const float* x, long long xs_b, long long xs_h, const float* weff, const float* bias, float* y,
                                  int B, int H, int T, int C, double* sums, void* stream) {
    if (int rc = ts_check(B, H, T, C)) return rc;
    if (!x || !weff || !bias || !y) return EEGCLIP_EINVAL;
    if (((uintptr_t)y & 15) != 0) return EEGCLIP_EALIGN;
    int grid = B * H < 1024 ? B * H : 1024;
    const size_t lds = (TS_U * TS_C + 256 + 720) * sizeof(float);
    EEG_LAUNCH(tsconv_fwd_kernel, dim3(grid), dim3(384), lds, stream, x, xs_b, xs_h, weff, bias, y, B, H, sums);
    return (int)hipGetLastError();
}

extern "C" int eegclip_tsconv_bwd_w(const float* x, long long xs_b, long long xs_h, const float* dy, float* dweff, int B, int H, int T,
                                    int C, void* stream) {
    if (int rc = ts_check(B, H, T, C)) return rc;
    if (!x || !dy || !dweff) return EEGCLIP_EINVAL;
    int grid = B * H < 768 ? B * H : 768;
    const size_t lds = (TS_C * 37 + 256) * sizeof(float);
    EEG_LAUNCH(tsconv_bwd_w_kernel, dim3(grid), dim3(256), lds, stream, x, xs_b, xs_h, dy, dweff, B, H);
    return (int)hipGetLastError();
}

extern "C" int eegclip_tsconv_bwd_x(const float* dy, const float* weff, float* dx, long long xs_b, long long xs_h, int B, int H, int T,
                                    int C, void* stream) {
    if (int rc = ts_check(B, H, T, C)) return rc;
    if (!dy || !weff || !dx) return EEGCLIP_EINVAL;
    const int groups = (B * H + TSX_ROWS - 1) / TSX_ROWS;
    int grid = groups < 2048 ? groups : 2048;
    const size_t lds = (TS_C * TS_U + TSX_ROWS * TS_C * TS_W) * sizeof(float);
    EEG_LAUNCH(tsconv_bwd_x_kernel, dim3(grid), dim3(256), lds, stream, dy, weff, dx, xs_b, xs_h, B, H);
    return (int)hipGetLastError();
}
