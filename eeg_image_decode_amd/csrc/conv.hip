// tsconv front: Conv2d(1,40,(1,25)) -> AvgPool2d((1,51),(1,5))  (Retrieval/ATMS_retrieval.py:102-103, the reference's
// CPU hotspot: 61 % of its step) folded into ONE 75-tap stride-5 temporal filter
//       y[b,c,h,w] = bias[c] + sum_{u<75} weff[c,u] * x[b,h,5w+u],   weff[c,u] = 1/51 * sum_{t<=24, 0<=u-t<=50} w[c,t]
// so the (B,40,63,226) intermediate (583 MB at B=256) never exists.  x is the encoder output (B,64,250) read in
// place (rows 0..62 of every sample: subject token + channels 0..61, ATMS_retrieval.py:91) -- no slice copy.
//
#include "eeg_common.h"

#include <stdlib.h>

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

// ===============================================================================================================
// All three heavy kernels are implicit GEMMs on the f32 matrix cores (v_mfma_f32_16x16x4_f32) over the flattened output-position
// index m = (row, w), row = (b, h), X[m][u] = x[row][5w + u] read straight from the token rows staged in LDS:
//   fwd   : Y^T[c][m]   = sum_u  weff[c][u] * X[m][u]         (M = 40->48 filters, N = positions, K = 75->76 taps)
//   bwd_w : dW[c][u]    = sum_m  dy[c][m]   * X[m][u]         (15 accumulator tiles live across the whole reduction)
//   bwd_x : dX[row][s]  = sum_{w,c} dy[row][c][w] * weff[c][s-5w]   (M = 16 EEG rows, N = samples, K = banded (w,c): see the kernel)
// Channel-major accumulators in fwd (rows = filters, cols = 16 consecutive positions) make every store a 64-byte run of y.
constexpr int TS_CP = 48;     // filters padded to 3 MFMA tiles
constexpr int TS_UP = 80;     // taps padded to 5 MFMA tiles (bwd) ; fwd uses 76 = 19 k-steps
constexpr int TS_XS = 256;    // staged token-row stride (250 samples + zero pad: windows may read up to index 254)

// one token row per pass, lanes walk the 250 samples (coalesced); (b, h) of a row is wave-uniform scalar math, no per-element division
// register-staged variant used by the software-pipelined loops: NR rows -> NR registers per thread, stored to LDS one iteration later
template <int NR>
__device__ __forceinline__ void load_x_rows(float (&v)[NR], const float* x, long long xs_b, long long xs_h, int row0, int rows, int H) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int row = row0 + j;
        v[j] = (row < rows && t < TS_T) ? x[(row / H) * xs_b + (row % H) * xs_h + t] : 0.f;
    }
}
template <int NR>
__device__ __forceinline__ void store_x_rows(float* xl, const float (&v)[NR]) {
#pragma unroll
    for (int j = 0; j < NR; ++j) xl[j * TS_XS + threadIdx.x] = v[j];
}

// Loads are issued in batches of 8 independent rows BEFORE any is consumed: with one workgroup per CU a load->LDS-store chain per
// row would pay the full HBM latency (~1 us) per row (measured: 32 serialized rows = half of the forward kernel's time).
__device__ __forceinline__ void stage_x_rows(float* xl, const float* x, long long xs_b, long long xs_h, int row0, int nrows, int rows, int H) {
    const int t = threadIdx.x;            // blockDim.x == TS_XS == 256
    for (int r0 = 0; r0 < nrows; r0 += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = row0 + r0 + j;
            v[j] = (r0 + j < nrows && row < rows && t < TS_T) ? x[(row / H) * xs_b + (row % H) * xs_h + t] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (r0 + j < nrows) xl[(r0 + j) * TS_XS + t] = v[j];
    }
}

// ---- forward -------------------------------------------------------------------------------------------------------
constexpr int TSF_R = 32;                       // token rows per work item: 32*36/16 = 72 position tiles, 18 per wave
__global__ __launch_bounds__(256) void tsconv_fwd_kernel(const float* __restrict__ x, long long xs_b, long long xs_h,
                                                          const float* __restrict__ weff, const float* __restrict__ bias,
                                                          float* __restrict__ y, int B, int H, double* __restrict__ sums) {
    EEG_LDS_BASE(float, lds);
    float* wl = lds;                             // [76][48]  taps-major: A operand (filters) read = 16 consecutive floats
    float* xl = wl + 76 * TS_CP;                 // [32][256]
    float* sc = xl + TSF_R * TS_XS;              // [4][2][48] per-wave channel sums
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int rows = B * H;
    for (int i = t; i < 76 * TS_CP; i += blockDim.x) {
        const int u = i / TS_CP, c = i % TS_CP;
        wl[i] = (u < TS_U && c < TS_C) ? weff[c * TS_U + u] : 0.f;
    }
    float bc[3][4], ss[3][4], sq[3][4];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * ct + 4 * g + r;
            bc[ct][r] = c < TS_C ? bias[c] : 0.f;
            ss[ct][r] = 0.f;
            sq[ct][r] = 0.f;
        }
    const int nchunks = (rows + TSF_R - 1) / TSF_R;
    float vx[TSF_R];                               // next chunk's token rows, in flight under the current chunk's MFMAs
    if ((int)blockIdx.x < nchunks) load_x_rows<TSF_R>(vx, x, xs_b, xs_h, blockIdx.x * TSF_R, rows, H);
    for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const int row0 = ch * TSF_R;
        __syncthreads();                          // previous chunk fully consumed (also orders the weight staging)
        store_x_rows<TSF_R>(xl, vx);
        __syncthreads();
        if (ch + (int)gridDim.x < nchunks) load_x_rows<TSF_R>(vx, x, xs_b, xs_h, (ch + gridDim.x) * TSF_R, rows, H);
        for (int mt = wv; mt < TSF_R * TS_W / 16; mt += 4) {
            const int m = 16 * mt + fr;          // B-operand column: output position
            const float* xp = xl + (m / TS_W) * TS_XS + 5 * (m % TS_W) + g;
            f32x4 acc[3];
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 19; ++kk) {
                const float xv = xp[4 * kk];
                const float* wp = wl + (4 * kk + g) * TS_CP + fr;
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) acc[ct] = mfma_f32_16x16x4(wp[16 * ct], xv, acc[ct]);   // D[c = 16ct+4g+r][m = 16mt+fr]
            }
            const int row = row0 + m / TS_W;
            if (row < rows) {
                const int w = m % TS_W;
                float* yp = y + ((long long)(row / H) * TS_C * H + (row % H)) * TS_W + w;
#pragma unroll
                for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 16 * ct + 4 * g + r;
                        if (c < TS_C) {
                            const float v = acc[ct][r] + bc[ct][r];
                            yp[(long long)c * H * TS_W] = v;
                            ss[ct][r] += v;
                            sq[ct][r] += v * v;
                        }
                    }
            }
        }
    }
    if (sums) {                                   // BatchNorm batch statistics fused into the producer (fp64 atomics, one per channel per block)
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = ss[ct][r], b2 = sq[ct][r];
#pragma unroll
                for (int msk = 8; msk >= 1; msk >>= 1) { a += __shfl_xor(a, msk, 64); b2 += __shfl_xor(b2, msk, 64); }
                if (fr == 0) { sc[(wv * 2 + 0) * TS_CP + 16 * ct + 4 * g + r] = a; sc[(wv * 2 + 1) * TS_CP + 16 * ct + 4 * g + r] = b2; }
            }
        __syncthreads();
        if (t < TS_C) {
            double s = 0.0, q = 0.0;
            for (int k = 0; k < 4; ++k) { s += sc[(k * 2 + 0) * TS_CP + t]; q += sc[(k * 2 + 1) * TS_CP + t]; }
            atomicAdd(sums + t, s);
            atomicAdd(sums + TS_C + t, q);
        }
    }
}

// ---- backward w.r.t. the taps -----------------------------------------------------------------------------------------
// R = token rows per work item.  LDS = dy slab [48][36R + 17] + token rows [R][256] (the cross-wave reduction tile aliases the slab):
// R = 9: 75 KB, R = 7: 59 KB (2 workgroups per CU, 63 = 9 * 7 rows split evenly), R = 6: 51 KB (3 per CU).
template <int R>
__global__ __launch_bounds__(256) void tsconv_bwd_w_kernel(const float* __restrict__ x, long long xs_b, long long xs_h,
                                                            const float* __restrict__ dy, float* __restrict__ partials, int B, int H) {
    constexpr int MS = R * TS_W + 17;            // dy slab row stride (odd: skewed banks for the 16 filter rows of an operand read)
    constexpr int NQ = (R * TS_W + 255) / 256;   // slab positions per thread
    EEG_LDS_BASE(float, lds);
    float* dl = lds;                             // [48][MS]  dy slab, filter-major (rows >= 40 zero)
    float* xl = dl + TS_CP * MS;                 // [R][256]
    float* red = dl;                             // 2 x [48][84] cross-wave reduction scratch (after the last item)
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int per = (H + R - 1) / R;             // work items per sample
    f32x4 acc[3][5];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int ut = 0; ut < 5; ++ut) acc[ct][ut] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = t; i < TS_CP * MS; i += blockDim.x) dl[i] = 0.f;         // pad filters 40..47 stay zero for ever
    // software pipeline: the next work item's dy slab (NQ x 40 floats per thread) and token rows are loaded into registers
    // before the MFMAs of the current item
    float vd[NQ][TS_C], vx[R];
    auto load_item = [&](int item) {
        const int b = item / per, h0 = (item % per) * R;
        const int nr = H - h0 < R ? H - h0 : R;
        const int mc = nr * TS_W;
        load_x_rows<R>(vx, x, xs_b, xs_h, b * H + h0, b * H + h0 + nr, H);
        const float* src = dy + (((long long)b * TS_C) * H + h0) * TS_W;      // (h, w) contiguous for a fixed (b, c): lanes walk it
        const long long cs = (long long)H * TS_W;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int m = t + 256 * q;
#pragma unroll
            for (int c = 0; c < TS_C; ++c) vd[q][c] = m < mc ? src[c * cs + m] : 0.f;
        }
    };
    if ((int)blockIdx.x < B * per) load_item(blockIdx.x);
    for (int item = blockIdx.x; item < B * per; item += gridDim.x) {
        const int h0 = (item % per) * R;
        const int nr = H - h0 < R ? H - h0 : R;
        const int mc = nr * TS_W;                 // positions in this slab (multiple of 4)
        __syncthreads();
        store_x_rows<R>(xl, vx);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int m = t + 256 * q;
            if (m < mc) {
#pragma unroll
                for (int c = 0; c < TS_C; ++c) dl[c * MS + m] = vd[q][c];
            }
        }
        __syncthreads();
        if (item + (int)gridDim.x < B * per) load_item(item + gridDim.x);
        for (int ks = wv; ks < mc / 4; ks += 4) {
            const int m = 4 * ks + g;            // this lane's k index (position)
            const float* xp = xl + (m / TS_W) * TS_XS + 5 * (m % TS_W) + fr;
            float av[3], bv[5];
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) av[ct] = dl[(16 * ct + fr) * MS + m];
#pragma unroll
            for (int ut = 0; ut < 5; ++ut) bv[ut] = xp[16 * ut];
#pragma unroll
            for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                for (int ut = 0; ut < 5; ++ut) acc[ct][ut] = mfma_f32_16x16x4(av[ct], bv[ut], acc[ct][ut]);   // D[c][u]
        }
    }
    // cross-wave sum of the four position-partial accumulator sets: two-level tree through LDS with plain stores (60 ds_add_f32 per
    // lane into one shared tile cost ~170 LDS cycles per wave instruction: a third of this kernel's time at 2 workgroups per CU)
    constexpr int RLD = 84;                      // 4 accumulator row groups land 16 banks apart
    auto put = [&](float* reg) {
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int ut = 0; ut < 5; ++ut)
#pragma unroll
                for (int r = 0; r < 4; ++r) reg[(16 * ct + 4 * g + r) * RLD + 16 * ut + fr] = acc[ct][ut][r];
    };
    auto add = [&](const float* reg) {
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int ut = 0; ut < 5; ++ut)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[ct][ut][r] += reg[(16 * ct + 4 * g + r) * RLD + 16 * ut + fr];
    };
    static_assert(2 * TS_CP * RLD <= TS_CP * MS + R * TS_XS, "reduction scratch must fit in the operand tiles");
    __syncthreads();
    if (wv >= 2) put(red + (wv - 2) * TS_CP * RLD);
    __syncthreads();
    if (wv < 2) add(red + wv * TS_CP * RLD);
    __syncthreads();
    if (wv == 1) put(red);
    __syncthreads();
    if (wv == 0) {
        add(red);
        float* out = partials + (long long)blockIdx.x * (TS_C * TS_U);
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int ut = 0; ut < 5; ++ut)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * ct + 4 * g + r, u = 16 * ut + fr;
                    if (c < TS_C && u < TS_U) out[c * TS_U + u] = acc[ct][ut][r];
                }
    }
}

// dweff[i] += sum over a slice of the workgroup partials (grid.y slices; dweff zeroed by the launcher): 3000 x 16 threads keep
// enough loads in flight -- a single thread walking 512 partials is pure memory latency
__global__ void tsconv_bwd_w_reduce_kernel(const float* __restrict__ partials, int nblk, float* __restrict__ dweff) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= TS_C * TS_U) return;
    float s = 0.f;
    for (int k = blockIdx.y; k < nblk; k += gridDim.y) s += partials[(long long)k * (TS_C * TS_U) + i];
    atomicAdd(dweff + i, s);
}

// ---- backward w.r.t. the token rows ------------------------------------------------------------------------------------
// dx[row][s] = sum_{c,w} dy[row][c][w] * weff[c][s - 5w]  as a GEMM whose M dimension is 16 different EEG rows, N = s and
// K = (w, c) restricted, per 16-wide s tile, to the <= 19 output positions w whose 75-tap window touches the tile (202 of the
// 16 x 36 (tile, w) pairs).  The overlap-add of the transposed convolution happens INSIDE the MFMA accumulation -- a first
// version that scattered per-position tiles with ds_add_f32 was LDS-atomic bound (SQ_LDS_IDX_ACTIVE 136 M cycles, 277 us).
// B operand = Toeplitz view of the taps: lane (s, c) reads weff[c][s - 5w] (zero outside 0..74) straight from LDS.
constexpr int TSX_R = 16;                       // EEG rows per work item = MFMA M
constexpr int TSX_WS = 17;                      // floats per (c, w) cell: 16 rows + 1 pad -> conflict-free staging writes
constexpr int TSX_CS = 624;                     // floats per channel: 36*17 = 612 padded to 16 (mod 32) for the operand reads
// CH = channels per LDS slab: 20 (2 slabs per item, 63 KB: 2 workgroups per CU) or 8 (5 slabs, 33 KB: 4 per CU)
template <int CH>
__global__ __launch_bounds__(256, CH == 20 ? 2 : 4) void tsconv_bwd_x_kernel(const float* __restrict__ dy, const float* __restrict__ weff,
                                                                            float* __restrict__ dx, long long xs_b, long long xs_h, int B, int H) {
    EEG_LDS_BASE(float, lds);
    float* wl = lds;                             // [40][80]  filter-major taps (cols >= 75 zero)
    float* dl = wl + TS_C * TS_UP;               // [CH][TSX_CS]  dy slab: dl[c][w][row]
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int rows = B * H;
    {
        float v[13];
#pragma unroll
        for (int j = 0; j < 13; ++j) { const int i = t + 256 * j; v[j] = (i < TS_C * TS_UP && i % TS_UP < TS_U) ? weff[(i / TS_UP) * TS_U + i % TS_UP] : 0.f; }
#pragma unroll
        for (int j = 0; j < 13; ++j) { const int i = t + 256 * j; if (i < TS_C * TS_UP) wl[i] = v[j]; }
    }
    const int nitems = (rows + TSX_R - 1) / TSX_R;
    // A slab = (work item, CH channels): 16 rows x CH channels x 36 positions.  For one channel the 16 rows are (h, w)-contiguous in
    // dy (144 float4, split in two where the item crosses a sample), so the slab is fetched as 16-byte loads, and the NEXT slab's
    // loads are issued before the MFMAs of the current one: the first version staged 2 rows at a time with a full memory round trip
    // per batch (16 serialized latencies per item -- that, not the matrix cores, set the kernel's time).
    // Thread -> float4 map: NT active threads, element e = t + NT * j with NT = 240 (CH 20) / 192 (CH 8): NT * 3 is a whole number
    // of channels (5 / 4), so j = 3m + r needs only THREE (channel, row, w) decodes per thread, m adds a constant LDS / HBM offset
    // (a t + 256 j map made every one of the 12 decodes loop invariant: 268 VGPRs, one wave per SIMD).
    constexpr int NT = CH == 20 ? 240 : 192, MCH = NT * 3 / 144, NM = CH / MCH;
    static_assert(MCH * NM == CH && CH % 4 == 0 && TS_C % CH == 0, "slab shape");
    const f32x4 zero4v{0.f, 0.f, 0.f, 0.f};
    int e_rl[3], e_lds[3], e_gl[3];              // per r: row in the item, LDS float offset, dy float offset relative to (b0, c0, h0)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int e = t + NT * r, cl = e / 144, rem = e % 144, rl = rem / 9, w4 = rem % 9;
        e_rl[r] = rl;
        e_lds[r] = cl * TSX_CS + 4 * w4 * TSX_WS + rl;
        e_gl[r] = (cl * H + rl) * TS_W + 4 * w4;
    }
    f32x4 v[NM][3];
    auto load_slab = [&](int item, int sl) {
        const int row0 = item * TSX_R, b0 = row0 / H, h0 = row0 % H;
        const float* base = dy + (((long long)b0 * TS_C + sl * CH) * H + h0) * TS_W;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            // row h0 + rl of sample b0, or -- past the sample's last row -- row h0 + rl - k H of sample b0 + k: + k * 39 H rows
            int wraps = 0;
            if (H >= TSX_R) wraps = h0 + e_rl[r] >= H ? 1 : 0;
            else wraps = (h0 + e_rl[r]) / H;
            const bool ok = t < NT && row0 + e_rl[r] < rows;
            const float* p = base + e_gl[r] + (long long)wraps * (TS_C - 1) * H * TS_W;
#pragma unroll
            for (int m = 0; m < NM; ++m) v[m][r] = ok ? *reinterpret_cast<const f32x4*>(p + (long long)m * MCH * H * TS_W) : zero4v;
        }
    };
    auto store_slab = [&]() {
        if (t < NT) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int m = 0; m < NM; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q) dl[e_lds[r] + m * MCH * TSX_CS + q * TSX_WS] = v[m][r][q];
        }
    };
    if ((int)blockIdx.x < nitems) load_slab(blockIdx.x, 0);
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int row0 = item * TSX_R;
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int sl = 0; sl < TS_C / CH; ++sl) {
            const int cbase = sl * CH;
            __syncthreads();                      // previous slab fully consumed (also orders the tap staging)
            store_slab();
            __syncthreads();
            if (sl + 1 < TS_C / CH) load_slab(item, sl + 1);
            else if (item + (int)gridDim.x < nitems) load_slab(item + gridDim.x, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = wv + 4 * q;                         // s tile: s = 16j .. 16j+15
                int w_lo = (16 * j - (TS_U - 1) + 4) / 5;         // ceil((16j - 74) / 5) for the positive case
                if (16 * j - (TS_U - 1) <= 0) w_lo = 0;
                int w_hi = (16 * j + 15) / 5;
                if (w_hi > TS_W - 1) w_hi = TS_W - 1;
                const int s = 16 * j + fr;
                for (int w = w_lo; w <= w_hi; ++w) {
                    const int u = s - 5 * w;
                    const bool inb = u >= 0 && u < TS_U;
#pragma unroll
                    for (int cc = 0; cc < CH / 4; ++cc) {
                        const int cl = 4 * cc + g;
                        const float a = dl[cl * TSX_CS + w * TSX_WS + fr];                     // A[row = fr][k = (w, c)]
                        const float bq = inb ? wl[(cbase + cl) * TS_UP + u] : 0.f;             // B[k][s] = weff[c][s - 5w]
                        acc[q] = mfma_f32_16x16x4(a, bq, acc[q]);                               // D[row = 4g + r][s = 16j + fr]
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = 16 * (wv + 4 * q) + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 4 * g + r;
                if (row < rows && s < TS_T) dx[(row / H) * xs_b + (row % H) * xs_h + s] = acc[q][r];
            }
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
    const int nchunks = (B * H + TSF_R - 1) / TSF_R;
    int grid = nchunks < 768 ? nchunks : 768;
    const size_t lds = (76 * TS_CP + TSF_R * TS_XS + 8 * TS_CP) * sizeof(float);
    EEG_LAUNCH(tsconv_fwd_kernel, dim3(grid), dim3(256), lds, stream, x, xs_b, xs_h, weff, bias, y, B, H, sums);
    return (int)hipGetLastError();
}

// rows per work item / resident workgroups per CU (tuning aid: EEGCLIP_TSW_R = 6 | 7 | 9)
static int tsw_rows() {
    static const int r = getenv("EEGCLIP_TSW_R") ? atoi(getenv("EEGCLIP_TSW_R")) : 7;
    return (r == 6 || r == 9) ? r : 7;
}
static int tsw_grid(int B, int H) {
    const int r = tsw_rows();
    const int items = B * ((H + r - 1) / r), cap = r == 6 ? 768 : 512;
    return items < cap ? items : cap;
}

extern "C" long long eegclip_tsconv_bwd_w_workspace_floats(int B, int H) { return (long long)tsw_grid(B, H) * TS_C * TS_U; }

template <int R>
static void tsw_launch(int grid, void* stream, const float* x, long long xs_b, long long xs_h, const float* dy, float* workspace, int B, int H) {
    const size_t lds = (TS_CP * (R * TS_W + 17) + R * TS_XS) * sizeof(float);
    EEG_LAUNCH(tsconv_bwd_w_kernel<R>, dim3(grid), dim3(256), lds, stream, x, xs_b, xs_h, dy, workspace, B, H);
}

extern "C" int eegclip_tsconv_bwd_w(const float* x, long long xs_b, long long xs_h, const float* dy, float* dweff, float* workspace, int B,
                                    int H, int T, int C, void* stream) {
    if (int rc = ts_check(B, H, T, C)) return rc;
    if (!x || !dy || !dweff || !workspace) return EEGCLIP_EINVAL;
    const int grid = tsw_grid(B, H), r = tsw_rows();
    if (r == 6)      tsw_launch<6>(grid, stream, x, xs_b, xs_h, dy, workspace, B, H);
    else if (r == 9) tsw_launch<9>(grid, stream, x, xs_b, xs_h, dy, workspace, B, H);
    else             tsw_launch<7>(grid, stream, x, xs_b, xs_h, dy, workspace, B, H);
    hipMemsetAsync(dweff, 0, TS_C * TS_U * sizeof(float), (hipStream_t)stream);
    EEG_LAUNCH(tsconv_bwd_w_reduce_kernel, dim3((TS_C * TS_U + 255) / 256, grid < 32 ? grid : 32), dim3(256), 0, stream, workspace, grid, dweff);
    return (int)hipGetLastError();
}

extern "C" int eegclip_tsconv_bwd_x(const float* dy, const float* weff, float* dx, long long xs_b, long long xs_h, int B, int H, int T,
                                    int C, void* stream) {
    if (int rc = ts_check(B, H, T, C)) return rc;
    if (!dy || !weff || !dx) return EEGCLIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(dy) & 15u) return EEGCLIP_EALIGN;
    if ((long long)B * TS_C * H * TS_W >= (1LL << 31)) return EEGCLIP_EINVAL;
    const int items = (B * H + TSX_R - 1) / TSX_R;
    static const int ch = getenv("EEGCLIP_TSX_CH") ? atoi(getenv("EEGCLIP_TSX_CH")) : 8;       // tuning aid: 20 | 8 (measured 78 / 66 us)
    if (ch == 8) {
        const size_t lds = (TS_C * TS_UP + 8 * TSX_CS) * sizeof(float);
        EEG_LAUNCH(tsconv_bwd_x_kernel<8>, dim3(items < 1024 ? items : 1024), dim3(256), lds, stream, dy, weff, dx, xs_b, xs_h, B, H);
    } else {
        const size_t lds = (TS_C * TS_UP + 20 * TSX_CS) * sizeof(float);
        EEG_LAUNCH(tsconv_bwd_x_kernel<20>, dim3(items < 1024 ? items : 1024), dim3(256), lds, stream, dy, weff, dx, xs_b, xs_h, B, H);
    }
    return (int)hipGetLastError();
}
