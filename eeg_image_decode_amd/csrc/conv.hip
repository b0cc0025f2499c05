// tsconv front: Conv2d(1,40,(1,25)) -> AvgPool2d((1,51),(1,5))  (Retrieval/ATMS_retrieval.py:102-103, the reference's
// CPU hotspot: 61 % of its step).  The pool is a box filter, and box filter and convolution commute:
//       y[b,c,h,w] = bias[c] + sum_{t<25} w[c,t] * S[b,h,5w+t],     S[b,h,j] = 1/51 * sum_{p<51} x[b,h,j+p]   (j < 200)
// S is ONE sliding-window sum per token row (a wave-level prefix sum, shared by all 40 filters), after which the stage is a 25-tap
// stride-5 convolution: K = 25 on the matrix cores instead of the 75 taps of the folded filter conv*box (a first version used that
// fold: 2.7x the MFMA work in all three kernels), and the (B,40,63,226) intermediate (583 MB at B=256) never exists either way.
// x is the encoder output (B,64,250) read in place (rows 0..62 of every sample: subject token + channels 0..61, :91) -- no slice copy.
// Backward:  dw[c,t] = sum dy[b,c,h,w] * S[b,h,5w+t] ;  dS[b,h,j] = sum_{c,w} dy[b,c,h,w] * w[c,j-5w] ;  dx = box^T(dS).
#include "conv_common.h"

#include <stdlib.h>

#include <type_traits>

namespace eeg {

constexpr int TS_C = 40;     // temporal filters
constexpr int TS_T = 250;    // samples per token row
constexpr int TS_W = 36;     // outputs per row
constexpr int TS_K1 = 25;    // conv taps
constexpr int TS_POOL = 51;
constexpr int TS_NS = 200;   // box-filtered samples per row that the 36 x 25 windows touch
constexpr int TS_CP = 48;    // filters padded to 3 MFMA tiles
constexpr int TS_KP = 28;    // taps padded to 7 MFMA k-steps
constexpr int TS_XS = 256;   // LDS row stride of a staged row (>= 208: padded taps read up to S[206], zeros)

typedef float f32x2 __attribute__((ext_vector_type(2)));

// lane l <- x[row][4l .. 4l+3] (zero past the 250 samples); rows are addressed through (xs_b, xs_h): element strides of a sample / a row
__device__ __forceinline__ void load_row4(float (&v)[4], const float* x, long long xs_b, long long xs_h, int row, int rows, int H, bool vec2) {
    const int lane = threadIdx.x & 63;
    const bool ok = row < rows;
    const float* xr = x + (ok ? (row / H) * xs_b + (row % H) * xs_h : 0);
    if (vec2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = 4 * lane + 2 * h;
            const f32x2 t = (ok && c < TS_T) ? *reinterpret_cast<const f32x2*>(xr + c) : f32x2{0.f, 0.f};
            v[2 * h] = t[0];
            v[2 * h + 1] = t[1];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (ok && 4 * lane + e < TS_T) ? xr[4 * lane + e] : 0.f;
    }
}

// One wave turns the token row held as v (lane l: samples 4l..4l+3) into its box-filtered row S[0..255] in LDS (zeros from j = 200):
// exclusive prefix sums P[i] = sum_{k<i} x[k] go to the wave's scratch row, S[j] = (P[j+51] - P[j]) / 51.
__device__ __forceinline__ void box_filter_row(float* __restrict__ srow, float* __restrict__ pscr, const float (&v)[4]) {
    const int lane = threadIdx.x & 63;
    const float p0 = v[0], p1 = p0 + v[1], p2 = p1 + v[2], p3 = p2 + v[3];
    const float base = wave_inclusive_scan(p3) - p3;           // sum of the samples of all lower lanes
    const f32x4 P{base, base + p0, base + p1, base + p2};      // P[4l .. 4l+3]
    *reinterpret_cast<f32x4*>(pscr + 4 * lane) = P;
    wave_sync();
    f32x4 S;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int j = 4 * lane + e;
        S[e] = j < TS_NS ? (pscr[j + TS_POOL] - P[e]) * (1.0f / TS_POOL) : 0.f;
    }
    wave_sync();                                               // scratch row is free for the wave's next row
    *reinterpret_cast<f32x4*>(srow + 4 * lane) = S;
}

// ===============================================================================================================
// The heavy kernels are implicit GEMMs on the f32 matrix cores (v_mfma_f32_16x16x4_f32) over the flattened output-position index
// m = (row, w), row = (b, h), with the im2col view X[m][t] = S[row][5w + t] read straight from the box-filtered rows in LDS:
//   fwd   : Y^T[c][m]   = sum_t  w[c][t]   * X[m][t]          (M = 40->48 filters, N = positions, K = 25->28 taps)
//   bwd_w : dW[c][t]    = sum_m  dy[c][m]  * X[m][t]          (6 accumulator tiles live across the whole reduction)
//   bwd_x : dS[row][j]  = sum_{w,c} dy[row][c][w] * w[c][j-5w]   (M = 16 EEG rows, N = box-filtered samples, K = banded (w,c))
// Channel-major accumulators in fwd (rows = filters, cols = 16 consecutive positions) make every store a 64-byte run of y.

// ---- forward -------------------------------------------------------------------------------------------------------
constexpr int TSF_R = 32;                       // token rows per work item: 32*36/16 = 72 position tiles, 18 per wave; 8 rows per wave
__global__ __launch_bounds__(256) void tsconv_fwd_kernel(const float* __restrict__ x, long long xs_b, long long xs_h,
                                                          const float* __restrict__ w25, const float* __restrict__ bias,
                                                          float* __restrict__ y, int B, int H, double* __restrict__ sums, double* __restrict__ partials,
                                                          int vec2) {
    EEG_LDS_BASE(float, lds);
    float* wl = lds;                             // [28][48]  taps-major: A operand (filters) read = 16 consecutive floats
    float* sl = wl + TS_KP * TS_CP;              // [32][256] box-filtered rows
    float* ps = sl + TSF_R * TS_XS;              // [4][256]  per-wave prefix scratch
    float* sc = ps + 4 * TS_XS;                  // [4][2][48] per-wave channel sums
    const int t = threadIdx.x, lane = t & 63, wv = wave_uniform(t >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int rows = B * H;
    for (int i = t; i < TS_KP * TS_CP; i += blockDim.x) {
        const int u = i / TS_CP, c = i % TS_CP;
        wl[i] = (u < TS_K1 && c < TS_C) ? w25[c * TS_K1 + u] : 0.f;
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
    float vx[TSF_R / 4][4];                        // next chunk's token rows of this wave, in flight under the current chunk's MFMAs
    auto load_chunk = [&](int ch) {
#pragma unroll
        for (int j = 0; j < TSF_R / 4; ++j) load_row4(vx[j], x, xs_b, xs_h, ch * TSF_R + wv * (TSF_R / 4) + j, rows, H, vec2 != 0);
    };
    if ((int)blockIdx.x < nchunks) load_chunk(blockIdx.x);
    for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const int row0 = ch * TSF_R;
        __syncthreads();                          // previous chunk fully consumed (also orders the weight staging)
#pragma unroll
        for (int j = 0; j < TSF_R / 4; ++j) box_filter_row(sl + (wv * (TSF_R / 4) + j) * TS_XS, ps + wv * TS_XS, vx[j]);
        __syncthreads();
        if (ch + (int)gridDim.x < nchunks) load_chunk(ch + gridDim.x);
        for (int mt = wv; mt < TSF_R * TS_W / 16; mt += 4) {
            const int m = 16 * mt + fr;          // B-operand column: output position
            const float* xp = sl + (m / TS_W) * TS_XS + 5 * (m % TS_W) + g;
            f32x4 acc[3];
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < TS_KP / 4; ++kk) {
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
            if (partials) {                       // a partial row per workgroup, column-summed by colsum_f64_kernel (504-way contended atomics otherwise)
                partials[(long long)blockIdx.x * 2 * TS_C + t] = s;
                partials[(long long)blockIdx.x * 2 * TS_C + TS_C + t] = q;
            } else {
                atomicAdd(sums + t, s);
                atomicAdd(sums + TS_C + t, q);
            }
        }
    }
}

// ---- backward w.r.t. the taps -----------------------------------------------------------------------------------------
// R = token rows per work item.  LDS = dy slab [48][36R + 17] + box-filtered rows [R][256] + scan scratch (the cross-wave reduction
// scratch aliases the slab): R = 7: 63 KB (2 workgroups per CU, 63 = 9 * 7 rows split evenly), R = 6: 55 KB.
template <int R>
__global__ __launch_bounds__(256) void tsconv_bwd_w_kernel(const float* __restrict__ x, long long xs_b, long long xs_h,
                                                            const float* __restrict__ dy, float* __restrict__ partials, int B, int H, int vec2) {
    constexpr int MS = R * TS_W + 17;            // dy slab row stride (odd: skewed banks for the 16 filter rows of an operand read)
    constexpr int NQ = (R * TS_W + 255) / 256;   // slab positions per thread
    constexpr int RW = (R + 3) / 4;              // rows box-filtered per wave
    EEG_LDS_BASE(float, lds);
    float* dl = lds;                             // [48][MS]  dy slab, filter-major (rows >= 40 zero)
    float* sl = dl + TS_CP * MS;                 // [R][256]  box-filtered token rows
    float* ps = sl + R * TS_XS;                  // [4][256]
    float* red = dl;                             // 2 x [48][36] cross-wave reduction scratch (after the last item)
    const int t = threadIdx.x, lane = t & 63, wv = wave_uniform(t >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int per = (H + R - 1) / R;             // work items per sample
    f32x4 acc[3][2];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int ut = 0; ut < 2; ++ut) acc[ct][ut] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = t; i < TS_CP * MS; i += blockDim.x) dl[i] = 0.f;         // pad filters 40..47 stay zero for ever
    for (int i = t; i < R * TS_XS; i += blockDim.x) sl[i] = 0.f;          // rows a short last item does not fill
    // software pipeline: the next work item's dy slab (NQ x 40 floats per thread) and token rows are loaded into registers
    // before the MFMAs of the current item
    float vd[NQ][TS_C], vx[RW][4];
    auto load_item = [&](int item) {
        const int b = item / per, h0 = (item % per) * R;
        const int nr = H - h0 < R ? H - h0 : R;
        const int mc = nr * TS_W;
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int rl = wv + 4 * j;
            load_row4(vx[j], x, xs_b, xs_h, rl < nr ? b * H + h0 + rl : B * H, B * H, H, vec2 != 0);
        }
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
#pragma unroll
        for (int j = 0; j < RW; ++j)
            if (wv + 4 * j < R) box_filter_row(sl + (wv + 4 * j) * TS_XS, ps + wv * TS_XS, vx[j]);
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
            const float* xp = sl + (m / TS_W) * TS_XS + 5 * (m % TS_W) + fr;
            float av[3], bv[2];
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) av[ct] = dl[(16 * ct + fr) * MS + m];
#pragma unroll
            for (int ut = 0; ut < 2; ++ut) bv[ut] = xp[16 * ut];              // S[5w + t], t = 16ut + fr (t >= 25: discarded columns)
#pragma unroll
            for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                for (int ut = 0; ut < 2; ++ut) acc[ct][ut] = mfma_f32_16x16x4(av[ct], bv[ut], acc[ct][ut]);   // D[c][t]
        }
    }
    // cross-wave sum of the four position-partial accumulator sets: two-level tree through LDS with plain stores (ds_add_f32 into
    // one shared tile cost ~170 LDS cycles per wave instruction)
    constexpr int RLD = 36;                      // 4 accumulator row groups land 16 banks apart
    auto put = [&](float* reg) {
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int ut = 0; ut < 2; ++ut)
#pragma unroll
                for (int r = 0; r < 4; ++r) reg[(16 * ct + 4 * g + r) * RLD + 16 * ut + fr] = acc[ct][ut][r];
    };
    auto add = [&](const float* reg) {
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int ut = 0; ut < 2; ++ut)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[ct][ut][r] += reg[(16 * ct + 4 * g + r) * RLD + 16 * ut + fr];
    };
    static_assert(2 * TS_CP * RLD <= TS_CP * MS, "reduction scratch must fit in the dy slab");
    __syncthreads();
    if (wv >= 2) put(red + (wv - 2) * TS_CP * RLD);
    __syncthreads();
    if (wv < 2) add(red + wv * TS_CP * RLD);
    __syncthreads();
    if (wv == 1) put(red);
    __syncthreads();
    if (wv == 0) {
        add(red);
        float* out = partials + (long long)blockIdx.x * (TS_C * TS_K1);
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int ut = 0; ut < 2; ++ut)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * ct + 4 * g + r, u = 16 * ut + fr;
                    if (c < TS_C && u < TS_K1) out[c * TS_K1 + u] = acc[ct][ut][r];
                }
    }
}

// dw25[i] += sum over the workgroup partials: one thread per (tap, slice of workgroups), slices combined with atomics
__global__ void tsconv_bwd_w_reduce_kernel(const float* __restrict__ partials, int nblk, float* __restrict__ dw25) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= TS_C * TS_K1) return;
    float s = 0.f;
#pragma unroll 8
    for (int k = blockIdx.y; k < nblk; k += gridDim.y) s += partials[(long long)k * (TS_C * TS_K1) + i];
    atomicAdd(dw25 + i, s);
}

// ---- backward w.r.t. the token rows ------------------------------------------------------------------------------------
// dS[row][j] = sum_{c,w} dy[row][c][w] * w[c][j - 5w]  as a GEMM whose M dimension is 16 different EEG rows, N = j (13 tiles of 16)
// and K = (w, c) restricted, per j tile, to the <= 8 output positions w whose 25-tap window touches the tile.  The overlap-add of the
// transposed convolution happens INSIDE the MFMA accumulation.  B operand = Toeplitz view of the taps: lane (j, c) reads
// w[c][j - 5w] (zero outside 0..24) from LDS.  Epilogue: dS rows -> LDS -> wave prefix sums Q -> dx[i] = (Q[i+1] - Q[i-50]) / 51,
// the transpose of the box filter (dS[j] = 0 for j >= 200, so Q is constant from 200 on and needs no clamp on the right).
constexpr int TSX_R = 16;                       // EEG rows per work item = MFMA M
constexpr int TSX_WS = 17;                      // floats per (c, w) cell: 16 rows + 1 pad -> conflict-free staging writes
constexpr int TSX_CS = 624;                     // floats per channel: 36*17 = 612 padded to 16 (mod 32) for the operand reads
constexpr int TSX_CH = 8;                       // channels per LDS slab (5 slabs per item)
constexpr int TSX_WL = 32;                      // tap row stride in LDS
constexpr int TSX_NJ = 13;                      // 16-wide j tiles covering the 200 box-filtered samples
__global__ __launch_bounds__(256, 4) void tsconv_bwd_x_kernel(const float* __restrict__ dy, const float* __restrict__ w25,
                                                               float* __restrict__ dx, long long xs_b, long long xs_h, int B, int H, int vec2) {
    constexpr int CH = TSX_CH;
    EEG_LDS_BASE(float, lds);
    float* wl = lds;                             // [40][32]  taps (cols >= 25 zero)
    float* dl = wl + TS_C * TSX_WL;              // [CH][TSX_CS]  dy slab: dl[c][w][row]
    float* ps = dl + CH * TSX_CS;                // [4][256]  prefix scratch
    float* dsl = dl;                             // [16][256] dS rows (aliases the slab once the accumulation is done)
    const int t = threadIdx.x, lane = t & 63, wv = wave_uniform(t >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int rows = B * H;
    for (int i = t; i < TS_C * TSX_WL; i += 256) wl[i] = (i % TSX_WL) < TS_K1 ? w25[(i / TSX_WL) * TS_K1 + i % TSX_WL] : 0.f;
    const int nitems = (rows + TSX_R - 1) / TSX_R;
    // A slab = (work item, CH channels): 16 rows x CH channels x 36 positions.  For one channel the 16 rows are (h, w)-contiguous in
    // dy (144 float4, split in two where the item crosses a sample), so the slab is fetched as 16-byte loads, and the NEXT slab's
    // loads are issued before the MFMAs of the current one.  Thread -> float4 map: 192 active threads, element e = t + 192 j:
    // 192 * 3 is a whole number of channels (4), so j = 3m + r needs only THREE (channel, row, w) decodes per thread and m adds a
    // constant LDS / HBM offset (a t + 256 j map made every decode loop invariant: 268 VGPRs, one wave per SIMD).
    constexpr int NT = 192, MCH = NT * 3 / 144, NM = CH / MCH;
    static_assert(MCH * NM == CH && CH % 4 == 0 && TS_C % CH == 0, "slab shape");
    static_assert(TSX_R * TS_XS <= CH * TSX_CS, "dS rows must fit in the slab");
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
            __syncthreads();                      // previous slab / dS rows fully consumed (also orders the tap staging)
            store_slab();
            __syncthreads();
            if (sl + 1 < TS_C / CH) load_slab(item, sl + 1);
            else if (item + (int)gridDim.x < nitems) load_slab(item + gridDim.x, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int jt = wv + 4 * q;                        // j tile: j = 16 jt .. 16 jt + 15
                if (jt >= TSX_NJ) continue;
                int w_lo = (16 * jt - (TS_K1 - 1) + 4) / 5;       // ceil((16 jt - 24) / 5) for the positive case
                if (16 * jt - (TS_K1 - 1) <= 0) w_lo = 0;
                int w_hi = (16 * jt + 15) / 5;
                if (w_hi > TS_W - 1) w_hi = TS_W - 1;
                const int j = 16 * jt + fr;
                for (int w = w_lo; w <= w_hi; ++w) {
                    const int u = j - 5 * w;
                    const bool inb = u >= 0 && u < TS_K1;
#pragma unroll
                    for (int cc = 0; cc < CH / 4; ++cc) {
                        const int cl = 4 * cc + g;
                        const float a = dl[cl * TSX_CS + w * TSX_WS + fr];                     // A[row = fr][k = (w, c)]
                        const float bq = inb ? wl[(cbase + cl) * TSX_WL + u] : 0.f;            // B[k][j] = w[c][j - 5w]
                        acc[q] = mfma_f32_16x16x4(a, bq, acc[q]);                               // D[row = 4g + r][j = 16 jt + fr]
                    }
                }
            }
        }
        __syncthreads();                          // every wave is done with the last slab: it becomes the dS rows
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int jt = wv + 4 * q;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (jt < TSX_NJ) dsl[(4 * g + r) * TS_XS + 16 * jt + fr] = acc[q][r];
        }
        __syncthreads();
        // transpose of the box filter, 4 rows per wave: Q[i] = sum_{k<i} dS[k] ; dx[i] = (Q[i+1] - Q[max(i-50, 0)]) / 51
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int rl = 4 * wv + rr, row = row0 + rl;
            const f32x4 d = 4 * lane < 16 * TSX_NJ ? *reinterpret_cast<const f32x4*>(dsl + rl * TS_XS + 4 * lane) : zero4v;
            const float p0 = d[0], p1 = p0 + d[1], p2 = p1 + d[2], p3 = p2 + d[3];
            const float basev = wave_inclusive_scan(p3) - p3;
            float* pq = ps + wv * TS_XS;
            const float q4[5] = {basev, basev + p0, basev + p1, basev + p2, basev + p3};                        // Q[4l .. 4l+4]
            *reinterpret_cast<f32x4*>(pq + 4 * lane) = f32x4{q4[0], q4[1], q4[2], q4[3]};
            wave_sync();
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * lane + e;
                const float lo = i >= TS_POOL - 1 ? pq[i - (TS_POOL - 1)] : 0.f;                                // Q[i - 50]
                o[e] = (q4[e + 1] - lo) * (1.0f / TS_POOL);
            }
            wave_sync();
            if (row < rows) {
                float* xr = dx + (row / H) * xs_b + (row % H) * xs_h;
                if (vec2) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        if (4 * lane + 2 * h < TS_T) *reinterpret_cast<f32x2*>(xr + 4 * lane + 2 * h) = f32x2{o[2 * h], o[2 * h + 1]};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (4 * lane + e < TS_T) xr[4 * lane + e] = o[e];
                }
            }
        }
    }
}


// ===============================================================================================================
// BatchNorm1-backward apply + the whole temporal-conv backward in ONE pass over y1 (round 4): the 93 MB gradient dy1 never exists in HBM.
//   round 3:  sconv_bwd_x<APPLY> (read y1, WRITE dy1)  ->  tsconv_bwd_w (READ dy1 + token rows)  +  tsconv_bwd_x (READ dy1, write token gradients)
//   here:     work item = (sample b, block of 16 EEG rows h): for slabs of 16 channels, each wave forms dy1[c][16 h][36 w] of its channels on the
//             matrix cores exactly as sconv_bwd_x<APPLY, X3> does (dz^T = dy2^T Ws on split-bf16 products, ELU' / BatchNorm backward in the
//             epilogue) and drops it into the LDS slab dl[c][w][row]; behind a barrier ALL waves consume the slab twice:
//               * dS[row][j] += sum_{w,c} dy1[c][w][row] taps[c][j - 5w]      the banded GEMM of tsconv_bwd_x (accumulators live across the slabs)
//               * dW[c][t]   += sum_{row,w} dy1[c][w][row] S[row][5w + t]     the GEMM of tsconv_bwd_w over the item's box-filtered token rows
//             then dS -> transpose of the box filter -> token-row gradients.  HBM: y1 (93 MB) + token rows in, token gradients out.
// LDS 80 KB: taps 5 KB | BatchNorm coefficients | dy2^T planes 13.5 KB | slab 39 KB (later the dS rows / the reduction scratch) | S rows 16 KB |
// scan scratch 4 KB: two workgroups per CU.
constexpr int CB_CH = 16;                        // channels per slab: one MFMA M tile of the taps gradient
constexpr int CB_NSLAB = (TS_C + CB_CH - 1) / CB_CH;
constexpr int CB_WS = 17, CB_CS = 624;          // slab cell (c, w) = 16 rows + 1 pad; floats per channel (16 mod 32: conflict-free banded-GEMM reads)
struct cb_args {
    const float* dy2;
    const unsigned short *wt_hi, *wt_lo;         // Ws^T planes [(c, h)][64 o]
    const float* y1;
    bn_affine bn;
    const double *sums, *sums_param;
    double count;
    float *dgamma, *dbeta;
    const float* x;                              // token rows (the tsconv input)
    long long xs_b, xs_h;
    const float* w25;
    float* dx;                                   // token-row gradients, same strides as x
    float* partials;                             // [workgroup][40][25] taps-gradient partials
    int B, H, vec2;
};

__global__ __launch_bounds__(256, 2) void conv_bwd_fused_kernel(const cb_args a) {
    // Every wave PRODUCES its channels of a dy1 slab (apply tasks: y1 / weight loads, split-bf16 MFMAs, ELU' + BatchNorm-backward epilogue), then every
    // wave CONSUMES the slab (the two f32-MFMA GEMMs); two workgroups per CU run out of phase.  Measured at B = 256 (tools/bench_conv_bwd.py): 109 us against
    // 49 + 44 + 37 us for the three kernels it replaces, HBM traffic 141 MB against 404 (profiles/r4_pmc_conv_bwd_fused.json).  A producer / consumer
    // split (512 threads: 4 waves produce slab k + 1 into a second buffer while 4 consume slab k) measured SLOWER, 139 us: half the waves issuing loads
    // halves the bytes in flight, and the kernel is bound by exactly that (44 % of all wave cycles parked on loads / barriers, matrix pipe 31 % busy).
    EEG_LDS_BASE(float, lds);
    float* wl = lds;                                                             // [40][32] taps (cols >= 25 zero)
    float* coef = wl + TS_C * TSX_WL;                                             // [40][8] mean, rstd, gamma, beta, S1 / n, S2 / n per channel
    unsigned char* dplane = reinterpret_cast<unsigned char*>(coef + 8 * SC_C);    // two bf16 planes [48 w][144 B] of dy2^T
    float* dl0 = reinterpret_cast<float*>(dplane + 2 * SC_OP * SCX_RS);            // [16][624] slab
    float* sl = dl0 + CB_CH * CB_CS;                                              // [16][256] box-filtered token rows
    float* ps = sl + TSX_R * TS_XS;                                               // [4][256] scan scratch
    float* dsl = dl0;                                                             // [16][256] dS rows (after the last slab)
    float* red = dl0;                                                             // reduction scratch (after the last item)
    const int t = threadIdx.x, lane = t & 63, wv = wave_uniform(t >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int H = a.H, MT = (H + 15) / 16, nitems = a.B * MT;
    const f32x4 zero4v{0.f, 0.f, 0.f, 0.f};
    const bf16x8 zero8{0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = t; i < TS_C * TSX_WL; i += 256) wl[i] = (i % TSX_WL) < TS_K1 ? a.w25[(i / TSX_WL) * TS_K1 + i % TSX_WL] : 0.f;
    for (int i = t; i < SC_OP * 64; i += 256) {                                   // o >= 40 and w >= 36 of the planes: zeros, for every item
        const int w = i >> 6, o = i & 63;
        if (o >= SC_C || w >= SC_W) {
            *reinterpret_cast<unsigned short*>(dplane + w * SCX_RS + 2 * o) = 0;
            *reinterpret_cast<unsigned short*>(dplane + SC_OP * SCX_RS + w * SCX_RS + 2 * o) = 0;
        }
    }
    if (t < SC_C) {                                                               // (once per workgroup: the apply tasks read one 32-byte row instead of four
        float* cf = coef + 8 * t;                                                 //  scalar loads + two fp64 divisions each)
        cf[0] = a.bn.mean[t]; cf[1] = a.bn.rstd[t]; cf[2] = a.bn.gamma[t]; cf[3] = a.bn.beta[t];
        cf[4] = (float)(a.sums[t] / a.count); cf[5] = (float)(a.sums[SC_C + t] / a.count);
    }
    if (blockIdx.x == 0 && t < SC_C) {                                            // BatchNorm1 parameter gradients from this rank's own sums
        atomicAdd(a.dgamma + t, (float)a.sums_param[SC_C + t]);
        atomicAdd(a.dbeta + t, (float)a.sums_param[t]);
    }
    f32x4 accw[CB_NSLAB][2];                                                      // taps gradient D[c = 16 s + 4 g + r][t = 16 ut + fr]
#pragma unroll
    for (int s = 0; s < CB_NSLAB; ++s)
#pragma unroll
        for (int ut = 0; ut < 2; ++ut) accw[s][ut] = zero4v;

    // apply task of channel c for the item (b, h0): two tasks of loads in flight per wave (a task is ~600 cycles, an HBM round trip 2000+)
    bf16x8 nbh[2][2], nbl[2][2];
    f32x4 nyv[2][3];
    auto load_task = [&](int b, int h0, int c, auto slot_c) {
        constexpr int SL = decltype(slot_c)::value;
        const int h = h0 + fr;
        const bool hok = h < H;
        const long long row = ((long long)c * H + (hok ? h : H - 1)) * 64;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            nbh[SL][s2] = *reinterpret_cast<const bf16x8*>(a.wt_hi + row + 32 * s2 + 8 * g);
            nbl[SL][s2] = *reinterpret_cast<const bf16x8*>(a.wt_lo + row + 32 * s2 + 8 * g);
            if (!hok) { nbh[SL][s2] = zero8; nbl[SL][s2] = zero8; }
        }
        const float* yr = a.y1 + (((long long)b * SC_C + c) * H + h) * SC_W;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int w = 16 * j + 4 * g;
            nyv[SL][j] = (hok && w < SC_W) ? *reinterpret_cast<const f32x4*>(yr + w) : zero4v;
        }
    };
    if ((int)blockIdx.x < nitems) {                                               // the first item's first two tasks: in flight under its staging
        const int b = (int)blockIdx.x / MT, h0 = 16 * ((int)blockIdx.x - b * MT);
        load_task(b, h0, wv, std::integral_constant<int, 0>{});
        load_task(b, h0, wv + 4, std::integral_constant<int, 1>{});
    }

    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int b = item / MT, mt = item - b * MT, h0 = 16 * mt;
        const int nitem = item + (int)gridDim.x, nb = nitem / MT, nh0 = 16 * (nitem - nb * MT);       // (this workgroup's next item)
        __syncthreads();                                                          // the previous item is fully consumed (planes, S rows, dS rows)
        for (int i = t; i < SC_C * SC_W; i += 256) {                              // dy2 of the sample -> transposed hi | lo planes
            const int o = i / SC_W, w = i % SC_W;
            const float v = a.dy2[(long long)b * SC_C * SC_W + i];
            const unsigned short hb = f32_to_bf16_bits(v);
            *reinterpret_cast<unsigned short*>(dplane + w * SCX_RS + 2 * o) = hb;
            *reinterpret_cast<unsigned short*>(dplane + SC_OP * SCX_RS + w * SCX_RS + 2 * o) = f32_to_bf16_bits(v - bf16_bits_to_f32(hb));
        }
        {
            float vx[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int h = h0 + 4 * wv + j;
                load_row4(vx[j], a.x, a.xs_b, a.xs_h, h < H ? b * H + h : a.B * H, a.B * H, H, a.vec2 != 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) box_filter_row(sl + (4 * wv + j) * TS_XS, ps + wv * TS_XS, vx[j]);
        }
        __syncthreads();
        const bool hok = h0 + fr < H;
        f32x4 accx[4];                                                            // dS tile rows
#pragma unroll
        for (int q = 0; q < 4; ++q) accx[q] = zero4v;

        // (the slab index is a compile-time constant: with a runtime index the compiler turned the per-slab accumulator choice into a dynamically
        //  indexed private array -- 112 bytes of scratch per lane, a scratch load + store around every taps-gradient MFMA: 490 us)
        auto produce = [&](auto s_c) {
            constexpr int s = decltype(s_c)::value;
            constexpr int c0 = CB_CH * s, nch = TS_C - c0 < CB_CH ? TS_C - c0 : CB_CH;
            float* dl = dl0;
#pragma unroll
            for (int ti = 0; ti < nch / 4; ++ti) {
                const int cl = wv + 4 * ti, c = c0 + cl;
                const int SL = ti & 1;                                            // task index within the item = 4 s + ti (compile-time after unrolling)
                f32x4 acc[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[j] = zero4v;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {                                 // D[w = 16 j + 4 g + r][h = h0 + fr]
                        // dy2^T fragments (rows w = 16 j + fr, k = o = 32 s2 + 8 g ..) re-read from the planes per task: held in registers (48) next to the
                        // two prefetch slots and the consumers' accumulators they made the kernel spill
                        const bf16x8 ahf = *reinterpret_cast<const bf16x8*>(dplane + (16 * j + fr) * SCX_RS + 2 * (32 * s2 + 8 * g));
                        const bf16x8 alf = *reinterpret_cast<const bf16x8*>(dplane + SC_OP * SCX_RS + (16 * j + fr) * SCX_RS + 2 * (32 * s2 + 8 * g));
                        acc[j] = mfma_bf16_16x16x32(ahf, SL ? nbl[1][s2] : nbl[0][s2], acc[j]);
                        acc[j] = mfma_bf16_16x16x32(alf, SL ? nbh[1][s2] : nbh[0][s2], acc[j]);
                        acc[j] = mfma_bf16_16x16x32(ahf, SL ? nbh[1][s2] : nbh[0][s2], acc[j]);
                    }
                const f32x4 cf0 = *reinterpret_cast<const f32x4*>(coef + 8 * c);
                const f32x2 cf1 = *reinterpret_cast<const f32x2*>(coef + 8 * c + 4);
                const float mean = cf0[0], rstd = cf0[1], gam = cf0[2], bet = cf0[3], m1 = cf1[0], m2 = cf1[1];
                float* cell = dl + cl * CB_CS + (fr ^ (cl >> 1));                   // (row index XOR (channel >> 1): see the taps-gradient reads below)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int w = 16 * j + 4 * g;
                    if (w < SC_W) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float xh = ((SL ? nyv[1][j][r] : nyv[0][j][r]) - mean) * rstd;
                            const float u = gam * xh + bet;
                            const float da = u > 0.f ? acc[j][r] : acc[j][r] * fast_exp(u);
                            cell[(w + r) * CB_WS] = hok ? gam * rstd * (da - m1 - xh * m2) : 0.f;      // dy1[c][h][w + r]; rows past H are zero rows
                        }
                    }
                }
                // the task after next into the slot this task has emptied: channel c + 8 of this item, or -- at its last two tasks -- the first two
                // of the workgroup's next item (in flight under the last slab's GEMMs and the staging of that item)
                if (c + 8 < TS_C) {
                    if (SL) load_task(b, h0, c + 8, std::integral_constant<int, 1>{});
                    else load_task(b, h0, c + 8, std::integral_constant<int, 0>{});
                } else if (nitem < nitems) {
                    if (SL) load_task(nb, nh0, c + 8 - TS_C, std::integral_constant<int, 1>{});
                    else load_task(nb, nh0, c + 8 - TS_C, std::integral_constant<int, 0>{});
                }
            }
        };
        auto consume = [&](auto s_c) {
            constexpr int s = decltype(s_c)::value;
            constexpr int c0 = CB_CH * s, nch = TS_C - c0 < CB_CH ? TS_C - c0 : CB_CH;
            const float* dl = dl0;
            // ---- token-row gradient: banded GEMM over this slab (csrc/conv.hip: tsconv_bwd_x_kernel)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int jt = wv + 4 * q;                                        // j tile: j = 16 jt .. 16 jt + 15
                if (jt >= TSX_NJ) continue;
                int w_lo = (16 * jt - (TS_K1 - 1) + 4) / 5;
                if (16 * jt - (TS_K1 - 1) <= 0) w_lo = 0;
                int w_hi = (16 * jt + 15) / 5;
                if (w_hi > TS_W - 1) w_hi = TS_W - 1;
                const int j = 16 * jt + fr;
                for (int w = w_lo; w <= w_hi; ++w) {
                    const int u = j - 5 * w;
                    const bool inb = u >= 0 && u < TS_K1;
#pragma unroll
                    for (int cc = 0; cc < nch / 4; ++cc) {
                        const int cl = 4 * cc + g;
                        const float av = dl[cl * CB_CS + w * CB_WS + (fr ^ (cl >> 1))];            // A[row = fr][k = (w, c)]
                        const float bq = inb ? wl[(c0 + cl) * TSX_WL + u] : 0.f;                   // B[k][j] = taps[c][j - 5 w]
                        accx[q] = mfma_f32_16x16x4(av, bq, accx[q]);                               // D[row = 4 g + r][j]
                    }
                }
            }
            // ---- taps gradient: K = the item's 16 x 36 positions; MFMA step ks = (w, rq), its 4 k slots = rows rq, rq + 8, rq + 4, rq + 12 of position w.
            //      The lanes of an A read are 16 CHANNELS (stride 624 floats = 16 banks): with the rows stored at row ^ (channel >> 1) the 8 even and the 8
            //      odd channels of a half-wave spread over 8 banks each, and its two k slots (rows 8 apart) over the two 8-bank halves: conflict-free
            //      (plain rows: 8 lanes per bank, 50 % of all LDS cycles of the first version were bank conflicts -- SQ_LDS_BANK_CONFLICT)
            for (int ks = wv; ks < TSX_R * TS_W / 4; ks += 4) {
                const int w = ks >> 2, row = (ks & 3) + 4 * (g >> 1) + 8 * (g & 1);
                const float av = fr < nch ? dl[fr * CB_CS + w * CB_WS + (row ^ (fr >> 1))] : 0.f;  // A[c = fr][(w, row)]  (a short last slab: zero channels)
                const float* xp = sl + row * TS_XS + 5 * w + fr;
#pragma unroll
                for (int ut = 0; ut < 2; ++ut) accw[s][ut] = mfma_f32_16x16x4(av, xp[16 * ut], accw[s][ut]);      // B[(w, row)][t] = S[row][5 w + t]
            }
        };
        static_assert(CB_NSLAB == 3, "three slabs written out");
        produce(std::integral_constant<int, 0>{});
        __syncthreads();
        consume(std::integral_constant<int, 0>{});
        __syncthreads();                                                          // the slab has been consumed
        produce(std::integral_constant<int, 1>{});
        __syncthreads();
        consume(std::integral_constant<int, 1>{});
        __syncthreads();
        produce(std::integral_constant<int, 2>{});
        __syncthreads();
        consume(std::integral_constant<int, 2>{});
        __syncthreads();                                                          // every wave is done with the last slab: it becomes the dS rows
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int jt = wv + 4 * q;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (jt < TSX_NJ) dsl[(4 * g + r) * TS_XS + 16 * jt + fr] = accx[q][r];
        }
        __syncthreads();
        // transpose of the box filter, 4 rows per wave: Q[i] = sum_{k<i} dS[k] ; dx[i] = (Q[i+1] - Q[max(i-50, 0)]) / 51
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int rl = 4 * wv + rr, hh = h0 + rl;
            const f32x4 d = 4 * lane < 16 * TSX_NJ ? *reinterpret_cast<const f32x4*>(dsl + rl * TS_XS + 4 * lane) : zero4v;
            const float p0 = d[0], p1 = p0 + d[1], p2 = p1 + d[2], p3 = p2 + d[3];
            const float basev = wave_inclusive_scan(p3) - p3;
            float* pq = ps + wv * TS_XS;
            const float q4[5] = {basev, basev + p0, basev + p1, basev + p2, basev + p3};
            *reinterpret_cast<f32x4*>(pq + 4 * lane) = f32x4{q4[0], q4[1], q4[2], q4[3]};
            wave_sync();
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * lane + e;
                const float lo = i >= TS_POOL - 1 ? pq[i - (TS_POOL - 1)] : 0.f;
                o[e] = (q4[e + 1] - lo) * (1.0f / TS_POOL);
            }
            wave_sync();
            if (hh < H) {
                float* xr = a.dx + b * a.xs_b + hh * a.xs_h;
                if (a.vec2) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
                        if (4 * lane + 2 * hf < TS_T) *reinterpret_cast<f32x2*>(xr + 4 * lane + 2 * hf) = f32x2{o[2 * hf], o[2 * hf + 1]};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (4 * lane + e < TS_T) xr[4 * lane + e] = o[e];
                }
            }
        }
    }
    // cross-wave sum of the four waves' position-partial taps gradients (as tsconv_bwd_w_kernel), then this workgroup's partial row
    constexpr int RLD = 36;
    auto put = [&](float* reg) {
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int ut = 0; ut < 2; ++ut)
#pragma unroll
                for (int r = 0; r < 4; ++r) reg[(16 * ct + 4 * g + r) * RLD + 16 * ut + fr] = accw[ct][ut][r];
    };
    auto add = [&](const float* reg) {
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int ut = 0; ut < 2; ++ut)
#pragma unroll
                for (int r = 0; r < 4; ++r) accw[ct][ut][r] += reg[(16 * ct + 4 * g + r) * RLD + 16 * ut + fr];
    };
    static_assert(2 * TS_CP * RLD <= CB_CH * CB_CS, "reduction scratch must fit in the slab");
    __syncthreads();
    if (wv >= 2) put(red + (wv - 2) * TS_CP * RLD);
    __syncthreads();
    if (wv < 2) add(red + wv * TS_CP * RLD);
    __syncthreads();
    if (wv == 1) put(red);
    __syncthreads();
    if (wv == 0) {
        add(red);
        float* out = a.partials + (long long)blockIdx.x * (TS_C * TS_K1);
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int ut = 0; ut < 2; ++ut)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * ct + 4 * g + r, u = 16 * ut + fr;
                    if (c < TS_C && u < TS_K1) out[c * TS_K1 + u] = accw[ct][ut][r];
                }
    }
}

}  // namespace eeg

using namespace eeg;

static int ts_check(int B, int H, int T, int C) {
    return (B < 1 || H < 1 || T != TS_T || C != TS_C) ? EEGCLIP_EINVAL : 0;
}
static int ts_vec2(const float* x, long long xs_b, long long xs_h) {
    return ((reinterpret_cast<uintptr_t>(x) & 7u) == 0 && (xs_b & 1) == 0 && (xs_h & 1) == 0) ? 1 : 0;
}

static int tsf_grid(int B, int H) {
    const int nchunks = (B * H + TSF_R - 1) / TSF_R;
    return nchunks < 1024 ? nchunks : 1024;
}
extern "C" long long eegclip_tsconv_fwd_workspace_floats(int B, int H) { return B < 1 || H < 1 ? 0 : 2LL * tsf_grid(B, H) * 2 * TS_C; }   // (doubles, in floats)

extern "C" int eegclip_tsconv_fwd(const float* x, long long xs_b, long long xs_h, const float* w25, const float* bias, float* y,
                                  int B, int H, int T, int C, double* sums, float* workspace, void* stream) {
    if (int rc = ts_check(B, H, T, C)) return rc;
    if (!x || !w25 || !bias || !y) return EEGCLIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(workspace) & 7u) return EEGCLIP_EALIGN;
    const int grid = tsf_grid(B, H);
    const size_t lds = (TS_KP * TS_CP + TSF_R * TS_XS + 4 * TS_XS + 8 * TS_CP) * sizeof(float);
    double* parts = sums ? reinterpret_cast<double*>(workspace) : nullptr;
    EEG_LAUNCH(tsconv_fwd_kernel, dim3(grid), dim3(256), lds, stream, x, xs_b, xs_h, w25, bias, y, B, H, sums, parts, ts_vec2(x, xs_b, xs_h));
    if (parts) EEG_COLSUM_F64((const double*)parts, grid, 2 * TS_C, sums, stream);
    return (int)hipGetLastError();
}

// rows per work item: 7 (63 KB of LDS, two workgroups per CU, 63 = 9 x 7 rows split evenly; 6 measured no faster)
static int tsw_rows() { return 7; }
static int tsw_grid(int B, int H) {
    const int r = tsw_rows();
    const int items = B * ((H + r - 1) / r), cap = 512;
    return items < cap ? items : cap;
}

extern "C" long long eegclip_tsconv_bwd_w_workspace_floats(int B, int H) { return (long long)tsw_grid(B, H) * TS_C * TS_K1; }

template <int R>
static void tsw_launch(int grid, void* stream, const float* x, long long xs_b, long long xs_h, const float* dy, float* workspace, int B, int H) {
    const size_t lds = (TS_CP * (R * TS_W + 17) + R * TS_XS + 4 * TS_XS) * sizeof(float);
    EEG_LAUNCH(tsconv_bwd_w_kernel<R>, dim3(grid), dim3(256), lds, stream, x, xs_b, xs_h, dy, workspace, B, H, ts_vec2(x, xs_b, xs_h));
}

extern "C" int eegclip_tsconv_bwd_w(const float* x, long long xs_b, long long xs_h, const float* dy, float* dw25, float* workspace, int B,
                                    int H, int T, int C, void* stream) {
    if (int rc = ts_check(B, H, T, C)) return rc;
    if (!x || !dy || !dw25 || !workspace) return EEGCLIP_EINVAL;
    const int grid = tsw_grid(B, H);
    tsw_launch<7>(grid, stream, x, xs_b, xs_h, dy, workspace, B, H);
    EEG_LAUNCH(tsconv_bwd_w_reduce_kernel, dim3((TS_C * TS_K1 + 255) / 256, grid < 32 ? grid : 32), dim3(256), 0, stream, workspace, grid, dw25);
    return (int)hipGetLastError();
}

extern "C" int eegclip_tsconv_bwd_x(const float* dy, const float* w25, float* dx, long long xs_b, long long xs_h, int B, int H, int T,
                                    int C, void* stream) {
    if (int rc = ts_check(B, H, T, C)) return rc;
    if (!dy || !w25 || !dx) return EEGCLIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(dy) & 15u) return EEGCLIP_EALIGN;
    if ((long long)B * TS_C * H * TS_W >= (1LL << 31)) return EEGCLIP_EINVAL;
    const int items = (B * H + TSX_R - 1) / TSX_R;
    const size_t lds = (TS_C * TSX_WL + TSX_CH * TSX_CS + 4 * TS_XS) * sizeof(float);
    EEG_LAUNCH(tsconv_bwd_x_kernel, dim3(items < 1280 ? items : 1280), dim3(256), lds, stream, dy, w25, dx, xs_b, xs_h, B, H, ts_vec2(dx, xs_b, xs_h));
    return (int)hipGetLastError();
}

static int cbf_grid(int B, int H, int limit) {
    const int items = B * ((H + 15) / 16), cap = limit > 0 ? limit : 512;      // default: two 80 KB workgroups per CU
    return items < cap ? items : cap;
}
extern "C" long long eegclip_conv_bwd_fused_workspace_floats(int B, int H) { return (B < 1 || H < 1) ? 0 : (long long)cbf_grid(B, H, 1 << 30) * TS_C * TS_K1; }

extern "C" int eegclip_conv_bwd_fused(const float* dy2, const void* WsT_hi, const void* WsT_lo, const float* y1, const float* mean, const float* rstd,
                                      const float* gamma, const float* beta, const double* sums, const double* sums_local, double count, float* dgamma,
                                      float* dbeta, const float* x, long long xs_b, long long xs_h, const float* w25, float* dx, float* dw25,
                                      float* workspace, int B, int H, int max_workgroups, void* stream) {
    if (B < 1 || H < 1 || H > 64 || max_workgroups < 0) return EEGCLIP_EINVAL;
    if (!dy2 || !WsT_hi || !WsT_lo || !y1 || !mean || !rstd || !gamma || !beta || !sums || !dgamma || !dbeta || !x || !w25 || !dx || !dw25 || !workspace ||
        count < 1.0)
        return EEGCLIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(y1) | reinterpret_cast<uintptr_t>(WsT_hi) | reinterpret_cast<uintptr_t>(WsT_lo)) & 15u) return EEGCLIP_EALIGN;
    const cb_args a{dy2, static_cast<const unsigned short*>(WsT_hi), static_cast<const unsigned short*>(WsT_lo), y1, bn_affine{mean, rstd, gamma, beta},
                    sums, sums_local ? sums_local : sums, count, dgamma, dbeta, x, xs_b, xs_h, w25, dx, workspace, B, H,
                    (ts_vec2(x, xs_b, xs_h) && ts_vec2(dx, xs_b, xs_h)) ? 1 : 0};
    const int grid = cbf_grid(B, H, max_workgroups);
    const size_t lds = (TS_C * TSX_WL + CB_CH * CB_CS + TSX_R * TS_XS + 4 * TS_XS + 8 * SC_C) * sizeof(float) + 2 * SC_OP * SCX_RS;
    EEG_LAUNCH(conv_bwd_fused_kernel, dim3(grid), dim3(256), lds, stream, a);
    EEG_LAUNCH(tsconv_bwd_w_reduce_kernel, dim3((TS_C * TS_K1 + 255) / 256, grid < 32 ? grid : 32), dim3(256), 0, stream, workspace, grid, dw25);
    return (int)hipGetLastError();
}
