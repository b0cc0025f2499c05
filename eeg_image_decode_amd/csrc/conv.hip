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

