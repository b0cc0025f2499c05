// The spatial stage of tsconv, fused around the only HBM-heavy tensor of the network, y1 = conv+pool output (B,40,H=63,36) fp32
// (93 MB at B=256):      BatchNorm2d(40) -> ELU -> Conv2d(40,40,(63,1))        (Retrieval/ATMS_retrieval.py:104-106)
//
// The reference materialises BN(y1), ELU(.) and their gradients (5 more tensors of that size per direction).  Here y1 is the ONLY
// big tensor that exists: z1 = ELU(BN(y1)) is re-evaluated inside the operand staging of every kernel that needs it, and the
// gradient w.r.t. z1 (a K=40 contraction) is recomputed on the matrix cores instead of being written and re-read:
//   sconv_fwd          y2[b,o,w]   = bs[o] + sum_{c,h} Ws[o,c,h] * z1[b,c,h,w]          reads y1 once (+ BN2 batch sums of the tiny y2)
//   sconv_bwd_w        dWs[o,c,h] += sum_{b,w} dy2[b,o,w] * z1[b,c,h,w]                  reads y1 once
//   sconv_bwd_x<false> BN1 backward sums of da = (Ws^T dy2) * ELU'(BN(y1))               reads y1 once
//   sconv_bwd_x<true>  dy1 = BN1 backward(da)                                            reads y1 once, writes dy1 once
// Contractions: v_mfma_f32_16x16x4_f32 (exact fp32), or -- sconv_bwd_x with pre-split weight planes -- split-bf16 products on
// v_mfma_f32_16x16x32_bf16.  HBM traffic: 5 passes over the 93 MB tensor per step instead of 14.
#include "conv_common.h"

#include <stdlib.h>

namespace eeg {


// ---------------------------------------------------------------------------------------------------------------------------
// forward: workgroup = (sample, K slice); K = (c,h) = 40*H is cut into SCF_KS slices, each streamed in chunks of 128 k, MFMA wave v owning
// k = 32v .. 32v+31 of a chunk.
//   B operand (activations): the y1 chunk [128 k][36 w] fp32 is 18 KB that lie contiguously in y1: copied as they are global -> LDS by LDS-DMA from a
//       PRODUCER wave (18 instructions of 1 KB, two stages); the MFMA waves read RAW y1 values as their fragment and evaluate z1 = ELU(BN(y1)) on it in
//       registers (every element is the B operand of exactly one lane), software-pipelined by one k-step and written branch-free.
//       Round 3's kernel staged the chunk through registers (load -> ELU -> ds_write, two barriers per chunk); its phases added up (42 / 46 / 46 us with
//       one of load | convert + LDS store | MFMA switched off, 60 with all).  Round 4 measured on this form: the DMA stream alone runs at 5.5 TB/s (17 us);
//       fragment reads + ELU add 15 us, the fp32 MFMAs another 13 -- the MFMA waves are the bound; 62 -> 57 us in the step (4 -> 2 K slices: 10 chunks per
//       workgroup amortise the pipeline fill), not the 30 us the ablation of round 3 suggested.
//   A operand (weights):     straight from global/L2 into registers in MFMA order -- lane (fr, g) fetches Ws[o = 16i + fr][k .. k+3] as
//       one 16-byte load and feeds four k-steps (the k-slot <-> lane-group assignment is free as long as both operands agree).
// The 4 MFMA waves keep private 3x3 accumulator tiles (o x w), combined through LDS at the end; the K slices leave slabs that
// sconv_merge_stats2_kernel sums (1440 device-scope float atomics per workgroup were 33 of 60 us).
constexpr int SCF_KS = 2;
constexpr int SCF_KC = 128;            // k rows per chunk
constexpr int SCD_NS = 2;
constexpr int SCD_STAGE = SCF_KC * SC_W * 4;             // bytes of a chunk: 18432
constexpr int SCD_DMA = SCD_STAGE / 1024;                // LDS-DMA instructions per chunk: 18
constexpr int SCD_LDS = SCD_NS * SCD_STAGE + 256 + 2 * SC_C * 4;
template <int NPW>
__global__ __launch_bounds__(256 + 64 * NPW, 3) void sconv_fwd_kernel(const float* __restrict__ y1, const bn_affine bn, const float* __restrict__ Ws,
                                                                const float* __restrict__ bs, float* __restrict__ y2, float* __restrict__ slabs, int B,
                                                                int H, int kper) {
    EEG_LDS_BASE(unsigned char, ldsb);
    float* aff = reinterpret_cast<float*>(ldsb + SCD_NS * SCD_STAGE + 256);      // [2][40] (256 bytes of slack behind the ring: the w >= 36 lanes of the
    float* red = reinterpret_cast<float*>(ldsb);                                 //  last row read past it)
    const int t = threadIdx.x, lane = t & 63, wv = wave_uniform(t >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int b = blockIdx.x;
    const int K = SC_C * H;
    const int kbeg = blockIdx.y * kper, kend = kbeg + kper < K ? kbeg + kper : K;
    const int nch = kend > kbeg ? (kend - kbeg + SCF_KC - 1) / SCF_KC : 0;
    if (t < SC_C) {
        const float sc = bn.gamma[t] * bn.rstd[t];
        aff[t] = sc;
        aff[SC_C + t] = bn.beta[t] - bn.mean[t] * sc;
    }
    __syncthreads();
    const float* yb = y1 + (long long)b * K * SC_W;
    if (wv >= 4) {                                          // ---------------- producers: wave p issues instructions p, p + NPW, ...
        constexpr int IPP = SCD_DMA / NPW;
        static_assert(SCD_DMA % NPW == 0, "18 DMA instructions per chunk");
        const int pw = wv - 4;
        const unsigned char* src0 = reinterpret_cast<const unsigned char*>(yb + (long long)kbeg * SC_W);
        auto issue = [&](int c) {
            const int rows = kend - kbeg - SCF_KC * c;
            unsigned char* st = ldsb + (c % SCD_NS) * SCD_STAGE;
            const unsigned char* sc_ = src0 + (long long)c * SCD_STAGE + 16 * lane;
            if (rows >= SCF_KC) {                           // (wave-uniform) a full chunk: no clamping
#pragma unroll
                for (int i = 0; i < IPP; ++i) lds_dma16(st + 1024 * (pw + NPW * i), sc_ + 1024 * (pw + NPW * i));
            } else {
                const int valid = rows * SC_W * 4;
#pragma unroll
                for (int i = 0; i < IPP; ++i) {
                    int off = 1024 * (pw + NPW * i) + 16 * lane;
                    off = off < valid ? off : valid - 16;   // past the slice: re-read its last 16 bytes (those rows are masked by the MFMA waves) -- never past y1
                    lds_dma16(st + 1024 * (pw + NPW * i), src0 + (long long)c * SCD_STAGE + off);
                }
            }
        };
#pragma unroll
        for (int p = 0; p < SCD_NS; ++p)
            if (p < nch) issue(p);
        for (int c = 0; c < nch; ++c) {
            const int fly = c == 0 ? (nch - 1 < SCD_NS - 1 ? nch - 1 : SCD_NS - 1) : (nch - 1 - c < SCD_NS - 2 ? nch - 1 - c : SCD_NS - 2);
            if (fly >= 3) wait_vmcnt<3 * IPP>();
            else if (fly == 2) wait_vmcnt<2 * IPP>();
            else if (fly == 1) wait_vmcnt<IPP>();
            else wait_vmcnt<0>();
            raw_barrier();                                  // chunk c has landed; the MFMA waves are done with chunk c - 1
            if (c >= 1 && c - 1 + SCD_NS < nch) issue(c - 1 + SCD_NS);
        }
        return;
    }
    f32x4 acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 va[3][2];
    const f32x4 zero4v{0.f, 0.f, 0.f, 0.f};
    auto load_weights = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int o = 16 * i + fr, k = k0 + 32 * wv + 16 * h + 4 * g;
                va[i][h] = (o < SC_C && k < kend) ? *reinterpret_cast<const f32x4*>(Ws + (long long)o * K + k) : zero4v;
            }
    };
    if (nch > 0) load_weights(kbeg);
    const float inv_h = 1.0f / (float)H;
    for (int c = 0; c < nch; ++c) {
        const int k0 = kbeg + SCF_KC * c;
        raw_barrier();
        f32x4 a[3][2];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) a[i][h] = va[i][h];
        if (c + 1 < nch) load_weights(k0 + SCF_KC);
        const float* st = reinterpret_cast<const float*>(ldsb + (c % SCD_NS) * SCD_STAGE);
        // k-step hs = (h, s4) of this wave's 32 k: software-pipelined by one step -- the LDS reads of step hs + 1 are issued before the MFMAs of step hs and
        // its ELU is evaluated behind them (the matrix pipe works through the 9 MFMAs meanwhile).  Written branch-free: a conditional around the channel
        // lookup / the ELU became an execz branch per k-step and serialised read -> wait -> ELU -> MFMA (66 us against the 56 of sconv_fwd_kernel).
        float raw[2][3], scv[2], shv[2];
        bool okv[2];
        auto ld = [&](int hs, int slot) {
            const int kk = 32 * wv + 16 * (hs >> 2) + 4 * g + (hs & 3), k = k0 + kk;
            okv[slot] = k < kend;
            const int kc = okv[slot] ? k : kend - 1;
            const int ch = (int)(((float)kc + 0.5f) * inv_h);      // k / H, exact for k < 2^20
            scv[slot] = aff[ch];
            shv[slot] = aff[SC_C + ch];
            const float* zp = st + kk * SC_W + fr;
#pragma unroll
            for (int j = 0; j < 3; ++j) raw[slot][j] = zp[16 * j];
        };
        auto fin = [&](int slot, float (&bv)[3]) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float u = raw[slot][j] * scv[slot] + shv[slot];
                const float e = fast_exp(u < 0.f ? u : 0.f) - 1.0f;
                const float z = u > 0.f ? u : e;                 // z1 = ELU(BN(y1)) on the fragment (columns w >= 36 hold neighbours' values: their
                bv[j] = okv[slot] ? z : 0.f;                     //  output columns are dropped below)
            }
        };
        float bv[2][3];
        ld(0, 0);
        fin(0, bv[0]);
#pragma unroll
        for (int hs = 0; hs < 8; ++hs) {
            const int cur = hs & 1, nxt = cur ^ 1;
            if (hs + 1 < 8) ld(hs + 1, nxt);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    acc[i][j] = mfma_f32_16x16x4(a[i][hs >> 2][hs & 3], bv[cur][j], acc[i][j]);      // D[o = 16i+4g+r][w = 16j+fr]
                }
            if (hs + 1 < 8) fin(nxt, bv[nxt]);
#if !defined(EEG_EMU)
            // issue order of the step: the LDS reads of step hs + 1 first, then its ~45 vector instructions dealt out BETWEEN the 9 MFMAs of step hs
            // (a wave issues in order and each fp32 MFMA holds the matrix pipe for 32 cycles: vector work placed behind the MFMAs waits for all nine)
            __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            }
#endif
        }
    }
    // cross-wave sum of the four k-partial accumulator sets: a two-level tree through LDS with plain stores (waves 2,3 -> 0,1, then 1 -> 0); the ring is
    // dead, LDS becomes reduction scratch.  (36 ds_add_f32 per lane into one tile: LDS float atomics retire at ~170 cycles per wave instruction.)
    constexpr int RLD = 52;
    auto put = [&](float* reg) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) reg[(16 * i + 4 * g + r) * RLD + 16 * j + fr] = acc[i][j][r];
    };
    auto add = [&](const float* reg) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += reg[(16 * i + 4 * g + r) * RLD + 16 * j + fr];
    };
    __syncthreads();
    if (wv >= 2) put(red + (wv - 2) * SC_OP * RLD);
    __syncthreads();
    if (wv < 2) add(red + wv * SC_OP * RLD);
    __syncthreads();
    if (wv == 1) put(red);
    __syncthreads();
    if (wv == 0) {
        add(red);
        float* yo = slabs ? slabs + ((long long)blockIdx.y * B + b) * SC_C * SC_W : y2 + (long long)b * SC_C * SC_W;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = 16 * i + 4 * g + r, w = 16 * j + fr;
                    if (o < SC_C && w < SC_W) {
                        const float v = acc[i][j][r] + (blockIdx.y == 0 ? bs[o] : 0.f);
                        if (slabs) yo[o * SC_W + w] = v;
                        else atomicAdd(yo + o * SC_W + w, v);
                    }
                }
    }
}

// y2 = sum of the K-slice slabs of sconv_fwd (slabs != NULL), and the BatchNorm2d #2 batch statistics of y2 (B,40,36) (sums2 != NULL):
// workgroup = (channel, slice of samples); fp64 atomics, 2 per workgroup
__global__ __launch_bounds__(256) void sconv_merge_stats2_kernel(const float* __restrict__ slabs, int nslab, float* __restrict__ y2,
                                                                  double* __restrict__ sums2, int B) {
    EEG_LDS_BASE(double, sh);                 // [2][4] per-wave partial sums
    const int o = blockIdx.x, t = threadIdx.x;
    const long long slab_stride = (long long)B * SC_C * SC_W;
    double s = 0.0, q = 0.0;
    for (int i = blockIdx.y * 256 + t; i < B * SC_W; i += gridDim.y * 256) {
        const long long e = ((long long)(i / SC_W) * SC_C + o) * SC_W + i % SC_W;
        float v;
        if (slabs) {
            v = slabs[e];
            for (int k = 1; k < nslab; ++k) v += slabs[k * slab_stride + e];       // slice order: a result independent of scheduling
            y2[e] = v;
        } else {
            v = y2[e];
        }
        s += v;
        q += (double)v * v;
    }
    if (!sums2) return;
    s = wave_sum(s);
    q = wave_sum(q);
    if ((t & 63) == 0) { sh[t >> 6] = s; sh[4 + (t >> 6)] = q; }
    __syncthreads();
    if (t == 0) {
        atomicAdd(sums2 + o, sh[0] + sh[1] + sh[2] + sh[3]);
        atomicAdd(sums2 + SC_C + o, sh[4] + sh[5] + sh[6] + sh[7]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// weight gradient: workgroup (n-slab of NS (c,h) columns, group of samples); per sample K = 36 positions; accumulators 3 x NS/16 tiles
// spread over the 4 waves; partial results go to the workspace, a second kernel sums the sample groups.  NS = 128 and ~1000
// workgroups (4 per CU) instead of 256-column slabs x 32 groups = 320 workgroups (1.25 per CU, two rounds on a quarter of the chip).
constexpr int SCW_L = 37;                    // LDS row stride for 36-float rows
template <int NS>
__global__ __launch_bounds__(256) void sconv_bwd_w_kernel(const float* __restrict__ y1, const bn_affine bn, const float* __restrict__ dy2,
                                                           float* __restrict__ partials, int B, int H, int bgroups) {
    constexpr int NT = NS / 64;               // n-tiles per wave
    constexpr int NV = (NS * SC_W / 4 + 255) / 256;   // float4 per thread of a slab (NS * 36 contiguous floats)
    EEG_LDS_BASE(float, lds);
    float* zl = lds;                          // [NS][37]  z1[n0 + n][w]
    float* dl = zl + NS * SCW_L;              // [48][37]   dy2[o][w]   (rows >= 40 zero)
    float* aff = dl + SC_OP * SCW_L;          // [2][40]
    const int t = threadIdx.x, lane = t & 63, wv = wave_uniform(t >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int K = SC_C * H;
    const int n0 = blockIdx.x * NS, bg = blockIdx.y;
    if (t < SC_C) {
        const float sc = bn.gamma[t] * bn.rstd[t];
        aff[t] = sc;
        aff[SC_C + t] = bn.beta[t] - bn.mean[t] * sc;
    }
    for (int i = t; i < SC_OP * SCW_L; i += 256) dl[i] = 0.f;
    f32x4 acc[3][NT];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ncols = K - n0 < NS ? K - n0 : NS;
    const f32x4 zero4v{0.f, 0.f, 0.f, 0.f};
    f32x4 vz[NV];
    float vd[6];
    auto load_sample = [&](int b) {
        const float* src = y1 + ((long long)b * K + n0) * SC_W;      // ncols*36 contiguous floats, 16-byte aligned
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int e = 4 * (t + 256 * j);
            vz[j] = (e < ncols * SC_W) ? *reinterpret_cast<const f32x4*>(src + e) : zero4v;
        }
        const float* dsrc = dy2 + (long long)b * SC_C * SC_W;
#pragma unroll
        for (int j = 0; j < 6; ++j) { const int i = t + 256 * j; vd[j] = i < SC_C * SC_W ? dsrc[i] : 0.f; }
    };
    auto store_sample = [&]() {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int e = 4 * (t + 256 * j), n = e / SC_W, w = e % SC_W;
            if (e >= NS * SC_W) continue;
            float sc = 0.f, sh = 0.f;
            if (n < ncols) {
                const int c = (n0 + n) / H;
                sc = aff[c];
                sh = aff[SC_C + c];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) zl[n * SCW_L + w + q] = n < ncols ? elu1_fast(vz[j][q] * sc + sh) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) { const int i = t + 256 * j; if (i < SC_C * SC_W) dl[(i / SC_W) * SCW_L + i % SC_W] = vd[j]; }
    };
    if (bg < B) load_sample(bg);
    for (int b = bg; b < B; b += bgroups) {
        __syncthreads();
        store_sample();
        __syncthreads();
        if (b + bgroups < B) load_sample(b + bgroups);          // next sample's loads land under this sample's MFMAs
#pragma unroll
        for (int kk = 0; kk < SC_W / 4; ++kk) {
            const int kq = 4 * kk + g;
            float av[3], bv[NT];
#pragma unroll
            for (int i = 0; i < 3; ++i) av[i] = dl[(16 * i + fr) * SCW_L + kq];
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = zl[(16 * NT * wv + 16 * j + fr) * SCW_L + kq];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma_f32_16x16x4(av[i], bv[j], acc[i][j]);      // D[o][n = n0 + 16 NT wv + 16j + fr]
        }
    }
    float* out = partials + (long long)bg * SC_C * K;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + 16 * NT * wv + 16 * j + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 16 * i + 4 * g + r;
                if (o < SC_C && n < K) out[(long long)o * K + n] = acc[i][j][r];
            }
        }
}

// weight gradient with split-bf16 products: both operands are contiguous along the contraction index (the 36 positions w of one sample), so the
// staging is a plain split -- 4 consecutive w -> one 8-byte LDS store per plane -- into planes [rows][64 w] (w >= 36 zero; row stride 144
// bytes: conflict-free 16-byte fragment reads).  Per sample and wave 3 x NT x 2 x 3 MFMAs of 16 cycles (K = 36 padded to 64) instead of
// 3 x NT x 9 exact-fp32 ones of 32 cycles.
constexpr int SWX_RS = 144;
template <int NS>
__global__ __launch_bounds__(256, 3) void sconv_bwd_w_x3_kernel(const float* __restrict__ y1, const bn_affine bn, const float* __restrict__ dy2,
                                                              float* __restrict__ partials, int B, int H, int bgroups) {
    constexpr int NT = NS / 64;
    constexpr int NV = (NS * SC_W / 4 + 255) / 256;
    constexpr int ZPL = NS * SWX_RS, DPL = SC_OP * SWX_RS;       // bytes per plane
    EEG_LDS_BASE(float, lds);
    unsigned char* zp = reinterpret_cast<unsigned char*>(lds);   // z1 planes hi | lo   [NS n][64 w]
    unsigned char* dp = zp + 2 * ZPL;                            // dy2 planes hi | lo  [48 o][64 w]
    float* aff = reinterpret_cast<float*>(dp + 2 * DPL);         // [2][40]
    const int t = threadIdx.x, lane = t & 63, wv = wave_uniform(t >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int K = SC_C * H;
    const int n0 = blockIdx.x * NS, bg = blockIdx.y;
    if (t < SC_C) {
        const float sc = bn.gamma[t] * bn.rstd[t];
        aff[t] = sc;
        aff[SC_C + t] = bn.beta[t] - bn.mean[t] * sc;
    }
    for (int i = t; i < (2 * ZPL + 2 * DPL) / 16; i += 256) reinterpret_cast<f32x4*>(zp)[i] = f32x4{0.f, 0.f, 0.f, 0.f};     // w >= 36, o >= 40: zero for good
    f32x4 acc[3][NT];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ncols = K - n0 < NS ? K - n0 : NS;
    const f32x4 zero4v{0.f, 0.f, 0.f, 0.f};
    f32x4 vz[NV], vd[2];
    auto load_sample = [&](int b) {
        const float* src = y1 + ((long long)b * K + n0) * SC_W;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int e = 4 * (t + 256 * j);
            vz[j] = (e < ncols * SC_W) ? *reinterpret_cast<const f32x4*>(src + e) : zero4v;
        }
        const float* dsrc = dy2 + (long long)b * SC_C * SC_W;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = 4 * (t + 256 * j);
            vd[j] = e < SC_C * SC_W ? *reinterpret_cast<const f32x4*>(dsrc + e) : zero4v;
        }
    };
    auto store_sample = [&]() {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int e = 4 * (t + 256 * j), n = e / SC_W, w = e % SC_W;
            if (e >= NS * SC_W) continue;
            f32x4 z = zero4v;
            if (n < ncols) {
                const int c = (n0 + n) / H;
                const float sc = aff[c], sh = aff[SC_C + c];
#pragma unroll
                for (int q = 0; q < 4; ++q) z[q] = elu1_fast(vz[j][q] * sc + sh);
            }
            u32x2_t hi, lo;
            x3_split4(z[0], z[1], z[2], z[3], hi, lo);
            *reinterpret_cast<u32x2_t*>(zp + n * SWX_RS + 2 * w) = hi;
            *reinterpret_cast<u32x2_t*>(zp + ZPL + n * SWX_RS + 2 * w) = lo;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = 4 * (t + 256 * j), o = e / SC_W, w = e % SC_W;
            if (e >= SC_C * SC_W) continue;
            u32x2_t hi, lo;
            x3_split4(vd[j][0], vd[j][1], vd[j][2], vd[j][3], hi, lo);
            *reinterpret_cast<u32x2_t*>(dp + o * SWX_RS + 2 * w) = hi;
            *reinterpret_cast<u32x2_t*>(dp + DPL + o * SWX_RS + 2 * w) = lo;
        }
    };
    if (bg < B) load_sample(bg);
    for (int b = bg; b < B; b += bgroups) {
        __syncthreads();
        store_sample();
        __syncthreads();
        if (b + bgroups < B) load_sample(b + bgroups);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 ah[3], al[3], bh[NT], bl[NT];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const unsigned char* src = dp + (16 * i + fr) * SWX_RS + 2 * (32 * s2 + 8 * g);
                ah[i] = *reinterpret_cast<const bf16x8*>(src);
                al[i] = *reinterpret_cast<const bf16x8*>(src + DPL);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const unsigned char* src = zp + (16 * NT * wv + 16 * j + fr) * SWX_RS + 2 * (32 * s2 + 8 * g);
                bh[j] = *reinterpret_cast<const bf16x8*>(src);
                bl[j] = *reinterpret_cast<const bf16x8*>(src + ZPL);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {                // D[o = 16i + 4g + r][n = n0 + 16 NT wv + 16j + fr]
                    acc[i][j] = mfma_bf16_16x16x32(ah[i], bl[j], acc[i][j]);
                    acc[i][j] = mfma_bf16_16x16x32(al[i], bh[j], acc[i][j]);
                    acc[i][j] = mfma_bf16_16x16x32(ah[i], bh[j], acc[i][j]);
                }
        }
    }
    float* out = partials + (long long)bg * SC_C * K;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + 16 * NT * wv + 16 * j + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 16 * i + 4 * g + r;
                if (o < SC_C && n < K) out[(long long)o * K + n] = acc[i][j][r];
            }
        }
}

// dW[i] += sum over the sample groups, in a FIXED order: workgroup = 64 elements x 4 group slices (thread (sg, e) sums groups sg, sg + 4, ...; the four
// slices meet in LDS and are added as (g0 + g1) + (g2 + g3)).  Round 3 split the groups over gridDim.y and combined the slices with float atomics --
// a scheduling-dependent sum; one thread per element walking all ~51 groups left most of the chip idle.
__global__ __launch_bounds__(256) void sconv_bwd_w_reduce_kernel(const float* __restrict__ partials, int groups, long long n, float* __restrict__ dW) {
    EEG_LDS_BASE(float, red);                                // [4][64]
    const int t = threadIdx.x, sg = t >> 6, e = t & 63;
    const long long i = (long long)blockIdx.x * 64 + e;
    float s = 0.f;
    if (i < n) {
#pragma unroll 8
        for (int k = sg; k < groups; k += 4) s += partials[(long long)k * n + i];      // independent loads: 8 in flight per thread
    }
    red[sg * 64 + e] = s;
    __syncthreads();
    if (sg == 0 && i < n) dW[i] += (red[e] + red[64 + e]) + (red[128 + e] + red[192 + e]);
}

// ---------------------------------------------------------------------------------------------------------------------------
// input gradient + BatchNorm1 backward.  gridDim.y workgroups per sample; wave tasks = (channel c, 16-row block of h):
// dz^T[w][h] = sum_o dy2[o][w] * Ws[o][c][h] on the matrix cores (K = 40), then da = dz * ELU'(u) with u = BN(y1).  The product is
// formed TRANSPOSED (MFMA rows = positions w, columns = rows h) so that the 4 accumulator registers of a lane are 4 consecutive w of
// one row: y1 is read and dy1 written as one 16-byte access per lane and tile instead of four 4-byte ones.
//   APPLY = false: accumulate sum(da), sum(da * xhat) per channel (LDS, then 80 fp64 atomics per workgroup)
//   APPLY = true : dy1 = gamma * rstd * (da - S1/n - xhat * S2/n)
// X3: the K = 40 contraction as split-bf16 products on v_mfma_f32_16x16x32_bf16 (same arithmetic as csrc/gemm_x3.hip: hi*lo + lo*hi + hi*hi, fp32
// accumulate) instead of 10 x 3 exact-fp32 MFMAs of 32 cycles each per task: 2 x 3 x 3 MFMAs of 16 cycles.  dy2^T is split once per workgroup
// into LDS planes [48 w][64 o] (row stride 144 bytes: the 16 rows of a fragment read land 4 banks apart), its fragments do not depend on the
// task and live in registers; the weights come PRE-SPLIT from eegclip_split_rows(transpose) as planes [(c,h)][64 o]: one 16-byte load per
// k-step and plane instead of 10 strided 4-byte gathers + a split in registers.
constexpr int SCX_GY = 5;                    // workgroups per sample: 20 waves x 2 channels x 4 row blocks
template <bool APPLY, bool X3>
__global__ __launch_bounds__(256, 3) void sconv_bwd_x_kernel(const float* __restrict__ dy2, const float* __restrict__ Ws,
                                                           const unsigned short* __restrict__ wt_hi, const unsigned short* __restrict__ wt_lo,
                                                           const float* __restrict__ y1, const bn_affine bn, double* __restrict__ sums,
                                                           const double* __restrict__ sums_param, double count, float* __restrict__ dy1,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, double* __restrict__ partials, int B, int H) {
    EEG_LDS_BASE(float, lds);
    float* dl = lds;                          // !X3: [40][48] dy2[o][w] (cols >= 36 zero)   X3: two bf16 planes [48 w][144 B] of dy2^T
    float* sl = dl + (X3 ? 2 * SC_OP * SCX_RS / 4 : SC_C * SC_OP);            // [80] per-workgroup channel sums
    unsigned char* dplane = reinterpret_cast<unsigned char*>(lds);
    const int t = threadIdx.x, lane = t & 63, wv = wave_uniform(t >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int b = blockIdx.x;
    if (X3) {
        for (int i = t; i < SC_OP * 64; i += 256) {               // o >= 40 and w >= 36: zeros
            const int w = i >> 6, o = i & 63;
            if (o >= SC_C || w >= SC_W) {
                *reinterpret_cast<unsigned short*>(dplane + w * SCX_RS + 2 * o) = 0;
                *reinterpret_cast<unsigned short*>(dplane + SC_OP * SCX_RS + w * SCX_RS + 2 * o) = 0;
            }
        }
        for (int i = t; i < SC_C * SC_W; i += 256) {              // coalesced read of the sample's dy2, transposed 2-byte stores
            const int o = i / SC_W, w = i % SC_W;
            const float v = dy2[(long long)b * SC_C * SC_W + i];
            const unsigned short h = f32_to_bf16_bits(v);
            *reinterpret_cast<unsigned short*>(dplane + w * SCX_RS + 2 * o) = h;
            *reinterpret_cast<unsigned short*>(dplane + SC_OP * SCX_RS + w * SCX_RS + 2 * o) = f32_to_bf16_bits(v - bf16_bits_to_f32(h));
        }
    } else {
        for (int i = t; i < SC_C * SC_OP; i += 256) {
            const int o = i / SC_OP, w = i % SC_OP;
            dl[i] = w < SC_W ? dy2[((long long)b * SC_C + o) * SC_W + w] : 0.f;
        }
    }
    if (t < 2 * SC_C) sl[t] = 0.f;
    if (APPLY && b == 0 && blockIdx.y == 0 && t < SC_C) {        // parameter gradients from this rank's own sums (see norm.hip: bn_elu_bwd_apply)
        atomicAdd(dgamma + t, (float)sums_param[SC_C + t]);
        atomicAdd(dbeta + t, (float)sums_param[t]);
    }
    __syncthreads();
    const int MT = (H + 15) / 16;
    const f32x4 zero4v{0.f, 0.f, 0.f, 0.f};
    // X3: A fragments (dy2^T rows w = 16j + fr, k = o = 32s + 8g ..): the same for every task of the wave
    bf16x8 ah[X3 ? 3 : 1][2], al[X3 ? 3 : 1][2];
    if (X3) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                ah[j][s2] = *reinterpret_cast<const bf16x8*>(dplane + (16 * j + fr) * SCX_RS + 2 * (32 * s2 + 8 * g));
                al[j][s2] = *reinterpret_cast<const bf16x8*>(dplane + SC_OP * SCX_RS + (16 * j + fr) * SCX_RS + 2 * (32 * s2 + 8 * g));
            }
    }
    // register double-buffer: the weight and 3 x 16-byte y1 loads of the wave's next task are issued before the MFMAs / epilogue of this one
    float nav[X3 ? 1 : 10];
    bf16x8 nbh[X3 ? 2 : 1], nbl[X3 ? 2 : 1];
    f32x4 nyv[3];
    const bf16x8 zero8{0, 0, 0, 0, 0, 0, 0, 0};
    auto load_task = [&](int p) {
        const int c = p / MT, mt = p % MT;
        const int h = 16 * mt + fr;                           // this lane's row: B-operand column and accumulator column
        if (X3) {
            const long long row = ((long long)c * H + (h < H ? h : H - 1)) * 64;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                nbh[s2] = *reinterpret_cast<const bf16x8*>(wt_hi + row + 32 * s2 + 8 * g);
                nbl[s2] = *reinterpret_cast<const bf16x8*>(wt_lo + row + 32 * s2 + 8 * g);
                if (h >= H) { nbh[s2] = zero8; nbl[s2] = zero8; }
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 10; ++kk) nav[kk] = h < H ? Ws[((long long)(4 * kk + g) * SC_C + c) * H + h] : 0.f;
        }
        const float* yr = y1 + (((long long)b * SC_C + c) * H + h) * SC_W;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int w = 16 * j + 4 * g;                     // accumulator rows w .. w+3 (36 % 4 == 0: all in or all out)
            nyv[j] = (h < H && w < SC_W) ? *reinterpret_cast<const f32x4*>(yr + w) : zero4v;
        }
    };
    // gridDim.y workgroups share a sample (more waves per CU: one wave's MFMA phase overlaps another's VALU epilogue); a wave owns WHOLE channels
    // -- all row blocks mt of channel c = wid + wps * i -- so that the statistics pass reduces its per-lane sums across the wave once per
    // channel instead of once per task (12 ds_bpermute steps per task were ~1000 of its ~2500 cycles)
    const int wps = 4 * gridDim.y, wid = 4 * blockIdx.y + wv;
    const int ntask = wid < SC_C ? ((SC_C - 1 - wid) / wps + 1) * MT : 0;
    auto task_of = [&](int q) { return (wid + wps * (q / MT)) * MT + q % MT; };
    if (ntask > 0) load_task(task_of(0));
    float s1 = 0.f, s2 = 0.f;
    for (int q = 0; q < ntask; ++q) {
        const int p = task_of(q);
        const int c = p / MT, mt = p % MT;
        float av[X3 ? 1 : 10];
        bf16x8 bh[X3 ? 2 : 1], bl[X3 ? 2 : 1];
        f32x4 yv[3];
        if (X3) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) { bh[s2] = nbh[s2]; bl[s2] = nbl[s2]; }
        } else {
#pragma unroll
            for (int kk = 0; kk < 10; ++kk) av[kk] = nav[kk];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) yv[j] = nyv[j];
        if (q + 1 < ntask) load_task(task_of(q + 1));
        f32x4 acc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = zero4v;
        if (X3) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 3; ++j) {                 // D[w = 16j + 4g + r][h = 16mt + fr]
                    acc[j] = mfma_bf16_16x16x32(ah[j][s2], bl[s2], acc[j]);
                    acc[j] = mfma_bf16_16x16x32(al[j][s2], bh[s2], acc[j]);
                    acc[j] = mfma_bf16_16x16x32(ah[j][s2], bh[s2], acc[j]);
                }
        } else {
#pragma unroll
            for (int kk = 0; kk < 10; ++kk) {
                const float* dp = dl + (4 * kk + g) * SC_OP + fr;
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[j] = mfma_f32_16x16x4(dp[16 * j], av[kk], acc[j]);      // D[w = 16j + 4g + r][h = 16mt + fr]
            }
        }
        const float mean = bn.mean[c], rstd = bn.rstd[c], gam = bn.gamma[c], bet = bn.beta[c];
        float m1 = 0.f, m2 = 0.f;
        if (APPLY) { m1 = (float)(sums[c] / count); m2 = (float)(sums[SC_C + c] / count); }
        const int h = 16 * mt + fr;
        float* dr = APPLY ? dy1 + (((long long)b * SC_C + c) * H + h) * SC_W : nullptr;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int w = 16 * j + 4 * g;
            if (h < H && w < SC_W) {
                f32x4 o4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xh = (yv[j][r] - mean) * rstd;
                    const float u = gam * xh + bet;
                    const float da = u > 0.f ? acc[j][r] : acc[j][r] * fast_exp(u);
                    o4[r] = gam * rstd * (da - m1 - xh * m2);
                    s1 += da;
                    s2 += da * xh;
                }
                if (APPLY) *reinterpret_cast<f32x4*>(dr + w) = o4;
            }
        }
        if (!APPLY && mt == MT - 1) {                          // channel complete
            s1 = wave_sum(s1);
            s2 = wave_sum(s2);
            if (lane == 0) { atomicAdd(sl + c, s1); atomicAdd(sl + SC_C + c, s2); }
            s1 = 0.f;
            s2 = 0.f;
        }
    }
    if (!APPLY) {
        __syncthreads();
        if (t < 2 * SC_C) {
            // (partials: one row of 80 sums per workgroup, column-summed by colsum_f64_kernel -- 1280 workgroups adding into 80 addresses were
            // ~20 us of atomics in this 65-us kernel)
            if (partials) partials[((long long)blockIdx.y * gridDim.x + blockIdx.x) * (2 * SC_C) + t] = (double)sl[t];
            else          atomicAdd(sums + t, (double)sl[t]);
        }
    }
}

}  // namespace eeg

using namespace eeg;

static bool sc_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static int sc_check(int B, int H) { return (B < 1 || H < 1 || H > 64) ? EEGCLIP_EINVAL : 0; }

extern "C" long long eegclip_sconv_fwd_workspace_floats(int B) { return B < 1 ? 0 : (long long)SCF_KS * B * SC_C * SC_W; }

extern "C" int eegclip_sconv_fwd(const float* y1, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* Ws,
                                 const float* bs, float* y2, double* sums2, int B, int H, int y2_is_zero, float* workspace, void* stream) {
    if (int rc = sc_check(B, H)) return rc;
    if (!y1 || !mean || !rstd || !gamma || !beta || !Ws || !bs || !y2) return EEGCLIP_EINVAL;
    if (!sc_aligned16(y1) || !sc_aligned16(Ws)) return EEGCLIP_EALIGN;
    const bn_affine bn{mean, rstd, gamma, beta};
    const int K = SC_C * H;
    // the K slices add their partial tiles into y2 with atomics: it must start at zero (callers that clear it together with their other
    // accumulators pass y2_is_zero != 0 and save the extra memset launch)
    if (!y2_is_zero && !workspace) (void)hipMemsetAsync(y2, 0, (size_t)B * SC_C * SC_W * sizeof(float), (hipStream_t)stream);
    {
        const int kper = ((K + SCF_KS - 1) / SCF_KS + 3) & ~3;                 // slices start on 16-byte boundaries of both operands
        EEG_LAUNCH((sconv_fwd_kernel<1>), dim3(B, SCF_KS), dim3(320), SCD_LDS, stream, y1, bn, Ws, bs, y2, workspace, B, H, kper);
    }
    if (sums2 || workspace)
        EEG_LAUNCH(sconv_merge_stats2_kernel, dim3(SC_C, 8), dim3(256), 8 * sizeof(double), stream, (const float*)workspace, SCF_KS, y2, sums2, B);
    return (int)hipGetLastError();
}

// slab width 128 k (a 256-wide slab measured no faster and its split-bf16 instantiation spilled 225 registers); ~4 workgroups per CU
constexpr int SCW_NS = 128;
static int scw_groups(int B, int H) {
    const int slabs = (SC_C * H + SCW_NS - 1) / SCW_NS;
    int gcap = 1024 / slabs;
    if (gcap < 1) gcap = 1;
    return B < gcap ? B : gcap;
}
extern "C" long long eegclip_sconv_bwd_w_workspace_floats(int B, int H) { return (long long)scw_groups(B, H) * SC_C * SC_C * H; }

extern "C" int eegclip_sconv_bwd_w(const float* y1, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* dy2,
                                   float* dWs, float* workspace, int B, int H, int precision, void* stream) {
    if (int rc = sc_check(B, H)) return rc;
    if (!y1 || !mean || !rstd || !gamma || !beta || !dy2 || !dWs || !workspace) return EEGCLIP_EINVAL;
    if (precision != EEGCLIP_PREC_F32 && precision != EEGCLIP_PREC_BF16X3) return EEGCLIP_EINVAL;
    if (!sc_aligned16(y1) || (precision == EEGCLIP_PREC_BF16X3 && !sc_aligned16(dy2))) return EEGCLIP_EALIGN;
    const bn_affine bn{mean, rstd, gamma, beta};
    const int K = SC_C * H, groups = scw_groups(B, H), ns = SCW_NS;
    const dim3 grid((K + ns - 1) / ns, groups);
    if (precision == EEGCLIP_PREC_BF16X3) {
        const size_t lds = (size_t)2 * (ns + SC_OP) * SWX_RS + 2 * SC_C * sizeof(float);
        EEG_LAUNCH(sconv_bwd_w_x3_kernel<SCW_NS>, grid, dim3(256), lds, stream, y1, bn, dy2, workspace, B, H, groups);
    } else {
        const size_t lds = (ns * SCW_L + SC_OP * SCW_L + 2 * SC_C) * sizeof(float);
        EEG_LAUNCH(sconv_bwd_w_kernel<SCW_NS>, grid, dim3(256), lds, stream, y1, bn, dy2, workspace, B, H, groups);
    }
    const long long n = (long long)SC_C * K;
    EEG_LAUNCH(sconv_bwd_w_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 256 * sizeof(float), stream, workspace, groups, n, dWs);
    return (int)hipGetLastError();
}

static bool scx_planes_ok(const void* hi, const void* lo) { return hi && lo && sc_aligned16(hi) && sc_aligned16(lo); }

extern "C" long long eegclip_sconv_bwd_x_stats_workspace_floats(int B) { return B < 1 ? 0 : 2LL * B * SCX_GY * 2 * SC_C; }     // (doubles, in floats)

extern "C" int eegclip_sconv_bwd_x_stats(const float* dy2, const float* Ws, const void* WsT_hi, const void* WsT_lo, const float* y1, const float* mean,
                                         const float* rstd, const float* gamma, const float* beta, double* sums, float* workspace, int B, int H,
                                         void* stream) {
    if (int rc = sc_check(B, H)) return rc;
    if (!dy2 || !Ws || !y1 || !mean || !rstd || !gamma || !beta || !sums) return EEGCLIP_EINVAL;
    if ((WsT_hi == nullptr) != (WsT_lo == nullptr)) return EEGCLIP_EINVAL;
    if (!sc_aligned16(y1) || (WsT_hi && !scx_planes_ok(WsT_hi, WsT_lo)) || (reinterpret_cast<uintptr_t>(workspace) & 7u)) return EEGCLIP_EALIGN;
    const bn_affine bn{mean, rstd, gamma, beta};
    const unsigned short *wh = (const unsigned short*)WsT_hi, *wl = (const unsigned short*)WsT_lo;
    double* parts = reinterpret_cast<double*>(workspace);
    if (wh) {
        const size_t lds = 2 * SC_OP * SCX_RS + 2 * SC_C * sizeof(float);
        EEG_LAUNCH((sconv_bwd_x_kernel<false, true>), dim3(B, SCX_GY), dim3(256), lds, stream, dy2, Ws, wh, wl, y1, bn, sums, (const double*)nullptr, 1.0,
                   (float*)nullptr, (float*)nullptr, (float*)nullptr, parts, B, H);
    } else {
        const size_t lds = (SC_C * SC_OP + 2 * SC_C) * sizeof(float);
        EEG_LAUNCH((sconv_bwd_x_kernel<false, false>), dim3(B, SCX_GY), dim3(256), lds, stream, dy2, Ws, wh, wl, y1, bn, sums, (const double*)nullptr, 1.0,
                   (float*)nullptr, (float*)nullptr, (float*)nullptr, parts, B, H);
    }
    if (parts) EEG_COLSUM_F64((const double*)parts, B * SCX_GY, 2 * SC_C, sums, stream);
    return (int)hipGetLastError();
}

extern "C" int eegclip_sconv_bwd_x_apply(const float* dy2, const float* Ws, const void* WsT_hi, const void* WsT_lo, const float* y1, const float* mean,
                                         const float* rstd, const float* gamma, const float* beta, const double* sums, const double* sums_local,
                                         double count, float* dy1, float* dgamma, float* dbeta, int B, int H, void* stream) {
    if (int rc = sc_check(B, H)) return rc;
    if (!dy2 || !Ws || !y1 || !mean || !rstd || !gamma || !beta || !sums || !dy1 || !dgamma || !dbeta || count < 1.0) return EEGCLIP_EINVAL;
    if ((WsT_hi == nullptr) != (WsT_lo == nullptr)) return EEGCLIP_EINVAL;
    if (!sc_aligned16(y1) || !sc_aligned16(dy1) || (WsT_hi && !scx_planes_ok(WsT_hi, WsT_lo))) return EEGCLIP_EALIGN;
    const bn_affine bn{mean, rstd, gamma, beta};
    const unsigned short *wh = (const unsigned short*)WsT_hi, *wl = (const unsigned short*)WsT_lo;
    if (wh) {
        const size_t lds = 2 * SC_OP * SCX_RS + 2 * SC_C * sizeof(float);
        EEG_LAUNCH((sconv_bwd_x_kernel<true, true>), dim3(B, SCX_GY), dim3(256), lds, stream, dy2, Ws, wh, wl, y1, bn, const_cast<double*>(sums),
                   sums_local ? sums_local : sums, count, dy1, dgamma, dbeta, (double*)nullptr, B, H);
    } else {
        const size_t lds = (SC_C * SC_OP + 2 * SC_C) * sizeof(float);
        EEG_LAUNCH((sconv_bwd_x_kernel<true, false>), dim3(B, SCX_GY), dim3(256), lds, stream, dy2, Ws, wh, wl, y1, bn, const_cast<double*>(sums),
                   sums_local ? sums_local : sums, count, dy1, dgamma, dbeta, (double*)nullptr, B, H);
    }
    return (int)hipGetLastError();
}
