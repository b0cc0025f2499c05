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
// All contractions run on v_mfma_f32_16x16x4_f32 (exact fp32).  HBM traffic: 5 passes over the 93 MB tensor per step instead of 14.
#include "eeg_common.h"

#include <stdlib.h>

namespace eeg {

constexpr int SC_C = 40;      // channels in and out
constexpr int SC_W = 36;      // positions per row
constexpr int SC_OP = 48;     // out channels / positions padded to 3 MFMA tiles

struct bn_affine {            // per-channel BatchNorm as y -> xhat -> u: xhat = (y - mean) * rstd ; u = gamma * xhat + beta
    const float *mean, *rstd, *gamma, *beta;
};

// ---------------------------------------------------------------------------------------------------------------------------
// forward: workgroup = (sample, K slice); K = (c,h) = 40*H is cut into SCF_KS slices (a whole sample per workgroup left ONE workgroup
// of 4 waves per CU: 126 us for a 93 MB read); each slice is streamed in chunks of 128 k, wave v owning k = 32v .. 32v+31 of a chunk.
//   B operand (activations): y1 chunk -> registers -> ELU(BN(.)) -> LDS [128][52] (16-byte stores; rows 4 apart land 16 banks apart)
//   A operand (weights):     straight from global/L2 into registers in MFMA order -- lane (fr, g) fetches Ws[o = 16i + fr][k .. k+3] as
//       one 16-byte load and feeds four k-steps (the k-slot <-> lane-group assignment is free as long as both operands agree).  A staged
//       [48][129] weight tile cost 3.2 LDS conflict cycles per LDS instruction (SQ_LDS_BANK_CONFLICT) and half the LDS traffic.
// The 4 waves keep private 3x3 accumulator tiles (o x w), combined through LDS at the end and added atomically into y2 (zeroed by
// the launcher).
constexpr int SCF_KC = 128;
constexpr int SCF_KS = 4;
constexpr int SCF_LZ = 52;             // activation tile row stride
__global__ __launch_bounds__(256) void sconv_fwd_kernel(const float* __restrict__ y1, const bn_affine bn, const float* __restrict__ Ws,
                                                         const float* __restrict__ bs, float* __restrict__ y2, int B, int H, int kper) {
    EEG_LDS_BASE(float, lds);
    float* zl = lds;                          // [128][52]      z1[k0 + kk][w]   (cols >= 36 zero)
    float* aff = zl + SCF_KC * SCF_LZ;        // [2][40]  BatchNorm folded to u = y * aff[c] + aff[40 + c] (LDS: no dependent global loads in staging)
    float* red = lds;                         // 2 x [48][52] cross-wave reduction scratch, aliases the activation tile after the last chunk
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int b = blockIdx.x;
    const int K = SC_C * H;
    const int kbeg = blockIdx.y * kper, kend = kbeg + kper < K ? kbeg + kper : K;
    if (t < SC_C) {
        const float sc = bn.gamma[t] * bn.rstd[t];
        aff[t] = sc;
        aff[SC_C + t] = bn.beta[t] - bn.mean[t] * sc;
    }
    for (int i = t; i < SCF_KC * SCF_LZ; i += 256) zl[i] = 0.f;
    f32x4 acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* yb = y1 + (long long)b * K * SC_W;
    // software pipeline: the global loads of chunk k0+128 (6 weight + 5 activation float4 per thread; kbeg, K and 36 are multiples of 4,
    // so a float4 never straddles a row of either operand) are issued before the MFMAs of chunk k0 and land under them
    f32x4 va[3][2], vy[5];
    const f32x4 zero4v{0.f, 0.f, 0.f, 0.f};
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int o = 16 * i + fr, k = k0 + 32 * wv + 16 * h + 4 * g;
                va[i][h] = (o < SC_C && k < kend) ? *reinterpret_cast<const f32x4*>(Ws + (long long)o * K + k) : zero4v;
            }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int e = 4 * (t + 256 * j), kk = e / SC_W;                          // 128 rows x 9 float4
            vy[j] = (e < SCF_KC * SC_W && k0 + kk < kend) ? *reinterpret_cast<const f32x4*>(yb + (long long)k0 * SC_W + e) : zero4v;
        }
    };
    auto store_chunk = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int e = 4 * (t + 256 * j), kk = e / SC_W, w = e % SC_W;
            if (e < SCF_KC * SC_W) {
                f32x4 z = zero4v;
                if (k0 + kk < kend) {
                    const int c = (k0 + kk) / H;
                    const float sc = aff[c], sh = aff[SC_C + c];
#pragma unroll
                    for (int q = 0; q < 4; ++q) z[q] = elu1_fast(vy[j][q] * sc + sh);    // z1 = ELU(BN(y1)) evaluated on the way into LDS
                }
                *reinterpret_cast<f32x4*>(zl + kk * SCF_LZ + w) = z;
            }
        }
    };
    if (kbeg < kend) load_chunk(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += SCF_KC) {
        __syncthreads();
        store_chunk(k0);
        f32x4 a[3][2];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) a[i][h] = va[i][h];
        __syncthreads();
        if (k0 + SCF_KC < kend) load_chunk(k0 + SCF_KC);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float* zp = zl + (32 * wv + 16 * h + 4 * g + s4) * SCF_LZ + fr;
                float bv[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) bv[j] = zp[16 * j];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[i][j] = mfma_f32_16x16x4(a[i][h][s4], bv[j], acc[i][j]);      // D[o = 16i+4g+r][w = 16j+fr]
            }
    }
    // cross-wave sum of the four k-partial accumulator sets: a two-level tree through LDS with plain stores (waves 2,3 -> 0,1, then
    // 1 -> 0).  The first version used 36 ds_add_f32 per lane into one tile: LDS float atomics retire at ~170 cycles per wave
    // instruction, and with 4 K-slice workgroups per sample that epilogue alone was ~40 % of the kernel.
    constexpr int RLD = 52;                   // 4 accumulator row groups land 16 banks apart: 2-way at most, free for stores
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
    __syncthreads();                          // every wave is done with the operand tiles: LDS becomes reduction scratch
    if (wv >= 2) put(red + (wv - 2) * SC_OP * RLD);
    __syncthreads();
    if (wv < 2) add(red + wv * SC_OP * RLD);
    __syncthreads();
    if (wv == 1) put(red);
    __syncthreads();
    if (wv == 0) {
        add(red);
        float* yo = y2 + (long long)b * SC_C * SC_W;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = 16 * i + 4 * g + r, w = 16 * j + fr;
                    if (o < SC_C && w < SC_W) atomicAdd(yo + o * SC_W + w, acc[i][j][r] + (blockIdx.y == 0 ? bs[o] : 0.f));
                }
    }
}

// BatchNorm2d #2 batch statistics of y2 (B,40,36): workgroup = (channel, slice of samples); fp64 atomics, 2 per workgroup
__global__ __launch_bounds__(256) void sconv_stats2_kernel(const float* __restrict__ y2, double* __restrict__ sums2, int B) {
    EEG_LDS_BASE(double, sh);                 // [2][4] per-wave partial sums
    const int o = blockIdx.x, t = threadIdx.x;
    double s = 0.0, q = 0.0;
    for (int i = blockIdx.y * 256 + t; i < B * SC_W; i += gridDim.y * 256) {
        const float v = y2[((long long)(i / SC_W) * SC_C + o) * SC_W + i % SC_W];
        s += v;
        q += (double)v * v;
    }
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
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
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

__global__ void sconv_bwd_w_reduce_kernel(const float* __restrict__ partials, int groups, long long n, float* __restrict__ dW) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < groups; ++k) s += partials[(long long)k * n + i];      // independent loads: keep 8 in flight per thread
    dW[i] += s;
}

// ---------------------------------------------------------------------------------------------------------------------------
// input gradient + BatchNorm1 backward.  gridDim.y workgroups per sample; wave tasks = (channel c, 16-row block of h):
// dz^T[w][h] = sum_o dy2[o][w] * Ws[o][c][h] on the matrix cores (K = 40), then da = dz * ELU'(u) with u = BN(y1).  The product is
// formed TRANSPOSED (MFMA rows = positions w, columns = rows h) so that the 4 accumulator registers of a lane are 4 consecutive w of
// one row: y1 is read and dy1 written as one 16-byte access per lane and tile instead of four 4-byte ones.
//   APPLY = false: accumulate sum(da), sum(da * xhat) per channel (LDS, then 80 fp64 atomics per workgroup)
//   APPLY = true : dy1 = gamma * rstd * (da - S1/n - xhat * S2/n)
template <bool APPLY>
__global__ __launch_bounds__(256) void sconv_bwd_x_kernel(const float* __restrict__ dy2, const float* __restrict__ Ws,
                                                           const float* __restrict__ y1, const bn_affine bn, double* __restrict__ sums,
                                                           const double* __restrict__ sums_param, double count, float* __restrict__ dy1,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int H) {
    EEG_LDS_BASE(float, lds);
    float* dl = lds;                          // [40][48]  dy2[o][w]  (cols >= 36 zero)
    float* sl = dl + SC_C * SC_OP;            // [80] per-workgroup channel sums
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int b = blockIdx.x;
    for (int i = t; i < SC_C * SC_OP; i += 256) {
        const int o = i / SC_OP, w = i % SC_OP;
        dl[i] = w < SC_W ? dy2[((long long)b * SC_C + o) * SC_W + w] : 0.f;
    }
    if (t < 2 * SC_C) sl[t] = 0.f;
    if (APPLY && b == 0 && blockIdx.y == 0 && t < SC_C) {        // parameter gradients from this rank's own sums (see norm.hip: bn_elu_bwd_apply)
        atomicAdd(dgamma + t, (float)sums_param[SC_C + t]);
        atomicAdd(dbeta + t, (float)sums_param[t]);
    }
    __syncthreads();
    const int MT = (H + 15) / 16;
    const f32x4 zero4v{0.f, 0.f, 0.f, 0.f};
    // register double-buffer: the 10 weight and 3 x 16-byte y1 loads of the wave's next task are issued before the MFMAs / epilogue of this one
    float nav[10];
    f32x4 nyv[3];
    auto load_task = [&](int p) {
        const int c = p / MT, mt = p % MT;
        const int h = 16 * mt + fr;                           // this lane's row: B-operand column and accumulator column
#pragma unroll
        for (int kk = 0; kk < 10; ++kk) nav[kk] = h < H ? Ws[((long long)(4 * kk + g) * SC_C + c) * H + h] : 0.f;
        const float* yr = y1 + (((long long)b * SC_C + c) * H + h) * SC_W;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int w = 16 * j + 4 * g;                     // accumulator rows w .. w+3 (36 % 4 == 0: all in or all out)
            nyv[j] = (h < H && w < SC_W) ? *reinterpret_cast<const f32x4*>(yr + w) : zero4v;
        }
    };
    const int pstep = 4 * gridDim.y, p0 = 4 * blockIdx.y + wv;      // gridDim.y workgroups share a sample: more waves per CU to overlap
    if (p0 < SC_C * MT) load_task(p0);                                 // one wave's MFMA phase with another's VALU epilogue
    for (int p = p0; p < SC_C * MT; p += pstep) {
        const int c = p / MT, mt = p % MT;
        float av[10];
        f32x4 yv[3];
#pragma unroll
        for (int kk = 0; kk < 10; ++kk) av[kk] = nav[kk];
#pragma unroll
        for (int j = 0; j < 3; ++j) yv[j] = nyv[j];
        if (p + pstep < SC_C * MT) load_task(p + pstep);
        f32x4 acc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = zero4v;
#pragma unroll
        for (int kk = 0; kk < 10; ++kk) {
            const float* dp = dl + (4 * kk + g) * SC_OP + fr;
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[j] = mfma_f32_16x16x4(dp[16 * j], av[kk], acc[j]);      // D[w = 16j + 4g + r][h = 16mt + fr]
        }
        const float mean = bn.mean[c], rstd = bn.rstd[c], gam = bn.gamma[c], bet = bn.beta[c];
        float m1 = 0.f, m2 = 0.f;
        if (APPLY) { m1 = (float)(sums[c] / count); m2 = (float)(sums[SC_C + c] / count); }
        float s1 = 0.f, s2 = 0.f;
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
        if (!APPLY) {
            s1 = wave_sum(s1);
            s2 = wave_sum(s2);
            if (lane == 0) { atomicAdd(sl + c, s1); atomicAdd(sl + SC_C + c, s2); }
        }
    }
    if (!APPLY) {
        __syncthreads();
        if (t < 2 * SC_C) atomicAdd(sums + t, (double)sl[t]);
    }
}

}  // namespace eeg

using namespace eeg;

static bool sc_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static int sc_check(int B, int H) { return (B < 1 || H < 1 || H > 64) ? EEGCLIP_EINVAL : 0; }

extern "C" int eegclip_sconv_fwd(const float* y1, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* Ws,
                                 const float* bs, float* y2, double* sums2, int B, int H, int y2_is_zero, void* stream) {
    if (int rc = sc_check(B, H)) return rc;
    if (!y1 || !mean || !rstd || !gamma || !beta || !Ws || !bs || !y2) return EEGCLIP_EINVAL;
    if (!sc_aligned16(y1) || !sc_aligned16(Ws)) return EEGCLIP_EALIGN;
    const bn_affine bn{mean, rstd, gamma, beta};
    const int K = SC_C * H, kper = ((K + SCF_KS - 1) / SCF_KS + 3) & ~3;       // slices start on 16-byte boundaries of both operands
    const size_t lds = (SCF_KC * SCF_LZ + 2 * SC_C) * sizeof(float);
    // the K slices add their partial tiles into y2 with atomics: it must start at zero (callers that clear it together with their other
    // accumulators pass y2_is_zero != 0 and save the extra memset launch)
    if (!y2_is_zero) (void)hipMemsetAsync(y2, 0, (size_t)B * SC_C * SC_W * sizeof(float), (hipStream_t)stream);
    EEG_LAUNCH(sconv_fwd_kernel, dim3(B, SCF_KS), dim3(256), lds, stream, y1, bn, Ws, bs, y2, B, H, kper);
    if (sums2) EEG_LAUNCH(sconv_stats2_kernel, dim3(SC_C, 8), dim3(256), 8 * sizeof(double), stream, y2, sums2, B);
    return (int)hipGetLastError();
}

// slab width / sample groups (tuning aids: EEGCLIP_SCW_NS = 128 | 256, EEGCLIP_SCW_G)
static int scw_ns() {
    static const int v = getenv("EEGCLIP_SCW_NS") ? atoi(getenv("EEGCLIP_SCW_NS")) : 128;
    return v == 256 ? 256 : 128;
}
static int scw_groups(int B, int H) {
    static const int forced = getenv("EEGCLIP_SCW_G") ? atoi(getenv("EEGCLIP_SCW_G")) : 0;
    const int slabs = (SC_C * H + scw_ns() - 1) / scw_ns();
    int gcap = forced > 0 ? forced : 1024 / slabs;             // ~4 workgroups per CU
    if (gcap < 1) gcap = 1;
    return B < gcap ? B : gcap;
}
extern "C" long long eegclip_sconv_bwd_w_workspace_floats(int B, int H) { return (long long)scw_groups(B, H) * SC_C * SC_C * H; }

extern "C" int eegclip_sconv_bwd_w(const float* y1, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* dy2,
                                   float* dWs, float* workspace, int B, int H, void* stream) {
    if (int rc = sc_check(B, H)) return rc;
    if (!y1 || !mean || !rstd || !gamma || !beta || !dy2 || !dWs || !workspace) return EEGCLIP_EINVAL;
    if (!sc_aligned16(y1)) return EEGCLIP_EALIGN;
    const bn_affine bn{mean, rstd, gamma, beta};
    const int K = SC_C * H, groups = scw_groups(B, H), ns = scw_ns();
    const size_t lds = (ns * SCW_L + SC_OP * SCW_L + 2 * SC_C) * sizeof(float);
    const dim3 grid((K + ns - 1) / ns, groups);
    if (ns == 256) EEG_LAUNCH(sconv_bwd_w_kernel<256>, grid, dim3(256), lds, stream, y1, bn, dy2, workspace, B, H, groups);
    else           EEG_LAUNCH(sconv_bwd_w_kernel<128>, grid, dim3(256), lds, stream, y1, bn, dy2, workspace, B, H, groups);
    const long long n = (long long)SC_C * K;
    EEG_LAUNCH(sconv_bwd_w_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, workspace, groups, n, dWs);
    return (int)hipGetLastError();
}

extern "C" int eegclip_sconv_bwd_x_stats(const float* dy2, const float* Ws, const float* y1, const float* mean, const float* rstd,
                                         const float* gamma, const float* beta, double* sums, int B, int H, void* stream) {
    if (int rc = sc_check(B, H)) return rc;
    if (!dy2 || !Ws || !y1 || !mean || !rstd || !gamma || !beta || !sums) return EEGCLIP_EINVAL;
    if (!sc_aligned16(y1)) return EEGCLIP_EALIGN;
    const bn_affine bn{mean, rstd, gamma, beta};
    const size_t lds = (SC_C * SC_OP + 2 * SC_C) * sizeof(float);
    EEG_LAUNCH((sconv_bwd_x_kernel<false>), dim3(B, 4), dim3(256), lds, stream, dy2, Ws, y1, bn, sums, (const double*)nullptr, 1.0, (float*)nullptr,
               (float*)nullptr, (float*)nullptr, B, H);
    return (int)hipGetLastError();
}

extern "C" int eegclip_sconv_bwd_x_apply(const float* dy2, const float* Ws, const float* y1, const float* mean, const float* rstd,
                                         const float* gamma, const float* beta, const double* sums, const double* sums_local, double count,
                                         float* dy1, float* dgamma, float* dbeta, int B, int H, void* stream) {
    if (int rc = sc_check(B, H)) return rc;
    if (!dy2 || !Ws || !y1 || !mean || !rstd || !gamma || !beta || !sums || !dy1 || !dgamma || !dbeta || count < 1.0) return EEGCLIP_EINVAL;
    if (!sc_aligned16(y1) || !sc_aligned16(dy1)) return EEGCLIP_EALIGN;
    const bn_affine bn{mean, rstd, gamma, beta};
    const size_t lds = (SC_C * SC_OP + 2 * SC_C) * sizeof(float);
    EEG_LAUNCH((sconv_bwd_x_kernel<true>), dim3(B, 4), dim3(256), lds, stream, dy2, Ws, y1, bn, const_cast<double*>(sums),
               sums_local ? sums_local : sums, count, dy1, dgamma, dbeta, B, H);
    return (int)hipGetLastError();
}
