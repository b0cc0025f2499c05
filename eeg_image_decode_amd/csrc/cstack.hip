// Conv stack of Enc_eeg, forward, recomputed from the token rows (see cstack_common.h; Retrieval/ATMS_retrieval.py:102-106):
//   cstack_stats1_kernel   BatchNorm1 batch sums of y1 = pool(conv(x)) WITHOUT writing y1: one partial row [sum | sumsq] per sample
//   cstack_fwd_kernel      per sample: y1 tile (tap contraction) -> BatchNorm1 -> ELU -> spatial (H x 1) conv, chained on the matrix cores:
//                          the accumulator tiles of the tap contraction, D[c][w] of one token row h, ARE the k = (c, h) operand fragments of the
//                          spatial contraction y2[o][w] += sum_c Ws[o][c][h] z1[c][h][w] once passed through BN1 / ELU and split in registers
//   cstack_pack_kernel     Ws (40,40,H) fp32 -> bf16 hi | lo planes in MFMA-fragment order for that chain, once per optimizer step
// Replaces tsconv_fwd (y1 write, 93 MB) + colsum + bn_finalize + sconv_fwd (y1 read) + sconv_merge_stats2 of rounds 1-4.
#include "cstack_common.h"

#include <stdlib.h>

namespace eeg {

// ---- packed spatial weights: see cstack_common.h (cs_pack_item) ----------------------------------------------------------------------------------------
inline int cs_pairs(int H) { return (H + 1) / 2; }

__global__ __launch_bounds__(256) void cstack_pack_kernel(const float* __restrict__ Ws, unsigned char* __restrict__ packed, int H) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 3 * ((H + 1) / 2) * 3 * 64) return;
    cs_pack_item(Ws, packed, H, id);
}

// ---- BatchNorm1 batch sums ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CS_NT) void cstack_stats1_kernel(const float* __restrict__ x, long long xs_b, long long xs_h, const float* __restrict__ w25,
                                                              const float* __restrict__ bias, double* __restrict__ rows, int B, int H, int vec2) {
    EEG_LDS_BASE(unsigned char, ldsb);
    cs_stats1_sample(ldsb, x, xs_b, xs_h, w25, bias, rows, blockIdx.x, H, vec2);
}

// ---- the fused forward ------------------------------------------------------------------------------------------------------------------------------
struct cs_fwd_args {
    const float* x;
    long long xs_b, xs_h;
    const float *w25, *bias1;
    const double* stat1;          // train: partial rows of BatchNorm1 sums (NULL: eval -- mean1 / rstd1 are inputs)
    int nstat1;
    double count1;
    float eps, momentum;
    const float *gamma1, *beta1;
    float *mean1, *rstd1;
    float *run_mean1, *run_var1;
    long long* nbt1;
    const unsigned char* packed;
    const float* bias2;
    float* y2;                    // [B][40][36]
    double* stat2;                // [B][80] BatchNorm2 partial rows of y2 (NULL: none)
    int B, H, vec2;
};

constexpr int CSF_RLD = 52;                                        // row stride of the cross-wave reduction tiles [48][52]
constexpr int CSF_LDS_RED = CS_NW * 48 * CSF_RLD * 4;              // 79,872 B: aliases the packed rows + scratch once the contractions are done
constexpr int CSF_LDS_MAIN = (CS_LDS_S + CS_LDS_PS) > CSF_LDS_RED ? (CS_LDS_S + CS_LDS_PS) : CSF_LDS_RED;
constexpr int CSF_LDS = CSF_LDS_MAIN + 2 * 48 * 4 + CS_C * CS_W * 4;        // + BatchNorm1 scale | shift + the y2 tile

__global__ __launch_bounds__(CS_NT) void cstack_fwd_kernel(const cs_fwd_args a) {
    EEG_LDS_BASE(unsigned char, ldsb);
    unsigned* S32 = reinterpret_cast<unsigned*>(ldsb);
    float* ps = reinterpret_cast<float*>(ldsb + CS_LDS_S);
    float* red = reinterpret_cast<float*>(ldsb);
    float* aff = reinterpret_cast<float*>(ldsb + CSF_LDS_MAIN);    // [48] scale | [48] shift (filters >= 40: 0 -> z1 = ELU(0) = 0)
    float* yt = aff + 96;                                          // [40][36] y2 of this sample
    const int t = threadIdx.x, lane = t & 63, wv = wave_uniform(t >> 6);
    const int n = lane & 15, kg = lane >> 4;
    const int b = blockIdx.x, H = a.H;
    double* bnscr = reinterpret_cast<double*>(yt);                 // [6][80] partial sums of the BatchNorm1 rows (the y2 tile is written much later)
    float vx[CS_RPW][4];
    cs_stage_load<true>(vx, a.x, a.xs_b, a.xs_h, b, H, a.vec2 != 0);
    if (a.stat1) cs_bn_rows_partial(a.stat1, a.nstat1, bnscr);      // (under the row loads)
    __syncthreads();
    if (t < 48) {
        float sc = 0.f, sh = 0.f;
        if (t < CS_C) {
            float mean, rstd;
            if (a.stat1) {
                double var;
                cs_bn_rows_finish(bnscr, a.count1, a.eps, t, mean, rstd, var);
                if (b == 0) {                                      // what bn_finalize did: statistics for the backward, running statistics, step counter
                    a.mean1[t] = mean;
                    a.rstd1[t] = rstd;
                    if (a.run_mean1) {
                        const double unb = a.count1 > 1.0 ? var * (a.count1 / (a.count1 - 1.0)) : var;
                        a.run_mean1[t] = (1.f - a.momentum) * a.run_mean1[t] + a.momentum * mean;
                        a.run_var1[t] = (1.f - a.momentum) * a.run_var1[t] + a.momentum * (float)unb;
                    }
                    if (t == 0 && a.nbt1) *a.nbt1 += 1;
                }
            } else {
                mean = a.mean1[t];
                rstd = a.rstd1[t];
            }
            sc = a.gamma1[t] * rstd;
            sh = a.beta1[t] + (a.bias1[t] - mean) * sc;           // u = gamma * (acc + bias - mean) * rstd + beta
        }
        aff[t] = sc;
        aff[48 + t] = sh;
    }
    __syncthreads();
    // taps scaled by gamma * rstd with the constant beta + (bias - mean) * gamma * rstd in the ones slot: the tap contraction yields u = BN1(y1) itself
    bf16x8 wh[3], wl[3];
    cs_tap_frags_affine(a.w25, [&](int c, float& sc, float& sh) { sc = aff[c]; sh = aff[48 + c]; }, wh, wl);
    cs_stage_finish<true>(S32, ps + wv * 256, vx, H);               // (a wave works on the row pairs it staged: no workgroup barrier)
    f32x4 acc2[3][3];                                              // y2 partial D[o = 16 ot + 4 kg + r][w = 16 wt + n] over this wave's rows
#pragma unroll
    for (int ot = 0; ot < 3; ++ot)
#pragma unroll
        for (int wt = 0; wt < 3; ++wt) acc2[ot][wt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 zero4{0.f, 0.f, 0.f, 0.f};
    const int npairs = (H + 1) / 2;
    for (int q = wv; q < npairs; q += CS_NW) {
        u32x2_t th[2][3], tl[2][3];                                // tile ct = 2 of both rows: the halves of the pair's third k-step
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int h = 2 * q + e;
            if (h >= H) {                                          // (odd H: the last pair has one row; wave-uniform)
#pragma unroll
                for (int wt = 0; wt < 3; ++wt) { th[e][wt] = u32x2_t{0u, 0u}; tl[e][wt] = u32x2_t{0u, 0u}; }
                continue;
            }
            bf16x8 a2h[3], a2l[3];                                 // Ws fragments of step 3 q + e: straight from L2 (every workgroup reads the same 576 KB)
#pragma unroll
            for (int ot = 0; ot < 3; ++ot) {
                a2h[ot] = *reinterpret_cast<const bf16x8*>(a.packed + csp_offset(3 * q + e, ot, 0) + 16 * lane);
                a2l[ot] = *reinterpret_cast<const bf16x8*>(a.packed + csp_offset(3 * q + e, ot, 1) + 16 * lane);
            }
#pragma unroll
            for (int wt = 0; wt < 3; ++wt) {
                bf16x8 xh, xl;
                cs_sfrag_ones(S32, h, wt, xh, xl);
                float z[3][4];
                f32x4 u[3] = {zero4, zero4, zero4};
                cs_mma3_a3(wh, wl, xh, xl, u);                                        // u = BN1(y1):  D[c = 16 ct + 4 kg + r][w = 16 wt + n]
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) {
                    const f32x2_t z0 = cs_elu2(cs_lo2(u[ct])), z1 = cs_elu2(cs_hi2(u[ct]));
                    z[ct][0] = z0[0]; z[ct][1] = z0[1]; z[ct][2] = z1[0]; z[ct][3] = z1[1];
                }
                bf16x8 zh, zl;
                {
                    const float v8[8] = {z[0][0], z[0][1], z[0][2], z[0][3], z[1][0], z[1][1], z[1][2], z[1][3]};
                    cs_split8(v8, zh, zl);
                }
                x3_split4(z[2][0], z[2][1], z[2][2], z[2][3], th[e][wt], tl[e][wt]);
                {
                    f32x4 t3[3] = {acc2[0][wt], acc2[1][wt], acc2[2][wt]};
                    cs_mma3_a3(a2h, a2l, zh, zl, t3);
                    acc2[0][wt] = t3[0]; acc2[1][wt] = t3[1]; acc2[2][wt] = t3[2];
                }
            }
        }
        {
            bf16x8 a2h[3], a2l[3];
#pragma unroll
            for (int ot = 0; ot < 3; ++ot) {
                a2h[ot] = *reinterpret_cast<const bf16x8*>(a.packed + csp_offset(3 * q + 2, ot, 0) + 16 * lane);
                a2l[ot] = *reinterpret_cast<const bf16x8*>(a.packed + csp_offset(3 * q + 2, ot, 1) + 16 * lane);
            }
#pragma unroll
            for (int wt = 0; wt < 3; ++wt) {
                const bf16x8 zh = cs_frag(th[0][wt][0], th[0][wt][1], th[1][wt][0], th[1][wt][1]);
                const bf16x8 zl = cs_frag(tl[0][wt][0], tl[0][wt][1], tl[1][wt][0], tl[1][wt][1]);
                f32x4 t3[3] = {acc2[0][wt], acc2[1][wt], acc2[2][wt]};
                cs_mma3_a3(a2h, a2l, zh, zl, t3);
                acc2[0][wt] = t3[0]; acc2[1][wt] = t3[1]; acc2[2][wt] = t3[2];
            }
        }
    }
    // cross-wave sum of the row-partial y2 tiles (fixed order), bias, the sample's BatchNorm2 partial row
    __syncthreads();
#pragma unroll
    for (int ot = 0; ot < 3; ++ot)
#pragma unroll
        for (int wt = 0; wt < 3; ++wt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wv * 48 + 16 * ot + 4 * kg + r) * CSF_RLD + 16 * wt + n] = acc2[ot][wt][r];
    __syncthreads();
    for (int i = t; i < CS_C * CS_W; i += CS_NT) {
        const int o = i / CS_W, w = i % CS_W;
        float v = a.bias2[o];
#pragma unroll
        for (int k = 0; k < CS_NW; ++k) v += red[(k * 48 + o) * CSF_RLD + w];
        a.y2[(long long)b * CS_C * CS_W + i] = v;
        yt[i] = v;
    }
    if (a.stat2) {
        __syncthreads();
        if (t < 2 * CS_C) {
            const int which = t / CS_C, o = t % CS_C;
            double s = 0.0;
            for (int w = 0; w < CS_W; ++w) {
                const double v = (double)yt[o * CS_W + w];
                s += which ? v * v : v;
            }
            a.stat2[(long long)b * 2 * CS_C + t] = s;
        }
    }
}

}  // namespace eeg

using namespace eeg;

static int cs_vec2(const float* x, long long xs_b, long long xs_h) {
    return ((reinterpret_cast<uintptr_t>(x) & 7u) == 0 && (xs_b & 1) == 0 && (xs_h & 1) == 0) ? 1 : 0;
}

extern "C" long long eegclip_cstack_packed_bytes(int H) { return (H < 1 || H > CS_MAXH) ? 0 : csp_offset(3 * cs_pairs(H), 0, 0); }

extern "C" int eegclip_cstack_pack(const float* Ws, void* packed, int H, void* stream) {
    if (!Ws || !packed || H < 1 || H > CS_MAXH) return EEGCLIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(packed) & 15u) return EEGCLIP_EALIGN;
    const int nthr = 3 * cs_pairs(H) * 3 * 64;
    EEG_LAUNCH(cstack_pack_kernel, dim3((nthr + 255) / 256), dim3(256), 0, stream, Ws, static_cast<unsigned char*>(packed), H);
    return (int)hipGetLastError();
}

extern "C" int eegclip_cstack_stats1(const float* x, long long xs_b, long long xs_h, const float* w25, const float* bias, double* rows, int B, int H,
                                     void* stream) {
    if (!x || !w25 || !bias || !rows || B < 1 || H < 1 || H > CS_MAXH) return EEGCLIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(rows) & 7u) return EEGCLIP_EALIGN;
    EEG_LAUNCH(cstack_stats1_kernel, dim3(B), dim3(CS_NT), CS_LDS_S + CS_LDS_PS, stream, x, xs_b, xs_h, w25, bias, rows, B, H, cs_vec2(x, xs_b, xs_h));
    return (int)hipGetLastError();
}

extern "C" int eegclip_cstack_fwd(const eegclip_cstack_fwd_desc* d, void* stream) {
    if (!d || d->B < 1 || d->H < 1 || d->H > CS_MAXH) return EEGCLIP_EINVAL;
    if (!d->x || !d->w25 || !d->bias1 || !d->gamma1 || !d->beta1 || !d->mean1 || !d->rstd1 || !d->packed || !d->bias2 || !d->y2) return EEGCLIP_EINVAL;
    if (d->stat1 && (d->nstat1 < 1 || d->count1 < 1.0)) return EEGCLIP_EINVAL;
    if ((d->run_mean1 == nullptr) != (d->run_var1 == nullptr)) return EEGCLIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d->packed) & 15u) || (reinterpret_cast<uintptr_t>(d->stat1) & 7u) || (reinterpret_cast<uintptr_t>(d->stat2) & 7u))
        return EEGCLIP_EALIGN;
    const cs_fwd_args a{d->x, d->xs_b, d->xs_h, d->w25, d->bias1, d->stat1, d->nstat1, d->count1, d->eps, d->momentum, d->gamma1, d->beta1, d->mean1,
                        d->rstd1, d->run_mean1, d->run_var1, d->nbt1, static_cast<const unsigned char*>(d->packed), d->bias2, d->y2, d->stat2,
                        d->B, d->H, cs_vec2(d->x, d->xs_b, d->xs_h)};
    EEG_LAUNCH(cstack_fwd_kernel, dim3(d->B), dim3(CS_NT), CSF_LDS, stream, a);
    return (int)hipGetLastError();
}
