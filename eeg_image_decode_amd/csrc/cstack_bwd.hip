// Conv stack of Enc_eeg, backward, recomputed from the token rows (see cstack_common.h; Retrieval/ATMS_retrieval.py:102-106).  Given dy2 (B,40,36), the
// gradient entering the spatial conv's output:
//   cstack_bwd_kernel<false>   BatchNorm1-backward sums: per sample and token row h the y1 tile (tap contraction) and dz1 = Ws^T dy2 (K = 40 out channels)
//                              on the matrix cores, da = dz1 * ELU'(BN1(y1)) in registers -> one partial row [sum da | sum da * xhat] per sample
//   cstack_bwd_kernel<true>    the same tiles again, dy1 = BatchNorm1-backward(da) in registers, then
//                                * E[w][t] = sum_c dy1[c][w] taps[c][t]  (the accumulator tiles ARE the operand), overlap-added through a wave-private LDS
//                                  tile into dS[j = 5w + t], transposed box filter -> the token-row gradient dx[b][h][:]
//                                * dW1[c][t] += sum_w dy1[c][w] S[h][5w + t]: dy1 transposed through a wave-private packed LDS tile
//   cstack_bwd_w2_kernel       dWs[o][c][h] = sum_{b,w} dy2[b][o][w] z1[b][c][h][w]: workgroup = (4 token rows, sample group), a wave owns ONE row h of
//                              every sample of its group: y1^T tile -> z1^T in registers = the k = w operand; per-group slabs + an ordered reduction
//   cstack_pack_t_kernel       Ws (40,40,H) -> Ws^T fragments (rows = filters c, k = out channels o), once per optimizer step
// Replaces sconv_bwd_w (+ reduce), sconv_bwd_x<stats> (+ colsum), sconv_bwd_x<apply> (dy1 write, 93 MB), tsconv_bwd_w (+ reduce), tsconv_bwd_x.
#include "cstack_common.h"

#include <stdlib.h>

namespace eeg {

// ---- Ws^T fragments: see cstack_common.h (cs_pack_t_item) -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cstack_pack_t_kernel(const float* __restrict__ Ws, unsigned char* __restrict__ packed, int H) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= H * 3 * 64) return;
    cs_pack_t_item(Ws, packed, H, id);
}
// both fragment sets of a step in ONE launch
__global__ __launch_bounds__(256) void cstack_pack_all_kernel(const float* __restrict__ Ws, unsigned char* __restrict__ packed, unsigned char* __restrict__ packed_t, int H) {
    cs_pack_both(Ws, packed, packed_t, H, blockIdx.x * blockDim.x + threadIdx.x);
}

__device__ __forceinline__ bf16x8 cs_half_frag(u32x2_t v) { return cs_frag(v[0], v[1], 0u, 0u); }

// ---- BatchNorm1-backward sums / apply + temporal-conv backward ------------------------------------------------------------------------------------------
struct cs_bwd_args {
    const float* x;
    long long xs_b, xs_h;
    const float *w25, *bias1;
    const float *mean1, *rstd1, *gamma1, *beta1;
    const unsigned char* packed_t;
    const float* dy2;
    double* rows_out;               // stats pass: [B][80]
    const double* stat;             // apply pass: partial rows [sum da | sum da * xhat] (nstat of them) and the element count
    int nstat;
    double count;
    const double* stat_local;       // this rank's own sums for dgamma / dbeta (NULL: stat)
    int nstat_local;
    float *dgamma, *dbeta;
    float* dx;
    float* dw_partials;             // [B][40 * 25]
    int B, H, vec2;
};

// E[w][t] = sum_c dy1[c][w] taps[c][t] is kept as FIVE PLANES indexed by the output sample: plane a = t / 5 holds E[w][t] at word 5 w + t (= 5 (w + a) + t % 5),
// so the overlap-add dS[j] = sum_a E[j / 5 - a][j % 5 + 5 a] is sum_a plane_a[j]: one 16-byte read per plane and lane (j = 4 lane + e), no index arithmetic.
// Words of a plane no E element lands on (j < 5 a and j >= 180 + 5 a) stay zero.  (Round 5 kept E as [36][27] and gathered 20 words per lane through
// computed, range-checked addresses: ~120 LDS cycles of a possible 40 per row, 2.5-way conflicts on average, and ~100 vector instructions.)
constexpr int CSB_EP = 200;                                 // words per plane = CS_NS
constexpr int CSB_E_BYTES = 5 * CSB_EP * 4;                 // 4000
constexpr int CSB_DT_BYTES = CS_C * CS_W * 4;               // 5760: dy1 as packed words [40 c][36 w]
constexpr int CSB_WAVE = CSB_E_BYTES + CSB_DT_BYTES;        // 9648
constexpr int CSB_DYF = 3 * 3072;                           // dy2 fragment images (3 position tiles)
constexpr int CSB_TAPF = 2 * 3072;                          // taps as the k = filter operand (2 tap tiles)
constexpr int CSB_NCOEF = 5;                                // gamma beta K -S1/n -S2/n

template <bool APPLY>
__global__ __launch_bounds__(CS_NT) void cstack_bwd_kernel(const cs_bwd_args a) {
    EEG_LDS_BASE(unsigned char, ldsb);
    const int H = a.H;
    unsigned* S32 = reinterpret_cast<unsigned*>(ldsb);
    unsigned char* p = ldsb + H * CS_RS * 4;
    float* coef = reinterpret_cast<float*>(p);               // [7][48]
    p += CSB_NCOEF * 48 * 4;
    unsigned char* dyf = p;                                  // dy2 fragments: per position tile  main hi | main lo | tail hi | tail lo
    p += CSB_DYF;
    unsigned char* tapf = p;                                 // (APPLY) taps fragments: per tap tile the same four images
    p += APPLY ? CSB_TAPF : 0;
    unsigned char* wreg = p;                                 // per-wave regions
    const int t = threadIdx.x, lane = t & 63, wv = wave_uniform(t >> 6);
    const int n = lane & 15, kg = lane >> 4;
    const int b = blockIdx.x;
    constexpr int WREG = APPLY ? CSB_WAVE : 1024;
    unsigned char* mine = wreg + wv * WREG;
    float* Et = reinterpret_cast<float*>(mine);              // [5][200] (CSB_EP)
    unsigned* DT = reinterpret_cast<unsigned*>(mine + (APPLY ? CSB_E_BYTES : 0));
    float* qscr = reinterpret_cast<float*>(mine);            // 256 floats of prefix-sum scratch (staging; the transposed box filter once E is consumed)
    double* bnscr = reinterpret_cast<double*>(wreg);         // [6][80] (before the staging uses the region)

    float vx[CS_RPW][4];
    cs_stage_load<false>(vx, a.x, a.xs_b, a.xs_h, b, H, a.vec2 != 0);
    if (APPLY) {
        cs_bn_rows_partial(a.stat, a.nstat, bnscr);
        if (b == 0 && a.stat_local) cs_bn_rows_partial(a.stat_local, a.nstat_local, bnscr + CS_BN_SLICES * 2 * CS_C);
    }
    // dy2 of this sample as the k = out-channel operand: lane (n, kg) of position tile wt <- dy2[o][w = 16 wt + n]
    if (t < 192) {
        const int wt = t >> 6, w = 16 * wt + n;
        const float* src = a.dy2 + (long long)b * CS_C * CS_W + w;
        float v[8], tl[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = w < CS_W ? src[(16 * (j >> 2) + 4 * kg + (j & 3)) * CS_W] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) tl[j] = (w < CS_W && 32 + 4 * kg + j < CS_C) ? src[(32 + 4 * kg + j) * CS_W] : 0.f;
        bf16x8 hi, lo;
        cs_split8(v, hi, lo);
        u32x2_t th, tlo;
        x3_split4(tl[0], tl[1], tl[2], tl[3], th, tlo);
        unsigned char* base = dyf + wt * 3072;
        *reinterpret_cast<bf16x8*>(base + 16 * lane) = hi;
        *reinterpret_cast<bf16x8*>(base + 1024 + 16 * lane) = lo;
        *reinterpret_cast<u32x2_t*>(base + 2048 + 8 * lane) = th;
        *reinterpret_cast<u32x2_t*>(base + 2560 + 8 * lane) = tlo;
    } else if (APPLY && t < 320) {
        // taps as the k = filter operand of E = dy1^T taps: lane (n, kg) of tap tile tt <- w25[c][t = 16 tt + n]
        const int tt = (t - 192) >> 6, tp = 16 * tt + n;
        float v[8], tl[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tp < CS_K1 ? a.w25[(16 * (j >> 2) + 4 * kg + (j & 3)) * CS_K1 + tp] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) tl[j] = (tp < CS_K1 && 32 + 4 * kg + j < CS_C) ? a.w25[(32 + 4 * kg + j) * CS_K1 + tp] : 0.f;
        bf16x8 hi, lo;
        cs_split8(v, hi, lo);
        u32x2_t th, tlo;
        x3_split4(tl[0], tl[1], tl[2], tl[3], th, tlo);
        unsigned char* base = tapf + tt * 3072;
        *reinterpret_cast<bf16x8*>(base + 16 * lane) = hi;
        *reinterpret_cast<bf16x8*>(base + 1024 + 16 * lane) = lo;
        *reinterpret_cast<u32x2_t*>(base + 2048 + 8 * lane) = th;
        *reinterpret_cast<u32x2_t*>(base + 2560 + 8 * lane) = tlo;
    }
    __syncthreads();                                          // bnscr complete
    if (t < 48) {
        // u = G * xhat + Bt;  dy1 = K * (da + NM1 + xhat * NM2)
        float G = 0.f, Bt = 0.f, K = 0.f, NM1 = 0.f, NM2 = 0.f;
        if (t < CS_C) {
            const float rstd = a.rstd1[t];
            G = a.gamma1[t];
            Bt = a.beta1[t];
            if (APPLY) {
                double s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int sl = 0; sl < CS_BN_SLICES; ++sl) { s1 += bnscr[sl * 2 * CS_C + t]; s2 += bnscr[sl * 2 * CS_C + CS_C + t]; }
                K = G * rstd;
                NM1 = -(float)(s1 / a.count);
                NM2 = -(float)(s2 / a.count);
                if (b == 0) {                                // BatchNorm1 parameter gradients from this rank's own sums (fixed summation order)
                    if (a.stat_local) {
                        s1 = 0.0;
                        s2 = 0.0;
                        const double* loc = bnscr + CS_BN_SLICES * 2 * CS_C;
#pragma unroll
                        for (int sl = 0; sl < CS_BN_SLICES; ++sl) { s1 += loc[sl * 2 * CS_C + t]; s2 += loc[sl * 2 * CS_C + CS_C + t]; }
                    }
                    atomicAdd(a.dbeta + t, (float)s1);
                    atomicAdd(a.dgamma + t, (float)s2);
                }
            }
        }
        coef[0 * 48 + t] = G; coef[1 * 48 + t] = Bt; coef[2 * 48 + t] = K; coef[3 * 48 + t] = NM1; coef[4 * 48 + t] = NM2;
    }
    // taps scaled by rstd with (bias - mean) * rstd in the ones slot: the tap contraction yields xhat = (y1 - mean) * rstd itself
    bf16x8 wh[3], wl[3];
    cs_tap_frags_affine(a.w25, [&](int c, float& sc, float& sh) { sc = a.rstd1[c]; sh = (a.bias1[c] - a.mean1[c]) * sc; }, wh, wl);
    __syncthreads();                                          // bnscr consumed (the per-wave regions may be used), coefficients + fragment images visible
    cs_stage_finish<false>(S32, qscr, vx, H);                 // (a wave works on the rows it staged: no workgroup barrier)

    const f32x4 zero4{0.f, 0.f, 0.f, 0.f};
    f32x2_t s1[3][2], s2[3][2];
    f32x4 acc5[3][2];                                         // (APPLY) taps gradient D[c = 16 ct + 4 kg + r][t = 16 tt + n]
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
#pragma unroll
        for (int k = 0; k < 2; ++k) { s1[ct][k] = f32x2_t{0.f, 0.f}; s2[ct][k] = f32x2_t{0.f, 0.f}; }
        acc5[ct][0] = zero4;
        acc5[ct][1] = zero4;
    }
    // the 20 words of every E plane that no element lands on (plane a: j < 5 a, j >= 180 + 5 a); the prefix-sum scratch shares the first 256 words with
    // the planes, so the row loop re-zeroes the 25 of them it overwrites (cs_e_border)
    auto cs_e_border = [&](int i) { const int pa = i / 20, idx = i % 20; return pa * CSB_EP + (idx < 5 * pa ? idx : 180 + idx); };
    if (APPLY) {
        Et[cs_e_border(lane)] = 0.f;
        if (lane < 36) Et[cs_e_border(64 + lane)] = 0.f;
    }
    const unsigned char* const dyf0 = dyf;
    const unsigned char* const tapf0 = tapf;
    const float* const coef0 = coef;
    for (int h = wv; h < H; h += CS_NW) {
        const int oz = cs_opaque_zero();                      // (see cstack_common.h: keeps the loop-invariant LDS operands out of registers)
        const unsigned char* dyf = dyf0 + oz;
        const unsigned char* tapf = tapf0 + oz;
        const float* coef = coef0 + oz;
        // Ws^T fragments of this row: straight from L2 (every workgroup reads the same rows)
        bf16x8 th_[3], tl_[3], uh_[3], ul_[3];
        {
            const unsigned char* base = a.packed_t + (long long)h * CST_ROW;
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) {
                th_[ct] = *reinterpret_cast<const bf16x8*>(base + ct * CST_TILE + 16 * lane);
                tl_[ct] = *reinterpret_cast<const bf16x8*>(base + ct * CST_TILE + 1024 + 16 * lane);
                uh_[ct] = cs_half_frag(*reinterpret_cast<const u32x2_t*>(base + ct * CST_TILE + 2048 + 8 * lane));
                ul_[ct] = cs_half_frag(*reinterpret_cast<const u32x2_t*>(base + ct * CST_TILE + 2560 + 8 * lane));
            }
        }
#pragma unroll
        for (int wt = 0; wt < 3; ++wt) {
            bf16x8 xh, xl;
            cs_sfrag_ones(S32, h, wt, xh, xl);
            const bf16x8 dh = *reinterpret_cast<const bf16x8*>(dyf + wt * 3072 + 16 * lane);
            const bf16x8 dl = *reinterpret_cast<const bf16x8*>(dyf + wt * 3072 + 1024 + 16 * lane);
            const bf16x8 eh = cs_half_frag(*reinterpret_cast<const u32x2_t*>(dyf + wt * 3072 + 2048 + 8 * lane));
            const bf16x8 el = cs_half_frag(*reinterpret_cast<const u32x2_t*>(dyf + wt * 3072 + 2560 + 8 * lane));
            f32x4 xhat[3] = {zero4, zero4, zero4}, dz[3] = {zero4, zero4, zero4};
            cs_mma3_a3(wh, wl, xh, xl, xhat);                 // xhat = (y1 - mean) * rstd:  D[c = 16 ct + 4 kg + r][w = 16 wt + n]
            cs_mma3_a3(th_, tl_, dh, dl, dz);                 // dz1 = Ws^T dy2, out channels 0 .. 31
            cs_mma3_a3(uh_, ul_, eh, el, dz);                 //                 out channels 32 .. 39   (exact zeros at w >= 36 and c >= 40: zero operands)
            f32x2_t dy1[3][2];
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) {
                const f32x4 G = *reinterpret_cast<const f32x4*>(coef + 0 * 48 + 16 * ct + 4 * kg), Bt = *reinterpret_cast<const f32x4*>(coef + 1 * 48 + 16 * ct + 4 * kg);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const f32x2_t xk = k ? cs_hi2(xhat[ct]) : cs_lo2(xhat[ct]), zk = k ? cs_hi2(dz[ct]) : cs_lo2(dz[ct]);
                    const f32x2_t u = cs_fma2(xk, k ? cs_hi2(G) : cs_lo2(G), k ? cs_hi2(Bt) : cs_lo2(Bt));
                    const f32x2_t da = zk * cs_elu_grad2(u);                           // da = dz1 * ELU'(u)
                    if (APPLY) {
                        const f32x4 K = *reinterpret_cast<const f32x4*>(coef + 2 * 48 + 16 * ct + 4 * kg), NM1 = *reinterpret_cast<const f32x4*>(coef + 3 * 48 + 16 * ct + 4 * kg);
                        const f32x4 NM2 = *reinterpret_cast<const f32x4*>(coef + 4 * 48 + 16 * ct + 4 * kg);
                        const f32x2_t tq = cs_fma2(xk, k ? cs_hi2(NM2) : cs_lo2(NM2), da) + (k ? cs_hi2(NM1) : cs_lo2(NM1));
                        dy1[ct][k] = tq * (k ? cs_hi2(K) : cs_lo2(K));                 // dy1 = gamma * rstd * (da - S1 / n - xhat * S2 / n)
                    } else {
                        s1[ct][k] += da;
                        s2[ct][k] = cs_fma2(da, xk, s2[ct][k]);
                    }
                }
            }
            if (APPLY) {
                if (wt == 2) {                                // positions >= 36 of the last tile: no gradient (their xhat is not a real position's)
                    const float wm = 32 + n < CS_W ? 1.f : 0.f;
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) { dy1[ct][0] = dy1[ct][0] * f32x2_t{wm, wm}; dy1[ct][1] = dy1[ct][1] * f32x2_t{wm, wm}; }
                }
                // split once: the halves are the A operand of E = dy1^T taps (row = position w = 16 wt + n, k slot j <-> c = 16 (j >> 2) + 4 kg + (j & 3)) AND, re-paired
                // into (hi << 16) | lo words, the transposed tile DT[c][w] the taps gradient contracts over w from
                u32x2_t sh_[3], sl_[3];
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) x3_split4(dy1[ct][0][0], dy1[ct][0][1], dy1[ct][1][0], dy1[ct][1][1], sh_[ct], sl_[ct]);
                const bf16x8 ah = cs_frag(sh_[0][0], sh_[0][1], sh_[1][0], sh_[1][1]), al = cs_frag(sl_[0][0], sl_[0][1], sl_[1][0], sl_[1][1]);
                const bf16x8 a2h = cs_half_frag(sh_[2]), a2l = cs_half_frag(sl_[2]);
                bf16x8 bh[2], bl[2], ch[2], cl[2];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    bh[tt] = *reinterpret_cast<const bf16x8*>(tapf + tt * 3072 + 16 * lane);
                    bl[tt] = *reinterpret_cast<const bf16x8*>(tapf + tt * 3072 + 1024 + 16 * lane);
                    ch[tt] = cs_half_frag(*reinterpret_cast<const u32x2_t*>(tapf + tt * 3072 + 2048 + 8 * lane));
                    cl[tt] = cs_half_frag(*reinterpret_cast<const u32x2_t*>(tapf + tt * 3072 + 2560 + 8 * lane));
                }
                f32x4 accE[2] = {zero4, zero4};               // D[w = 16 wt + 4 kg + r][t = 16 tt + n], product-major over the two tap tiles
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) accE[tt] = mfma_bf16_16x16x32(ah, bl[tt], accE[tt]);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) accE[tt] = mfma_bf16_16x16x32(a2h, cl[tt], accE[tt]);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) accE[tt] = mfma_bf16_16x16x32(al, bh[tt], accE[tt]);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) accE[tt] = mfma_bf16_16x16x32(a2l, ch[tt], accE[tt]);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) accE[tt] = mfma_bf16_16x16x32(ah, bh[tt], accE[tt]);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) accE[tt] = mfma_bf16_16x16x32(a2h, ch[tt], accE[tt]);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int tp = 16 * tt + n;
                    if (tp < CS_K1) {
                        float* ep = Et + (tp / 5) * CSB_EP + tp + 5 * (16 * wt + 4 * kg);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int w = 16 * wt + 4 * kg + r;
                            if (w < CS_W) ep[5 * r] = accE[tt][r];
                        }
                    }
                }
                if (wt < 2 || 32 + n < CS_W) {
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) {
                        const int c0 = 16 * ct + 4 * kg;
                        if (c0 < CS_C) {                      // (40 = 10 groups of 4 filters: a group is all in or all out)
                            unsigned* d = DT + c0 * CS_W + 16 * wt + n;
                            d[0 * CS_W] = cs_pair_lo(sh_[ct][0], sl_[ct][0]);       // (hi(r) << 16) | lo(r) of r = 0 .. 3
                            d[1 * CS_W] = cs_pair_hi(sh_[ct][0], sl_[ct][0]);
                            d[2 * CS_W] = cs_pair_lo(sh_[ct][1], sl_[ct][1]);
                            d[3 * CS_W] = cs_pair_hi(sh_[ct][1], sl_[ct][1]);
                        }
                    }
                }
            }
        }
        if (APPLY) {
            wave_sync();
            // overlap-add: dS[j] = sum_{a < 5} E[j / 5 - a][j % 5 + 5 a] = sum_a plane_a[j], lane l < 50 owns j = 4 l .. 4 l + 3
            float ds[4];
            {
                const float* ep = Et + 4 * (lane < CS_NS / 4 ? lane : 0);
                f32x4 s = *reinterpret_cast<const f32x4*>(ep);
#pragma unroll
                for (int k = 1; k < 5; ++k) s += *reinterpret_cast<const f32x4*>(ep + k * CSB_EP);
#pragma unroll
                for (int e = 0; e < 4; ++e) ds[e] = lane < CS_NS / 4 ? s[e] : 0.f;
            }
            wave_sync();                                      // E is consumed: its first KB becomes the prefix-sum scratch
            // transposed box filter: Q[i] = sum_{k < i} dS[k];  dx[i] = (Q[i + 1] - Q[max(i - 50, 0)]) / 51
            {
                const float p0 = ds[0], p1 = p0 + ds[1], p2 = p1 + ds[2], p3 = p2 + ds[3];
                const float basev = cs_wave_scan(p3) - p3;
                const float q4[5] = {basev, basev + p0, basev + p1, basev + p2, basev + p3};
                *reinterpret_cast<f32x4*>(qscr + 4 * lane) = f32x4{q4[0], q4[1], q4[2], q4[3]};
                wave_sync();
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * lane + e;
                    const float lo = i >= CS_POOL - 1 ? qscr[i - (CS_POOL - 1)] : 0.f;
                    o[e] = (q4[e + 1] - lo) * (1.0f / CS_POOL);
                }
                float* xr = a.dx + (long long)b * a.xs_b + (long long)h * a.xs_h;
                if (a.vec2) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
                        if (4 * lane + 2 * hf < CS_T) *reinterpret_cast<f32x2_t*>(xr + 4 * lane + 2 * hf) = f32x2_t{o[2 * hf], o[2 * hf + 1]};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (4 * lane + e < CS_T) xr[4 * lane + e] = o[e];
                }
            }
            // taps gradient: dW1[c][t] += sum_w dy1[c][w] S[h][5 w + t];  A = DT rows, k slot j <-> w = 16 (kg & 1) + 8 (kg >> 1) + j (the lane groups a 4-byte
            // LDS read serves together, kg = 0 | 1 and 2 | 3, sit 80 words apart: their 16-word runs share no bank); tail w = 32 + j in lane group 0 (the other
            // groups' A slots are zero: they read the same words, a broadcast)
            bf16x8 bh[2], bl[2], ch[2], cl[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const unsigned* sp = S32 + h * CS_RS + 16 * tt + n;
                unsigned w8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) w8[j] = sp[5 * (cs_tap_base(kg) + j)];
                cs_words_to_frags(w8, bh[tt], bl[tt]);
                unsigned w4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) w4[j] = sp[5 * (32 + j)];
                ch[tt] = cs_frag(cs_pair_hi(w4[1], w4[0]), cs_pair_hi(w4[3], w4[2]), 0u, 0u);
                cl[tt] = cs_frag(cs_pair_lo(w4[1], w4[0]), cs_pair_lo(w4[3], w4[2]), 0u, 0u);
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                bf16x8 ah[3], al[3], a2h[3], a2l[3];
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) {
                    const int c = 16 * ct + n < CS_C ? 16 * ct + n : CS_C - 1;          // (filters >= 40: discarded output rows)
                    const u32x4_t m0 = *reinterpret_cast<const u32x4_t*>(DT + c * CS_W + cs_tap_base(kg));
                    const u32x4_t m1 = *reinterpret_cast<const u32x4_t*>(DT + c * CS_W + cs_tap_base(kg) + 4);
                    u32x4_t m2 = *reinterpret_cast<const u32x4_t*>(DT + c * CS_W + 32);
                    if (kg != 0) m2 = u32x4_t{0u, 0u, 0u, 0u};                            // positions >= 36
                    const unsigned w8[8] = {m0[0], m0[1], m0[2], m0[3], m1[0], m1[1], m1[2], m1[3]};
                    cs_words_to_frags(w8, ah[ct], al[ct]);
                    a2h[ct] = cs_frag(cs_pair_hi(m2[1], m2[0]), cs_pair_hi(m2[3], m2[2]), 0u, 0u);
                    a2l[ct] = cs_frag(cs_pair_lo(m2[1], m2[0]), cs_pair_lo(m2[3], m2[2]), 0u, 0u);
                }
                f32x4 t3[3] = {acc5[0][tt], acc5[1][tt], acc5[2][tt]};
                cs_mma3_a3(ah, al, bh[tt], bl[tt], t3);
                cs_mma3_a3(a2h, a2l, ch[tt], cl[tt], t3);
                acc5[0][tt] = t3[0]; acc5[1][tt] = t3[1]; acc5[2][tt] = t3[2];
            }
            wave_sync();                                      // the next row rewrites E / DT
            if (lane < 25) Et[cs_e_border(lane)] = 0.f;        // the scratch lay over plane 0's words 180 .. 199 and plane 1's words 0 .. 4
        }
    }

    __syncthreads();                                          // every wave is done: LDS becomes reduction scratch
    if (!APPLY) {
        float* sc = reinterpret_cast<float*>(ldsb);           // [NW][2][48]
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u = s1[ct][r >> 1][r & 1], v = s2[ct][r >> 1][r & 1];
#pragma unroll
                for (int msk = 8; msk >= 1; msk >>= 1) { u += __shfl_xor(u, msk, 64); v += __shfl_xor(v, msk, 64); }
                if (n == 0) { sc[(wv * 2 + 0) * 48 + 16 * ct + 4 * kg + r] = u; sc[(wv * 2 + 1) * 48 + 16 * ct + 4 * kg + r] = v; }
            }
        __syncthreads();
        if (t < 2 * CS_C) {
            const int which = t / CS_C, c = t % CS_C;
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < CS_NW; ++k) s += (double)sc[(k * 2 + which) * 48 + c];
            a.rows_out[(long long)b * 2 * CS_C + t] = s;
        }
    } else {
        float* red = reinterpret_cast<float*>(ldsb);          // [NW][48][33]
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(wv * 48 + 16 * ct + 4 * kg + r) * 33 + 16 * tt + n] = acc5[ct][tt][r];
        __syncthreads();
        for (int i = t; i < CS_C * CS_K1; i += CS_NT) {
            const int c = i / CS_K1, tp = i % CS_K1;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < CS_NW; ++k) s += red[(k * 48 + c) * 33 + tp];
            a.dw_partials[(long long)b * CS_C * CS_K1 + i] = s;
        }
    }
}

// out[i] += sum_r partials[r][i], rows in a fixed order: workgroup = 16 elements x 16 row slices (thread (sl, e) sums rows sl, sl + 16, ...: 16 independent
// loads in flight for 256 rows), the slices added pairwise in a fixed tree
__global__ __launch_bounds__(256) void cstack_rows_reduce_kernel(const float* __restrict__ partials, int nrows, int n, float* __restrict__ out) {
    EEG_LDS_BASE(float, red);                                 // [16][16]
    const int t = threadIdx.x, sl = t >> 4, e = t & 15;
    const int i = blockIdx.x * 16 + e;
    float s = 0.f;
    if (i < n) {
#pragma unroll 16
        for (int k = sl; k < nrows; k += 16) s += partials[(long long)k * n + i];
    }
    red[sl * 16 + e] = s;
    __syncthreads();
#pragma unroll
    for (int half = 8; half >= 1; half >>= 1) {
        if (sl < half) red[sl * 16 + e] += red[(sl + half) * 16 + e];
        __syncthreads();
    }
    if (sl == 0 && i < n) out[i] += red[e];
}

// dWs[oc][h] += sum_g slabs[g][h][oc] (oc = o * 40 + c; groups in order) as a tiled transpose: workgroup = 32 rows h x 32 columns oc; reads are 128-byte
// runs of a slab, the sum meets in an LDS tile, writes are 128-byte runs of dWs
__global__ __launch_bounds__(256) void cstack_w2_reduce_kernel(const float* __restrict__ slabs, int G, int H, float* __restrict__ dWs) {
    EEG_LDS_BASE(float, tile);                                // [32 h][33]
    const int t = threadIdx.x, col = t & 31, r8 = t >> 5;
    const int oc0 = 32 * blockIdx.x, h0 = 32 * blockIdx.y;
    constexpr int N = CS_C * CS_C;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int h = h0 + r8 + 8 * k;
        float s = 0.f;
        if (h < H) {
            const float* src = slabs + (long long)h * N + oc0 + col;
#pragma unroll 8
            for (int g = 0; g < G; ++g) s += src[(long long)g * H * N];
        }
        tile[(r8 + 8 * k) * 33 + col] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int oc = oc0 + r8 + 8 * k, h = h0 + col;
        if (h < H) dWs[(long long)oc * H + h] += tile[col * 33 + r8 + 8 * k];
    }
}

// ---- spatial-conv weight gradient -----------------------------------------------------------------------------------------------------------------------
// workgroup (row block of 4 token rows, sample group g): wave v owns row h = 4 blockIdx.x + v and walks the samples g, g + G, ...
//   u^T tile D[w][c] = BN1(y1)^T: the operands of the forward's tap contraction swapped (taps scaled by gamma * rstd, the BatchNorm constant in the ones
//   slot) -> z1^T = ELU(.) in registers = the k = w operand (k slot j <-> w = 16 (j >> 2) + 4 kg + (j & 3); positions 32 .. 35 as a second, mostly empty
//   k-step) of dWs[o][c] += sum_w dy2[b][o][w] z1[c][w]; the dy2 fragments (rows o, the same k slots) come from global memory, 16 bytes per quarter.
// workgroup = (2 token rows, 4 sample sub-groups): the sub-groups' tiles are summed in LDS in a fixed order, so only SG <= 8 slabs [SG][H][40 o][40 c]
// leave (3.2 MB; 32 slabs of one group each cost a 72-us reduction in the step), summed by cstack_w2_reduce_kernel, a tiled transpose to dWs[o][c][h].
constexpr int CSW_NW = 8;                                      // 2 token rows x 4 sample sub-groups per workgroup
__global__ __launch_bounds__(64 * CSW_NW) void cstack_bwd_w2_kernel(const float* __restrict__ x, long long xs_b, long long xs_h, const float* __restrict__ w25,
                                                                    const float* __restrict__ bias1, const float* __restrict__ mean1,
                                                                    const float* __restrict__ rstd1, const float* __restrict__ gamma1,
                                                                    const float* __restrict__ beta1, const float* __restrict__ dy2, float* __restrict__ slabs,
                                                                    int B, int H, int SG, int vec2) {
    EEG_LDS_BASE(unsigned char, ldsb);
    const int t = threadIdx.x, lane = t & 63, wv = wave_uniform(t >> 6);
    const int n = lane & 15, kg = lane >> 4;
    unsigned* srow = reinterpret_cast<unsigned*>(ldsb) + wv * (CS_RS + 256);      // the wave's packed row + its prefix-sum scratch
    float* pscr = reinterpret_cast<float*>(srow + CS_RS);
    float* tile = reinterpret_cast<float*>(ldsb) + CSW_NW * (CS_RS + 256);       // [8 waves][1600]: the sub-groups' results before they are summed
    const int G = 4 * SG;                                      // sample groups in all: wave (row, q) walks the samples g = 4 sg + q, g + G, ...
    const int h = 2 * blockIdx.x + (wv & 1), g = 4 * blockIdx.y + (wv >> 1);
    const bool active = h < H && g < B;                        // (a last row block may be short, a small batch may not fill the sub-groups)
    const f32x4 zero4{0.f, 0.f, 0.f, 0.f};
    f32x4 acc[3][3];                                          // D[o = 16 ot + 4 kg + r][c = 16 ct + n]
#pragma unroll
    for (int ot = 0; ot < 3; ++ot)
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) acc[ot][ct] = zero4;
    if (active) {
        bf16x8 wh[3], wl[3];
        cs_tap_frags_affine(w25, [&](int c, float& sc, float& sh) { sc = gamma1[c] * rstd1[c]; sh = beta1[c] + (bias1[c] - mean1[c]) * sc; }, wh, wl);
        float vx[4];
        f32x4 vd[3][3];
        auto load_sample = [&](int b) {
            cs_load_row(vx, x + (long long)b * xs_b + (long long)h * xs_h, true, vec2 != 0);
            const float* src = dy2 + (long long)b * CS_C * CS_W;
#pragma unroll
            for (int ot = 0; ot < 3; ++ot) {
                const int o = 16 * ot + n;
                const bool ok = o < CS_C;
                const float* r = src + (ok ? o : 0) * CS_W;
                vd[ot][0] = ok ? *reinterpret_cast<const f32x4*>(r + 4 * kg) : zero4;
                vd[ot][1] = ok ? *reinterpret_cast<const f32x4*>(r + 16 + 4 * kg) : zero4;
                vd[ot][2] = (ok && kg == 0) ? *reinterpret_cast<const f32x4*>(r + 32) : zero4;
            }
        };
        if (g < B) load_sample(g);
        for (int b = g; b < B; b += G) {
            cs_box_row(srow, pscr, vx);
            bf16x8 dh[3], dl[3], eh[3], el[3];
#pragma unroll
            for (int ot = 0; ot < 3; ++ot) {
                const float v8[8] = {vd[ot][0][0], vd[ot][0][1], vd[ot][0][2], vd[ot][0][3], vd[ot][1][0], vd[ot][1][1], vd[ot][1][2], vd[ot][1][3]};
                cs_split8(v8, dh[ot], dl[ot]);
                u32x2_t a_, b_;
                x3_split4(vd[ot][2][0], vd[ot][2][1], vd[ot][2][2], vd[ot][2][3], a_, b_);
                eh[ot] = cs_half_frag(a_);
                el[ot] = cs_half_frag(b_);
            }
            if (b + G < B) load_sample(b + G);                // the next sample's loads land under this sample's MFMAs
            wave_sync();
            f32x2_t z[3][3][2];                               // z1^T[w = 16 wt + 4 kg + r][c = 16 ct + n]
#pragma unroll
            for (int wt = 0; wt < 3; ++wt) {
                bf16x8 xh, xl;
                cs_sfrag_ones(srow, 0, wt, xh, xl);
                f32x4 u[3] = {zero4, zero4, zero4};
                cs_mma3_b3(xh, xl, wh, wl, u);                // u^T = BN1(y1)^T
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) { z[wt][ct][0] = cs_elu2(cs_lo2(u[ct])); z[wt][ct][1] = cs_elu2(cs_hi2(u[ct])); }
            }
            wave_sync();                                      // the row is consumed: the next sample may overwrite it
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) {
                bf16x8 zh, zl;
                {
                    const float v8[8] = {z[0][ct][0][0], z[0][ct][0][1], z[0][ct][1][0], z[0][ct][1][1], z[1][ct][0][0], z[1][ct][0][1], z[1][ct][1][0], z[1][ct][1][1]};
                    cs_split8(v8, zh, zl);
                }
                u32x2_t th, tl;
                x3_split4(z[2][ct][0][0], z[2][ct][0][1], z[2][ct][1][0], z[2][ct][1][1], th, tl);     // (positions >= 36: finite, and the dy2 operand is zero there)
                f32x4 t3[3] = {acc[0][ct], acc[1][ct], acc[2][ct]};
                cs_mma3_a3(dh, dl, zh, zl, t3);
                cs_mma3_a3(eh, el, cs_half_frag(th), cs_half_frag(tl), t3);
                acc[0][ct] = t3[0]; acc[1][ct] = t3[1]; acc[2][ct] = t3[2];
            }
        }
    }
    // the four sub-groups of a row meet in LDS (q = 0 .. 3 in order) and leave as slab[sg][h][o][c]: 8 slabs instead of 32
#pragma unroll
    for (int ot = 0; ot < 3; ++ot)
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 16 * ot + 4 * kg + r, c = 16 * ct + n;
                if (o < CS_C && c < CS_C) tile[wv * CS_C * CS_C + o * CS_C + c] = acc[ot][ct][r];      // (idle waves: zeros)
            }
    __syncthreads();
    for (int i = t; i < 2 * CS_C * CS_C; i += 64 * CSW_NW) {
        const int row = i / (CS_C * CS_C), oc = i % (CS_C * CS_C), hh = 2 * blockIdx.x + row;
        if (hh < H) {
            const float* p = tile + row * CS_C * CS_C + oc;
            slabs[((long long)blockIdx.y * H + hh) * CS_C * CS_C + oc] = ((p[0] + p[2 * CS_C * CS_C]) + p[4 * CS_C * CS_C]) + p[6 * CS_C * CS_C];
        }
    }
}

}  // namespace eeg

using namespace eeg;

static int csb_vec2(const float* x, long long xs_b, long long xs_h) {
    return ((reinterpret_cast<uintptr_t>(x) & 7u) == 0 && (xs_b & 1) == 0 && (xs_h & 1) == 0) ? 1 : 0;
}
static int csw_groups(int B) { const int sg = (B + 3) / 4; return sg < 8 ? sg : 8; }      // slabs: 32 row blocks x 8 x 8 waves = 2 waves per SIMD

extern "C" long long eegclip_cstack_packed_t_bytes(int H) { return (H < 1 || H > CS_MAXH) ? 0 : (long long)H * CST_ROW; }

extern "C" int eegclip_cstack_pack_all(const float* Ws, void* packed, void* packed_t, int H, void* stream) {
    if (!Ws || !packed || H < 1 || H > CS_MAXH) return EEGCLIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(packed_t)) & 15u) return EEGCLIP_EALIGN;
    const int nthr = cs_pack_items(H, packed_t != nullptr);
    EEG_LAUNCH(cstack_pack_all_kernel, dim3((nthr + 255) / 256), dim3(256), 0, stream, Ws, static_cast<unsigned char*>(packed), static_cast<unsigned char*>(packed_t), H);
    return (int)hipGetLastError();
}

extern "C" int eegclip_cstack_pack_t(const float* Ws, void* packed_t, int H, void* stream) {
    if (!Ws || !packed_t || H < 1 || H > CS_MAXH) return EEGCLIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(packed_t) & 15u) return EEGCLIP_EALIGN;
    EEG_LAUNCH(cstack_pack_t_kernel, dim3((H * 192 + 255) / 256), dim3(256), 0, stream, Ws, static_cast<unsigned char*>(packed_t), H);
    return (int)hipGetLastError();
}

static int csb_check(const eegclip_cstack_bwd_desc* d) {
    if (!d || d->B < 1 || d->H < 1 || d->H > CS_MAXH) return EEGCLIP_EINVAL;
    if (!d->x || !d->w25 || !d->bias1 || !d->mean1 || !d->rstd1 || !d->gamma1 || !d->beta1 || !d->packed_t || !d->dy2) return EEGCLIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d->packed_t) & 15u)) return EEGCLIP_EALIGN;
    return 0;
}
static cs_bwd_args csb_args(const eegclip_cstack_bwd_desc* d) {
    return cs_bwd_args{d->x, d->xs_b, d->xs_h, d->w25, d->bias1, d->mean1, d->rstd1, d->gamma1, d->beta1, static_cast<const unsigned char*>(d->packed_t), d->dy2,
                       d->rows_out, d->stat, d->nstat, d->count, d->stat_local, d->nstat_local, d->dgamma, d->dbeta, d->dx, d->dw_partials, d->B, d->H,
                       (csb_vec2(d->x, d->xs_b, d->xs_h) && (!d->dx || csb_vec2(d->dx, d->xs_b, d->xs_h))) ? 1 : 0};
}

extern "C" int eegclip_cstack_bwd_stats(const eegclip_cstack_bwd_desc* d, void* stream) {
    if (int rc = csb_check(d)) return rc;
    if (!d->rows_out) return EEGCLIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(d->rows_out) & 7u) return EEGCLIP_EALIGN;
    const size_t lds = (size_t)d->H * CS_RS * 4 + CSB_NCOEF * 48 * 4 + CSB_DYF + CS_NW * 1024;
    EEG_LAUNCH(cstack_bwd_kernel<false>, dim3(d->B), dim3(CS_NT), lds, stream, csb_args(d));
    return (int)hipGetLastError();
}

extern "C" long long eegclip_cstack_bwd_workspace_floats(int B) { return B < 1 ? 0 : (long long)B * CS_C * CS_K1; }

extern "C" int eegclip_cstack_bwd_apply(const eegclip_cstack_bwd_desc* d, void* stream) {
    if (int rc = csb_check(d)) return rc;
    if (!d->stat || d->nstat < 1 || d->count < 1.0 || !d->dgamma || !d->dbeta || !d->dx || !d->dw_partials) return EEGCLIP_EINVAL;
    if (d->stat_local && d->nstat_local < 1) return EEGCLIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d->stat) | reinterpret_cast<uintptr_t>(d->stat_local)) & 7u) return EEGCLIP_EALIGN;
    const size_t lds = (size_t)d->H * CS_RS * 4 + CSB_NCOEF * 48 * 4 + CSB_DYF + CSB_TAPF + CS_NW * CSB_WAVE;
    EEG_LAUNCH(cstack_bwd_kernel<true>, dim3(d->B), dim3(CS_NT), lds, stream, csb_args(d));
    if (!d->dw25) return (int)hipGetLastError();                 // the caller sums the tap-gradient rows itself (eegclip_cstack_bwd_taps_reduce, any stream)
    const int n = CS_C * CS_K1;
    EEG_LAUNCH(cstack_rows_reduce_kernel, dim3((n + 15) / 16), dim3(256), 256 * sizeof(float), stream, (const float*)d->dw_partials, d->B, n, d->dw25);
    return (int)hipGetLastError();
}

// dw25 (+)= the B partial rows eegclip_cstack_bwd_apply left in `dw_partials` (a launch of its own so that it can leave the dX chain: only the optimizer reads it)
extern "C" int eegclip_cstack_bwd_taps_reduce(const float* dw_partials, int B, float* dw25, void* stream) {
    if (!dw_partials || !dw25 || B < 1) return EEGCLIP_EINVAL;
    const int n = CS_C * CS_K1;
    EEG_LAUNCH(cstack_rows_reduce_kernel, dim3((n + 15) / 16), dim3(256), 256 * sizeof(float), stream, dw_partials, B, n, dw25);
    return (int)hipGetLastError();
}

extern "C" long long eegclip_cstack_bwd_w2_workspace_floats(int B, int H) { return (B < 1 || H < 1 || H > CS_MAXH) ? 0 : (long long)csw_groups(B) * H * CS_C * CS_C; }

extern "C" int eegclip_cstack_bwd_w2(const float* x, long long xs_b, long long xs_h, const float* w25, const float* bias1, const float* mean1, const float* rstd1,
                                     const float* gamma1, const float* beta1, const float* dy2, float* dWs, float* workspace, int B, int H, void* stream) {
    if (!x || !w25 || !bias1 || !mean1 || !rstd1 || !gamma1 || !beta1 || !dy2 || !dWs || !workspace || B < 1 || H < 1 || H > CS_MAXH) return EEGCLIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(dy2) & 15u) return EEGCLIP_EALIGN;
    const int G = csw_groups(B);
    const size_t lds = (size_t)CSW_NW * (CS_RS + 256) * 4 + (size_t)CSW_NW * CS_C * CS_C * 4;
    EEG_LAUNCH(cstack_bwd_w2_kernel, dim3((H + 1) / 2, G), dim3(64 * CSW_NW), lds, stream, x, xs_b, xs_h, w25, bias1, mean1, rstd1, gamma1, beta1, dy2, workspace, B,
               H, G, csb_vec2(x, xs_b, xs_h));
    EEG_LAUNCH(cstack_w2_reduce_kernel, dim3(CS_C * CS_C / 32, (H + 31) / 32), dim3(256), 32 * 33 * sizeof(float), stream, (const float*)workspace, G, H, dWs);
    return (int)hipGetLastError();
}
