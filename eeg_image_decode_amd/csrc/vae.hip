// The SDXL VAE's arithmetic (low-level reconstruction path: Generation/custom_pipeline_low_level.py:8-31 `vae.encode` of the start image,
// Generation/custom_pipeline.py:421 `vae.decode` of the final latents; the module is diffusers' AutoencoderKL -- 3 x 3 convolutions, GroupNorm(32) + SiLU,
// nearest 2 x upsampling / stride-2 downsampling, one single-head self-attention in the middle) on 16-bit activations, fp32 accumulation.
//
// Layout: activations are PADDED NHWC -- [image][H + 2][W + 2][C], 16-bit, the one-pixel border is zero and never written -- so a 3 x 3 tap is a
// constant pixel offset with no boundary case, and a pixel's channels are one contiguous run: the k-slices of the implicit GEMM
//     out[p][co] = sum_{tap, ci} in[p + off(tap)][ci] W[co][tap][ci]          (M = pixels, N = Cout, K = 9 Cin)
// go global -> LDS by LDS-DMA exactly like the rows of a plain GEMM (csrc/gemm16.hip, whose tile loop this is: 128 pixels x 128 output channels per
// 256-thread workgroup, 64-k tiles, 4 stages with counted vmcnt, XOR-swizzled chunks, v_mfma_f32_32x32x16_{f16,bf16}); only the row ADDRESS differs -- it
// is recomputed per tap from the output pixel (stride 1 or 2; `up`: the source is the nearest-2x upsampled image, i.e. pixel >> 1 -- the Upsample2D +
// conv pair of the decoder never materialises the upsampled tensor).  Epilogue: + bias, + residual (a tensor shaped like the output).
// The few layers with 3 / 4 / 8 channels on one side (conv_in, conv_out, quant convs) run on a direct fp32-accumulate kernel, weights in LDS.
// GroupNorm: one statistics pass (fp64 partial sums per (image, group), atomics onto 64 addresses per image) and one apply pass (+ SiLU) that writes
// the next convolution's padded operand.
#include "eeg_common.h"

#include <stdlib.h>

namespace eeg {

// LDS stages of the implicit-GEMM conv: 2 (64 KB: two workgroups per CU, the other's MFMAs cover this one's barrier and DMA waits) measured 19.9-20.0 ms per
// 1024 x 1024 decode against 20.8-20.9 with 3 or 4 stages (one workgroup per CU), three builds alternated on one box
#ifndef CV_NS_BUILD
#define CV_NS_BUILD 2
#endif
constexpr int CV_T = 128, CV_K = 64, CV_NS = CV_NS_BUILD;
constexpr int CV_ROWB = 2 * CV_K;
constexpr int CV_TILE_B = CV_T * CV_ROWB;
constexpr int CV_STAGE_B = 2 * CV_TILE_B;

typedef _Float16 cv_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short cv_u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short cv_u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 cv_mfma_f16(bf16x8 a, bf16x8 b, f32x16 c) {
#if defined(EEG_EMU)
    struct AB { bf16x8 a, b; } in{a, b};
    auto all = hipemu::wave_allgather(&in, sizeof(in));
    const int l = hipemu::cur->lane, col = l & 31, hb = 4 * (l >> 5);
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + hb;
        float acc = c[r];
        for (int h = 0; h < 2; ++h) {
            AB ra, rbv;
            memcpy(&ra, all[row + 32 * h], sizeof(AB));
            memcpy(&rbv, all[col + 32 * h], sizeof(AB));
            for (int e = 0; e < 8; ++e) {
                _Float16 x, y;
                short sx = ra.a[e], sy = rbv.b[e];
                memcpy(&x, &sx, 2);
                memcpy(&y, &sy, 2);
                acc += (float)x * (float)y;
            }
        }
        d[r] = acc;
    }
    return d;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cv_f16x8, a), __builtin_bit_cast(cv_f16x8, b), c, 0, 0, 0);
#endif
}
template <bool F16>
__device__ __forceinline__ float cv_to_f32(unsigned short u) {
    if (F16) {
        _Float16 h;
        memcpy(&h, &u, 2);
        return (float)h;
    }
    return bf16_bits_to_f32(u);
}
template <bool F16>
__device__ __forceinline__ unsigned short cv_from_f32(float v) {
    if (F16) {
        const _Float16 h = (_Float16)v;
        unsigned short u;
        memcpy(&u, &h, 2);
        return u;
    }
    return f32_to_bf16_bits(v);
}

struct cv_args {
    const unsigned short* in;          // padded NHWC input, pixel (0, 0) of the padded image 0
    const unsigned short* W;           // [Cout][KS * KS][Cin]
    unsigned short* out;               // output image 0, pixel (0, 0) of ITS (optionally padded) frame
    const unsigned short* bias;        // [Cout] or null
    const unsigned short* R;           // residual in the output's layout, or null
    int N, Ho, Wo;                     // images, output height / width
    int Hp, Wp, Cin;                   // input frame (padded) and channels
    int Hop, Wop, opad, Cout;          // output frame, its padding (0 or 1), channels
    int KS, stride, oy, ox, up;        // source pixel of output (y, x), tap (ky, kx): (y * stride + ky + oy, x * stride + kx + ox), or with `up`
                                       // (((y + ky - 1) >> 1) + 1, ((x + kx - 1) >> 1) + 1) in the padded input frame
    int M, tiles_n, ntiles, chunk;
};

// SPEC (round 6): four extra PRODUCER waves issue every LDS-DMA instruction (and do the address arithmetic in front of it); the four MFMA waves only read
// fragments and feed the matrix pipe, and meet the producers at the one barrier per k-tile -- a wave that does both is in order and stalls in every DMA issue
// with its MFMAs unissued behind it (csrc/infonce_fused.hip, NPRD).
template <bool F16, bool SPEC>
__global__ __launch_bounds__(SPEC ? 512 : 256) void conv16_kernel(const cv_args a) {
    EEG_LDS_BASE(unsigned char, lds);
    const int logical = (int)(blockIdx.x & 7) * a.chunk + (int)(blockIdx.x >> 3);
    if (logical >= a.ntiles) return;
    const int m0 = (logical / a.tiles_n) * CV_T, n0 = (logical % a.tiles_n) * CV_T;
    const int t = threadIdx.x, lane = t & 63, wave_ = wave_uniform(t >> 6);
    const bool producer = SPEC && wave_ >= 4;                 // (wave-uniform)
    const int wave = wave_ & 3;                               // index among the MFMA waves / among the DMA waves
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;
    auto swz = [](int row) { return (row >> 1) & 7; };
    // DMA roles (csrc/gemm16.hip): DMA wave w deposits rows 32 w .. 32 w + 31 of both tiles, 8 rows (1 KB) per instruction
    const int drow = lane >> 3, dpos = lane & 7;
    int pn[4], py[4], px[4], acol[4], pbase[4];
    const unsigned short* wsrc[4];
    const int hw = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 32 * wave + 8 * i + drow;
        const int col = 8 * (dpos ^ swz(row));
        const int m = m0 + row < a.M ? m0 + row : a.M - 1;          // rows beyond M: a clamped copy that nobody stores
        pn[i] = m / hw;
        const int rem = m - pn[i] * hw;
        py[i] = rem / a.Wo;
        px[i] = rem - py[i] * a.Wo;
        acol[i] = col;
        wsrc[i] = a.W + (long long)(n0 + row) * a.KS * a.KS * a.Cin + col;
        // pixel index of tap (0, 0) in the padded input frame (plain form) / of the frame row 0 of this image (`up`: the source row depends on the tap's parity)
        pbase[i] = a.up ? pn[i] * a.Hp : (pn[i] * a.Hp + py[i] * a.stride + a.oy) * a.Wp + px[i] * a.stride + a.ox;
    }
    const int cpt = a.Cin / CV_K;                             // k-tiles per tap
    // (tap, channel offset) of the NEXT k-tile to be issued, stepped once per tile: k-tile kt is tap kt / cpt, channels (kt % cpt) CV_K .. -- the first version
    // took those two integer divisions (and the tap's / KS) inside every one of the 8 DMA instructions of a tile: ~65 instructions of address arithmetic per
    // instruction, ~520 per k-tile against the 16 MFMAs they sit between (240 TFLOP/s for the decode); the weights' offset is simply kt CV_K
    int t_ky = 0, t_kx = 0, t_c0 = 0;
    auto advance = [&]() {
        t_c0 += CV_K;
        if (t_c0 == a.Cin) {
            t_c0 = 0;
            if (++t_kx == a.KS) { t_kx = 0; ++t_ky; }
        }
    };
    auto issue_one = [&](int kt, int dnum) {                  // DMA instruction `dnum` of k-tile kt == the tile (t_ky, t_kx, t_c0) describes
        const int o = dnum >> 2, i = dnum & 3;
        unsigned char* st = lds + (kt % CV_NS) * CV_STAGE_B + 32 * wave * CV_ROWB;
        if (o == 0) {
            int pix;
            if (a.up) {
                const int sy = ((py[i] + t_ky - 1) >> 1) + 1, sx = ((px[i] + t_kx - 1) >> 1) + 1;
                pix = (pbase[i] + sy) * a.Wp + sx;
            } else {
                pix = pbase[i] + t_ky * a.Wp + t_kx;
            }
            lds_dma16(st + 8 * i * CV_ROWB, a.in + (long long)pix * a.Cin + t_c0 + acol[i]);
        } else {
            lds_dma16(st + CV_TILE_B + 8 * i * CV_ROWB, wsrc[i] + (long long)kt * CV_K);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
    int fom[4][2], fon[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rm = wm * 64 + 32 * i + r32, rn = wn * 64 + 32 * i + r32;
            fom[s][i] = rm * CV_ROWB + (((2 * s + h) ^ swz(rm)) & 7) * 16;
            fon[s][i] = CV_TILE_B + rn * CV_ROWB + (((2 * s + h) ^ swz(rn)) & 7) * 16;
        }
    const int ktiles = a.KS * a.KS * cpt;
    if (!SPEC || producer) {
#pragma unroll
        for (int p = 0; p < CV_NS - 1; ++p)
            if (p < ktiles) {
#pragma unroll
                for (int dnum = 0; dnum < 8; ++dnum) issue_one(p, dnum);
                advance();
            }
    }
    if (producer) {
        for (int kt = 0; kt < ktiles; ++kt) {
            const int newer = ktiles - 1 - kt < CV_NS - 2 ? ktiles - 1 - kt : CV_NS - 2;
            if (newer >= 2) wait_vmcnt<16>();
            else if (newer == 1) wait_vmcnt<8>();
            else wait_vmcnt<0>();
            raw_barrier();                                    // the one meeting point of the two kinds of waves per k-tile
            if (kt + CV_NS - 1 < ktiles) {
#pragma unroll
                for (int dnum = 0; dnum < 8; ++dnum) issue_one(kt + CV_NS - 1, dnum);
                advance();
            }
        }
        return;                                               // (the epilogue has no barrier)
    }
    for (int kt = 0; kt < ktiles; ++kt) {
        if (!SPEC) {
            const int newer = ktiles - 1 - kt < CV_NS - 2 ? ktiles - 1 - kt : CV_NS - 2;
            if (newer >= 2) wait_vmcnt<16>();
            else if (newer == 1) wait_vmcnt<8>();
            else wait_vmcnt<0>();
        }
        raw_barrier();
        const bool refill = !SPEC && kt + CV_NS - 1 < ktiles;
        const unsigned char* st = lds + (kt % CV_NS) * CV_STAGE_B;
        bf16x8 am[2][2], wf[2][2];
        // (what a step's first MFMA takes is read last and the next step's reads go out behind that MFMA: csrc/infonce_fused.hip, round 6)
        auto read_step = [&](int s, int set) {
#pragma unroll
            for (int i = 1; i >= 0; --i) {
                am[set][i] = *reinterpret_cast<const bf16x8*>(st + fom[s][i]);
                wf[set][i] = *reinterpret_cast<const bf16x8*>(st + fon[s][i]);
            }
        };
        read_step(0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#if !defined(EEG_EMU)
            __builtin_amdgcn_sched_barrier(0);
#endif
            const int set = s & 1;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    acc[j][i] = F16 ? cv_mfma_f16(wf[set][j], am[set][i], acc[j][i]) : mfma_bf16_32x32x16(wf[set][j], am[set][i], acc[j][i]);
                    const int mi = 4 * s + 2 * j + i;
                    if (j == 0 && i == 0 && s + 1 < 4) {
#if !defined(EEG_EMU)
                        __builtin_amdgcn_sched_barrier(0);
#endif
                        read_step(s + 1, (s + 1) & 1);
#if !defined(EEG_EMU)
                        __builtin_amdgcn_sched_barrier(0);
#endif
                    }
                    if (refill && (mi & 1)) issue_one(kt + CV_NS - 1, mi >> 1);
                }
#if !defined(EEG_EMU)
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        if (refill) advance();
    }
    // ---- epilogue: lane (r32, h) owns pixel m = m0 + 64 wm + 32 i + r32; registers 4 eq .. 4 eq + 3 of n tile j are channels n0 + 64 wn + 32 j + 8 eq + 4 h ..
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 64 + 32 * i + r32;
        if (m >= a.M) continue;
        const int n_ = m / hw, rem = m - n_ * hw, y = rem / a.Wo, x = rem - y * a.Wo;
        const long long opix = (((long long)n_ * a.Hop + y + a.opad) * a.Wop + x + a.opad) * a.Cout;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int eq = 0; eq < 4; ++eq) {
                const int n = n0 + wn * 64 + 32 * j + 8 * eq + 4 * h;
                cv_u16x4 bv = cv_u16x4{0, 0, 0, 0}, rv = cv_u16x4{0, 0, 0, 0};
                if (a.bias) bv = *reinterpret_cast<const cv_u16x4*>(a.bias + n);
                if (a.R) rv = *reinterpret_cast<const cv_u16x4*>(a.R + opix + n);
                cv_u16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[j][i][4 * eq + e];
                    if (a.bias) v += cv_to_f32<F16>(bv[e]);
                    if (a.R) v += cv_to_f32<F16>(rv[e]);
                    o[e] = cv_from_f32<F16>(v);
                }
                *reinterpret_cast<cv_u16x4*>(a.out + opix + n) = o;
            }
    }
}

// ---- direct convolution for the layers with 3 / 4 / 8 channels on one side: thread = (pixel, output channel), weights (16-bit) in LDS, fp32 accumulate
template <bool F16>
__global__ __launch_bounds__(256) void conv_small16_kernel(const cv_args a) {
    EEG_LDS_BASE(unsigned short, wl);                        // [Cout][KS*KS][Cin]
    const int kk = a.KS * a.KS, wn = a.Cout * kk * a.Cin;
    for (int i = threadIdx.x; i < wn; i += 256) wl[i] = a.W[i];
    __syncthreads();
    const long long total = (long long)a.M * a.Cout;
    const int hw = a.Ho * a.Wo;
    const bool vec8 = (a.Cin & 7) == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15u) == 0;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += 256LL * gridDim.x) {
        const int m = (int)(q / a.Cout), co = (int)(q - (long long)m * a.Cout);
        const int n_ = m / hw, rem = m - n_ * hw, y = rem / a.Wo, x = rem - y * a.Wo;
        float acc = a.bias ? cv_to_f32<F16>(a.bias[co]) : 0.f;
        for (int ky = 0; ky < a.KS; ++ky)
            for (int kx = 0; kx < a.KS; ++kx) {
                const int sy = a.up ? (((y + ky - 1) >> 1) + 1) : y * a.stride + ky + a.oy;
                const int sx = a.up ? (((x + kx - 1) >> 1) + 1) : x * a.stride + kx + a.ox;
                const unsigned short* p = a.in + (((long long)n_ * a.Hp + sy) * a.Wp + sx) * a.Cin;
                const unsigned short* w = wl + (co * kk + ky * a.KS + kx) * a.Cin;
                if (vec8) {                                  // 16-byte reads of both operands (conv_out, 128 -> 3 at 1024 x 1024: 6.1 -> ms with 2-byte reads); same summation order
                    for (int ci = 0; ci < a.Cin; ci += 8) {
                        const cv_u16x8 pv = *reinterpret_cast<const cv_u16x8*>(p + ci), wv = *reinterpret_cast<const cv_u16x8*>(w + ci);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc += cv_to_f32<F16>(pv[e]) * cv_to_f32<F16>(wv[e]);
                    }
                } else {
                    for (int ci = 0; ci < a.Cin; ++ci) acc += cv_to_f32<F16>(p[ci]) * cv_to_f32<F16>(w[ci]);
                }
            }
        const long long opix = (((long long)n_ * a.Hop + y + a.opad) * a.Wop + x + a.opad) * a.Cout;
        if (a.R) acc += cv_to_f32<F16>(a.R[opix + co]);
        a.out[opix + co] = cv_from_f32<F16>(acc);
    }
}

// ---- GroupNorm over a padded NHWC tensor (interior pixels only).  Statistics: workgroup = (image, block of GN_PB pixels); thread = (8 consecutive channels, pixel lane):
// 16-byte reads, the pixels of a lane independent of one another (loads in flight), the two 4-channel halves summed apart (a group has >= 4 channels); the sums meet
// per group in LDS, one fp64 atomic pair per (workgroup, group).  (The first version gave a thread one CHANNEL and walked its 64 pixels one 2-byte read after the
// other, an integer division each: 7.5 ms of a 35 ms decode for ~3 GB of reads.)  Channel counts the vector form does not take (C % 8, 256 % (C / 8)) walk as before.
constexpr int GN_PB = 256;
template <bool F16>
__global__ __launch_bounds__(256) void gn_stats16_kernel(const unsigned short* __restrict__ x, int H, int W, int C, int pad, int groups, double* __restrict__ sums) {
    EEG_LDS_BASE(float, red);                                // [2][groups]
    const int n_ = blockIdx.y, t = threadIdx.x, hw = H * W, Wp = W + 2 * pad, Hp = H + 2 * pad;
    for (int i = t; i < 2 * groups; i += 256) red[i] = 0.f;
    __syncthreads();
    const int p0 = blockIdx.x * GN_PB, p1 = p0 + GN_PB < hw ? p0 + GN_PB : hw;
    const int cpg = C / groups, c8n = C / 8;
    if ((C & 7) == 0 && c8n <= 256 && 256 % c8n == 0 && (groups & 1) == 0 && groups <= 256 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0) {
        const int cc = t % c8n, pl = t / c8n, ppi = 256 / c8n;
        float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll 4
        for (int p = p0 + pl; p < p1; p += ppi) {
            const int y = p / W, xx = p - y * W;
            const cv_u16x8 v = *reinterpret_cast<const cv_u16x8*>(x + (((long long)n_ * Hp + y + pad) * Wp + xx + pad) * C + 8 * cc);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = cv_to_f32<F16>(v[e]), b = cv_to_f32<F16>(v[4 + e]);
                s0 += a; q0 += a * a;
                s1 += b; q1 += b * b;
            }
        }
        // fixed-order combine (two decodes of one latent are bit-identical: tests/test_vae_gpu.py): every thread leaves its two half-chunk pairs in LDS, thread g
        // adds group g's entries -- half-chunks g cpg / 4 .. of every pixel lane -- in index order
        float* part = red + 2 * groups;                      // [256][4]
        *reinterpret_cast<f32x4*>(part + 4 * t) = f32x4{s0, q0, s1, q1};
        __syncthreads();
        if (t < groups) {
            const int hpg = cpg / 4;                         // 4-channel half-chunks per group
            float s_ = 0.f, q = 0.f;
            for (int l = 0; l < ppi; ++l)
                for (int k = 0; k < hpg; ++k) {
                    const int hc = t * hpg + k;              // half-chunk index along the channels: thread (l, hc >> 1), half hc & 1
                    const float* e = part + 4 * (l * c8n + (hc >> 1)) + 2 * (hc & 1);
                    s_ += e[0];
                    q += e[1];
                }
            red[t] = s_;
            red[groups + t] = q;
        }
    } else {
        for (int c = t; c < C; c += 256) {
            float s_ = 0.f, q = 0.f;
            for (int p = p0; p < p1; ++p) {
                const int y = p / W, xx = p - y * W;
                const float v = cv_to_f32<F16>(x[(((long long)n_ * Hp + y + pad) * Wp + xx + pad) * C + c]);
                s_ += v;
                q += v * v;
            }
            atomicAdd(red + c / cpg, s_);
            atomicAdd(red + groups + c / cpg, q);
        }
    }
    __syncthreads();
    for (int g = t; g < groups; g += 256) {
        atomicAdd(sums + ((long long)n_ * groups + g) * 2, (double)red[g]);
        atomicAdd(sums + ((long long)n_ * groups + g) * 2 + 1, (double)red[groups + g]);
    }
}
// apply: y = (x - mean) * rstd * gamma + beta (then SiLU if `silu`), written into another frame (its own padding); 4 consecutive channels per thread.  Every workgroup
// first turns the (image, group) sums into a (mean, 1 / sqrt(var + eps)) table in LDS (the first version redid the fp64 division and square root per 4 channels)
template <bool F16>
__global__ __launch_bounds__(256) void gn_apply16_kernel(const unsigned short* __restrict__ x, int N, int H, int W, int C, int pad, int groups,
                                                          const double* __restrict__ sums, const unsigned short* __restrict__ gamma,
                                                          const unsigned short* __restrict__ beta, float eps, int silu_on, unsigned short* __restrict__ y, int opad) {
    EEG_LDS_BASE(float, tab);                                // [N * groups][2]
    const int c4n = C / 4, hw = H * W, Wp = W + 2 * pad, Hp = H + 2 * pad, Wop = W + 2 * opad, Hop = H + 2 * opad, cpg = C / groups;
    const long long total = (long long)N * hw * c4n;
    const double cnt = (double)hw * cpg;
    for (int i = threadIdx.x; i < N * groups; i += 256) {
        const double mu = sums[2 * (long long)i] / cnt;
        double var = sums[2 * (long long)i + 1] / cnt - mu * mu;
        if (var < 0.0) var = 0.0;
        tab[2 * i] = (float)mu;
        tab[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += 256LL * gridDim.x) {
        const int c = 4 * (int)(q % c4n);
        const long long pq = q / c4n;
        const int n_ = (int)(pq / hw), p = (int)(pq - (long long)n_ * hw), yy = p / W, xx = p - yy * W;
        const int g = c / cpg;                               // (cpg % 4 == 0: the 4 channels share a group)
        const float mean = tab[2 * (n_ * groups + g)], rstd = tab[2 * (n_ * groups + g) + 1];
        const cv_u16x4 xv = *reinterpret_cast<const cv_u16x4*>(x + (((long long)n_ * Hp + yy + pad) * Wp + xx + pad) * C + c);
        const cv_u16x4 gv = *reinterpret_cast<const cv_u16x4*>(gamma + c), bv = *reinterpret_cast<const cv_u16x4*>(beta + c);
        cv_u16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = (cv_to_f32<F16>(xv[e]) - mean) * rstd * cv_to_f32<F16>(gv[e]) + cv_to_f32<F16>(bv[e]);
            if (silu_on) v = silu(v);
            o[e] = cv_from_f32<F16>(v);
        }
        *reinterpret_cast<cv_u16x4*>(y + (((long long)n_ * Hop + yy + opad) * Wop + xx + opad) * C + c) = o;
    }
}

// ---- softmax over the rows of a 16-bit matrix, in place, logits scaled by `scale` first (the mid-block attention's (HW, HW) score matrix): one workgroup
// per row, fp32 arithmetic
template <bool F16>
__global__ __launch_bounds__(256) void softmax_rows16_kernel(unsigned short* __restrict__ s, int cols, long long ld, float scale) {
    EEG_LDS_BASE(float, red);                                // [8]
    unsigned short* row = s + (long long)blockIdx.x * ld;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    float mx = -3.0e38f;
    for (int c = t; c < cols; c += 256) mx = fmaxf(mx, scale * cv_to_f32<F16>(row[c]));
    mx = wave_max(mx);
    if (lane == 0) red[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int c = t; c < cols; c += 256) sum += expf(scale * cv_to_f32<F16>(row[c]) - mx);
    sum = wave_sum(sum);
    if (lane == 0) red[4 + w] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
    for (int c = t; c < cols; c += 256) row[c] = cv_from_f32<F16>(expf(scale * cv_to_f32<F16>(row[c]) - mx) * inv);
}

// ---- DiagonalGaussianDistribution.sample of the encoder's moments (padded-free NHWC (N, H, W, 2 L): mean | logvar): z = mean + exp(0.5 clamp(logvar)) noise
template <bool F16>
__global__ __launch_bounds__(256) void vae_sample16_kernel(const unsigned short* __restrict__ mom, const unsigned short* __restrict__ noise,
                                                            unsigned short* __restrict__ z, long long pixels, int L) {
    const long long total = pixels * L;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += 256LL * gridDim.x) {
        const long long p = q / L;
        const int c = (int)(q - p * L);
        const float mean = cv_to_f32<F16>(mom[p * 2 * L + c]);
        float lv = cv_to_f32<F16>(mom[p * 2 * L + L + c]);
        lv = fminf(fmaxf(lv, -30.f), 20.f);
        const float nz = noise ? cv_to_f32<F16>(noise[q]) : 0.f;
        z[q] = cv_from_f32<F16>(mean + expf(0.5f * lv) * nz);
    }
}

}  // namespace eeg

using namespace eeg;

static int cv_check(const eegclip_conv16_desc* d) {
    if (!d || !d->in || !d->W || !d->out || d->N < 1 || d->Ho < 1 || d->Wo < 1 || d->Cin < 1 || d->Cout < 1 || (d->KS != 1 && d->KS != 3) ||
        (d->stride != 1 && d->stride != 2) || d->in_pad < 0 || d->in_pad > 1 || d->out_pad < 0 || d->out_pad > 1 || d->Hi < 1 || d->Wi < 1 ||
        (d->dtype != EEGCLIP_DT_BF16 && d->dtype != EEGCLIP_DT_F16))
        return EEGCLIP_EINVAL;
    if (d->upsample && (d->KS != 3 || d->stride != 1 || d->in_pad != 1 || d->Ho != 2 * d->Hi || d->Wo != 2 * d->Wi)) return EEGCLIP_EINVAL;
    // every tap of every output pixel must land inside the padded input frame
    const int pt = d->pad_top, pl = d->pad_left;
    if (pt < 0 || pl < 0 || pt > d->in_pad || pl > d->in_pad) return EEGCLIP_EINVAL;
    if (!d->upsample) {
        const int ymax = (d->Ho - 1) * d->stride + d->KS - 1 - pt, xmax = (d->Wo - 1) * d->stride + d->KS - 1 - pl;
        if (ymax > d->Hi - 1 + d->in_pad || xmax > d->Wi - 1 + d->in_pad) return EEGCLIP_EINVAL;
    }
    return 0;
}

extern "C" int eegclip_conv16(const eegclip_conv16_desc* d, void* stream) {
    if (const int rc = cv_check(d)) return rc;
    cv_args a;
    a.in = static_cast<const unsigned short*>(d->in);
    a.W = static_cast<const unsigned short*>(d->W);
    a.out = static_cast<unsigned short*>(d->out);
    a.bias = static_cast<const unsigned short*>(d->bias);
    a.R = static_cast<const unsigned short*>(d->residual);
    a.N = d->N; a.Ho = d->Ho; a.Wo = d->Wo;
    a.Hp = d->Hi + 2 * d->in_pad; a.Wp = d->Wi + 2 * d->in_pad; a.Cin = d->Cin;
    a.Hop = d->Ho + 2 * d->out_pad; a.Wop = d->Wo + 2 * d->out_pad; a.opad = d->out_pad; a.Cout = d->Cout;
    a.KS = d->KS; a.stride = d->stride; a.oy = d->in_pad - d->pad_top; a.ox = d->in_pad - d->pad_left; a.up = d->upsample ? 1 : 0;
    const long long M = (long long)d->N * d->Ho * d->Wo;
    if (M > 0x7fffffffLL) return EEGCLIP_EINVAL;
    a.M = (int)M;
    const bool f16 = d->dtype == EEGCLIP_DT_F16;
    const bool mfma = d->Cin % CV_K == 0 && d->Cout % CV_T == 0;
    if (mfma) {
        if ((reinterpret_cast<uintptr_t>(d->in) | reinterpret_cast<uintptr_t>(d->W)) & 15u) return EEGCLIP_EALIGN;
        if ((reinterpret_cast<uintptr_t>(d->out) | reinterpret_cast<uintptr_t>(d->bias) | reinterpret_cast<uintptr_t>(d->residual)) & 7u) return EEGCLIP_EALIGN;
        a.tiles_n = d->Cout / CV_T;
        a.ntiles = a.tiles_n * ((a.M + CV_T - 1) / CV_T);
        a.chunk = (a.ntiles + 7) / 8;
        const size_t lds = (size_t)CV_NS * CV_STAGE_B;
        static const bool spec = [] { const char* e = getenv("EEGCLIP_CONV16_PRODUCERS"); return !(e && e[0] == '0'); }();      // (A/B aid)
        if (spec) {
            if (f16) EEG_LAUNCH((conv16_kernel<true, true>), dim3((unsigned)(8 * a.chunk)), dim3(512), lds, stream, a);
            else     EEG_LAUNCH((conv16_kernel<false, true>), dim3((unsigned)(8 * a.chunk)), dim3(512), lds, stream, a);
        } else {
            if (f16) EEG_LAUNCH((conv16_kernel<true, false>), dim3((unsigned)(8 * a.chunk)), dim3(256), lds, stream, a);
            else     EEG_LAUNCH((conv16_kernel<false, false>), dim3((unsigned)(8 * a.chunk)), dim3(256), lds, stream, a);
        }
        return (int)hipGetLastError();
    }
    const size_t wbytes = (size_t)d->Cout * d->KS * d->KS * d->Cin * 2;
    if (wbytes > 150 * 1024) return EEGCLIP_EINVAL;           // (the small-channel layers of the VAE: at most 512 -> 8, 72 KB)
    a.tiles_n = a.ntiles = a.chunk = 0;
    long long g = (M * d->Cout + 255) / 256;
    if (g > 16384) g = 16384;
    if (f16) EEG_LAUNCH((conv_small16_kernel<true>), dim3((unsigned)g), dim3(256), wbytes, stream, a);
    else     EEG_LAUNCH((conv_small16_kernel<false>), dim3((unsigned)g), dim3(256), wbytes, stream, a);
    return (int)hipGetLastError();
}

extern "C" int eegclip_groupnorm16(const void* x, int N, int H, int W, int C, int pad, int groups, const void* gamma, const void* beta, float eps, int silu_on,
                                   void* y, int out_pad, double* sums, int dtype, void* stream) {
    if (!x || !gamma || !beta || !y || !sums || N < 1 || H < 1 || W < 1 || C < 4 || groups < 1 || C % groups || (C / groups) % 4 || pad < 0 || pad > 1 ||
        out_pad < 0 || out_pad > 1 || (dtype != EEGCLIP_DT_BF16 && dtype != EEGCLIP_DT_F16))
        return EEGCLIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(y) |
         reinterpret_cast<uintptr_t>(sums)) & 7u)
        return EEGCLIP_EALIGN;
    const bool f16 = dtype == EEGCLIP_DT_F16;
    if ((long long)N * groups > 4096) return EEGCLIP_EINVAL;     // (the apply kernel keeps the (mean, rstd) table of every (image, group) in LDS)
    hipError_t e = hipMemsetAsync(sums, 0, (size_t)N * groups * 2 * sizeof(double), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    const dim3 gs((unsigned)((H * W + GN_PB - 1) / GN_PB), (unsigned)N);
    const size_t lds = (2 * (size_t)groups + 256 * 4) * sizeof(float);
    if (f16) EEG_LAUNCH((gn_stats16_kernel<true>), gs, dim3(256), lds, stream, static_cast<const unsigned short*>(x), H, W, C, pad, groups, sums);
    else     EEG_LAUNCH((gn_stats16_kernel<false>), gs, dim3(256), lds, stream, static_cast<const unsigned short*>(x), H, W, C, pad, groups, sums);
    long long g = ((long long)N * H * W * (C / 4) + 255) / 256;
    if (g > 16384) g = 16384;
#define EEG_GNA(F)                                                                                                                                        \
    EEG_LAUNCH((gn_apply16_kernel<F>), dim3((unsigned)g), dim3(256), (size_t)N * groups * 2 * sizeof(float), stream, static_cast<const unsigned short*>(x), N, H, W, C, pad, groups, sums,      \
               static_cast<const unsigned short*>(gamma), static_cast<const unsigned short*>(beta), eps, silu_on, static_cast<unsigned short*>(y), out_pad)
    if (f16) EEG_GNA(true);
    else     EEG_GNA(false);
#undef EEG_GNA
    return (int)hipGetLastError();
}

extern "C" int eegclip_softmax_rows16(void* s, int rows, int cols, long long ld, float scale, int dtype, void* stream) {
    if (!s || rows < 1 || cols < 1 || ld < cols || (dtype != EEGCLIP_DT_BF16 && dtype != EEGCLIP_DT_F16)) return EEGCLIP_EINVAL;
    if (dtype == EEGCLIP_DT_F16) EEG_LAUNCH((softmax_rows16_kernel<true>), dim3((unsigned)rows), dim3(256), 8 * sizeof(float), stream, static_cast<unsigned short*>(s), cols, ld, scale);
    else                         EEG_LAUNCH((softmax_rows16_kernel<false>), dim3((unsigned)rows), dim3(256), 8 * sizeof(float), stream, static_cast<unsigned short*>(s), cols, ld, scale);
    return (int)hipGetLastError();
}

extern "C" int eegclip_vae_sample16(const void* moments, const void* noise, void* z, long long pixels, int latent_channels, int dtype, void* stream) {
    if (!moments || !z || pixels < 1 || latent_channels < 1 || (dtype != EEGCLIP_DT_BF16 && dtype != EEGCLIP_DT_F16)) return EEGCLIP_EINVAL;
    long long g = (pixels * latent_channels + 255) / 256;
    if (g > 4096) g = 4096;
    if (dtype == EEGCLIP_DT_F16) EEG_LAUNCH((vae_sample16_kernel<true>), dim3((unsigned)g), dim3(256), 0, stream, static_cast<const unsigned short*>(moments), static_cast<const unsigned short*>(noise), static_cast<unsigned short*>(z), pixels, latent_channels);
    else                         EEG_LAUNCH((vae_sample16_kernel<false>), dim3((unsigned)g), dim3(256), 0, stream, static_cast<const unsigned short*>(moments), static_cast<const unsigned short*>(noise), static_cast<unsigned short*>(z), pixels, latent_channels);
    return (int)hipGetLastError();
}
