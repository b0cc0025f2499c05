// The conv stack of Enc_eeg recomputed from the token rows (round 5):
//     Conv2d(1,40,(1,25)) -> AvgPool2d((1,51),(1,5)) -> BatchNorm2d(40) -> ELU -> Conv2d(40,40,(63,1))      (Retrieval/ATMS_retrieval.py:102-106)
// Rounds 1-4 kept y1 = conv + pool output (B,40,63,36) fp32 in HBM (93 MB at B = 256) and made seven passes over it (and over its gradient) per
// step; each pass sat at 0.2-0.44 of the HBM roof on the 1/16-rate fp32 matrix pipe.  y1 is a 25-tap stride-5 convolution of the box-filtered
// token row -- 4.5 MFLOP per sample -- and the token rows of a sample are 63 KB: every kernel of this family re-derives the y1 tile it needs on the
// bf16 matrix cores (split-bf16 products, csrc/gemm_x3.hip arithmetic) from rows staged ONCE per workgroup in LDS, and chains the next contraction
// straight from the accumulator registers.  y1, z1 = ELU(BN(y1)), dz1 and dy1 never exist in HBM.
//
// Staging: S[h][j] = 1/51 sum_{p<51} x[h][j+p] (j < 200), one wave-level prefix sum per row; stored as ONE 32-bit word per sample,
// (bf16 hi << 16) | bf16 lo  -- the im2col fragment of the tap contraction, X[(h,w)][t] = S[h][5w + t], starts at an arbitrary word (5w), so both
// planes come out of the same eight 4-byte reads (a bf16 plane would need 2-byte-aligned 16-byte reads) and two v_perm_b32 per pair re-pack them.
//
// MFMA fragment conventions (v_mfma_f32_16x16x32_bf16, eeg_common.h): lane = 16 * kg + n; an operand fragment holds, for row / column n, the eight
// k slots 8 kg .. 8 kg + 7; the accumulator holds D[4 kg + r][n] in register r.  The k-slot <-> contraction-index assignment is free as long as
// both operands agree, which is what lets an ACCUMULATOR tile pair (rows 4 kg + r of tiles 0 and 1) be the next product's operand with no data
// movement: slot j <-> row 16 (j >> 2) + 4 kg + (j & 3).
#pragma once
#include "conv_common.h"

namespace eeg {

constexpr int CS_C = 40;       // temporal filters = channels of the spatial conv (in and out)
constexpr int CS_W = 36;       // pooled positions per row
constexpr int CS_K1 = 25;      // taps
constexpr int CS_T = 250;      // samples per token row
constexpr int CS_POOL = 51;
constexpr int CS_NS = 200;     // box-filtered samples the 36 x 25 windows touch
constexpr int CS_RS = 268;     // words per packed row (a multiple of 4): fragment reads of the padded tiles (w < 48, t < 32) reach word 5 * 47 + 31 = 266
constexpr int CS_MAXH = 64;
constexpr int CS_NW = 8;       // waves per workgroup of the sample-major kernels
constexpr int CS_NT = 64 * CS_NW;

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// (w1 & 0xffff0000) | (w0 >> 16)  and  (w1 << 16) | (w0 & 0xffff): the hi / lo bf16 halves of two packed words as one fragment dword (v_perm_b32)
__device__ __forceinline__ unsigned cs_pair_hi(unsigned w1, unsigned w0) {
#if defined(EEG_EMU)
    return (w1 & 0xffff0000u) | (w0 >> 16);
#else
    return __builtin_amdgcn_perm(w1, w0, 0x07060302u);
#endif
}
__device__ __forceinline__ unsigned cs_pair_lo(unsigned w1, unsigned w0) {
#if defined(EEG_EMU)
    return (w1 << 16) | (w0 & 0xffffu);
#else
    return __builtin_amdgcn_perm(w1, w0, 0x05040100u);
#endif
}
// fp32 -> (bf16 hi << 16) | bf16 lo,  lo = bf16(v - hi)
__device__ __forceinline__ unsigned cs_pack_word(float v) {
    const float hi = __uint_as_float(x3_pack2(v, v) & 0xffff0000u);
    return x3_pack2(v - hi, v);
}
__device__ __forceinline__ float cs_word_value(unsigned w) { return __uint_as_float(w & 0xffff0000u) + __uint_as_float(w << 16); }

__device__ __forceinline__ bf16x8 cs_frag(unsigned a, unsigned b, unsigned c, unsigned d) { return __builtin_bit_cast(bf16x8, u32x4_t{a, b, c, d}); }

// eight consecutive packed words -> the hi and lo operand fragments (slot i <-> word i)
__device__ __forceinline__ void cs_words_to_frags(const unsigned (&w)[8], bf16x8& hi, bf16x8& lo) {
    hi = cs_frag(cs_pair_hi(w[1], w[0]), cs_pair_hi(w[3], w[2]), cs_pair_hi(w[5], w[4]), cs_pair_hi(w[7], w[6]));
    lo = cs_frag(cs_pair_lo(w[1], w[0]), cs_pair_lo(w[3], w[2]), cs_pair_lo(w[5], w[4]), cs_pair_lo(w[7], w[6]));
}
// The taps a lane group holds in its eight k slots: t = cs_tap_base(kg) + i with bases 0, 16, 8, 24 -- NOT 8 kg.  The im2col reads below are 4-byte
// LDS reads at word 5 n + base + i; ds_read_b32 / ds_read2_b32 serve lanes 0-31 (kg = 0 | 1) and 32-63 (kg = 2 | 3) as one group each on 32 banks, and
// {5 n} and {5 n + d}, n < 16, share no bank only for d = 16 (mod 32): with bases 8 kg every read was a 2-way conflict (0.8 M conflict cycles per
// launch in each kernel of the family, round-5 PMC); the k-slot <-> tap assignment is free as long as the tap operand agrees, so the fix costs nothing.
__device__ __forceinline__ int cs_tap_base(int kg) { return 16 * (kg & 1) + 8 * (kg >> 1); }
// im2col fragment of row h at position tile wt: lane (n, kg) <- S[h][5 (16 wt + n) + cs_tap_base(kg) + i], i < 8
__device__ __forceinline__ void cs_sfrag(const unsigned* __restrict__ S32, int h, int wt, bf16x8& hi, bf16x8& lo) {
    const int lane = threadIdx.x & 63, n = lane & 15, kg = lane >> 4;
    const unsigned* p = S32 + h * CS_RS + 5 * (16 * wt + n) + cs_tap_base(kg);
    unsigned w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = p[i];
    cs_words_to_frags(w, hi, lo);
}

// eight fp32 values -> hi / lo fragments (slot i <-> v[i])
__device__ __forceinline__ void cs_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
    u32x2_t h0, l0, h1, l1;
    x3_split4(v[0], v[1], v[2], v[3], h0, l0);
    x3_split4(v[4], v[5], v[6], v[7], h1, l1);
    hi = cs_frag(h0[0], h0[1], h1[0], h1[1]);
    lo = cs_frag(l0[0], l0[1], l1[0], l1[1]);
}

// a += x_hi * y_lo + x_lo * y_hi + x_hi * y_hi   (the split-bf16 product, small terms first)
__device__ __forceinline__ f32x4 cs_mma3(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x4 acc) {
    acc = mfma_bf16_16x16x32(ah, bl, acc);
    acc = mfma_bf16_16x16x32(al, bh, acc);
    return mfma_bf16_16x16x32(ah, bh, acc);
}
// the same for three tiles that share one operand, PRODUCT-major: the three MFMAs that follow one another write different accumulators, so none of
// them waits for its predecessor's result (a dependent MFMA issued back to back stalls the wave for the pipe latency: SQ_WAIT_INST_ANY was 28 % of the
// wave cycles of the first version of these kernels, whose order was tile-major)
__device__ __forceinline__ void cs_mma3_a3(const bf16x8 (&ah)[3], const bf16x8 (&al)[3], bf16x8 bh, bf16x8 bl, f32x4 (&acc)[3]) {      // three A tiles, one B
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = mfma_bf16_16x16x32(ah[i], bl, acc[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = mfma_bf16_16x16x32(al[i], bh, acc[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = mfma_bf16_16x16x32(ah[i], bh, acc[i]);
}
__device__ __forceinline__ void cs_mma3_b3(bf16x8 ah, bf16x8 al, const bf16x8 (&bh)[3], const bf16x8 (&bl)[3], f32x4 (&acc)[3]) {      // one A, three B tiles
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = mfma_bf16_16x16x32(ah, bl[i], acc[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = mfma_bf16_16x16x32(al, bh[i], acc[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = mfma_bf16_16x16x32(ah, bh[i], acc[i]);
}
// the taps as the filter-side operand: lane (n, kg), tile ct <- scale[c] * w25[c = 16 ct + n][t = cs_tap_base(kg) + i] (zero past 40 filters / 25 taps), and in the
// otherwise unused k slot t = 25 the per-filter constant shift[c]: against an activation operand whose slot 25 is 1.0 (cs_sfrag_ones) the contraction
// yields  scale[c] * (conv)[c][w] + shift[c]  -- a BatchNorm affine (or the conv bias) costs no vector instruction in the epilogue.
constexpr int CS_ONE_SLOT = 25;
template <class F>
__device__ __forceinline__ void cs_tap_frags_affine(const float* __restrict__ w25, F&& scale_shift, bf16x8 (&wh)[3], bf16x8 (&wl)[3]) {
    const int lane = threadIdx.x & 63, n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
        const int c = 16 * ct + n;
        float sc = 0.f, sh = 0.f;
        if (c < CS_C) scale_shift(c, sc, sh);
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = cs_tap_base(kg) + i;
            v[i] = (c < CS_C && t < CS_K1) ? sc * w25[c * CS_K1 + t] : (t == CS_ONE_SLOT ? sh : 0.f);
        }
        cs_split8(v, wh[ct], wl[ct]);
    }
}
__device__ __forceinline__ void cs_tap_frags(const float* __restrict__ w25, bf16x8 (&wh)[3], bf16x8 (&wl)[3]) {
    cs_tap_frags_affine(w25, [](int, float& sc, float& sh) { sc = 1.f; sh = 0.f; }, wh, wl);
}
// cs_sfrag with k slot 25 (lane group 3 -- tap base 24 --, word 1) replaced by 1.0: the partner of cs_tap_frags_affine's shift slot
__device__ __forceinline__ void cs_sfrag_ones(const unsigned* __restrict__ S32, int h, int wt, bf16x8& hi, bf16x8& lo) {
    const int lane = threadIdx.x & 63, n = lane & 15, kg = lane >> 4;
    const unsigned* p = S32 + h * CS_RS + 5 * (16 * wt + n) + cs_tap_base(kg);
    unsigned w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = p[i];
    const unsigned keep = kg == 3 ? 0x0000ffffu : 0xffffffffu, one = kg == 3 ? 0x3f800000u : 0u;     // bf16(1.0) = 0x3f80 in the upper half of dword 0
    hi = cs_frag((cs_pair_hi(w[1], w[0]) & keep) | one, cs_pair_hi(w[3], w[2]), cs_pair_hi(w[5], w[4]), cs_pair_hi(w[7], w[6]));
    lo = cs_frag(cs_pair_lo(w[1], w[0]) & keep, cs_pair_lo(w[3], w[2]), cs_pair_lo(w[5], w[4]), cs_pair_lo(w[7], w[6]));
}

// ---- packed fp32 helpers: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 process two elements per issue slot (the elementwise epilogues of this family are
// bound by vector-instruction issue, ~4 cycles per wave instruction) ----
__device__ __forceinline__ f32x2_t cs_fma2(f32x2_t a, f32x2_t b, f32x2_t c) {
#if defined(EEG_EMU)
    return f32x2_t{fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])};
#else
    return __builtin_elementwise_fma(a, b, c);
#endif
}
__device__ __forceinline__ float cs_exp2(float x) {
#if defined(EEG_EMU)
    return exp2f(x);
#else
    return __builtin_amdgcn_exp2f(x);
#endif
}
constexpr float CS_LOG2E = 1.4426950408889634f;
constexpr float CS_LN2 = 0.6931471805599453f;
// ELU'(u) = exp(min(u, 0)) for two elements (1 for u >= 0); mneg <- min(u * log2(e), 0).  The min is taken of the PRODUCT, a value the compiler knows to be
// canonical: fminf / fmaxf of a raw MFMA result cost an extra canonicalising v_max each -- and no inline asm here: an asm statement that reads an MFMA
// result gets none of the wait states the hazard recogniser inserts for compiler-visible instructions (wrong values on the hardware, right ones on the
// emulator).
__device__ __forceinline__ f32x2_t cs_elu_grad2(f32x2_t u, f32x2_t& mneg) {
    const f32x2_t m = u * f32x2_t{CS_LOG2E, CS_LOG2E};
    mneg = f32x2_t{fminf(m[0], 0.f), fminf(m[1], 0.f)};
    return f32x2_t{cs_exp2(mneg[0]), cs_exp2(mneg[1])};
}
__device__ __forceinline__ f32x2_t cs_elu_grad2(f32x2_t u) {
    f32x2_t mneg;
    return cs_elu_grad2(u, mneg);
}
// ELU(u) = max(u, 0) + (exp(min(u, 0)) - 1) for two elements, with max(u, 0) = u - ln(2) * min(u * log2(e), 0) as ONE packed fma (exact u for u > 0; the
// product round trip leaves <= 1.2e-7 |u| for u < 0, cf. elu1_fast's 2e-7)
__device__ __forceinline__ f32x2_t cs_elu2(f32x2_t u) {
    f32x2_t mneg;
    const f32x2_t e = cs_elu_grad2(u, mneg);
    return cs_fma2(mneg, f32x2_t{-CS_LN2, -CS_LN2}, u) + (e + f32x2_t{-1.f, -1.f});
}
__device__ __forceinline__ f32x2_t cs_lo2(f32x4 v) { return f32x2_t{v[0], v[1]}; }
__device__ __forceinline__ f32x2_t cs_hi2(f32x4 v) { return f32x2_t{v[2], v[3]}; }

// an integer zero the compiler cannot see through: added to the address of loop-invariant LDS operands it keeps their loads INSIDE the loop (hoisted, the
// per-sample fragments and coefficient rows of the backward kernels occupied ~90 VGPRs for the whole row loop and spilled)
__device__ __forceinline__ int cs_opaque_zero() {
    int z = 0;
#if !defined(EEG_EMU)
    asm volatile("" : "+s"(z));
#endif
    return z;
}

// inclusive prefix sum over the wave in DPP data movement: Hillis-Steele inside the four 16-lane rows (row_shr 1, 2, 4, 8), then the row totals
// (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) -- 6 VALU instructions; the __shfl_up form is 6 dependent ds_bpermute round trips
__device__ __forceinline__ float cs_wave_scan(float v) {
#if defined(EEG_EMU)
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int s = 1; s < 16; s <<= 1) {
        const float o = hipemu::shfl_idx(v, (lane - s) & 63);
        if ((lane & 15) >= s) v += o;
    }
    {
        const float o = hipemu::shfl_idx(v, ((lane & ~15) - 1) & 63);
        if ((lane >> 4) & 1) v += o;
    }
    {
        const float o = hipemu::shfl_idx(v, 31);
        if (lane >= 32) v += o;
    }
    return v;
#else
    v += dpp_mov<0x111, 0xf>(0.f, v);      // row_shr:1 (lanes without a source add the `old` operand: 0)
    v += dpp_mov<0x112, 0xf>(0.f, v);      // row_shr:2
    v += dpp_mov<0x114, 0xf>(0.f, v);      // row_shr:4
    v += dpp_mov<0x118, 0xf>(0.f, v);      // row_shr:8
    v += dpp_mov<0x142, 0xa>(0.f, v);      // row_bcast:15 -> rows 1, 3
    v += dpp_mov<0x143, 0xc>(0.f, v);      // row_bcast:31 -> rows 2, 3
    return v;
#endif
}

// lane l <- x[4l .. 4l+3] of one token row (zero past the 250 samples)
__device__ __forceinline__ void cs_load_row(float (&v)[4], const float* __restrict__ xr, bool ok, bool vec2) {
    const int lane = threadIdx.x & 63;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    if (vec2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = 4 * lane + 2 * h;
            const f32x2 t = (ok && c < CS_T) ? *reinterpret_cast<const f32x2*>(xr + c) : f32x2{0.f, 0.f};
            v[2 * h] = t[0];
            v[2 * h + 1] = t[1];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (ok && 4 * lane + e < CS_T) ? xr[4 * lane + e] : 0.f;
    }
}

// one wave: token row (lane l holds samples 4l .. 4l+3) -> its packed box-filtered row [CS_RS words] (zeros from j = 200); pscr = the wave's 256-float
// scratch row for the exclusive prefix sums P[i] = sum_{k<i} x[k]:  S[j] = (P[j + 51] - P[j]) / 51
__device__ __forceinline__ void cs_box_row(unsigned* __restrict__ srow, float* __restrict__ pscr, const float (&v)[4]) {
    const int lane = threadIdx.x & 63;
    const float p0 = v[0], p1 = p0 + v[1], p2 = p1 + v[2], p3 = p2 + v[3];
    const float base = cs_wave_scan(p3) - p3;
    const f32x4 P{base, base + p0, base + p1, base + p2};
    *reinterpret_cast<f32x4*>(pscr + 4 * lane) = P;
    wave_sync();
    u32x4_t S;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int j = 4 * lane + e;
        S[e] = j < CS_NS ? cs_pack_word((pscr[j + CS_POOL] - P[e]) * (1.0f / CS_POOL)) : 0u;
    }
    wave_sync();
    *reinterpret_cast<u32x4_t*>(srow + 4 * lane) = S;
    if (lane < CS_RS - 256) srow[256 + lane] = 0u;             // words 256 .. 267: read by the padded tiles only, must be finite
}

// The H token rows of sample b -> S32[h][CS_RS].  Every wave stages exactly the rows it later works on (rows wv, wv + 8, ... -- or, PAIRS, the row
// pairs (2q, 2q + 1), q = wv, wv + 8, ...), so no workgroup barrier separates the staging from the contractions: a wave_sync does.  Two halves so that
// other prologue loads can be issued while the rows are in flight.
constexpr int CS_RPW = CS_MAXH / CS_NW;
template <bool PAIRS>
__device__ __forceinline__ int cs_stage_row(int wv, int j) { return PAIRS ? 2 * (wv + CS_NW * (j >> 1)) + (j & 1) : wv + CS_NW * j; }
template <bool PAIRS>
__device__ __forceinline__ void cs_stage_load(float (&vx)[CS_RPW][4], const float* __restrict__ x, long long xs_b, long long xs_h, int b, int H, bool vec2) {
    const int wv = wave_uniform(threadIdx.x >> 6);
#pragma unroll
    for (int j = 0; j < CS_RPW; ++j) {
        const int h = cs_stage_row<PAIRS>(wv, j);
        cs_load_row(vx[j], x + (long long)b * xs_b + (long long)(h < H ? h : 0) * xs_h, h < H, vec2);
    }
}
template <bool PAIRS>
__device__ __forceinline__ void cs_stage_finish(unsigned* __restrict__ S32, float* __restrict__ pscr, const float (&vx)[CS_RPW][4], int H) {
    const int wv = wave_uniform(threadIdx.x >> 6);
#pragma unroll
    for (int j = 0; j < CS_RPW; ++j) {
        const int h = cs_stage_row<PAIRS>(wv, j);
        if (h < H) cs_box_row(S32 + h * CS_RS, pscr, vx[j]);
    }
    wave_sync();
}

// BatchNorm batch statistics from `nrows` partial rows [sum(40) | sumsq(40)] (fp64), summed in a FIXED order by every workgroup that needs them: no
// atomics, nothing to clear, the same value everywhere.  nrows = 1: the sums themselves (after a data-parallel all-reduce).  All CS_NT threads take
// part: thread (slice, col) sums rows slice, slice + 6, ...; the six slices are then added in slice order.  scratch: [6][80] doubles of LDS.
constexpr int CS_BN_SLICES = 6;
__device__ __forceinline__ void cs_bn_rows_partial(const double* __restrict__ rows, int nrows, double* __restrict__ scratch) {
    const int t = threadIdx.x;
    if (t < CS_BN_SLICES * 2 * CS_C) {
        const int sl = t / (2 * CS_C), col = t % (2 * CS_C);
        // (22 rows in flight: the 43 rows of a slice at B = 256 are two L2 round trips at the head of every sample-major kernel, not six)
        scratch[t] = ordered_column_sum<CS_BN_SLICES, 22>(rows, nrows, sl, 2 * CS_C, col);
    }
}
// (behind a barrier) channel c's batch mean, 1 / sqrt(var + eps) and biased variance
__device__ __forceinline__ void cs_bn_rows_finish(const double* __restrict__ scratch, double count, float eps, int c, float& mean, float& rstd, double& var_out) {
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int sl = 0; sl < CS_BN_SLICES; ++sl) {
        s += scratch[sl * 2 * CS_C + c];
        q += scratch[sl * 2 * CS_C + CS_C + c];
    }
    const double m = s / count;
    double var = q / count - m * m;
    if (var < 0.0) var = 0.0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
    var_out = var;
}

// ---- packed spatial weights of the forward chain ---------------------------------------------------------------------------------------------------
// Token rows are taken in pairs (h0, h1) = (2 q, 2 q + 1); pair q has three k-steps of 32:
//   step 3q     k slot j of lane group kg <-> c = 16 (j >> 2) + 4 kg + (j & 3) (filters 0 .. 31) of row h0      = accumulator tiles ct 0 | 1 of h0
//   step 3q + 1 the same of row h1
//   step 3q + 2 slots 0 .. 3 <-> c = 32 + 4 kg + j of h0, slots 4 .. 7 <-> c = 32 + 4 kg + (j - 4) of h1 (c >= 40: zero)   = tile ct 2 of both rows
// i.e. 2.5 accumulator tiles of real filters per row in 1.5 MFMA k-steps.  Fragment (step, ot, plane): 64 lanes x 16 bytes, lane (n, kg) <- o = 16 ot + n.
constexpr int CSP_FRAG = 1024;                       // bytes
__host__ __device__ inline long long csp_offset(int step, int ot, int plane) { return (((long long)step * 3 + ot) * 2 + plane) * CSP_FRAG; }
__device__ __forceinline__ void cs_pack_item(const float* __restrict__ Ws, unsigned char* __restrict__ packed, int H, int id) {
    const int lane = id & 63, ot = (id >> 6) % 3, step = id / 192;
    const int n = lane & 15, kg = lane >> 4, o = 16 * ot + n;
    const int q = step / 3, kind = step % 3;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int c, h;
        if (kind < 2) { c = 16 * (j >> 2) + 4 * kg + (j & 3); h = 2 * q + kind; }
        else          { c = 32 + 4 * kg + (j & 3);            h = 2 * q + (j >> 2); }
        v[j] = (o < CS_C && c < CS_C && h < H) ? Ws[((long long)o * CS_C + c) * H + h] : 0.f;
    }
    bf16x8 hi, lo;
    cs_split8(v, hi, lo);
    *reinterpret_cast<bf16x8*>(packed + csp_offset(step, ot, 0) + 16 * lane) = hi;
    *reinterpret_cast<bf16x8*>(packed + csp_offset(step, ot, 1) + 16 * lane) = lo;
}

// ---- Ws^T fragments of the backward (dz1 = Ws^T dy2: rows = filters c, k = out channels o) -------------------------------------------------------------
// per token row h and filter tile ct:  main hi | main lo (1024 B each: lane (n, kg) slot j <-> o = 16 (j >> 2) + 4 kg + (j & 3), filter c = 16 ct + n)
//                                      tail hi | tail lo (512 B each: slot j < 4 <-> o = 32 + 4 kg + j, zero from o = 40)
constexpr int CST_TILE = 3072, CST_ROW = 3 * CST_TILE;
__device__ __forceinline__ void cs_pack_t_item(const float* __restrict__ Ws, unsigned char* __restrict__ packed, int H, int id) {
    const int lane = id & 63, ct = (id >> 6) % 3, h = id / 192;
    const int n = lane & 15, kg = lane >> 4, c = 16 * ct + n;
    float v[8], tl[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int o = 16 * (j >> 2) + 4 * kg + (j & 3);
        v[j] = c < CS_C ? Ws[((long long)o * CS_C + c) * H + h] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int o = 32 + 4 * kg + j;
        tl[j] = (c < CS_C && o < CS_C) ? Ws[((long long)o * CS_C + c) * H + h] : 0.f;
    }
    bf16x8 hi, lo;
    cs_split8(v, hi, lo);
    u32x2_t th, tlo;
    x3_split4(tl[0], tl[1], tl[2], tl[3], th, tlo);
    unsigned char* base = packed + (long long)h * CST_ROW + ct * CST_TILE;
    *reinterpret_cast<bf16x8*>(base + 16 * lane) = hi;
    *reinterpret_cast<bf16x8*>(base + 1024 + 16 * lane) = lo;
    *reinterpret_cast<u32x2_t*>(base + 2048 + 8 * lane) = th;
    *reinterpret_cast<u32x2_t*>(base + 2560 + 8 * lane) = tlo;
}
// work item `id` of both packs: the forward's fragments first, then the backward's (packed_t may be null)
__host__ __device__ inline int cs_pack_items(int H, bool with_t) { return 3 * ((H + 1) / 2) * 3 * 64 + (with_t ? H * 3 * 64 : 0); }
__device__ __forceinline__ void cs_pack_both(const float* __restrict__ Ws, unsigned char* __restrict__ packed, unsigned char* __restrict__ packed_t, int H, int id) {
    const int n1 = 3 * ((H + 1) / 2) * 3 * 64;
    if (id < n1) cs_pack_item(Ws, packed, H, id);
    else if (packed_t && id - n1 < H * 3 * 64) cs_pack_t_item(Ws, packed_t, H, id - n1);
}


// ---- BatchNorm1 batch sums of ONE sample (workgroup of CS_NT threads, CS_LDS_S + CS_LDS_PS bytes of LDS at `ldsb`): y1 = pool(conv(x)) of the sample's H
// token rows through the tap contraction, its per-channel [sum | sumsq] as one fp64 partial row rows[b][80].  The body of cstack_stats1_kernel (csrc/cstack.hip)
// -- and, round 6, the tail of token_block_fwd_kernel (csrc/token_block.hip): the rows it has just written are still in L2, a launch and a 16 MB read go.
constexpr int CS_LDS_S = CS_MAXH * CS_RS * 4;        // packed rows
constexpr int CS_LDS_PS = CS_NW * 256 * 4;           // prefix-sum scratch
__device__ __forceinline__ void cs_stats1_sample(unsigned char* __restrict__ ldsb, const float* __restrict__ x, long long xs_b, long long xs_h,
                                                 const float* __restrict__ w25, const float* __restrict__ bias, double* __restrict__ rows, int b, int H, int vec2) {

    unsigned* S32 = reinterpret_cast<unsigned*>(ldsb);
    float* ps = reinterpret_cast<float*>(ldsb + CS_LDS_S);
    float* sc = ps;                                   // [NW][2][48] per-wave filter sums (after the staging)
    const int t = threadIdx.x, lane = t & 63, wv = wave_uniform(t >> 6);
    const int n = lane & 15, kg = lane >> 4;
    float vx[CS_RPW][4];
    cs_stage_load<false>(vx, x, xs_b, xs_h, b, H, vec2 != 0);
    bf16x8 wh[3], wl[3];
    cs_tap_frags_affine(w25, [&](int c, float& sc, float& sh) { sc = 1.f; sh = bias[c]; }, wh, wl);      // the conv bias rides in the ones slot
    cs_stage_finish<false>(S32, ps + wv * 256, vx, H);          // (a wave works on the rows it staged: no workgroup barrier)
    f32x2_t ss[3][2], sq[3][2];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int k = 0; k < 2; ++k) { ss[ct][k] = f32x2_t{0.f, 0.f}; sq[ct][k] = f32x2_t{0.f, 0.f}; }
    for (int h = wv; h < H; h += CS_NW) {
#pragma unroll
        for (int wt = 0; wt < 3; ++wt) {
            bf16x8 xh, xl;
            cs_sfrag_ones(S32, h, wt, xh, xl);
            const float wm = 16 * wt + n < CS_W ? 1.f : 0.f;                                               // (only the last tile has columns w >= 36)
            f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            cs_mma3_a3(wh, wl, xh, xl, acc);                                                             // y1: D[c = 16 ct + 4 kg + r][w = 16 wt + n]
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) {
                f32x2_t v0 = cs_lo2(acc[ct]), v1 = cs_hi2(acc[ct]);
                if (wt == 2) { v0 = v0 * f32x2_t{wm, wm}; v1 = v1 * f32x2_t{wm, wm}; }
                ss[ct][0] += v0;
                ss[ct][1] += v1;
                sq[ct][0] = cs_fma2(v0, v0, sq[ct][0]);
                sq[ct][1] = cs_fma2(v1, v1, sq[ct][1]);
            }
        }
    }
    __syncthreads();                                  // the scratch rows are dead: they become the per-wave sums
#pragma unroll
    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = ss[ct][r >> 1][r & 1], q = sq[ct][r >> 1][r & 1];
#pragma unroll
            for (int msk = 8; msk >= 1; msk >>= 1) { a += __shfl_xor(a, msk, 64); q += __shfl_xor(q, msk, 64); }
            if (n == 0) { sc[(wv * 2 + 0) * 48 + 16 * ct + 4 * kg + r] = a; sc[(wv * 2 + 1) * 48 + 16 * ct + 4 * kg + r] = q; }
        }
    __syncthreads();
    if (t < 2 * CS_C) {
        const int which = t / CS_C, c = t % CS_C;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < CS_NW; ++k) s += (double)sc[(k * 2 + which) * 48 + c];
        rows[(long long)b * 2 * CS_C + t] = s;
    }
}

}  // namespace eeg
