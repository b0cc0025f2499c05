// Attention backward with split-bf16 products (SelfAttention_Family.py:59-70 differentiated; the exact-fp32 kernel is csrc/attention.hip).
//
// One 256-thread workgroup per (sample, head), L = 64 tokens, E <= 64 dims.  Q, K, V, dO live in LDS as bf16 hi / lo planes in their NATURAL
// layout [token][dim] (row stride 144 B: conflict-free 16-byte fragment reads): 4 x 18 KB = 72 KB, two workgroups per CU.  Every product is
// three v_mfma_f32_16x16x32_bf16 (hi*lo + lo*hi + hi*hi, fp32 accumulate) instead of eight v_mfma_f32_16x16x4_f32 per 32 k.
//
//   phase T (wave w owns queries 16w .. 16w+15):   S^T = K Q^T and dPd^T = V dO^T (contraction over the dims: natural planes are k-contiguous),
//       softmax over the keys in-lane + 2 shuffles, dropout mask regenerated (one Philox block = the lane's 4 consecutive keys),
//       dP = dPd o mask / (1-p),  D_q = sum_k P dP,  dS = P o (dP - D_q);  dQ^T = K^T dS^T.
//   phase N (wave w owns keys 16w .. 16w+15):      dV^T = dO^T Pd,  dK^T = Q^T dS   (contraction over the QUERIES).
//
// The products that contract over tokens need operands with the token index contiguous per lane while the planes are [token][dim]: those fragments
// come through ds_read_b64_tr_b16 -- the LDS transpose read of gfx950: within a 16-lane group, lane i receives as element j the (i & 3)-th 16-bit
// element of the 8 bytes addressed by lane 4 j + (i >> 2) (tools/micro/tr_read_probe.hip checks exactly this on the hardware), so when lane l
// addresses row r0 + (l >> 2), columns c0 + 4 (l & 3) .., lane i ends up with rows r0 .. r0 + 3 of column c0 + i.  The other operand of those
// products is what the lane already holds: an accumulator tile has 4 consecutive rows of one column per lane, i.e. 4 consecutive k of its column
// once the MFMA k slots are assigned accordingly (slots 0-3 <-> tile 2m, 4-7 <-> tile 2m+1) -- no LDS round trip for P / dS in phase T; for
// phase N the dropped probabilities and dS are parked as planes [query][key] in the (by then dead) K and V regions.
#include "eeg_common.h"

namespace eeg {

constexpr int AX_L = 64;
constexpr int AX_RS = 144;                        // bytes per plane row: 64 bf16 + 16 pad
constexpr int AX_PLANE = AX_L * AX_RS;            // 9216
constexpr int AX_TENSOR = 2 * AX_PLANE;           // hi | lo

struct ax_args {
    const float* qkv;     // (B*L, ld): q at col h*E+e, k at HE + h*E+e, v at 2HE + h*E+e
    const float* dctx;    // (B*L, H*E)
    float* dqkv;          // (B*L, ld), or null:
    unsigned char* dqkvp; // dq | dk | dv as token planes (csrc/wgrad_tok.hip: per sample [hi | lo][64][256], channel 64 head + e), groups B * 64 KB apart
    int B, H, E, ld;
    float scale, drop_p;
    unsigned long long seed;
    unsigned site;
};

typedef short ax_s4 __attribute__((ext_vector_type(4)));
typedef float ax_f2 __attribute__((ext_vector_type(2)));
typedef float ax_f4u __attribute__((ext_vector_type(4), aligned(4)));      // 16-byte global access from a dword-aligned address

// LDS transpose read (see the header): 4 bf16 for this lane
__device__ __forceinline__ ax_s4 ax_tr_read(const unsigned char* p) {
#if defined(EEG_EMU)
    const int lane = hipemu::cur->lane, g = lane >> 4, i = lane & 15;
    ax_s4 r;
    for (int j = 0; j < 4; ++j) {
        const unsigned long long src = hipemu::shfl_idx((unsigned long long)(uintptr_t)p, 16 * g + 4 * j + (i >> 2));
        r[j] = reinterpret_cast<const short*>((uintptr_t)src)[i & 3];
    }
    return r;
#else
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ax_s4*)(p));
#endif
}
// fragment with k slots 0-3 <-> rows r0 .. r0+3 and 4-7 <-> rows r1 .. r1+3 of column c0 + (lane & 15) of a [row][col] plane
__device__ __forceinline__ bf16x8 ax_tr_frag(const unsigned char* plane, int r0, int r1, int c0, int lane) {
    const int l = lane & 15;
    const ax_s4 a = ax_tr_read(plane + (r0 + (l >> 2)) * AX_RS + 2 * (c0 + 4 * (l & 3)));
    const ax_s4 b = ax_tr_read(plane + (r1 + (l >> 2)) * AX_RS + 2 * (c0 + 4 * (l & 3)));
    return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
__device__ __forceinline__ bf16x8 ax_frag(const unsigned char* plane, int row, int k0) {      // 8 consecutive k of one row
    return *reinterpret_cast<const bf16x8*>(plane + row * AX_RS + 2 * k0);
}
__device__ __forceinline__ bf16x8 ax_pack8(u32x2_t a, u32x2_t b) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 v{a[0], a[1], b[0], b[1]};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ f32x4 ax_mfma3(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x4 c) {
    c = mfma_bf16_16x16x32(al, bh, c);
    c = mfma_bf16_16x16x32(ah, bl, c);
    return mfma_bf16_16x16x32(ah, bh, c);
}
// 4 floats (e0 ..) of one output row: `valid` of them exist
__device__ __forceinline__ void ax_store4(float* p, int valid, f32x4 v) {
    if (valid >= 4) *reinterpret_cast<ax_f4u*>(p) = ax_f4u{v[0], v[1], v[2], v[3]};
    else if (valid >= 2) {
        *reinterpret_cast<ax_f2*>(p) = ax_f2{v[0], v[1]};
        if (valid == 3) p[2] = v[2];
    } else if (valid == 1) p[0] = v[0];
}

// 4 consecutive dims (e0 ..) of one token row as token planes: 8 bytes into the hi plane, 8 into the lo plane 32 KB behind.  Dims >= E are exact
// zeros in `v` (the operand planes are zero there), and they must be WRITTEN: the consumers read all 64 channels of a head.
__device__ __forceinline__ void ax_store4_planes(unsigned char* row_base, int e0, f32x4 v) {
    u32x2_t hi, lo;
    x3_split4(v[0], v[1], v[2], v[3], hi, lo);
    *reinterpret_cast<u32x2_t*>(row_base + 2 * e0) = hi;
    *reinterpret_cast<u32x2_t*>(row_base + 32768 + 2 * e0) = lo;
}

template <bool TRAIN>
__global__ __launch_bounds__(256, 2) void attention_bwd_x3_kernel(const ax_args a) {
    EEG_LDS_BASE(float, ldsf);
    unsigned char* lds = reinterpret_cast<unsigned char*>(ldsf);
    unsigned char* Qp = lds;
    unsigned char* Kp = lds + AX_TENSOR;           // phase N: Pd planes [query][key]
    unsigned char* Vp = lds + 2 * AX_TENSOR;       // phase N: dS planes [query][key]
    unsigned char* Op = lds + 3 * AX_TENSOR;
    const int t = threadIdx.x, lane = t & 63, w = wave_uniform(t >> 6), fr = lane & 15, g = lane >> 4;
    const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
    const int E = a.E, HE = a.H * E;
    const float* qrow0 = a.qkv + (long long)b * AX_L * a.ld + h * E;
    const float* orow0 = a.dctx + (long long)b * AX_L * HE + h * E;

    // ---- stage Q, K, V, dO: (64 x E) fp32 -> hi / lo planes, columns E .. 63 zero.  All 32 loads of a thread before its first LDS store.
    {
        ax_f2 v[4][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = t + 256 * j, row = idx >> 5, col = 2 * (idx & 31);
            const bool ok = col < E;
            const float* qr = qrow0 + (long long)row * a.ld + col;
            v[0][j] = ok ? *reinterpret_cast<const ax_f2*>(qr) : ax_f2{0.f, 0.f};
            v[1][j] = ok ? *reinterpret_cast<const ax_f2*>(qr + HE) : ax_f2{0.f, 0.f};
            v[2][j] = ok ? *reinterpret_cast<const ax_f2*>(qr + 2 * HE) : ax_f2{0.f, 0.f};
            v[3][j] = ok ? *reinterpret_cast<const ax_f2*>(orow0 + (long long)row * HE + col) : ax_f2{0.f, 0.f};
        }
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = t + 256 * j, row = idx >> 5, col = 2 * (idx & 31);
                const unsigned hi = x3_pack2(v[x][j][0], v[x][j][1]);
                const float r0 = v[x][j][0] - __uint_as_float(hi << 16), r1 = v[x][j][1] - __uint_as_float(hi & 0xffff0000u);
                unsigned char* dst = lds + x * AX_TENSOR + row * AX_RS + 2 * col;
                *reinterpret_cast<unsigned*>(dst) = hi;
                *reinterpret_cast<unsigned*>(dst + AX_PLANE) = x3_pack2(r0, r1);
            }
    }
    __syncthreads();

    // ---- phase T: this wave's 16 queries against all 64 keys
    f32x4 s[4], dp[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) { s[kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[kt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int k0 = 32 * ks + 8 * g;
        const bf16x8 qh = ax_frag(Qp, 16 * w + fr, k0), ql = ax_frag(Qp + AX_PLANE, 16 * w + fr, k0);
        const bf16x8 oh = ax_frag(Op, 16 * w + fr, k0), ol = ax_frag(Op + AX_PLANE, 16 * w + fr, k0);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const bf16x8 kh = ax_frag(Kp, 16 * kt + fr, k0), kl = ax_frag(Kp + AX_PLANE, 16 * kt + fr, k0);
            const bf16x8 vh = ax_frag(Vp, 16 * kt + fr, k0), vl = ax_frag(Vp + AX_PLANE, 16 * kt + fr, k0);
            s[kt] = ax_mfma3(kh, kl, qh, ql, s[kt]);              // S^T[key 16 kt + 4 g + r][query 16 w + fr]
            dp[kt] = ax_mfma3(vh, vl, oh, ol, dp[kt]);            // (dO V^T)^T, same layout
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[kt][r] *= a.scale;
            mx = fmaxf(mx, s[kt][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[kt][r] = expf(s[kt][r] - mx);
            sum += s[kt][r];
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    const bool drop = TRAIN && a.drop_p > 0.f;
    const float ksc = drop ? 1.f / (1.f - a.drop_p) : 1.f;
    f32x4 pd[4];                                                   // dropped, rescaled probabilities (feed dV)
    float dq_sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        bool keep[4] = {true, true, true, true};
        if (drop) {
            const unsigned long long idx0 = ((unsigned long long)bh * AX_L + 16 * w + fr) * AX_L + 16 * kt + 4 * g;
            dropout_keep4(a.seed, a.site, idx0, a.drop_p, keep);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = s[kt][r] * inv;
            const float d = keep[r] ? dp[kt][r] * ksc : 0.f;       // gradient w.r.t. the un-dropped probability
            pd[kt][r] = keep[r] ? p * ksc : 0.f;
            s[kt][r] = p;
            dp[kt][r] = d;
            dq_sum += p * d;
        }
    }
    dq_sum += __shfl_xor(dq_sum, 16, 64);
    dq_sum += __shfl_xor(dq_sum, 32, 64);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dp[kt][r] = s[kt][r] * (dp[kt][r] - dq_sum);      // dS (w.r.t. the scaled scores)

    // dQ^T[e][q] = sum_key K^T[e][key] dS^T[key][q]: B = the lane's own dS values (k slots: tile 2m rows 4g.., tile 2m+1 rows 4g..)
    {
        f32x4 dq[4];
#pragma unroll
        for (int et = 0; et < 4; ++et) dq[et] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            u32x2_t h0, l0, h1, l1;
            x3_split4(dp[2 * m][0], dp[2 * m][1], dp[2 * m][2], dp[2 * m][3], h0, l0);
            x3_split4(dp[2 * m + 1][0], dp[2 * m + 1][1], dp[2 * m + 1][2], dp[2 * m + 1][3], h1, l1);
            const bf16x8 bhf = ax_pack8(h0, h1), blf = ax_pack8(l0, l1);
#pragma unroll
            for (int et = 0; et < 4; ++et) {
                const bf16x8 ah = ax_tr_frag(Kp, 32 * m + 4 * g, 32 * m + 16 + 4 * g, 16 * et, lane);
                const bf16x8 al = ax_tr_frag(Kp + AX_PLANE, 32 * m + 4 * g, 32 * m + 16 + 4 * g, 16 * et, lane);
                dq[et] = ax_mfma3(ah, al, bhf, blf, dq[et]);      // dQ^T[e 16 et + 4 g + r][query 16 w + fr]
            }
        }
        float* out = a.dqkv + ((long long)b * AX_L + 16 * w + fr) * a.ld + h * E;
        unsigned char* outp = a.dqkvp + (long long)b * 65536 + (16 * w + fr) * 512 + 128 * h;
#pragma unroll
        for (int et = 0; et < 4; ++et) {
            const int e0 = 16 * et + 4 * g;
            f32x4 v = dq[et];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= a.scale;
            if (a.dqkvp) ax_store4_planes(outp, e0, v);
            else if (e0 < E) ax_store4(out + e0, E - e0, v);
        }
    }
    __syncthreads();                                               // every wave is done with K and V
    // Pd -> the K region, dS -> the V region, as planes [query][key] (the lane holds 4 consecutive keys of its query)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        u32x2_t hi, lo;
        const int off = (16 * w + fr) * AX_RS + 2 * (16 * kt + 4 * g);
        x3_split4(pd[kt][0], pd[kt][1], pd[kt][2], pd[kt][3], hi, lo);
        *reinterpret_cast<u32x2_t*>(Kp + off) = hi;
        *reinterpret_cast<u32x2_t*>(Kp + AX_PLANE + off) = lo;
        x3_split4(dp[kt][0], dp[kt][1], dp[kt][2], dp[kt][3], hi, lo);
        *reinterpret_cast<u32x2_t*>(Vp + off) = hi;
        *reinterpret_cast<u32x2_t*>(Vp + AX_PLANE + off) = lo;
    }
    __syncthreads();

    // ---- phase N: this wave's 16 keys; contraction over the queries (k slots 0-7 <-> queries 32 ks + 8 g + 0..7 on both operands)
    f32x4 dv[4], dk[4];
#pragma unroll
    for (int et = 0; et < 4; ++et) { dv[et] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[et] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int q0 = 32 * ks + 8 * g;
        const bf16x8 ph = ax_tr_frag(Kp, q0, q0 + 4, 16 * w, lane), pl = ax_tr_frag(Kp + AX_PLANE, q0, q0 + 4, 16 * w, lane);
        const bf16x8 sh = ax_tr_frag(Vp, q0, q0 + 4, 16 * w, lane), sl = ax_tr_frag(Vp + AX_PLANE, q0, q0 + 4, 16 * w, lane);
#pragma unroll
        for (int et = 0; et < 4; ++et) {
            const bf16x8 oh = ax_tr_frag(Op, q0, q0 + 4, 16 * et, lane), ol = ax_tr_frag(Op + AX_PLANE, q0, q0 + 4, 16 * et, lane);
            const bf16x8 qh = ax_tr_frag(Qp, q0, q0 + 4, 16 * et, lane), ql = ax_tr_frag(Qp + AX_PLANE, q0, q0 + 4, 16 * et, lane);
            dv[et] = ax_mfma3(oh, ol, ph, pl, dv[et]);            // dV^T[e 16 et + 4 g + r][key 16 w + fr]
            dk[et] = ax_mfma3(qh, ql, sh, sl, dk[et]);            // dK^T, same layout
        }
    }
    if (a.dqkvp) {
        const long long grp = (long long)a.B * 65536;
        unsigned char* outp = a.dqkvp + (long long)b * 65536 + (16 * w + fr) * 512 + 128 * h;
#pragma unroll
        for (int et = 0; et < 4; ++et) {
            const int e0 = 16 * et + 4 * g;
            f32x4 v = dk[et];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= a.scale;
            ax_store4_planes(outp + grp, e0, v);
            ax_store4_planes(outp + 2 * grp, e0, dv[et]);
        }
        return;
    }
    float* outk = a.dqkv + ((long long)b * AX_L + 16 * w + fr) * a.ld + HE + h * E;
#pragma unroll
    for (int et = 0; et < 4; ++et) {
        const int e0 = 16 * et + 4 * g;
        if (e0 < E) {
            f32x4 v = dk[et];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= a.scale;
            ax_store4(outk + e0, E - e0, v);
            ax_store4(outk + HE + e0, E - e0, dv[et]);
        }
    }
}

}  // namespace eeg

using namespace eeg;

// same contract as eegclip_attention_bwd (csrc/attention.hip); split-bf16 products.  E and ld even, qkv / dctx / dqkv 8-byte aligned.
// dqkv_planes = 1: `dqkv` receives dq | dk | dv as token planes instead (3 groups of B * 64 KB, 16-byte aligned; needs H * 64 <= 256)
extern "C" int eegclip_attention_bwd_x3(const float* qkv, const float* dctx, void* dqkv, int dqkv_planes, int B, int L, int H, int E, int ld, float scale,
                                        float drop_p, unsigned long long seed, unsigned site, void* stream) {
    if (dqkv_planes && (H > 4 || (reinterpret_cast<uintptr_t>(dqkv) & 15u))) return dqkv && H <= 4 ? EEGCLIP_EALIGN : EEGCLIP_EINVAL;
    if (!qkv || !dctx || !dqkv || B < 1 || L != AX_L || H < 1 || E < 2 || E > 64 || (E & 1) || (ld & 1) || ld < 3 * H * E || drop_p < 0.f || drop_p >= 1.f)
        return EEGCLIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(dctx) | reinterpret_cast<uintptr_t>(dqkv)) & 7u) return EEGCLIP_EALIGN;
    const ax_args a{qkv, dctx, dqkv_planes ? nullptr : static_cast<float*>(dqkv), dqkv_planes ? static_cast<unsigned char*>(dqkv) : nullptr, B, H, E, ld, scale,
                    drop_p, seed, site};
    const size_t lds = 4 * AX_TENSOR;
    if (drop_p > 0.f) EEG_LAUNCH(attention_bwd_x3_kernel<true>, dim3(B * H), dim3(256), lds, stream, a);
    else              EEG_LAUNCH(attention_bwd_x3_kernel<false>, dim3(B * H), dim3(256), lds, stream, a);
    return (int)hipGetLastError();
}
