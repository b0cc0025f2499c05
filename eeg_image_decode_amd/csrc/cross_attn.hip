// SDXL UNet cross-attention with the IP-Adapter image branch fused in (call site Generation/custom_pipeline.py:365-373; the
// arithmetic is diffusers' AttnProcessor2_0 / IPAdapterAttnProcessor2_0, head_dim 64):
//
//     O = softmax(Q K_txt^T / 8) V_txt  +  ip_scale * softmax(Q K_ip^T / 8) V_ip          Q (B,HW,h*64); K/V (B,S,h*64), S = 77 (+ 4 image tokens)
//
// The key/value set is tiny (77 + 4 tokens), so one workgroup keeps the whole K and V^T of its (sample, head) in LDS and
// streams queries through the 16-bit matrix cores (v_mfma_f32_16x16x32_{bf16,f16}); HBM traffic is exactly Q in + O out.
// Scores are computed TRANSPOSED (S^T = K Q^T): a lane then owns one query column, the softmax is in-lane plus two
// xor-shuffles (16, 32), and the probabilities are already in A-operand order for P V -- no LDS round trip for P.
// The k-slot permutation this implies (tile pair 2u/2u+1, rows 4g+r) is matched on the V side by reading V^T with two
// 8-byte LDS reads per lane.  Both branches accumulate into the same fp32 accumulators (ip probabilities pre-scaled).
#include "eeg_common.h"

#include <stdlib.h>

namespace eeg {

constexpr int CA_D = 64;          // head_dim
constexpr int CA_MAXT = 8;        // key tiles of 16 -> S <= 128
constexpr int CA_KLD = CA_D + 16; // K row stride in halfs (160 B = 16 B * 10: the 16 lanes of a ds_read_b128 service group hit 16 distinct slots;
                                  // 144 B measured 5.8 conflict cycles per LDS instruction)
constexpr int CA_QB = 512;        // queries per workgroup (4 waves x 8 tiles of 16): the K / V^T staging of a (sample, head) is amortised over them

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma_f16_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
#if defined(EEG_EMU)
    struct AB { bf16x8 a, b; } in{a, b};
    auto all = hipemu::wave_allgather(&in, sizeof(in));
    const int l = hipemu::cur->lane, col = l & 15, rb = (l >> 4) * 4;
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        float acc = c[r];
        for (int q = 0; q < 4; ++q) {
            AB ra, rbv;
            memcpy(&ra, all[(rb + r) + 16 * q], sizeof(AB));
            memcpy(&rbv, all[col + 16 * q], sizeof(AB));
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
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
#endif
}

template <bool F16>
__device__ __forceinline__ unsigned short to_h(float v) {
    if (F16) {
        _Float16 h = (_Float16)v;
        unsigned short u;
        memcpy(&u, &h, 2);
        return u;
    }
    return f32_to_bf16_bits(v);
}
// two fp32 -> one dword of two 16-bit floats (round to nearest even): one v_cvt_pk_{f16,bf16}_f32 on gfx950
template <bool F16>
__device__ __forceinline__ unsigned pack2(float a, float b) {
#if defined(EEG_EMU)
    return (unsigned)to_h<F16>(a) | ((unsigned)to_h<F16>(b) << 16);
#else
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef _Float16 h16x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 b16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ v{a, b};
    if (F16) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h16x2_));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b16x2_));
#endif
}
__device__ __forceinline__ float fast_exp2(float x) {
#if defined(EEG_EMU)
    return exp2f(x);
#else
    return __builtin_amdgcn_exp2f(x);        // v_exp_f32
#endif
}
typedef unsigned int ca_u32x4 __attribute__((ext_vector_type(4)));

template <bool F16>
__device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) {
    return F16 ? mfma_f16_16x16x32(a, b, c) : mfma_bf16_16x16x32(a, b, c);
}

// V^T row stride in halfs for nt key tiles: >= 32 * ceil(nt / 2) keys and = 4 (mod 64), i.e. 2 dwords (mod 32): the 16 lanes of a
// service group of the paired 8-byte reads (ds_read2_b64, 32 banks) then walk all 32 banks (a 208-byte stride measured 4.7 conflict
// cycles per LDS instruction)
__host__ __device__ inline int ca_ldv(int nt) {
    const int need = ((nt + 1) / 2) * 32;
    return need <= 4 ? 4 : ((need - 4 + 63) / 64) * 64 + 4;
}

struct ca_args {
    const unsigned short *q, *k, *v, *k_ip, *v_ip;
    unsigned short* out;
    int B, HW, heads, S, S_ip;
    float scale, ip_scale;
};

// stage K rows [S][64] (padded stride) and V transposed [64][ldv] for one (b, head); rows >= S are zero.  Two phases: every global load of
// BOTH branches is issued before the first LDS store (kv_issue ... kv_issue, kv_commit ... kv_commit).  As one load-then-store loop per
// operand each iteration waited for its own load: ~10 dependent HBM round trips per workgroup, about half of the launch (the 1280
// workgroups of a UNet attention call each stage their own 20 KB of keys and values).
constexpr int CA_KIT = (CA_MAXT * 16 * (CA_D / 8) + 255) / 256;       // 16-byte pieces of K per thread
constexpr int CA_VIT = (132 * (CA_D / 8) + 255) / 256;                // ... of V (ca_ldv() never exceeds 132 keys)
struct kv_stage {
    uint4 k[CA_KIT], v[CA_VIT];
};

__device__ __forceinline__ void kv_issue(kv_stage& r, int ldv, const unsigned short* k, const unsigned short* v, int S, int S_pad,
                                         long long row_stride) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < CA_KIT; ++j) {
        const int i = t + 256 * j, row = i / (CA_D / 8), c8 = i % (CA_D / 8);
        r.k[j] = make_uint4(0, 0, 0, 0);
        if (i < S_pad * (CA_D / 8) && row < S) r.k[j] = *reinterpret_cast<const uint4*>(k + row * row_stride + c8 * 8);
    }
#pragma unroll
    for (int j = 0; j < CA_VIT; ++j) {
        const int i = t + 256 * j, key = i / (CA_D / 8), d8 = 8 * (i % (CA_D / 8));
        r.v[j] = make_uint4(0, 0, 0, 0);
        if (i < ldv * (CA_D / 8) && key < S) r.v[j] = *reinterpret_cast<const uint4*>(v + key * row_stride + d8);
    }
}

__device__ __forceinline__ void kv_commit(const kv_stage& r, unsigned short* Ks, unsigned short* Vt, int ldv, int S_pad) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < CA_KIT; ++j) {
        const int i = t + 256 * j, row = i / (CA_D / 8), c8 = i % (CA_D / 8);
        if (i < S_pad * (CA_D / 8)) *reinterpret_cast<uint4*>(Ks + row * CA_KLD + c8 * 8) = r.k[j];
    }
#pragma unroll
    for (int j = 0; j < CA_VIT; ++j) {                                 // V^T[d][key], zero padded along keys: transposed 2-byte stores
        const int i = t + 256 * j, key = i / (CA_D / 8), d8 = 8 * (i % (CA_D / 8));
        if (i < ldv * (CA_D / 8)) {
            const unsigned w[4] = {r.v[j].x, r.v[j].y, r.v[j].z, r.v[j].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Vt[(d8 + 2 * e) * ldv + key] = (unsigned short)(w[e] & 0xffffu);
                Vt[(d8 + 2 * e + 1) * ldv + key] = (unsigned short)(w[e] >> 16);
            }
        }
    }
}

// one branch (text or ip) for a 16-query tile: scores^T, softmax over keys, P V accumulated into acc[4].  `scale2` = scale * log2(e):
// the softmax runs in base 2 (v_exp_f32); only the last key tile can hold padding keys and is the only one masked.  (The kernel was
// VALU-bound, not HBM-bound: 880 VALU instructions per 16-query tile -- libm expf, element-wise 16-bit packing and unpacking.)
template <bool F16>
__device__ __forceinline__ void branch(const unsigned short* Ks, const unsigned short* Vt, int ldv, int S, int ntile, const bf16x8 (&bq)[2],
                                       float scale2, float pscale, f32x4 (&acc)[4], int lane) {
    const int fr = lane & 15, g = lane >> 4;
    f32x4 s[CA_MAXT];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < CA_MAXT; ++t) {
        if (t >= ntile) break;
        f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const bf16x8 ak = *reinterpret_cast<const bf16x8*>(Ks + (16 * t + fr) * CA_KLD + 32 * st + 8 * g);
            c = mma<F16>(ak, bq[st], c);                               // S^T[key = 16t + 4g + r][query = fr]
        }
        if (t == ntile - 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) c[r] = (16 * t + 4 * g + r) < S ? c[r] : -INFINITY;
        }
        mx = fmaxf(fmaxf(fmaxf(c[0], c[1]), fmaxf(c[2], c[3])), mx);          // max of the RAW scores: the (positive) scale is folded into
        s[t] = c;                                                             // the exponent's FMA below, one instruction per score less
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mxs = mx * scale2;
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < CA_MAXT; ++t) {
        if (t >= ntile) break;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = fast_exp2(fmaf(s[t][r], scale2, -mxs));
            s[t][r] = p;
            sum += p;
        }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float nrm = pscale / sum;
    // P V: k-step u covers key tiles 2u and 2u+1; lane (query fr, group g) supplies keys {32u+4g+r} U {32u+16+4g+r}
#pragma unroll
    for (int u = 0; u < CA_MAXT / 2; ++u) {
        if (2 * u >= ntile) break;
        const bool two = 2 * u + 1 < ntile;
        const f32x4 lo4 = s[2 * u], hi4 = two ? s[2 * u + 1] : f32x4{0.f, 0.f, 0.f, 0.f};
        const ca_u32x4 pw{pack2<F16>(lo4[0] * nrm, lo4[1] * nrm), pack2<F16>(lo4[2] * nrm, lo4[3] * nrm),
                          pack2<F16>(hi4[0] * nrm, hi4[1] * nrm), pack2<F16>(hi4[2] * nrm, hi4[3] * nrm)};
        const bf16x8 pa = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
        for (int dn = 0; dn < 4; ++dn) {
            const unsigned short* vp = Vt + (16 * dn + fr) * ldv + 32 * u + 4 * g;
            const uint2 lo = *reinterpret_cast<const uint2*>(vp);
            const uint2 hi = *reinterpret_cast<const uint2*>(vp + 16);
            const bf16x8 bv = __builtin_bit_cast(bf16x8, (ca_u32x4{lo.x, lo.y, hi.x, hi.y}));
            acc[dn] = mma<F16>(bv, pa, acc[dn]);                       // O^T[d = 16dn + 4g + r][query = fr]: 4 consecutive d per lane
        }
    }
}

template <bool F16>
__global__ __launch_bounds__(256) void cross_attn_kernel(const ca_args a) {
    EEG_LDS_BASE(unsigned short, lds);
    const int nt = (a.S + 15) / 16, nt_ip = (a.S_ip + 15) / 16;
    const int ldv = ca_ldv(nt), ldv_ip = ca_ldv(nt_ip);
    unsigned short* Ks = lds;                                    // [nt*16][CA_KLD]
    unsigned short* Vt = Ks + nt * 16 * CA_KLD;                  // [64][ldv]
    unsigned short* Kip = Vt + CA_D * ldv;                       // [nt_ip*16][CA_KLD]
    unsigned short* Vip = Kip + nt_ip * 16 * CA_KLD;             // [64][ldv_ip]
    const int b = blockIdx.z, h = blockIdx.y;
    const int C = a.heads * CA_D;
    const long long rs = C;
    {
        kv_stage rt, ri;
        kv_issue(rt, ldv, a.k + ((long long)b * a.S) * rs + h * CA_D, a.v + ((long long)b * a.S) * rs + h * CA_D, a.S, nt * 16, rs);
        if (nt_ip > 0)
            kv_issue(ri, ldv_ip, a.k_ip + ((long long)b * a.S_ip) * rs + h * CA_D, a.v_ip + ((long long)b * a.S_ip) * rs + h * CA_D, a.S_ip, nt_ip * 16, rs);
        kv_commit(rt, Ks, Vt, ldv, nt * 16);
        if (nt_ip > 0) kv_commit(ri, Kip, Vip, ldv_ip, nt_ip * 16);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, g = lane >> 4;
    for (int it = 0; it < CA_QB / 64; ++it) {
        const int q0 = blockIdx.x * CA_QB + (it * 4 + wave) * 16;
        if (q0 >= a.HW) break;                                   // wave-uniform
        // Q as the B operand of S^T = K Q^T: lane (query fr, group g) holds Q[q][32 st + 8g .. +7]
        bf16x8 bq[2];
        const int qrow = q0 + fr;
        const unsigned short* qp = a.q + ((long long)b * a.HW + (qrow < a.HW ? qrow : a.HW - 1)) * rs + h * CA_D + 8 * g;
        bq[0] = *reinterpret_cast<const bf16x8*>(qp);
        bq[1] = *reinterpret_cast<const bf16x8*>(qp + 32);
        f32x4 acc[4];
#pragma unroll
        for (int dn = 0; dn < 4; ++dn) acc[dn] = f32x4{0.f, 0.f, 0.f, 0.f};
        branch<F16>(Ks, Vt, ldv, a.S, nt, bq, a.scale, 1.0f, acc, lane);
        if (nt_ip > 0) branch<F16>(Kip, Vip, ldv_ip, a.S_ip, nt_ip, bq, a.scale, a.ip_scale, acc, lane);
        if (qrow < a.HW) {                                       // the output tile was formed transposed: 8-byte stores of 4 consecutive d
            unsigned short* op = a.out + ((long long)b * a.HW + qrow) * rs + h * CA_D + 4 * g;
#pragma unroll
            for (int dn = 0; dn < 4; ++dn) {
                uint2 w;
                w.x = pack2<F16>(acc[dn][0], acc[dn][1]);
                w.y = pack2<F16>(acc[dn][2], acc[dn][3]);
                *reinterpret_cast<uint2*>(op + 16 * dn) = w;
            }
        }
    }
}

// ---- SDXL's own shape (77 text tokens = 5 key tiles, 0 or 4 image tokens = 0 / 1 tile): K and V^T fragments live in REGISTERS -----------
// A wave walks 8 query tiles against the same keys.  Re-reading the K / V^T fragments from LDS for every tile cost 28 LDS instructions and
// their address arithmetic per tile on top of a VALU-heavy softmax (wave64 VALU issues over 4 cycles: ~400 instructions per tile were
// ~27 us of the 66 us launch, the MFMAs 7.5 us, the HBM roofline 28 us).  With the tile counts as template parameters the fragments are
// loaded once per wave (112 VGPRs) and the tile loop touches LDS not at all.
template <int NT>
struct kv_frag {
    bf16x8 k[NT > 0 ? NT : 1][2];
    bf16x8 v[NT > 0 ? (NT + 1) / 2 : 1][4];
};

template <int NT, bool VREG>
__device__ __forceinline__ void load_frag(kv_frag<NT>& f, const unsigned short* Ks, const unsigned short* Vt, int ldv, int lane) {
    const int fr = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int st = 0; st < 2; ++st) f.k[t][st] = *reinterpret_cast<const bf16x8*>(Ks + (16 * t + fr) * CA_KLD + 32 * st + 8 * g);
    if (!VREG) return;
#pragma unroll
    for (int u = 0; u < (NT + 1) / 2; ++u)
#pragma unroll
        for (int dn = 0; dn < 4; ++dn) {
            const unsigned short* vp = Vt + (16 * dn + fr) * ldv + 32 * u + 4 * g;
            const uint2 lo = *reinterpret_cast<const uint2*>(vp);
            const uint2 hi = *reinterpret_cast<const uint2*>(vp + 16);
            f.v[u][dn] = __builtin_bit_cast(bf16x8, (ca_u32x4{lo.x, lo.y, hi.x, hi.y}));
        }
}

// same arithmetic, in the same order, as branch() -- the two are interchangeable bit for bit.  TWO query tiles advance in lockstep through
// every phase (scores, max, exp, sum, P V): the phases of one tile are a dependent chain through MFMA results and cross-lane shuffles, and
// at 2 waves per SIMD nothing else hides those latencies.
template <bool F16, int NT, bool VREG>
__device__ __forceinline__ void branch_reg2(const kv_frag<NT>& f, const unsigned short* Vt, int ldv, int S, const bf16x8 (&bq)[2][2], float scale2,
                                            float pscale, f32x4 (&acc)[2][4], int lane) {
    const int g = lane >> 4, fr = lane & 15;
    f32x4 s[2][NT];
    float mx[2] = {-INFINITY, -INFINITY};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 2; ++st) c = mma<F16>(f.k[t][st], bq[p][st], c);
            if (t == NT - 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) c[r] = (16 * t + 4 * g + r) < S ? c[r] : -INFINITY;
            }
            mx[p] = fmaxf(fmaxf(fmaxf(c[0], c[1]), fmaxf(c[2], c[3])), mx[p]);
            s[p][t] = c;
        }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) mx[p] = fmaxf(mx[p], __shfl_xor(mx[p], 16, 64));
#pragma unroll
    for (int p = 0; p < 2; ++p) mx[p] = fmaxf(mx[p], __shfl_xor(mx[p], 32, 64));
    const float mxs[2] = {mx[0] * scale2, mx[1] * scale2};
    float sum[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = fast_exp2(fmaf(s[p][t][r], scale2, -mxs[p]));
                s[p][t][r] = e;
                sum[p] += e;
            }
#pragma unroll
    for (int p = 0; p < 2; ++p) sum[p] += __shfl_xor(sum[p], 16, 64);
#pragma unroll
    for (int p = 0; p < 2; ++p) sum[p] += __shfl_xor(sum[p], 32, 64);
    const float nrm[2] = {pscale / sum[0], pscale / sum[1]};
#pragma unroll
    for (int u = 0; u < (NT + 1) / 2; ++u) {
        const bool two = 2 * u + 1 < NT;
        bf16x8 pa[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const f32x4 lo4 = s[p][2 * u], hi4 = two ? s[p][two ? 2 * u + 1 : 0] : f32x4{0.f, 0.f, 0.f, 0.f};
            const ca_u32x4 pw{pack2<F16>(lo4[0] * nrm[p], lo4[1] * nrm[p]), pack2<F16>(lo4[2] * nrm[p], lo4[3] * nrm[p]),
                              pack2<F16>(hi4[0] * nrm[p], hi4[1] * nrm[p]), pack2<F16>(hi4[2] * nrm[p], hi4[3] * nrm[p])};
            pa[p] = __builtin_bit_cast(bf16x8, pw);
        }
#pragma unroll
        for (int dn = 0; dn < 4; ++dn) {
            bf16x8 bv;
            if (VREG) {
                bv = f.v[u][dn];
            } else {                                     // V^T fragment from LDS, shared by the two tiles
                const unsigned short* vp = Vt + (16 * dn + fr) * ldv + 32 * u + 4 * g;
                const uint2 lo = *reinterpret_cast<const uint2*>(vp);
                const uint2 hi = *reinterpret_cast<const uint2*>(vp + 16);
                bv = __builtin_bit_cast(bf16x8, (ca_u32x4{lo.x, lo.y, hi.x, hi.y}));
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) acc[p][dn] = mma<F16>(bv, pa[p], acc[p][dn]);
        }
    }
}

template <bool F16, int NT, int NT_IP, bool VREG>
__global__ __launch_bounds__(256, VREG ? 2 : 3) void cross_attn_reg_kernel(const ca_args a) {
    EEG_LDS_BASE(unsigned short, lds);
    const int ldv = ca_ldv(NT), ldv_ip = ca_ldv(NT_IP);
    unsigned short* Ks = lds;
    unsigned short* Vt = Ks + NT * 16 * CA_KLD;
    unsigned short* Kip = Vt + CA_D * ldv;
    unsigned short* Vip = Kip + NT_IP * 16 * CA_KLD;
    const int b = blockIdx.z, h = blockIdx.y;
    const long long rs = a.heads * CA_D;
    {
        kv_stage rt, ri;
        kv_issue(rt, ldv, a.k + ((long long)b * a.S) * rs + h * CA_D, a.v + ((long long)b * a.S) * rs + h * CA_D, a.S, NT * 16, rs);
        if (NT_IP > 0)
            kv_issue(ri, ldv_ip, a.k_ip + ((long long)b * a.S_ip) * rs + h * CA_D, a.v_ip + ((long long)b * a.S_ip) * rs + h * CA_D, a.S_ip, NT_IP * 16, rs);
        kv_commit(rt, Ks, Vt, ldv, NT * 16);
        if (NT_IP > 0) kv_commit(ri, Kip, Vip, ldv_ip, NT_IP * 16);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, g = lane >> 4;
    kv_frag<NT> ft;
    kv_frag<NT_IP> fi;
    load_frag<NT, VREG>(ft, Ks, Vt, ldv, lane);
    if (NT_IP > 0) load_frag<NT_IP, VREG>(fi, Kip, Vip, ldv_ip, lane);
    // a wave owns query tiles (it*4 + wave), it = 0..7, taken two at a time (it, it + 1); rows beyond HW are clamped on load, skipped on store
    auto q_ptr = [&](int it) {
        const int qrow = blockIdx.x * CA_QB + (it * 4 + wave) * 16 + fr;
        return a.q + ((long long)b * a.HW + (qrow < a.HW ? qrow : a.HW - 1)) * rs + h * CA_D + 8 * g;
    };
    bf16x8 nq[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        nq[p][0] = *reinterpret_cast<const bf16x8*>(q_ptr(p));
        nq[p][1] = *reinterpret_cast<const bf16x8*>(q_ptr(p) + 32);
    }
    for (int it = 0; it < CA_QB / 64; it += 2) {
        const int q0 = blockIdx.x * CA_QB + (it * 4 + wave) * 16;
        if (q0 >= a.HW) break;                                   // wave-uniform
        const bf16x8 bq[2][2] = {{nq[0][0], nq[0][1]}, {nq[1][0], nq[1][1]}};
        if (it + 2 < CA_QB / 64) {                               // the next pair's queries are in flight while this pair is computed
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                nq[p][0] = *reinterpret_cast<const bf16x8*>(q_ptr(it + 2 + p));
                nq[p][1] = *reinterpret_cast<const bf16x8*>(q_ptr(it + 2 + p) + 32);
            }
        }
        f32x4 acc[2][4];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int dn = 0; dn < 4; ++dn) acc[p][dn] = f32x4{0.f, 0.f, 0.f, 0.f};
        branch_reg2<F16, NT, VREG>(ft, Vt, ldv, a.S, bq, a.scale, 1.0f, acc, lane);
        if (NT_IP > 0)
            branch_reg2<F16, (NT_IP > 0 ? NT_IP : 1), VREG>(reinterpret_cast<const kv_frag<(NT_IP > 0 ? NT_IP : 1)>&>(fi), Vip, ldv_ip, a.S_ip, bq, a.scale,
                                                            a.ip_scale, acc, lane);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int qrow = q0 + 64 * p + fr;
            if (qrow < a.HW) {
                unsigned short* op = a.out + ((long long)b * a.HW + qrow) * rs + h * CA_D + 4 * g;
#pragma unroll
                for (int dn = 0; dn < 4; ++dn) {
                    uint2 w;
                    w.x = pack2<F16>(acc[p][dn][0], acc[p][dn][1]);
                    w.y = pack2<F16>(acc[p][dn][2], acc[p][dn][3]);
                    *reinterpret_cast<uint2*>(op + 16 * dn) = w;
                }
            }
        }
    }
}

}  // namespace eeg

using namespace eeg;

extern "C" int eegclip_cross_attn_fwd(const void* q, const void* k, const void* v, const void* k_ip, const void* v_ip, void* out, int B, int HW,
                                      int heads, int head_dim, int S, int S_ip, float ip_scale, int dtype, void* stream) {
    if (!q || !k || !v || !out || B < 1 || HW < 1 || heads < 1 || head_dim != CA_D || S < 1 || S > 16 * CA_MAXT || S_ip < 0 || S_ip > 16 * CA_MAXT)
        return EEGCLIP_EINVAL;
    if (S_ip > 0 && (!k_ip || !v_ip)) return EEGCLIP_EINVAL;
    if (dtype != EEGCLIP_DT_BF16 && dtype != EEGCLIP_DT_F16) return EEGCLIP_EINVAL;
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)k_ip | (uintptr_t)v_ip | (uintptr_t)out) & 15) != 0) return EEGCLIP_EALIGN;
    ca_args a{(const unsigned short*)q, (const unsigned short*)k, (const unsigned short*)v, (const unsigned short*)k_ip, (const unsigned short*)v_ip,
              (unsigned short*)out, B, HW, heads, S, S_ip, 0.125f * 1.44269504088896340736f, ip_scale};      // scale * log2(e): base-2 softmax
    const int nt = (S + 15) / 16, nt_ip = (S_ip + 15) / 16;
    const int ldv = ca_ldv(nt), ldv_ip = ca_ldv(nt_ip);
    const size_t lds = sizeof(unsigned short) * ((size_t)nt * 16 * CA_KLD + CA_D * ldv + (size_t)nt_ip * 16 * CA_KLD + (nt_ip ? CA_D * ldv_ip : 0));
    const dim3 grid((HW + CA_QB - 1) / CA_QB, heads, B);
    const bool allow_reg = true;
    if (allow_reg && nt == 5 && nt_ip <= 1) {            // SDXL: 77 text tokens, 0 / 4 image tokens -- K (and optionally V^T) fragments in registers
        const bool vreg = false;                              // (V^T fragments in registers as well: measured slower)
        const bool f16 = dtype == EEGCLIP_DT_F16;
#define EEG_CA_GO(F, I, V) EEG_LAUNCH((cross_attn_reg_kernel<F, 5, I, V>), grid, dim3(256), lds, stream, a)
        if (vreg) {
            if (nt_ip == 1) { if (f16) EEG_CA_GO(true, 1, true); else EEG_CA_GO(false, 1, true); }
            else            { if (f16) EEG_CA_GO(true, 0, true); else EEG_CA_GO(false, 0, true); }
        } else {
            if (nt_ip == 1) { if (f16) EEG_CA_GO(true, 1, false); else EEG_CA_GO(false, 1, false); }
            else            { if (f16) EEG_CA_GO(true, 0, false); else EEG_CA_GO(false, 0, false); }
        }
#undef EEG_CA_GO
        return (int)hipGetLastError();
    }
    if (dtype == EEGCLIP_DT_F16) EEG_LAUNCH((cross_attn_kernel<true>), grid, dim3(256), lds, stream, a);
    else                         EEG_LAUNCH((cross_attn_kernel<false>), grid, dim3(256), lds, stream, a);
    return (int)hipGetLastError();
}
