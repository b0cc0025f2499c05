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

// stage K rows [S][64] (padded stride) and V transposed [64][ldv] for one (b, head); rows >= S are zero
__device__ __forceinline__ void stage_kv(unsigned short* Ks, unsigned short* Vt, int ldv, const unsigned short* k, const unsigned short* v,
                                         int S, int S_pad, long long row_stride) {
    const int t = threadIdx.x;
    for (int i = t; i < S_pad * (CA_D / 8); i += blockDim.x) {      // 16-byte pieces of K rows
        const int r = i / (CA_D / 8), c8 = i % (CA_D / 8);
        uint4 val = make_uint4(0, 0, 0, 0);
        if (r < S) val = *reinterpret_cast<const uint4*>(k + r * row_stride + c8 * 8);
        *reinterpret_cast<uint4*>(Ks + r * CA_KLD + c8 * 8) = val;
    }
    for (int i = t; i < ldv * (CA_D / 8); i += blockDim.x) {        // V^T[d][key], zero padded along keys: 16-byte reads of V rows, transposed stores
        const int key = i / (CA_D / 8), d8 = 8 * (i % (CA_D / 8));
        uint4 val = make_uint4(0, 0, 0, 0);
        if (key < S) val = *reinterpret_cast<const uint4*>(v + key * row_stride + d8);
        const unsigned w[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Vt[(d8 + 2 * e) * ldv + key] = (unsigned short)(w[e] & 0xffffu);
            Vt[(d8 + 2 * e + 1) * ldv + key] = (unsigned short)(w[e] >> 16);
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
            for (int r = 0; r < 4; ++r) c[r] = (16 * t + 4 * g + r) < S ? c[r] * scale2 : -INFINITY;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) c[r] *= scale2;
        }
        mx = fmaxf(fmaxf(fmaxf(c[0], c[1]), fmaxf(c[2], c[3])), mx);
        s[t] = c;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < CA_MAXT; ++t) {
        if (t >= ntile) break;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = fast_exp2(s[t][r] - mx);
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
    stage_kv(Ks, Vt, ldv, a.k + ((long long)b * a.S) * rs + h * CA_D, a.v + ((long long)b * a.S) * rs + h * CA_D, a.S, nt * 16, rs);
    if (nt_ip > 0)
        stage_kv(Kip, Vip, ldv_ip, a.k_ip + ((long long)b * a.S_ip) * rs + h * CA_D, a.v_ip + ((long long)b * a.S_ip) * rs + h * CA_D, a.S_ip,
                 nt_ip * 16, rs);
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
    if (dtype == EEGCLIP_DT_F16) EEG_LAUNCH((cross_attn_kernel<true>), grid, dim3(256), lds, stream, a);
    else                         EEG_LAUNCH((cross_attn_kernel<false>), grid, dim3(256), lds, stream, a);
    return (int)hipGetLastError();
}
