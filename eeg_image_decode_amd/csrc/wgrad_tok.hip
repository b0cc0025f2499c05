// Weight gradients of the transformer block from TOKEN-MAJOR bf16 planes:  dW[o][i] += sum_t dY[t][o] X[t][i], t = the 64 B token rows of a batch
// (models/subject_layers/Transformer_EncDec.py:48-49, SelfAttention_Family.py:199-213, Embed.py:146 differentiated w.r.t. the weights).
//
// Operands are what the fused per-sample kernels (csrc/token_block.hip, attention_x3.hip) hold in LDS anyway and now write out instead of the
// fp32 tensors the old weight-gradient GEMMs split again: per sample one 64 KB block = [hi | lo][64 tokens][256 channels] bf16 (the AP image
// of token_block, unswizzled; channel 255 of an X operand's hi plane is 1.0, so column 255 of dW is the bias gradient sum_t dY[t][o]).
//
// Both MFMA operands contract over TOKENS while the planes are channel-contiguous: a k-tile [32 tokens][128 channels] of each operand plane goes
// global -> LDS by LDS-DMA (16 bytes per lane, no VGPRs, no VALU: the round-3 GEMM spent ~1000 vector instructions per wave and k-tile on the
// fp32 -> hi | lo split + register transposition), and the fragments come out through ds_read_b64_tr_b16, the LDS transpose read of gfx950
// (lane mapping: csrc/attention_x3.hip).  LDS rows are 256 B (DMA deposits lane-linear); the 16-byte chunk index is XOR-ed with 2 (row & 7) on
// the DMA source address and on the read address: the 8 rows x 32 bytes a half-wave's transpose read touches then cover the 64 banks once.
//
//   wgrad_tok_kernel<WN>   workgroup = (problem, 128 x 128 output tile, K slice), 2 x WN waves, wave tile 64 x (128 / WN) as 16x16x32 MFMAs, three
//                          products per multiply-add (hi lo + lo hi + hi hi, fp32 accumulate: the arithmetic of every Linear of the step);
//                          4 LDS stages of 32 KB (A hi | A lo | B hi | B lo), counted vmcnt, the refill of tile kt + 3 issued between the MFMAs
//                          of tile kt (the pipeline of csrc/infonce_fused.hip).  Several problems per launch (the three gradients that become
//                          ready together are ONE launch); all tiles of a K slice run on one XCD (each operand byte enters one L2).
//   wgrad_tok_reduce_kernel  out[m][n] += sum_s slab[s][m][n] in slice order (bit-reproducible; the round-3 GEMM added 32 slices with fp32
//                          atomics: 19.8 MB of write traffic per launch for a 0.25 MB result and a scheduling-dependent sum).
#include "eeg_common.h"

#include <string.h>

#include <type_traits>

namespace eeg {

constexpr int WK_NS = 4;                                     // LDS stages
constexpr int WK_BK = 32;                                    // tokens per k-tile = one MFMA k-step
constexpr int WK_ROWB = 256;                                 // bytes per LDS row: 128 channels
constexpr int WK_TILE = WK_BK * WK_ROWB;                     // one operand-plane tile: 8 KB
constexpr int WK_STAGE = 4 * WK_TILE;                        // A hi | A lo | B hi | B lo
constexpr int WK_SAMPLE = 65536, WK_PLANE = 32768, WK_TOKB = 512;      // bytes: sample block, plane, token row of the global layout
constexpr int WK_MAXP = 4;

struct wk_problem {
    const unsigned char* a;                                  // dY planes
    const unsigned char* b;                                  // X planes
    float* slab;                                             // [slices][128 m_tiles][256] partial tiles, then [slices][128 m_tiles] bias partials
    long long a_group_stride;                                // bytes between 256-channel groups of A (dq | dk | dv)
    int m_tiles;                                             // 128-channel tiles of A: 2 per group
    int bias_mfma;                                           // bias gradient through an all-ones fragment (X has no spare column)
    int first_block;                                         // first workgroup of this problem in the launch
};
struct wk_table {
    wk_problem p[WK_MAXP];
    int n;
};

typedef short wk_s4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ wk_s4 wk_tr_read(const unsigned char* p) {
#if defined(EEG_EMU)
    const int lane = hipemu::cur->lane, g = lane >> 4, i = lane & 15;
    wk_s4 r;
    for (int j = 0; j < 4; ++j) {
        const unsigned long long src = hipemu::shfl_idx((unsigned long long)(uintptr_t)p, 16 * g + 4 * j + (i >> 2));
        r[j] = reinterpret_cast<const short*>((uintptr_t)src)[i & 3];
    }
    return r;
#else
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wk_s4*)(p));
#endif
}

template <int WN>
__global__ __launch_bounds__(128 * WN) void wgrad_tok_kernel(const wk_table tb, int ktiles_all, int slices) {
    constexpr int NWAVE = 2 * WN, NT = 8 / WN;               // n-tiles of 16 per wave
    constexpr int IPW = 32 / NWAVE;                          // DMA instructions (1 KB each) per wave and k-tile
    constexpr int MPT = 4 * NT * 3;                          // MFMAs per wave and k-tile
    EEG_LDS_BASE(unsigned char, lds);
    const int t = threadIdx.x, lane = t & 63, wave = wave_uniform(t >> 6), wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, g = lane >> 4;

    int prob = 0;
#pragma unroll
    for (int p = 1; p < WK_MAXP; ++p)
        if (p < tb.n && (int)blockIdx.x >= tb.p[p].first_block) prob = p;
    const wk_problem& P = tb.p[prob];
    const int tiles = 2 * P.m_tiles;
    const int local = (int)blockIdx.x - P.first_block;
    int slice, tile;
    if ((slices & 7) == 0) {                                 // workgroup b runs on XCD b % 8 (observed dispatch; speed only): a slice's tiles share one L2
        const int xcd = local & 7, j = local >> 3;
        slice = xcd + 8 * (j / tiles);
        tile = j % tiles;
    } else {
        slice = local / tiles;
        tile = local - slice * tiles;
    }
    const int tm = tile >> 1, tn = tile & 1;
    const int kt0 = (int)((long long)slice * ktiles_all / slices), kt1 = (int)((long long)(slice + 1) * ktiles_all / slices);
    const int nk = kt1 - kt0;
    const unsigned char* const abase = P.a + (long long)(tm >> 1) * P.a_group_stride + (tm & 1) * 256;
    const unsigned char* const bbase = P.b + tn * 256;
    const bool bias = P.bias_mfma && tn == 0 && wn == 0;     // (wave-uniform)

    // ---- DMA roles: instruction q = wave IPW + i of a k-tile deposits rows 4 (q & 7) .. + 3 of operand-plane tile q >> 3 (A hi, A lo, B hi, B lo)
    int doff[IPW];                                           // lane's source offset within the k-tile's 16 KB token block of its plane
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int q = wave * IPW + i, row = 4 * (q & 7) + (lane >> 4), pos = lane & 15;
        doff[i] = row * WK_TOKB + ((pos ^ ((row & 7) << 1)) << 4);
    }
    auto issue_one = [&](int kt, int i) {                    // kt relative to kt0
        const int q = wave * IPW + i, o = q >> 3;
        const int k = kt0 + kt;
        const long long koff = (long long)(k >> 1) * WK_SAMPLE + (k & 1) * (WK_BK * WK_TOKB) + (o & 1) * WK_PLANE;
        const unsigned char* src = (o < 2 ? abase : bbase) + koff + doff[i];
        lds_dma16(lds + (kt % WK_NS) * WK_STAGE + q * 1024, src);
    };

    // ---- fragment offsets (k-tile independent): lane (fr, g) addresses row 4 g + (fr >> 2) (+ 16), channels c0 + 4 (fr & 3) .. + 3 and receives tokens
    //      4 g .. 4 g + 3 (+ 16) of channel c0 + fr: k slots 0-3 / 4-7 of a 16x16x32 step, the same assignment on both operands
    int foa[4], fob[NT];
    {
        const int row = 4 * g + (fr >> 2), sw = (row & 7) << 1, sub = 8 * (fr & 1), cp = (fr & 3) >> 1;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) foa[mt] = row * WK_ROWB + (((8 * wm + 2 * mt + cp) ^ sw) << 4) + sub;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) fob[nt] = 2 * WK_TILE + row * WK_ROWB + (((2 * NT * wn + 2 * nt + cp) ^ sw) << 4) + sub;
    }
    auto frag = [&](const unsigned char* p) {
        const wk_s4 x = wk_tr_read(p), y = wk_tr_read(p + 16 * WK_ROWB);
        return bf16x8{x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
    };

    f32x4 acc[4][NT], bacc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        bacc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;     // bf16 1.0

#pragma unroll
    for (int p = 0; p < WK_NS - 1; ++p)
        if (p < nk) {
#pragma unroll
            for (int i = 0; i < IPW; ++i) issue_one(p, i);
        }
    // one k-tile; REFILL (compile-time: the steady state has no branches between its MFMAs) = tile kt + 3 exists and is requested between the MFMAs
    auto step = [&](int kt, auto refill_c) {
        constexpr bool REFILL = decltype(refill_c)::value;
        if (REFILL) wait_vmcnt<2 * IPW>();                   // tiles kt + 1, kt + 2 may stay in flight
        else {
            const int newer = nk - 1 - kt;                   // < 3 here
            if (newer >= 2) wait_vmcnt<2 * IPW>();
            else if (newer == 1) wait_vmcnt<IPW>();
            else wait_vmcnt<0>();
        }
        raw_barrier();                                       // tile kt has landed for every wave; the stage about to be refilled is drained
        const unsigned char* st = lds + (kt % WK_NS) * WK_STAGE;
        bf16x8 ah[4], al[4], bh[NT], bl[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bh[nt] = frag(st + fob[nt]);
            bl[nt] = frag(st + WK_TILE + fob[nt]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            ah[mt] = frag(st + foa[mt]);
            al[mt] = frag(st + WK_TILE + foa[mt]);
        }
        // product-major: the three MFMAs of one accumulator are 4 NT instructions apart; MFMA rows = X channels (n), columns = dY channels (m), so a
        // lane holds 4 CONSECUTIVE n of one m: 16-byte slab stores
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const bf16x8 bw = pr == 0 ? bl[nt] : bh[nt];
                    const bf16x8 aw = pr == 1 ? al[mt] : ah[mt];
                    acc[mt][nt] = mfma_bf16_16x16x32(bw, aw, acc[mt][nt]);      // D[n = .. + 4 g + r][m = .. + fr]
                    const int idx = (pr * 4 + mt) * NT + nt;
                    if (REFILL && ((idx + 1) * IPW) / MPT > (idx * IPW) / MPT) issue_one(kt + WK_NS - 1, (idx * IPW) / MPT);
                }
        if (bias) {                                          // column sums of the dY tile: every MFMA row of the product holds them
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                bacc[mt] = mfma_bf16_16x16x32(ones, al[mt], bacc[mt]);
                bacc[mt] = mfma_bf16_16x16x32(ones, ah[mt], bacc[mt]);
            }
        }
    };
    int kt = 0;
    for (; kt + WK_NS - 1 < nk; ++kt) step(kt, std::true_type{});
    for (; kt < nk; ++kt) step(kt, std::false_type{});

    // ---- partial tile -> this slice's slab (plain stores; the reduce kernel sums the slices in order)
    const int Mp = 128 * P.m_tiles;
    float* out = P.slab + (long long)slice * Mp * 256;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = 128 * tm + 64 * wm + 16 * mt + fr;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = 128 * tn + 16 * NT * wn + 16 * nt + 4 * g;
            *reinterpret_cast<f32x4*>(out + (long long)m * 256 + n) = acc[mt][nt];
        }
    }
    if (bias && g == 0) {
        float* bo = P.slab + (long long)slices * Mp * 256 + (long long)slice * Mp;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) bo[128 * tm + 64 * wm + 16 * mt + fr] = bacc[mt][0];
    }
}

struct wk_reduce_problem {
    const float* slab;
    float* out;
    float* bias_out;
    long long ldo;
    int M, N, Mp;                                            // rows / columns of out; slab rows
    int heads_m, heads_n;                                    // index i of out <-> slab index 256 (i / 248) + 64 ((i % 248) / 62) + (i % 62)
    int bias_mfma;
    int first;                                               // first thread of this problem
};
struct wk_reduce_table {
    wk_reduce_problem p[WK_MAXP];
    int n;
};

__device__ __forceinline__ int wk_slab_index(int i, int heads) {
    if (!heads) return i;
    const int grp = i / 248, r = i - 248 * grp, hd = r / 62;
    return 256 * grp + 64 * hd + (r - 62 * hd);
}

// one thread per output element, 16 slices of loads in flight (consecutive threads = consecutive columns of a slab row: coalesced)
__global__ __launch_bounds__(256) void wgrad_tok_reduce_kernel(const wk_reduce_table tb, int slices) {
    const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    int prob = 0;
#pragma unroll
    for (int p = 1; p < WK_MAXP; ++p)
        if (p < tb.n && i >= tb.p[p].first) prob = p;
    const wk_reduce_problem& P = tb.p[prob];
    const int e = i - P.first, total = P.M * P.N;
    const long long stride = (long long)P.Mp * 256;
    const float* p;
    long long step;
    float* dst;
    if (e < total) {
        const int m = e / P.N, n = e - m * P.N;
        p = P.slab + (long long)wk_slab_index(m, P.heads_m) * 256 + wk_slab_index(n, P.heads_n);
        step = stride;
        dst = P.out + (long long)m * P.ldo + n;
    } else if (P.bias_out && e < total + P.M) {
        const int m = e - total, sm = wk_slab_index(m, P.heads_m);
        if (P.bias_mfma) {
            p = P.slab + (long long)slices * stride + sm;
            step = P.Mp;
        } else {
            p = P.slab + (long long)sm * 256 + 255;
            step = stride;
        }
        dst = P.bias_out + m;
    } else
        return;
    float s = 0.f;
    int k = 0;
    for (; k + 16 <= slices; k += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = p[(long long)(k + u) * step];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; k < slices; ++k) s += p[(long long)k * step];
    *dst += s;
}

}  // namespace eeg

using namespace eeg;

static int wk_check(const eegclip_wgrad_tok_problem* p, int n_prob, int B) {
    if (!p || n_prob < 1 || n_prob > WK_MAXP || B < 1) return EEGCLIP_EINVAL;
    for (int i = 0; i < n_prob; ++i) {
        const eegclip_wgrad_tok_problem& q = p[i];
        if (!q.a || !q.b || !q.out || q.m_groups < 1 || q.m_groups > 3 || q.M < 1 || q.N < 1 || q.ldo < q.N) return EEGCLIP_EINVAL;
        if (q.M > (q.heads_m ? 248 : 256) * q.m_groups || q.N > (q.heads_n ? 248 : 256)) return EEGCLIP_EINVAL;
        if (q.bias_out && !q.bias_mfma && q.N > (q.heads_n ? 248 : 255)) return EEGCLIP_EINVAL;      // column 255 must be the ones column, not data
        if (q.m_groups > 1 && q.a_group_stride < (long long)B * WK_SAMPLE) return EEGCLIP_EINVAL;
        if ((reinterpret_cast<uintptr_t>(q.a) | reinterpret_cast<uintptr_t>(q.b) | (uintptr_t)q.a_group_stride) & 15u) return EEGCLIP_EALIGN;
    }
    return 0;
}

extern "C" int eegclip_wgrad_tok_slices(int total_m_groups, int B) {
    if (total_m_groups < 1 || B < 1) return 0;
    const int tiles = 4 * total_m_groups, kt = 2 * B;
    int s = 256 / tiles;                                         // one workgroup (128 KB of LDS) per CU
    if (s > kt / 4) s = kt / 4;                                  // at least 4 k-tiles per workgroup
    if (s >= 8) s = s / 8 * 8;                                   // a multiple of 8: the XCD-aware order of the kernel
    return s < 1 ? 1 : s;
}

extern "C" long long eegclip_wgrad_tok_workspace_floats(const eegclip_wgrad_tok_problem* p, int n_prob, int B, int slices) {
    if (wk_check(p, n_prob, B) || slices < 1 || slices > 2 * B) return 0;
    long long total = 0;
    for (int i = 0; i < n_prob; ++i) total += (long long)slices * (256LL * p[i].m_groups * 256 + 256LL * p[i].m_groups);
    return total;
}

static int wk_tables(const eegclip_wgrad_tok_problem* p, int n_prob, int B, int slices, float* workspace, wk_table& tb, wk_reduce_table& rt, int& blocks,
                     int& threads) {
    const int rc = wk_check(p, n_prob, B);
    if (rc) return rc;
    if (!workspace || slices < 1 || slices > 2 * B) return EEGCLIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(workspace) & 15u) return EEGCLIP_EALIGN;
    memset(&tb, 0, sizeof(tb));
    memset(&rt, 0, sizeof(rt));
    tb.n = rt.n = n_prob;
    blocks = threads = 0;
    float* ws = workspace;
    for (int i = 0; i < n_prob; ++i) {
        const eegclip_wgrad_tok_problem& q = p[i];
        const int Mp = 256 * q.m_groups;
        tb.p[i] = wk_problem{static_cast<const unsigned char*>(q.a), static_cast<const unsigned char*>(q.b), ws, q.a_group_stride, 2 * q.m_groups,
                             (q.bias_out && q.bias_mfma) ? 1 : 0, blocks};
        rt.p[i] = wk_reduce_problem{ws, q.out, q.bias_out, q.ldo, q.M, q.N, Mp, q.heads_m ? 1 : 0, q.heads_n ? 1 : 0, q.bias_mfma ? 1 : 0, threads};
        blocks += 4 * q.m_groups * slices;
        threads += (q.M * q.N + (q.bias_out ? q.M : 0) + 255) / 256 * 256;
        ws += (long long)slices * ((long long)Mp * 256 + Mp);
    }
    return 0;
}

// the partial tiles of every K slice -> workspace (ONE kernel) ...
extern "C" int eegclip_wgrad_tok(const eegclip_wgrad_tok_problem* p, int n_prob, int B, int slices, float* workspace, int variant, void* stream) {
    wk_table tb;
    wk_reduce_table rt;
    int blocks, threads;
    const int rc = wk_tables(p, n_prob, B, slices, workspace, tb, rt, blocks, threads);
    if (rc) return rc;
    if (variant == 1) EEG_LAUNCH(wgrad_tok_kernel<2>, dim3((unsigned)blocks), dim3(256), WK_NS * WK_STAGE, stream, tb, 2 * B, slices);
    else EEG_LAUNCH(wgrad_tok_kernel<4>, dim3((unsigned)blocks), dim3(512), WK_NS * WK_STAGE, stream, tb, 2 * B, slices);
    return (int)hipGetLastError();
}

// ... and out += their sum in slice order (ONE kernel); same arguments
extern "C" int eegclip_wgrad_tok_reduce(const eegclip_wgrad_tok_problem* p, int n_prob, int B, int slices, float* workspace, void* stream) {
    wk_table tb;
    wk_reduce_table rt;
    int blocks, threads;
    const int rc = wk_tables(p, n_prob, B, slices, workspace, tb, rt, blocks, threads);
    if (rc) return rc;
    EEG_LAUNCH(wgrad_tok_reduce_kernel, dim3((unsigned)(threads / 256)), dim3(256), 0, stream, rt, slices);
    return (int)hipGetLastError();
}

// fp32 [rows = 64 B][cols] (row stride ld) -> the token-major plane layout above ([B][hi | lo][64][256], columns >= cols zero, `ones`: hi[.][255] = 1.0;
// heads: column c = 62 head + d goes to channel 64 head + d).  For operands no fused kernel produces (tests, the unfused fallback plans).
namespace eeg {
__global__ __launch_bounds__(256) void tok_planes_from_f32_kernel(const float* __restrict__ src, long long ld, int rows, int cols, int heads, int ones,
                                                                  unsigned short* __restrict__ dst) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;       // (row, channel pair)
    if (i >= (long long)rows * 128) return;
    const int row = (int)(i >> 7), c = 2 * (int)(i & 127);
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int ch = c + e;
        int col = ch;
        bool ok = ch < cols;
        if (heads) {
            const int hd = ch >> 6, d = ch & 63;
            col = 62 * hd + d;
            ok = d < 62 && col < cols;
        }
        v[e] = ok ? src[(long long)row * ld + col] : 0.f;
    }
    unsigned hi = x3_pack2(v[0], v[1]);
    const float r0 = v[0] - __uint_as_float(hi << 16), r1 = v[1] - __uint_as_float(hi & 0xffff0000u);
    const unsigned lo = x3_pack2(r0, r1);
    if (ones && c == 254) hi = (hi & 0xffffu) | 0x3F800000u;
    unsigned char* base = reinterpret_cast<unsigned char*>(dst) + (long long)(row >> 6) * WK_SAMPLE + (row & 63) * WK_TOKB + 2 * c;
    *reinterpret_cast<unsigned*>(base) = hi;
    *reinterpret_cast<unsigned*>(base + WK_PLANE) = lo;
}
}  // namespace eeg

extern "C" int eegclip_tok_planes_from_f32(const float* src, long long ld, int rows, int cols, int heads, int ones, void* dst, void* stream) {
    if (!src || !dst || rows < 64 || (rows & 63) || cols < 1 || cols > 256 || ld < cols || (heads && cols > 248) || (ones && !heads && cols > 255)) return EEGCLIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(dst) & 15u) return EEGCLIP_EALIGN;
    const long long n = (long long)rows * 128;
    EEG_LAUNCH(tok_planes_from_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, ld, rows, cols, heads, ones,
               static_cast<unsigned short*>(dst));
    return (int)hipGetLastError();
}
