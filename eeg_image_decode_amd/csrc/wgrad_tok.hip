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
//                          4 LDS stages of 32 KB (A hi | A lo | B hi | B lo), counted vmcnt; the fragment reads of tile kt + 1 and the refill of
//                          tile kt's stage (tile kt + 4) are issued under the MFMAs of tile kt.  Several problems per launch (the three gradients that become
//                          ready together are ONE launch); all tiles of a K slice run on one XCD (each operand byte enters one L2).
//   wgrad_tok_reduce_kernel  out[m][n] += sum_s slab[s][m][n] in slice order (bit-reproducible; the round-3 GEMM added 32 slices with fp32
//                          atomics: 19.8 MB of write traffic per launch for a 0.25 MB result and a scheduling-dependent sum).
// Template switches of the kernel: IDX = the contraction of a problem runs over a sample range of an index list (the joint-subject value embedding: one
// problem per subject); L2D = plain row-major planes [rows][ld] and direct accumulation into the gradient (the diffusion prior, eegclip_wgrad_planes);
// SPEC = four extra producer waves issue all LDS-DMA (variants 2 / 3 of eegclip_wgrad_tok: measured within 5 % of the default -- the kernel is bound by
// operand delivery, 3.4 TB/s on the 192 CUs its 12 tiles x 16 XCD-aligned slices occupy, not by its wave layout).
#include "eeg_common.h"

#include <math.h>
#include <string.h>

#include <type_traits>

namespace eeg {

constexpr int WK_BK = 32;                                    // tokens per k-tile = one MFMA k-step
constexpr int WK_ROWB = 256;                                 // bytes per LDS row: 128 channels
constexpr int WK_TILE = WK_BK * WK_ROWB;                     // one operand-plane tile: 8 KB
constexpr int WK_STAGE = 4 * WK_TILE;                        // A hi | A lo | B hi | B lo
constexpr int WK_SAMPLE = 65536, WK_PLANE = 32768, WK_TOKB = 512;      // bytes: sample block, plane, token row of the global layout
constexpr int WK_MAXP = 24;                                // problems per launch of the plain-planes form (the diffusion prior's 22 weight gradients are ONE launch)
constexpr int WK_MAXP_TOK = 12;                            // ... of the token-plane form (the joint-subject value embedding: one problem per subject of the batch)

struct wk_problem {
    const unsigned char* a;                                  // dY planes
    const unsigned char* b;                                  // X planes
    float* slab;                                             // [slices][128 m_tiles][256] partial tiles, then [slices][128 m_tiles] bias partials
    long long a_group_stride;                                // bytes between 256-channel groups of A (dq | dk | dv)
    int m_tiles;                                             // 128-channel tiles of A: 2 per group
    int bias_mfma;                                           // bias gradient through an all-ones fragment (X has no spare column)
    int first_block;                                         // first workgroup of this problem in the launch
    int sample0, ktiles;                                     // the contraction runs over samples sample0 .. of the (indexed) batch: 2 k-tiles per sample
    // PLAIN 2-D planes (eegclip_wgrad_planes: [rows][ld] bf16, hi and lo planes apart) -- the template flag L2D selects these fields:
    long long a_lo, b_lo;                                    // bytes from the hi to the lo plane
    int a_tok, b_tok;                                        // bytes per k-row of A / B
    int n_tiles, slices;                                     // 128-column tiles of the output; K slices of THIS problem
    float* out;                                              // (M, N) row stride ldo, accumulated into directly (atomically when slices > 1)
    float* bias_out;
    long long ldo;
    int M, N;
};
struct wk_table {
    wk_problem p[WK_MAXP];
    const int* index;                                        // IDX: position -> sample of the batch (one subject's samples are a range of positions)
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

// SPEC (round 4): four extra PRODUCER waves issue every LDS-DMA instruction and hold the counted vmcnt waits; the MFMA waves issue no vector-memory
// instruction in the k-loop.  A wave that does both is in order: it stalls in each LDS-DMA issue while the CU's vector-memory path (~64 B/clk) works
// through its queue, with its MFMAs unissued behind it -- the two costs add (tools/micro/tile_chain.hip; the InfoNCE tile kernel).
template <int WN, int WK_NS, bool IDX, bool L2D = false, bool SPEC = false>
__global__ __launch_bounds__(128 * WN + (SPEC ? 256 : 0)) void wgrad_tok_kernel(const wk_table tb, int slices_all) {
    constexpr int NWAVE = 2 * WN, NT = 8 / WN;               // MFMA waves; n-tiles of 16 per wave
    constexpr int NDW = SPEC ? 4 : NWAVE;                    // waves that issue LDS-DMA
    constexpr int IPW = 32 / NDW;                            // DMA instructions (1 KB each) per DMA wave and k-tile
    constexpr int MPT = 4 * NT * 3;                          // MFMAs per wave and k-tile
    EEG_LDS_BASE(unsigned char, lds);
    const int t = threadIdx.x, lane = t & 63, wave = wave_uniform(t >> 6), wm = wave / WN, wn = wave % WN;
    const bool producer = SPEC && wave >= NWAVE;             // (wave-uniform)
    const int dw = SPEC ? (wave - NWAVE) & 3 : wave;         // index among the DMA waves
    const int fr = lane & 15, g = lane >> 4;

    int prob = 0;
#pragma unroll
    for (int p = 1; p < WK_MAXP; ++p)
        if (p < tb.n && (int)blockIdx.x >= tb.p[p].first_block) prob = p;
    const wk_problem& P = tb.p[prob];
    const int n_tiles = L2D ? P.n_tiles : 2;
    const int tiles = n_tiles * P.m_tiles;
    const int slices = L2D ? P.slices : slices_all;
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
    const int tm = tile / n_tiles, tn = tile - tm * n_tiles;
    const int ktiles_all = P.ktiles;
    const int kt0 = (int)((long long)slice * ktiles_all / slices), kt1 = (int)((long long)(slice + 1) * ktiles_all / slices);
    const int sample0 = P.sample0;
    const int* const index = tb.index;
    const int nk = kt1 - kt0;
    const unsigned char* const abase = L2D ? P.a + tm * 256 : P.a + (long long)(tm >> 1) * P.a_group_stride + (tm & 1) * 256;
    const unsigned char* const bbase = P.b + tn * 256;
    const bool bias = P.bias_mfma && tn == 0 && wn == 0;     // (wave-uniform)

    // ---- DMA roles: instruction q = wave IPW + i of a k-tile deposits rows 4 (q & 7) .. + 3 of operand-plane tile q >> 3 (A hi, A lo, B hi, B lo)
    int doff[IPW];                                           // lane's source offset within the k-tile's 16 KB token block of its plane
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int q = dw * IPW + i, row = 4 * (q & 7) + (lane >> 4), pos = lane & 15;
        doff[i] = row * (L2D ? ((q >> 3) < 2 ? P.a_tok : P.b_tok) : WK_TOKB) + ((pos ^ ((row & 7) << 1)) << 4);
    }
    auto issue_one = [&](int kt, int i) {                    // kt relative to kt0
        const int q = dw * IPW + i, o = q >> 3;
        const int k = kt0 + kt;
        const int smp = IDX ? index[sample0 + (k >> 1)] : sample0 + (k >> 1);      // (wave-uniform: a scalar load)
        long long koff;
        if (L2D) koff = (long long)k * WK_BK * (o < 2 ? P.a_tok : P.b_tok) + ((o & 1) ? (o < 2 ? P.a_lo : P.b_lo) : 0);
        else koff = (long long)smp * WK_SAMPLE + (k & 1) * (WK_BK * WK_TOKB) + (o & 1) * WK_PLANE;
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

    // ---- pipeline: all 4 stages are requested up front; iteration kt: [tile kt + 1 landed -> barrier] -> fragment reads of tile kt + 1 into the OTHER
    //      register set -> MFMAs of tile kt (fragments read one iteration ago), the refill of tile kt's stage with tile kt + 4 issued between them.
    //      The LDS reads of a k-tile run under the previous tile's MFMAs: with the reads and the MFMAs of ONE tile back to back behind each barrier
    //      (first version) every wave of the workgroup read at once and nobody computed -- 0.92 us per k-tile for 0.32 us of MFMA time.
    bf16x8 ah[2][4], al[2][4], bh[2][NT], bl[2][NT];
    auto read_frags = [&](int kt, auto set_c) {
        constexpr int S = decltype(set_c)::value;
        const unsigned char* st = lds + (kt % WK_NS) * WK_STAGE;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bh[S][nt] = frag(st + fob[nt]);
            bl[S][nt] = frag(st + WK_TILE + fob[nt]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            ah[S][mt] = frag(st + foa[mt]);
            al[S][mt] = frag(st + WK_TILE + foa[mt]);
        }
    };
    auto wait_tiles = [&](int n) {                               // at most n k-tiles of this wave's DMA still in flight
        if (n >= 4) wait_vmcnt<4 * IPW>();
        else if (n == 3) wait_vmcnt<3 * IPW>();
        else if (n == 2) wait_vmcnt<2 * IPW>();
        else if (n == 1) wait_vmcnt<IPW>();
        else wait_vmcnt<0>();
    };
    auto mfmas = [&](int kt, auto set_c, auto refill_c) {
        constexpr int S = decltype(set_c)::value;
        constexpr bool REFILL = decltype(refill_c)::value;
        // product-major: the three MFMAs of one accumulator are 4 NT instructions apart; MFMA rows = X channels (n), columns = dY channels (m), so a
        // lane holds 4 CONSECUTIVE n of one m: 16-byte slab stores
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const bf16x8 bw = pr == 0 ? bl[S][nt] : bh[S][nt];
                    const bf16x8 aw = pr == 1 ? al[S][mt] : ah[S][mt];
                    acc[mt][nt] = mfma_bf16_16x16x32(bw, aw, acc[mt][nt]);      // D[n = .. + 4 g + r][m = .. + fr]
                    const int idx = (pr * 4 + mt) * NT + nt;
                    if (!SPEC && REFILL && ((idx + 1) * IPW) / MPT > (idx * IPW) / MPT) issue_one(kt + WK_NS, (idx * IPW) / MPT);
                }
        if (bias) {                                          // column sums of the dY tile: every MFMA row of the product holds them
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                bacc[mt] = mfma_bf16_16x16x32(ones, al[S][mt], bacc[mt]);
                bacc[mt] = mfma_bf16_16x16x32(ones, ah[S][mt], bacc[mt]);
            }
        }
    };
    auto step = [&](int kt, auto set_c, auto refill_c) {
        constexpr int S = decltype(set_c)::value;
        constexpr bool REFILL = decltype(refill_c)::value;       // compile-time: the steady state has no branches between its MFMAs
        const bool next = REFILL || kt + 1 < nk;
        if (next) {
            if (REFILL) wait_vmcnt<(WK_NS - 2) * IPW>();         // tiles kt + 2 .. kt + NS - 1 may stay in flight
            else wait_tiles(nk - 2 - kt);                        // tiles requested after kt + 1 (< NS - 1 here: nothing was refilled since)
            raw_barrier();                                       // tile kt + 1 is visible to every wave; every wave's reads of tile kt are complete
            read_frags(kt + 1, std::integral_constant<int, 1 - S>{});
        }
        mfmas(kt, set_c, refill_c);
    };
    if (SPEC) {
        // ---- producer / consumer form: ONE fragment register set (two MFMA waves per SIMD cover each other's LDS reads; 12 waves leave 168 registers)
        //      per k-tile: [producers: tile kt has landed] barrier [producers: refill the stage of tile kt - 1] fragment reads of tile kt, its MFMAs
        if (producer) {
#pragma unroll
            for (int p = 0; p < WK_NS; ++p)
                if (p < nk) {
#pragma unroll
                    for (int i = 0; i < IPW; ++i) issue_one(p, i);
                }
            for (int kt = 0; kt < nk; ++kt) {
                const int fly = nk - 1 - kt < WK_NS - 2 ? nk - 1 - kt : WK_NS - 2;      // tiles kt + 1 .. kt + NS - 2 may stay in flight (kt - 1 + NS comes below)
                wait_tiles(kt == 0 ? (nk - 1 < WK_NS - 1 ? nk - 1 : WK_NS - 1) : fly);
                raw_barrier();
                if (kt >= 1 && kt - 1 + WK_NS < nk) {
#pragma unroll
                    for (int i = 0; i < IPW; ++i) issue_one(kt - 1 + WK_NS, i);
                }
            }
            return;
        }
        for (int kt = 0; kt < nk; ++kt) {
            raw_barrier();
            read_frags(kt, std::integral_constant<int, 0>{});
            mfmas(kt, std::integral_constant<int, 0>{}, std::false_type{});
        }
    } else {
#pragma unroll
    for (int p = 0; p < WK_NS; ++p)
        if (p < nk) {
#pragma unroll
            for (int i = 0; i < IPW; ++i) issue_one(p, i);
        }
    if (nk > 0) {
        wait_tiles(nk - 1 < WK_NS - 1 ? nk - 1 : WK_NS - 1);
        raw_barrier();
        read_frags(0, std::integral_constant<int, 0>{});
    }
    int kt = 0;
    for (; kt + 1 + WK_NS < nk; kt += 2) {                       // steady state, two k-tiles per trip (the register sets alternate)
        step(kt, std::integral_constant<int, 0>{}, std::true_type{});
        step(kt + 1, std::integral_constant<int, 1>{}, std::true_type{});
    }
    if (kt + WK_NS < nk) {                                       // one more tile to request
        step(kt, std::integral_constant<int, 0>{}, std::true_type{});
        ++kt;
        for (; kt + 1 < nk; kt += 2) {
            step(kt, std::integral_constant<int, 1>{}, std::false_type{});
            step(kt + 1, std::integral_constant<int, 0>{}, std::false_type{});
        }
        if (kt < nk) step(kt, std::integral_constant<int, 1>{}, std::false_type{});
    } else {
        for (; kt + 1 < nk; kt += 2) {
            step(kt, std::integral_constant<int, 0>{}, std::false_type{});
            step(kt + 1, std::integral_constant<int, 1>{}, std::false_type{});
        }
        if (kt < nk) step(kt, std::integral_constant<int, 0>{}, std::false_type{});
    }
    }

    if (L2D) {
        // ---- plain planes: straight into the gradient (fp32 atomics only when this problem is K-sliced: small outputs, few addresses)
        const bool atomic = slices > 1;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = 128 * tm + 64 * wm + 16 * mt + fr;
            if (m >= P.M) continue;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = 128 * tn + 16 * NT * wn + 16 * nt + 4 * g;
                if (n >= P.N) continue;
                float* o = P.out + (long long)m * P.ldo + n;
                if (atomic) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(o + e, acc[mt][nt][e]);
                } else {
                    f32x4 v = *reinterpret_cast<const f32x4*>(o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += acc[mt][nt][e];
                    *reinterpret_cast<f32x4*>(o) = v;
                }
            }
        }
        if (bias && g == 0) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int m = 128 * tm + 64 * wm + 16 * mt + fr;
                if (m < P.M) {
                    if (atomic) atomicAdd(P.bias_out + m, bacc[mt][0]);
                    else P.bias_out[m] += bacc[mt][0];
                }
            }
        }
        return;
    }
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
    int heads_m, heads_n;                                    // slab index 64 head + d (d < 62) of a 256-group <-> index 62 head + d (+ 248 group) of out
    int bias_mfma;
    int first;                                               // first workgroup of this problem
};
struct wk_reduce_table {
    wk_reduce_problem p[WK_MAXP];
    int n;
};

// slab index -> index of `out` (or -1: padding)
__device__ __forceinline__ int wk_out_index(int i, int heads, int limit) {
    int o = i;
    if (heads) {
        const int grp = i >> 8, r = i & 255, hd = r >> 6, d = r & 63;
        o = d < 62 ? 248 * grp + 62 * hd + d : -1;
    }
    return o < limit ? o : -1;
}

// out += sum over the slices, in slice order.  Workgroup = 4 slab rows x 64 column quads x 4 slice groups: thread (row, quad, sg) sums slices sg,
// sg + 4, ... of its 4 consecutive columns (16-byte loads, all of them in flight), the four groups meet in LDS and are added in a fixed order --
// the same bits on every run.  (One thread per element walking all slices took 8 .. 18 us per launch: 64 dependent-latency rounds on 250
// workgroups.)  The bias gradient is column 255 of the slab rows (ones column of X) or the extra per-slice rows behind the tiles (bias_mfma).
// ADAM (eegclip_wgrad_tok_reduce_adamw): the sums do not land in the gradient, they STEP THE OPTIMIZER -- element x of the gradient run [G, G + n) is
// x = old gradient + slices (what the plain form stores), then AdamW on the parameter / moments at the same offset and G[x] = 0 (the zero_grad() that opens
// the next iteration).  The elements of the run that no problem covers (LayerNorm rows, embeddings: their gradients came from other kernels) are stepped
// by the workgroups behind the reduction's, range by range.  One launch instead of reduction -> optimizer at the end of the training step.
constexpr int WK_GAPS = 2 * WK_MAXP + 2;
struct wk_adam {
    float *G, *P, *M, *V;
    float lr, b1, b2, eps, wd, step, bc2s;                   // step = lr / (1 - beta1^t)
    int first_gap_wg, gap_wgs, n_gaps;
    long long gap0[WK_GAPS], gapn[WK_GAPS];                  // uncovered ranges of the run (offset, count)
};
template <bool ADAM>
__device__ __forceinline__ void wk_land(const wk_adam& A, float* g, float r) {
    if constexpr (ADAM) {
        const long long off = g - A.G;
        const float gi = *g + r;
        *g = 0.f;
        adamw_element(A.P + off, A.M + off, A.V + off, gi, A.lr, A.b1, A.b2, A.eps, A.wd, A.step, A.bc2s);
    } else {
        *g += r;
    }
}
template <bool ADAM>
__global__ __launch_bounds__(256) void wgrad_tok_reduce_kernel(const wk_reduce_table tb, int slices, const wk_adam A) {
    EEG_LDS_BASE(f32x4, red);                                // [4 slice groups][64 threads]
    if constexpr (ADAM) {
        if ((int)blockIdx.x >= A.first_gap_wg) {             // plain AdamW over the uncovered ranges
            const int w = (int)blockIdx.x - A.first_gap_wg;
            for (int r = 0; r < A.n_gaps; ++r)
                for (long long i = (long long)w * 256 + threadIdx.x; i < A.gapn[r]; i += 256LL * A.gap_wgs) {
                    const long long x = A.gap0[r] + i;
                    const float gi = A.G[x];
                    A.G[x] = 0.f;
                    adamw_element(A.P + x, A.M + x, A.V + x, gi, A.lr, A.b1, A.b2, A.eps, A.wd, A.step, A.bc2s);
                }
            return;
        }
    }
    int prob = 0;
#pragma unroll
    for (int p = 1; p < WK_MAXP; ++p)
        if (p < tb.n && (int)blockIdx.x >= tb.p[p].first) prob = p;
    const wk_reduce_problem& P = tb.p[prob];
    const int t = threadIdx.x, sg = t >> 6, q = t & 63;
    const int local = (int)blockIdx.x - P.first;
    const int tiles_rows = P.Mp / 4;                         // workgroups that cover the slab tiles; then the bias_mfma rows (Mp / 256 workgroups)
    const long long stride = (long long)P.Mp * 256;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool tile_wg = local < tiles_rows;
    int m = 0, c4 = 0;
    const float* p;
    long long step;
    if (tile_wg) {
        m = 4 * local + (q >> 4);                            // 4 rows x 16 quads per 64 threads ... x 4 column blocks below
        c4 = q & 15;
        p = P.slab + (long long)m * 256;
        step = stride;
    } else {
        m = 256 * (local - tiles_rows) + 4 * q;              // bias_mfma partial rows: 4 consecutive m per thread
        p = P.slab + (long long)slices * stride + m;
        step = P.Mp;
    }
    f32x4 acc4[4];                                           // tile workgroups: 4 column blocks of 64 columns (quad c4 + 16 cb)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc4[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tile_wg) {
#pragma unroll 4
        for (int k = sg; k < slices; k += 4) {
            f32x4 v[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) v[cb] = *reinterpret_cast<const f32x4*>(p + (long long)k * step + 4 * (c4 + 16 * cb));
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc4[cb][e] += v[cb][e];
        }
    } else if (P.bias_mfma && P.bias_out) {
        for (int k = sg; k < slices; k += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + (long long)k * step);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] += v[e];
        }
        acc4[0] = s;
    }
    // the four slice groups in fixed order: ((g0 + g1) + (g2 + g3))
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        if (cb > 0 && !tile_wg) break;
        red[sg * 64 + q] = acc4[cb];
        __syncthreads();
        if (sg == 0) {
            const f32x4 a0 = red[q], a1 = red[64 + q], a2 = red[128 + q], a3 = red[192 + q];
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = (a0[e] + a1[e]) + (a2[e] + a3[e]);
            if (tile_wg) {
                const int om = wk_out_index(m, P.heads_m, P.M);
                if (om >= 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int n = 4 * (c4 + 16 * cb) + e;
                        const int on = wk_out_index(n, P.heads_n, P.N);
                        if (on >= 0) wk_land<ADAM>(A, P.out + (long long)om * P.ldo + on, r[e]);
                        else if (n == 255 && P.bias_out && !P.bias_mfma) wk_land<ADAM>(A, P.bias_out + om, r[e]);
                    }
                }
            } else if (P.bias_mfma && P.bias_out) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int om = wk_out_index(m + e, P.heads_m, P.M);
                    if (om >= 0) wk_land<ADAM>(A, P.bias_out + om, r[e]);
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace eeg

using namespace eeg;

static int wk_check(const eegclip_wgrad_tok_problem* p, int n_prob, int B) {
    if (!p || n_prob < 1 || n_prob > WK_MAXP_TOK || B < 1) return EEGCLIP_EINVAL;
    for (int i = 0; i < n_prob; ++i) {
        const eegclip_wgrad_tok_problem& q = p[i];
        if (!q.a || !q.b || !q.out || q.m_groups < 1 || q.m_groups > 3 || q.M < 1 || q.N < 1 || q.ldo < q.N) return EEGCLIP_EINVAL;
        if (q.M > (q.heads_m ? 248 : 256) * q.m_groups || q.N > (q.heads_n ? 248 : 256)) return EEGCLIP_EINVAL;
        if (q.bias_out && !q.bias_mfma && q.N > (q.heads_n ? 248 : 255)) return EEGCLIP_EINVAL;      // column 255 must be the ones column, not data
        if (q.m_groups > 1 && q.a_group_stride < (long long)B * WK_SAMPLE) return EEGCLIP_EINVAL;
        if ((reinterpret_cast<uintptr_t>(q.a) | reinterpret_cast<uintptr_t>(q.b) | (uintptr_t)q.a_group_stride) & 15u) return EEGCLIP_EALIGN;
        // a sample range (the joint-subject value embedding: one problem per subject) must lie inside the batch; ONE index list per launch
        if (q.sample0 < 0 || q.samples < 0 || q.sample0 + q.samples > B || (q.samples == 0 && q.sample0 != 0)) return EEGCLIP_EINVAL;
        if (q.sample_index != p[0].sample_index || (reinterpret_cast<uintptr_t>(q.sample_index) & 3u)) return EEGCLIP_EINVAL;
    }
    return 0;
}

extern "C" int eegclip_wgrad_tok_slices(int total_m_groups, int B) {
    if (total_m_groups < 1 || B < 1) return 0;
    const int tiles = 4 * total_m_groups, kt = 2 * B;
    int s = 256 / tiles;                                         // one workgroup (128 KB of LDS) per CU
    if (s > 32) s = 32;                                          // (more slices: the slab traffic of the reduction outgrows what the extra workgroups gain)
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
    tb.index = p[0].sample_index;
    blocks = threads = 0;
    float* ws = workspace;
    for (int i = 0; i < n_prob; ++i) {
        const eegclip_wgrad_tok_problem& q = p[i];
        const int Mp = 256 * q.m_groups;
        tb.p[i] = wk_problem{static_cast<const unsigned char*>(q.a), static_cast<const unsigned char*>(q.b), ws, q.a_group_stride, 2 * q.m_groups,
                             (q.bias_out && q.bias_mfma) ? 1 : 0, blocks, q.sample0, 2 * (q.samples ? q.samples : B)};
        rt.p[i] = wk_reduce_problem{ws, q.out, q.bias_out, q.ldo, q.M, q.N, Mp, q.heads_m ? 1 : 0, q.heads_n ? 1 : 0, q.bias_mfma ? 1 : 0, threads};
        blocks += 4 * q.m_groups * slices;
        threads += Mp / 4 + ((q.bias_out && q.bias_mfma) ? q.m_groups : 0);      // (workgroups of the reduction)
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
    // measured on the MI355X (tools/bench_wgrad_tok.py, B = 256, 16 slices, us): 512-thread / 4 stages 25.0 (q|k|v) 28.7 (FFN + out-projection);
    // 256-thread 28.9 / 30.7; 5 stages (all 160 KB of LDS) 31.1 / 34.5 -- more bytes in flight do not help: the kernel moves its operands at
    // 3.3 TB/s (+ 0.8 of slab writes) with the matrix pipe of the CUs it occupies 50 % busy (profiles/r4_pmc_wgrad_tok.json)
    if (tb.index) EEG_LAUNCH((wgrad_tok_kernel<4, 4, true>), dim3((unsigned)blocks), dim3(512), 4 * WK_STAGE, stream, tb, slices);
    else if (variant == 2) EEG_LAUNCH((wgrad_tok_kernel<4, 4, false, false, true>), dim3((unsigned)blocks), dim3(768), 4 * WK_STAGE, stream, tb, slices);
    else if (variant == 3) EEG_LAUNCH((wgrad_tok_kernel<2, 4, false, false, true>), dim3((unsigned)blocks), dim3(512), 4 * WK_STAGE, stream, tb, slices);
    else if (variant == 1) EEG_LAUNCH((wgrad_tok_kernel<2, 4, false>), dim3((unsigned)blocks), dim3(256), 4 * WK_STAGE, stream, tb, slices);
    else EEG_LAUNCH((wgrad_tok_kernel<4, 4, false>), dim3((unsigned)blocks), dim3(512), 4 * WK_STAGE, stream, tb, slices);
    return (int)hipGetLastError();
}

// ... and out += their sum in slice order (ONE kernel); same arguments
extern "C" int eegclip_wgrad_tok_reduce(const eegclip_wgrad_tok_problem* p, int n_prob, int B, int slices, float* workspace, void* stream) {
    wk_table tb;
    wk_reduce_table rt;
    int blocks, threads;
    const int rc = wk_tables(p, n_prob, B, slices, workspace, tb, rt, blocks, threads);
    if (rc) return rc;
    wk_adam none;
    memset(&none, 0, sizeof(none));
    EEG_LAUNCH(wgrad_tok_reduce_kernel<false>, dim3((unsigned)threads), dim3(256), 256 * sizeof(f32x4), stream, rt, slices, none);
    return (int)hipGetLastError();
}

// ... or, at the end of a training step: the sums step the optimizer instead (see wk_adam).  [G, G + n) is one run of the flat gradient buffer with its
// parameters P and moments M, V at the same offsets; every problem's out (dense: ldo == N) and bias_out must lie inside it, disjoint.
extern "C" int eegclip_wgrad_tok_reduce_adamw(const eegclip_wgrad_tok_problem* p, int n_prob, int B, int slices, float* workspace, float* P, float* G, float* M,
                                              float* V, long long n, float lr, float beta1, float beta2, float eps, float weight_decay, long long step,
                                              void* stream) {
    if (!P || !G || !M || !V || n < 1 || step < 1) return EEGCLIP_EINVAL;
    wk_table tb;
    wk_reduce_table rt;
    int blocks, threads;
    const int rc = wk_tables(p, n_prob, B, slices, workspace, tb, rt, blocks, threads);
    if (rc) return rc;
    // covered ranges of the run, sorted; what lies between them are the gaps
    long long c0[2 * WK_MAXP], cn[2 * WK_MAXP];
    int nc = 0;
    for (int i = 0; i < n_prob; ++i) {
        if (p[i].ldo != p[i].N) return EEGCLIP_EINVAL;
        c0[nc] = p[i].out - G;
        cn[nc++] = (long long)p[i].M * p[i].N;
        if (p[i].bias_out) {
            c0[nc] = p[i].bias_out - G;
            cn[nc++] = p[i].M;
        }
    }
    for (int i = 1; i < nc; ++i)
        for (int j = i; j > 0 && c0[j] < c0[j - 1]; --j) {
            const long long a = c0[j], b = cn[j];
            c0[j] = c0[j - 1]; cn[j] = cn[j - 1];
            c0[j - 1] = a; cn[j - 1] = b;
        }
    wk_adam A;
    memset(&A, 0, sizeof(A));
    long long at = 0, gap_total = 0;
    for (int i = 0; i <= nc; ++i) {
        const long long upto = i < nc ? c0[i] : n;
        if (upto < at || (i < nc && c0[i] + cn[i] > n)) return EEGCLIP_EINVAL;       // outside the run, or overlapping ranges
        if (upto > at) {
            A.gap0[A.n_gaps] = at;
            A.gapn[A.n_gaps++] = upto - at;
            gap_total += upto - at;
        }
        if (i < nc) at = c0[i] + cn[i];
    }
    A.G = G; A.P = P; A.M = M; A.V = V;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    A.lr = lr; A.b1 = beta1; A.b2 = beta2; A.eps = eps; A.wd = weight_decay;
    A.step = lr / (float)bc1;
    A.bc2s = (float)sqrt(bc2);
    A.first_gap_wg = threads;
    A.gap_wgs = gap_total ? (int)((gap_total + 1023) / 1024 < 64 ? (gap_total + 1023) / 1024 : 64) : 0;
    EEG_LAUNCH(wgrad_tok_reduce_kernel<true>, dim3((unsigned)(threads + A.gap_wgs)), dim3(256), 256 * sizeof(f32x4), stream, rt, slices, A);
    return (int)hipGetLastError();
}

// ---- the same kernel over PLAIN 2-D planes: dW[m][n] += sum_r dY[r][m] X[r][n], dY (rows, M) and X (rows, N) each a hi and a lo bf16 plane with the
// channel index contiguous (rows lda / ldb elements apart) -- the weight gradients of the diffusion prior's Linears (Generation/diffusion_prior.py:
// 167-203 differentiated w.r.t. the weights; rows = the batch).  The general split-bf16 GEMM took 37 us for each of them whatever its size (both
// operands k-strided: transposed in registers while staged), 27 launches per step on the plan's second stream = the critical path of the backward.
extern "C" int eegclip_wgrad_planes(const eegclip_wgrad_planes_problem* p, int n_prob, void* stream) {
    if (!p || n_prob < 1 || n_prob > WK_MAXP) return EEGCLIP_EINVAL;
    wk_table tb;
    memset(&tb, 0, sizeof(tb));
    tb.n = n_prob;
    int blocks = 0;
    for (int i = 0; i < n_prob; ++i) {
        const eegclip_wgrad_planes_problem& q = p[i];
        if (!q.a_hi || !q.a_lo || !q.b_hi || !q.b_lo || !q.out || q.rows < WK_BK || q.rows % WK_BK || q.M < 1 || q.N < 1 || q.lda < q.M || q.ldb < q.N ||
            q.ldo < q.N || (q.ldo & 3) || (q.N & 3) || (q.lda & 7) || (q.ldb & 7) || q.slices < 1 || q.slices > q.rows / WK_BK)
            return EEGCLIP_EINVAL;
        if ((reinterpret_cast<uintptr_t>(q.a_hi) | reinterpret_cast<uintptr_t>(q.a_lo) | reinterpret_cast<uintptr_t>(q.b_hi) | reinterpret_cast<uintptr_t>(q.b_lo) |
             reinterpret_cast<uintptr_t>(q.out)) & 15u)
            return EEGCLIP_EALIGN;
        wk_problem& w = tb.p[i];
        w.a = static_cast<const unsigned char*>(q.a_hi);
        w.b = static_cast<const unsigned char*>(q.b_hi);
        w.a_lo = static_cast<const unsigned char*>(q.a_lo) - w.a;
        w.b_lo = static_cast<const unsigned char*>(q.b_lo) - w.b;
        w.a_tok = (int)(2 * q.lda);
        w.b_tok = (int)(2 * q.ldb);
        w.m_tiles = (q.M + 127) / 128;
        w.n_tiles = (q.N + 127) / 128;
        w.slices = q.slices;
        w.bias_mfma = q.bias_out ? 1 : 0;
        w.first_block = blocks;
        w.sample0 = 0;
        w.ktiles = q.rows / WK_BK;
        w.out = q.out;
        w.bias_out = q.bias_out;
        w.ldo = q.ldo;
        w.M = q.M;
        w.N = q.N;
        blocks += w.m_tiles * w.n_tiles * q.slices;
    }
    EEG_LAUNCH((wgrad_tok_kernel<4, 4, false, true, true>), dim3((unsigned)blocks), dim3(768), 4 * WK_STAGE, stream, tb, 1);
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
