// Fused CLIP-symmetric InfoNCE (models/loss.py:100-141): the N x N logits never exist in HBM on the forward.
//
// One "block" is S = s * Q K^T with Q (n, D) the rows that are scored and K (N, D) what they are scored against; its loss term is the
// cross-entropy of every row against column col0 + row (loss.py:129-130: labels = arange(n) + n * rank).  The reference's symmetric loss is
// two such blocks, (Q, K) = (A, B) and (B, A) -- it literally computes both logit matrices (loss.py:122-123); the row-sharded data-parallel
// form is the same two blocks with K = the gathered features (loss.py:113-115).  So ONE kernel shape covers everything:
//
//   infonce_tile_kernel<FWD>   logits tile on the bf16 matrix cores -> per-row (max, sum exp) partial of the tile, the positive logit
//   infonce_finalize_kernel    partials -> log-sum-exp per row, loss += w / n_total * sum_rows (lse - positive)
//   infonce_tile_kernel<GRAD>  recomputes the tile, writes G = s * w / n_total * (exp(S - lse_row) [+ exp(S - lse_key)] - [1|2] delta) in fp32 --
//                              the ONLY N x N array of the step, written once and read once by the dQ = G K GEMM -- and d loss / d s
//
// Arithmetic (template NP): NP = 1 "throughput": features rounded to bf16, one product; NP = 2 "parity": features split hi + lo
// (eegclip_split_bf16), q*k = q_hi k_hi + q_hi k_lo + q_lo k_hi with fp32 accumulation -- logits within ~5e-5 of exact fp32 products
// (budget 1e-3), like gemm_x3.hip.
//
// Tile = TM x TM logits per 256-thread workgroup (TM = 128, or 64 when the block has few tiles: a rank's 256 x 2048 block is 128 workgroups of
// 64 x 64), 2 x 2 wavefronts, each wave (TM/2)^2 as 32x32 MFMA tiles (v_mfma_f32_32x32x16_bf16).  The product is formed TRANSPOSED
// (MFMA rows = keys, columns = queries): a lane then owns ONE query row and its 16 accumulator registers per tile run over keys, so the
// row max / row sum are in-lane reductions plus one exchange with lane ^ 32 -- no cross-lane reduction trees.  (The column statistics of the
// same tile would need them; they are the row statistics of the swapped block, which is a second block of the same launch.)
// Operand tiles go global -> LDS by LDS-DMA (16 bytes per lane, no VGPRs), 4 stages deep with counted vmcnt, exactly the pipeline of
// logits_bf16.hip; LDS rows are unpadded (DMA deposits lane-linear), bank conflicts are avoided by XOR-swizzling the 16-byte chunk index on
// the DMA SOURCE address and on the ds_read address: f(row) = (row >> 1) & 7 for 128-byte rows (BK = 64), (row >> 2) & 3 for 64-byte rows
// (BK = 32) -- conflict free for the 32x32x16 fragment fetch "lane (r = lane & 31, h = lane >> 5) <- chunk 2 step + h of row r".
#include "eeg_common.h"

#include <stdlib.h>
#include <string.h>

namespace eeg {

constexpr int IF_NS = 4;                 // LDS stages
constexpr float IF_LOG2E = 1.4426950408889634f, IF_LN2 = 0.6931471805599453f;
__device__ __forceinline__ float if_exp2(float x) {
#if defined(EEG_EMU)
    return exp2f(x);
#else
    return __builtin_amdgcn_exp2f(x);      // v_exp_f32
#endif
}
constexpr int IF_MAX_PROB = 8;

struct if_problem {                      // device copy of eegclip_infonce_problem (pointers only what the kernels use)
    const unsigned short* q_hi;
    const unsigned short* q_lo;
    const unsigned short* k_hi;
    const unsigned short* k_lo;
    float* part;                         // [2][P][n]: max plane, sum plane
    float* diag;                         // [n]
    float* lse;                          // [n]
    const float* lse_k;                  // [N] or null
    float* G;
    long long ldg;
    int col0;
    float weight;
    const float* part_k;                 // gradient pass with the finalize folded in: partials / positives of the SWAPPED block (its rows = this block's keys)
    const float* diag_k;
    unsigned short* G_hi;                // G again as bf16 hi | lo planes (or null)
    unsigned short* G_lo;
};
struct if_table {
    if_problem p[IF_MAX_PROB];
};

// k elements per LDS stage: 64 with one product, 32 with three (two planes per operand) -- and 32 for the 256 x 256 tile, whose stage then is 32 KB like the others
constexpr int if_bk(int NP, int TM) { return (NP == 1 && TM <= 128) ? 64 : 32; }
template <int BK_>
struct if_geom {
    static constexpr int BK = BK_;                            // bf16 elements per row of a stage tile
    static constexpr int ROWB = 2 * BK;                       // bytes per LDS row
    static constexpr int NCH = BK / 8;                        // 16-byte chunks per row
    static constexpr int RPI = 1024 / ROWB;                   // rows one DMA instruction deposits
    __device__ static __forceinline__ int swz(int row) { return BK == 64 ? (row >> 1) & 7 : (row >> 2) & 3; }
};

// MODE 0 = forward partials, 1 = gradient tile.  (Round 3 also built a variant that staged the operand tiles through registers -- global_load_dwordx4
// -> ds_write_b128, two LDS stages -- to test whether the DMA fill is the limit at N = 2048: measured SLOWER than the 4-stage DMA pipeline, 20.8 vs
// 18.3 us per logits block; removed in round 4.)
// NW = 4 | 8 MFMA waves per workgroup.  8 (TM = 128 only; round 4): 2 query halves x 4 key quarters, each wave 64 x 32 logits -- two waves per SIMD, so one
// wave's fragment reads / waits run under the other's MFMAs.
// NPRD = 0 | 4 PRODUCER waves (round 4).  What bounds the k-loop is the CU's vector-memory path, not the matrix pipe: a 128 x 128 x 64 bf16 k-tile is 32 KB
// through the texture addresser / L1 at ~64 B/clk (measured 49-69 B/clk, tools/micro/tile_chain.hip, tools/micro/dma_rate.hip) = ~500-650 cycles, the same
// ~512 cycles its 16 MFMAs take -- and a wave that issues both is IN ORDER: it stalls ~60 cycles in every LDS-DMA issue (the queue is full) with its MFMAs
// unissued behind it, so the two costs add (1390 cycles per k-tile: 108 vmcnt wait + 76 barrier + 119 first fragments + 1085 reads/MFMAs/DMA issues; 830 of
// the 1085 without the refills, 480 with the refills alone).  With NPRD = 4 the LDS-DMA instructions, their counted vmcnt waits and nothing else live in
// four extra waves (one per SIMD, parked in the memory queue almost all the time); the MFMA waves issue no vector-memory instruction at all and meet the
// producers at the one barrier per k-tile: 1090 cycles per k-tile in the micro-benchmark, 13.8 -> 10.1 us for the k-loops of one N = 2048 block.
// PF (round 6; the 8-wave forms: 256-tiles and 128-tiles): the barrier of k-tile kt + 1 is met INSIDE k-tile kt, in front of its last MFMA step, and the first fragments of tile kt + 1 are
// requested right behind it -- so every wave leaves the barrier with a step's MFMAs ready to issue and the fragment-read latency of the next tile under them.
// (Without it both waves of a SIMD come out of the barrier with nothing to feed the matrix pipe until their first six ds_read_b128 return.)  The refill DMA of
// tile kt + NS - 1 is issued behind that barrier too (its stage held tile kt - 1, whose reads every wave has consumed by then), so two tiles instead of
// three are in flight.
template <int NP, int TM, int MODE, int NW, int NPRD = 0, bool PF = false>
__global__ __launch_bounds__(64 * (NW + NPRD)) void infonce_tile_kernel(const if_table tb, int n, int N, int D, int tiles_q, int tiles_k, const float* __restrict__ scale,
                                                            float inv_total, float* __restrict__ dscale, float* __restrict__ loss) {
    using Gm = if_geom<if_bk(NP, TM)>;
    constexpr int BK = Gm::BK, ROWB = Gm::ROWB, NCH = Gm::NCH, RPI = Gm::RPI;
    constexpr bool SPEC = NPRD > 0;
    constexpr int NDW = SPEC ? NPRD : NW;                     // waves that issue LDS-DMA
    constexpr int NWK = NW / 2;                               // waves along the keys
    constexpr int WT = TM / 64;                               // 32x32 MFMA tiles per wave along the queries
    constexpr int WTK = TM / (32 * NWK);                      // ... along the keys
    constexpr int SPT = TM == 256 ? 1 : NWK;                  // partial slots per (row, key tile): the 256-tile combines its key quarters in LDS first
    constexpr int TILE_B = TM * ROWB;                         // bytes of one operand-plane tile
    constexpr int STAGE_B = 2 * NP * TILE_B;                  // q_hi | k_hi | (q_lo | k_lo)
    constexpr int IPT = TM / RPI / NDW;                       // DMA instructions per DMA wave, tile and operand plane
    constexpr int DPT = 2 * NP * IPT;                         // ... per DMA wave and k-tile (what vmcnt counts)
    static_assert(IPT >= 1 && WTK >= 1 && 2 * DPT < 64, "tile too small for this many waves");
    EEG_LDS_BASE(unsigned char, lds);

    const int tiles = tiles_q * tiles_k;
    const int prob = (int)blockIdx.x / tiles;
    const int rem = (int)blockIdx.x - prob * tiles;
    // (An XCD-aware tile order -- every XCD owning whole 4 x 8 supertiles so that its 32 workgroups share 12 operand tile rows instead of 18 -- was tried
    //  in round 4 on the reading that fabric traffic bounds the kernel at N = 2048 (FETCH_SIZE 77 MB for 8 MB of operands): no gain, 16.9 vs 15.7 us one
    //  product, 32.2 vs 29.7 us three products.  A single 128 x 128 tile on an otherwise idle chip takes the same ~0.6 us per 64-k tile as 256 of them:
    //  the bound is the per-tile chain wait -> barrier -> fragment reads -> MFMAs, not bandwidth.  DESIGN.md section 4.)
    const if_problem& P = tb.p[prob];
    const int q0 = (rem / tiles_k) * TM, k0r = (rem % tiles_k) * TM;
    const int t = threadIdx.x, lane = t & 63, wave = wave_uniform(t >> 6);
    const bool producer = SPEC && wave >= NW;                 // (wave-uniform)
    const int dw = SPEC ? wave - NW : wave;                   // index among the DMA waves (negative for the MFMA waves of the specialised form: unused there)
    const int wq = wave / NWK, wk = wave % NWK;
    const int r32 = lane & 31, h = lane >> 5;

    // ---- DMA roles: DMA wave w deposits rows [w * TM/NDW, +TM/NDW) of every operand-plane tile, instruction i = rows + RPI * i
    const int drow = lane / NCH, dpos = lane % NCH;
    const unsigned short* src[2 * NP][IPT];
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const int row = (dw & (NDW - 1)) * (TM / NDW) + RPI * i + drow;
        const int col = 8 * (dpos ^ Gm::swz(row));
        src[0][i] = P.q_hi + (long long)(q0 + row) * D + col;
        src[1][i] = P.k_hi + (long long)(k0r + row) * D + col;
        if (NP == 2) {
            src[2 * NP - 2][i] = P.q_lo + (long long)(q0 + row) * D + col;
            src[2 * NP - 1][i] = P.k_lo + (long long)(k0r + row) * D + col;
        }
    }
    auto issue_one = [&](int kt, int dnum) {                  // DMA instruction `dnum` (0 .. DPT-1) of k-tile kt
        const int o = dnum / IPT, i = dnum % IPT;
        unsigned char* st = lds + (kt % IF_NS) * STAGE_B + (dw & (NDW - 1)) * (TM / NDW) * ROWB;
        lds_dma16(st + o * TILE_B + RPI * i * ROWB, src[o][i] + kt * BK);
    };
    auto issue_tile = [&](int kt) {
#pragma unroll
        for (int dnum = 0; dnum < DPT; ++dnum) issue_one(kt, dnum);
    };

    f32x16 acc[WTK][WT];                                      // acc[j][i]: key tile j (MFMA rows), query tile i (MFMA columns)
#pragma unroll
    for (int j = 0; j < WTK; ++j)
#pragma unroll
        for (int i = 0; i < WT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

    // fragment addresses (do not depend on the k-tile): byte offset of chunk 2 s + h of this lane's row in a q / k tile
    int foq[BK / 16][WT], fok[BK / 16][WTK];
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
#pragma unroll
        for (int i = 0; i < WT; ++i) {
            const int rq = wq * (TM / 2) + 32 * i + r32;
            foq[s][i] = rq * ROWB + (((2 * s + h) ^ Gm::swz(rq)) & (NCH - 1)) * 16;
        }
#pragma unroll
        for (int j = 0; j < WTK; ++j) {
            const int rk = wk * (TM / NWK) + 32 * j + r32;
            fok[s][j] = TILE_B + rk * ROWB + (((2 * s + h) ^ Gm::swz(rk)) & (NCH - 1)) * 16;
        }
    }

    // One k-tile = NSTEP MFMA k-steps.  With ONE wave per SIMD nothing else covers this wave's issue slots, so the order inside the tile
    // is what overlaps the three pipes (first version: barrier -> 8 DMA issues -> 16 fragment reads -> wait -> 16 MFMAs, strictly one after
    // the other: 0.9 us per k-tile, of which the DMA issues alone ~0.4 -- MI355X_MICROARCH.md prices an LDS-DMA issue at 60-185 cycles):
    //   * the fragment reads of step s + 1 are issued BEFORE the MFMAs of step s (two register sets),
    //   * the DMA instructions of tile kt + 3 are spread between the MFMAs of the whole tile (the matrix pipe works through its queue
    //     while the wave issues them).
    constexpr int NSTEP = BK / 16, MPS = WTK * WT * (NP == 2 ? 3 : 1), TOTAL = NSTEP * MPS;
    const int ktiles = D / BK;
    auto wait_tile = [&](int kt) {                            // this wave's DMA of tile kt has landed (the newer tiles stay in flight)
        const int newer = ktiles - 1 - kt < IF_NS - 2 ? ktiles - 1 - kt : IF_NS - 2;
        if (newer >= 2) wait_vmcnt<2 * DPT>();
        else if (newer == 1) wait_vmcnt<DPT>();
        else wait_vmcnt<0>();
    };
    auto mfma_tile = [&](int kt, bool refill) {               // the MFMAs of k-tile kt (+ the refill DMA of tile kt + NS - 1 between them when this wave does both)
        const unsigned char* st = lds + (kt % IF_NS) * STAGE_B;
        bf16x8 qh[2][WT], kh[2][WTK], ql[2][WT], kl[2][WTK];
        // Issue order: the operands of a step's FIRST MFMA are read LAST, and the next step's reads go out BEHIND that first MFMA.  The compiler's LDS counting
        // is conservative around this loop (it emitted s_waitcnt lgkmcnt(0) in front of a step's MFMAs with the NEXT step's reads already issued, i.e. no
        // overlap at all); this way its one full wait sits in front of the first MFMA and covers only reads issued a whole step earlier.
        auto read_step = [&](int s, int set) {
#pragma unroll
            for (int j = WTK - 1; j >= 0; --j) {
                if (NP == 2 && j > 0) kl[set][j] = *reinterpret_cast<const bf16x8*>(st + 2 * TILE_B + fok[s][j]);
                if (NP == 2 || j > 0) kh[set][j] = *reinterpret_cast<const bf16x8*>(st + fok[s][j]);
            }
#pragma unroll
            for (int i = WT - 1; i >= 0; --i) {
                if (NP == 2) ql[set][i] = *reinterpret_cast<const bf16x8*>(st + 2 * TILE_B + foq[s][i]);
                qh[set][i] = *reinterpret_cast<const bf16x8*>(st + foq[s][i]);
            }
            if (NP == 2) kl[set][0] = *reinterpret_cast<const bf16x8*>(st + 2 * TILE_B + fok[s][0]);
            else kh[set][0] = *reinterpret_cast<const bf16x8*>(st + fok[s][0]);
        };
        read_step(0, 0);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
#if !defined(EEG_EMU)
            __builtin_amdgcn_sched_barrier(0);
#endif
            const int set = s & 1;
#pragma unroll
            for (int j = 0; j < WTK; ++j)
#pragma unroll
                for (int i = 0; i < WT; ++i) {
                    const int m0_ = s * MPS + (j * WT + i) * (NP == 2 ? 3 : 1);      // index of this accumulator's first MFMA within the tile
                    if (NP == 2) {
                        acc[j][i] = mfma_bf16_32x32x16(kl[set][j], qh[set][i], acc[j][i]);
                        if (!SPEC && refill && ((m0_ + 1) * DPT) / TOTAL > (m0_ * DPT) / TOTAL) issue_one(kt + IF_NS - 1, (m0_ * DPT) / TOTAL);
                        acc[j][i] = mfma_bf16_32x32x16(kh[set][j], ql[set][i], acc[j][i]);
                        if (!SPEC && refill && ((m0_ + 2) * DPT) / TOTAL > ((m0_ + 1) * DPT) / TOTAL) issue_one(kt + IF_NS - 1, ((m0_ + 1) * DPT) / TOTAL);
                    }
                    constexpr int last = NP == 2 ? 2 : 0;
                    acc[j][i] = mfma_bf16_32x32x16(kh[set][j], qh[set][i], acc[j][i]);      // D[key 32j + row(reg, h)][query 32i + r32]
                    if (!SPEC && refill && ((m0_ + last + 1) * DPT) / TOTAL > ((m0_ + last) * DPT) / TOTAL) issue_one(kt + IF_NS - 1, ((m0_ + last) * DPT) / TOTAL);
                    if (j == 0 && i == 0 && s + 1 < NSTEP) {
#if !defined(EEG_EMU)
                        __builtin_amdgcn_sched_barrier(0);
#endif
                        read_step(s + 1, (s + 1) & 1);
#if !defined(EEG_EMU)
                        __builtin_amdgcn_sched_barrier(0);
#endif
                    }
                }
#if !defined(EEG_EMU)
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    };
    if (PF) {
        // (measured with producer waves too -- they promise tile kt + 1 at the barrier inside tile kt and refill behind it: neutral on the 4 + 4 form,
        //  16.6 against 16.4 us at N = 2048, 383.5 against 385.2 with three products at N = 8192; not kept)
        static_assert(!PF || (!SPEC && NSTEP % 2 == 0), "the prefetching loop is the un-specialised form; fragment sets alternate by step parity");
        bf16x8 qh[2][WT], kh[2][WTK], ql[2][WT], kl[2][WTK];
        // (issue order: what a step's FIRST MFMA takes -- key fragment 0, query fragment 0 -- is read LAST, so the one wait in front of that MFMA covers the
        //  whole set and no later MFMA of the step waits again behind the next set's reads)
        auto read_step = [&](int kt, int s, int set) {
            const unsigned char* st = lds + (kt % IF_NS) * STAGE_B;
#pragma unroll
            for (int j = WTK - 1; j >= 0; --j) {
                if (NP == 2 && j > 0) kl[set][j] = *reinterpret_cast<const bf16x8*>(st + 2 * TILE_B + fok[s][j]);
                if (NP == 2 || j > 0) kh[set][j] = *reinterpret_cast<const bf16x8*>(st + fok[s][j]);
            }
#pragma unroll
            for (int i = WT - 1; i >= 0; --i) {
                if (NP == 2) ql[set][i] = *reinterpret_cast<const bf16x8*>(st + 2 * TILE_B + foq[s][i]);
                qh[set][i] = *reinterpret_cast<const bf16x8*>(st + foq[s][i]);
            }
            if (NP == 2) kl[set][0] = *reinterpret_cast<const bf16x8*>(st + 2 * TILE_B + fok[s][0]);
            else kh[set][0] = *reinterpret_cast<const bf16x8*>(st + fok[s][0]);
        };
#pragma unroll
        for (int p = 0; p < IF_NS - 1; ++p)
            if (p < ktiles) issue_tile(p);
        wait_tile(0);
        raw_barrier();
        read_step(0, 0, 0);
        constexpr int MPSP = MPS;                                 // MFMAs of one step
        // the refill's DPT DMA instructions spread over the last step's MPSP MFMAs: behind MFMA m go pieces [m DPT / MPSP, (m + 1) DPT / MPSP)
        auto refill_after = [&](int kt, int m) {
#pragma unroll
            for (int d = 0; d < DPT; ++d)
                if (d >= (m * DPT) / MPSP && d < ((m + 1) * DPT) / MPSP) issue_one(kt + IF_NS - 1, d);
        };
        for (int kt = 0; kt < ktiles; ++kt) {
            const bool more = kt + 1 < ktiles, refill = kt + IF_NS - 1 < ktiles;      // (workgroup-uniform)
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                const int set = s & 1;
                if (s + 1 == NSTEP && more) {
                    // this wave's part of tile kt + 1 has landed (tile kt + 2, issued a tile ago, may stay in flight)
                    if (kt + 2 < ktiles) wait_vmcnt<DPT>();
                    else wait_vmcnt<0>();
                    raw_barrier();                                // tile kt + 1 visible to every wave; every wave is past its reads of tile kt - 1
                }
                int m = 0;
#pragma unroll
                for (int j = 0; j < WTK; ++j)
#pragma unroll
                    for (int i = 0; i < WT; ++i) {
                        if (NP == 2) {
                            acc[j][i] = mfma_bf16_32x32x16(kl[set][j], qh[set][i], acc[j][i]);
                            if (s == NSTEP - 1 && refill) refill_after(kt, m);
                            ++m;
                            acc[j][i] = mfma_bf16_32x32x16(kh[set][j], ql[set][i], acc[j][i]);
                            if (s == NSTEP - 1 && refill) refill_after(kt, m);
                            ++m;
                        }
                        acc[j][i] = mfma_bf16_32x32x16(kh[set][j], qh[set][i], acc[j][i]);
                        if (s == NSTEP - 1 && refill) refill_after(kt, m);
                        ++m;
                        if (j == 0 && i == 0) {
                            // the next step's fragment reads go out BEHIND this step's first MFMA: the s_waitcnt lgkmcnt(0) the compiler puts in front of that MFMA
                            // (it does not count LDS reads across the loop edge) then covers only reads issued a whole step ago
#if !defined(EEG_EMU)
                            __builtin_amdgcn_sched_barrier(0);
#endif
                            if (s + 1 < NSTEP) read_step(kt, s + 1, set ^ 1);
                            else read_step(more ? kt + 1 : kt, 0, set ^ 1);      // (unconditional: a branch here makes the compiler's LDS counts conservative; the last tile re-reads itself)
#if !defined(EEG_EMU)
                            __builtin_amdgcn_sched_barrier(0);
#endif
                        }
                    }
#if !defined(EEG_EMU)
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        }
    } else if (!SPEC) {
#pragma unroll
        for (int p = 0; p < IF_NS - 1; ++p)
            if (p < ktiles) issue_tile(p);
        for (int kt = 0; kt < ktiles; ++kt) {
            wait_tile(kt);
            raw_barrier();                                        // tile kt has landed for every wave; the stage about to be refilled is drained
            mfma_tile(kt, kt + IF_NS - 1 < ktiles);               // (workgroup-uniform)
        }
    } else if (producer) {
#pragma unroll
        for (int p = 0; p < IF_NS - 1; ++p)
            if (p < ktiles) issue_tile(p);
        for (int kt = 0; kt < ktiles; ++kt) {
            wait_tile(kt);
            raw_barrier();                                        // the one meeting point of the two kinds of waves per k-tile
            if (kt + IF_NS - 1 < ktiles) issue_tile(kt + IF_NS - 1);
        }
    } else {
        for (int kt = 0; kt < ktiles; ++kt) {
            raw_barrier();
            mfma_tile(kt, false);
        }
    }

    // ---- epilogue: lane (r32, h) owns query row q = q0 + wq TM/2 + 32 i + r32 of tile i; register e of key tile j is key
    //      k = k0r + wk TM/NWK + 32 j + (e & 3) + 8 (e >> 2) + 4 h.
    // (every field of the block descriptor is read into a register HERE: `tb.p[prob]` is a dynamically indexed kernel argument, and a use
    //  inside a conditional costs a scalar load + wait per use -- 64 of them per lane in the first version, ~5 us of a 17 us kernel)
    const float s = *scale;
    float* const p_part = P.part;
    float* const p_diag = P.diag;
    const float* const p_lse = P.lse;
    const float* const p_lse_k = P.lse_k;
    float* const p_G = P.G;
    unsigned short* const p_Ghi = P.G_hi;
    unsigned short* const p_Glo = P.G_lo;
    const long long p_ldg = P.ldg;
    const int p_col0 = P.col0;
    const float p_weight = P.weight;
    if (MODE == 0) {
        if (producer) return;
        const int Pn = SPT * tiles_k;                         // partial slots per row: (key tile, wk) -- or one per key tile
        const int slot = SPT * (rem % tiles_k) + (SPT == 1 ? 0 : wk);
        float* const comb = reinterpret_cast<float*>(lds);    // (SPT == 1) [NWK][TM][2]: the key quarters' (max, sum) of every row, in base-2 units
        if (SPT == 1) __syncthreads();                        // every wave is done with the operand stages
        // Four vector instructions + one v_exp_f32 per logit (round 6; the first version spent eleven: max AND min of the raw accumulators -- each with a
        // canonicalising extra --, s * acc twice, the subtraction, __expf's own multiply, and the positive's compare / select on every element of every tile;
        // at 128 logits per thread of the 256-tile that epilogue was ~12 % of the tile's time): the accumulators are scaled IN PLACE to base-2 logits
        // w = acc * (s log2 e), monotone in the logit whatever the sign of s, so one max serves; exp2(w - max) is a subtract + the hardware exp2; the positive is
        // looked for only by the waves whose key range meets the diagonal.  Partials leave in natural units (max * ln 2).
        const float sl2 = s * IF_LOG2E;
        const int kb = k0r + wk * (TM / NWK);
#pragma unroll
        for (int i = 0; i < WT; ++i) {
            const int qw = q0 + wq * (TM / 2) + 32 * i, q = qw + r32;
            float mx = -3.0e38f;
#pragma unroll
            for (int j = 0; j < WTK; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    acc[j][i][e] *= sl2;
                    mx = fmaxf(mx, acc[j][i][e]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));           // the other half-wave holds the other 16 keys of every 32-key tile
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < WTK; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) sum += if_exp2(acc[j][i][e] - mx);
            sum += __shfl_xor(sum, 32, 64);
            if (p_col0 + qw + 31 >= kb && p_col0 + qw < kb + 32 * WTK) {      // (wave-uniform) the positives of these 32 rows fall among this wave's keys
                const int pos = p_col0 + q - kb - 4 * h;      // key index of the positive in this lane's register numbering, if any
                float dv = 0.f;
                bool have = false;
#pragma unroll
                for (int j = 0; j < WTK; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const bool hit = 32 * j + (e & 3) + 8 * (e >> 2) == pos;
                        dv = hit ? acc[j][i][e] : dv;
                        have = have || hit;
                    }
                if (have) p_diag[q] = dv * IF_LN2;            // exactly one lane of the launch per row
            }
            if (h == 0) {
                if (SPT == 1) {
                    *reinterpret_cast<f32x2_t*>(comb + ((wk * TM) + (q - q0)) * 2) = f32x2_t{mx, sum};
                } else {
                    p_part[(long long)slot * n + q] = mx * IF_LN2;
                    p_part[(long long)(Pn + slot) * n + q] = sum;
                }
            }
        }
        if (SPT == 1) {
            // one partial per row and key TILE: the four key quarters meet here (at N = 8192 the finalize launch read 128 slots per row through its slow
            // path, 16 us of a 150 us logits block; 32 slots take its all-loads-in-flight path, and the partial planes shrink 4 x)
            __syncthreads();
            if (t < TM) {
                f32x2_t v[NWK];
#pragma unroll
                for (int k = 0; k < NWK; ++k) v[k] = *reinterpret_cast<const f32x2_t*>(comb + (k * TM + t) * 2);
                float m = v[0][0];
#pragma unroll
                for (int k = 1; k < NWK; ++k) m = fmaxf(m, v[k][0]);
                float l = 0.f;
#pragma unroll
                for (int k = 0; k < NWK; ++k) l += v[k][1] * if_exp2(v[k][0] - m);
                p_part[(long long)slot * n + q0 + t] = m * IF_LN2;
                p_part[(long long)(Pn + slot) * n + q0 + t] = l;
            }
        }
    } else {
        // `fin` (eegclip_infonce_fused_grad_finalize): no finalize launch ran -- every workgroup forms the log-sum-exp of ITS TM rows and TM keys from the
        // forward's per-tile partials (rows: this block's, keys: the swapped block's) in the dead operand stages, and the first workgroup of every row /
        // key block adds their loss terms.  No cross-workgroup ordering is involved: the partials are the previous launch's output.
        const float* const p_part_k = P.part_k;
        const float* const p_diag_k = P.diag_k;
        const bool fin = p_part_k != nullptr;
        float* const fin_q = reinterpret_cast<float*>(lds) + 64;
        float* const fin_k = fin_q + TM;
        if (fin) {
            __syncthreads();                                         // every wave is done with the operand stages
            float contrib = 0.f;
            if (t < 2 * TM) {
                const bool keys = t >= TM;
                const int idx = keys ? k0r + (t - TM) : q0 + t;
                const float* const part = keys ? p_part_k : p_part;
                const long long ld = keys ? N : n;
                const int Pn = SPT * (keys ? tiles_q : tiles_k);
                float m = -3.0e38f;
                for (int sl = 0; sl < Pn; ++sl) m = fmaxf(m, part[(long long)sl * ld + idx]);
                float l = 0.f;
                for (int sl = 0; sl < Pn; ++sl) l += part[(long long)(Pn + sl) * ld + idx] * fast_exp(part[(long long)sl * ld + idx] - m);
                const float lse = m + logf(l);
                (keys ? fin_k : fin_q)[keys ? t - TM : t] = lse;
                const bool mine = keys ? rem / tiles_k == 0 : rem % tiles_k == 0;
                if (mine) contrib = (lse - (keys ? p_diag_k : p_diag)[idx]) * (p_weight * inv_total);
            }
            if (t < 2 * TM) {                                        // (whole waves: 2 TM is a multiple of 64)
                contrib = wave_sum(contrib);
                if (lane == 0 && contrib != 0.f) atomicAdd(loss, contrib);
            }
            __syncthreads();
        }
        // key_only (round 6): no row normaliser given -- G[i][j] = c s (exp(S_ij - lse_k[j]) - [j == col0 + i]): the gradient matrix of the SWAPPED block,
        // produced already transposed (rows = the gathered queries, columns = this rank's targets), i.e. k-contiguous for the GEMM that contracts over the
        // rank's targets (the gathered-copy gradient of the row-sharded loss, models/loss.py:52-58,129-130)
        const bool key_only = !fin && p_lse == nullptr;
        const bool two_norm = (p_lse_k != nullptr && !key_only) || fin;
        const float two = two_norm ? 2.f : 1.f;
        const float c = p_weight * inv_total;
        float ds = 0.f;
#pragma unroll
        for (int i = 0; i < (producer ? 0 : WT); ++i) {       // (the producer waves only take part in the reduction barriers below)
            const int q = q0 + wq * (TM / 2) + 32 * i + r32;
            const float lq = fin ? fin_q[q - q0] : (key_only ? 0.f : p_lse[q]);
            const int kb = k0r + wk * (TM / NWK);
            const int pos = p_col0 + q - kb;
            const long long goff = (long long)q * p_ldg + kb;
            float* grow = p_G + goff;
#pragma unroll
            for (int j = 0; j < WTK; ++j)
#pragma unroll
                for (int eq = 0; eq < 4; ++eq) {              // registers 4 eq .. 4 eq + 3 are 4 CONSECUTIVE keys: one 16-byte store
                    const int kk = 32 * j + 8 * eq + 4 * h;
                    f32x4 lk = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (fin) lk = *reinterpret_cast<const f32x4*>(fin_k + (kb - k0r) + kk);
                    else if (two_norm || key_only) lk = *reinterpret_cast<const f32x4*>(p_lse_k + kb + kk);
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float raw = acc[j][i][4 * eq + e];
                        const float v = s * raw;
                        float g = key_only ? fast_exp(v - lk[e]) : fast_exp(v - lq);
                        if (two_norm) g += fast_exp(v - lk[e]);
                        g -= (kk + e == pos) ? two : 0.f;
                        g *= c;
                        ds += g * raw;
                        o[e] = g * s;
                    }
                    if (p_G) *reinterpret_cast<f32x4*>(grow + kk) = o;
                    if (p_Ghi) {                              // the query-gradient GEMM's A operand, already split
                        u32x2_t hi, lo;
                        x3_split4(o[0], o[1], o[2], o[3], hi, lo);
                        *reinterpret_cast<u32x2_t*>(p_Ghi + goff + kk) = hi;
                        *reinterpret_cast<u32x2_t*>(p_Glo + goff + kk) = lo;
                    }
                }
        }
        // one atomic per WORKGROUP (same-address atomics retire at ~12 ns each: one per wave was 1024 of them = ~13 us at N = 2048);
        // the operand stages are dead here, the partial sums go through the first bytes of the LDS
        ds = wave_sum(ds);
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds);
        if (lane == 0 && !producer) red[wave] = ds;
        __syncthreads();
        if (t == 0) {
            float tot = (red[0] + red[1]) + (red[2] + red[3]);
            if (NW == 8) tot += (red[4] + red[5]) + (red[6] + red[7]);
            atomicAdd(dscale, tot);
        }
    }
}

// partials -> lse; loss.  One workgroup per 64 rows (grid.y = problem): lane = row, the four waves split the Pn partial slots (slot p of wave
// p & 3), so every load instruction is 64 consecutive rows of one slot (256 contiguous bytes) and ALL of a thread's loads (<= 2 x 16) are in flight
// at once; two LDS exchanges (max, then scaled sum) combine the waves.  One round trip of memory latency per launch -- the first version walked the
// slots with 32 lanes per row, 8 rows per workgroup pass and 8 passes of two dependent load rounds each: 11.3 us on average at N = 2048, as long as
// the tile kernel it follows.  One atomic per workgroup on the loss.
constexpr int IF_FIN_MAX = 16;                                   // partial slots per wave: Pn <= 64 (N <= 2048 with 64-wide tiles; larger N: strided loop)
__global__ __launch_bounds__(256) void infonce_finalize_kernel(const if_table tb, int n, int Pn, float inv_total, float* __restrict__ loss) {
    EEG_LDS_BASE(float, red);                                    // [2][4][64]
    const if_problem& P = tb.p[blockIdx.y];
    const float* const part = P.part;
    const float* const diag = P.diag;
    float* const lse_out = P.lse;
    const float w = P.weight * inv_total;
    const int lane = threadIdx.x & 63, s = threadIdx.x >> 6;
    const int q = (int)blockIdx.x * 64 + lane;
    const bool live = q < n;
    float mx[IF_FIN_MAX], sm[IF_FIN_MAX];
    float m = -3.0e38f;
    if (Pn <= 4 * IF_FIN_MAX) {
#pragma unroll
        for (int i = 0; i < IF_FIN_MAX; ++i) {
            const int p = s + 4 * i;
            const bool ok = live && p < Pn;
            mx[i] = ok ? part[(long long)p * n + q] : -3.0e38f;
            sm[i] = ok ? part[(long long)(Pn + p) * n + q] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < IF_FIN_MAX; ++i) m = fmaxf(m, mx[i]);
    } else if (live) {
        for (int p = s; p < Pn; p += 4) m = fmaxf(m, part[(long long)p * n + q]);
    }
    red[s * 64 + lane] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[lane], red[64 + lane]), fmaxf(red[128 + lane], red[192 + lane]));
    float l = 0.f;
    if (Pn <= 4 * IF_FIN_MAX) {
#pragma unroll
        for (int i = 0; i < IF_FIN_MAX; ++i) l += sm[i] * fast_exp(mx[i] - m);
    } else if (live) {
        for (int p = s; p < Pn; p += 4) l += part[(long long)(Pn + p) * n + q] * fast_exp(part[(long long)p * n + q] - m);
    }
    red[256 + s * 64 + lane] = l;
    __syncthreads();
    if (s == 0) {
        l = (red[256 + lane] + red[320 + lane]) + (red[384 + lane] + red[448 + lane]);
        float contrib = 0.f;
        if (live) {
            const float lse = m + logf(l);
            lse_out[q] = lse;
            contrib = (lse - diag[q]) * w;
        }
        contrib = wave_sum(contrib);
        if (lane == 0) atomicAdd(loss, contrib);
    }
}

// fp32 -> bf16 hi (+ lo = bf16(x - hi)) planes, 8 elements per thread
typedef unsigned int ifu32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ hi, unsigned short* __restrict__ lo,
                                                          long long n8) {
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n8; q += (long long)gridDim.x * blockDim.x) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + 8 * q), b = *reinterpret_cast<const f32x4*>(x + 8 * q + 4);
        float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        unsigned short hb[8], lb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            hb[e] = f32_to_bf16_bits(v[e]);
            lb[e] = f32_to_bf16_bits(v[e] - bf16_bits_to_f32(hb[e]));
        }
        ifu32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = hb[2 * e] | ((unsigned)hb[2 * e + 1] << 16);
        *reinterpret_cast<ifu32x4*>(hi + 8 * q) = o;
        if (lo) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = lb[2 * e] | ((unsigned)lb[2 * e + 1] << 16);
            *reinterpret_cast<ifu32x4*>(lo + 8 * q) = o;
        }
    }
}

// `force`: bits 8..15 of the `planes` argument (tuning / tests): 0 = chosen here, 64, 128 or 255 (= 256 x 256 tiles: one product only)
constexpr long long IF_T256_MIN_TILES = 256;                     // 256-tiles once every CU gets one (N = 4096 square; measured 40.0 us against 48.9 with 128-tiles)
static inline int if_tile(int n, int N, int force, int nplanes, bool auto256 = true) {
    const bool can256 = nplanes == 1 && n % 256 == 0 && N % 256 == 0;
    if (force == 255 && can256) return 256;
    if (force == 128 && n % 128 == 0 && N % 128 == 0) return 128;
    if (force == 64) return 64;
    // 256 x 256 tiles (8 waves of 128 x 64 logits: 0.75 fragment reads per MFMA instead of 1 -- 1.5, a k-loop with twice the matrix work per barrier) where
    // the block is large enough to give every CU several of them
    if (force == 0 && auto256 && can256 && (long long)(n / 256) * (N / 256) >= IF_T256_MIN_TILES) return 256;
    // 128 x 128 tiles once they fill the chip on their own; 64 x 64 otherwise (a rank's 256 x 2048 block: 128 workgroups per block)
    return (n % 128 == 0 && N % 128 == 0 && (long long)(n / 128) * (N / 128) >= 256) ? 128 : 64;
}

static int if_table_from(const eegclip_infonce_problem* probs, int nprob, int planes, bool grad, if_table& tb, bool want_fin = false) {
    if (!probs || nprob < 1 || nprob > IF_MAX_PROB) return EEGCLIP_EINVAL;
    for (int i = 0; i < nprob; ++i) {
        const eegclip_infonce_problem& p = probs[i];
        const bool fin = grad && p.part_k != nullptr;
        if (!p.q_hi || !p.k_hi || (planes == 2 && (!p.q_lo || !p.k_lo)) || (!p.lse && !fin && !(grad && p.lse_k)) || (!grad && (!p.part || !p.diag)) || (grad && ((!p.G && !p.G_hi) || (p.ldg & 3))))
            return EEGCLIP_EINVAL;
        if ((p.G_hi != nullptr) != (p.G_lo != nullptr) || ((reinterpret_cast<uintptr_t>(p.G_hi) | reinterpret_cast<uintptr_t>(p.G_lo)) & 7u)) return EEGCLIP_EINVAL;
        if (fin != want_fin || (fin && (!p.part || !p.diag || !p.diag_k))) return EEGCLIP_EINVAL;
        uintptr_t al = reinterpret_cast<uintptr_t>(p.q_hi) | reinterpret_cast<uintptr_t>(p.k_hi) | reinterpret_cast<uintptr_t>(p.q_lo) |
                       reinterpret_cast<uintptr_t>(p.k_lo) | reinterpret_cast<uintptr_t>(p.G) | reinterpret_cast<uintptr_t>(p.lse_k);
        if (al & 15u) return EEGCLIP_EALIGN;
        tb.p[i] = if_problem{static_cast<const unsigned short*>(p.q_hi), static_cast<const unsigned short*>(p.q_lo), static_cast<const unsigned short*>(p.k_hi),
                             static_cast<const unsigned short*>(p.k_lo), p.part, p.diag, p.lse, p.lse_k, p.G, p.ldg, p.col0, p.weight, fin ? p.part_k : nullptr,
                             fin ? p.diag_k : nullptr, grad ? static_cast<unsigned short*>(p.G_hi) : nullptr, grad ? static_cast<unsigned short*>(p.G_lo) : nullptr};
    }
    return 0;
}

}  // namespace eeg

using namespace eeg;

extern "C" int eegclip_split_bf16(const float* x, void* hi, void* lo, long long n, void* stream) {
    if (!x || !hi || n < 0 || (n & 7)) return EEGCLIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(hi) | reinterpret_cast<uintptr_t>(lo)) & 15u) return EEGCLIP_EALIGN;
    if (n == 0) return 0;
    long long g = (n / 8 + 255) / 256;
    if (g > 4096) g = 4096;
    EEG_LAUNCH(split_bf16_kernel, dim3((unsigned)g), dim3(256), 0, stream, x, static_cast<unsigned short*>(hi), static_cast<unsigned short*>(lo), n / 8);
    return (int)hipGetLastError();
}

extern "C" int eegclip_infonce_fused_supported(int n, int N, int D) { return n >= 64 && N >= 64 && n % 64 == 0 && N % 64 == 0 && D >= 64 && D % 64 == 0; }

extern "C" long long eegclip_infonce_fused_workspace_floats(int n, int N) {
    if (n < 1 || N < 1) return 0;
    return 2LL * (2 * (N / 64)) * n;                              // [2 planes][2 slots per key tile][n], sized for the smaller tile
}

#define EEG_IF_GO2(NP_, TM_, MODE_, NW_, NPRD_, PF_)                                                                                                       \
    EEG_LAUNCH((infonce_tile_kernel<NP_, TM_, MODE_, NW_, NPRD_, PF_>), dim3((unsigned)(nprob * tq * tk)), dim3(64 * (NW_ + NPRD_)),                       \
               (size_t)IF_NS * 2 * NP_ * TM_ * if_geom<if_bk(NP_, TM_)>::ROWB, stream, tb, n, N, D, tq, tk, scale, inv_total, dscale, loss)
#define EEG_IF_GO(NP_, TM_, NW_, NPRD_)                                                                                                                    \
    do {                                                                                                                                                   \
        if (mode == 0) EEG_IF_GO2(NP_, TM_, 0, NW_, NPRD_, false);                                                                                         \
        else EEG_IF_GO2(NP_, TM_, 1, NW_, NPRD_, false);                                                                                                   \
    } while (0)
#define EEG_IF_GO_PF(NP_, TM_, NW_)                                                                                                                        \
    do {                                                                                                                                                   \
        if (mode == 0) EEG_IF_GO2(NP_, TM_, 0, NW_, 0, true);                                                                                              \
        else EEG_IF_GO2(NP_, TM_, 1, NW_, 0, true);                                                                                                        \
    } while (0)

// wave layout: bits 16..17 of `planes` (tests / benches) 1 = 4 waves, 2 = 8 waves (128-tiles only), 3 = 4 MFMA waves + 4 producer waves,
// 0 = the library's choice: the producer form, except 8 waves for 128-tiles with one product (kernel durations at N = 2048, 256 workgroups, one product:
// 12.9 / 12.0 / 12.6 us for 4 / 8 / 4+4 waves; three products: 27.2 / 26.8 / 26.6; 64-tiles, a rank's two 256 x 2048 blocks: 7.8 -> 7.2 and 12.8 -> 10.8 us)
static int if_wsel(int TM, int planes) {
    const int w = (planes >> 16) & 3;
    if (TM == 256) return 2;                                     // the 256-tile exists with 8 waves only (128 accumulator registers per wave)
    if (w == 0) return (TM == 128 && (planes & 0xff) == 1) ? 2 : 3;
    return (w == 2 && TM != 128) ? 1 : w;
}
// MFMA waves of a workgroup; partial slots per key tile = the waves along the keys, 1 for the 256-tile (combined in the workgroup)
static int if_waves(int TM, int planes) { return if_wsel(TM, planes) == 2 ? 8 : 4; }
static int if_slots(int TM, int planes) { return TM == 256 ? 1 : if_waves(TM, planes) / 2; }

static int if_launch_tiles(const if_table& tb, int nprob, int n, int N, int D, int planes, int mode, const float* scale, float inv_total, float* dscale,
                           void* stream, float* loss = nullptr) {
    // the gradient pass keeps 128-tiles unless it finalises the forward's partials itself (their slot layout is the forward tile's): its 256-tile
    // instantiation spills (128 accumulators + the G store's operands) and measured 254 against 231 us at N = 8192
    const int TM = if_tile(n, N, (planes >> 8) & 0xff, planes & 0xff, mode == 0 || loss != nullptr), tq = n / TM, tk = N / TM, wsel = if_wsel(TM, planes);
    const bool no_pf = ((planes >> 18) & 1) != 0;                // bit 18 (benches): the 8-wave forms without the cross-barrier fragment prefetch
    planes &= 0xff;
    if (planes == 1) {
        if (TM == 256) { if (no_pf) EEG_IF_GO(1, 256, 8, 0); else EEG_IF_GO_PF(1, 256, 8); }
        else if (TM == 128) {
            if (wsel == 3) EEG_IF_GO(1, 128, 4, 4);
            else if (wsel == 2) { if (no_pf) EEG_IF_GO(1, 128, 8, 0); else EEG_IF_GO_PF(1, 128, 8); }
            else EEG_IF_GO(1, 128, 4, 0);
        }
        else           { if (wsel == 3) EEG_IF_GO(1, 64, 4, 4); else EEG_IF_GO(1, 64, 4, 0); }
    } else {
        if (TM == 128) {
            if (wsel == 3) EEG_IF_GO(2, 128, 4, 4);
            else if (wsel == 2) { if (no_pf) EEG_IF_GO(2, 128, 8, 0); else EEG_IF_GO_PF(2, 128, 8); }
            else EEG_IF_GO(2, 128, 4, 0);
        }
        else           { if (wsel == 3) EEG_IF_GO(2, 64, 4, 4); else EEG_IF_GO(2, 64, 4, 0); }
    }
    return (int)hipGetLastError();
}

extern "C" int eegclip_infonce_fused_fwd(const eegclip_infonce_problem* probs, int nprob, int n, int N, int D, int planes, int n_total,
                                         const float* scale, float* loss, void* stream) {
    if (!eegclip_infonce_fused_supported(n, N, D) || ((planes & 0xff) != 1 && (planes & 0xff) != 2) || !scale || n_total < 1) return EEGCLIP_EINVAL;
    if_table tb;
    int rc = if_table_from(probs, nprob, planes & 0xff, false, tb);
    if (rc) return rc;
    const float inv_total = 1.0f / (float)n_total;
    rc = if_launch_tiles(tb, nprob, n, N, D, planes, 0, scale, inv_total, nullptr, stream);
    if (rc || !loss) return rc;                                  // loss == NULL: partials only -- eegclip_infonce_fused_grad_finalize finishes them
    const int TMsel = if_tile(n, N, (planes >> 8) & 0xff, planes & 0xff);
    const int Pn = if_slots(TMsel, planes) * (N / TMsel);
    EEG_LAUNCH(infonce_finalize_kernel, dim3((unsigned)((n + 63) / 64), (unsigned)nprob), dim3(256), 512 * sizeof(float), stream, tb, n, Pn, inv_total, loss);
    return (int)hipGetLastError();
}

extern "C" int eegclip_infonce_fused_grad(const eegclip_infonce_problem* probs, int nprob, int n, int N, int D, int planes, int n_total,
                                          const float* scale, float* dscale, void* stream) {
    if (!eegclip_infonce_fused_supported(n, N, D) || ((planes & 0xff) != 1 && (planes & 0xff) != 2) || !scale || !dscale || n_total < 1) return EEGCLIP_EINVAL;
    if_table tb;
    const int rc = if_table_from(probs, nprob, planes & 0xff, true, tb);
    if (rc) return rc;
    return if_launch_tiles(tb, nprob, n, N, D, planes, 1, scale, 1.0f / (float)n_total, dscale, stream);
}

// the gradient pass with the forward's finalize folded in (training: every block of the loss is also differentiated): `blocks` as for eegclip_infonce_fused_grad
// plus part / diag (this block's forward partials) and part_k / diag_k (the swapped block's); lse / lse_k are not read.  Adds the loss terms of BOTH
// blocks of every entry to *loss.  Preceded by eegclip_infonce_fused_fwd(..., loss = NULL) over all the blocks.
extern "C" int eegclip_infonce_fused_grad_finalize(const eegclip_infonce_problem* probs, int nprob, int n, int N, int D, int planes, int n_total,
                                                   const float* scale, float* loss, float* dscale, void* stream) {
    if (!eegclip_infonce_fused_supported(n, N, D) || ((planes & 0xff) != 1 && (planes & 0xff) != 2) || !scale || !loss || !dscale || n_total < 1 || n != N)
        return EEGCLIP_EINVAL;
    if_table tb;
    const int rc = if_table_from(probs, nprob, planes & 0xff, true, tb, true);
    if (rc) return rc;
    return if_launch_tiles(tb, nprob, n, N, D, planes, 1, scale, 1.0f / (float)n_total, dscale, stream, loss);
}
