// The projection head's GEMMs (Retrieval/ATMS_retrieval.py:157-167: Linear 1440 -> 1024, GELU, Linear 1024 -> 1024) and the GEMMs of its backward and of
// the loss gradient at M = the batch (256): C[m][n] = sum_k A[m][k] B[n][k] from bf16 hi | lo PLANES (three products per multiply-add, fp32 accumulate:
// EEGCLIP_PREC_BF16X3), K-PARALLEL ACROSS WORKGROUPS without atomics and without any in-launch hand-off.
//
// Why its own kernel: at M = 256 a 64 x 64 tile grid is 64 .. 92 workgroups on 256 CUs and each walks its K = 1024 .. 1440 serially -- 12 - 20 us per launch
// whatever feeds the tiles (csrc/gemm_x3.hip with fp32 operands, csrc/gemm_planes.hip with planes: a k-tile costs ~0.3 us of fill latency with one workgroup
// per CU).  Splitting K with fp32 atomics on C (what the head did until round 5) pays 2 M atomics for one 256 x 1024 output.  Here a workgroup is (64 x 64
// output tile, K slice) and slice s simply stores its partial tile into SLAB s of the output (slab_stride floats apart): 256 - 368 workgroups of 6 - 11
// k-tiles each, 5 - 7 us per launch.  The slabs are added by the launch that consumes the result anyway -- the head's bias + GELU kernel, its LayerNorm, the
// LayerNorm / GELU backward kernels, the 1x1-conv backward (each takes `slabs, slab_stride`) -- in slice order: bit-reproducible, no extra launch.
// Measured alternative (round 6, tools/round6/bench_head_gemm.py): combining inside the launch -- slabs + agent-scope release fence + ticket, last arriver
// acquires, sums and runs the epilogue -- costs 18 - 34 us at 2 - 8 slices against 14 - 17 unsplit: the release is an L2 write-back per workgroup on this
// eight-L2 part (MI355X_MICROARCH.md: splitk-seam, "cut GEMM -> GEMM seams at this size").  With slices = 1 the kernel runs the epilogue itself (bias,
// pre-activation copy, GELU / GELU', residual, fp32 result, result again as planes).
// The k-loop is csrc/gemm_planes.hip's: 32-k tiles of the four operand planes by LDS-DMA from four producer waves, four MFMA waves (2 x 2, one 32 x 32
// accumulator each, formed transposed: a lane owns one row m and four consecutive n per register quad), one barrier per k-tile, 4 stages of 16 KB.
// Edges: rows >= M / columns >= N of a tile read clamped operand rows and are not stored (any batch size; N = 1440 of the head's input gradient).
// b_kmajor: the B operand given as B[k][n] planes (n contiguous, rows ldb apart) -- what the INPUT-gradient GEMMs contract over: dX = dY W reads the weight
// planes of the forward (W[n_out][k_in]: the contraction index is the ROW), the query gradient the stacked target planes as they are.  The k-tile lands in
// LDS as 32 rows x 128 B and the fragments come out through ds_read_b64_tr_b16 (the LDS transpose read of gfx950, csrc/attention_x3.hip): no transposed
// copy of the weights or the targets is ever made (round 6's first version split them transposed at the start of every step: 40 us of second-stream
// work beside the fused forward, which slowed it by 20 us).
#include "eeg_common.h"

#include <type_traits>

namespace eeg {

constexpr int HG_TM = 64, HG_BK = 32, HG_ROWB = 64, HG_NS = 4, HG_RPI = 16;
constexpr int HG_TILE_B = HG_TM * HG_ROWB;                 // one operand-plane tile: 4 KB
constexpr int HG_STAGE_B = 4 * HG_TILE_B;                  // A hi | B hi | A lo | B lo
constexpr int HG_LDS = HG_NS * HG_STAGE_B;
constexpr int HG_MAX_SLICES = 16;

struct hg_args {
    const unsigned short *a_hi, *a_lo, *b_hi, *b_lo;
    long long lda, ldb;
    int M, N, K, slices;
    long long slab_stride;
    const float* bias;
    float* Cpre;
    long long ldcpre;
    int act;
    const float* aux;
    long long ldaux;
    const float* R;
    long long ldr;
    float* C;
    long long ldc;
    unsigned short *p_hi, *p_lo;
    long long ldp;
    int tiles_n, tiles, chunk;
};

__device__ __forceinline__ int hg_swz_t(int row) { return ((row >> 1) & 1) << 2; }      // k-major B tile: 128-byte rows, 8 chunks; rows r, r + 2 swap chunk halves


__device__ __forceinline__ int hg_swz(int row) { return (row >> 2) & 3; }

typedef short hg_s4 __attribute__((ext_vector_type(4)));
// LDS transpose read: within a 16-lane group, lane i receives as element j the (i & 3)-th 16-bit element of the 8 bytes addressed by lane 4 j + (i >> 2)
__device__ __forceinline__ hg_s4 hg_tr_read(const unsigned char* p) {
#if defined(EEG_EMU)
    const int lane = hipemu::cur->lane, g = lane >> 4, i = lane & 15;
    hg_s4 r;
    for (int j = 0; j < 4; ++j) {
        const unsigned long long src = hipemu::shfl_idx((unsigned long long)(uintptr_t)p, 16 * g + 4 * j + (i >> 2));
        r[j] = reinterpret_cast<const short*>((uintptr_t)src)[i & 3];
    }
    return r;
#else
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) hg_s4*)(p));
#endif
}

template <bool BT>
__global__ __launch_bounds__(512, 2) void head_gemm_kernel(const hg_args a) {
    EEG_LDS_BASE(unsigned char, lds);
    // workgroup b runs on XCD b % 8 (observed dispatch; speed only): an XCD takes a contiguous range of (tile, slice) pairs, all slices of a tile on one XCD
    const int logical = (int)(blockIdx.x & 7) * a.chunk + (int)(blockIdx.x >> 3);
    const int tile = logical / a.slices, slice = logical - tile * a.slices;
    if (tile >= a.tiles) return;
    const int m0 = (tile / a.tiles_n) * HG_TM, n0 = (tile % a.tiles_n) * HG_TM;
    const int t = threadIdx.x, lane = t & 63, wave = wave_uniform(t >> 6);
    const int ktiles_all = a.K / HG_BK;
    const int kt0 = (int)((long long)ktiles_all * slice / a.slices), kt1 = (int)((long long)ktiles_all * (slice + 1) / a.slices);
    const int ktiles = kt1 - kt0;
    auto wait_tile = [&](int kt, auto dpt_c) {
        constexpr int DPT = decltype(dpt_c)::value;
        const int newer = ktiles - 1 - kt < HG_NS - 2 ? ktiles - 1 - kt : HG_NS - 2;
        if (newer >= 2) wait_vmcnt<2 * DPT>();
        else if (newer == 1) wait_vmcnt<DPT>();
        else wait_vmcnt<0>();
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int wq = (wave & 3) >> 1, wk = wave & 1, r32 = lane & 31, h = lane >> 5;
    if (wave >= 4) {                                         // ---------------- producer: plane tile o = wave - 4 (A hi, B hi, A lo, B lo), 4 DMA instructions of 16 rows
        const int o = wave - 4;
        const unsigned short* base = o == 0 ? a.a_hi : o == 1 ? a.b_hi : o == 2 ? a.a_lo : a.b_lo;
        const unsigned short* src[4];
        long long kstep;                                     // elements from one k-tile to the next
        if (BT && (o & 1)) {
            // k-major B: the tile is 32 contraction rows x 64 columns (128 B per row): a DMA instruction covers 8 rows of 8 chunks
            const int drow = lane >> 3, dpos = lane & 7;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 8 * i + drow;
                int col = n0 + 8 * (dpos ^ hg_swz_t(row));
                col = col < a.N - 8 ? col : a.N - 8;                       // (edge tiles: a clamped chunk, its columns are never stored; N % 8 == 0)
                src[i] = base + ((long long)kt0 * HG_BK + row) * a.ldb + col;
            }
            kstep = (long long)HG_BK * a.ldb;
        } else {
            const int drow = lane >> 2, dpos = lane & 3;
            const long long ld = (o & 1) ? a.ldb : a.lda;
            const int r0 = (o & 1) ? n0 : m0, rmax = ((o & 1) ? a.N : a.M) - 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = HG_RPI * i + drow;
                const int gr = r0 + row < rmax ? r0 + row : rmax;          // (edge tiles: clamped rows, never stored)
                src[i] = base + (long long)gr * ld + 8 * (dpos ^ hg_swz(row)) + (long long)kt0 * HG_BK;
            }
            kstep = HG_BK;
        }
        auto issue_tile = [&](int kt) {
            unsigned char* st = lds + (kt % HG_NS) * HG_STAGE_B + o * HG_TILE_B;
#pragma unroll
            for (int i = 0; i < 4; ++i) lds_dma16(st + i * 1024, src[i] + kt * kstep);
        };
#pragma unroll
        for (int p = 0; p < HG_NS - 1; ++p)
            if (p < ktiles) issue_tile(p);
        for (int kt = 0; kt < ktiles; ++kt) {
            wait_tile(kt, std::integral_constant<int, 4>{});
            raw_barrier();                                   // the one meeting point of the two kinds of waves per k-tile
            if (kt + HG_NS - 1 < ktiles) issue_tile(kt + HG_NS - 1);
        }
    } else {                                                 // ---------------- MFMA waves: wave (wq, wk) owns rows m0 + 32 wq .., columns n0 + 32 wk ..
        int foa[2], fob[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ra = wq * 32 + r32, rb = wk * 32 + r32;
            foa[s] = ra * HG_ROWB + (((2 * s + h) ^ hg_swz(ra)) & 3) * 16;
            fob[s] = HG_TILE_B + rb * HG_ROWB + (((2 * s + h) ^ hg_swz(rb)) & 3) * 16;
        }
        if (BT) {
            // k-major B tile [32 k][64 n]: MFMA row n = 32 wk + (lane & 31) = 32 wk + 16 cg + li, k slots 0 .. 7 of step s = tile rows 16 s + 8 h + 0 .. 7: two
            // transpose reads (4 rows x 16 columns per 16-lane group each); lane li addresses row .. + (li >> 2), columns 16 cg + 4 (li & 3) ..
            const int li = lane & 15, cg = (lane >> 4) & 1, row0 = 8 * h + (li >> 2);
            const int chunk = 4 * wk + 2 * cg + ((li & 3) >> 1);
            fob[0] = HG_TILE_B + row0 * 128 + ((chunk ^ hg_swz_t(row0)) << 4) + 8 * (li & 1);          // (rows + 4 / + 16: the same swizzle term)
            fob[1] = fob[0] + 16 * 128;
        }
        for (int kt = 0; kt < ktiles; ++kt) {
            raw_barrier();
            const unsigned char* st = lds + (kt % HG_NS) * HG_STAGE_B;
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                ah[s] = *reinterpret_cast<const bf16x8*>(st + foa[s]);
                al[s] = *reinterpret_cast<const bf16x8*>(st + 2 * HG_TILE_B + foa[s]);
                if (BT) {
                    const hg_s4 x = hg_tr_read(st + fob[s]), y = hg_tr_read(st + fob[s] + 4 * 128);
                    const hg_s4 xl = hg_tr_read(st + 2 * HG_TILE_B + fob[s]), yl = hg_tr_read(st + 2 * HG_TILE_B + fob[s] + 4 * 128);
                    bh[s] = bf16x8{x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
                    bl[s] = bf16x8{xl[0], xl[1], xl[2], xl[3], yl[0], yl[1], yl[2], yl[3]};
                } else {
                    bh[s] = *reinterpret_cast<const bf16x8*>(st + fob[s]);
                    bl[s] = *reinterpret_cast<const bf16x8*>(st + 2 * HG_TILE_B + fob[s]);
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                acc = mfma_bf16_32x32x16(bl[s], ah[s], acc);     // D[n = .. + row(reg, h)][m = .. + r32]
                acc = mfma_bf16_32x32x16(bh[s], al[s], acc);
                acc = mfma_bf16_32x32x16(bh[s], ah[s], acc);
            }
        }
    }
    // lane (r32, h) of MFMA wave (wq, wk) owns tile row 32 wq + r32; registers 4 eq .. 4 eq + 3 are tile columns 32 wk + 8 eq + 4 h .. + 3
    if (wave >= 4) return;
    const int tr = wq * 32 + r32, tc = wk * 32 + 4 * h;
    const int m = m0 + tr;
    if (m >= a.M) return;
    if (a.slices > 1) {                                      // ---------------- partial tile -> slab `slice` of C; the consumer launch adds the slabs
        float* mine = a.C + (long long)slice * a.slab_stride + (long long)m * a.ldc;
#pragma unroll
        for (int eq = 0; eq < 4; ++eq) {
            const int n = n0 + tc + 8 * eq;
            if (n < a.N) *reinterpret_cast<f32x4*>(mine + n) = f32x4{acc[4 * eq], acc[4 * eq + 1], acc[4 * eq + 2], acc[4 * eq + 3]};
        }
        return;
    }
    // ---------------- epilogue (slices == 1)
#pragma unroll
    for (int eq = 0; eq < 4; ++eq) {
        const int n = n0 + tc + 8 * eq;
        if (n >= a.N) continue;
        f32x4 v = f32x4{acc[4 * eq], acc[4 * eq + 1], acc[4 * eq + 2], acc[4 * eq + 3]};
        if (a.bias) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bv[e];
        }
        if (a.Cpre) *reinterpret_cast<f32x4*>(a.Cpre + (long long)m * a.ldcpre + n) = v;
        if (a.act == EEGCLIP_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        } else if (a.act == EEGCLIP_ACT_GELU_GRAD) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(a.aux + (long long)m * a.ldaux + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= gelu_erf_grad(xv[e]);
        }
        if (a.R) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(a.R + (long long)m * a.ldr + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += rv[e];
        }
        if (a.C) *reinterpret_cast<f32x4*>(a.C + (long long)m * a.ldc + n) = v;
        if (a.p_hi) {
            u32x2_t hi, lo;
            x3_split4(v[0], v[1], v[2], v[3], hi, lo);
            *reinterpret_cast<u32x2_t*>(a.p_hi + (long long)m * a.ldp + n) = hi;
            *reinterpret_cast<u32x2_t*>(a.p_lo + (long long)m * a.ldp + n) = lo;
        }
    }
}

}  // namespace eeg

using namespace eeg;

static int hg_tiles(const eegclip_head_gemm_desc* d) { return ((d->M + HG_TM - 1) / HG_TM) * ((d->N + HG_TM - 1) / HG_TM); }

extern "C" int eegclip_head_gemm_slices(int M, int N, int K) {
    // (tile, slice) workgroups for about one per CU (measured at M = 256: 3 - 4 slices of N = 1024, 2 of N = 1440; more slices only add slab traffic), at least
    // four 32-k tiles per slice
    if (M < 1 || N < 1 || K < HG_BK) return 1;
    const int tiles = ((M + HG_TM - 1) / HG_TM) * ((N + HG_TM - 1) / HG_TM);
    int s = 256 / tiles;
    const int cap = K / (4 * HG_BK);
    if (s > cap) s = cap;
    if (s > HG_MAX_SLICES) s = HG_MAX_SLICES;
    return s < 1 ? 1 : s;
}

extern "C" int eegclip_head_gemm(const eegclip_head_gemm_desc* d, void* stream) {
    if (!d || !d->a_hi || !d->a_lo || !d->b_hi || !d->b_lo || d->M < 1 || d->N < 4 || d->K < HG_BK || (d->N & 3) || d->K % HG_BK) return EEGCLIP_EINVAL;
    if (d->lda < d->K || (d->lda & 7) || (d->ldb & 7)) return EEGCLIP_EINVAL;
    if (d->b_kmajor ? (d->ldb < d->N || (d->N & 7)) : d->ldb < d->K) return EEGCLIP_EINVAL;
    if (d->slices < 1 || d->slices > HG_MAX_SLICES || d->slices > d->K / HG_BK) return EEGCLIP_EINVAL;
    if (d->slices > 1) {                                     // slabs of C only: the consumer runs the epilogue
        if (!d->C || d->slab_stride < (long long)(d->M - 1) * d->ldc + d->N || (d->slab_stride & 3) || d->bias || d->Cpre || d->act != EEGCLIP_ACT_NONE || d->R || d->p_hi ||
            d->p_lo)
            return EEGCLIP_EINVAL;
    }
    if (!d->C && !d->Cpre && !d->p_hi) return EEGCLIP_EINVAL;
    if ((d->p_hi != nullptr) != (d->p_lo != nullptr)) return EEGCLIP_EINVAL;
    if ((d->C && (d->ldc < d->N || (d->ldc & 3))) || (d->Cpre && (d->ldcpre < d->N || (d->ldcpre & 3))) || (d->R && (d->ldr < d->N || (d->ldr & 3))) ||
        (d->p_hi && (d->ldp < d->N || (d->ldp & 3))))
        return EEGCLIP_EINVAL;
    if (d->act != EEGCLIP_ACT_NONE && d->act != EEGCLIP_ACT_GELU && d->act != EEGCLIP_ACT_GELU_GRAD) return EEGCLIP_EINVAL;
    if (d->act == EEGCLIP_ACT_GELU_GRAD && (!d->aux || d->ldaux < d->N || (d->ldaux & 3))) return EEGCLIP_EINVAL;
    const uintptr_t al16 = reinterpret_cast<uintptr_t>(d->a_hi) | reinterpret_cast<uintptr_t>(d->a_lo) | reinterpret_cast<uintptr_t>(d->b_hi) |
                           reinterpret_cast<uintptr_t>(d->b_lo) | reinterpret_cast<uintptr_t>(d->C) | reinterpret_cast<uintptr_t>(d->Cpre) |
                           reinterpret_cast<uintptr_t>(d->bias) | reinterpret_cast<uintptr_t>(d->R) | reinterpret_cast<uintptr_t>(d->aux);
    const uintptr_t al8 = reinterpret_cast<uintptr_t>(d->p_hi) | reinterpret_cast<uintptr_t>(d->p_lo);
    if ((al16 & 15u) || (al8 & 7u)) return EEGCLIP_EALIGN;
    hg_args a{static_cast<const unsigned short*>(d->a_hi), static_cast<const unsigned short*>(d->a_lo), static_cast<const unsigned short*>(d->b_hi),
              static_cast<const unsigned short*>(d->b_lo), d->lda, d->ldb, d->M, d->N, d->K, d->slices, d->slab_stride, d->bias, d->Cpre, d->ldcpre,
              d->act, d->aux, d->ldaux, d->R, d->ldr, d->C, d->ldc, static_cast<unsigned short*>(d->p_hi), static_cast<unsigned short*>(d->p_lo), d->ldp,
              0, 0, 0};
    a.tiles_n = (d->N + HG_TM - 1) / HG_TM;
    a.tiles = hg_tiles(d);
    a.chunk = (a.tiles + 7) / 8 * d->slices;                 // whole tiles per XCD range: the slices of a tile share one L2's view of its operand rows
    if (d->b_kmajor) EEG_LAUNCH(head_gemm_kernel<true>, dim3(8 * a.chunk), dim3(512), HG_LDS, stream, a);
    else             EEG_LAUNCH(head_gemm_kernel<false>, dim3(8 * a.chunk), dim3(512), HG_LDS, stream, a);
    return (int)hipGetLastError();
}
