// C = A B^T from bf16 hi | lo PLANES:  C[m][n] = sum_k A[m][k] B[n][k], A (M, K) and B (N, K) each given as two k-contiguous bf16 planes (value = hi + lo,
// eegclip_split_bf16 / eegclip_split_rows / the plane outputs of the producing kernels), three products per multiply-add (hi hi + hi lo + lo hi, fp32
// accumulate: the arithmetic of EEGCLIP_PREC_BF16X3).  The Linear layers of the diffusion prior (Generation/diffusion_prior.py:167-203): M = the batch
// (1024), N = 64 .. 2880 features, K = 64 .. 1024 -- shapes on which the general split-bf16 GEMM (csrc/gemm_x3.hip: fp32 operands split into planes by
// every workgroup while it stages them, ~1 us per 64-k tile of vector work) took ~20 us per launch whatever the size, 64 launches per training step.
// Here nothing is converted: the k-loop is the InfoNCE tile kernel's (csrc/infonce_fused.hip) -- 64 x 64 output tile per workgroup, 32-k tiles of the four
// operand planes go global -> LDS by LDS-DMA from four PRODUCER waves (the vector-memory path of a CU fills ~64 B/clk; a wave that issues the DMA and
// the MFMAs serialises the two, tools/micro/tile_chain.hip), four MFMA waves (2 x 2, one 32 x 32 accumulator each) meet them at one barrier per k-tile,
// 4 LDS stages of 16 KB (two workgroups per CU).  The product is formed transposed (MFMA rows = n): a lane owns one row m and four consecutive n per
// register quad -- 16-byte stores of C and 8-byte stores of its planes.
// Epilogue: + bias[n], optional copy of the pre-activation (Cpre), SiLU, + R[m][n], accumulate onto C, and the result (or Cpre) again as hi | lo planes
// for the next GEMM that takes it as an operand.
#include "eeg_common.h"

#include <type_traits>

namespace eeg {

constexpr int GP_TM = 64, GP_BK = 32, GP_ROWB = 64, GP_NS = 4, GP_RPI = 16;       // tile edge, k-tile, bytes per LDS row, stages, rows per DMA instruction
constexpr int GP_TILE_B = GP_TM * GP_ROWB;                                         // one operand-plane tile: 4 KB
constexpr int GP_STAGE_B = 4 * GP_TILE_B;                                          // A hi | B hi | A lo | B lo
constexpr int GP_LDS = GP_NS * GP_STAGE_B;

struct gp_args {
    const unsigned short *a_hi, *a_lo, *b_hi, *b_lo;
    long long lda, ldb;                                      // elements between rows
    int M, N, K;
    float* C;
    long long ldc;
    float* Cpre;
    long long ldcpre;
    const float* bias;
    const float* R;
    long long ldr;
    unsigned short *p_hi, *p_lo;
    long long ldp;
    int act, accumulate, planes_of;                          // planes_of: 0 none, 1 the result C, 2 Cpre
    int tiles_n, tiles, chunk;
};

__device__ __forceinline__ int gp_swz(int row) { return (row >> 2) & 3; }          // 64-byte rows: 4 chunks of 16 bytes (if_geom<2>)

__global__ __launch_bounds__(512, 2) void gemm_planes_kernel(const gp_args a) {
    EEG_LDS_BASE(unsigned char, lds);
    // workgroup b runs on XCD b % 8 (observed dispatch; speed only): every XCD takes a contiguous range of tiles -- neighbours share A rows / B rows in one L2
    const int logical = (int)(blockIdx.x & 7) * a.chunk + (int)(blockIdx.x >> 3);
    if (logical >= a.tiles) return;
    const int m0 = (logical / a.tiles_n) * GP_TM, n0 = (logical % a.tiles_n) * GP_TM;
    const int t = threadIdx.x, lane = t & 63, wave = wave_uniform(t >> 6);
    const int ktiles = a.K / GP_BK;
    auto wait_tile = [&](int kt, auto dpt_c) {
        constexpr int DPT = decltype(dpt_c)::value;
        const int newer = ktiles - 1 - kt < GP_NS - 2 ? ktiles - 1 - kt : GP_NS - 2;
        if (newer >= 2) wait_vmcnt<2 * DPT>();
        else if (newer == 1) wait_vmcnt<DPT>();
        else wait_vmcnt<0>();
    };
    if (wave >= 4) {                                         // ---------------- producer: plane tile o = wave - 4 (A hi, B hi, A lo, B lo), 4 DMA instructions of 16 rows
        const int o = wave - 4, drow = lane >> 2, dpos = lane & 3;
        const unsigned short* base = o == 0 ? a.a_hi : o == 1 ? a.b_hi : o == 2 ? a.a_lo : a.b_lo;
        const long long ld = (o & 1) ? a.ldb : a.lda;
        const int r0 = (o & 1) ? n0 : m0;
        const unsigned short* src[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = GP_RPI * i + drow;
            src[i] = base + (long long)(r0 + row) * ld + 8 * (dpos ^ gp_swz(row));
        }
        auto issue_tile = [&](int kt) {
            unsigned char* st = lds + (kt % GP_NS) * GP_STAGE_B + o * GP_TILE_B;
#pragma unroll
            for (int i = 0; i < 4; ++i) lds_dma16(st + GP_RPI * i * GP_ROWB, src[i] + kt * GP_BK);
        };
#pragma unroll
        for (int p = 0; p < GP_NS - 1; ++p)
            if (p < ktiles) issue_tile(p);
        for (int kt = 0; kt < ktiles; ++kt) {
            wait_tile(kt, std::integral_constant<int, 4>{});
            raw_barrier();                                   // the one meeting point of the two kinds of waves per k-tile
            if (kt + GP_NS - 1 < ktiles) issue_tile(kt + GP_NS - 1);
        }
        return;
    }
    // ---------------- MFMA waves: wave (wq, wk) owns rows m0 + 32 wq .., columns n0 + 32 wk ..
    const int wq = wave >> 1, wk = wave & 1, r32 = lane & 31, h = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    int foa[2], fob[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int ra = wq * 32 + r32, rb = wk * 32 + r32;
        foa[s] = ra * GP_ROWB + (((2 * s + h) ^ gp_swz(ra)) & 3) * 16;
        fob[s] = GP_TILE_B + rb * GP_ROWB + (((2 * s + h) ^ gp_swz(rb)) & 3) * 16;
    }
    for (int kt = 0; kt < ktiles; ++kt) {
        raw_barrier();
        const unsigned char* st = lds + (kt % GP_NS) * GP_STAGE_B;
        bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            ah[s] = *reinterpret_cast<const bf16x8*>(st + foa[s]);
            bh[s] = *reinterpret_cast<const bf16x8*>(st + fob[s]);
            al[s] = *reinterpret_cast<const bf16x8*>(st + 2 * GP_TILE_B + foa[s]);
            bl[s] = *reinterpret_cast<const bf16x8*>(st + 2 * GP_TILE_B + fob[s]);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            acc = mfma_bf16_32x32x16(bl[s], ah[s], acc);     // D[n = .. + row(reg, h)][m = .. + r32]
            acc = mfma_bf16_32x32x16(bh[s], al[s], acc);
            acc = mfma_bf16_32x32x16(bh[s], ah[s], acc);
        }
    }
    // ---------------- epilogue: lane (r32, h) owns row m; registers 4 eq .. 4 eq + 3 are columns nb + 8 eq + 4 h .. + 3
    const int m = m0 + wq * 32 + r32, nb = n0 + wk * 32;
    const bool acc_c = a.accumulate != 0;
#pragma unroll
    for (int eq = 0; eq < 4; ++eq) {
        const int n = nb + 8 * eq + 4 * h;
        f32x4 v = f32x4{acc[4 * eq], acc[4 * eq + 1], acc[4 * eq + 2], acc[4 * eq + 3]};
        if (a.bias) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bv[e];
        }
        if (a.Cpre) *reinterpret_cast<f32x4*>(a.Cpre + (long long)m * a.ldcpre + n) = v;
        if (a.planes_of == 2) {
            u32x2_t hi, lo;
            x3_split4(v[0], v[1], v[2], v[3], hi, lo);
            *reinterpret_cast<u32x2_t*>(a.p_hi + (long long)m * a.ldp + n) = hi;
            *reinterpret_cast<u32x2_t*>(a.p_lo + (long long)m * a.ldp + n) = lo;
        }
        if (a.act == EEGCLIP_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = silu(v[e]);
        }
        if (a.R) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(a.R + (long long)m * a.ldr + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += rv[e];
        }
        if (a.C) {
            float* cp = a.C + (long long)m * a.ldc + n;
            if (acc_c) {
                const f32x4 cv = *reinterpret_cast<const f32x4*>(cp);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += cv[e];
            }
            *reinterpret_cast<f32x4*>(cp) = v;
        }
        if (a.planes_of == 1) {
            u32x2_t hi, lo;
            x3_split4(v[0], v[1], v[2], v[3], hi, lo);
            *reinterpret_cast<u32x2_t*>(a.p_hi + (long long)m * a.ldp + n) = hi;
            *reinterpret_cast<u32x2_t*>(a.p_lo + (long long)m * a.ldp + n) = lo;
        }
    }
}

// fp32 (rows, cols) matrices -> bf16 hi | lo planes of their TRANSPOSES ([cols][ld_out]): the weights a dX plane GEMM contracts over the OUTPUT index
// (dX = dY W: B operand = W^T, k-contiguous).  64 x 64 tiles through LDS: 16-byte reads along the source rows, 8-byte plane writes along the source
// columns.  (eegclip_split_rows' transposing path reads one element per lane with the lanes ld_src apart -- fine for the encoder's 250-wide weights,
// 4 M elements per step here.)
constexpr int GPT_MAX = 24;
struct gpt_entry {
    const float* src;
    unsigned short *hi, *lo;
    long long ld_src, ld_out;
    int rows, cols, first_block, tiles_c;
};
struct gpt_table {
    gpt_entry e[GPT_MAX];
    int n;
};
__global__ __launch_bounds__(256) void split_transpose_kernel(const gpt_table tb) {
    EEG_LDS_BASE(float, tile);                               // [64 source columns][65]
    int ei = 0;
    for (int i = 1; i < tb.n; ++i) ei += (int)blockIdx.x >= tb.e[i].first_block ? 1 : 0;
    const gpt_entry& E = tb.e[ei];
    const int local = (int)blockIdx.x - E.first_block;
    const int r0 = (local / E.tiles_c) * 64, c0 = (local % E.tiles_c) * 64;
    const int t = threadIdx.x, q = t & 15, rr = t >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = rr + 16 * i;
        const f32x4 v = c0 + 4 * q < E.cols ? *reinterpret_cast<const f32x4*>(E.src + (long long)(r0 + r) * E.ld_src + c0 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[(4 * q + e) * 65 + r] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = rr + 16 * i;
        if (c0 + c >= E.cols) continue;                      // (cols need only be a multiple of 4: the head's 1440-column weight)
        const float* tr = tile + c * 65 + 4 * q;
        u32x2_t h, l;
        x3_split4(tr[0], tr[1], tr[2], tr[3], h, l);
        const long long o = (long long)(c0 + c) * E.ld_out + r0 + 4 * q;
        *reinterpret_cast<u32x2_t*>(E.hi + o) = h;
        *reinterpret_cast<u32x2_t*>(E.lo + o) = l;
    }
}

}  // namespace eeg

using namespace eeg;

extern "C" int eegclip_split_transpose(const eegclip_split_item* items, int n, void* stream) {
    if (!items || n < 1 || n > GPT_MAX) return EEGCLIP_EINVAL;
    gpt_table tb;
    tb.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const eegclip_split_item& it = items[i];
        if (!it.src || !it.hi || !it.lo || it.rows < 64 || it.cols < 4 || (it.rows & 63) || (it.cols & 3) || it.ld_src < it.cols || it.ld_out < it.rows ||
            (it.ld_src & 3) || (it.ld_out & 3) || it.copy)
            return EEGCLIP_EINVAL;
        if ((reinterpret_cast<uintptr_t>(it.src) & 15u) || ((reinterpret_cast<uintptr_t>(it.hi) | reinterpret_cast<uintptr_t>(it.lo)) & 7u)) return EEGCLIP_EALIGN;
        tb.e[i] = gpt_entry{it.src, static_cast<unsigned short*>(it.hi), static_cast<unsigned short*>(it.lo), it.ld_src, it.ld_out, it.rows, it.cols, blocks,
                            (it.cols + 63) / 64};
        blocks += (it.rows / 64) * ((it.cols + 63) / 64);
    }
    EEG_LAUNCH(split_transpose_kernel, dim3((unsigned)blocks), dim3(256), 64 * 65 * sizeof(float), stream, tb);
    return (int)hipGetLastError();
}

extern "C" int eegclip_gemm_planes(const eegclip_gemm_planes_desc* d, void* stream) {
    if (!d || !d->a_hi || !d->a_lo || !d->b_hi || !d->b_lo || d->M < GP_TM || d->N < GP_TM || d->K < GP_BK || d->M % GP_TM || d->N % GP_TM || d->K % GP_BK)
        return EEGCLIP_EINVAL;
    if (d->lda < d->K || d->ldb < d->K || (d->lda & 7) || (d->ldb & 7)) return EEGCLIP_EINVAL;
    if (!d->C && !d->Cpre && !d->planes_of) return EEGCLIP_EINVAL;
    if ((d->C && (d->ldc < d->N || (d->ldc & 3))) || (d->Cpre && (d->ldcpre < d->N || (d->ldcpre & 3))) || (d->R && (d->ldr < d->N || (d->ldr & 3)))) return EEGCLIP_EINVAL;
    if (d->planes_of < 0 || d->planes_of > 2 || (d->planes_of && (!d->p_hi || !d->p_lo || d->ldp < d->N || (d->ldp & 3)))) return EEGCLIP_EINVAL;
    if (d->accumulate && !d->C) return EEGCLIP_EINVAL;
    if (d->act != EEGCLIP_ACT_NONE && d->act != EEGCLIP_ACT_SILU) return EEGCLIP_EINVAL;
    const uintptr_t al16 = reinterpret_cast<uintptr_t>(d->a_hi) | reinterpret_cast<uintptr_t>(d->a_lo) | reinterpret_cast<uintptr_t>(d->b_hi) |
                           reinterpret_cast<uintptr_t>(d->b_lo) | reinterpret_cast<uintptr_t>(d->C) | reinterpret_cast<uintptr_t>(d->Cpre) |
                           reinterpret_cast<uintptr_t>(d->bias) | reinterpret_cast<uintptr_t>(d->R);
    const uintptr_t al8 = reinterpret_cast<uintptr_t>(d->p_hi) | reinterpret_cast<uintptr_t>(d->p_lo);
    if ((al16 & 15u) || (al8 & 7u)) return EEGCLIP_EALIGN;
    gp_args a{static_cast<const unsigned short*>(d->a_hi), static_cast<const unsigned short*>(d->a_lo), static_cast<const unsigned short*>(d->b_hi),
              static_cast<const unsigned short*>(d->b_lo), d->lda, d->ldb, d->M, d->N, d->K, d->C, d->ldc, d->Cpre, d->ldcpre, d->bias, d->R, d->ldr,
              static_cast<unsigned short*>(d->p_hi), static_cast<unsigned short*>(d->p_lo), d->ldp, d->act, d->accumulate, d->planes_of, 0, 0, 0};
    a.tiles_n = d->N / GP_TM;
    a.tiles = a.tiles_n * (d->M / GP_TM);
    a.chunk = (a.tiles + 7) / 8;
    EEG_LAUNCH(gemm_planes_kernel, dim3(8 * a.chunk), dim3(512), GP_LDS, stream, a);
    return (int)hipGetLastError();
}
