// Generic strided fp32 GEMM on the f32 matrix cores (v_mfma_f32_16x16x4_f32: exact f32, 157 TF peak).
//
//   C[m,n] (+)= epilogue(alpha * sum_k A[m,k] B[k,n])        -- see include/eegclip.h: eegclip_gemm_desc
//
// Two kernels share one epilogue:
//   gemm_f32_fast_kernel  plain-stride operands with even leading dimensions (every Linear of the encoder, forward and backward):
//                         8-byte global loads, no branches in the staging path, ds_read_b64 operand fetches, XCD-aware tile order
//   gemm_f32_kernel       everything else (two-level index maps, odd sizes): element-wise guarded staging
// Tiling (CDNA4): 64 x 64 output tile per 256-thread workgroup = 4 wavefronts in a 2x2 grid, each wave owning 32x32 as 2x2 MFMA
// 16x16 accumulators; BK = 32 (128x128 / 128x64 tiles and an LDS double buffer were built and measured slower on every shape of
// this path: at K ~ 250 occupancy, not per-wave reuse, hides the latency).  Global->register prefetch of k-tile t+1 overlaps the
// MFMAs of tile t (a second register stage -- tile t+2 in flight as well -- costs 32 VGPRs and two waves/SIMD: 94 vs 84 us on
// 16384x744x250, measured).  None of the encoder's dimensions (250, 248, 63, 36, 1440, 2520) is tile aligned: padding lives only in LDS
// (zeros) or in clamped addresses, never in HBM.
#include "eeg_common.h"
#include "gemm_epilogue.h"

#include <stdio.h>
#include <stdlib.h>

namespace eeg {

constexpr int G_BK = 32;
constexpr int G_BT = 64;        // tile edge (M and N)
constexpr int G_THREADS = 256;

// =================================================================================================================================
// general kernel.  LDS image: element (k, m) of a BK x 64 operand tile lives at k*64 + (m ^ swz(k)), swz(k) = ((k&1)<<4) | (k>>1):
//   * MFMA operand read (16 lanes walk m at k, the next 16 at k+1, ds_read_b32 services 32 lanes per cycle over 32 banks): the
//     two 16-float runs land in opposite halves of the 32 banks (bit 4 of swz = k&1)                           -> conflict free
//   * k-contiguous staging write (32 lanes walk k at one m): swz is a bijection of 0..31                         -> conflict free
//   * m-contiguous staging write (lanes walk m at one k): XOR with a constant permutes the 32 banks              -> conflict free
// (a padded stride of 17 mod 32 measured 1.5 conflict cycles per LDS instruction: SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS)
__device__ __forceinline__ int g_swz(int k) { return ((k & 1) << 4) | ((k >> 1) & 15); }

template <bool A_KC, bool B_KC, bool PLAIN>
__global__ __launch_bounds__(G_THREADS) void gemm_f32_kernel(const eegclip_gemm_desc d) {
    constexpr int E = (G_BT * G_BK) / G_THREADS;   // staged elements per thread and operand
    EEG_LDS_BASE(float, lds);
    float* As = lds;                    // [G_BK][64]
    float* Bs = lds + G_BK * G_BT;      // [G_BK][64]

    const int t = threadIdx.x;
    const int lane = t & 63, wave = wave_uniform(t >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.y * G_BT, n0 = blockIdx.x * G_BT;
    int kt_begin, kt_end;
    gemm_k_slice<G_BK>(d, blockIdx.z, kt_begin, kt_end);

    // ---- per-thread staging coordinates -------------------------------------------------------------
    // m-contiguous operand: lane walks m (coalesced), k = t / 64 + 4 i.   k-contiguous: lane walks k, m = (t>>5) + 8 i.
    int a_row[E], a_k[E];
    long long a_off[E];
    bool a_ok[E];
    int b_col[E], b_k[E];
    long long b_off[E];
    bool b_ok[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        if (A_KC) { a_k[i] = t & 31; a_row[i] = (t >> 5) + 8 * i; }
        else      { a_row[i] = t % G_BT; a_k[i] = t / G_BT + (G_THREADS / G_BT) * i; }
        a_ok[i] = (m0 + a_row[i]) < d.M;
        a_off[i] = a_ok[i] ? goff<PLAIN>(d.Am, m0 + a_row[i]) : 0;
    }
#pragma unroll
    for (int i = 0; i < E; ++i) {
        if (B_KC) { b_k[i] = t & 31; b_col[i] = (t >> 5) + 8 * i; }
        else      { b_col[i] = t % G_BT; b_k[i] = t / G_BT + (G_THREADS / G_BT) * i; }
        b_ok[i] = (n0 + b_col[i]) < d.N;
        b_off[i] = b_ok[i] ? goff<PLAIN>(d.Bn, n0 + b_col[i]) : 0;
    }

    float ra[E], rb[E];
    auto load_tile = [&](int kt) {
        const int k0 = kt * G_BK;
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int ka = k0 + a_k[i];
            ra[i] = (a_ok[i] && ka < d.K) ? d.A[a_off[i] + goff<PLAIN>(d.Ak, ka)] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int kb = k0 + b_k[i];
            rb[i] = (b_ok[i] && kb < d.K) ? d.B[b_off[i] + goff<PLAIN>(d.Bk, kb)] : 0.f;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < E; ++i) As[a_k[i] * G_BT + (a_row[i] ^ g_swz(a_k[i]))] = ra[i];
#pragma unroll
        for (int i = 0; i < E; ++i) Bs[b_k[i] * G_BT + (b_col[i] ^ g_swz(b_k[i]))] = rb[i];
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool do_rowsum = d.rowsum_a != nullptr && blockIdx.x == 0;
    float rowsum = 0.f;
    if (kt_begin < kt_end) load_tile(kt_begin);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        store_tile();
        __syncthreads();
        if (kt + 1 < kt_end) load_tile(kt + 1);     // global -> registers, in flight under the MFMAs of tile kt
        if (do_rowsum && t < G_BT) {                // first wave of the n = 0 tiles: row sums of the staged A tile (bias gradients)
#pragma unroll
            for (int k = 0; k < G_BK; ++k) rowsum += As[k * G_BT + (t ^ g_swz(k))];
        }
        const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < G_BK / 4; ++kk) {
            const int kq = kk * 4 + fq;
            const int sw = g_swz(kq);
            float av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = As[kq * G_BT + ((wr * 32 + 16 * i + fr) ^ sw)];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = Bs[kq * G_BT + ((wc * 32 + 16 * j + fr) ^ sw)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma_f32_16x16x4(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
    if (do_rowsum && t < G_BT && m0 + t < d.M) atomicAdd(d.rowsum_a + m0 + t, rowsum);
    gemm_epilogue<PLAIN>(d, acc, m0, n0, wr, wc, lane, blockIdx.z == 0);
}

// =================================================================================================================================
// fast kernel: plain strides, leading dimensions / contiguous extents even, base pointers 8-byte aligned, offsets < 2^31.
//
// LDS images:
//   k-contiguous operand ("KC", e.g. X and W of Y = X W^T): [row][40]; staged with ds_write_b64 (16 lanes = one 128-byte row),
//       fetched with ds_read_b128: lane (fr, g) takes k = 16h + 4g + {0..3} of row fr.  Row stride 40 = 4 * 10: in each 16-lane
//       service group of ds_read_b128 ({0-3,12-15,20-27}, ...) the 16-byte slots fr * 10 + g (mod 16) are all distinct.
//       (A [row][36] image read with ds_read_b64 pairs -- which the compiler fuses into ds_read2_b64, served 16 lanes at a time over
//       32 banks -- measured 4 conflict cycles per LDS instruction: SQ_LDS_BANK_CONFLICT / SQ_INSTS_LDS on a 4096^3 problem.)
//   row-contiguous operand ("MC", e.g. dY^T and X of dW = dY^T X, W of dX = dY W): [k][64] with column ^ (((k>>2)&1) << 4);
//       staged with ds_write_b64 along the row, fetched with ds_read_b32 at k = 16h + 4g + s (g and g+1 land in opposite bank halves)
// Any bijection between the 4 lane groups x 8 MFMA steps and the 32 k of a tile is a valid contraction order as long as A and B
// agree; "group g, step 4h+s  <->  k = 16h + 4g + s" is the one that lets the KC side fetch four steps per LDS instruction.
//
// Staging never branches: rows beyond M / N are clamped to the last valid row (their products only reach accumulator rows that the
// epilogue never stores), k beyond K loads from a clamped in-bounds address and is replaced by 0.
//
// Tile order: the dispatcher places workgroup b on XCD b % 8 and every XCD has a private 4 MiB L2; logical tile = (b % 8) * chunk
// + b / 8 gives each XCD a contiguous run of tiles (n fastest), so the 4..12 column tiles that share one 64-row slab of A hit the
// same L2 instead of fetching the slab into up to 8 of them.
constexpr int F_LDK = 40;
__device__ __forceinline__ int f_swz(int k) { return ((k >> 2) & 1) << 4; }

typedef float f32x2 __attribute__((ext_vector_type(2)));

// `bid` = index of this workgroup within its problem (== blockIdx.x for a single GEMM; the grouped launch below subtracts the first
// workgroup of the group, always a multiple of 8, so bid % 8 is still the XCD the dispatcher placed the workgroup on)
template <bool A_KC, bool B_KC, bool C_PLAIN, bool K2>
__device__ __forceinline__ void gemm_f32_fast_body(const eegclip_gemm_desc& d, int gx, int ntiles, int chunk, int bid) {
    constexpr int A_FLOATS = A_KC ? G_BT * F_LDK : G_BK * G_BT;
    EEG_LDS_BASE(float, lds);
    float* As = lds;
    float* Bs = lds + A_FLOATS;

    int logical, slice = 0;
    if (d.split_k == 1) {
        logical = (bid & 7) * chunk + (bid >> 3);
        if (logical >= ntiles) return;                   // whole workgroup leaves before any barrier
    } else {
        // split-K (weight gradients: few output tiles, K = all rows of the batch): ALL tiles of one K slice go to ONE XCD, so the
        // slice of A and B is fetched from HBM once and re-read from that XCD's L2 by the other tiles (a tile-major order spread the
        // 16 tiles of a 250 x 256 gradient over all 8 XCDs: 114 MB of HBM traffic for 33 MB of operands, rocprofv3 FETCH_SIZE)
        const int slot = bid >> 3;
        slice = (bid & 7) + 8 * (slot / ntiles);
        logical = slot % ntiles;
        if (slice >= d.split_k) return;
    }
    const int m0 = (logical / gx) * G_BT, n0 = (logical % gx) * G_BT;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = wave_uniform(t >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, g = lane >> 4;
    int kt_begin, kt_end;
    gemm_k_slice<G_BK>(d, slice, kt_begin, kt_end);

    // staging roles: KC operand -> thread (row = r0 + 16 i, k pair kp);  MC operand -> thread (k = kr0 + 8 i, row pair mp)
    const int kp = t & 15, r0 = t >> 4, mp = t & 31, kr0 = t >> 5;
    const int a_ld = A_KC ? (int)d.Am.si : (int)d.Ak.si, b_ld = B_KC ? (int)d.Bn.si : (int)d.Bk.si;
    int a_fix[4], b_fix[4];          // KC: element offset of the (clamped) row;  MC: running element offset of the k row
    int a_col = 0, b_col = 0;        // MC: clamped first row of the pair
    // K2 (both operands row-contiguous, k through a two-level map {div, so, si}: the value-embedding weight gradient contracts over
    // the 63 channel rows of every 64-row sample): the k row advances by 32 per tile, so (offset, remainder) are carried along and
    // wrapped by subtraction -- no division in the loop
    int a_rem[K2 ? 4 : 1], b_rem[K2 ? 4 : 1];
    const int a_div = K2 ? (d.Ak.div > 0x7fffffffLL ? 0x7fffffff : (int)d.Ak.div) : 0;      // plain map: one block that never wraps
    const int b_div = K2 ? (d.Bk.div > 0x7fffffffLL ? 0x7fffffff : (int)d.Bk.div) : 0;
    const int a_wrap = K2 ? (int)(d.Ak.so - d.Ak.div * d.Ak.si) : 0, b_wrap = K2 ? (int)(d.Bk.so - d.Bk.div * d.Bk.si) : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kfirst = kt_begin * G_BK + kr0 + 8 * i;
        if (A_KC) { const int m = m0 + r0 + 16 * i; a_fix[i] = (m < d.M ? m : d.M - 1) * a_ld; }
        else if (K2) { a_rem[i] = kfirst % a_div; a_fix[i] = (kfirst / a_div) * (int)d.Ak.so + a_rem[i] * a_ld; }
        else      a_fix[i] = kfirst * a_ld;
        if (B_KC) { const int n = n0 + r0 + 16 * i; b_fix[i] = (n < d.N ? n : d.N - 1) * b_ld; }
        else if (K2) { b_rem[i] = kfirst % b_div; b_fix[i] = (kfirst / b_div) * (int)d.Bk.so + b_rem[i] * b_ld; }
        else      b_fix[i] = kfirst * b_ld;
    }
    if (!A_KC) { const int m = m0 + 2 * mp; a_col = m < d.M ? m : d.M - 2; }
    if (!B_KC) { const int n = n0 + 2 * mp; b_col = n < d.N ? n : d.N - 2; }

    // the "k beyond K -> 0" select is applied when the registers are written to LDS (store_tile), not where the loads are issued:
    // a select next to the load makes the wave wait for its prefetch before the MFMAs it is supposed to overlap with
    f32x2 ra[4], rb[4];
    unsigned okbits = 0;             // bit i: A pass i in range, bit 4+i: B pass i
    auto load_tile = [&](int kt) {
        const int k0 = kt * G_BK;
        const bool kc_ok = k0 + 2 * kp < d.K;
        const int kc_k = kc_ok ? k0 + 2 * kp : 0;
        okbits = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (A_KC) {
                ra[i] = *reinterpret_cast<const f32x2*>(d.A + a_fix[i] + kc_k);
                okbits |= (kc_ok ? 1u : 0u) << i;
            } else {
                const bool ok = k0 + kr0 + 8 * i < d.K;
                ra[i] = *reinterpret_cast<const f32x2*>(d.A + (ok ? a_fix[i] : 0) + a_col);
                okbits |= (ok ? 1u : 0u) << i;
                a_fix[i] += G_BK * a_ld;
                if (K2) {
                    a_rem[i] += G_BK;
                    while (a_rem[i] >= a_div) { a_rem[i] -= a_div; a_fix[i] += a_wrap; }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (B_KC) {
                rb[i] = *reinterpret_cast<const f32x2*>(d.B + b_fix[i] + kc_k);
                okbits |= (kc_ok ? 1u : 0u) << (4 + i);
            } else {
                const bool ok = k0 + kr0 + 8 * i < d.K;
                rb[i] = *reinterpret_cast<const f32x2*>(d.B + (ok ? b_fix[i] : 0) + b_col);
                okbits |= (ok ? 1u : 0u) << (4 + i);
                b_fix[i] += G_BK * b_ld;
                if (K2) {
                    b_rem[i] += G_BK;
                    while (b_rem[i] >= b_div) { b_rem[i] -= b_div; b_fix[i] += b_wrap; }
                }
            }
        }
    };
    auto store_tile = [&]() {
        const f32x2 zero{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 v = (okbits >> i) & 1u ? ra[i] : zero;
            if (A_KC) *reinterpret_cast<f32x2*>(As + (r0 + 16 * i) * F_LDK + 2 * kp) = v;
            else      *reinterpret_cast<f32x2*>(As + (kr0 + 8 * i) * G_BT + ((2 * mp) ^ f_swz(kr0))) = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 v = (okbits >> (4 + i)) & 1u ? rb[i] : zero;
            if (B_KC) *reinterpret_cast<f32x2*>(Bs + (r0 + 16 * i) * F_LDK + 2 * kp) = v;
            else      *reinterpret_cast<f32x2*>(Bs + (kr0 + 8 * i) * G_BT + ((2 * mp) ^ f_swz(kr0))) = v;
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool do_rowsum = d.rowsum_a != nullptr && n0 == 0;
    float rowsum = 0.f;
    if (kt_begin < kt_end) load_tile(kt_begin);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        store_tile();
        __syncthreads();
        if (kt + 1 < kt_end) load_tile(kt + 1);
        if (do_rowsum && t < G_BT) {                     // first wave of the n = 0 tiles: row sums of the staged A tile (bias gradients)
            if (A_KC) {
#pragma unroll
                for (int k = 0; k < G_BK; k += 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(As + t * F_LDK + k);
                    rowsum += (v[0] + v[1]) + (v[2] + v[3]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < G_BK; ++k) rowsum += As[k * G_BT + (t ^ f_swz(k))];
            }
        }
        const int sw = (g & 1) << 4;                     // f_swz(16h + 4g + s)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float av[2][4], bv[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (A_KC) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(As + (wr * 32 + 16 * i + fr) * F_LDK + 16 * h + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[i][e] = v[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[i][e] = As[(16 * h + 4 * g + e) * G_BT + ((wr * 32 + 16 * i + fr) ^ sw)];
                }
                if (B_KC) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(Bs + (wc * 32 + 16 * i + fr) * F_LDK + 16 * h + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[i][e] = v[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[i][e] = Bs[(16 * h + 4 * g + e) * G_BT + ((wc * 32 + 16 * i + fr) ^ sw)];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma_f32_16x16x4(av[i][e], bv[j][e], acc[i][j]);
        }
        __syncthreads();
    }
    if (do_rowsum && t < G_BT && m0 + t < d.M) atomicAdd(d.rowsum_a + m0 + t, rowsum);
    gemm_epilogue<C_PLAIN>(d, acc, m0, n0, wr, wc, lane, slice == 0);
}

template <bool A_KC, bool B_KC, bool C_PLAIN, bool K2 = false>
__global__ __launch_bounds__(G_THREADS) void gemm_f32_fast_kernel(const eegclip_gemm_desc d, int gx, int ntiles, int chunk) {
    gemm_f32_fast_body<A_KC, B_KC, C_PLAIN, K2>(d, gx, ntiles, chunk, (int)blockIdx.x);
}

// Grouped launch: up to GEMM_MAX_GROUPS problems that differ only in M, K, split_k and their pointers (the joint-subject model's
// per-subject value embeddings: SURVEY 8f row 1, models/subject_layers/Embed.py:142-144) run as ONE grid.  Group g owns workgroups
// [first[g], first[g+1]); everything else comes from the shared descriptor.  Ten 100-tile launches of 10 us each become one 1000-tile
// launch that fills the chip.
constexpr int GEMM_MAX_GROUPS = 16;
struct gemm_group {
    int M, K, split_k, pad;
    const float* A;
    const float* B;
    float* C;
    const float* bias_n;
    float* rowsum_a;
};
struct gemm_group_table {
    int n;
    int first[GEMM_MAX_GROUPS + 1];
    gemm_group g[GEMM_MAX_GROUPS];
};

template <bool A_KC, bool B_KC, bool C_PLAIN, bool K2>
__global__ __launch_bounds__(G_THREADS) void gemm_f32_grouped_kernel(const eegclip_gemm_desc d0, const gemm_group_table tb) {
    const int b = (int)blockIdx.x;
    int gi = 0;
    for (int i = 1; i < tb.n; ++i) gi += b >= tb.first[i] ? 1 : 0;          // workgroup-uniform
    eegclip_gemm_desc d = d0;
    d.M = tb.g[gi].M;
    d.K = tb.g[gi].K;
    d.split_k = tb.g[gi].split_k;
    d.A = tb.g[gi].A;
    d.B = tb.g[gi].B;
    d.C = tb.g[gi].C;
    d.bias_n = tb.g[gi].bias_n;
    d.rowsum_a = tb.g[gi].rowsum_a;
    const int gx = (d.N + G_BT - 1) / G_BT, ntiles = gx * ((d.M + G_BT - 1) / G_BT);
    gemm_f32_fast_body<A_KC, B_KC, C_PLAIN, K2>(d, gx, ntiles, (ntiles + 7) / 8, b - tb.first[gi]);
}

// =================================================================================================================================
// skinny kernel: M <= 32 rows against a k-contiguous weight matrix (the diffusion prior's sampling chain, Generation/diffusion_prior.py:
// 340-378: 16 rows -- 8 embeddings x the classifier-free-guidance pair -- through ~34 Linear layers per DDPM step, 50 steps).  With
// 64 x 64 tiles such a GEMM is at most 16 workgroups walking K = 1024 in 32 dependent k-tiles: 22 us each, 87 % of the chain
// (rocprofv3), on a chip that could stream the 4 MB of weights in a microsecond.  Here one workgroup owns 16 output columns and its
// 16 waves split K into interleaved 32-element chunks; each lane feeds the MFMA straight from two 16-byte global loads (lane (n, g)
// holds W[n][k + 4g .. 4g+3], the A lane X[m][same k]: MFMA step e contracts k + 4g + e over the four lane groups), no LDS staging.
// The 16 partial tiles are summed through LDS and one wave per 16-row block runs the general epilogue.
constexpr int SK_WAVES = 16;
constexpr int SK_N = 16;

template <int MB>
__global__ __launch_bounds__(SK_WAVES * 64) void gemm_f32_skinny_kernel(const eegclip_gemm_desc d) {
    EEG_LDS_BASE(float, lds);                                   // [wave][MB][lane] float4 partial accumulators
    const int t = threadIdx.x, lane = t & 63, wave = wave_uniform(t >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * SK_N;
    const int nrow = n0 + fr < d.N ? n0 + fr : d.N - 1;         // clamped rows: their products land in columns / rows nobody stores
    const float* bp = d.B + (long long)nrow * d.Bn.si + 4 * g;
    const float* ap[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int m = 16 * i + fr < d.M ? 16 * i + fr : d.M - 1;
        ap[i] = d.A + (long long)m * d.Am.si + 4 * g;
    }
    f32x4 acc[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 zero{0.f, 0.f, 0.f, 0.f};
    for (int k0 = wave * 32; k0 < d.K; k0 += SK_WAVES * 32) {   // (requesting two chunks before the first MFMA measured slower: 6.3 vs 5.3 us)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = k0 + 16 * h;
            const bool ok = k + 4 * g < d.K;                    // K % 4 == 0: a lane's four k are in range together
            const int kk = ok ? k : 0;
            f32x4 bv = *reinterpret_cast<const f32x4*>(bp + kk);
            bv = ok ? bv : zero;
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                f32x4 av = *reinterpret_cast<const f32x4*>(ap[i] + kk);
                av = ok ? av : zero;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i] = mfma_f32_16x16x4(av[e], bv[e], acc[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MB; ++i) *reinterpret_cast<f32x4*>(lds + ((wave * MB + i) * 64 + lane) * 4) = acc[i];
    __syncthreads();
    if (wave >= MB) return;
    f32x4 sum = zero;                                           // wave i finishes 16-row block i
#pragma unroll
    for (int w = 0; w < SK_WAVES; ++w) sum += *reinterpret_cast<const f32x4*>(lds + ((w * MB + wave) * 64 + lane) * 4);
    const float keep_scale = d.drop_p > 0.f ? 1.0f / (1.0f - d.drop_p) : 1.0f;
    const int n = n0 + fr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = 16 * wave + 4 * g + r;
        if (m < d.M && n < d.N)
            gemm_epilogue_element<true>(d, d.alpha * sum[r], m, n, (long long)m * d.Cm.si + (long long)n * d.Cn.si,
                                        d.R ? (long long)m * d.Rm.si + (long long)n * d.Rn.si : 0, true, keep_scale);
    }
}

static inline bool is_plain(const eegclip_dim& x) { return x.div > (1LL << 40); }
static inline bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }

// operand eligibility for the fast kernel: `rows` = M or N, row / k strides in elements
static bool fast_operand_ok(const float* p, long long row_si, long long k_si, int rows, int K, bool& kc) {
    if (!aligned8(p) || row_si < 0 || k_si < 0) return false;
    const long long reach = (long long)(rows - 1) * row_si + (long long)(K - 1) * k_si + 2;
    if (reach >= (1LL << 31)) return false;
    if (k_si == 1 && (row_si & 1) == 0 && (K & 1) == 0) { kc = true; return true; }
    if (row_si == 1 && (k_si & 1) == 0 && (rows & 1) == 0 && rows >= 2) { kc = false; return true; }
    return false;
}

// row-contiguous operand whose k index goes through a (possibly two-level) map: even strides, even row count, 32-bit reach
static bool fast_k2_ok(const float* p, const eegclip_dim& kd, int rows, int K) {
    if (!aligned8(p) || kd.si < 0 || kd.so < 0 || (kd.si & 1) || (kd.so & 1) || (rows & 1) || rows < 2) return false;
    const long long div = kd.div > (1LL << 30) ? (1LL << 30) : kd.div;                 // plain map: never wraps
    const long long reach = ((K - 1) / div) * kd.so + ((K - 1) % div) * kd.si + rows + 2;
    return kd.div >= 1 && reach < (1LL << 31) && (kd.div > (1LL << 30) || kd.so - kd.div * kd.si < (1LL << 31));
}

// which fast instantiation (if any) a problem can use: 0 = none, 1 = plain k maps (akc / bkc say which operands are k-contiguous),
// 2 = K2 (row-contiguous operands, two-level k maps)
static int fast_class(const eegclip_gemm_desc& d, bool& akc, bool& bkc, bool& c_plain) {
    const int gx = (d.N + G_BT - 1) / G_BT, gy = (d.M + G_BT - 1) / G_BT;
    const bool ab_plain = is_plain(d.Am) && is_plain(d.Ak) && is_plain(d.Bk) && is_plain(d.Bn);
    c_plain = is_plain(d.Cm) && is_plain(d.Cn) && (!d.R || (is_plain(d.Rm) && is_plain(d.Rn)));
    akc = bkc = false;
    if (d.K < 2 || (long long)gx * gy >= (1LL << 28)) return 0;
    if (ab_plain && fast_operand_ok(d.A, d.Am.si, d.Ak.si, d.M, d.K, akc) && fast_operand_ok(d.B, d.Bn.si, d.Bk.si, d.N, d.K, bkc)) return 1;
    akc = bkc = false;
    if (c_plain && is_plain(d.Am) && is_plain(d.Bn) && d.Am.si == 1 && d.Bn.si == 1 && fast_k2_ok(d.A, d.Ak, d.M, d.K) && fast_k2_ok(d.B, d.Bk, d.N, d.K))
        return 2;
    return 0;
}

static int launch_gemm(const eegclip_gemm_desc& d, void* stream, long long* ws_query = nullptr);

// everything a group cannot override must be the same in all members (Cpre / bias_m are per-problem buffers: not groupable when set)
static bool same_dim(const eegclip_dim& a, const eegclip_dim& b) { return a.div == b.div && a.so == b.so && a.si == b.si; }
static bool same_shared(const eegclip_gemm_desc& a, const eegclip_gemm_desc& b) {
    return a.N == b.N && same_dim(a.Am, b.Am) && same_dim(a.Ak, b.Ak) && same_dim(a.Bk, b.Bk) && same_dim(a.Bn, b.Bn) && same_dim(a.Cm, b.Cm) &&
           same_dim(a.Cn, b.Cn) && !a.Cpre && !b.Cpre && !a.bias_m && !b.bias_m && a.R == b.R && (!a.R || (same_dim(a.Rm, b.Rm) && same_dim(a.Rn, b.Rn))) &&
           a.alpha == b.alpha && a.accumulate == b.accumulate && a.act == b.act && a.drop_p == b.drop_p && a.seed == b.seed &&
           a.drop_site == b.drop_site && (a.bias_n == nullptr) == (b.bias_n == nullptr) && (a.rowsum_a == nullptr) == (b.rowsum_a == nullptr);
}

static int launch_gemm_grouped(const eegclip_gemm_desc* ds, int n, void* stream) {
    const bool allow = true;
    bool akc = false, bkc = false, cpl = false;
    int cls = (allow && n >= 2 && n <= GEMM_MAX_GROUPS) ? fast_class(ds[0], akc, bkc, cpl) : 0;
    gemm_group_table tb;
    tb.n = 0;
    tb.first[0] = 0;
    for (int i = 0; cls && i < n; ++i) {
        bool a2, b2, c2;
        if (fast_class(ds[i], a2, b2, c2) != cls || a2 != akc || b2 != bkc || c2 != cpl || !same_shared(ds[i], ds[0])) cls = 0;   // same instantiation
        const long long ntiles = (long long)((ds[i].N + G_BT - 1) / G_BT) * ((ds[i].M + G_BT - 1) / G_BT);
        const long long wgs = ds[i].split_k == 1 ? 8 * ((ntiles + 7) / 8) : 8LL * ((ds[i].split_k + 7) / 8) * ntiles;
        if (tb.first[i] + wgs >= (1LL << 30)) cls = 0;
        if (!cls) break;
        tb.first[i + 1] = tb.first[i] + (int)wgs;
        tb.g[i] = gemm_group{ds[i].M, ds[i].K, ds[i].split_k, 0, ds[i].A, ds[i].B, ds[i].C, ds[i].bias_n, ds[i].rowsum_a};
    }
    if (!cls) {                                           // not groupable: the same work as n launches
        for (int i = 0; i < n; ++i) {
            const int rc = launch_gemm(ds[i], stream);
            if (rc) return rc;
        }
        return 0;
    }
    tb.n = n;
    const dim3 grid(tb.first[n]), block(G_THREADS);
    const size_t lds = ((akc ? G_BT * F_LDK : G_BK * G_BT) + (bkc ? G_BT * F_LDK : G_BK * G_BT)) * sizeof(float);
#define EEG_GROUP_GO(AK, BK_, CP, K2_) EEG_LAUNCH((gemm_f32_grouped_kernel<AK, BK_, CP, K2_>), grid, block, lds, stream, ds[0], tb)
    if (cls == 2)                EEG_GROUP_GO(false, false, true, true);
    else if (akc && bkc && cpl)  EEG_GROUP_GO(true, true, true, false);
    else if (akc && bkc)         EEG_GROUP_GO(true, true, false, false);
    else if (akc && cpl)         EEG_GROUP_GO(true, false, true, false);
    else if (akc)                EEG_GROUP_GO(true, false, false, false);
    else if (bkc && cpl)         EEG_GROUP_GO(false, true, true, false);
    else if (bkc)                EEG_GROUP_GO(false, true, false, false);
    else if (cpl)                EEG_GROUP_GO(false, false, true, false);
    else                         EEG_GROUP_GO(false, false, false, false);
#undef EEG_GROUP_GO
    return (int)hipGetLastError();
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ws_query: do not launch; report the split-K workspace the launch would use (0 unless it is routed to the BF16X3 kernels)
static int launch_gemm(const eegclip_gemm_desc& d, void* stream, long long* ws_query) {
    if (ws_query) *ws_query = 0;
    const bool allow_skinny = true;
    // skinny: few rows against k-contiguous operands, every map a plain stride, 16-byte loads legal, K long enough for the 16-way split
    if (allow_skinny && d.M <= 32 && d.split_k == 1 && d.K >= 64 && (d.K & 3) == 0 && is_plain(d.Am) && is_plain(d.Ak) && is_plain(d.Bk) &&
        is_plain(d.Bn) && is_plain(d.Cm) && is_plain(d.Cn) && (!d.R || (is_plain(d.Rm) && is_plain(d.Rn))) && d.Ak.si == 1 && d.Bk.si == 1 &&
        d.Am.si >= 0 && d.Bn.si >= 0 && (d.Am.si & 3) == 0 && (d.Bn.si & 3) == 0 && aligned16(d.A) && aligned16(d.B)) {
        if (ws_query) return 0;
        const dim3 grid((d.N + SK_N - 1) / SK_N), block(SK_WAVES * 64);
        if (d.M <= 16) EEG_LAUNCH((gemm_f32_skinny_kernel<1>), grid, block, SK_WAVES * 1 * 64 * 4 * sizeof(float), stream, d);
        else           EEG_LAUNCH((gemm_f32_skinny_kernel<2>), grid, block, SK_WAVES * 2 * 64 * 4 * sizeof(float), stream, d);
        return (int)hipGetLastError();
    }
    const int gx = (d.N + G_BT - 1) / G_BT, gy = (d.M + G_BT - 1) / G_BT;
    const dim3 block(G_THREADS);
    const bool ab_plain = is_plain(d.Am) && is_plain(d.Ak) && is_plain(d.Bk) && is_plain(d.Bn);
    const bool c_plain = is_plain(d.Cm) && is_plain(d.Cn) && (!d.R || (is_plain(d.Rm) && is_plain(d.Rn)));
    const bool plain = ab_plain && c_plain;
    const bool allow_fast = true;
    bool akc = false, bkc = false;
    const int ntiles = gx * gy, chunk = (ntiles + 7) / 8;
    const dim3 fgrid(d.split_k == 1 ? 8 * chunk : 8 * ((d.split_k + 7) / 8) * ntiles);
    if (allow_fast && ab_plain && d.K >= 2 && (long long)gx * gy < (1LL << 28) && fast_operand_ok(d.A, d.Am.si, d.Ak.si, d.M, d.K, akc) &&
        fast_operand_ok(d.B, d.Bn.si, d.Bk.si, d.N, d.K, bkc)) {
        if (ws_query) {
            if ((d.precision & 0xff) == EEGCLIP_PREC_BF16X3) *ws_query = gemm_x3_workspace_bytes(d);
            return 0;
        }
        if ((d.precision & 0xff) == EEGCLIP_PREC_BF16X3) return launch_gemm_x3(d, akc, bkc, c_plain, false, stream);
        const size_t lds = ((akc ? G_BT * F_LDK : G_BK * G_BT) + (bkc ? G_BT * F_LDK : G_BK * G_BT)) * sizeof(float);
#define EEG_FAST_GO(AK, BK_)                                                                                                         \
    do {                                                                                                                             \
        if (c_plain) EEG_LAUNCH((gemm_f32_fast_kernel<AK, BK_, true>), fgrid, block, lds, stream, d, gx, ntiles, chunk);            \
        else         EEG_LAUNCH((gemm_f32_fast_kernel<AK, BK_, false>), fgrid, block, lds, stream, d, gx, ntiles, chunk);           \
    } while (0)
        if (akc && bkc)        EEG_FAST_GO(true, true);
        else if (akc && !bkc)  EEG_FAST_GO(true, false);
        else if (!akc && bkc)  EEG_FAST_GO(false, true);
        else                   EEG_FAST_GO(false, false);
#undef EEG_FAST_GO
        return (int)hipGetLastError();
    }
    // both operands row-contiguous with k running through two-level maps (plain row maps, plain C): the K2 instantiation
    if (allow_fast && c_plain && is_plain(d.Am) && is_plain(d.Bn) && d.Am.si == 1 && d.Bn.si == 1 && d.K >= 2 && (long long)gx * gy < (1LL << 28) &&
        fast_k2_ok(d.A, d.Ak, d.M, d.K) && fast_k2_ok(d.B, d.Bk, d.N, d.K)) {
        if (ws_query) {
            if ((d.precision & 0xff) == EEGCLIP_PREC_BF16X3) *ws_query = gemm_x3_workspace_bytes(d);
            return 0;
        }
        if ((d.precision & 0xff) == EEGCLIP_PREC_BF16X3) return launch_gemm_x3(d, false, false, true, true, stream);
        const size_t lds = 2 * G_BK * G_BT * sizeof(float);
        EEG_LAUNCH((gemm_f32_fast_kernel<false, false, true, true>), fgrid, block, lds, stream, d, gx, ntiles, chunk);
        return (int)hipGetLastError();
    }
    if (ws_query) return 0;
    const dim3 grid(gx, gy, d.split_k);
    const size_t lds = 2 * G_BK * G_BT * sizeof(float);
    akc = (d.Ak.si == 1);
    bkc = (d.Bk.si == 1);
#define EEG_GEMM_GO(AK, BK_)                                                                              \
    do {                                                                                                  \
        if (plain) EEG_LAUNCH((gemm_f32_kernel<AK, BK_, true>), grid, block, lds, stream, d);            \
        else       EEG_LAUNCH((gemm_f32_kernel<AK, BK_, false>), grid, block, lds, stream, d);           \
    } while (0)
    if (akc && bkc)        EEG_GEMM_GO(true, true);
    else if (akc && !bkc)  EEG_GEMM_GO(true, false);
    else if (!akc && bkc)  EEG_GEMM_GO(false, true);
    else                   EEG_GEMM_GO(false, false);
#undef EEG_GEMM_GO
    return (int)hipGetLastError();
}

}  // namespace eeg

static int gemm_desc_check(const eegclip_gemm_desc& d);

extern "C" int eegclip_gemm_f32(const eegclip_gemm_desc* dp, void* stream) {
    using namespace eeg;
    if (!dp) return EEGCLIP_EINVAL;
    const eegclip_gemm_desc d = *dp;
    const int rc = gemm_desc_check(d);
    if (rc || d.M == 0 || d.N == 0) return rc;
    return launch_gemm(d, stream);
}

extern "C" long long eegclip_gemm_workspace_bytes(const eegclip_gemm_desc* dp) {
    using namespace eeg;
    if (!dp) return 0;
    const eegclip_gemm_desc d = *dp;
    if (gemm_desc_check(d) || d.M == 0 || d.N == 0 || d.split_k <= 1) return 0;
    long long bytes = 0;
    launch_gemm(d, nullptr, &bytes);
    return bytes;
}

extern "C" int eegclip_gemm_f32_grouped(const eegclip_gemm_desc* descs, int n, void* stream) {
    using namespace eeg;
    if (!descs || n < 0) return EEGCLIP_EINVAL;
    bool empty = false;
    for (int i = 0; i < n; ++i) {
        const int rc = gemm_desc_check(descs[i]);
        if (rc) return rc;
        empty = empty || descs[i].M == 0 || descs[i].N == 0;
    }
    if (n == 0) return 0;
    if (empty || n == 1) {                               // degenerate members: plain launches of the non-empty ones
        for (int i = 0; i < n; ++i) {
            if (descs[i].M == 0 || descs[i].N == 0) continue;
            const int rc = launch_gemm(descs[i], stream);
            if (rc) return rc;
        }
        return 0;
    }
    return launch_gemm_grouped(descs, n, stream);
}

static int gemm_desc_check(const eegclip_gemm_desc& d) {
    if (d.M < 0 || d.N < 0 || d.K < 0 || !d.C) return EEGCLIP_EINVAL;
    if (d.M == 0 || d.N == 0) return 0;                  /* nothing to compute (callers skip the launch) */
    if (d.K > 0 && (!d.A || !d.B)) return EEGCLIP_EINVAL;
    if (d.split_k < 1) return EEGCLIP_EINVAL;
    if (d.split_k > 1 && (d.act != EEGCLIP_ACT_NONE || d.drop_p > 0.f || d.R || d.Cpre || d.accumulate == 2)) return EEGCLIP_EINVAL;
    if (d.accumulate < 0 || d.accumulate > 2) return EEGCLIP_EINVAL;
    if (d.drop_p < 0.f || d.drop_p >= 1.f || d.act < 0 || d.act > EEGCLIP_ACT_GELU_GRAD) return EEGCLIP_EINVAL;
    if ((d.precision & 0xff) != EEGCLIP_PREC_F32 && (d.precision & 0xff) != EEGCLIP_PREC_BF16X3) return EEGCLIP_EINVAL;
    if ((d.precision >> 8) < 0 || (d.precision >> 8) > 6) return EEGCLIP_EINVAL;
    if (d.workspace && d.workspace_bytes < 0) return EEGCLIP_EINVAL;
    if (d.act == EEGCLIP_ACT_GELU_GRAD && (!d.R || d.split_k > 1)) return EEGCLIP_EINVAL;
    if (d.Am.div <= 0 || d.Ak.div <= 0 || d.Bk.div <= 0 || d.Bn.div <= 0 || d.Cm.div <= 0 || d.Cn.div <= 0) return EEGCLIP_EINVAL;
    if (d.R && (d.Rm.div <= 0 || d.Rn.div <= 0)) return EEGCLIP_EINVAL;
    return 0;
}

extern "C" int eegclip_abi_version(void) { return EEGCLIP_ABI_VERSION; }
