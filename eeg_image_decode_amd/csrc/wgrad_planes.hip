// Weight gradients as "long K, small output" GEMMs over bf16 planes:  dW[o][i] (+)= sum_t dY[t][o] X[t][i]   (t = the 16384 token rows of a batch).
//
// In the plan GEMM (csrc/gemm_x3.hip) both operands of a weight gradient are k-STRIDED (k = t runs down the rows of dY and X): every workgroup
// transposes 4 k x 2 m register blocks and splits fp32 -> bf16 hi | lo while it stages -- ~1000 VALU instructions per wave and k-tile; the five
// weight gradients of the transformer block cost 35 .. 73 us each (60 TF) although they are 2 .. 6 GFLOP (PMC, profiles/r2_pmc_step.json: MFMA
// pipe 19 % busy).  Here the transposition and the split happen ONCE per operand in a bandwidth-bound pass (split_transpose_kernel: fp32 [t][c] ->
// bf16 planes [c][t], t contiguous), and the GEMM is a pure planes kernel: 16-byte global loads -> ds_write_b128 -> ds_read_b128 fragments ->
// v_mfma_f32_32x32x16_bf16, three products per multiply-add like every other split-bf16 contraction of the library.
//
//   wgrad_planes_kernel   workgroup = (128 x 64 output tile, K slice); 4 waves as 2 x 2, each 64 x 32 outputs = two 32x32 MFMA tiles; operand tiles
//                         [128 | 64 rows][64 k] per plane, two LDS stages, the next k-tile's loads in flight under the MFMAs; 128-byte LDS rows with
//                         the 16-byte chunk index XOR ((row >> 1) & 7) (conflict-free fragment reads, as csrc/infonce_fused.hip).  The bias
//                         gradient (column sums of dY = row sums of the A planes) rides on the matrix cores: one extra accumulator against an
//                         all-ones B fragment in the waves of the first output-tile column.  Partial tiles go to per-slice slabs.
//   wgrad_reduce_kernel   out[m][n] += sum_s slab[s][m][n] (ordered: bit-reproducible), bias[m] += sum_s ...
#include "eeg_common.h"

namespace eeg {

constexpr int WG_TM = 128, WG_TN = 64, WG_BK = 64;
constexpr int WG_ROWB = 2 * WG_BK;                               // bytes per LDS row of a plane tile
constexpr int WG_TILE_A = WG_TM * WG_ROWB, WG_TILE_B = WG_TN * WG_ROWB;
constexpr int WG_STAGE = 2 * (WG_TILE_A + WG_TILE_B);            // A hi | A lo | B hi | B lo
typedef unsigned wg_u4 __attribute__((ext_vector_type(4)));
typedef float wg_f4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ int wg_swz(int row) { return (row >> 1) & 7; }

struct wgrad_args {
    const unsigned short *a_hi, *a_lo, *b_hi, *b_lo;             // planes [rows][K], k contiguous; rows padded to the tile (zeros)
    float* slab;                                                 // [slices][Mp][Np] partial tiles, then [slices][Mp] bias partials
    int Mp, Np, K, slices, want_bias;
    long long ld;                                                // elements between plane rows (>= K; NOT a multiple of 2048: see eegclip_wgrad_planes)
};

__global__ __launch_bounds__(256) void wgrad_planes_kernel(const wgrad_args a) {
    EEG_LDS_BASE(unsigned char, lds);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;
    const int tiles_n = a.Np / WG_TN, tiles = (a.Mp / WG_TM) * tiles_n;
    // XCD-aware order: workgroup b runs on XCD b % 8 (observed dispatch; speed only).  All tiles of a K slice read the same 64-k columns of both
    // operands, so a slice's tiles go to ONE XCD (slices is a multiple of 8: XCD x owns slices x, x + 8, ...) and every operand byte enters exactly
    // one L2.  (tile-major order put one output tile per XCD: each L2 fetched its tile's rows over ALL of K -- 96 MB instead of 32 per GEMM.)
    int slice, tile;
    if ((a.slices & 7) == 0) {
        const int xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
        slice = xcd + 8 * (j / tiles);
        tile = j % tiles;
    } else {
        slice = (int)blockIdx.x / tiles;
        tile = (int)blockIdx.x - slice * tiles;
    }
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int ktiles_all = a.K / WG_BK;
    const int kt0 = (int)((long long)slice * ktiles_all / a.slices), kt1 = (int)((long long)(slice + 1) * ktiles_all / a.slices);
    const bool bias = a.want_bias && tn == 0 && wn == 0;

    // staging: chunk c = t + 256 i of the stage image [A hi (128 rows) | A lo | B hi (64 rows) | B lo], 8 chunks of 16 bytes per row
    constexpr int CH_A = WG_TM * 8, CH_B = WG_TN * 8, CH = 2 * (CH_A + CH_B), CPT = CH / 256;      // 3072 chunks, 12 per thread
    const unsigned short* gsrc[CPT];
    int loff[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        int c = t + 256 * i;
        const unsigned short* base;
        int row0, tile_off;
        if (c < CH_A) { base = a.a_hi; row0 = tm * WG_TM; tile_off = 0; }
        else if (c < 2 * CH_A) { c -= CH_A; base = a.a_lo; row0 = tm * WG_TM; tile_off = WG_TILE_A; }
        else if (c < 2 * CH_A + CH_B) { c -= 2 * CH_A; base = a.b_hi; row0 = tn * WG_TN; tile_off = 2 * WG_TILE_A; }
        else { c -= 2 * CH_A + CH_B; base = a.b_lo; row0 = tn * WG_TN; tile_off = 2 * WG_TILE_A + WG_TILE_B; }
        const int row = c >> 3, pos = c & 7;
        gsrc[i] = base + (long long)(row0 + row) * a.ld + 8 * pos;
        loff[i] = tile_off + row * WG_ROWB + ((pos ^ wg_swz(row)) << 4);
    }
    // three k-tiles of operand loads in flight per thread (a register ring: tile j sits in ring[j % 3] until it is written to LDS stage j & 1):
    // with ONE k-tile in flight the kernel was latency-bound -- 47 us for 250 x 256 x 16384, a memory round trip per k-tile and workgroup
    // The loads are inline asm with hand-counted s_waitcnt (cdna_hip_programming.md 5.7 form (ii)): left to hipcc, every wait in this loop came out as
    // vmcnt(11) .. vmcnt(0) -- the runtime `if (j + 3 < nk)` around a refill makes its counter bookkeeping give up at the merge, so each k-tile waited
    // for the tiles requested a moment earlier (a full memory round trip per k-tile: 16 us for 250 x 256 x 16384 at any prefetch depth).
    wg_u4 ring[3][CPT];
    auto gload = [&](int kt, wg_u4 (&r)[CPT]) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const unsigned short* p = gsrc[i] + (long long)kt * WG_BK;
#if defined(EEG_EMU)
            r[i] = *reinterpret_cast<const wg_u4*>(p);
#else
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[i]) : "v"(p) : "memory");
#endif
        }
    };
    // at most N of this wave's asm loads still in flight; names every register of the tile about to be consumed so that no use is scheduled above it
    auto wait_tile = [&](int younger_tiles, wg_u4 (&r)[CPT]) {
#if !defined(EEG_EMU)
        static_assert(CPT == 12, "the wait statement lists 12 destination registers");
#define WG_WAIT(N)                                                                                                                                    \
    asm volatile("s_waitcnt vmcnt(" #N ")"                                                                                                           \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]),   \
                   "+v"(r[11])                                                                                                                        \
                 :                                                                                                                                    \
                 : "memory")
        if (younger_tiles >= 2) WG_WAIT(24);
        else if (younger_tiles == 1) WG_WAIT(12);
        else WG_WAIT(0);
#undef WG_WAIT
#endif
    };
    auto lstore = [&](int stg, const wg_u4 (&r)[CPT]) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) *reinterpret_cast<wg_u4*>(lds + stg * WG_STAGE + loff[i]) = r[i];
    };
    // fragment offsets: A rows wm 64 + 32 i + r32, B rows wn 32 + r32; chunk 2 s + h of the row (k-step s of 16)
    int foa[4][2], fob[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ra = wm * 64 + 32 * i + r32;
            foa[s][i] = ra * WG_ROWB + (((2 * s + h) ^ wg_swz(ra)) << 4);
        }
        const int rb = wn * 32 + r32;
        fob[s] = 2 * WG_TILE_A + rb * WG_ROWB + (((2 * s + h) ^ wg_swz(rb)) << 4);
    }
    f32x16 acc[2], bacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[i][e] = 0.f; bacc[i][e] = 0.f; }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;         // bf16 1.0

    const int nk = kt1 - kt0;
    if (nk > 0) gload(kt0, ring[0]);
    if (nk > 1) gload(kt0 + 1, ring[1]);
    if (nk > 2) gload(kt0 + 2, ring[2]);
    if (nk > 0) {
        // ONE wait form here (everything issued so far): with the three-way choice of wait_tile() the compiler gave the vmcnt(0) branch other
        // registers for the "+v" operands and COPIED the in-flight ring registers into them ahead of the wait (tests/test_asm_rings.py)
        wait_tile(0, ring[0]);
        lstore(0, ring[0]);
    }
    raw_barrier();
    auto compute = [&](int j) {
        const unsigned char* st = lds + (j & 1) * WG_STAGE;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 ah[2], al[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8*>(st + foa[s][i]);
                al[i] = *reinterpret_cast<const bf16x8*>(st + WG_TILE_A + foa[s][i]);
            }
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(st + fob[s]), bl = *reinterpret_cast<const bf16x8*>(st + WG_TILE_B + fob[s]);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = mfma_bf16_32x32x16(al[i], bh, acc[i]);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = mfma_bf16_32x32x16(ah[i], bl, acc[i]);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = mfma_bf16_32x32x16(ah[i], bh, acc[i]);        // D[m = wm 64 + 32 i + row(reg, h)][n = wn 32 + r32]
            if (bias) {                                           // (wave-uniform) row sums of the A tile: every column of the tile holds them
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    bacc[i] = mfma_bf16_32x32x16(al[i], ones, bacc[i]);
                    bacc[i] = mfma_bf16_32x32x16(ah[i], ones, bacc[i]);
                }
            }
        }
    };
    // step U of a group of three: tile j = j0 + U is in LDS stage j & 1, tile j + 1 in ring[(U + 1) % 3], tile j + 2 in flight in ring[(U + 2) % 3];
    // ring[U] is free (stored before tile j was computed): refill it with tile j + 3 FIRST, so that load runs under the MFMAs
#define WG_STEP(U)                                                              \
    if (j0 + U < nk) {                                                          \
        const int j = j0 + U;                                                   \
        if (j + 3 < nk) gload(kt0 + j + 3, ring[U]);                            \
        compute(j);                                                             \
        if (j + 1 < nk) {                                                       \
            wait_tile((j + 3 < nk) + (j + 2 < nk), ring[(U + 1) % 3]);          \
            lstore((j + 1) & 1, ring[(U + 1) % 3]);                             \
        }                                                                       \
        raw_barrier();   /* NOT __syncthreads(): that drains vmcnt(0), i.e. waits for the two k-tiles just requested */ \
    }
    for (int j0 = 0; j0 < nk; j0 += 3) {
        WG_STEP(0)
        WG_STEP(1)
        WG_STEP(2)
    }
#undef WG_STEP
    // partial tile -> this slice's slab (plain stores: the reduce kernel sums the slices in order)
    float* out = a.slab + (long long)slice * a.Mp * a.Np;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = tm * WG_TM + wm * 64 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
            out[(long long)m * a.Np + tn * WG_TN + wn * 32 + r32] = acc[i][e];
        }
    if (bias && r32 == 0) {
        float* bo = a.slab + (long long)a.slices * a.Mp * a.Np + (long long)slice * a.Mp;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) bo[tm * WG_TM + wm * 64 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h] = bacc[i][e];
    }
}

// one thread per output element, 16 slices of loads in flight (consecutive threads = consecutive columns of a slab row: coalesced).  (A float4 per
// thread left 63 workgroups for a 250 x 256 output: 9 us of latency on a quarter of the chip; so did walking the slices 8 at a time.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int slices, int Mp, int Np, int M, int N, float* __restrict__ out,
                                                            long long ldo, float* __restrict__ bias_out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)M * N;
    const long long stride = (long long)Mp * Np;
    if (i < total) {
        const int m = (int)(i / N), n = (int)(i - (long long)m * N);
        const float* p = slab + (long long)m * Np + n;
        float s = 0.f;
        int k = 0;
        for (; k + 16 <= slices; k += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = p[(long long)(k + u) * stride];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
        for (; k < slices; ++k) s += p[(long long)k * stride];
        out[(long long)m * ldo + n] += s;
    } else if (bias_out && i < total + M) {
        const int m = (int)(i - total);
        const float* bs = slab + (long long)slices * stride;
        float s = 0.f;
#pragma unroll 16
        for (int k = 0; k < slices; ++k) s += bs[(long long)k * Mp + m];
        bias_out[m] += s;
    }
}

// fp32 [rows = k][cols = c] (row stride ld) -> bf16 planes [c][k] (row stride ldo elements, k contiguous); rows of the output beyond `cols` (up to
// gridDim.y * 64) are written as zeros.  One workgroup per 64 k x 64 c tile, transposed through LDS ([64][65] floats).
__global__ __launch_bounds__(256) void split_transpose_kernel(const float* __restrict__ src, long long ld, int rows, int cols, unsigned short* __restrict__ hi,
                                                               unsigned short* __restrict__ lo, long long ldo) {
    EEG_LDS_BASE(float, tile);
    const int t = threadIdx.x;
    const int k0 = (int)blockIdx.x * 64, c0 = (int)blockIdx.y * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = (t >> 4) + 16 * i, c = 4 * (t & 15);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (k0 + k < rows) {
            const float* p = src + (long long)(k0 + k) * ld + c0 + c;
            if (c0 + c + 3 < cols) {
                const wg_f4u q = *reinterpret_cast<const wg_f4u*>(p);
                v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = c0 + c + e < cols ? p[e] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[k * 65 + c + e] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int item = t + 256 * i, c = item & 63, kg = item >> 6;          // 8 consecutive k of one output row
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tile[(8 * kg + e) * 65 + c];
        u32x2_t h0, l0, h1, l1;
        x3_split4(v[0], v[1], v[2], v[3], h0, l0);
        x3_split4(v[4], v[5], v[6], v[7], h1, l1);
        const long long o = (long long)(c0 + c) * ldo + k0 + 8 * kg;
        *reinterpret_cast<wg_u4*>(hi + o) = wg_u4{h0[0], h0[1], h1[0], h1[1]};
        *reinterpret_cast<wg_u4*>(lo + o) = wg_u4{l0[0], l0[1], l1[0], l1[1]};
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------------------
// The same weight gradient from planes in their NATURAL layout [t][c] (what a producer holds anyway: no transposing pass): operand tiles
// [32 t][128 | 64 c] go through LDS unchanged (16-byte copies) and both MFMA operands are fetched with ds_read_b64_tr_b16, the LDS transpose read
// of gfx950 (see csrc/attention_x3.hip for the lane mapping): lane (fr, g) of a 16x16x32 step receives tokens 4g .. 4g+3 and 16 + 4g .. 16 + 4g+3
// of column c0 + fr -- the same k-slot assignment on both operands.  Row strides 288 B (A) and 160 B (B): the 4 rows x 32 bytes a 16-lane group
// addresses land 8 banks apart and the groups 32 apart (every bank exactly twice per 512-byte read: the minimum).
//   wgrad_tr_kernel        workgroup = (128 x 64 output tile, K slice); 4 waves as 2 x 2, each 64 x 32 outputs = 4 x 2 MFMA tiles of 16 x 16;
//                          k-tile = 32 tokens = one MFMA k-step; the next k-tile's 6 x 16-byte loads per thread are in flight under the MFMAs.
//   split_rows_natural_kernel   fp32 [t][c] -> bf16 planes [t][ldp] (ldp a multiple of 8, columns >= c zero): the operand of the kernel above.
// Opt-in (EEGCLIP_WGRAD_TR=1 in the encoder's backward plan): parity-tested on the emulator and the GPU; first timing, untuned and with a plain split
// launch per operand in front: 1.09 vs 1.01 ms per step -- it needs the producers to write the planes (they hold them in LDS) and a tuning pass.
constexpr int WT_BK = 32;
constexpr int WT_RA = 288, WT_RB = 160;                                  // bytes per LDS row of an A / B plane tile
constexpr int WT_TILE_A = WT_BK * WT_RA, WT_TILE_B = WT_BK * WT_RB;
constexpr int WT_STAGE = 2 * (WT_TILE_A + WT_TILE_B);                   // A hi | A lo | B hi | B lo
typedef short wt_s4 __attribute__((ext_vector_type(4)));

struct wgrad_tr_args {
    const unsigned short *a_hi, *a_lo, *b_hi, *b_lo;             // planes [K][lda] / [K][ldb], channel contiguous
    float* slab;
    int M, N, Mp, Np, K, slices, want_bias;
    long long lda, ldb;
};

__device__ __forceinline__ wt_s4 wt_tr_read(const unsigned char* p) {
#if defined(EEG_EMU)
    const int lane = hipemu::cur->lane, g = lane >> 4, i = lane & 15;
    wt_s4 r;
    for (int j = 0; j < 4; ++j) {
        const unsigned long long src = hipemu::shfl_idx((unsigned long long)(uintptr_t)p, 16 * g + 4 * j + (i >> 2));
        r[j] = reinterpret_cast<const short*>((uintptr_t)src)[i & 3];
    }
    return r;
#else
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wt_s4*)(p));
#endif
}
// k slots 0-3 <-> tile rows 4g .. 4g+3, slots 4-7 <-> rows 16 + 4g .. of column c0 + (lane & 15)
__device__ __forceinline__ bf16x8 wt_frag(const unsigned char* plane, int rs, int c0, int lane) {
    const int l = lane & 15, g = lane >> 4;
    const unsigned char* p = plane + (4 * g + (l >> 2)) * rs + 2 * (c0 + 4 * (l & 3));
    const wt_s4 a = wt_tr_read(p), b = wt_tr_read(p + 16 * rs);
    return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

__global__ __launch_bounds__(256) void wgrad_tr_kernel(const wgrad_tr_args a) {
    EEG_LDS_BASE(unsigned char, lds);
    const int t = threadIdx.x, lane = t & 63, wave = wave_uniform(t >> 6), wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, g = lane >> 4;
    const int tiles_n = a.Np / WG_TN, tiles = (a.Mp / WG_TM) * tiles_n;
    int slice, tile;
    if ((a.slices & 7) == 0) {                                   // all tiles of a K slice on one XCD (see wgrad_planes_kernel)
        const int xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
        slice = xcd + 8 * (j / tiles);
        tile = j % tiles;
    } else {
        slice = (int)blockIdx.x / tiles;
        tile = (int)blockIdx.x - slice * tiles;
    }
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int ktiles_all = a.K / WT_BK;
    const int kt0 = (int)((long long)slice * ktiles_all / a.slices), kt1 = (int)((long long)(slice + 1) * ktiles_all / a.slices);
    const bool bias = a.want_bias && tn == 0 && wn == 0;

    // staging: chunk c = t + 256 i of [A hi: 32 rows x 16 chunks | A lo | B hi: 32 rows x 8 chunks | B lo]; a chunk past the plane's columns is zero
    constexpr int CH_A = WT_BK * 16, CH_B = WT_BK * 8, CPT = (2 * CH_A + 2 * CH_B) / 256;      // 6
    const unsigned short* gsrc[CPT];
    int loff[CPT];
    bool gok[CPT];
    long long gstep[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        int c = t + 256 * i;
        const unsigned short* base;
        int row, col, ncols, off;
        long long ld;
        if (c < 2 * CH_A) {
            const bool lo = c >= CH_A;
            c -= lo ? CH_A : 0;
            row = c >> 4; col = tm * WG_TM + 8 * (c & 15);
            base = lo ? a.a_lo : a.a_hi; ld = a.lda; ncols = (int)a.lda;
            off = (lo ? WT_TILE_A : 0) + row * WT_RA + 16 * (c & 15);
        } else {
            c -= 2 * CH_A;
            const bool lo = c >= CH_B;
            c -= lo ? CH_B : 0;
            row = c >> 3; col = tn * WG_TN + 8 * (c & 7);
            base = lo ? a.b_lo : a.b_hi; ld = a.ldb; ncols = (int)a.ldb;
            off = 2 * WT_TILE_A + (lo ? WT_TILE_B : 0) + row * WT_RB + 16 * (c & 7);
        }
        gok[i] = col + 8 <= ncols;                                // (plane rows are padded to multiples of 8 columns)
        gsrc[i] = base + (long long)row * ld + (gok[i] ? col : 0);
        gstep[i] = (long long)WT_BK * ld;
        loff[i] = off;
    }
    wg_u4 regs[CPT];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            regs[i] = gok[i] ? *reinterpret_cast<const wg_u4*>(gsrc[i] + (long long)kt * gstep[i]) : wg_u4{0u, 0u, 0u, 0u};
    };
    auto lstore = [&](int stg) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) *reinterpret_cast<wg_u4*>(lds + stg * WT_STAGE + loff[i]) = regs[i];
    };
    f32x4 acc[4][2], bacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;         // bf16 1.0
    const int nk = kt1 - kt0;
    if (nk > 0) {
        gload(kt0);
        lstore(0);
    }
    __syncthreads();
    for (int j = 0; j < nk; ++j) {
        if (j + 1 < nk) gload(kt0 + j + 1);                      // in flight under this k-tile's MFMAs
        const unsigned char* st = lds + (j & 1) * WT_STAGE;
        bf16x8 bh[2], bl[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            bh[nt] = wt_frag(st + 2 * WT_TILE_A, WT_RB, 32 * wn + 16 * nt, lane);
            bl[nt] = wt_frag(st + 2 * WT_TILE_A + WT_TILE_B, WT_RB, 32 * wn + 16 * nt, lane);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const bf16x8 ah = wt_frag(st, WT_RA, 64 * wm + 16 * mt, lane), al = wt_frag(st + WT_TILE_A, WT_RA, 64 * wm + 16 * mt, lane);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {                     // D[m = 64 wm + 16 mt + 4 g + r][n = 32 wn + 16 nt + fr]
                acc[mt][nt] = mfma_bf16_16x16x32(al, bh[nt], acc[mt][nt]);
                acc[mt][nt] = mfma_bf16_16x16x32(ah, bl[nt], acc[mt][nt]);
                acc[mt][nt] = mfma_bf16_16x16x32(ah, bh[nt], acc[mt][nt]);
            }
            if (bias) {                                           // (wave-uniform) column sums of the A tile: every column of the product holds them
                bacc[mt] = mfma_bf16_16x16x32(al, ones, bacc[mt]);
                bacc[mt] = mfma_bf16_16x16x32(ah, ones, bacc[mt]);
            }
        }
        if (j + 1 < nk) lstore((j + 1) & 1);                     // the other stage: nobody reads it during this k-tile
        __syncthreads();
    }
    float* out = a.slab + (long long)slice * a.Mp * a.Np;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = tm * WG_TM + 64 * wm + 16 * mt + 4 * g + r;
                out[(long long)m * a.Np + tn * WG_TN + 32 * wn + 16 * nt + fr] = acc[mt][nt][r];
            }
    if (bias && fr == 0) {
        float* bo = a.slab + (long long)a.slices * a.Mp * a.Np + (long long)slice * a.Mp;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) bo[tm * WG_TM + 64 * wm + 16 * mt + 4 * g + r] = bacc[mt][r];
    }
}

// fp32 [rows][cols] (row stride ld, 8-byte aligned rows) -> bf16 planes [rows][ldp]; columns cols .. ldp-1 zero.  One thread per 4 columns.
__global__ __launch_bounds__(256) void split_rows_natural_kernel(const float* __restrict__ src, long long ld, int rows, int cols, unsigned short* __restrict__ hi,
                                                                  unsigned short* __restrict__ lo, int ldp) {
    const int q = ldp / 4;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)rows * q) return;
    const int r = (int)(i / q), c = 4 * (int)(i - (long long)r * q);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const float* p = src + (long long)r * ld + c;
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
        if (c + e + 1 < cols) {
            const f32x2_t w = *reinterpret_cast<const f32x2_t*>(p + e);
            v[e] = w[0]; v[e + 1] = w[1];
        } else if (c + e < cols) v[e] = p[e];
    }
    u32x2_t h, l;
    x3_split4(v[0], v[1], v[2], v[3], h, l);
    *reinterpret_cast<u32x2_t*>(hi + (long long)r * ldp + c) = h;
    *reinterpret_cast<u32x2_t*>(lo + (long long)r * ldp + c) = l;
}

}  // namespace eeg

using namespace eeg;

static inline int wg_pad(int v, int m) { return (v + m - 1) / m * m; }

extern "C" int eegclip_split_transpose(const float* src, long long ld, int rows, int cols, int out_rows, void* hi, void* lo, long long ldo, void* stream) {
    if (!src || !hi || !lo || rows < 1 || cols < 1 || out_rows < cols || (out_rows & 63) || (rows & 63) || ldo < rows || (ldo & 7)) return EEGCLIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(src) & 3u) || ((reinterpret_cast<uintptr_t>(hi) | reinterpret_cast<uintptr_t>(lo)) & 15u)) return EEGCLIP_EALIGN;
    EEG_LAUNCH(split_transpose_kernel, dim3((unsigned)(rows / 64), (unsigned)(out_rows / 64)), dim3(256), 64 * 65 * sizeof(float), stream, src, ld, rows, cols,
               static_cast<unsigned short*>(hi), static_cast<unsigned short*>(lo), ldo);
    return (int)hipGetLastError();
}

static int wg_slices(int Mp, int Np, int K) {
    const int tiles = (Mp / WG_TM) * (Np / WG_TN), kt = K / WG_BK;
    int s = (256 + tiles - 1) / tiles;                           // ~one workgroup (96 KB of LDS) per CU
    if (s > kt / 4) s = kt / 4;                                  // at least 4 k-tiles per workgroup
    if (s >= 8) s = (s + 4) / 8 * 8;                             // a multiple of 8: the XCD-aware order of the kernel
    if (s > kt) s = kt;
    return s < 1 ? 1 : s;
}

extern "C" long long eegclip_wgrad_planes_workspace_floats(int M, int N, int K) {
    if (M < 1 || N < 1 || K < WG_BK) return 0;
    const int Mp = wg_pad(M, WG_TM), Np = wg_pad(N, WG_TN);
    return (long long)wg_slices(Mp, Np, K) * ((long long)Mp * Np + Mp);
}

// out (M x N, row stride ldo) += A^T B with A = planes [>= pad128(M)][K], B = planes [>= pad64(N)][K] (rows beyond M / N zero); bias_out[m] += sum_k A[m][k].
// ld = elements between plane rows.  K * 2 bytes is a power of two for the batch sizes of the path (16384 tokens -> 32 KB): with ld == K every row's
// k-tile sits at the same offset modulo the L2 / HBM channel interleave and a workgroup's 192 row segments hammer a few channels (measured: 16 us
// for the 250 x 256 x 16384 kernel); ld = K + 64 (one 128-byte line of skew per row) spreads them.
extern "C" int eegclip_wgrad_planes(const void* a_hi, const void* a_lo, const void* b_hi, const void* b_lo, long long ld, int M, int N, int K, float* out,
                                    long long ldo, float* bias_out, float* workspace, void* stream) {
    if (!a_hi || !a_lo || !b_hi || !b_lo || !out || !workspace || M < 1 || N < 1 || K < WG_BK || (K % WG_BK) || ldo < N || ld < K || (ld & 7)) return EEGCLIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(a_hi) | reinterpret_cast<uintptr_t>(a_lo) | reinterpret_cast<uintptr_t>(b_hi) | reinterpret_cast<uintptr_t>(b_lo)) & 15u)
        return EEGCLIP_EALIGN;
    const int Mp = wg_pad(M, WG_TM), Np = wg_pad(N, WG_TN);
    wgrad_args a{static_cast<const unsigned short*>(a_hi), static_cast<const unsigned short*>(a_lo), static_cast<const unsigned short*>(b_hi),
                 static_cast<const unsigned short*>(b_lo), workspace, Mp, Np, K, wg_slices(Mp, Np, K), bias_out ? 1 : 0, ld};
    const int tiles = (Mp / WG_TM) * (Np / WG_TN);
    EEG_LAUNCH(wgrad_planes_kernel, dim3((unsigned)(tiles * a.slices)), dim3(256), 2 * WG_STAGE, stream, a);
    const long long total = (long long)M * N + (bias_out ? M : 0);
    EEG_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, workspace, a.slices, Mp, Np, M, N, out, ldo, bias_out);
    return (int)hipGetLastError();
}

extern "C" int eegclip_split_rows_natural(const float* src, long long ld, int rows, int cols, void* hi, void* lo, int ldp, void* stream) {
    if (!src || !hi || !lo || rows < 1 || cols < 1 || ld < cols || (ld & 1) || ldp < cols || (ldp & 7)) return EEGCLIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(src) & 7u) || ((reinterpret_cast<uintptr_t>(hi) | reinterpret_cast<uintptr_t>(lo)) & 15u)) return EEGCLIP_EALIGN;
    const long long n = (long long)rows * (ldp / 4);
    EEG_LAUNCH(split_rows_natural_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, ld, rows, cols, static_cast<unsigned short*>(hi),
               static_cast<unsigned short*>(lo), ldp);
    return (int)hipGetLastError();
}

static int wt_slices(int Mp, int Np, int K) {
    const int tiles = (Mp / WG_TM) * (Np / WG_TN), kt = K / WT_BK;
    int s = (512 + tiles - 1) / tiles;                           // ~two workgroups (53 KB of LDS each) per CU
    if (s > kt / 4) s = kt / 4;
    if (s >= 8) s = (s + 4) / 8 * 8;
    if (s > kt) s = kt;
    return s < 1 ? 1 : s;
}

extern "C" long long eegclip_wgrad_tr_workspace_floats(int M, int N, int K) {
    if (M < 1 || N < 1 || K < WT_BK) return 0;
    const int Mp = wg_pad(M, WG_TM), Np = wg_pad(N, WG_TN);
    return (long long)wt_slices(Mp, Np, K) * ((long long)Mp * Np + Mp);
}

// out (M x N, row stride ldo) += A^T B with A = planes [K][lda], B = planes [K][ldb] in natural (token-major) layout, lda / ldb multiples of 8 with
// zeros in the columns past M / N (eegclip_split_rows_natural writes them); bias_out[m] += sum_k A[k][m].  K a multiple of 32.
extern "C" int eegclip_wgrad_tr(const void* a_hi, const void* a_lo, long long lda, const void* b_hi, const void* b_lo, long long ldb, int M, int N, int K,
                                float* out, long long ldo, float* bias_out, float* workspace, void* stream) {
    if (!a_hi || !a_lo || !b_hi || !b_lo || !out || !workspace || M < 1 || N < 1 || K < WT_BK || (K % WT_BK) || ldo < N || lda < M || ldb < N || (lda & 7) ||
        (ldb & 7))
        return EEGCLIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(a_hi) | reinterpret_cast<uintptr_t>(a_lo) | reinterpret_cast<uintptr_t>(b_hi) | reinterpret_cast<uintptr_t>(b_lo)) & 15u)
        return EEGCLIP_EALIGN;
    const int Mp = wg_pad(M, WG_TM), Np = wg_pad(N, WG_TN);
    wgrad_tr_args a{static_cast<const unsigned short*>(a_hi), static_cast<const unsigned short*>(a_lo), static_cast<const unsigned short*>(b_hi),
                    static_cast<const unsigned short*>(b_lo), workspace, M, N, Mp, Np, K, wt_slices(Mp, Np, K), bias_out ? 1 : 0, lda, ldb};
    const int tiles = (Mp / WG_TM) * (Np / WG_TN);
    EEG_LAUNCH(wgrad_tr_kernel, dim3((unsigned)(tiles * a.slices)), dim3(256), 2 * WT_STAGE, stream, a);
    const long long total = (long long)M * N + (bias_out ? M : 0);
    EEG_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, workspace, a.slices, Mp, Np, M, N, out, ldo, bias_out);
    return (int)hipGetLastError();
}
