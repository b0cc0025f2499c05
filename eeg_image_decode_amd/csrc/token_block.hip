// The transformer block of the ATM-S encoder as ONE workgroup per sample (models/subject_layers/Embed.py:141-162, SelfAttention_Family.py:56-75,
// 194-213, Transformer_EncDec.py:39-51,61-80): value embedding + positional embedding + subject token + dropout, fused Q|K|V projection, 4-head
// softmax attention with probability dropout, output projection, dropout + residual + LayerNorm, FFN 250 -> 256 (GELU, dropout) -> 250,
// dropout + residual + LayerNorm, final LayerNorm.
//
// Why one kernel: a sample is 64 tokens x 250 features (64 KB of fp32) -- it fits the 160 KB LDS of a CU with room to spare, and every Linear of
// the block has K ~ 250: as ten separate launches (round 2) the path was bound by per-launch prologue / epilogue and by ~600 MB per step of
// activation round trips through HBM, not by the contractions (16 of 27 us of a 16384 x 256 x 250 launch with an EMPTY k-loop).
// Here activations stay in LDS between the stages; HBM sees the EEG sample once and each tensor the backward needs once (written, never re-read,
// except the 64 KB residual `h` that comes back from L2).
//
// Arithmetic = the plan GEMMs' (csrc/gemm_x3.hip): fp32 in / out / accumulate, products as split bf16 (a = a_hi + a_lo, a b ~ a_hi b_hi + a_hi b_lo
// + a_lo b_hi on v_mfma_f32_16x16x32_bf16).  Activations are split ONCE by their producer into hi | lo planes in LDS; weights are split once per
// step by eegclip_token_block_pack into MFMA-fragment order, so a wave's B operand is one coalesced 1 KB load per (16 outputs x 32 k) tile
// straight into registers (each weight element is used by exactly one wave of a workgroup: LDS staging would be a pure detour).
//
// Workgroup = 512 threads = 8 waves (2 per SIMD, 256 VGPRs each).  LDS (160 KB, all of a CU):
//   AP  [2 planes][64 rows][256 k] bf16   A operand of the current Linear; 512-byte rows, 16-byte chunk index XOR (row & 15): conflict-free
//                                         ds_read_b128 fragment fetches
//   XF  [64][256] fp32                    row-wise work (dropout, residual, LayerNorm); 1 KB rows, 16-byte chunk index XOR (row & 15)
//   HQ  2 x 48 KB behind AP (over XF, which is dead then): the q | k | v^T (| P) planes of two heads ([64][64] bf16 hi | lo, 128-byte rows).
// Barriers are s_barrier + lgkmcnt(0) only (raw_barrier): __syncthreads() also drains vmcnt, and on CDNA4 that counts the STORES -- every
// stage writes something the backward needs, so each barrier would wait for a full HBM write round trip (the first version: 64 % of all wave
// cycles waiting, 123 us; PMC in profiles/r3_pmc_token_block.json).
// Dropout masks are Philox(seed, site, flat element index) exactly as in the unfused kernels (csrc/elementwise.hip, norm.hip, attention.hip,
// gemm_epilogue.h): the backward regenerates them, tests regenerate them in numpy.
#include "cstack_common.h"

#include <stdio.h>
#include <stdlib.h>

namespace eeg {

constexpr int TB_THREADS = 512;
constexpr int TB_L = 64, TB_D = 250, TB_T = 250, TB_NCH = 63, TB_H = 4, TB_E = 62, TB_HE = 248, TB_FF = 256;
constexpr int TB_AP_PLANE = 64 * 512;                       // bytes of one A plane
constexpr int TB_AP_BYTES = 2 * TB_AP_PLANE;
constexpr int TB_XF_BYTES = 64 * 1024;
constexpr int TB_HQ_PLANE = 64 * 128;                       // one [64][64] bf16 plane of a head
constexpr int TB_HQ_BYTES = 6 * TB_HQ_PLANE;                // q | k | v^T, hi | lo each
constexpr int TB_LDS = TB_AP_BYTES + 2 * TB_HQ_BYTES;       // 160 KB: the h planes + two heads' planes (XF is the first 64 KB behind AP)
// packed weights: per (n-tile of 16, k-step of 32): 64 lanes x 8 bf16 of the hi plane, then of the lo plane (MFMA B / A fragment order)
constexpr int TB_KS = 8;
constexpr int TB_TILE = 1024;                               // bf16 elements per (n-tile, k-step)
constexpr int TB_NT_V = 16, TB_NT_QKV = 48, TB_NT_O = 16, TB_NT_1 = 16, TB_NT_2 = 16;
constexpr int TB_NMAT = 11;                                 // 5 forward operands, 6 backward (transposed) ones of 16 n-tiles each
constexpr long long TB_OFF_V = 0;
constexpr long long TB_OFF_QKV = TB_OFF_V + (long long)TB_NT_V * TB_KS * TB_TILE;
constexpr long long TB_OFF_O = TB_OFF_QKV + (long long)TB_NT_QKV * TB_KS * TB_TILE;
constexpr long long TB_OFF_1 = TB_OFF_O + (long long)TB_NT_O * TB_KS * TB_TILE;
constexpr long long TB_OFF_2 = TB_OFF_1 + (long long)TB_NT_1 * TB_KS * TB_TILE;
constexpr long long TB_MAT_ELEMS = 16LL * TB_KS * TB_TILE;  // one 16-tile operand
constexpr long long TB_OFF_2T = TB_OFF_2 + TB_MAT_ELEMS;    // W2^T   (dg1 = df2 W2)
constexpr long long TB_OFF_1T = TB_OFF_2T + TB_MAT_ELEMS;   // W1^T   (dn1 += df1 W1)
constexpr long long TB_OFF_OT = TB_OFF_1T + TB_MAT_ELEMS;   // Wo^T   (dctx = da1 Wo), output columns 64 head + d
constexpr long long TB_OFF_QT = TB_OFF_OT + TB_MAT_ELEMS;   // Wq^T | Wk^T | Wv^T (dh = dq Wq + dk Wk + dv Wv), k = 64 head + d: three operands
constexpr long long TB_PACKED_ELEMS = TB_OFF_QT + 3 * TB_MAT_ELEMS;

typedef float tb_f4u __attribute__((ext_vector_type(4), aligned(4)));      // 16-byte global access from a dword-aligned address
typedef float tb_f2 __attribute__((ext_vector_type(2)));
typedef unsigned tb_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int ap_off(int row, int k) { return row * 512 + ((((k >> 3) ^ (row & 15))) << 4) + ((k & 7) << 1); }
__device__ __forceinline__ int xf_off(int row, int col) { return row * 1024 + ((((col >> 2) ^ (row & 15))) << 4) + ((col & 3) << 2); }
__device__ __forceinline__ int hq_off(int row, int k) { return row * 128 + ((((k >> 3) ^ ((row >> 1) & 7))) << 4) + ((k & 7) << 1); }

// source element of the packed matrix `mat` at (n, k), or -1 (zero padding)
//   0  value embedding  (n < 250, k < 250)            Wv[n][k]
//   1  q | k | v        n = 64 seg + d, seg = 3 head + which, d < 62     Wqkv[which 248 + head 62 + d][k]
//   2  out projection   (n < 250), k = 64 head + d, d < 62               Wo[n][62 head + d]
//   3  FFN 1            (n < 256, k < 250)            W1[n][k]
//   4  FFN 2            (n < 250, k < 256)            W2[n][k]
// backward operands (B[k][n] = W[k][n]: the Linear's weight read transposed):
//   5  dg1 = df2 W2     (n < 256, k < 250)            W2[k][n]
//   6  dn1 += df1 W1    (n < 250, k < 256)            W1[k][n]
//   7  dctx = da1 Wo    n = 64 head + d, d < 62, (k < 250)               Wo[k][62 head + d]
//   8+which  dh += d{q,k,v} W{q,k,v}   (n < 250), k = 64 head + d        Wqkv[which 248 + 62 head + d][n]
__device__ __forceinline__ long long tb_src_index(int mat, int n, int k) {
    switch (mat) {
        case 0: return (n < TB_D && k < TB_T) ? (long long)n * TB_T + k : -1;
        case 1: {
            const int seg = n >> 6, d = n & 63, head = seg / 3, which = seg - 3 * head;
            return (seg < 12 && d < TB_E && k < TB_D) ? (long long)(which * TB_HE + head * TB_E + d) * TB_D + k : -1;
        }
        case 2: {
            const int head = k >> 6, d = k & 63;
            return (n < TB_D && head < TB_H && d < TB_E) ? (long long)n * TB_HE + head * TB_E + d : -1;
        }
        case 3: return (n < TB_FF && k < TB_D) ? (long long)n * TB_D + k : -1;
        case 4: return (n < TB_D && k < TB_FF) ? (long long)n * TB_FF + k : -1;
        case 5: return (n < TB_FF && k < TB_D) ? (long long)k * TB_FF + n : -1;
        case 6: return (n < TB_D && k < TB_FF) ? (long long)k * TB_D + n : -1;
        case 7: {
            const int head = n >> 6, d = n & 63;
            return (head < TB_H && d < TB_E && k < TB_D) ? (long long)k * TB_HE + head * TB_E + d : -1;
        }
        default: {
            const int head = k >> 6, d = k & 63;
            return (n < TB_D && head < TB_H && d < TB_E) ? (long long)((mat - 8) * TB_HE + head * TB_E + d) * TB_D + n : -1;
        }
    }
}

struct tb_pack_args {
    const float* w[TB_NMAT];
    unsigned short* out;
    // (eegclip_weight_prep) the conv stack's weight fragments ride in the same launch: workgroups tb_tiles .. run csrc/cstack_common.h: cs_pack_both
    int tb_tiles;
    const float* Ws;
    unsigned char *cs_packed, *cs_packed_t;
    int H;
};

// one 64-thread workgroup per (matrix, n-tile, k-step): lane l owns the 8 k of fragment lane l
__global__ __launch_bounds__(64) void token_block_pack_kernel(const tb_pack_args a) {
    if ((int)blockIdx.x >= a.tb_tiles) {
        cs_pack_both(a.Ws, a.cs_packed, a.cs_packed_t, a.H, ((int)blockIdx.x - a.tb_tiles) * 64 + (int)(threadIdx.x & 63));
        return;
    }
    int tile = blockIdx.x;
    int mat = 0;
    while (mat < TB_NMAT - 1 && tile >= (mat == 1 ? TB_NT_QKV : 16) * TB_KS) { tile -= (mat == 1 ? TB_NT_QKV : 16) * TB_KS; ++mat; }
    const int nt = tile / TB_KS, ks = tile % TB_KS;
    const int lane = threadIdx.x & 63;
    const int n = 16 * nt + (lane & 15), k0 = 32 * ks + 8 * (lane >> 4);
    const float* src = a.w[mat];
    unsigned hb[8], lb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const long long si = tb_src_index(mat, n, k0 + e);
        const float v = si >= 0 ? src[si] : 0.f;
        const unsigned short h = f32_to_bf16_bits(v);
        hb[e] = h;
        lb[e] = f32_to_bf16_bits(v - bf16_bits_to_f32(h));
    }
    unsigned short* dst = a.out + (long long)blockIdx.x * TB_TILE + lane * 8;
    tb_u4 oh, ol;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        oh[e] = hb[2 * e] | (hb[2 * e + 1] << 16);
        ol[e] = lb[2 * e] | (lb[2 * e + 1] << 16);
    }
    *reinterpret_cast<tb_u4*>(dst) = oh;
    *reinterpret_cast<tb_u4*>(dst + 512) = ol;
}

// ---- C[64 x 16 NT] of one wave: A = the AP planes (all 64 rows, K = 256), B = NT packed n-tiles starting at `wp`.
//      bit j of TM set: tile j is formed TRANSPOSED (MFMA rows = n): lane (fr, g) then holds n = 16 j + 4 g + i of row m = 16 mt + fr;
//      clear: standard, lane holds rows m = 16 mt + 4 g + i of column n = 16 j + fr.
#ifndef TB_FWD_RING
#define TB_FWD_RING false                                   // the forward kernel's GEMMs: compiler-scheduled weight loads (see tb_gemm)
#endif
#ifndef TB_QKV_RING
#define TB_QKV_RING false                                   // ... and its Q | K | V GEMM (NT = 3: 72 ring registers on top of 48 accumulators)
#endif
#ifndef TB_BWD_PF
#define TB_BWD_PF 2                                         // k-steps of weight fragments in flight in the backward kernels' GEMMs
#endif
template <int NT, unsigned TM, int PF = 2, bool ZERO = true, bool RING = false>
__device__ __forceinline__ void tb_gemm(const unsigned char* AP, const unsigned short* wp, int lane, f32x4 (&acc)[4][NT]) {
    const int fr = lane & 15, g = lane >> 4;
    if (ZERO) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // the B fragments of k-step ks live in ring slot ks % (PF + 1): PF k-steps of weight loads (L2 round trips) in flight under the MFMAs.
    // RING: the loads are inline asm with hand-counted waits: left to the compiler, each load is scheduled 4..20 MFMAs ahead of its use (it sinks
    // prefetches to shorten live ranges) -- a fraction of the L2 latency -- and every k-step stalls.  Loads retire in order, so "at most
    // 2 NT x (younger k-steps) operations outstanding" means the fragments of this k-step have landed; the wait names them as read-write so
    // that no use is scheduled above it (tests/test_asm_rings.py checks the generated code: the compiler must not copy or spill a register whose
    // load is in flight).  Measured: backward part 0 73 -> 64 us, part 1 37 -> 31 us; the FORWARD kernel, already at the 256-VGPR limit, spills
    // more with the ring pinned and gets slower (144 -> 151 us train, 107 -> 135 us eval), so it keeps the compiler's schedule.
    bf16x8 bh[PF + 1][NT], bl[PF + 1][NT];
    auto issue = [&](int ks) {
        const int slot = ks % (PF + 1);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const bf16x8* q = reinterpret_cast<const bf16x8*>(wp + (long long)(j * TB_KS + ks) * TB_TILE) + lane;
#if defined(EEG_EMU)
            bh[slot][j] = q[0];
            bl[slot][j] = q[64];
#else
            if constexpr (RING) {
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bh[slot][j]) : "v"(q) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(bl[slot][j]) : "v"(q) : "memory");
            } else {
                bh[slot][j] = q[0];
                bl[slot][j] = q[64];
            }
#endif
        }
    };
    auto landed = [&](int ks) {
#if !defined(EEG_EMU)
        if constexpr (!RING) return;
        static_assert(NT == 2 || NT == 3, "the wait statements list 2 NT fragment registers");
        const int slot = ks % (PF + 1);
        const int younger = TB_KS - 1 - ks < PF ? TB_KS - 1 - ks : PF;          // k-steps issued after this one
#define TB_WAIT(N)                                                                                                                       \
    do {                                                                                                                                 \
        if constexpr (NT == 2)                                                                                                           \
            asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(bh[slot][0]), "+v"(bl[slot][0]), "+v"(bh[slot][1]), "+v"(bl[slot][1])::"memory"); \
        else                                                                                                                             \
            asm volatile("s_waitcnt vmcnt(" #N ")"                                                                                       \
                         : "+v"(bh[slot][0]), "+v"(bl[slot][0]), "+v"(bh[slot][1]), "+v"(bl[slot][1]), "+v"(bh[slot][NT - 1]), "+v"(bl[slot][NT - 1]) \
                         :                                                                                                               \
                         : "memory");                                                                                                    \
    } while (0)
        const int n = 2 * NT * younger;
        if (n == 0) TB_WAIT(0);
        else if (n == 4) TB_WAIT(4);
        else if (n == 6) TB_WAIT(6);
        else if (n == 8) TB_WAIT(8);
        else if (n == 12) TB_WAIT(12);
        else if (n == 16) TB_WAIT(16);
        else if (n == 18) TB_WAIT(18);
        else TB_WAIT(0);
#undef TB_WAIT
#endif
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) issue(p);
#pragma unroll
    for (int ks = 0; ks < TB_KS; ++ks) {
        const int cur = ks % (PF + 1);
        if (ks + PF < TB_KS) issue(ks + PF);
        landed(ks);
        // (pinning the prefetch issue here with sched_barrier(0) measured SLOWER -- 140 vs 120 us per launch: it also stops the scheduler from
        //  running the next k-step's fragment reads under this k-step's MFMAs)
        bf16x8 ah[4], al[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int off = (16 * mt + fr) * 512 + (((4 * ks + g) ^ fr) << 4);
            ah[mt] = *reinterpret_cast<const bf16x8*>(AP + off);
            al[mt] = *reinterpret_cast<const bf16x8*>(AP + TB_AP_PLANE + off);
        }
        // product-major: the three MFMAs of one accumulator are 4 NT instructions apart (back to back they wait for each other's result)
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const bf16x8 bw = pr == 0 ? bl[cur][j] : bh[cur][j];
                    const bf16x8 aw = pr == 1 ? al[mt] : ah[mt];
                    if ((TM >> j) & 1u) acc[mt][j] = mfma_bf16_16x16x32(bw, aw, acc[mt][j]);
                    else acc[mt][j] = mfma_bf16_16x16x32(aw, bw, acc[mt][j]);
                }
    }
}

// 4 consecutive-k fp32 -> hi | lo bf16 (8 bytes each) at the same offset of both planes
__device__ __forceinline__ void tb_store_planes4(unsigned char* hi_plane, int plane_stride, int off, float v0, float v1, float v2, float v3) {
    u32x2_t h, l;
    x3_split4(v0, v1, v2, v3, h, l);
    *reinterpret_cast<u32x2_t*>(hi_plane + off) = h;
    *reinterpret_cast<u32x2_t*>(hi_plane + plane_stride + off) = l;
}
__device__ __forceinline__ void tb_store_planes2(unsigned char* hi_plane, int plane_stride, int off, float v0, float v1) {
    const unsigned h = x3_pack2(v0, v1);
    const float r0 = v0 - __uint_as_float(h << 16), r1 = v1 - __uint_as_float(h & 0xffff0000u);
    *reinterpret_cast<unsigned*>(hi_plane + off) = h;
    *reinterpret_cast<unsigned*>(hi_plane + plane_stride + off) = x3_pack2(r0, r1);
}

// 4 floats p[0..3] of which the first `valid` (2 or 4) exist; p is 8-byte aligned
__device__ __forceinline__ f32x4 tb_ld4(const float* p, int valid) {
    if (valid >= 4) {
        const tb_f4u t = *reinterpret_cast<const tb_f4u*>(p);
        return f32x4{t[0], t[1], t[2], t[3]};
    }
    const tb_f2 t = *reinterpret_cast<const tb_f2*>(p);
    return f32x4{t[0], t[1], 0.f, 0.f};
}
__device__ __forceinline__ void tb_st4(float* p, int valid, f32x4 v) {
    if (valid >= 4) *reinterpret_cast<tb_f4u*>(p) = tb_f4u{v[0], v[1], v[2], v[3]};
    else *reinterpret_cast<tb_f2*>(p) = tb_f2{v[0], v[1]};
}


// the AP image of this sample (both planes, 64 rows x 256 channels) -> its 64 KB block of a token-plane tensor (csrc/wgrad_tok.hip: [hi | lo][64][256],
// unswizzled): chunk q = t + 512 j of 16 bytes = (plane q >> 11, row (q >> 5) & 63, chunk q & 31).  ONES_FROM < 64: channel 255 of the hi plane of rows
// >= ONES_FROM := 1.0 -- column 255 of X^T dY is then the bias gradient.  Called AFTER the k-loop of the GEMM that consumes the planes (its weight
// loads have been issued: a load behind these stores would wait for them) and before the barrier that lets the next stage overwrite AP.
template <int ONES_FROM>
__device__ __forceinline__ void tb_planes_out(const unsigned char* AP, unsigned char* dst, int t) {
    // row = (t >> 5) + 16 (j & 3), so row & 15 = t >> 5 for every j: ONE per-lane LDS offset and ONE per-lane global offset, everything else is an
    // immediate / a scalar add (eight hoisted address pairs made the forward kernel spill 200 registers)
    const unsigned lds_off = (unsigned)((t >> 5) * 512 + (((t & 31) ^ (t >> 5)) << 4)), g_off = (unsigned)t * 16u;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        tb_u4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = *reinterpret_cast<const tb_u4*>(AP + half * TB_AP_PLANE + j * 8192 + lds_off);
            if (ONES_FROM < 64 && half == 0 && (t & 31) == 31 && (ONES_FROM == 0 || j > 0 || (t >> 5) >= ONES_FROM)) v[j][3] = (v[j][3] & 0xffffu) | 0x3F800000u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<tb_u4*>(dst + (half * 4 + j) * 8192 + g_off) = v[j];
    }
}

struct tb_fwd_args {
    const float* x;
    const unsigned short* packed;
    const float *bv, *pe, *tokens;
    const long long* ids;
    const float *bqkv, *bo, *ln1_g, *ln1_b, *b1, *b2, *ln2_g, *ln2_b, *ln3_g, *ln3_b;
    float *h, *qkv, *r1, *mu1, *rs1, *f1, *r2, *n2, *mu2, *rs2, *n3, *mu3, *rs3;
    unsigned char *xp, *hp, *ctxp, *n1p, *g1p;                   // token planes (64 KB per sample) of the weight-gradient GEMMs' X operands; xp may be null
    float drop_p, eps, scale;
    unsigned long long seed;
    unsigned site_embed, site_attn, site_attn_out, site_ffn_act, site_ffn_out;
    unsigned dbg;                                                // diagnosis (EEGCLIP_TB_DEBUG): bit 0 = leave out the activation stores (timing ablation only)
    unsigned long long* tstamp;                                  // diagnosis (bit 1): [B][16] s_memtime stamps of wave 0 at the phase boundaries
    const unsigned short* packed_embed;                          // joint-subject model: one packed value embedding per subject (null: the one in `packed`)
    const int* embed_subject;                                    // ... (B) the sample's subject: matrix and bias row
    long long bv_stride;                                         // ... floats between the subjects' biases
    const float *cs_w25, *cs_bias;                               // the conv stack's BatchNorm1 batch sums of this sample as the kernel's tail (cs_rows non-null):
    double* cs_rows;                                             // ... cs_stats1_sample (cstack_common.h) over the n3 rows just written
    int cs_H;
};

// The forward kernel takes ~50 pointers (backward part A: 25).  hipcc loads every kernel argument in the entry block and keeps it in scalar registers until its last use, so the
// phases ran with 104 SGPRs + 99 spilled to vector lanes: every use of a spilled value (the Philox round keys of each dropout site, the per-row store
// bases) was a v_readlane + hazard nops inside the row loops.  Each phase therefore reads ITS arguments again from the kernel-argument segment through a
// pointer the optimiser cannot see through: scalar loads at the phase's start, dead at its end.
template <class ARGS>
__device__ __forceinline__ ARGS tb_args_again(const ARGS& a) {
#if defined(EEG_EMU) || !defined(__HIP_DEVICE_COMPILE__)
    return a;                                                    // (the emulator, and the host pass of hipcc over this device function)
#else
    auto p = (const __attribute__((address_space(4))) ARGS*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *p;                                                   // by value: one scalar load per field the phase uses, none for the others
#endif
}

// phase stamp of the diagnosis build path: the shader clock as wave 0 passes phase boundary k
__device__ __forceinline__ void tb_stamp(const tb_fwd_args& a, int b, int t, int k) {
#if !defined(EEG_EMU)
    if (a.tstamp && t == 0) a.tstamp[b * 16 + k] = __builtin_amdgcn_s_memtime();
#endif
}

// one QKV tile of this wave (+ bias, handed in: TRANS = the 4 consecutive dims of the lane, else its one dim in all four) -> global qkv (natural
// (row, 744) layout); values stay in `v` (biased)
template <bool TRANS>
__device__ __forceinline__ void tb_qkv_tile_out(const tb_fwd_args& a, int b, int head, int which, int tt, int lane, f32x4 bias, f32x4 (&v)[4]) {
    const int fr = lane & 15, g = lane >> 4;
    const int cbase = which * TB_HE + head * TB_E;
    if (TRANS) {
        const int d0 = 16 * tt + 4 * g, valid = TB_E - d0;              // >= 4, or 2 at d0 = 60
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = 16 * mt + fr;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[mt][i] = i < valid ? v[mt][i] + bias[i] : 0.f;
            if (!(a.dbg & 1u)) tb_st4(a.qkv + (long long)b * (TB_L * 3 * TB_HE) + (unsigned)(m * (3 * TB_HE) + cbase + d0), valid, v[mt]);
        }
    } else {
        const int d = 16 * tt + fr;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[mt][i] = d < TB_E ? v[mt][i] + bias[0] : 0.f;
                if (d < TB_E && !(a.dbg & 1u)) (a.qkv + (long long)b * (TB_L * 3 * TB_HE))[(unsigned)((16 * mt + 4 * g + i) * (3 * TB_HE) + cbase + d)] = v[mt][i];
            }
    }
}
// ... -> the head's planes in LDS: q / k as [token][dim], v as [dim][token] (hi plane, lo plane 8 KB behind)
template <bool TRANS>
__device__ __forceinline__ void tb_qkv_tile_lds(unsigned char* HQ, int which, int tt, int lane, const f32x4 (&v)[4]) {
    const int fr = lane & 15, g = lane >> 4;
    unsigned char* base = HQ + which * (2 * TB_HQ_PLANE);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int off = TRANS ? hq_off(16 * mt + fr, 16 * tt + 4 * g) : hq_off(16 * tt + fr, 16 * mt + 4 * g);
        tb_store_planes4(base, TB_HQ_PLANE, off, v[mt][0], v[mt][1], v[mt][2], v[mt][3]);
    }
}

// V = w & 3: which 3 of the 12 (q t0..3 | k t0..3 | v t0..3) tiles of a head this wave owns: tiles 3 V .. 3 V + 2
template <int V>
struct tb_qkv_variant {
    static constexpr unsigned mask = V == 0 ? 7u : V == 1 ? 7u : V == 2 ? 3u : 0u;      // q and k transposed, v standard
};

template <int V>
__device__ __forceinline__ void tb_qkv_pass(const tb_fwd_args& a, const unsigned char* AP, int b, int head, int lane, f32x4 (&acc)[4][3]) {
    tb_gemm<3, tb_qkv_variant<V>::mask, 2, true, TB_QKV_RING>(AP, a.packed + TB_OFF_QKV + (long long)(head * 12 + 3 * V) * TB_KS * TB_TILE, lane, acc);
    // the three tiles' biases first (loads), then every store of the pass: a bias load behind the previous tile's stores would wait for them
    const int fr = lane & 15, g = lane >> 4;
    f32x4 bias[3];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        const int j = 3 * V + jj, which = j >> 2, tt = j & 3, cbase = which * TB_HE + head * TB_E;
        if ((tb_qkv_variant<V>::mask >> jj) & 1u) {
            const int d0 = 16 * tt + 4 * g;
            bias[jj] = tb_ld4(a.bqkv + cbase + d0, TB_E - d0 >= 4 ? 4 : 2);
        } else {
            const int d = 16 * tt + fr;
            const float bv = d < TB_E ? a.bqkv[cbase + d] : 0.f;
            bias[jj] = f32x4{bv, bv, bv, bv};
        }
    }
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        const int j = 3 * V + jj, which = j >> 2, tt = j & 3;
        f32x4 v[4] = {acc[0][jj], acc[1][jj], acc[2][jj], acc[3][jj]};
        if ((tb_qkv_variant<V>::mask >> jj) & 1u) tb_qkv_tile_out<true>(a, b, head, which, tt, lane, bias[jj], v);
        else tb_qkv_tile_out<false>(a, b, head, which, tt, lane, bias[jj], v);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][jj] = v[mt];
    }
}
template <int V>
__device__ __forceinline__ void tb_qkv_to_lds(unsigned char* HQ, int lane, const f32x4 (&acc)[4][3]) {
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        const int j = 3 * V + jj, which = j >> 2, tt = j & 3;
        const f32x4 v[4] = {acc[0][jj], acc[1][jj], acc[2][jj], acc[3][jj]};
        if ((tb_qkv_variant<V>::mask >> jj) & 1u) tb_qkv_tile_lds<true>(HQ, which, tt, lane, v);
        else tb_qkv_tile_lds<false>(HQ, which, tt, lane, v);
    }
}

// one head, one query tile (16 queries) per wave: S^T = K Q^T (rows = keys), softmax over keys in-lane + 2 shuffles, probability dropout,
// P -> the wave's own (dead) q rows, ctx^T = V^T P^T.  c[dt][i] = ctx[query 16 qt + fr][dim 16 dt + 4 g + i]
template <bool TRAIN>
__device__ __forceinline__ void tb_attention(const tb_fwd_args& a, unsigned char* HQ, int b, int head, int qt, int lane, f32x4 (&c)[4]) {
    const int fr = lane & 15, g = lane >> 4;
    unsigned char* Q = HQ;
    const unsigned char* K = HQ + 2 * TB_HQ_PLANE;
    const unsigned char* VT = HQ + 4 * TB_HQ_PLANE;
    f32x4 s[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int qo = hq_off(16 * qt + fr, 32 * ks + 8 * g);
        const bf16x8 qh = *reinterpret_cast<const bf16x8*>(Q + qo), ql = *reinterpret_cast<const bf16x8*>(Q + TB_HQ_PLANE + qo);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int ko = hq_off(16 * kt + fr, 32 * ks + 8 * g);
            const bf16x8 kh = *reinterpret_cast<const bf16x8*>(K + ko), kl = *reinterpret_cast<const bf16x8*>(K + TB_HQ_PLANE + ko);
            s[kt] = mfma_bf16_16x16x32(kl, qh, s[kt]);
            s[kt] = mfma_bf16_16x16x32(kh, ql, s[kt]);
            s[kt] = mfma_bf16_16x16x32(kh, qh, s[kt]);          // S[key 16 kt + 4 g + i][query 16 qt + fr]
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s[kt][i] *= a.scale;
            mx = fmaxf(mx, s[kt][i]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s[kt][i] = expf(s[kt][i] - mx);
            sum += s[kt][i];
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    const float ksc = (TRAIN && a.drop_p > 0.f) ? 1.f / (1.f - a.drop_p) : 1.f;
    wave_sync();                                                 // every lane has its q fragments: the rows may be overwritten
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        bool keep[4] = {true, true, true, true};
        if (TRAIN && a.drop_p > 0.f) {
            const unsigned long long idx0 = (((unsigned long long)(b * TB_H + head) * TB_L + 16 * qt + fr) * TB_L) + 16 * kt + 4 * g;
            dropout_keep4(a.seed, a.site_attn, idx0, a.drop_p, keep);
        }
        float p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = keep[i] ? s[kt][i] * inv * ksc : 0.f;
        tb_store_planes4(Q, TB_HQ_PLANE, hq_off(16 * qt + fr, 16 * kt + 4 * g), p[0], p[1], p[2], p[3]);
    }
    wave_sync();
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) c[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int po = hq_off(16 * qt + fr, 32 * ks + 8 * g);
        const bf16x8 ph = *reinterpret_cast<const bf16x8*>(Q + po), pl = *reinterpret_cast<const bf16x8*>(Q + TB_HQ_PLANE + po);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int vo = hq_off(16 * dt + fr, 32 * ks + 8 * g);
            const bf16x8 vh = *reinterpret_cast<const bf16x8*>(VT + vo), vl = *reinterpret_cast<const bf16x8*>(VT + TB_HQ_PLANE + vo);
            c[dt] = mfma_bf16_16x16x32(vl, ph, c[dt]);
            c[dt] = mfma_bf16_16x16x32(vh, pl, c[dt]);
            c[dt] = mfma_bf16_16x16x32(vh, ph, c[dt]);
        }
    }
}

// row pass of the two post-LN sublayer tails: v = resid + dropout(x), y = LN(v) [, y2 = LN(y)].  Wave w owns rows 8 w .. 8 w + 7; a lane owns the 4
// columns of ONE Philox block of the flat (B, 64, 250) index: c0 = 4 lane - (row odd ? 2 : 0) (row starts are 2 mod 4 for odd rows).
// x comes from `xs` (an XF-layout LDS image), resid from `resid_lds` (XF layout) or `resid_g` (global rows of 250)
template <bool TRAIN, bool DOUBLE>
__device__ __forceinline__ void tb_ln_rows(const tb_fwd_args& a, int b, int w, int lane, const unsigned char* xs, const unsigned char* resid_lds,
                                           const float* resid_g, unsigned site, float* r_out, const float* g1, const float* be1, float* y_out, float* mu_out,
                                           float* rs_out, const float* g2, const float* be2, float* y2_out, float* mu2_out, float* rs2_out,
                                           unsigned char* y_lds /* XF layout or null */, unsigned char* ap /* planes of y or null */) {
    const long long sb = (long long)b * (TB_L * TB_D);            // uniform: this sample's offset in every (B, 64, 250) tensor

    const float ksc = (TRAIN && a.drop_p > 0.f) ? 1.f / (1.f - a.drop_p) : 1.f;
    const float inv = 1.0f / (float)TB_D;
    // EVERY global load of the pass is issued here, before the first store: vmcnt retires in order and counts stores, so a load behind a store
    // waits for that store's HBM round trip -- with the affine parameters fetched inside the row loop each of the 8 rows paid one (~2 us: the two
    // row passes were 19 + 20 us of a 147 us launch; per-phase s_memtime stamps, EEGCLIP_TB_DEBUG=2)
    tb_f2 pg1[2][2], pb1[2][2], pg2[2][2], pb2[2][2];            // [row parity][pair]
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int cp = 4 * lane - 2 * par + 2 * p;
            const bool okc = cp >= 0 && cp < TB_D;
            pg1[par][p] = okc ? *reinterpret_cast<const tb_f2*>(g1 + cp) : tb_f2{0.f, 0.f};
            pb1[par][p] = okc ? *reinterpret_cast<const tb_f2*>(be1 + cp) : tb_f2{0.f, 0.f};
            if (DOUBLE) {
                pg2[par][p] = okc ? *reinterpret_cast<const tb_f2*>(g2 + cp) : tb_f2{0.f, 0.f};
                pb2[par][p] = okc ? *reinterpret_cast<const tb_f2*>(be2 + cp) : tb_f2{0.f, 0.f};
            }
        }
    tb_f2 rg[8][2];
    if (resid_g) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int r = 8 * w + rr, c0 = 4 * lane - ((r & 1) ? 2 : 0);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int cp = c0 + 2 * p;
                rg[rr][p] = (cp >= 0 && cp < TB_D) ? *reinterpret_cast<const tb_f2*>(resid_g + sb + (unsigned)(r * TB_D + cp)) : tb_f2{0.f, 0.f};
            }
        }
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int r = 8 * w + rr, c0 = 4 * lane - ((r & 1) ? 2 : 0);
        const int rbase = r * TB_D;                                 // within the sample: every global access below is (uniform sample base) + 32-bit offset
        float v[4];
        bool ok[2];
        bool keep[4] = {true, true, true, true};
        if (TRAIN && a.drop_p > 0.f && c0 < TB_D) dropout_keep4(a.seed, site, (unsigned long long)(sb + (rbase + c0)), a.drop_p, keep);
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int cp = c0 + 2 * p;
            ok[p] = cp >= 0 && cp < TB_D;
            tb_f2 xv = tb_f2{0.f, 0.f}, rv = tb_f2{0.f, 0.f};
            if (ok[p]) {
                xv = *reinterpret_cast<const tb_f2*>(xs + xf_off(r, cp));
                rv = resid_g ? rg[rr][p] : *reinterpret_cast<const tb_f2*>(resid_lds + xf_off(r, cp));
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float d = xv[e];
                if (TRAIN && a.drop_p > 0.f) d = keep[2 * p + e] ? d * ksc : 0.f;
                v[2 * p + e] = ok[p] ? d + rv[e] : 0.f;
                s += v[2 * p + e];
            }
            if (ok[p] && r_out && !(a.dbg & 1u)) *reinterpret_cast<tb_f2*>(r_out + sb + (unsigned)(rbase + cp)) = tb_f2{v[2 * p], v[2 * p + 1]};
        }
        const float mean = wave_sum(s) * inv;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float dl = ok[e >> 1] ? v[e] - mean : 0.f;
            q += dl * dl;
        }
        const float rstd = rsqrtf(wave_sum(q) * inv + a.eps);
        float y[4];
        float s2 = 0.f;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int cp = c0 + 2 * p;
            const tb_f2 gg = pg1[rr & 1][p], bb = pb1[rr & 1][p];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                y[2 * p + e] = ok[p] ? (v[2 * p + e] - mean) * rstd * gg[e] + bb[e] : 0.f;
                s2 += y[2 * p + e];
            }
            if (ok[p]) {
                if (y_out && !(a.dbg & 1u)) *reinterpret_cast<tb_f2*>(y_out + sb + (unsigned)(rbase + cp)) = tb_f2{y[2 * p], y[2 * p + 1]};
                if (y_lds) *reinterpret_cast<tb_f2*>(y_lds + xf_off(r, cp)) = tb_f2{y[2 * p], y[2 * p + 1]};
            }
            if (ap && cp >= 0 && cp < 256) tb_store_planes2(ap, TB_AP_PLANE, ap_off(r, cp), y[2 * p], y[2 * p + 1]);      // (zeros past column 249)
        }
        if (ap && (r & 1) && lane == 63) tb_store_planes2(ap, TB_AP_PLANE, ap_off(r, 254), 0.f, 0.f);                       // odd rows: 254, 255 belong to no lane
        if (lane == 0) {
            (mu_out + (long long)b * TB_L)[(unsigned)r] = mean;
            (rs_out + (long long)b * TB_L)[(unsigned)r] = rstd;
        }
        if (DOUBLE) {
            const float mean2 = wave_sum(s2) * inv;
            float q2 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dl = ok[e >> 1] ? y[e] - mean2 : 0.f;
                q2 += dl * dl;
            }
            const float rstd2 = rsqrtf(wave_sum(q2) * inv + a.eps);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int cp = c0 + 2 * p;
                if (ok[p]) {
                    const tb_f2 gg = pg2[rr & 1][p], bb = pb2[rr & 1][p];
                    *reinterpret_cast<tb_f2*>(y2_out + sb + (unsigned)(rbase + cp)) =
                        tb_f2{(y[2 * p] - mean2) * rstd2 * gg[0] + bb[0], (y[2 * p + 1] - mean2) * rstd2 * gg[1] + bb[1]};
                }
            }
            if (lane == 0) {
                (mu2_out + (long long)b * TB_L)[(unsigned)r] = mean2;
                (rs2_out + (long long)b * TB_L)[(unsigned)r] = rstd2;
            }
        }
    }
}

// epilogue of the three 64 x 256 Linears whose tiles are all transposed (NT = 2: wave w owns n = 32 w .. 32 w + 31): calls f(m, n0, valid, acc4)
// for every (row m, 4 consecutive columns n0..) group this lane holds; valid = how many of the 4 columns exist (N = 250: 2 at n0 = 248)
template <class F>
__device__ __forceinline__ void tb_for_tiles(int w, int lane, int N, f32x4 (&acc)[4][2], F&& f) {
    const int fr = lane & 15, g = lane >> 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n0 = 32 * w + 16 * j + 4 * g;
        if (n0 >= N) continue;
        const int valid = N - n0 >= 4 ? 4 : N - n0;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) f(16 * mt + fr, n0, valid, acc[mt][j]);
    }
}

template <bool TRAIN>
__global__ __launch_bounds__(TB_THREADS, 2) void token_block_fwd_kernel(const tb_fwd_args a0) {
    EEG_LDS_BASE(unsigned char, lds);
    unsigned char* const AP = lds;
    unsigned char* const XF = lds + TB_AP_BYTES;
    const int t = threadIdx.x, lane = t & 63, w = wave_uniform(t >> 6), fr = lane & 15, g = lane >> 4;
    const int b = blockIdx.x;
    const float ksc = (TRAIN && a0.drop_p > 0.f) ? 1.f / (1.f - a0.drop_p) : 1.f;

    tb_stamp(a0, b, t, 0);
    const tb_fwd_args p0 = tb_args_again(a0);                 // this phase's arguments, re-read from the kernel-argument segment (see tb_args_again)
    // ---- S0: the EEG sample (63 x 250 fp32) -> A planes, token row 1 + channel; row 0 and k >= 250 are zero
    {
        const float* xb = p0.x + (long long)b * (TB_NCH * TB_T);
        tb_f2 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int u = t + TB_THREADS * j;
            v[j] = u < TB_NCH * TB_T / 2 ? *reinterpret_cast<const tb_f2*>(xb + 2 * u) : tb_f2{0.f, 0.f};
        }
        if (t < 64) *reinterpret_cast<tb_u4*>(AP + (t >> 5) * TB_AP_PLANE + (t & 31) * 16) = tb_u4{0u, 0u, 0u, 0u};      // row 0 of both planes
        else if (t < 192) {
            const int row = (t - 64) & 63, plane = (t - 64) >> 6;
#pragma unroll
            for (int k = 250; k < 256; k += 2) *reinterpret_cast<unsigned*>(AP + plane * TB_AP_PLANE + ap_off(row, k)) = 0u;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int u = t + TB_THREADS * j;
            if (u < TB_NCH * TB_T / 2) {
                const int flat = 2 * u, c = flat / TB_T, k = flat - TB_T * c;
                tb_store_planes2(AP, TB_AP_PLANE, ap_off(c + 1, k), v[j][0], v[j][1]);
            }
        }
    }
    raw_barrier();

    tb_stamp(a0, b, t, 1);
    const tb_fwd_args p1 = tb_args_again(a0);                 // this phase's arguments, re-read from the kernel-argument segment (see tb_args_again)
    // ---- S1: value embedding + bias + positional embedding (row = channel), subject token in row 0      (Embed.py:146-160)
    {
        f32x4 acc[4][2];
        // (joint-subject model, Embed.py:142-144: the sample's subject picks the Linear -- a workgroup-uniform base, nothing else changes)
        const int vs = p1.embed_subject ? p1.embed_subject[b] : 0;
        const unsigned short* const wv = p1.embed_subject ? p1.packed_embed + (long long)vs * TB_MAT_ELEMS : p1.packed + TB_OFF_V;
        const float* const bvp = p1.bv + (long long)vs * p1.bv_stride;
        tb_gemm<2, 3u, 2, true, TB_FWD_RING>(AP, wv + (long long)(2 * w) * TB_KS * TB_TILE, lane, acc);
        const long long id = p1.ids ? p1.ids[b] : 0;
        tb_for_tiles(w, lane, TB_D, acc, [&](int m, int n0, int valid, f32x4& v) {
            f32x4 o;
            if (m == 0) o = tb_ld4(p1.tokens + id * TB_D + n0, valid);
            else {
                const f32x4 bias = tb_ld4(bvp + n0, valid), pe = tb_ld4(p1.pe + (long long)(m - 1) * TB_D + n0, valid);
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (v[i] + bias[i]) + pe[i];
            }
            if (valid < 4) { o[2] = 0.f; o[3] = 0.f; }
            *reinterpret_cast<f32x4*>(XF + xf_off(m, n0)) = o;
        });
        // the EEG sample as token planes (row 0 zero: the subject token has no EEG row) = the X operand of the value embedding's weight gradient
        if (p1.xp && !(p1.dbg & 1u)) tb_planes_out<1>(AP, p1.xp + (long long)b * (2 * TB_AP_PLANE), t);
    }
    raw_barrier();

    tb_stamp(a0, b, t, 2);
    const tb_fwd_args p2 = tb_args_again(a0);                 // this phase's arguments, re-read from the kernel-argument segment (see tb_args_again)
    // ---- S2: embedding dropout over the flat (64 x 250) sample (one Philox block = 4 consecutive flat elements); h -> HBM and -> A planes
    {
        float* hb = p2.h + (long long)b * (TB_L * TB_D);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = t + TB_THREADS * j;
            if (q < TB_L * TB_D / 4) {
                const int i0 = 4 * q, row = i0 / TB_D, col = i0 - TB_D * row;
                const bool wrap = col + 2 >= TB_D;                     // the second pair starts the next row
                const int row2 = wrap ? row + 1 : row, col2 = wrap ? col + 2 - TB_D : col + 2;
                const tb_f2 p0 = *reinterpret_cast<const tb_f2*>(XF + xf_off(row, col)), p1 = *reinterpret_cast<const tb_f2*>(XF + xf_off(row2, col2));
                float v[4] = {p0[0], p0[1], p1[0], p1[1]};
                if (TRAIN && p2.drop_p > 0.f) {
                    bool keep[4];
                    dropout_keep4(p2.seed, p2.site_embed, (unsigned long long)b * (TB_L * TB_D) + i0, p2.drop_p, keep);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = keep[e] ? v[e] * ksc : 0.f;
                }
                if (!(p2.dbg & 1u)) *reinterpret_cast<f32x4*>(hb + i0) = f32x4{v[0], v[1], v[2], v[3]};
                tb_store_planes2(AP, TB_AP_PLANE, ap_off(row, col), v[0], v[1]);
                tb_store_planes2(AP, TB_AP_PLANE, ap_off(row2, col2), v[2], v[3]);
            }
        }
    }
    raw_barrier();

    tb_stamp(a0, b, t, 3);
    const tb_fwd_args p3 = tb_args_again(a0);                 // this phase's arguments, re-read from the kernel-argument segment (see tb_args_again)
    // ---- S3: q | k | v (SelfAttention_Family.py:199-207) and attention, two heads at a time: 24 tiles of the pair's q | k | v -> their planes
    //          behind the (still live) h planes, then waves 0-3 run one head and waves 4-7 the other
    f32x4 ctxr[2][4];
    {
        const int hip = w >> 2;                                      // head within the pair
        unsigned char* const HQ = lds + TB_AP_BYTES + hip * TB_HQ_BYTES;
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            f32x4 cur[4][3];
            switch (w & 3) {
                case 0: tb_qkv_pass<0>(p3, AP, b, 2 * rd + hip, lane, cur); tb_qkv_to_lds<0>(HQ, lane, cur); break;
                case 1: tb_qkv_pass<1>(p3, AP, b, 2 * rd + hip, lane, cur); tb_qkv_to_lds<1>(HQ, lane, cur); break;
                case 2: tb_qkv_pass<2>(p3, AP, b, 2 * rd + hip, lane, cur); tb_qkv_to_lds<2>(HQ, lane, cur); break;
                default: tb_qkv_pass<3>(p3, AP, b, 2 * rd + hip, lane, cur); tb_qkv_to_lds<3>(HQ, lane, cur); break;
            }
            if (rd == 1 && !(p3.dbg & 1u)) tb_planes_out<0>(AP, p3.hp + (long long)b * (2 * TB_AP_PLANE), t);      // (the last reader of the h planes has its weights)
            raw_barrier();
            tb_stamp(a0, b, t, 10 + 2 * rd);                         // (diagnosis: q | k | v of the head pair done)
            tb_attention<TRAIN>(p3, HQ, b, 2 * rd + hip, w & 3, lane, ctxr[rd]);
            raw_barrier();                                           // the pair's planes are dead (and, second round, the h planes too)
            tb_stamp(a0, b, t, 11 + 2 * rd);                         // (diagnosis: attention of the head pair done)
        }
        // context -> A planes with 64 columns per head (dims 62, 63 are exact zeros); S4 copies the planes to HBM
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int head = 2 * rd + hip, m = 16 * (w & 3) + fr;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int d0 = 16 * dt + 4 * g;
                const f32x4 c = ctxr[rd][dt];
                tb_store_planes4(AP, TB_AP_PLANE, ap_off(m, 64 * head + d0), c[0], c[1], c[2], c[3]);
            }
        }
    }
    raw_barrier();

    tb_stamp(a0, b, t, 4);
    const tb_fwd_args p4 = tb_args_again(a0);                 // this phase's arguments, re-read from the kernel-argument segment (see tb_args_again)
    // ---- S4: output projection + bias -> XF      (SelfAttention_Family.py:213)
    {
        f32x4 acc[4][2];
        tb_gemm<2, 3u, 2, true, TB_FWD_RING>(AP, p4.packed + TB_OFF_O + (long long)(2 * w) * TB_KS * TB_TILE, lane, acc);
        tb_for_tiles(w, lane, TB_D, acc, [&](int m, int n0, int valid, f32x4& v) {
            const f32x4 bias = tb_ld4(p4.bo + n0, valid);
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = i < valid ? v[i] + bias[i] : 0.f;
            *reinterpret_cast<f32x4*>(XF + xf_off(m, n0)) = o;
        });
        if (!(p4.dbg & 1u)) tb_planes_out<0>(AP, p4.ctxp + (long long)b * (2 * TB_AP_PLANE), t);      // ctx planes (channel 255 = dim 63 of head 3: the ones column)
    }
    __syncthreads();                                                 // (the one barrier that also waits for global stores: S5 re-reads h)
    tb_stamp(a0, b, t, 5);
    const tb_fwd_args p5 = tb_args_again(a0);                 // this phase's arguments, re-read from the kernel-argument segment (see tb_args_again)
    // ---- S5: r1 = h + dropout(attention output), n1 = LayerNorm1(r1); n1 stays in XF (FFN residual) and goes to the A planes
    tb_ln_rows<TRAIN, false>(p5, b, w, lane, XF, nullptr, p5.h, p5.site_attn_out, p5.r1, p5.ln1_g, p5.ln1_b, nullptr, p5.mu1, p5.rs1, nullptr, nullptr, nullptr, nullptr,
                             nullptr, XF, AP);
    raw_barrier();

    tb_stamp(a0, b, t, 6);
    const tb_fwd_args p6 = tb_args_again(a0);                 // this phase's arguments, re-read from the kernel-argument segment (see tb_args_again)
    // ---- S6: FFN 1 + bias -> f1 (pre-activation, kept for the backward), g1 = dropout(gelu(f1))      (Transformer_EncDec.py:48)
    {
        f32x4 acc[4][2];
        tb_gemm<2, 3u, 2, true, TB_FWD_RING>(AP, p6.packed + TB_OFF_1 + (long long)(2 * w) * TB_KS * TB_TILE, lane, acc);
        f32x4 bias1[2];                                              // (loads before the first store: see tb_ln_rows)
#pragma unroll
        for (int j = 0; j < 2; ++j) bias1[j] = *reinterpret_cast<const f32x4*>(p6.b1 + 32 * w + 16 * j + 4 * g);
        tb_for_tiles(w, lane, TB_FF, acc, [&](int m, int n0, int valid, f32x4& v) {
            const f32x4 bias = bias1[(n0 >> 4) & 1];
            const long long ob = (long long)b * (TB_L * TB_FF);          // uniform: the stores take it as a scalar base + a 32-bit lane offset
            const unsigned ol = (unsigned)(m * TB_FF + n0);
            f32x4 f, gq;
#pragma unroll
            for (int i = 0; i < 4; ++i) f[i] = v[i] + bias[i];
            if (!(p6.dbg & 1u)) *reinterpret_cast<f32x4*>(p6.f1 + ob + ol) = f;
            bool keep[4] = {true, true, true, true};
            if (TRAIN && p6.drop_p > 0.f) dropout_keep4(p6.seed, p6.site_ffn_act, (unsigned long long)ob + ol, p6.drop_p, keep);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float ge = gelu_erf(f[i]);
                gq[i] = (TRAIN && p6.drop_p > 0.f) ? (keep[i] ? ge * ksc : 0.f) : ge;
            }
            v = gq;
        });
        if (!(p6.dbg & 1u)) tb_planes_out<0>(AP, p6.n1p + (long long)b * (2 * TB_AP_PLANE), t);
        raw_barrier();                                               // every wave is done with the n1 planes
        tb_for_tiles(w, lane, TB_FF, acc, [&](int m, int n0, int valid, f32x4& v) {
            tb_store_planes4(AP, TB_AP_PLANE, ap_off(m, n0), v[0], v[1], v[2], v[3]);
        });
    }
    raw_barrier();

    tb_stamp(a0, b, t, 7);
    const tb_fwd_args p7 = tb_args_again(a0);                 // this phase's arguments, re-read from the kernel-argument segment (see tb_args_again)
    // ---- S7: FFN 2 + bias; the result replaces the (dead) g1 planes as an fp32 image      (Transformer_EncDec.py:49)
    {
        f32x4 acc[4][2];
        tb_gemm<2, 3u, 2, true, TB_FWD_RING>(AP, p7.packed + TB_OFF_2 + (long long)(2 * w) * TB_KS * TB_TILE, lane, acc);
        f32x4 bias2[2];                                              // (loads before the stores below)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n0 = 32 * w + 16 * j + 4 * g;
            bias2[j] = n0 < TB_D ? tb_ld4(p7.b2 + n0, TB_D - n0 >= 4 ? 4 : 2) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (!(p7.dbg & 1u)) tb_planes_out<64>(AP, p7.g1p + (long long)b * (2 * TB_AP_PLANE), t);      // g1 planes: 256 real channels, no ones column
        raw_barrier();
#pragma unroll
        for (int j = 0; j < 2; ++j) {                                // (tb_for_tiles with the tile index j a compile-time constant: bias2[j] stays in registers)
            const int n0 = 32 * w + 16 * j + 4 * g;
            if (n0 >= TB_D) continue;
            const int valid = TB_D - n0 >= 4 ? 4 : TB_D - n0;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = i < valid ? acc[mt][j][i] + bias2[j][i] : 0.f;
                *reinterpret_cast<f32x4*>(AP + xf_off(16 * mt + fr, n0)) = o;
            }
        }
    }
    raw_barrier();
    tb_stamp(a0, b, t, 8);
    const tb_fwd_args p8 = tb_args_again(a0);                 // this phase's arguments, re-read from the kernel-argument segment (see tb_args_again)
    // ---- S8: r2 = n1 + dropout(FFN output), n2 = LayerNorm2(r2), n3 = final LayerNorm(n2)      (Transformer_EncDec.py:51,77-78)
    tb_ln_rows<TRAIN, true>(p8, b, w, lane, AP, XF, nullptr, p8.site_ffn_out, p8.r2, p8.ln2_g, p8.ln2_b, p8.n2, p8.mu2, p8.rs2, p8.ln3_g, p8.ln3_b, p8.n3, p8.mu3, p8.rs3,
                            nullptr, nullptr);
    tb_stamp(a0, b, t, 9);
    // ---- S9 (training plans): the conv stack's BatchNorm1 batch sums of this sample.  The statistics need y1 of EVERY sample before cstack_fwd can normalise
    // one, so they were a launch of their own (cstack_stats1: 18 - 20 us, 16 MB of token rows read back from HBM); here the workgroup goes on with its own
    // sample while the rows are in L2 and the chip is its anyway.  The LDS images of the block are dead: the packed rows overlay them.
    if (a0.cs_rows) {
        __syncthreads();                                         // (vmcnt(0) + barrier: every wave's n3 rows are in L2, every LDS read of the row pass is done)
        const tb_fwd_args p9 = tb_args_again(a0);
        cs_stats1_sample(lds, p9.n3, (long long)(TB_L * TB_D), (long long)TB_D, p9.cs_w25, p9.cs_bias, p9.cs_rows, b, p9.cs_H, 1);
    }
}


// =====================================================================================================================================
// Backward of the block's dX chain, one workgroup per sample, in two launches around the (unchanged) attention backward:
//   A  final LayerNorm', LayerNorm2' (+ FFN-output dropout'), dg1 = df2 W2 with dropout' gelu', dn1 = dr2 + df1 W1, LayerNorm1' (+ attention-output
//      dropout'), dctx = da1 Wo                                                   (Transformer_EncDec.py:45-51,77-78 read backwards)
//   B  dh = dropout'_embed(dr1 + dq Wq + dk Wk + dv Wv)                            (SelfAttention_Family.py:199-207, Embed.py:162)
// The weight gradients stay GEMMs over the whole batch on the second stream: they read what these kernels leave in HBM (df2, dg1 = df1, da1,
// dqkv, dr1).  LayerNorm gamma / beta gradients leave the workgroup as ONE partial row per sample ([6][256]: g3 b3 g2 b2 g1 b1);
// token_block_param_reduce_kernel sums the rows.
struct tb_bwd_args {
    const unsigned short* packed;
    const float *dn3, *n2, *r2, *r1, *f1, *mu1, *rs1, *mu2, *rs2, *mu3, *rs3, *ln1_g, *ln2_g, *ln2_b, *ln3_g;
    float *dr1, *dctx, *partials;
    unsigned char *df2p, *dg1p, *da1p, *dr1p;                    // token planes: the dY operands of the weight-gradient GEMMs (dr1p may be null)
    const unsigned char* dqkvp;                                  // dq | dk | dv token planes (eegclip_attention_bwd_x3), B * 64 KB apart
    float drop_p;
    unsigned long long seed;
    unsigned site_embed, site_attn_out, site_ffn_act, site_ffn_out;
};

// LayerNorm backward of one row slice held by a lane (4 columns, ok[p] per pair): dx = rs (g - mean(g) - xh mean(g xh)), g = dy gamma
__device__ __forceinline__ void tb_ln_bwd_row(const float (&dy)[4], const float (&x)[4], const float (&gam)[4], const bool (&ok)[2], float mu, float rs,
                                              float (&dx)[4], float (&pg)[4], float (&pb)[4]) {
    float xh[4], gq[4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        xh[e] = ok[e >> 1] ? (x[e] - mu) * rs : 0.f;
        gq[e] = dy[e] * gam[e];
        s1 += gq[e];
        s2 += gq[e] * xh[e];
        pg[e] += dy[e] * xh[e];
        pb[e] += dy[e];
    }
    const float c1 = wave_sum(s1) * (1.0f / (float)TB_D), c2 = wave_sum(s2) * (1.0f / (float)TB_D);
#pragma unroll
    for (int e = 0; e < 4; ++e) dx[e] = ok[e >> 1] ? rs * (gq[e] - c1 - xh[e] * c2) : 0.f;
}

// the per-lane partial sums of NV parameter-gradient vectors (even rows own columns 4 lane.., odd rows 4 lane - 2..) -> one row per vector and
// sample in `out` ([NV][256]); `slab` = 8 x NV x 256 floats of LDS scratch
template <int NV>
__device__ __forceinline__ void tb_reduce_partials(float* slab, int w, int lane, int t, const float (&pe)[NV][4], const float (&po)[NV][4], float* out) {
    float* mine = slab + w * (NV * 256);
#pragma unroll
    for (int v = 0; v < NV; ++v) *reinterpret_cast<f32x4*>(mine + v * 256 + 4 * lane) = f32x4{pe[v][0], pe[v][1], pe[v][2], pe[v][3]};
    wave_sync();
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * lane - 2 + e;
            if (c >= 0) mine[v * 256 + c] += po[v][e];              // (a lane's four odd-row columns belong to no other lane's odd-row set)
        }
    raw_barrier();
    for (int i = t; i < NV * 256; i += TB_THREADS) {
        float sum = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) sum += slab[ww * (NV * 256) + i];
        out[i] = sum;
    }
}

template <bool TRAIN>
__global__ __launch_bounds__(TB_THREADS, 2) void token_block_bwd_a_kernel(const tb_bwd_args a0) {
    EEG_LDS_BASE(unsigned char, lds);
    unsigned char* const AP = lds;
    unsigned char* const XF = lds + TB_AP_BYTES;
    float* const SC = reinterpret_cast<float*>(lds + TB_AP_BYTES + TB_XF_BYTES);       // 32 KB of scratch for the parameter-gradient rows
    const int t = threadIdx.x, lane = t & 63, w = wave_uniform(t >> 6);
    const int b = blockIdx.x;
    const float ksc = (TRAIN && a0.drop_p > 0.f) ? 1.f / (1.f - a0.drop_p) : 1.f;
    float* const part = a0.partials + (long long)b * (6 * 256);

    const tb_bwd_args q1 = tb_args_again(a0);                 // (see tb_args_again)
    // ---- T1: final LayerNorm' and LayerNorm2' chained per row in registers; df2 = dropout'(dr2) -> A planes (T2 copies them to HBM), dr2 -> XF
    {
        float pe[4][4], po[4][4];                                    // [g3 b3 g2 b2] x 4 columns, even / odd rows
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) { pe[v][e] = 0.f; po[v][e] = 0.f; }
        // every global load of the pass before its first store (a load behind a store waits for the store's round trip: vmcnt is in order)
        tb_f2 pg3[2][2], pg2[2][2], pb2[2][2];                       // LayerNorm parameters, [row parity][pair]
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int cp = 4 * lane - 2 * par + 2 * p;
                const bool okc = cp >= 0 && cp < TB_D;
                pg3[par][p] = okc ? *reinterpret_cast<const tb_f2*>(q1.ln3_g + cp) : tb_f2{0.f, 0.f};
                pg2[par][p] = okc ? *reinterpret_cast<const tb_f2*>(q1.ln2_g + cp) : tb_f2{0.f, 0.f};
                pb2[par][p] = (okc && !q1.n2) ? *reinterpret_cast<const tb_f2*>(q1.ln2_b + cp) : tb_f2{0.f, 0.f};
            }
        tb_f2 in_dy[8][2], in_r2[8][2], in_n2[8][2];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int cp = 4 * lane - ((rr & 1) ? 2 : 0) + 2 * p;
                const long long sb = (long long)b * (TB_L * TB_D);           // uniform sample base + a 32-bit lane offset (see tb_ln_rows)
                const unsigned o = (unsigned)((8 * w + rr) * TB_D + cp);
                const bool okc = cp >= 0 && cp < TB_D;
                in_dy[rr][p] = okc ? *reinterpret_cast<const tb_f2*>(q1.dn3 + sb + o) : tb_f2{0.f, 0.f};
                in_r2[rr][p] = okc ? *reinterpret_cast<const tb_f2*>(q1.r2 + sb + o) : tb_f2{0.f, 0.f};
                in_n2[rr][p] = (okc && q1.n2) ? *reinterpret_cast<const tb_f2*>(q1.n2 + sb + o) : tb_f2{0.f, 0.f};
            }
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int r = 8 * w + rr, c0 = 4 * lane - ((rr & 1) ? 2 : 0);
            const long long row = (long long)b * TB_L + r, rbase = row * TB_D;
            float dy[4], x3[4], x2[4], g3[4], g2[4];
            bool ok[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int cp = c0 + 2 * p;
                ok[p] = cp >= 0 && cp < TB_D;
                const tb_f2 v0 = in_dy[rr][p], v2 = in_r2[rr][p], v3 = pg3[rr & 1][p], v4 = pg2[rr & 1][p];
                tb_f2 v1 = in_n2[rr][p];
                if (ok[p] && !q1.n2) {
                    // n2 = LayerNorm2(r2) is re-evaluated from r2 and the row statistics (the forward did not store it: 16 MB per step less each way)
                    const tb_f2 bb = pb2[rr & 1][p];
                    const float mu = q1.mu2[row], rs = q1.rs2[row];
                    v1 = tb_f2{(v2[0] - mu) * rs * v4[0] + bb[0], (v2[1] - mu) * rs * v4[1] + bb[1]};
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) { dy[2 * p + e] = v0[e]; x3[2 * p + e] = v1[e]; x2[2 * p + e] = v2[e]; g3[2 * p + e] = v3[e]; g2[2 * p + e] = v4[e]; }
            }
            float dn2[4], dr2[4];
            if (rr & 1) {
                tb_ln_bwd_row(dy, x3, g3, ok, q1.mu3[row], q1.rs3[row], dn2, po[0], po[1]);
                tb_ln_bwd_row(dn2, x2, g2, ok, q1.mu2[row], q1.rs2[row], dr2, po[2], po[3]);
            } else {
                tb_ln_bwd_row(dy, x3, g3, ok, q1.mu3[row], q1.rs3[row], dn2, pe[0], pe[1]);
                tb_ln_bwd_row(dn2, x2, g2, ok, q1.mu2[row], q1.rs2[row], dr2, pe[2], pe[3]);
            }
            bool keep[4] = {true, true, true, true};
            if (TRAIN && q1.drop_p > 0.f && c0 < TB_D) dropout_keep4(q1.seed, q1.site_ffn_out, (unsigned long long)(rbase + c0), q1.drop_p, keep);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int cp = c0 + 2 * p;
                const float d0 = (TRAIN && q1.drop_p > 0.f) ? (keep[2 * p] ? dr2[2 * p] * ksc : 0.f) : dr2[2 * p];
                const float d1 = (TRAIN && q1.drop_p > 0.f) ? (keep[2 * p + 1] ? dr2[2 * p + 1] * ksc : 0.f) : dr2[2 * p + 1];
                if (ok[p]) *reinterpret_cast<tb_f2*>(XF + xf_off(r, cp)) = tb_f2{dr2[2 * p], dr2[2 * p + 1]};
                if (cp >= 0 && cp < 256) tb_store_planes2(AP, TB_AP_PLANE, ap_off(r, cp), ok[p] ? d0 : 0.f, ok[p] ? d1 : 0.f);
            }
            if ((rr & 1) && lane == 63) tb_store_planes2(AP, TB_AP_PLANE, ap_off(r, 254), 0.f, 0.f);
        }
        tb_reduce_partials<4>(SC, w, lane, t, pe, po, part);
    }
    raw_barrier();

    const tb_bwd_args q2 = tb_args_again(a0);                 // (see tb_args_again)
    // ---- T2: dg1 = df2 W2, then the FFN activation's dropout' and gelu' (pre-activation f1 from HBM): df1 -> A planes (T3 copies them to HBM)
    {
        f32x4 acc[4][2];
        tb_gemm<2, 3u, TB_BWD_PF, true, true>(AP, q2.packed + TB_OFF_2T + (long long)(2 * w) * TB_KS * TB_TILE, lane, acc);
        f32x4 fpre[4][2];                                            // the pre-activations of this lane's 8 groups: loaded before the first store
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                fpre[mt][j] = *reinterpret_cast<const f32x4*>(q2.f1 + (long long)b * (TB_L * TB_FF) + (unsigned)((16 * mt + (lane & 15)) * TB_FF + 32 * w + 16 * j + 4 * (lane >> 4)));
        tb_planes_out<64>(AP, q2.df2p + (long long)b * (2 * TB_AP_PLANE), t);      // df2 planes (bias gradient: all-ones fragment in the GEMM, g1 has 256 channels)
        tb_for_tiles(w, lane, TB_FF, acc, [&](int m, int n0, int valid, f32x4& v) {
            const long long ob = (long long)b * (TB_L * TB_FF);
            const unsigned ol = (unsigned)(m * TB_FF + n0);
            const f32x4 f = fpre[m >> 4][(n0 >> 4) & 1];
            bool keep[4] = {true, true, true, true};
            if (TRAIN && q2.drop_p > 0.f) dropout_keep4(q2.seed, q2.site_ffn_act, (unsigned long long)ob + ol, q2.drop_p, keep);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = (TRAIN && q2.drop_p > 0.f) ? (keep[i] ? v[i] * ksc : 0.f) : v[i];
                v[i] = d * gelu_erf_grad(f[i]);
            }
        });
        raw_barrier();                                               // every wave is done with the df2 planes
        tb_for_tiles(w, lane, TB_FF, acc, [&](int m, int n0, int valid, f32x4& v) {
            tb_store_planes4(AP, TB_AP_PLANE, ap_off(m, n0), v[0], v[1], v[2], v[3]);
        });
    }
    raw_barrier();

    const tb_bwd_args q3 = tb_args_again(a0);                 // (see tb_args_again)
    // ---- T3: dn1 = dr2 + df1 W1 (in place in XF)
    {
        f32x4 acc[4][2];
        tb_gemm<2, 3u, TB_BWD_PF, true, true>(AP, q3.packed + TB_OFF_1T + (long long)(2 * w) * TB_KS * TB_TILE, lane, acc);
        tb_for_tiles(w, lane, TB_D, acc, [&](int m, int n0, int valid, f32x4& v) {
            f32x4* p = reinterpret_cast<f32x4*>(XF + xf_off(m, n0));
            f32x4 o = *p;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = i < valid ? o[i] + v[i] : 0.f;
            *p = o;
        });
        tb_planes_out<64>(AP, q3.dg1p + (long long)b * (2 * TB_AP_PLANE), t);      // dg1 = df1 planes
    }
    raw_barrier();

    const tb_bwd_args q4 = tb_args_again(a0);                 // (see tb_args_again)
    // ---- T4: LayerNorm1': dr1 -> HBM (the residual path into dh), da1 = dropout'(dr1) -> A planes (T5 copies them to HBM)
    {
        float pe[2][4], po[2][4];
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) { pe[v][e] = 0.f; po[v][e] = 0.f; }
        tb_f2 pg1[2][2], in_r1[8][2];                                // (loads before the first store)
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int cp = 4 * lane - 2 * par + 2 * p;
                pg1[par][p] = (cp >= 0 && cp < TB_D) ? *reinterpret_cast<const tb_f2*>(q4.ln1_g + cp) : tb_f2{0.f, 0.f};
            }
#pragma unroll
        for (int rr = 0; rr < 8; ++rr)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int cp = 4 * lane - ((rr & 1) ? 2 : 0) + 2 * p;
                in_r1[rr][p] = (cp >= 0 && cp < TB_D) ? *reinterpret_cast<const tb_f2*>(q4.r1 + (long long)b * (TB_L * TB_D) + (unsigned)((8 * w + rr) * TB_D + cp)) : tb_f2{0.f, 0.f};
            }
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int r = 8 * w + rr, c0 = 4 * lane - ((rr & 1) ? 2 : 0);
            const long long row = (long long)b * TB_L + r, rbase = row * TB_D;
            float dy[4], x1[4], g1[4];
            bool ok[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int cp = c0 + 2 * p;
                ok[p] = cp >= 0 && cp < TB_D;
                tb_f2 v0 = tb_f2{0.f, 0.f};
                const tb_f2 v1 = in_r1[rr][p], v2 = pg1[rr & 1][p];
                if (ok[p]) v0 = *reinterpret_cast<const tb_f2*>(XF + xf_off(r, cp));
#pragma unroll
                for (int e = 0; e < 2; ++e) { dy[2 * p + e] = v0[e]; x1[2 * p + e] = v1[e]; g1[2 * p + e] = v2[e]; }
            }
            float dr1[4];
            if (rr & 1) tb_ln_bwd_row(dy, x1, g1, ok, q4.mu1[row], q4.rs1[row], dr1, po[0], po[1]);
            else tb_ln_bwd_row(dy, x1, g1, ok, q4.mu1[row], q4.rs1[row], dr1, pe[0], pe[1]);
            bool keep[4] = {true, true, true, true};
            if (TRAIN && q4.drop_p > 0.f && c0 < TB_D) dropout_keep4(q4.seed, q4.site_attn_out, (unsigned long long)(rbase + c0), q4.drop_p, keep);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int cp = c0 + 2 * p;
                const float d0 = (TRAIN && q4.drop_p > 0.f) ? (keep[2 * p] ? dr1[2 * p] * ksc : 0.f) : dr1[2 * p];
                const float d1 = (TRAIN && q4.drop_p > 0.f) ? (keep[2 * p + 1] ? dr1[2 * p + 1] * ksc : 0.f) : dr1[2 * p + 1];
                if (ok[p]) {
                    *reinterpret_cast<tb_f2*>(q4.dr1 + rbase + (unsigned)cp) = tb_f2{dr1[2 * p], dr1[2 * p + 1]};
                }
                if (cp >= 0 && cp < 256) tb_store_planes2(AP, TB_AP_PLANE, ap_off(r, cp), ok[p] ? d0 : 0.f, ok[p] ? d1 : 0.f);
            }
            if ((rr & 1) && lane == 63) tb_store_planes2(AP, TB_AP_PLANE, ap_off(r, 254), 0.f, 0.f);
        }
        raw_barrier();                                               // (the scratch rows of T1 have been read by every thread)
        tb_reduce_partials<2>(SC, w, lane, t, pe, po, part + 4 * 256);
    }
    raw_barrier();

    const tb_bwd_args q5 = tb_args_again(a0);                 // (see tb_args_again)
    // ---- T5: dctx = da1 Wo -> HBM, natural (row, 248) layout (the packed operand's columns are 64 head + d)
    {
        f32x4 acc[4][2];
        tb_gemm<2, 3u, TB_BWD_PF, true, true>(AP, q5.packed + TB_OFF_OT + (long long)(2 * w) * TB_KS * TB_TILE, lane, acc);
        tb_for_tiles(w, lane, 256, acc, [&](int m, int n0, int valid, f32x4& v) {
            const int head = n0 >> 6, d0 = n0 & 63;
            if (d0 < TB_E) tb_st4(q5.dctx + (long long)b * (TB_L * TB_HE) + (unsigned)(m * TB_HE + head * TB_E + d0), TB_E - d0 >= 4 ? 4 : 2, v);
        });
        tb_planes_out<64>(AP, q5.da1p + (long long)b * (2 * TB_AP_PLANE), t);
    }
}

template <bool TRAIN>
__global__ __launch_bounds__(TB_THREADS, 2) void token_block_bwd_b_kernel(const tb_bwd_args a) {
    EEG_LDS_BASE(unsigned char, lds);
    unsigned char* const AP = lds;
    unsigned char* const XF = lds + TB_AP_BYTES;
    const int t = threadIdx.x, lane = t & 63, w = wave_uniform(t >> 6);
    const int b = blockIdx.x;
    const float ksc = (TRAIN && a.drop_p > 0.f) ? 1.f / (1.f - a.drop_p) : 1.f;
    f32x4 acc[4][2];
    // dq | dk | dv of the sample arrive as token planes = AP images (csrc/attention_x3.hip writes them; columns 64 head + 62, 63 are exact zeros): LDS-DMA
    // copies, no VGPRs and no fp32 -> hi | lo split here.  Wave w deposits rows 8 w .. 8 w + 7 of both planes (1 KB = 2 rows per instruction; the
    // lane picks the source chunk so that the AP swizzle comes out).  dq -> AP and dk -> the XF region at once, dv -> AP when the first GEMM is done.
    auto dma_planes = [&](int which, unsigned char* dst) {
        const unsigned char* src = a.dqkvp + ((long long)which * gridDim.x + b) * (2 * TB_AP_PLANE);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int plane = i >> 2, row = 8 * w + 2 * (i & 3) + (lane >> 5), c = lane & 31;
            lds_dma16(dst + plane * TB_AP_PLANE + (8 * w + 2 * (i & 3)) * 512, src + plane * TB_AP_PLANE + row * 512 + ((c ^ (row & 15)) << 4));
        }
    };
    dma_planes(0, AP);
    dma_planes(1, XF);
    wait_vmcnt<8>();
    raw_barrier();
    tb_gemm<2, 3u, TB_BWD_PF, true, true>(AP, a.packed + TB_OFF_QT + (long long)(2 * w) * TB_KS * TB_TILE, lane, acc);
    raw_barrier();                                                   // every wave has read the dq planes
    dma_planes(2, AP);
    wait_vmcnt<8>();
    raw_barrier();
    tb_gemm<2, 3u, TB_BWD_PF, false, true>(XF, a.packed + TB_OFF_QT + TB_MAT_ELEMS + (long long)(2 * w) * TB_KS * TB_TILE, lane, acc);
    wait_vmcnt<0>();
    raw_barrier();
    tb_gemm<2, 3u, TB_BWD_PF, false, true>(AP, a.packed + TB_OFF_QT + 2 * TB_MAT_ELEMS + (long long)(2 * w) * TB_KS * TB_TILE, lane, acc);
    raw_barrier();                                                   // (the dk planes in the XF region are dead: the epilogue writes its fp32 image there)
    // + the residual-path gradient, then the embedding dropout' over the flat sample (one Philox block = 4 consecutive flat elements)
    tb_for_tiles(w, lane, TB_D, acc, [&](int m, int n0, int valid, f32x4& v) {
        const f32x4 r = tb_ld4(a.dr1 + (long long)b * (TB_L * TB_D) + (unsigned)(m * TB_D + n0), valid);
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = i < valid ? r[i] + v[i] : 0.f;
        *reinterpret_cast<f32x4*>(XF + xf_off(m, n0)) = o;
    });
    __syncthreads();                                                 // (also: every load of the old dr1 has returned before it is overwritten)
    float* ob = a.dr1 + (long long)b * (TB_L * TB_D);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int q = t + TB_THREADS * j;
        if (q < TB_L * TB_D / 4) {
            const int i0 = 4 * q, row = i0 / TB_D, col = i0 - TB_D * row;
            const bool wrap = col + 2 >= TB_D;
            const int row2 = wrap ? row + 1 : row, col2 = wrap ? col + 2 - TB_D : col + 2;
            const tb_f2 p0 = *reinterpret_cast<const tb_f2*>(XF + xf_off(row, col)), p1 = *reinterpret_cast<const tb_f2*>(XF + xf_off(row2, col2));
            float v[4] = {p0[0], p0[1], p1[0], p1[1]};
            if (TRAIN && a.drop_p > 0.f) {
                bool keep[4];
                dropout_keep4(a.seed, a.site_embed, (unsigned long long)b * (TB_L * TB_D) + i0, a.drop_p, keep);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = keep[e] ? v[e] * ksc : 0.f;
            }
            *reinterpret_cast<f32x4*>(ob + i0) = f32x4{v[0], v[1], v[2], v[3]};
            if (a.dr1p) {                                            // dh as token planes: the dY operand of the value embedding's weight gradient
                tb_store_planes2(AP, TB_AP_PLANE, ap_off(row, col), v[0], v[1]);
                tb_store_planes2(AP, TB_AP_PLANE, ap_off(row2, col2), v[2], v[3]);
            }
        }
    }
    if (a.dr1p) {
        if (t < 128) {                                               // channels 250 .. 255 of both planes: zero
            const int row = t & 63, plane = t >> 6;
#pragma unroll
            for (int k = 250; k < 256; k += 2) *reinterpret_cast<unsigned*>(AP + plane * TB_AP_PLANE + ap_off(row, k)) = 0u;
        }
        raw_barrier();
        tb_planes_out<64>(AP, a.dr1p + (long long)b * (2 * TB_AP_PLANE), t);
    }
}

// dgamma / dbeta of the three LayerNorms: out[v][c] += sum_b partials[b][v][c]; one workgroup per (vector, 64 columns, slice of <= B/16 samples)
struct tb_param_args {
    const float* partials;
    float* out[6];
    int B;
};
__global__ __launch_bounds__(256) void token_block_param_reduce_kernel(const tb_param_args a) {
    EEG_LDS_BASE(float, red);
    const int v = blockIdx.y, lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int per = (a.B + (int)gridDim.z - 1) / (int)gridDim.z, b0 = (int)blockIdx.z * per, b1 = b0 + per < a.B ? b0 + per : a.B;
    float s = 0.f;
#pragma unroll 4
    for (int b = b0 + g; b < b1; b += 4) s += a.partials[((long long)b * 6 + v) * 256 + c];
    red[g * 64 + lane] = s;
    __syncthreads();
    if (g == 0 && c < TB_D) atomicAdd(a.out[v] + c, (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]));      // gridDim.z adds per address
}

}  // namespace eeg

using namespace eeg;

extern "C" long long eegclip_token_block_packed_bytes(void) { return TB_PACKED_ELEMS * 2; }

extern "C" int eegclip_token_block_pack(const float* wv, const float* wqkv, const float* wo, const float* w1, const float* w2, void* packed, void* stream) {
    if (!wv || !wqkv || !wo || !w1 || !w2 || !packed) return EEGCLIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(packed) & 15u) return EEGCLIP_EALIGN;
    const int tiles = (int)(TB_PACKED_ELEMS / TB_TILE);
    tb_pack_args a{{wv, wqkv, wo, w1, w2, w2, w1, wo, wqkv, wqkv, wqkv}, static_cast<unsigned short*>(packed), tiles, nullptr, nullptr, nullptr, 0};
    EEG_LAUNCH(token_block_pack_kernel, dim3(tiles), dim3(64), 0, stream, a);
    return (int)hipGetLastError();
}

// the per-step weight preparation of the encoder as ONE launch: eegclip_token_block_pack + eegclip_cstack_pack_all (Ws (40,40,H) -> cs_packed, and
// cs_packed_t unless NULL)
extern "C" int eegclip_weight_prep(const float* wv, const float* wqkv, const float* wo, const float* w1, const float* w2, void* packed, const float* Ws,
                                   void* cs_packed, void* cs_packed_t, int H, void* stream) {
    if (!wv || !wqkv || !wo || !w1 || !w2 || !packed || !Ws || !cs_packed || H < 1 || H > CS_MAXH) return EEGCLIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(cs_packed) | reinterpret_cast<uintptr_t>(cs_packed_t)) & 15u) return EEGCLIP_EALIGN;
    const int tiles = (int)(TB_PACKED_ELEMS / TB_TILE);
    tb_pack_args a{{wv, wqkv, wo, w1, w2, w2, w1, wo, wqkv, wqkv, wqkv}, static_cast<unsigned short*>(packed), tiles, Ws, static_cast<unsigned char*>(cs_packed),
                   static_cast<unsigned char*>(cs_packed_t), H};
    const int extra = (cs_pack_items(H, cs_packed_t != nullptr) + 63) / 64;
    EEG_LAUNCH(token_block_pack_kernel, dim3(tiles + extra), dim3(64), 0, stream, a);
    return (int)hipGetLastError();
}

// the value embeddings of a joint-subject model: matrix s (nn.Linear (250, 250) at w0 + s * w_stride floats) -> out + s * one packed operand
namespace eeg {
__global__ __launch_bounds__(64) void token_block_pack_embed_kernel(const float* __restrict__ w0, long long w_stride, unsigned short* __restrict__ out) {
    const int tile = blockIdx.x, s = blockIdx.y;
    const int nt = tile / TB_KS, ks = tile % TB_KS;
    const int lane = threadIdx.x & 63;
    const int n = 16 * nt + (lane & 15), k0 = 32 * ks + 8 * (lane >> 4);
    const float* src = w0 + (long long)s * w_stride;
    unsigned hb[8], lb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const long long si = tb_src_index(0, n, k0 + e);
        const float v = si >= 0 ? src[si] : 0.f;
        const unsigned short h = f32_to_bf16_bits(v);
        hb[e] = h;
        lb[e] = f32_to_bf16_bits(v - bf16_bits_to_f32(h));
    }
    unsigned short* dst = out + (long long)s * TB_MAT_ELEMS + (long long)tile * TB_TILE + lane * 8;
    tb_u4 oh, ol;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        oh[e] = hb[2 * e] | (hb[2 * e + 1] << 16);
        ol[e] = lb[2 * e] | (lb[2 * e + 1] << 16);
    }
    *reinterpret_cast<tb_u4*>(dst) = oh;
    *reinterpret_cast<tb_u4*>(dst + 512) = ol;
}
}  // namespace eeg

extern "C" long long eegclip_token_block_packed_embed_bytes(int n_subjects) { return n_subjects < 1 ? 0 : (long long)n_subjects * TB_MAT_ELEMS * 2; }

extern "C" int eegclip_token_block_pack_embed(const float* w0, long long w_stride, int n_subjects, void* out, void* stream) {
    if (!w0 || !out || n_subjects < 1 || n_subjects > 65535 || w_stride < (long long)TB_D * TB_T) return EEGCLIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(out) & 15u) return EEGCLIP_EALIGN;
    EEG_LAUNCH(token_block_pack_embed_kernel, dim3(16 * TB_KS, (unsigned)n_subjects), dim3(64), 0, stream, w0, w_stride, static_cast<unsigned short*>(out));
    return (int)hipGetLastError();
}

extern "C" int eegclip_token_block_fwd(const eegclip_token_block_desc* d, void* stream) {
    if (!d || d->B < 1 || !d->x || !d->packed || !d->bv || !d->pe || !d->tokens || !d->bqkv || !d->bo || !d->ln1_g || !d->ln1_b || !d->b1 || !d->b2 ||
        !d->ln2_g || !d->ln2_b || !d->ln3_g || !d->ln3_b || !d->h || !d->qkv || !d->ctxp || !d->r1 || !d->n1p || !d->mu1 || !d->rs1 || !d->f1 || !d->g1p ||
        !d->hp || !d->r2 || !d->mu2 || !d->rs2 || !d->n3 || !d->mu3 || !d->rs3 || d->drop_p < 0.f || d->drop_p >= 1.f)
        return EEGCLIP_EINVAL;
    const uintptr_t al16 = reinterpret_cast<uintptr_t>(d->packed) | reinterpret_cast<uintptr_t>(d->h) | reinterpret_cast<uintptr_t>(d->f1) |
                           reinterpret_cast<uintptr_t>(d->b1) | reinterpret_cast<uintptr_t>(d->xp) | reinterpret_cast<uintptr_t>(d->hp) |
                           reinterpret_cast<uintptr_t>(d->ctxp) | reinterpret_cast<uintptr_t>(d->n1p) | reinterpret_cast<uintptr_t>(d->g1p);
    const uintptr_t al8 = reinterpret_cast<uintptr_t>(d->x) | reinterpret_cast<uintptr_t>(d->qkv) |
                          reinterpret_cast<uintptr_t>(d->r1) | reinterpret_cast<uintptr_t>(d->r2) |
                          reinterpret_cast<uintptr_t>(d->n2) | reinterpret_cast<uintptr_t>(d->n3) | reinterpret_cast<uintptr_t>(d->bv) |
                          reinterpret_cast<uintptr_t>(d->pe) | reinterpret_cast<uintptr_t>(d->tokens) | reinterpret_cast<uintptr_t>(d->bqkv) |
                          reinterpret_cast<uintptr_t>(d->bo) | reinterpret_cast<uintptr_t>(d->b2) | reinterpret_cast<uintptr_t>(d->ln1_g) |
                          reinterpret_cast<uintptr_t>(d->ln1_b) | reinterpret_cast<uintptr_t>(d->ln2_g) | reinterpret_cast<uintptr_t>(d->ln2_b) |
                          reinterpret_cast<uintptr_t>(d->ln3_g) | reinterpret_cast<uintptr_t>(d->ln3_b);
    if ((al16 & 15u) || (al8 & 7u)) return EEGCLIP_EALIGN;
    tb_fwd_args a{d->x, static_cast<const unsigned short*>(d->packed), d->bv, d->pe, d->tokens, d->ids, d->bqkv, d->bo, d->ln1_g, d->ln1_b, d->b1, d->b2,
                  d->ln2_g, d->ln2_b, d->ln3_g, d->ln3_b, d->h, d->qkv, d->r1, d->mu1, d->rs1, d->f1, d->r2, d->n2, d->mu2, d->rs2,
                  d->n3, d->mu3, d->rs3, static_cast<unsigned char*>(d->xp), static_cast<unsigned char*>(d->hp), static_cast<unsigned char*>(d->ctxp),
                  static_cast<unsigned char*>(d->n1p), static_cast<unsigned char*>(d->g1p), d->drop_p, d->eps, d->scale, d->seed, d->site_embed, d->site_attn,
                  d->site_attn_out, d->site_ffn_act, d->site_ffn_out, 0u, nullptr, static_cast<const unsigned short*>(d->packed_embed), d->embed_subject,
                  d->embed_subject ? d->bv_stride : 0, d->cs_w25, d->cs_bias, d->cs_rows, d->cs_H};
    if (d->cs_rows && (!d->cs_w25 || !d->cs_bias || d->cs_H < 1 || d->cs_H > CS_MAXH || (reinterpret_cast<uintptr_t>(d->cs_rows) & 7u))) return EEGCLIP_EINVAL;
    static_assert(CS_LDS_S + CS_LDS_PS <= TB_LDS && CS_NT == TB_THREADS, "the conv stack's statistics pass runs in the block kernel's workgroup");
    if (d->embed_subject && (!d->packed_embed || (reinterpret_cast<uintptr_t>(d->packed_embed) & 15u) || d->bv_stride < 0 || (d->bv_stride & 1))) return EEGCLIP_EINVAL;
    static const unsigned dbg = getenv("EEGCLIP_TB_DEBUG") ? (unsigned)atoi(getenv("EEGCLIP_TB_DEBUG")) : 0u;
    //  bit 0 leaves out every activation / plane store -- a timing ablation.  The weight-gradient operands exist ONLY as those planes, so a stray
    // environment variable would make training read uninitialised memory: refuse the combination instead of producing silent garbage.
    if ((dbg & 1u) && (d->xp || d->hp || d->ctxp || d->n1p || d->g1p)) return EEGCLIP_EINVAL;
    a.dbg = dbg;
    a.tstamp = nullptr;
#if !defined(EEG_EMU)
    static unsigned long long* stamps = nullptr;                 // DIAGNOSIS ONLY (EEGCLIP_TB_DEBUG & 2): the one allocation and sync of the library
    static int calls = 0;
    if ((dbg & 2u) && d->B <= 4096) {
        if (!stamps && hipMalloc(&stamps, 4096 * 16 * sizeof(unsigned long long)) != hipSuccess) stamps = nullptr;
        a.tstamp = stamps;
    }
#endif
    if (d->drop_p > 0.f) EEG_LAUNCH(token_block_fwd_kernel<true>, dim3(d->B), dim3(TB_THREADS), TB_LDS, stream, a);
    else EEG_LAUNCH(token_block_fwd_kernel<false>, dim3(d->B), dim3(TB_THREADS), TB_LDS, stream, a);
#if !defined(EEG_EMU)
    if (a.tstamp && ++calls % 20 == 0) {                         // phase lengths in shader clocks, mean over the workgroups, every 20th launch
        hipStreamSynchronize((hipStream_t)stream);
        static unsigned long long host[4096 * 16];
        hipMemcpy(host, stamps, (size_t)d->B * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double ph[9] = {0};
        for (int b = 0; b < d->B; ++b)
            for (int k = 0; k < 9; ++k) ph[k] += (double)(host[b * 16 + k + 1] - host[b * 16 + k]);
        fprintf(stderr, "token_block_fwd phases (s_memtime ticks, mean of %d workgroups): x->planes %.0f | embed GEMM %.0f | dropout+h %.0f | qkv+attention %.0f | out-proj %.0f | "
                        "LN1 %.0f | FFN1 %.0f | FFN2 %.0f | LN2+LN3 %.0f\n", d->B, ph[0] / d->B, ph[1] / d->B, ph[2] / d->B, ph[3] / d->B, ph[4] / d->B, ph[5] / d->B,
                ph[6] / d->B, ph[7] / d->B, ph[8] / d->B);
        double q[4] = {0};                                         // inside qkv+attention: [qkv pair 0, attention pair 0, qkv pair 1, attention pair 1]
        for (int b = 0; b < d->B; ++b) {
            const unsigned long long* h = host + b * 16;
            q[0] += (double)(h[10] - h[3]); q[1] += (double)(h[11] - h[10]); q[2] += (double)(h[12] - h[11]); q[3] += (double)(h[13] - h[12]);
        }
        fprintf(stderr, "  qkv+attention: qkv(heads 0,1) %.0f | attention %.0f | qkv(heads 2,3) + h planes out %.0f | attention %.0f\n", q[0] / d->B, q[1] / d->B, q[2] / d->B,
                q[3] / d->B);
    }
#endif
    return (int)hipGetLastError();
}

extern "C" long long eegclip_token_block_bwd_workspace_floats(int B) { return B < 1 ? 0 : (long long)B * 6 * 256; }

extern "C" int eegclip_token_block_bwd(const eegclip_token_block_bwd_desc* d, int part, void* stream) {
    if (!d || d->B < 1 || !d->packed || d->drop_p < 0.f || d->drop_p >= 1.f || part < 0 || part > 2) return EEGCLIP_EINVAL;
    tb_bwd_args a{static_cast<const unsigned short*>(d->packed), d->dn3, d->n2, d->r2, d->r1, d->f1, d->mu1, d->rs1, d->mu2, d->rs2, d->mu3, d->rs3,
                  d->ln1_g, d->ln2_g, d->ln2_b, d->ln3_g, d->dr1, d->dctx, d->partials, static_cast<unsigned char*>(d->df2p), static_cast<unsigned char*>(d->dg1p),
                  static_cast<unsigned char*>(d->da1p), static_cast<unsigned char*>(d->dr1p), static_cast<const unsigned char*>(d->dqkvp), d->drop_p, d->seed,
                  d->site_embed, d->site_attn_out, d->site_ffn_act, d->site_ffn_out};
    if (part == 0) {
        if (!d->dn3 || (!d->n2 && !d->ln2_b) || !d->r2 || !d->r1 || !d->f1 || !d->mu1 || !d->rs1 || !d->mu2 || !d->rs2 || !d->mu3 || !d->rs3 || !d->ln1_g || !d->ln2_g ||
            !d->ln3_g || !d->df2p || !d->dg1p || !d->da1p || !d->dr1 || !d->dctx || !d->partials)
            return EEGCLIP_EINVAL;
        const uintptr_t al16 = reinterpret_cast<uintptr_t>(d->packed) | reinterpret_cast<uintptr_t>(d->f1) | reinterpret_cast<uintptr_t>(d->df2p) |
                               reinterpret_cast<uintptr_t>(d->dg1p) | reinterpret_cast<uintptr_t>(d->da1p);
        const uintptr_t al8 = reinterpret_cast<uintptr_t>(d->dn3) | reinterpret_cast<uintptr_t>(d->n2) | reinterpret_cast<uintptr_t>(d->r2) |
                              reinterpret_cast<uintptr_t>(d->r1) | reinterpret_cast<uintptr_t>(d->ln1_g) | reinterpret_cast<uintptr_t>(d->ln2_g) |
                              reinterpret_cast<uintptr_t>(d->ln3_g) | reinterpret_cast<uintptr_t>(d->dr1) | reinterpret_cast<uintptr_t>(d->dctx);
        if ((al16 & 15u) || (al8 & 7u)) return EEGCLIP_EALIGN;
        if (d->drop_p > 0.f) EEG_LAUNCH(token_block_bwd_a_kernel<true>, dim3(d->B), dim3(TB_THREADS), TB_LDS, stream, a);
        else EEG_LAUNCH(token_block_bwd_a_kernel<false>, dim3(d->B), dim3(TB_THREADS), TB_LDS, stream, a);
    } else if (part == 1) {
        if (!d->dqkvp || !d->dr1) return EEGCLIP_EINVAL;
        if ((reinterpret_cast<uintptr_t>(d->packed) | reinterpret_cast<uintptr_t>(d->dr1) | reinterpret_cast<uintptr_t>(d->dqkvp) | reinterpret_cast<uintptr_t>(d->dr1p)) & 15u)
            return EEGCLIP_EALIGN;
        if (d->drop_p > 0.f) EEG_LAUNCH(token_block_bwd_b_kernel<true>, dim3(d->B), dim3(TB_THREADS), TB_AP_BYTES + TB_XF_BYTES, stream, a);
        else EEG_LAUNCH(token_block_bwd_b_kernel<false>, dim3(d->B), dim3(TB_THREADS), TB_AP_BYTES + TB_XF_BYTES, stream, a);
    } else {
        if (!d->partials || !d->dln3_g || !d->dln3_b || !d->dln2_g || !d->dln2_b || !d->dln1_g || !d->dln1_b) return EEGCLIP_EINVAL;
        tb_param_args pa{d->partials, {d->dln3_g, d->dln3_b, d->dln2_g, d->dln2_b, d->dln1_g, d->dln1_b}, d->B};
        EEG_LAUNCH(token_block_param_reduce_kernel, dim3(4, 6, d->B < 16 ? d->B : 16), dim3(256), 256 * sizeof(float), stream, pa);
    }
    return (int)hipGetLastError();
}
