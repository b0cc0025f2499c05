// Shared pieces of the GEMM kernels (gemm.hip: exact-f32 MFMA; gemm_x3.hip: split-bf16 MFMA): index maps, the fused epilogue over
// 16x16 MFMA accumulator tiles (the C/D fragment layout is the same for every MFMA shape x dtype on gfx950) and the split-K slice bounds.
#pragma once
#include "eeg_common.h"

namespace eeg {

// PLAIN = every index map is a plain stride (div = 2^62): offsets are one multiply, and the integer-division path of the two-level
// maps is not even instantiated (it was >1000 instructions of the unrolled staging/epilogue code)
template <bool PLAIN>
__device__ __forceinline__ long long goff(const eegclip_dim& d, int i) {
    if (PLAIN) return (long long)i * d.si;
    return dim_off(d, i);
}

// ---- epilogue: D[row = (lane>>4)*4 + r][col = lane&15] of each 16x16 accumulator -----------------------------------------------
// Three paths, chosen by workgroup-uniform tests BEFORE the element loops: at K ~ 250 a workgroup runs only 8 k-tiles, and a fully
// general per-element epilogue (eight uniform branches and 64-bit index maps per output) was ~28 % of its instructions.
//   split-K slices: atomicAdd of alpha*acc (+ bias on slice 0)
//   plain stride C, nothing but (bias_n, accumulate): pointer-bump stores
//   everything else: the general form
// the general form for one output element: v = alpha * acc on entry
// keep_known: -1 = evaluate the dropout mask here, 0 / 1 = the caller already has this element's decision (quad-shared Philox blocks)
// coff / roff: element offsets into C (and Cpre) / R, formed by the caller from per-row and per-column parts (a two-level map costs an integer
// division: a lane's 16 outputs share 8 rows and 2 columns)
template <bool PLAIN>
__device__ __forceinline__ void gemm_epilogue_element(const eegclip_gemm_desc& d, float v, int m, int n, long long coff, long long roff,
                                                      bool first_slice, float keep_scale, int keep_known = -1) {
    if (first_slice) {
        if (d.bias_n) v += d.bias_n[n];
        if (d.bias_m) v += d.bias_m[m];
    }
    if (d.split_k > 1) {
        atomicAdd(d.C + coff, v);
        return;
    }
    if (d.accumulate == 2) v += d.C[coff];               // accumulate FIRST: the old C goes through the rest of the chain (dropout of a sum)
    if (d.Cpre) d.Cpre[coff] = v;
    if (d.act == EEGCLIP_ACT_GELU) v = gelu_erf(v);
    else if (d.act == EEGCLIP_ACT_SILU) v = silu(v);
    if (d.drop_p > 0.f) {
        const bool keep = keep_known >= 0 ? keep_known != 0 : dropout_keep(d.seed, d.drop_site, (unsigned long long)m * (unsigned)d.N + (unsigned)n, d.drop_p);
        v = keep ? v * keep_scale : 0.f;
    }
    if (d.act == EEGCLIP_ACT_GELU_GRAD) v *= gelu_erf_grad(d.R[roff]);
    else if (d.R) v += d.R[roff];
    if (d.accumulate == 1) v += d.C[coff];
    d.C[coff] = v;
}

template <bool PLAIN>
__device__ __forceinline__ void gemm_epilogue(const eegclip_gemm_desc& d, const f32x4 (&acc)[2][2], int m0, int n0, int wr, int wc,
                                              int lane, bool first_slice) {
    const int nsplit = d.split_k;
    const int mb = m0 + wr * 32 + (lane >> 4) * 4, nb = n0 + wc * 32 + (lane & 15);
    if (PLAIN && !d.bias_m && (nsplit > 1 || (!d.Cpre && d.act == EEGCLIP_ACT_NONE && !(d.drop_p > 0.f) && !d.R))) {
        const long long ldc = d.Cm.si, ldn = d.Cn.si;
        const bool use_bias = d.bias_n != nullptr && first_slice;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = nb + nt * 16;
            if (n >= d.N) continue;
            const float bn = use_bias ? d.bias_n[n] : 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                float* cp = d.C + (long long)(mb + mt * 16) * ldc + (long long)n * ldn;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (mb + mt * 16 + r < d.M) {
                        const float v = d.alpha * acc[mt][nt][r] + bn;
                        if (nsplit > 1) atomicAdd(cp + r * ldc, v);
                        else cp[r * ldc] = d.accumulate ? cp[r * ldc] + v : v;
                    }
                }
            }
        }
        return;
    }
    const float keep_scale = d.drop_p > 0.f ? 1.0f / (1.0f - d.drop_p) : 1.0f;
    // dropout with N % 4 == 0: the 4 lanes of a quad (4 consecutive columns) sit in the same Philox block of every row, so for a lane's 4
    // accumulator rows the quad evaluates 4 blocks instead of 16 (a lane's 16 outputs otherwise need 16 whole blocks: ~1400 VALU
    // instructions per wave tile, +12 us on a 32 us GEMM)
    const bool quad_mask = d.drop_p > 0.f && nsplit == 1 && (d.N & 3) == 0;
    long long ccol[2], rcol[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = nb + nt * 16 < d.N ? nb + nt * 16 : 0;
        ccol[nt] = goff<PLAIN>(d.Cn, n);
        rcol[nt] = d.R ? goff<PLAIN>(d.Rn, n) : 0;
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        long long crow[4], rrow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mb + mt * 16 + r < d.M ? mb + mt * 16 + r : 0;
            crow[r] = goff<PLAIN>(d.Cm, m);
            rrow[r] = d.R ? goff<PLAIN>(d.Rm, m) : 0;
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = nb + nt * 16;
            bool kq[4] = {true, true, true, true};
            if (quad_mask) {
                const int j = lane & 3;
                const unsigned long long blk = ((unsigned long long)(mb + mt * 16 + j) * (unsigned)d.N + (unsigned)(n - j)) >> 2;
                dropout_keep_quad_blocks(d.seed, d.drop_site, blk, j, d.drop_p, kq);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mb + mt * 16 + r;
                if (m >= d.M || n >= d.N) continue;
                gemm_epilogue_element<PLAIN>(d, d.alpha * acc[mt][nt][r], m, n, crow[r] + ccol[nt], rrow[r] + rcol[nt], first_slice, keep_scale,
                                             quad_mask ? (int)kq[r] : -1);
            }
        }
    }
}

// ---- epilogue over TRANSPOSED accumulator tiles: the product was formed as D' = B_tile A_tile^T (MFMA rows = n, columns = m), so lane
// (fr = lane & 15, g = lane >> 4) holds, in the 4 registers of tile (mt, nt), the 4 CONSECUTIVE columns n = nb + 16 nt + 4 g + {0..3} of ONE
// output row m = mb + 16 mt + fr.  Everything per-row is then per-lane and everything along n is a 16-byte access: bias / residual / C are
// read and written as f32x4 (a dword-aligned address is enough for global 16-byte accesses), one Philox block covers the lane's 4 outputs.
// Two measured reasons (tools/xp_abl.py, a 16384 x 256 x 250 launch: 27 us, of which an EMPTY k-loop left 16, no epilogue 6, plain 16-byte
// stores of the tile 8): (1) with the standard layout a lane's 16 outputs are 16 separate 4-byte stores; (2) the epilogue was a chain of
// "load bias -> wait -> compute -> store" blocks, one per 16x16 tile, each behind its own uniform branches: every block paid a full memory
// latency.  Here ALL loads of a wave tile row are issued before the first use: the bias vectors once per call, residual / accumulate operands
// per row of tiles.
typedef float f32x4a4_t __attribute__((ext_vector_type(4), aligned(4)));

template <bool PLAIN, int MT, int NT>
__device__ __forceinline__ void gemm_epilogue_t(const eegclip_gemm_desc& d, const f32x4 (&acc)[MT][NT], int mb, int nb, int lane, bool first_slice,
                                                bool split, float* __restrict__ C) {
    const int fr = lane & 15, g = lane >> 4;
    const float keep_scale = d.drop_p > 0.f ? 1.0f / (1.0f - d.drop_p) : 1.0f;
    const bool vec_c = PLAIN ? d.Cn.si == 1 : (d.Cn.si == 1 && d.Cn.div > (1LL << 40));          // columns contiguous in C (and Cpre)
    const bool vec_r = d.R != nullptr && (PLAIN ? d.Rn.si == 1 : (d.Rn.si == 1 && d.Rn.div > (1LL << 40)));
    const bool use_bias = first_slice && d.bias_n != nullptr;
    // (split: this call adds one K slice's share atomically; false also for the workspace reduction, whose last workgroup owns the tile)
    // wave-uniform: does the 16-column tile nt lie inside N entirely (vector accesses) or is it the ragged last tile (per-element, clamped)?
    bool tile_full[NT];
    int ncol[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        tile_full[nt] = nb + 16 * nt + 16 <= d.N;
        ncol[nt] = nb + 16 * nt + 4 * g;
    }
    auto load4 = [&](const float* base, long long row_off, const eegclip_dim& cn, int nt, bool vec) -> f32x4a4_t {
        const int n = ncol[nt];
        if (vec && tile_full[nt]) return *reinterpret_cast<const f32x4a4_t*>(base + row_off + n);
        f32x4a4_t r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ne = n + e < d.N ? n + e : d.N - 1;          // clamped: never dereferenced out of range, never stored
            r[e] = base[row_off + goff<PLAIN>(cn, ne)];
        }
        return r;
    };
    f32x4a4_t bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias[nt] = f32x4a4_t{0.f, 0.f, 0.f, 0.f};
    if (use_bias) {
        const eegclip_dim unit{1LL << 62, 0, 1};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (nb + 16 * nt < d.N) {
                const int n = ncol[nt];
                if (tile_full[nt]) bias[nt] = *reinterpret_cast<const f32x4a4_t*>(d.bias_n + n);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) bias[nt][e] = d.bias_n[n + e < d.N ? n + e : d.N - 1];
                }
            }
        }
        (void)unit;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int mm = mb + 16 * mt + fr;
        const bool row_ok = mm < d.M;
        const int m = row_ok ? mm : d.M - 1;                       // clamped row: loads stay in range, stores are predicated
        const long long crow = goff<PLAIN>(d.Cm, m);
        const long long rrow = d.R ? goff<PLAIN>(d.Rm, m) : 0;
        const float bm = (first_slice && d.bias_m) ? d.bias_m[m] : 0.f;
        // phase 1: every load of this row of tiles
        f32x4a4_t rr[NT], oo[NT];
        if (d.R && !split) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                if (nb + 16 * nt < d.N) rr[nt] = load4(d.R, rrow, d.Rn, nt, vec_r);
        }
        if (d.accumulate && !split) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                if (nb + 16 * nt < d.N) oo[nt] = load4(C, crow, d.Cn, nt, vec_c);
        }
        // phase 2: arithmetic and stores
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (nb + 16 * nt >= d.N) continue;                     // (wave uniform)
            const int n = ncol[nt];
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = d.alpha * acc[mt][nt][e] + bm + bias[nt][e];
            const bool vc = vec_c && tile_full[nt];
            const long long c0 = crow + goff<PLAIN>(d.Cn, n < d.N ? n : d.N - 1);
            if (split) {
                if (row_ok) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < d.N) atomicAdd(C + (vc ? c0 + e : crow + goff<PLAIN>(d.Cn, n + e)), v[e]);
                }
                continue;
            }
            if (d.accumulate == 2) {                               // accumulate FIRST (see gemm_epilogue_element)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += oo[nt][e];
            }
            if (d.Cpre && row_ok) {
                if (vc) *reinterpret_cast<f32x4a4_t*>(d.Cpre + c0) = f32x4a4_t{v[0], v[1], v[2], v[3]};
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < d.N) d.Cpre[crow + goff<PLAIN>(d.Cn, n + e)] = v[e];
                }
            }
            if (d.act == EEGCLIP_ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            } else if (d.act == EEGCLIP_ACT_SILU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = silu(v[e]);
            }
            if (d.drop_p > 0.f) {
                const unsigned long long idx = (unsigned long long)mm * (unsigned)d.N + (unsigned)n;
                bool keep[4];
                if ((d.N & 3) == 0) dropout_keep4(d.seed, d.drop_site, idx, d.drop_p, keep);       // the lane's 4 outputs are ONE Philox block
                else {                                                                             // rows start off a block boundary: at most two blocks
                    const unsigned off = (unsigned)(idx & 3ull);
                    bool k0[4], k1[4];
                    dropout_keep4(d.seed, d.drop_site, idx - off, d.drop_p, k0);
                    if (off) dropout_keep4(d.seed, d.drop_site, idx - off + 4, d.drop_p, k1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) keep[e] = (off + e < 4) ? k0[(off + e) & 3] : k1[(off + e) & 3];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = keep[e] ? v[e] * keep_scale : 0.f;
            }
            if (d.R) {
                if (d.act == EEGCLIP_ACT_GELU_GRAD) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= gelu_erf_grad(rr[nt][e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rr[nt][e];
                }
            }
            if (d.accumulate == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += oo[nt][e];
            }
            if (!row_ok) continue;
            if (vc) *reinterpret_cast<f32x4a4_t*>(C + c0) = f32x4a4_t{v[0], v[1], v[2], v[3]};
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < d.N) C[crow + goff<PLAIN>(d.Cn, n + e)] = v[e];
            }
        }
    }
}

template <int BK>
__device__ __forceinline__ void gemm_k_slice(const eegclip_gemm_desc& d, int slice, int& kt_begin, int& kt_end) {
    const int ktiles = (d.K + BK - 1) / BK;
    const int tiles_per = (ktiles + d.split_k - 1) / d.split_k;
    kt_begin = slice * tiles_per;
    kt_end = kt_begin + tiles_per;
    if (kt_end > ktiles) kt_end = ktiles;
}

// gemm_x3.hip: the split-bf16 kernel for the plain-stride operand classes (akc / bkc: operand is k-contiguous)
// k2: both operands row-contiguous, k through two-level maps (akc = bkc = false, c_plain = true)
int launch_gemm_x3(const eegclip_gemm_desc& d, bool akc, bool bkc, bool c_plain, bool k2, void* stream);
long long gemm_x3_workspace_bytes(const eegclip_gemm_desc& d);

}  // namespace eeg
