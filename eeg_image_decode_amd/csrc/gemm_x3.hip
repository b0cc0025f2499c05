// Split-precision fp32 GEMM on the bf16 matrix cores (eegclip_gemm_desc.precision = EEGCLIP_PREC_BF16X3).
//
//   C[m,n] (+)= epilogue(alpha * sum_k A[m,k] B[k,n]),   a*b  :=  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi   (fp32 accumulate)
//
// with a = a_hi + a_lo, a_hi = bf16(a), a_lo = bf16(a - a_hi) (a - a_hi is exact in fp32).  v_mfma_f32_16x16x4_f32 runs at 1/16 of
// the bf16 rate (157 TF vs 2.5 PF); three bf16 products cost 3/16 of an exact one, and the GEMMs of this path (K ~ 250, N ~ 250..1024)
// stop being bound by the matrix pipe at all: what is left is operand staging, which this kernel shares with gemm.hip's fast kernel.
// Same operand classes as that kernel (plain strides, even leading dimensions, 8-byte aligned bases, 32-bit reach), same XCD-aware
// tile order, same split-K scheme, same epilogue (gemm_epilogue.h) -- the C/D fragment layout of a 16x16 MFMA does not depend on dtype.
//
// The split happens ONCE per staged element, in registers, between the global load and the LDS store (v_cvt_pk_bf16_f32 x2, one
// shift / and, one subtract per element): HBM and L2 only ever see fp32, and the LDS image costs the same 4 bytes per element.
//
// LDS image of a BT x BK operand tile (both operand classes end up in the SAME image, so the MFMA loop has one form):
//     row r (= m or n within the tile):  [ hi plane: BK bf16 | lo plane: BK bf16 | 32 bytes pad ]        row stride RS = 4 BK + 32
//     inside a plane the 16-byte chunk c (8 consecutive k) sits at chunk slot c ^ ((r >> 2) & (BK/8 - 1))
//   * MFMA operand fetch: lane (fr = lane & 15, g = lane >> 4) of k-step s reads the 16 bytes of row fr, chunk 4 s + g with one
//     ds_read_b128 per plane -- conflict free (tools/micro/lds_layout_search.py enumerates the lane groups of MI355X_MICROARCH.md)
//   * k-contiguous operand ("KC": X and W of Y = X W^T): a thread loads 4 consecutive k of one row (two 8-byte loads: rows of 250 floats
//     are only 8-byte aligned), splits them and writes 8 bytes per plane (ds_write_b64, 2-way at worst)
//   * row-contiguous operand ("MC": dY^T and X of dW = dY^T X; W of dX = dY W): a thread loads 4 consecutive k ROWS x 2 adjacent m
//     (coalesced along m), i.e. a 4 x 2 block -- the transpose the k-contiguous image needs happens in registers -- and writes 8 bytes
//     per (m, plane)
//
// Tile shapes (template): BT x BT output per 256-thread workgroup, 2 x 2 wavefronts, each wave (BT/32)^2 MFMA 16x16x32 tiles;
// BK = 32 or 64; DB = two LDS images (one barrier per k-tile: the next tile is converted and stored behind the current tile's MFMAs)
// or one (two barriers).  launch_gemm_x3() picks per problem.
#include "eeg_common.h"
#include "gemm_epilogue.h"

#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

namespace eeg {

constexpr int X3_THREADS = 256;

template <int BK>
struct x3_geom {
    static constexpr int RS = 4 * BK + 32;       // row stride in bytes
    static constexpr int NCH = BK / 8;           // 16-byte chunks per plane
    // byte offset of k-quad kq (4 consecutive k = 8 bytes) of row r in plane p
    __device__ static __forceinline__ int quad(int r, int p, int kq) {
        return r * RS + p * (2 * BK) + ((((kq >> 1) ^ (r >> 2)) & (NCH - 1)) << 4) + ((kq & 1) << 3);
    }
    // byte offset of 16-byte chunk c of row r in plane p
    __device__ static __forceinline__ int chunk(int r, int p, int c) { return r * RS + p * (2 * BK) + (((c ^ (r >> 2)) & (NCH - 1)) << 4); }
};

typedef float f32x4u_t __attribute__((ext_vector_type(4), aligned(4)));      // 16-byte global load from a dword-aligned address

// split-K workspace: split_k plain [M][ld] slabs of partial products, ld = N rounded up to 4 floats
static inline long long x3_ws_ld(int N) { return ((long long)N + 3) & ~3LL; }
static inline long long x3_workspace_bytes(int M, int N, int split_k) { return 4LL * split_k * M * x3_ws_ld(N); }

// C[m,n] (+)= sum_s slab_s[m,n] + bias_n[n] + bias_m[m], slabs added in slice order (a result independent of scheduling); one thread per 4 columns
template <bool PLAIN>
__global__ __launch_bounds__(256) void x3_splitk_reduce_kernel(const eegclip_gemm_desc d, const float* __restrict__ slabs, int ld) {
    const int q = ld >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)d.M * q) return;
    const int m = (int)(idx / q), n = (int)(idx % q) * 4;
    const size_t slab = (size_t)d.M * ld;
    const float* p = slabs + (size_t)m * ld + n;
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    for (int s = 1; s < d.split_k; ++s) v += *reinterpret_cast<const f32x4*>(p + s * slab);
    const float bm = d.bias_m ? d.bias_m[m] : 0.f;
    const long long crow = goff<PLAIN>(d.Cm, m);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (n + e >= d.N) break;
        float o = v[e] + bm + (d.bias_n ? d.bias_n[n + e] : 0.f);
        float* c = d.C + crow + goff<PLAIN>(d.Cn, n + e);
        if (d.accumulate) o += *c;
        *c = o;
    }
}

// K2: both operands row-contiguous with k running through two-level maps {div, so, si} (the value-embedding weight gradient contracts over
// the 63 channel rows of every 64-row sample): a thread's k rows advance by BK per tile, so (offset, remainder) are carried along and
// wrapped by subtraction -- no division in the loop (same scheme as gemm.hip's K2 instantiation)
// TRANS: the product is formed transposed (MFMA rows = n) and finished by gemm_epilogue_t (16-byte accesses along n); the split-K launches keep
// the standard orientation: their epilogue is atomics only, and a wave's atomic instruction over 4 rows x 16 consecutive words coalesces where
// 16 rows x 4 strided words does not (measured: the weight-gradient GEMMs 62 -> 174 us with the transposed layout).
template <int BT, int BK, bool DB, bool A_KC, bool B_KC, bool C_PLAIN, bool K2 = false, bool TRANS = false>
__global__ __launch_bounds__(X3_THREADS, (BT == 64 && !K2) ? 4 : (BT == 64 ? 3 : 1)) void gemm_x3_kernel(const eegclip_gemm_desc d, int gx, int ntiles, int chunk) {
    static_assert(!K2 || (!A_KC && !B_KC), "two-level k maps are implemented for row-contiguous operands");
    using G = x3_geom<BK>;
    constexpr int WT = BT / 32;                  // MFMA tiles per wave and dimension
    constexpr int IMG = BT * G::RS;              // bytes of one operand image
    constexpr int NQ = BK / 4;                   // k-quads per row
    // KC staging: thread -> (k-quad t % NQ, row t / NQ + RPP i)
    constexpr int RPP = X3_THREADS / NQ, KC_PASS = BT / RPP;
    // MC staging: thread -> (m-pair t % NP, k-quad t / NP + QPP i)
    constexpr int NP = BT / 2, QPP = X3_THREADS / NP, MC_PASS = NQ / QPP;
    static_assert(KC_PASS >= 1 && MC_PASS >= 1 && BT % RPP == 0 && NQ % QPP == 0, "tile shape");
    EEG_LDS_BASE(unsigned char, lds);            // [DB ? 2 : 1][A image | B image]

    const int bid = (int)blockIdx.x;
    int logical, slice = 0;
    if (d.split_k == 1) {
        logical = (bid & 7) * chunk + (bid >> 3);                   // XCD b % 8 owns a contiguous run of tiles (n fastest)
        if (logical >= ntiles) return;
    } else {
        const int slot = bid >> 3;                                  // all tiles of one K slice on one XCD (see gemm.hip)
        slice = (bid & 7) + 8 * (slot / ntiles);
        logical = slot % ntiles;
        if (slice >= d.split_k) return;
    }
    const int m0 = (logical / gx) * BT, n0 = (logical % gx) * BT;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = wave_uniform(t >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, g = lane >> 4;
    int kt_begin, kt_end;
    gemm_k_slice<BK>(d, slice, kt_begin, kt_end);

    const int a_ld = A_KC ? (int)d.Am.si : (int)d.Ak.si, b_ld = B_KC ? (int)d.Bn.si : (int)d.Bk.si;
    const int kc_kq = t % NQ, kc_r0 = t / NQ, mc_mp = t % NP, mc_q0 = t / NP;
    int a_fix[A_KC ? KC_PASS : 1], b_fix[B_KC ? KC_PASS : 1];      // KC: element offset of the (clamped) row
    int a_col = 0, b_col = 0;                                      // MC: clamped first row of the pair
    if (A_KC) {
#pragma unroll
        for (int i = 0; i < KC_PASS; ++i) { const int m = m0 + kc_r0 + RPP * i; a_fix[i] = (m < d.M ? m : d.M - 1) * a_ld; }
    } else { const int m = m0 + 2 * mc_mp; a_col = m < d.M ? m : d.M - 2; a_fix[0] = 0; }
    if (B_KC) {
#pragma unroll
        for (int i = 0; i < KC_PASS; ++i) { const int n = n0 + kc_r0 + RPP * i; b_fix[i] = (n < d.N ? n : d.N - 1) * b_ld; }
    } else { const int n = n0 + 2 * mc_mp; b_col = n < d.N ? n : d.N - 2; b_fix[0] = 0; }
    // K2: element offset / remainder of each k row this thread stages (MC_PASS x 4 rows per operand), for the FIRST tile of the slice
    constexpr int K2R = K2 ? MC_PASS * 4 : 1;
    int a_koff[K2R], a_rem[K2R], b_koff[K2R], b_rem[K2R];
    const int a_div = K2 ? (d.Ak.div > 0x7fffffffLL ? 0x7fffffff : (int)d.Ak.div) : 1;      // plain map: one block that never wraps
    const int b_div = K2 ? (d.Bk.div > 0x7fffffffLL ? 0x7fffffff : (int)d.Bk.div) : 1;
    const int a_wrap = K2 ? (int)(d.Ak.so - d.Ak.div * d.Ak.si) : 0, b_wrap = K2 ? (int)(d.Bk.so - d.Bk.div * d.Bk.si) : 0;
    if (K2) {
#pragma unroll
        for (int i = 0; i < MC_PASS; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kt_begin * BK + 4 * (mc_q0 + QPP * i) + j;
                a_rem[4 * i + j] = k % a_div;
                a_koff[4 * i + j] = (k / a_div) * (int)d.Ak.so + a_rem[4 * i + j] * a_ld;
                b_rem[4 * i + j] = k % b_div;
                b_koff[4 * i + j] = (k / b_div) * (int)d.Bk.so + b_rem[4 * i + j] * b_ld;
            }
    }

    // registers of the tile in flight; the "k beyond K -> 0" select is applied at LDS-store time (a select next to the load would
    // make the wave wait for its prefetch before the MFMAs it overlaps with)
    constexpr int A_REGS = A_KC ? KC_PASS * 2 : MC_PASS * 4, B_REGS = B_KC ? KC_PASS * 2 : MC_PASS * 4;
    f32x2_t ra[A_REGS], rb[B_REGS];
    unsigned a_ok = 0, b_ok = 0;

    auto load_op = [&](auto kc_tag, const float* P, const int* fix, int col, int ld, f32x2_t* r, unsigned& ok, int k0, int* koff, int* rem,
                       int div, int wrap) {
        constexpr bool KC = decltype(kc_tag)::value;
        ok = 0;
        if (KC) {
            const int ka = k0 + 4 * kc_kq;
            if (k0 + BK <= d.K) {
                // interior tile (workgroup-uniform test): one 16-byte load per 4 consecutive k.  Rows of 250 floats are only 8-byte
                // aligned; the hardware needs dword alignment for global_load_dwordx4, and half the vector-memory instructions go away
                ok = 3u;
#pragma unroll
                for (int i = 0; i < KC_PASS; ++i) {
                    const f32x4u_t v = *reinterpret_cast<const f32x4u_t*>(P + fix[i] + ka);
                    r[2 * i] = f32x2_t{v[0], v[1]};
                    r[2 * i + 1] = f32x2_t{v[2], v[3]};
                }
            } else {
                const bool ok0 = ka < d.K, ok1 = ka + 2 < d.K;      // K is even: a pair is in or out as a whole
                const int k_0 = ok0 ? ka : 0, k_1 = ok1 ? ka + 2 : 0;
                ok = (ok0 ? 1u : 0u) | (ok1 ? 2u : 0u);
#pragma unroll
                for (int i = 0; i < KC_PASS; ++i) {
                    r[2 * i] = *reinterpret_cast<const f32x2_t*>(P + fix[i] + k_0);
                    r[2 * i + 1] = *reinterpret_cast<const f32x2_t*>(P + fix[i] + k_1);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < MC_PASS; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = k0 + 4 * (mc_q0 + QPP * i) + j;
                    const bool okk = k < d.K;
                    const int off = K2 ? koff[4 * i + j] : k * ld;
                    r[4 * i + j] = *reinterpret_cast<const f32x2_t*>(P + (okk ? off : 0) + col);
                    ok |= (okk ? 1u : 0u) << (4 * i + j);
                    if (K2) {
                        koff[4 * i + j] += BK * ld;
                        rem[4 * i + j] += BK;
                        while (rem[4 * i + j] >= div) { rem[4 * i + j] -= div; koff[4 * i + j] += wrap; }
                    }
                }
        }
    };
    auto store_op = [&](auto kc_tag, unsigned char* img, const f32x2_t* r, unsigned ok) {
        constexpr bool KC = decltype(kc_tag)::value;
        const f32x2_t zero{0.f, 0.f};
        if (KC) {
#pragma unroll
            for (int i = 0; i < KC_PASS; ++i) {
                const f32x2_t p0 = ok & 1u ? r[2 * i] : zero, p1 = ok & 2u ? r[2 * i + 1] : zero;
                u32x2_t hi, lo;
                x3_split4(p0[0], p0[1], p1[0], p1[1], hi, lo);
                const int row = kc_r0 + RPP * i;
                *reinterpret_cast<u32x2_t*>(img + G::quad(row, 0, kc_kq)) = hi;
                *reinterpret_cast<u32x2_t*>(img + G::quad(row, 1, kc_kq)) = lo;
            }
        } else {
#pragma unroll
            for (int i = 0; i < MC_PASS; ++i) {
                f32x2_t v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (ok >> (4 * i + j)) & 1u ? r[4 * i + j] : zero;
                const int kq = mc_q0 + QPP * i;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    u32x2_t hi, lo;
                    x3_split4(v[0][e], v[1][e], v[2][e], v[3][e], hi, lo);
                    *reinterpret_cast<u32x2_t*>(img + G::quad(2 * mc_mp + e, 0, kq)) = hi;
                    *reinterpret_cast<u32x2_t*>(img + G::quad(2 * mc_mp + e, 1, kq)) = lo;
                }
            }
        }
    };
    using akc_t = std::integral_constant<bool, A_KC>;
    using bkc_t = std::integral_constant<bool, B_KC>;
    auto load_tile = [&](int kt) {
        load_op(akc_t{}, d.A, a_fix, a_col, a_ld, ra, a_ok, kt * BK, a_koff, a_rem, a_div, a_wrap);
        load_op(bkc_t{}, d.B, b_fix, b_col, b_ld, rb, b_ok, kt * BK, b_koff, b_rem, b_div, b_wrap);
    };
    auto store_tile = [&](unsigned char* buf) {
        store_op(akc_t{}, buf, ra, a_ok);
        store_op(bkc_t{}, buf + IMG, rb, b_ok);
    };

    f32x4 acc[WT][WT];
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool do_rowsum = d.rowsum_a != nullptr && n0 == 0;
    float rowsum = 0.f;
    auto compute = [&](const unsigned char* buf) {
        if (do_rowsum && t < BT) {                               // bias gradients: row sums of the staged A tile (hi + lo = a to 2^-17)
#pragma unroll
            for (int c = 0; c < G::NCH; ++c) {
                const bf16x8 h = *reinterpret_cast<const bf16x8*>(buf + G::chunk(t, 0, c));
                const bf16x8 l = *reinterpret_cast<const bf16x8*>(buf + G::chunk(t, 1, c));
#pragma unroll
                for (int e = 0; e < 8; ++e) rowsum += bf16_bits_to_f32((unsigned short)h[e]) + bf16_bits_to_f32((unsigned short)l[e]);
            }
        }
#pragma unroll
        for (int s = 0; s < BK / 32; ++s) {
            bf16x8 bh[WT], bl[WT];
#pragma unroll
            for (int j = 0; j < WT; ++j) {
                const int row = wc * (BT / 2) + 16 * j + fr;
                bh[j] = *reinterpret_cast<const bf16x8*>(buf + IMG + G::chunk(row, 0, 4 * s + g));
                bl[j] = *reinterpret_cast<const bf16x8*>(buf + IMG + G::chunk(row, 1, 4 * s + g));
            }
#pragma unroll
            for (int i = 0; i < WT; ++i) {
                const int row = wr * (BT / 2) + 16 * i + fr;
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(buf + G::chunk(row, 0, 4 * s + g));
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(buf + G::chunk(row, 1, 4 * s + g));
#pragma unroll
                for (int j = 0; j < WT; ++j) {
                    if (TRANS) {
                        acc[i][j] = mfma_bf16_16x16x32(bh[j], al, acc[i][j]);
                        acc[i][j] = mfma_bf16_16x16x32(bl[j], ah, acc[i][j]);
                        acc[i][j] = mfma_bf16_16x16x32(bh[j], ah, acc[i][j]);
                    } else {
                        acc[i][j] = mfma_bf16_16x16x32(al, bh[j], acc[i][j]);
                        acc[i][j] = mfma_bf16_16x16x32(ah, bl[j], acc[i][j]);
                        acc[i][j] = mfma_bf16_16x16x32(ah, bh[j], acc[i][j]);
                    }
                }
            }
        }
    };

    if (DB) {
        int cur = 0;
        if (kt_begin < kt_end) {
            load_tile(kt_begin);
            store_tile(lds);
        }
        __syncthreads();
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const bool more = kt + 1 < kt_end;
            if (more) load_tile(kt + 1);                         // global -> registers, in flight under this tile's MFMAs
            compute(lds + cur * 2 * IMG);
            if (more) store_tile(lds + (cur ^ 1) * 2 * IMG);     // split + store into the OTHER image: one barrier per k-tile
            __syncthreads();
            cur ^= 1;
        }
    } else {
        if (kt_begin < kt_end) load_tile(kt_begin);
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            store_tile(lds);
            __syncthreads();
            if (kt + 1 < kt_end) load_tile(kt + 1);
            compute(lds);
            __syncthreads();
        }
    }
    if (do_rowsum && t < BT && m0 + t < d.M) atomicAdd(d.rowsum_a + m0 + t, rowsum);
    if constexpr (TRANS) {
        // split_k > 1 here = the workspace route: the launcher pointed C at a stack of split_k plain [M][ld] slabs, slice s fills slab s with
        // its partial product (no bias, no accumulate) and eegclip's reduce kernel adds the slabs in slice order afterwards
        float* cbase = d.C + (d.split_k > 1 ? (size_t)slice * (size_t)d.M * (size_t)d.Cm.si : (size_t)0);
        gemm_epilogue_t<C_PLAIN, WT, WT>(d, acc, m0 + wr * (BT / 2), n0 + wc * (BT / 2), lane, slice == 0, false, cbase);
    } else {
        // (explicit 32x32 sub-blocks: a loop over them is too large for the unroller and would index `acc` at run time -> scratch)
#define EEG_X3_EPI(SI, SJ)                                                                                                          \
    {                                                                                                                               \
        f32x4 a2[2][2] = {{acc[2 * SI][2 * SJ], acc[2 * SI][2 * SJ + 1]}, {acc[2 * SI + 1][2 * SJ], acc[2 * SI + 1][2 * SJ + 1]}};  \
        gemm_epilogue<C_PLAIN>(d, a2, m0 + wr * (BT / 2) + 32 * SI, n0 + wc * (BT / 2) + 32 * SJ, 0, 0, lane, slice == 0);          \
    }
        EEG_X3_EPI(0, 0)
        if constexpr (WT == 4) {
            EEG_X3_EPI(0, 1)
            EEG_X3_EPI(1, 0)
            EEG_X3_EPI(1, 1)
        }
#undef EEG_X3_EPI
    }
}

typedef unsigned xp_u32x4 __attribute__((ext_vector_type(4)));

// rows of an fp32 matrix -> bf16 planes hi / lo, [rows][ld_out] with zeros beyond `cols`; TRANSPOSE: the planes of the transposed matrix.
// A table of up to 24 matrices per launch (every Linear weight of the encoder, both orientations, in ONE launch per step).
struct xp_split_entry {
    const float* src;
    unsigned short* hi;
    unsigned short* lo;
    int rows, cols;                  // of the SOURCE matrix
    long long ld_src, ld_out;
    int transpose, first_block;      // workgroups [first_block, next first_block) belong to this entry
    float* copy;                     // optional fp32 copy of the source, rows ld_copy apart (several sources stacked into one GEMM operand)
    long long ld_copy;
};
constexpr int XP_SPLIT_MAX = 24;
struct xp_split_table {
    int n;
    xp_split_entry e[XP_SPLIT_MAX];
};
__global__ __launch_bounds__(256) void split_rows_kernel(const xp_split_table tb) {
    int ei = 0;
    for (int i = 1; i < tb.n; ++i) ei += (int)blockIdx.x >= tb.e[i].first_block ? 1 : 0;
    const xp_split_entry& E = tb.e[ei];
    const int orow_n = E.transpose ? E.cols : E.rows, ocol_n = E.transpose ? E.rows : E.cols;
    const long long per_row = E.ld_out;                              // output elements per row incl. padding
    const long long total = (long long)orow_n * per_row;
    const int span = (ei + 1 < tb.n ? tb.e[ei + 1].first_block : (int)gridDim.x) - E.first_block;
    if (!E.transpose && E.ld_src == E.cols && E.ld_out == E.cols && (total & 3) == 0 && (!E.copy || (E.ld_copy == E.cols && !(reinterpret_cast<uintptr_t>(E.copy) & 15u))) &&
        !((reinterpret_cast<uintptr_t>(E.src) & 15u) | ((reinterpret_cast<uintptr_t>(E.hi) | reinterpret_cast<uintptr_t>(E.lo)) & 7u))) {
        // a dense matrix split in place-order (the diffusion prior's 1024-wide inputs): 4 elements per lane, 16-byte loads, 8-byte plane stores
        for (long long q4 = (long long)((int)blockIdx.x - E.first_block) * 256 + threadIdx.x; 4 * q4 < total; q4 += 256LL * span) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(E.src + 4 * q4);
            u32x2_t h, l;
            x3_split4(v[0], v[1], v[2], v[3], h, l);
            *reinterpret_cast<u32x2_t*>(E.hi + 4 * q4) = h;
            *reinterpret_cast<u32x2_t*>(E.lo + 4 * q4) = l;
            if (E.copy) *reinterpret_cast<f32x4*>(E.copy + 4 * q4) = v;          // (this path: ld_copy == cols, 16-byte aligned -- checked by the host)
        }
        return;
    }
    if (E.transpose == 2) {
        // out[c][r .. r + 3] = src[r .. r + 3][c]: a transposing split WITHOUT LDS and without padding writes (rows % 4 == 0; the output may be a column block of
        // a wider matrix).  Consecutive lanes take consecutive source columns: four coalesced row reads, one 8-byte store per plane and lane.  LDS-free and
        // ~20 VGPRs on purpose: at the start of a training step it runs on the second stream BESIDE the fused transformer-block forward, whose workgroups hold
        // every CU's LDS and all but ~30 VGPRs per SIMD lane -- the tiled eegclip_split_transpose waited for it to end (109 us in the step's trace)
        const long long items = (long long)(E.rows / 4) * E.cols;
        for (long long q = (long long)((int)blockIdx.x - E.first_block) * 256 + threadIdx.x; q < items; q += 256LL * span) {
            const int c = (int)(q % E.cols), r = 4 * (int)(q / E.cols);
            const float* sp = E.src + (long long)r * E.ld_src + c;
            const float v0 = sp[0], v1 = sp[E.ld_src], v2 = sp[2 * E.ld_src], v3 = sp[3 * E.ld_src];
            u32x2_t h, l;
            x3_split4(v0, v1, v2, v3, h, l);
            *reinterpret_cast<u32x2_t*>(E.hi + (long long)c * E.ld_out + r) = h;
            *reinterpret_cast<u32x2_t*>(E.lo + (long long)c * E.ld_out + r) = l;
        }
        return;
    }
    for (long long q = (long long)((int)blockIdx.x - E.first_block) * 256 + threadIdx.x; q < total;
         q += 256LL * ((ei + 1 < tb.n ? tb.e[ei + 1].first_block : (int)gridDim.x) - E.first_block)) {
        const int r = (int)(q / per_row), c = (int)(q % per_row);
        float v = 0.f;
        if (c < ocol_n) v = E.transpose ? E.src[(long long)c * E.ld_src + r] : E.src[(long long)r * E.ld_src + c];
        const unsigned short h = f32_to_bf16_bits(v);
        E.hi[q] = h;
        E.lo[q] = f32_to_bf16_bits(v - bf16_bits_to_f32(h));
        if (E.copy && c < ocol_n) E.copy[(long long)r * E.ld_copy + c] = v;      // (never with transpose: refused by the host)
    }
}

template <int BT, int BK, bool DB>
static int x3_launch_cfg(const eegclip_gemm_desc& d_in, bool akc, bool bkc, bool c_plain_in, bool k2, void* stream) {
    bool c_plain = c_plain_in;
    const int gx = (d_in.N + BT - 1) / BT, gy = (d_in.M + BT - 1) / BT;
    const int ntiles = gx * gy, chunk = (ntiles + 7) / 8;
    const dim3 grid(d_in.split_k == 1 ? 8 * chunk : 8 * ((d_in.split_k + 7) / 8) * ntiles), block(X3_THREADS);
    const size_t lds = (size_t)(DB ? 2 : 1) * 2 * BT * x3_geom<BK>::RS;
    // split-K: through the workspace (transposed product into per-slice slabs + one ordered reduction) when the caller gave one, else atomics
    const bool ws = d_in.split_k > 1 && d_in.workspace != nullptr && d_in.workspace_bytes >= x3_workspace_bytes(d_in.M, d_in.N, d_in.split_k) &&
                    (reinterpret_cast<uintptr_t>(d_in.workspace) & 15u) == 0;
    eegclip_gemm_desc d = d_in;
    if (ws) {
        d.C = d_in.workspace;
        d.Cm = eegclip_dim{1LL << 62, 0, x3_ws_ld(d_in.N)};
        d.Cn = eegclip_dim{1LL << 62, 0, 1};
        d.bias_n = d.bias_m = nullptr;
        d.accumulate = 0;
        c_plain = true;
    }
    const bool trans = d.split_k == 1 || ws;
#define EEG_X3_GO(AK, BK_)                                                                                                        \
    do {                                                                                                                          \
        if (!trans) {                                                                                                             \
            if (c_plain) EEG_LAUNCH((gemm_x3_kernel<BT, BK, DB, AK, BK_, true, false, false>), grid, block, lds, stream, d, gx, ntiles, chunk);   \
            else         EEG_LAUNCH((gemm_x3_kernel<BT, BK, DB, AK, BK_, false, false, false>), grid, block, lds, stream, d, gx, ntiles, chunk);  \
        } else {                                                                                                                  \
            if (c_plain) EEG_LAUNCH((gemm_x3_kernel<BT, BK, DB, AK, BK_, true, false, true>), grid, block, lds, stream, d, gx, ntiles, chunk);    \
            else         EEG_LAUNCH((gemm_x3_kernel<BT, BK, DB, AK, BK_, false, false, true>), grid, block, lds, stream, d, gx, ntiles, chunk);   \
        }                                                                                                                         \
    } while (0)
    if (k2) {
        if (ws) EEG_LAUNCH((gemm_x3_kernel<BT, BK, DB, false, false, true, true, true>), grid, block, lds, stream, d, gx, ntiles, chunk);
        else    EEG_LAUNCH((gemm_x3_kernel<BT, BK, DB, false, false, true, true, false>), grid, block, lds, stream, d, gx, ntiles, chunk);
    }
    else if (akc && bkc)   EEG_X3_GO(true, true);
    else if (akc && !bkc)  EEG_X3_GO(true, false);
    else if (!akc && bkc)  EEG_X3_GO(false, true);
    else                   EEG_X3_GO(false, false);
#undef EEG_X3_GO
    if (ws) {
        const int ld = (int)x3_ws_ld(d_in.N);
        const long long threads = (long long)d_in.M * (ld >> 2);
        const dim3 rgrid((unsigned)((threads + 255) / 256));
        if (c_plain_in) EEG_LAUNCH((x3_splitk_reduce_kernel<true>), rgrid, dim3(256), 0, stream, d_in, (const float*)d_in.workspace, ld);
        else            EEG_LAUNCH((x3_splitk_reduce_kernel<false>), rgrid, dim3(256), 0, stream, d_in, (const float*)d_in.workspace, ld);
    }
    return (int)hipGetLastError();
}

// Problem -> tile shape.  cfg: 0 = 64x64x32, 1 = 64x64x32 double-buffered, 2 = 64x64x64, 3 = 64x64x64 double-buffered,
// 4 = 128x128x32, 5 = 128x128x32 double-buffered (bits 8..15 of desc.precision pin one: tests, tools/bench_gemm_x3.py).
long long gemm_x3_workspace_bytes(const eegclip_gemm_desc& d) { return d.split_k > 1 ? x3_workspace_bytes(d.M, d.N, d.split_k) : 0; }

int launch_gemm_x3(const eegclip_gemm_desc& d, bool akc, bool bkc, bool c_plain, bool k2, void* stream) {
    int cfg = ((d.precision >> 8) & 0xff) - 1;                  // explicit tile configuration in the descriptor (tuning / tests)
    if (cfg < 0) {
        // measured on the shapes of a training step (profiles/r2_gemm_x3_sweep_*.json): 64x64x64 with one LDS image is the best or within 3 % of
        // the best everywhere at K ~ 250 and for the split-K weight gradients; 128x128 tiles only pay once K is long and the grid is large (4096^3:
        // 510 vs 733 us)
        const long long big_tiles = (long long)((d.M + 127) / 128) * ((d.N + 127) / 128);
        cfg = (d.split_k == 1 && big_tiles >= 1024 && d.K >= 1024) ? 5 : 2;
    }
    switch (cfg) {
        case 0: return x3_launch_cfg<64, 32, false>(d, akc, bkc, c_plain, k2, stream);
        case 2: return x3_launch_cfg<64, 64, false>(d, akc, bkc, c_plain, k2, stream);
        case 3: return x3_launch_cfg<64, 64, true>(d, akc, bkc, c_plain, k2, stream);
        case 4: return x3_launch_cfg<128, 32, false>(d, akc, bkc, c_plain, k2, stream);
        case 5: return x3_launch_cfg<128, 32, true>(d, akc, bkc, c_plain, k2, stream);
        default: return x3_launch_cfg<64, 32, true>(d, akc, bkc, c_plain, k2, stream);
    }
}

}  // namespace eeg

using namespace eeg;

extern "C" int eegclip_split_rows(const eegclip_split_item* items, int n, void* stream) {
    if (!items || n < 0 || n > XP_SPLIT_MAX) return EEGCLIP_EINVAL;
    if (n == 0) return 0;
    xp_split_table tb;
    tb.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const eegclip_split_item& it = items[i];
        const int orow = it.transpose ? it.cols : it.rows, ocol = it.transpose ? it.rows : it.cols;
        if (!it.src || !it.hi || !it.lo || it.rows < 1 || it.cols < 1 || it.ld_src < it.cols || it.ld_out < ocol) return EEGCLIP_EINVAL;
        if (it.copy && (it.transpose || it.ld_copy < it.cols)) return EEGCLIP_EINVAL;
        if (it.transpose == 2 && ((it.rows & 3) || (it.ld_out & 3) || ((reinterpret_cast<uintptr_t>(it.hi) | reinterpret_cast<uintptr_t>(it.lo)) & 7u))) return EEGCLIP_EINVAL;
        if (it.transpose < 0 || it.transpose > 2) return EEGCLIP_EINVAL;
        tb.e[i] = xp_split_entry{it.src, static_cast<unsigned short*>(it.hi), static_cast<unsigned short*>(it.lo), it.rows, it.cols, it.ld_src, it.ld_out,
                                 it.transpose, blocks, it.copy, it.copy ? it.ld_copy : 0};
        long long b = ((long long)orow * it.ld_out + 1023) / 1024;          // ~4 elements per thread
        if (b > 256) b = 256;
        blocks += (int)b;
    }
    EEG_LAUNCH(split_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, tb);
    return (int)hipGetLastError();
}
