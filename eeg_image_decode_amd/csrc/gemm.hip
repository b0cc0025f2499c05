// Generic strided fp32 GEMM on the f32 matrix cores (v_mfma_f32_16x16x4_f32: exact f32, 157 TF peak).
//
//   C[m,n] (+)= epilogue(alpha * sum_k A[m,k] B[k,n])        -- see include/eegclip.h: eegclip_gemm_desc
//
// Tiling (CDNA4): BM x BN output tile (128x128, 128x64 or 64x64, chosen per problem so the grid still fills 256 CUs) per
// 256-thread workgroup = 4 wavefronts in a 2x2 grid, each wave owning (BM/2)x(BN/2) as (BM/32)x(BN/32) MFMA 16x16
// accumulators; BK = 32.  Both operand tiles are staged k-major in LDS (As[k][m ^ swz(k)], Bs[k][n ^ swz(k)], see g_swz):
// the MFMA operand read "lane l -> (row l&15, k l>>4)" is then a ds_read_b32 of 16 consecutive floats per 16-lane group,
// and stride = 17 (mod 32) keeps both that read and the k-contiguous staging write (lanes walk k) off each other's banks.  Global->register prefetch of tile t+1
// overlaps the MFMAs of tile t.  None of the encoder's dimensions (250, 248, 63, 36, 1440, 2520) is tile
// aligned: every load and store is guarded, padding lives only in LDS (zeros), never in HBM.
#include "eeg_common.h"

#include <stdlib.h>

namespace eeg {

constexpr int G_BK = 32;
constexpr int G_THREADS = 256;

// LDS image: element (k, m) of a BK x BT operand tile lives at k*BT + (m ^ swz(k)), swz(k) = ((k&1)<<4) | (k>>1):
//   * MFMA operand read (16 lanes walk m at k, the next 16 at k+1, ds_read_b32 services 32 lanes per cycle over 32 banks): the
//     two 16-float runs land in opposite halves of the 32 banks (bit 4 of swz = k&1)                           -> conflict free
//   * k-contiguous staging write (32 lanes walk k at one m): swz is a bijection of 0..31                         -> conflict free
//   * m-contiguous staging write (lanes walk m at one k): XOR with a constant permutes the 32 banks              -> conflict free
// (a padded stride of 17 mod 32 measured 1.5 conflict cycles per LDS instruction: SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS)
template <int BT> struct g_ld { static constexpr int v = BT; };
__device__ __forceinline__ int g_swz(int k) { return ((k & 1) << 4) | ((k >> 1) & 15); }

// PLAIN = every index map is a plain stride (div = 2^62): offsets are one multiply, and the integer-division path of the two-level
// maps is not even instantiated (it was >1000 instructions of the unrolled staging/epilogue code)
template <bool PLAIN>
__device__ __forceinline__ long long goff(const eegclip_dim& d, int i) {
    if (PLAIN) return (long long)i * d.si;
    return dim_off(d, i);
}
__device__ __forceinline__ bool gemm_dropout_keep(unsigned long long seed, unsigned site, unsigned long long idx, float p) {
    return dropout_keep(seed, site, idx, p);
}

template <int BM, int BN, bool A_KC, bool B_KC, bool PLAIN>
__global__ __launch_bounds__(G_THREADS) void gemm_f32_kernel(const eegclip_gemm_desc d) {
    constexpr int LDA = g_ld<BM>::v, LDB = g_ld<BN>::v;
    constexpr int EA = (BM * G_BK) / G_THREADS, EB = (BN * G_BK) / G_THREADS;   // staged elements per thread
    constexpr int MT = BM / 32, NT = BN / 32;                                    // 16x16 MFMA tiles per wave (2x2 wave grid)
    EEG_LDS_BASE(float, lds);
    float* As = lds;                    // [G_BK][LDA]
    float* Bs = lds + G_BK * LDA;       // [G_BK][LDB]

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    // K slice of this workgroup (split-K along blockIdx.z)
    const int nsplit = d.split_k;
    const int ktiles = (d.K + G_BK - 1) / G_BK;
    const int tiles_per = (ktiles + nsplit - 1) / nsplit;
    const int kt_begin = blockIdx.z * tiles_per;
    int kt_end = kt_begin + tiles_per;
    if (kt_end > ktiles) kt_end = ktiles;

    // ---- per-thread staging coordinates -------------------------------------------------------------
    // m-contiguous operand: lane walks m (coalesced), k = t / BM + (256/BM) i.   k-contiguous: lane walks k, m = (t>>5) + 8 i.
    int a_row[EA], a_k[EA];
    long long a_off[EA];
    bool a_ok[EA];
    int b_col[EB], b_k[EB];
    long long b_off[EB];
    bool b_ok[EB];
#pragma unroll
    for (int i = 0; i < EA; ++i) {
        if (A_KC) { a_k[i] = t & 31; a_row[i] = (t >> 5) + 8 * i; }
        else      { a_row[i] = t % BM; a_k[i] = t / BM + (G_THREADS / BM) * i; }
        a_ok[i] = (m0 + a_row[i]) < d.M;
        a_off[i] = a_ok[i] ? goff<PLAIN>(d.Am, m0 + a_row[i]) : 0;
    }
#pragma unroll
    for (int i = 0; i < EB; ++i) {
        if (B_KC) { b_k[i] = t & 31; b_col[i] = (t >> 5) + 8 * i; }
        else      { b_col[i] = t % BN; b_k[i] = t / BN + (G_THREADS / BN) * i; }
        b_ok[i] = (n0 + b_col[i]) < d.N;
        b_off[i] = b_ok[i] ? goff<PLAIN>(d.Bn, n0 + b_col[i]) : 0;
    }

    float ra[EA], rb[EB];
    auto load_tile = [&](int kt) {
        const int k0 = kt * G_BK;
#pragma unroll
        for (int i = 0; i < EA; ++i) {
            const int ka = k0 + a_k[i];
            ra[i] = (a_ok[i] && ka < d.K) ? d.A[a_off[i] + goff<PLAIN>(d.Ak, ka)] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < EB; ++i) {
            const int kb = k0 + b_k[i];
            rb[i] = (b_ok[i] && kb < d.K) ? d.B[b_off[i] + goff<PLAIN>(d.Bk, kb)] : 0.f;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < EA; ++i) As[a_k[i] * LDA + (a_row[i] ^ g_swz(a_k[i]))] = ra[i];
#pragma unroll
        for (int i = 0; i < EB; ++i) Bs[b_k[i] * LDB + (b_col[i] ^ g_swz(b_k[i]))] = rb[i];
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // (a double-buffered LDS variant with one barrier per k-tile measured 10-25 % SLOWER on MI355X: the second stage halves the
    //  workgroups per CU and occupancy, not barrier count, is what hides latency for these short-K problems)
    if (kt_begin < kt_end) load_tile(kt_begin);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        store_tile();
        __syncthreads();
        if (kt + 1 < kt_end) load_tile(kt + 1);     // global -> registers, in flight under the MFMAs of tile kt
        const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < G_BK / 4; ++kk) {
            const int kq = kk * 4 + fq;
            const int sw = g_swz(kq);
            float av[MT], bv[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) av[i] = As[kq * LDA + ((wr * (BM / 2) + 16 * i + fr) ^ sw)];
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = Bs[kq * LDB + ((wc * (BN / 2) + 16 * j + fr) ^ sw)];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma_f32_16x16x4(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }

    // ---- epilogue: D[row = (lane>>4)*4 + r][col = lane&15] ------------------------------------------
    const bool first_slice = (blockIdx.z == 0);
    const float keep_scale = d.drop_p > 0.f ? 1.0f / (1.0f - d.drop_p) : 1.0f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = n0 + wc * (BN / 2) + nt * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr * (BM / 2) + mt * 16 + (lane >> 4) * 4 + r;
                if (m >= d.M || n >= d.N) continue;
                float v = d.alpha * acc[mt][nt][r];
                if (first_slice) {
                    if (d.bias_n) v += d.bias_n[n];
                    if (d.bias_m) v += d.bias_m[m];
                }
                const long long coff = goff<PLAIN>(d.Cm, m) + goff<PLAIN>(d.Cn, n);
                if (nsplit > 1) {
                    atomicAdd(d.C + coff, v);
                    continue;
                }
                if (d.Cpre) d.Cpre[coff] = v;
                if (d.act == EEGCLIP_ACT_GELU) v = gelu_erf(v);
                else if (d.act == EEGCLIP_ACT_SILU) v = silu(v);
                if (d.drop_p > 0.f)
                    v = gemm_dropout_keep(d.seed, d.drop_site, (unsigned long long)m * (unsigned)d.N + (unsigned)n, d.drop_p) ? v * keep_scale : 0.f;
                if (d.R) v += d.R[goff<PLAIN>(d.Rm, m) + goff<PLAIN>(d.Rn, n)];
                if (d.accumulate) v += d.C[coff];
                d.C[coff] = v;
            }
        }
    }
}

template <int BM, int BN>
static int launch_gemm(const eegclip_gemm_desc& d, void* stream) {
    const dim3 grid((d.N + BN - 1) / BN, (d.M + BM - 1) / BM, d.split_k);
    const dim3 block(G_THREADS);
    const size_t lds = G_BK * (g_ld<BM>::v + g_ld<BN>::v) * sizeof(float);
    const bool akc = (d.Ak.si == 1), bkc = (d.Bk.si == 1);
    const long long big = 1LL << 40;
    const bool plain = d.Am.div > big && d.Ak.div > big && d.Bk.div > big && d.Bn.div > big && d.Cm.div > big && d.Cn.div > big &&
                       (!d.R || (d.Rm.div > big && d.Rn.div > big));
#define EEG_GEMM_GO(AK, BK_)                                                                                         \
    do {                                                                                                             \
        if (plain) EEG_LAUNCH((gemm_f32_kernel<BM, BN, AK, BK_, true>), grid, block, lds, stream, d);               \
        else       EEG_LAUNCH((gemm_f32_kernel<BM, BN, AK, BK_, false>), grid, block, lds, stream, d);              \
    } while (0)
    if (akc && bkc)        EEG_GEMM_GO(true, true);
    else if (akc && !bkc)  EEG_GEMM_GO(true, false);
    else if (!akc && bkc)  EEG_GEMM_GO(false, true);
    else                   EEG_GEMM_GO(false, false);
#undef EEG_GEMM_GO
    return (int)hipGetLastError();
}

}  // namespace eeg

extern "C" int eegclip_gemm_f32(const eegclip_gemm_desc* dp, void* stream) {
    using namespace eeg;
    if (!dp) return EEGCLIP_EINVAL;
    const eegclip_gemm_desc d = *dp;
    if (d.M < 0 || d.N < 0 || d.K < 0 || !d.C) return EEGCLIP_EINVAL;
    if (d.M == 0 || d.N == 0) return 0;
    if (d.K > 0 && (!d.A || !d.B)) return EEGCLIP_EINVAL;
    if (d.split_k < 1) return EEGCLIP_EINVAL;
    if (d.split_k > 1 && (d.act != EEGCLIP_ACT_NONE || d.drop_p > 0.f || d.R || d.Cpre)) return EEGCLIP_EINVAL;
    if (d.drop_p < 0.f || d.drop_p >= 1.f || d.act < 0 || d.act > EEGCLIP_ACT_SILU) return EEGCLIP_EINVAL;
    if (d.Am.div <= 0 || d.Ak.div <= 0 || d.Bk.div <= 0 || d.Bn.div <= 0 || d.Cm.div <= 0 || d.Cn.div <= 0) return EEGCLIP_EINVAL;
    if (d.R && (d.Rm.div <= 0 || d.Rn.div <= 0)) return EEGCLIP_EINVAL;
    // tile choice: 128x128 (4x4 MFMA tiles per wave, 4 MFMAs per LDS operand read) once the grid still fills the 256 CUs;
    // 128x64 for narrow N; 64x64 for small problems where occupancy matters more than per-wave reuse.
    const long long b128 = (long long)((d.M + 127) / 128) * ((d.N + 127) / 128) * d.split_k;
    const long long b12864 = (long long)((d.M + 127) / 128) * ((d.N + 63) / 64) * d.split_k;
    static const int force = getenv("EEGCLIP_GEMM_TILE") ? atoi(getenv("EEGCLIP_GEMM_TILE")) : 0;   // tuning aid: 64 | 12864 | 128
    static const long long thr = getenv("EEGCLIP_GEMM_THR") ? atoll(getenv("EEGCLIP_GEMM_THR")) : (1LL << 40);   // measured on MI355X: the 64x64 tile wins on every shape of this path (occupancy > reuse)
    if (force == 128) return launch_gemm<128, 128>(d, stream);
    if (force == 12864) return launch_gemm<128, 64>(d, stream);
    if (force == 64) return launch_gemm<64, 64>(d, stream);
    if (b128 >= thr && d.N > 64) return launch_gemm<128, 128>(d, stream);
    if (b12864 >= thr && d.M > 64) return launch_gemm<128, 64>(d, stream);
    return launch_gemm<64, 64>(d, stream);
}

extern "C" int eegclip_abi_version(void) { return EEGCLIP_ABI_VERSION; }
