// Generic strided fp32 GEMM on the f32 matrix cores (v_mfma_f32_16x16x4_f32: exact f32, 157 TF peak).
//
//   C[m,n] (+)= epilogue(alpha * sum_k A[m,k] B[k,n])        -- see include/eegclip.h: eegclip_gemm_desc
//
// Tiling (CDNA4): 64x64 output tile per 256-thread workgroup = 4 wavefronts in a 2x2 grid, each wave owning a
// 32x32 sub-tile as 2x2 MFMA 16x16 accumulators; BK = 32.  Both operand tiles are staged k-major in LDS
// (As[k][m], Bs[k][n], row stride 81 floats): the MFMA operand read "lane l -> (row l&15, k l>>4)" is then a
// ds_read_b32 of 16 consecutive floats per 16-lane group, and 81 = 17 (mod 32) keeps both that read and the
// k-contiguous staging write (lanes walk k) off each other's banks.  Global->register prefetch of tile t+1
// overlaps the MFMAs of tile t.  None of the encoder's dimensions (250, 248, 63, 36, 1440, 2520) is tile
// aligned: every load and store is guarded, padding lives only in LDS (zeros), never in HBM.
#include "eeg_common.h"

namespace eeg {

constexpr int G_BM = 64, G_BN = 64, G_BK = 32, G_LD = 81;
constexpr int G_THREADS = 256;
constexpr int G_EPT = (G_BM * G_BK) / G_THREADS;   // 8 staged elements per thread per operand

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(G_THREADS) void gemm_f32_kernel(const eegclip_gemm_desc d) {
    EEG_LDS_BASE(float, lds);
    float* As = lds;                    // [G_BK][G_LD]
    float* Bs = lds + G_BK * G_LD;      // [G_BK][G_LD]

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.y * G_BM, n0 = blockIdx.x * G_BN;

    // K slice of this workgroup (split-K along blockIdx.z)
    const int nsplit = d.split_k;
    const int ktiles = (d.K + G_BK - 1) / G_BK;
    const int tiles_per = (ktiles + nsplit - 1) / nsplit;
    const int kt_begin = blockIdx.z * tiles_per;
    int kt_end = kt_begin + tiles_per;
    if (kt_end > ktiles) kt_end = ktiles;

    // ---- per-thread staging coordinates -------------------------------------------------------------
    // m-contiguous operand: lane walks m (coalesced), k = (t>>6) + 4 i.   k-contiguous: lane walks k, m = (t>>5) + 8 i.
    int a_row[G_EPT], a_k[G_EPT];
    long long a_off[G_EPT];
    bool a_ok[G_EPT];
    int b_col[G_EPT], b_k[G_EPT];
    long long b_off[G_EPT];
    bool b_ok[G_EPT];
#pragma unroll
    for (int i = 0; i < G_EPT; ++i) {
        if (A_KC) { a_k[i] = t & 31; a_row[i] = (t >> 5) + 8 * i; }
        else      { a_row[i] = t & 63; a_k[i] = (t >> 6) + 4 * i; }
        a_ok[i] = (m0 + a_row[i]) < d.M;
        a_off[i] = a_ok[i] ? dim_off(d.Am, m0 + a_row[i]) : 0;
        if (B_KC) { b_k[i] = t & 31; b_col[i] = (t >> 5) + 8 * i; }
        else      { b_col[i] = t & 63; b_k[i] = (t >> 6) + 4 * i; }
        b_ok[i] = (n0 + b_col[i]) < d.N;
        b_off[i] = b_ok[i] ? dim_off(d.Bn, n0 + b_col[i]) : 0;
    }

    float ra[G_EPT], rb[G_EPT];
    auto load_tile = [&](int kt) {
        const int k0 = kt * G_BK;
#pragma unroll
        for (int i = 0; i < G_EPT; ++i) {
            const int ka = k0 + a_k[i];
            ra[i] = (a_ok[i] && ka < d.K) ? d.A[a_off[i] + dim_off(d.Ak, ka)] : 0.f;
            const int kb = k0 + b_k[i];
            rb[i] = (b_ok[i] && kb < d.K) ? d.B[b_off[i] + dim_off(d.Bk, kb)] : 0.f;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < G_EPT; ++i) {
            As[a_k[i] * G_LD + a_row[i]] = ra[i];
            Bs[b_k[i] * G_LD + b_col[i]] = rb[i];
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (kt_begin < kt_end) load_tile(kt_begin);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        store_tile();
        __syncthreads();
        if (kt + 1 < kt_end) load_tile(kt + 1);     // prefetch into registers under the MFMAs
        const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < G_BK / 4; ++kk) {
            const int kq = kk * 4 + fq;
            const float a0 = As[kq * G_LD + wr * 32 + fr];
            const float a1 = As[kq * G_LD + wr * 32 + 16 + fr];
            const float b0 = Bs[kq * G_LD + wc * 32 + fr];
            const float b1 = Bs[kq * G_LD + wc * 32 + 16 + fr];
            acc[0][0] = mfma_f32_16x16x4(a0, b0, acc[0][0]);
            acc[0][1] = mfma_f32_16x16x4(a0, b1, acc[0][1]);
            acc[1][0] = mfma_f32_16x16x4(a1, b0, acc[1][0]);
            acc[1][1] = mfma_f32_16x16x4(a1, b1, acc[1][1]);
        }
        __syncthreads();
    }

    // ---- epilogue: D[row = (lane>>4)*4 + r][col = lane&15] ------------------------------------------
    const bool first_slice = (blockIdx.z == 0);
    const float keep_scale = d.drop_p > 0.f ? 1.0f / (1.0f - d.drop_p) : 1.0f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = n0 + wc * 32 + nt * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr * 32 + mt * 16 + (lane >> 4) * 4 + r;
                if (m >= d.M || n >= d.N) continue;
                float v = d.alpha * acc[mt][nt][r];
                if (first_slice) {
                    if (d.bias_n) v += d.bias_n[n];
                    if (d.bias_m) v += d.bias_m[m];
                }
                const long long coff = dim_off(d.Cm, m) + dim_off(d.Cn, n);
                if (nsplit > 1) {
                    atomicAdd(d.C + coff, v);
                    continue;
                }
                if (d.Cpre) d.Cpre[coff] = v;
                if (d.act == EEGCLIP_ACT_GELU) v = gelu_erf(v);
                else if (d.act == EEGCLIP_ACT_SILU) v = silu(v);
                if (d.drop_p > 0.f)
                    v = dropout_keep(d.seed, d.drop_site, (unsigned long long)m * (unsigned)d.N + (unsigned)n, d.drop_p) ? v * keep_scale : 0.f;
                if (d.R) v += d.R[dim_off(d.Rm, m) + dim_off(d.Rn, n)];
                if (d.accumulate) v += d.C[coff];
                d.C[coff] = v;
            }
        }
    }
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
    const dim3 grid((d.N + G_BN - 1) / G_BN, (d.M + G_BM - 1) / G_BM, d.split_k);
    const dim3 block(G_THREADS);
    const size_t lds = 2 * G_BK * G_LD * sizeof(float);
    const bool akc = (d.Ak.si == 1), bkc = (d.Bk.si == 1);
    if (akc && bkc)        EEG_LAUNCH((gemm_f32_kernel<true, true>), grid, block, lds, stream, d);
    else if (akc && !bkc)  EEG_LAUNCH((gemm_f32_kernel<true, false>), grid, block, lds, stream, d);
    else if (!akc && bkc)  EEG_LAUNCH((gemm_f32_kernel<false, true>), grid, block, lds, stream, d);
    else                   EEG_LAUNCH((gemm_f32_kernel<false, false>), grid, block, lds, stream, d);
    return (int)hipGetLastError();
}

extern "C" int eegclip_abi_version(void) { return EEGCLIP_ABI_VERSION; }
