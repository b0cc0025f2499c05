// InfoNCE logits on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16, 2.5 PFLOP/s dense peak):
//       C[m][n] = s * sum_k A[m][k] * B[n][k]          A (M,K), B (N,K) bf16 row-major, C (M,N) fp32, s = *scale (device scalar)
// = logit_scale * Z_eeg @ Z_img^T of models/loss.py:122-123 for the large-batch configuration (global batch 2048, D = 1024), where
// the fp32-exact GEMM of gemm.hip (157 TF peak) is the wrong tool.  The features are rounded to bf16 once (eegclip_cast_bf16); the
// accumulation and the logits stay fp32 (a bf16 logit of magnitude ~80 would carry an error of 0.3 into the softmax).
//
// Tiling: 128 x 128 output tile per 256-thread workgroup (N = 2048: 256 tiles = one per CU), BK = 64, 4 waves in a 2 x 2 grid, each
// wave 64 x 64 = 4 x 4 MFMA tiles (16 accumulators).  The product is formed TRANSPOSED (MFMA rows = n) so a lane's 4 accumulator registers
// are 4 consecutive columns of one output row: 16-byte stores of C.  (A register-staged two-stage variant was the first version; the LDS-DMA
// pipeline below replaced it in round 2 and the old kernel was removed in round 4.)
#include "eeg_common.h"

#include <stdlib.h>

namespace eeg {

constexpr int LB_T = 128;      // tile edge
constexpr int LB_K = 64;       // k-tile
constexpr int LB_LD = 80;      // LDS row stride in bf16

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- the operand tiles go global -> LDS by LDS-DMA, without touching VGPRs, NS stages deep -------------------------------------------
// Stage image: [row][64 bf16] = 128-byte rows, NO padding (an LDS-DMA instruction deposits wave-uniform base + 16 * lane: 8 rows of 128
// contiguous bytes); bank conflicts are avoided by an XOR swizzle of the 16-byte chunk index with (row & 7), applied on the SOURCE
// address of the DMA and again on the ds_read address (same involution on both sides): the 16 lanes of every ds_read_b128 service group
// then cover all 16 slots of the 256-byte bank row.  Pipeline per k-tile: wait until this wave's DMA of tile kt has landed (counted
// vmcnt: the newer tiles stay in flight), barrier (everybody's part of tile kt is in LDS; everybody is done reading the stage that is
// about to be refilled), issue the DMA of tile kt + NS - 1, then 2 x 16 MFMAs on tile kt.
constexpr int LD_NS = 4;                        // stages: 4 x 32 KB = 128 KB of the 160 KB LDS
constexpr int LD_STAGE = 2 * LB_T * LB_K;       // bf16 elements per stage (A tile + B tile)

__global__ __launch_bounds__(256) void logits_bf16_dma_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B,
                                                               float* __restrict__ C, int M, int N, int K, long long ldc,
                                                               const float* __restrict__ scale, int gx, int ntiles, int chunk) {
    EEG_LDS_BASE(unsigned short, lds);
    const int logical = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (logical >= ntiles) return;
    const int m0 = (logical / gx) * LB_T, n0 = (logical % gx) * LB_T;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, g = lane >> 4;
    // DMA role: wave w fills rows 32w .. 32w+31 of both tiles, instruction q (0..3) = rows 32w + 8q .. +7; lane -> (row, swizzled chunk)
    const int drow = lane >> 3, dpos = lane & 7;
    const unsigned short* asrc[4];
    const unsigned short* bsrc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = 32 * wave + 8 * q + drow;
        const int col = 8 * (dpos ^ (row & 7));
        asrc[q] = A + (long long)(m0 + row) * K + col;
        bsrc[q] = B + (long long)(n0 + row) * K + col;
    }
    auto issue_tile = [&](int kt) {
        unsigned short* st = lds + (kt % LD_NS) * LD_STAGE + (32 * wave) * LB_K;
#pragma unroll
        for (int q = 0; q < 4; ++q) lds_dma16(st + 8 * q * LB_K, asrc[q] + kt * LB_K);
#pragma unroll
        for (int q = 0; q < 4; ++q) lds_dma16(st + LB_T * LB_K + 8 * q * LB_K, bsrc[q] + kt * LB_K);
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ktiles = K / LB_K;
#pragma unroll
    for (int p = 0; p < LD_NS - 1; ++p)
        if (p < ktiles) issue_tile(p);
    const int sw = fr & 7;                       // (row & 7) of every operand row this lane fetches (rows are 16 i + fr)
    for (int kt = 0; kt < ktiles; ++kt) {
        const int newer = ktiles - 1 - kt < LD_NS - 2 ? ktiles - 1 - kt : LD_NS - 2;      // tiles issued after kt that may stay in flight
        if (newer >= 2) wait_vmcnt<16>();
        else if (newer == 1) wait_vmcnt<8>();
        else wait_vmcnt<0>();
        raw_barrier();
        if (kt + LD_NS - 1 < ktiles) issue_tile(kt + LD_NS - 1);
        const unsigned short* as = lds + (kt % LD_NS) * LD_STAGE;
        const unsigned short* bs = as + LB_T * LB_K;
        // operands of BOTH k-steps are requested before the first MFMA: LDS returns in order, so step 0 waits for its own 8 reads only
        // and the reads of step 1 land under the 16 MFMAs of step 0 (issued per step the compiler serialised read -> wait -> MFMA)
        bf16x8 av[2][4], bv[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int pc = 8 * ((4 * s + g) ^ sw);                   // swizzled 16-byte chunk -> bf16 column
#pragma unroll
            for (int i = 0; i < 4; ++i) av[s][i] = *reinterpret_cast<const bf16x8*>(as + (wr * 64 + 16 * i + fr) * LB_K + pc);
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[s][j] = *reinterpret_cast<const bf16x8*>(bs + (wc * 64 + 16 * j + fr) * LB_K + pc);
        }
#if !defined(EEG_EMU)
        __builtin_amdgcn_sched_barrier(0);       // keep the 16 reads ahead of the MFMAs (the scheduler otherwise sinks each read to its use)
#endif
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = mfma_bf16_16x16x32(bv[s][j], av[s][i], acc[j][i]);
    }
    const float s = scale ? *scale : 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wr * 64 + 16 * i + fr;
        float* cr = C + (long long)m * ldc + n0 + wc * 64 + 4 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 v = acc[j][i];
            *reinterpret_cast<f32x4*>(cr + 16 * j) = f32x4{v[0] * s, v[1] * s, v[2] * s, v[3] * s};
        }
    }
}

// fp32 -> bf16 (round to nearest even), 8 elements per thread
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long long n8) {
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n8; q += (long long)gridDim.x * blockDim.x) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + 8 * q), b = *reinterpret_cast<const f32x4*>(x + 8 * q + 4);
        u32x4 o;
        o[0] = f32_to_bf16_bits(a[0]) | ((unsigned)f32_to_bf16_bits(a[1]) << 16);
        o[1] = f32_to_bf16_bits(a[2]) | ((unsigned)f32_to_bf16_bits(a[3]) << 16);
        o[2] = f32_to_bf16_bits(b[0]) | ((unsigned)f32_to_bf16_bits(b[1]) << 16);
        o[3] = f32_to_bf16_bits(b[2]) | ((unsigned)f32_to_bf16_bits(b[3]) << 16);
        *reinterpret_cast<u32x4*>(y + 8 * q) = o;
    }
}

}  // namespace eeg

using namespace eeg;

extern "C" int eegclip_cast_bf16(const float* x, void* y, long long n, void* stream) {
    if (!x || !y || n < 0 || (n & 7)) return EEGCLIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) return EEGCLIP_EALIGN;
    if (n == 0) return 0;
    long long g = (n / 8 + 255) / 256;
    if (g > 4096) g = 4096;
    EEG_LAUNCH(cast_bf16_kernel, dim3((unsigned)g), dim3(256), 0, stream, x, static_cast<unsigned short*>(y), n / 8);
    return (int)hipGetLastError();
}

extern "C" int eegclip_logits_bf16(const void* a, const void* b, float* c, int M, int N, int K, long long ldc, const float* scale, void* stream) {
    if (!a || !b || !c || M < 1 || N < 1 || K < 1 || ldc < N) return EEGCLIP_EINVAL;
    if (M % LB_T || N % LB_T || K % LB_K || (ldc & 3)) return EEGCLIP_EINVAL;          // whole tiles only: ragged sizes take the fp32 GEMM
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15u) return EEGCLIP_EALIGN;
    const int gx = N / LB_T, ntiles = gx * (M / LB_T), chunk = (ntiles + 7) / 8;
    const size_t lds = (size_t)LD_NS * LD_STAGE * sizeof(unsigned short);              // 128 KB
    EEG_LAUNCH(logits_bf16_dma_kernel, dim3(8 * chunk), dim3(256), lds, stream, static_cast<const unsigned short*>(a),
               static_cast<const unsigned short*>(b), c, M, N, K, ldc, scale, gx, ntiles, chunk);
    return (int)hipGetLastError();
}
