// micro-benchmark / experiment: C[M][N] = A[M][K] * B[N][K]^T (both operands k-contiguous, fp32, v_mfma_f32_16x16x4_f32) WITHOUT LDS --
// every wave feeds its MFMAs straight from 8-byte global loads (lane (row fr, group g) holds X[row][k + 2g .. 2g+1]: MFMA steps e = 0, 1
// contract k + 2g + e over the four lane groups), no barriers, register double buffer.  Question: with no workgroup-wide phases to
// lock-step, does the matrix pipe get busier than the LDS-tiled kernel's 48 % on the encoder's 16384 x 744 x 250?
//   hipcc -O3 --offload-arch=gfx950 tools/micro/gemm_direct.hip -o tools/micro/gemm_direct && tools/micro/gemm_direct
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// wave tile = (16 WM) x (16 WN); workgroup = 4 waves as 2 x 2
template <int WM, int WN>
__global__ __launch_bounds__(256) void gemm_direct(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N,
                                                   int K, int gx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int b = blockIdx.x;
    const int chunk = (gridDim.x + 7) / 8;
    const int logical = (b & 7) * chunk + (b >> 3);                 // XCD-contiguous tile order, n fastest
    const int tm = logical / gx, tn = logical % gx;
    if (tm * 32 * WM >= M) return;
    const int m0 = tm * 32 * WM + (wave >> 1) * 16 * WM, n0 = tn * 32 * WN + (wave & 1) * 16 * WN;
    const float* ap[WM];
    const float* bp[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) { int m = m0 + 16 * i + fr; m = m < M ? m : M - 1; ap[i] = A + (long long)m * K + 2 * g; }
#pragma unroll
    for (int j = 0; j < WN; ++j) { int n = n0 + 16 * j + fr; n = n < N ? n : N - 1; bp[j] = B + (long long)n * K + 2 * g; }
    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x2 ra[2][WM][4], rb[2][WN][4];
    auto load = [&](int k0, int s) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = k0 + 8 * c;
            const bool ok = k + 2 * g < K;
            const int kk = ok ? k : 0;
#pragma unroll
            for (int i = 0; i < WM; ++i) { f32x2 v = *reinterpret_cast<const f32x2*>(ap[i] + kk); ra[s][i][c] = ok ? v : f32x2{0.f, 0.f}; }
#pragma unroll
            for (int j = 0; j < WN; ++j) { f32x2 v = *reinterpret_cast<const f32x2*>(bp[j] + kk); rb[s][j][c] = ok ? v : f32x2{0.f, 0.f}; }
        }
    };
    auto mma = [&](int s) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[s][i][c][e], rb[s][j][c][e], acc[i][j], 0, 0, 0);
    };
    load(0, 0);
    int k0 = 0;
    for (; k0 + 64 < K + 32; k0 += 64) {          // two 32-k steps per trip, stages alternate statically
        if (k0 + 32 < K) load(k0 + 32, 1);
        mma(0);
        if (k0 + 32 >= K) break;
        if (k0 + 64 < K) load(k0 + 64, 0);
        mma(1);
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 16 * i + 4 * g + r, n = n0 + 16 * j + fr;
                if (m < M && n < N) C[(long long)m * N + n] = acc[i][j][r];
            }
}

template <int WM, int WN>
void run(const float* A, const float* B, float* C, int M, int N, int K, const std::vector<float>& hA, const std::vector<float>& hB) {
    const int gx = (N + 32 * WN - 1) / (32 * WN), gy = (M + 32 * WM - 1) / (32 * WM);
    const int tiles = gx * gy, grid = 8 * ((tiles + 7) / 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_direct<WM, WN>), dim3(grid), dim3(256), 0, 0, A, B, C, M, N, K, gx);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm_direct<WM, WN>), dim3(grid), dim3(256), 0, 0, A, B, C, M, N, K, gx);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    std::vector<float> hC((size_t)M * N);
    hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int t = 0; t < 200; ++t) {
        const int m = (t * 7919) % M, n = (t * 104729) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k];
        maxerr = fmax(maxerr, fabs(ref - hC[(size_t)m * N + n]));
    }
    printf("%dx%dx%d wave tile %dx%d: %7.1f us %6.1f TF/s  max err %.2e\n", M, N, K, 16 * WM, 16 * WN, ms * 1e3, 2.0 * M * N * K / ms / 1e9, maxerr);
}

int main() {
    const int shapes[][3] = {{16384, 744, 250}, {16384, 250, 250}, {16384, 256, 250}, {16384, 250, 744}};
    for (auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
        for (auto& v : hA) v = (float)(rand() % 2001 - 1000) * 1e-3f;
        for (auto& v : hB) v = (float)(rand() % 2001 - 1000) * 1e-3f;
        float *A, *B, *C;
        hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, (size_t)M * N * 4);
        hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
        run<2, 2>(A, B, C, M, N, K, hA, hB);
        run<4, 2>(A, B, C, M, N, K, hA, hB);
        run<2, 4>(A, B, C, M, N, K, hA, hB);
        hipFree(A); hipFree(B); hipFree(C);
    }
    return 0;
}
