// micro-benchmark: sustained v_mfma_f32_16x16x4_f32 rate vs independent accumulators per wave and waves per SIMD (no memory traffic)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0 + blockIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu, float* out) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, 10, 1.f, 2.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 8 * NACC * 2048.0;
    printf("nacc %2d  waves/SIMD %d : %8.3f ms  %7.1f TF/s\n", NACC, blocks_per_cu, ms, flops / ms / 1e9);
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w = 1; w <= 8; w *= 2) { run<1>(w, out); run<2>(w, out); run<4>(w, out); run<8>(w, out); run<16>(w, out); }
    return 0;
}
