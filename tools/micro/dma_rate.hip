// micro-benchmark: how many bytes per second can ONE CU pull from L2 / HBM into LDS (global_load_lds_dwordx4, "LDS-DMA") or into registers
// (global_load_dwordx4), as a function of waves per workgroup and of the bytes each wave keeps in flight?  One workgroup per CU (grid = 256).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/dma_rate.hip -o /tmp/dma_rate && /tmp/dma_rate
// L2 mode: every workgroup re-reads its own 64 KB region (32 workgroups x 64 KB = 2 MB per XCD: L2 resident after the first pass);
// HBM mode: every workgroup streams its own 8 MB region once per launch (2 GB total: no reuse anywhere).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int DEPTH>   // 1 KB DMA instructions in flight per wave
__global__ __launch_bounds__(1024) void k_dma(const unsigned char* __restrict__ src, long long region, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const unsigned char* base = src + (long long)blockIdx.x * region;
    const long long per_it = (long long)nw * 1024;           // bytes the workgroup requests per "instruction round"
    const int rounds = (int)(region / per_it);
    unsigned char* my = lds + wave * DEPTH * 1024;            // the wave's own ring of DEPTH KB
    for (int it = 0; it < iters; ++it) {
        for (int r = 0; r < rounds; ++r) {
            const unsigned char* p = base + (long long)r * per_it + wave * 1024 + lane * 16;
            const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(my + (r % DEPTH) * 1024));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
            if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (DEPTH == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else if (DEPTH == 16) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(31)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = lds[blockIdx.x & 1023];
}

template <int DEPTH>   // 16-byte register loads in flight per lane
__global__ __launch_bounds__(1024) void k_reg(const unsigned char* __restrict__ src, long long region, int iters, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned char* base = src + (long long)blockIdx.x * region;
    const long long per_it = (long long)nw * 1024 * DEPTH;
    const int rounds = (int)(region / per_it);
    u4 acc = u4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        for (int r = 0; r < rounds; ++r) {
            u4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = *reinterpret_cast<const u4*>(base + (long long)r * per_it + (long long)(d * nw + wave) * 1024 + lane * 16);
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
        }
    }
    if (acc[0] == 0x12345678u) sink[blockIdx.x] = acc[1] + acc[2] + acc[3];
}

int main() {
    const int grid = 256;
    unsigned char* buf;
    unsigned* sink;
    const long long total = 2LL << 30;
    hipMalloc(&buf, total);
    hipMalloc(&sink, grid * 4);
    hipMemset(buf, 1, total);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch, long long region, int iters, int threads, int depth) {
        launch(region, iters, threads);      // warm
        hipDeviceSynchronize();
        hipEventRecord(e0);
        launch(region, iters, threads);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)grid * region * iters;
        printf("%-8s %-3s waves/CU %2d  in flight/wave %5d B  -> %7.1f GB/s per CU  (%6.2f TB/s chip, %.3f ms)\n", name, region > (1 << 20) ? "HBM" : "L2", threads / 64,
               depth * 1024, bytes / ms / 1e6 / grid, bytes / ms / 1e9, ms);
    };
#define DMA(D) run("lds-dma", [&](long long rg, int it, int th) { hipLaunchKernelGGL(k_dma<D>, dim3(grid), dim3(th), (th / 64) * D * 1024, 0, buf, rg, it, sink); }, region, iters, threads, D)
#define REG(D) run("regs", [&](long long rg, int it, int th) { hipLaunchKernelGGL(k_reg<D>, dim3(grid), dim3(th), 0, 0, buf, rg, it, sink); }, region, iters, threads, D)
    for (int mode = 0; mode < 2; ++mode) {
        const long long region = mode == 0 ? (64 << 10) : (8 << 20);
        const int iters = mode == 0 ? 400 : 1;
        for (int threads : {256, 512, 1024}) {
            DMA(1); DMA(2); DMA(4); DMA(8); DMA(16);
            if (threads <= 512) { REG(1); REG(2); REG(4); REG(8); }
        }
    }
    return 0;
}
