// micro-benchmark: WHERE does a k-tile of the LDS-DMA tile pipeline (csrc/infonce_fused.hip: 128 x 128 logits tile, one bf16 plane per operand, 64-k tiles, 4 stages,
// 4 waves) spend its time?  Wave 0 of every workgroup stamps s_memtime at: loop top | after the counted vmcnt wait | after the barrier | after the first fragment
// set has arrived | after the last MFMA of the tile.  Prints the mean segment lengths (shader cycles) for a full grid (256 workgroups) and for ONE workgroup.
//   hipcc -O3 --offload-arch=gfx950 -I eeg_image_decode_amd/csrc tools/micro/tile_chain.hip -o tools/micro/tile_chain
#include "eeg_common.h"
#include <stdio.h>
#include <vector>
using namespace eeg;
constexpr int TM = 128, BK = 64, ROWB = 128, NCH = 8, RPI = 8, NS = 4, D = 1024;
constexpr int TILE_B = TM * ROWB, STAGE_B = 2 * TILE_B, IPT = TM / RPI / 4, DPT = 2 * IPT;
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

template <int VAR>   // 0: as the product kernel; 1: no MFMAs; 2: no fragment reads (and no MFMAs); 3: no DMA refill (tiles re-read from whatever landed)
__global__ __launch_bounds__(256) void k(const unsigned short* q, const unsigned short* kk, float* out, unsigned long long* stamps, int tiles_k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), wq = wave >> 1, wk = wave & 1, r32 = lane & 31, h = lane >> 5;
    const int q0 = ((int)blockIdx.x / tiles_k) * TM, k0r = ((int)blockIdx.x % tiles_k) * TM;
    const int drow = lane / NCH, dpos = lane % NCH;
    const unsigned short* src[2][IPT];
    for (int i = 0; i < IPT; ++i) {
        const int row = wave * (TM / 4) + RPI * i + drow, col = 8 * (dpos ^ swz(row));
        src[0][i] = q + (long long)(q0 + row) * D + col;
        src[1][i] = kk + (long long)(k0r + row) * D + col;
    }
    auto issue_one = [&](int kt, int dnum) {
        const int o = dnum / IPT, i = dnum % IPT;
        lds_dma16(lds + (kt % NS) * STAGE_B + wave * (TM / 4) * ROWB + o * TILE_B + RPI * i * ROWB, src[o][i] + kt * BK);
    };
    f32x16 acc[2][2];
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
    int foq[4][2], fok[4][2];
    for (int s = 0; s < 4; ++s) for (int i = 0; i < 2; ++i) {
        const int rq = wq * 64 + 32 * i + r32, rk = wk * 64 + 32 * i + r32;
        foq[s][i] = rq * ROWB + (((2 * s + h) ^ swz(rq)) & 7) * 16;
        fok[s][i] = TILE_B + rk * ROWB + (((2 * s + h) ^ swz(rk)) & 7) * 16;
    }
    const int ktiles = D / BK;
    unsigned long long seg[5] = {0, 0, 0, 0, 0};
    for (int p = 0; p < NS - 1; ++p) for (int d = 0; d < DPT; ++d) issue_one(p, d);
    for (int kt = 0; kt < ktiles; ++kt) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const int newer = ktiles - 1 - kt < NS - 2 ? ktiles - 1 - kt : NS - 2;
        if (newer >= 2) wait_vmcnt<2 * DPT>(); else if (newer == 1) wait_vmcnt<DPT>(); else wait_vmcnt<0>();
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        raw_barrier();
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        const bool refill = VAR != 3 && kt + NS - 1 < ktiles;
        const unsigned char* st = lds + (kt % NS) * STAGE_B;
        bf16x8 qh[2][2], kh[2][2];
        auto read_step = [&](int s, int set) {
            for (int i = 0; i < 2; ++i) {
                qh[set][i] = *reinterpret_cast<const bf16x8*>(st + foq[s][i]);
                kh[set][i] = *reinterpret_cast<const bf16x8*>(st + fok[s][i]);
            }
        };
        if (VAR != 2) read_step(0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t3 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (VAR != 2 && s + 1 < 4) read_step(s + 1, (s + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const int set = s & 1;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int m0 = s * 4 + j * 2 + i;
                    if (VAR == 0 || VAR == 3) acc[j][i] = mfma_bf16_32x32x16(kh[set][j], qh[set][i], acc[j][i]);
                    if (refill && ((m0 + 1) * DPT) / 16 > (m0 * DPT) / 16) issue_one(kt + NS - 1, (m0 * DPT) / 16);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        // (the accumulators are read here so that "end of the tile" includes the matrix pipe draining)
        asm volatile("" : "+v"(acc[1][1]));
        const unsigned long long t4 = __builtin_amdgcn_s_memtime();
        seg[0] += t1 - t0; seg[1] += t2 - t1; seg[2] += t3 - t2; seg[3] += t4 - t3; seg[4] += 1;
    }
    float ssum = 0.f;
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) ssum += acc[j][i][e];
    out[blockIdx.x * 256 + t] = ssum;
    if (t == 0) for (int i = 0; i < 4; ++i) stamps[blockIdx.x * 4 + i] = seg[i];
}

// ---- wave-specialised form: NC consumer waves (fragment reads + MFMAs only) and NPRD producer waves (LDS-DMA issue + counted vmcnt wait only), one barrier per k-tile
template <int NC, int NPRD>
__global__ __launch_bounds__(64 * (NC + NPRD)) void kspec(const unsigned short* q, const unsigned short* kk, float* out, unsigned long long* stamps, int tiles_k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), r32 = lane & 31, h = lane >> 5;
    const int q0 = ((int)blockIdx.x / tiles_k) * TM, k0r = ((int)blockIdx.x % tiles_k) * TM;
    const int ktiles = D / BK;
    if (wave >= NC) {                                        // ---------------- producer
        constexpr int PI = 32 / NPRD;                         // DMA instructions per producer and k-tile (2 operands x 16 instructions of 8 rows)
        const int pw = wave - NC, drow = lane / NCH, dpos = lane % NCH;
        const unsigned short* src[PI];
        int dst[PI];
        for (int i = 0; i < PI; ++i) {
            const int inst = pw + NPRD * i, o = inst / 16, row = RPI * (inst % 16) + drow, col = 8 * (dpos ^ swz(row));
            src[i] = (o ? kk + (long long)(k0r + row) * D : q + (long long)(q0 + row) * D) + col;
            dst[i] = o * TILE_B + RPI * (inst % 16) * ROWB;
        }
        auto issue_tile = [&](int kt) {
#pragma unroll
            for (int i = 0; i < PI; ++i) lds_dma16(lds + (kt % NS) * STAGE_B + dst[i], src[i] + kt * BK);
        };
        for (int p = 0; p < NS - 1; ++p) issue_tile(p);
        for (int kt = 0; kt < ktiles; ++kt) {
            const int newer = ktiles - 1 - kt < NS - 2 ? ktiles - 1 - kt : NS - 2;
            if (newer >= 2) wait_vmcnt<2 * PI>(); else if (newer == 1) wait_vmcnt<PI>(); else wait_vmcnt<0>();
            raw_barrier();
            if (kt + NS - 1 < ktiles) issue_tile(kt + NS - 1);
        }
        return;
    }
    constexpr int NWK = NC / 2, WT = 2, WTK = TM / (32 * NWK);
    const int wq = wave / NWK, wk = wave % NWK;
    f32x16 acc[WTK][WT];
    for (int j = 0; j < WTK; ++j) for (int i = 0; i < WT; ++i) for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
    int foq[4][WT], fok[4][WTK];
    for (int s = 0; s < 4; ++s) {
        for (int i = 0; i < WT; ++i) { const int rq = wq * 64 + 32 * i + r32; foq[s][i] = rq * ROWB + (((2 * s + h) ^ swz(rq)) & 7) * 16; }
        for (int j = 0; j < WTK; ++j) { const int rk = wk * (TM / NWK) + 32 * j + r32; fok[s][j] = TILE_B + rk * ROWB + (((2 * s + h) ^ swz(rk)) & 7) * 16; }
    }
    unsigned long long seg[4] = {0, 0, 0, 0};
    for (int kt = 0; kt < ktiles; ++kt) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        raw_barrier();
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        const unsigned char* st = lds + (kt % NS) * STAGE_B;
        bf16x8 qh[2][WT], kh[2][WTK];
        auto read_step = [&](int s, int set) {
            for (int i = 0; i < WT; ++i) qh[set][i] = *reinterpret_cast<const bf16x8*>(st + foq[s][i]);
            for (int j = 0; j < WTK; ++j) kh[set][j] = *reinterpret_cast<const bf16x8*>(st + fok[s][j]);
        };
        read_step(0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t3 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s + 1 < 4) read_step(s + 1, (s + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const int set = s & 1;
#pragma unroll
            for (int j = 0; j < WTK; ++j)
#pragma unroll
                for (int i = 0; i < WT; ++i) acc[j][i] = mfma_bf16_32x32x16(kh[set][j], qh[set][i], acc[j][i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        const unsigned long long t4 = __builtin_amdgcn_s_memtime();
        seg[1] += t2 - t1; seg[2] += t3 - t2; seg[3] += t4 - t3;
    }
    float ssum = 0.f;
    for (int j = 0; j < WTK; ++j) for (int i = 0; i < WT; ++i) for (int e = 0; e < 16; ++e) ssum += acc[j][i][e];
    out[blockIdx.x * 64 * NC + t] = ssum;
    if (t == 0) for (int i = 0; i < 4; ++i) stamps[blockIdx.x * 4 + i] = seg[i];
}

int main() {
    const int N = 2048;
    unsigned short *q, *kk;
    float* out;
    unsigned long long* st;
    hipMalloc(&q, (size_t)N * D * 2); hipMalloc(&kk, (size_t)N * D * 2); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&st, 256 * 4 * 8);
    hipMemset(q, 0x3c, (size_t)N * D * 2); hipMemset(kk, 0x3c, (size_t)N * D * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch, int grid) {
        for (int w = 0; w < 3; ++w) launch(grid);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 20; ++r) launch(grid);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(grid * 4);
        hipMemcpy(h.data(), st, grid * 4 * 8, hipMemcpyDeviceToHost);
        double s[4] = {0, 0, 0, 0};
        for (int b = 0; b < grid; ++b) for (int i = 0; i < 4; ++i) s[i] += (double)h[b * 4 + i];
        const double nt = (double)grid * 16;
        printf("%-34s grid %3d: %6.2f us per launch | per k-tile (cycles, wave 0): vmcnt wait %5.0f | barrier %5.0f | first fragments %5.0f | steps (reads + MFMAs + DMA issue) %5.0f | total %5.0f\n",
               name, grid, ms / 20 * 1e3, s[0] / nt, s[1] / nt, s[2] / nt, s[3] / nt, (s[0] + s[1] + s[2] + s[3]) / nt);
    };
#define GO(V, name) for (int grid : {256, 1}) run(name, [&](int g) { hipLaunchKernelGGL(k<V>, dim3(g), dim3(256), NS * STAGE_B, 0, q, kk, out, st, 16); }, grid)
    GO(0, "product loop");
    GO(1, "no MFMAs");
    GO(2, "no fragment reads, no MFMAs");
    GO(3, "no DMA refill");
#define GOS(NC_, NP_, name) for (int grid : {256, 1}) run(name, [&](int g) { hipLaunchKernelGGL((kspec<NC_, NP_>), dim3(g), dim3(64 * (NC_ + NP_)), NS * STAGE_B, 0, q, kk, out, st, 16); }, grid)
    GOS(4, 2, "4 consumers + 2 producers");
    GOS(4, 4, "4 consumers + 4 producers");
    GOS(8, 2, "8 consumers + 2 producers");
    GOS(8, 4, "8 consumers + 4 producers");
    return 0;
}
