// probe of ds_read_b64_tr_b16 (gfx950): which 16-bit element of which lane's 8-byte chunk does lane i, element j receive?
// LDS holds element e = e (u16); lane L supplies the address of chunk `chunk_of[L]` (8 bytes = elements 4c .. 4c+3).  Test 1: linear (chunk = L).
// Test 2: a [rows][64] plane with row stride 144 bytes, group g reads rows 4g .. 4g+3, columns 16 .. 31: lane l = L & 15 -> row 4g + (l >> 2),
// chunk 4 + (l & 3); expectation: lane i gets column 16 + i of rows 4g .. 4g+3.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned short u16;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(u16* out, int mode) {
    __shared__ __attribute__((aligned(16))) u16 lds[8192];
    const int L = threadIdx.x;
    for (int i = L; i < 8192; i += 64) lds[i] = (u16)i;
    __syncthreads();
    unsigned addr;
    if (mode == 0) addr = 8 * L;
    else { const int g = L >> 4, l = L & 15; addr = (4 * g + (l >> 2)) * 144 + 2 * (16 + 4 * (l & 3)); }
    u32x2 v;
    const unsigned base = (unsigned)(size_t)lds;          // LDS offset of the array (generic -> local truncation)
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + addr) : "memory");
    out[4 * L + 0] = (u16)(v[0] & 0xffff); out[4 * L + 1] = (u16)(v[0] >> 16);
    out[4 * L + 2] = (u16)(v[1] & 0xffff); out[4 * L + 3] = (u16)(v[1] >> 16);
}
int main() {
    u16 *d, h[256];
    hipMalloc(&d, sizeof(h));
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int ok = 1;
        for (int L = 0; L < 64; ++L)
            for (int j = 0; j < 4; ++j) {
                const int g = L >> 4, i = L & 15;
                // hypothesis: element j of lane i = element (i & 3) of the chunk supplied by lane 16 g + 4 j + (i >> 2)
                const int src = 16 * g + 4 * j + (i >> 2);
                int chunk_elem0;
                if (mode == 0) chunk_elem0 = 4 * src;
                else { const int sg = src >> 4, sl = src & 15; chunk_elem0 = ((4 * sg + (sl >> 2)) * 144 + 2 * (16 + 4 * (sl & 3))) / 2; }
                if (h[4 * L + j] != (u16)(chunk_elem0 + (i & 3))) ok = 0;
            }
        printf("mode %d: hypothesis %s\n", mode, ok ? "HOLDS" : "FAILS");
        if (!ok || mode == 0) {
            for (int L = 0; L < 20; ++L) printf("  lane %2d: %5d %5d %5d %5d\n", L, h[4 * L], h[4 * L + 1], h[4 * L + 2], h[4 * L + 3]);
        }
    }
    return 0;
}
