// hipemu -- a tiny lane-accurate CPU emulator for the HIP kernels in eeg_image_decode_amd/csrc.
//
// TEST INFRASTRUCTURE ONLY.  The build container has no GPU, so the kernel *sources* (the very same
// .hip files that hipcc compiles for gfx950) are also compiled for x86 with -DEEG_EMU and run here
// under a cooperative fiber scheduler: one fiber per work-item, 64-lane wavefronts, __syncthreads,
// wave shuffles and MFMA (16x16x4 f32 / 16x16x32 bf16, gfx950 fragment layouts) are emulated exactly.
// It checks indexing, tiling, guards, LDS layouts and epilogues before a kernel ever reaches the
// MI355X.  It is never loaded by the product package (eeg_image_decode_amd/_lib.py only loads the
// hipcc-built library and raises if it is missing).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
#define hipMemcpyDeviceToDevice 3

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }

namespace hipemu {

struct Block;
struct Lane {
    void* sp;          // saved stack pointer (fiber)
    char* stack;
    uint3_emu tid;
    int lin;           // linear thread id in block
    int wave, lane;
    bool done;
    Block* blk;
};
struct WaveState {
    int nlanes, alive, arrived;
    unsigned gen;
    alignas(16) unsigned char xbuf[2][64][64];   // double-buffered 64-byte slot per lane
};
struct Block {
    uint3_emu bid, bdim, gdim;
    char* smem;
    int nthreads, alive, arrived;
    unsigned gen;
    unsigned long progress;
    const std::function<void()>* body;
    WaveState waves[16];
};

extern thread_local Lane* cur;

void syncthreads();
void wave_sync();
// every lane deposits `bytes` (<=64) and receives a pointer to the 64 slots (valid until the next wave op)
const unsigned char (*wave_allgather(const void* in, size_t bytes))[64];
void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body);
inline char* smem() { return cur->blk->smem; }

template <class T>
inline T shfl_idx(T v, int src) {
    static_assert(sizeof(T) <= 64, "slot");
    auto all = wave_allgather(&v, sizeof(T));
    T r;
    memcpy(&r, all[src & 63], sizeof(T));
    return r;
}

float atomic_add(float* p, float v);
double atomic_add(double* p, double v);
int atomic_add(int* p, int v);
unsigned atomic_add(unsigned* p, unsigned v);
float atomic_max(float* p, float v);

}  // namespace hipemu

#define threadIdx (hipemu::cur->tid)
#define blockIdx (hipemu::cur->blk->bid)
#define blockDim (hipemu::cur->blk->bdim)
#define gridDim (hipemu::cur->blk->gdim)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __syncthreads() hipemu::syncthreads()

template <class T> inline T __shfl_xor(T v, int mask, int = 64) { return hipemu::shfl_idx(v, hipemu::cur->lane ^ mask); }
template <class T> inline T __shfl_down(T v, int d, int = 64) { int s = hipemu::cur->lane + d; return hipemu::shfl_idx(v, s < 64 ? s : hipemu::cur->lane); }
template <class T> inline T __shfl(T v, int src, int = 64) { return hipemu::shfl_idx(v, src); }
template <class T> inline T __shfl_up(T v, int d, int = 64) { int s = hipemu::cur->lane - d; return hipemu::shfl_idx(v, s >= 0 ? s : hipemu::cur->lane); }
inline float atomicAdd(float* p, float v) { return hipemu::atomic_add(p, v); }
inline double atomicAdd(double* p, double v) { return hipemu::atomic_add(p, v); }
inline int atomicAdd(int* p, int v) { return hipemu::atomic_add(p, v); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return hipemu::atomic_add(p, v); }
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline float __expf(float x) { return expf(x); }
inline float __logf(float x) { return logf(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
