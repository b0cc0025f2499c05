// Common device helpers for the eegclip HIP kernels (gfx950 / CDNA4: 64-lane wavefronts, MFMA, LDS).
//
// The kernels are written against a handful of thin wrappers (LDS base, launch, wave shuffles, MFMA)
// so that the same source also builds under the test-only lane emulator (tests/hipemu, -DEEG_EMU).
// Product builds (hipcc --offload-arch=gfx950) never define EEG_EMU.
#pragma once

#if defined(EEG_EMU)
#include "hipemu.h"
#else
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#endif

#include <stdint.h>

#include "../../include/eegclip.h"

namespace eeg {

#if !defined(EEG_EMU)
struct launch_events {
    hipEvent_t start, stop;
    int launches;                         // kernels launched by this thread since the field was last cleared (csrc/plan_exec.hip: fork events)
};
inline launch_events& tls_launch_events() {
    static thread_local launch_events e{nullptr, nullptr, 0};
    return e;
}
#endif

constexpr int kWave = 64;   // CDNA wavefront width -- hard-coded on purpose (guide section 1)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#if defined(EEG_EMU)
#define EEG_LDS_BASE(T, name) T* name = reinterpret_cast<T*>(hipemu::smem())
#define EEG_LAUNCH(kern, grid, block, smem, stream, ...) \
    hipemu::launch((grid), (block), (smem), [&]() { kern(__VA_ARGS__); })
#else
// one dynamic LDS region per kernel, 16-byte aligned (guide G17); never mix with static __shared__
#define EEG_LDS_BASE(T, name)                                                         \
    extern __shared__ __attribute__((aligned(16))) unsigned char eeg_lds_raw_[];      \
    T* name = reinterpret_cast<T*>(eeg_lds_raw_)
// eegclip_time_next_launch(start, stop) arms a pair of events for the NEXT kernel this thread launches: the launch then goes through
// hipExtLaunchKernelGGL, which stamps the events with the kernel's own begin / end timestamps (what rocprofv3 reports) -- no marker packets
// around the kernel, so neither the dispatch bubbles of an event bracket (~5 us per launch) nor their serialisation end up in the measurement.
// A stop event alone (csrc/plan_exec.hip) is BOUND to the kernel's own completion: the fork point of a launch plan without a marker packet on the main queue.
#define EEG_LAUNCH(kern, grid, block, smem, stream, ...)                                                                                  \
    do {                                                                                                                                 \
        eeg::launch_events& ev_ = eeg::tls_launch_events();                                                                              \
        ++ev_.launches;                                                                                                                  \
        if (ev_.start || ev_.stop) {                                                                                                     \
            hipExtLaunchKernelGGL(kern, (grid), (block), (smem), (hipStream_t)(stream), ev_.start, ev_.stop, 0, __VA_ARGS__);            \
            ev_.start = ev_.stop = nullptr;                                                                                              \
        } else {                                                                                                                         \
            hipLaunchKernelGGL(kern, (grid), (block), (smem), (hipStream_t)(stream), __VA_ARGS__);                                       \
        }                                                                                                                                \
    } while (0)
#endif

// sum of the partial slabs of a K-parallel GEMM (csrc/head_gemm.hip) at float offset i, 4 consecutive floats, in SLICE ORDER: p[0] + p[1] + ... -- the loads
// of up to 8 slabs issued before the first add (as a runtime-length `v += slab[s][i]` loop every slab waits for the previous one's L2 round trip: the
// compiler does not pipeline loads across iterations of a loop it cannot unroll).  Slabs past nslabs re-read slab 0 (a cache hit) and are not added.
__device__ __forceinline__ f32x4 slab_sum4_inflight(const float* __restrict__ slabs, int nslabs, long long stride, long long i) {
    f32x4 v;
    for (int s0 = 0; s0 < nslabs; s0 += 8) {
        f32x4 p[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) p[k] = *reinterpret_cast<const f32x4*>(slabs + (long long)(s0 + k < nslabs ? s0 + k : 0) * stride + i);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (s0 + k < nslabs) {
                if (s0 + k == 0) v = p[k];
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += p[k][e];
                }
            }
    }
    return v;
}

// torch.optim.AdamW single-step math of ONE element (decoupled weight decay, bias correction; bc1 = 1 - beta1^t, bc2s = sqrt(1 - beta2^t) from the host
// in double): shared by csrc/elementwise.hip (adamw_kernel) and csrc/wgrad_tok.hip (the slab reduction that steps the optimizer) -- one expression tree.
__device__ __forceinline__ void adamw_element(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, float gi, float lr, float b1, float b2,
                                              float eps, float wd, float step, float bc2s) {
    float pi = *p;
    pi *= (1.f - lr * wd);
    const float mi = b1 * *m + (1.f - b1) * gi;
    const float vi = b2 * *v + (1.f - b2) * gi * gi;
    *m = mi;
    *v = vi;
    const float denom = sqrtf(vi) / bc2s + eps;
    *p = pi - step * (mi / denom);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// all lanes of the calling wave have executed everything before this point (LDS traffic included) before any lane goes on.  On the
// hardware a wavefront IS in lockstep and LDS operations of one wave retire in order; what has to be pinned is the COMPILER, which
// otherwise moves a lane's LDS load above its own earlier store to a provably different address (the value another lane wrote
// there is invisible to single-thread alias analysis): wavefront-scope release / acquire fences around the scheduling barrier.
__device__ __forceinline__ void wave_sync() {
#if defined(EEG_EMU)
    hipemu::wave_sync();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// LDS-DMA: every lane copies 16 bytes global -> LDS without passing through VGPRs; the destination is wave-uniform base + 16 * lane
// (NOT a per-lane scatter), the source address is per lane.  Asynchronous: tracked by vmcnt; a ds_read may only follow a counted
// s_waitcnt vmcnt AND a workgroup barrier (MI355X_MICROARCH.md, LDS-DMA ordering).  Inline asm on purpose: with the builtin the compiler
// drains vmcnt(0) at every barrier / LDS read, which serialises the pipeline this is used for.
__device__ __forceinline__ void lds_dma16(void* lds_wave_base, const void* gsrc) {
#if defined(EEG_EMU)
    memcpy(static_cast<char*>(lds_wave_base) + 16 * (threadIdx.x & 63), gsrc, 16);
#else
    // LDS byte address = low 32 bits of the generic pointer; wave-uniform by contract, pinned into an SGPR for M0
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
#endif
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {      // at most N vector-memory operations of this wave still in flight
#if !defined(EEG_EMU)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
// workgroup barrier WITHOUT the memory waits __syncthreads() implies (LDS-DMA pipelines count their own vmcnt)
__device__ __forceinline__ void raw_barrier() {
#if defined(EEG_EMU)
    hipemu::syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// inclusive prefix sum over the 64 lanes of a wave
__device__ __forceinline__ float wave_inclusive_scan(float v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

// ---- wave-level reductions (64 lanes) ----
// float sum / max: DPP data movement inside the VALU (quad_perm xor 1, xor 2, row_ror 4, row_ror 8: every lane of a 16-lane row holds the row's
// result; row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3: lane 63 holds the wave's; v_readlane broadcasts it) -- 6 VALU
// instructions instead of the 6 dependent ds_bpermute round trips of a __shfl_xor butterfly (~1000 cycles per pair of sums: it was the
// longest dependency chain of every LayerNorm row and of the BatchNorm-statistics tasks).
#if !defined(EEG_EMU)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
#endif
// sum of rows first, first + STEP, ... (< nrows) of column `col` of a row-major fp64 matrix, IN THAT ORDER, with BATCH loads in flight: written as
// load-all-then-add because hipcc serialised `s += rows[...]` loops into load / s_waitcnt vmcnt(0) / add per row whenever it chose to reuse the
// address registers (csrc/proj1x1.hip: 85 dependent L2 round trips at the head of a 58 k-MAC kernel)
template <int STEP, int BATCH>
__device__ __forceinline__ double ordered_column_sum(const double* __restrict__ rows, int nrows, int first, long long ld, int col) {
    double s = 0.0;
    int r = first;
    for (; r + (BATCH - 1) * STEP < nrows; r += BATCH * STEP) {
        double v[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) v[j] = rows[(long long)(r + j * STEP) * ld + col];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) s += v[j];
    }
    {
        double v[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) v[j] = r + j * STEP < nrows ? rows[(long long)(r + j * STEP) * ld + col] : 0.0;
#pragma unroll
        for (int j = 0; j < BATCH; ++j)
            if (r + j * STEP < nrows) s += v[j];
    }
    return s;
}

__device__ __forceinline__ float wave_sum(float v) {
#if defined(EEG_EMU)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
#else
    v += dpp_mov<0xb1, 0xf>(0.f, v);       // quad_perm:[1,0,3,2]
    v += dpp_mov<0x4e, 0xf>(0.f, v);       // quad_perm:[2,3,0,1]
    v += dpp_mov<0x124, 0xf>(0.f, v);      // row_ror:4
    v += dpp_mov<0x128, 0xf>(0.f, v);      // row_ror:8
    v += dpp_mov<0x142, 0xa>(0.f, v);      // row_bcast:15 -> rows 1, 3 (the others add the `old` operand: 0)
    v += dpp_mov<0x143, 0xc>(0.f, v);      // row_bcast:31 -> rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
#endif
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#if defined(EEG_EMU)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
#else
    v = fmaxf(v, dpp_mov<0xb1, 0xf>(v, v));
    v = fmaxf(v, dpp_mov<0x4e, 0xf>(v, v));
    v = fmaxf(v, dpp_mov<0x124, 0xf>(v, v));
    v = fmaxf(v, dpp_mov<0x128, 0xf>(v, v));
    v = fmaxf(v, dpp_mov<0x142, 0xa>(v, v));       // (rows that do not receive keep `old` = their own value)
    v = fmaxf(v, dpp_mov<0x143, 0xc>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
#endif
}

// ---- MFMA wrappers (gfx950 fragment layouts, guide section 3) ----
// 16x16x4 f32: lane l supplies A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D[row=(l>>4)*4+r][col=l&15] in reg r.
__device__ __forceinline__ f32x4 mfma_f32_16x16x4(float a, float b, f32x4 c) {
#if defined(EEG_EMU)
    struct AB { float a, b; } in{a, b};
    auto all = hipemu::wave_allgather(&in, sizeof(in));
    const int l = hipemu::cur->lane, col = l & 15, rb = (l >> 4) * 4;
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            AB ra, rbv;
            memcpy(&ra, all[(rb + r) + 16 * k], sizeof(AB));
            memcpy(&rbv, all[col + 16 * k], sizeof(AB));
            acc = fmaf(ra.a, rbv.b, acc);
        }
        d[r] = acc;
    }
    return d;
#else
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
// round-to-nearest-even f32 -> bf16 bits (inputs are finite here)
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// 16x16x32 bf16: lane l supplies A[i=l&15][k=8*(l>>4)..+7], B[k=8*(l>>4)..+7][j=l&15]; D as above.
__device__ __forceinline__ f32x4 mfma_bf16_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
#if defined(EEG_EMU)
    struct AB { bf16x8 a, b; } in{a, b};
    auto all = hipemu::wave_allgather(&in, sizeof(in));
    const int l = hipemu::cur->lane, col = l & 15, rb = (l >> 4) * 4;
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        float acc = c[r];
        for (int q = 0; q < 4; ++q) {
            AB ra, rbv;
            memcpy(&ra, all[(rb + r) + 16 * q], sizeof(AB));
            memcpy(&rbv, all[col + 16 * q], sizeof(AB));
            for (int e = 0; e < 8; ++e)
                acc += bf16_bits_to_f32((unsigned short)ra.a[e]) * bf16_bits_to_f32((unsigned short)rbv.b[e]);
        }
        d[r] = acc;
    }
    return d;
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}

// 32x32x16 bf16: lane l supplies A[i=l&31][k=8*(l>>5)..+7], B[k=8*(l>>5)..+7][j=l&31]; D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31] in reg r (16 regs).
__device__ __forceinline__ f32x16 mfma_bf16_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
#if defined(EEG_EMU)
    struct AB { bf16x8 a, b; } in{a, b};
    auto all = hipemu::wave_allgather(&in, sizeof(in));
    const int l = hipemu::cur->lane, col = l & 31, hb = 4 * (l >> 5);
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + hb;
        float acc = c[r];
        for (int h = 0; h < 2; ++h) {
            AB ra, rbv;
            memcpy(&ra, all[row + 32 * h], sizeof(AB));
            memcpy(&rbv, all[col + 32 * h], sizeof(AB));
            for (int e = 0; e < 8; ++e)
                acc += bf16_bits_to_f32((unsigned short)ra.a[e]) * bf16_bits_to_f32((unsigned short)rbv.b[e]);
        }
        d[r] = acc;
    }
    return d;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}

// ---- two-level index -> element offset  (eegclip_dim: offset(i) = (i / div) * so + (i % div) * si) ----
__device__ __forceinline__ long long dim_off(const eegclip_dim& d, int i) {
    if ((long long)i < d.div) return (long long)i * d.si;      // plain strided dimension (div = 2^62) or first run
    const int dv = (int)d.div;                                  // here div <= i <= INT_MAX
    return (long long)(i / dv) * d.so + (long long)(i % dv) * d.si;
}

// a value every lane of the wave agrees on, made provably so for the compiler (scalar branches, SGPR addressing): the wave index
// threadIdx.x >> 6 is uniform, but the compiler only knows that of values it can trace to SGPRs -- without this, everything derived from it (task
// indices, row bases, 64-bit addresses) is computed per lane on the vector ALU
__device__ __forceinline__ int wave_uniform(int v) {
#if defined(EEG_EMU)
    return v;
#else
    return __builtin_amdgcn_readfirstlane(v);
#endif
}

// ---- Philox4x32 counter-based RNG: dropout masks are a pure function of (seed, site, element) so the
//      backward pass regenerates them instead of storing them.  7 rounds (round 4; 10 before): Philox4x32-7 is the fewest-round variant Salmon et al.
//      (SC'11, "Parallel random numbers: as easy as 1, 2, 3") report as passing BigCrush -- ample for a keep / drop decision -- and the generator was
//      ~20 % of the fused transformer-block forward (5 dropout sites).  The masks cannot match torch's in any case; tests regenerate them in numpy
//      (tests/philox_np.py, same round count) so that the oracle runs train mode under the kernels' masks. ----
constexpr int PHILOX_ROUNDS = 7;
struct philox4 { unsigned x, y, z, w; };
__device__ __forceinline__ philox4 philox4x32(unsigned long long seed, unsigned long long ctr_lo, unsigned ctr_hi) {
    const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    unsigned c0 = (unsigned)ctr_lo, c1 = (unsigned)(ctr_lo >> 32), c2 = ctr_hi, c3 = 0u;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < PHILOX_ROUNDS; ++r) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a v_mul_hi_u32 + v_mul_lo_u32 pair: same bits, half the multiplies
        const unsigned long long p0 = (unsigned long long)M0 * c0, p1 = (unsigned long long)M1 * c2;
        const unsigned h0 = (unsigned)(p0 >> 32), l0 = (unsigned)p0, h1 = (unsigned)(p1 >> 32), l1 = (unsigned)p1;
        unsigned n0 = h1 ^ c1 ^ k0, n1 = l1, n2 = h0 ^ c3 ^ k1, n3 = l0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    return philox4{c0, c1, c2, c3};
}
// keep-decision for element `idx` of dropout site `site`: uniform u = (bits >> 8) / 2^24 in [0,1) from 32 Philox bits, keep iff u >= p.  Evaluated on the
// integers: u >= p  <=>  (bits >> 8) >= ceil(p 2^24)  <=>  bits >= ceil(p 2^24) << 8 (p 2^24 is exact in fp32 and < 2^24 for p < 1) -- one compare per
// element instead of shift + convert + multiply + compare; the same decisions bit for bit (tests/philox_np.py keeps the float form).
__device__ __forceinline__ unsigned dropout_threshold(float p) { return (unsigned)ceilf(p * 16777216.0f) << 8; }
__device__ __forceinline__ bool dropout_keep(unsigned long long seed, unsigned site, unsigned long long idx, float p) {
    philox4 r = philox4x32(seed, idx >> 2, site);
    unsigned sel = (unsigned)(idx & 3);
    unsigned bits = sel == 0 ? r.x : sel == 1 ? r.y : sel == 2 ? r.z : r.w;
    return bits >= dropout_threshold(p);
}

// dropout_keep() for the 4 consecutive elements idx4 .. idx4+3 (idx4 % 4 == 0): they share ONE Philox block
__device__ __forceinline__ void dropout_keep4(unsigned long long seed, unsigned site, unsigned long long idx4, float p, bool (&keep)[4]) {
    const philox4 r = philox4x32(seed, idx4 >> 2, site);
    const unsigned thr = dropout_threshold(p);
    keep[0] = r.x >= thr;
    keep[1] = r.y >= thr;
    keep[2] = r.z >= thr;
    keep[3] = r.w >= thr;
}

// value held by lane T of the caller's quad (lanes 4q..4q+3): one DPP move, no LDS crossbar
template <int T>
__device__ __forceinline__ unsigned quad_bcast(unsigned v) {
#if defined(EEG_EMU)
    return hipemu::shfl_idx(v, (hipemu::cur->lane & ~3) | T);
#else
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, T * 0x55, 0xF, 0xF, true);      // quad_perm [T,T,T,T]
#endif
}
// dropout_keep() for the four elements base + 16 t + fr (t = 0..3; base % 4 == 0, fr = lane & 15) of a 16-lane row group, bit-identical
// to four dropout_keep calls: element (t, fr) lives in Philox block (base >> 2) + 4t + (fr >> 2), word fr & 3, so the four lanes of a
// quad run ONE block each (lane j takes t = j) and trade words -- 4x fewer Philox evaluations in the attention kernels
__device__ __forceinline__ void dropout_keep_quad(unsigned long long seed, unsigned site, unsigned long long base, int fr, float p, bool (&keep)[4]) {
    const int q = fr >> 2, j = fr & 3;
    const philox4 r = philox4x32(seed, (base >> 2) + 4 * j + q, site);
    unsigned bits[4];
#define EEG_QUAD_PICK(T)                                                                                      \
    {                                                                                                         \
        const unsigned x = quad_bcast<T>(r.x), y = quad_bcast<T>(r.y), z = quad_bcast<T>(r.z), w = quad_bcast<T>(r.w); \
        bits[T] = j == 0 ? x : j == 1 ? y : j == 2 ? z : w;                                                   \
    }
    EEG_QUAD_PICK(0) EEG_QUAD_PICK(1) EEG_QUAD_PICK(2) EEG_QUAD_PICK(3)
#undef EEG_QUAD_PICK
#pragma unroll
    for (int t = 0; t < 4; ++t) keep[t] = bits[t] >= dropout_threshold(p);
}

// the same trade for four elements that share the quad's lane-in-quad word: lane j hands in the Philox block of ITS element t = j
// (`my_block`), gets back keep[t] = decision of word j in the block lane t handed in.  (GEMM epilogue: a lane's 4 accumulator rows at one
// column; with N % 4 == 0 the four lanes of a quad sit in the same block of every row.)
__device__ __forceinline__ void dropout_keep_quad_blocks(unsigned long long seed, unsigned site, unsigned long long my_block, int j, float p,
                                                         bool (&keep)[4]) {
    const philox4 r = philox4x32(seed, my_block, site);
    unsigned bits[4];
#define EEG_QUAD_PICK(T)                                                                                      \
    {                                                                                                         \
        const unsigned x = quad_bcast<T>(r.x), y = quad_bcast<T>(r.y), z = quad_bcast<T>(r.z), w = quad_bcast<T>(r.w); \
        bits[T] = j == 0 ? x : j == 1 ? y : j == 2 ? z : w;                                                   \
    }
    EEG_QUAD_PICK(0) EEG_QUAD_PICK(1) EEG_QUAD_PICK(2) EEG_QUAD_PICK(3)
#undef EEG_QUAD_PICK
#pragma unroll
    for (int t = 0; t < 4; ++t) keep[t] = bits[t] >= dropout_threshold(p);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    return cdf + x * pdf;
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float silu_grad(float x) {
    const float sg = 1.0f / (1.0f + expf(-x));
    return sg * (1.0f + x * (1.0f - sg));
}
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1f(x); }
// ELU for the fused spatial stage, evaluated 23 M times per pass inside operand staging: hardware exp2 (v_exp_f32) instead of the
// ~40-instruction expm1f; absolute error <= 2e-7 (the subtraction only loses RELATIVE precision near 0, where ELU(x) ~ x ~ 0)
__device__ __forceinline__ float fast_exp(float x) {
#if defined(EEG_EMU)
    return expf(x);
#else
    return __expf(x);
#endif
}
__device__ __forceinline__ float elu1_fast(float x) { return x > 0.f ? x : fast_exp(x) - 1.0f; }

// ---- split-bf16 staging helpers (shared by csrc/gemm_x3.hip and the bf16x3 conv kernels) ----
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

// two fp32 -> packed bf16 pair (round to nearest even), low half = first
__device__ __forceinline__ unsigned x3_pack2(float a, float b) {
#if defined(EEG_EMU)
    return (unsigned)f32_to_bf16_bits(a) | ((unsigned)f32_to_bf16_bits(b) << 16);
#else
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const f32x2_t v{a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));          // v_cvt_pk_bf16_f32
#endif
}
// 4 consecutive-k values -> hi / lo planes (4 bf16 = 8 bytes each)
__device__ __forceinline__ void x3_split4(float v0, float v1, float v2, float v3, u32x2_t& hi, u32x2_t& lo) {
    const unsigned h01 = x3_pack2(v0, v1), h23 = x3_pack2(v2, v3);
    const float r0 = v0 - __uint_as_float(h01 << 16), r1 = v1 - __uint_as_float(h01 & 0xffff0000u);
    const float r2 = v2 - __uint_as_float(h23 << 16), r3 = v3 - __uint_as_float(h23 & 0xffff0000u);
    hi = u32x2_t{h01, h23};
    lo = u32x2_t{x3_pack2(r0, r1), x3_pack2(r2, r3)};
}


// ---- column sums of per-workgroup partial rows (fp64): out[c] += sum_p partials[p][c].  The BatchNorm-statistics producers used to add their 80
// per-channel sums with fp64 atomics straight from every workgroup: 500..1300-way contention per address, ~0.2 ns per atomic device-wide, i.e.
// 8..20 us inside kernels of 45..65 us.  grid (column blocks of 64, COLSUM_SLICES slices of the rows): COLSUM_SLICES atomics per address.
constexpr int COLSUM_SLICES = 8;
template <int UNUSED>
__global__ __launch_bounds__(256) void colsum_f64_kernel(const double* __restrict__ partials, int nparts, int ncols, double* __restrict__ out) {
    EEG_LDS_BASE(double, red);   // [4][64]
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    double s = 0.0;
    if (c < ncols) {
        const int per = (nparts + COLSUM_SLICES - 1) / COLSUM_SLICES;
        const int p0 = blockIdx.y * per, p1 = p0 + per < nparts ? p0 + per : nparts;
#pragma unroll 8
        for (int p = p0 + g; p < p1; p += 4) s += partials[(long long)p * ncols + c];
    }
    red[g * 64 + lane] = s;
    __syncthreads();
    if (g == 0 && c < ncols) atomicAdd(out + c, (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]));
}
#define EEG_COLSUM_F64(partials, nparts, ncols, out, stream)                                                                               \
    EEG_LAUNCH((eeg::colsum_f64_kernel<0>), dim3(((ncols) + 63) / 64, (nparts) < eeg::COLSUM_SLICES ? (nparts) : eeg::COLSUM_SLICES), dim3(256),       \
               256 * sizeof(double), stream, partials, nparts, ncols, out)

}  // namespace eeg
