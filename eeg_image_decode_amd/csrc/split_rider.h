// "Riders": dense fp32 -> bf16 hi | lo plane splits executed by EXTRA workgroups of a main-chain kernel that leaves most of the chip idle.
//
// The projection head's plane GEMMs (csrc/head_gemm.hip) need this step's head weights and the loss targets as planes.  As launches of their own on the second
// stream at the start of the step (round 6, first version) they ran beside the fused transformer-block forward -- 21 us of splits slowed it by 12 us -- and
// the main stream paid a join (~8 us of idle queue) in front of the conv stack.  The 1x1-conv tail of the conv stack (csrc/proj1x1.hip) is one small
// workgroup per sample, 15 us of latency with the memory system idle: a second set of workgroups of the SAME launch performs the splits meanwhile.  Nothing
// is ordered inside the launch: the riders' outputs are read by later launches only.
#pragma once
#include "eeg_common.h"

namespace eeg {

constexpr int RIDER_MAX = 4;
struct rider_item {
    const float* src;
    unsigned short *hi, *lo;
    long long n4;                         // float4 groups (the item is dense: n4 * 4 contiguous elements)
};
struct rider_table {
    rider_item it[RIDER_MAX];
    int n;
};

// workgroup `rb` of `nrb` rider workgroups (256 threads each): a grid-stride loop over the concatenated items
__device__ __forceinline__ void split_rider(const rider_table& tb, int rb, int nrb) {
    for (int i = 0; i < tb.n; ++i) {
        const rider_item& E = tb.it[i];
        for (long long q = (long long)rb * 256 + threadIdx.x; q < E.n4; q += 256LL * nrb) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(E.src + 4 * q);
            u32x2_t h, l;
            x3_split4(v[0], v[1], v[2], v[3], h, l);
            *reinterpret_cast<u32x2_t*>(E.hi + 4 * q) = h;
            *reinterpret_cast<u32x2_t*>(E.lo + 4 * q) = l;
        }
    }
}

// host side: dense eegclip_split_item entries (transpose = 0, no copy, ld_src == ld_out == cols, rows * cols % 4 == 0, 16- / 8-byte aligned) -> table
inline int rider_table_from(const eegclip_split_item* items, int n, rider_table& tb) {
    tb.n = 0;
    if (n == 0) return 0;
    if (!items || n < 0 || n > RIDER_MAX) return EEGCLIP_EINVAL;
    for (int i = 0; i < n; ++i) {
        const eegclip_split_item& it = items[i];
        const long long total = (long long)it.rows * it.cols;
        if (!it.src || !it.hi || !it.lo || it.rows < 1 || it.cols < 1 || it.transpose || it.copy || it.ld_src != it.cols || it.ld_out != it.cols || (total & 3))
            return EEGCLIP_EINVAL;
        if ((reinterpret_cast<uintptr_t>(it.src) & 15u) || ((reinterpret_cast<uintptr_t>(it.hi) | reinterpret_cast<uintptr_t>(it.lo)) & 7u)) return EEGCLIP_EALIGN;
        tb.it[i] = rider_item{it.src, static_cast<unsigned short*>(it.hi), static_cast<unsigned short*>(it.lo), total / 4};
    }
    tb.n = n;
    return 0;
}

}  // namespace eeg
