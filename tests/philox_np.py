"""numpy restatement of the kernels' dropout RNG (csrc/eeg_common.h: philox4x32 with PHILOX_ROUNDS = 7 / dropout_keep).
TEST ONLY: lets the oracle run train-mode forward/backward with exactly the masks the HIP kernels draw."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


ROUNDS = 7          # csrc/eeg_common.h: PHILOX_ROUNDS


def philox4x32(seed, ctr_lo, ctr_hi):
    """seed: python int (64 bit); ctr_lo: uint64 array; ctr_hi: python int (32 bit).  Returns 4 uint32 arrays."""
    ctr_lo = np.asarray(ctr_lo, dtype=np.uint64)
    c0 = (ctr_lo & MASK).astype(np.uint64)
    c1 = (ctr_lo >> np.uint64(32)).astype(np.uint64)
    c2 = np.full_like(c0, ctr_hi & MASK)
    c3 = np.zeros_like(c0)
    k0, k1 = seed & MASK, (seed >> 32) & MASK
    for _ in range(ROUNDS):
        p0 = np.uint64(M0) * c0
        p1 = np.uint64(M1) * c2
        h0, l0 = p0 >> np.uint64(32), p0 & np.uint64(MASK)
        h1, l1 = p1 >> np.uint64(32), p1 & np.uint64(MASK)
        n0 = h1 ^ c1 ^ np.uint64(k0)
        n2 = h0 ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, l1, n2, l0
        k0 = (k0 + W0) & MASK
        k1 = (k1 + W1) & MASK
    return c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32)


def keep_mask(seed, site, n, p):
    """Boolean keep mask for elements 0..n-1 of dropout site `site` (flat, logical index order)."""
    idx = np.arange(n, dtype=np.uint64)
    r = philox4x32(seed, idx >> np.uint64(2), site)
    sel = (idx & np.uint64(3)).astype(np.int64)
    bits = np.choose(sel, r)
    u = (bits >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return u >= np.float32(p)
