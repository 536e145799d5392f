"""Philox4x32-10 + Box-Muller in numpy -- the counter-based noise stream shared by
the oracle sampler and the HIP sampler (synference_amd/csrc/sf_rng.h).

TEST INFRASTRUCTURE ONLY.  The reference draws its base noise from torch's global
generator ([UPSTREAM] nflows StandardNormal._sample -> torch.randn); that stream
cannot be reproduced on a GPU, so both sides of the parity test use this one.

Stream definition (must match sf_rng.h bit for bit in the integer part):
  key      = (seed & 0xffffffff, (seed >> 32) ^ stream)
  counter  = (slot & 0xffffffff, slot >> 32, attempt, d // 4)
  r[0..3]  = philox4x32_10(counter, key)
  u_i      = ((r_i >> 9) + 0.5) * 2**-23                       (exact in fp32)
  pair (r0,r1): z0 = sqrt(-2 ln u0) * cos(2 pi u1), z1 = ... * sin(2 pi u1); same for (r2,r3)
  noise for dimension d = z[d % 4] of block d // 4
"""
from __future__ import annotations

import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(a, dtype=np.uint32) for a in (c0, c1, c2, c3))
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32(k0 + W0)
            k1 = np.uint32(k1 + W1)
    return c0, c1, c2, c3


def _u01(r):
    return ((r >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23)


def normal(seed: int, slot, attempt, D: int, stream: int = 0) -> np.ndarray:
    """Standard-normal noise [n, D] float32 for ``slot`` (uint64 array) and ``attempt`` (int array)."""
    slot = np.asarray(slot, dtype=np.uint64)
    attempt = np.broadcast_to(np.asarray(attempt, dtype=np.uint32), slot.shape)
    k0 = seed & 0xFFFFFFFF
    k1 = ((seed >> 32) & 0xFFFFFFFF) ^ (stream & 0xFFFFFFFF)
    out = np.empty(slot.shape + (D,), dtype=np.float32)
    two_pi = np.float32(6.2831855)
    for blk in range((D + 3) // 4):
        r = philox4x32_10((slot & MASK).astype(np.uint32), (slot >> np.uint64(32)).astype(np.uint32),
                          attempt, np.full(slot.shape, blk, dtype=np.uint32), k0, k1)
        z = []
        for a, b in ((r[0], r[1]), (r[2], r[3])):
            rad = np.sqrt(np.float32(-2.0) * np.log(_u01(a)))
            ang = two_pi * _u01(b)
            z += [rad * np.cos(ang), rad * np.sin(ang)]
        for j in range(4):
            d = 4 * blk + j
            if d < D:
                out[..., d] = z[j]
    return out
