// sf_rng.h -- Philox4x32-10 counter RNG + Box-Muller, in-kernel base noise of the sampler.
// Stream definition is shared with the test oracle (oracle/philox.py):
//   key = (seed_lo, seed_hi ^ stream), counter = (slot_lo, slot_hi, attempt, d/4)
//   u = ((r >> 9) + 0.5) * 2^-23 ; (r0,r1) -> z0,z1 ; (r2,r3) -> z2,z3 ; dim d uses z[d%4].
// Replaces [UPSTREAM] nflows StandardNormal._sample (torch.randn), reached from
// ref: src/synference/sbi_runner.py:6442.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ void sf_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

__device__ __forceinline__ float sf_u01(uint32_t r) {
  return ((float)(r >> 9) + 0.5f) * 1.1920928955078125e-07f;  // 2^-23
}

// four standard normals for (slot, attempt, block)
__device__ __forceinline__ void sf_normal4(uint32_t k0, uint32_t k1, uint64_t slot, uint32_t attempt,
                                           uint32_t blk, float (&z)[4]) {
  uint32_t r[4];
  sf_philox4x32_10((uint32_t)slot, (uint32_t)(slot >> 32), attempt, blk, k0, k1, r);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    // hardware forms: v_log_f32 (log2), v_sqrt_f32, and v_sin/v_cos_f32, which take the angle in turns -- the
    // uniform itself, no 2 pi multiply and no range reduction (|dz| <~ 1e-6 against libm; the oracle uses libm)
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(sf_u01(r[2 * q])));
    const float turn = sf_u01(r[2 * q + 1]);
    z[2 * q] = rad * __builtin_amdgcn_cosf(turn);
    z[2 * q + 1] = rad * __builtin_amdgcn_sinf(turn);
  }
}
