// sf_fixacc.h -- order-independent gradient accumulation for the cooperative training kernels.
//
// A workgroup's weight-gradient blocks have to be summed over all workgroups of a step.  Per-workgroup partials + a gather are
// bitwise reproducible but cost (workgroups x gradient image) bytes of HBM traffic per step (MAF cfg1 at batch 16 384: 256 x
// 145 KB written and read back; NSF cfg3: 512 x 0.6 MB -- out of the question); f32 atomics into a shared image stay in L2
// but add in whatever order the hardware serves them.  Here a contribution is converted to 2^-40 FIXED POINT in an int64 and
// added with a 64-bit integer atomic into the replica of the workgroup's XCD (SF_FIX_REPLICAS zeroed images, each touched by
// one XCD only, so the atomics execute in that XCD's L2 and the images never travel): integer addition is associative, the sum
// does not depend on the order, and the gather adds the replicas in integers before it converts back -- same inputs, same bits,
// at any batch size.  Resolution 2^-40 = 9e-13 absolute per contribution (fp32 accumulators of |value| >= 2^-17 convert
// exactly), range +-8.4e6 for a sum.
#pragma once
#include <hip/hip_runtime.h>

#define SF_FIX_REPLICAS 8
#define SF_FIX_SCALE 1099511627776.0   // 2^40

struct SfAcc {
  int mode;           // 0: plain store into the workgroup's partial, 1: add to it (later chunks), 3: fixed-point atomics
  const float* base;  // the float pointer the job descriptors' gw / gb were built from (index 0 of a gradient image)
  long long* fix;     // mode 3: this workgroup's replica
};

// Atomics on a replica are issued at WORKGROUP scope: an agent-scope atomic carries sc1 and is forwarded by the L2 to the
// fabric (measured: WRITE_SIZE = every atomic's bytes, profiles/r04_pmc_nsfatomic.csv), a workgroup-scope one is executed by
// the XCD's own L2 -- which is the only L2 that ever touches this replica; the end-of-kernel write-back publishes the sums.
__device__ __forceinline__ void sf_l2_add(float* p, float v) {
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void sf_fix_add(long long* p, float v) {
  const long long q = __double2ll_rn((double)v * SF_FIX_SCALE);
  (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(p), (unsigned long long)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int sf_xcc_id() {
  int xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  return xcc & (SF_FIX_REPLICAS - 1);
}
