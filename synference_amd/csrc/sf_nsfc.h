// sf_nsfc.h -- cooperative 16-row NSF training kernel (sf_nsfc.hip): argument block and launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sf_layout.h"


struct SfNscArgs {
  SfNscDev c;
  const float* img;   // cooperative operand image (SfLayout::srcC1), all transforms
  const float* cst;   // constants image (sf_layout.cpp)
  int D, C, T, K;
  float tail_bound, min_w, min_h, min_d, lu_eps, inv_sqrt_h, deriv_const, logdet0;
  int c_pscale, c_pshift, c_xmean, c_xstd;
  const float* theta;
  const float* x;
  const long long* idx;  // optional [B]: batch row b reads library row idx[b]
  const float* wts;      // optional per-sample weights [B] (multiplied by w)
  long B, n_chunks;      // n_chunks = ceil(B / 32)
  float w;
  float* loss;           // [B] or null
  double* loss_sum;      // optional device scalar -- or, loss_mask != 0, loss_mask + 1 scalars: workgroup i adds to [i & loss_mask]
  int loss_mask;
  // gradient accumulation target: one partial of gpart_stride floats per workgroup (plain stores, summed by k_gather_c2 in
  // workgroup order), or -- fix != null -- SF_FIX_REPLICAS zeroed int64 images of gpart_stride entries: the workgroup adds
  // 2^-40 fixed-point contributions into the replica of its XCD with integer atomics that stay in that XCD's L2
  // (sf_fixacc.h: the NSF partial is 0.6 MB -- 512 of them per step would be 300 MB of HBM traffic for a 0.37 MB gradient),
  // or float contributions with f32 atomics (fix_mode 2: twice as fast in the L2s, but the order of the adds is the hardware's).
  float* gpart;
  long gpart_stride;
  long long* fix;
  int fix_mode;          // 3: int64 fixed point (sf_fixacc.h); 2: the replicas are FLOAT images added to with f32 atomics
  float* ustash;         // [n_chunks * 32][T][16]: u entering transform t (8) and the spline's outputs u' (8)
#ifdef SF_NSC_TRACE
  unsigned long long* trace;  // developer build: [4 waves][512] cycle stamps of workgroup 0
#endif
};

#ifdef SF_NSC_TRACE
#define SF_NC(slot)                                                                                         \
  do {                                                                                                      \
    if (a.trace && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (slot) < 512)                              \
      a.trace[(threadIdx.x >> 6) * 512 + (slot)] = __builtin_readcyclecounter();                            \
  } while (0)
#else
#define SF_NC(slot) do { } while (0)
#endif

size_t sf_nsfc_lds_bytes(const SfNscDev& c);
bool sf_nsfc_eligible(const SfLayout& L, bool want_dctx);
int sf_nsfc_grid(long B, int NT);
int sf_nsfc_acc_mode(long B, int grid, long n_gradC);
hipError_t sf_launch_nsf_trainc(const SfNscArgs& a, int grid, hipStream_t st);
