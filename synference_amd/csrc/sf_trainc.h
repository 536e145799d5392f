// sf_trainc.h -- cooperative 16-row MAF training kernel (sf_trainc.hip): argument block and launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sf_layout.h"

#define SF_TRC_TS 5       // transforms whose a1 / a2 tiles a wave keeps in registers without spilling
#define SF_TRC_TS_MAX 8   // flows with more transforms take k_maf_train
// the instantiation (stash depth) a flow of T transforms runs
inline int sf_trc_ts(int T) { return T <= 5 ? 5 : (T <= 6 ? 6 : 8); }

struct SfTrcArgs {
  SfTrcDev c;
  const float* img;   // cooperative operand image (SfLayout::srcC1/2), all transforms
  const float* cst;   // constants image (sf_layout.cpp)
  int D, C, T, scale_fn;
  float eps, logdet0;
  int c_pscale, c_pshift, c_tdim, c_xmean, c_xstd;
  const float* theta;
  const float* x;
  const long long* idx;  // optional [B]: batch row b reads library row idx[b]
  const float* wts;      // optional per-sample weights [B] (multiplied by w)
  long B, n_chunks;      // n_chunks = ceil(B / (32 * groups per workgroup))
  float w;
  float* loss;           // [B] or null
  double* loss_sum;      // optional device scalar -- or, loss_mask != 0, loss_mask + 1 scalars: workgroup i adds to [i & loss_mask]
  int loss_mask;         // (hundreds of same-address double atomics serialise in one L2 channel: ~6 us of a 90 us step)
  float* dctx;           // [B, C] or null (only with one input tile)
  float* gpart;          // [grid] gradient partials of gpart_stride floats; plain stores, summed by k_gather_c
  long gpart_stride;
  long long* fix;        // != null: SF_FIX_REPLICAS zeroed int64 images of gpart_stride entries instead (sf_fixacc.h)
#ifdef SF_TRC_TRACE
  unsigned long long* trace;  // developer build: [8 waves][256] cycle stamps of workgroup 0
#endif
};

#ifdef SF_TRC_TRACE
#define SF_TC(slot)                                                                                         \
  do {                                                                                                      \
    if (a.trace && blockIdx.x == 0 && (threadIdx.x & 63) == 0)                                              \
      a.trace[(threadIdx.x >> 6) * 256 + (slot)] = __builtin_readcyclecounter();                            \
  } while (0)
// fine stamps inside one phase of one transform: `dep` (a VGPR value) must be complete before the stamp is taken
#define SF_TCX(cond, slot, dep)                                                                             \
  do {                                                                                                      \
    if (cond) {                                                                                             \
      float d_ = (dep);                                                                                     \
      asm volatile("v_mov_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(d_) :: "memory");                     \
      SF_TC(slot);                                                                                          \
    }                                                                                                       \
  } while (0)
#else
#define SF_TC(slot) do { } while (0)
#define SF_TCX(cond, slot, dep) do { } while (0)
#endif

size_t sf_trainc_lds_bytes(const SfTrcDev& c, int TS, int NG);
int sf_trainc_groups(long B, const SfTrcDev* c = nullptr, int T = 0);
bool sf_trainc_eligible(const SfLayout& L, bool want_dctx);
int sf_trainc_grid(long B, const SfTrcDev* c = nullptr, int T = 0);
bool sf_trainc_fix(int grid, long n_gradC);
hipError_t sf_launch_maf_trainc(const SfTrcArgs& a, int grid, hipStream_t st);
hipError_t sf_launch_gather_c(const float* gpart, long stride, int nwg, const int32_t* gdst, float* grad, long n, hipStream_t st);
hipError_t sf_launch_gather_fix(const long long* gfix, long stride, int nrep, const int32_t* gsrc, const int32_t* gzero, long n_zero,
                                float* grad, hipStream_t st);
// sq_part (optional, sf_gather_c2_blocks floats): per-block shares of |grad|^2 for the optimiser step (sf_launch_adam)
long sf_gather_c2_blocks(long stride, long n_zero);
hipError_t sf_launch_gather_c2(const float* gpart, long stride, int nwg, const int32_t* gsrc, const int32_t* gzero, long n_zero,
                               float* grad, hipStream_t st, float* sq_part = nullptr);
