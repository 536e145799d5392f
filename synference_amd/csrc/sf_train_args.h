// sf_train_args.h -- argument block of the training kernels.
#pragma once
#include <hip/hip_runtime.h>

#define SF_GCOPIES 8

struct SfTrainArgs {
  const float* theta;
  const float* x;
  long B;
  float w;           // gradient weight of every sample (grad_scale)
  const float* wts;  // optional per-sample weights [B] (multiplied by w)
  const long long* idx;  // optional [B]: batch row b reads theta / x row idx[b] (gather fused into the kernel)
  double* loss_sum;  // optional device scalar: += sum_b loss_b (unweighted), one f64 atomic per tile
  float* loss;       // [B] or null
  float* dctx;       // [B,C] or null: += d(sum_b w_b loss_b)/d x[b,:] (context gradient, raw x units)
  float* gimg;       // gradient image replicas of gimg_stride floats, summed in fixed order by the gather:
  long gimg_stride;  //   det == 0: SF_GCOPIES replicas, one per XCD, f32 atomics
  int det;           //   det == 1: one replica per 32-sample tile, plain stores -> bitwise reproducible gradients
  float4* act;       // activation stash
  long act_per_wave; // float4 per wave
#ifdef SF_TRAIN_TRACE
  unsigned long long* trace;  // developer build only: [2 waves][128] time stamps of one workgroup (scripts/train_trace.sh)
#endif
};

#ifdef SF_TRAIN_TRACE
#define SF_TR(slot)                                                                                                   \
  do {                                                                                                                \
    if (a.trace && blockIdx.x == gridDim.x / 2 && (threadIdx.x & 63) == 0)                                            \
      a.trace[(threadIdx.x >> 6) * 128 + (slot)] = __builtin_readcyclecounter();                                      \
  } while (0)
#else
#define SF_TR(slot) do { } while (0)
#endif
