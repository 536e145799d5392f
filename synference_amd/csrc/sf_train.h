// sf_train.h -- training-side launchers (forward+backward of -log_prob, fused clip+Adam).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "sf_layout.h"

// Forward + backward; lazily builds the transposed operand image / gradient tables on first use.
int sf_train_loss_grad(const SfLayout& L, SfDev dev, float** d_packedT, int32_t** d_t1, int32_t** d_t2,
                       float** d_gpacked, int32_t** d_gdst, int32_t** d_gdst2, float** d_act, size_t* act_cap,
                       const int32_t* d_s1, const int32_t* d_s2, float* d_packed, const float* flat,
                       const float* theta, const float* x, long B, float grad_scale, float* loss, float* grad,
                       hipStream_t st, std::string& err);

hipError_t sf_launch_adam(float* params, const float* grad, float* m, float* v, float* norm_scratch, long n,
                          const sf_adam_desc& d, float bc1, float bc2, float max_norm, float* grad_norm_out,
                          hipStream_t st);
