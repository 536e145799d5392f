// sf_train.h -- training-side launchers (forward+backward of -log_prob, fused clip+Adam).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "sf_layout.h"

struct sf_flow;
// Forward + backward of -log_prob; lazily builds the transposed operand image, the gradient image
// and the activation stash on first use.  Returns 0 or a negative sf_status with `err` set.
int sf_train_loss_grad(sf_flow* f, const float* flat, const float* theta, const float* x, const long long* idx, long B,
                       float grad_scale, const float* weights, float* loss, double* loss_sum, float* grad, float* dctx,
                       hipStream_t st,
                       std::string& err);

// bc_dev != null: the two bias corrections are read from device memory (captured training step) instead of bc1 / bc2
hipError_t sf_launch_fold_loss(double* part, int n, double* out, hipStream_t st);
// sq_part != null: n_sq shares of |grad|^2 left by the gather (k_gather_c2) -- summed in index order instead of reading the
// whole gradient again for the clipping norm
hipError_t sf_launch_adam(float* params, const float* grad, float* m, float* v, float* norm_scratch, long n,
                          const sf_adam_desc& d, float bc1, float bc2, float max_norm, float* grad_norm_out,
                          hipStream_t st, const float* bc_dev = nullptr, const float* sq_part = nullptr, int n_sq = 0);
hipError_t sf_launch_step_begin(const long long* order, long long* ctr, long batch, long long* rows_buf, float beta1, float beta2,
                                float* bc, hipStream_t st);
hipError_t sf_launch_step_end(long long* ctr, hipStream_t st);
