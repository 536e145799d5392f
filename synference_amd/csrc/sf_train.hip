// sf_train.hip -- training kernels.
//   (1) fused global-norm clip + Adam/AdamW      ref: custom_runner.py:613-618
//   (2) forward + backward of -log_prob          ref: custom_runner.py:604-610
#include "sf_train.h"

#include <hip/hip_runtime.h>

#include "sf_flows.h"
#include "sf_internal.h"

// ---------------------------------------------------------------------------------------------
// clip_grad_norm_ + Adam
// ---------------------------------------------------------------------------------------------
__global__ void k_sqnorm(const float* __restrict__ g, long n, float* __restrict__ acc) {
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    s += g[i] * g[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  __shared__ float part[16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) part[w] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += part[i];
    atomicAdd(acc, t);
  }
}

__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, const float* __restrict__ sq, long n, sf_adam_desc d, float bc1,
                       float bc2, float max_norm, float* __restrict__ norm_out) {
  const float total = sqrtf(*sq);
  float coef = 1.f;
  if (max_norm > 0.f) coef = fminf(max_norm / (total + 1e-6f), 1.0f);  // torch clip_grad_norm_
  if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = total;
  const float step_size = d.lr / bc1;
  const float inv_sqrt_bc2 = rsqrtf(bc2);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float pi = p[i];
    float gi = g[i] * coef;
    if (d.decoupled) pi *= (1.f - d.lr * d.weight_decay);
    else if (d.weight_decay != 0.f) gi += d.weight_decay * pi;
    const float mi = d.beta1 * m[i] + (1.f - d.beta1) * gi;
    const float vi = d.beta2 * v[i] + (1.f - d.beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + d.eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

hipError_t sf_launch_adam(float* params, const float* grad, float* m, float* v, float* norm_scratch, long n,
                          const sf_adam_desc& d, float bc1, float bc2, float max_norm, float* grad_norm_out,
                          hipStream_t st) {
  hipError_t e = hipMemsetAsync(norm_scratch, 0, sizeof(float), st);
  if (e != hipSuccess) return e;
  const int blocks = (int)((n + 1023) / 1024 < 256 ? (n + 1023) / 1024 : 256);
  hipLaunchKernelGGL(k_sqnorm, dim3(blocks), dim3(256), 0, st, grad, n, norm_scratch);
  hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, st, params, grad, m, v, norm_scratch, n, d, bc1, bc2,
                     max_norm, grad_norm_out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// forward + backward  (built in the next milestone)
// ---------------------------------------------------------------------------------------------
int sf_train_loss_grad(const SfLayout&, SfDev, float**, int32_t**, int32_t**, float**, int32_t**, int32_t**,
                       float**, size_t*, const int32_t*, const int32_t*, float*, const float*, const float*,
                       const float*, long, float, float*, float*, hipStream_t, std::string& err) {
  err = "sf_flow_loss_grad: backward kernels not built in this library";
  return SF_ERR_STATE;
}
