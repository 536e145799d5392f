// sf_nsf1.h -- the one-parameter NSF (sf_nsf1.hip): state and entry points used by sf_api.hip / sf_train.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/synference_hip.h"

#define SF_NSF1_TMAX 16   // transforms (the spline chain of the training kernel is unrolled over them)

struct SfNsf1 {
  sf_mlp* mlp = nullptr;    // ContextSplineMap of ONE transform; transform t runs it on flat + t * P_mlp
  int64_t P_mlp = 0;
  int T = 0, C = 0, H = 0, K = 0, NP = 0;
  float tail_bound = 3.f, min_w = 1e-3f, min_h = 1e-3f, min_d = 1e-3f, inv_sqrt_h = 1.f, deriv_const = 0.f;
  float th_mean = 0.f, th_std = 1.f;
  float* d_q = nullptr;     // [T][rows][3K - 1] raw spline parameters
  size_t q_cap = 0;
  float* d_dq = nullptr;    // their gradients
  size_t dq_cap = 0;
  float* d_xg = nullptr;    // gathered mini-batch rows (sf_flow_loss_grad_rows)
  size_t xg_cap = 0;
  float* d_thg = nullptr;
  size_t thg_cap = 0;
  unsigned int* d_cnt = nullptr;
};

int sf_nsf1_create(const sf_flow_desc& d, SfNsf1** out, std::string& err);
void sf_nsf1_destroy(SfNsf1* n);
int sf_nsf1_log_prob(SfNsf1* n, const float* flat, const float* theta, const float* x, long B, float* out, hipStream_t st, std::string& err);
int sf_nsf1_inverse(SfNsf1* n, const float* flat, const float* z, const float* x, long B, float* theta, float* logdet, hipStream_t st,
                    std::string& err);
int sf_nsf1_loss_grad(SfNsf1* n, const float* flat, const float* theta, const float* x, const long long* idx, long B, float grad_scale,
                      const float* weights, float* loss, double* loss_sum, float* grad, hipStream_t st, std::string& err);
int sf_nsf1_sample(SfNsf1* n, const float* flat, const float* x, long M, long S, const uint32_t* slots, long n_slots, const float* lo,
                   const float* hi, uint32_t k0, uint32_t k1, unsigned long long slot_offset, int max_attempts, float* out,
                   int32_t* n_drawn, int64_t* n_unfilled, hipStream_t st, std::string& err);
int sf_nsf1_acceptance(SfNsf1* n, const float* flat, const float* x, long M, long cnt, const float* lo, const float* hi, uint32_t k0,
                       uint32_t k1, unsigned long long slot_offset, int32_t* count, hipStream_t st, std::string& err);
