// sf_nsf1.hip -- the one-parameter NSF (D = 1).
//
// [UPSTREAM] sbi build_nsf, `if x_numel == 1` (reached through the same load_nde_sbi call as every other NSF,
// ref: src/synference/sbi_runner.py:5121-5146): the coupling mask is [1] in every block -- the single dimension is always
// transformed, there is nothing to condition on -- the conditioner is ContextSplineMap(hidden_layers = 1), i.e.
//     q_t = Linear(H, 3K - 1)(relu(Linear(H, H)(relu(Linear(C, H)(e(x))))))                t = 0 .. T - 1
// on the (standardised, embedded) context ALONE, and no LULinear is appended.  The flow is therefore
//     u_0 = (theta - mean) / std ;  u_{t+1} = RQS(u_t ; q_t(x)) ;  log p = log N(u_T) + sum_t log|RQS'| - log std
// -- T small MLPs on the context (they run on the register-tile MLP engine of the embedding net, sf_mlp.hip: forward, and
// backward with the weight gradients) and a chain of T scalar splines per sample (one thread per sample / per draw; a
// draw costs T inverse splines and NO conditioner evaluation: the parameters of a galaxy are computed once).
// Flat layout per transform: W0[H,C] b0[H] W1[H,H] b1[H] W2[3K-1,H] b2[3K-1]  (= the MLP engine's own layout).
#include <hip/hip_runtime.h>

#include <string>

#include "sf_internal.h"
#include "sf_nsf1.h"
#include "sf_rng.h"
#include "sf_spline_flat.h"

namespace {

constexpr int KM1 = 16, NQ1 = 48;  // slots: widths [0, 16), heights [16, 32), derivatives [32, 47)
using Spl1 = NSpl<KM1, NQ1>;

struct Nsf1C {
  NSplC sc;
  int T, NP;
  float th_scale, th_shift, logdet0;  // u = theta * scale + shift
};

// the 3K - 1 raw parameters of (transform t, row b) -> spline slots
__device__ __forceinline__ void n1_load_q(const float* __restrict__ q, long n_rows, int t, long b, const Nsf1C& c, float (&s)[NQ1]) {
  const float* p = q + ((size_t)t * n_rows + b) * c.NP;
  const int K = c.sc.K;
#pragma unroll
  for (int k = 0; k < KM1; ++k) {
    s[k] = k < K ? p[k] : 0.f;
    s[KM1 + k] = k < K ? p[K + k] : 0.f;
    s[2 * KM1 + k] = k < K - 1 ? p[2 * K + k] : 0.f;
  }
}

__global__ void k_nsf1_logprob(const float* __restrict__ q, const float* __restrict__ theta, long B, Nsf1C c, float* __restrict__ out) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float u = theta[b] * c.th_scale + c.th_shift, ld = c.logdet0;
  for (int t = 0; t < c.T; ++t) {
    float s[NQ1];
    n1_load_q(q, B, t, b, c, s);
    float v, lad;
    Spl1::fwd(c.sc, s, u, v, lad);
    u = v;
    ld += lad;
  }
  out[b] = -0.5f * u * u - 0.5f * 1.8378770664093453f + ld;
}

__global__ void k_nsf1_inverse(const float* __restrict__ q, const float* __restrict__ z, long B, Nsf1C c, float* __restrict__ theta,
                               float* __restrict__ logdet) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float u = z[b], ld = -c.logdet0;
  for (int t = c.T - 1; t >= 0; --t) {
    float s[NQ1];
    n1_load_q(q, B, t, b, c, s);
    float v, lad;
    Spl1::inv(c.sc, s, u, v, lad);
    u = v;
    ld += lad;
  }
  theta[b] = (u - c.th_shift) / c.th_scale;
  if (logdet) logdet[b] = ld;
}

// forward + backward of -log p through the spline chain: dq[t][b][:] = d(w_b * loss_b) / d q_t[b][:]
__global__ void k_nsf1_train(const float* __restrict__ q, const float* __restrict__ theta, long B, Nsf1C c, float w,
                             const float* __restrict__ wts, float* __restrict__ dq, float* __restrict__ loss,
                             double* __restrict__ loss_sum) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = b < B;
  float nll = 0.f;
  if (valid) {
    float uin[SF_NSF1_TMAX];
    float u = theta[b] * c.th_scale + c.th_shift, ld = c.logdet0;
#pragma unroll
    for (int t = 0; t < SF_NSF1_TMAX; ++t) {
      uin[t] = u;
      if (t < c.T) {
        float s[NQ1];
        n1_load_q(q, B, t, b, c, s);
        float v, lad;
        Spl1::fwd(c.sc, s, u, v, lad);
        u = v;
        ld += lad;
      }
    }
    nll = 0.5f * u * u + 0.5f * 1.8378770664093453f - ld;
    if (loss) loss[b] = nll;
    const float wb = wts ? w * wts[b] : w;
    float G = wb * u;
#pragma unroll
    for (int tt = 0; tt < SF_NSF1_TMAX; ++tt) {
      const int t = SF_NSF1_TMAX - 1 - tt;
      if (t < c.T) {
        float s[NQ1], ds[NQ1];
        n1_load_q(q, B, t, b, c, s);
        float dv;
        Spl1::bwd(c.sc, s, uin[t], G, -wb, dv, ds);
        float* p = dq + ((size_t)t * B + b) * c.NP;
        const int K = c.sc.K;
#pragma unroll
        for (int k = 0; k < KM1; ++k) {
          if (k < K) { p[k] = ds[k]; p[K + k] = ds[KM1 + k]; }
          if (k < K - 1) p[2 * K + k] = ds[2 * KM1 + k];
        }
        G = dv;
      }
    }
  }
  if (loss_sum) {
    float tsum = valid ? nll : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tsum += __shfl_xor(tsum, o, 64);
    // values on a 2^-20 grid add exactly in double: the sum does not depend on the order of the atomics
    if ((threadIdx.x & 63) == 0) atomicAdd(loss_sum, (double)rintf(tsum * 1048576.0f) * (1.0 / 1048576.0));
  }
}

// one thread per output slot: attempts 0, 1, 2, ... of the slot's Philox stream until one lands in [lo, hi]
// (the lowest accepted attempt is the draw: the per-slot restatement of sbi's rejection loop, DESIGN.md "Sampler")
__global__ void k_nsf1_sample(const float* __restrict__ qg, long M, long S, const uint32_t* __restrict__ slots, long n_slots,
                              Nsf1C c, const float* __restrict__ lo, const float* __restrict__ hi, uint32_t k0, uint32_t k1,
                              unsigned long long slot_offset, uint32_t max_attempts, float* __restrict__ out,
                              int32_t* __restrict__ n_drawn, unsigned int* __restrict__ n_unfilled) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_slots) return;
  const unsigned long long slot = slots ? (unsigned long long)slots[i] : (unsigned long long)i;
  const long g = (long)(slot / (unsigned long long)S);
  const float vlo = lo ? lo[0] : -3.0e38f, vhi = hi ? hi[0] : 3.0e38f;
  float th = __builtin_nanf("");
  uint32_t att = 0;
  bool ok = false;
  for (; att < max_attempts && !ok; ++att) {
    float z4[4];
    sf_normal4(k0, k1, slot + slot_offset, att, 0u, z4);
    float u = z4[0];
    for (int t = c.T - 1; t >= 0; --t) {
      float s[NQ1];
      n1_load_q(qg, M, t, g, c, s);
      float v, lad;
      Spl1::inv(c.sc, s, u, v, lad);
      u = v;
    }
    const float cand = (u - c.th_shift) / c.th_scale;
    ok = (cand == cand) && fabsf(cand) < 3.0e38f && cand >= vlo && cand <= vhi;
    if (ok) th = cand;
  }
  out[slot] = th;
  if (n_drawn) sf_sat_add(n_drawn + g, (int32_t)att);
  if (!ok) atomicAdd(n_unfilled, 1u);
}

// leakage correction: how many of n unconstrained draws of row g fall in the box (stream id 1, attempt 0)
__global__ void k_nsf1_accept(const float* __restrict__ qg, long M, long n, Nsf1C c, const float* __restrict__ lo,
                              const float* __restrict__ hi, uint32_t k0, uint32_t k1, unsigned long long slot_offset,
                              int32_t* __restrict__ count) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * n) return;
  const long g = i / n;
  float z4[4];
  sf_normal4(k0, k1, (unsigned long long)i + slot_offset, 0u, 0u, z4);
  float u = z4[0];
  for (int t = c.T - 1; t >= 0; --t) {
    float s[NQ1];
    n1_load_q(qg, M, t, g, c, s);
    float v, lad;
    Spl1::inv(c.sc, s, u, v, lad);
    u = v;
  }
  const float cand = (u - c.th_shift) / c.th_scale;
  if ((cand == cand) && cand >= lo[0] && cand <= hi[0]) atomicAdd(count + g, 1);
}

__global__ void k_nsf1_gather(const float* __restrict__ theta, const float* __restrict__ x, const long long* __restrict__ idx, long B,
                              int C, float* __restrict__ thg, float* __restrict__ xg) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * (C + 1)) return;
  const long b = i / (C + 1);
  const int j = (int)(i - b * (C + 1));
  const long src = (long)idx[b];
  if (j == C) thg[b] = theta[src];
  else xg[b * C + j] = x[src * C + j];
}

Nsf1C consts_of(const SfNsf1& n) {
  Nsf1C c;
  c.sc = NSplC{n.K, n.tail_bound, n.min_w, n.min_h, n.min_d, n.inv_sqrt_h, n.deriv_const};
  c.T = n.T;
  c.NP = n.NP;
  c.th_scale = 1.0f / n.th_std;
  c.th_shift = -n.th_mean / n.th_std;
  c.logdet0 = -logf(fabsf(n.th_std));
  return c;
}

#define N1_HIP(call)                                                        \
  do {                                                                      \
    hipError_t e_ = (call);                                                 \
    if (e_ != hipSuccess) {                                                 \
      err = std::string(#call) + ": " + hipGetErrorString(e_);              \
      return SF_ERR_HIP;                                                    \
    }                                                                       \
  } while (0)
#define N1_RC(call)                                                         \
  do {                                                                      \
    int rc_ = (call);                                                       \
    if (rc_) {                                                              \
      err = std::string(#call) + ": " + sf_last_error();                    \
      return rc_;                                                           \
    }                                                                       \
  } while (0)

int grow(float*& p, size_t& cap, size_t need, std::string& err) {
  if (need <= cap) return SF_OK;
  if (p) N1_HIP(hipFree(p));
  p = nullptr;
  cap = 0;
  N1_HIP(hipMalloc(&p, need * sizeof(float)));
  cap = need;
  return SF_OK;
}

// q[t][row][:] = MLP_t(x[row]) for every transform, with the parameters in `flat` (device)
int conditioner(SfNsf1& n, const float* flat, const float* x, long rows, float* q, hipStream_t st, std::string& err) {
  for (int t = 0; t < n.T; ++t)
    N1_RC(sf_mlp_forward(n.mlp, flat + (size_t)t * n.P_mlp, x, rows, q + (size_t)t * rows * n.NP, st));
  return SF_OK;
}

}  // namespace

int sf_nsf1_create(const sf_flow_desc& d, SfNsf1** out, std::string& err) {
  if (d.T < 1 || d.T > SF_NSF1_TMAX) { err = "one-parameter NSF: T must be in 1.." + std::to_string(SF_NSF1_TMAX); return SF_ERR_INVALID; }
  if (d.K < 2 || d.K > 16) { err = "K must be in 2..16"; return SF_ERR_INVALID; }
  if (d.H < 1 || d.H > 128 || d.C < 1 || d.C > 512) { err = "H must be in 1..128, C in 1..512"; return SF_ERR_INVALID; }
  if (!d.theta_mean || !d.theta_std || !d.x_mean || !d.x_std) { err = "z-score buffers must be given"; return SF_ERR_INVALID; }
  SfNsf1* n = new SfNsf1();
  n->T = d.T; n->C = d.C; n->H = d.H; n->K = d.K; n->NP = 3 * d.K - 1;
  n->tail_bound = d.tail_bound; n->min_w = d.min_bin_width; n->min_h = d.min_bin_height; n->min_d = d.min_derivative;
  n->inv_sqrt_h = (float)(1.0 / std::sqrt((double)d.H));
  n->deriv_const = (float)std::log(std::exp(1.0 - (double)d.min_derivative) - 1.0);
  n->th_mean = d.theta_mean[0];
  n->th_std = d.theta_std[0];
  sf_mlp_desc md{};
  md.n_in = d.C; md.n_layers = 3; md.widths[0] = d.H; md.widths[1] = d.H; md.widths[2] = n->NP; md.act = SF_ACT_RELU;
  md.x_mean = d.x_mean; md.x_std = d.x_std;
  int rc = sf_mlp_create(&md, &n->mlp);
  if (rc) { err = sf_last_error(); delete n; return rc; }
  n->P_mlp = sf_mlp_num_params(n->mlp);
  *out = n;
  return SF_OK;
}

void sf_nsf1_destroy(SfNsf1* n) {
  if (!n) return;
  sf_mlp_destroy(n->mlp);
  (void)hipFree(n->d_q); (void)hipFree(n->d_dq); (void)hipFree(n->d_xg); (void)hipFree(n->d_thg); (void)hipFree(n->d_cnt);
  delete n;
}

int sf_nsf1_log_prob(SfNsf1* n, const float* flat, const float* theta, const float* x, long B, float* out, hipStream_t st,
                     std::string& err) {
  int rc = grow(n->d_q, n->q_cap, (size_t)n->T * B * n->NP, err);
  if (rc) return rc;
  rc = conditioner(*n, flat, x, B, n->d_q, st, err);
  if (rc) return rc;
  hipLaunchKernelGGL(k_nsf1_logprob, dim3((unsigned)((B + 127) / 128)), dim3(128), 0, st, n->d_q, theta, B, consts_of(*n), out);
  N1_HIP(hipGetLastError());
  return SF_OK;
}

int sf_nsf1_inverse(SfNsf1* n, const float* flat, const float* z, const float* x, long B, float* theta, float* logdet,
                    hipStream_t st, std::string& err) {
  int rc = grow(n->d_q, n->q_cap, (size_t)n->T * B * n->NP, err);
  if (rc) return rc;
  rc = conditioner(*n, flat, x, B, n->d_q, st, err);
  if (rc) return rc;
  hipLaunchKernelGGL(k_nsf1_inverse, dim3((unsigned)((B + 127) / 128)), dim3(128), 0, st, n->d_q, z, B, consts_of(*n), theta, logdet);
  N1_HIP(hipGetLastError());
  return SF_OK;
}

int sf_nsf1_loss_grad(SfNsf1* n, const float* flat, const float* theta, const float* x, const long long* idx, long B,
                      float grad_scale, const float* weights, float* loss, double* loss_sum, float* grad, hipStream_t st,
                      std::string& err) {
  if (B == 0) {
    N1_HIP(hipMemsetAsync(grad, 0, (size_t)n->T * n->P_mlp * sizeof(float), st));
    return SF_OK;
  }
  int rc;
  if (idx) {  // the mini-batch gather (the MLP engine reads contiguous rows)
    if ((rc = grow(n->d_xg, n->xg_cap, (size_t)B * n->C, err)) || (rc = grow(n->d_thg, n->thg_cap, (size_t)B, err))) return rc;
    const long tot = B * (n->C + 1);
    hipLaunchKernelGGL(k_nsf1_gather, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, theta, x, idx, B, n->C, n->d_thg, n->d_xg);
    N1_HIP(hipGetLastError());
    theta = n->d_thg;
    x = n->d_xg;
  }
  if ((rc = grow(n->d_q, n->q_cap, (size_t)n->T * B * n->NP, err)) || (rc = grow(n->d_dq, n->dq_cap, (size_t)n->T * B * n->NP, err))) return rc;
  rc = conditioner(*n, flat, x, B, n->d_q, st, err);
  if (rc) return rc;
  hipLaunchKernelGGL(k_nsf1_train, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, n->d_q, theta, B, consts_of(*n), grad_scale, weights,
                     n->d_dq, loss, loss_sum);
  N1_HIP(hipGetLastError());
  for (int t = 0; t < n->T; ++t)
    N1_RC(sf_mlp_backward(n->mlp, flat + (size_t)t * n->P_mlp, x, n->d_dq + (size_t)t * B * n->NP, B, grad + (size_t)t * n->P_mlp, st));
  return SF_OK;
}

int sf_nsf1_sample(SfNsf1* n, const float* flat, const float* x, long M, long S, const uint32_t* slots, long n_slots,
                   const float* lo, const float* hi, uint32_t k0, uint32_t k1, unsigned long long slot_offset, int max_attempts,
                   float* out, int32_t* n_drawn, int64_t* n_unfilled, hipStream_t st, std::string& err) {
  int rc = grow(n->d_q, n->q_cap, (size_t)n->T * M * n->NP, err);
  if (rc) return rc;
  rc = conditioner(*n, flat, x, M, n->d_q, st, err);
  if (rc) return rc;
  if (!n->d_cnt) N1_HIP(hipMalloc(&n->d_cnt, sizeof(unsigned int)));
  N1_HIP(hipMemsetAsync(n->d_cnt, 0, sizeof(unsigned int), st));
  // no ceiling asked for: 2^22 attempts per slot (the engine's other samplers give a galaxy up once 1e5 attempts of its open
  // slots brought no draw; a scalar slot that failed four million attempts is in the same state)
  const uint32_t cap = max_attempts > 0 ? (uint32_t)max_attempts : (1u << 22);
  hipLaunchKernelGGL(k_nsf1_sample, dim3((unsigned)((n_slots + 127) / 128)), dim3(128), 0, st, n->d_q, M, S, slots, n_slots, consts_of(*n),
                     lo, hi, k0, k1, slot_offset, cap, out, n_drawn, n->d_cnt);
  N1_HIP(hipGetLastError());
  unsigned int h = 0;
  N1_HIP(hipMemcpyAsync(&h, n->d_cnt, sizeof(h), hipMemcpyDeviceToHost, st));
  N1_HIP(hipStreamSynchronize(st));
  if (n_unfilled) *n_unfilled = (int64_t)h;
  return SF_OK;
}

int sf_nsf1_acceptance(SfNsf1* n, const float* flat, const float* x, long M, long cnt, const float* lo, const float* hi, uint32_t k0,
                       uint32_t k1, unsigned long long slot_offset, int32_t* count, hipStream_t st, std::string& err) {
  int rc = grow(n->d_q, n->q_cap, (size_t)n->T * M * n->NP, err);
  if (rc) return rc;
  rc = conditioner(*n, flat, x, M, n->d_q, st, err);
  if (rc) return rc;
  const long tot = M * cnt;
  hipLaunchKernelGGL(k_nsf1_accept, dim3((unsigned)((tot + 127) / 128)), dim3(128), 0, st, n->d_q, M, cnt, consts_of(*n), lo, hi, k0, k1,
                     slot_offset, count);
  N1_HIP(hipGetLastError());
  return SF_OK;
}
