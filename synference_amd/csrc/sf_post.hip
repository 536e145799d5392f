// sf_post.hip -- posterior summaries on the device: per (galaxy, parameter)
// quantiles of the S draws with numpy's default 'linear' rule, NaN draws ignored (all-NaN -> NaN).
// Replaces the host pass  np.quantile(samples_i, [0.16, 0.5, 0.84], axis=1)
// (ref: src/synference/sbi_runner.py:3250-3282) without materialising (N,S,D) on the host.
#include <hip/hip_runtime.h>

#include <string>

#include "sf_internal.h"

// one 256-thread workgroup per (galaxy, dim); P = S padded to a power of two (<= 8192), bitonic sort in LDS
__global__ __launch_bounds__(256) void k_quantiles(const float* __restrict__ samples, long S, int D, int P,
                                                   const float* __restrict__ q, int Q, float* __restrict__ out) {
  extern __shared__ float v[];
  const long g = blockIdx.x / D;
  const int d = blockIdx.x % D;
  const float* src = samples + g * S * D + d;
  const float INF = __builtin_inff();
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    float x = i < S ? src[(long)i * D] : INF;
    v[i] = (x == x) ? x : INF;  // NaN -> +inf: sorted to the end and not counted
  }
  __shared__ int n_valid;
  if (threadIdx.x == 0) n_valid = 0;
  __syncthreads();
  int cnt = 0;
  for (int i = threadIdx.x; i < S; i += blockDim.x) cnt += (v[i] < INF) ? 1 : 0;
  atomicAdd(&n_valid, cnt);
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int i = threadIdx.x; i < P; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const float a = v[i], b = v[ixj];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { v[i] = b; v[ixj] = a; }
        }
      }
    }
  }
  __syncthreads();
  const int n = n_valid;
  for (int t = threadIdx.x; t < Q; t += blockDim.x) {
    float r = __builtin_nanf("");
    if (n > 0) {
      const float hpos = (float)(n - 1) * q[t];
      int lo = (int)floorf(hpos);
      lo = lo < 0 ? 0 : (lo > n - 1 ? n - 1 : lo);
      const int hi = lo + 1 < n ? lo + 1 : n - 1;
      r = v[lo] + (hpos - (float)lo) * (v[hi] - v[lo]);
    }
    out[(g * D + d) * Q + t] = r;
  }
}

extern "C" int sf_quantiles(const float* samples, int64_t N, int64_t S, int32_t D, const float* q_dev, int32_t Q,
                            float* out, void* stream) {
  if (N == 0) return SF_OK;
  if (!samples || !q_dev || !out) { sf_set_error("null argument"); return SF_ERR_INVALID; }
  if (S < 1 || S > 8192 || D < 1 || Q < 1 || Q > 256) {
    sf_set_error("sf_quantiles: need 1 <= S <= 8192, D >= 1, 1 <= Q <= 256");
    return SF_ERR_INVALID;
  }
  if ((uint64_t)N * (uint64_t)D > 0x7fffffffull) { sf_set_error("N*D too large for one launch"); return SF_ERR_INVALID; }
  int P = 1;
  while (P < S) P <<= 1;
  hipLaunchKernelGGL(k_quantiles, dim3((unsigned)(N * D)), dim3(256), (size_t)P * sizeof(float), (hipStream_t)stream,
                     samples, (long)S, (int)D, P, q_dev, (int)Q, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { sf_set_error(std::string("k_quantiles: ") + hipGetErrorString(e)); return SF_ERR_HIP; }
  return SF_OK;
}
