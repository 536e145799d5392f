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

// ---------------------------------------------------------------------------------------------
// Feature transform on the device: fluxes in nJy -> AB magnitudes,
//   mag = -2.5 log10(f / 1000) + 23.9 ; f < 0 (or non-finite result) -> mag_limit ; mag > mag_limit -> mag_limit
// and, optionally, flux errors -> magnitude errors  2.5 sigma / (ln 10 f).
// Replaces the host numpy pass at ref: src/synference/sbi_runner.py:1698-1716, 1927-1932.  HBM-bound, float4 I/O.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sf_abmag(float f_njy, float lim) {
  float m = -2.5f * log10f(f_njy * 1.0e-3f) + 23.9f;
  if (!(f_njy >= 0.f) || !(m == m) || m > lim) m = lim;  // negative flux, NaN and the faint limit
  return m;
}
__global__ void k_flux_to_abmag(const float* __restrict__ flux, const float* __restrict__ err, long n, float lim,
                                float* __restrict__ mag, float* __restrict__ mag_err) {
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      const float4 f = *reinterpret_cast<const float4*>(flux + i);
      *reinterpret_cast<float4*>(mag + i) = make_float4(sf_abmag(f.x, lim), sf_abmag(f.y, lim), sf_abmag(f.z, lim), sf_abmag(f.w, lim));
      if (err) {
        const float4 e = *reinterpret_cast<const float4*>(err + i);
        const float c = 1.0857362047581294f;  // 2.5 / ln 10
        *reinterpret_cast<float4*>(mag_err + i) = make_float4(c * e.x / f.x, c * e.y / f.y, c * e.z / f.z, c * e.w / f.w);
      }
    } else {
      for (long j = i; j < n; ++j) {
        mag[j] = sf_abmag(flux[j], lim);
        if (err) mag_err[j] = 1.0857362047581294f * err[j] / flux[j];
      }
    }
  }
}

extern "C" int sf_flux_to_abmag(const float* flux_njy, const float* err_njy, int64_t n, float mag_limit, float* mag,
                                float* mag_err, void* stream) {
  if (n == 0) return SF_OK;
  if (!flux_njy || !mag || (err_njy && !mag_err)) { sf_set_error("null argument"); return SF_ERR_INVALID; }
  if ((((uintptr_t)flux_njy | (uintptr_t)mag | (uintptr_t)err_njy | (uintptr_t)mag_err) & 15) != 0) {
    sf_set_error("sf_flux_to_abmag: buffers must be 16-byte aligned");
    return SF_ERR_INVALID;
  }
  long blocks = (n / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(k_flux_to_abmag, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, flux_njy, err_njy, (long)n,
                     mag_limit, mag, mag_err);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { sf_set_error(std::string("k_flux_to_abmag: ") + hipGetErrorString(e)); return SF_ERR_HIP; }
  return SF_OK;
}
