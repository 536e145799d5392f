// sf_post.hip -- posterior summaries on the device: per (galaxy, parameter)
// quantiles of the S draws with numpy's default 'linear' rule, NaN draws ignored (all-NaN -> NaN).
// Replaces the host pass  np.quantile(samples_i, [0.16, 0.5, 0.84], axis=1)
// (ref: src/synference/sbi_runner.py:3250-3282) without materialising (N,S,D) on the host.
#include <hip/hip_runtime.h>

#include <string>

#include "sf_internal.h"
#include "sf_rng.h"

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
//   mag = -2.5 log10(f / 1000) + 23.9 ; f < 0 -> mag_limit ; mag > mag_limit (incl. f == 0 -> +inf) -> mag_limit ;
//   a NaN flux stays NaN (the reference only replaces negative fluxes, sbi_runner.py:1706-1714, so that rows with a
//   missing band are still recognised and masked downstream)
// and, optionally, flux errors -> magnitude errors  2.5 sigma / (ln 10 f).
// Replaces the host numpy pass at ref: src/synference/sbi_runner.py:1698-1716, 1927-1932.  HBM-bound, float4 I/O.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sf_abmag(float f_njy, float lim) {
  if (f_njy != f_njy) return f_njy;  // missing band: NaN propagates
  float m = -2.5f * log10f(f_njy * 1.0e-3f) + 23.9f;
  if (f_njy < 0.f || m > lim) m = lim;  // negative flux and the faint limit (f == 0 gives +inf > lim)
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

// ---------------------------------------------------------------------------------------------
// asinh ("luptitude") magnitudes, per-band softening f_b (ref: src/synference/utils.py:647-704, used at
// sbi_runner.py:1718-1731):
//   mag = -2.5 log10(e) * ( asinh(f / (2 f_b)) + ln(f_b / 3631 Jy) ),  err = 2.5 log10(e) * sigma / sqrt(f^2 + (2 f_b)^2)
// flux, err [N,C] row-major in nJy; f_b [C] in nJy (device).  HBM-bound elementwise pass.
// ---------------------------------------------------------------------------------------------
__global__ void k_flux_to_asinh(const float* __restrict__ flux, const float* __restrict__ err, long n, int C,
                                const float* __restrict__ f_b, float* __restrict__ mag, float* __restrict__ mag_err) {
  const float k = 1.0857362047581294f;  // 2.5 log10(e)
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float b = f_b[i % C], f = flux[i];
    const float r = f / (2.f * b);
    mag[i] = -k * (asinhf(r) + logf(b * (1.0e-9f / 3631.f)));
    if (err) mag_err[i] = k * err[i] / sqrtf(f * f + 4.f * b * b);
  }
}

extern "C" int sf_flux_to_asinh(const float* flux_njy, const float* err_njy, int64_t N, int32_t C, const float* f_b_njy,
                                float* mag, float* mag_err, void* stream) {
  if (N == 0) return SF_OK;
  if (!flux_njy || !mag || !f_b_njy || (err_njy && !mag_err)) { sf_set_error("null argument"); return SF_ERR_INVALID; }
  if (N < 0 || C < 1) { sf_set_error("sf_flux_to_asinh: bad shape"); return SF_ERR_INVALID; }
  const long n = (long)N * C;
  long blocks = (n + 255) / 256;
  blocks = blocks > 4096 ? 4096 : blocks;
  hipLaunchKernelGGL(k_flux_to_asinh, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, flux_njy, err_njy, n, (int)C,
                     f_b_njy, mag, mag_err);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { sf_set_error(std::string("k_flux_to_asinh: ") + hipGetErrorString(e)); return SF_ERR_HIP; }
  return SF_OK;
}

// ---------------------------------------------------------------------------------------------
// Depth-noise scatter (ref: sbi_runner.py:580-691 `_apply_depths`, 1-D depths): every library row is repeated
// n_scatters times and perturbed by N(0, sigma_c), sigma_c = max(depth_c / depth_sigma, |flux| * min_pc / 100);
// the sigma used is returned as the error column.  out row = row * n_scatters + s.  Noise: Philox stream
// (seed, stream 2; counter = (out_row, band block)), so a catalogue is reproducible and independent of launch shape
// (the reference draws from numpy's global generator).
// ---------------------------------------------------------------------------------------------
__global__ void k_scatter_depths(const float* __restrict__ flux, long N, int C, const float* __restrict__ sigma, int n_sig,
                                 int n_scatters, float min_pc, uint32_t k0, uint32_t k1, float* __restrict__ out,
                                 float* __restrict__ err_out) {
  const int CB = (C + 3) / 4;
  const long total = N * n_scatters * CB;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long orow = i / CB;
    const int cb = (int)(i % CB);
    const long row = orow / n_scatters;
    const float* sg = sigma + (size_t)((orow % n_scatters) % n_sig) * C;  // one sigma row, or one per scatter copy
    float z[4];
    sf_normal4(k0, k1, (uint64_t)orow, 0u, (uint32_t)cb, z);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = cb * 4 + j;
      if (c < C) {
        const float f = flux[row * C + c];
        const float s = fmaxf(sg[c], fabsf(f) * min_pc * 0.01f);
        out[orow * C + c] = f + s * z[j];
        if (err_out) err_out[orow * C + c] = s;
      }
    }
  }
}

extern "C" int sf_scatter_depths(const float* flux, int64_t N, int32_t C, const float* sigma, int32_t n_sigma_rows,
                                 int32_t n_scatters, float min_flux_pc_error, uint64_t seed, float* out, float* err_out,
                                 void* stream) {
  if (N == 0) return SF_OK;
  if (!flux || !sigma || !out) { sf_set_error("null argument"); return SF_ERR_INVALID; }
  if (N < 0 || C < 1 || n_scatters < 1 || (n_sigma_rows != 1 && n_sigma_rows != n_scatters)) {
    sf_set_error("sf_scatter_depths: bad shape (sigma rows must be 1 or n_scatters)");
    return SF_ERR_INVALID;
  }
  const long total = (long)N * n_scatters * ((C + 3) / 4);
  long blocks = (total + 255) / 256;
  blocks = blocks > 8192 ? 8192 : blocks;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ 2u;
  hipLaunchKernelGGL(k_scatter_depths, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, flux, (long)N, (int)C, sigma,
                     (int)n_sigma_rows, (int)n_scatters, min_flux_pc_error, k0, k1, out, err_out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { sf_set_error(std::string("k_scatter_depths: ") + hipGetErrorString(e)); return SF_ERR_HIP; }
  return SF_OK;
}

// ---------------------------------------------------------------------------------------------
// PIT ranks (ref: sbi_runner.py:7153-7158): rank[g,d] = #{ s : draw[g,s,d] < truth[g,d] } / #{finite draws}
// one wave per (galaxy, dim); NaN draws ignored (all-NaN -> NaN).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pit_ranks(const float* __restrict__ samples, const float* __restrict__ truth, long N,
                                                   long S, int D, float* __restrict__ out) {
  const long w = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (w >= N * D) return;
  const long g = w / D;
  const int d = (int)(w % D);
  const float t = truth[g * D + d];
  int below = 0, valid = 0;
  for (long s = lane; s < S; s += 64) {
    const float v = samples[(g * S + s) * D + d];
    valid += (v == v) ? 1 : 0;
    below += (v < t) ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    below += __shfl_xor(below, o, 64);
    valid += __shfl_xor(valid, o, 64);
  }
  if (lane == 0) out[w] = valid > 0 ? (float)below / (float)valid : __builtin_nanf("");
}

extern "C" int sf_pit_ranks(const float* samples, const float* truth, int64_t N, int64_t S, int32_t D, float* out,
                            void* stream) {
  if (N == 0) return SF_OK;
  if (!samples || !truth || !out) { sf_set_error("null argument"); return SF_ERR_INVALID; }
  if (N < 0 || S < 1 || D < 1) { sf_set_error("sf_pit_ranks: bad shape"); return SF_ERR_INVALID; }
  const long waves = (long)N * D;
  hipLaunchKernelGGL(k_pit_ranks, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, samples, truth, (long)N,
                     (long)S, (int)D, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { sf_set_error(std::string("k_pit_ranks: ") + hipGetErrorString(e)); return SF_ERR_HIP; }
  return SF_OK;
}
