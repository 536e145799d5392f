// sf_kernels.hip -- utility kernels and the (kind, HT) dispatch onto the per-TU instantiations
// of sf_inst.hip.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "sf_device.h"
#include "sf_internal.h"

// ---------------------------------------------------------------------------------------------
// small utility kernels
// ---------------------------------------------------------------------------------------------
__global__ void k_pack(const float* __restrict__ flat, const int32_t* __restrict__ s1,
                       const int32_t* __restrict__ s2, float* __restrict__ packed, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int a = s1[i], b = s2[i];
  float v = 0.f;
  if (a >= 0) v = flat[a];
  if (b >= 0) v += flat[b];
  else if (b == SF_PACK_TANH_SCALE) v *= SF_TANH_PRESCALE;
  packed[i] = v;
}

__global__ void k_pack_bf16(const float* __restrict__ flat, const int32_t* __restrict__ src,
                            unsigned short* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int a = src[i];
  const __bf16 v = (__bf16)(a >= 0 ? flat[a] : 0.f);  // round to nearest even
  out[i] = __builtin_bit_cast(unsigned short, v);
}

// split-bf16 pack: src = logical index | (part << 30); part 0 -> hi = bf16(w) (round to nearest even), part 1 ->
// lo = bf16(w - hi).  Two bf16 per 32-bit word of the image.
__global__ void k_pack_bf16_split(const float* __restrict__ flat, const int32_t* __restrict__ src,
                                  unsigned short* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int a = src[i];
  unsigned short r = 0;
  if (a >= 0) {
    const float w = flat[a & SF_PACK_SPLIT_INDEX] * ((a & SF_PACK_SPLIT_SCALED) ? SF_TANH_PRESCALE : 1.0f);
    const __bf16 hi = (__bf16)w;
    r = __builtin_bit_cast(unsigned short, (a >> 30) & 1 ? (__bf16)(w - (float)hi) : hi);
  }
  out[i] = r;
}

__global__ void k_fill_nan_rows(float* __restrict__ out, const uint32_t* __restrict__ slots, long n, int D, int out_f64) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int d = 0; d < D; ++d) {
    if (out_f64) reinterpret_cast<double*>(out)[(long)slots[i] * D + d] = __builtin_nan("");
    else out[(long)slots[i] * D + d] = __builtin_nanf("");
  }
}

// Progress rule between the stages of an uncapped sampling call (sf_api.hip): one workgroup walks the survivor list;
// slots of galaxies that got no draw accepted in the stage are written as NaN rows, the rest are compacted in place
// (stable order), and the survivor count is updated.  The list is short (it is what more than 64 attempts left over).
__global__ __launch_bounds__(1024) void k_filter_survivors(uint32_t* __restrict__ list, unsigned int* __restrict__ n_surv,
                                                          long S, const int32_t* __restrict__ gal_acc,
                                                          float* __restrict__ out, int D, int out_f64) {
  __shared__ unsigned int warp_cnt[16];
  __shared__ unsigned int base_s;
  const unsigned int n = *n_surv;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) base_s = 0;
  __syncthreads();
  for (unsigned int i0 = 0; i0 < n; i0 += 1024) {
    const unsigned int i = i0 + threadIdx.x;
    uint32_t slot = 0;
    bool keep = false;
    if (i < n) {
      slot = list[i];
      keep = gal_acc[(long)(slot / (uint32_t)S)] > 0;
      if (!keep)
        for (int d = 0; d < D; ++d) {
          if (out_f64) reinterpret_cast<double*>(out)[(size_t)slot * D + d] = __builtin_nan("");
          else out[(size_t)slot * D + d] = __builtin_nanf("");
        }
    }
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) warp_cnt[w] = (unsigned)__popcll(bal);
    __syncthreads();  // every read of this chunk of the list has happened
    unsigned int off = base_s;
    for (int q = 0; q < w; ++q) off += warp_cnt[q];
    if (keep) list[off + (unsigned)__popcll(bal & ((1ull << lane) - 1ull))] = slot;  // off + rank <= i: never ahead of the reads
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned int t = 0;
      for (int q = 0; q < 16; ++q) t += warp_cnt[q];
      base_s += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    n_surv[1] = n - base_s;  // SfQueue::dropped sits right behind n_surv
    *n_surv = base_s;
  }
}

// Bookkeeping between the find and the resolve launch of a deep-tail window (sf_api.hip): attempts consumed per galaxy
// and the per-galaxy progress counter.
__global__ void k_account_window(const uint32_t* __restrict__ list, const uint32_t* __restrict__ best, long n, long S,
                                 uint32_t a_lo, uint32_t A, int32_t* __restrict__ n_drawn, int32_t* __restrict__ gal_acc) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long g = (long)(list[i] / (uint32_t)S);
  const uint32_t b = best[i];
  if (n_drawn) sf_sat_add(&n_drawn[g], b != 0xffffffffu ? (int)(b - a_lo + 1u) : (int)A);
  if (gal_acc && b != 0xffffffffu) atomicAdd(&gal_acc[g], 1);
}

__global__ void k_fill_i32(int32_t* p, long n, int32_t v) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ---------------------------------------------------------------------------------------------
// dispatch on (kind, HT); PT and NS are resolved inside the instantiation TUs
// ---------------------------------------------------------------------------------------------
#define SF_DECL(K, H)                                                                                   \
  hipError_t sf_launch_logprob_k##K##_h##H(const SfDev&, int, const float*, const float*, long, float*, \
                                           hipStream_t);                                                \
  hipError_t sf_launch_inverse_k##K##_h##H(const SfDev&, int, const SfSampleArgsHost&, hipStream_t);
SF_DECL(0, 1) SF_DECL(0, 2) SF_DECL(0, 3) SF_DECL(0, 4)
SF_DECL(1, 1) SF_DECL(1, 2) SF_DECL(1, 3) SF_DECL(1, 4)
hipError_t sf_launch_ctab_k1_h1(const SfDev&, const float*, long, float*, hipStream_t);
hipError_t sf_launch_ctab_k1_h2(const SfDev&, const float*, long, float*, hipStream_t);
hipError_t sf_launch_ctab_k1_h3(const SfDev&, const float*, long, float*, hipStream_t);
hipError_t sf_launch_ctab_k1_h4(const SfDev&, const float*, long, float*, hipStream_t);

// sample tiles per wave: SF_NS=1|2 overrides (diagnostics); default 2 while HT <= 2
int sf_pick_ns(const SfDev& m, bool inverse) {
  static int forced = -1;
  if (forced < 0) {
    const char* e = std::getenv("SF_NS");
    forced = e ? std::atoi(e) : 0;
  }
  if (m.HT > 2) return 1;
  if (forced == 1 || forced == 2) return forced;
  // measured on MI355X: with the operand image in LDS (512-thread workgroups) one 32-sample tile per wave
  // is faster for every kernel (fewer registers, no spills); two tiles per wave only pay off when the
  // weights stream from L2 (oversized images)
  if (m.n_parts > 0) return 1;
  return 2;
}

#define SF_CASE(K, H, FN, ...) \
  case K * 10 + H: return FN##_k##K##_h##H(__VA_ARGS__);

hipError_t sf_launch_logprob(const SfDev& m, const float* theta, const float* x, long B, float* out,
                             hipStream_t st) {
  if (B <= 0) return hipSuccess;
  const int ns = sf_pick_ns(m, false);
  switch (m.kind * 10 + m.HT) {
    SF_CASE(0, 1, sf_launch_logprob, m, ns, theta, x, B, out, st)
    SF_CASE(0, 2, sf_launch_logprob, m, ns, theta, x, B, out, st)
    SF_CASE(0, 3, sf_launch_logprob, m, ns, theta, x, B, out, st)
    SF_CASE(0, 4, sf_launch_logprob, m, ns, theta, x, B, out, st)
    SF_CASE(1, 1, sf_launch_logprob, m, ns, theta, x, B, out, st)
    SF_CASE(1, 2, sf_launch_logprob, m, ns, theta, x, B, out, st)
    SF_CASE(1, 3, sf_launch_logprob, m, ns, theta, x, B, out, st)
    SF_CASE(1, 4, sf_launch_logprob, m, ns, theta, x, B, out, st)
  }
  return hipErrorInvalidValue;
}

hipError_t sf_launch_inverse(const SfDev& m, const SfSampleArgsHost& a_in, hipStream_t st) {
  if (a_in.n_items <= 0) return hipSuccess;
  SfSampleArgsHost a = a_in;
  a.log2_attempts = 0;
  while ((1 << a.log2_attempts) < a.attempts_per_slot) ++a.log2_attempts;
  if (sf_maf16_enabled(m, a)) return sf_launch_maf_inv16(m, a, st);
  const int ns = sf_pick_ns(m, true);
  switch (m.kind * 10 + m.HT) {
    SF_CASE(0, 1, sf_launch_inverse, m, ns, a, st)
    SF_CASE(0, 2, sf_launch_inverse, m, ns, a, st)
    SF_CASE(0, 3, sf_launch_inverse, m, ns, a, st)
    SF_CASE(0, 4, sf_launch_inverse, m, ns, a, st)
    SF_CASE(1, 1, sf_launch_inverse, m, ns, a, st)
    SF_CASE(1, 2, sf_launch_inverse, m, ns, a, st)
    SF_CASE(1, 3, sf_launch_inverse, m, ns, a, st)
    SF_CASE(1, 4, sf_launch_inverse, m, ns, a, st)
  }
  return hipErrorInvalidValue;
}

// per-galaxy context table (sampling): which flows have one, and its builder
void sf_ctab_shape(const SfDev& m, int& R, int& NV) {
  R = 0; NV = 0;
  if (m.hidden_bf16 == 1) return;
  if (m.kind == SF_NSF) { R = m.HT * 32; NV = 1 + m.NB; }
  // (MAF with the fused first layer: a row holds c0 and, behind it, c0' = b1 + (W1 o M) c0)
  else if (m.kind == SF_MAF && m.m16_ok && m.packed16 != nullptr) { R = m.nT16 * 16 * (m.o16_wp >= 0 ? 2 : 1); NV = 1; }
}
hipError_t sf_launch_ctab(const SfDev& m, const float* x, long M, float* tab, hipStream_t st) {
  if (M <= 0) return hipSuccess;
  if (m.kind == SF_MAF) return sf_launch_maf_ctab16(m, x, M, tab, st);
  switch (m.HT) {
    case 1: return sf_launch_ctab_k1_h1(m, x, M, tab, st);
    case 2: return sf_launch_ctab_k1_h2(m, x, M, tab, st);
    case 3: return sf_launch_ctab_k1_h3(m, x, M, tab, st);
    case 4: return sf_launch_ctab_k1_h4(m, x, M, tab, st);
  }
  return hipErrorInvalidValue;
}
hipError_t sf_launch_pack(const float* flat, const int32_t* s1, const int32_t* s2, float* packed, long n,
                          hipStream_t st) {
  hipLaunchKernelGGL(k_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, flat, s1, s2, packed, n);
  return hipGetLastError();
}
hipError_t sf_launch_pack_bf16(const float* flat, const int32_t* src, unsigned short* out, long n, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_pack_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, flat, src, out, n);
  return hipGetLastError();
}
hipError_t sf_launch_pack_bf16_split(const float* flat, const int32_t* src, unsigned short* out, long n, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_pack_bf16_split, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, flat, src, out, n);
  return hipGetLastError();
}
hipError_t sf_launch_fill_nan_rows(float* out, const uint32_t* slots, long n, int D, hipStream_t st, int out_f64) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_fill_nan_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, slots, n, D, out_f64);
  return hipGetLastError();
}
hipError_t sf_launch_filter_survivors(uint32_t* list, unsigned int* n_surv, long S, const int32_t* gal_acc, float* out,
                                      int D, hipStream_t st, int out_f64) {
  hipLaunchKernelGGL(k_filter_survivors, dim3(1), dim3(1024), 0, st, list, n_surv, S, gal_acc, out, D, out_f64);
  return hipGetLastError();
}
hipError_t sf_launch_account_window(const uint32_t* list, const uint32_t* best, long n, long S, uint32_t a_lo, uint32_t A,
                                    int32_t* n_drawn, int32_t* gal_acc, hipStream_t st) {
  if (n <= 0 || (!n_drawn && !gal_acc)) return hipSuccess;
  hipLaunchKernelGGL(k_account_window, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, list, best, n, S, a_lo, A, n_drawn, gal_acc);
  return hipGetLastError();
}
hipError_t sf_launch_fill_i32(int32_t* p, long n, int32_t v, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n, v);
  return hipGetLastError();
}
