// sf_inst.hip -- one translation unit per (SF_KIND, SF_HT): instantiates the inference kernels
// for that hidden-tile count and exports plain launchers (sf_internal.h, sf_launch_*_k?_h?).
// NS (32-sample tiles per wave) is a runtime choice among the instantiated values.
#include <cstdlib>

#include "sf_inst_templates.h"

#ifndef SF_KIND
#error "compile with -DSF_KIND=0|1 -DSF_HT=1..4"
#endif

#define SF_CAT_(a, b, c, d) a##b##c##d
#define SF_CAT(a, b, c, d) SF_CAT_(a, b, c, d)

// LDS-staged variant when one transform's operand image fits the 160 KiB LDS (with slack)
static inline bool sf_fits_lds(const SfDev& m) { return m.n_parts > 0; }
// waves per workgroup of the LDS-staged kernels: 8 (one workgroup per CU) unless two 4-wave workgroups
// fit the LDS side by side -- then one workgroup's staging barriers overlap the other's compute.
// SF_WPB=4|8 overrides (diagnostics).
static inline int sf_lds_wpb(size_t shmem_bytes) {
  static int forced = -1;
  if (forced < 0) { const char* e = std::getenv("SF_WPB"); forced = e ? std::atoi(e) : 0; }
  if (forced == 4 || forced == 8) return forced;
  return 2 * shmem_bytes <= 156 * 1024 ? 4 : 8;
}

template <class K>
static hipError_t set_shmem(K kernel, size_t bytes, SfAttrCache& done) {
  int dev;
  if (!done.need(dev)) return hipSuccess;
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) done.set(dev);
  return e;
}


// ---- persistent sampler launches (a.q != nullptr): grid = what the chip holds at once ------------------------
static int sf_cu_count() {
  int dev = 0;
  hipDeviceProp_t pr;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
    return pr.multiProcessorCount;
  return 256;
}
template <class Ops, int NS, bool LDSW, int WPB>
static hipError_t launch_persist_w(const SfDev& m, const SfSampleArgsHost& a, size_t image_bytes, hipStream_t st) {
  static SfAttrCache attr;
  static SfResidentCache rcache;
  constexpr int IPW = WPB * 32 * NS;
  const size_t sh = image_bytes + SF_Q_WORDS(IPW) * sizeof(unsigned int);
  hipError_t e = set_shmem(k_sample_persist<Ops, NS, LDSW, WPB>, sh, attr);
  if (e != hipSuccess) return e;
  int resident = 0, cur_dev = 0;
  (void)hipGetDevice(&cur_dev);
  if (!rcache.get(cur_dev, sh, resident)) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_sample_persist<Ops, NS, LDSW, WPB>, 64 * WPB, sh) !=
            hipSuccess || occ < 1)
      occ = 1;
    resident = sf_cu_count() * occ;
    rcache.put(cur_dev, sh, resident);
  }
  long grid = (a.n_items + IPW - 1) / IPW;
  if (grid > resident) grid = resident;
  SfSampArgs args;
  args.m = m;
  args.a = a;
  hipLaunchKernelGGL((k_sample_persist<Ops, NS, LDSW, WPB>), dim3((unsigned)grid), dim3(64 * WPB), sh, st, args,
                     (int)(image_bytes / sizeof(float)));
  return hipGetLastError();
}
template <class OpsL>
static hipError_t launch_persist_lds(const SfDev& m, const SfSampleArgsHost& a, size_t image_bytes, hipStream_t st) {
  image_bytes = (image_bytes + 15) & ~(size_t)15;
  if (sf_lds_wpb(image_bytes + SF_Q_WORDS(128) * sizeof(unsigned int)) == 4) return launch_persist_w<OpsL, 1, true, 4>(m, a, image_bytes, st);
  return launch_persist_w<OpsL, 1, true, 8>(m, a, image_bytes, st);
}

template <class OpsB>
static hipError_t launch_logprob_bf16(const SfDev& m, const float* theta, const float* x, long B, float* out,
                                      hipStream_t st) {
  static SfAttrCache attr;
  const size_t sh = (size_t)m.part_bytes_max;
  hipError_t e = set_shmem(k_logprob<OpsB, 1, true>, sh, attr);
  if (e != hipSuccess) return e;
  const int wpb = sf_lds_wpb(sh);
  hipLaunchKernelGGL((k_logprob<OpsB, 1, true>), dim3((unsigned)((B + 32 * wpb - 1) / (32 * wpb))), dim3(64 * wpb), sh, st,
                     m, theta, x, B, out);
  return hipGetLastError();
}
template <class OpsB>
static hipError_t launch_inverse_bf16(const SfDev& m, const SfSampleArgsHost& a, hipStream_t st) {
  static SfAttrCache attr;
  const size_t sh = (size_t)m.part_bytes_max;
  if (a.q) return launch_persist_lds<OpsB>(m, a, sh, st);
  hipError_t e = set_shmem(k_inverse<OpsB, 1, true>, sh, attr);
  if (e != hipSuccess) return e;
  const int wpb = sf_lds_wpb(sh);
  hipLaunchKernelGGL((k_inverse<OpsB, 1, true>), dim3((unsigned)((a.n_items + 32 * wpb - 1) / (32 * wpb))), dim3(64 * wpb),
                     sh, st, m, a);
  return hipGetLastError();
}

template <class OpsG, class OpsL, int NS>
static hipError_t launch_logprob(const SfDev& m, const float* theta, const float* x, long B, float* out,
                                 hipStream_t st) {
  if (sf_fits_lds(m)) {
    static SfAttrCache attr;
    const size_t sh = (size_t)m.part_max * sizeof(float);
    hipError_t e = set_shmem(k_logprob<OpsL, NS, true>, sh, attr);
    if (e != hipSuccess) return e;
    const int wpb = sf_lds_wpb(sh);
    const long per_block = (long)wpb * 32 * NS;
    hipLaunchKernelGGL((k_logprob<OpsL, NS, true>), dim3((unsigned)((B + per_block - 1) / per_block)), dim3(64 * wpb),
                       sh, st, m, theta, x, B, out);
  } else {
    const long per_block = 4L * 32 * NS;
    hipLaunchKernelGGL((k_logprob<OpsG, NS, false>), dim3((unsigned)((B + per_block - 1) / per_block)), dim3(256),
                       0, st, m, theta, x, B, out);
  }
  return hipGetLastError();
}
template <class OpsG, class OpsL, int NS>
static hipError_t launch_inverse(const SfDev& m, const SfSampleArgsHost& a, hipStream_t st) {
  if constexpr (NS == 1) {
    if (a.q) {  // persistent sampler (one 32-sample tile per wave)
      if (sf_fits_lds(m)) return launch_persist_lds<OpsL>(m, a, (size_t)m.part_max * sizeof(float), st);
      return launch_persist_w<OpsG, 1, false, 4>(m, a, 0, st);
    }
  }
  if (sf_fits_lds(m)) {
    static SfAttrCache attr;
    const size_t sh = (size_t)m.part_max * sizeof(float);
    hipError_t e = set_shmem(k_inverse<OpsL, NS, true>, sh, attr);
    if (e != hipSuccess) return e;
    const int wpb = sf_lds_wpb(sh);
    const long per_block = (long)wpb * 32 * NS;
    hipLaunchKernelGGL((k_inverse<OpsL, NS, true>), dim3((unsigned)((a.n_items + per_block - 1) / per_block)),
                       dim3(64 * wpb), sh, st, m, a);
  } else {
    const long per_block = 4L * 32 * NS;
    hipLaunchKernelGGL((k_inverse<OpsG, NS, false>), dim3((unsigned)((a.n_items + per_block - 1) / per_block)),
                       dim3(256), 0, st, m, a);
  }
  return hipGetLastError();
}

#if SF_KIND == 0
#define SF_BF_SWITCH(FN, ...) { using OB = MafOps<SF_HT, 1, true, true>; return FN<OB>(__VA_ARGS__); }
#define SF_BFS_SWITCH(FN, ...) SF_BF_SWITCH(FN, __VA_ARGS__)
#else
// (the last template argument of NsfOps: the transform's image is staged in several parts -- sf_flows.h)
#define SF_NSF_MP(FN, PT_, BF_, ...)                                                                      \
  if (m.n_parts > 1) { using OB = NsfOps<SF_HT, PT_, 1, true, BF_, true>; return FN<OB>(__VA_ARGS__); }   \
  else { using OB = NsfOps<SF_HT, PT_, 1, true, BF_, false>; return FN<OB>(__VA_ARGS__); }
#define SF_BF_SWITCH(FN, ...)                                                              \
  switch (m.PT) {                                                                          \
    case 2: SF_NSF_MP(FN, 2, 1, __VA_ARGS__)                                               \
    case 3: SF_NSF_MP(FN, 3, 1, __VA_ARGS__)                                               \
    default: return hipErrorInvalidValue;                                                  \
  }
// sampling kernels: single bf16 (opt-in) or split bf16 x3 (the NSF sampler image)
#define SF_BFS_SWITCH(FN, ...)                                                             \
  switch (m.PT * 10 + m.hidden_bf16) {                                                     \
    case 21: SF_NSF_MP(FN, 2, 1, __VA_ARGS__)                                              \
    case 31: SF_NSF_MP(FN, 3, 1, __VA_ARGS__)                                              \
    case 22: SF_NSF_MP(FN, 2, 2, __VA_ARGS__)                                              \
    case 32: SF_NSF_MP(FN, 3, 2, __VA_ARGS__)                                              \
    default: return hipErrorInvalidValue;                                                  \
  }
#endif

#if SF_KIND == 0
#define SF_PT_SWITCH(NS, FN, ...)                                                          \
  { using OG = MafOps<SF_HT, NS, false>; using OL = MafOps<SF_HT, NS, true>;               \
    return FN<OG, OL, NS>(__VA_ARGS__); }
#else
#define SF_NSF_PT(NS, FN, PT_, ...)                                                                                              \
  if (m.n_parts > 1) { using OG = NsfOps<SF_HT, PT_, NS, false>; using OL = NsfOps<SF_HT, PT_, NS, true, 0, true>;               \
                       return FN<OG, OL, NS>(__VA_ARGS__); }                                                                     \
  else { using OG = NsfOps<SF_HT, PT_, NS, false>; using OL = NsfOps<SF_HT, PT_, NS, true, 0, false>;                            \
         return FN<OG, OL, NS>(__VA_ARGS__); }
#define SF_PT_SWITCH(NS, FN, ...)                                                          \
  switch (m.PT) {                                                                          \
    case 2: SF_NSF_PT(NS, FN, 2, __VA_ARGS__)                                              \
    case 3: SF_NSF_PT(NS, FN, 3, __VA_ARGS__)                                              \
    default: return hipErrorInvalidValue;                                                  \
  }
#endif

hipError_t SF_CAT(sf_launch_logprob_k, SF_KIND, _h, SF_HT)(const SfDev& m, int ns, const float* theta,
                                                           const float* x, long B, float* out,
                                                           hipStream_t st) {
  if (m.hidden_bf16) SF_BF_SWITCH(launch_logprob_bf16, m, theta, x, B, out, st)
#if SF_HT <= 2
  if (ns == 2) SF_PT_SWITCH(2, launch_logprob, m, theta, x, B, out, st)
#endif
  SF_PT_SWITCH(1, launch_logprob, m, theta, x, B, out, st)
}

hipError_t SF_CAT(sf_launch_inverse_k, SF_KIND, _h, SF_HT)(const SfDev& m, int ns, const SfSampleArgsHost& a,
                                                           hipStream_t st) {
  if (m.hidden_bf16) SF_BFS_SWITCH(launch_inverse_bf16, m, a, st)
#if SF_HT <= 2
  if (ns == 2 && !a.q) SF_PT_SWITCH(2, launch_inverse, m, a, st)
#endif
  SF_PT_SWITCH(1, launch_inverse, m, a, st)
}

#if SF_KIND == 1
hipError_t SF_CAT(sf_launch_ctab_k, SF_KIND, _h, SF_HT)(const SfDev& m, const float* x, long M, float* tab, hipStream_t st) {
  hipLaunchKernelGGL((k_nsf_ctab<SF_HT>), dim3((unsigned)((M + 127) / 128)), dim3(256), 0, st, m, x, M, tab);
  return hipGetLastError();
}
#endif
