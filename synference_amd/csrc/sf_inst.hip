// sf_inst.hip -- one translation unit per (SF_KIND, SF_HT): instantiates the inference kernels
// for that hidden-tile count and exports plain launchers (sf_internal.h, sf_launch_*_k?_h?).
// NS (32-sample tiles per wave) is a runtime choice among the instantiated values.
#include "sf_inst_templates.h"

#ifndef SF_KIND
#error "compile with -DSF_KIND=0|1 -DSF_HT=1..4"
#endif

#define SF_CAT_(a, b, c, d) a##b##c##d
#define SF_CAT(a, b, c, d) SF_CAT_(a, b, c, d)

template <class Ops, int NS>
static hipError_t launch_logprob(const SfDev& m, const float* theta, const float* x, long B, float* out,
                                 hipStream_t st) {
  const long per_block = 4L * 32 * NS;
  const long grid = (B + per_block - 1) / per_block;
  hipLaunchKernelGGL((k_logprob<Ops, NS>), dim3((unsigned)grid), dim3(256), 0, st, m, theta, x, B, out);
  return hipGetLastError();
}
template <class Ops, int NS>
static hipError_t launch_inverse(const SfDev& m, const SfSampleArgsHost& a, hipStream_t st) {
  const long per_block = 4L * 32 * NS;
  const long grid = (a.n_items + per_block - 1) / per_block;
  hipLaunchKernelGGL((k_inverse<Ops, NS>), dim3((unsigned)grid), dim3(256), 0, st, m, a);
  return hipGetLastError();
}

#if SF_KIND == 0
#define OPS(NS) MafOps<SF_HT, NS>
#define SF_PT_SWITCH(NS, CALL) { using O = OPS(NS); return CALL; }
#else
#define SF_PT_SWITCH(NS, CALL)                                          \
  switch (m.PT) {                                                       \
    case 2: { using O = NsfOps<SF_HT, 2, NS>; return CALL; }            \
    case 3: { using O = NsfOps<SF_HT, 3, NS>; return CALL; }            \
    default: return hipErrorInvalidValue;                               \
  }
#endif

hipError_t SF_CAT(sf_launch_logprob_k, SF_KIND, _h, SF_HT)(const SfDev& m, int ns, const float* theta,
                                                           const float* x, long B, float* out,
                                                           hipStream_t st) {
#if SF_HT <= 2
  if (ns == 2) SF_PT_SWITCH(2, (launch_logprob<O, 2>(m, theta, x, B, out, st)))
#endif
  SF_PT_SWITCH(1, (launch_logprob<O, 1>(m, theta, x, B, out, st)))
}

hipError_t SF_CAT(sf_launch_inverse_k, SF_KIND, _h, SF_HT)(const SfDev& m, int ns, const SfSampleArgsHost& a,
                                                           hipStream_t st) {
#if SF_HT <= 2
  if (ns == 2) SF_PT_SWITCH(2, (launch_inverse<O, 2>(m, a, st)))
#endif
  SF_PT_SWITCH(1, (launch_inverse<O, 1>(m, a, st)))
}
