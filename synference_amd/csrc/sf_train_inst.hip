// sf_train_inst.hip -- one translation unit per SF_HT: instantiates the training kernels.
#include "sf_train_kernels.h"
#include <cstdio>

#ifndef SF_HT
#error "compile with -DSF_HT=1..4"
#endif
#define SF_CAT_(a, b) a##b
#define SF_CAT(a, b) SF_CAT_(a, b)

template <class K>
static hipError_t set_shmem(K kernel, size_t bytes) {
  return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

hipError_t SF_CAT(sf_launch_maf_train_h, SF_HT)(const SfDev& m, const SfTrainArgs& a, hipStream_t st) {
  const long grid = (a.B + 31) / 32;  // one workgroup (producer + consumer wave) per 32-sample tile
  const size_t shmem = (size_t)2 * (SF_JOB_HDR + (2 * SF_HT) * SF_TL) * sizeof(float);
  static SfAttrCache attr;
  int attr_dev;
  if (attr.need(attr_dev)) {
    hipError_t e = set_shmem(k_maf_train<SF_HT>, shmem);
    if (e != hipSuccess) return e;
    attr.set(attr_dev);
  }
#ifdef SF_TRAIN_TRACE
  {  // developer build: time stamps of the middle workgroup, printed as microsecond deltas (100 MHz counter)
    static unsigned long long* d_tr = nullptr;
    if (!d_tr && hipMalloc(&d_tr, 256 * 8) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipMemsetAsync(d_tr, 0, 256 * 8, st);
    SfTrainArgs b = a;
    b.trace = d_tr;
    hipLaunchKernelGGL((k_maf_train<SF_HT>), dim3((unsigned)grid), dim3(128), shmem, st, m, b);
    (void)hipStreamSynchronize(st);
    unsigned long long h[256];
    (void)hipMemcpy(h, d_tr, sizeof(h), hipMemcpyDeviceToHost);
    static int calls = 0;
    if (++calls % 8 == 0) {
      fprintf(stderr, "[train trace] B=%ld grid=%ld\n", a.B, grid);
      for (int w = 0; w < 2; ++w) {
        fprintf(stderr, "  wave %d:", w);
        for (int i = 0; i < 128; ++i)
          if (h[w * 128 + i]) fprintf(stderr, " %d:%.2f", i, (double)(h[w * 128 + i] - h[0]) * 0.01);
        fprintf(stderr, "\n");
      }
    }
    return hipGetLastError();
  }
#endif
  hipLaunchKernelGGL((k_maf_train<SF_HT>), dim3((unsigned)grid), dim3(128), shmem, st, m, a);
  return hipGetLastError();
}

template <int PT>
static hipError_t launch_nsf(const SfDev& m, const SfTrainArgs& a, hipStream_t st) {
  const long grid = (a.B + 31) / 32;
  const size_t shmem = (size_t)2 * (SF_JOB_HDR + SfNsfLds<SF_HT, PT>::tiles * SF_TL) * sizeof(float);
  static SfAttrCache attr;
  int attr_dev;
  if (attr.need(attr_dev)) {
    hipError_t e = set_shmem(k_nsf_train<SF_HT, PT>, shmem);
    if (e != hipSuccess) return e;
    attr.set(attr_dev);
  }
  hipLaunchKernelGGL((k_nsf_train<SF_HT, PT>), dim3((unsigned)grid), dim3(128), shmem, st, m, a);
  return hipGetLastError();
}

hipError_t SF_CAT(sf_launch_nsf_train_h, SF_HT)(const SfDev& m, const SfTrainArgs& a, hipStream_t st) {
  switch (m.PT) {
    case 2: return launch_nsf<2>(m, a, st);
    case 3: return launch_nsf<3>(m, a, st);
  }
  return hipErrorInvalidValue;
}
