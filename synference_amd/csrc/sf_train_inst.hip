// sf_train_inst.hip -- one translation unit per SF_HT: instantiates the training kernels.
#include "sf_train_kernels.h"
#include <cstdio>
#include <cstdlib>

#ifndef SF_HT
#error "compile with -DSF_HT=1..4"
#endif
#define SF_CAT_(a, b) a##b
#define SF_CAT(a, b) SF_CAT_(a, b)

template <class K>
static hipError_t set_shmem(K kernel, size_t bytes) {
  return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <bool LDSW>
static hipError_t launch_maf(const SfDev& m, const SfTrainArgs& a, size_t shmem, hipStream_t st) {
  const long grid = (a.B + 31) / 32;  // one workgroup (producer + consumer wave) per 32-sample tile
  static SfAttrCache attr;
  static size_t attr_bytes = 0;
  int attr_dev;
  if (attr.need(attr_dev) || shmem > attr_bytes) {
    hipError_t e = set_shmem(k_maf_train<SF_HT, LDSW>, shmem);
    if (e != hipSuccess) return e;
    attr.set(attr_dev);
    attr_bytes = shmem;
  }
#ifdef SF_TRAIN_TRACE
  {  // developer build: time stamps of the middle workgroup, printed in units of 100 shader cycles
    static unsigned long long* d_tr = nullptr;
    if (!d_tr && hipMalloc(&d_tr, 256 * 8) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipMemsetAsync(d_tr, 0, 256 * 8, st);
    SfTrainArgs b = a;
    b.trace = d_tr;
    hipLaunchKernelGGL((k_maf_train<SF_HT, LDSW>), dim3((unsigned)grid), dim3(128), shmem, st, m, b);
    (void)hipStreamSynchronize(st);
    unsigned long long h[256];
    (void)hipMemcpy(h, d_tr, sizeof(h), hipMemcpyDeviceToHost);
    static int calls = 0;
    if (++calls % 8 == 0) {
      fprintf(stderr, "[train trace] B=%ld grid=%ld ldsw=%d shmem=%zu\n", a.B, grid, (int)LDSW, shmem);
      for (int w = 0; w < 2; ++w) {
        fprintf(stderr, "  wave %d:", w);
        for (int i = 0; i < 128; ++i)
          if (h[w * 128 + i]) fprintf(stderr, " %d:%.2f", i, (double)(long long)(h[w * 128 + i] - h[0]) * 0.01);
        fprintf(stderr, "\n");
      }
    }
    return hipGetLastError();
  }
#endif
  hipLaunchKernelGGL((k_maf_train<SF_HT, LDSW>), dim3((unsigned)grid), dim3(128), shmem, st, m, a);
  return hipGetLastError();
}

hipError_t SF_CAT(sf_launch_maf_train_h, SF_HT)(const SfDev& m, const SfTrainArgs& a, hipStream_t st) {
  const size_t pipe = (size_t)2 * (SF_JOB_HDR + (2 * SF_HT) * SF_TL) * sizeof(float);
  // Operands of one transform in LDS (forward image without the sampler-only head rows, transposed image without the
  // context block) when two workgroups still fit a CU (80 KiB each) AND the whole batch is resident at that
  // occupancy (512 tiles = batch 16 384): measured 3 % at batch 64 and 2 % at 16 384 (the forward sweep of a
  // transform 24.0 -> 21.6 k cycles); at 65 536 rows the halved occupancy costs 8 %, so larger batches keep streaming
  // operands from L2 with four workgroups per CU.
  const size_t wimg = (size_t)(m.o_hv > m.oT_wc ? m.o_hv : m.oT_wc) * sizeof(float);
  static int use_lds = -1;
  if (use_lds < 0) { const char* e = std::getenv("SF_TRAIN_LDSW"); use_lds = e ? std::atoi(e) : 1; }
  if (use_lds && (a.B + 31) / 32 <= 512 && (m.o_hv & 3) == 0 && (m.oT_wc & 3) == 0 && pipe + wimg <= (size_t)80 * 1024)
    return launch_maf<true>(m, a, pipe + wimg, st);
  return launch_maf<false>(m, a, pipe, st);
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
