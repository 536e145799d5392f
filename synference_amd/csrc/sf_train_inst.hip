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

template <int NBUF, int NC, int DM>
static hipError_t launch_maf(const SfDev& m, const SfTrainArgs& a, size_t shmem, hipStream_t st) {
  const long grid = (a.B + 31) / 32;  // one workgroup (producer + consumer waves) per 32-sample tile
  static SfAttrCache attr;
  static size_t attr_bytes = 0;
  int attr_dev;
  if (attr.need(attr_dev) || shmem > attr_bytes) {
    hipError_t e = set_shmem(k_maf_train<SF_HT, NBUF, NC, DM>, shmem);
    if (e != hipSuccess) return e;
    attr.set(attr_dev);
    attr_bytes = shmem;
  }
  const dim3 block(64 * (1 + NC));
#ifdef SF_TRAIN_TRACE
  {  // developer build: time stamps of the middle workgroup, printed in units of 100 shader cycles
    static unsigned long long* d_tr = nullptr;
    if (!d_tr && hipMalloc(&d_tr, 384 * 8) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipMemsetAsync(d_tr, 0, 384 * 8, st);
    SfTrainArgs b = a;
    b.trace = d_tr;
    hipLaunchKernelGGL((k_maf_train<SF_HT, NBUF, NC, DM>), dim3((unsigned)grid), block, shmem, st, m, b);
    (void)hipStreamSynchronize(st);
    unsigned long long h[384];
    (void)hipMemcpy(h, d_tr, sizeof(h), hipMemcpyDeviceToHost);
    static int calls = 0;
    if (++calls % 8 == 0) {
      fprintf(stderr, "[train trace] B=%ld grid=%ld nbuf=%d nc=%d shmem=%zu\n", a.B, grid, NBUF, NC, shmem);
      for (int w = 0; w < 1 + NC; ++w) {
        fprintf(stderr, "  wave %d:", w);
        for (int i = 0; i < 128; ++i)
          if (h[w * 128 + i]) fprintf(stderr, " %d:%.2f", i, (double)(long long)(h[w * 128 + i] - h[0]) * 0.01);
        fprintf(stderr, "\n");
      }
    }
    return hipGetLastError();
  }
#endif
  hipLaunchKernelGGL((k_maf_train<SF_HT, NBUF, NC, DM>), dim3((unsigned)grid), block, shmem, st, m, a);
  return hipGetLastError();
}

hipError_t SF_CAT(sf_launch_maf_train_h, SF_HT)(const SfDev& m, const SfTrainArgs& a, hipStream_t st) {
  const size_t buf = (size_t)(SF_JOB_HDR + (2 * SF_HT) * SF_TL) * sizeof(float), ctl = SF_PIPE_CTL * sizeof(int);
  const long tiles = (a.B + 31) / 32;
  static int two = -1;  // SF_TRAIN_CONSUMERS=1 keeps one consumer / two buffers at every batch size
  if (two < 0) { const char* e = std::getenv("SF_TRAIN_CONSUMERS"); two = e ? (std::atoi(e) >= 2) : 1; }
  // While the whole batch is resident with two workgroups per CU (512 tiles = batch 16 384) LDS is plentiful: four job
  // buffers and TWO consumer waves, so that the producer never waits at a hand-over (with one consumer and two buffers
  // it stalled at every job: the consumer is busy 75 % of the backward sweep).  Measured: 182 -> 172 us at batch
  // 16 384; the sweep is then bound by the producer's own chain.  Larger batches keep two buffers and one consumer:
  // four workgroups per CU matter more there (65 536 rows: 562 vs 577 us).
  const bool d8 = m.D <= 8;  // (every BASELINE shape; D up to 16 runs the same code with longer theta loops)
  if (two && tiles <= 512 && ctl + 4 * buf <= (size_t)80 * 1024)
    return d8 ? launch_maf<4, 2, 8>(m, a, ctl + 4 * buf, st) : launch_maf<4, 2, SF_DMAX>(m, a, ctl + 4 * buf, st);
  return d8 ? launch_maf<2, 1, 8>(m, a, ctl + 2 * buf, st) : launch_maf<2, 1, SF_DMAX>(m, a, ctl + 2 * buf, st);
}

template <int PT>
static hipError_t launch_nsf(const SfDev& m, const SfTrainArgs& a, hipStream_t st) {
  const long grid = (a.B + 31) / 32;
  const size_t shmem = SF_PIPE_CTL * sizeof(int) + (size_t)2 * (SF_JOB_HDR + SfNsfLds<SF_HT, PT>::tiles * SF_TL) * sizeof(float);
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
