// sf_hostio.hip -- device fp32 -> host float64 hand-over of posterior draws (sf_copy_to_host_f64).
//
// The reference returns `posterior.sample(...)` as a host float64 array filled galaxy by galaxy from `.cpu().numpy()` copies
// (ref: src/synference/sbi_runner.py:6436-6457).  Here the draws of a whole catalogue sit in HBM as fp32; moving them costs more
// than drawing them (cfg2: 1.9 ms of kernel, 40 MB over PCIe, 80 MB of float64 written on the host), so the hand-over is a
// native pipeline and not a Python loop:
//   * the source is cut into pieces of SF_HOSTIO_PIECE bytes; piece k goes D2H on a private copy stream into slot k mod NBUF of
//     a ring of PINNED staging buffers (allocated once per process);
//   * when its copy event has completed the piece is widened fp32 -> float64 by a persistent pool of host threads, each taking
//     a contiguous part: AVX2 convert + NON-TEMPORAL stores (the destination is written once and read by somebody else later:
//     streaming stores skip the read-for-ownership, 80 MB of traffic instead of 160), while the next pieces are on the bus.
// PCIe carries fp32; the values are not touched (float -> double is exact).  One call at a time per process (a mutex says so).
#include <hip/hip_runtime.h>
#include <immintrin.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/synference_hip.h"
#include "sf_internal.h"

namespace {

constexpr int NBUF = 4;
size_t piece_bytes() {
  static size_t v = 0;
  if (!v) {
    const char* e = std::getenv("SF_HOSTIO_PIECE_MB");
    const long mb = e ? std::atol(e) : 4;
    v = (size_t)(mb < 1 ? 1 : (mb > 64 ? 64 : mb)) << 20;
  }
  return v;
}

int usable_cores() {
  int n = (int)std::thread::hardware_concurrency();
  if (n < 1) n = 1;
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {   // a GPU box shows 256 cores and grants 16
    char q[64] = {0};
    long period = 0;
    if (std::fscanf(f, "%63s %ld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) {
      const long c = std::atol(q) / period;
      if (c >= 1 && c < n) n = (int)c;
    }
    std::fclose(f);
  }
  return n;
}

__attribute__((target("avx2"))) void widen_avx2(const float* __restrict__ src, double* __restrict__ dst, size_t n) {
  size_t i = 0;
  while (i < n && ((uintptr_t)(dst + i) & 31u)) { dst[i] = (double)src[i]; ++i; }
  for (; i + 8 <= n; i += 8) {
    const __m256 v = _mm256_loadu_ps(src + i);
    _mm256_stream_pd(dst + i, _mm256_cvtps_pd(_mm256_castps256_ps128(v)));
    _mm256_stream_pd(dst + i + 4, _mm256_cvtps_pd(_mm256_extractf128_ps(v, 1)));
  }
  for (; i < n; ++i) dst[i] = (double)src[i];
  _mm_sfence();
}
void widen(const float* src, double* dst, size_t n) {
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2) { widen_avx2(src, dst, n); return; }
  for (size_t i = 0; i < n; ++i) dst[i] = (double)src[i];
}

// a fixed pool of workers; a "job" is P parts of one piece, counted down on the piece's counter
struct Pool {
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv;
  std::vector<std::function<void()>> q;
  bool stop = false;
  explicit Pool(int n) {
    for (int i = 0; i < n; ++i)
      th.emplace_back([this] {
        for (;;) {
          std::function<void()> job;
          {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [this] { return stop || !q.empty(); });
            if (stop && q.empty()) return;
            job = std::move(q.back());
            q.pop_back();
          }
          job();
        }
      });
  }
  void submit(std::function<void()> f) {
    { std::lock_guard<std::mutex> lk(m); q.insert(q.begin(), std::move(f)); }
    cv.notify_one();
  }
  ~Pool() {
    { std::lock_guard<std::mutex> lk(m); stop = true; }
    cv.notify_all();
    for (auto& t : th) t.join();
  }
};

struct State {
  std::mutex call;
  int device = -1;
  float* stage[NBUF] = {nullptr, nullptr, nullptr, nullptr};
  size_t stage_bytes = 0;
  hipStream_t cs = nullptr;
  hipEvent_t ev[NBUF] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_src = nullptr;
  Pool* pool = nullptr;
  int workers = 0;
  std::atomic<int> left[NBUF];
  std::mutex dm;
  std::condition_variable dcv;
};
State g;
#define HIO(call)                                                            \
  do {                                                                       \
    hipError_t e_ = (call);                                                  \
    if (e_ != hipSuccess) {                                                  \
      err = std::string(#call) + ": " + hipGetErrorString(e_);               \
      return SF_ERR_HIP;                                                     \
    }                                                                        \
  } while (0)

void wait_slot(int b) {
  std::unique_lock<std::mutex> lk(g.dm);
  g.dcv.wait(lk, [b] { return g.left[b].load() == 0; });
}

}  // namespace

int sf_hostio_copy_f64(const float* dev_src, double* host_dst, int64_t n, hipStream_t stream, std::string& err) {
  std::lock_guard<std::mutex> call(g.call);
  int dev = 0;
  HIO(hipGetDevice(&dev));
  const size_t pb = piece_bytes();
  if (g.device != dev || g.stage_bytes != pb) {   // first call (or another device / piece size): build the pipeline
    for (int b = 0; b < NBUF; ++b) {
      if (g.stage[b]) (void)hipHostFree(g.stage[b]);
      g.stage[b] = nullptr;
      if (g.ev[b]) (void)hipEventDestroy(g.ev[b]);
      g.ev[b] = nullptr;
      g.left[b].store(0);
    }
    if (g.cs) (void)hipStreamDestroy(g.cs);
    g.cs = nullptr;
    if (g.ev_src) (void)hipEventDestroy(g.ev_src);
    g.ev_src = nullptr;
    g.stage_bytes = 0;
    for (int b = 0; b < NBUF; ++b) {
      HIO(hipHostMalloc((void**)&g.stage[b], pb, hipHostMallocDefault));
      HIO(hipEventCreateWithFlags(&g.ev[b], hipEventDisableTiming));
    }
    HIO(hipStreamCreateWithFlags(&g.cs, hipStreamNonBlocking));
    HIO(hipEventCreateWithFlags(&g.ev_src, hipEventDisableTiming));
    g.device = dev;
    g.stage_bytes = pb;
  }
  if (!g.pool) {
    const char* e = std::getenv("SF_HOSTIO_THREADS");
    int w = e ? std::atoi(e) : usable_cores();
    w = w < 1 ? 1 : (w > 8 ? 8 : w);
    g.pool = new Pool(w);
    g.workers = w;
  }
  // the draws were written on the caller's stream
  HIO(hipEventRecord(g.ev_src, stream));
  HIO(hipStreamWaitEvent(g.cs, g.ev_src, 0));
  const size_t per = pb / sizeof(float);
  const int64_t n_pieces = (int64_t)(((size_t)n + per - 1) / per);
  auto issue = [&](int64_t k) -> hipError_t {
    const int b = (int)(k % NBUF);
    const size_t off = (size_t)k * per, cnt = (size_t)n - off < per ? (size_t)n - off : per;
    hipError_t e = hipMemcpyAsync(g.stage[b], dev_src + off, cnt * sizeof(float), hipMemcpyDeviceToHost, g.cs);
    if (e != hipSuccess) return e;
    return hipEventRecord(g.ev[b], g.cs);
  };
  int rc = SF_OK;
  int64_t issued = 0;
  for (; issued < n_pieces && issued < NBUF; ++issued) {
    hipError_t e = issue(issued);
    if (e != hipSuccess) { err = std::string("hipMemcpyAsync(D2H piece): ") + hipGetErrorString(e); rc = SF_ERR_HIP; break; }
  }
  const int parts = g.workers;
  for (int64_t k = 0; rc == SF_OK && k < n_pieces; ++k) {
    const int b = (int)(k % NBUF);
    hipError_t e = hipEventSynchronize(g.ev[b]);
    if (e != hipSuccess) { err = std::string("hipEventSynchronize(piece): ") + hipGetErrorString(e); rc = SF_ERR_HIP; break; }
    const size_t off = (size_t)k * per, cnt = (size_t)n - off < per ? (size_t)n - off : per;
    g.left[b].store(parts);
    for (int p = 0; p < parts; ++p) {
      // parts begin on multiples of 8 elements: every part but the first starts 32-byte aligned relative to the piece
      const size_t a0 = (cnt * (size_t)p / (size_t)parts) & ~(size_t)7, a1 = p + 1 == parts ? cnt : ((cnt * (size_t)(p + 1) / (size_t)parts) & ~(size_t)7);
      const float* s = g.stage[b] + a0;
      double* d = host_dst + off + a0;
      g.pool->submit([s, d, a0, a1, b] {
        if (a1 > a0) widen(s, d, a1 - a0);
        if (g.left[b].fetch_sub(1) == 1) {
          std::lock_guard<std::mutex> lk(g.dm);
          g.dcv.notify_all();
        }
      });
    }
    // the slot of piece k - 1 is needed by piece k - 1 + NBUF: wait for its widening, then put the next copy on the bus
    if (k >= 1 && issued < n_pieces) {
      wait_slot((int)((k - 1) % NBUF));
      hipError_t e2 = issue(issued);
      if (e2 != hipSuccess) { err = std::string("hipMemcpyAsync(D2H piece): ") + hipGetErrorString(e2); rc = SF_ERR_HIP; break; }
      ++issued;
    }
  }
  for (int b = 0; b < NBUF; ++b) wait_slot(b);
  if (rc != SF_OK) (void)hipStreamSynchronize(g.cs);
  return rc;
}
