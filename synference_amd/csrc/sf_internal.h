// sf_internal.h -- launcher prototypes shared by the kernel translation units and sf_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sf_layout.h"
#include "sf_queue.h"

struct SfSampleArgsHost {
  const float* x = nullptr;
  const float* z_in = nullptr;
  long S = 1;
  const uint32_t* slots = nullptr;
  long slot_base = 0;
  long n_items = 0;
  uint32_t attempt = 0, k0 = 0, k1 = 0;
  // added to the slot id in the Philox counter only: row_offset * S, so that a block of rows of a larger catalogue
  // (a chunk, a rank's shard) draws exactly what those rows draw in a single call over the whole catalogue
  unsigned long long rng_slot_offset = 0;
  int attempts_per_slot = 1;  // A: consecutive attempts evaluated per listed slot (power of two <= 32)
  int log2_attempts = 0;      // log2(A) (set by sf_launch_inverse)
  const float* lo = nullptr;
  const float* hi = nullptr;
  float* out = nullptr;
  int out_f64 = 0;            // != 0: `out` is a double array (sf_flow_set_sample_output_f64): accepted draws are widened in the store
  float* logdet_out = nullptr;
  uint32_t* rejected = nullptr;
  uint32_t* n_rejected = nullptr;
  int32_t* n_drawn = nullptr;
  int32_t* count = nullptr;
  // find mode (plain kernels): listed slot i = item >> log2_attempts (any power of two of attempts per slot); an accepted
  // attempt only lowers best[i] (atomic min), nothing else is written.  att_list (sampling mode, attempts_per_slot 1):
  // listed slot i tries attempt att_list[i] instead of `attempt` (0xffffffff = none: goes straight to the rejected list)
  uint32_t* best = nullptr;
  const uint32_t* att_list = nullptr;
  // persistent mode (q != nullptr): one launch works the dense list AND its retries to the end (sf_queue.h)
  struct SfQueue* q = nullptr;
  unsigned long long* ring = nullptr;  // retry ring, ring_mask + 1 entries
  uint32_t ring_mask = 0;
  uint32_t attempt_limit = 0xffffffffu;  // attempts [attempt, attempt_limit) are tried by this launch; then -> rejected[]
  uint32_t n_total = 0;                // slots of the dense list (== n_items)
  uint32_t out_slots = 0;              // M * S: every slot id must be below this
  // != 0 (whole-catalogue launches: slots == nullptr, slot_base == 0, n_total == M * S): the dense list walks ACROSS the
  // galaxies inside blocks of dense_G consecutive galaxies -- item i lies in block b = i / (dense_G * S); with Gb =
  // min(dense_G, M - b * dense_G) galaxies in that block and j = i - b * dense_G * S it is slot
  // (b * dense_G + j % Gb) * S + j / Gb.  In plain slot order the ~8 workgroups that draw the ranges of a low-acceptance
  // galaxy grind through its retries alone while the rest of the chip runs dry (their own retries come first and the dense
  // phase does not share work); interleaved, every workgroup carries the same mix, and a block keeps the context rows
  // in flight few enough to stay cached.  The draws do not depend on the order (streams are keyed by slot and attempt).
  uint32_t dense_G = 0;
  // ... in RUNS of dense_run consecutive draws of one galaxy (dense_run divides S; 1 = draw by draw): with a run = one tile
  // of draws (16 / 32) the lanes of a tile share their galaxy's context-table rows and write one contiguous piece of the
  // output, and a galaxy's S slots still spread over S / dense_run ranges.  Within a block of Gb galaxies item j is draw
  // (j / run / Gb) * run + j % run of galaxy (j / run) % Gb.
  uint32_t dense_run = 1;
  // != 0 (an explicit slot list, e.g. an ensemble member's share, sorted by slot): item i of the dense list is list entry
  // walk(i), walk = i -> i * list_mul mod 2^list_log2, repeated until the result is < n_total (cycle walking: a
  // permutation of [0, n_total)) -- consecutive items lie ~n_total / 128 entries apart, for the same reason as dense_G
  uint32_t list_mul = 0, list_log2 = 0;
  // tail mode: entries that have failed at least this many attempts are tried at the full speculation width at once
  // (0 = the width only grows with the attempt number)
  uint32_t spec_full_after = 0;
  // tail mode: speculation fills at most this many items of a workgroup iteration (0 = all of them).  A wave walks its
  // tiles one after the other, so items beyond one tile per wave double the latency of the iteration.
  uint32_t tail_cap = 0;
  int32_t* gal_acc = nullptr;          // optional [M]: += 1 per accepted slot of the galaxy (progress test between stages)
#ifdef SF_Q_STATS
  // developer build: [workgroup][256][4] = {start of the iteration (10 ns since the first workgroup's start is taken on
  // the host), entries, log2 attempts per entry, end of its epilogue} of every iteration of every workgroup
  uint32_t* qtrace = nullptr;
#endif
};
#ifdef __HIPCC__
// one value of an accepted draw: fp32 into a float array or -- out_f64 -- widened into a double array (wave-uniform choice)
__device__ __forceinline__ void sf_out_store(const SfSampleArgsHost& a, size_t idx, float v) {
  if (a.out_f64) reinterpret_cast<double*>(a.out)[idx] = (double)v;
  else a.out[idx] = v;
}
#endif


hipError_t sf_launch_logprob(const SfDev& m, const float* theta, const float* x, long B, float* out,
                             hipStream_t st);
hipError_t sf_launch_inverse(const SfDev& m, const SfSampleArgsHost& a, hipStream_t st);
bool sf_maf16_enabled(const SfDev& m, const SfSampleArgsHost& a);
hipError_t sf_launch_maf_inv16(const SfDev& m, const SfSampleArgsHost& a, hipStream_t st);
bool sf_maf16b_available(const SfDev& m);
hipError_t sf_launch_maf_inv16b_hook(const SfDev& m, const float* z, const float* x, long n, float* out, hipStream_t st);
void sf_sampler_fp32_set(int on);
int sf_sampler_fp32_get();        // 1 fp32, 0 split bf16 x3, -1 unset: per flow kind
int sf_sampler_fp32_for(int kind);  // the mode a flow of this kind samples in (unset: MAF fp32, NSF split)
// per-galaxy context table: rows/variants for this flow (0 = the flow has no table path), builder
void sf_ctab_shape(const SfDev& m, int& R, int& NV);
hipError_t sf_launch_ctab(const SfDev& m, const float* x, long M, float* tab, hipStream_t st);
hipError_t sf_launch_maf_ctab16(const SfDev& m, const float* x, long M, float* tab, hipStream_t st);
hipError_t sf_launch_maf_fuse16(const SfDev& m, hipStream_t st);
int sf_maf16_fused_d(const SfDev& m);   // D (3..5) when the fused-first-layer fp32 kernels apply to this view (table with c0' rows), else 0
hipError_t sf_launch_maf_find16_zin(const SfDev& m, const SfSampleArgsHost& a, hipStream_t st);   // W' = (W1 o M)(W0 o M0) into o16_wp of every transform
hipError_t sf_launch_pack(const float* flat, const int32_t* s1, const int32_t* s2, float* packed, long n,
                          hipStream_t st);
hipError_t sf_launch_pack_bf16(const float* flat, const int32_t* src, unsigned short* out, long n, hipStream_t st);
hipError_t sf_launch_pack_bf16_split(const float* flat, const int32_t* src, unsigned short* out, long n, hipStream_t st);
hipError_t sf_launch_fill_nan_rows(float* out, const uint32_t* slots, long n, int D, hipStream_t st, int out_f64 = 0);
hipError_t sf_launch_fill_i32(int32_t* p, long n, int32_t v, hipStream_t st);
hipError_t sf_launch_account_window(const uint32_t* list, const uint32_t* best, long n, long S, uint32_t a_lo, uint32_t A,
                                    int32_t* n_drawn, int32_t* gal_acc, hipStream_t st);
// survivors of galaxies with gal_acc == 0 become NaN rows; the others are compacted in place; *n_surv updated
hipError_t sf_launch_filter_survivors(uint32_t* list, unsigned int* n_surv, long S, const int32_t* gal_acc, float* out,
                                      int D, hipStream_t st, int out_f64 = 0);

#include <string>
// per-device "attribute already set" cache for hipFuncSetAttribute(MaxDynamicSharedMemorySize): function attributes are
// per device, so a process that launches on a second device must set them there too
struct SfAttrCache {
  bool done[16] = {false};
  bool need(int& dev) {
    dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return true;
    return !done[dev];
  }
  void set(int dev) { if (dev >= 0 && dev < 16) done[dev] = true; }
};
// resident workgroups of a persistent kernel, cached per (device, dynamic LDS bytes): the occupancy of one template instance
// depends on the flow's LDS footprint and the grid on the device's CU count -- a process that samples flows of different
// sizes, or on a second GPU, must not reuse the first answer
struct SfResidentCache {
  int dev[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
  size_t sh[8] = {0};
  int val[8] = {0};
  int n = 0;
  bool get(int d, size_t s, int& v) const {
    for (int i = 0; i < 8; ++i)
      if (dev[i] == d && sh[i] == s) { v = val[i]; return true; }
    return false;
  }
  void put(int d, size_t s, int v) { dev[n & 7] = d; sh[n & 7] = s; val[n & 7] = v; ++n; }
};
void sf_set_error(const std::string& msg);
struct sf_comm;
int sf_comm_all_reduce_impl(sf_comm* c, float* buf, long n, hipStream_t st);  // sf_comm.hip: ncclAllReduce(SUM, fp32) in place  // thread-local message behind sf_last_error()

// ---- the handle (shared by sf_api.hip and sf_train.hip) -------------------------------------
#define SF_LOSS_PARTS 64
#define SF_MAX_ROUNDS 80
struct SfNsf1;
struct sf_flow {
  SfLayout L;
  struct SfNsfAr* nsfar = nullptr;   // != null: the autoregressive NSF (sf_nsfar.hip); L carries the shape and n_params only
  SfNsf1* nsf1 = nullptr;       // != null: the one-parameter NSF (sf_nsf1.hip); L then only carries the shape and n_params
  bool dev_ready = false;
  bool params_set = false;
  bool train_ready = false;     // every lazily built training buffer exists (sf_train.hip)
  bool flat_valid = false;      // d_flat holds the vector last given to sf_flow_set_params
  float* d_packed = nullptr;    // forward operand image
  float* d_packedT = nullptr;   // transposed operand image (training, lazily built)
  float* d_cst = nullptr;
  unsigned short* d_packedB = nullptr;  // bf16 hidden operand image (hidden_bf16)
  int32_t* d_bsrc = nullptr;
  float* d_packed16 = nullptr;          // 16-row image of the incremental MAF inverse (sf_maf16.hip)
  bool packed16_stale = false;          // loss_grad refreshed only the forward image
  unsigned short* d_packed16B = nullptr;  // split-bf16 hidden blocks of the 16-row sampler (sf_layout.h)
  int32_t* d_s16B = nullptr;
  int32_t *d_s16a = nullptr, *d_s16b = nullptr;
  int32_t *d_s1 = nullptr, *d_s2 = nullptr, *d_t1 = nullptr, *d_t2 = nullptr;
  float* d_flat = nullptr;      // staging for host-sourced parameters
  float* d_gpacked = nullptr;   // gradient image replicas (accumulation target)
  size_t gpacked_cap = 0;       // floats
  int32_t* d_gdst = nullptr;    // logical parameter -> gradient image index
  // cooperative 16-row training path (sf_trainc.hip): operand image, its gather table, gradient partials
  float* d_imgC = nullptr;
  int32_t *d_sC1 = nullptr, *d_sC2 = nullptr, *d_gdstC = nullptr;
  int32_t* d_gsrcC = nullptr;  // inverse of gdstC: [n_gradC][2] parameters fed by a position of the gradient partial (-1: none); nullptr: not invertible
  int32_t* d_gzeroC = nullptr; // parameters without a position (their gradient is 0)
  long n_gzeroC = 0;
  float* d_gpartC = nullptr;
  long long* d_gfixC = nullptr; // SF_FIX_REPLICAS int64 gradient images (fixed-point accumulation, sf_fixacc.h)
  size_t gpartC_cap = 0;        // floats
  bool trainc_ready = false;
  // captured training step of sf_flow_train_epoch (HIP graph: step_begin -> prep -> flow -> gather -> clip + Adam -> step_end)
  hipGraph_t step_graph = nullptr;
  hipGraphExec_t step_exec = nullptr;
  hipStream_t step_stream = nullptr;     // the capture / replay stream (the caller's may be the legacy default stream)
  hipEvent_t step_ev[2] = {nullptr, nullptr};
  long long* d_step_ctr = nullptr;       // [2]: batch number within the epoch, Adam steps taken
  float* d_step_bc = nullptr;            // [2]: Adam's bias corrections of the running step
  long long* d_step_rows = nullptr;      // the running batch's rows
  size_t step_rows_cap = 0;
  unsigned long long step_key[20] = {0}; // what the graph was captured for
  float* d_ustash = nullptr;    // cooperative NSF training (sf_nsfc.hip): u / u' of every transform, [rows][T][16]
  size_t ustash_cap = 0;        // floats
  float* d_act = nullptr;       // activation stash (training)
  size_t act_cap = 0;           // floats
  uint32_t* d_rej[2] = {nullptr, nullptr};
  size_t rej_cap = 0;
  SfQueue* d_queue = nullptr;    // work-queue words of the persistent sampler (sf_queue.h)
  SfQueue* h_queue = nullptr;    // pinned host mirror, read once per stage
  unsigned long long* d_ring = nullptr;  // retry ring
  uint64_t ring_cap = 0;         // entries (power of two)
  bool ring_dirty = false;       // a persistent launch did not end cleanly: clear the ring before the next one
  int32_t* d_galacc = nullptr;   // per-galaxy accepted-slot counter of a stage (progress rule)
  uint32_t* d_best = nullptr;    // deep-tail windows: lowest accepted attempt per survivor (find launch -> resolve launch)
  size_t best_cap = 0;
  size_t galacc_cap = 0;
  // per-block shares of |grad|^2 from the gather of the last loss_grad (epoch loop only: want_sq); n_sqpart = 0: none
  float* d_sqpart = nullptr;
  size_t sqpart_cap = 0;
  int n_sqpart = 0;
  bool want_sq = false;
  bool prep_lite = false;     // epoch call, not its last step: only the cooperative training image is re-tiled per step
  bool packed_stale = false;  // the density / sampler images lag behind the training image (cleared by the next full re-tiling)
  // loss sums of an epoch call spread over SF_LOSS_PARTS device scalars (non-null only inside sf_flow_train_epoch), folded into
  // the caller's scalar at its end
  double* d_losspart = nullptr;
  double* d_losspart_mem = nullptr;
  bool losspart_used = false;
  uint32_t* d_cnt = nullptr;     // SF_MAX_ROUNDS rejected-slot counters (one per round of a sf_flow_sample call)
  uint32_t* h_cnt = nullptr;     // pinned host mirror for the per-round read-back
  bool wp_stale = true;              // the fused first-layer blocks (o16_wp) lag behind packed16: recomputed before the next context table
  bool sample_out_f64 = false;       // sf_flow_set_sample_output_f64: `out` of sf_flow_sample / _slots is a double array
  long long sample_row_offset = 0;   // sf_flow_set_sample_row_offset: first row of the next sampling calls in its catalogue
  double sample_time_limit_s = 0.0;  // > 0: sf_flow_sample* stop opening new attempt windows after this much wall time
  bool profiling = false;         // sf_flow_set_profiling: bracket the training flow kernel with HIP events
  hipEvent_t ev_train[2] = {nullptr, nullptr};
  bool ev_train_valid = false;
  hipEvent_t ev_dense[2] = {nullptr, nullptr};  // brackets round 0 of the last sf_flow_sample call
  float last_stats[4] = {0.f, 0.f, 0.f, 0.f};   // dense-round ms, rounds, rejected after round 0, items evaluated
  float* d_ctab = nullptr;       // per-galaxy context table (sf_flow_prepare_context)
  size_t ctab_cap = 0;           // floats
  const float* ctab_x = nullptr; // context rows the table was built from (NULL = no valid table)
  int64_t ctab_M = 0;
  SfDev dev() const {
    SfDev v = L.dev;
    v.packed = d_packed;
    v.packedT = d_packedT;
    v.cst = d_cst;
    v.packedB = d_packedB;
    v.packed16 = packed16_stale ? nullptr : d_packed16;
    v.packed16B = reinterpret_cast<const uint32_t*>(d_packed16B);
    v.ctab = nullptr;  // set per launch by the sampler entry points
    return v;
  }
};

// device fp32 -> host float64 hand-over (sf_hostio.hip)
int sf_hostio_copy_f64(const float* dev_src, double* host_dst, int64_t n, hipStream_t stream, std::string& err);
