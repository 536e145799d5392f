// sf_nsfar.h -- the autoregressive NSF of the `backend="lampe"` route (sf_nsfar.hip): state and entry points used by sf_api.hip /
// sf_train.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/synference_hip.h"

struct SfNsfAr {
  int D = 0, C = 0, H = 0, Hp = 0, T = 0, K = 0, NP = 0;
  float bound = 5.f, cw = 0.f, cd = 0.f, logdet0 = 0.f;
  float th_scale[16], th_shift[16];
  std::vector<float> h_xmean, h_xstd;
  // hidden rows sorted by type; perm[p] = logical unit of physical row p (-1: padding), ptype[p] its type, tend[r] = rows of type <= r
  std::vector<int32_t> perm, ptype, tend, ord, dimof, dwave, src;
  int64_t n_params = 0;
  long P_t = 0, t_stride = 0;
  int l_W0 = 0, l_b0 = 0, l_W1 = 0, l_b1 = 0, l_W2 = 0, l_b2 = 0;                             // logical offsets in a transform
  int o_L0t = 0, o_b0 = 0, o_L1t = 0, o_L1m = 0, o_b1 = 0, o_L2t = 0, o_b2 = 0, o_L0m = 0, o_L2m = 0;   // image offsets in a transform
  // 16-sample register-tile sampler (sf_nsfar16.hip): s16_nt = hidden tiles (= D x s16_tpt: type r is tile r, or tiles 2 r and 2 r + 1
  // when it has 17..32 units) or 0 when the shape does not take it;
  // fragment-ordered blocks of a transform (lane l of block (ot, kt): W[16 ot + (l & 15)][16 kt + 4 (l >> 4) + 0..3]) in the sampler's
  // own hidden order (s16_ks = ceil(units per type / 4), at least 2: the units of a type sit on the rows with (row & 3) < s16_ks of
  // its tile), biases in that order
  int s16_nt = 0, s16_ni = 0, s16_ks = 4, s16_tpt = 1, o_F0 = 0, o_fb0 = 0, o_F1 = 0, o_fb1 = 0, o_F2 = 0;
  bool dev_ready = false;
  float* d_img = nullptr;
  int32_t *d_src = nullptr, *d_none = nullptr, *d_perm = nullptr, *d_ptype = nullptr, *d_tend = nullptr, *d_ord = nullptr, *d_dimof = nullptr, *d_dwave = nullptr;
  float *d_xmean = nullptr, *d_xstd = nullptr;
  int affine = 0;               // SF_MAF_AR (zuko MAF): MonotonicAffineTransform instead of the spline
  float* d_ustash = nullptr;
  size_t ustash_cap = 0;
  unsigned char* d_live = nullptr;   // [n_params] 1 = the parameter has a gradient (unmasked weight or bias)
  float* d_gpart = nullptr;          // per-workgroup gradient partials of the training kernel (small batches)
  size_t gpart_cap = 0;
  int32_t* d_gal = nullptr;              // [2][M]: attempts / accepted draws per row (progress rule of the uncapped sampler)
  size_t gal_cap = 0;
  unsigned long long* h_ctr = nullptr;   // pinned host copy of d_ctr (read back after the persistent launch and after every round)
  unsigned long long* d_ctr = nullptr;   // [0] work cursor of the sampler, [1] slots written off, [2] evaluations, [3] first attempts rejected
  uint32_t *d_surv[2] = {nullptr, nullptr}, *d_best = nullptr;   // survivor lists of the find / resolve rounds, lowest accepted attempt
  size_t surv_cap = 0;
  double last_evals = 0.0, last_rej0 = 0.0;   // of the last sampling call that read the counters back
};

int sf_nsfar_create(const sf_flow_desc& d, SfNsfAr** out, std::string& err);
void sf_nsfar_destroy(SfNsfAr* n);
size_t sf_nsfar_lds_bytes(const SfNsfAr& n, int hidden_buffers, int waves = 1);
// re-tiles the logical vector into the masked images (every entry point below reads the images of the LAST pack)
int sf_nsfar_pack(SfNsfAr* n, const float* flat, hipStream_t st, std::string& err);
int sf_nsfar_log_prob(SfNsfAr* n, const float* theta, const float* x, long B, float* out, hipStream_t st, std::string& err);
int sf_nsfar_inverse(SfNsfAr* n, const float* z, const float* x, long B, float* theta, float* logdet, hipStream_t st, std::string& err);
// count != null: acceptance counting (one attempt per item, item i belongs to row i / S); else the rejection sampler
int sf_nsfar_sample(SfNsfAr* n, const float* x, long M, long S, const uint32_t* slots, long n_slots, const float* lo, const float* hi,
                    uint32_t k0, uint32_t k1, unsigned long long slot_offset, int max_attempts, float* out, int32_t* n_drawn,
                    int32_t* count, int64_t* n_unfilled, hipStream_t st, std::string& err, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
// sf_nsfar16.hip: the persistent first launch of sf_nsfar_sample on 16-sample register tiles (same arguments and hand-over as k_ar_sample)
struct SfAr16Launch {
  const float* x; long S; const uint32_t* slots; long n_slots; const float* lo; const float* hi; uint32_t k0, k1; unsigned long long slot_offset;
  uint32_t max_attempts; float* out; int32_t* n_drawn; int32_t* count; unsigned long long* cursor; unsigned int* n_unfilled; int32_t* g_try;
  int32_t* g_acc; unsigned long long walk_R, walk_C; uint32_t window_end; uint32_t* surv; unsigned int* n_surv;
};
bool sf_nsfar16_eligible(const SfNsfAr& n);
hipError_t sf_nsfar16_launch(const SfNsfAr& n, const SfAr16Launch& L, int cus, hipStream_t st);
// the chip-wide rounds on the same candidate routine (arguments of k_ar_find / k_ar_resolve; L carries x, S, the box, the keys and the outputs)
hipError_t sf_nsfar16_find(const SfNsfAr& n, const SfAr16Launch& L, const uint32_t* surv, unsigned int n_surv, uint32_t base, uint32_t chunks,
                           uint32_t att_end, uint32_t* best, unsigned long long* ctr, hipStream_t st);
hipError_t sf_nsfar16_resolve(const SfNsfAr& n, const SfAr16Launch& L, const uint32_t* surv, unsigned int n_surv, const uint32_t* best,
                              uint32_t tried_end, uint32_t tried_now, uint32_t* next, unsigned int* n_next, hipStream_t st);
int sf_nsfar_loss_grad(SfNsfAr* n, const float* flat, const float* theta, const float* x, const long long* idx, long B, float grad_scale,
                       const float* weights, float* loss, double* loss_sum, float* grad, hipStream_t st, std::string& err, hipEvent_t ev0 = nullptr,
                       hipEvent_t ev1 = nullptr);
