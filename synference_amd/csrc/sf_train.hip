// sf_train.hip -- training kernels.
//   (1) fused global-norm clip + Adam/AdamW      ref: custom_runner.py:613-618
//   (2) forward + backward of -log_prob          ref: custom_runner.py:604-610
#include "sf_train.h"

#include <hip/hip_runtime.h>

#include <cstdlib>

#include "sf_internal.h"
#include "sf_train_args.h"
#include "sf_fixacc.h"
#include "sf_nsf1.h"
#include "sf_nsfar.h"
#include "sf_nsfc.h"
#include "sf_trainc.h"

// ---------------------------------------------------------------------------------------------
// clip_grad_norm_ + Adam
// ---------------------------------------------------------------------------------------------
__global__ void k_sqnorm(const float* __restrict__ g, long n, float* __restrict__ acc) {
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    s += g[i] * g[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  __shared__ float part[16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) part[w] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += part[i];
    atomicAdd(acc, t);
  }
}

__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, const float* __restrict__ sq, long n, sf_adam_desc d, float bc1,
                       float bc2, float max_norm, float* __restrict__ norm_out, const float* __restrict__ bc_dev) {
  if (bc_dev) { bc1 = bc_dev[0]; bc2 = bc_dev[1]; }   // (captured training step: the step counter lives on the device)
  const float total = sqrtf(*sq);
  float coef = 1.f;
  if (max_norm > 0.f) coef = fminf(max_norm / (total + 1e-6f), 1.0f);  // torch clip_grad_norm_
  if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = total;
  const float step_size = d.lr / bc1;
  const float inv_sqrt_bc2 = rsqrtf(bc2);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float pi = p[i];
    float gi = g[i] * coef;
    if (d.decoupled) pi *= (1.f - d.lr * d.weight_decay);
    else if (d.weight_decay != 0.f) gi += d.weight_decay * pi;
    const float mi = d.beta1 * m[i] + (1.f - d.beta1) * gi;
    const float vi = d.beta2 * v[i] + (1.f - d.beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + d.eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

// One-launch form for small parameter vectors (n <= 131072): every workgroup first reduces the WHOLE gradient to
// its squared norm (L2-resident re-reads, a fixed summation order -> the clip coefficient is deterministic), then
// updates its own slice.  Replaces memset + k_sqnorm + k_adam.
__global__ __launch_bounds__(1024) void k_adam_fused(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, float* __restrict__ sq_out, long n,
                                                    sf_adam_desc d, float bc1, float bc2, float max_norm,
                                                    float* __restrict__ norm_out, const float* __restrict__ bc_dev,
                                                    const float* __restrict__ sq_part, int n_sq) {
  if (bc_dev) { bc1 = bc_dev[0]; bc2 = bc_dev[1]; }
  // Everything this thread will need is requested before anything is waited for: its own elements of p / m / v / g (one
  // float4 each for n <= 4096 x blocks: the common case) AND its eight float4 of the gradient for the norm -- the kernel
  // is a chain of L2 round trips (32 k parameters: 12 us when the norm took two trips and the update a third).
  const long n4 = n >> 2;
  const float4* __restrict__ g4 = reinterpret_cast<const float4*>(g);
  const long own = (long)blockIdx.x * blockDim.x + threadIdx.x;  // first float4 this thread updates (n % 4 == 0 path)
  const bool vec = (n & 3) == 0 && (((uintptr_t)p | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
  const bool has_own = vec && own < n4;
  float4 po, mo, vo, go;
  if (has_own) {
    po = reinterpret_cast<const float4*>(p)[own]; mo = reinterpret_cast<const float4*>(m)[own];
    vo = reinterpret_cast<const float4*>(v)[own]; go = g4[own];
  }
  float s = 0.f;
  if (sq_part) {   // the gather left the norm in n_sq shares: one short strided read instead of the whole gradient
    for (int i = threadIdx.x; i < n_sq; i += 1024) s += sq_part[i];
  } else {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long i = threadIdx.x; i < n4; i += 8 * 1024) {
      float4 q[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) q[k] = i + k * 1024 < n4 ? g4[i + k * 1024] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += q[k].x * q[k].x + q[k].y * q[k].y + q[k].z * q[k].z + q[k].w * q[k].w;
    }
    s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += 1024) s += g[i] * g[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  __shared__ float part[16];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  float sq = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) sq += part[w];
  const float total = sqrtf(sq);
  float coef = 1.f;
  if (max_norm > 0.f) coef = fminf(max_norm / (total + 1e-6f), 1.0f);  // torch clip_grad_norm_
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *sq_out = sq;
    if (norm_out) *norm_out = total;
  }
  const float step_size = d.lr / bc1;
  const float inv_sqrt_bc2 = rsqrtf(bc2);
  auto upd = [&](float& pi, float& mi, float& vi, float gi_raw) {
    float gi = gi_raw * coef;
    if (d.decoupled) pi *= (1.f - d.lr * d.weight_decay);
    else if (d.weight_decay != 0.f) gi += d.weight_decay * pi;
    mi = d.beta1 * mi + (1.f - d.beta1) * gi;
    vi = d.beta2 * vi + (1.f - d.beta2) * gi * gi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + d.eps;
    pi = pi - step_size * (mi / denom);
  };
  if (vec) {
    if (has_own) {
      upd(po.x, mo.x, vo.x, go.x); upd(po.y, mo.y, vo.y, go.y); upd(po.z, mo.z, vo.z, go.z); upd(po.w, mo.w, vo.w, go.w);
      reinterpret_cast<float4*>(p)[own] = po; reinterpret_cast<float4*>(m)[own] = mo; reinterpret_cast<float4*>(v)[own] = vo;
    }
    // (more float4 than threads: the rest in a strided loop)
    for (long i = own + (long)gridDim.x * blockDim.x; i < n4; i += (long)gridDim.x * blockDim.x) {
      float4 pp = reinterpret_cast<const float4*>(p)[i], mm = reinterpret_cast<const float4*>(m)[i];
      float4 vv = reinterpret_cast<const float4*>(v)[i];
      const float4 gg = g4[i];
      upd(pp.x, mm.x, vv.x, gg.x); upd(pp.y, mm.y, vv.y, gg.y); upd(pp.z, mm.z, vv.z, gg.z); upd(pp.w, mm.w, vv.w, gg.w);
      reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
    }
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
      float pi = p[i], mi = m[i], vi = v[i];
      upd(pi, mi, vi, g[i]);
      p[i] = pi; m[i] = mi; v[i] = vi;
    }
  }
}

// the spread loss sums of an epoch call (SfTrcArgs::loss_mask) -> the caller's scalar; the parts are zeroed for the next call
// (values on a 2^-20 grid: the double adds are exact in any order)
__global__ void k_fold_loss(double* __restrict__ part, int n, double* __restrict__ out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) { s += part[i]; part[i] = 0.0; }
    *out += s;
  }
}
hipError_t sf_launch_fold_loss(double* part, int n, double* out, hipStream_t st) {
  hipLaunchKernelGGL(k_fold_loss, dim3(1), dim3(64), 0, st, part, n, out);
  return hipGetLastError();
}

// start of a captured training step (sf_flow_train_epoch): the rows of batch number ctr[0] of the epoch's order -> rows_buf,
// the bias corrections of Adam step ctr[1] + 1 -> bc[0..1]; both counters advance
__global__ void k_step_begin(const long long* __restrict__ order, long long* __restrict__ ctr, long batch, long long* __restrict__ rows_buf,
                             float beta1, float beta2, float* __restrict__ bc) {
  const long long b = ctr[0];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < batch; i += (long)gridDim.x * blockDim.x)
    rows_buf[i] = order[b * batch + i];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const double step = (double)(ctr[1] + 1);
    bc[0] = (float)(1.0 - pow((double)beta1, step));
    bc[1] = (float)(1.0 - pow((double)beta2, step));
  }
}
__global__ void k_step_end(long long* __restrict__ ctr) {
  ctr[0] += 1;
  ctr[1] += 1;
}
hipError_t sf_launch_step_begin(const long long* order, long long* ctr, long batch, long long* rows_buf, float beta1, float beta2,
                                float* bc, hipStream_t st) {
  const int blocks = (int)((batch + 255) / 256 < 64 ? (batch + 255) / 256 : 64);
  hipLaunchKernelGGL(k_step_begin, dim3(blocks < 1 ? 1 : blocks), dim3(256), 0, st, order, ctr, batch, rows_buf, beta1, beta2, bc);
  return hipGetLastError();
}
hipError_t sf_launch_step_end(long long* ctr, hipStream_t st) {
  hipLaunchKernelGGL(k_step_end, dim3(1), dim3(1), 0, st, ctr);
  return hipGetLastError();
}

hipError_t sf_launch_adam(float* params, const float* grad, float* m, float* v, float* norm_scratch, long n,
                          const sf_adam_desc& d, float bc1, float bc2, float max_norm, float* grad_norm_out,
                          hipStream_t st, const float* bc_dev, const float* sq_part, int n_sq) {
  if (n <= 131072 && ((uintptr_t)grad & 15) == 0) {  // (the fused kernel reads the gradient as float4)
    const int nb = (int)((n + 4095) / 4096);
    hipLaunchKernelGGL(k_adam_fused, dim3(nb < 1 ? 1 : nb), dim3(1024), 0, st, params, grad, m, v, norm_scratch, n, d, bc1,
                       bc2, max_norm, grad_norm_out, bc_dev, sq_part, n_sq);
    return hipGetLastError();
  }
  hipError_t e = hipMemsetAsync(norm_scratch, 0, sizeof(float), st);
  if (e != hipSuccess) return e;
  const int blocks = (int)((n + 1023) / 1024 < 256 ? (n + 1023) / 1024 : 256);
  hipLaunchKernelGGL(k_sqnorm, dim3(blocks), dim3(256), 0, st, grad, n, norm_scratch);
  hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, st, params, grad, m, v, norm_scratch, n, d, bc1, bc2,
                     max_norm, grad_norm_out, bc_dev);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// forward + backward of  L = w * (-log p(theta | x))
//
// One wave = one 32-sample tile, end to end: forward with the block activations stashed in HBM
// (tile-register layout, float4 per lane, coalesced), then the backward sweep.
//   data gradients   : delta_in = W^T delta_out  -- same MFMA chaining as the forward, on the
//                      transposed operand image (packedT)
//   weight gradients : dW^T tile = in_tile . delta_tile^T over the 32 samples -- both operands
//                      are transposed through a per-wave LDS scratch ([row][33] floats, conflict
//                      free both ways); the accumulator registers ARE the packed float4 groups, so
//                      they are added to the gradient image with 256-byte-contiguous f32 atomics
//   bias gradients   : row sums of the delta tile, taken from the B operands already read
// ---------------------------------------------------------------------------------------------
__global__ void k_grad_gather(const float* __restrict__ gimg, long stride, int copies, const int32_t* __restrict__ gdst,
                              float* __restrict__ grad, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int g = gdst[i];
  float v = 0.f;
  if (g >= 0) {
    for (int c = 0; c < copies; ++c) v += gimg[(size_t)c * stride + g];  // fixed order over the replicas
  }
  grad[i] = v;
}

// one launch for everything a training step needs before the flow kernel: forward image, transposed image,
// zeroed gradient image (and zeroed context-gradient rows)
__global__ void k_train_prep(const float* __restrict__ flat, const int32_t* __restrict__ s1, const int32_t* __restrict__ s2,
                             float* __restrict__ packed, long n1, const int32_t* __restrict__ t1,
                             const int32_t* __restrict__ t2, float* __restrict__ packedT, long n2,
                             float* __restrict__ gimg, int copies, float* __restrict__ dctx, long n4,
                             const int32_t* __restrict__ u1, const int32_t* __restrict__ u2, float* __restrict__ packed16,
                             long n5, const int32_t* __restrict__ sB, unsigned short* __restrict__ packed16B, long n6,
                             const int32_t* __restrict__ c1, const int32_t* __restrict__ c2, float* __restrict__ imgC, long n7) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n1) {
    const int a = s1[i], b = s2[i];
    float v = 0.f;
    if (a >= 0) v = flat[a];
    if (b >= 0) v += flat[b];
    packed[i] = v;
    for (int c = 0; c < copies; ++c) gimg[(size_t)c * n1 + i] = 0.f;
    return;
  }
  i -= n1;
  if (i < n2) {
    const int a = t1[i], b = t2[i];
    float v = 0.f;
    if (a >= 0) v = flat[a];
    if (b >= 0) v += flat[b];
    packedT[i] = v;
    return;
  }
  i -= n2;
  if (i < n4) { dctx[i] = 0.f; return; }
  i -= n4;
  if (i < n5) {  // 16-row sampler image (so that sampling right after a training step needs no set_params)
    const int a = u1[i], b = u2[i];
    float v = 0.f;
    if (a >= 0) v = flat[a];
    if (b >= 0) v += flat[b];
    else if (b == SF_PACK_TANH_SCALE) v *= SF_TANH_PRESCALE;
    packed16[i] = v;
    return;
  }
  i -= n5;
  if (i < n6) {  // split-bf16 hidden blocks of the persistent sampler (k_pack_bf16_split)
    const int a = sB[i];
    unsigned short r = 0;
    if (a >= 0) {
      const float w = flat[a & SF_PACK_SPLIT_INDEX] * ((a & SF_PACK_SPLIT_SCALED) ? SF_TANH_PRESCALE : 1.0f);
      const __bf16 hi = (__bf16)w;
      r = __builtin_bit_cast(unsigned short, (a >> 30) & 1 ? (__bf16)(w - (float)hi) : hi);
    }
    packed16B[i] = r;
    return;
  }
  i -= n6;
  if (i < n7) {  // cooperative training image (sf_trainc.hip)
    const int a = c1[i], b = c2[i];
    float v = 0.f;
    if (a >= 0) v = flat[a];
    if (b >= 0) v += flat[b];
    imgC[i] = v;
  }
}

#define SF_TDECL(H)                                                                         \
  hipError_t sf_launch_maf_train_h##H(const SfDev&, const SfTrainArgs&, hipStream_t);       \
  hipError_t sf_launch_nsf_train_h##H(const SfDev&, const SfTrainArgs&, hipStream_t);
SF_TDECL(1) SF_TDECL(2) SF_TDECL(3) SF_TDECL(4)

#define SF_TRY(call)                                                         \
  do {                                                                       \
    hipError_t e_ = (call);                                                  \
    if (e_ != hipSuccess) {                                                  \
      err = std::string(#call) + ": " + hipGetErrorString(e_);               \
      return SF_ERR_HIP;                                                     \
    }                                                                        \
  } while (0)

// The epoch loop (sf_flow_train_epoch) asks the gather for the per-block shares of |grad|^2: returns the buffer (grown on demand)
// and notes how many shares the gradient of THIS call comes with; null when nobody asked or the buffer cannot be had.
static float* sq_for(sf_flow* f, const SfLayout& L) {
  f->n_sqpart = 0;
  if (!f->want_sq) return nullptr;
  const long nb = sf_gather_c2_blocks((long)L.n_gradC, f->n_gzeroC);
  if ((size_t)nb > f->sqpart_cap) {
    (void)hipFree(f->d_sqpart);
    f->d_sqpart = nullptr; f->sqpart_cap = 0;
    if (hipMalloc(&f->d_sqpart, (size_t)nb * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    f->sqpart_cap = (size_t)nb;
  }
  f->n_sqpart = (int)nb;
  return f->d_sqpart;
}

int sf_train_loss_grad(sf_flow* f, const float* flat, const float* theta, const float* x, const long long* idx, long B,
                       float grad_scale, const float* weights, float* loss, double* loss_sum, float* grad, float* dctx,
                       hipStream_t st, std::string& err) {
  const SfLayout& L = f->L;
  if (f->nsf1) {  // one-parameter NSF: T context MLPs + a scalar spline chain (sf_nsf1.hip); the handle keeps the vector it was given
    if (dctx) { err = "the one-parameter NSF has no context-gradient path"; return SF_ERR_INVALID; }
    hipError_t e = hipMemcpyAsync(f->d_flat, flat, (size_t)L.n_params * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) { err = std::string("hipMemcpyAsync: ") + hipGetErrorString(e); return SF_ERR_HIP; }
    return sf_nsf1_loss_grad(f->nsf1, flat, theta, x, idx, B, grad_scale, weights, loss, loss_sum, grad, st, err);
  }
  if (f->nsfar) {
    if (dctx) { err = "the autoregressive NSF has no context-gradient path"; return SF_ERR_INVALID; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (f->profiling) {
      if (!f->ev_train[0]) { SF_TRY(hipEventCreate(&f->ev_train[0])); SF_TRY(hipEventCreate(&f->ev_train[1])); }
      e0 = f->ev_train[0]; e1 = f->ev_train[1];
    }
    const int rc = sf_nsfar_loss_grad(f->nsfar, flat, theta, x, idx, B, grad_scale, weights, loss, loss_sum, grad, st, err, e0, e1);
    if (!rc && f->profiling) f->ev_train_valid = true;
    return rc;
  }
  // ---- cooperative 16-row kernels: MAF (sf_trainc.hip: two blocks, D <= 8, T <= SF_TRC_TS, <= 4 hidden tiles) and
  //      NSF (sf_nsfc.hip: two blocks, D <= 8, H <= 64, K <= 11); they share the image / gather machinery
  const bool coop_maf = B > 0 && sf_trainc_eligible(L, dctx != nullptr);
  const bool coop_nsf = B > 0 && !coop_maf && sf_nsfc_eligible(L, dctx != nullptr);
  if (coop_maf || coop_nsf) {
    if (!f->trainc_ready) {
      auto undo = [&]() {
        (void)hipFree(f->d_imgC); (void)hipFree(f->d_sC1); (void)hipFree(f->d_sC2); (void)hipFree(f->d_gdstC);
        (void)hipFree(f->d_gsrcC); (void)hipFree(f->d_gzeroC);
        f->d_imgC = nullptr; f->d_sC1 = f->d_sC2 = f->d_gdstC = f->d_gsrcC = f->d_gzeroC = nullptr; f->n_gzeroC = 0;
      };
#define SF_TRY_C(call)                                                       \
  do {                                                                       \
    hipError_t e_ = (call);                                                  \
    if (e_ != hipSuccess) {                                                  \
      err = std::string(#call) + ": " + hipGetErrorString(e_);               \
      undo();                                                                \
      return SF_ERR_HIP;                                                     \
    }                                                                        \
  } while (0)
      SF_TRY_C(hipMalloc(&f->d_imgC, (size_t)L.n_imgC * sizeof(float)));
      SF_TRY_C(hipMalloc(&f->d_sC1, (size_t)L.n_imgC * sizeof(int32_t)));
      SF_TRY_C(hipMalloc(&f->d_sC2, (size_t)L.n_imgC * sizeof(int32_t)));
      SF_TRY_C(hipMemcpy(f->d_sC1, L.srcC1.data(), (size_t)L.n_imgC * sizeof(int32_t), hipMemcpyHostToDevice));
      SF_TRY_C(hipMemcpy(f->d_sC2, L.srcC2.data(), (size_t)L.n_imgC * sizeof(int32_t), hipMemcpyHostToDevice));
      SF_TRY_C(hipMalloc(&f->d_gdstC, (size_t)L.n_params * sizeof(int32_t)));
      SF_TRY_C(hipMemcpy(f->d_gdstC, L.gdstC.data(), (size_t)L.n_params * sizeof(int32_t), hipMemcpyHostToDevice));
      {
        // position -> parameter(s), for the gather that walks the partials in THEIR order (coalesced reads): a position
        // feeds at most two parameters (b0 and bc share one); if the packer ever maps more, the parameter-order gather stays
        std::vector<int32_t> inv((size_t)L.n_gradC * 2, -1), zero;
        bool ok = true;
        for (long i = 0; i < (long)L.n_params && ok; ++i) {
          const int32_t g = L.gdstC[(size_t)i];
          if (g < 0) { zero.push_back((int32_t)i); continue; }
          if (g >= (int32_t)L.n_gradC) { ok = false; break; }
          if (inv[(size_t)g * 2] < 0) inv[(size_t)g * 2] = (int32_t)i;
          else if (inv[(size_t)g * 2 + 1] < 0) inv[(size_t)g * 2 + 1] = (int32_t)i;
          else ok = false;
        }
        if (ok) {
          SF_TRY_C(hipMalloc(&f->d_gsrcC, inv.size() * sizeof(int32_t)));
          SF_TRY_C(hipMemcpy(f->d_gsrcC, inv.data(), inv.size() * sizeof(int32_t), hipMemcpyHostToDevice));
          f->n_gzeroC = (long)zero.size();
          if (!zero.empty()) {
            SF_TRY_C(hipMalloc(&f->d_gzeroC, zero.size() * sizeof(int32_t)));
            SF_TRY_C(hipMemcpy(f->d_gzeroC, zero.data(), zero.size() * sizeof(int32_t), hipMemcpyHostToDevice));
          }
        }
      }
#undef SF_TRY_C
      f->trainc_ready = true;
    }
    const int grid = coop_maf ? sf_trainc_grid(B, &L.trc, L.dev.T) : sf_nsfc_grid(B, L.nsc.NT);
    // Gradient accumulation (sf_fixacc.h): per-workgroup partials + gather while they are few, above that 2^-40 fixed-point
    // contributions added with int64 atomics into one zeroed replica per XCD (L2-resident, order independent).  The gather
    // needs the position -> parameter table (d_gsrcC); SF_GRAD_ACC=partial | fix forces one form.
    const int nsf_mode = coop_nsf ? sf_nsfc_acc_mode(B, grid, (long)L.n_gradC) : 0;
    bool use_fix = coop_nsf ? nsf_mode != 0 : sf_trainc_fix(grid, (long)L.n_gradC);
    if (!f->d_gsrcC) use_fix = false;
    if (use_fix && !f->d_gfixC) {
      SF_TRY(hipMalloc(&f->d_gfixC, (size_t)SF_FIX_REPLICAS * (size_t)L.n_gradC * sizeof(long long)));
    }
    if (use_fix) SF_TRY(hipMemsetAsync(f->d_gfixC, 0, (size_t)SF_FIX_REPLICAS * (size_t)L.n_gradC * sizeof(long long), st));
    const int n_part = use_fix ? 1 : grid;   // (fix: d_gpartC is only the base the job descriptors count from)
    const size_t need = (size_t)n_part * (size_t)L.n_gradC;
    if (need > f->gpartC_cap) {
      if (f->d_gpartC) SF_TRY(hipFree(f->d_gpartC));
      f->d_gpartC = nullptr; f->gpartC_cap = 0;
      SF_TRY(hipMalloc(&f->d_gpartC, need * sizeof(float)));
      f->gpartC_cap = need;
    }
    {
      // inside an epoch call (sf_flow_train_epoch, all steps but its last) only the cooperative image is re-tiled: the density
      // and sampler images are not read by the training kernels, and nobody can look at them before the call returns
      const bool lite = f->prep_lite;
      const long n1 = lite ? 0 : (long)L.n_packed;
      const long n4 = dctx ? B * (long)L.dev.C : 0;
      const long n5 = (!lite && f->d_packed16) ? (long)L.n_packed16 : 0;
      const long n6 = (!lite && f->d_packed16B) ? (long)L.n_packed16B : 0;
      const long n7 = (long)L.n_imgC;
      const long tot = n1 + n4 + n5 + n6 + n7;
      hipLaunchKernelGGL(k_train_prep, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, flat, f->d_s1, f->d_s2, f->d_packed,
                         n1, (const int32_t*)nullptr, (const int32_t*)nullptr, (float*)nullptr, 0L, (float*)nullptr, 0,
                         dctx, n4, f->d_s16a, f->d_s16b, f->d_packed16, n5, f->d_s16B, f->d_packed16B, n6, f->d_sC1, f->d_sC2,
                         f->d_imgC, n7);
      SF_TRY(hipGetLastError());
      f->packed16_stale = lite;
      f->packed_stale = lite;
      f->wp_stale = true;
    }
    if (L.n_packedB > 0) SF_TRY(sf_launch_pack_bf16(flat, f->d_bsrc, f->d_packedB, (long)L.n_packedB, st));
    if (coop_nsf) {
      const long n_chunks = (B + 31) / 32;
      const size_t ust_need = (size_t)n_chunks * 32 * (size_t)L.dev.T * 16;
      if (ust_need > f->ustash_cap) {
        if (f->d_ustash) SF_TRY(hipFree(f->d_ustash));
        f->d_ustash = nullptr; f->ustash_cap = 0;
        SF_TRY(hipMalloc(&f->d_ustash, ust_need * sizeof(float)));
        f->ustash_cap = ust_need;
      }
      const SfDev& v = L.dev;
      SfNscArgs a;
      a.c = L.nsc;
      a.img = f->d_imgC; a.cst = f->d_cst;
      a.D = v.D; a.C = v.C; a.T = v.T; a.K = v.K;
      a.tail_bound = v.tail_bound; a.min_w = v.min_w; a.min_h = v.min_h; a.min_d = v.min_d; a.lu_eps = v.lu_eps;
      a.inv_sqrt_h = v.inv_sqrt_h; a.deriv_const = v.deriv_const; a.logdet0 = v.logdet0;
      a.c_pscale = v.c_pscale; a.c_pshift = v.c_pshift; a.c_xmean = v.c_xmean; a.c_xstd = v.c_xstd;
      a.theta = theta; a.x = x; a.idx = idx; a.wts = weights; a.B = B; a.n_chunks = n_chunks; a.w = grad_scale;
      a.loss = loss; a.loss_sum = loss_sum; a.loss_mask = 0;
      if (loss_sum && f->d_losspart) { a.loss_sum = f->d_losspart; a.loss_mask = SF_LOSS_PARTS - 1; f->losspart_used = true; }
      a.gpart = f->d_gpartC; a.gpart_stride = (long)L.n_gradC; a.fix = use_fix ? f->d_gfixC : nullptr; a.fix_mode = nsf_mode;
      a.ustash = f->d_ustash;
#ifdef SF_NSC_TRACE
      a.trace = nullptr;
#endif
      if (f->profiling) {
        if (!f->ev_train[0]) { SF_TRY(hipEventCreate(&f->ev_train[0])); SF_TRY(hipEventCreate(&f->ev_train[1])); }
        SF_TRY(hipEventRecord(f->ev_train[0], st));
      }
      SF_TRY(sf_launch_nsf_trainc(a, grid, st));
      if (f->profiling) {
        SF_TRY(hipEventRecord(f->ev_train[1], st));
        f->ev_train_valid = true;
      }
      if (use_fix && nsf_mode == 2)   // float replicas: the partial gather over SF_FIX_REPLICAS images
        SF_TRY(sf_launch_gather_c2(reinterpret_cast<const float*>(f->d_gfixC), (long)L.n_gradC, SF_FIX_REPLICAS, f->d_gsrcC, f->d_gzeroC, f->n_gzeroC, grad, st, sq_for(f, L)));
      else if (use_fix) SF_TRY(sf_launch_gather_fix(f->d_gfixC, (long)L.n_gradC, SF_FIX_REPLICAS, f->d_gsrcC, f->d_gzeroC, f->n_gzeroC, grad, st));
      else if (f->d_gsrcC) SF_TRY(sf_launch_gather_c2(f->d_gpartC, (long)L.n_gradC, n_part, f->d_gsrcC, f->d_gzeroC, f->n_gzeroC, grad, st, sq_for(f, L)));
      else SF_TRY(sf_launch_gather_c(f->d_gpartC, (long)L.n_gradC, n_part, f->d_gdstC, grad, (long)L.n_params, st));
      return SF_OK;
    }
    SfTrcArgs a;
    a.c = L.trc;
    a.img = f->d_imgC; a.cst = f->d_cst;
    a.D = L.dev.D; a.C = L.dev.C; a.T = L.dev.T; a.scale_fn = L.dev.scale_fn;
    a.eps = L.dev.eps; a.logdet0 = L.dev.logdet0;
    a.c_pscale = L.dev.c_pscale; a.c_pshift = L.dev.c_pshift; a.c_tdim = L.dev.c_tdim; a.c_xmean = L.dev.c_xmean; a.c_xstd = L.dev.c_xstd;
    a.theta = theta; a.x = x; a.idx = idx; a.wts = weights; a.B = B; a.n_chunks = (B + 32L * sf_trainc_groups(B, &L.trc, L.dev.T) - 1) / (32L * sf_trainc_groups(B, &L.trc, L.dev.T)); a.w = grad_scale;
    a.loss = loss; a.loss_sum = loss_sum; a.dctx = dctx; a.loss_mask = 0;
    if (loss_sum && f->d_losspart) { a.loss_sum = f->d_losspart; a.loss_mask = SF_LOSS_PARTS - 1; f->losspart_used = true; }
    a.gpart = f->d_gpartC; a.gpart_stride = (long)L.n_gradC; a.fix = use_fix ? f->d_gfixC : nullptr;
    if (f->profiling) {
      if (!f->ev_train[0]) { SF_TRY(hipEventCreate(&f->ev_train[0])); SF_TRY(hipEventCreate(&f->ev_train[1])); }
      SF_TRY(hipEventRecord(f->ev_train[0], st));
    }
    SF_TRY(sf_launch_maf_trainc(a, grid, st));
    if (f->profiling) {
      SF_TRY(hipEventRecord(f->ev_train[1], st));
      f->ev_train_valid = true;
    }
    if (use_fix) SF_TRY(sf_launch_gather_fix(f->d_gfixC, (long)L.n_gradC, SF_FIX_REPLICAS, f->d_gsrcC, f->d_gzeroC, f->n_gzeroC, grad, st));
    else if (f->d_gsrcC) SF_TRY(sf_launch_gather_c2(f->d_gpartC, (long)L.n_gradC, grid, f->d_gsrcC, f->d_gzeroC, f->n_gzeroC, grad, st, sq_for(f, L)));
    else SF_TRY(sf_launch_gather_c(f->d_gpartC, (long)L.n_gradC, grid, f->d_gdstC, grad, (long)L.n_params, st));
    return SF_OK;
  }
  // ---- lazily built training state
  if (!f->train_ready) {
    // all-or-nothing: a failed allocation frees what was already taken, so that the next call starts over instead of
    // launching kernels on half-built state
    auto undo = [&]() {
      (void)hipFree(f->d_packedT); (void)hipFree(f->d_t1); (void)hipFree(f->d_t2); (void)hipFree(f->d_gpacked); (void)hipFree(f->d_gdst);
      f->d_packedT = nullptr; f->d_t1 = f->d_t2 = nullptr; f->d_gpacked = nullptr; f->d_gdst = nullptr; f->gpacked_cap = 0;
    };
#define SF_TRY_U(call)                                                       \
  do {                                                                       \
    hipError_t e_ = (call);                                                  \
    if (e_ != hipSuccess) {                                                  \
      err = std::string(#call) + ": " + hipGetErrorString(e_);               \
      undo();                                                                \
      return SF_ERR_HIP;                                                     \
    }                                                                        \
  } while (0)
    SF_TRY_U(hipMalloc(&f->d_packedT, (size_t)L.n_packedT * sizeof(float)));
    SF_TRY_U(hipMalloc(&f->d_t1, (size_t)L.n_packedT * sizeof(int32_t)));
    SF_TRY_U(hipMalloc(&f->d_t2, (size_t)L.n_packedT * sizeof(int32_t)));
    SF_TRY_U(hipMemcpy(f->d_t1, L.srcT1.data(), (size_t)L.n_packedT * sizeof(int32_t), hipMemcpyHostToDevice));
    SF_TRY_U(hipMemcpy(f->d_t2, L.srcT2.data(), (size_t)L.n_packedT * sizeof(int32_t), hipMemcpyHostToDevice));
    SF_TRY_U(hipMalloc(&f->d_gpacked, (size_t)SF_GCOPIES * L.n_packed * sizeof(float)));
    f->gpacked_cap = (size_t)SF_GCOPIES * L.n_packed;
    SF_TRY_U(hipMalloc(&f->d_gdst, (size_t)L.n_params * sizeof(int32_t)));
    SF_TRY_U(hipMemcpy(f->d_gdst, L.gdst.data(), (size_t)L.n_params * sizeof(int32_t), hipMemcpyHostToDevice));
#undef SF_TRY_U
    f->train_ready = true;
  }
  const SfDev& v = L.dev;
  const long waves = (B + 31) / 32;
  const long tiles_per_wave = (long)v.T * (v.kind == SF_MAF ? ((v.NB + 1) * v.HT + 1) : (2 + (3 * v.NB + 1) * v.HT));
  const long act_per_wave = tiles_per_wave * 4 * 64;  // float4
  const size_t need = (size_t)waves * act_per_wave * 4;
  if (need > f->act_cap) {
    if (f->d_act) SF_TRY(hipFree(f->d_act));
    f->d_act = nullptr;
    f->act_cap = 0;
    SF_TRY(hipMalloc(&f->d_act, need * sizeof(float)));
    f->act_cap = need;
  }
  // gradient accumulation: one replica per tile + plain stores (bitwise reproducible, and cheaper than atomics) up to
  // 16 tiles (batch 512: the reference's batch sizes), or whenever SF_DETERMINISTIC=1 and the replicas fit 2 GiB
  // (batch 2048: +15 % step time for zeroing and summing 64 replicas); else per-XCD replicas + f32 atomics
  static int force_det = -1;
  if (force_det < 0) { const char* e = std::getenv("SF_DETERMINISTIC"); force_det = e ? std::atoi(e) : 0; }
  const bool det = waves > 0 && (waves <= 16 || (force_det == 1 && (size_t)waves * L.n_packed * sizeof(float) <= ((size_t)2 << 30)));
  const int copies = det ? (int)waves : SF_GCOPIES;
  if ((size_t)copies * L.n_packed > f->gpacked_cap) {
    SF_TRY(hipFree(f->d_gpacked));
    f->d_gpacked = nullptr; f->gpacked_cap = 0;
    SF_TRY(hipMalloc(&f->d_gpacked, (size_t)copies * L.n_packed * sizeof(float)));
    f->gpacked_cap = (size_t)copies * L.n_packed;
  }
  {
    const long n4 = (dctx && B > 0) ? B * (long)L.dev.C : 0;
    const long n5 = f->d_packed16 ? (long)L.n_packed16 : 0;
    const long n6 = f->d_packed16B ? (long)L.n_packed16B : 0;
    const long tot = (long)L.n_packed + (long)L.n_packedT + n4 + n5 + n6;
    hipLaunchKernelGGL(k_train_prep, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, flat, f->d_s1, f->d_s2, f->d_packed,
                       (long)L.n_packed, f->d_t1, f->d_t2, f->d_packedT, (long)L.n_packedT, f->d_gpacked, copies, dctx, n4,
                       f->d_s16a, f->d_s16b, f->d_packed16, n5, f->d_s16B, f->d_packed16B, n6, (const int32_t*)nullptr,
                       (const int32_t*)nullptr, (float*)nullptr, 0L);
    SF_TRY(hipGetLastError());
  }
  f->packed16_stale = false;
  f->wp_stale = true;
  if (L.n_packedB > 0) SF_TRY(sf_launch_pack_bf16(flat, f->d_bsrc, f->d_packedB, (long)L.n_packedB, st));
  if (B > 0) {
    SfTrainArgs a;
    a.theta = theta; a.x = x; a.idx = idx; a.loss_sum = loss_sum; a.B = B; a.w = grad_scale; a.wts = weights; a.loss = loss; a.dctx = dctx; a.gimg = f->d_gpacked; a.gimg_stride = (long)L.n_packed; a.det = det ? 1 : 0;
    a.act = reinterpret_cast<float4*>(f->d_act); a.act_per_wave = act_per_wave;
    const SfDev m = f->dev();
    const bool maf = m.kind == SF_MAF;
    if (f->profiling) {
      if (!f->ev_train[0]) { SF_TRY(hipEventCreate(&f->ev_train[0])); SF_TRY(hipEventCreate(&f->ev_train[1])); }
      SF_TRY(hipEventRecord(f->ev_train[0], st));
    }
    switch (m.HT) {
      case 1: SF_TRY(maf ? sf_launch_maf_train_h1(m, a, st) : sf_launch_nsf_train_h1(m, a, st)); break;
      case 2: SF_TRY(maf ? sf_launch_maf_train_h2(m, a, st) : sf_launch_nsf_train_h2(m, a, st)); break;
      case 3: SF_TRY(maf ? sf_launch_maf_train_h3(m, a, st) : sf_launch_nsf_train_h3(m, a, st)); break;
      case 4: SF_TRY(maf ? sf_launch_maf_train_h4(m, a, st) : sf_launch_nsf_train_h4(m, a, st)); break;
      default: err = "bad HT"; return SF_ERR_INVALID;
    }
    if (f->profiling) {
      SF_TRY(hipEventRecord(f->ev_train[1], st));
      f->ev_train_valid = true;
    }
  }
  hipLaunchKernelGGL(k_grad_gather, dim3((unsigned)((L.n_params + 255) / 256)), dim3(256), 0, st,
                     f->d_gpacked, (long)L.n_packed, copies, f->d_gdst, grad, (long)L.n_params);
  SF_TRY(hipGetLastError());
  return SF_OK;
}
