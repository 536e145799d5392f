// sf_train.hip -- training kernels.
//   (1) fused global-norm clip + Adam/AdamW      ref: custom_runner.py:613-618
//   (2) forward + backward of -log_prob          ref: custom_runner.py:604-610
#include "sf_train.h"

#include <hip/hip_runtime.h>

#include "sf_flows.h"
#include "sf_internal.h"

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
                       float bc2, float max_norm, float* __restrict__ norm_out) {
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

hipError_t sf_launch_adam(float* params, const float* grad, float* m, float* v, float* norm_scratch, long n,
                          const sf_adam_desc& d, float bc1, float bc2, float max_norm, float* grad_norm_out,
                          hipStream_t st) {
  hipError_t e = hipMemsetAsync(norm_scratch, 0, sizeof(float), st);
  if (e != hipSuccess) return e;
  const int blocks = (int)((n + 1023) / 1024 < 256 ? (n + 1023) / 1024 : 256);
  hipLaunchKernelGGL(k_sqnorm, dim3(blocks), dim3(256), 0, st, grad, n, norm_scratch);
  hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, st, params, grad, m, v, norm_scratch, n, d, bc1, bc2,
                     max_norm, grad_norm_out);
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
#define SF_TL 1056  // floats per transposed tile in LDS: 32 rows x 33

struct SfTrainArgs {
  const float* theta;
  const float* x;
  long B;
  float w;           // gradient weight of every sample (grad_scale)
  const float* wts;  // optional per-sample weights [B] (multiplied by w)
  float* loss;       // [B] or null
  float* gimg;       // gradient image
  float4* act;       // activation stash
  long act_per_wave; // float4 per wave
};

__device__ __forceinline__ void sf_tile_to_lds(float* __restrict__ dst, const f32x16& t, int c, int h) {
#pragma unroll
  for (int r = 0; r < 16; ++r) dst[sf_row(r, h) * 33 + c] = t[r];
}
__device__ __forceinline__ void sf_stash_store(float4* __restrict__ base, int tile, const f32x16& t, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
    base[(tile * 4 + q) * 64 + lane] = make_float4(t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]);
}
__device__ __forceinline__ void sf_stash_load(const float4* __restrict__ base, int tile, f32x16& t, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 v = base[(tile * 4 + q) * 64 + lane];
    t[4 * q] = v.x; t[4 * q + 1] = v.y; t[4 * q + 2] = v.z; t[4 * q + 3] = v.w;
  }
}

// gradient of one linear layer's weights (and optionally bias):
//   gw block [mt][kg][j][lane] += sum_s in[i][s] * delta[o][s]
// lds: (IT + OT) transposed tiles; in tiles first.
template <int OT, int IT>
__device__ __forceinline__ void sf_grad_w(float* __restrict__ lds, const f32x16 (&delta)[OT][1],
                                          const f32x16 (&in)[IT][1], float* __restrict__ gw,
                                          float* __restrict__ gb, int nGtot, int kg0, int ng, int lane) {
  const int c = lane & 31, h = lane >> 5;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int kt = 0; kt < IT; ++kt) sf_tile_to_lds(lds + kt * SF_TL, in[kt][0], c, h);
#pragma unroll
  for (int mt = 0; mt < OT; ++mt) sf_tile_to_lds(lds + (IT + mt) * SF_TL, delta[mt][0], c, h);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int rd = c * 33 + h;
#pragma unroll
  for (int mt = 0; mt < OT; ++mt) {
    float bsum = 0.f;
    const float* ld = lds + (IT + mt) * SF_TL + rd;
#pragma unroll
    for (int kt = 0; kt < IT; ++kt) {
      if (kt * 4 < ng) {
        const float* li = lds + kt * SF_TL + rd;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const float a = li[2 * k];
          const float b = ld[2 * k];
          if (kt == 0) bsum += b;
          acc = SF_MFMA(a, b, acc);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (kt * 4 + g < ng) {
            float* dst = gw + (((size_t)mt * nGtot + kg0 + kt * 4 + g) * 4) * 64 + lane;
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(dst + j * 64, acc[4 * g + j]);
          }
      }
    }
    if (gb) {
      bsum += sf_xhalf(bsum);
      if (h == 0) atomicAdd(gb + mt * 32 + c, bsum);
    }
  }
  __builtin_amdgcn_wave_barrier();
}

template <int HT>
__global__ __launch_bounds__(256) void k_maf_train(SfDev m, SfTrainArgs a) {
  extern __shared__ float lds_all[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const long wid = (long)blockIdx.x * 4 + wave;
  const long base = wid * 32;
  if (base >= a.B) return;
  float* lds = lds_all + wave * (2 * HT) * SF_TL;
  float4* stash = a.act + wid * a.act_per_wave;
  const int TPT = (m.NB + 1) * HT + 1;  // stash tiles per transform: u, h0, a_1..a_NB

  const long row = base + c;
  const bool valid = row < a.B;
  const long ii = valid ? row : a.B - 1;
  const float* xr[1] = {a.x + ii * m.C};
  float u[1][SF_DMAX];
  float logdet[1] = {m.logdet0};
#pragma unroll
  for (int p = 0; p < SF_DMAX; ++p) {
    u[0][p] = 0.f;
    if (p < m.D) {
      const int td = (int)m.cst[m.c_tdim + p];
      u[0][p] = a.theta[ii * m.D + td] * m.cst[m.c_pscale + p] + m.cst[m.c_pshift + p];
    }
  }
  using Ops = MafOps<HT, 1>;

  // ------------------------------------------------------------------ forward (with stash)
  for (int t = 0; t < m.T; ++t) {
    const float* tp = m.packed + (size_t)t * m.t_stride;
    {
      f32x16 ut;
#pragma unroll
      for (int p = 0; p < SF_DMAX; ++p) ut[p] = u[0][p];
      sf_stash_store(stash, t * TPT, ut, lane);
    }
    f32x16 act[HT][1];
    sf_init_bias<HT, 1>(act, tp + m.o_b0, h);
    {
      f32x16 ut[1][1];
      sf_build_u_tile<1>(ut, u, h);
      sf_mm_acc<HT, 1, 1, false>(act, ut, tp + m.o_w0, m.nGu, 0, m.nGu, lane);
    }
    sf_ctx_mm<HT, 1>(act, xr, m, tp + m.o_wc, lane);
#pragma unroll
    for (int mt = 0; mt < HT; ++mt) sf_stash_store(stash, t * TPT + 1 + mt, act[mt][0], lane);
#pragma unroll
    for (int k = 0; k < SF_NBMAX; ++k) {
      if (k < m.NB) {
        f32x16 b[HT][1];
        sf_init_bias<HT, 1>(b, tp + m.o_bk[k], h);
        sf_mm_acc<HT, 1, HT, false>(b, act, tp + m.o_wk[k], m.nGh, 0, m.nGh, lane);
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) {
#pragma unroll
          for (int r = 0; r < 16; ++r) act[mt][0][r] = sf_tanh(b[mt][0][r]);
          sf_stash_store(stash, t * TPT + 1 + (k + 1) * HT + mt, act[mt][0], lane);
        }
      }
    }
    f32x16 fin[1][1];
    sf_init_bias<1, 1>(fin, tp + m.o_bf, h);
    sf_mm_acc<1, 1, HT, false>(fin, act, tp + m.o_wf, m.nGh, 0, m.nGh, lane);
    float ld = 0.f;
#pragma unroll
    for (int p = 0; p < SF_DMAX; ++p) {
      if (p < m.D) {
        const float s = Ops::scale(m, fin[0][0][2 * (p >> 1)]);
        const float val = s * u[0][p] + fin[0][0][2 * (p >> 1) + 1];
        const bool mine = (h == (p & 1));
        const float oth = sf_xhalf(val);
        u[0][p] = mine ? val : oth;
        ld += mine ? logf(s) : 0.f;
      }
    }
    logdet[0] += ld + sf_xhalf(ld);
  }
  float G[SF_DMAX];  // dL/d(output of the current transform), replicated in both halves
  const float w = valid ? (a.wts ? a.w * a.wts[row] : a.w) : 0.f;
  {
    float ss = 0.f;
#pragma unroll
    for (int p = 0; p < SF_DMAX; ++p) {
      G[p] = 0.f;
      if (p < m.D) {
        ss += u[0][p] * u[0][p];
        G[p] = w * u[0][p];
      }
    }
    if (a.loss && valid && h == 0)
      a.loss[row] = 0.5f * ss + 0.5f * (float)m.D * 1.8378770664093453f - logdet[0];
  }

  // ------------------------------------------------------------------ backward
  for (int t = m.T - 1; t >= 0; --t) {
    const float* tp = m.packed + (size_t)t * m.t_stride;
    const float* tpT = m.packedT + (size_t)t * m.tT_stride;
    float* gp = a.gimg + (size_t)t * m.t_stride;
    float uin[1][SF_DMAX];
    {
      f32x16 ut;
      sf_stash_load(stash, t * TPT, ut, lane);
#pragma unroll
      for (int p = 0; p < SF_DMAX; ++p) uin[0][p] = ut[p];
    }
    f32x16 ak[HT][1];  // activation feeding the layer whose gradient is being formed
#pragma unroll
    for (int mt = 0; mt < HT; ++mt) sf_stash_load(stash, t * TPT + 1 + m.NB * HT + mt, ak[mt][0], lane);
    // recompute the head
    f32x16 fin[1][1];
    sf_init_bias<1, 1>(fin, tp + m.o_bf, h);
    sf_mm_acc<1, 1, HT, false>(fin, ak, tp + m.o_wf, m.nGh, 0, m.nGh, lane);
    f32x16 dfin[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) dfin[0][0][r] = 0.f;
    float Gd[SF_DMAX];
#pragma unroll
    for (int p = 0; p < SF_DMAX; ++p) {
      Gd[p] = 0.f;
      if (p < m.D) {
        const float av = fin[0][0][2 * (p >> 1)];
        const float s = Ops::scale(m, av);
        const float dsda = (m.scale_fn == 0) ? sf_sigmoid(av)
                                             : sf_sigmoid(av + 2.0f) * (1.0f - sf_sigmoid(av + 2.0f));
        const float ds = G[p] * uin[0][p] - w / s;
        const bool mine = (h == (p & 1));
        dfin[0][0][2 * (p >> 1)] = mine ? ds * dsda : dfin[0][0][2 * (p >> 1)];
        dfin[0][0][2 * (p >> 1) + 1] = mine ? G[p] : dfin[0][0][2 * (p >> 1) + 1];
        const float gd = G[p] * s;
        const float oth = sf_xhalf(gd);
        Gd[p] = mine ? gd : oth;
      }
    }
    // head: dWf, dbf ; delta_h = Wf^T dfin
    sf_grad_w<1, HT>(lds, dfin, ak, gp + m.o_wf, gp + m.o_bf, m.nGh, 0, m.nGh, lane);
    f32x16 dh[HT][1];
#pragma unroll
    for (int mt = 0; mt < HT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dh[mt][0][r] = 0.f;
    sf_mm_acc<HT, 1, 1, false>(dh, dfin, tpT + m.oT_wf, m.nGf, 0, m.nGf, lane);
#pragma unroll
    for (int kk = 0; kk < SF_NBMAX; ++kk) {
      const int k = SF_NBMAX - 1 - kk;
      if (k < m.NB) {
        f32x16 dpre[HT][1];
#pragma unroll
        for (int mt = 0; mt < HT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) dpre[mt][0][r] = dh[mt][0][r] * (1.0f - ak[mt][0][r] * ak[mt][0][r]);
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) sf_stash_load(stash, t * TPT + 1 + k * HT + mt, ak[mt][0], lane);
        sf_grad_w<HT, HT>(lds, dpre, ak, gp + m.o_wk[k], gp + m.o_bk[k], m.nGh, 0, m.nGh, lane);
#pragma unroll
        for (int mt = 0; mt < HT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) dh[mt][0][r] = 0.f;
        sf_mm_acc<HT, 1, HT, false>(dh, dpre, tpT + m.oT_wk[k], m.nGh, 0, m.nGh, lane);
      }
    }
    // initial layer: dW0 (u tile), dWc (context tiles), d(b0+bc)
    {
      f32x16 ut[1][1];
      sf_build_u_tile<1>(ut, uin, h);
      sf_grad_w<HT, 1>(lds, dh, ut, gp + m.o_w0, gp + m.o_b0, m.nGu, 0, m.nGu, lane);
    }
    for (int kt = 0; kt * 4 < m.nGc; ++kt) {
      f32x16 ct[1][1];
      sf_build_ctx_tile<1>(ct, xr, m, kt, h);
      sf_grad_w<HT, 1>(lds, dh, ct, gp + m.o_wc, nullptr, m.nGc, kt * 4, min(4, m.nGc - kt * 4), lane);
    }
    // delta_u = W0^T delta_h0
    f32x16 du[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) du[0][0][r] = 0.f;
    sf_mm_acc<1, 1, HT, false>(du, dh, tpT + m.oT_w0, m.nGh, 0, m.nGh, lane);
#pragma unroll
    for (int p = 0; p < SF_DMAX; ++p) {
      if (p < m.D) {
        const float v = du[0][0][(p & 3) + 4 * (p >> 3)];
        const float oth = sf_xhalf(v);
        G[p] = Gd[p] + ((h == ((p >> 2) & 1)) ? v : oth);
      }
    }
  }
}

__global__ void k_grad_gather(const float* __restrict__ gimg, const int32_t* __restrict__ gdst,
                              float* __restrict__ grad, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int g = gdst[i];
  grad[i] = g >= 0 ? gimg[g] : 0.f;
}

template <int HT>
static hipError_t launch_maf_train(const SfDev& m, const SfTrainArgs& a, hipStream_t st) {
  const long waves = (a.B + 31) / 32;
  const long grid = (waves + 3) / 4;
  const size_t shmem = (size_t)4 * (2 * HT) * SF_TL * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_maf_train<HT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)shmem);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((k_maf_train<HT>), dim3((unsigned)grid), dim3(256), shmem, st, m, a);
  return hipGetLastError();
}

#define SF_TRY(call)                                                         \
  do {                                                                       \
    hipError_t e_ = (call);                                                  \
    if (e_ != hipSuccess) {                                                  \
      err = std::string(#call) + ": " + hipGetErrorString(e_);               \
      return SF_ERR_HIP;                                                     \
    }                                                                        \
  } while (0)

int sf_train_loss_grad(sf_flow* f, const float* flat, const float* theta, const float* x, long B,
                       float grad_scale, const float* weights, float* loss, float* grad, hipStream_t st,
                       std::string& err) {
  const SfLayout& L = f->L;
  if (L.dev.kind != SF_MAF) {
    err = "sf_flow_loss_grad: the NSF backward kernels are not built in this library yet";
    return SF_ERR_STATE;
  }
  // ---- lazily built training state
  if (!f->d_packedT) {
    SF_TRY(hipMalloc(&f->d_packedT, (size_t)L.n_packedT * sizeof(float)));
    SF_TRY(hipMalloc(&f->d_t1, (size_t)L.n_packedT * sizeof(int32_t)));
    SF_TRY(hipMalloc(&f->d_t2, (size_t)L.n_packedT * sizeof(int32_t)));
    SF_TRY(hipMemcpy(f->d_t1, L.srcT1.data(), (size_t)L.n_packedT * sizeof(int32_t), hipMemcpyHostToDevice));
    SF_TRY(hipMemcpy(f->d_t2, L.srcT2.data(), (size_t)L.n_packedT * sizeof(int32_t), hipMemcpyHostToDevice));
    SF_TRY(hipMalloc(&f->d_gpacked, (size_t)L.n_packed * sizeof(float)));
    SF_TRY(hipMalloc(&f->d_gdst, (size_t)L.n_params * sizeof(int32_t)));
    SF_TRY(hipMemcpy(f->d_gdst, L.gdst.data(), (size_t)L.n_params * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  const SfDev& v = L.dev;
  const long waves = (B + 31) / 32;
  const long tiles_per_wave = (long)v.T * ((v.NB + 1) * v.HT + 1);
  const long act_per_wave = tiles_per_wave * 4 * 64;  // float4
  const size_t need = (size_t)waves * act_per_wave * 4;
  if (need > f->act_cap) {
    if (f->d_act) SF_TRY(hipFree(f->d_act));
    f->d_act = nullptr;
    f->act_cap = 0;
    SF_TRY(hipMalloc(&f->d_act, need * sizeof(float)));
    f->act_cap = need;
  }
  SF_TRY(sf_launch_pack(flat, f->d_s1, f->d_s2, f->d_packed, (long)L.n_packed, st));
  SF_TRY(sf_launch_pack(flat, f->d_t1, f->d_t2, f->d_packedT, (long)L.n_packedT, st));
  SF_TRY(hipMemsetAsync(f->d_gpacked, 0, (size_t)L.n_packed * sizeof(float), st));
  if (B > 0) {
    SfTrainArgs a;
    a.theta = theta; a.x = x; a.B = B; a.w = grad_scale; a.wts = weights; a.loss = loss; a.gimg = f->d_gpacked;
    a.act = reinterpret_cast<float4*>(f->d_act); a.act_per_wave = act_per_wave;
    const SfDev m = f->dev();
    switch (m.HT) {
      case 1: SF_TRY(launch_maf_train<1>(m, a, st)); break;
      case 2: SF_TRY(launch_maf_train<2>(m, a, st)); break;
      case 3: SF_TRY(launch_maf_train<3>(m, a, st)); break;
      case 4: SF_TRY(launch_maf_train<4>(m, a, st)); break;
      default: err = "bad HT"; return SF_ERR_INVALID;
    }
  }
  hipLaunchKernelGGL(k_grad_gather, dim3((unsigned)((L.n_params + 255) / 256)), dim3(256), 0, st,
                     f->d_gpacked, f->d_gdst, grad, (long)L.n_params);
  SF_TRY(hipGetLastError());
  return SF_OK;
}
