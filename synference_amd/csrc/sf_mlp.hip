// sf_mlp.hip -- the fully connected embedding net (context path) on the
// register-tile engine: forward, and backward (weight gradients through the same LDS-transposed MFMA
// + f32-atomic gradient image as the flow training kernels).  Standardisation of x is fused in.
#include <hip/hip_runtime.h>

#include <string>

#include "sf_train_kernels.h"

struct SfMlpArgs {
  const float* x;
  const float* dout;  // backward only
  float* out;         // forward only
  long B;
  float* gimg;
  float4* act;
  long act_per_wave;
};

__device__ __forceinline__ float sf_act_f(int act, float z) {
  if (act == SF_ACT_SILU) return z * sf_sigmoid(z);
  if (act == SF_ACT_RELU) return fmaxf(z, 0.f);
  return sf_tanh(z);
}
__device__ __forceinline__ float sf_act_d(int act, float z) {
  if (act == SF_ACT_SILU) {
    const float s = sf_sigmoid(z);
    return s * (1.f + z * (1.f - s));
  }
  if (act == SF_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  const float t = sf_tanh(z);
  return 1.f - t * t;
}

// input tile kt of the standardised features
__device__ __forceinline__ void sf_mlp_in_tile(f32x16 (&ct)[1][1], const float* __restrict__ xr, const SfMlpDev& m,
                                               int kt, int h) {
  const int pad = ((m.n_in + 3) / 4) * 4;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rho = kt * 32 + sf_row(r, h);
    const bool ok = rho < m.n_in;
    const int rr = ok ? rho : 0;
    const float v = (xr[rr] - m.cst[rr]) / m.cst[pad + rr];
    ct[0][0][r] = ok ? v : 0.f;
  }
}

// z_0 .. z_{L-1} (pre-activations); z[L-1] is the output.  STASH: write z_l (l < L-1) to the stash.
template <int HT, bool STASH>
__device__ __forceinline__ void sf_mlp_forward(const SfMlpDev& m, const float* __restrict__ xr, f32x16 (&z)[HT][1],
                                               float4* stash, int lane) {
  const int h = lane >> 5;
  sf_init_bias<HT, 1>(z, m.packed + m.o_b[0], h);
  for (int kt = 0; kt * 4 < m.nG[0]; ++kt) {
    f32x16 ct[1][1];
    sf_mlp_in_tile(ct, xr, m, kt, h);
    sf_mm_acc<HT, 1, 1, false, false, true>(z, ct, m.packed + m.o_w[0], m.nG[0], kt * 4, min(4, m.nG[0] - kt * 4), lane);
  }
#pragma unroll
  for (int l = 1; l < SF_MLP_LMAX; ++l) {
    if (l < m.L) {
      f32x16 a[HT][1];
#pragma unroll
      for (int mt = 0; mt < HT; ++mt) {
        if (STASH) sf_stash_store(stash, (l - 1) * HT + mt, z[mt][0], lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) a[mt][0][r] = sf_act_f(m.act, z[mt][0][r]);
      }
      sf_init_bias<HT, 1>(z, m.packed + m.o_b[l], h);
      sf_mm_acc<HT, 1, HT, false, false, true>(z, a, m.packed + m.o_w[l], m.nG[l], 0, m.nG[l], lane);
    }
  }
}

template <int HT>
__global__ __launch_bounds__(256) void k_mlp_fwd(SfMlpDev m, SfMlpArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const long base = ((long)blockIdx.x * 4 + wave) * 32;
  if (base >= a.B) return;
  const long row = base + c;
  const bool valid = row < a.B;
  const float* xr = a.x + (valid ? row : a.B - 1) * m.n_in;
  f32x16 z[HT][1];
  sf_mlp_forward<HT, false>(m, xr, z, nullptr, lane);
  if (valid) {
#pragma unroll
    for (int mt = 0; mt < HT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = mt * 32 + sf_row(r, h);
        if (f < m.n_out) a.out[row * m.n_out + f] = z[mt][0][r];
      }
  }
}

template <int HT>
__global__ __launch_bounds__(256) void k_mlp_bwd(SfMlpDev m, SfMlpArgs a) {
  extern __shared__ float lds_all[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const long wid = (long)blockIdx.x * 4 + wave;
  const long base = wid * 32;
  if (base >= a.B) return;
  float* lds = lds_all + wave * (2 * HT) * SF_TL;
  float4* stash = a.act + wid * a.act_per_wave;
  const long row = base + c;
  const bool valid = row < a.B;
  const long ii = valid ? row : a.B - 1;
  const float* xr = a.x + ii * m.n_in;
  {
    f32x16 z[HT][1];
    sf_mlp_forward<HT, true>(m, xr, z, stash, lane);
  }
  f32x16 d[HT][1];  // dL/d z_l, starting from the output gradient
#pragma unroll
  for (int mt = 0; mt < HT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = mt * 32 + sf_row(r, h);
      d[mt][0][r] = (valid && f < m.n_out) ? a.dout[row * m.n_out + f] : 0.f;
    }
#pragma unroll
  for (int ll = 0; ll < SF_MLP_LMAX - 1; ++ll) {
    const int l = SF_MLP_LMAX - 1 - ll;  // 3, 2, 1
    if (l < m.L) {
      f32x16 zin[HT][1], ain[HT][1];
#pragma unroll
      for (int mt = 0; mt < HT; ++mt) {
        sf_stash_load(stash, (l - 1) * HT + mt, zin[mt][0], lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) ain[mt][0][r] = sf_act_f(m.act, zin[mt][0][r]);
      }
      sf_grad_w_local<HT, HT>(lds, d, ain, a.gimg + m.o_w[l], a.gimg + m.o_b[l], m.nG[l], 0, m.nG[l], lane);
      f32x16 din[HT][1];
#pragma unroll
      for (int mt = 0; mt < HT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) din[mt][0][r] = 0.f;
      sf_mm_acc<HT, 1, HT, false, false, true>(din, d, m.packedT + m.oT_w[l], m.nGo[l], 0, m.nGo[l], lane);
#pragma unroll
      for (int mt = 0; mt < HT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) d[mt][0][r] = din[mt][0][r] * sf_act_d(m.act, zin[mt][0][r]);
    }
  }
  for (int kt = 0; kt * 4 < m.nG[0]; ++kt) {
    f32x16 ct[1][1];
    sf_mlp_in_tile(ct, xr, m, kt, h);
    sf_grad_w_local<HT, 1>(lds, d, ct, a.gimg + m.o_w[0], kt == 0 ? a.gimg + m.o_b[0] : nullptr, m.nG[0], kt * 4,
                     min(4, m.nG[0] - kt * 4), lane);
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct sf_mlp {
  SfMlpLayout L;
  bool dev_ready = false;
  float *d_packed = nullptr, *d_packedT = nullptr, *d_cst = nullptr, *d_gimg = nullptr, *d_act = nullptr;
  int32_t *d_s1 = nullptr, *d_s2 = nullptr, *d_t1 = nullptr, *d_t2 = nullptr, *d_gdst = nullptr;
  size_t act_cap = 0;
  SfMlpDev dev() const {
    SfMlpDev v = L.dev;
    v.packed = d_packed; v.packedT = d_packedT; v.cst = d_cst;
    return v;
  }
};

__global__ void k_mlp_gather(const float* __restrict__ gimg, const int32_t* __restrict__ gdst,
                             float* __restrict__ grad, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) grad[i] = gdst[i] >= 0 ? gimg[gdst[i]] : 0.f;
}

namespace {
#define SF_MTRY(call)                                                              \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess) {                                                        \
      sf_set_error(std::string(#call) + ": " + hipGetErrorString(e_));             \
      return SF_ERR_HIP;                                                           \
    }                                                                              \
  } while (0)

int mlp_ensure(sf_mlp* m) {
  if (m->dev_ready) return SF_OK;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    sf_set_error("no HIP device visible: the embedding kernels have no CPU fallback");
    return SF_ERR_NO_DEVICE;
  }
  const SfMlpLayout& L = m->L;
  auto up = [&](const std::vector<int32_t>& v, int32_t** d) -> hipError_t {
    hipError_t e = hipMalloc(d, std::max<size_t>(v.size(), 1) * sizeof(int32_t));
    if (e != hipSuccess) return e;
    return hipMemcpy(*d, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice);
  };
  SF_MTRY(hipMalloc(&m->d_packed, (size_t)L.n_packed * sizeof(float)));
  SF_MTRY(hipMalloc(&m->d_packedT, std::max<size_t>((size_t)L.n_packedT, 64) * sizeof(float)));
  SF_MTRY(hipMalloc(&m->d_gimg, (size_t)L.n_packed * sizeof(float)));
  SF_MTRY(hipMalloc(&m->d_cst, L.cst.size() * sizeof(float)));
  SF_MTRY(hipMemcpy(m->d_cst, L.cst.data(), L.cst.size() * sizeof(float), hipMemcpyHostToDevice));
  SF_MTRY(up(L.src1, &m->d_s1)); SF_MTRY(up(L.src2, &m->d_s2));
  SF_MTRY(up(L.srcT1, &m->d_t1)); SF_MTRY(up(L.srcT2, &m->d_t2));
  SF_MTRY(up(L.gdst, &m->d_gdst));
  m->dev_ready = true;
  return SF_OK;
}

template <int HT>
hipError_t launch_fwd(const SfMlpDev& d, const SfMlpArgs& a, hipStream_t st) {
  const long grid = ((a.B + 31) / 32 + 3) / 4;
  hipLaunchKernelGGL((k_mlp_fwd<HT>), dim3((unsigned)grid), dim3(256), 0, st, d, a);
  return hipGetLastError();
}
template <int HT>
hipError_t launch_bwd(const SfMlpDev& d, const SfMlpArgs& a, hipStream_t st) {
  const long grid = ((a.B + 31) / 32 + 3) / 4;
  const size_t shmem = (size_t)4 * (2 * HT) * SF_TL * sizeof(float);
  static SfAttrCache attr;
  int attr_dev;
  if (attr.need(attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)k_mlp_bwd<HT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    attr.set(attr_dev);
  }
  hipLaunchKernelGGL((k_mlp_bwd<HT>), dim3((unsigned)grid), dim3(256), shmem, st, d, a);
  return hipGetLastError();
}
}  // namespace

extern "C" {
int sf_mlp_create(const sf_mlp_desc* d, sf_mlp** out) {
  if (!d || !out) { sf_set_error("null argument"); return SF_ERR_INVALID; }
  sf_mlp* m = new sf_mlp();
  if (!sf_build_mlp_layout(*d, m->L)) {
    sf_set_error(m->L.error);
    delete m;
    return SF_ERR_INVALID;
  }
  *out = m;
  return SF_OK;
}
void sf_mlp_destroy(sf_mlp* m) {
  if (!m) return;
  if (m->dev_ready) {
    (void)hipFree(m->d_packed); (void)hipFree(m->d_packedT); (void)hipFree(m->d_cst); (void)hipFree(m->d_gimg);
    (void)hipFree(m->d_act); (void)hipFree(m->d_s1); (void)hipFree(m->d_s2); (void)hipFree(m->d_t1);
    (void)hipFree(m->d_t2); (void)hipFree(m->d_gdst);
  }
  delete m;
}
int64_t sf_mlp_num_params(const sf_mlp* m) { return m ? m->L.n_params : 0; }

int sf_mlp_forward(sf_mlp* m, const float* flat, const float* x, int64_t B, float* out, void* stream) {
  if (!m || !flat) { sf_set_error("null argument"); return SF_ERR_INVALID; }
  if (B == 0) return SF_OK;
  if (!x || !out || B < 0) { sf_set_error("bad argument"); return SF_ERR_INVALID; }
  int rc = mlp_ensure(m);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  SF_MTRY(sf_launch_pack(flat, m->d_s1, m->d_s2, m->d_packed, (long)m->L.n_packed, st));
  SfMlpArgs a{};
  a.x = x; a.out = out; a.B = B;
  const SfMlpDev d = m->dev();
  switch (d.HT) {
    case 1: SF_MTRY(launch_fwd<1>(d, a, st)); break;
    case 2: SF_MTRY(launch_fwd<2>(d, a, st)); break;
    case 3: SF_MTRY(launch_fwd<3>(d, a, st)); break;
    default: SF_MTRY(launch_fwd<4>(d, a, st)); break;
  }
  return SF_OK;
}

int sf_mlp_backward(sf_mlp* m, const float* flat, const float* x, const float* dout, int64_t B, float* grad,
                    void* stream) {
  if (!m || !flat || !grad) { sf_set_error("null argument"); return SF_ERR_INVALID; }
  if (B > 0 && (!x || !dout)) { sf_set_error("null argument"); return SF_ERR_INVALID; }
  int rc = mlp_ensure(m);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const SfMlpLayout& L = m->L;
  const long waves = (B + 31) / 32;
  const long act_per_wave = (long)std::max(1, (L.dev.L - 1) * L.dev.HT) * 4 * 64;
  const size_t need = (size_t)std::max<long>(waves, 1) * act_per_wave * 4;
  if (need > m->act_cap) {
    if (m->d_act) SF_MTRY(hipFree(m->d_act));
    m->d_act = nullptr; m->act_cap = 0;
    SF_MTRY(hipMalloc(&m->d_act, need * sizeof(float)));
    m->act_cap = need;
  }
  SF_MTRY(sf_launch_pack(flat, m->d_s1, m->d_s2, m->d_packed, (long)L.n_packed, st));
  if (L.n_packedT > 0) SF_MTRY(sf_launch_pack(flat, m->d_t1, m->d_t2, m->d_packedT, (long)L.n_packedT, st));
  SF_MTRY(hipMemsetAsync(m->d_gimg, 0, (size_t)L.n_packed * sizeof(float), st));
  if (B > 0) {
    SfMlpArgs a{};
    a.x = x; a.dout = dout; a.B = B; a.gimg = m->d_gimg;
    a.act = reinterpret_cast<float4*>(m->d_act); a.act_per_wave = act_per_wave;
    const SfMlpDev d = m->dev();
    switch (d.HT) {
      case 1: SF_MTRY(launch_bwd<1>(d, a, st)); break;
      case 2: SF_MTRY(launch_bwd<2>(d, a, st)); break;
      case 3: SF_MTRY(launch_bwd<3>(d, a, st)); break;
      default: SF_MTRY(launch_bwd<4>(d, a, st)); break;
    }
  }
  hipLaunchKernelGGL(k_mlp_gather, dim3((unsigned)((L.n_params + 255) / 256)), dim3(256), 0, st, m->d_gimg, m->d_gdst,
                     grad, (long)L.n_params);
  SF_MTRY(hipGetLastError());
  return SF_OK;
}
}  // extern "C"
