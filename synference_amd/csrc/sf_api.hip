// sf_api.hip -- the C ABI (include/synference_hip.h): handle management and call sequencing.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sf_internal.h"
#include "sf_train.h"
#include "sf_nsf1.h"
#include "sf_nsfar.h"
#include "sf_nsfc.h"
#include "sf_trainc.h"

namespace {
thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
int hip_fail(hipError_t e, const char* what) {
  return fail(SF_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define SF_HIP(call)                                  \
  do {                                                \
    hipError_t e_ = (call);                           \
    if (e_ != hipSuccess) return hip_fail(e_, #call); \
  } while (0)
}  // namespace

static int ensure_device(sf_flow* f) {
  if (f->nsf1 || f->nsfar) {
    if (f->dev_ready) return SF_OK;
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0) return fail(SF_ERR_NO_DEVICE, "no HIP device visible");
    hipError_t e = hipMalloc(&f->d_flat, (size_t)f->L.n_params * sizeof(float));
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(d_flat)");
    f->dev_ready = true;
    return SF_OK;
  }
  if (f->dev_ready) return SF_OK;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
    return fail(SF_ERR_NO_DEVICE, "no HIP device visible: the gfx950 flow engine has no CPU fallback");
  const size_t np = (size_t)f->L.n_packed;
  SF_HIP(hipMalloc(&f->d_packed, np * sizeof(float)));
  SF_HIP(hipMemset(f->d_packed, 0, np * sizeof(float)));
  SF_HIP(hipMalloc(&f->d_cst, f->L.cst.size() * sizeof(float)));
  SF_HIP(hipMemcpy(f->d_cst, f->L.cst.data(), f->L.cst.size() * sizeof(float), hipMemcpyHostToDevice));
  SF_HIP(hipMalloc(&f->d_s1, np * sizeof(int32_t)));
  SF_HIP(hipMalloc(&f->d_s2, np * sizeof(int32_t)));
  SF_HIP(hipMemcpy(f->d_s1, f->L.src1.data(), np * sizeof(int32_t), hipMemcpyHostToDevice));
  SF_HIP(hipMemcpy(f->d_s2, f->L.src2.data(), np * sizeof(int32_t), hipMemcpyHostToDevice));
  if ((f->L.dev.m16_ok || f->L.nsfS.ok) && f->L.n_packed16 > 0) {
    const size_t n16 = (size_t)f->L.n_packed16;
    SF_HIP(hipMalloc(&f->d_packed16, n16 * sizeof(float)));
    SF_HIP(hipMalloc(&f->d_s16a, n16 * sizeof(int32_t)));
    SF_HIP(hipMalloc(&f->d_s16b, n16 * sizeof(int32_t)));
    SF_HIP(hipMemcpy(f->d_s16a, f->L.src16a.data(), n16 * sizeof(int32_t), hipMemcpyHostToDevice));
    SF_HIP(hipMemcpy(f->d_s16b, f->L.src16b.data(), n16 * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  if ((f->L.dev.m16_ok || f->L.nsfS.ok) && f->L.n_packed16B > 0) {
    const size_t nB = (size_t)f->L.n_packed16B;
    SF_HIP(hipMalloc(&f->d_packed16B, nB * sizeof(unsigned short)));
    SF_HIP(hipMalloc(&f->d_s16B, nB * sizeof(int32_t)));
    SF_HIP(hipMemcpy(f->d_s16B, f->L.src16B.data(), nB * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  if (f->L.n_packedB > 0) {
    SF_HIP(hipMalloc(&f->d_packedB, (size_t)f->L.n_packedB * sizeof(unsigned short)));
    SF_HIP(hipMalloc(&f->d_bsrc, (size_t)f->L.n_packedB * sizeof(int32_t)));
    SF_HIP(hipMemcpy(f->d_bsrc, f->L.srcB.data(), (size_t)f->L.n_packedB * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  SF_HIP(hipMalloc(&f->d_flat, (size_t)f->L.n_params * sizeof(float)));
  SF_HIP(hipMalloc(&f->d_cnt, SF_MAX_ROUNDS * sizeof(uint32_t)));
  SF_HIP(hipHostMalloc((void**)&f->h_cnt, SF_MAX_ROUNDS * sizeof(uint32_t), hipHostMallocDefault));
  SF_HIP(hipEventCreate(&f->ev_dense[0]));
  SF_HIP(hipEventCreate(&f->ev_dense[1]));
  f->dev_ready = true;
  return SF_OK;
}

extern "C" {

const char* sf_last_error(void) { return g_err.c_str(); }
const char* sf_version(void) { return "synference_hip 0.2 (gfx950; mfma_f32_32x32x2 / 16x16x4)"; }
int sf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int sf_flow_create(const sf_flow_desc* desc, sf_flow** out) {
  if (!desc || !out) return fail(SF_ERR_INVALID, "null argument");
  if (desc->kind == SF_NSF && desc->D == 1) {  // sbi's ContextSplineMap flow (sf_nsf1.hip)
    sf_flow* f1 = new sf_flow();
    std::string err;
    int rc = sf_nsf1_create(*desc, &f1->nsf1, err);
    if (rc) { delete f1; return fail(rc, err); }
    std::memset(&f1->L.dev, 0, sizeof(f1->L.dev));
    std::memset(&f1->L.trc, 0, sizeof(f1->L.trc));
    std::memset(&f1->L.nsc, 0, sizeof(f1->L.nsc));
    std::memset(&f1->L.nsfS, 0, sizeof(f1->L.nsfS));
    SfDev& v = f1->L.dev;
    v.kind = SF_NSF; v.D = 1; v.C = desc->C; v.H = desc->H; v.T = desc->T; v.K = desc->K; v.NB = desc->NB;
    f1->L.n_params = (int64_t)desc->T * f1->nsf1->P_mlp;
    *out = f1;
    return SF_OK;
  }
  if (desc->kind == SF_NSF_AR || desc->kind == SF_MAF_AR) {  // the autoregressive flows of the lampe / zuko backend (sf_nsfar.hip)
    sf_flow* fa = new sf_flow();
    std::string err;
    int rc = sf_nsfar_create(*desc, &fa->nsfar, err);
    if (rc) { delete fa; return fail(rc, err); }
    std::memset(&fa->L.dev, 0, sizeof(fa->L.dev));
    std::memset(&fa->L.trc, 0, sizeof(fa->L.trc));
    std::memset(&fa->L.nsc, 0, sizeof(fa->L.nsc));
    std::memset(&fa->L.nsfS, 0, sizeof(fa->L.nsfS));
    SfDev& v = fa->L.dev;
    v.kind = desc->kind; v.D = desc->D; v.C = desc->C; v.H = desc->H; v.T = desc->T; v.K = desc->K; v.NB = desc->NB;
    fa->L.n_params = fa->nsfar->n_params;
    *out = fa;
    return SF_OK;
  }
  sf_flow* f = new sf_flow();
  if (!sf_build_layout(*desc, f->L)) {
    std::string e = f->L.error;
    delete f;
    return fail(SF_ERR_INVALID, e);
  }
  *out = f;
  return SF_OK;
}

void sf_flow_destroy(sf_flow* f) {
  if (!f) return;
  if (f->nsf1 || f->nsfar) {
    sf_nsf1_destroy(f->nsf1);
    sf_nsfar_destroy(f->nsfar);
    if (f->dev_ready) (void)hipFree(f->d_flat);
    (void)hipFree(f->d_losspart_mem);   // (sf_flow_train_epoch allocates it for every kind)
    if (f->ev_train[0]) (void)hipEventDestroy(f->ev_train[0]);
    if (f->ev_train[1]) (void)hipEventDestroy(f->ev_train[1]);
    if (f->ev_dense[0]) (void)hipEventDestroy(f->ev_dense[0]);
    if (f->ev_dense[1]) (void)hipEventDestroy(f->ev_dense[1]);
    delete f;
    return;
  }
  if (f->dev_ready) {
    (void)hipFree(f->d_packed); (void)hipFree(f->d_packedT); (void)hipFree(f->d_cst); (void)hipFree(f->d_packedB); (void)hipFree(f->d_bsrc);
    (void)hipHostFree(f->h_cnt); if (f->ev_train[0]) (void)hipEventDestroy(f->ev_train[0]); if (f->ev_train[1]) (void)hipEventDestroy(f->ev_train[1]); if (f->ev_dense[0]) (void)hipEventDestroy(f->ev_dense[0]); if (f->ev_dense[1]) (void)hipEventDestroy(f->ev_dense[1]);
    (void)hipFree(f->d_ctab); (void)hipFree(f->d_packed16B); (void)hipFree(f->d_s16B); (void)hipFree(f->d_packed16); (void)hipFree(f->d_s16a); (void)hipFree(f->d_s16b);
    (void)hipFree(f->d_s1); (void)hipFree(f->d_s2); (void)hipFree(f->d_t1); (void)hipFree(f->d_t2);
    (void)hipFree(f->d_flat); (void)hipFree(f->d_gpacked); (void)hipFree(f->d_gdst);
    (void)hipFree(f->d_imgC); (void)hipFree(f->d_sC1); (void)hipFree(f->d_sC2); (void)hipFree(f->d_gdstC); (void)hipFree(f->d_gsrcC); (void)hipFree(f->d_gzeroC); (void)hipFree(f->d_gpartC); (void)hipFree(f->d_gfixC); (void)hipFree(f->d_ustash);
    (void)hipFree(f->d_queue); (void)hipFree(f->d_ring); (void)hipFree(f->d_galacc); (void)hipFree(f->d_sqpart); (void)hipFree(f->d_losspart_mem); (void)hipFree(f->d_best); (void)hipHostFree(f->h_queue);
    (void)hipFree(f->d_act); (void)hipFree(f->d_rej[0]); (void)hipFree(f->d_rej[1]); (void)hipFree(f->d_cnt);
    if (f->step_exec) (void)hipGraphExecDestroy(f->step_exec);
    if (f->step_graph) (void)hipGraphDestroy(f->step_graph);
    if (f->step_stream) (void)hipStreamDestroy(f->step_stream);
    if (f->step_ev[0]) (void)hipEventDestroy(f->step_ev[0]);
    if (f->step_ev[1]) (void)hipEventDestroy(f->step_ev[1]);
    (void)hipFree(f->d_step_ctr); (void)hipFree(f->d_step_bc); (void)hipFree(f->d_step_rows);
  }
  delete f;
}

int64_t sf_flow_num_params(const sf_flow* f) { return f ? f->L.n_params : 0; }
int64_t sf_flow_packed_size(const sf_flow* f) { return !f ? 0 : (f->nsfar ? (int64_t)f->nsfar->src.size() : f->L.n_packed); }

int sf_flow_pack_table(const sf_flow* f, int32_t* src1, int32_t* src2, int64_t n_packed) {
  if (!f || !src1 || !src2) return fail(SF_ERR_INVALID, "null argument");
  if (f->nsfar) {   // one gather table (sf_nsfar.hip); the second is "none" everywhere
    if (n_packed != (int64_t)f->nsfar->src.size()) return fail(SF_ERR_INVALID, "n_packed mismatch");
    std::memcpy(src1, f->nsfar->src.data(), (size_t)n_packed * sizeof(int32_t));
    for (int64_t i = 0; i < n_packed; ++i) src2[i] = -1;
    return SF_OK;
  }
  if (n_packed != f->L.n_packed) return fail(SF_ERR_INVALID, "n_packed mismatch");
  if (n_packed == 0) return SF_OK;   // (empty tables have no storage to copy from)
  std::memcpy(src1, f->L.src1.data(), (size_t)n_packed * sizeof(int32_t));
  std::memcpy(src2, f->L.src2.data(), (size_t)n_packed * sizeof(int32_t));
  return SF_OK;
}

int64_t sf_flow_packed16_size(const sf_flow* f) { return (f && f->L.dev.m16_ok) ? f->L.n_packed16 : 0; }

int sf_flow_pack_table16(const sf_flow* f, int32_t* src1, int32_t* src2, int64_t n_packed16) {
  if (!f || !src1 || !src2) return fail(SF_ERR_INVALID, "null argument");
  if (!f->L.dev.m16_ok || n_packed16 != f->L.n_packed16) return fail(SF_ERR_INVALID, "no 16-row image or size mismatch");
  std::memcpy(src1, f->L.src16a.data(), (size_t)n_packed16 * sizeof(int32_t));
  std::memcpy(src2, f->L.src16b.data(), (size_t)n_packed16 * sizeof(int32_t));
  return SF_OK;
}

int64_t sf_flow_packed16b_size(const sf_flow* f) { return (f && f->L.dev.m16_ok) ? f->L.n_packed16B : 0; }

int sf_flow_pack_table16b(const sf_flow* f, int32_t* src, int64_t n) {
  if (!f || !src) return fail(SF_ERR_INVALID, "null argument");
  if (!f->L.dev.m16_ok || n != f->L.n_packed16B) return fail(SF_ERR_INVALID, "no split-bf16 image or size mismatch");
  std::memcpy(src, f->L.src16B.data(), (size_t)n * sizeof(int32_t));
  return SF_OK;
}

int64_t sf_flow_trainc_size(const sf_flow* f) { return (f && (f->L.trc.ok || f->L.nsc.ok)) ? f->L.n_imgC : 0; }
int64_t sf_flow_trainc_grad_size(const sf_flow* f) { return (f && (f->L.trc.ok || f->L.nsc.ok)) ? f->L.n_gradC : 0; }
int64_t sf_flow_cst_size(const sf_flow* f) { return f ? (int64_t)f->L.cst.size() : 0; }
int sf_flow_trainc_table(const sf_flow* f, int32_t* src1, int32_t* src2, int64_t n, int32_t* gdst, int64_t n_params,
                         int32_t* desc, float* cst, int64_t n_cst) {
  if (!f || !src1 || !src2 || !gdst || !desc || !cst) return fail(SF_ERR_INVALID, "null argument");
  if ((!f->L.trc.ok && !f->L.nsc.ok) || n != f->L.n_imgC || n_params != f->L.n_params || n_cst != (int64_t)f->L.cst.size())
    return fail(SF_ERR_INVALID, "no cooperative training image or size mismatch");
  static_assert(sizeof(SfTrcDev) <= 64 * sizeof(int32_t) && sizeof(SfNscDev) <= 64 * sizeof(int32_t), "descriptor words");
  std::memcpy(src1, f->L.srcC1.data(), (size_t)n * sizeof(int32_t));
  std::memcpy(src2, f->L.srcC2.data(), (size_t)n * sizeof(int32_t));
  std::memcpy(gdst, f->L.gdstC.data(), (size_t)n_params * sizeof(int32_t));
  std::memset(desc, 0, 64 * sizeof(int32_t));
  if (f->L.trc.ok) std::memcpy(desc, &f->L.trc, sizeof(SfTrcDev));   // MAF: SfTrcDev, NSF: SfNscDev (sf_layout.h)
  else std::memcpy(desc, &f->L.nsc, sizeof(SfNscDev));
  std::memcpy(cst, f->L.cst.data(), (size_t)n_cst * sizeof(float));
  return SF_OK;
}

int sf_flow_describe(const sf_flow* f, char* buf, size_t buflen) {
  if (!f || !buf) return fail(SF_ERR_INVALID, "null argument");
  const SfDev& v = f->L.dev;
  std::string s = "{";
  auto add = [&](const char* k, long val) { s += "\"" + std::string(k) + "\": " + std::to_string(val) + ", "; };
  if (f->nsfar) {   // the autoregressive NSF has its own images (sf_nsfar.h)
    const SfNsfAr& n = *f->nsfar;
    add("kind", n.affine ? SF_MAF_AR : SF_NSF_AR); add("D", n.D); add("C", n.C); add("H", n.H); add("T", n.T); add("K", n.K); add("NB", 2);
    add("Hp", n.Hp); add("t_stride", n.t_stride); add("n_params", (long)n.n_params); add("n_packed", (long)n.src.size());
    add("o_L0t", n.o_L0t); add("o_b0", n.o_b0); add("o_L1t", n.o_L1t); add("o_L1m", n.o_L1m); add("o_b1", n.o_b1);
    add("o_L2t", n.o_L2t); add("o_b2", n.o_b2); add("o_L0m", n.o_L0m); add("o_L2m", n.o_L2m); add("lds_bytes_train", (long)sf_nsfar_lds_bytes(n, 3));
    add("sampler_tiles16", sf_nsfar16_eligible(n) ? 1 : 0); add("s16_nt", n.s16_nt); add("s16_ni", n.s16_ni); add("s16_ks", n.s16_ks); add("s16_tpt", n.s16_tpt);
    add("o_F0", n.o_F0); add("o_fb0", n.o_fb0); add("o_F1", n.o_F1); add("o_fb1", n.o_fb1); add("o_F2", n.o_F2);
    auto arr = [&](const char* k, const std::vector<int32_t>& a, bool last) {
      s += "\"" + std::string(k) + "\": [";
      for (size_t i = 0; i < a.size(); ++i) s += std::to_string(a[i]) + (i + 1 < a.size() ? ", " : "");
      s += last ? "]}" : "], ";
    };
    arr("perm", n.perm, false); arr("ptype", n.ptype, false); arr("tend", n.tend, false); arr("ord", n.ord, false); arr("dimof", n.dimof, true);
    if (s.size() + 1 > buflen) return fail(SF_ERR_INVALID, "buffer too small");
    std::memcpy(buf, s.c_str(), s.size() + 1);
    return SF_OK;
  }
  add("kind", v.kind); add("D", v.D); add("C", v.C); add("H", v.H); add("T", v.T); add("K", v.K);
  add("NB", v.NB); add("HT", v.HT); add("PT", v.PT); add("KMAX", v.KMAX); add("JP", v.JP);
  add("nGu", v.nGu); add("nGc", v.nGc); add("nGh", v.nGh); add("t_stride", v.t_stride);
  add("o_w0", v.o_w0); add("o_wc", v.o_wc); add("o_b0", v.o_b0); add("o_wk0", v.o_wk[0]);
  add("o_bk0", v.o_bk[0]); add("o_wk1", v.o_wk[1]); add("o_bk1", v.o_bk[1]); add("o_wf", v.o_wf);
  add("o_wk2", v.o_wk[2]); add("o_bk2", v.o_bk[2]); add("o_wk3", v.o_wk[3]); add("o_bk3", v.o_bk[3]);
  add("scale_fn", v.scale_fn);
  add("o_bf", v.o_bf); add("o_winu", v.o_winu); add("o_winc", v.o_winc); add("o_bin", v.o_bin);
  add("o_wg0", v.o_wg[0]); add("o_bg0", v.o_bg[0]); add("o_w10", v.o_w1[0]); add("o_b10", v.o_b1[0]);
  add("o_w20", v.o_w2[0]); add("o_b20", v.o_b2[0]); add("o_wg1", v.o_wg[1]); add("o_bg1", v.o_bg[1]);
  add("o_w11", v.o_w1[1]); add("o_b11", v.o_b1[1]); add("o_w21", v.o_w2[1]); add("o_b21", v.o_b2[1]);
  add("o_wout", v.o_wout); add("o_bout", v.o_bout); add("o_lu", v.o_lu);
  add("c_pscale", v.c_pscale); add("c_pshift", v.c_pshift); add("c_tdim", v.c_tdim);
  add("c_xmean", v.c_xmean); add("c_xstd", v.c_xstd); add("c_dslot", v.c_dslot);
  add("nsf_split_sampler", f->L.nsfS.ok); add("trainc_ok", f->L.trc.ok); add("nsfc_ok", f->L.nsc.ok);
  add("inc_ok", v.inc_ok); add("hidden_bf16", v.hidden_bf16); add("tB_stride", v.tB_stride); add("n_parts", v.n_parts); add("part_max", v.part_max);
  add("n_params", f->L.n_params); add("n_packed", f->L.n_packed);
  {
    int R = 0, NV = 0;
    SfDev vv = v;
    vv.packed16 = reinterpret_cast<const float*>(1);  // shape query only: as if the 16-row image were fresh
    sf_ctab_shape(vv, R, NV);
    add("ctab_floats_per_galaxy", (long)v.T * NV * R);
  }
  add("o16_w0", v.o16_w0); add("o16_wc", v.o16_wc); add("o16_b0", v.o16_b0); add("o16_wk0", v.o16_wk[0]);
  add("o16_wk1", v.o16_wk[1]); add("o16_bk0", v.o16_bk[0]); add("o16_bk1", v.o16_bk[1]); add("o16_hv", v.o16_hv);
  add("o16_hvb", v.o16_hvb); add("o16_wh", v.o16_wh); add("o16_bh", v.o16_bh); add("o16_wp", v.o16_wp); add("t16_a_tab", v.t16_a_tab);
  add("t16_a", v.t16_a); add("t16B_stride", v.t16B_stride); add("nP16", v.nP16); add("o16B_wk0", v.o16B_wk[0]); add("o16B_wk1", v.o16B_wk[1]);
  add("m16_ok", v.m16_ok); add("m16_span", v.m16_span); add("nT16", v.nT16); add("nC16", v.nC16); add("t16_stride", v.t16_stride);
  s += "\"g16_tile\": [";
  for (int i = 0; i < SF_DMAX; ++i) s += std::to_string(v.g16_tile[i]) + (i + 1 < SF_DMAX ? ", " : "], ");
  s += "\"g16_lo\": [";
  for (int i = 0; i < SF_DMAX; ++i) s += std::to_string(v.g16_lo[i]) + (i + 1 < SF_DMAX ? ", " : "], ");
  s += "\"g_kend\": [";
  for (int i = 0; i < SF_DMAX; ++i) s += std::to_string(v.g_kend[i]) + (i + 1 < SF_DMAX ? ", " : "], ");
  s += "\"g_lo\": [";
  for (int i = 0; i < SF_DMAX; ++i) s += std::to_string(v.g_lo[i]) + (i + 1 < SF_DMAX ? ", " : "], ");
  s += "\"g_tile\": [";
  for (int i = 0; i < SF_DMAX; ++i) s += std::to_string(v.g_tile[i]) + (i + 1 < SF_DMAX ? ", " : "], ");
  s += "\"mt_kend\": [";
  for (int i = 0; i < 4; ++i) s += std::to_string(v.mt_kend[i]) + (i + 1 < 4 ? ", " : "], ");
  s += "\"cst\": [";
  for (size_t i = 0; i < f->L.cst.size(); ++i) {
    char t[40];
    std::snprintf(t, sizeof(t), "%.9g%s", f->L.cst[i], i + 1 < f->L.cst.size() ? ", " : "");
    s += t;
  }
  s += "]}";
  if (s.size() + 1 > buflen) return fail(SF_ERR_INVALID, "buffer too small");
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return SF_OK;
}

int sf_flow_set_params(sf_flow* f, const float* flat, int64_t n, int is_device, void* stream) {
  if (!f || !flat) return fail(SF_ERR_INVALID, "null argument");
  if (n != f->L.n_params) return fail(SF_ERR_INVALID, "parameter count mismatch");
  int rc = ensure_device(f);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  // the handle keeps its own copy of the logical vector (sf_flow_get_params)
  SF_HIP(hipMemcpyAsync(f->d_flat, flat, (size_t)n * sizeof(float), is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
  const float* src = f->d_flat;
  f->flat_valid = true;
  if (f->nsf1) { f->params_set = true; return SF_OK; }   // (the MLP engine re-tiles per call)
  if (f->nsfar) {
    std::string err;
    rc = sf_nsfar_pack(f->nsfar, src, st, err);
    if (rc) return fail(rc, err);
    f->params_set = true;
    return SF_OK;
  }
  SF_HIP(sf_launch_pack(src, f->d_s1, f->d_s2, f->d_packed, (long)f->L.n_packed, st));
  if (f->d_packed16) SF_HIP(sf_launch_pack(src, f->d_s16a, f->d_s16b, f->d_packed16, (long)f->L.n_packed16, st));
  f->wp_stale = true;
  if (f->d_packed16B) SF_HIP(sf_launch_pack_bf16_split(src, f->d_s16B, f->d_packed16B, (long)f->L.n_packed16B, st));
  f->packed16_stale = false;
  if (f->L.n_packedB > 0) SF_HIP(sf_launch_pack_bf16(src, f->d_bsrc, f->d_packedB, (long)f->L.n_packedB, st));
  f->params_set = true;
  f->ctab_x = nullptr;  // a context table built from the old parameters is stale
  return SF_OK;
}

int sf_flow_get_params(sf_flow* f, float* flat, int64_t n, int is_device, void* stream) {
  if (!f || !flat) return fail(SF_ERR_INVALID, "null argument");
  if (n != f->L.n_params) return fail(SF_ERR_INVALID, "parameter count mismatch");
  if (!f->params_set || !f->flat_valid)
    return fail(SF_ERR_STATE, "no parameters held by the handle: after sf_flow_loss_grad the caller's vector is the master copy "
                              "(call sf_flow_set_params first)");
  hipStream_t st = (hipStream_t)stream;
  SF_HIP(hipMemcpyAsync(flat, f->d_flat, (size_t)n * sizeof(float), is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
  if (!is_device) SF_HIP(hipStreamSynchronize(st));
  return SF_OK;
}

int sf_flow_log_prob(sf_flow* f, const float* theta, const float* x, int64_t B, float* out, void* stream) {
  if (!f) return fail(SF_ERR_INVALID, "null handle");
  if (B == 0) return SF_OK;
  if (!theta || !x || !out) return fail(SF_ERR_INVALID, "null argument");
  if (!f->params_set) return fail(SF_ERR_STATE, "sf_flow_set_params has not been called");
  if (B < 0) return fail(SF_ERR_INVALID, "B < 0");
  if (f->nsf1) {
    std::string err;
    int rc = sf_nsf1_log_prob(f->nsf1, f->d_flat, theta, x, (long)B, out, (hipStream_t)stream, err);
    return rc ? fail(rc, err) : SF_OK;
  }
  if (f->nsfar) {
    std::string err;
    int rc = sf_nsfar_log_prob(f->nsfar, theta, x, (long)B, out, (hipStream_t)stream, err);
    return rc ? fail(rc, err) : SF_OK;
  }
  SF_HIP(sf_launch_logprob(f->dev(), theta, x, (long)B, out, (hipStream_t)stream));
  return SF_OK;
}

int sf_flow_inverse_from_noise(sf_flow* f, const float* z, const float* x, int64_t B, float* theta,
                               float* logdet, void* stream) {
  if (!f) return fail(SF_ERR_INVALID, "null handle");
  if (B == 0) return SF_OK;
  if (!z || !x || !theta) return fail(SF_ERR_INVALID, "null argument");
  if (!f->params_set) return fail(SF_ERR_STATE, "sf_flow_set_params has not been called");
  if (f->nsf1) {
    std::string err;
    int rc = sf_nsf1_inverse(f->nsf1, f->d_flat, z, x, (long)B, theta, logdet, (hipStream_t)stream, err);
    return rc ? fail(rc, err) : SF_OK;
  }
  if (f->nsfar) {
    std::string err;
    int rc = sf_nsfar_inverse(f->nsfar, z, x, (long)B, theta, logdet, (hipStream_t)stream, err);
    return rc ? fail(rc, err) : SF_OK;
  }
  SfSampleArgsHost a;
  a.x = x; a.z_in = z; a.n_items = (long)B; a.out = theta; a.logdet_out = logdet;
  SF_HIP(sf_launch_inverse(f->dev(), a, (hipStream_t)stream));
  return SF_OK;
}

static void nsf_sampler_view(const sf_flow* f, SfDev& m);
static SfDev sampler_dev(const sf_flow* f, const float* x);
int sf_flow_inverse_from_noise_sampler(sf_flow* f, const float* z, const float* x, int64_t B, float* theta, void* stream) {
  if (!f) return fail(SF_ERR_INVALID, "null handle");
  if (B == 0) return SF_OK;
  if (!z || !x || !theta) return fail(SF_ERR_INVALID, "null argument");
  if (!f->params_set) return fail(SF_ERR_STATE, "sf_flow_set_params has not been called");
  if (f->nsf1 || f->nsfar) {  // one fp32 path
    int rc = sf_flow_inverse_from_noise(f, z, x, B, theta, nullptr, stream);
    return rc ? rc : 1;
  }
  if (f->L.dev.kind == SF_NSF) {  // NSF: the sampling kernels themselves, on the sampler image when the flow has one
    SfDev ms = f->dev();
    nsf_sampler_view(f, ms);
    SfSampleArgsHost a;
    a.x = x; a.z_in = z; a.n_items = (long)B; a.out = theta;
    SF_HIP(sf_launch_inverse(ms, a, (hipStream_t)stream));
    return ms.hidden_bf16 == 2 ? SF_OK : 1;
  }
  const SfDev m = f->dev();
  if (!sf_maf16b_available(m)) {  // the sampler of this flow is the 32-row fp32 path
    SfSampleArgsHost a;
    a.x = x; a.z_in = z; a.n_items = (long)B; a.out = theta;
    SF_HIP(sf_launch_inverse(m, a, (hipStream_t)stream));
    return 1;  // (positive: "fp32 path", not an error)
  }
  if (sf_sampler_fp32_for(SF_MAF)) {
    // the default sampler of a flow with the unrolled kernels runs the FUSED first layer off the context table: the hook builds
    // the table for these rows and runs the find kernel's pass functions on the given noise
    int rc = sf_flow_prepare_context(f, x, B, stream);
    if (rc) return rc;
    const SfDev mt = sampler_dev(f, x);
    if (sf_maf16_fused_d(mt) > 0) {
      SfSampleArgsHost a;
      a.x = x; a.z_in = z; a.n_items = (long)B; a.out = theta; a.S = 1;
      hipError_t e = sf_launch_maf_find16_zin(mt, a, (hipStream_t)stream);
      (void)sf_flow_release_context(f);
      if (e != hipSuccess) return hip_fail(e, "sf_launch_maf_find16_zin");
      return 3;   // 3: the 16-row sampler's fp32 pass functions with the fused first layer
    }
    (void)sf_flow_release_context(f);
  }
  SF_HIP(sf_launch_maf_inv16b_hook(m, z, x, (long)B, theta, (hipStream_t)stream));
  return sf_sampler_fp32_for(SF_MAF) ? 2 : SF_OK;   // 2: the 16-row sampler's fp32 pass functions
}

int sf_set_sampler_fp32(int on) {
  sf_sampler_fp32_set(on);
  return SF_OK;
}

int sf_flow_train_path(const sf_flow* f, int64_t B, int want_dctx) {
  if (!f) return fail(SF_ERR_INVALID, "null handle");
  if (f->nsf1) return 4;   // MLP engine + scalar spline chain (sf_nsf1.hip)
  if (f->nsfar) return 5;  // thread-per-sample masked hyper-network (sf_nsfar.hip)
  if (B > 0 && sf_trainc_eligible(f->L, want_dctx != 0)) return sf_trainc_groups((long)B, &f->L.trc, f->L.dev.T);
  if (B > 0 && sf_nsfc_eligible(f->L, want_dctx != 0)) return 3;
  return 0;
}

static void seed_keys(uint64_t seed, uint32_t stream_id, uint32_t& k0, uint32_t& k1) {
  k0 = (uint32_t)(seed & 0xffffffffu);
  k1 = (uint32_t)(seed >> 32) ^ stream_id;
}

// ---- per-galaxy context table ---------------------------------------------------------------
static size_t ctab_limit_bytes() {
  static long v = -1;
  if (v < 0) {
    const char* e = std::getenv("SF_CTAB_MAX_MB");  // 0 disables the table
    v = e ? std::atol(e) : 4096;
  }
  return (size_t)v << 20;
}

int sf_flow_prepare_context(sf_flow* f, const float* x, int64_t M, void* stream) {
  if (!f) return fail(SF_ERR_INVALID, "null handle");
  f->ctab_x = nullptr;
  f->ctab_M = 0;
  if (M == 0) return SF_OK;
  if (!x) return fail(SF_ERR_INVALID, "null argument");
  if (M < 0) return fail(SF_ERR_INVALID, "M < 0");
  if (!f->params_set) return fail(SF_ERR_STATE, "sf_flow_set_params has not been called");
  if (f->nsf1 || f->nsfar) return SF_OK;   // (no per-row table: nsf1 evaluates the conditioner once per row anyway, nsf_ar's depends on theta)
  SfDev m = f->dev();
  int R = 0, NV = 0;
  sf_ctab_shape(m, R, NV);
  const size_t need = (size_t)M * m.T * NV * R;
  if (need == 0 || need * sizeof(float) > ctab_limit_bytes()) return SF_OK;  // no table: kernels evaluate the context per draw
  if (f->ctab_cap < need) {
    (void)hipFree(f->d_ctab);
    f->d_ctab = nullptr;
    f->ctab_cap = 0;
    SF_HIP(hipMalloc(&f->d_ctab, need * sizeof(float)));
    f->ctab_cap = need;
  }
  m.ctab_R = R; m.ctab_NV = NV;
  if (f->wp_stale && m.kind == SF_MAF && m.m16_ok && m.packed16 && m.o16_wp >= 0) {   // the fused first layer follows the parameters
    SF_HIP(sf_launch_maf_fuse16(m, (hipStream_t)stream));
    f->wp_stale = false;
  }
  SF_HIP(sf_launch_ctab(m, x, (long)M, f->d_ctab, (hipStream_t)stream));
  f->ctab_x = x;
  f->ctab_M = M;
  return SF_OK;
}

int sf_flow_release_context(sf_flow* f) {
  if (!f) return fail(SF_ERR_INVALID, "null handle");
  f->ctab_x = nullptr;
  f->ctab_M = 0;
  return SF_OK;
}

// device view for a sampling launch over context rows x: attaches the table when it was prepared for x
// NSF: the sampling kernels run on the sampler image (fp32 blocks without W1 / W2 + their split-bf16 form, SfNsfSamp):
// the descriptor handed to them has its image pointer, stride and block offsets replaced; the kernels themselves only see
// hidden_bf16 == 2.  sf_set_sampler_fp32(1) / SF_SAMPLER_FP32=1 keeps the all-fp32 image.
static void nsf_sampler_view(const sf_flow* f, SfDev& m) {
  const SfNsfSamp& s = f->L.nsfS;
  if (m.kind != SF_NSF || !s.ok || m.hidden_bf16 || !f->d_packed16 || !f->d_packed16B || f->packed16_stale || sf_sampler_fp32_for(SF_NSF)) return;
  m.packed = f->d_packed16;
  m.t_stride = s.t_stride;
  m.o_winu = s.o_winu; m.o_winc = s.o_winc; m.o_bin = s.o_bin; m.o_wout = s.o_wout; m.o_bout = s.o_bout; m.o_lu = s.o_lu;
  for (int k = 0; k < SF_NBMAX; ++k) {
    m.o_wg[k] = s.o_wg[k]; m.o_bg[k] = s.o_bg[k]; m.o_b1[k] = s.o_b1[k]; m.o_b2[k] = s.o_b2[k];
    m.o_w1[k] = 0; m.o_w2[k] = 0;  // (not in this image)
    m.oB_w1[k] = s.oB_w1[k]; m.oB_w2[k] = s.oB_w2[k];
  }
  m.n_parts = s.n_parts; m.part_max = s.part_floats_max; m.part_bytes_max = s.part_bytes_max; m.head_part = s.head_part;
  for (int p = 0; p < 5; ++p) { m.part_off[p] = s.part_off[p]; m.partB_off[p] = s.partB_off[p]; }
  for (int k = 0; k < SF_NBMAX; ++k) m.blk_part[k] = s.blk_part[k];
  m.packedB = f->d_packed16B;
  m.tB_stride = s.tB_stride;
  m.hidden_bf16 = 2;
  m.packed16 = nullptr; m.packed16B = nullptr;
}
static SfDev sampler_dev(const sf_flow* f, const float* x) {
  SfDev m = f->dev();
  if (f->ctab_x != nullptr && f->ctab_x == x) {
    sf_ctab_shape(m, m.ctab_R, m.ctab_NV);
    m.ctab = f->d_ctab;
  }
  nsf_sampler_view(f, m);
  return m;
}

int sf_flow_sample_round(sf_flow* f, const float* x, int64_t S, const uint32_t* slots, int64_t slot_base,
                         int64_t n_slots, uint32_t attempt, int32_t attempts_per_slot, uint64_t seed,
                         uint32_t stream_id,
                         const float* lo, const float* hi, float* out, uint32_t* rejected,
                         uint32_t* n_rejected, int32_t* n_drawn, void* stream) {
  if (!f || !x || !out || !rejected || !n_rejected) return fail(SF_ERR_INVALID, "null argument");
  if (f->nsf1 || f->nsfar) return fail(SF_ERR_INVALID, "sf_flow_sample_round is not offered for the one-parameter / autoregressive NSF: use sf_flow_sample / sf_flow_sample_slots");
  if (!f->params_set) return fail(SF_ERR_STATE, "sf_flow_set_params has not been called");
  if (S < 1 || S > 0x7fffffffll) return fail(SF_ERR_INVALID, "S must be in 1 .. 2^31-1");
  if ((lo == nullptr) != (hi == nullptr)) return fail(SF_ERR_INVALID, "lo and hi must be given together");
  if ((uint64_t)(slot_base + n_slots) > 0xffffffffull)
    return fail(SF_ERR_INVALID, "slot ids must fit 32 bits: split the catalogue");
  const int A = attempts_per_slot;
  if (A < 1 || A > 32 || (A & (A - 1))) return fail(SF_ERR_INVALID, "attempts_per_slot must be 1,2,4,8,16 or 32");
  SfSampleArgsHost a;
  a.x = x; a.S = (long)S; a.slots = slots; a.slot_base = (long)slot_base; a.n_items = (long)n_slots * A;
  a.attempts_per_slot = A; a.attempt = attempt; seed_keys(seed, stream_id, a.k0, a.k1);
  a.rng_slot_offset = (unsigned long long)f->sample_row_offset * (unsigned long long)S;
  a.lo = lo; a.hi = hi; a.out = out; a.rejected = rejected; a.n_rejected = n_rejected; a.n_drawn = n_drawn;
  if (f->ctab_x == x && (uint64_t)(slot_base + n_slots) > (uint64_t)f->ctab_M * (uint64_t)S && slots == nullptr)
    return fail(SF_ERR_INVALID, "slots reach past the rows given to sf_flow_prepare_context");
  SF_HIP(sf_launch_inverse(sampler_dev(f, x), a, (hipStream_t)stream));
  return SF_OK;
}

// ---- persistent sampler ------------------------------------------------------------------------
// Launches of one sampling call.  The first persistent launch resolves the dense slot list -- first attempts and
// retries -- up to 1024 attempts per slot (or the caller's ceiling).  Slots that are still empty afterwards
// ("survivors": the flow's mass for that galaxy lies almost entirely outside the prior box) go through further
// launches with the attempt windows [1024, 16384), [16384, 262144), ... until they are filled, the caller's ceiling
// `max_attempts` is reached, or -- when the caller set no ceiling (max_attempts <= 0) -- a galaxy got NOT ONE draw
// accepted between its 64th attempt and the end of a window: its acceptance is then zero to within
// 1 / (window x open slots) and its open slots become NaN rows, which is what the reference's timeout / error path
// produces (ref: sbi_runner.py:6443-6460); [UPSTREAM] accept_reject_sample itself would loop forever on such a galaxy.
static int ensure_queue(sf_flow* f, int64_t n_slots, int64_t M) {
  if (!f->d_queue) SF_HIP(hipMalloc(&f->d_queue, sizeof(SfQueue)));
  if (!f->h_queue) SF_HIP(hipHostMalloc((void**)&f->h_queue, sizeof(SfQueue), hipHostMallocDefault));
  // ring positions are tickets of idle workgroups: at most (resident workgroups x items per iteration) are in flight at a
  // time (<= 2048 x 256), whatever the size of the catalogue -- 2^20 entries (8 MiB) never alias
  uint64_t cap = 1u << 16;
  while (cap < (uint64_t)n_slots && cap < (1ull << 20)) cap <<= 1;
  if (f->ring_cap < cap) {
    (void)hipFree(f->d_ring);
    f->d_ring = nullptr; f->ring_cap = 0;
    SF_HIP(hipMalloc(&f->d_ring, cap * sizeof(unsigned long long)));
    SF_HIP(hipMemset(f->d_ring, 0, cap * sizeof(unsigned long long)));  // consumers clear what they take: stays zero
    f->ring_cap = cap;
    f->ring_dirty = false;
  }
  if (f->rej_cap < (size_t)n_slots) {
    (void)hipFree(f->d_rej[0]); (void)hipFree(f->d_rej[1]);
    f->d_rej[0] = f->d_rej[1] = nullptr; f->rej_cap = 0;
    SF_HIP(hipMalloc(&f->d_rej[0], (size_t)n_slots * sizeof(uint32_t)));
    SF_HIP(hipMalloc(&f->d_rej[1], (size_t)n_slots * sizeof(uint32_t)));
    f->rej_cap = (size_t)n_slots;
  }
  if (f->galacc_cap < (size_t)M) {
    (void)hipFree(f->d_galacc);
    f->d_galacc = nullptr; f->galacc_cap = 0;
    SF_HIP(hipMalloc(&f->d_galacc, (size_t)M * sizeof(int32_t)));
    f->galacc_cap = (size_t)M;
  }
  return SF_OK;
}

// x [M,C]; slots: device list of n_slots output slots (NULL = all of 0 .. M*S-1); see sf_flow_sample_slots
static int sample_persistent(sf_flow* f, const float* x, int64_t M, int64_t S, const uint32_t* slots, int64_t n_slots,
                             const float* lo, const float* hi, uint64_t seed, int32_t max_attempts, float* out,
                             int32_t* n_drawn, int64_t* n_unfilled, hipStream_t st) {
  if (n_unfilled) *n_unfilled = 0;
  if (n_slots == 0) return SF_OK;
  if ((uint64_t)(M * S) > 0xfff00000ull) return fail(SF_ERR_INVALID, "M*S must stay below 2^32 - 2^20: split the catalogue");
  int rc = ensure_queue(f, n_slots, M);
  if (rc) return rc;
  const bool capped = max_attempts > 0;
  const uint32_t ceiling = capped ? (uint32_t)max_attempts : 0x40000000u;
  {
    rc = sf_flow_prepare_context(f, x, M, (void*)st);
    if (rc) return rc;
  }
  const SfDev m = sampler_dev(f, x);
  SfSampleArgsHost a;
  a.x = x; a.S = (long)S; seed_keys(seed, 0, a.k0, a.k1);
  a.rng_slot_offset = (unsigned long long)f->sample_row_offset * (unsigned long long)S;
  a.lo = lo; a.hi = hi; a.out = out; a.n_drawn = n_drawn; a.out_f64 = f->sample_out_f64 ? 1 : 0;
  a.q = f->d_queue; a.ring = f->d_ring; a.ring_mask = (uint32_t)(f->ring_cap - 1);
  a.out_slots = (uint32_t)(M * S);
  const uint32_t* cur = slots;
  int64_t pending = n_slots;
  const auto t_start = std::chrono::steady_clock::now();
  auto out_of_time = [&]() {
    return f->sample_time_limit_s > 0.0 &&
           std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > f->sample_time_limit_s;
  };
  // the 16-row MAF kernel tries a slot 64 times per workgroup iteration, the 32-row kernels 32 times per tile: the
  // first (persistent) launch goes as deep as a short sequential chain allows, the windows beyond are chip-wide
  const bool fast16 = m.kind == SF_MAF && m.m16_ok && !m.hidden_bf16 && m.packed16 != nullptr;
  // (32-row kernels, NSF cfg3 per 2e7 draws with the interleaved dense order and find launches that double the attempts
  //  instead of covering a whole window: 78 ms at 256, 81 at 1 024 -- an iteration of that kernel costs 180 us, so a
  //  slot that needs hundreds of attempts is cheaper side by side in a find launch than 32 at a time in the tail; the
  //  16-row MAF kernel, 35 us per sparse iteration and 64 attempts wide, is best left at 1 024)
  uint32_t first_window = fast16 ? 1024u : 256u;
  {
    static int env_w = -1;  // developer knob: SF_FIRST_WINDOW=<attempts> (power of two)
    if (env_w < 0) { const char* e = std::getenv("SF_FIRST_WINDOW"); env_w = e ? std::atoi(e) : 0; }
    if (env_w >= 64) first_window = (uint32_t)env_w;
  }
  uint32_t attempt = 0, limit = ceiling < first_window ? ceiling : first_window;
  int buf = 0, stage = 0;
  double evals = 0.0;
  float rej0 = 0.f;
  int64_t dropped = 0;
  const bool progress_rule = !capped;
  // The progress rule is looked at where a window ends (attempts 1024, 16384, 262144, ...), and only once the galaxy's
  // open slots have seen enough attempts since the last look for "no draw accepted" to mean something (1e5 attempts
  // over S slots: acceptance below ~3e-5 at 95 %); otherwise the counters carry over into the next window.
  uint32_t acc_from = 64;
  // Evidence = attempts since the last look x slots of the galaxy that are actually being worked.  sf_flow_sample works all S
  // slots of a galaxy; sf_flow_sample_slots (an ensemble member's share) may list ONE slot of a galaxy, so there the rule
  // counts one slot per galaxy: a listed slot is given up only after ~1e5 attempts of its own (with S in its place a
  // single-slot galaxy of acceptance 1e-3 was written off after one 1 024-attempt window).
  const uint64_t S_rule = slots ? 1ull : (uint64_t)S;
  auto rule_due = [&](uint32_t att_now) { return progress_rule && (uint64_t)(att_now - acc_from) * S_rule >= 100000ull; };
  // ---- the persistent kernel -- first attempts and retries of every slot, attempts [0, limit); then, while MANY slots are
  // still open (a catalogue of 1e5 galaxies leaves tens of thousands past 1 024 attempts: its few-per-mille galaxies of
  // acceptance ~1e-4 need ~1e4 attempts for each of their S slots), further persistent launches over the survivor list
  // with the windows [1 024, 16 384), [16 384, 262 144) ...: the queue keeps every lane busy with speculation 64 wide
  // at the sampler kernel's cost per evaluation.  (Chip-wide find / resolve launches, below, are for the FEW slots that
  // remain: they spread one slot's attempts over the whole chip, on the plain fp32 kernels.)
  int64_t persist_min = 8192;
  {
    static long env_pm = -1;  // developer knob: SF_PERSIST_MIN=<slots> (1 = every window on the persistent kernel: tests)
    if (env_pm < 0) { const char* e = std::getenv("SF_PERSIST_MIN"); env_pm = e ? std::atol(e) : 0; }
    if (env_pm > 0) persist_min = env_pm;
  }
  for (;;) {
    // The retry ring stays all-zero only while every launch ends cleanly (consumers clear what they take).  A launch that
    // ended on a queue error, or never completed, may have left donated entries behind: clear the ring before it is reused.
    if (f->ring_dirty) SF_HIP(hipMemsetAsync(f->d_ring, 0, f->ring_cap * sizeof(unsigned long long), st));
    f->ring_dirty = true;
    SF_HIP(hipMemsetAsync(f->d_queue, 0, sizeof(SfQueue), st));
#ifdef SF_Q_STATS
    SF_HIP(hipMemsetAsync(&f->d_queue->stats[10], 0xff, sizeof(unsigned long long), st));  // atomicMin target
#endif
    if (progress_rule && stage == 0) SF_HIP(hipMemsetAsync(f->d_galacc, 0, (size_t)M * sizeof(int32_t), st));  // (later: carried over)
    a.slots = cur; a.slot_base = 0; a.n_items = (long)pending; a.n_total = (uint32_t)pending;
    {
      static int env_il = -1;  // developer knob: SF_INTERLEAVE=<galaxies per block>, 0 = plain slot order (A-B runs)
      if (env_il < 0) { const char* e = std::getenv("SF_INTERLEAVE"); env_il = e ? std::atoi(e) : 128; }
      a.dense_G = (!cur && env_il > 0 && pending == M * S && M > 1 && (int64_t)env_il * S < (int64_t)1 << 31) ? (uint32_t)env_il : 0u;
      {
        static int env_run = -1;  // developer knob: SF_DENSE_RUN=<draws> (0 = the kernel's tile: 16 / 32)
        if (env_run < 0) { const char* e = std::getenv("SF_DENSE_RUN"); env_run = e ? std::atoi(e) : 0; }
        uint32_t run = env_run > 0 ? (uint32_t)env_run : (fast16 ? 16u : 32u);
        // (S = 1000 does not divide into tiles of 16 / 32: the largest power of two that divides S -- 8 -- still gives every
        //  tile runs of consecutive draws of few galaxies; round 4 fell back to draw-by-draw order there)
        while (run > 1 && S % (int64_t)run != 0) run >>= 1;
        a.dense_run = run;
      }
      a.list_mul = 0; a.list_log2 = 0;
      if (cur && stage == 0 && env_il > 0 && pending >= 4096 && pending < (1ll << 31)) {  // the caller's list (sorted by slot): strided walk
        uint32_t k = 12;
        while ((1ull << k) < (uint64_t)pending) ++k;
        a.list_log2 = k;
        a.list_mul = (uint32_t)(((1ull << k) / 128u) | 1u);   // odd: a bijection of [0, 2^k)
      }
      static int env_sp = -1;  // developer knob: SF_SPEC_AFTER=<attempts> (0 = width grows with the attempt number only)
      if (env_sp < 0) { const char* e = std::getenv("SF_SPEC_AFTER"); env_sp = e ? std::atoi(e) : 0; }
      a.spec_full_after = (uint32_t)env_sp;
      static int env_tc = -1;  // developer knob: SF_TAIL_CAP=<items> (0 = the whole iteration)
      if (env_tc < 0) { const char* e = std::getenv("SF_TAIL_CAP"); env_tc = e ? std::atoi(e) : 0; }
      a.tail_cap = (uint32_t)env_tc;
    }
    a.attempt = attempt; a.attempt_limit = limit; a.attempts_per_slot = 1;
    a.rejected = f->d_rej[buf];
    a.gal_acc = progress_rule ? f->d_galacc : nullptr;
#ifdef SF_Q_STATS
    static uint32_t* d_qtrace = nullptr;
    const size_t qtrace_bytes = (size_t)2048 * 256 * 4 * sizeof(uint32_t);
    if (std::getenv("SF_Q_TRACE")) {
      if (!d_qtrace) SF_HIP(hipMalloc(&d_qtrace, qtrace_bytes));
      SF_HIP(hipMemsetAsync(d_qtrace, 0, qtrace_bytes, st));
      a.qtrace = d_qtrace;
    }
#endif
    if (stage == 0) SF_HIP(hipEventRecord(f->ev_dense[0], st));  // (the reported launch time is the first window's)
    hipError_t e = sf_launch_inverse(m, a, st);
    if (e != hipSuccess) { f->ctab_x = nullptr; return hip_fail(e, "persistent sampler launch"); }
    if (stage == 0) SF_HIP(hipEventRecord(f->ev_dense[1], st));
    if (limit >= 1024u && rule_due(limit)) {  // drop the open slots of galaxies that made no progress (NaN rows), in place
      SF_HIP(sf_launch_filter_survivors(f->d_rej[buf], &f->d_queue->n_surv, (long)S, f->d_galacc, out, f->L.dev.D, st, a.out_f64));
      SF_HIP(hipMemsetAsync(f->d_galacc, 0, (size_t)M * sizeof(int32_t), st));
      acc_from = limit;
    }
    SF_HIP(hipMemcpyAsync(f->h_queue, f->d_queue, sizeof(SfQueue), hipMemcpyDeviceToHost, st));  // pinned
    SF_HIP(hipStreamSynchronize(st));
    if (f->h_queue->error) {
      f->ctab_x = nullptr;
      return fail(SF_ERR_STATE, "persistent sampler: work queue inconsistent (code " + std::to_string(f->h_queue->error) +
                                    ": 1 = a device-side wait exceeded its bound, 2 = foreign ring entry, 3 = survivor "
                                    "list overflow, 4 = listed slot outside M*S, 5 = claimed entry never arrived, 6/7 = "
                                    "claim contention); head " + std::to_string(f->h_queue->head) + " reserve " +
                                    std::to_string(f->h_queue->reserve) + " resolved " + std::to_string(f->h_queue->resolved) +
                                    " of " + std::to_string((unsigned)pending) + " survivors " + std::to_string(f->h_queue->n_surv) +
                                    " dense_next " + std::to_string(f->h_queue->dense_next));
    }
    f->ring_dirty = false;  // clean end: every ring entry was consumed
    evals += (double)f->h_queue->evals;
    dropped += (int64_t)f->h_queue->dropped;
#ifdef SF_Q_STATS
    if (a.qtrace) {  // the last launch's trace stays in the file
      std::vector<uint32_t> h(qtrace_bytes / sizeof(uint32_t));
      SF_HIP(hipMemcpy(h.data(), a.qtrace, qtrace_bytes, hipMemcpyDeviceToHost));
      if (h[1] != 0u) {  // (kernels without the trace code leave the buffer empty: keep the previous file)
        if (FILE* fp = std::fopen(std::getenv("SF_Q_TRACE"), "wb")) { std::fwrite(h.data(), 1, qtrace_bytes, fp); std::fclose(fp); }
      }
      a.qtrace = nullptr;
    }
#endif
    if (std::getenv("SF_Q_STATS")) {
      const unsigned long long* q = f->h_queue->stats;
      std::fprintf(stderr, "[sf_queue] stage %d: entry-barrier %.3e cyc, serial %.3e cyc, exit-barrier %.3e cyc, idle %.3e cyc in %llu "
                   "episodes, %llu donations, iterations dense %llu tail %llu, entries %llu, evals %llu, resolved %u/%u\n", stage,
                   (double)q[0], (double)q[1], (double)q[2], (double)q[3], q[4], q[5], q[6], q[7], q[8],
                   (unsigned long long)f->h_queue->evals, f->h_queue->resolved, (unsigned)a.n_total);
      std::fprintf(stderr, "[sf_queue]   max iterations of a workgroup %llu; last flow evaluation ended %.1f us, last exit %.1f us after the first start\n",
                   q[11], ((double)q[12] - (double)q[10]) * 0.01, ((double)q[13] - (double)q[10]) * 0.01);
      std::fprintf(stderr, "[sf_queue]   wave 0 of every workgroup, summed, dense mode (us): fetch %.0f, prologue %.0f, staging %.0f, passes %.0f, epilogue %.0f\n",
                   (double)q[14] * 0.01, (double)q[15] * 0.01, (double)q[16] * 0.01, (double)q[17] * 0.01, (double)q[18] * 0.01);
      std::fprintf(stderr, "[sf_queue]   the same in tail mode (us): fetch %.0f, prologue %.0f, staging %.0f, passes %.0f, epilogue %.0f\n",
                   (double)q[19] * 0.01, (double)q[20] * 0.01, (double)q[21] * 0.01, (double)q[22] * 0.01, (double)q[23] * 0.01);
    }
    if (stage == 0) rej0 = (float)f->h_queue->rej0;
    pending = (int64_t)f->h_queue->n_surv;
    cur = f->d_rej[buf];
    buf ^= 1;
    ++stage;
    attempt = limit;
    if (pending < persist_min || attempt >= ceiling || out_of_time()) break;
    limit = (attempt > ceiling / 16u) ? ceiling : attempt * 16u;
  }
  // ---- deep tail: the few slots that used up `limit` attempts (their galaxies accept less than ~1 draw in a
  // thousand).  One slot's attempts are now spread over the whole chip instead of over one workgroup: a FIND launch
  // evaluates attempts [a_lo, a_lo + A) of every survivor side by side (plain kernel, an accepted attempt only lowers
  // best[i] with an atomic min), a RESOLVE launch re-evaluates exactly attempt best[i] and writes the draw -- the
  // slot still keeps its LOWEST accepted attempt, so the result does not depend on A or on the schedule.
  if (pending > 0 && attempt < ceiling) {
    if (f->best_cap < (size_t)pending) {
      (void)hipFree(f->d_best);
      f->d_best = nullptr; f->best_cap = 0;
      SF_HIP(hipMalloc(&f->d_best, (size_t)pending * sizeof(uint32_t)));
      f->best_cap = (size_t)pending;
    }
    SfSampleArgsHost p = a;  // plain launches
    p.q = nullptr; p.ring = nullptr; p.gal_acc = nullptr; p.n_drawn = nullptr;
    uint32_t window_end = attempt < 1024u ? 1024u : ((attempt > ceiling / 16u) ? ceiling : attempt * 16u);
    if (window_end > ceiling) window_end = ceiling;
    while (pending > 0 && attempt < ceiling && !out_of_time()) {
      // as many new attempts per survivor as it has already failed (a slot that has failed n attempts needs ~n more on
      // average: doubling wastes at most half of a launch, a launch over the whole window up to 15/16 of it), within a
      // budget of 4M items per launch
      uint32_t A = 32;
      static int find_div = -1;  // developer knob: SF_FIND_DIV=<d>: a round tries at most attempt / d new attempts per survivor
      if (find_div < 0) { const char* e = std::getenv("SF_FIND_DIV"); find_div = e ? std::atoi(e) : 1; if (find_div < 1) find_div = 1; }
      const uint32_t a_max = attempt / (uint32_t)find_div < 64u ? 64u : attempt / (uint32_t)find_div;
      while ((uint64_t)(2u * A) * (uint64_t)pending <= (1ull << 22) && 2u * A <= 65536u && 2u * A <= a_max) A *= 2;
      while (A > 1 && (uint64_t)attempt + A > (uint64_t)window_end) A /= 2;  // windows (and the caller's ceiling) are exact
      SF_HIP(hipMemsetAsync(f->d_best, 0xff, (size_t)pending * sizeof(uint32_t), st));
      p.slots = cur; p.slot_base = 0; p.n_items = (long)pending * A; p.attempts_per_slot = (int)A; p.attempt = attempt;
      p.best = f->d_best; p.att_list = nullptr; p.rejected = nullptr; p.n_rejected = nullptr;
      hipError_t e = sf_launch_inverse(m, p, st);
      if (e != hipSuccess) { f->ctab_x = nullptr; return hip_fail(e, "find launch"); }
      SF_HIP(sf_launch_account_window(cur, f->d_best, (long)pending, (long)S, attempt, A, n_drawn,
                                      progress_rule ? f->d_galacc : nullptr, st));
      SF_HIP(hipMemsetAsync(&f->d_queue->n_surv, 0, 2 * sizeof(unsigned int), st));  // n_surv, dropped
      p.n_items = (long)pending; p.attempts_per_slot = 1; p.best = nullptr; p.att_list = f->d_best;
      p.rejected = f->d_rej[buf]; p.n_rejected = &f->d_queue->n_surv;
      e = sf_launch_inverse(m, p, st);
      if (e != hipSuccess) { f->ctab_x = nullptr; return hip_fail(e, "resolve launch"); }
      evals += (double)pending * A + (double)pending;
      attempt += A;
      const bool window_done = attempt >= window_end || attempt >= ceiling;
      const bool look = window_done && rule_due(attempt);
      if (look) {
        SF_HIP(sf_launch_filter_survivors(f->d_rej[buf], &f->d_queue->n_surv, (long)S, f->d_galacc, out, f->L.dev.D, st, a.out_f64));
        SF_HIP(hipMemsetAsync(f->d_galacc, 0, (size_t)M * sizeof(int32_t), st));
        acc_from = attempt;
      }
      SF_HIP(hipMemcpyAsync(f->h_queue, f->d_queue, sizeof(SfQueue), hipMemcpyDeviceToHost, st));
      SF_HIP(hipStreamSynchronize(st));
      if (look) dropped += (int64_t)f->h_queue->dropped;
      if (window_done) window_end = (window_end > ceiling / 16u) ? ceiling : window_end * 16u;
      pending = (int64_t)f->h_queue->n_surv;
      cur = f->d_rej[buf];
      buf ^= 1;
      ++stage;
    }
  }
  {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, f->ev_dense[0], f->ev_dense[1]) != hipSuccess) ms = 0.f;
    f->last_stats[0] = ms; f->last_stats[1] = (float)stage; f->last_stats[2] = rej0; f->last_stats[3] = (float)evals;
  }
  f->ctab_x = nullptr;
  if (pending > 0) SF_HIP(sf_launch_fill_nan_rows(out, cur, (long)pending, f->L.dev.D, st, a.out_f64));
  if (n_unfilled) *n_unfilled = pending + dropped;
  return SF_OK;
}

int sf_flow_sample(sf_flow* f, const float* x, int64_t M, int64_t S, const float* lo, const float* hi,
                   uint64_t seed, int32_t max_attempts, float* out, int32_t* n_drawn, int64_t* n_unfilled,
                   void* stream) {
  if (!f) return fail(SF_ERR_INVALID, "null handle");
  if (n_unfilled) *n_unfilled = 0;
  if (M == 0) return SF_OK;
  if (!x || !out) return fail(SF_ERR_INVALID, "null argument");
  if (!f->params_set) return fail(SF_ERR_STATE, "sf_flow_set_params has not been called");
  if (M < 0 || S < 1) return fail(SF_ERR_INVALID, "bad M or S");
  if ((lo == nullptr) != (hi == nullptr)) return fail(SF_ERR_INVALID, "lo and hi must be given together");
  hipStream_t st = (hipStream_t)stream;
  if (f->nsf1) {
    if (n_drawn) SF_HIP(sf_launch_fill_i32(n_drawn, (long)M, 0, st));
    uint32_t k0, k1;
    seed_keys(seed, 0u, k0, k1);
    std::string err;
    int rc = sf_nsf1_sample(f->nsf1, f->d_flat, x, (long)M, (long)S, nullptr, (long)(M * S), lo, hi, k0, k1,
                            (unsigned long long)f->sample_row_offset * (unsigned long long)S, max_attempts, out, n_drawn, n_unfilled, st, err);
    return rc ? fail(rc, err) : SF_OK;
  }
  if (f->nsfar) {
    if (n_drawn) SF_HIP(sf_launch_fill_i32(n_drawn, (long)M, 0, st));
    uint32_t k0, k1;
    seed_keys(seed, 0u, k0, k1);
    std::string err;
    if (!f->ev_dense[0]) { SF_HIP(hipEventCreate(&f->ev_dense[0])); SF_HIP(hipEventCreate(&f->ev_dense[1])); }
    int64_t unf = 0;
    int rc = sf_nsfar_sample(f->nsfar, x, (long)M, (long)S, nullptr, (long)(M * S), lo, hi, k0, k1,
                             (unsigned long long)f->sample_row_offset * (unsigned long long)S, max_attempts, out, n_drawn, nullptr, &unf, st, err,
                             f->ev_dense[0], f->ev_dense[1]);
    if (rc) return fail(rc, err);
    if (n_unfilled) *n_unfilled = unf;
    float ms = 0.f;   // (the counters' read-back has synchronised the stream)
    SF_HIP(hipEventElapsedTime(&ms, f->ev_dense[0], f->ev_dense[1]));
    f->last_stats[0] = ms; f->last_stats[1] = 1.f; f->last_stats[2] = (float)f->nsfar->last_rej0; f->last_stats[3] = (float)f->nsfar->last_evals;
    return SF_OK;
  }
  if (n_drawn) SF_HIP(sf_launch_fill_i32(n_drawn, (long)M, (int32_t)S, st));
  return sample_persistent(f, x, M, S, nullptr, M * S, lo, hi, seed, max_attempts, out, n_drawn, n_unfilled, st);
}

int sf_flow_sample_slots(sf_flow* f, const float* x, int64_t M, int64_t S, const uint32_t* slots, int64_t n_slots,
                         const float* lo, const float* hi, uint64_t seed, int32_t max_attempts, float* out,
                         int64_t* n_unfilled, void* stream) {
  if (!f) return fail(SF_ERR_INVALID, "null handle");
  if (n_unfilled) *n_unfilled = 0;
  if (n_slots == 0) return SF_OK;
  if (!x || !out || !slots) return fail(SF_ERR_INVALID, "null argument");
  if (!f->params_set) return fail(SF_ERR_STATE, "sf_flow_set_params has not been called");
  if (M < 1 || S < 1 || n_slots < 0 || n_slots > M * S) return fail(SF_ERR_INVALID, "bad M, S or n_slots");
  if ((lo == nullptr) != (hi == nullptr)) return fail(SF_ERR_INVALID, "lo and hi must be given together");
  if (f->nsf1) {
    uint32_t k0, k1;
    seed_keys(seed, 0u, k0, k1);
    std::string err;
    int rc = sf_nsf1_sample(f->nsf1, f->d_flat, x, (long)M, (long)S, slots, (long)n_slots, lo, hi, k0, k1,
                            (unsigned long long)f->sample_row_offset * (unsigned long long)S, max_attempts, out, nullptr, n_unfilled,
                            (hipStream_t)stream, err);
    return rc ? fail(rc, err) : SF_OK;
  }
  if (f->nsfar) {
    uint32_t k0, k1;
    seed_keys(seed, 0u, k0, k1);
    std::string err;
    int rc = sf_nsfar_sample(f->nsfar, x, (long)M, (long)S, slots, (long)n_slots, lo, hi, k0, k1,
                             (unsigned long long)f->sample_row_offset * (unsigned long long)S, max_attempts, out, nullptr, nullptr, n_unfilled,
                             (hipStream_t)stream, err);
    return rc ? fail(rc, err) : SF_OK;
  }
  return sample_persistent(f, x, M, S, slots, n_slots, lo, hi, seed, max_attempts, out, nullptr, n_unfilled,
                           (hipStream_t)stream);
}

int sf_flow_sample_stats(const sf_flow* f, float* stats4) {
  if (!f || !stats4) return fail(SF_ERR_INVALID, "null argument");
  for (int i = 0; i < 4; ++i) stats4[i] = f->last_stats[i];
  return SF_OK;
}

int sf_flow_acceptance(sf_flow* f, const float* x, int64_t M, int64_t n, const float* lo, const float* hi,
                       uint64_t seed, int32_t* count, void* stream) {
  if (!f || !x || !count || !lo || !hi) return fail(SF_ERR_INVALID, "null argument");
  if (!f->params_set) return fail(SF_ERR_STATE, "sf_flow_set_params has not been called");
  if (M < 0 || n < 1 || n > 0x7fffffffll) return fail(SF_ERR_INVALID, "bad M or n");
  if ((uint64_t)(M * n) > 0xffffffffull) return fail(SF_ERR_INVALID, "M*n must fit 32 bits");
  hipStream_t st = (hipStream_t)stream;
  SF_HIP(sf_launch_fill_i32(count, (long)M, 0, st));
  if (f->nsf1) {
    uint32_t k0, k1;
    seed_keys(seed, 1u, k0, k1);
    std::string err;
    int rc = sf_nsf1_acceptance(f->nsf1, f->d_flat, x, (long)M, (long)n, lo, hi, k0, k1,
                                (unsigned long long)f->sample_row_offset * (unsigned long long)n, count, st, err);
    return rc ? fail(rc, err) : SF_OK;
  }
  if (f->nsfar) {
    uint32_t k0, k1;
    seed_keys(seed, 1u, k0, k1);
    std::string err;
    int rc = sf_nsfar_sample(f->nsfar, x, (long)M, (long)n, nullptr, (long)(M * n), lo, hi, k0, k1,
                             (unsigned long long)f->sample_row_offset * (unsigned long long)n, 1, nullptr, nullptr, count, nullptr, st, err);
    return rc ? fail(rc, err) : SF_OK;
  }
  SfSampleArgsHost a;
  a.x = x; a.S = (long)n; a.n_items = (long)(M * n); seed_keys(seed, 1u, a.k0, a.k1);
  a.rng_slot_offset = (unsigned long long)f->sample_row_offset * (unsigned long long)n;
  a.lo = lo; a.hi = hi; a.count = count;
  {
    int rc = sf_flow_prepare_context(f, x, M, stream);
    if (rc) return rc;
  }
  const SfDev m = sampler_dev(f, x);
  f->ctab_x = nullptr;
  SF_HIP(sf_launch_inverse(m, a, st));
  return SF_OK;
}

int sf_copy_to_host_f64(const float* dev_src, double* host_dst, int64_t n, void* stream) {
  if (n == 0) return SF_OK;
  if (!dev_src || !host_dst || n < 0) return fail(SF_ERR_INVALID, "null argument or n < 0");
  std::string err;
  int rc = sf_hostio_copy_f64(dev_src, host_dst, n, (hipStream_t)stream, err);
  return rc ? fail(rc, err) : SF_OK;
}

// ---- training ------------------------------------------------------------------------------
int sf_flow_loss_grad_weighted(sf_flow* f, const float* flat, const float* theta, const float* x, int64_t B,
                               float grad_scale, const float* weights, float* loss, float* grad,
                               float* dctx, void* stream) {
  if (!f || !flat || !grad) return fail(SF_ERR_INVALID, "null argument");
  if (B > 0 && (!theta || !x)) return fail(SF_ERR_INVALID, "null argument");
  if (B < 0) return fail(SF_ERR_INVALID, "B < 0");
  int rc = ensure_device(f);
  if (rc) return rc;
  std::string err;
  rc = sf_train_loss_grad(f, flat, theta, x, nullptr, (long)B, grad_scale, weights, loss, nullptr, grad, dctx,
                          (hipStream_t)stream, err);
  if (rc) return fail(rc, err);
  f->params_set = true;  // the forward image now holds `flat`
  f->flat_valid = false; // ... but the handle's own copy of the logical vector does not
  f->ctab_x = nullptr;
  return SF_OK;
}

int sf_flow_loss_grad_rows(sf_flow* f, const float* flat, const float* theta, const float* x, const int64_t* rows,
                           int64_t B, float grad_scale, const float* weights, float* loss, double* loss_sum,
                           float* grad, float* dctx, void* stream) {
  if (!f || !flat || !grad) return fail(SF_ERR_INVALID, "null argument");
  if (B > 0 && (!theta || !x || !rows)) return fail(SF_ERR_INVALID, "null argument");
  if (B < 0) return fail(SF_ERR_INVALID, "B < 0");
  int rc = ensure_device(f);
  if (rc) return rc;
  std::string err;
  rc = sf_train_loss_grad(f, flat, theta, x, reinterpret_cast<const long long*>(rows), (long)B, grad_scale, weights, loss,
                          loss_sum, grad, dctx, (hipStream_t)stream, err);
  if (rc) return fail(rc, err);
  f->params_set = true;
  f->flat_valid = false;
  f->ctab_x = nullptr;
  return SF_OK;
}

int sf_flow_train_epoch(sf_flow* f, float* flat, const float* theta, const float* x, const int64_t* order,
                        int64_t n_batches, int64_t batch, float grad_scale, float* exp_avg, float* exp_avg_sq,
                        const sf_adam_desc* d, int64_t step0, float max_norm, float* scratch, float* grad,
                        double* loss_sum, void* stream) {
  return sf_flow_train_epoch_dp(f, flat, theta, x, order, n_batches, batch, grad_scale, exp_avg, exp_avg_sq, d, step0, max_norm,
                                scratch, grad, loss_sum, nullptr, stream);
}

int sf_flow_train_epoch_dp(sf_flow* f, float* flat, const float* theta, const float* x, const int64_t* order,
                           int64_t n_batches, int64_t batch, float grad_scale, float* exp_avg, float* exp_avg_sq,
                           const sf_adam_desc* d, int64_t step0, float max_norm, float* scratch, float* grad,
                           double* loss_sum, sf_comm* comm, void* stream) {
  if (!f || !flat || !theta || !x || !order || !exp_avg || !exp_avg_sq || !d || !scratch || !grad)
    return fail(SF_ERR_INVALID, "null argument");
  if (n_batches < 0 || batch < 1 || step0 < 0) return fail(SF_ERR_INVALID, "bad n_batches, batch or step0");
  // (the gather of the step leaves |grad|^2 in per-block shares for the clip: the optimiser kernel then skips its pass over the
  //  whole gradient -- one L2 round trip less in a step that is a chain of them)
  //  (data parallel: the norm that clips is the REDUCED gradient's, so the optimiser kernel takes its own pass over it)
  struct WantSq { sf_flow* f; WantSq(sf_flow* f_, bool on) : f(f_) { f->want_sq = on; } ~WantSq() { f->want_sq = false; f->n_sqpart = 0; } } want_sq(f, comm == nullptr);
  // ... and the kernels add their loss sums to one of SF_LOSS_PARTS scalars instead of all to the caller's (folded in below)
  if (loss_sum && !f->d_losspart_mem) {
    if (hipMalloc(&f->d_losspart_mem, SF_LOSS_PARTS * sizeof(double)) == hipSuccess)
      (void)hipMemsetAsync(f->d_losspart_mem, 0, SF_LOSS_PARTS * sizeof(double), (hipStream_t)stream);
    else { f->d_losspart_mem = nullptr; (void)hipGetLastError(); }
  }
  struct Spread {
    sf_flow* f; double* out; hipStream_t st;
    Spread(sf_flow* f_, double* o, hipStream_t s) : f(f_), out(o), st(s) { f->d_losspart = o ? f->d_losspart_mem : nullptr; f->losspart_used = false; }
    ~Spread() {
      if (f->d_losspart && f->losspart_used) (void)sf_launch_fold_loss(f->d_losspart, SF_LOSS_PARTS, out, st);
      f->d_losspart = nullptr;
    }
  } spread(f, loss_sum, (hipStream_t)stream);
  auto plain_step = [&](int64_t b) -> int {
    f->n_sqpart = 0;
    f->prep_lite = b + 1 < n_batches;   // (the last step of the call re-tiles every image)
    int rc = sf_flow_loss_grad_rows(f, flat, theta, x, order + b * batch, batch, grad_scale, nullptr, nullptr, loss_sum, grad,
                                    nullptr, stream);
    if (rc) {   // (the density / sampler images may lag behind: nothing may use them before the next sf_flow_set_params)
      f->prep_lite = false;
      if (f->packed_stale) f->params_set = false;
      return rc;
    }
    if (comm) {   // the exchange step: sum of the ranks' shard gradients, in place, on the same stream
      rc = sf_comm_all_reduce_impl(comm, grad, (long)f->L.n_params, (hipStream_t)stream);
      if (rc) { f->prep_lite = false; if (f->packed_stale) f->params_set = false; return rc; }
      f->n_sqpart = 0;
    }
    const int64_t step = step0 + b + 1;
    const double bc1 = 1.0 - std::pow((double)d->beta1, (double)step), bc2 = 1.0 - std::pow((double)d->beta2, (double)step);
    hipError_t e = sf_launch_adam(flat, grad, exp_avg, exp_avg_sq, scratch, (long)f->L.n_params, *d, (float)bc1, (float)bc2, max_norm,
                                  scratch + 1, (hipStream_t)stream, nullptr, f->n_sqpart > 0 ? f->d_sqpart : nullptr, f->n_sqpart);
    f->prep_lite = false;
    if (e != hipSuccess) return hip_fail(e, "sf_launch_adam");
    return SF_OK;
  };
  // ---- the step as ONE captured HIP graph, replayed per batch.  What changes from step to step -- the batch's rows and Adam's
  // bias correction -- lives on the device: k_step_begin copies rows [ctr[0] * batch, ...) of `order` into a fixed buffer the
  // training kernels read and computes 1 - beta^(ctr[1] + 1); k_step_end advances both counters.  The first step of a call
  // runs the plain way (lazy allocations, function attributes); the graph is kept on the handle and captured again only when
  // an argument changes.  OPT-IN (SF_TRAIN_GRAPH=1): measured on the bench workloads the replay is SLOWER than the four
  // back-to-back launches it replaces -- MAF cfg1 at batch 16 384: 0.117 ms per step against 0.091, NSF cfg3: 0.370 against
  // 0.351 (hipGraphLaunch of a 6-node graph costs more on this runtime than the launches' own overhead, which the GPU hides
  // behind the previous step anyway).  A flow kind whose step is not capturable (the one-parameter NSF allocates inside its
  // MLP calls) never uses it.
  static int use_graph = -1;
  if (use_graph < 0) { const char* e = std::getenv("SF_TRAIN_GRAPH"); use_graph = e ? std::atoi(e) : 0; }
  int64_t b0 = 0;
  if (use_graph && !comm && n_batches >= 4 && !f->nsf1 && !f->nsfar && !f->profiling) {
    int rc = plain_step(0);
    if (rc) return rc;
    b0 = 1;
    hipStream_t user = (hipStream_t)stream;
    auto hip_ok = [&](hipError_t e) { return e == hipSuccess; };
    bool ok = true;
    if (!f->step_stream) {
      ok = hip_ok(hipStreamCreateWithFlags(&f->step_stream, hipStreamNonBlocking)) &&
           hip_ok(hipEventCreateWithFlags(&f->step_ev[0], hipEventDisableTiming)) &&
           hip_ok(hipEventCreateWithFlags(&f->step_ev[1], hipEventDisableTiming)) &&
           hip_ok(hipMalloc(&f->d_step_ctr, 2 * sizeof(long long))) && hip_ok(hipMalloc(&f->d_step_bc, 2 * sizeof(float)));
    }
    if (ok && f->step_rows_cap < (size_t)batch) {
      (void)hipFree(f->d_step_rows);
      f->d_step_rows = nullptr; f->step_rows_cap = 0;
      ok = hip_ok(hipMalloc(&f->d_step_rows, (size_t)batch * sizeof(long long)));
      if (ok) f->step_rows_cap = (size_t)batch;
      if (f->step_exec) { (void)hipGraphExecDestroy(f->step_exec); f->step_exec = nullptr; }   // (the buffer moved)
    }
    if (ok) {
      // (the handle's lazily grown buffers are baked into the captured kernels' arguments: a loss_grad call at a larger batch
      //  between two epoch calls moves them, and the key must notice)
      unsigned long long key[20] = {(unsigned long long)flat, (unsigned long long)theta, (unsigned long long)x, (unsigned long long)order,
                                    (unsigned long long)batch, 0, (unsigned long long)exp_avg, (unsigned long long)exp_avg_sq, 0, 0, 0,
                                    (unsigned long long)scratch, (unsigned long long)grad, (unsigned long long)loss_sum, 0, 0,
                                    (unsigned long long)f->d_gpartC, (unsigned long long)f->d_ustash, (unsigned long long)f->d_sqpart,
                                    (unsigned long long)f->d_gfixC};
      std::memcpy(&key[5], &grad_scale, sizeof(float));
      std::memcpy(&key[8], d, sizeof(sf_adam_desc) < 24 ? sizeof(sf_adam_desc) : 24);
      std::memcpy(&key[14], &max_norm, sizeof(float));
      // the step runs on the handle's own stream, ordered behind the caller's
      ok = hip_ok(hipEventRecord(f->step_ev[0], user)) && hip_ok(hipStreamWaitEvent(f->step_stream, f->step_ev[0], 0));
      const long long ctr0[2] = {1, (long long)(step0 + 1)};   // next batch, Adam steps already taken
      ok = ok && hip_ok(hipMemcpyAsync(f->d_step_ctr, ctr0, sizeof(ctr0), hipMemcpyHostToDevice, f->step_stream)) &&
           hip_ok(hipStreamSynchronize(f->step_stream));   // (ctr0 is a stack variable)
      if (ok && (!f->step_exec || std::memcmp(key, f->step_key, sizeof(key)) != 0)) {
        if (f->step_exec) { (void)hipGraphExecDestroy(f->step_exec); f->step_exec = nullptr; }
        if (f->step_graph) { (void)hipGraphDestroy(f->step_graph); f->step_graph = nullptr; }
        hipStream_t cs = f->step_stream;
        bool cap = hip_ok(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        int rc2 = 0;
        if (cap) {
          if (!hip_ok(sf_launch_step_begin(reinterpret_cast<const long long*>(order), f->d_step_ctr, (long)batch, f->d_step_rows, d->beta1,
                                           d->beta2, f->d_step_bc, cs))) rc2 = SF_ERR_HIP;
          if (!rc2) rc2 = sf_flow_loss_grad_rows(f, flat, theta, x, reinterpret_cast<const int64_t*>(f->d_step_rows), batch, grad_scale, nullptr,
                                                 nullptr, loss_sum, grad, nullptr, cs);
          if (!rc2 && !hip_ok(sf_launch_adam(flat, grad, exp_avg, exp_avg_sq, scratch, (long)f->L.n_params, *d, 1.f, 1.f, max_norm,
                                             scratch + 1, cs, f->d_step_bc, f->n_sqpart > 0 ? f->d_sqpart : nullptr, f->n_sqpart))) rc2 = SF_ERR_HIP;
          if (!rc2 && !hip_ok(sf_launch_step_end(f->d_step_ctr, cs))) rc2 = SF_ERR_HIP;
          hipGraph_t g = nullptr;
          const bool ended = hip_ok(hipStreamEndCapture(cs, &g));
          if (!rc2 && ended && g && hip_ok(hipGraphInstantiate(&f->step_exec, g, nullptr, nullptr, 0))) {
            f->step_graph = g;
            std::memcpy(f->step_key, key, sizeof(key));
          } else {
            if (g) (void)hipGraphDestroy(g);
            f->step_exec = nullptr;
            (void)hipGetLastError();
          }
        }
      }
      if (ok && f->step_exec) {
        for (int64_t b = b0; b < n_batches && ok; ++b) ok = hip_ok(hipGraphLaunch(f->step_exec, f->step_stream));
        if (!ok) return hip_fail(hipGetLastError(), "hipGraphLaunch (training step)");
        SF_HIP(hipEventRecord(f->step_ev[1], f->step_stream));
        SF_HIP(hipStreamWaitEvent(user, f->step_ev[1], 0));
        f->params_set = true; f->flat_valid = false; f->ctab_x = nullptr;
        f->packed_stale = false; f->packed16_stale = false;   // (the captured step re-tiles every image)
        return SF_OK;
      }
    }
    (void)hipGetLastError();   // capture not possible here: the rest of the epoch the plain way
  }
  for (int64_t b = b0; b < n_batches; ++b) {
    int rc = plain_step(b);
    if (rc) return rc;
  }
  return SF_OK;
}

int sf_flow_set_sample_row_offset(sf_flow* f, int64_t row_offset) {
  if (!f) return fail(SF_ERR_INVALID, "null handle");
  if (row_offset < 0) return fail(SF_ERR_INVALID, "row_offset < 0");
  f->sample_row_offset = (long long)row_offset;
  return SF_OK;
}

int sf_flow_set_sample_output_f64(sf_flow* f, int on) {
  if (!f) return fail(SF_ERR_INVALID, "null handle");
  if (on && (f->nsf1 || f->nsfar))
    return fail(SF_ERR_INVALID, "float64 sampler output is not offered for the one-parameter / autoregressive NSF: sample fp32 and "
                                "use sf_copy_to_host_f64");
  f->sample_out_f64 = on != 0;
  return SF_OK;
}

int sf_flow_set_sample_time_limit(sf_flow* f, double seconds) {
  if (!f) return fail(SF_ERR_INVALID, "null handle");
  f->sample_time_limit_s = seconds > 0.0 ? seconds : 0.0;
  return SF_OK;
}

int sf_flow_set_profiling(sf_flow* f, int on) {
  if (!f) return fail(SF_ERR_INVALID, "null handle");
  f->profiling = on != 0;
  f->ev_train_valid = false;
  return SF_OK;
}

int sf_flow_train_stats(sf_flow* f, float* kernel_ms) {
  if (!f || !kernel_ms) return fail(SF_ERR_INVALID, "null argument");
  if (!f->profiling || !f->ev_train_valid) return fail(SF_ERR_STATE, "no profiled training call yet (sf_flow_set_profiling)");
  SF_HIP(hipEventSynchronize(f->ev_train[1]));
  SF_HIP(hipEventElapsedTime(kernel_ms, f->ev_train[0], f->ev_train[1]));
  return SF_OK;
}

int sf_flow_loss_grad(sf_flow* f, const float* flat, const float* theta, const float* x, int64_t B,
                      float grad_scale, float* loss, float* grad, void* stream) {
  return sf_flow_loss_grad_weighted(f, flat, theta, x, B, grad_scale, nullptr, loss, grad, nullptr, stream);
}

}  // extern "C"
void sf_set_error(const std::string& msg) { g_err = msg; }
extern "C" {

struct sf_opt {
  int64_t n = 0;
  sf_adam_desc d{};
  float* m = nullptr;
  float* v = nullptr;
  float* norm = nullptr;  // device scalar: sum of squares
  int64_t step = 0;
};

void sf_opt_destroy(sf_opt* o);
int sf_opt_create(int64_t n, const sf_adam_desc* d, sf_opt** out) {
  if (n < 1 || !d || !out) return fail(SF_ERR_INVALID, "bad argument");
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0) return fail(SF_ERR_NO_DEVICE, "no HIP device visible");
  sf_opt* o = new sf_opt();
  o->n = n;
  o->d = *d;
  hipError_t e = hipMalloc(&o->m, (size_t)n * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&o->v, (size_t)n * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&o->norm, sizeof(float));
  if (e == hipSuccess) e = hipMemset(o->m, 0, (size_t)n * sizeof(float));
  if (e == hipSuccess) e = hipMemset(o->v, 0, (size_t)n * sizeof(float));
  if (e != hipSuccess) {
    sf_opt_destroy(o);
    return hip_fail(e, "sf_opt_create");
  }
  *out = o;
  return SF_OK;
}
void sf_opt_destroy(sf_opt* o) {
  if (!o) return;
  (void)hipFree(o->m); (void)hipFree(o->v); (void)hipFree(o->norm);
  delete o;
}
int sf_adam_step(sf_opt* o, float* params, const float* grad, float max_norm, float* grad_norm_out, void* stream) {
  if (!o || !params || !grad) return fail(SF_ERR_INVALID, "null argument");
  o->step += 1;
  const double bc1 = 1.0 - std::pow((double)o->d.beta1, (double)o->step);
  const double bc2 = 1.0 - std::pow((double)o->d.beta2, (double)o->step);
  hipError_t e = sf_launch_adam(params, grad, o->m, o->v, o->norm, (long)o->n, o->d, (float)bc1, (float)bc2,
                                max_norm, grad_norm_out, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "sf_launch_adam");
  return SF_OK;
}
int sf_adam_apply(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  const sf_adam_desc* d, int64_t step, float max_norm, float* scratch, void* stream) {
  if (!params || !grad || !exp_avg || !exp_avg_sq || !d || !scratch) return fail(SF_ERR_INVALID, "null argument");
  if (n < 1 || step < 1) return fail(SF_ERR_INVALID, "bad n or step");
  const double bc1 = 1.0 - std::pow((double)d->beta1, (double)step);
  const double bc2 = 1.0 - std::pow((double)d->beta2, (double)step);
  hipError_t e = sf_launch_adam(params, grad, exp_avg, exp_avg_sq, scratch, (long)n, *d, (float)bc1, (float)bc2,
                                max_norm, scratch + 1, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "sf_launch_adam");
  return SF_OK;
}
int sf_opt_state(sf_opt* o, float** exp_avg, float** exp_avg_sq, int64_t** step_host) {
  if (!o) return fail(SF_ERR_INVALID, "null argument");
  if (exp_avg) *exp_avg = o->m;
  if (exp_avg_sq) *exp_avg_sq = o->v;
  if (step_host) *step_host = &o->step;
  return SF_OK;
}

}  // extern "C"
