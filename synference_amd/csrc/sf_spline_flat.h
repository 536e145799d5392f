// sf_spline_flat.h -- rational-quadratic spline with linear tails on a thread-local parameter array (forward, inverse,
// backward).  Shared by the cooperative NSF training kernel (sf_nsfc.hip: lane (sample, row group g) owns the parameters of
// transformed dimension g) and the one-parameter NSF (sf_nsf1.hip: one thread per sample).
#pragma once
#include <hip/hip_runtime.h>

#include "sf_device.h"

// ---------------------------------------------------------------------------------------------------------------------
// Rational-quadratic spline with linear tails on a lane-local parameter array ([UPSTREAM] nflows
// unconstrained_rational_quadratic_spline; the same arithmetic as SfSpline / SfSplineBwd in sf_flows.h, whose numpy twin
// tests/spline_bwd_model.py is checked against autograd).  Slots: widths [0, KM), heights [KM, 2 KM), derivatives [2 KM, 3 KM - 1).
// ---------------------------------------------------------------------------------------------------------------------
struct NSplC {
  int K;
  float B, min_w, min_h, min_d, inv_sqrt_h, dconst;
};

template <int KM, int NQV>
struct NSpl {
  // softmax -> knots of one family; p[] = bin probabilities; BY_VALUE: bin = largest k with v >= knot_k, else k == idx
  template <int OFF, bool BY_VALUE>
  static __device__ __forceinline__ void family(const NSplC& c, const float (&q)[NQV], float min_size, float v, int& idx,
                                                float& left, float& size, float (&p)[KM]) {
    const int K = c.K;
    float mx = -3.0e38f;
#pragma unroll
    for (int k = 0; k < KM; ++k)
      if (k < K) {
        p[k] = q[OFF + k] * c.inv_sqrt_h;
        mx = fmaxf(mx, p[k]);
      }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < KM; ++k)
      if (k < K) {
        p[k] = sf_exp(p[k] - mx);
        sum += p[k];
      }
    const float rs = __builtin_amdgcn_rcpf(sum);
    const float scale = 1.0f - min_size * (float)K;
    float cs = 0.f, c_lo = -c.B;
    left = -c.B;
    size = 1.f;
    if (BY_VALUE) idx = 0;
#pragma unroll
    for (int k = 0; k < KM; ++k)
      if (k < K) {
        p[k] *= rs;
        cs += min_size + scale * p[k];
        const float c_hi = (k == K - 1) ? c.B : (2.0f * c.B * cs - c.B);
        const bool sel = BY_VALUE ? (v >= c_lo) : (k == idx);
        if (sel) {
          left = c_lo;
          size = c_hi - c_lo;
          if (BY_VALUE) idx = k;
        }
        c_lo = c_hi;
      }
  }
  // gradient wrt the raw logits of a family given dL/d(left knot) and dL/d(bin size)
  template <int OFF>
  static __device__ __forceinline__ void family_bwd(const NSplC& c, const float (&p)[KM], int idx, float L_left, float L_size,
                                                    float min_size, float (&dq)[NQV]) {
    const int K = c.K;
    const float Lc0 = L_left - L_size, Lc1 = L_size;
    const float f = 2.0f * c.B * (1.0f - min_size * (float)K);
    const bool c1_interior = idx <= K - 2;
    float S = 0.f;
#pragma unroll
    for (int i = 0; i < KM; ++i)
      if (i < K) {
        const float dp = f * ((i < idx ? Lc0 : 0.f) + ((c1_interior && i <= idx) ? Lc1 : 0.f));
        S += p[i] * dp;
      }
#pragma unroll
    for (int i = 0; i < KM; ++i)
      if (i < K) {
        const float dp = f * ((i < idx ? Lc0 : 0.f) + ((c1_interior && i <= idx) ? Lc1 : 0.f));
        dq[OFF + i] = p[i] * (dp - S) * c.inv_sqrt_h;
      }
  }
  // density direction only: out = spline(v), lad = log |d out / d v|
  static __device__ __forceinline__ void fwd(const NSplC& c, const float (&q)[NQV], float v, float& out, float& lad) {
    const int K = c.K;
    const bool inside = (v >= -c.B) && (v <= c.B);
    const float vc = fminf(fmaxf(v, -c.B), c.B);
    int idx = 0;
    float x_k, w_k, y_k, h_k;
    float pw[KM], ph[KM];
    family<0, true>(c, q, c.min_w, vc, idx, x_k, w_k, pw);
    family<KM, false>(c, q, c.min_h, vc, idx, y_k, h_k, ph);
    float r_k = c.dconst, r_k1 = c.dconst;
#pragma unroll
    for (int j = 1; j < KM; ++j)
      if (j < K) {
        const float rj = q[2 * KM + j - 1];
        r_k = (j == idx) ? rj : r_k;
        r_k1 = (j == idx + 1) ? rj : r_k1;
      }
    const float d_k = c.min_d + sf_softplus(r_k), d_k1 = c.min_d + sf_softplus(r_k1);
    const float s_k = sf_div(h_k, w_k);
    const float xi = sf_div(vc - x_k, w_k);
    const float om = xi * (1.f - xi);
    const float num = h_k * (s_k * xi * xi + d_k * om);
    const float den = s_k + (d_k + d_k1 - 2.f * s_k) * om;
    const float o_in = y_k + sf_div(num, den);
    const float dnum = s_k * s_k * (d_k1 * xi * xi + 2.f * s_k * om + d_k * (1.f - xi) * (1.f - xi));
    const float l_in = sf_log(dnum) - 2.f * sf_log(den);
    out = inside ? o_in : v;
    lad = inside ? l_in : 0.f;
  }
  // sampling direction: out = spline^{-1}(v), lad = log |d out / d v|  ([UPSTREAM] nflows rational_quadratic_spline(inverse=True):
  // bin search on the heights' knots, root 2c / (-b - sqrt(b^2 - 4ac)))
  static __device__ __forceinline__ void inv(const NSplC& c, const float (&q)[NQV], float v, float& out, float& lad) {
    const int K = c.K;
    const bool inside = (v >= -c.B) && (v <= c.B);
    const float vc = fminf(fmaxf(v, -c.B), c.B);
    int idx = 0;
    float x_k, w_k, y_k, h_k;
    float pw[KM], ph[KM];
    family<KM, true>(c, q, c.min_h, vc, idx, y_k, h_k, ph);
    family<0, false>(c, q, c.min_w, vc, idx, x_k, w_k, pw);
    float r_k = c.dconst, r_k1 = c.dconst;
#pragma unroll
    for (int j = 1; j < KM; ++j)
      if (j < K) {
        const float rj = q[2 * KM + j - 1];
        r_k = (j == idx) ? rj : r_k;
        r_k1 = (j == idx + 1) ? rj : r_k1;
      }
    const float d_k = c.min_d + sf_softplus(r_k), d_k1 = c.min_d + sf_softplus(r_k1);
    const float s_k = sf_div(h_k, w_k);
    const float dy = vc - y_k;
    const float tmp = dy * (d_k + d_k1 - 2.f * s_k);
    const float aa = tmp + h_k * (s_k - d_k);
    const float bb = h_k * d_k - tmp;
    const float cc = -s_k * dy;
    const float disc = bb * bb - 4.f * aa * cc;
    const float xi = sf_div(2.f * cc, -bb - __builtin_amdgcn_sqrtf(disc));
    const float o_in = xi * w_k + x_k;
    const float om = xi * (1.f - xi);
    const float den = s_k + (d_k + d_k1 - 2.f * s_k) * om;
    const float dnum = s_k * s_k * (d_k1 * xi * xi + 2.f * s_k * om + d_k * (1.f - xi) * (1.f - xi));
    const float l_in = -(sf_log(dnum) - 2.f * sf_log(den));
    out = inside ? o_in : v;
    lad = inside ? l_in : 0.f;
  }
  // L = Go * out + Gl * lad  ->  dv = dL/dv, dq[slot] = dL/d(raw parameter in that slot)
  static __device__ __forceinline__ void bwd(const NSplC& c, const float (&q)[NQV], float v, float Go, float Gl, float& dv,
                                             float (&dq)[NQV]) {
    const int K = c.K;
#pragma unroll
    for (int i = 0; i < NQV; ++i) dq[i] = 0.f;
    const bool inside = (v >= -c.B) && (v <= c.B);
    const float vc = fminf(fmaxf(v, -c.B), c.B);
    int idx = 0;
    float x_k, w_k, y_k, h_k;
    float pw[KM], ph[KM];
    family<0, true>(c, q, c.min_w, vc, idx, x_k, w_k, pw);
    family<KM, false>(c, q, c.min_h, vc, idx, y_k, h_k, ph);
    float r_k = c.dconst, r_k1 = c.dconst;
#pragma unroll
    for (int j = 1; j < KM; ++j)
      if (j < K) {
        const float rj = q[2 * KM + j - 1];
        r_k = (j == idx) ? rj : r_k;
        r_k1 = (j == idx + 1) ? rj : r_k1;
      }
    const float d_k = c.min_d + sf_softplus(r_k), d_k1 = c.min_d + sf_softplus(r_k1);
    const float inv_w = __builtin_amdgcn_rcpf(w_k);
    const float s = h_k * inv_w;
    const float xi = (vc - x_k) * inv_w;
    const float om = xi * (1.f - xi);
    const float A = d_k + d_k1 - 2.f * s;
    const float N = s * xi * xi + d_k * om;
    const float den = s + A * om;
    const float Mq = d_k1 * xi * xi + 2.f * s * om + d_k * (1.f - xi) * (1.f - xi);
    const float dnum = s * s * Mq;
    const float go = inside ? Go : 0.f, gl = inside ? Gl : 0.f;
    const float inv_den = __builtin_amdgcn_rcpf(den), inv_dnum = __builtin_amdgcn_rcpf(dnum);
    const float cN = go * h_k * inv_den;
    const float cD = -go * h_k * N * inv_den * inv_den - 2.f * gl * inv_den;
    const float cQ = gl * inv_dnum;
    const float L_s = cN * (xi * xi) + cD * (1.f - 2.f * om) + cQ * (2.f * s * Mq + 2.f * s * s * om);
    const float L_dk = cN * om + cD * om + cQ * (s * s * (1.f - xi) * (1.f - xi));
    const float L_dk1 = cD * om + cQ * (s * s * xi * xi);
    const float L_xi = cN * (2.f * s * xi + d_k * (1.f - 2.f * xi)) + cD * (A * (1.f - 2.f * xi)) +
                       cQ * (s * s * (2.f * d_k1 * xi + 2.f * s * (1.f - 2.f * xi) - 2.f * d_k * (1.f - xi)));
    const float L_y = go;
    const float L_h = go * N * inv_den + L_s * inv_w;
    const float L_w = -(L_s * s + L_xi * xi) * inv_w;
    const float L_x = -L_xi * inv_w;
    dv = inside ? L_xi * inv_w : Go;
    family_bwd<0>(c, pw, idx, L_x, L_w, c.min_w, dq);
    family_bwd<KM>(c, ph, idx, L_y, L_h, c.min_h, dq);
    const float g_k = L_dk * sf_sigmoid(r_k), g_k1 = L_dk1 * sf_sigmoid(r_k1);
#pragma unroll
    for (int j = 1; j < KM; ++j)
      if (j < K) dq[2 * KM + j - 1] = ((j == idx) ? g_k : 0.f) + ((j == idx + 1) ? g_k1 : 0.f);
  }
};


// ---------------------------------------------------------------------------------------------------------------------
// zuko's MonotonicRQSTransform (the univariate of zuko.flows.NSF, i.e. of `backend="lampe"`, ref: sbi_runner.py:5123-5125)
// on a thread-local parameter array -- same rational-quadratic bin arithmetic, another parametrisation: logits soft-clipped
// a / (1 + |2 a / log slope|) (derivatives: a / (1 + |a / log slope|)), softmax without a minimum bin size, knot derivatives
// exp(.) with 1 at both ends, identity outside (-B, B]; bin = searchsorted(knots, v) - 1 (the largest k with knot_k < v).
// Slots as above.  [UPSTREAM: restated from the published zuko sources.]
// ---------------------------------------------------------------------------------------------------------------------
struct ZSplC {
  int K;
  float B, cw, cd;  // cw = 2 / |log slope|, cd = 1 / |log slope|
};

template <int KM, int NQV>
struct ZSpl {
  static __device__ __forceinline__ float clip(float a, float c) { return a * __builtin_amdgcn_rcpf(1.f + c * fabsf(a)); }
  static __device__ __forceinline__ float dclip(float a, float c) {
    const float r = __builtin_amdgcn_rcpf(1.f + c * fabsf(a));
    return r * r;
  }
  template <int OFF, bool BY_VALUE>
  static __device__ __forceinline__ void family(const ZSplC& c, const float (&q)[NQV], float v, int& idx, float& left, float& size,
                                                float (&p)[KM]) {
    const int K = c.K;
    float mx = -3.0e38f;
#pragma unroll
    for (int k = 0; k < KM; ++k)
      if (k < K) {
        p[k] = clip(q[OFF + k], c.cw);
        mx = fmaxf(mx, p[k]);
      }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < KM; ++k)
      if (k < K) {
        p[k] = sf_exp(p[k] - mx);
        sum += p[k];
      }
    const float rs = __builtin_amdgcn_rcpf(sum);
    float cs = 0.f, c_lo = -c.B;
    left = -c.B;
    size = 1.f;
    if (BY_VALUE) idx = 0;
#pragma unroll
    for (int k = 0; k < KM; ++k)
      if (k < K) {
        p[k] *= rs;
        cs += p[k];
        const float c_hi = (k == K - 1) ? c.B : (2.0f * c.B * cs - c.B);
        const bool sel = BY_VALUE ? (v > c_lo) : (k == idx);
        if (sel) {
          left = c_lo;
          size = c_hi - c_lo;
          if (BY_VALUE) idx = k;
        }
        c_lo = c_hi;
      }
  }
  template <int OFF>
  static __device__ __forceinline__ void family_bwd(const ZSplC& c, const float (&q)[NQV], const float (&p)[KM], int idx, float L_left,
                                                    float L_size, float (&dq)[NQV]) {
    const int K = c.K;
    const float Lc0 = L_left - L_size, Lc1 = L_size;
    const float f = 2.0f * c.B;
    const bool c1_interior = idx <= K - 2;
    float S = 0.f;
#pragma unroll
    for (int i = 0; i < KM; ++i)
      if (i < K) {
        const float dp = f * ((i < idx ? Lc0 : 0.f) + ((c1_interior && i <= idx) ? Lc1 : 0.f));
        S += p[i] * dp;
      }
#pragma unroll
    for (int i = 0; i < KM; ++i)
      if (i < K) {
        const float dp = f * ((i < idx ? Lc0 : 0.f) + ((c1_interior && i <= idx) ? Lc1 : 0.f));
        dq[OFF + i] = p[i] * (dp - S) * dclip(q[OFF + i], c.cw);
      }
  }
  // the bin's two knot derivatives and the raw parameters behind them (raw = 0 and d = 1 at the two ends)
  static __device__ __forceinline__ void derivs(const ZSplC& c, const float (&q)[NQV], int idx, float& r_k, float& r_k1, float& d_k,
                                                float& d_k1) {
    const int K = c.K;
    r_k = 0.f;
    r_k1 = 0.f;
#pragma unroll
    for (int j = 1; j < KM; ++j)
      if (j < K) {
        const float rj = q[2 * KM + j - 1];
        r_k = (j == idx) ? rj : r_k;
        r_k1 = (j == idx + 1) ? rj : r_k1;
      }
    d_k = idx >= 1 ? sf_exp(clip(r_k, c.cd)) : 1.f;
    d_k1 = idx + 1 <= K - 1 ? sf_exp(clip(r_k1, c.cd)) : 1.f;
  }
  static __device__ __forceinline__ void fwd(const ZSplC& c, const float (&q)[NQV], float v, float& out, float& lad) {
    const bool inside = (v > -c.B) && (v <= c.B);
    const float vc = fminf(fmaxf(v, -c.B), c.B);
    int idx = 0;
    float x_k, w_k, y_k, h_k, r_k, r_k1, d_k, d_k1;
    float pw[KM], ph[KM];
    family<0, true>(c, q, vc, idx, x_k, w_k, pw);
    family<KM, false>(c, q, vc, idx, y_k, h_k, ph);
    derivs(c, q, idx, r_k, r_k1, d_k, d_k1);
    const float s_k = sf_div(h_k, w_k);
    const float xi = sf_div(vc - x_k, w_k);
    const float om = xi * (1.f - xi);
    const float den = s_k + (d_k + d_k1 - 2.f * s_k) * om;
    const float o_in = y_k + sf_div(h_k * (s_k * xi * xi + d_k * om), den);
    const float dnum = s_k * s_k * (d_k1 * xi * xi + 2.f * s_k * om + d_k * (1.f - xi) * (1.f - xi));
    out = inside ? o_in : v;
    lad = inside ? sf_log(dnum) - 2.f * sf_log(den) : 0.f;
  }
  static __device__ __forceinline__ void inv(const ZSplC& c, const float (&q)[NQV], float v, float& out, float& lad) {
    const bool inside = (v > -c.B) && (v <= c.B);
    const float vc = fminf(fmaxf(v, -c.B), c.B);
    int idx = 0;
    float x_k, w_k, y_k, h_k, r_k, r_k1, d_k, d_k1;
    float pw[KM], ph[KM];
    family<KM, true>(c, q, vc, idx, y_k, h_k, ph);
    family<0, false>(c, q, vc, idx, x_k, w_k, pw);
    derivs(c, q, idx, r_k, r_k1, d_k, d_k1);
    const float s_k = sf_div(h_k, w_k);
    const float dy = vc - y_k;
    const float tmp = dy * (d_k + d_k1 - 2.f * s_k);
    const float aa = tmp + h_k * (s_k - d_k);
    const float bb = h_k * d_k - tmp;
    const float cc = -s_k * dy;
    const float xi = sf_div(2.f * cc, -bb - __builtin_amdgcn_sqrtf(bb * bb - 4.f * aa * cc));
    const float om = xi * (1.f - xi);
    const float den = s_k + (d_k + d_k1 - 2.f * s_k) * om;
    const float dnum = s_k * s_k * (d_k1 * xi * xi + 2.f * s_k * om + d_k * (1.f - xi) * (1.f - xi));
    out = inside ? xi * w_k + x_k : v;
    lad = inside ? -(sf_log(dnum) - 2.f * sf_log(den)) : 0.f;
  }
  // L = Go * out + Gl * lad  ->  dv = dL/dv, dq[slot] = dL/d(raw parameter in that slot)
  static __device__ __forceinline__ void bwd(const ZSplC& c, const float (&q)[NQV], float v, float Go, float Gl, float& dv,
                                             float (&dq)[NQV]) {
    const int K = c.K;
#pragma unroll
    for (int i = 0; i < NQV; ++i) dq[i] = 0.f;
    const bool inside = (v > -c.B) && (v <= c.B);
    const float vc = fminf(fmaxf(v, -c.B), c.B);
    int idx = 0;
    float x_k, w_k, y_k, h_k, r_k, r_k1, d_k, d_k1;
    float pw[KM], ph[KM];
    family<0, true>(c, q, vc, idx, x_k, w_k, pw);
    family<KM, false>(c, q, vc, idx, y_k, h_k, ph);
    derivs(c, q, idx, r_k, r_k1, d_k, d_k1);
    const float inv_w = __builtin_amdgcn_rcpf(w_k);
    const float s = h_k * inv_w;
    const float xi = (vc - x_k) * inv_w;
    const float om = xi * (1.f - xi);
    const float A = d_k + d_k1 - 2.f * s;
    const float N = s * xi * xi + d_k * om;
    const float den = s + A * om;
    const float Mq = d_k1 * xi * xi + 2.f * s * om + d_k * (1.f - xi) * (1.f - xi);
    const float dnum = s * s * Mq;
    const float go = inside ? Go : 0.f, gl = inside ? Gl : 0.f;
    const float inv_den = __builtin_amdgcn_rcpf(den), inv_dnum = __builtin_amdgcn_rcpf(dnum);
    const float cN = go * h_k * inv_den;
    const float cD = -go * h_k * N * inv_den * inv_den - 2.f * gl * inv_den;
    const float cQ = gl * inv_dnum;
    const float L_s = cN * (xi * xi) + cD * (1.f - 2.f * om) + cQ * (2.f * s * Mq + 2.f * s * s * om);
    const float L_dk = cN * om + cD * om + cQ * (s * s * (1.f - xi) * (1.f - xi));
    const float L_dk1 = cD * om + cQ * (s * s * xi * xi);
    const float L_xi = cN * (2.f * s * xi + d_k * (1.f - 2.f * xi)) + cD * (A * (1.f - 2.f * xi)) +
                       cQ * (s * s * (2.f * d_k1 * xi + 2.f * s * (1.f - 2.f * xi) - 2.f * d_k * (1.f - xi)));
    const float L_y = go;
    const float L_h = go * N * inv_den + L_s * inv_w;
    const float L_w = -(L_s * s + L_xi * xi) * inv_w;
    const float L_x = -L_xi * inv_w;
    dv = inside ? L_xi * inv_w : Go;
    family_bwd<0>(c, q, pw, idx, L_x, L_w, dq);
    family_bwd<KM>(c, q, ph, idx, L_y, L_h, dq);
    const float g_k = idx >= 1 ? L_dk * d_k * dclip(r_k, c.cd) : 0.f;
    const float g_k1 = idx + 1 <= K - 1 ? L_dk1 * d_k1 * dclip(r_k1, c.cd) : 0.f;
#pragma unroll
    for (int j = 1; j < KM; ++j)
      if (j < K) dq[2 * KM + j - 1] = ((j == idx) ? g_k : 0.f) + ((j == idx + 1) ? g_k1 : 0.f);
  }
};
