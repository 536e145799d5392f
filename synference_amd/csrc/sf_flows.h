// sf_flows.h -- the two flow families on the register-tile engine.
//   MafOps: [UPSTREAM] nflows MaskedAffineAutoregressiveTransform + RandomPermutation
//           (SURVEY.md B.3), permutations folded into the weight image (sf_layout.cpp).
//   NsfOps: [UPSTREAM] PiecewiseRationalQuadraticCouplingTransform (ResidualNet conditioner,
//           GLU context gates) + LULinear.
// State per lane: u[ns][p] = value of physical slot p for the lane's sample of sample tile ns,
// replicated in both row halves; logdet[ns] replicated likewise.
#pragma once
#include "sf_device.h"

// =============================================================================================
// MAF
// =============================================================================================
template <int HT, int NS, bool LDSW = false, bool BF = false>
struct MafOps {
  // MADE: (a_p, m_p) for every slot p land in fin[0][ns][2*(p>>1)], [2*(p>>1)+1] on half p&1
  static __device__ __forceinline__ void made(const SfDev& m, const float* __restrict__ tp,
                                              const float (&u)[NS][SF_DMAX],
                                              const float* const (&xr)[NS], f32x16 (&fin)[1][NS],
                                              int lane, const unsigned short* __restrict__ tpB = nullptr,
                                              const f32x16 (*pre)[1][NS] = nullptr) {
    const int h = lane >> 5;
    f32x16 a[HT][NS];
    sf_init_bias<HT, NS>(a, tp + m.o_b0, h);
    {
      f32x16 ut[1][NS];
      sf_build_u_tile<NS>(ut, u, h);
      sf_mm_acc<HT, NS, 1, false>(a, ut, tp + m.o_w0, m.nGu, 0, m.nGu, lane);
    }
    sf_ctx_mm<HT, NS>(a, xr, m, tp + m.o_wc, lane, pre);
#pragma unroll
    for (int k = 0; k < SF_NBMAX; ++k) {
      if (k < m.NB) {
        f32x16 b[HT][NS];
        sf_init_bias<HT, NS>(b, tp + m.o_bk[k], h);
        if (BF)
          sf_mm_acc_bf16<HT, NS, HT, false, true>(b, a, tpB + m.oB_wk[k], m.nKS, m.nKS, lane,
                                                  SfKLim{{(m.mt_kend[0] + 1) >> 1, (m.mt_kend[1] + 1) >> 1,
                                                          (m.mt_kend[2] + 1) >> 1, (m.mt_kend[3] + 1) >> 1}});
        else
          sf_mm_acc<HT, NS, HT, false, true>(b, a, tp + m.o_wk[k], m.nGh, 0, m.nGh, lane,
                                             SfKLim{{m.mt_kend[0], m.mt_kend[1], m.mt_kend[2], m.mt_kend[3]}});
#pragma unroll
        for (int mt = 0; mt < HT; ++mt)
#pragma unroll
          for (int ns = 0; ns < NS; ++ns)
#pragma unroll
            for (int r = 0; r < 16; ++r) a[mt][ns][r] = sf_tanh(b[mt][ns][r]);
      }
    }
    sf_init_bias<1, NS>(fin, tp + m.o_bf, h);
    sf_mm_acc<1, NS, HT, false>(fin, a, tp + m.o_wf, m.nGh, 0, m.nGh, lane);
  }

  static __device__ __forceinline__ float scale(const SfDev& m, float a) {
    return (m.scale_fn == 0 ? sf_softplus(a) : sf_sigmoid(a + 2.0f)) + m.eps;
  }

  // density direction: u <- s*u + m for every transform, logdet += sum log s
  static __device__ __forceinline__ void forward(const SfDev& m, float (&u)[NS][SF_DMAX],
                                                 const float* const (&xr)[NS], float (&logdet)[NS],
                                                 int lane, float* lds = nullptr) {
    const int h = lane >> 5;
    // first standardised context tile, built once for all transforms (one sample tile per wave only: registers)
    f32x16 ct0[1][NS];
    if (NS == 1) sf_build_ctx_tile<NS>(ct0, xr, m, 0, lane >> 5);
    const f32x16 (*pre)[1][NS] = (NS == 1) ? &ct0 : nullptr;
    for (int t = 0; t < m.T; ++t) {
      const float* tp = sf_stage<LDSW>(m, t, lds);
      f32x16 fin[1][NS];
      made(m, tp, u, xr, fin, lane, sf_bf16_base<LDSW>(m, t, lds), pre);
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        float ld = 0.f;
#pragma unroll
        for (int p = 0; p < SF_DMAX; ++p) {
          if (p < m.D) {
            const float s = scale(m, fin[0][ns][2 * (p >> 1)]);
            const float val = s * u[ns][p] + fin[0][ns][2 * (p >> 1) + 1];
            const bool mine = (h == (p & 1));
            const float oth = sf_xhalf(val);
            u[ns][p] = mine ? val : oth;
            ld += mine ? sf_log(s) : 0.f;
          }
        }
        logdet[ns] += ld + sf_xhalf(ld);
      }
    }
  }

  // sampling direction: transforms in reverse, D MADE passes each (AutoregressiveTransform.inverse)
  static __device__ __forceinline__ void inverse_full(const SfDev& m, float (&u)[NS][SF_DMAX],
                                                      const float* const (&xr)[NS], float (&logdet)[NS],
                                                      int lane, float* lds = nullptr) {
    const int h = lane >> 5;
    // first standardised context tile, built once for all transforms (one sample tile per wave only: registers)
    f32x16 ct0[1][NS];
    if (NS == 1) sf_build_ctx_tile<NS>(ct0, xr, m, 0, lane >> 5);
    const f32x16 (*pre)[1][NS] = (NS == 1) ? &ct0 : nullptr;
    for (int t = m.T - 1; t >= 0; --t) {
      const float* tp = sf_stage<LDSW>(m, t, lds);
      float w[NS][SF_DMAX];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int p = 0; p < SF_DMAX; ++p) w[ns][p] = 0.f;
      float ldl[NS];
      for (int pass = 0; pass < m.D; ++pass) {
        f32x16 fin[1][NS];
        made(m, tp, w, xr, fin, lane, sf_bf16_base<LDSW>(m, t, lds), pre);
        affine_inverse(m, fin, u, w, ldl, h);
      }
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        logdet[ns] -= ldl[ns];
#pragma unroll
        for (int p = 0; p < SF_DMAX; ++p) u[ns][p] = w[ns][p];
      }
    }
  }

  // w <- (v - m)/s for every slot from the head tile; ldl = sum log s
  static __device__ __forceinline__ void affine_inverse(const SfDev& m, const f32x16 (&fin)[1][NS],
                                                        const float (&v)[NS][SF_DMAX], float (&w)[NS][SF_DMAX],
                                                        float (&ldl)[NS], int h) {
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      float ld = 0.f;
#pragma unroll
      for (int p = 0; p < SF_DMAX; ++p) {
        if (p < m.D) {
          const float s = scale(m, fin[0][ns][2 * (p >> 1)]);
          const float val = sf_div(v[ns][p] - fin[0][ns][2 * (p >> 1) + 1], s);
          const bool mine = (h == (p & 1));
          const float oth = sf_xhalf(val);
          w[ns][p] = mine ? val : oth;
          ld += mine ? sf_log(s) : 0.f;
        }
      }
      ldl[ns] = ld + sf_xhalf(ld);
    }
  }

  // Incremental form of the same D passes.  Hidden units are stored sorted by MADE degree with each
  // degree group inside one tile (sf_layout.cpp), so pass p only has to (re)compute the ONE hidden
  // tile holding the units of degree p-1 -- their inputs became final in pass p-1 -- over the input
  // groups of degree <= p-1; everything of lower degree is already final and is kept in registers.
  // Masked weights are structural zeros, so every value that the D full passes would end with is
  // produced by the identical fma chain: results are bit-identical to inverse_full.
  // The context product (b0 + bc + Wc e) is hoisted out of the passes; the two head rows needed per
  // pass are VALU dot products (so the head differs from inverse_full by fp32 summation order only).
  static __device__ __forceinline__ void inverse_incremental(const SfDev& m, float (&u)[NS][SF_DMAX],
                                                             const float* const (&xr)[NS],
                                                             float (&logdet)[NS], int lane, float* lds = nullptr) {
    const int h = lane >> 5;
    // first standardised context tile, built once for all transforms (one sample tile per wave only: registers)
    f32x16 ct0[1][NS];
    if (NS == 1) sf_build_ctx_tile<NS>(ct0, xr, m, 0, lane >> 5);
    const f32x16 (*pre)[1][NS] = (NS == 1) ? &ct0 : nullptr;
    for (int t = m.T - 1; t >= 0; --t) {
      const float* tp = sf_stage<LDSW>(m, t, lds);
      f32x16 c0[HT][NS];
      sf_init_bias<HT, NS>(c0, tp + m.o_b0, h);
      sf_ctx_mm<HT, NS>(c0, xr, m, tp + m.o_wc, lane, pre);
      f32x16 act[3][HT][NS];  // act[0] = initial layer, act[k+1] = output of block k (NB <= 2)
#pragma unroll
      for (int k = 0; k <= 2; ++k)
#pragma unroll
        for (int mt = 0; mt < HT; ++mt)
#pragma unroll
          for (int ns = 0; ns < NS; ++ns)
#pragma unroll
            for (int r = 0; r < 16; ++r) act[k][mt][ns][r] = 0.f;
      float w[NS][SF_DMAX];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int p = 0; p < SF_DMAX; ++p) w[ns][p] = 0.f;
      float ldl[NS];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) ldl[ns] = 0.f;
      for (int p = 1; p <= m.D; ++p) {
        if (p >= 2) {
          // the degree group of this pass occupies tiles lo..hi (one tile when the layout is aligned, several when
          // the units are packed contiguously); every layer is recomputed for all of them before the next layer
          // starts -- units of one degree feed each other -- over the input groups of degree <= p-1
          const int hi = m.g_tile[p - 1], lo = m.g_lo[p - 1];
          const int kend = m.g_kend[p - 1];
          f32x16 ut[1][NS];
          sf_build_u_tile<NS>(ut, w, h);
#pragma unroll
          for (int mt = 0; mt < HT; ++mt) {
            if (mt >= lo && mt <= hi) {
#pragma unroll
              for (int ns = 0; ns < NS; ++ns) act[0][mt][ns] = c0[mt][ns];
              sf_mm_acc_tile<NS, 1, false>(act[0][mt], ut, tp + m.o_w0, m.nGu, mt, m.nGu, lane);
            }
          }
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            if (k < m.NB) {
#pragma unroll
              for (int mt = 0; mt < HT; ++mt) {
                if (mt >= lo && mt <= hi) {
                  f32x16 b[NS];
                  sf_init_bias_tile<NS>(b, tp + m.o_bk[k], mt, h);
                  if (BF)
                    sf_mm_acc_bf16_tile<NS, HT, false>(b, act[k], sf_bf16_base<LDSW>(m, t, lds) + m.oB_wk[k], m.nKS, mt,
                                                       (kend + 1) >> 1, lane);
                  else
                    sf_mm_acc_tile<NS, HT, false>(b, act[k], tp + m.o_wk[k], m.nGh, mt, kend, lane);
#pragma unroll
                  for (int ns = 0; ns < NS; ++ns)
#pragma unroll
                    for (int r = 0; r < 16; ++r) act[k + 1][mt][ns][r] = sf_tanh(b[ns][r]);
                }
              }
            }
          }
        }
        // head: only the (a, m) rows of the dimension that becomes final in this pass are needed, so
        // instead of a 32-row MFMA tile they are two dot products over the lane's own activation
        // registers (each row half holds half of the hidden units; weights broadcast from the image)
        const int slot = (int)m.cst[m.c_dslot + t * SF_DMAX + (p - 1)];
        const float* hv = tp + m.o_hv + (size_t)slot * 4 * (HT * 16) + h * (HT * 16);
        float pa[NS], pm[NS];
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) pa[ns] = pm[ns] = 0.f;
        if (p >= 2) {
#pragma unroll
          for (int mt = 0; mt < HT; ++mt)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
              const float4 wa = *reinterpret_cast<const float4*>(hv + mt * 16 + q4 * 4);
              const float4 wm = *reinterpret_cast<const float4*>(hv + 2 * (HT * 16) + mt * 16 + q4 * 4);
#pragma unroll
              for (int ns = 0; ns < NS; ++ns) {
                const f32x16& av = (m.NB == 1) ? act[1][mt][ns] : act[2][mt][ns];
                pa[ns] += wa.x * av[q4 * 4] + wa.y * av[q4 * 4 + 1] + wa.z * av[q4 * 4 + 2] + wa.w * av[q4 * 4 + 3];
                pm[ns] += wm.x * av[q4 * 4] + wm.y * av[q4 * 4 + 1] + wm.z * av[q4 * 4 + 2] + wm.w * av[q4 * 4 + 3];
              }
            }
        }
        const float ba = tp[m.o_hvb + 2 * slot], bm = tp[m.o_hvb + 2 * slot + 1];
#pragma unroll
        for (int q = 0; q < SF_DMAX; ++q) {
          if (q == slot) {
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) {
              const float av = ba + pa[ns] + sf_xhalf(pa[ns]);
              const float mv = bm + pm[ns] + sf_xhalf(pm[ns]);
              const float s = scale(m, av);
              w[ns][q] = sf_div(u[ns][q] - mv, s);
              ldl[ns] += sf_log(s);
            }
          }
        }
      }
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        logdet[ns] -= ldl[ns];
#pragma unroll
        for (int p = 0; p < SF_DMAX; ++p) u[ns][p] = w[ns][p];
      }
    }
  }

  static __device__ __forceinline__ void inverse(const SfDev& m, float (&u)[NS][SF_DMAX],
                                                 const float* const (&xr)[NS], float (&logdet)[NS],
                                                 int lane, float* lds = nullptr,
                                                 const float* const (*cg)[NS] = nullptr,  // (no table on this path)
                                                 bool active = true) {                    // (MAF: every wave runs)
    if (m.inc_ok && m.NB <= 2) inverse_incremental(m, u, xr, logdet, lane, lds);
    else inverse_full(m, u, xr, logdet, lane, lds);
  }
};

// =============================================================================================
// NSF
// =============================================================================================
// One rational-quadratic spline evaluation with linear tails
// ([UPSTREAM] nflows unconstrained_rational_quadratic_spline).  Parameter slots of the PT tiles:
// widths [0,KM), heights [KM,2KM), interior derivatives [2KM, 3KM-1), KM = (16*PT+1)/3.
template <int PT>
struct SfSpline {
  static constexpr int KM = (PT * 16 + 1) / 3;

  template <int NS>
  static __device__ __forceinline__ float Q(const f32x16 (&q)[PT][NS], int ns, int s) {
    return q[s >> 4][ns][s & 15];
  }

  // softmax-normalised bin sizes of one parameter family -> knots; selects the bin
  //   BY_VALUE: largest k with v >= knot_k (searchsorted semantics incl. the +1e-6 on the last knot)
  //   else    : k == idx
  template <int NS, int OFF, bool BY_VALUE>
  static __device__ __forceinline__ void knots(const SfDev& m, const f32x16 (&q)[PT][NS], int ns,
                                               float min_size, float v, int& idx, float& left,
                                               float& size) {
    const int K = m.K;
    const float B = m.tail_bound;
    float e[KM];
    float mx = -3.0e38f;
#pragma unroll
    for (int k = 0; k < KM; ++k)
      if (k < K) {
        e[k] = Q<NS>(q, ns, OFF + k) * m.inv_sqrt_h;
        mx = fmaxf(mx, e[k]);
      }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < KM; ++k)
      if (k < K) {
        e[k] = sf_exp(e[k] - mx);
        sum += e[k];
      }
    const float scale = (1.0f - min_size * (float)K) * __builtin_amdgcn_rcpf(sum);  // (one reciprocal per family)
    float cs = 0.f, c_lo = -B;
    left = -B;
    size = 1.f;
    if (BY_VALUE) idx = 0;
#pragma unroll
    for (int k = 0; k < KM; ++k)
      if (k < K) {
        cs += min_size + scale * e[k];
        const float c_hi = (k == K - 1) ? B : (2.0f * B * cs - B);
        const bool sel = BY_VALUE ? (v >= c_lo) : (k == idx);
        if (sel) {
          left = c_lo;
          size = c_hi - c_lo;
          if (BY_VALUE) idx = k;
        }
        c_lo = c_hi;
      }
  }

  template <int NS>
  static __device__ __forceinline__ void eval(const SfDev& m, const f32x16 (&q)[PT][NS], int ns,
                                              float v, bool inverse, float& out, float& lad) {
    const int K = m.K;
    const float B = m.tail_bound;
    const bool inside = (v >= -B) && (v <= B);
    const float vc = fminf(fmaxf(v, -B), B);
    int idx = 0;
    float x_k, w_k, y_k, h_k;
    if (!inverse) {
      knots<NS, 0, true>(m, q, ns, m.min_w, vc, idx, x_k, w_k);
      knots<NS, KM, false>(m, q, ns, m.min_h, vc, idx, y_k, h_k);
    } else {
      knots<NS, KM, true>(m, q, ns, m.min_h, vc, idx, y_k, h_k);
      knots<NS, 0, false>(m, q, ns, m.min_w, vc, idx, x_k, w_k);
    }
    // derivatives at the bin's two knots (boundary knots use the padded constant): pick the two raw values first,
    // then ONE softplus each -- not a softplus (exp + log) per interior knot of which K-3 are thrown away
    float r_k = m.deriv_const, r_k1 = m.deriv_const;
#pragma unroll
    for (int j = 1; j < KM; ++j)
      if (j < K) {
        const float rj = Q<NS>(q, ns, 2 * KM + j - 1);
        r_k = (j == idx) ? rj : r_k;
        r_k1 = (j == idx + 1) ? rj : r_k1;
      }
    const float d_k = m.min_d + sf_softplus(r_k), d_k1 = m.min_d + sf_softplus(r_k1);
    const float s_k = sf_div(h_k, w_k);
    float xi;
    if (!inverse) {
      xi = sf_div(vc - x_k, w_k);
      const float om = xi * (1.f - xi);
      const float num = h_k * (s_k * xi * xi + d_k * om);
      const float den = s_k + (d_k + d_k1 - 2.f * s_k) * om;
      out = y_k + sf_div(num, den);
    } else {
      const float dy = vc - y_k;
      const float tmp = dy * (d_k + d_k1 - 2.f * s_k);
      const float a = tmp + h_k * (s_k - d_k);
      const float b = h_k * d_k - tmp;
      const float c = -s_k * dy;
      const float disc = b * b - 4.f * a * c;
      xi = sf_div(2.f * c, -b - __builtin_amdgcn_sqrtf(disc));
      out = xi * w_k + x_k;
    }
    const float om = xi * (1.f - xi);
    const float den = s_k + (d_k + d_k1 - 2.f * s_k) * om;
    const float dnum = s_k * s_k * (d_k1 * xi * xi + 2.f * s_k * om + d_k * (1.f - xi) * (1.f - xi));
    lad = sf_log(dnum) - 2.f * sf_log(den);
    if (inverse) lad = -lad;
    out = inside ? out : v;
    lad = inside ? lad : 0.f;
  }
};

// Forward + backward of one spline evaluation (density direction), hand-derived; the numpy twin
// is tests/spline_bwd_model.py (checked against torch.autograd on the oracle).
//   L = Go*out + Gl*lad  ->  dv = dL/dv, dq[slot] = dL/d(raw parameter in that slot)
template <int PT>
struct SfSplineBwd {
  static constexpr int KM = SfSpline<PT>::KM;

  // one parameter family (widths or heights): softmax -> knots, bin select, and the pieces needed
  // to back-propagate (p_k kept in e[], idx)
  template <int NS, int OFF, bool BY_VALUE>
  static __device__ __forceinline__ void family_fwd(const SfDev& m, const f32x16 (&q)[PT][NS], int ns,
                                                    float min_size, float v, int& idx, float& left,
                                                    float& size, float (&e)[KM]) {
    const int K = m.K;
    const float B = m.tail_bound;
    float mx = -3.0e38f;
#pragma unroll
    for (int k = 0; k < KM; ++k)
      if (k < K) {
        e[k] = SfSpline<PT>::template Q<NS>(q, ns, OFF + k) * m.inv_sqrt_h;
        mx = fmaxf(mx, e[k]);
      }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < KM; ++k)
      if (k < K) {
        e[k] = sf_exp(e[k] - mx);
        sum += e[k];
      }
    const float scale = (1.0f - min_size * (float)K);
    float cs = 0.f, c_lo = -B;
    left = -B;
    size = 1.f;
    if (BY_VALUE) idx = 0;
#pragma unroll
    for (int k = 0; k < KM; ++k)
      if (k < K) {
        e[k] = sf_div(e[k], sum);  // p_k
        cs += min_size + scale * e[k];
        const float c_hi = (k == K - 1) ? B : (2.0f * B * cs - B);
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
  template <int NS, int OFF>
  static __device__ __forceinline__ void family_bwd(const SfDev& m, const float (&p)[KM], int idx, float L_left,
                                                    float L_size, float min_size, f32x16 (&dq)[PT][NS], int ns) {
    const int K = m.K;
    const float Lc0 = L_left - L_size, Lc1 = L_size;
    const float f = 2.0f * m.tail_bound * (1.0f - min_size * (float)K);
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
        dq[(OFF + i) >> 4][ns][(OFF + i) & 15] = p[i] * (dp - S) * m.inv_sqrt_h;
      }
  }

  template <int NS>
  static __device__ __forceinline__ void eval(const SfDev& m, const f32x16 (&q)[PT][NS], int ns, float v,
                                              float Go, float Gl, float& out, float& lad, float& dv,
                                              f32x16 (&dq)[PT][NS]) {
    const int K = m.K;
    const float B = m.tail_bound;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[pt][ns][r] = 0.f;
    const bool inside = (v >= -B) && (v <= B);
    const float vc = fminf(fmaxf(v, -B), B);
    int idx = 0;
    float x_k, w_k, y_k, h_k;
    float pw[KM], ph[KM];
    family_fwd<NS, 0, true>(m, q, ns, m.min_w, vc, idx, x_k, w_k, pw);
    family_fwd<NS, KM, false>(m, q, ns, m.min_h, vc, idx, y_k, h_k, ph);
    float r_k = m.deriv_const, r_k1 = m.deriv_const;  // raw derivative parameters of the bin's two knots
#pragma unroll
    for (int j = 1; j < KM; ++j)
      if (j < K) {
        const float rj = SfSpline<PT>::template Q<NS>(q, ns, 2 * KM + j - 1);
        r_k = (j == idx) ? rj : r_k;
        r_k1 = (j == idx + 1) ? rj : r_k1;
      }
    const float d_k = m.min_d + sf_softplus(r_k), d_k1 = m.min_d + sf_softplus(r_k1);
    const float s = h_k / w_k;
    const float xi = (vc - x_k) / w_k;
    const float om = xi * (1.f - xi);
    const float A = d_k + d_k1 - 2.f * s;
    const float N = s * xi * xi + d_k * om;
    const float den = s + A * om;
    const float Mq = d_k1 * xi * xi + 2.f * s * om + d_k * (1.f - xi) * (1.f - xi);
    const float dnum = s * s * Mq;
    const float o_in = y_k + h_k * N / den;
    const float l_in = sf_log(dnum) - 2.f * sf_log(den);
    out = inside ? o_in : v;
    lad = inside ? l_in : 0.f;
    // partials wrt z in {s, d_k, d_k1, xi}
    const float go = inside ? Go : 0.f, gl = inside ? Gl : 0.f;
    const float inv_den = 1.f / den, inv_dnum = 1.f / dnum;
    const float cN = go * h_k * inv_den;             // coefficient of N_z
    const float cD = -go * h_k * N * inv_den * inv_den - 2.f * gl * inv_den;  // coefficient of D_z
    const float cQ = gl * inv_dnum;                  // coefficient of Q_z
    const float L_s = cN * (xi * xi) + cD * (1.f - 2.f * om) + cQ * (2.f * s * Mq + 2.f * s * s * om);
    const float L_dk = cN * om + cD * om + cQ * (s * s * (1.f - xi) * (1.f - xi));
    const float L_dk1 = cD * om + cQ * (s * s * xi * xi);
    const float L_xi = cN * (2.f * s * xi + d_k * (1.f - 2.f * xi)) + cD * (A * (1.f - 2.f * xi)) +
                       cQ * (s * s * (2.f * d_k1 * xi + 2.f * s * (1.f - 2.f * xi) - 2.f * d_k * (1.f - xi)));
    const float inv_w = 1.f / w_k;
    const float L_y = go;
    const float L_h = go * N * inv_den + L_s * inv_w;
    const float L_w = -(L_s * s + L_xi * xi) * inv_w;
    const float L_x = -L_xi * inv_w;
    dv = inside ? L_xi * inv_w : Go;
    family_bwd<NS, 0>(m, pw, idx, L_x, L_w, m.min_w, dq, ns);
    family_bwd<NS, KM>(m, ph, idx, L_y, L_h, m.min_h, dq, ns);
    // d softplus / d raw = sigmoid(raw): only the bin's two knots carry gradient -- two sigmoids, not K-1
    const float g_k = L_dk * sf_sigmoid(r_k), g_k1 = L_dk1 * sf_sigmoid(r_k1);
#pragma unroll
    for (int j = 1; j < KM; ++j)
      if (j < K)
        dq[(2 * KM + j - 1) >> 4][ns][(2 * KM + j - 1) & 15] = ((j == idx) ? g_k : 0.f) + ((j == idx + 1) ? g_k1 : 0.f);
  }
};

// BF: 0 = fp32 hidden blocks; 1 = single bf16 operands (sf_flow_desc.hidden_bf16); 2 = split bf16 x3 (sampler image)
// MP: the image of a transform is staged in several PARTS (split sampler image of a wide flow: sf_layout.cpp, SfNsfSamp).  A template
// argument, not a run-time test: with the part switch inside the block loop every persistent sampler spilled ~100 B per lane more
// (staging loops and barriers between the blocks cut every live range), also where a transform is ONE part -- the usual case.
template <int HT, int PT, int NS, bool LDSW = false, int BF = 0, bool MP = false>
struct NsfOps {
  // ResidualNet conditioner -> hidden tiles
  // returns the (possibly LDS) base pointer valid for the spline head
  static __device__ __forceinline__ const float* resnet(const SfDev& m, int t, float* lds, const float* tp0,
                                                        const float (&u)[NS][SF_DMAX],
                                                        const float* const (&xr)[NS], f32x16 (&hid)[HT][NS],
                                                        int lane, const f32x16 (*pre)[1][NS] = nullptr,
                                                        const float* const (*cg)[NS] = nullptr, bool active = true) {
    // active (wave-uniform): false for a wave of the persistent sampler whose tile holds no item -- it only takes part
    // in the staging barriers, its issue slots go to the waves that have work
    const int h = lane >> 5;
    int part = 0;
    const float* tp = tp0;  // part 0 was staged by the caller
    if (active) {
    if (cg) sf_ctab_load<HT, NS>(hid, *cg, (t * m.ctab_NV) * m.ctab_R, h);  // bin + Win_c e(x) from the galaxy table
    else sf_init_bias<HT, NS>(hid, tp + m.o_bin, h);
    {
      f32x16 ut[1][NS];
      sf_build_u_tile<NS>(ut, u, h);
      sf_mm_acc<HT, NS, 1, false>(hid, ut, tp + m.o_winu, m.nGu, 0, m.nGu, lane);
    }
    if (!cg) sf_ctx_mm<HT, NS>(hid, xr, m, tp + m.o_winc, lane, pre);
    }
#pragma unroll
    for (int k = 0; k < SF_NBMAX; ++k) {
      if (k < m.NB) {
        if (MP && LDSW && m.blk_part[k] != part) {
          part = m.blk_part[k];
          tp = sf_stage_part<LDSW>(m, t, part, lds);
        }
        if (!active) continue;
        f32x16 t2[HT][NS];
        {
          f32x16 t1[HT][NS];
          sf_init_bias<HT, NS>(t1, tp + m.o_b1[k], h);
          sf_init_bias<HT, NS>(t2, tp + m.o_b2[k], h);
          if (BF == 2) {
            const unsigned short* tpB = sf_bf16_base<LDSW>(m, t, lds, MP ? part : 0);
            sf_mm_acc_bf16_split<HT, NS, HT, true>(t1, hid, tpB + m.oB_w1[k], m.nKS, m.nKS, lane);
            sf_mm_acc_bf16_split<HT, NS, HT, true>(t2, t1, tpB + m.oB_w2[k], m.nKS, m.nKS, lane);
          } else if (BF == 1) {
            const unsigned short* tpB = sf_bf16_base<LDSW>(m, t, lds);
            sf_mm_acc_bf16<HT, NS, HT, true>(t1, hid, tpB + m.oB_w1[k], m.nKS, m.nKS, lane);
            sf_mm_acc_bf16<HT, NS, HT, true>(t2, t1, tpB + m.oB_w2[k], m.nKS, m.nKS, lane);
          } else {
            sf_mm_acc<HT, NS, HT, true>(t1, hid, tp + m.o_w1[k], m.nGh, 0, m.nGh, lane);
            sf_mm_acc<HT, NS, HT, true>(t2, t1, tp + m.o_w2[k], m.nGh, 0, m.nGh, lane);
          }
        }
        // GLU gate, one output tile at a time: hid += t2 * sigmoid(Wg e + bg)
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) {
          f32x16 g[1][NS];
          if (cg) {
            sf_ctab_load<1, NS>(g, *cg, (t * m.ctab_NV + 1 + k) * m.ctab_R + mt * 32, h);
          } else {
            sf_init_bias<1, NS>(g, tp + m.o_bg[k] + mt * 32, h);
            sf_ctx_mm<1, NS>(g, xr, m, tp + m.o_wg[k] + mt * m.nGc * 256, lane, pre);
          }
#pragma unroll
          for (int ns = 0; ns < NS; ++ns)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              hid[mt][ns][r] += t2[mt][ns][r] * sf_sigmoid(g[0][ns][r]);
        }
      }
    }
    if (MP && LDSW && m.head_part != part) tp = sf_stage_part<LDSW>(m, t, m.head_part, lds);
    return tp;
  }

  // spline head + RQ spline on the transform dims of parity t&1, given the conditioner state
  static __device__ __forceinline__ void spline_apply(const SfDev& m, const float* __restrict__ tp, int t,
                                                      const f32x16 (&hid)[HT][NS], float (&u)[NS][SF_DMAX],
                                                      float (&logdet)[NS], bool inverse, int lane) {
    const int h = lane >> 5;
    const int start = t & 1;
    const int d_tr = (m.D - start + 1) / 2;
    for (int jp = 0; jp * 2 < d_tr; ++jp) {
      f32x16 q[PT][NS];
      sf_init_bias<PT, NS>(q, tp + m.o_bout + jp * PT * 32, h);
      sf_mm_acc<PT, NS, HT, false>(q, hid, tp + m.o_wout + jp * PT * m.nGh * 256, m.nGh, 0, m.nGh, lane);
      // the two row halves evaluate the two transform dims of this pair: half 0 the dim in physical slot
      // sA, half 1 the one in slot sB (wave-uniform slot numbers -> scalar compares, no per-lane masks)
      const int sA = start + 4 * jp, sB = sA + 2;
      const bool haveA = 2 * jp < d_tr, haveB = 2 * jp + 1 < d_tr;
      const bool have = h == 0 ? haveA : haveB;
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        float uA = 0.f, uB = 0.f;
#pragma unroll
        for (int p = 0; p < SF_DMAX; ++p) {
          uA = (p == sA) ? u[ns][p] : uA;
          uB = (p == sB) ? u[ns][p] : uB;
        }
        const float vin = h == 0 ? uA : uB;
        float vout, lad;
        SfSpline<PT>::template eval<NS>(m, q, ns, vin, inverse, vout, lad);
        lad = have ? lad : 0.f;
        const float vo = sf_xhalf(vout);
        const float nA = h == 0 ? vout : vo, nB = h == 0 ? vo : vout;
#pragma unroll
        for (int p = 0; p < SF_DMAX; ++p) {
          u[ns][p] = (haveA && p == sA) ? nA : u[ns][p];
          u[ns][p] = (haveB && p == sB) ? nB : u[ns][p];
        }
        logdet[ns] += lad + sf_xhalf(lad);
      }
    }
  }

  // coupling: conditioner on the identity dims, spline on the others
  static __device__ __forceinline__ void coupling(const SfDev& m, int t, float* lds, const float* tp0,
                                                  float (&u)[NS][SF_DMAX], const float* const (&xr)[NS],
                                                  float (&logdet)[NS], bool inverse, int lane,
                                                  const f32x16 (*pre)[1][NS] = nullptr,
                                                  const float* const (*cg)[NS] = nullptr, bool active = true) {
    f32x16 hid[HT][NS];
    const float* tp = resnet(m, t, lds, tp0, u, xr, hid, lane, pre, cg, active);
    if (active) spline_apply(m, tp, t, hid, u, logdet, inverse, lane);
  }

  // LULinear:  y = L (U u) + b ;  diag(U) = softplus(udiag) + eps
  static __device__ __forceinline__ void lu_forward(const SfDev& m, const float* __restrict__ lp,
                                                    float (&u)[NS][SF_DMAX], float (&logdet)[NS]) {
    const int D = m.D;
    const float* Lm = lp;
    const float* Um = lp + D * D;
    const float* ud = lp + 2 * D * D;
    const float* bb = ud + D;
    float ld = 0.f;
    float t[NS][SF_DMAX];
#pragma unroll
    for (int i = 0; i < SF_DMAX; ++i)
      if (i < D) {
        const float dg = sf_softplus(ud[i]) + m.lu_eps;
        ld += sf_log(dg);
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) t[ns][i] = dg * u[ns][i];
#pragma unroll
        for (int j = 0; j < SF_DMAX; ++j)
          if (j > i && j < D) {
            const float w = Um[i * D + j];
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) t[ns][i] += w * u[ns][j];
          }
      }
#pragma unroll
    for (int i = 0; i < SF_DMAX; ++i)
      if (i < D) {
        const float b = bb[i];
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) u[ns][i] = t[ns][i] + b;
#pragma unroll
        for (int j = 0; j < SF_DMAX; ++j)
          if (j < i) {
            const float w = Lm[i * D + j];
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) u[ns][i] += w * t[ns][j];
          }
      }
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) logdet[ns] += ld;
  }

  // inverse: u = U^{-1} L^{-1} (y - b)
  static __device__ __forceinline__ void lu_inverse(const SfDev& m, const float* __restrict__ lp,
                                                    float (&u)[NS][SF_DMAX], float (&logdet)[NS]) {
    const int D = m.D;
    const float* Lm = lp;
    const float* Um = lp + D * D;
    const float* ud = lp + 2 * D * D;
    const float* bb = ud + D;
    float ld = 0.f;
    float t[NS][SF_DMAX];
    // forward substitution with unit-lower L
#pragma unroll
    for (int i = 0; i < SF_DMAX; ++i)
      if (i < D) {
        const float b = bb[i];
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) t[ns][i] = u[ns][i] - b;
#pragma unroll
        for (int j = 0; j < SF_DMAX; ++j)
          if (j < i) {
            const float w = Lm[i * D + j];
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) t[ns][i] -= w * t[ns][j];
          }
      }
    // back substitution with U
#pragma unroll
    for (int ii = 0; ii < SF_DMAX; ++ii) {
      const int i = SF_DMAX - 1 - ii;
      if (i < D) {
        const float dg = sf_softplus(ud[i]) + m.lu_eps;
        ld += sf_log(dg);
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) u[ns][i] = t[ns][i];
#pragma unroll
        for (int j = 0; j < SF_DMAX; ++j)
          if (j > i && j < D) {
            const float w = Um[i * D + j];
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) u[ns][i] -= w * u[ns][j];
          }
        const float rdg = __builtin_amdgcn_rcpf(dg);
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) u[ns][i] *= rdg;
      }
    }
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) logdet[ns] -= ld;
  }

  static __device__ __forceinline__ void forward(const SfDev& m0, float (&u)[NS][SF_DMAX],
                                                 const float* const (&xr)[NS], float (&logdet)[NS],
                                                 int lane, float* lds = nullptr) {
    // first standardised context tile, built once for all transforms (one sample tile per wave only: registers)
    f32x16 ct0[1][NS];
    if (NS == 1) sf_build_ctx_tile<NS>(ct0, xr, m0, 0, lane >> 5);
    const f32x16 (*pre)[1][NS] = (NS == 1) ? &ct0 : nullptr;
    for (int t = 0; t < m0.T; ++t) {
      const SfDev m = sf_iter_view(m0);
      const float* tp0 = sf_stage_part<LDSW>(m, t, 0, lds);
      coupling(m, t, lds, tp0, u, xr, logdet, false, lane, pre);
      // LU parameters: from the staged image when the whole transform is one part, else from global
      // (two call sites, not a pointer select, so each keeps its address space)
      if (m.D > 1) {
        if (LDSW && (!MP || m.n_parts == 1)) lu_forward(m, tp0 + m.o_lu, u, logdet);
        else lu_forward(m, m.packed + (size_t)t * m.t_stride + m.o_lu, u, logdet);
      }
    }
  }
  static __device__ __forceinline__ void inverse(const SfDev& m0, float (&u)[NS][SF_DMAX],
                                                 const float* const (&xr)[NS], float (&logdet)[NS],
                                                 int lane, float* lds = nullptr,
                                                 const float* const (*cg)[NS] = nullptr, bool active = true) {
    // first standardised context tile, built once for all transforms (one sample tile per wave only: registers)
    f32x16 ct0[1][NS];
    if (NS == 1) sf_build_ctx_tile<NS>(ct0, xr, m0, 0, lane >> 5);
    const f32x16 (*pre)[1][NS] = (NS == 1) ? &ct0 : nullptr;
    for (int t = m0.T - 1; t >= 0; --t) {
      const SfDev m = sf_iter_view(m0);
      const float* tp0 = sf_stage_part<LDSW>(m, t, 0, lds);
      if (active && m.D > 1) {
        if (LDSW && (!MP || m.n_parts == 1)) lu_inverse(m, tp0 + m.o_lu, u, logdet);
        else lu_inverse(m, m.packed + (size_t)t * m.t_stride + m.o_lu, u, logdet);
      }
      coupling(m, t, lds, tp0, u, xr, logdet, true, lane, pre, cg, active);
    }
  }
};
