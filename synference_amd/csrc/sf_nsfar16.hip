// sf_nsfar16.hip -- the rejection sampler of the autoregressive flows (`backend="lampe"`: zuko NSF / MAF) on 16-sample REGISTER tiles.
// ref: src/synference/sbi_runner.py:5123-5125 (load_nde_lampe), 6438-6442 (posterior.sample per object); [UPSTREAM] zuko
// MaskedAutoregressiveTransform, restated in oracle/flows.py ("nsf_ar" / "maf_ar", parity unpinned).
//
// Why (round 5).  k_ar_sample (sf_nsfar.hip) keeps a wave's 64 samples in LDS rows -- 56 KB for the bench shape: TWO waves per CU,
// each walking D x T dependent steps of three tile passes and a spline with every L2 / LDS round trip exposed (0.03 of the fp32
// MFMA roof).  Here a wave owns SIXTEEN candidates and nothing of them lives in LDS:
//   * activations are 16 x 16 tiles in the accumulator layout of v_mfma_f32_16x16x4_f32 (lane l: rows 4 (l >> 4) + 0..3, sample
//     l & 15), which IS the B-operand order of the next product when the weight block is stored lane-major with the same
//     k-permutation (lane l: W[16 ot + (l & 15)][16 kt + 4 (l >> 4) + j], one 16-byte load per block, four products);
//   * the sampler has its own hidden order: type r (the units that become final once the dimensions ordered before r are known)
//     IS tile r (tiles 2 r and 2 r + 1 when the type has 17..32 units: TPT = 2), its units on the rows with (row & 3) < KS =
//     ceil(units per tile / 4), so that the k-steps KS..3 of a block over a hidden tile multiply zeros and are not issued.  Step r of
//     the sweep is then static code -- the input block(s) of tile r, blocks (r, 0..r) of the masked layer, blocks 0..r of the two head
//     tiles of the dimension ordered r -- with no masks, selects or row bounds; the steps of a transform are the cases of a switch
//     inside the rolled loops over transforms and order values;
//   * the spline inverse runs on the head tiles as the MFMA left them (a16_spline_inv: the width pair and the height pair of lane
//     groups work on four bins each and exchange the selected bin: twelve cross-lane moves, no transposition through memory);
//   * occupancy is set by registers alone: three waves per SIMD (two / one for the two-tile shapes) instead of half a wave.
// Control flow (work queue, retry compaction, speculation once the list has run dry, survivor hand-over to the find / resolve
// rounds, which run the same candidate routine) is k_ar_sample's with 16 entries per wave; the four waves of a workgroup never
// meet; the evaluation statistics stay in the wave until it leaves (per round they were two atomics on the queue head's cache line).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "sf_device.h"
#include "sf_internal.h"
#include "sf_nsfar.h"
#include "sf_rng.h"
#include "sf_spline_flat.h"

namespace {

constexpr int ARK = 8, ARQ = 24;
typedef float f32x4 __attribute__((ext_vector_type(4)));
using ZS = ZSpl<ARK, ARQ>;

struct Ar16Args {
  const float* img;
  const int32_t* dimof;   // [T][D]
  const float* xmean;
  const float* xstd;
  int D, C, T, K, affine;
  long t_stride;
  int o_F0, o_fb0, o_F1, o_fb1, o_F2, o_b2;
  float B, cw, cd;
  float th_scale[8], th_shift[8];
  SfAr16Launch L;
#ifdef SF_A16_TRACE
  unsigned long long* trace;   // developer build: cycle stamps of wave 0 of workgroup 0
#endif
};
#ifdef SF_A16_TRACE
#define A16_TS(slot) do { if (a.trace && blockIdx.x == 0 && threadIdx.x == 0 && (slot) < 512) a.trace[slot] = __builtin_readcyclecounter(); } while (0)
#else
#define A16_TS(slot) do { } while (0)
#endif

__device__ __forceinline__ float4 a16_frag(const float* base, int blk, int lane) {
  return reinterpret_cast<const float4*>(base)[(unsigned)(blk * 64 + lane)];
}
__device__ __forceinline__ f32x4 a16_ld4(const float* p) {
  const float4 b = *reinterpret_cast<const float4*>(p);
  f32x4 r;
  r[0] = b.x; r[1] = b.y; r[2] = b.z; r[3] = b.w;
  return r;
}
// KS k-steps of a block: the sampler's hidden order puts a type's units on rows with (row & 3) < KS of its tile (KS = ceil(units / 4)),
// so the k-steps KS..3 of every block whose INPUT is a hidden tile multiply zeros and are not issued (ten units per type: three of four)
template <int KS = 4>
__device__ __forceinline__ f32x4 a16_mma(const float4 w, const f32x4 in, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, in[0], acc, 0, 0, 0);
  if constexpr (KS > 1) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, in[1], acc, 0, 0, 0);
  if constexpr (KS > 2) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, in[2], acc, 0, 0, 0);
  if constexpr (KS > 3) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, in[3], acc, 0, 0, 0);
  return acc;
}
// two independent products, alternating (a dependent 16x16x4 MFMA waits 40 cycles, an independent one issues after 32)
template <int KS>
__device__ __forceinline__ void a16_mma2(const float4 wa, const float4 wb, const f32x4 in, f32x4& acca, f32x4& accb) {
  acca = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.x, in[0], acca, 0, 0, 0);
  accb = __builtin_amdgcn_mfma_f32_16x16x4f32(wb.x, in[0], accb, 0, 0, 0);
  if constexpr (KS > 1) {
    acca = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.y, in[1], acca, 0, 0, 0);
    accb = __builtin_amdgcn_mfma_f32_16x16x4f32(wb.y, in[1], accb, 0, 0, 0);
  }
  if constexpr (KS > 2) {
    acca = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.z, in[2], acca, 0, 0, 0);
    accb = __builtin_amdgcn_mfma_f32_16x16x4f32(wb.z, in[2], accb, 0, 0, 0);
  }
  if constexpr (KS > 3) {
    acca = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.w, in[3], acca, 0, 0, 0);
    accb = __builtin_amdgcn_mfma_f32_16x16x4f32(wb.w, in[3], accb, 0, 0, 0);
  }
}
__device__ __forceinline__ f32x4 a16_relu(f32x4 v) {
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
  return v;
}

// Step R of the sweep of one transform: the hidden units of type R (tile R) from the inputs known so far, then the two head tiles
// (slots 0..15 / 16..23) of dimension d, the one ordered R.  Every fragment of the step is requested before the first product.
template <int R, int DD, int NI, int KS>
__device__ __forceinline__ void a16_step(const Ar16Args& a, const float* __restrict__ tp, int d, int lane, const f32x4 (&E)[NI], f32x4 (&H1)[DD],
                                         f32x4 (&H2)[DD], f32x4& q0, f32x4& q1) {
  const int g4 = lane >> 4;
  float4 w0[NI], w1[R + 1], wa[R + 1], wb[R + 1];
#pragma unroll
  for (int ti = 0; ti < NI; ++ti) w0[ti] = a16_frag(tp + a.o_F0, R * NI + ti, lane);
#pragma unroll
  for (int kt = 0; kt <= R; ++kt) w1[kt] = a16_frag(tp + a.o_F1, R * DD + kt, lane);
#pragma unroll
  for (int kt = 0; kt <= R; ++kt) {
    wa[kt] = a16_frag(tp + a.o_F2, (d * 2) * DD + kt, lane);
    wb[kt] = a16_frag(tp + a.o_F2, (d * 2 + 1) * DD + kt, lane);
  }
  f32x4 h = a16_ld4(tp + a.o_fb0 + 16 * R + 4 * g4);
  f32x4 acc = a16_ld4(tp + a.o_fb1 + 16 * R + 4 * g4);
  q0 = a16_ld4(tp + a.o_b2 + d * ARQ + 4 * g4);
  q1 = a16_ld4(tp + a.o_b2 + d * ARQ + 16 + 4 * (g4 & 1));   // (slots 16..23; the upper row groups repeat them: never read)
#pragma unroll
  for (int ti = 0; ti < NI; ++ti) h = a16_mma<4>(w0[ti], E[ti], h);
  // what does not depend on this step's tile: the lower blocks of the masked layer and of the head
  f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < R; ++kt) {
    if (kt & 1) acc1 = a16_mma<KS>(w1[kt], H1[kt], acc1);
    else acc = a16_mma<KS>(w1[kt], H1[kt], acc);
  }
#pragma unroll
  for (int kt = 0; kt < R; ++kt) a16_mma2<KS>(wa[kt], wb[kt], H2[kt], q0, q1);
  H1[R] = a16_relu(h);
  acc = a16_mma<KS>(w1[R], H1[R], acc);
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] += acc1[r];
  H2[R] = a16_relu(acc);
  a16_mma2<KS>(wa[R], wb[R], H2[R], q0, q1);
}

// The same step when a type takes TWO tiles (17..32 hidden units per parameter: the reference's own lampe example trains 180 / 150 / 120
// hidden units): tiles 2 R and 2 R + 1 from the 2 R + 2 tiles known by then, within the 256 registers of two workgroups per CU.
// sum over input tiles [K0, K1) of two block rows that share the input tiles (rows ra / rb of the block array `base`, NT blocks per row):
// four input tiles at a time -- eight fragments requested, then their products -- fenced, so that the scheduler does not hoist the
// loads of a whole step (up to 64 fragments) above its first product and spill the hidden tiles to make room
template <int K0, int K1, int NT, int KS>
__device__ __forceinline__ void a16_rows2(const float* __restrict__ base, int ra, int rb, int lane, const f32x4 (&H)[NT], f32x4& acca, f32x4& accb) {
#pragma unroll
  for (int c0 = K0; c0 < K1; c0 += 4) {
    float4 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (c0 + i < K1) {
        fa[i] = a16_frag(base, ra * NT + c0 + i, lane);
        fb[i] = a16_frag(base, rb * NT + c0 + i, lane);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (c0 + i < K1) a16_mma2<KS>(fa[i], fb[i], H[c0 + i], acca, accb);
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <int R, int DD, int NI, int KS>
__device__ __forceinline__ void a16_step2(const Ar16Args& a, const float* __restrict__ tp, int d, int lane, const f32x4 (&E)[NI],
                                          f32x4 (&H1)[2 * DD], f32x4 (&H2)[2 * DD], f32x4& q0, f32x4& q1) {
  constexpr int NT = 2 * DD, O0 = 2 * R, O1 = 2 * R + 1;
  const int g4 = lane >> 4;
  f32x4 h0 = a16_ld4(tp + a.o_fb0 + 16 * O0 + 4 * g4), h1 = a16_ld4(tp + a.o_fb0 + 16 * O1 + 4 * g4);
#pragma unroll
  for (int ti = 0; ti < NI; ++ti) {
    h0 = a16_mma<4>(a16_frag(tp + a.o_F0, O0 * NI + ti, lane), E[ti], h0);
    h1 = a16_mma<4>(a16_frag(tp + a.o_F0, O1 * NI + ti, lane), E[ti], h1);
  }
  f32x4 accA = a16_ld4(tp + a.o_fb1 + 16 * O0 + 4 * g4), accB = a16_ld4(tp + a.o_fb1 + 16 * O1 + 4 * g4);
  H1[O0] = a16_relu(h0);
  H1[O1] = a16_relu(h1);
  a16_rows2<0, O1 + 1, NT, KS>(tp + a.o_F1, O0, O1, lane, H1, accA, accB);
  H2[O0] = a16_relu(accA);
  H2[O1] = a16_relu(accB);
  q0 = a16_ld4(tp + a.o_b2 + d * ARQ + 4 * g4);
  q1 = a16_ld4(tp + a.o_b2 + d * ARQ + 16 + 4 * (g4 & 1));
  a16_rows2<0, O1 + 1, NT, KS>(tp + a.o_F2, d * 2, d * 2 + 1, lane, H2, q0, q1);
}

// The inverse of zuko's MonotonicRQSTransform (ZSpl::inv, sf_spline_flat.h) on the head tiles AS THE MFMA LEFT THEM: of a sample's
// slots, lane group g' = 0 / 1 holds widths 0..3 / 4..7 (q0), g' = 2 / 3 heights 0..3 / 4..7, and g' = 0 / 1 the knot derivatives
// 1..4 / 5..7 (q1).  The width pair and the height pair run the SAME instructions on their four bins each -- soft clip, softmax
// (maximum and sum completed across the pair), cumulative knots -- the height pair's count of knots below v gives the bin, each
// pair selects its bin and the pairs exchange (left, size); twelve cross-lane moves in all, and a quarter of the exponentials
// and selects of the one-thread-per-sample form (which would first need all 24 slots gathered into every lane).
__device__ __forceinline__ float a16_spline_inv(const ZSplC& c, const f32x4 q0, const f32x4 q1, float v, int lane) {
  const int g4 = lane >> 4, s = lane & 15, hi = g4 & 1;
  const int K = c.K;
  const float B = c.B;
  const bool inside = (v > -B) && (v <= B);
  const float vc = fminf(fmaxf(v, -B), B);
  float p[4];
  float mx = -3.0e38f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    p[j] = ZS::clip(q0[j], c.cw);
    mx = (4 * hi + j < K) ? fmaxf(mx, p[j]) : mx;
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  float own = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    p[j] = (4 * hi + j < K) ? sf_exp(p[j] - mx) : 0.f;
    own += p[j];
  }
  const float oth = __shfl_xor(own, 16, 64);
  const float rs = __builtin_amdgcn_rcpf(own + oth);
  // knots of my four bins: c_lo[j], c_hi[j] (bin kk = 4 hi + j; the last bin ends at B, bins past K are empty at B)
  float clo[4], chi[4];
  float run = hi ? oth : 0.f;
  float lo_k = hi ? (4 >= K ? B : 2.0f * B * (run * rs) - B) : -B;
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    run += p[j];
    chi[j] = (4 * hi + j >= K - 1) ? B : 2.0f * B * (run * rs) - B;
    clo[j] = lo_k;
    lo_k = chi[j];
    cnt += (vc > clo[j]) ? 1 : 0;
  }
  // the bin: knots of the HEIGHT family below v, minus one
  int idx = __shfl(cnt, 32 + s, 64) + __shfl(cnt, 48 + s, 64) - 1;
  idx = idx < 0 ? 0 : idx;
  const bool mine = (idx >> 2) == hi;
  const int jj = idx & 3;
  float left = clo[0], size = chi[0] - clo[0];
#pragma unroll
  for (int j = 1; j < 4; ++j) {
    left = (jj == j) ? clo[j] : left;
    size = (jj == j) ? chi[j] - clo[j] : size;
  }
  left = mine ? left : 0.f;
  size = mine ? size : 0.f;
  left += __shfl_xor(left, 16, 64);
  size += __shfl_xor(size, 16, 64);
  const float left2 = __shfl_xor(left, 32, 64), size2 = __shfl_xor(size, 32, 64);
  const bool is_h = g4 >= 2;
  const float y_k = is_h ? left : left2, h_k = is_h ? size : size2;
  const float x_k = is_h ? left2 : left, w_k = is_h ? size2 : size;
  // raw derivatives of knots idx and idx + 1: slots idx - 1 and idx of q1 (lane groups 0 and 1)
  const int sa = idx - 1, sb = idx;
  float ra = 0.f, rb = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ra = (sa == 4 * g4 + j) ? q1[j] : ra;
    rb = (sb == 4 * g4 + j) ? q1[j] : rb;
  }
  const float r_k = __shfl(ra, s, 64) + __shfl(ra, 16 + s, 64);
  const float r_k1 = __shfl(rb, s, 64) + __shfl(rb, 16 + s, 64);
  const float d_k = idx >= 1 ? sf_exp(ZS::clip(r_k, c.cd)) : 1.f;
  const float d_k1 = idx + 1 <= K - 1 ? sf_exp(ZS::clip(r_k1, c.cd)) : 1.f;
  const float s_k = sf_div(h_k, w_k);
  const float dy = vc - y_k;
  const float tmp = dy * (d_k + d_k1 - 2.f * s_k);
  const float aa = tmp + h_k * (s_k - d_k);
  const float bb = h_k * d_k - tmp;
  const float cc = -s_k * dy;
  const float xi = sf_div(2.f * cc, -bb - __builtin_amdgcn_sqrtf(bb * bb - 4.f * aa * cc));
  return inside ? xi * w_k + x_k : v;
}

template <int R, int DD, int NI, int KS, int TPT>
__device__ __forceinline__ void a16_stepT(const Ar16Args& a, const float* __restrict__ tp, int d, int lane, const f32x4 (&E)[NI],
                                          f32x4 (&H1)[DD * TPT], f32x4 (&H2)[DD * TPT], f32x4& q0, f32x4& q1) {
  if constexpr (TPT == 1) a16_step<R, DD, NI, KS>(a, tp, d, lane, E, H1, H2, q0, q1);
  else a16_step2<R, DD, NI, KS>(a, tp, d, lane, E, H1, H2, q0, q1);
}

// one candidate per lane group member: noise of (slot, attempt) through the inverse flow, prior-box test; th = the candidate
template <int DD, int NI, int KS, int TPT>
__device__ __forceinline__ bool a16_candidate(const Ar16Args& a, const ZSplC& sc, long g, unsigned long long slot, uint32_t att, int lane,
                                              bool active, float (&th)[DD]) {
  const int g4 = lane >> 4;
  const SfAr16Launch& L = a.L;
  A16_TS(0);
  f32x4 E[NI], H1[DD * TPT], H2[DD * TPT];
#pragma unroll
  for (int ti = 0; ti < NI; ++ti)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = 16 * ti + 4 * g4 + j - DD;
      const int cc = c < 0 ? 0 : (c < a.C ? c : a.C - 1);
      const float v = (L.x[g * a.C + cc] - a.xmean[cc]) / a.xstd[cc];
      E[ti][j] = (c >= 0 && c < a.C) ? v : 0.f;
    }
#pragma unroll
  for (int k = 0; k < DD * TPT; ++k) { H1[k] = f32x4{0.f, 0.f, 0.f, 0.f}; H2[k] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  float vv[DD];
#pragma unroll
  for (int d0 = 0; d0 < DD; d0 += 4) {
    float z4[4];
    sf_normal4(L.k0, L.k1, slot + L.slot_offset, att, (uint32_t)(d0 >> 2), z4);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (d0 + j < DD) vv[d0 + j] = z4[j];
  }
  A16_TS(1);
  for (int t = a.T - 1; t >= 0; --t) {
    const float* tp = a.img + (size_t)t * a.t_stride;
    int dm[DD];
#pragma unroll
    for (int r = 0; r < DD; ++r) dm[r] = __builtin_amdgcn_readfirstlane(a.dimof[t * DD + r]);
    // the dimensions are not known yet: their input rows are zeros (masked weights are zeros, the values must be finite)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * g4 + j < DD) E[0][j] = 0.f;
    for (int r = 0; r < DD; ++r) {
      int d = 0;
#pragma unroll
      for (int k = 0; k < DD; ++k) d = (k == r) ? dm[k] : d;
      f32x4 q0, q1;
      A16_TS(2 + ((a.T - 1 - t) * DD + r) * 3);
      switch (r) {
        case 0: a16_stepT<0, DD, NI, KS, TPT>(a, tp, d, lane, E, H1, H2, q0, q1); break;
        case 1: if constexpr (DD > 1) a16_stepT<1, DD, NI, KS, TPT>(a, tp, d, lane, E, H1, H2, q0, q1); break;
        case 2: if constexpr (DD > 2) a16_stepT<2, DD, NI, KS, TPT>(a, tp, d, lane, E, H1, H2, q0, q1); break;
        case 3: if constexpr (DD > 3) a16_stepT<3, DD, NI, KS, TPT>(a, tp, d, lane, E, H1, H2, q0, q1); break;
        case 4: if constexpr (DD > 4) a16_stepT<4, DD, NI, KS, TPT>(a, tp, d, lane, E, H1, H2, q0, q1); break;
        case 5: if constexpr (DD > 5) a16_stepT<5, DD, NI, KS, TPT>(a, tp, d, lane, E, H1, H2, q0, q1); break;
        case 6: if constexpr (DD > 6) a16_stepT<6, DD, NI, KS, TPT>(a, tp, d, lane, E, H1, H2, q0, q1); break;
        default: if constexpr (DD > 7) a16_stepT<7, DD, NI, KS, TPT>(a, tp, d, lane, E, H1, H2, q0, q1); break;
      }
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < DD; ++k) v = (k == d) ? vv[k] : v;
#ifdef SF_A16_TRACE
      asm volatile("" :: "v"(q0[0]), "v"(q1[0]));
#endif
      A16_TS(3 + ((a.T - 1 - t) * DD + r) * 3);
      float w;
      if (a.affine) {   // zuko MAF: slot 0 = shift, slot 1 = log-scale logit (rows 0, 1 of the first head tile: lane group 0)
        const int s = lane & 15;
        const float sh = __shfl(q0[0], s, 64), lg = __shfl(q0[1], s, 64);
        w = (v - sh) * sf_exp(-ZS::clip(lg, sc.cd));
      } else {
        w = a16_spline_inv(sc, q0, q1, v, lane);
      }
#ifdef SF_A16_TRACE
      asm volatile("" :: "v"(w));
#endif
      A16_TS(4 + ((a.T - 1 - t) * DD + r) * 3);
#pragma unroll
      for (int k = 0; k < DD; ++k) vv[k] = (k == d) ? w : vv[k];
#pragma unroll
      for (int j = 0; j < 4; ++j) E[0][j] = (4 * g4 + j == d) ? w : E[0][j];
    }
  }
  bool ok = active;
#pragma unroll
  for (int k = 0; k < DD; ++k) {
    const float t_ = (vv[k] - a.th_shift[k]) / a.th_scale[k];
    th[k] = t_;
    ok = ok && (t_ == t_) && fabsf(t_) < 3.0e38f && (!L.lo || (t_ >= L.lo[k] && t_ <= L.hi[k]));
  }
  return ok;
}

// order the LDS traffic of ONE wave (the retry list is wave-private)
__device__ __forceinline__ void a16_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Rejection sampler: every wave works (slot, attempt) items, sixteen at a time -- its own rejects first -- until the slot list is
// exhausted (k_ar_sample's queue, walk order, speculation and survivor hand-over; see there).  Lane l serves entry l & 15; the
// four lanes of an entry compute the same candidate, lane < 16 carries the side effects.
#ifndef SF_A16_WGS
#define SF_A16_WGS 3
#endif
template <int DD, int NI, int KS, int TPT>
__global__ __launch_bounds__(256, TPT == 1 ? SF_A16_WGS : (DD <= 5 ? 2 : 1)) void k_ar_samp16(Ar16Args a) {
  __shared__ unsigned long long r_slot[4][16];
  __shared__ uint32_t r_att[4][16];
  const SfAr16Launch& L = a.L;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s = lane & 15;
  const bool lead = lane < 16;
  const ZSplC sc = {a.K, a.B, a.cw, a.cd};
  const long S = L.S;
  const bool interleave = L.walk_R > 1;
  const unsigned long long n_index = interleave ? L.walk_R * L.walk_C : (unsigned long long)L.n_slots;
  const bool small = n_index <= 0xffffffffull && (unsigned long long)L.n_slots <= 0xffffffffull && (unsigned long long)S <= 0xffffffffull;
  int n_retry = 0;
  bool list_done = false;
  unsigned int n_ev = 0, n_rej0 = 0;
#ifdef SF_A16_TRACE
  if (a.trace && lane == 0) atomicMin(a.trace + 509, (unsigned long long)wall_clock64());
  bool stamped = false;
#endif
  for (;;) {
#ifdef SF_A16_TRACE
    if (a.trace && lane == 0 && list_done && !stamped) { atomicMin(a.trace + 510, (unsigned long long)wall_clock64()); atomicMax(a.trace + 508, (unsigned long long)wall_clock64()); stamped = true; }
#endif
    const int take = list_done ? 0 : 16 - n_retry;
    unsigned long long base = n_index;
    if (take > 0) {
      if (lane == 0) base = atomicAdd(L.cursor, (unsigned long long)take);
      base = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    }
    int n_fresh = 0;
    if (take > 0) {
      if (base >= n_index) list_done = true;
      else {
        const unsigned long long left = n_index - base;
        n_fresh = left < (unsigned long long)take ? (int)left : take;
        if (n_fresh < take) list_done = true;
      }
    }
    const int n_ent = n_retry + n_fresh;
#ifdef SF_A16_TRACE
    if (n_ent == 0 && a.trace && lane == 0) atomicMax(a.trace + 511, (unsigned long long)wall_clock64());
#endif
    if (n_ent == 0) {
      if (lane == 0) {
        if (n_ev) atomicAdd(L.cursor + 2, (unsigned long long)n_ev);
        if (n_rej0) atomicAdd(L.cursor + 3, (unsigned long long)n_rej0);
      }
      break;
    }
    int lw = 0;   // log2 of the speculation width
    if (list_done && !L.count)
      while ((n_ent << (lw + 1)) <= 16) ++lw;
    const int W = 1 << lw;
    const int e = s >> lw, sub = s & (W - 1);
    unsigned long long slot = 0;
    uint32_t att0 = 0;
    bool exists = e < n_ent;
    if (exists) {
      if (e < n_retry) { slot = r_slot[wv][e]; att0 = r_att[wv][e]; }
      else {
        const unsigned long long idx = base + (unsigned)(e - n_retry);
        unsigned long long pos = idx;
        if (interleave) {   // (32-bit division where the cover fits: a 64-bit one is ~200 instructions)
          if (small) { const uint32_t i32 = (uint32_t)idx, r32 = (uint32_t)L.walk_R; pos = (unsigned long long)(i32 % r32) * L.walk_C + i32 / r32; }
          else pos = (idx % L.walk_R) * L.walk_C + idx / L.walk_R;
        }
        if (pos >= (unsigned long long)L.n_slots) exists = false;   // (a hole of the R x C cover)
        else slot = L.slots ? (unsigned long long)L.slots[pos] : pos;
      }
    }
    const uint32_t att = att0 + (uint32_t)sub;
    const bool active = exists && att < L.window_end;
    a16_wave_sync();   // (the retry list has been read)
    const long g = !exists ? 0 : (small ? (long)((uint32_t)slot / (uint32_t)S) : (long)(slot / (unsigned long long)S));
    float th[DD];
    const bool ok = a16_candidate<DD, NI, KS, TPT>(a, sc, g, slot, att, lane, active, th);
    const unsigned long long m_ok = __ballot(ok && lead);
    // (statistics: kept in the wave and added once when it leaves -- per round they were two more atomics on the cache line of the
    //  queue head that every wave's next fetch waits for)
    n_ev += (unsigned)__popcll(__ballot(active && lead));
    n_rej0 += (unsigned)__popcll(__ballot(active && lead && !ok && att == 0u));
    if (L.count) {   // acceptance counting (leakage correction): one attempt per item, nothing written
      // (items are consecutive slots: the sixteen of a wave belong to one row, two at a row boundary -- one add for the first
      //  row's hits instead of up to sixteen same-address atomics, the others on their own)
      const long g0 = ((long)__builtin_amdgcn_readfirstlane((int)(g >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)g);
      const unsigned long long same = __ballot(lead && g == g0);
      const int c0 = __popcll(m_ok & same);
      if (lane == 0 && c0) atomicAdd(L.count + g0, c0);
      if (ok && lead && g != g0) atomicAdd(L.count + g, 1);
      n_retry = 0;
      continue;
    }
    // the entry's lanes: [e W, e W + W) of the first sixteen; its lowest accepted attempt
    const unsigned long long grp = (m_ok >> (e * W)) & ((1ull << W) - 1ull);
    const bool resolved = grp != 0ull;
    const int win = resolved ? __builtin_ctzll(grp) : 0;
    const bool leader = exists && sub == 0 && lead;
    const uint32_t tried_now = att0 + (uint32_t)W < L.window_end ? (uint32_t)W : L.window_end - att0;
    const bool window_out = leader && !resolved && att0 + (uint32_t)W >= L.window_end;
    bool give_up = window_out && L.window_end >= L.max_attempts;
    if (L.g_try && leader) {   // no ceiling asked for: the row-level progress rule (k_ar_sample)
      const int tried = atomicAdd(L.g_try + g, (int)(resolved ? win + 1 : (int)tried_now)) + (int)(resolved ? win + 1 : (int)tried_now);
      if (resolved) atomicAdd(L.g_acc + g, 1);
      else if (tried >= 100000 && __hip_atomic_load(L.g_acc + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) give_up = true;
    }
    if (ok && lead && sub == win) {   // the winner writes the draw
#pragma unroll
      for (int k = 0; k < DD; ++k) L.out[slot * DD + k] = th[k];
      if (L.n_drawn) sf_sat_add(L.n_drawn + g, (int32_t)(att + 1u));
    }
    if (give_up) {
#pragma unroll
      for (int k = 0; k < DD; ++k) L.out[slot * DD + k] = __builtin_nanf("");
      if (L.n_drawn) sf_sat_add(L.n_drawn + g, (int32_t)(att0 + tried_now));
      atomicAdd(L.n_unfilled, 1u);
    }
    if (window_out && !give_up) L.surv[atomicAdd(L.n_surv, 1u)] = (uint32_t)slot;
    const bool again = leader && !resolved && !give_up && !window_out;
    const unsigned long long m = __ballot(again);
    if (again) {
      const int pos = __popcll(m & ((1ull << lane) - 1ull));
      r_slot[wv][pos] = slot;
      r_att[wv][pos] = att0 + (uint32_t)W;
    }
    n_retry = __popcll(m);
    a16_wave_sync();
  }
}

// FIND / RESOLVE of the chip-wide rounds (k_ar_find / k_ar_resolve of sf_nsfar.hip, same arguments) on the 16-sample candidate:
// workgroup (e, j) of FIND tries attempts base + 64 j + 16 wave + (lane & 15) of survivor e and lowers best[e]; RESOLVE re-evaluates
// exactly attempt best[e] of sixteen survivors per wave and writes the draw.
template <int DD, int NI, int KS, int TPT>
__global__ __launch_bounds__(256, TPT == 1 ? SF_A16_WGS : (DD <= 5 ? 2 : 1)) void k_ar_find16(Ar16Args a, const uint32_t* __restrict__ surv, unsigned int n_surv, uint32_t base,
                                                               uint32_t chunks, uint32_t att_end, uint32_t* __restrict__ best,
                                                               unsigned long long* __restrict__ ctr) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const ZSplC sc = {a.K, a.B, a.cw, a.cd};
  const unsigned int e = blockIdx.x / chunks, j = blockIdx.x - e * chunks;
  if (e >= n_surv) return;
  const unsigned long long slot = surv[e];
  const uint32_t att = base + 64u * j + 16u * (uint32_t)wv + (uint32_t)(lane & 15);
  const bool active = att < att_end;
  const long g = (long)(slot / (unsigned long long)a.L.S);
  float th[DD];
  const bool ok = a16_candidate<DD, NI, KS, TPT>(a, sc, g, slot, att, lane, active, th);
  if (ok && lane < 16) atomicMin(best + e, att);
  const unsigned long long ma = __ballot(active && lane < 16);
  if (lane == 0) atomicAdd(ctr + 2, (unsigned long long)__popcll(ma));
}
template <int DD, int NI, int KS, int TPT>
__global__ __launch_bounds__(256, TPT == 1 ? SF_A16_WGS : (DD <= 5 ? 2 : 1)) void k_ar_resolve16(Ar16Args a, const uint32_t* __restrict__ surv, unsigned int n_surv,
                                                                  const uint32_t* __restrict__ best, uint32_t tried_end, uint32_t tried_now,
                                                                  uint32_t* __restrict__ next, unsigned int* __restrict__ n_next) {
  const SfAr16Launch& L = a.L;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const ZSplC sc = {a.K, a.B, a.cw, a.cd};
  const unsigned int e = blockIdx.x * 64u + 16u * (unsigned)wv + (unsigned)(lane & 15);
  const bool exists = e < n_surv, lead = lane < 16;
  const unsigned long long slot = exists ? surv[e] : 0ull;
  const uint32_t b = exists ? best[e] : 0xffffffffu;
  const bool found = exists && b != 0xffffffffu;
  const long g = exists ? (long)(slot / (unsigned long long)L.S) : 0;
  float th[DD];
  const bool ok = a16_candidate<DD, NI, KS, TPT>(a, sc, g, slot, found ? b : 0u, lane, found, th);
  if (!lead) return;
  if (found) {   // (ok by construction: the find launch accepted this very attempt)
#pragma unroll
    for (int k = 0; k < DD; ++k) L.out[slot * DD + k] = ok ? th[k] : __builtin_nanf("");
    if (L.n_drawn) sf_sat_add(L.n_drawn + g, (int32_t)(b + 1u));
    if (L.g_acc) atomicAdd(L.g_acc + g, 1);
    if (L.g_try) atomicAdd(L.g_try + g, (int)tried_now);
  } else if (exists) {
    bool give_up = tried_end >= L.max_attempts;
    if (L.g_try) {   // (uncapped: the progress rule, as in the persistent launch)
      const int tried = atomicAdd(L.g_try + g, (int)tried_now) + (int)tried_now;
      if (tried >= 100000 && __hip_atomic_load(L.g_acc + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) give_up = true;
    }
    if (give_up) {
#pragma unroll
      for (int k = 0; k < DD; ++k) L.out[slot * DD + k] = __builtin_nanf("");
      if (L.n_drawn) sf_sat_add(L.n_drawn + g, (int32_t)tried_end);
      atomicAdd(L.n_unfilled, 1u);
    } else {
      next[atomicAdd(n_next, 1u)] = (uint32_t)slot;
    }
  }
}

template <int DD, int NI, int KS, int TPT>
hipError_t launch16(const Ar16Args& a, int cus, hipStream_t st) {
  static int occ = 0;
  if (!occ) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_ar_samp16<DD, NI, KS, TPT>, 256, 0) != hipSuccess || nb < 1) nb = 2;
    occ = nb > 8 ? 8 : nb;
  }
  long grid = (long)cus * occ;
  const long need = (a.L.n_slots + 63) / 64;
  if (grid > need) grid = need;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((k_ar_samp16<DD, NI, KS, TPT>), dim3((unsigned)grid), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace

bool sf_nsfar16_eligible(const SfNsfAr& n) {
  static int off = -1;
  if (off < 0) { const char* e = std::getenv("SF_AR_SAMP16"); off = (e && e[0] == '0') ? 1 : 0; }
  return !off && n.s16_nt >= 2 && n.D >= 2 && n.D <= 8 && (n.s16_tpt == 1 || n.s16_tpt == 2) && n.s16_nt == n.D * n.s16_tpt &&
         (n.s16_ni == 1 || n.s16_ni == 2);
}

static Ar16Args a16_args(const SfNsfAr& n, const SfAr16Launch& L) {
  Ar16Args a;
  a.img = n.d_img; a.dimof = n.d_dimof; a.xmean = n.d_xmean; a.xstd = n.d_xstd;
  a.D = n.D; a.C = n.C; a.T = n.T; a.K = n.K; a.affine = n.affine;
  a.t_stride = n.t_stride;
  a.o_F0 = n.o_F0; a.o_fb0 = n.o_fb0; a.o_F1 = n.o_F1; a.o_fb1 = n.o_fb1; a.o_F2 = n.o_F2; a.o_b2 = n.o_b2;
  a.B = n.bound; a.cw = n.cw; a.cd = n.cd;
  for (int d = 0; d < 8; ++d) { a.th_scale[d] = n.th_scale[d]; a.th_shift[d] = n.th_shift[d]; }
  a.L = L;
#ifdef SF_A16_TRACE
  a.trace = nullptr;
#endif
  return a;
}
// (DD, NI, KS, TPT) of a flow -> the instantiation: F is called with four integral constants.  Two tiles per type are built for
// three and four k-steps only (17..32 units per type: 9..16 per tile).
template <typename F>
static hipError_t a16_dispatch(const SfNsfAr& n, F f) {
  auto with_ks = [&](auto dd, auto ni) {
    if (n.s16_tpt == 2) {
      if (n.s16_ks == 3) return f(dd, ni, std::integral_constant<int, 3>(), std::integral_constant<int, 2>());
      return f(dd, ni, std::integral_constant<int, 4>(), std::integral_constant<int, 2>());
    }
    switch (n.s16_ks) {
      case 2: return f(dd, ni, std::integral_constant<int, 2>(), std::integral_constant<int, 1>());
      case 3: return f(dd, ni, std::integral_constant<int, 3>(), std::integral_constant<int, 1>());
      default: return f(dd, ni, std::integral_constant<int, 4>(), std::integral_constant<int, 1>());
    }
  };
#define A16_CASE(DD)                                                                                                             \
  case DD:                                                                                                                       \
    return n.s16_ni == 1 ? with_ks(std::integral_constant<int, DD>(), std::integral_constant<int, 1>())                         \
                         : with_ks(std::integral_constant<int, DD>(), std::integral_constant<int, 2>());
  switch (n.D) {
    A16_CASE(2) A16_CASE(3) A16_CASE(4) A16_CASE(5) A16_CASE(6) A16_CASE(7) A16_CASE(8)
    default: return hipErrorInvalidValue;
  }
#undef A16_CASE
}

hipError_t sf_nsfar16_launch(const SfNsfAr& n, const SfAr16Launch& L, int cus, hipStream_t st) {
  Ar16Args a = a16_args(n, L);
#ifdef SF_A16_TRACE
  static unsigned long long* d_tr = nullptr;
  if (!d_tr && hipMalloc(&d_tr, 512 * 8) != hipSuccess) return hipErrorOutOfMemory;
  (void)hipMemsetAsync(d_tr, 0, 512 * 8, st);
  (void)hipMemsetAsync(d_tr + 509, 0xff, 2 * 8, st);
  a.trace = d_tr;
  struct Dump { unsigned long long* p; hipStream_t st; ~Dump() {
    (void)hipStreamSynchronize(st);
    unsigned long long h[512];
    (void)hipMemcpy(h, p, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr, "[a16 timeline] first wave sees the list dry at %.1f us, last wave sees it at %.1f us, last wave leaves at %.1f us (100 MHz clock, from the first wave's start)\n",
            (double)(long long)(h[510] - h[509]) * 0.01, (double)(long long)(h[508] - h[509]) * 0.01, (double)(long long)(h[511] - h[509]) * 0.01);
    fprintf(stderr, "[a16 trace] cycles since stamp 0:");
    for (int i = 0; i < 512; ++i) if (h[i]) fprintf(stderr, " %d:%lld", i, (long long)(h[i] - h[0]));
    fprintf(stderr, "\n");
  } } dump{d_tr, st};
#endif
  return a16_dispatch(n, [&](auto dd, auto ni, auto ks, auto tpt) { return launch16<decltype(dd)::value, decltype(ni)::value, decltype(ks)::value, decltype(tpt)::value>(a, cus, st); });
}

hipError_t sf_nsfar16_find(const SfNsfAr& n, const SfAr16Launch& L, const uint32_t* surv, unsigned int n_surv, uint32_t base, uint32_t chunks,
                           uint32_t att_end, uint32_t* best, unsigned long long* ctr, hipStream_t st) {
  const Ar16Args a = a16_args(n, L);
  return a16_dispatch(n, [&](auto dd, auto ni, auto ks, auto tpt) {
    hipLaunchKernelGGL((k_ar_find16<decltype(dd)::value, decltype(ni)::value, decltype(ks)::value, decltype(tpt)::value>), dim3(n_surv * chunks), dim3(256), 0, st, a, surv, n_surv, base, chunks,
                       att_end, best, ctr);
    return hipGetLastError();
  });
}

hipError_t sf_nsfar16_resolve(const SfNsfAr& n, const SfAr16Launch& L, const uint32_t* surv, unsigned int n_surv, const uint32_t* best,
                              uint32_t tried_end, uint32_t tried_now, uint32_t* next, unsigned int* n_next, hipStream_t st) {
  const Ar16Args a = a16_args(n, L);
  return a16_dispatch(n, [&](auto dd, auto ni, auto ks, auto tpt) {
    hipLaunchKernelGGL((k_ar_resolve16<decltype(dd)::value, decltype(ni)::value, decltype(ks)::value, decltype(tpt)::value>), dim3((n_surv + 63u) / 64u), dim3(256), 0, st, a, surv, n_surv, best,
                       tried_end, tried_now, next, n_next);
    return hipGetLastError();
  });
}
