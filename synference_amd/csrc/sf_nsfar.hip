// sf_nsfar.hip -- the autoregressive NSF of the reference's second backend (`backend="lampe"`: ili.utils.load_nde_lampe ->
// zuko.flows.NSF; ref: src/synference/sbi_runner.py:5123-5125, examples/sbi/scripts/basic_model.py:31-41).
//
// [UPSTREAM, restated from the published zuko sources -- oracle/flows.py "nsf_ar" is the CPU twin, parity unpinned]
//   T MaskedAutoregressiveTransforms with alternating orderings; the hyper-network of a transform is a MaskedMLP
//   [theta ; context] -> H -> H -> D (3K - 1) (ReLU) whose masks make the spline parameters of a dimension depend on the
//   dimensions ordered before it and on the context; the univariate map is zuko's MonotonicRQSTransform (sf_spline_flat.h, ZSpl).
//
// One THREAD per sample for the univariate maps, the wave's 64 samples side by side for the products: activations live in LDS
// as rows of 65 floats ([unit][sample]), every layer product is a set of 16 x 16 tiles on v_mfma_f32_16x16x4_f32 -- A = sixteen
// output units x four inputs straight from the k-major masked weight image in L2 (four 64-byte segments per load), B = four
// input rows x sixteen samples from LDS, four sample tiles per weight load.  The same tile routine runs the backward data
// products on the transposed images (L1m, L2m, L0m), and the weight gradients are 16 x 16 blocks over the 64 samples (A = delta
// rows, B = input rows), added with f32 atomics.  Hidden units are stored SORTED BY TYPE (zuko: unit h has type h mod D and sees
// the inputs ordered before its type), every type padded to a multiple of eight rows (all of them to a multiple of 16), so that
//   * the masked layers are block lower-triangular: an output tile whose highest type is r reads the first tend[r] rows only
//     (masked entries of the images are zeros);
//   * the SAMPLING direction costs one hyper-network evaluation per transform, not D: the units of type r become final as
//     soon as the dimensions ordered before r are inverted, so the sweep r = 0 .. D-1 computes the hidden units of type r,
//     the head of the dimension with order r, inverts that dimension, and goes on (zuko sweeps the whole network D times);
//   * rejected draws are retried by compaction: a wave keeps taking (slot, attempt) items -- its own rejects first -- until
//     the catalogue's slot list is exhausted.
// Training: forward with every transform's inputs stashed, then per transform (top down) the hyper-network is recomputed from
// the stash and back-propagated by hand.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <ctime>
#include <cstdio>
#include <string>
#include <vector>

#include "sf_device.h"
#include "sf_internal.h"
#include "sf_nsfar.h"
#include "sf_rng.h"
#include "sf_spline_flat.h"

namespace {

// one element of a weight gradient: k_ar_train<NWV, PART> -- PART: the workgroup's own partial (every element of a transform's
// gradient is produced exactly once per workgroup: plain stores, summed in workgroup order by k_ar_gather: no global atomics -- the
// LDS adds of the hidden deltas inside a workgroup keep the hardware's order, so runs agree to rounding, not to the bit); else f32 atomics into the one gradient (256 workgroups adding the same 57 k elements in lockstep: 0.18 of 0.57 ms)
#define AR_GADD(p, v) do { if (PART && first_chunk) *(p) = (v); else unsafeAtomicAdd((p), (v)); } while (0)
constexpr int ARK = 8, ARQ = 24;   // bins capacity / parameter slots per dimension (K <= 8: 3K - 1 <= 23)
constexpr int RS = 65;             // floats per LDS row (64 samples + 1: the MFMA operand reads of the training kernel walk rows
                                   // with the lane index -- a stride of 64 would put sixteen rows on one bank)
typedef float ar_f32x4 __attribute__((ext_vector_type(4)));
using ZS = ZSpl<ARK, ARQ>;

struct ArArgs {
  const float* img;       // packed images, all transforms
  const int32_t* perm;    // [Hp] physical hidden row -> logical unit (-1: padding)
  const int32_t* ptype;   // [Hp] type of a physical row (padding rows: the type of their block)
  const int32_t* tend;    // [D]  rows of type <= r
  const int32_t* ord;     // [T][D] order value of dimension d
  const int32_t* dimof;   // [T][D] dimension with order value r
  const int32_t* dwave;   // [T][D] wave (of four) that runs dimension d of transform t in the multi-wave density / training kernels
  const float* xmean;     // [C]
  const float* xstd;      // [C]
  int D, C, H, Hp, T, K, NP, NIN16, NIN4;
  int affine;             // SF_MAF_AR: the univariate map is zuko's MonotonicAffineTransform (slots 0 = shift, 1 = log-scale logit)
  long t_stride;          // floats per transform in img
  int o_L0t, o_b0, o_L1t, o_L1m, o_b1, o_L2t, o_b2, o_L0m, o_L2m;
  long P_t;               // logical parameters per transform
  int l_W0, l_b0, l_W1, l_b1, l_W2, l_b2;
  float B, cw, cd, logdet0;
  float th_scale[16], th_shift[16];
  short tendk[16];        // = tend[] (kernel-argument copy: no dependent global load)
  short tile_kend[24];    // per 16-row tile of the hidden rows: rows its highest type reads (tend of that type)
  short tile_kbeg[24];    // ... first row of its lowest type
#ifdef SF_AR_TRACE
  unsigned long long* trace;   // developer build: cycle stamps of block 0 (k_ar_logprob)
#endif
};
#ifdef SF_AR_TRACE
#define AR_TS(slot) do { if (a.trace && blockIdx.x == 0 && threadIdx.x == 0 && (slot) < 256) a.trace[slot] = __builtin_readcyclecounter(); } while (0)
#else
#define AR_TS(slot) do { } while (0)
#endif

// 16 x 16 block of a weight gradient: sum over the wave's 64 samples of A[o0 + i][s] * B[k0 + j][s] (sixteen steps of four
// samples); lane l holds rows 4 (l >> 4) + r, column l & 15
__device__ __forceinline__ ar_f32x4 ar_dw16(const float* A, int o0, const float* Bm, int k0, int lane) {
  const float* pa = A + (o0 + (lane & 15)) * RS + (lane >> 4);
  const float* pb = Bm + (k0 + (lane & 15)) * RS + (lane >> 4);
  // all 32 operands first, then two interleaved accumulator chains (a dependent 16x16x4 MFMA waits 40 cycles, an independent
  // one issues after 32): no LDS round trip and no dependent wait between the products
  float av[16], bw[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { av[i] = pa[4 * i]; bw[i] = pb[4 * i]; }
  __builtin_amdgcn_sched_barrier(0);
  ar_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 16; i += 2) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bw[i], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i + 1], bw[i + 1], acc1, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) acc0[r] += acc1[r];
  return acc0;
}
// the row sums of the same A rows (B = ones): every column holds them
__device__ __forceinline__ ar_f32x4 ar_rowsum16(const float* A, int o0, int lane) {
  const float* pa = A + (o0 + (lane & 15)) * RS + (lane >> 4);
  float av[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) av[i] = pa[4 * i];
  __builtin_amdgcn_sched_barrier(0);
  ar_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 16; i += 2) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], 1.0f, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i + 1], 1.0f, acc1, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) acc0[r] += acc1[r];
  return acc0;
}

// The univariate map of dimension d from its parameter slots q: the rational-quadratic spline (zuko NSF), or -- a.affine, zuko
// MAF -- MonotonicAffineTransform: out = v exp(ls) + q[0], ls = q[1] / (1 + |q[1] / log slope|), log|d out / d v| = ls.
__device__ __forceinline__ void ar_uni_fwd(const ArArgs& a, const ZSplC& sc, const float (&q)[ARQ], float v, float& out, float& lad) {
  if (a.affine) {
    const float ls = ZS::clip(q[1], sc.cd);
    out = v * sf_exp(ls) + q[0];
    lad = ls;
  } else {
    ZS::fwd(sc, q, v, out, lad);
  }
}
__device__ __forceinline__ void ar_uni_inv(const ArArgs& a, const ZSplC& sc, const float (&q)[ARQ], float v, float& out, float& lad) {
  if (a.affine) {
    const float ls = ZS::clip(q[1], sc.cd);
    out = (v - q[0]) * sf_exp(-ls);
    lad = -ls;
  } else {
    ZS::inv(sc, q, v, out, lad);
  }
}
// Go = dL / d out, Gl = dL / d lad  ->  dv = dL / d v, dq = dL / d q
__device__ __forceinline__ void ar_uni_bwd(const ArArgs& a, const ZSplC& sc, const float (&q)[ARQ], float v, float Go, float Gl, float& dv,
                                           float (&dq)[ARQ]) {
  if (a.affine) {
#pragma unroll
    for (int i = 0; i < ARQ; ++i) dq[i] = 0.f;
    const float e = sf_exp(ZS::clip(q[1], sc.cd));
    dv = Go * e;
    dq[0] = Go;
    dq[1] = (Go * v * e + Gl) * ZS::dclip(q[1], sc.cd);
  } else {
    ZS::bwd(sc, q, v, Go, Gl, dv, dq);
  }
}
// parameter slot sl of a dimension (family sl >> 3, index sl & 7) -> row of the dimension's block in the logical head, -1 = none
__device__ __forceinline__ int ar_slot_row(const ArArgs& a, int sl) {
  const int fam = sl >> 3, kk = sl & 7;
  if (a.affine) return (fam == 0 && kk < 2) ? kk : -1;
  return (sl < ARQ - 1 && kk < (fam < 2 ? a.K : a.K - 1)) ? fam * a.K + kk : -1;
}

// workgroup barrier of the multi-wave kernels of this file
__device__ __forceinline__ void ar_barrier() {
  __syncthreads();   // (-DSF_FUZZ_SCHED: the macro of sf_device.h, barrier + delay)
}

// order the LDS traffic of ONE wave (its LDS operations execute in program order: a read issued after a write of another lane
// of the same wave sees it; this only keeps the compiler from reordering them)
__device__ __forceinline__ void ar_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// NTL x sixteen output rows [p0, p0 + 16 NTL) of a product for the wave's 64 samples:
//   out[p - oshift][s] = act(bias[p] (or the old value: ACC) + sum_{k in [kbeg, kend)} wt[k][p] * in[k][s])
// wt: k-major image (row k, ldo floats per row); kbeg, kend multiples of four; rows p >= row_lim are not written.  Columns past
// the image's row end (edge tiles) read the neighbouring floats of the same transform's image: finite, and never written.
// The tiles of a call share every activation read; the weights come four k-steps at a time, the next four requested before the
// products of the current ones are issued, and the bias (requested first) is only added at the end: what a call exposes is ONE
// L2 round trip, not three.
template <bool RELU, bool ACC, int NTL, bool ATOM = false>
__device__ __forceinline__ void ar_tiles(const float* __restrict__ wt, int ldo, const float* __restrict__ bias, int p0, int kbeg, int kend,
                                         const float* in, float* out, int row_lim, int lane, int oshift = 0) {
  const int i4 = 4 * (lane >> 4), j = lane & 15;
  float* orow = out + (p0 - oshift + i4) * RS + j;
  const float* wa = wt + (size_t)(lane >> 4) * ldo + p0 + j;
  const float* pb = in + (lane >> 4) * RS + j;
  // Four k-steps (sixteen input rows) per round: the next round's weights are requested first -- UNCONDITIONALLY, the row index
  // clamped to the last valid one, so that the compiler can count them in vmcnt and does not drain them where this round's
  // weights are waited for -- then all sixteen activation reads of the round, then the products (skipped past kend).
  constexpr int CH = 4;
  const int klast = kend - 4;
  float wq[CH][NTL];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int kc = kbeg + 4 * c < klast ? kbeg + 4 * c : klast;
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl) wq[c][tl] = wa[(size_t)kc * ldo + 16 * tl];
  }
  float4 bv[NTL];
#pragma unroll
  for (int tl = 0; tl < NTL; ++tl) bv[tl] = bias ? *reinterpret_cast<const float4*>(bias + p0 + 16 * tl + i4) : make_float4(0.f, 0.f, 0.f, 0.f);
  ar_f32x4 acc[NTL][4];
#pragma unroll
  for (int tl = 0; tl < NTL; ++tl)
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[tl][st][r] = (ACC && !ATOM && p0 + 16 * tl + i4 + r < row_lim) ? orow[(16 * tl + r) * RS + st * 16] : 0.f;
  for (int k0 = kbeg; k0 < kend; k0 += 4 * CH) {
    float wn[CH][NTL];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int kn = k0 + 4 * CH + 4 * c;
      const int kc = kn < klast ? kn : klast;
#pragma unroll
      for (int tl = 0; tl < NTL; ++tl) wn[c][tl] = wa[(size_t)kc * ldo + 16 * tl];
    }
    float bq[CH][4];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int st = 0; st < 4; ++st) bq[c][st] = pb[(k0 + 4 * c) * RS + st * 16];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (k0 + 4 * c < kend) {   // (wave-uniform)
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
          for (int tl = 0; tl < NTL; ++tl) acc[tl][st] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[c][tl], bq[c][st], acc[tl][st], 0, 0, 0);
      }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int tl = 0; tl < NTL; ++tl) wq[c][tl] = wn[c][tl];
  }
#pragma unroll
  for (int tl = 0; tl < NTL; ++tl) {
    const float bb[4] = {bv[tl].x, bv[tl].y, bv[tl].z, bv[tl].w};
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (p0 + 16 * tl + i4 + r < row_lim) {
          const float v = acc[tl][st][r] + bb[r];
          if (ATOM) atomicAdd(orow + (16 * tl + r) * RS + st * 16, v);   // (several waves add into these rows: ds_add_f32)
          else orow[(16 * tl + r) * RS + st * 16] = RELU ? fmaxf(v, 0.f) : v;
        }
  }
}
// Warm the vector L1 for a tile pass that comes later: lane l requests the first floats of row kbeg + l (+ 64 ...) of the image
// at column p0 (and p0 + 16): one load instruction per 64 rows touches the 64-byte segments the pass will read.  The values
// are only "used" by ar_keep, placed behind the work that is to cover the L2 round trip.
struct ArPre { float v[6]; };
__device__ __forceinline__ void ar_touch(ArPre& P, int slot, const float* __restrict__ wt, int ldo, int p0, int kbeg, int kend, int lane) {
  const int k = kbeg + lane;
  P.v[slot] = k < kend ? wt[(size_t)k * ldo + p0] : 0.f;
  P.v[slot + 1] = k < kend ? wt[(size_t)k * ldo + p0 + 16] : 0.f;
  if (kbeg + 64 < kend) {
    const int k2 = k + 64;
    P.v[slot] += k2 < kend ? wt[(size_t)k2 * ldo + p0] : 0.f;
    P.v[slot + 1] += k2 < kend ? wt[(size_t)k2 * ldo + p0 + 16] : 0.f;
  }
}
__device__ __forceinline__ void ar_keep(const ArPre& P) {
#pragma unroll
  for (int i = 0; i < 6; ++i) asm volatile("" ::"v"(P.v[i]));
}

// rows [p_lo, p_hi) in pairs of tiles, a single one at the end; kend_of(p0, n): inputs the n tiles from p0 on read
template <bool RELU, typename KE>
__device__ __forceinline__ void ar_rows(const float* __restrict__ wt, int ldo, const float* __restrict__ bias, int p_lo, int p_hi, int kbeg,
                                        KE kend_of, const float* in, float* out, int row_lim, int lane, int wid = 0, int nwv = 1) {
  // (wave wid of nwv takes every nwv-th pair of tiles)
  for (int p0 = p_lo + 32 * wid; p0 < p_hi; p0 += 32 * nwv) {
    if (p0 + 16 < p_hi) ar_tiles<RELU, false, 2>(wt, ldo, bias, p0, kbeg, kend_of(p0, 2), in, out, row_lim, lane);
    else ar_tiles<RELU, false, 1>(wt, ldo, bias, p0, kbeg, kend_of(p0, 1), in, out, row_lim, lane);
  }
}

// x[r][lane] <- x[r][lane] where gate[r][lane] > 0, rows [0, n) (n a multiple of 8): eight rows per round, every read of a round
// before its writes (row by row, each write would wait for the LDS round trip of its own reads)
__device__ __forceinline__ void ar_mask_rows(float* x, const float* gate, int n, int lane, int wid = 0, int nwv = 1) {
  for (int r0 = 8 * wid; r0 < n; r0 += 8 * nwv) {
    float xv[8], gv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { xv[i] = x[(r0 + i) * RS + lane]; gv[i] = gate[(r0 + i) * RS + lane]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) x[(r0 + i) * RS + lane] = gv[i] > 0.f ? xv[i] : 0.f;
  }
}

// bits[r] = lanes (samples) with x[r][lane] > 0, rows [0, n): two 32-bit words per row
__device__ __forceinline__ void ar_sign_bits(const float* x, unsigned int* bits, int n, int lane, int wid = 0, int nwv = 1) {
  for (int r0 = 8 * wid; r0 < n; r0 += 8 * nwv) {
    float xv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) xv[i] = x[(r0 + i) * RS + lane];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned long long b = __ballot(xv[i] > 0.f);
      if (lane == 0) { bits[2 * (r0 + i)] = (unsigned int)b; bits[2 * (r0 + i) + 1] = (unsigned int)(b >> 32); }
    }
  }
}
// x[r][lane] <- x[r][lane] where bit `lane` of bits[r] is set
__device__ __forceinline__ void ar_mask_bits(float* x, const unsigned int* bits, int n, int lane, int wid = 0, int nwv = 1) {
  for (int r0 = 8 * wid; r0 < n; r0 += 8 * nwv) {
    float xv[8];
    unsigned int bw[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { xv[i] = x[(r0 + i) * RS + lane]; bw[i] = bits[2 * (r0 + i) + (lane >> 5)]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) x[(r0 + i) * RS + lane] = ((bw[i] >> (lane & 31)) & 1u) ? xv[i] : 0.f;
  }
}
// the first hidden layer alone (the training sweep recomputes it where it needs it again: it is the cheap one, NIN4 inputs)
__device__ __forceinline__ void ar_hidden0(const ArArgs& a, const float* __restrict__ tp, const float* E0, float* H1, int lane, int wid = 0,
                                           int nwv = 1) {
  ar_rows<true>(tp + a.o_L0t, a.Hp, tp + a.o_b0, 0, a.Hp, 0, [&](int, int) { return a.NIN4; }, E0, H1, a.Hp, lane, wid, nwv);
}

// both hidden layers of a transform from the inputs in E0 (rows [0, NIN4): u, context, zeros)
__device__ __forceinline__ void ar_hidden(const ArArgs& a, const float* __restrict__ tp, const float* E0, float* H1, float* H2, int lane,
                                          int wid = 0, int nwv = 1) {
  ar_barrier();   // (E0 was written sample by sample)
  ar_rows<true>(tp + a.o_L0t, a.Hp, tp + a.o_b0, 0, a.Hp, 0, [&](int, int) { return a.NIN4; }, E0, H1, a.Hp, lane, wid, nwv);
  ar_barrier();
  ar_rows<true>(tp + a.o_L1t, a.Hp, tp + a.o_b1, 0, a.Hp, 0, [&](int p0, int n) { return (int)a.tile_kend[(p0 >> 4) + n - 1]; }, H1, H2, a.Hp, lane,
                wid, nwv);
  ar_barrier();
}
// the 24 parameter slots of dimension d from the last hidden layer (rows < kend) -> q (through the 32 rows of QB)
__device__ __forceinline__ void ar_head(const ArArgs& a, const float* __restrict__ tp, int d, int kend, const float* H2, float* QB, int lane,
                                        float (&q)[ARQ]) {
  // (QB belongs to the calling wave: wave-local ordering is all it needs)
  ar_wave_sync();
  // rows d * ARQ + i of the head -> QB row i (two tiles: rows 24 .. 31 of the second are the next dimension's, not written)
  ar_tiles<false, false, 2>(tp + a.o_L2t, a.D * ARQ, tp + a.o_b2, d * ARQ, 0, kend, H2, QB, d * ARQ + ARQ, lane, d * ARQ);
  ar_wave_sync();
#pragma unroll
  for (int sl = 0; sl < ARQ; ++sl) q[sl] = QB[sl * RS + lane];
}

__device__ __forceinline__ void ar_load_inputs(const ArArgs& a, const float* __restrict__ theta, const float* __restrict__ x, long row,
                                               float* E0, int lane, int wid = 0, int nwv = 1) {
  for (int d = wid; d < a.D; d += nwv) E0[d * RS + lane] = theta[row * a.D + d] * a.th_scale[d] + a.th_shift[d];
  for (int c = wid; c < a.C; c += nwv) E0[(a.D + c) * RS + lane] = (x[row * a.C + c] - a.xmean[c]) / a.xstd[c];
  for (int r = a.D + a.C + wid; r < a.NIN16; r += nwv) E0[r * RS + lane] = 0.f;   // (rows the k-steps of four run over)
}

// NWV waves per 64 samples: the tile pairs of a layer and the dimensions of a transform (head + spline) are dealt round robin
template <int NWV>
__global__ __launch_bounds__(64 * NWV) void k_ar_logprob(ArArgs a, const float* __restrict__ theta, const float* __restrict__ x, long B,
                                                          float* __restrict__ out) {
  extern __shared__ float lds[];
  float* E0 = lds;
  float* H1 = E0 + a.NIN16 * RS;
  float* H2 = H1 + a.Hp * RS;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float* QB = H2 + a.Hp * RS + wid * 32 * RS;   // (one 32-row head buffer per wave)
  const long b = (long)blockIdx.x * 64 + lane;
  const long row = b < B ? b : B - 1;
  AR_TS(0);
  ar_load_inputs(a, theta, x, row, E0, lane, wid, NWV);
  const ZSplC sc = {a.K, a.B, a.cw, a.cd};
  float ld = 0.f;
  for (int t = 0; t < a.T; ++t) {
    const float* tp = a.img + (size_t)t * a.t_stride;
    AR_TS(1 + t * 40);
    ar_hidden(a, tp, E0, H1, H2, lane, wid, NWV);
    AR_TS(2 + t * 40);
    for (int d = 0; d < a.D; ++d) {
      if (NWV > 1 && a.dwave[t * a.D + d] != wid) continue;   // (dimensions dealt by cost: see sf_nsfar_create)
      float q[ARQ];
      ar_head(a, tp, d, (int)a.tendk[a.ord[t * a.D + d]], H2, QB, lane, q);
      AR_TS(3 + t * 40 + 2 * d);
      float v, lad;
      ar_uni_fwd(a, sc, q, E0[d * RS + lane], v, lad);
      E0[d * RS + lane] = v;   // (every parameter of this transform has been taken from the inputs already)
      ld += lad;
      AR_TS(4 + t * 40 + 2 * d);
    }
  }
  ar_barrier();
  if (NWV > 1) {   // the log-determinants of the waves' dimensions
    QB[lane] = ld;
    ar_barrier();
    if (wid == 0)
      for (int w2 = 1; w2 < NWV; ++w2) ld += H2[a.Hp * RS + w2 * 32 * RS + lane];
  }
  if (wid == 0) {
    float ss = 0.f;
    for (int d = 0; d < a.D; ++d) ss += E0[d * RS + lane] * E0[d * RS + lane];
    if (b < B) out[b] = -0.5f * ss - 0.5f * (float)a.D * 1.8378770664093453f + ld + a.logdet0;
  }
  AR_TS(250);
}

// the inverse of transform t in ONE sweep over the order values: V = the transform's outputs, E0[0 .. D) receives its inputs
__device__ __forceinline__ float ar_inverse_transform(const ArArgs& a, const ZSplC& sc, int t, float* E0, const float* V, float* H1,
                                                      float* H2, float* QB, int lane) {
  const float* tp = a.img + (size_t)t * a.t_stride;
  float ld = 0.f;
  for (int d = 0; d < a.D; ++d) E0[d * RS + lane] = 0.f;   // (not yet known: masked weights are zeros, the values must be finite)
  for (int r = 0; r < a.D; ++r) {
    const int p_lo = r ? (int)a.tendk[r - 1] : 0, p_hi = (int)a.tendk[r];
    ar_barrier();   // (E0 row of the dimension inverted last)
    ar_rows<true>(tp + a.o_L0t, a.Hp, tp + a.o_b0, p_lo, p_hi, 0, [&](int, int) { return a.NIN4; }, E0, H1, p_hi, lane);
    ar_barrier();
    ar_rows<true>(tp + a.o_L1t, a.Hp, tp + a.o_b1, p_lo, p_hi, 0, [&](int, int) { return p_hi; }, H1, H2, p_hi, lane);
    const int d = a.dimof[t * a.D + r];
    // the step after this one (the next order value, or the first of the transform below): which rows, which dimension
    const bool more = r + 1 < a.D || t > 0;
    const int tn = r + 1 < a.D ? t : t - 1, rn = r + 1 < a.D ? r + 1 : 0;
    const int dn = more ? a.dimof[tn * a.D + rn] : 0;
    float q[ARQ];
    ar_head(a, tp, d, p_hi, H2, QB, lane, q);
    ArPre pre = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
    if (more) {   // its weights are asked for now, under this step's spline
      const float* tq = a.img + (size_t)tn * a.t_stride;
      const int q_lo = rn ? (int)a.tendk[rn - 1] : 0, q_hi = (int)a.tendk[rn];
      ar_touch(pre, 0, tq + a.o_L0t, a.Hp, q_lo, 0, a.NIN4, lane);
      ar_touch(pre, 2, tq + a.o_L1t, a.Hp, q_lo, 0, q_hi, lane);
      ar_touch(pre, 4, tq + a.o_L2t, a.D * ARQ, dn * ARQ, 0, q_hi, lane);
    }
    float w, lad;
    ar_uni_inv(a, sc, q, V[d * RS + lane], w, lad);
    E0[d * RS + lane] = w;
    ld += lad;
    ar_keep(pre);
  }
  return ld;
}

__global__ __launch_bounds__(64) void k_ar_inverse(ArArgs a, const float* __restrict__ z, const float* __restrict__ x, long B,
                                                    float* __restrict__ theta, float* __restrict__ logdet) {
  extern __shared__ float lds[];
  float* E0 = lds;
  float* H1 = E0 + a.NIN16 * RS;
  float* H2 = H1 + a.Hp * RS;
  float* QB = H2 + a.Hp * RS;
  float* V = QB + 32 * RS;
  const int lane = threadIdx.x;
  const long b = (long)blockIdx.x * 64 + lane;
  const long row = b < B ? b : B - 1;
  for (int c = 0; c < a.C; ++c) E0[(a.D + c) * RS + lane] = (x[row * a.C + c] - a.xmean[c]) / a.xstd[c];
  for (int r = a.D + a.C; r < a.NIN16; ++r) E0[r * RS + lane] = 0.f;
  for (int d = 0; d < a.D; ++d) V[d * RS + lane] = z[row * a.D + d];
  const ZSplC sc = {a.K, a.B, a.cw, a.cd};
  float ld = -a.logdet0;
  for (int t = a.T - 1; t >= 0; --t) {
    ld += ar_inverse_transform(a, sc, t, E0, V, H1, H2, QB, lane);
    for (int d = 0; d < a.D; ++d) V[d * RS + lane] = E0[d * RS + lane];
  }
  if (b < B) {
    for (int d = 0; d < a.D; ++d) theta[b * a.D + d] = (V[d * RS + lane] - a.th_shift[d]) / a.th_scale[d];
    if (logdet) logdet[b] = ld;
  }
}

// one candidate per lane: noise of (slot, attempt) through the inverse flow, box test; the candidate is left in V rows [0, D)
__device__ __forceinline__ bool ar_candidate(const ArArgs& a, const ZSplC& sc, const float* __restrict__ x, long g, unsigned long long slot,
                                             uint32_t att, uint32_t k0, uint32_t k1, unsigned long long slot_offset,
                                             const float* __restrict__ lo, const float* __restrict__ hi, float* E0, float* V, float* H1, float* H2,
                                             float* QB, int lane, bool active) {
  for (int c = 0; c < a.C; ++c) E0[(a.D + c) * RS + lane] = (x[g * a.C + c] - a.xmean[c]) / a.xstd[c];
  for (int d0 = 0; d0 < a.D; d0 += 4) {
    float z4[4];
    sf_normal4(k0, k1, slot + slot_offset, att, (uint32_t)(d0 >> 2), z4);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (d0 + j < a.D) V[(d0 + j) * RS + lane] = z4[j];
  }
  for (int t = a.T - 1; t >= 0; --t) {
    (void)ar_inverse_transform(a, sc, t, E0, V, H1, H2, QB, lane);
    for (int d = 0; d < a.D; ++d) V[d * RS + lane] = E0[d * RS + lane];
  }
  bool ok = active;
  for (int d = 0; d < a.D; ++d) {
    const float th = (V[d * RS + lane] - a.th_shift[d]) / a.th_scale[d];
    V[d * RS + lane] = th;
    ok = ok && (th == th) && fabsf(th) < 3.0e38f && (!lo || (th >= lo[d] && th <= hi[d]));
  }
  return ok;
}

// rejection sampler: a wave works (slot, attempt) items until the slot list is exhausted; rejected items come back first
__global__ __launch_bounds__(64) void k_ar_sample(ArArgs a, const float* __restrict__ x, long S, const uint32_t* __restrict__ slots,
                                                   long n_slots, const float* __restrict__ lo, const float* __restrict__ hi, uint32_t k0,
                                                   uint32_t k1, unsigned long long slot_offset, uint32_t max_attempts,
                                                   float* __restrict__ out, int32_t* __restrict__ n_drawn, int32_t* __restrict__ count,
                                                   unsigned long long* __restrict__ cursor, unsigned int* __restrict__ n_unfilled,
                                                   int32_t* __restrict__ g_try, int32_t* __restrict__ g_acc, unsigned long long walk_R,
                                                   unsigned long long walk_C, uint32_t window_end, uint32_t* __restrict__ surv,
                                                   unsigned int* __restrict__ n_surv) {
  // window_end < max_attempts: an entry that has used up attempts [0, window_end) is appended to surv[] -- the find / resolve
  // launches of sf_nsfar_sample spread its further attempts over the whole chip -- instead of being carried on by this wave
  // cursor[2]: evaluations (items worked), cursor[3]: first attempts rejected -- statistics for sf_flow_sample_stats
  extern __shared__ float lds[];
  float* E0 = lds;
  float* H1 = E0 + a.NIN16 * RS;
  float* H2 = H1 + a.Hp * RS;
  float* QB = H2 + a.Hp * RS;
  float* V = QB + 32 * RS;
  __shared__ unsigned long long r_slot[64];
  __shared__ uint32_t r_att[64];
  const int lane = threadIdx.x;
  const ZSplC sc = {a.K, a.B, a.cw, a.cd};
  for (int r = a.D + a.C; r < a.NIN16; ++r) E0[r * RS + lane] = 0.f;
  // The slot list is walked ACROSS its rows: item i is entry (i % R) * C + i / R of the list seen as R x C (whole catalogue:
  // R = rows, C = draws per row, i.e. draw i / M of row i % M; an explicit list, sorted by slot: R = 4096 pieces) -- the 64 items
  // of a wave then belong to 64 different rows, so a row that accepts one draw in hundreds leaves ONE stubborn entry in many
  // waves instead of 64 in a few.  Once the list has run dry a wave spends its idle lanes on its open entries: W = 2^k <= 64 /
  // entries attempts of each side by side, the LOWEST accepted attempt wins (what the one-at-a-time order would have kept).
  const bool interleave = walk_R > 1;
  const unsigned long long n_index = interleave ? walk_R * walk_C : (unsigned long long)n_slots;   // (>= n_slots: holes are skipped)
  int n_retry = 0;
  bool list_done = false;
  unsigned int n_ev = 0, n_rej0 = 0;
  for (;;) {
    const int take = list_done ? 0 : 64 - n_retry;
    unsigned long long base = n_index;
    if (take > 0) {
      if (lane == 0) base = atomicAdd(cursor, (unsigned long long)take);
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
    if (n_ent == 0) {
      if (lane == 0) {
        if (n_ev) atomicAdd(cursor + 2, (unsigned long long)n_ev);
        if (n_rej0) atomicAdd(cursor + 3, (unsigned long long)n_rej0);
      }
      break;
    }
    int lw = 0;   // log2 of the speculation width
    if (list_done && !count)
      while ((n_ent << (lw + 1)) <= 64) ++lw;
    const int W = 1 << lw;
    const int e = lane >> lw, sub = lane & (W - 1);
    unsigned long long slot = 0;
    uint32_t att0 = 0;
    bool exists = e < n_ent;   // (an entry of this round: a carried one, or a fresh item that is not a hole of the cover)
    if (exists) {
      if (e < n_retry) { slot = r_slot[e]; att0 = r_att[e]; }
      else {
        const unsigned long long idx = base + (unsigned)(e - n_retry);
        const unsigned long long pos = interleave ? (idx % walk_R) * walk_C + idx / walk_R : idx;
        if (pos >= (unsigned long long)n_slots) exists = false;   // (a hole of the R x C cover)
        else slot = slots ? (unsigned long long)slots[pos] : pos;
      }
    }
    const uint32_t att = att0 + (uint32_t)sub;
    const bool active = exists && att < window_end;
    ar_barrier();   // (the retry list has been read)
    const long g = exists ? (long)(slot / (unsigned long long)S) : 0;
    const bool ok = ar_candidate(a, sc, x, g, slot, att, k0, k1, slot_offset, lo, hi, E0, V, H1, H2, QB, lane, active);
    const unsigned long long m_ok = __ballot(ok);
    // (statistics stay in the wave until it leaves: per round they were two more atomics on the cache line of the queue head)
    n_ev += (unsigned)__popcll(__ballot(active));
    n_rej0 += (unsigned)__popcll(__ballot(active && !ok && att == 0u));
    if (count) {   // acceptance counting (leakage correction): one attempt per item, nothing written
      if (ok) atomicAdd(count + g, 1);
      n_retry = 0;
      continue;
    }
    // the entry's lanes: [e W, e W + W); its lowest accepted attempt
    const unsigned long long grp = W == 64 ? m_ok : ((m_ok >> (e * W)) & ((1ull << W) - 1ull));
    const bool resolved = grp != 0ull;
    const int win = resolved ? __builtin_ctzll(grp) : 0;
    const bool leader = exists && sub == 0;
    const uint32_t tried_now = att0 + (uint32_t)W < window_end ? (uint32_t)W : window_end - att0;   // attempts of this round that count
    const bool window_out = leader && !resolved && att0 + (uint32_t)W >= window_end;
    bool give_up = window_out && window_end >= max_attempts;
    if (g_try && leader) {   // no ceiling asked for: a row whose open slots spent 1e5 attempts without ONE accepted draw is written off
      const int tried = atomicAdd(g_try + g, (int)(resolved ? win + 1 : (int)tried_now)) + (int)(resolved ? win + 1 : (int)tried_now);
      if (resolved) atomicAdd(g_acc + g, 1);
      else if (tried >= 100000 && __hip_atomic_load(g_acc + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) give_up = true;
    }
    if (ok && sub == win) {   // the winner writes the draw
      for (int d = 0; d < a.D; ++d) out[slot * a.D + d] = V[d * RS + lane];
      if (n_drawn) sf_sat_add(n_drawn + g, (int32_t)(att + 1u));
    }
    if (give_up) {
      for (int d = 0; d < a.D; ++d) out[slot * a.D + d] = __builtin_nanf("");
      if (n_drawn) sf_sat_add(n_drawn + g, (int32_t)(att0 + tried_now));
      atomicAdd(n_unfilled, 1u);
    }
    if (window_out && !give_up) surv[atomicAdd(n_surv, 1u)] = (uint32_t)slot;   // (hand-over; a row written off by the progress rule is not)
    const bool again = leader && !resolved && !give_up && !window_out;
    const unsigned long long m = __ballot(again);
    if (again) {
      const int pos = __popcll(m & ((1ull << lane) - 1ull));
      r_slot[pos] = slot;
      r_att[pos] = att0 + (uint32_t)W;
    }
    n_retry = __popcll(m);
    ar_barrier();
  }
}

// FIND: attempts [base, base + A) of every survivor side by side -- workgroup (e, j) tries attempts base + 64 j + lane of
// survivor e; an accepted attempt only lowers best[e].  RESOLVE re-evaluates exactly attempt best[e] and writes the draw: the
// slot keeps its LOWEST accepted attempt, whatever A and the schedule were.
__global__ __launch_bounds__(64) void k_ar_find(ArArgs a, const float* __restrict__ x, long S, const uint32_t* __restrict__ surv,
                                                 unsigned int n_surv, uint32_t base, uint32_t chunks, uint32_t att_end,
                                                 const float* __restrict__ lo, const float* __restrict__ hi, uint32_t k0, uint32_t k1,
                                                 unsigned long long slot_offset, uint32_t* __restrict__ best, unsigned long long* __restrict__ ctr) {
  extern __shared__ float lds[];
  float* E0 = lds;
  float* H1 = E0 + a.NIN16 * RS;
  float* H2 = H1 + a.Hp * RS;
  float* QB = H2 + a.Hp * RS;
  float* V = QB + 32 * RS;
  const int lane = threadIdx.x;
  const ZSplC sc = {a.K, a.B, a.cw, a.cd};
  for (int r = a.D + a.C; r < a.NIN16; ++r) E0[r * RS + lane] = 0.f;
  const unsigned int e = blockIdx.x / chunks, j = blockIdx.x - e * chunks;
  if (e >= n_surv) return;
  const unsigned long long slot = surv[e];
  const uint32_t att = base + 64u * j + (uint32_t)lane;
  const bool active = att < att_end;
  const long g = (long)(slot / (unsigned long long)S);
  const bool ok = ar_candidate(a, sc, x, g, slot, att, k0, k1, slot_offset, lo, hi, E0, V, H1, H2, QB, lane, active);
  if (ok) atomicMin(best + e, att);
  const unsigned long long ma = __ballot(active);
  if (lane == 0) atomicAdd(ctr + 2, (unsigned long long)__popcll(ma));
}
__global__ __launch_bounds__(64) void k_ar_resolve(ArArgs a, const float* __restrict__ x, long S, const uint32_t* __restrict__ surv,
                                                    unsigned int n_surv, const uint32_t* __restrict__ best, uint32_t tried_end,
                                                    uint32_t max_attempts, const float* __restrict__ lo, const float* __restrict__ hi,
                                                    uint32_t k0, uint32_t k1, unsigned long long slot_offset, float* __restrict__ out,
                                                    int32_t* __restrict__ n_drawn, int32_t* __restrict__ g_try, int32_t* __restrict__ g_acc,
                                                    uint32_t tried_now, uint32_t* __restrict__ next, unsigned int* __restrict__ n_next,
                                                    unsigned int* __restrict__ n_unfilled) {
  extern __shared__ float lds[];
  float* E0 = lds;
  float* H1 = E0 + a.NIN16 * RS;
  float* H2 = H1 + a.Hp * RS;
  float* QB = H2 + a.Hp * RS;
  float* V = QB + 32 * RS;
  const int lane = threadIdx.x;
  const ZSplC sc = {a.K, a.B, a.cw, a.cd};
  for (int r = a.D + a.C; r < a.NIN16; ++r) E0[r * RS + lane] = 0.f;
  const unsigned int e = blockIdx.x * 64u + (unsigned)lane;
  const bool exists = e < n_surv;
  const unsigned long long slot = exists ? surv[e] : 0ull;
  const uint32_t b = exists ? best[e] : 0xffffffffu;
  const bool found = exists && b != 0xffffffffu;
  const long g = exists ? (long)(slot / (unsigned long long)S) : 0;
  const bool ok = ar_candidate(a, sc, x, g, slot, found ? b : 0u, k0, k1, slot_offset, lo, hi, E0, V, H1, H2, QB, lane, found);
  if (found) {   // (ok by construction: the find launch accepted this very attempt)
    for (int d = 0; d < a.D; ++d) out[slot * a.D + d] = ok ? V[d * RS + lane] : __builtin_nanf("");
    if (n_drawn) sf_sat_add(n_drawn + g, (int32_t)(b + 1u));
    if (g_acc) atomicAdd(g_acc + g, 1);
    if (g_try) atomicAdd(g_try + g, (int)tried_now);
  } else if (exists) {
    bool give_up = tried_end >= max_attempts;
    if (g_try) {   // (uncapped: the progress rule, as in k_ar_sample)
      const int tried = atomicAdd(g_try + g, (int)tried_now) + (int)tried_now;
      if (tried >= 100000 && __hip_atomic_load(g_acc + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) give_up = true;
    }
    if (give_up) {
      for (int d = 0; d < a.D; ++d) out[slot * a.D + d] = __builtin_nanf("");
      if (n_drawn) sf_sat_add(n_drawn + g, (int32_t)tried_end);
      atomicAdd(n_unfilled, 1u);
    } else {
      next[atomicAdd(n_next, 1u)] = (uint32_t)slot;
    }
  }
}

// forward (with the inputs of every transform stashed) + loss, then the backward sweep
// Two workgroups per CU (round 5): 24 head rows per wave -> 74.6 KB of LDS for the bench shape, the four-wave form held to 256 registers
// (219, no scratch): 3.1 -> 2.3 ms per 131 072 rows (one 64-row chunk per workgroup: only batches above 64 x CUs rows gain).  The first
// build of it computed WRONG losses on thousands of rows whenever two workgroups really shared a CU: a latent race since round 4 (the
// barrier behind the loss, see there) that one workgroup per CU never lost.  -DSF_AR_TRAIN_WGS=1 -DSF_AR_QBR=32 rebuilds the old form.
#ifndef SF_AR_TRAIN_WGS
#define SF_AR_TRAIN_WGS 2
#endif
#ifndef SF_AR_QBR
#define SF_AR_QBR 24
#endif
template <int NWV, bool PART>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? SF_AR_TRAIN_WGS : 1) void k_ar_train(ArArgs a, const float* __restrict__ theta, const float* __restrict__ x,
                                                        const long long* __restrict__ idx, long B, float w, const float* __restrict__ wts,
                                                        float* __restrict__ loss, double* __restrict__ loss_sum, float* __restrict__ grad_in,
                                                        long part_stride, float* __restrict__ ustash) {
  extern __shared__ float lds[];
  float* grad = grad_in + (PART ? (size_t)blockIdx.x * part_stride : 0);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float* E0 = lds;                       // [NIN16]: u, context, zeros
  float* H1 = E0 + a.NIN16 * RS;         // [Hp]
  float* H2 = H1 + a.Hp * RS;            // [Hp]
  float* QB0 = H2 + a.Hp * RS;           // [NWV][32] per wave: head outputs of ONE dimension, then their deltas; wave 0: the input deltas
  // (24 rows per wave: the head's 24 slots.  The 16-row tile routines read rows 24..31 of a wave's second tile -- the next wave's
  //  buffer, or GG behind the last -- and everything computed from them lands in output rows that are never stored)
  constexpr int QBR = SF_AR_QBR;
  float* QB = QB0 + wid * QBR * RS;
  // TWO hidden buffers serve the backward sweep (round 5; three before: 102 KB for cfg1's shape, one workgroup per CU, and the
  // reference's own lampe example -- 180 hidden units, examples/sbi/scripts/basic_model.py:31-41 -- did not fit at all):
  //   head phase     H2 = last hidden layer (operand of the head's weight gradients), DH = its delta, accumulated in H1's rows
  //                  (H1 is dead once H2 exists; the samples where it was positive are kept as BITS, two words per row);
  //   second layer   H1 is RECOMPUTED into H2's rows (dead after it has gated DH): weight gradients DH x H1;
  //   first layer    delta_h1 = L1m^T DH overwrites the recomputed H1, gated by the bits.
  float* DH = H1;                        // (alias: see above)
  float* GG = QB0 + NWV * QBR * RS;      // [D] dL/du at the transform's output
  float* DV = GG + a.D * RS;             // [D] what reaches the transform's input through the splines
  int* PERM = reinterpret_cast<int*>(DV + a.D * RS);   // [Hp] perm, [Hp] ptype: read per weight-gradient block
  int* PTYP = PERM + a.Hp;
  unsigned int* M1 = reinterpret_cast<unsigned int*>(PTYP + a.Hp);   // [Hp][2] samples with H1 > 0
  for (int i = threadIdx.x; i < a.Hp; i += 64 * NWV) { PERM[i] = a.perm[i]; PTYP[i] = a.ptype[i]; }
  const ZSplC sc = {a.K, a.B, a.cw, a.cd};
  // (the launches give every workgroup ONE chunk of 64 rows; a workgroup that is given more -- a smaller grid -- stores its first
  //  chunk's gradient into its partial and adds the later ones with f32 atomics nobody contends for)
  const long n_chunks = (B + 63) / 64;
  for (long chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
  const bool first_chunk = chunk == (long)blockIdx.x;
  (void)first_chunk;
  if (!first_chunk) ar_barrier();   // (the previous chunk's last reads of the LDS rows)
  const long b = chunk * 64 + lane;
  const bool valid = b < B;
  const long bb = valid ? b : B - 1;
  const long row = idx ? (long)idx[bb] : bb;
  AR_TS(0);
  ar_load_inputs(a, theta, x, row, E0, lane, wid, NWV);
  float* ust = ustash + (size_t)bb * a.T * a.D;   // (an invalid lane shares the last row's stash: same values)
  float ld = 0.f;
  for (int t = 0; t < a.T; ++t) {
    const float* tp = a.img + (size_t)t * a.t_stride;
    ar_barrier();   // (E0 complete)
    for (int d = wid; d < a.D; d += NWV) ust[t * a.D + d] = E0[d * RS + lane];
    ar_hidden(a, tp, E0, H1, H2, lane, wid, NWV);
    for (int d = 0; d < a.D; ++d) {
      if (NWV > 1 && a.dwave[t * a.D + d] != wid) continue;
      float q[ARQ];
      ar_head(a, tp, d, (int)a.tendk[a.ord[t * a.D + d]], H2, QB, lane, q);
      float v, lad;
      ar_uni_fwd(a, sc, q, E0[d * RS + lane], v, lad);
      E0[d * RS + lane] = v;
      ld += lad;
    }
  }
  ar_barrier();
  if (NWV > 1) {   // the log-determinants of the waves' dimensions
    QB[lane] = ld;
    ar_barrier();
    if (wid == 0)
      for (int w2 = 1; w2 < NWV; ++w2) ld += QB0[w2 * QBR * RS + lane];
  }
  if (wid == 0) {
    float ss = 0.f;
    for (int d = 0; d < a.D; ++d) ss += E0[d * RS + lane] * E0[d * RS + lane];
    const float nll = 0.5f * ss + 0.5f * (float)a.D * 1.8378770664093453f - (ld + a.logdet0);
    if (loss && valid) loss[b] = nll;
    if (loss_sum) {
      float tsum = valid ? nll : 0.f;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) tsum += __shfl_xor(tsum, o, 64);
      // values on a 2^-20 grid add exactly in double: the sum does not depend on the order of the atomics
      if (lane == 0) atomicAdd(loss_sum, (double)rintf(tsum * 1048576.0f) * (1.0 / 1048576.0));
    }
  }
  const float wb = valid ? (wts ? w * wts[b] : w) : 0.f;
  for (int d = wid; d < a.D; d += NWV) GG[d * RS + lane] = wb * E0[d * RS + lane];
  // wave 0 has summed EVERY row of E0 for the loss above; the backward sweep below starts by putting the last transform's inputs back
  // into the rows of the other waves.  Without this barrier a wave that runs ahead overwrites rows wave 0 has not read yet: the
  // gradients stay right (every wave has taken its own rows for GG), the LOSS of some rows does not -- never seen with one workgroup
  // per CU (wave 0's path is the short one), thousands of rows with two (round 5: the "two workgroups per CU" experiment).
  if (NWV > 1) ar_barrier();
  AR_TS(98);
  const int nin = a.D + a.C;
  for (int t = a.T - 1; t >= 0; --t) {
    const float* tp = a.img + (size_t)t * a.t_stride;
    float* gt = grad + (size_t)t * a.P_t;
    for (int d = wid; d < a.D; d += NWV) E0[d * RS + lane] = ust[t * a.D + d];
    AR_TS(99);
    ar_hidden(a, tp, E0, H1, H2, lane, wid, NWV);
    AR_TS(100);
    ar_sign_bits(H1, M1, a.Hp, lane, wid, NWV);   // (same rows per wave as the clearing below: no barrier between the two)
    for (int r0 = 8 * wid; r0 < a.Hp; r0 += 8 * NWV)
#pragma unroll
      for (int i = 0; i < 8; ++i) DH[(r0 + i) * RS + lane] = 0.f;
    ar_barrier();
    AR_TS(101);
    // ---- head + splines, dimension by dimension (wave wid: dimensions wid, wid + NWV, ...)
    for (int d = 0; d < a.D; ++d) {
      if (NWV > 1 && a.dwave[t * a.D + d] != wid) continue;
      const int kend = (int)a.tendk[a.ord[t * a.D + d]];
      float q[ARQ], dq[ARQ];
      ar_head(a, tp, d, kend, H2, QB, lane, q);
      AR_TS(102 + 4 * d);
      float dv;
      ar_uni_bwd(a, sc, q, E0[d * RS + lane], GG[d * RS + lane], -wb, dv, dq);
      AR_TS(103 + 4 * d);
      DV[d * RS + lane] = dv;
      dq[ARQ - 1] = 0.f;
      ar_wave_sync();   // (every lane has taken its q)
#pragma unroll
      for (int sl = 0; sl < ARQ; ++sl) QB[sl * RS + lane] = dq[sl];
      ar_wave_sync();
      // weight gradients of this dimension's head rows: (24 slots x 64 samples) x (64 samples x kend hidden rows)
      for (int it = 0; it < 2; ++it) {
        for (int k0 = 0; k0 < kend; k0 += 16) {
          const ar_f32x4 g4 = ar_dw16(QB, it * 16, H2, k0, lane);
          const int k = k0 + (lane & 15);
          const int kl = k < kend ? PERM[k] : -1;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int lr = ar_slot_row(a, it * 16 + 4 * (lane >> 4) + r);
            if (kl >= 0 && lr >= 0) AR_GADD(gt + a.l_W2 + (size_t)(d * a.NP + lr) * a.H + kl, g4[r]);
          }
        }
        const ar_f32x4 b4 = ar_rowsum16(QB, it * 16, lane);
        if ((lane & 15) == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int lr = ar_slot_row(a, it * 16 + 4 * (lane >> 4) + r);
            if (lr >= 0) AR_GADD(gt + a.l_b2 + d * a.NP + lr, b4[r]);
          }
        }
      }
      AR_TS(104 + 4 * d);
      // delta of the last hidden layer: DH[k] += sum_slots W2[(d, slot)][k] dq[slot]   (k-major image of the head's transpose)
      {
        const float* wm = tp + a.o_L2m + (size_t)d * ARQ * a.Hp;
        int p0 = 0;
        for (; p0 + 16 < kend; p0 += 32) ar_tiles<false, true, 2, (NWV > 1)>(wm, a.Hp, nullptr, p0, 0, ARQ, QB, DH, a.Hp, lane);
        if (p0 < kend) ar_tiles<false, true, 1, (NWV > 1)>(wm, a.Hp, nullptr, p0, 0, ARQ, QB, DH, a.Hp, lane);
      }
      AR_TS(105 + 4 * d);
    }
    // ---- second hidden layer: delta through the ReLU, weight gradients, delta of the first hidden layer (into H2's rows)
    ar_barrier();
    ar_mask_rows(DH, H2, a.Hp, lane, wid, NWV);   // DH <- DH where H2 > 0
    ar_barrier();
    float* H1r = H2;                              // H2 is dead: the first hidden layer once more, into its rows
    ar_hidden0(a, tp, E0, H1r, lane, wid, NWV);
    ar_barrier();
    AR_TS(130);
    for (int o0 = 16 * wid; o0 < a.Hp; o0 += 16 * NWV) {
      const int kend = (int)a.tile_kend[o0 >> 4];
      int ol[4], oty[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { ol[r] = PERM[o0 + 4 * (lane >> 4) + r]; oty[r] = PTYP[o0 + 4 * (lane >> 4) + r]; }
      for (int k0 = 0; k0 < kend; k0 += 16) {
        const ar_f32x4 g4 = ar_dw16(DH, o0, H1r, k0, lane);
        const int k = k0 + (lane & 15), kl = PERM[k], kty = PTYP[k];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kl >= 0 && ol[r] >= 0 && kty <= oty[r]) AR_GADD(gt + a.l_W1 + (size_t)ol[r] * a.H + kl, g4[r]);
      }
      const ar_f32x4 b4 = ar_rowsum16(DH, o0, lane);
      if ((lane & 15) == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ol[r] >= 0) AR_GADD(gt + a.l_b1 + ol[r], b4[r]);
      }
    }
    ar_barrier();   // (the recomputed H1 is overwritten next)
    AR_TS(131);
    // delta_h1[k] = sum_{o: type(o) >= type(k)} W1[o][k] delta_h2[o]
    for (int p0 = 32 * wid; p0 < a.Hp; p0 += 32 * NWV) {
      if (p0 + 16 < a.Hp) ar_tiles<false, false, 2>(tp + a.o_L1m, a.Hp, nullptr, p0, (int)a.tile_kbeg[p0 >> 4], a.Hp, DH, H2, a.Hp, lane);
      else ar_tiles<false, false, 1>(tp + a.o_L1m, a.Hp, nullptr, p0, (int)a.tile_kbeg[p0 >> 4], a.Hp, DH, H2, a.Hp, lane);
    }
    ar_barrier();
    AR_TS(132);
    ar_mask_bits(H2, M1, a.Hp, lane, wid, NWV);   // H2 (delta_h1) <- where H1 > 0 (the bits taken before DH took its rows)
    ar_barrier();
    AR_TS(133);
    // ---- first hidden layer: weight gradients, and what reaches the inputs
    for (int o0 = 16 * wid; o0 < a.Hp; o0 += 16 * NWV) {
      int ol[4], oty[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { ol[r] = PERM[o0 + 4 * (lane >> 4) + r]; oty[r] = PTYP[o0 + 4 * (lane >> 4) + r]; }
      for (int i0 = 0; i0 < a.NIN16; i0 += 16) {
        const ar_f32x4 g4 = ar_dw16(H2, o0, E0, i0, lane);
        const int i = i0 + (lane & 15);
        const int io = i < a.D ? a.ord[t * a.D + i] : -1;   // (context columns: seen by every unit)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (i < nin && ol[r] >= 0 && io < oty[r]) AR_GADD(gt + a.l_W0 + (size_t)ol[r] * nin + i, g4[r]);
      }
      const ar_f32x4 b4 = ar_rowsum16(H2, o0, lane);
      if ((lane & 15) == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ol[r] >= 0) AR_GADD(gt + a.l_b0 + ol[r], b4[r]);
      }
    }
    AR_TS(134);
    // d input[i] = sum_o W0[o][i] delta_h1[o], i < D  (L0m: [o][16]); the last wave has the fewest weight-gradient tiles
    if (wid == NWV - 1) ar_tiles<false, false, 1>(tp + a.o_L0m, 16, nullptr, 0, 0, a.Hp, H2, QB0, 16, lane);
    ar_barrier();
    AR_TS(135);
    for (int d = wid; d < a.D; d += NWV) GG[d * RS + lane] = DV[d * RS + lane] + QB0[d * RS + lane];
  }
  }   // chunks
}

// grad[i] = sum over the workgroups' partials, in workgroup order; 0 where no workgroup writes (masked weights: live[i] = 0)
__global__ __launch_bounds__(256) void k_ar_gather(const float* __restrict__ part, long stride, int nwg, const unsigned char* __restrict__ live,
                                                   float* __restrict__ grad, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (live[i]) {
    const float* p = part + i;
    int w = 0;
    for (; w + 4 <= nwg; w += 4) {
      s0 += p[(size_t)w * stride];
      s1 += p[(size_t)(w + 1) * stride];
      s2 += p[(size_t)(w + 2) * stride];
      s3 += p[(size_t)(w + 3) * stride];
    }
    for (; w < nwg; ++w) s0 += p[(size_t)w * stride];
  }
  grad[i] = (s0 + s1) + (s2 + s3);
}

#define AR_HIP(call)                                                        \
  do {                                                                      \
    hipError_t e_ = (call);                                                 \
    if (e_ != hipSuccess) {                                                 \
      err = std::string(#call) + ": " + hipGetErrorString(e_);              \
      return SF_ERR_HIP;                                                    \
    }                                                                       \
  } while (0)

ArArgs args_of(const SfNsfAr& n) {
  ArArgs a;
  a.img = n.d_img; a.perm = n.d_perm; a.ptype = n.d_ptype; a.tend = n.d_tend; a.ord = n.d_ord; a.dimof = n.d_dimof; a.dwave = n.d_dwave;
  a.xmean = n.d_xmean; a.xstd = n.d_xstd;
  a.affine = n.affine;
  a.D = n.D; a.C = n.C; a.H = n.H; a.Hp = n.Hp; a.T = n.T; a.K = n.K; a.NP = n.NP; a.NIN16 = (n.D + n.C + 15) / 16 * 16; a.NIN4 = (n.D + n.C + 3) / 4 * 4;
  a.t_stride = n.t_stride;
  a.o_L0t = n.o_L0t; a.o_b0 = n.o_b0; a.o_L1t = n.o_L1t; a.o_L1m = n.o_L1m; a.o_b1 = n.o_b1; a.o_L2t = n.o_L2t; a.o_b2 = n.o_b2; a.o_L0m = n.o_L0m; a.o_L2m = n.o_L2m;
  a.P_t = n.P_t; a.l_W0 = n.l_W0; a.l_b0 = n.l_b0; a.l_W1 = n.l_W1; a.l_b1 = n.l_b1; a.l_W2 = n.l_W2; a.l_b2 = n.l_b2;
  a.B = n.bound; a.cw = n.cw; a.cd = n.cd; a.logdet0 = n.logdet0;
  for (int d = 0; d < 16; ++d) { a.th_scale[d] = n.th_scale[d]; a.th_shift[d] = n.th_shift[d]; a.tendk[d] = d < n.D ? (short)n.tend[d] : (short)n.Hp; }
  for (int i = 0; i < 24; ++i) { a.tile_kend[i] = 0; a.tile_kbeg[i] = 0; }
#ifdef SF_AR_TRACE
  a.trace = nullptr;
#endif
  for (int p0 = 0; p0 < n.Hp; p0 += 16) {
    const int ta = n.ptype[p0], tb = n.ptype[p0 + 15];
    a.tile_kend[p0 >> 4] = (short)n.tend[tb];
    a.tile_kbeg[p0 >> 4] = (short)(ta ? n.tend[ta - 1] : 0);
  }
  return a;
}

template <typename Kern>
hipError_t set_lds(Kern k, size_t bytes) {
  return hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

int sf_nsfar_create(const sf_flow_desc& d, SfNsfAr** out, std::string& err) {
  if (d.D < 1 || d.D > 16) { err = "autoregressive NSF: D must be in 1..16"; return SF_ERR_INVALID; }
  const bool affine = d.kind == SF_MAF_AR;
  if (!affine && (d.K < 2 || d.K > ARK)) { err = "autoregressive NSF: K (bins) must be in 2..8"; return SF_ERR_INVALID; }
  if (d.NB != 2) { err = "autoregressive NSF: two hidden layers (NB = 2: lampe / ltu-ili's hyper-network) are built"; return SF_ERR_INVALID; }
  if (d.H < d.D || d.H > 192) { err = "autoregressive NSF: H must be in D..192"; return SF_ERR_INVALID; }
  if (d.C < 1 || d.C > 256 || d.T < 1 || d.T > 64) { err = "autoregressive NSF: C in 1..256, T in 1..64"; return SF_ERR_INVALID; }
  if (!d.theta_mean || !d.theta_std || !d.x_mean || !d.x_std) { err = "z-score buffers must be given"; return SF_ERR_INVALID; }
  if (!(d.ar_slope > 0.f && d.ar_slope < 1.f)) { err = "ar_slope must be in (0, 1)"; return SF_ERR_INVALID; }
  SfNsfAr* n = new SfNsfAr();
  // (affine = zuko MAF: two parameters per dimension, carried in slots 0 and 1 of family 0 -- "K = 2, widths only")
  const int D = d.D, C = d.C, H = d.H, T = d.T, K = affine ? 2 : d.K, NP = affine ? 2 : 3 * K - 1;
  n->D = D; n->C = C; n->H = H; n->T = T; n->K = K; n->NP = NP; n->affine = affine ? 1 : 0;
  n->bound = d.tail_bound;
  const double ls = std::fabs(std::log((double)d.ar_slope));
  n->cw = (float)(2.0 / ls); n->cd = (float)(1.0 / ls);
  double ld0 = 0;
  for (int i = 0; i < 16; ++i) { n->th_scale[i] = 1.f; n->th_shift[i] = 0.f; }
  for (int i = 0; i < D; ++i) {
    n->th_scale[i] = 1.0f / d.theta_std[i];
    n->th_shift[i] = -d.theta_mean[i] / d.theta_std[i];
    ld0 += std::log(std::fabs(1.0 / (double)d.theta_std[i]));
  }
  n->logdet0 = (float)ld0;
  n->h_xmean.assign(d.x_mean, d.x_mean + C);
  n->h_xstd.assign(d.x_std, d.x_std + C);
  // hidden rows sorted by type (unit h: type h mod D), every type padded to a multiple of eight rows
  n->perm.clear(); n->ptype.clear(); n->tend.assign(D, 0);
  for (int r = 0; r < D; ++r) {
    int cnt = 0;
    for (int h = r; h < H; h += D) { n->perm.push_back(h); n->ptype.push_back(r); ++cnt; }
    while (cnt % 8) { n->perm.push_back(-1); n->ptype.push_back(r); ++cnt; }
    n->tend[r] = (int)n->perm.size();
  }
  if (n->perm.size() % 16) {   // whole 16-row tiles for the MFMA blocks of the training kernel: one more padding block of the last type
    for (int i = 0; i < 8; ++i) { n->perm.push_back(-1); n->ptype.push_back(D - 1); }
    n->tend[D - 1] = (int)n->perm.size();
  }
  const int Hp = (int)n->perm.size();
  n->Hp = Hp;
  // logical layout of a transform
  const int nin = D + C;
  n->l_W0 = 0; n->l_b0 = n->l_W0 + H * nin; n->l_W1 = n->l_b0 + H; n->l_b1 = n->l_W1 + H * H;
  n->l_W2 = n->l_b1 + H; n->l_b2 = n->l_W2 + D * NP * H; n->P_t = n->l_b2 + D * NP;
  n->n_params = (int64_t)T * n->P_t;
  // images of a transform (offsets in floats, every block a multiple of 8)
  long o = 0;
  n->o_L0t = (int)o; o += (long)((nin + 3) / 4 * 4) * Hp;   // (k-steps of four: zero rows behind the last input)
  n->o_b0 = (int)o; o += Hp;
  n->o_L1t = (int)o; o += (long)Hp * Hp;
  n->o_L1m = (int)o; o += (long)Hp * Hp;
  n->o_b1 = (int)o; o += Hp;
  n->o_L2t = (int)o; o += (long)Hp * D * ARQ;
  n->o_b2 = (int)o; o += (long)D * ARQ;
  n->o_L0m = (int)o; o += (long)Hp * 16;
  n->o_L2m = (int)o; o += (long)D * ARQ * Hp;   // the head transposed: row (d, slot), column k
  // the 16-sample sampler's blocks (sf_nsfar16.hip): a type fits one 16-row tile, or two (17..32 units per type: the reference's lampe
  // example), D steps, one or two input tiles
  {
    const int cnt_max = (H + D - 1) / D;
    n->s16_tpt = cnt_max <= 16 ? 1 : 2;
    const int per_tile = (cnt_max + n->s16_tpt - 1) / n->s16_tpt;
    n->s16_ks = (per_tile + 3) / 4 < 2 ? 2 : (per_tile + 3) / 4;
    if (n->s16_tpt == 2 && n->s16_ks < 3) n->s16_ks = 3;   // (two-tile types are built for three and four k-steps)
    n->s16_ni = (nin + 15) / 16;
    n->s16_nt = (cnt_max <= 32 && D >= 2 && D <= 8 && n->s16_ni <= 2) ? D * n->s16_tpt : 0;
  }
  if (n->s16_nt) {
    const long NTs = n->s16_nt, NI = n->s16_ni;
    n->o_F0 = (int)o; o += NTs * NI * 256;
    n->o_fb0 = (int)o; o += NTs * 16;
    n->o_F1 = (int)o; o += NTs * NTs * 256;
    n->o_fb1 = (int)o; o += NTs * 16;
    n->o_F2 = (int)o; o += (long)D * 2 * NTs * 256;
  }
  n->t_stride = (o + 63) / 64 * 64;
  if (std::max(sf_nsfar_lds_bytes(*n, 3, 1), sf_nsfar_lds_bytes(*n, 2, 1)) > (size_t)160 * 1024 - 1024) {
    err = "autoregressive NSF: (3 D + C + 2 Hp + 32) x 260 bytes of LDS per wave (training: 24 head rows + 16 Hp bytes of tables) exceed the 160 KB of a CU (Hp = H with every type padded to a multiple of 8, in all a multiple of 16)";
    delete n;
    return SF_ERR_INVALID;
  }
  n->src.assign((size_t)T * n->t_stride, -1);
  n->ord.assign((size_t)T * D, 0); n->dimof.assign((size_t)T * D, 0); n->dwave.assign((size_t)T * D, 0);
  for (int t = 0; t < T; ++t) {
    int32_t* s = n->src.data() + (size_t)t * n->t_stride;
    const long base = (long)t * n->P_t;
    for (int dd = 0; dd < D; ++dd) {
      const int r = (t % 2 == 0) ? dd : D - 1 - dd;
      n->ord[(size_t)t * D + dd] = r;
      n->dimof[(size_t)t * D + r] = dd;
    }
    const int32_t* ord = n->ord.data() + (size_t)t * D;
    {   // Dimensions over the four waves of k_ar_logprob<4> / k_ar_train<4>: the head, spline and (training) gradient work of a dimension
        // grows with the hidden rows it reads (tend[ord]) -- the one ordered last costs five times the first at D = 5.  Longest first
        // onto the least loaded wave (round robin gave one wave the cheapest AND the dearest: 6 units of 15 against 5).
      std::vector<int> by_cost(D);
      for (int dd = 0; dd < D; ++dd) by_cost[dd] = dd;
      std::sort(by_cost.begin(), by_cost.end(), [&](int x, int y) { return ord[x] != ord[y] ? ord[x] > ord[y] : x < y; });
      long load[4] = {0, 0, 0, 0};
      for (int dd : by_cost) {
        int w = 0;
        for (int k = 1; k < 4; ++k)
          if (load[k] < load[w]) w = k;
        n->dwave[(size_t)t * D + dd] = w;
        load[w] += n->tend[ord[dd]] + 24;   // (+ the spline: the same for every dimension)
      }
    }
    for (int p = 0; p < Hp; ++p) {
      const int h = n->perm[p], ty = n->ptype[p];
      if (h < 0) continue;
      for (int i = 0; i < nin; ++i) {
        const bool on = i >= D || ord[i] < ty;
        if (!on) continue;
        s[n->o_L0t + (long)i * Hp + p] = (int32_t)(base + n->l_W0 + (long)h * nin + i);
        if (i < D) s[n->o_L0m + (long)p * 16 + i] = (int32_t)(base + n->l_W0 + (long)h * nin + i);
      }
      s[n->o_b0 + p] = (int32_t)(base + n->l_b0 + h);
      s[n->o_b1 + p] = (int32_t)(base + n->l_b1 + h);
      for (int k = 0; k < Hp; ++k) {
        const int hk = n->perm[k];
        if (hk < 0 || n->ptype[k] > ty) continue;
        s[n->o_L1t + (long)k * Hp + p] = (int32_t)(base + n->l_W1 + (long)h * H + hk);
        s[n->o_L1m + (long)p * Hp + k] = (int32_t)(base + n->l_W1 + (long)h * H + hk);
      }
    }
    for (int dd = 0; dd < D; ++dd)
      for (int fam = 0; fam < (affine ? 1 : 3); ++fam)
        for (int kk = 0; kk < (fam < 2 ? K : K - 1); ++kk) {
          const int slot = dd * ARQ + fam * 8 + kk, lrow = dd * NP + fam * K + kk;
          s[n->o_b2 + slot] = (int32_t)(base + n->l_b2 + lrow);
          for (int k = 0; k < Hp; ++k) {
            const int hk = n->perm[k];
            if (hk < 0 || n->ptype[k] > ord[dd]) continue;
            s[n->o_L2t + (long)k * D * ARQ + slot] = (int32_t)(base + n->l_W2 + (long)lrow * H + hk);
            s[n->o_L2m + (long)slot * Hp + k] = (int32_t)(base + n->l_W2 + (long)lrow * H + hk);
          }
        }
    if (n->s16_nt) {   // fragment blocks in the sampler's hidden order (type r = tile r, or tiles 2 r and 2 r + 1), -1 on the rows that hold no unit
      const int NTs = n->s16_nt, NI = n->s16_ni, TPT = n->s16_tpt;
      const int KS = n->s16_ks;   // a tile holds 4 KS units: the n-th unit of its tile sits on row 4 (n / KS) + n % KS
      auto unit = [&](int p) {
        const int tl = p >> 4, r = tl / TPT, sub = tl % TPT, i = p & 15, j = i & 3;
        if (j >= KS) return -1;
        const int h = r + (sub * 4 * KS + (i >> 2) * KS + j) * D;
        return h < H ? h : -1;
      };
      for (int ot = 0; ot < NTs; ++ot)
        for (int l = 0; l < 64; ++l) {
          const int p = 16 * ot + (l & 15), h = unit(p), ty = ot / TPT;
          if (h < 0) continue;
          if (l < 16) {
            s[n->o_fb0 + p] = (int32_t)(base + n->l_b0 + h);
            s[n->o_fb1 + p] = (int32_t)(base + n->l_b1 + h);
          }
          for (int j = 0; j < 4; ++j) {
            for (int ti = 0; ti < NI; ++ti) {
              const int i = 16 * ti + 4 * (l >> 4) + j;
              if (i < nin && (i >= D || ord[i] < ty)) s[n->o_F0 + (long)(ot * NI + ti) * 256 + l * 4 + j] = (int32_t)(base + n->l_W0 + (long)h * nin + i);
            }
            for (int kt = 0; kt < (ty + 1) * TPT; ++kt) {   // (type of row k = kt / TPT <= type of row p)
              const int hk = unit(16 * kt + 4 * (l >> 4) + j);
              if (hk >= 0) s[n->o_F1 + (long)(ot * NTs + kt) * 256 + l * 4 + j] = (int32_t)(base + n->l_W1 + (long)h * H + hk);
            }
          }
        }
      for (int dd = 0; dd < D; ++dd)
        for (int o2 = 0; o2 < 2; ++o2)
          for (int l = 0; l < 64; ++l) {
            const int slot = 16 * o2 + (l & 15), fam = slot >> 3, kk = slot & 7;
            if (slot >= ARQ || fam >= (affine ? 1 : 3) || kk >= (fam < 2 ? K : K - 1)) continue;
            const int lrow = dd * NP + fam * K + kk;
            for (int kt = 0; kt < NTs; ++kt) {
              if (kt / TPT > ord[dd]) continue;
              for (int j = 0; j < 4; ++j) {
                const int hk = unit(16 * kt + 4 * (l >> 4) + j);
                if (hk >= 0) s[n->o_F2 + (long)((dd * 2 + o2) * NTs + kt) * 256 + l * 4 + j] = (int32_t)(base + n->l_W2 + (long)lrow * H + hk);
              }
            }
          }
    }
  }
  *out = n;
  return SF_OK;
}

void sf_nsfar_destroy(SfNsfAr* n) {
  if (!n) return;
  (void)hipFree(n->d_img); (void)hipFree(n->d_src); (void)hipFree(n->d_none); (void)hipFree(n->d_perm); (void)hipFree(n->d_ptype); (void)hipFree(n->d_tend);
  (void)hipFree(n->d_ord); (void)hipFree(n->d_dimof); (void)hipFree(n->d_dwave); (void)hipFree(n->d_xmean); (void)hipFree(n->d_xstd); (void)hipFree(n->d_ustash);
  (void)hipFree(n->d_ctr); (void)hipHostFree(n->h_ctr); (void)hipFree(n->d_live); (void)hipFree(n->d_gpart); (void)hipFree(n->d_gal); (void)hipFree(n->d_surv[0]); (void)hipFree(n->d_surv[1]); (void)hipFree(n->d_best);
  delete n;
}

static int ar_ensure(SfNsfAr* n, std::string& err) {
  if (n->dev_ready) return SF_OK;
  auto up = [&](auto*& dst, const auto& v) -> hipError_t {
    using T = typename std::remove_reference<decltype(v[0])>::type;
    hipError_t e = hipMalloc(&dst, v.size() * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemcpy(dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  };
  AR_HIP(hipMalloc(&n->d_img, n->src.size() * sizeof(float)));
  AR_HIP(up(n->d_src, n->src)); AR_HIP(up(n->d_perm, n->perm)); AR_HIP(up(n->d_ptype, n->ptype)); AR_HIP(up(n->d_tend, n->tend));
  AR_HIP(up(n->d_ord, n->ord)); AR_HIP(up(n->d_dimof, n->dimof)); AR_HIP(up(n->d_dwave, n->dwave)); AR_HIP(up(n->d_xmean, n->h_xmean)); AR_HIP(up(n->d_xstd, n->h_xstd));
  {   // live[i] = 1 where logical parameter i appears in an image (an unmasked weight or a bias): the entries a training workgroup writes
    std::vector<unsigned char> live((size_t)n->n_params, 0);
    for (int32_t v : n->src)
      if (v >= 0) live[(size_t)v] = 1;
    AR_HIP(up(n->d_live, live));
  }
  AR_HIP(hipMalloc(&n->d_ctr, 8 * sizeof(unsigned long long)));
  AR_HIP(hipHostMalloc((void**)&n->h_ctr, 8 * sizeof(unsigned long long), hipHostMallocDefault));   // [0] cursor [1] unfilled [2] evaluations [3] rejected first attempts [4], [5] survivor counts
  {
    const std::vector<int32_t> none(n->src.size(), -1);
    AR_HIP(up(n->d_none, none));
  }
  const size_t lds = (size_t)160 * 1024 - 1024;   // (k_ar_sample also has 768 static bytes)
  AR_HIP(set_lds(k_ar_logprob<1>, lds)); AR_HIP(set_lds(k_ar_logprob<4>, lds)); AR_HIP(set_lds(k_ar_inverse, lds)); AR_HIP(set_lds(k_ar_sample, lds));
  AR_HIP(set_lds(k_ar_train<1, false>, lds)); AR_HIP(set_lds(k_ar_train<4, false>, lds));
  AR_HIP(set_lds(k_ar_train<1, true>, lds)); AR_HIP(set_lds(k_ar_train<4, true>, lds));
  AR_HIP(set_lds(k_ar_find, lds)); AR_HIP(set_lds(k_ar_resolve, lds));
  n->dev_ready = true;
  return SF_OK;
}

size_t sf_nsfar_lds_bytes(const SfNsfAr& n, int hidden_buffers, int waves) {
  // inputs (padded to whole 16-row tiles), hidden buffers, 32 rows of one dimension's head per wave, V or GG + DV
  // (hidden_buffers == 3 names the TRAINING kernel: it runs on two hidden buffers too since round 5, plus the row tables and the
  //  sign bits of the first hidden layer)
  const int hb = hidden_buffers == 3 ? 2 : hidden_buffers;
  return (size_t)((n.D + n.C + 15) / 16 * 16 + hb * n.Hp + (hidden_buffers == 3 ? SF_AR_QBR : 32) * waves + 2 * n.D) * RS * sizeof(float) +
         (hidden_buffers == 3 ? (size_t)4 * n.Hp * sizeof(int) : 0);
}
// waves per 64 samples of the density / training kernels: four (tile pairs and dimensions dealt round robin) when the LDS takes it
static int ar_waves(const SfNsfAr& n, int hidden_buffers) {
  static int forced = -1;
  if (forced < 0) { const char* e = std::getenv("SF_NSFAR_WAVES"); forced = e ? std::atoi(e) : 0; }
  if (forced == 1 || forced == 4) return sf_nsfar_lds_bytes(n, hidden_buffers, forced) <= (size_t)160 * 1024 - 1024 ? forced : 1;
  return sf_nsfar_lds_bytes(n, hidden_buffers, 4) <= (size_t)160 * 1024 - 1024 ? 4 : 1;
}

int sf_nsfar_pack(SfNsfAr* n, const float* flat, hipStream_t st, std::string& err) {
  int rc = ar_ensure(n, err);
  if (rc) return rc;
  // (k_pack sums two gather tables; the second is "none" everywhere)
  AR_HIP(sf_launch_pack(flat, n->d_src, n->d_none, n->d_img, (long)n->src.size(), st));
  return SF_OK;
}

int sf_nsfar_log_prob(SfNsfAr* n, const float* theta, const float* x, long B, float* out, hipStream_t st, std::string& err) {
#ifdef SF_AR_TRACE
  {
    static unsigned long long* d_tr = nullptr;
    if (!d_tr) AR_HIP(hipMalloc(&d_tr, 256 * 8));
    AR_HIP(hipMemsetAsync(d_tr, 0, 256 * 8, st));
    ArArgs aa = args_of(*n);
    aa.trace = d_tr;
    if (ar_waves(*n, 2) == 4)
      hipLaunchKernelGGL(k_ar_logprob<4>, dim3((unsigned)((B + 63) / 64)), dim3(256), sf_nsfar_lds_bytes(*n, 2, 4), st, aa, theta, x, B, out);
    else
      hipLaunchKernelGGL(k_ar_logprob<1>, dim3((unsigned)((B + 63) / 64)), dim3(64), sf_nsfar_lds_bytes(*n, 2, 1), st, aa, theta, x, B, out);
    AR_HIP(hipStreamSynchronize(st));
    unsigned long long h[256];
    AR_HIP(hipMemcpy(h, d_tr, sizeof(h), hipMemcpyDeviceToHost));
    static int calls = 0;
    if (++calls % 8 == 0) {
      fprintf(stderr, "[nsfar trace] B=%ld (units of 100 cycles since stamp 0):", B);
      for (int i = 0; i < 256; ++i) if (h[i]) fprintf(stderr, " %d:%.1f", i, (double)(long long)(h[i] - h[0]) * 0.01);
      fprintf(stderr, "\n");
    }
    return SF_OK;
  }
#endif
  if (ar_waves(*n, 2) == 4)
    hipLaunchKernelGGL(k_ar_logprob<4>, dim3((unsigned)((B + 63) / 64)), dim3(256), sf_nsfar_lds_bytes(*n, 2, 4), st, args_of(*n), theta, x, B, out);
  else
    hipLaunchKernelGGL(k_ar_logprob<1>, dim3((unsigned)((B + 63) / 64)), dim3(64), sf_nsfar_lds_bytes(*n, 2, 1), st, args_of(*n), theta, x, B, out);
  AR_HIP(hipGetLastError());
  return SF_OK;
}

int sf_nsfar_inverse(SfNsfAr* n, const float* z, const float* x, long B, float* theta, float* logdet, hipStream_t st, std::string& err) {
  hipLaunchKernelGGL(k_ar_inverse, dim3((unsigned)((B + 63) / 64)), dim3(64), sf_nsfar_lds_bytes(*n, 2, 1), st, args_of(*n), z, x, B, theta, logdet);
  AR_HIP(hipGetLastError());
  return SF_OK;
}

int sf_nsfar_sample(SfNsfAr* n, const float* x, long M, long S, const uint32_t* slots, long n_slots, const float* lo, const float* hi,
                    uint32_t k0, uint32_t k1, unsigned long long slot_offset, int max_attempts, float* out, int32_t* n_drawn,
                    int32_t* count, int64_t* n_unfilled, hipStream_t st, std::string& err, hipEvent_t ev0, hipEvent_t ev1) {
  AR_HIP(hipMemsetAsync(n->d_ctr, 0, 8 * sizeof(unsigned long long), st));
  static int cus = 0;   // (asked once: the query costs more than a small sampling call)
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t pr;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
  }
  const size_t lds = sf_nsfar_lds_bytes(*n, 2, 1);
  const long per_cu = (long)((size_t)160 * 1024 / (lds + 1024));
  long grid = (long)cus * (per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu));
  if (grid > (n_slots + 63) / 64) grid = (n_slots + 63) / 64;
  // no ceiling asked for: the row-level progress rule of the kernel, and 2^20 attempts per slot at the very most
  const uint32_t cap = count ? 1u : (max_attempts > 0 ? (uint32_t)max_attempts : (1u << 20));
  int32_t* g_try = nullptr;
  if (!count && max_attempts <= 0 && lo) {
    if ((size_t)(2 * M) > n->gal_cap) {
      if (n->d_gal) AR_HIP(hipFree(n->d_gal));
      n->d_gal = nullptr; n->gal_cap = 0;
      AR_HIP(hipMalloc(&n->d_gal, (size_t)(2 * M) * sizeof(int32_t)));
      n->gal_cap = (size_t)(2 * M);
    }
    AR_HIP(hipMemsetAsync(n->d_gal, 0, (size_t)(2 * M) * sizeof(int32_t), st));
    g_try = n->d_gal;
  }
  if (ev0) AR_HIP(hipEventRecord(ev0, st));
  unsigned long long walk_R = 0, walk_C = 0;
  if (!count && n_slots >= 4096) {
    if (!slots && M > 1 && (long)(M * S) == n_slots) { walk_R = (unsigned long long)M; walk_C = (unsigned long long)S; }
    else { walk_R = 4096; walk_C = ((unsigned long long)n_slots + walk_R - 1) / walk_R; }
  }
  // The persistent launch takes every slot through attempts [0, 256); what is still open then (rows that accept less than
  // about one draw in a hundred) goes on in FIND / RESOLVE rounds that spread ONE slot's attempts over the whole chip -- as many new
  // attempts per round as the slot has already failed, within 4 M candidates per launch.  In one wave a slot of a row with
  // acceptance 7e-4 needed ~150 rounds of 64 attempts while the chip idled (the bench flow: 90 ms instead of 25).
  uint32_t window = cap;
  const bool rounds = !count && lo && cap > 256u && n_slots <= (1l << 31) && (unsigned long long)M * (unsigned long long)S < (1ull << 32);
  if (rounds) {
    // (16-sample waves: an open entry costs a whole 16-candidate round of ONE wave per 16 attempts once the list has run dry, so the
    //  hand-over comes early -- measured on the bench flow: window 256 / 64 / 32 / 16 -> 8.35 / 7.84 / 6.93 / 7.29 ms per catalogue)
    window = sf_nsfar16_eligible(*n) ? 32u : 256u;
    {   // (developer knob: attempts of a slot inside the persistent launch before the chip-wide rounds take it)
      static int w_env = -1;
      if (w_env < 0) { const char* e = std::getenv("SF_AR_WINDOW"); w_env = e ? std::atoi(e) : 0; }
      if (w_env >= 16 && (uint32_t)w_env < cap) window = (uint32_t)w_env / 16u * 16u;
    }
    if ((size_t)n_slots > n->surv_cap) {
      (void)hipFree(n->d_surv[0]); (void)hipFree(n->d_surv[1]); (void)hipFree(n->d_best);
      n->d_surv[0] = n->d_surv[1] = n->d_best = nullptr; n->surv_cap = 0;
      AR_HIP(hipMalloc(&n->d_surv[0], (size_t)n_slots * sizeof(uint32_t)));
      AR_HIP(hipMalloc(&n->d_surv[1], (size_t)n_slots * sizeof(uint32_t)));
      AR_HIP(hipMalloc(&n->d_best, (size_t)n_slots * sizeof(uint32_t)));
      n->surv_cap = (size_t)n_slots;
    }
  }
  unsigned int* d_ns = reinterpret_cast<unsigned int*>(n->d_ctr + 4);   // [0], [1]: survivor counts of the two lists
  const bool tiles16 = sf_nsfar16_eligible(*n);   // 16-sample register tiles, four independent waves per workgroup (sf_nsfar16.hip)
  const SfAr16Launch L = {x, S, slots, n_slots, lo, hi, k0, k1, slot_offset, cap, out, n_drawn, count, n->d_ctr,
                          reinterpret_cast<unsigned int*>(n->d_ctr + 1), g_try, g_try ? g_try + M : nullptr, walk_R, walk_C, window, n->d_surv[0], d_ns};
  if (tiles16) {
    AR_HIP(sf_nsfar16_launch(*n, L, cus, st));
  } else {
    hipLaunchKernelGGL(k_ar_sample, dim3((unsigned)grid), dim3(64), lds, st, args_of(*n), x, S, slots, n_slots, lo, hi, k0, k1, slot_offset, cap, out,
                       n_drawn, count, n->d_ctr, reinterpret_cast<unsigned int*>(n->d_ctr + 1), g_try, g_try ? g_try + M : nullptr, walk_R, walk_C,
                       window, n->d_surv[0], d_ns);
  }
  AR_HIP(hipGetLastError());
  if (ev1) AR_HIP(hipEventRecord(ev1, st));
  unsigned long long* h = n->h_ctr;   // (pinned: a read-back into pageable memory is a staged, host-synchronous copy)
  for (int i = 0; i < 8; ++i) h[i] = 0;
  if (rounds) {
    int cur = 0;
    uint32_t base = window;
    for (;;) {
      AR_HIP(hipMemcpyAsync(h, n->d_ctr, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
      AR_HIP(hipStreamSynchronize(st));
      const unsigned int ns = reinterpret_cast<const unsigned int*>(h + 4)[cur];
      {
        static int dbg = -1;
        if (dbg < 0) dbg = std::getenv("SF_AR_DEBUG") ? 1 : 0;
        if (dbg) {
          static double t_prev = 0.0;
          timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
          const double tn = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
          fprintf(stderr, "[nsfar rounds] base %u survivors %u evaluations %llu  (+%.3f ms since the previous read-back)\n", base, ns, h[2], tn - t_prev);
          t_prev = tn;
        }
      }
      if (ns == 0 || base >= cap) break;
      uint32_t A = 64;
      static int grow = -1;
      if (grow < 0) { const char* e = std::getenv("SF_AR_GROW"); grow = e ? std::atoi(e) : 1; if (grow < 1) grow = 1; }
      while (2u * A <= (uint32_t)grow * base && (uint64_t)(2u * A) * ns <= (1ull << 22) && 2u * A <= 65536u) A *= 2;
      if ((uint64_t)base + A > cap) A = (uint32_t)(((uint64_t)cap - base + 63u) / 64u * 64u);
      const uint32_t att_end = (uint64_t)base + A > cap ? cap : base + A;
      const uint32_t chunks = A / 64u;
      AR_HIP(hipMemsetAsync(n->d_best, 0xff, (size_t)ns * sizeof(uint32_t), st));
      AR_HIP(hipMemsetAsync(d_ns + (cur ^ 1), 0, sizeof(unsigned int), st));
      if (tiles16) {
        AR_HIP(sf_nsfar16_find(*n, L, n->d_surv[cur], ns, base, chunks, att_end, n->d_best, n->d_ctr, st));
        AR_HIP(sf_nsfar16_resolve(*n, L, n->d_surv[cur], ns, n->d_best, att_end, att_end - base, n->d_surv[cur ^ 1], d_ns + (cur ^ 1), st));
      } else {
        hipLaunchKernelGGL(k_ar_find, dim3(ns * chunks), dim3(64), lds, st, args_of(*n), x, S, n->d_surv[cur], ns, base, chunks, att_end, lo, hi, k0, k1,
                           slot_offset, n->d_best, n->d_ctr);
        hipLaunchKernelGGL(k_ar_resolve, dim3((ns + 63u) / 64u), dim3(64), lds, st, args_of(*n), x, S, n->d_surv[cur], ns, n->d_best, att_end, cap, lo, hi,
                           k0, k1, slot_offset, out, n_drawn, g_try, g_try ? g_try + M : nullptr, att_end - base, n->d_surv[cur ^ 1], d_ns + (cur ^ 1),
                           reinterpret_cast<unsigned int*>(n->d_ctr + 1));
      }
      AR_HIP(hipGetLastError());
      base = att_end;
      cur ^= 1;
    }
    const unsigned int left = reinterpret_cast<const unsigned int*>(h + 4)[cur];
    if (left > 0) {   // (the ceiling was reached with slots still open: NaN rows)
      AR_HIP(hipMemsetAsync(n->d_best, 0xff, (size_t)left * sizeof(uint32_t), st));
      if (tiles16) {
        SfAr16Launch Lf = L;
        Lf.g_try = nullptr; Lf.g_acc = nullptr;
        AR_HIP(sf_nsfar16_resolve(*n, Lf, n->d_surv[cur], left, n->d_best, cap, 0u, n->d_surv[cur ^ 1], d_ns + (cur ^ 1), st));
      } else {
        hipLaunchKernelGGL(k_ar_resolve, dim3((left + 63u) / 64u), dim3(64), lds, st, args_of(*n), x, S, n->d_surv[cur], left, n->d_best, cap, cap, lo, hi,
                           k0, k1, slot_offset, out, n_drawn, (int32_t*)nullptr, (int32_t*)nullptr, 0u, n->d_surv[cur ^ 1], d_ns + (cur ^ 1),
                           reinterpret_cast<unsigned int*>(n->d_ctr + 1));
      }
      AR_HIP(hipGetLastError());
    }
    AR_HIP(hipMemcpyAsync(h, n->d_ctr, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    AR_HIP(hipStreamSynchronize(st));
  } else if (n_unfilled) {
    AR_HIP(hipMemcpyAsync(h, n->d_ctr, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    AR_HIP(hipStreamSynchronize(st));
  }
  if (n_unfilled) *n_unfilled = (int64_t)(unsigned int)h[1];
  n->last_evals = (double)h[2];
  n->last_rej0 = (double)h[3];
  return SF_OK;
}

int sf_nsfar_loss_grad(SfNsfAr* n, const float* flat, const float* theta, const float* x, const long long* idx, long B, float grad_scale,
                       const float* weights, float* loss, double* loss_sum, float* grad, hipStream_t st, std::string& err, hipEvent_t ev0,
                       hipEvent_t ev1) {
  int rc = sf_nsfar_pack(n, flat, st, err);
  if (rc) return rc;
  if (B == 0) {
    AR_HIP(hipMemsetAsync(grad, 0, (size_t)n->n_params * sizeof(float), st));
    return SF_OK;
  }
  const size_t need = (size_t)B * n->T * n->D;
  if (need > n->ustash_cap) {
    if (n->d_ustash) AR_HIP(hipFree(n->d_ustash));
    n->d_ustash = nullptr; n->ustash_cap = 0;
    AR_HIP(hipMalloc(&n->d_ustash, need * sizeof(float)));
    n->ustash_cap = need;
  }
  // Gradient accumulation: one partial per 64-row chunk + k_ar_gather (plain stores, summed in chunk order) while the partials fit
  // 512 MiB -- cfg1 shape: up to 2 300 chunks = 147 000 rows; 16 384 rows: 0.57 -> 0.36 ms, 131 072 rows: 3.0 -> 1.6 ms -- else f32
  // atomics into the one gradient (SF_AR_GRAD=atomic forces them).  Measured and dropped: 2 x CUs persistent workgroups that own a
  // partial and ADD their later chunks -- with plain read-modify-writes 3.1 ms per 131 072 rows (an L2 round trip per weight-gradient
  // block), with uncontended atomics 2.9: the atomic units, not the contention, bound that form.
  const long n_chunks = (B + 63) / 64;
  static int force = -1;
  if (force < 0) { const char* e = std::getenv("SF_AR_GRAD"); force = !e ? 0 : (e[0] == 'a' ? 1 : (e[0] == 'p' ? 2 : 0)); }
  const size_t part_bytes = (size_t)n_chunks * (size_t)n->n_params * sizeof(float);
  const bool part = force != 1 && part_bytes <= ((size_t)512 << 20);
  const long nwg = n_chunks;
  if (part) {
    if (part_bytes > n->gpart_cap) {
      if (n->d_gpart) AR_HIP(hipFree(n->d_gpart));
      n->d_gpart = nullptr; n->gpart_cap = 0;
      AR_HIP(hipMalloc(&n->d_gpart, part_bytes));
      n->gpart_cap = part_bytes;
    }
  } else {
    AR_HIP(hipMemsetAsync(grad, 0, (size_t)n->n_params * sizeof(float), st));
  }
  float* gdst = part ? n->d_gpart : grad;
  const long gstride = part ? (long)n->n_params : 0;
  ArArgs aa = args_of(*n);
#ifdef SF_AR_TRACE
  static unsigned long long* d_tr = nullptr;
  if (!d_tr) AR_HIP(hipMalloc(&d_tr, 256 * 8));
  AR_HIP(hipMemsetAsync(d_tr, 0, 256 * 8, st));
  aa.trace = d_tr;
#endif
  if (ev0) AR_HIP(hipEventRecord(ev0, st));
  const int nwv = ar_waves(*n, 3);
  const dim3 grid((unsigned)nwg), block(64 * nwv);
  size_t lds = sf_nsfar_lds_bytes(*n, 3, nwv);
  {
    static long pad = -1;
    if (pad < 0) { const char* e = std::getenv("SF_AR_LDS_PAD"); pad = e ? std::atol(e) : 0; }
    lds += (size_t)pad;
  }
  {
    static int dbg = -1;
    if (dbg < 0) dbg = std::getenv("SF_AR_DEBUG") ? 1 : 0;
    if (dbg == 1) {
      dbg = 2;
      int occ = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_ar_train<4, true>, 256, lds);
      fprintf(stderr, "[nsfar train] %d waves, %zu bytes of LDS per workgroup, %d workgroups per CU by the occupancy query, partial mode %d\n", nwv, lds, occ, (int)part);
    }
  }
  if (nwv == 4) {
    if (part) hipLaunchKernelGGL((k_ar_train<4, true>), grid, block, lds, st, aa, theta, x, idx, B, grad_scale, weights, loss, loss_sum, gdst, gstride, n->d_ustash);
    else hipLaunchKernelGGL((k_ar_train<4, false>), grid, block, lds, st, aa, theta, x, idx, B, grad_scale, weights, loss, loss_sum, gdst, gstride, n->d_ustash);
  } else {
    if (part) hipLaunchKernelGGL((k_ar_train<1, true>), grid, block, lds, st, aa, theta, x, idx, B, grad_scale, weights, loss, loss_sum, gdst, gstride, n->d_ustash);
    else hipLaunchKernelGGL((k_ar_train<1, false>), grid, block, lds, st, aa, theta, x, idx, B, grad_scale, weights, loss, loss_sum, gdst, gstride, n->d_ustash);
  }
  AR_HIP(hipGetLastError());
  if (ev1) AR_HIP(hipEventRecord(ev1, st));
  if (part) {
    hipLaunchKernelGGL(k_ar_gather, dim3((unsigned)((n->n_params + 255) / 256)), dim3(256), 0, st, n->d_gpart, gstride, (int)nwg, n->d_live, grad, (long)n->n_params);
    AR_HIP(hipGetLastError());
  }
#ifdef SF_AR_TRACE
  {
    AR_HIP(hipStreamSynchronize(st));
    unsigned long long h[256];
    AR_HIP(hipMemcpy(h, d_tr, sizeof(h), hipMemcpyDeviceToHost));
    static int calls = 0;
    if (++calls % 8 == 0) {
      fprintf(stderr, "[nsfar train trace] B=%ld (units of 100 cycles since stamp 0):", B);
      for (int i = 0; i < 256; ++i) if (h[i]) fprintf(stderr, " %d:%.1f", i, (double)(long long)(h[i] - h[0]) * 0.01);
      fprintf(stderr, "\n");
    }
  }
#endif
  return SF_OK;
}
