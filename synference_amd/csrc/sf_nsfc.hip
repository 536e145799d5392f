// sf_nsfc.hip -- cooperative NSF training kernel: forward + backward of -log_prob on 16-row tiles
// (v_mfma_f32_16x16x4_f32), four waves per 32 samples.            ref: custom_runner.py:585-618 (the training step)
//
// Why it exists (round 4).  k_nsf_train (sf_train_kernels.h) gives a 32-sample tile to ONE producer wave: one wave per SIMD
// at batch 16 384, a ~1.5 M-cycle dependent chain per tile, 188 MB of activations stashed in HBM and read back, f32 atomics
// for every tile's 0.6 MB weight gradient (rocprof, round 3: 0.074 of the fp32 MFMA roof, 606 MB of HBM traffic per launch
// for 2.6 MB of algorithmic bytes).  Here, following k_maf_trainc (sf_trainc.hip):
//   * a workgroup (4 waves) owns 32 samples = two 16-sample subtiles; wave j owns hidden tile j (16 units) of BOTH subtiles,
//     so every weight fragment it fetches from L2 feeds two independent MFMA chains, and the waves exchange activation tiles
//     through LDS at every layer (one barrier per layer);
//   * nothing but u (16 floats per sample and transform) is stashed: the backward sweep of a transform first RECOMPUTES its
//     conditioner (h0, t1, t2, gate of both residual blocks: 56 VGPRs per wave), i.e. 4 instead of 3 forward-equivalents of
//     matrix work and no activation traffic at all;
//   * spline parameters never leave the lane that uses them: the head's output tiles are laid out so that lane (sample s,
//     row group g) receives the 3K - 1 parameters of transformed dimension g of sample s (sf_layout.h), and the spline
//     forward / backward is lane-local VALU work on the two "spline waves" of the workgroup;
//   * weight gradients: the delta tiles stay in the padded B layout they were exchanged in (read transposed: 68-float row
//     groups make that conflict free), the layer inputs are written transposed once ([sample >> 2][row][sample & 3]) --
//     products over the 32 samples, spread over the four waves; small batches store into the workgroup's own partial (plain
//     stores, summed by k_gather_c2), large ones add 2^-40 fixed-point contributions into the int64 gradient replica of
//     their XCD with integer atomics that never leave that XCD's L2 (sf_fixacc.h): bitwise reproducible either way;
//   * LULinear: forward / backward on replicated registers (every lane of a sample holds all D values); its weight
//     gradients ride the same 16 x 16 block products (two blocks per transform).
// Tile = 16 rows x 16 samples in 4 VGPRs: lane l = sample (l & 15) + 16 * row group (l >> 4), register r = row
// 4 * (l >> 4) + r.  Image layout: sf_layout.h (SfNscDev).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "sf_device.h"
#include "sf_fixacc.h"
#include "sf_internal.h"
#include "sf_nsfc.h"
#include "sf_spline_flat.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SF_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace {

constexpr int PBT = 272;  // floats of a padded B-layout tile: lane l's float4 at 4 l + 4 (l >> 4)
constexpr int TT = 272;   // floats of a transposed tile: four sample groups of 68 (64 + 4 of padding: conflict-free writes)

__device__ __forceinline__ f32x4 n_ld4(const float* p) {
  const float4 b = *reinterpret_cast<const float4*>(p);
  f32x4 r;
  r[0] = b.x; r[1] = b.y; r[2] = b.z; r[3] = b.w;
  return r;
}
__device__ __forceinline__ void n_st4(float* p, const f32x4 v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ f32x4 n_zero() {
  f32x4 z;
  z[0] = 0.f; z[1] = 0.f; z[2] = 0.f; z[3] = 0.f;
  return z;
}
__device__ __forceinline__ f32x4 n_relu(const f32x4 v) {
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = fmaxf(v[i], 0.f);
  return r;
}
// weight fragment of block (a, b) of a [.][nb] block array (wave-uniform base, one shared per-lane offset)
__device__ __forceinline__ float4 n_frag(const float* wp, int nb, int a, int b, int lane) {
  const float4* base = reinterpret_cast<const float4*>(wp) + (a * nb + b) * 64;
  return base[(unsigned)lane];
}
// one weight fragment through the tiles of both subtiles: two independent accumulator chains, their MFMAs alternating;
// components [k0, kc) of the fragment are in use (rows k, 4 + k, 8 + k, 12 + k of the input tile: sf_layout.h)
__device__ __forceinline__ void n_mma2(const float4 w, const f32x4 ia, const f32x4 ib, f32x4& aa, f32x4& ab, int kc, int k0 = 0) {
  if (k0 < 1) { aa = SF_MFMA16(w.x, ia[0], aa); ab = SF_MFMA16(w.x, ib[0], ab); }
  if (k0 < 2 && kc > 1) { aa = SF_MFMA16(w.y, ia[1], aa); ab = SF_MFMA16(w.y, ib[1], ab); }
  if (kc > 2) { aa = SF_MFMA16(w.z, ia[2], aa); ab = SF_MFMA16(w.z, ib[2], ab); }
  if (kc > 3) { aa = SF_MFMA16(w.w, ia[3], aa); ab = SF_MFMA16(w.w, ib[3], ab); }
}
// one tile: the k-steps alternate between two partial accumulators (the caller adds them)
__device__ __forceinline__ void n_mma1(const float4 w, const f32x4 in, f32x4& a0, f32x4& a1, int kc) {
  a0 = SF_MFMA16(w.x, in[0], a0);
  if (kc > 1) a1 = SF_MFMA16(w.y, in[1], a1);
  if (kc > 2) a0 = SF_MFMA16(w.z, in[2], a0);
  if (kc > 3) a1 = SF_MFMA16(w.w, in[3], a1);
}
// transposed tile: element (row, sample) at (sample >> 2) * 68 + row * 4 + (sample & 3); lane l = 16 kk + i then reads
// row i, samples 4 kk .. 4 kk + 3 with one ds_read_b128 at kk * 68 + i * 4.  (68, not 64: the four ds_write_b32 of a lane
// then hit 4 (s >> 2) + 16 g4 + (s & 3) + 4 r -- 64 different banks; with 64 the lanes s, s + 4, s + 8, s + 12 collided, 37 % of
// the kernel's LDS cycles were bank conflicts.)
__device__ __forceinline__ int n_T_rd(int lane) { return (lane >> 4) * 68 + (lane & 15) * 4; }
__device__ __forceinline__ void n_put_T(float* tile, const f32x4 v, int s, int g4) {
  float* p = tile + (s >> 2) * 68 + (s & 3) + 16 * g4;
  p[0] = v[0]; p[4] = v[1]; p[8] = v[2]; p[12] = v[3];
}
__device__ __forceinline__ float n_sel4(int g, float a, float b, float c, float d) { return g == 0 ? a : (g == 1 ? b : (g == 2 ? c : d)); }

// ---------------------------------------------------------------------------------------------------------------------
// Weight-gradient blocks: acc[ro][ri] = sum over the 32 samples of delta[ot rows][s] * in[it rows][s].  The delta tile is read
// TRANSPOSED out of its padded B layout (element (row i, sample sm) at 4 sm + 68 (i >> 2) + (i & 3): lane (i, kk) takes
// samples 4 kk .. 4 kk + 3 with four ds_read_b32 that hit 64 different banks), the input tile with one ds_read_b128.
// ---------------------------------------------------------------------------------------------------------------------
struct NJob {
  const float* Xd;  // padded B-layout delta tiles [.][2]
  const float* Ti;  // transposed input tiles [.][2]
  float* gw;        // gradient block (256 floats)
  float* gb;        // bias gradient rows of tile ot (or null)
  int ot, it;
};
__device__ __forceinline__ void n_dw_finish(const NJob& J, f32x4 acc, float bs, const SfAcc& A, int lane) {
  // A.mode 0: plain store, 1: add to the workgroup's partial (later chunks), 2: f32 atomics into the XCD's replica,
  // 3: fixed-point (int64) atomics into it
  if (A.mode == 2) {
    float* q = reinterpret_cast<float*>(A.fix) + (J.gw - A.base);
#pragma unroll
    for (int r = 0; r < 4; ++r) sf_l2_add(q + r * 64 + lane, acc[r]);
  } else if (A.mode == 3) {
    long long* q = A.fix + (J.gw - A.base);
#pragma unroll
    for (int r = 0; r < 4; ++r) sf_fix_add(q + r * 64 + lane, acc[r]);
  } else {
    if (A.mode == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] += J.gw[r * 64 + lane];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) J.gw[r * 64 + lane] = acc[r];
  }
  if (J.gb) {
    bs += __shfl_xor(bs, 16, 64);
    bs += __shfl_xor(bs, 32, 64);
    if (lane < 16) {
      if (A.mode == 2) sf_l2_add(reinterpret_cast<float*>(A.fix) + (J.gb - A.base) + J.ot * 16 + lane, bs);
      else if (A.mode == 3) sf_fix_add(A.fix + (J.gb - A.base) + J.ot * 16 + lane, bs);
      else J.gb[J.ot * 16 + lane] = A.mode == 1 ? J.gb[J.ot * 16 + lane] + bs : bs;
    }
  }
}
__device__ __forceinline__ void n_dw_jobs(const NJob& A, const NJob& B, bool two, const SfAcc& mode, int lane) {
  const int dofs = 16 * (lane >> 4) + 68 * ((lane & 15) >> 2) + (lane & 3);
  f32x4 a0 = n_zero(), a1 = n_zero(), b0 = n_zero(), b1 = n_zero();
  float bsA = 0.f, bsB = 0.f;
  if (two) {  // two blocks at once: four accumulator chains in flight
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float* dA = A.Xd + (A.ot * 2 + q) * PBT + dofs;
      const float* dB = B.Xd + (B.ot * 2 + q) * PBT + dofs;
      const float dA0 = dA[0], dA1 = dA[4], dA2 = dA[8], dA3 = dA[12];
      const float dB0 = dB[0], dB1 = dB[4], dB2 = dB[8], dB3 = dB[12];
      const float4 iA = *reinterpret_cast<const float4*>(A.Ti + (A.it * 2 + q) * TT + n_T_rd(lane));
      const float4 iB = *reinterpret_cast<const float4*>(B.Ti + (B.it * 2 + q) * TT + n_T_rd(lane));
      a0 = SF_MFMA16(dA0, iA.x, a0);
      b0 = SF_MFMA16(dB0, iB.x, b0);
      a1 = SF_MFMA16(dA1, iA.y, a1);
      b1 = SF_MFMA16(dB1, iB.y, b1);
      a0 = SF_MFMA16(dA2, iA.z, a0);
      b0 = SF_MFMA16(dB2, iB.z, b0);
      a1 = SF_MFMA16(dA3, iA.w, a1);
      b1 = SF_MFMA16(dB3, iB.w, b1);
      bsA += (dA0 + dA1) + (dA2 + dA3);
      bsB += (dB0 + dB1) + (dB2 + dB3);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { a0[r] += a1[r]; b0[r] += b1[r]; }
    n_dw_finish(A, a0, bsA, mode, lane);
    n_dw_finish(B, b0, bsB, mode, lane);
  } else {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float* dA = A.Xd + (A.ot * 2 + q) * PBT + dofs;
      const float dA0 = dA[0], dA1 = dA[4], dA2 = dA[8], dA3 = dA[12];
      const float4 iA = *reinterpret_cast<const float4*>(A.Ti + (A.it * 2 + q) * TT + n_T_rd(lane));
      a0 = SF_MFMA16(dA0, iA.x, a0);
      a1 = SF_MFMA16(dA1, iA.y, a1);
      a0 = SF_MFMA16(dA2, iA.z, a0);
      a1 = SF_MFMA16(dA3, iA.w, a1);
      bsA += (dA0 + dA1) + (dA2 + dA3);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) a0[r] += a1[r];
    n_dw_finish(A, a0, bsA, mode, lane);
  }
}

}  // namespace

// argument block through the kernarg segment pointer, laundered where a phase starts (see k_maf_trainc)
__device__ __forceinline__ const SfNscArgs& n_args() {
  auto kp = __builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));
  return *(const SfNscArgs*)kp;
}
// workgroup barrier for LDS hand-overs: wait for the wave's own LDS operations only (outstanding global loads -- weight
// fragments requested ahead -- stay in flight), then s_barrier
__device__ __forceinline__ void n_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  SF_FUZZ();
}

// NW waves: 4, or 5 when the flow has five hidden tiles (64 < H <= 80: the reference's production NSF has H = 69); one hidden
// tile per wave.
template <int NT, int OTQ>
__global__ __launch_bounds__(64 * (NT > 4 ? NT : 4), (NT > 4 ? 1 : 2)) void k_nsf_trainc(SfNscArgs a_in) {
  extern __shared__ float lds[];
  constexpr int NQ = 2;
  constexpr int NW = NT > 4 ? NT : 4;
  constexpr int KM = OTQ == 6 ? 8 : 11, NQV = 4 * OTQ;
  constexpr int NOWN = OTQ / NW;       // spline-head tiles a wave produces for both subtiles (tile = wave + NW k)
  // the remaining OTQ % NW tiles: four waves split them into (tile, subtile) units, one per wave (OTQ = 6: two tiles = four
  // units); five waves give whole tiles to the first waves
  constexpr int NEXT = NW == 4 ? (OTQ % 4) * 2 : OTQ % NW;
  using Spl = NSpl<KM, NQV>;
  const SfNscArgs& a = n_args();
  const SfNscDev& c = a.c;
  // wave = the wave's ROLE (tile ownership, spline wave of subtile `wave` for roles 0 and 1).  The second half of the grid
  // rotates the roles by two: the two workgroups that share a CU (w and w + grid / 2: the dispatcher fills every CU once
  // before it starts a second round) then keep their spline waves -- the waves with the extra VALU phases -- on different
  // SIMD pairs
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) ^ ((NW == 4 && 2 * blockIdx.x >= gridDim.x && gridDim.x > 1) ? 2 : 0));
  const int s = lane & 15, g4 = lane >> 4;
  const int NI = c.NI, D = a.D, T = a.T;
  const bool has = wave < NT;   // owns hidden tile `wave` of both subtiles
  const int j = has ? wave : 0;
  const bool spl = wave < NQ;   // spline wave of subtile `wave`
  const int pbo = lane * 4 + 4 * g4;
  const int kc_h = c.kc_h;
  // ---- LDS map (floats)
  float* XIN = lds;                      // padded B: input tiles [NI][NQ]
  float* XA = XIN + NI * NQ * PBT;       // padded B: [NT][NQ]
  float* XB = XA + NT * NQ * PBT;
  float* XG = XB + NT * NQ * PBT;        // gate deltas
  float* XQ = XG + NT * NQ * PBT;        // spline-head outputs / their deltas [OTQ][NQ]
  float* XL = XQ + OTQ * NQ * PBT;       // LU deltas [2][NQ]
  float* TIN = XL + 2 * NQ * PBT;        // transposed: input tiles [NI][NQ]
  float* TI = TIN + NI * NQ * TT;        // transposed layer inputs [NT][NQ], two buffers
  float* TI2 = TI + NT * NQ * TT;
  float* TL = TI2 + NT * NQ * TT;        // transposed LU inputs [2][NQ]
  float* LUC = TL + 2 * NQ * TT;         // LU block of the current transform (144 floats)
  // per-sample state of the spline waves between their phases, [subtile][16 samples][24]: u (forward) / dL/du (backward) [0, 8),
  // u entering the transform [8, 16), the spline's outputs u' [16, 24).  (Held in registers, replicated over the four row
  // groups of a sample, they were 24-32 live VGPRs through every matrix phase of every wave.)
  float* SST = LUC + 160;
  const NSplC sc = {a.K, a.tail_bound, a.min_w, a.min_h, a.min_d, a.inv_sqrt_h, a.deriv_const};
  // gradient target: this workgroup's partial, or -- a.fix -- the fixed-point replica of its XCD (sf_fixacc.h; the job
  // descriptors then carry offsets from a.gpart that are never dereferenced as floats)
  float* gpart = a.fix ? a.gpart : a.gpart + (size_t)blockIdx.x * a.gpart_stride;
  // (a.fix_mode 2: the replicas are float images of the same stride)
  long long* gfix = !a.fix ? nullptr
                           : (a.fix_mode == 2 ? reinterpret_cast<long long*>(reinterpret_cast<float*>(a.fix) + (size_t)sf_xcc_id() * a.gpart_stride)
                                              : a.fix + (size_t)sf_xcc_id() * a.gpart_stride);
  auto kc_in = [&](int it) { return it == 0 ? 4 : (it == 1 ? c.kc_in[1] : c.kc_in[2]); };

  for (long chunk = blockIdx.x, iter = 0; chunk < a.n_chunks; chunk += gridDim.x, ++iter) {
    const SfAcc mode = {gfix ? a.fix_mode : (iter > 0 ? 1 : 0), a.gpart, gfix};
    SF_NC(0);
    // ------------------------------------------------------------------ per-sample inputs (spline waves)
    const long row = chunk * 32 + (spl ? wave : 0) * 16 + s;
    const bool valid = spl && row < a.B;
    float wgt = 0.f;
    float ld = 0.f, ldlu = 0.f;
    float* sst = SST + ((spl ? wave : 0) * 16 + s) * 24;   // my sample's state
    auto st8 = [&](int off, const float (&v)[8]) {   // (one lane of the four that hold identical copies writes)
      if (g4 == 0) {
        *reinterpret_cast<float4*>(sst + off) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(sst + off + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
    };
    auto ld8 = [&](int off, float (&v)[8]) {
      const float4 a0 = *reinterpret_cast<const float4*>(sst + off), a1 = *reinterpret_cast<const float4*>(sst + off + 4);
      v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
    };
    if (spl) {
      const long ii = row < a.B ? row : a.B - 1;
      const long src = a.idx ? (long)a.idx[ii] : ii;
      wgt = valid ? (a.wts ? a.w * a.wts[row] : a.w) : 0.f;
      const float* xr = a.x + src * a.C;
      const float* th = a.theta + src * D;
      float u[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        u[p] = 0.f;
        if (p < D) u[p] = th[p] * a.cst[a.c_pscale + p] + a.cst[a.c_pshift + p];
      }
      st8(0, u);
      for (int it = 0; it < NI; ++it) {
        f32x4 e = n_zero();
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int f = it == 0 ? (m - 2) * 4 + g4 : 8 + (it - 1) * 16 + 4 * m + g4;
          const bool on = (it > 0 || m >= 2) && f < a.C;
          const int ff = on ? f : 0;
          const float v = (xr[ff] - a.cst[a.c_xmean + ff]) / a.cst[a.c_xstd + ff];
          e[m] = on ? v : 0.f;
        }
        if (it == 0) {
          e[0] = n_sel4(g4, u[0], u[2], u[4], u[6]);
          e[1] = n_sel4(g4, u[1], u[3], u[5], u[7]);
        }
        n_st4(XIN + (it * NQ + wave) * PBT + pbo, e);
        n_put_T(TIN + (it * NQ + wave) * TT, e, s, g4);
      }
    }
    n_barrier();

    // The fragments of the HIDDEN layers (NT per phase) and their bias are requested ONE PHASE AHEAD, into the register set the
    // running phase does not use (wA / bA and wB / bB alternate): their L2 round trip hides behind the running phase's
    // products and the barrier.  Input-layer / gate fragments are requested at the top of their own phase (they are used after
    // the hidden product).  (Everything one phase ahead: 80 live registers, 200 spilled.)
    f32x4 h0[2], t1[2], t2[2], sg[2], t1b[2], t2b[2], sgb[2];
    auto bias4 = [&](const float* tp, int off, int tile) { return n_ld4(tp + off + tile * 16 + 4 * g4); };
    auto ld_hid = [&](const float* tp, int o_w, int o_b, float4 (&W)[NT], f32x4& b) {   // (unconditional: a wave without a tile
#pragma unroll                                                                          //  loads tile 0's and drops them)
      for (int it = 0; it < NT; ++it) W[it] = n_frag(tp + o_w, NT, j, it, lane);
      b = bias4(tp, o_b, j);
    };
    // acc[q] = b + W[j][:] . X[:][q] over the NT hidden input tiles
    auto go_hid = [&](const float4 (&W)[NT], const f32x4 b, const float* X, f32x4& acc0, f32x4& acc1) {
      acc0 = b;
      acc1 = b;
#pragma unroll
      for (int it = 0; it < NT; ++it) {
        const f32x4 i0 = n_ld4(X + (it * NQ + 0) * PBT + pbo), i1 = n_ld4(X + (it * NQ + 1) * PBT + pbo);
        n_mma2(W[it], i0, i1, acc0, acc1, it == NT - 1 ? kc_h : 4);
      }
    };
    // the same over the input tiles; k0: first component in use (2 for the gates: context rows only)
    auto ld_in = [&](const float* tp, int o_w, int o_b, float4 (&W)[3], f32x4& b) {
#pragma unroll
      for (int it = 0; it < 3; ++it)
        if (it < NI) W[it] = n_frag(tp + o_w, NI, j, it, lane);
      b = bias4(tp, o_b, j);
    };
    auto go_in = [&](const float4 (&W)[3], const f32x4 b, int k0, f32x4& acc0, f32x4& acc1) {
      acc0 = b;
      acc1 = b;
#pragma unroll
      for (int it = 0; it < 3; ++it)
        if (it < NI) {
          const f32x4 i0 = n_ld4(XIN + (it * NQ + 0) * PBT + pbo), i1 = n_ld4(XIN + (it * NQ + 1) * PBT + pbo);
          n_mma2(W[it], i0, i1, acc0, acc1, kc_in(it), it == 0 ? k0 : 0);
        }
    };
    auto put2 = [&](float* X, const f32x4 v0, const f32x4 v1) {
      n_st4(X + (j * NQ + 0) * PBT + pbo, v0);
      n_st4(X + (j * NQ + 1) * PBT + pbo, v1);
    };
    auto putT2 = [&](float* Tb, const f32x4 v0, const f32x4 v1) {
      n_put_T(Tb + (j * NQ + 0) * TT, v0, s, g4);
      n_put_T(Tb + (j * NQ + 1) * TT, v1, s, g4);
    };
    auto img_of = [&](int t) { const SfNscArgs& a = n_args(); return a.img + (size_t)t * a.c.t_stride + sf_opaque_zero(); };

    // F1 .. F6 of transform t: conditioner forward; leaves the head outputs in XQ.  bw: recomputation at the start of the
    // transform's backward sweep (also writes h2 transposed for the head's weight gradient).
    auto fwd_mat = [&](int t, bool bw) {
      const SfNscArgs& a = n_args();
      const SfNscDev& c = a.c;
      const float* tp = img_of(t);
      float4 wA[NT], wB[NT];
      f32x4 bA, bB;
      // LU block of this transform -> LDS (read by the spline waves many barriers later)
      if (threadIdx.x < 144) LUC[threadIdx.x] = tp[c.o_lu + threadIdx.x];
      // F1: h0 = bin + Win . [u ; e(x)]
      {
        float4 wi[3];
        f32x4 bi;
        ld_in(tp, c.o_win, c.o_bin, wi, bi);
        ld_hid(tp, c.o_w1[0], c.o_b1[0], wB, bB);
        if (has) {
          go_in(wi, bi, 0, h0[0], h0[1]);
          put2(XA, n_relu(h0[0]), n_relu(h0[1]));
        }
      }
      SF_NC(300 + (bw ? 10 : 0) + 0);
      n_barrier();
      SF_NC(300 + (bw ? 10 : 0) + 1);
      // F2: t1 = b1 + W1 relu(h0)
      ld_hid(tp, c.o_w2[0], c.o_b2[0], wA, bA);
      if (has) {
        go_hid(wB, bB, XA, t1[0], t1[1]);
        put2(XB, n_relu(t1[0]), n_relu(t1[1]));
      }
      n_barrier();
      SF_NC(300 + (bw ? 10 : 0) + 2);
      // F3: t2 = b2 + W2 relu(t1); gate = sigmoid(bg + Wg e); h1 = h0 + t2 * gate
      ld_hid(tp, c.o_w1[1], c.o_b1[1], wB, bB);
      if (has) {
        float4 wi[3];
        f32x4 bi;
        ld_in(tp, c.o_wg[0], c.o_bg[0], wi, bi);
        go_hid(wA, bA, XB, t2[0], t2[1]);
        f32x4 ga0, ga1;
        go_in(wi, bi, 2, ga0, ga1);
        f32x4 h1a, h1b;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sg[0][r] = sf_sigmoid(ga0[r]); sg[1][r] = sf_sigmoid(ga1[r]);
          h1a[r] = fmaxf(h0[0][r] + t2[0][r] * sg[0][r], 0.f);
          h1b[r] = fmaxf(h0[1][r] + t2[1][r] * sg[1][r], 0.f);
        }
        put2(XA, h1a, h1b);
      }
      n_barrier();
      SF_NC(300 + (bw ? 10 : 0) + 3);
      // F4: t1' = b1' + W1' relu(h1)
      ld_hid(tp, c.o_w2[1], c.o_b2[1], wA, bA);
      if (has) {
        go_hid(wB, bB, XA, t1b[0], t1b[1]);
        put2(XB, n_relu(t1b[0]), n_relu(t1b[1]));
      }
      n_barrier();
      SF_NC(300 + (bw ? 10 : 0) + 4);
      // F5: t2' = b2' + W2' relu(t1'); gate'; h2 = h1 + t2' * gate'  (no activation in front of the head).  Ahead: the head
      // fragments of tile `wave` (both subtiles)
#pragma unroll
      for (int it = 0; it < NT; ++it) wB[it] = n_frag(tp + c.o_wout, NT, wave, it, lane);
      bB = bias4(tp, c.o_bout, wave);
      if (has) {
        float4 wi[3];
        f32x4 bi;
        ld_in(tp, c.o_wg[1], c.o_bg[1], wi, bi);
        go_hid(wA, bA, XB, t2b[0], t2b[1]);
        f32x4 ga0, ga1;
        go_in(wi, bi, 2, ga0, ga1);
        f32x4 h2a, h2b;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sgb[0][r] = sf_sigmoid(ga0[r]); sgb[1][r] = sf_sigmoid(ga1[r]);
          h2a[r] = h0[0][r] + t2[0][r] * sg[0][r] + t2b[0][r] * sgb[0][r];
          h2b[r] = h0[1][r] + t2[1][r] * sg[1][r] + t2b[1][r] * sgb[1][r];
        }
        put2(XA, h2a, h2b);
        if (bw) putT2(TI2, h2a, h2b);
      }
      n_barrier();
      SF_NC(300 + (bw ? 10 : 0) + 5);
      // F6: q = bout + Wout h2: tiles wave, wave + 4 for both subtiles; the remaining tiles one (tile, subtile) unit per wave
      {
        // second portion of the head's fragments: tile wave + 4 (OTQ = 8) or this wave's extra unit (OTQ = 6)
        constexpr bool OWN2 = NOWN > 1 || NW > 4;   // the second portion is a whole tile (both subtiles)
        const int tile2 = NOWN > 1 ? wave + NW : (NW > 4 ? NW * NOWN + wave : 4 * NOWN + (wave >> 1));
        const bool do2 = NOWN > 1 || (NEXT > 0 && wave < NEXT);
        float4 w2[NT];
        f32x4 b2 = n_zero();
        if (do2) {
#pragma unroll
          for (int it = 0; it < NT; ++it) w2[it] = n_frag(tp + c.o_wout, NT, tile2, it, lane);
          b2 = bias4(tp, c.o_bout, tile2);
        }
        f32x4 acc0 = bB, acc1 = bB;
#pragma unroll
        for (int it = 0; it < NT; ++it) {
          const f32x4 i0 = n_ld4(XA + (it * NQ + 0) * PBT + pbo), i1 = n_ld4(XA + (it * NQ + 1) * PBT + pbo);
          n_mma2(wB[it], i0, i1, acc0, acc1, it == NT - 1 ? kc_h : 4);
        }
        n_st4(XQ + (wave * NQ + 0) * PBT + pbo, acc0);
        n_st4(XQ + (wave * NQ + 1) * PBT + pbo, acc1);
        if (do2) {
          if (OWN2) {
            acc0 = b2;
            acc1 = b2;
#pragma unroll
            for (int it = 0; it < NT; ++it) {
              const f32x4 i0 = n_ld4(XA + (it * NQ + 0) * PBT + pbo), i1 = n_ld4(XA + (it * NQ + 1) * PBT + pbo);
              n_mma2(w2[it], i0, i1, acc0, acc1, it == NT - 1 ? kc_h : 4);
            }
            n_st4(XQ + (tile2 * NQ + 0) * PBT + pbo, acc0);
            n_st4(XQ + (tile2 * NQ + 1) * PBT + pbo, acc1);
          } else {
            const int sub = wave & 1;
            acc0 = b2;
            acc1 = n_zero();
#pragma unroll
            for (int it = 0; it < NT; ++it) n_mma1(w2[it], n_ld4(XA + (it * NQ + sub) * PBT + pbo), acc0, acc1, it == NT - 1 ? kc_h : 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc0[r] += acc1[r];
            n_st4(XQ + (tile2 * NQ + sub) * PBT + pbo, acc0);
          }
        }
      }
      SF_NC(300 + (bw ? 10 : 0) + 6);
      n_barrier();
      SF_NC(300 + (bw ? 10 : 0) + 7);
    };
    // LU constants of the transform whose block sits in LUC
    auto lu_diag = [&](float (&dg)[8]) {
#pragma unroll
      for (int i = 0; i < 8; ++i) dg[i] = i < D ? sf_softplus(LUC[128 + i]) + a.lu_eps : 1.f;
    };
    auto write_in0 = [&](const float (&uu)[8]) {   // input tile 0 of my subtile: theta rows from uu, context rows as they are
      f32x4 e = n_ld4(XIN + (0 * NQ + wave) * PBT + pbo);
      e[0] = n_sel4(g4, uu[0], uu[2], uu[4], uu[6]);
      e[1] = n_sel4(g4, uu[1], uu[3], uu[5], uu[7]);
      n_st4(XIN + (0 * NQ + wave) * PBT + pbo, e);
      n_put_T(TIN + (0 * NQ + wave) * TT, e, s, g4);
    };
    float* ust = a.ustash + ((size_t)(chunk * 32 + (spl ? wave : 0) * 16 + s) * T) * 16;
    // F7 (spline waves): spline on my (sample, transformed dimension g4), then LULinear on replicated registers.
    // keep: the transform's backward sweep follows at once (the top transform): nothing is stashed, the input tile stays
    auto fwd_spline = [&](int t, bool keep, float (&u)[8], float (&ui)[8], float (&up)[8]) {
      const int start = t & 1, d_tr = (D - start + 1) >> 1;
      const bool have = g4 < d_tr;
      float q[NQV];
#pragma unroll
      for (int jt = 0; jt < OTQ; ++jt) {
        const f32x4 v = n_ld4(XQ + (jt * NQ + wave) * PBT + pbo);
        q[4 * jt] = v[0]; q[4 * jt + 1] = v[1]; q[4 * jt + 2] = v[2]; q[4 * jt + 3] = v[3];
      }
      ld8(0, u);
      SF_NC(320 + (keep ? 10 : 0) + 0);
#pragma unroll
      for (int p = 0; p < 8; ++p) ui[p] = u[p];
      const float vin = start ? n_sel4(g4, u[1], u[3], u[5], u[7]) : n_sel4(g4, u[0], u[2], u[4], u[6]);
      float vout, lad;
      Spl::fwd(sc, q, vin, vout, lad);
      SF_NC(320 + (keep ? 10 : 0) + 1);
      ld += have ? lad : 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float vg = __shfl(vout, s + 16 * g, 64);
        const bool on = g < d_tr;
        u[2 * g] = (on && start == 0) ? vg : u[2 * g];
        u[2 * g + 1] = (on && start == 1) ? vg : u[2 * g + 1];
      }
#pragma unroll
      for (int p = 0; p < 8; ++p) up[p] = u[p];
      if (!keep && g4 == 0) {
        float4* dst = reinterpret_cast<float4*>(ust + t * 16);
        dst[0] = make_float4(ui[0], ui[1], ui[2], ui[3]); dst[1] = make_float4(ui[4], ui[5], ui[6], ui[7]);
        dst[2] = make_float4(up[0], up[1], up[2], up[3]); dst[3] = make_float4(up[4], up[5], up[6], up[7]);
      }
      // y = L (U u') + b
      SF_NC(320 + (keep ? 10 : 0) + 2);
      float dg[8], tt[8];
      lu_diag(dg);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        tt[i] = dg[i] * up[i];
        if (i < D) ldlu += sf_log(dg[i]);
#pragma unroll
        for (int jj = i + 1; jj < 8; ++jj) tt[i] += LUC[64 + i * 8 + jj] * up[jj];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float y = tt[i] + LUC[136 + i];
#pragma unroll
        for (int jj = 0; jj < i; ++jj) y += LUC[i * 8 + jj] * tt[jj];
        u[i] = i < D ? y : 0.f;
      }
      SF_NC(320 + (keep ? 10 : 0) + 3);
      if (!keep) {
        write_in0(u);
        st8(0, u);
      }
      SF_NC(320 + (keep ? 10 : 0) + 4);
    };

    // ------------------------------------------------------------------ forward sweep (all transforms but the top one)
    for (int t = 0; t < T - 1; ++t) {
      SF_NC(1 + t);
      fwd_mat(t, false);
      if (spl) {
        float u[8], ui[8], up[8];
        fwd_spline(t, false, u, ui, up);
      }
      n_barrier();
    }

    // ------------------------------------------------------------------ backward sweep (the top transform: first evaluation)
    for (int t = T - 1; t >= 0; --t) {
      const SfNscArgs& a = n_args();
      const SfNscDev& c = a.c;
      const float* tp = img_of(t);
      float* gp = gpart + (size_t)t * c.g_stride;
      const bool top = t == T - 1;
      SF_NC(40 + 10 * (T - 1 - t));
      if (!top) {
        if (spl) {
          float ui[8];
          ld8(8, ui);
          write_in0(ui);
        }
        n_barrier();
      }
      fwd_mat(t, true);
      SF_NC(41 + 10 * (T - 1 - t));
      if (spl && top) {
        // ---------------------------------------------------------------- loss, dL/du_T
        float u[8], ui[8], up[8], G[8];
        fwd_spline(t, true, u, ui, up);
        float ss = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p) ss += u[p] * u[p];
        float lds_ = ld;
        lds_ += __shfl_xor(lds_, 16, 64);
        lds_ += __shfl_xor(lds_, 32, 64);
        const float nll = 0.5f * ss + 0.5f * (float)D * 1.8378770664093453f - (a.logdet0 + ldlu + lds_);
        if (a.loss && valid && g4 == 0) a.loss[row] = nll;
        if (a.loss_sum) {
          float tsum = (valid && g4 == 0) ? nll : 0.f;
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) tsum += __shfl_xor(tsum, o, 64);
          // values on a 2^-20 grid add exactly in double: the sum does not depend on the order of the atomics
          if (lane == 0) atomicAdd(a.loss_sum + (blockIdx.x & a.loss_mask), (double)rintf(tsum * 1048576.0f) * (1.0 / 1048576.0));
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) G[p] = wgt * u[p];
        st8(0, G);
        st8(8, ui);
        st8(16, up);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the four lanes of a sample read what lane g4 == 0 wrote)
      }
      // B0 (spline waves): LULinear backward, spline backward; deltas of the head into XQ
      if (spl) {
        const int start = t & 1, d_tr = (D - start + 1) >> 1;
        const bool have = g4 < d_tr;
        float G[8], ui[8], up[8];
        ld8(0, G);
        ld8(8, ui);
        ld8(16, up);
        float dg[8], tt[8], dt[8], Gn[8], z[8];
        lu_diag(dg);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          tt[i] = dg[i] * up[i];
#pragma unroll
          for (int jj = i + 1; jj < 8; ++jj) tt[i] += LUC[64 + i * 8 + jj] * up[jj];
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          dt[jj] = G[jj];
#pragma unroll
          for (int i = jj + 1; i < 8; ++i) dt[jj] += LUC[i * 8 + jj] * G[i];
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          Gn[jj] = dg[jj] * dt[jj];
#pragma unroll
          for (int i = 0; i < jj; ++i) Gn[jj] += LUC[64 + i * 8 + jj] * dt[i];
          z[jj] = jj < D ? (dt[jj] * up[jj] - sf_div(wgt, dg[jj])) * sf_sigmoid(LUC[128 + jj]) : 0.f;
        }
        {
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = n_sel4(g4, G[r], G[4 + r], z[r], z[4 + r]);
          n_st4(XL + (0 * NQ + wave) * PBT + pbo, v);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = g4 == 0 ? dt[r] : (g4 == 1 ? dt[4 + r] : 0.f);
          n_st4(XL + (1 * NQ + wave) * PBT + pbo, v);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = g4 == 0 ? tt[r] : (g4 == 1 ? tt[4 + r] : 0.f);
          n_put_T(TL + (0 * NQ + wave) * TT, v, s, g4);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = g4 == 0 ? up[r] : (g4 == 1 ? up[4 + r] : 0.f);
          n_put_T(TL + (1 * NQ + wave) * TT, v, s, g4);
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) G[p] = p < D ? Gn[p] : 0.f;
        // (the spline's 2 x (3K - 1) parameter registers are not wanted while the LU arrays above are alive)
        __builtin_amdgcn_sched_barrier(0);
        SF_NC(340);
        float q[NQV], dq[NQV];
#pragma unroll
        for (int jt = 0; jt < OTQ; ++jt) {
          const f32x4 v = n_ld4(XQ + (jt * NQ + wave) * PBT + pbo);
          q[4 * jt] = v[0]; q[4 * jt + 1] = v[1]; q[4 * jt + 2] = v[2]; q[4 * jt + 3] = v[3];
        }
        const float vin = start ? n_sel4(g4, ui[1], ui[3], ui[5], ui[7]) : n_sel4(g4, ui[0], ui[2], ui[4], ui[6]);
        const float Gsel = start ? n_sel4(g4, G[1], G[3], G[5], G[7]) : n_sel4(g4, G[0], G[2], G[4], G[6]);
        float dv;
        Spl::bwd(sc, q, vin, have ? Gsel : 0.f, have ? -wgt : 0.f, dv, dq);
        SF_NC(341);
#pragma unroll
        for (int jt = 0; jt < OTQ; ++jt) {
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = have ? dq[4 * jt + r] : 0.f;
          n_st4(XQ + (jt * NQ + wave) * PBT + pbo, v);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float vg = __shfl(dv, s + 16 * g, 64);
          const bool on = g < d_tr;
          G[2 * g] = (on && start == 0) ? vg : G[2 * g];
          G[2 * g + 1] = (on && start == 1) ? vg : G[2 * g + 1];
        }
        st8(0, G);
        SF_NC(342);
      }
      n_barrier();
      SF_NC(42 + 10 * (T - 1 - t));
      // generic job runner: blocks n = wave, wave + NW, ... of a list, two at a time
      // (atomic modes: the i-th workgroup of an XCD starts the list at block i -- the workgroups of an XCD run the same phase at about the same
      //  time, and in list order all of them would add to the SAME gradient block, i.e. the same few L2 channels, at once)
      auto run_jobs = [&](int total, auto mk) {
        const int rot = mode.mode >= 2 ? (int)((blockIdx.x >> 3) % (unsigned)total) : 0;   // (blockIdx & 7 = the XCD: its workgroups share a replica)
        auto at = [&](int n) { const int m = n + rot; return m >= total ? m - total : m; };
        for (int n = wave; n < total; n += 2 * NW) {
          const bool two = n + NW < total;
          const NJob A = mk(at(n)), B = mk(at(two ? n + NW : n));
          n_dw_jobs(A, B, two, mode, lane);
        }
      };
      // Phases B1 .. B6: the phase's transposed fragments are requested first, then the weight-gradient blocks of the layer
      // above run (LDS operands only: they cover the fragments' L2 round trip, and their stores / atomics are issued AFTER
      // the loads, so that waiting for the fragments never waits for a store acknowledgement), then the data product.
      auto ld_hidT = [&](int o_wT, float4 (&W)[NT]) {
        if (has) {
#pragma unroll
          for (int ot = 0; ot < NT; ++ot) W[ot] = n_frag(tp + o_wT, NT, j, ot, lane);
        }
      };
      auto go_hidT = [&](const float4 (&W)[NT], const float* X, f32x4& acc0, f32x4& acc1) {
        acc0 = n_zero();
        acc1 = n_zero();
#pragma unroll
        for (int ot = 0; ot < NT; ++ot) {
          const f32x4 i0 = n_ld4(X + (ot * NQ + 0) * PBT + pbo), i1 = n_ld4(X + (ot * NQ + 1) * PBT + pbo);
          n_mma2(W[ot], i0, i1, acc0, acc1, ot == NT - 1 ? kc_h : 4);
        }
      };
      f32x4 dh[2];
      // B1: dh2 = Wout^T dq; deltas of block 1's second layer and gate; weight gradients of the head and of LULinear
      {
        float4 wf[OTQ];
        if (has) {
#pragma unroll
          for (int ot = 0; ot < OTQ; ++ot) wf[ot] = n_frag(tp + c.o_woutT, OTQ, j, ot, lane);
        }
        const int nw = OTQ * NT;
        run_jobs(nw + 2, [&](int n) -> NJob {
          if (n < nw) {
            const int ot = n / NT, it = n - ot * NT;
            return NJob{XQ, TI2, gp + c.g_wout + n * 256, it == 0 ? gp + c.g_bout : nullptr, ot, it};
          }
          const int b = n - nw;
          return NJob{XL, TL, gp + c.g_lu + b * 256, b == 0 ? gp + c.g_lu + 512 : nullptr, b, b};
        });
        SF_NC(350);
        if (has) {
          dh[0] = n_zero();
          dh[1] = n_zero();
#pragma unroll
          for (int ot = 0; ot < OTQ; ++ot) {
            const f32x4 i0 = n_ld4(XQ + (ot * NQ + 0) * PBT + pbo), i1 = n_ld4(XQ + (ot * NQ + 1) * PBT + pbo);
            n_mma2(wf[ot], i0, i1, dh[0], dh[1], 4);
          }
          f32x4 d2[2], dgt[2];
#pragma unroll
          for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              d2[qq][r] = dh[qq][r] * sgb[qq][r];
              dgt[qq][r] = dh[qq][r] * t2b[qq][r] * sgb[qq][r] * (1.f - sgb[qq][r]);
            }
          put2(XB, d2[0], d2[1]);
          put2(XG, dgt[0], dgt[1]);
          putT2(TI, n_relu(t1b[0]), n_relu(t1b[1]));
        }
      }
      SF_NC(351);
      n_barrier();
      SF_NC(43 + 10 * (T - 1 - t));
      // the two residual blocks, top down
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int k = 1 - kk;
        const int o_w2T = k ? c.o_w2T[1] : c.o_w2T[0], o_w1T = k ? c.o_w1T[1] : c.o_w1T[0];
        const int g_w2 = k ? c.g_w2[1] : c.g_w2[0], g_b2 = k ? c.g_b2[1] : c.g_b2[0];
        const int g_wg = k ? c.g_wg[1] : c.g_wg[0], g_bg = k ? c.g_bg[1] : c.g_bg[0];
        const int g_w1 = k ? c.g_w1[1] : c.g_w1[0], g_b1 = k ? c.g_b1[1] : c.g_b1[0];
        // B2 / B4: delta of the block's first layer; weight gradients of its second layer and of its gate
        {
          float4 wf[NT];
          ld_hidT(o_w2T, wf);
          const int n2 = NT * NT;
          run_jobs(n2 + NT * NI, [&](int n) -> NJob {
            if (n < n2) {
              const int ot = n / NT, it = n - ot * NT;
              return NJob{XB, TI, gp + g_w2 + n * 256, it == 0 ? gp + g_b2 : nullptr, ot, it};
            }
            const int m = n - n2;
            const int ot = m / NI, it = m - ot * NI;
            return NJob{XG, TIN, gp + g_wg + m * 256, it == 0 ? gp + g_bg : nullptr, ot, it};
          });
          SF_NC(360 + 10 * kk);
          if (has) {
            f32x4 a0, a1;
            go_hidT(wf, XB, a0, a1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              a0[r] = (k ? t1b[0][r] : t1[0][r]) > 0.f ? a0[r] : 0.f;
              a1[r] = (k ? t1b[1][r] : t1[1][r]) > 0.f ? a1[r] : 0.f;
            }
            put2(XA, a0, a1);
            // the block's input, transposed, for the first layer's weight gradient: relu(h1) (block 1) / relu(h0) (block 0)
            f32x4 hin[2];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
#pragma unroll
              for (int r = 0; r < 4; ++r) hin[qq][r] = fmaxf(k ? h0[qq][r] + t2[qq][r] * sg[qq][r] : h0[qq][r], 0.f);
            putT2(TI2, hin[0], hin[1]);
          }
        }
        SF_NC(361 + 10 * kk);
        n_barrier();
        SF_NC(362 + 10 * kk);
        // B3 / B5: gradient at the block's input; (block 1) deltas of block 0's second layer and gate; weight gradients
        // of the block's first layer
        {
          float4 wf[NT];
          ld_hidT(o_w1T, wf);
          run_jobs(NT * NT, [&](int n) -> NJob {
            const int ot = n / NT, it = n - ot * NT;
            return NJob{XA, TI2, gp + g_w1 + n * 256, it == 0 ? gp + g_b1 : nullptr, ot, it};
          });
          SF_NC(363 + 10 * kk);
          if (has) {
            f32x4 a0, a1;
            go_hidT(wf, XA, a0, a1);
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float hin = k ? h0[qq][r] + t2[qq][r] * sg[qq][r] : h0[qq][r];
                const float av = qq ? a1[r] : a0[r];
                dh[qq][r] += hin > 0.f ? av : 0.f;
              }
            if (k == 1) {
              f32x4 d2[2], dgt[2];
#pragma unroll
              for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  d2[qq][r] = dh[qq][r] * sg[qq][r];
                  dgt[qq][r] = dh[qq][r] * t2[qq][r] * sg[qq][r] * (1.f - sg[qq][r]);
                }
              put2(XB, d2[0], d2[1]);
              put2(XG, dgt[0], dgt[1]);
              putT2(TI, n_relu(t1[0]), n_relu(t1[1]));
            } else {
              put2(XB, dh[0], dh[1]);  // delta of the initial layer
            }
          }
        }
        SF_NC(364 + 10 * kk);
        n_barrier();
        SF_NC(365 + 10 * kk);
      }
      SF_NC(44 + 10 * (T - 1 - t));
      // B6: partial sums of Win^T delta over my hidden tile (operand straight from registers); weight gradients of Win;
      // spline waves: u / u' of the transform below (used in B7 and in its backward sweep)
      {
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has) w = n_frag(tp + c.o_winT, NT, 0, j, lane);
        float4 s0 = w, s1 = w, s2 = w, s3 = w;
        run_jobs(NT * NI, [&](int n) -> NJob {
          const int ot = n / NI, it = n - ot * NI;
          return NJob{XB, TIN, gp + c.g_win + n * 256, it == 0 ? gp + c.g_bin : nullptr, ot, it};
        });
        if (spl && t > 0) {
          const float4* srcp = reinterpret_cast<const float4*>(ust + (t - 1) * 16);
          s0 = srcp[0]; s1 = srcp[1]; s2 = srcp[2]; s3 = srcp[3];
        }
        if (has) {
          f32x4 p0 = n_zero(), p1 = n_zero();
          n_mma2(w, dh[0], dh[1], p0, p1, j == NT - 1 ? kc_h : 4);
          put2(XA, p0, p1);
        }
        if (spl && t > 0 && g4 == 0) {   // u / u' of the transform below -> my sample's state
          *reinterpret_cast<float4*>(sst + 8) = s0; *reinterpret_cast<float4*>(sst + 12) = s1;
          *reinterpret_cast<float4*>(sst + 16) = s2; *reinterpret_cast<float4*>(sst + 20) = s3;
        }
      }
      n_barrier();
      SF_NC(45 + 10 * (T - 1 - t));
      // B7 (spline waves): dL/du of the transform below = what came through the spline / the identity + Win^T delta
      if (spl) {
        float G[8];
        ld8(0, G);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float2 v = *reinterpret_cast<const float2*>(XA + (jt * NQ + wave) * PBT + (s + 16 * g) * 4 + 4 * g);
            G[2 * g] += v.x;
            G[2 * g + 1] += v.y;
          }
#pragma unroll
        for (int p = 0; p < 8; ++p) G[p] = p < D ? G[p] : 0.f;
        st8(0, G);
      }
    }
    n_barrier();  // the last reads of XA / XIN are done before the next chunk's inputs are written
    SF_NC(200);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
size_t sf_nsfc_lds_bytes(const SfNscDev& c) {
  const size_t pb = (size_t)(c.NI + 3 * c.NT + c.OTQ + 2) * 2 * PBT;
  const size_t tt = (size_t)(c.NI + 2 * c.NT + 2) * 2 * TT;
  return (pb + tt + 160 + 2 * 16 * 24) * sizeof(float);
}

bool sf_nsfc_eligible(const SfLayout& L, bool want_dctx) {
  const SfNscDev& c = L.nsc;
  static int env = -1;
  if (env < 0) { const char* e = std::getenv("SF_NSFC"); env = e ? std::atoi(e) : 1; }
  if (!env || !c.ok || want_dctx) return false;
  if (c.NT < 2 || c.NT > 5 || (c.OTQ != 6 && c.OTQ != 8) || c.NI < 1 || c.NI > 3) return false;
  return sf_nsfc_lds_bytes(c) <= (size_t)160 * 1024;
}

int sf_nsfc_grid(long B, int NT) {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t pr;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
              ? pr.multiProcessorCount : 256;
  }
  const long chunks = (B + 31) / 32, cap = (NT > 4 ? 1L : 2L) * cus;   // (five-wave workgroups: one per CU)
  return (int)(chunks < cap ? chunks : cap);
}

// Gradient accumulation of a step (measured on cfg3, kernel / step):
//   0 = per-workgroup partials + gather: plain stores, bitwise reproducible.  ONE chunk per workgroup (batch <= 32 x grid =
//       16 384 rows): 301 / 352 us at 16 384 rows -- the fastest form although 512 partials of 0.6 MB cross the HBM twice
//       (the L2s take stores at sixteen times the rate of atomics); later chunks of a persistent workgroup would read-modify-
//       write their partial (3.15 ms per 131 072 rows), so beyond one chunk:
//   2 = f32 atomics into one replica per XCD (385 / 418 us at 16 384, 2.56 ms per 131 072 rows): nothing leaves the L2s, but
//       the order of the adds is the hardware's;
//   3 = the same with 2^-40 fixed-point int64 atomics (sf_fixacc.h): order independent, bitwise reproducible, half the atomic
//       rate (520 us at 16 384) -- taken instead of 2 under SF_DETERMINISTIC=1.
// SF_GRAD_ACC=partial | atomic | fix forces one.
int sf_nsfc_acc_mode(long B, int grid, long n_gradC) {
  static int force = -1, det = -1;
  if (force < 0) {
    const char* e = std::getenv("SF_GRAD_ACC");
    force = !e ? 0 : (e[0] == 'p' ? 1 : (e[0] == 'a' ? 2 : (e[0] == 'f' ? 3 : 0)));
    const char* d = std::getenv("SF_DETERMINISTIC");
    det = d ? std::atoi(d) : 0;
  }
  if (force) return force == 1 ? 0 : force;
  const long chunks = (B + 31) / 32;
  if (chunks <= (long)grid && (size_t)grid * (size_t)n_gradC * sizeof(float) <= ((size_t)1 << 30)) return 0;
  return det == 1 ? 3 : 2;
}

template <int NT, int OTQ>
static hipError_t n_launch(const SfNscArgs& a, int grid, hipStream_t st) {
  static SfAttrCache attr;
  int attr_dev;
  if (attr.need(attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)k_nsf_trainc<NT, OTQ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr.set(attr_dev);
  }
  const size_t sh = sf_nsfc_lds_bytes(a.c);
#ifdef SF_NSC_TRACE
  {
    static unsigned long long* d_tr = nullptr;
    if (!d_tr && hipMalloc(&d_tr, 4 * 512 * 8) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipMemsetAsync(d_tr, 0, 4 * 512 * 8, st);
    SfNscArgs b = a;
    b.trace = d_tr;
    hipLaunchKernelGGL((k_nsf_trainc<NT, OTQ>), dim3((unsigned)grid), dim3(64 * (NT > 4 ? NT : 4)), sh, st, b);
    (void)hipStreamSynchronize(st);
    static unsigned long long h[4 * 512];
    (void)hipMemcpy(h, d_tr, sizeof(h), hipMemcpyDeviceToHost);
    static int calls = 0;
    if (++calls % 8 == 0) {
      fprintf(stderr, "[nsfc trace] B=%ld grid=%d (units of 100 cycles since stamp 0 of wave 0)\n", a.B, grid);
      for (int w = 0; w < 4; ++w) {
        fprintf(stderr, "  wave %d:", w);
        for (int i = 0; i < 512; ++i)
          if (h[w * 512 + i]) fprintf(stderr, " %d:%.1f", i, (double)(long long)(h[w * 512 + i] - h[0]) * 0.01);
        fprintf(stderr, "\n");
      }
    }
    return hipGetLastError();
  }
#endif
  hipLaunchKernelGGL((k_nsf_trainc<NT, OTQ>), dim3((unsigned)grid), dim3(64 * (NT > 4 ? NT : 4)), sh, st, a);
  return hipGetLastError();
}

hipError_t sf_launch_nsf_trainc(const SfNscArgs& a, int grid, hipStream_t st) {
  switch (a.c.NT * 10 + a.c.OTQ) {
    case 26: return n_launch<2, 6>(a, grid, st);
    case 28: return n_launch<2, 8>(a, grid, st);
    case 36: return n_launch<3, 6>(a, grid, st);
    case 38: return n_launch<3, 8>(a, grid, st);
    case 46: return n_launch<4, 6>(a, grid, st);
    case 48: return n_launch<4, 8>(a, grid, st);
    case 56: return n_launch<5, 6>(a, grid, st);
    case 58: return n_launch<5, 8>(a, grid, st);
  }
  return hipErrorInvalidValue;
}
