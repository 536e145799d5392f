// sf_trainc.hip -- cooperative MAF training kernel: forward + backward of -log_prob on 16-row tiles
// (v_mfma_f32_16x16x4_f32), eight waves per 64 samples.        ref: custom_runner.py:585-618 (the training step)
//
// Why it exists (round 3).  k_maf_train (sf_train_kernels.h) gives one 32-sample tile to ONE producer wave: at batch
// 16 384 that is 1.5 waves per SIMD, each running a 270 k-cycle dependent chain, with the activations stashed in HBM
// (72 MB) and every tile adding its own 210 KB weight gradient with f32 atomics.  Here
//   * a workgroup (8 waves = 2 groups of 4) owns 64 samples; inside a group wave j owns ONE hidden tile of 16 units for
//     the group's 32 samples (group 0: tile j, group 1: tile NT-1-j, so that the block-triangular MADE layers load the
//     four SIMDs evenly), and the waves exchange activation tiles through LDS at every layer: a layer's chain is a
//     quarter of the single-wave one;
//   * nothing is stashed in HBM: a1 / a2 of the wave's own tile stay in registers for all T transforms (16 VGPRs per
//     transform), u and the head outputs (a few floats per sample) in LDS; h0 is recomputed (one 16-row product);
//   * weight gradients are products over all 64 samples (K = 64: delta and input tiles transposed through LDS once per
//     layer, read back with conflict-free ds_read_b128), spread over the eight waves, and written with PLAIN stores
//     into the workgroup's own gradient partial: no atomics, no zeroing pass, bitwise reproducible; k_gather_c sums
//     the partials in a fixed order.  A workgroup that is given several chunks adds into its partial.
//   * weight fragments come straight from L2 (each wave needs 4-5 KB per layer), issued a phase ahead.
// Tile = 16 rows x 16 samples in 4 VGPRs: lane l = sample (l & 15) + 16 * row group (l >> 4), register r = row
// 4 * (l >> 4) + r -- the C/D layout of the MFMA and the B-operand order of the next layer.  Image layout: sf_layout.h.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "sf_device.h"
#include "sf_fixacc.h"
#include "sf_internal.h"
#include "sf_trainc.h"

// One translation unit per (stash depth, fragment slots) (Makefile: -DSF_TRC_TU=5 | 6 | 8 for five slots, 15 | 16 | 18 for eight):
// up to sixteen instantiations of the kernel each -- all of them in one unit compile for a quarter of an hour on one core.  Unit 5
// also holds the host side and the gather kernels.
#ifndef SF_TRC_TU
#define SF_TRC_TU 5
#endif
#define SF_TRC_TS_OF_TU (SF_TRC_TU % 10)
#define SF_TRC_NFS_OF_TU (SF_TRC_TU >= 10 ? 8 : 5)

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SF_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace {

__device__ __forceinline__ f32x4 c_mma(const float4 w, const f32x4 in, f32x4 acc) {
  acc = SF_MFMA16(w.x, in[0], acc);
  acc = SF_MFMA16(w.y, in[1], acc);
  acc = SF_MFMA16(w.z, in[2], acc);
  acc = SF_MFMA16(w.w, in[3], acc);
  return acc;
}
// two independent products, their MFMAs alternating (neither chain is dependent back to back)
__device__ __forceinline__ void c_mma2(const float4 wa, const f32x4 ina, f32x4& acca, const float4 wb, const f32x4 inb, f32x4& accb) {
  acca = SF_MFMA16(wa.x, ina[0], acca);
  accb = SF_MFMA16(wb.x, inb[0], accb);
  acca = SF_MFMA16(wa.y, ina[1], acca);
  accb = SF_MFMA16(wb.y, inb[1], accb);
  acca = SF_MFMA16(wa.z, ina[2], acca);
  accb = SF_MFMA16(wb.z, inb[2], accb);
  acca = SF_MFMA16(wa.w, ina[3], acca);
  accb = SF_MFMA16(wb.w, inb[3], accb);
}
__device__ __forceinline__ f32x4 c_ld4(const float* p) {
  const float4 b = *reinterpret_cast<const float4*>(p);
  f32x4 r;
  r[0] = b.x; r[1] = b.y; r[2] = b.z; r[3] = b.w;
  return r;
}
__device__ __forceinline__ void c_st4(float* p, const f32x4 v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
// weight fragment of block (a, b) of a [.][nb] block array
// (block index first, as a wave-uniform 64-bit base; the lane enters as an unsigned 32-bit offset: the load then takes
//  the SGPR-base + VGPR-offset form and every fragment address shares ONE per-lane register)
__device__ __forceinline__ float4 c_frag(const float* wp, int nb, int a, int b, int lane) {
  const float4* base = reinterpret_cast<const float4*>(wp) + (a * nb + b) * 64;
  return base[(unsigned)lane];
}
__device__ __forceinline__ f32x4 c_zero() {
  f32x4 z;
  z[0] = 0.f; z[1] = 0.f; z[2] = 0.f; z[3] = 0.f;
  return z;
}

// One block product split over two partial accumulators (k-steps 0, 2 -> acc0; 1, 3 -> acc1): two MFMAs on one accumulator
// are always two issues apart, so the 40-cycle dependent latency of v_mfma_f32_16x16x4_f32 (issue: 32) never stalls the
// wave, without any extra control flow.  The caller adds the partials once per layer.
__device__ __forceinline__ void c_mma_alt(const float4 w, const f32x4 in, f32x4& acc0, f32x4& acc1) {
  acc0 = SF_MFMA16(w.x, in[0], acc0);
  acc1 = SF_MFMA16(w.y, in[1], acc1);
  acc0 = SF_MFMA16(w.z, in[2], acc0);
  acc1 = SF_MFMA16(w.w, in[3], acc1);
}
// LDS tile addressing: a tile (16 rows x 16 samples of subtile q) is 256 floats
//   B layout: lane l holds float4 at l*4 (rows 4*(l>>4)+r of sample l&15): the MFMA B operand of a data product
//   T layout: [sample >> 2][row][sample & 3]: lane l = 16*kk + i reads float4 at l*4 = row i, samples 4kk..4kk+3:
//             A / B operand of a weight-gradient product (k = sample)
//   (round 4: sample groups of 68 floats, not 64 -- TTS = 272 floats per tile: the four ds_write_b32 of a lane then hit
//   4 (s >> 2) + 16 g4 + (s & 3) + 4 r = 64 different banks; with 64 the lanes s, s + 4, s + 8, s + 12 collided and 41 % of the
//   kernel's LDS cycles were bank conflicts, profiles/r03_pmc_summary.json)
#define TTS 272
__device__ __forceinline__ int c_T_rd(int lane) { return (lane >> 4) * 68 + (lane & 15) * 4; }
__device__ __forceinline__ void c_put_T(float* tile, const f32x4 v, int s, int g4) {
  float* p = tile + (s >> 2) * 68 + (s & 3) + 16 * g4;
  p[0] = v[0]; p[4] = v[1]; p[8] = v[2]; p[12] = v[3];
}
__device__ __forceinline__ float c_scale(int scale_fn, float av, float eps) {
  return (scale_fn == 0 ? sf_softplus(av) : sf_sigmoid(av + 2.0f)) + eps;
}
__device__ __forceinline__ float c_dscale(int scale_fn, float av) {
  if (scale_fn == 0) return sf_sigmoid(av);
  const float sg = sf_sigmoid(av + 2.0f);
  return sg * (1.0f - sg);
}

// Weight-gradient blocks: acc = sum over the 64 samples of delta[ot rows][s] * in[it rows][s]; plain store (first chunk
// of the workgroup) or add into the workgroup's partial.  gb != nullptr: also the bias gradient of tile ot.  A wave that
// holds two blocks of a batch runs them together (independent accumulators hide the MFMA's dependent latency).
struct CJob {
  const float* Td;
  const float* Ti;
  float* gw;
  float* gb;
  int ot, it;
};
__device__ __forceinline__ void c_dw_finish(const CJob& J, f32x4 acc, float bs, const SfAcc& A, int lane) {
  if (A.mode == 3) {  // fixed-point atomics into the XCD's replica (sf_fixacc.h)
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
      if (A.mode == 3) sf_fix_add(A.fix + (J.gb - A.base) + J.ot * 16 + lane, bs);
      else J.gb[J.ot * 16 + lane] = A.mode == 1 ? J.gb[J.ot * 16 + lane] + bs : bs;
    }
  }
}
template <int NQ>
__device__ __forceinline__ void c_dw_jobs(const CJob& A, const CJob& B, bool two, const SfAcc& accumulate, int lane) {
  f32x4 accA = c_zero(), accB = c_zero();
  float bsA = 0.f, bsB = 0.f;
  if (two) {  // (two blocks at once: their chains alternate, the 40-cycle dependent latency of the MFMA is hidden)
#pragma unroll 2
    for (int q = 0; q < NQ; ++q) {  // (two subtiles' operands in flight at a time: 32 registers, not 64)
      const float4 dA = *reinterpret_cast<const float4*>(A.Td + (A.ot * NQ + q) * TTS + c_T_rd(lane));
      const float4 iA = *reinterpret_cast<const float4*>(A.Ti + (A.it * NQ + q) * TTS + c_T_rd(lane));
      const float4 dB = *reinterpret_cast<const float4*>(B.Td + (B.ot * NQ + q) * TTS + c_T_rd(lane));
      const float4 iB = *reinterpret_cast<const float4*>(B.Ti + (B.it * NQ + q) * TTS + c_T_rd(lane));
      accA = SF_MFMA16(dA.x, iA.x, accA);
      accB = SF_MFMA16(dB.x, iB.x, accB);
      accA = SF_MFMA16(dA.y, iA.y, accA);
      accB = SF_MFMA16(dB.y, iB.y, accB);
      accA = SF_MFMA16(dA.z, iA.z, accA);
      accB = SF_MFMA16(dB.z, iB.z, accB);
      accA = SF_MFMA16(dA.w, iA.w, accA);
      accB = SF_MFMA16(dB.w, iB.w, accB);
      bsA += (dA.x + dA.y) + (dA.z + dA.w);
      bsB += (dB.x + dB.y) + (dB.z + dB.w);
    }
    c_dw_finish(A, accA, bsA, accumulate, lane);
    c_dw_finish(B, accB, bsB, accumulate, lane);
  } else {  // one block: two partial accumulators over alternating k-steps, for the same reason
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const float4 dA = *reinterpret_cast<const float4*>(A.Td + (A.ot * NQ + q) * TTS + c_T_rd(lane));
      const float4 iA = *reinterpret_cast<const float4*>(A.Ti + (A.it * NQ + q) * TTS + c_T_rd(lane));
      accA = SF_MFMA16(dA.x, iA.x, accA);
      accB = SF_MFMA16(dA.y, iA.y, accB);
      accA = SF_MFMA16(dA.z, iA.z, accA);
      accB = SF_MFMA16(dA.w, iA.w, accB);
      bsA += (dA.x + dA.y) + (dA.z + dA.w);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) accA[r] += accB[r];
    c_dw_finish(A, accA, bsA, accumulate, lane);
  }
}

}  // namespace

// The argument block is read through the kernarg segment pointer, laundered where a phase starts: descriptor fields
// are then scalar loads next to their use (scalar-cache hits) instead of ~80 SGPRs held -- and spilled -- for the
// life of the kernel (same device as k_maf_samp16).
__device__ __forceinline__ const SfTrcArgs& c_args() {
  auto kp = __builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));
  return *(const SfTrcArgs*)kp;
}

// Workgroup barrier for the LDS hand-overs.  __syncthreads() carries a workgroup-scope fence that also drains the
// wave's outstanding GLOBAL loads (s_waitcnt vmcnt(0)): every weight fragment issued a phase ahead would be waited for
// at the very next barrier.  The exchanged data is LDS only: wait for the wave's own LDS operations, then s_barrier.
__device__ __forceinline__ void c_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  SF_FUZZ();
}

// Block offsets of the cooperative image and of a gradient partial: functions of (NT, NI) alone (sf_layout.cpp emits the blocks
// in this order; sf_trainc_eligible checks the descriptor against these formulas), so the kernel carries them as
// immediates instead of ~25 descriptor words re-read from the kernarg segment in every phase.
template <int NT, int NI>
struct CLay {
  static constexpr int o_win = 0;
  static constexpr int o_b0 = o_win + NT * NI * 256;
  static constexpr int o_w1 = o_b0 + NT * 16;
  static constexpr int o_b1 = o_w1 + NT * NT * 256;
  static constexpr int o_w2 = o_b1 + NT * 16;
  static constexpr int o_b2 = o_w2 + NT * NT * 256;
  static constexpr int o_wf = o_b2 + NT * 16;
  static constexpr int o_bf = o_wf + NT * 256;
  static constexpr int o_wfT = o_bf + 16;
  static constexpr int o_w2T = o_wfT + NT * 256;
  static constexpr int o_w1T = o_w2T + NT * NT * 256;
  static constexpr int o_winT = o_w1T + NT * NT * 256;
  static constexpr int t_end = o_winT + NI * NT * 256;
  static constexpr int t_stride = (t_end + 255) / 256 * 256;
  static constexpr int g_win = 0;
  static constexpr int g_b0 = g_win + NT * NI * 256;
  static constexpr int g_w1 = g_b0 + NT * 16;
  static constexpr int g_b1 = g_w1 + NT * NT * 256;
  static constexpr int g_w2 = g_b1 + NT * 16;
  static constexpr int g_b2 = g_w2 + NT * NT * 256;
  static constexpr int g_wf = g_b2 + NT * 16;
  static constexpr int g_bf = g_wf + NT * 256;
  static constexpr int g_end = g_bf + 16;
  static constexpr int g_stride = (g_end + 63) / 64 * 64;
  static bool matches(const SfTrcDev& c, int T) {
    return c.o_win == o_win && c.o_b0 == o_b0 && c.o_w1 == o_w1 && c.o_b1 == o_b1 && c.o_w2 == o_w2 && c.o_b2 == o_b2 &&
           c.o_wf == o_wf && c.o_bf == o_bf && c.o_wfT == o_wfT && c.o_w2T == o_w2T && c.o_w1T == o_w1T && c.o_winT == o_winT &&
           (T == 1 || c.t_stride == t_stride) && c.g_win == g_win && c.g_b0 == g_b0 && c.g_w1 == g_w1 && c.g_b1 == g_b1 &&
           c.g_w2 == g_w2 && c.g_b2 == g_b2 && c.g_wf == g_wf && c.g_bf == g_bf && c.g_stride == g_stride;
  }
};

// floats of the constant block kept in LDS for the life of the workgroup
__host__ __device__ inline int sf_trc_nbias(int NT) { return 3 * NT * 16 + 16; }
__host__ __device__ inline int sf_trc_cb_floats(int NT, int NI, int TS) { return TS * sf_trc_nbias(NT) + 16 + NI * 16 * 3 + 8 * 3; }

// NG groups of 4 waves per workgroup, 32 samples (two 16-sample subtiles) per group.  Wave (p, qq) of a group owns
// hidden tiles p and NT-1-p of subtile qq: with the block-triangular MADE layers that is (p + 1) + (NT - p) = NT + 1
// blocks per layer for EVERY wave (round 3's first version gave wave j tile j for both subtiles: the wave with the
// last tile ran four blocks while the first ran one, and every barrier waited for it).
// NFS = fragment slots per masked layer and wave: 5 covers the aligned degree placement ((p + 1) + (NT - p) = NT + 1 blocks), 8 the
// contiguous ("span") placements, whose degree groups straddle tiles and unmask more blocks (up to NT per tile: D = 6 with H = 50,
// H = 64 with D >= 6 -- the width of the reference's example CLI); the three extra fragments per layer cost registers (scratch).
// workgroups per CU the register allocation is made for: the 8-wave form (NG = 2) holds 145 KB of LDS -- ONE per CU, two waves per
// SIMD, so a wave may take 256 registers (round 3-4 compiled it for two per CU: 128 registers, 48 B of scratch)
#ifndef SF_TRC_WGS
#define SF_TRC_WGS(NG) ((NG) == 2 ? 1 : 2)
#endif
template <int TS, int NI, int NT, int NG, int NFS = 5>
__global__ __launch_bounds__(256 * NG, SF_TRC_WGS(NG)) void k_maf_trainc(SfTrcArgs a_in) {
  extern __shared__ float lds[];
  constexpr int NQ = 2 * NG, NW = 4 * NG, NTH = 256 * NG;
  constexpr int NP = (NT + 1) / 2;  // wave rows (p) that own tiles
  using L = CLay<NT, NI>;
  const SfTrcArgs& a = c_args();
  const SfTrcDev& c = a.c;
  // (the wave index is made visibly wave-uniform: tile indices, buffer bases and fragment bases then live in SGPRs and the
  //  per-lane address part is one register)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2, p = (wave >> 1) & 1, qq = wave & 1;
  const int q = 2 * grp + qq;  // my subtile
  const int s = lane & 15, g4 = lane >> 4;
  const int D = a.D;
  const int tA = p, tB = NT - 1 - p;
  const bool has0 = tA <= tB, has1 = tA < tB;  // tiles this wave owns: tA, and tB when different
  const int tB_ = has1 ? tB : tA;
  int keA = 0, keB = 0, kbA = 0, kbB = 0;  // tile bounds (select chains: the descriptor stays in SGPRs)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    keA = (i == tA) ? c.kend[i] : keA;
    keB = (i == tB_) ? c.kend[i] : keB;
    kbA = (i == tA) ? c.kbeg[i] : kbA;
    kbB = (i == tB_) ? c.kbeg[i] : kbB;
  }
  // block slots of a masked layer: tile A's blocks first, then tile B's (at most 5 in all: sf_trainc_eligible)
  const int nfA = has0 ? keA + 1 : 0, nfT = nfA + (has1 ? keB + 1 : 0);        // forward: input tiles 0..ke
  const int nbA = has0 ? NT - kbA : 0, nbT = nbA + (has1 ? NT - kbB : 0);      // backward: output tiles kb..NT-1
  // ---- LDS map (floats)
  constexpr int HSZ = NT * NQ * 256;     // one hidden-size tensor: NT tiles x NQ subtiles x 256
  float* XBa = lds;                      // B layout: h0 (fwd) / dpre2 (bwd); backward partial sums of du
  float* XBb = XBa + HSZ;                // B layout: a1 (fwd) / dpre1 (bwd)
  constexpr int HST = NT * NQ * TTS;     // ... in the (padded) T layout
  float* TB0 = XBb + HSZ;                // T layout x5: TD2, TA2 (later TD0), TH0, TD1, TA1; forward: head partial sums
  float* TD2 = TB0, *TA2 = TB0 + HST, *TH0 = TB0 + 2 * HST, *TD1 = TB0 + 3 * HST, *TA1 = TB0 + 4 * HST;
  float* TD0 = TA2;
  float* PBf = TB0;                      // [NP][NQ] tiles
  float* PBb = XBa;                      // [NP][NQ] tiles
  float* TDF = TB0 + 5 * HST;            // T layout: head delta, 1 tile x NQ
  float* TIN = TDF + NQ * TTS;           // T layout: input tiles, NI tiles x NQ
  float* USt = TIN + NI * NQ * TTS;      // [TS][16*NQ samples][8]: u entering transform t
  float* ASt = USt + TS * NQ * 128;      // [TS][16*NQ][8]: head output a (slots)
  // constants of the whole call: biases of every transform, the weight-gradient block list, per input-row and
  // per-slot standardisation constants (read from LDS in the phases instead of through dependent global loads)
  constexpr int NBIAS = 3 * NT * 16 + 16;
  float* CBb = ASt + TS * NQ * 128;      // [T][b0 NT*16 | b1 | b2 | bf 16]
  int* CBj = reinterpret_cast<int*>(CBb + TS * NBIAS);  // [16] (ot * 4 + it)
  float* CBx = reinterpret_cast<float*>(CBj + 16);      // [NI*16][feature, mean, 1/std]
  float* CBs = CBx + NI * 16 * 3;                       // [8 slots][theta column, scale, shift]
  {
    const float* cst = a.cst;
    for (int i = threadIdx.x; i < a.T * NBIAS; i += NTH) {
      const int t = i / NBIAS, k = i - t * NBIAS;
      const int sec = k / (NT * 16), kk = k - sec * NT * 16;
      const int off = sec == 0 ? L::o_b0 : (sec == 1 ? L::o_b1 : (sec == 2 ? L::o_b2 : L::o_bf));
      CBb[i] = a.img[(size_t)t * L::t_stride + off + kk];
    }
    if (threadIdx.x < 16) CBj[threadIdx.x] = threadIdx.x < c.n_jobs ? (int)cst[c.c_jobs + threadIdx.x] : 0;
    if (threadIdx.x < NI * 16) {
      const int f = (int)cst[c.c_insrc + threadIdx.x];
      const int ff = f >= 0 ? f : 0;
      CBx[threadIdx.x * 3] = (float)f;
      CBx[threadIdx.x * 3 + 1] = cst[a.c_xmean + ff];
      CBx[threadIdx.x * 3 + 2] = 1.0f / cst[a.c_xstd + ff];
    }
    if (threadIdx.x < 8) {
      const int pp = threadIdx.x < D ? threadIdx.x : 0;
      CBs[threadIdx.x * 3] = cst[a.c_tdim + pp];
      CBs[threadIdx.x * 3 + 1] = cst[a.c_pscale + pp];
      CBs[threadIdx.x * 3 + 2] = cst[a.c_pshift + pp];
    }
  }
  __syncthreads();
  const bool s0on = 2 * g4 < D, s1on = 2 * g4 + 1 < D;
  static_assert(NG <= 2, "groups");
  const bool swapped = NG == 2 ? grp == 1 : (blockIdx.x & 1) != 0;  // (one group: alternate workgroups instead)
  // slot i of a masked layer -> (which of my tiles, block index); wave-uniform
  auto fslot_tile = [&](int i) { return i < nfA ? tA : tB_; };
  auto fslot_it = [&](int i) { const int ii = i < nfT ? i : (nfT > 0 ? nfT - 1 : 0); return ii < nfA ? ii : ii - nfA; };
  auto bslot_tile = [&](int i) { return i < nbA ? tA : tB_; };
  auto bslot_ot = [&](int i) { const int ii = i < nbT ? i : (nbT > 0 ? nbT - 1 : 0); return ii < nbA ? kbA + ii : kbB + ii - nbA; };

  for (long chunk = blockIdx.x, iter = 0; chunk < a.n_chunks; chunk += gridDim.x, ++iter) {
    SF_TC(0);
    // first fragments of the forward sweep: in flight behind the input loads
    float4 pwin[2][NI];
    {
      const float* tp0 = a.img + sf_opaque_zero();
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        pwin[0][it] = c_frag(tp0 + L::o_win, NI, tA, it, lane);
        pwin[1][it] = c_frag(tp0 + L::o_win, NI, tB_, it, lane);
      }
    }
    // ------------------------------------------------------------------ per-sample inputs (one subtile per wave)
    const long row = chunk * (16 * NQ) + q * 16 + s;
    const bool valid = row < a.B;
    float wgt;
    f32x4 inx[NI];  // context part of the input tiles (slot rows 0 here)
    float u0, u1;
    {
      const long ii = valid ? row : a.B - 1;
      const long src = a.idx ? (long)a.idx[ii] : ii;
      wgt = valid ? (a.wts ? a.w * a.wts[row] : a.w) : 0.f;
      const float* xr = a.x + src * a.C;
#pragma unroll
      for (int it = 0; it < NI; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* cb = CBx + (it * 16 + 4 * g4 + r) * 3;
          const int f = (int)cb[0];
          const float v = (xr[f >= 0 ? f : 0] - cb[1]) * cb[2];
          inx[it][r] = f >= 0 ? v : 0.f;
        }
      }
      const float* sb = CBs + (2 * g4) * 3;
      // (unconditional loads, selected afterwards: a load under a per-lane condition gets its own wait)
      const float th0 = a.theta[src * D + (int)sb[0]], th1 = a.theta[src * D + (int)sb[3]];
      u0 = s0on ? th0 * sb[1] + sb[2] : 0.f;
      u1 = s1on ? th1 * sb[4] + sb[5] : 0.f;
    }
    float ld = 0.f;
    f32x4 a1s[TS][2], a2s[TS][2];  // my two tiles of a1 / a2 for every transform

    // ------------------------------------------------------------------ forward
#pragma unroll
    for (int t = 0; t < TS; ++t) {
      if (t < a.T) {
        const SfTrcArgs& a = c_args();
        const SfTrcDev& c = a.c;
        const float* tp = a.img + (size_t)t * L::t_stride + sf_opaque_zero();  // (not loop-invariant: see sf_device.h)
        const float* cb = CBb + t * NBIAS;
        // F1: h0 = b0 + bc + Win . [u ; e(x)]
        SF_TC(1 + 5 * t);
        float4 w1f[NFS], w2f[NFS];
        if (has0) {
          // next phase's fragments: in flight across the barrier
#pragma unroll
          for (int i = 0; i < NFS; ++i) w1f[i] = c_frag(tp + L::o_w1, NT, fslot_tile(i), fslot_it(i), lane);
          f32x4 in0 = inx[0];
          in0[0] = s0on ? u0 : in0[0];
          in0[1] = s1on ? u1 : in0[1];
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            if (k == 0 || has1) {
              const int tk = k == 0 ? tA : tB_;
              f32x4 h0 = c_ld4(cb + (tk * 4 + g4) * 4);
              h0 = c_mma(pwin[k][0], in0, h0);
              if constexpr (NI > 1) h0 = c_mma(pwin[k][NI - 1], inx[NI - 1], h0);
              c_st4(XBa + (tk * NQ + q) * 256 + lane * 4, h0);
            }
          }
        }
        SF_TC(2 + 5 * t);
        c_barrier();
        // F2: a1 = tanh(b1 + W1 h0)
        float4 wff[2];
        SF_TCX(t == 1, 200, u0);
        if (has0) {
#pragma unroll
          for (int i = 0; i < NFS; ++i) w2f[i] = c_frag(tp + L::o_w2, NT, fslot_tile(i), fslot_it(i), lane);
          wff[0] = c_frag(tp + L::o_wf, NT, 0, tA, lane);
          wff[1] = c_frag(tp + L::o_wf, NT, 0, tB_, lane);
          f32x4 accA = c_ld4(cb + NT * 16 + (tA * 4 + g4) * 4), accB = c_ld4(cb + NT * 16 + (tB_ * 4 + g4) * 4);
          SF_TCX(t == 1, 201, accA[0] + accB[0]);
          SF_TCX(t == 1, 202, w1f[0].x + w1f[4].x);
{
            f32x4 accA1 = c_zero(), accB1 = c_zero();
#pragma unroll
            for (int i = 0; i < NFS; ++i) {
              if (i < nfT) {
                const f32x4 tv = c_ld4(XBa + (fslot_it(i) * NQ + q) * 256 + lane * 4);
                if (i < nfA) c_mma_alt(w1f[i], tv, accA, accA1);
                else c_mma_alt(w1f[i], tv, accB, accB1);
              }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { accA[r] += accA1[r]; accB[r] += accB1[r]; }
          }
          SF_TCX(t == 1, 203, accA[0] + accB[0] + accA[3] + accB[3]);
#pragma unroll
          for (int r = 0; r < 4; ++r) { accA[r] = sf_tanh(accA[r]); accB[r] = sf_tanh(accB[r]); }
          SF_TCX(t == 1, 204, accA[0] + accB[0] + accA[3] + accB[3]);
          a1s[t][0] = accA; a1s[t][1] = accB;
          c_st4(XBb + (tA * NQ + q) * 256 + lane * 4, accA);
          if (has1) c_st4(XBb + (tB_ * NQ + q) * 256 + lane * 4, accB);
          SF_TCX(t == 1, 205, accA[0]);
        }
        SF_TC(3 + 5 * t);
        c_barrier();
        // F3: a2 = tanh(b2 + W2 a1); head partial sums over my hidden rows
        if (has0) {
          {  // the next transform's first fragments (the last transform reloads its own: no branch around a load)
            const float* tpn = tp + (t + 1 < a.T ? L::t_stride : 0);
#pragma unroll
            for (int it = 0; it < NI; ++it) {
              pwin[0][it] = c_frag(tpn + L::o_win, NI, tA, it, lane);
              pwin[1][it] = c_frag(tpn + L::o_win, NI, tB_, it, lane);
            }
          }
          f32x4 accA = c_ld4(cb + 2 * NT * 16 + (tA * 4 + g4) * 4), accB = c_ld4(cb + 2 * NT * 16 + (tB_ * 4 + g4) * 4);
{
            f32x4 accA1 = c_zero(), accB1 = c_zero();
#pragma unroll
            for (int i = 0; i < NFS; ++i) {
              if (i < nfT) {
                const f32x4 tv = c_ld4(XBb + (fslot_it(i) * NQ + q) * 256 + lane * 4);
                if (i < nfA) c_mma_alt(w2f[i], tv, accA, accA1);
                else c_mma_alt(w2f[i], tv, accB, accB1);
              }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { accA[r] += accA1[r]; accB[r] += accB1[r]; }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) { accA[r] = sf_tanh(accA[r]); accB[r] = sf_tanh(accB[r]); }
          a2s[t][0] = accA; a2s[t][1] = accB;
          f32x4 hp = c_mma(wff[0], accA, c_zero());
          if (has1) hp = c_mma(wff[1], accB, hp);
          c_st4(PBf + (p * NQ + q) * 256 + lane * 4, hp);
        }
        SF_TC(4 + 5 * t);
        c_barrier();
        SF_TC(5 + 5 * t);
        // F4 (both waves of the subtile, replicated): head = bf + partial sums; affine update of this lane's two slots
        {
          f32x4 fin = c_ld4(cb + 3 * NT * 16 + g4 * 4);
#pragma unroll
          for (int pp = 0; pp < NP; ++pp) {
            const f32x4 pv = c_ld4(PBf + (pp * NQ + q) * 256 + lane * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) fin[r] += pv[r];
          }
          if (p == 0) {  // one wave per subtile keeps what the backward sweep needs
            const int sidx = ((t * NQ + q) * 16 + s) * 8 + 2 * g4;
            *reinterpret_cast<float2*>(USt + sidx) = make_float2(u0, u1);
            *reinterpret_cast<float2*>(ASt + sidx) = make_float2(fin[0], fin[1]);
          }
          const float sc0 = c_scale(a.scale_fn, fin[0], a.eps), sc1 = c_scale(a.scale_fn, fin[1], a.eps);
          u0 = s0on ? sc0 * u0 + fin[2] : 0.f;
          u1 = s1on ? sc1 * u1 + fin[3] : 0.f;
          ld += (s0on ? sf_log(sc0) : 0.f) + (s1on ? sf_log(sc1) : 0.f);
        }
      }
    }
    // first fragments of the backward sweep (top transform): in flight behind the loss
    float4 pwfT[2], pw2T[NFS], pwinB[2][NI];
    {
      const SfTrcArgs& a = c_args();
      const SfTrcDev& c = a.c;
      const float* tpl = a.img + (size_t)(a.T - 1) * L::t_stride + sf_opaque_zero();
      pwfT[0] = c_frag(tpl + L::o_wfT, 1, tA, 0, lane);
      pwfT[1] = c_frag(tpl + L::o_wfT, 1, tB_, 0, lane);
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        pwinB[0][it] = c_frag(tpl + L::o_win, NI, tA, it, lane);
        pwinB[1][it] = c_frag(tpl + L::o_win, NI, tB_, it, lane);
      }
#pragma unroll
      for (int i = 0; i < NFS; ++i) pw2T[i] = c_frag(tpl + L::o_w2T, NT, bslot_tile(i), bslot_ot(i), lane);
    }
    // ------------------------------------------------------------------ loss, dL/du_T
    float G0, G1;
    {
      float ss = u0 * u0 + u1 * u1;
      float lds_ = ld;
      ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);
      lds_ += __shfl_xor(lds_, 16, 64); lds_ += __shfl_xor(lds_, 32, 64);
      const float nll = 0.5f * ss + 0.5f * (float)D * 1.8378770664093453f - (a.logdet0 + lds_);
      if (p == 0) {
        if (a.loss && valid && g4 == 0) a.loss[row] = nll;
        if (a.loss_sum) {
          float tsum = (valid && g4 == 0) ? nll : 0.f;
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) tsum += __shfl_xor(tsum, o, 64);
          // values on a 2^-20 grid add exactly in double: the sum does not depend on the order of the atomics
          if (lane == 0) atomicAdd(a.loss_sum + (blockIdx.x & a.loss_mask), (double)rintf(tsum * 1048576.0f) * (1.0 / 1048576.0));
        }
      }
      G0 = wgt * u0;
      G1 = wgt * u1;
    }
    SF_TC(39);
    c_barrier();  // the last head sums have been read (PBf is TD2's buffer), the u / a stash is complete

    // ------------------------------------------------------------------ backward
    // gradient target: this workgroup's partial (plain stores; adds from its second chunk on), or -- a.fix -- the fixed-point
    // replica of its XCD (the job descriptors then carry offsets from a.gpart that are never dereferenced as floats)
    float* gpart = a.fix ? a.gpart : a.gpart + (size_t)blockIdx.x * a.gpart_stride;
    const SfAcc accumulate = {a.fix ? 3 : (iter > 0 ? 1 : 0), a.gpart, a.fix ? a.fix + (size_t)sf_xcc_id() * a.gpart_stride : nullptr};
#pragma unroll
    for (int tt = 0; tt < TS; ++tt) {
      const int t = TS - 1 - tt;
      if (t < a.T) {
        const SfTrcArgs& a = c_args();
        const SfTrcDev& c = a.c;
        const float* tp = a.img + (size_t)t * L::t_stride + sf_opaque_zero();
        const float* cb = CBb + t * NBIAS;
        float* gp = gpart + (size_t)t * L::g_stride;
        // B1: head backward, delta of block 2, h0 recomputed
        SF_TC(40 + 12 * tt);
        float Gd0, Gd1;
        {
          const int sidx = ((t * NQ + q) * 16 + s) * 8 + 2 * g4;
          const float2 uu = *reinterpret_cast<const float2*>(USt + sidx);
          const float2 aa = *reinterpret_cast<const float2*>(ASt + sidx);
          const float sc0 = c_scale(a.scale_fn, aa.x, a.eps), sc1 = c_scale(a.scale_fn, aa.y, a.eps);
          f32x4 dfin;
          dfin[0] = s0on ? (G0 * uu.x - sf_div(wgt, sc0)) * c_dscale(a.scale_fn, aa.x) : 0.f;
          dfin[1] = s1on ? (G1 * uu.y - sf_div(wgt, sc1)) * c_dscale(a.scale_fn, aa.y) : 0.f;
          dfin[2] = s0on ? G0 : 0.f;
          dfin[3] = s1on ? G1 : 0.f;
          Gd0 = s0on ? G0 * sc0 : 0.f;
          Gd1 = s1on ? G1 * sc1 : 0.f;
          // input tiles of this transform (slot rows from the stash)
          f32x4 in0 = inx[0];
          in0[0] = s0on ? uu.x : in0[0];
          in0[1] = s1on ? uu.y : in0[1];
          if (p == 0) {
            c_put_T(TDF + q * TTS, dfin, s, g4);
            c_put_T(TIN + q * TTS, in0, s, g4);
            if constexpr (NI > 1) c_put_T(TIN + (NQ + q) * TTS, inx[NI - 1], s, g4);
          }
          if (has0) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              if (k == 0 || has1) {
                const int tk = k == 0 ? tA : tB_;
                f32x4 dp2 = c_mma(pwfT[k], dfin, c_zero());
#pragma unroll
                for (int r = 0; r < 4; ++r) dp2[r] *= 1.0f - a2s[t][k][r] * a2s[t][k][r];
                c_st4(XBa + (tk * NQ + q) * 256 + lane * 4, dp2);
                c_put_T(TD2 + (tk * NQ + q) * TTS, dp2, s, g4);
                c_put_T(TA2 + (tk * NQ + q) * TTS, a2s[t][k], s, g4);
                f32x4 h0 = c_mma(pwinB[k][0], in0, c_ld4(cb + (tk * 4 + g4) * 4));
                if constexpr (NI > 1) h0 = c_mma(pwinB[k][NI - 1], inx[NI - 1], h0);
                c_put_T(TH0 + (tk * NQ + q) * TTS, h0, s, g4);
              }
            }
          }
        }
        SF_TC(41 + 12 * tt);
        c_barrier();
        SF_TC(42 + 12 * tt);
        // B2: delta of block 1; weight gradients of the head.  The two halves of a backward phase (data path, weight-
        // gradient blocks) are independent: odd groups run them in the opposite order, so that the two waves that share
        // a SIMD are not in the same kind of work at the same time (MFMA chain + tanh' epilogue vs LDS-fed block products)
        float4 w1T[NFS], wiT[2];
        auto b2_data = [&]() {
        if (has0) {
#pragma unroll
          for (int i = 0; i < NFS; ++i) w1T[i] = c_frag(tp + L::o_w1T, NT, bslot_tile(i), bslot_ot(i), lane);
          wiT[0] = c_frag(tp + L::o_winT, NT, 0, tA, lane);
          wiT[1] = c_frag(tp + L::o_winT, NT, 0, tB_, lane);
          f32x4 accA = c_zero(), accB = c_zero();
{
            f32x4 accA1 = c_zero(), accB1 = c_zero();
#pragma unroll
            for (int i = 0; i < NFS; ++i) {
              if (i < nbT) {
                const f32x4 tv = c_ld4(XBa + (bslot_ot(i) * NQ + q) * 256 + lane * 4);
                if (i < nbA) c_mma_alt(pw2T[i], tv, accA, accA1);
                else c_mma_alt(pw2T[i], tv, accB, accB1);
              }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { accA[r] += accA1[r]; accB[r] += accB1[r]; }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            accA[r] *= 1.0f - a1s[t][0][r] * a1s[t][0][r];
            accB[r] *= 1.0f - a1s[t][1][r] * a1s[t][1][r];
          }
          c_st4(XBb + (tA * NQ + q) * 256 + lane * 4, accA);
          c_put_T(TD1 + (tA * NQ + q) * TTS, accA, s, g4);
          c_put_T(TA1 + (tA * NQ + q) * TTS, a1s[t][0], s, g4);
          if (has1) {
            c_st4(XBb + (tB_ * NQ + q) * 256 + lane * 4, accB);
            c_put_T(TD1 + (tB_ * NQ + q) * TTS, accB, s, g4);
            c_put_T(TA1 + (tB_ * NQ + q) * TTS, a1s[t][1], s, g4);
          }
        }
        };
        auto b2_jobs = [&]() {
        for (int n = wave; n < NT; n += 2 * NW) {  // dWf[head tile][hidden tile]; bias with the first block
          const bool two = n + NW < NT;
          const int n2 = two ? n + NW : n;
          const CJob A = {TDF, TA2, gp + L::g_wf + n * 256, n == 0 ? gp + L::g_bf : nullptr, 0, n};
          const CJob B = {TDF, TA2, gp + L::g_wf + n2 * 256, nullptr, 0, n2};
          c_dw_jobs<NQ>(A, B, two, accumulate, lane);
        }
        };
        if (swapped) { b2_jobs(); SF_TC(43 + 12 * tt); b2_data(); } else { b2_data(); SF_TC(43 + 12 * tt); b2_jobs(); }
        SF_TC(44 + 12 * tt);
        c_barrier();
        SF_TC(45 + 12 * tt);
        // B3: delta of the initial layer, partial sums of W_in^T delta; weight gradients of block 2
        auto b3_data = [&]() {
        if (has0) {
          f32x4 accA = c_zero(), accB = c_zero();
{
            f32x4 accA1 = c_zero(), accB1 = c_zero();
#pragma unroll
            for (int i = 0; i < NFS; ++i) {
              if (i < nbT) {
                const f32x4 tv = c_ld4(XBb + (bslot_ot(i) * NQ + q) * 256 + lane * 4);
                if (i < nbA) c_mma_alt(w1T[i], tv, accA, accA1);
                else c_mma_alt(w1T[i], tv, accB, accB1);
              }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { accA[r] += accA1[r]; accB[r] += accB1[r]; }
          }
          c_put_T(TD0 + (tA * NQ + q) * TTS, accA, s, g4);
          f32x4 dup = c_mma(wiT[0], accA, c_zero());
          if (has1) {
            c_put_T(TD0 + (tB_ * NQ + q) * TTS, accB, s, g4);
            dup = c_mma(wiT[1], accB, dup);
          }
          c_st4(PBb + (p * NQ + q) * 256 + lane * 4, dup);
        }
        };
        auto b3_jobs = [&]() {
          const int nj = c.n_jobs;
          for (int n = wave; n < nj; n += 2 * NW) {
            const bool two = n + NW < nj;
            const int cA = CBj[n], cB = CBj[two ? n + NW : n];
            const CJob A = {TD2, TA1, gp + L::g_w2 + ((cA >> 2) * NT + (cA & 3)) * 256, (cA & 3) == 0 ? gp + L::g_b2 : nullptr, cA >> 2, cA & 3};
            const CJob B = {TD2, TA1, gp + L::g_w2 + ((cB >> 2) * NT + (cB & 3)) * 256, (cB & 3) == 0 ? gp + L::g_b2 : nullptr, cB >> 2, cB & 3};
            c_dw_jobs<NQ>(A, B, two, accumulate, lane);
          }
        };
        if (swapped) { b3_jobs(); SF_TC(46 + 12 * tt); b3_data(); } else { b3_data(); SF_TC(46 + 12 * tt); b3_jobs(); }
        SF_TC(47 + 12 * tt);
        c_barrier();
        SF_TC(48 + 12 * tt);
        // B4 (both waves of the subtile, replicated): dL/du of the transform below; weight gradients of block 1 and
        // the initial layer
        {  // first fragments of the transform below (the bottom transform reloads its own)
          const float* tpn = tp - (t >= 1 ? L::t_stride : 0);
          pwfT[0] = c_frag(tpn + L::o_wfT, 1, tA, 0, lane);
          pwfT[1] = c_frag(tpn + L::o_wfT, 1, tB_, 0, lane);
#pragma unroll
          for (int it = 0; it < NI; ++it) {
            pwinB[0][it] = c_frag(tpn + L::o_win, NI, tA, it, lane);
            pwinB[1][it] = c_frag(tpn + L::o_win, NI, tB_, it, lane);
          }
#pragma unroll
          for (int i = 0; i < NFS; ++i) pw2T[i] = c_frag(tpn + L::o_w2T, NT, bslot_tile(i), bslot_ot(i), lane);
        }
        {
          f32x4 du = c_zero();
#pragma unroll
          for (int pp = 0; pp < NP; ++pp) {
            const f32x4 pv = c_ld4(PBb + (pp * NQ + q) * 256 + lane * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) du[r] += pv[r];
          }
          G0 = Gd0 + (s0on ? du[0] : 0.f);
          G1 = Gd1 + (s1on ? du[1] : 0.f);
          if (a.dctx && p == 0 && valid) {  // context gradient of the rows of input tile 0 that hold features
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float* cbx = CBx + (4 * g4 + r) * 3;
              const int f = (int)cbx[0];
              const bool slot_row = (r == 0 && s0on) || (r == 1 && s1on);
              if (f >= 0 && !slot_row) a.dctx[row * a.C + f] += du[r] * cbx[2];
            }
          }
        }
        SF_TC(49 + 12 * tt);
        {
          // jobs 0..n1-1: block 1 (TD1 x TH0); n1..n1+n2-1: initial layer (TD0 x TIN)
          const int n1 = c.n_jobs, n2 = NT * NI, ntot = n1 + n2;
          auto mk = [&](int n) -> CJob {
            if (n < n1) {
              const int code = CBj[n];
              const int ot = code >> 2, it = code & 3;
              return CJob{TD1, TH0, gp + L::g_w1 + (ot * NT + it) * 256, it == 0 ? gp + L::g_b1 : nullptr, ot, it};
            }
            const int m = n - n1;
            const int ot = m / NI, it = m - ot * NI;
            return CJob{TD0, TIN, gp + L::g_win + (ot * NI + it) * 256, it == 0 ? gp + L::g_b0 : nullptr, ot, it};
          };
          for (int n = wave; n < ntot; n += 2 * NW) {
            const bool two = n + NW < ntot;
            const CJob A = mk(n), B = mk(two ? n + NW : n);
            c_dw_jobs<NQ>(A, B, two, accumulate, lane);
          }
        }
        SF_TC(50 + 12 * tt);
        c_barrier();
        SF_TC(51 + 12 * tt);
      }
    }
    SF_TC(120);
  }
}

#if SF_TRC_TU == 5
// sum of the workgroups' gradient partials in a fixed order: block = 64 parameters x 16 groups of partials (1024 threads);
// a thread has up to eight of its group's loads in flight (the kernel is a chain of L2 / HBM round trips over 33 MB at
// batch 16 384: 15 us with 64 partials per thread four at a time, a third of that with 16 per thread eight at a time)
__global__ __launch_bounds__(1024) void k_gather_c(const float* __restrict__ gpart, long stride, int nwg,
                                                    const int32_t* __restrict__ gdst, float* __restrict__ grad, long n) {
  __shared__ float part[16][64];
  const int px = threadIdx.x & 63, qy = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + px;
  const int g = i < n ? gdst[i] : -1;
  float v = 0.f;
  if (g >= 0) {
    const int per = (nwg + 15) / 16;
    const int lo = qy * per, hi = min(nwg, lo + per);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int w = lo;
    for (; w + 8 <= hi; w += 8) {
      float q[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) q[k] = gpart[(size_t)(w + k) * stride + g];
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] += q[k];
    }
    for (int k = 0; w < hi; ++w, ++k) a[k] += gpart[(size_t)w * stride + g];
    v = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  part[qy][px] = v;
  __syncthreads();
  if (qy == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][px];
    grad[i] = t;
  }
}

// The same sum walking the partials in THEIR order: thread j owns position j of a partial -- consecutive threads read
// consecutive floats of every partial (k_gather_c's parameter order scatters its 4-byte reads over the fragment layout:
// 15 us for 33 MB at batch 16 384) -- and writes the one or two parameters that position feeds; the tail of the grid zeroes
// the parameters without a position.  Order of the sum: partial 0, 1, 2 ... through eight interleaved accumulators.
// sq_part (optional): the block's share of |grad|^2 -> sq_part[blockIdx.x], so that the optimiser step that follows need not read
// the whole gradient again for the clipping norm (sf_launch_adam sums the shares in block order: deterministic).
__global__ __launch_bounds__(256) void k_gather_c2(const float* __restrict__ gpart, long stride, int nwg,
                                                    const int32_t* __restrict__ gsrc, const int32_t* __restrict__ gzero,
                                                    long n_zero, float* __restrict__ grad, float* __restrict__ sq_part) {
  // block = 64 consecutive positions x 4 groups of partials (every CU gets blocks: 33 k positions alone are 130 blocks)
  __shared__ float part[4][64];
  const int px = threadIdx.x & 63, qy = threadIdx.x >> 6;
  const long j = (long)blockIdx.x * 64 + px;
  if ((long)blockIdx.x * 64 >= stride) {  // the tail of the grid: parameters without a position
    const long z = ((long)blockIdx.x * 64 - (stride + 63) / 64 * 64) * 4 + threadIdx.x;
    if (z < n_zero) grad[gzero[z]] = 0.f;
    if (sq_part && threadIdx.x == 0) sq_part[blockIdx.x] = 0.f;
    return;
  }
  const int p0 = j < stride ? gsrc[2 * j] : -1, p1 = j < stride ? gsrc[2 * j + 1] : -1;
  float v = 0.f;
  if (p0 >= 0) {
    const int per = (nwg + 3) / 4;
    const int lo = qy * per, hi = min(nwg, lo + per);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int w = lo;
    for (; w + 8 <= hi; w += 8) {
      float q[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) q[k] = gpart[(size_t)(w + k) * stride + j];
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] += q[k];
    }
    for (int k = 0; w < hi; ++w, ++k) a[k] += gpart[(size_t)w * stride + j];
    v = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  part[qy][px] = v;
  __syncthreads();
  float sq = 0.f;
  if (qy == 0 && p0 >= 0) {
    const float t = (part[0][px] + part[1][px]) + (part[2][px] + part[3][px]);
    grad[p0] = t;
    if (p1 >= 0) grad[p1] = t;
    sq = p1 >= 0 ? 2.f * t * t : t * t;   // (a position that feeds two parameters counts twice in the norm)
  }
  if (sq_part && qy == 0) {   // wave 0 holds the block's 64 positions
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    if (px == 0) sq_part[blockIdx.x] = sq;
  }
}

// fixed-point form (sf_fixacc.h): position j of the SF_FIX_REPLICAS int64 images -> the one or two parameters it feeds;
// integer sum of the replicas, one conversion
__global__ __launch_bounds__(256) void k_gather_fix(const long long* __restrict__ gfix, long stride, int nrep,
                                                     const int32_t* __restrict__ gsrc, const int32_t* __restrict__ gzero,
                                                     long n_zero, float* __restrict__ grad) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < stride) {
    const int p0 = gsrc[2 * j], p1 = gsrc[2 * j + 1];
    if (p0 >= 0) {
      long long sacc = 0;
      for (int r = 0; r < nrep; ++r) sacc += gfix[(size_t)r * stride + j];
      const float v = (float)((double)sacc * (1.0 / SF_FIX_SCALE));
      grad[p0] = v;
      if (p1 >= 0) grad[p1] = v;
    }
  } else {
    const long z = j - stride;
    if (z < n_zero) grad[gzero[z]] = 0.f;
  }
}

size_t sf_trainc_lds_bytes(const SfTrcDev& c, int TS, int NG) {
  const size_t NQ = 2 * (size_t)NG;
  return ((size_t)c.NT * NQ * (256 * 2 + TTS * 5) + NQ * TTS + (size_t)c.NI * NQ * TTS + (size_t)TS * NQ * 256 +
          (size_t)sf_trc_cb_floats(c.NT, c.NI, TS)) * sizeof(float);
}

// groups of 4 waves (32 samples) per workgroup.  1 = 4-wave workgroups: a 32-sample chain is shorter (batch 64: 48 us
// against 63, batch 2 048: 55 against 66) -- taken while the batch leaves CUs idle anyway (<= 8 192 rows: one workgroup per
// CU).  2 = one 8-wave workgroup per CU: weight-gradient products over 64 samples and HALF the gradient partials.  At batch
// 16 384 the flow kernel alone is faster with 4-wave workgroups, two per CU (67.6 us against 73.4: two independent
// workgroups overlap better than two groups that meet at every barrier), but the 512 partials cost the gather kernel more
// than that (step 107 us against 97): the step decides.  SF_TRC_NG=1|2 overrides.
int sf_trainc_groups(long B, const SfTrcDev* c, int T) {
  // (a deep stash -- T = 7, 8 with four hidden tiles -- leaves LDS for one group of four waves only)
  if (c && sf_trainc_lds_bytes(*c, sf_trc_ts(T), 2) > (size_t)160 * 1024) return 1;
  static int forced = -1;
  if (forced < 0) { const char* e = std::getenv("SF_TRC_NG"); forced = e ? std::atoi(e) : 0; }
  if (forced == 1 || forced == 2) return forced;
  return B <= 8192 ? 1 : 2;
}

// fixed-point replicas instead of per-workgroup partials (sf_fixacc.h): only on request (SF_GRAD_ACC=fix).  Measured at batch
// 16 384 (cfg1): 100 us against 74 with the partials (145 KB each, plain stores) -- the L2s serve about one 64-bit atomic per
// channel and clock, a sixteenth of their store rate.
bool sf_trainc_fix(int grid, long n_gradC) {
  static int force = -1;
  if (force < 0) {
    const char* e = std::getenv("SF_GRAD_ACC");
    force = !e ? 0 : (e[0] == 'p' ? 1 : (e[0] == 'f' ? 2 : 0));
  }
  (void)grid; (void)n_gradC;
  return force == 2;
}

bool sf_trainc_eligible(const SfLayout& L, bool want_dctx) {
  const SfTrcDev& c = L.trc;
  static int env = -1;
  if (env < 0) { const char* e = std::getenv("SF_TRAINC"); env = e ? std::atoi(e) : 1; }
  if (!env || !c.ok) return false;
  if (L.dev.T > SF_TRC_TS_MAX || c.NI > 2 || c.NT < 1 || c.NT > 4) return false;
  if (want_dctx && c.NI > 1) return false;
  {
    bool ok = false;
    switch (c.NI * 10 + c.NT) {
      case 11: ok = CLay<1, 1>::matches(c, L.dev.T); break; case 12: ok = CLay<2, 1>::matches(c, L.dev.T); break;
      case 13: ok = CLay<3, 1>::matches(c, L.dev.T); break; case 14: ok = CLay<4, 1>::matches(c, L.dev.T); break;
      case 21: ok = CLay<1, 2>::matches(c, L.dev.T); break; case 22: ok = CLay<2, 2>::matches(c, L.dev.T); break;
      case 23: ok = CLay<3, 2>::matches(c, L.dev.T); break; case 24: ok = CLay<4, 2>::matches(c, L.dev.T); break;
    }
    if (!ok) return false;  // (the packer and the kernel disagree about the block order: never launch on that)
  }
  // a wave holds the fragments of tiles p and NT-1-p in five slots per layer (aligned MADE placement: NT + 1 blocks)
  for (int p = 0; 2 * p < c.NT; ++p) {
    const int tA = p, tB = c.NT - 1 - p;
    const int nf = (c.kend[tA] + 1) + (tB > tA ? c.kend[tB] + 1 : 0);
    const int nb = (c.NT - c.kbeg[tA]) + (tB > tA ? c.NT - c.kbeg[tB] : 0);
    if (nf > 8 || nb > 8) return false;
  }
  return sf_trainc_lds_bytes(c, sf_trc_ts(L.dev.T), 1) <= (size_t)160 * 1024;
}

int sf_trainc_grid(long B, const SfTrcDev* c, int T) {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t pr;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
              ? pr.multiProcessorCount : 256;
  }
  const int ng = sf_trainc_groups(B, c, T);
  const long per = 32L * ng, chunks = (B + per - 1) / per;
  const long cap = (long)cus * (ng == 1 ? 2 : 1);
  return (int)(chunks < cap ? chunks : cap);
}

#endif  // SF_TRC_TU == 5

template <int NI, int NT, int NG, int TS, int NFS>
static hipError_t c_launch_ts(const SfTrcArgs& a, int grid, hipStream_t st) {
  static SfAttrCache attr;
  int attr_dev;
  if (attr.need(attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)k_maf_trainc<TS, NI, NT, NG, NFS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr.set(attr_dev);
  }
  const size_t sh = sf_trainc_lds_bytes(a.c, TS, NG);
#ifdef SF_TRC_TRACE
  {
    static unsigned long long* d_tr = nullptr;
    if (!d_tr && hipMalloc(&d_tr, 8 * 256 * 8) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipMemsetAsync(d_tr, 0, 8 * 256 * 8, st);
    SfTrcArgs b = a;
    b.trace = d_tr;
    hipLaunchKernelGGL((k_maf_trainc<TS, NI, NT, NG, NFS>), dim3((unsigned)grid), dim3(256 * NG), sh, st, b);
    (void)hipStreamSynchronize(st);
    static unsigned long long h[8 * 256];
    (void)hipMemcpy(h, d_tr, sizeof(h), hipMemcpyDeviceToHost);
    static int calls = 0;
    if (++calls % 8 == 0) {
      fprintf(stderr, "[trainc trace] B=%ld grid=%d (units of 100 cycles since stamp 0 of wave 0)\n", a.B, grid);
      for (int w = 0; w < 4 * NG; ++w) {
        fprintf(stderr, "  wave %d:", w);
        for (int i = 0; i < 256; ++i)
          if (h[w * 256 + i]) fprintf(stderr, " %d:%.1f", i, (double)(long long)(h[w * 256 + i] - h[0]) * 0.01);
        fprintf(stderr, "\n");
      }
    }
    return hipGetLastError();
  }
#endif
  hipLaunchKernelGGL((k_maf_trainc<TS, NI, NT, NG, NFS>), dim3((unsigned)grid), dim3(256 * NG), sh, st, a);
  return hipGetLastError();
}
template <int NG>
static hipError_t c_dispatch(const SfTrcArgs& a, int grid, hipStream_t st) {
  const int key = a.c.NI * 10 + a.c.NT;
  constexpr int TS_ = SF_TRC_TS_OF_TU, NFS_ = SF_TRC_NFS_OF_TU;
  switch (key) {
#if SF_TRC_TU < 10   // (one or two hidden tiles never need more than five slots)
    case 11: return c_launch_ts<1, 1, NG, TS_, NFS_>(a, grid, st);
    case 12: return c_launch_ts<1, 2, NG, TS_, NFS_>(a, grid, st);
    case 21: return c_launch_ts<2, 1, NG, TS_, NFS_>(a, grid, st);
    case 22: return c_launch_ts<2, 2, NG, TS_, NFS_>(a, grid, st);
#endif
    case 13: return c_launch_ts<1, 3, NG, TS_, NFS_>(a, grid, st);
    case 14: return c_launch_ts<1, 4, NG, TS_, NFS_>(a, grid, st);
    case 23: return c_launch_ts<2, 3, NG, TS_, NFS_>(a, grid, st);
    case 24: return c_launch_ts<2, 4, NG, TS_, NFS_>(a, grid, st);
  }
  return hipErrorInvalidValue;
}
#define SF_TRC_CAT2(a, b) a##b
#define SF_TRC_CAT(a, b) SF_TRC_CAT2(a, b)
// this unit's launcher: sf_trainc_launch_ts5 / _ts6 / _ts8 (five fragment slots), _ts15 / _ts16 / _ts18 (eight)
hipError_t SF_TRC_CAT(sf_trainc_launch_ts, SF_TRC_TU)(const SfTrcArgs& a, int grid, int ng, hipStream_t st) {
  return ng == 1 ? c_dispatch<1>(a, grid, st) : c_dispatch<2>(a, grid, st);
}

#if SF_TRC_TU == 5
hipError_t sf_trainc_launch_ts6(const SfTrcArgs& a, int grid, int ng, hipStream_t st);
hipError_t sf_trainc_launch_ts8(const SfTrcArgs& a, int grid, int ng, hipStream_t st);
hipError_t sf_trainc_launch_ts15(const SfTrcArgs& a, int grid, int ng, hipStream_t st);
hipError_t sf_trainc_launch_ts16(const SfTrcArgs& a, int grid, int ng, hipStream_t st);
hipError_t sf_trainc_launch_ts18(const SfTrcArgs& a, int grid, int ng, hipStream_t st);
// fragment slots a flow's placement needs per masked layer and wave (sf_trainc_eligible admits up to eight)
static int c_slots_needed(const SfTrcDev& c) {
  int mx = 0;
  for (int p = 0; 2 * p < c.NT; ++p) {
    const int tA = p, tB = c.NT - 1 - p;
    const int nf = (c.kend[tA] + 1) + (tB > tA ? c.kend[tB] + 1 : 0);
    const int nb = (c.NT - c.kbeg[tA]) + (tB > tA ? c.NT - c.kbeg[tB] : 0);
    mx = nf > mx ? nf : mx;
    mx = nb > mx ? nb : mx;
  }
  return mx;
}
// T <= 5: every transform's a1 / a2 tiles in registers (80 VGPRs of stash); T = 6 and T = 7..8 are instantiations of their own --
// the reference's example CLI trains num_transforms = 6 (examples/sbi/scripts/train_model.py:56-57) -- whose longer stash the
// compiler parks partly in scratch: same cooperative decomposition (no activation in HBM), measured 0.173 of the fp32 roof at
// T = 6 and batch 16 384 (86.9 us; the generic kernel: 187.5 us)
hipError_t sf_launch_maf_trainc(const SfTrcArgs& a, int grid, hipStream_t st) {
  const int ng = sf_trainc_groups(a.B, &a.c, a.T);
  if (c_slots_needed(a.c) > 5) {   // contiguous ("span") placement: the eight-slot instantiations
    switch (sf_trc_ts(a.T)) {
      case 5: return sf_trainc_launch_ts15(a, grid, ng, st);
      case 6: return sf_trainc_launch_ts16(a, grid, ng, st);
      default: return sf_trainc_launch_ts18(a, grid, ng, st);
    }
  }
  switch (sf_trc_ts(a.T)) {
    case 5: return sf_trainc_launch_ts5(a, grid, ng, st);
    case 6: return sf_trainc_launch_ts6(a, grid, ng, st);
    default: return sf_trainc_launch_ts8(a, grid, ng, st);
  }
}

long sf_gather_c2_blocks(long stride, long n_zero) { return (stride + 63) / 64 + (n_zero + 255) / 256; }
hipError_t sf_launch_gather_c2(const float* gpart, long stride, int nwg, const int32_t* gsrc, const int32_t* gzero, long n_zero,
                               float* grad, hipStream_t st, float* sq_part) {
  const long blocks = sf_gather_c2_blocks(stride, n_zero);  // 64 positions per block, then 256 unmapped parameters per block
  hipLaunchKernelGGL(k_gather_c2, dim3((unsigned)blocks), dim3(256), 0, st, gpart, stride, nwg, gsrc, gzero, n_zero, grad, sq_part);
  return hipGetLastError();
}
hipError_t sf_launch_gather_fix(const long long* gfix, long stride, int nrep, const int32_t* gsrc, const int32_t* gzero, long n_zero,
                                float* grad, hipStream_t st) {
  const long tot = stride + n_zero;
  hipLaunchKernelGGL(k_gather_fix, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, gfix, stride, nrep, gsrc, gzero, n_zero, grad);
  return hipGetLastError();
}
hipError_t sf_launch_gather_c(const float* gpart, long stride, int nwg, const int32_t* gdst, float* grad, long n, hipStream_t st) {
  hipLaunchKernelGGL(k_gather_c, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, st, gpart, stride, nwg, gdst, grad, n);
  return hipGetLastError();
}
#endif  // SF_TRC_TU == 5
