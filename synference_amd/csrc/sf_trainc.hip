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
#include "sf_internal.h"
#include "sf_trainc.h"

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
__device__ __forceinline__ float4 c_frag(const float* wp, int nb, int a, int b, int lane) {
  return reinterpret_cast<const float4*>(wp)[(a * nb + b) * 64 + lane];
}
__device__ __forceinline__ f32x4 c_zero() {
  f32x4 z;
  z[0] = 0.f; z[1] = 0.f; z[2] = 0.f; z[3] = 0.f;
  return z;
}
// LDS tile addressing: a tile (16 rows x 16 samples of subtile q) is 256 floats
//   B layout: lane l holds float4 at l*4 (rows 4*(l>>4)+r of sample l&15): the MFMA B operand of a data product
//   T layout: [sample >> 2][row][sample & 3]: lane l = 16*kk + i reads float4 at l*4 = row i, samples 4kk..4kk+3:
//             A / B operand of a weight-gradient product (k = sample)
__device__ __forceinline__ void c_put_T(float* tile, const f32x4 v, int s, int g4) {
  float* p = tile + (s >> 2) * 64 + (s & 3) + 16 * g4;
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
__device__ __forceinline__ void c_dw_finish(const CJob& J, f32x4 acc, float bs, bool accumulate, int lane) {
  if (accumulate) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += J.gw[r * 64 + lane];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) J.gw[r * 64 + lane] = acc[r];
  if (J.gb) {
    bs += __shfl_xor(bs, 16, 64);
    bs += __shfl_xor(bs, 32, 64);
    if (lane < 16) J.gb[J.ot * 16 + lane] = accumulate ? J.gb[J.ot * 16 + lane] + bs : bs;
  }
}
__device__ __forceinline__ void c_dw_jobs(const CJob& A, const CJob& B, bool two, bool accumulate, int lane) {
  f32x4 accA = c_zero(), accB = c_zero();
  float bsA = 0.f, bsB = 0.f;
  if (two) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 dA = *reinterpret_cast<const float4*>(A.Td + (A.ot * 4 + q) * 256 + lane * 4);
      const float4 iA = *reinterpret_cast<const float4*>(A.Ti + (A.it * 4 + q) * 256 + lane * 4);
      const float4 dB = *reinterpret_cast<const float4*>(B.Td + (B.ot * 4 + q) * 256 + lane * 4);
      const float4 iB = *reinterpret_cast<const float4*>(B.Ti + (B.it * 4 + q) * 256 + lane * 4);
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
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 dA = *reinterpret_cast<const float4*>(A.Td + (A.ot * 4 + q) * 256 + lane * 4);
      const float4 iA = *reinterpret_cast<const float4*>(A.Ti + (A.it * 4 + q) * 256 + lane * 4);
      accA = SF_MFMA16(dA.x, iA.x, accA);
      accA = SF_MFMA16(dA.y, iA.y, accA);
      accA = SF_MFMA16(dA.z, iA.z, accA);
      accA = SF_MFMA16(dA.w, iA.w, accA);
      bsA += (dA.x + dA.y) + (dA.z + dA.w);
    }
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
}

// floats of the constant block kept in LDS for the life of the workgroup
__host__ __device__ inline int sf_trc_nbias(int NT) { return 3 * NT * 16 + 16; }
__host__ __device__ inline int sf_trc_cb_floats(int NT, int NI, int TS) { return TS * sf_trc_nbias(NT) + 16 + NI * 16 * 3 + 8 * 3; }

template <int TS, int NI, int NT>
__global__ __launch_bounds__(512, 2) void k_maf_trainc(SfTrcArgs a_in) {
  extern __shared__ float lds[];
  const SfTrcArgs& a = c_args();
  const SfTrcDev& c = a.c;
  // (the wave index is made visibly wave-uniform: tile indices, buffer bases and fragment bases then live in SGPRs and the
  //  per-lane address part is one register)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2, j = wave & 3;
  const int s = lane & 15, g4 = lane >> 4;
  const int D = a.D;
  const int mt_raw = grp == 0 ? j : NT - 1 - j;
  const bool has = NT == 4 || (mt_raw >= 0 && mt_raw < NT);  // this wave owns hidden tile mt (always, with four tiles)
  const int mt = has ? mt_raw : 0;
  int ke = 0, kb = 0;  // tile bounds of my tile (select chains: the descriptor stays in SGPRs)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    ke = (q == mt) ? c.kend[q] : ke;
    kb = (q == mt) ? c.kbeg[q] : kb;
  }
  // ---- LDS map (floats)
  constexpr int HSZ = NT * 1024;         // one hidden-size tensor: NT tiles x 4 subtiles x 256
  float* XBa = lds;                      // B layout: h0 (fwd) / dpre2 (bwd); backward partial sums of du
  float* XBb = XBa + HSZ;                // B layout: a1 (fwd) / dpre1 (bwd)
  float* TB0 = XBb + HSZ;                // T layout x5: TD2, TA2 (later TD0), TH0, TD1, TA1; forward: head partial sums
  float* TD2 = TB0, *TA2 = TB0 + HSZ, *TH0 = TB0 + 2 * HSZ, *TD1 = TB0 + 3 * HSZ, *TA1 = TB0 + 4 * HSZ;
  float* TD0 = TA2;
  float* PBf = TB0;
  float* PBb = XBa;
  float* TDF = TB0 + 5 * HSZ;            // T layout: head delta, 1 tile
  float* TIN = TDF + 1024;               // T layout: input tiles, NI tiles
  float* USt = TIN + NI * 1024;          // [TS][64 samples][8]: u entering transform t
  float* ASt = USt + TS * 512;           // [TS][64][8]: head output a (slots)
  // constants of the whole call: biases of every transform, the weight-gradient block list, per input-row and
  // per-slot standardisation constants (read from LDS in the phases instead of through dependent global loads)
  constexpr int NBIAS = 3 * NT * 16 + 16;
  float* CBb = ASt + TS * 512;           // [T][b0 NT*16 | b1 | b2 | bf 16]
  int* CBj = reinterpret_cast<int*>(CBb + TS * NBIAS);  // [16] (ot * 4 + it)
  float* CBx = reinterpret_cast<float*>(CBj + 16);      // [NI*16][feature, mean, 1/std]
  float* CBs = CBx + NI * 16 * 3;                       // [8 slots][theta column, scale, shift]
  {
    const float* cst = a.cst;
    for (int i = threadIdx.x; i < a.T * NBIAS; i += 512) {
      const int t = i / NBIAS, k = i - t * NBIAS;
      const int sec = k / (NT * 16), kk = k - sec * NT * 16;
      const int off = sec == 0 ? c.o_b0 : (sec == 1 ? c.o_b1 : (sec == 2 ? c.o_b2 : c.o_bf));
      CBb[i] = a.img[(size_t)t * c.t_stride + off + kk];
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
      const int p = threadIdx.x < D ? threadIdx.x : 0;
      CBs[threadIdx.x * 3] = cst[a.c_tdim + p];
      CBs[threadIdx.x * 3 + 1] = cst[a.c_pscale + p];
      CBs[threadIdx.x * 3 + 2] = cst[a.c_pshift + p];
    }
  }
  __syncthreads();
  // my two subtiles
  const int q0 = 2 * grp;
  const bool s0on = 2 * g4 < D, s1on = 2 * g4 + 1 < D;

  for (long chunk = blockIdx.x, iter = 0; chunk < a.n_chunks; chunk += gridDim.x, ++iter) {
    const bool accumulate = iter > 0;
    SF_TC(0);
    // first fragments of the forward sweep: in flight behind the input loads
    float4 pwin[NI];
    {
      const float* tp0 = a.img + sf_opaque_zero();
#pragma unroll
      for (int it = 0; it < NI; ++it) pwin[it] = c_frag(tp0 + c.o_win, NI, mt, it, lane);
    }
    // ------------------------------------------------------------------ per-sample inputs
    bool valid[2];
    float wgt[2];
    f32x4 inx[NI][2];  // [input tile][subtile]: context part of the input tiles (slot rows 0 here)
    float u0[2], u1[2];
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const long row = chunk * 64 + (q0 + qq) * 16 + s;
      valid[qq] = row < a.B;
      const long ii = valid[qq] ? row : a.B - 1;
      const long src = a.idx ? (long)a.idx[ii] : ii;
      wgt[qq] = valid[qq] ? (a.wts ? a.w * a.wts[row] : a.w) : 0.f;
      const float* xr = a.x + src * a.C;
#pragma unroll
      for (int it = 0; it < NI; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* cb = CBx + (it * 16 + 4 * g4 + r) * 3;
          const int f = (int)cb[0];
          const float v = (xr[f >= 0 ? f : 0] - cb[1]) * cb[2];
          inx[it][qq][r] = f >= 0 ? v : 0.f;
        }
      }
      const float* sb = CBs + (2 * g4) * 3;
      // (unconditional loads, selected afterwards: a load under a per-lane condition gets its own wait)
      const float th0 = a.theta[src * D + (int)sb[0]], th1 = a.theta[src * D + (int)sb[3]];
      u0[qq] = s0on ? th0 * sb[1] + sb[2] : 0.f;
      u1[qq] = s1on ? th1 * sb[4] + sb[5] : 0.f;
    }
    float ld[2] = {0.f, 0.f};
    f32x4 a1s[TS][2], a2s[TS][2];  // the wave's own tile of a1 / a2 for every transform

    // ------------------------------------------------------------------ forward
#pragma unroll
    for (int t = 0; t < TS; ++t) {
      if (t < a.T) {
        const SfTrcArgs& a = c_args();
        const SfTrcDev& c = a.c;
        const float* tp = a.img + (size_t)t * c.t_stride + sf_opaque_zero();  // (not loop-invariant: see sf_device.h)
        const float* cb = CBb + t * NBIAS;
        // F1: h0 = b0 + bc + Win . [u ; e(x)]
        SF_TC(1 + 5 * t);
        float4 w1f[4], w2f[4];
        if (has) {
          // next phase's fragments: in flight across the barrier
#pragma unroll
          for (int it = 0; it < 4; ++it) w1f[it] = c_frag(tp + c.o_w1, NT, mt, it <= ke ? it : ke, lane);
          f32x4 h0[2];
          const f32x4 bv = c_ld4(cb + (mt * 4 + g4) * 4);
          h0[0] = bv; h0[1] = bv;
#pragma unroll
          for (int it = 0; it < NI; ++it) {
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
              f32x4 in = inx[it][qq];
              if (it == 0) {
                in[0] = s0on ? u0[qq] : in[0];
                in[1] = s1on ? u1[qq] : in[1];
              }
              h0[qq] = c_mma(pwin[it], in, h0[qq]);
            }
          }
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) c_st4(XBa + (mt * 4 + q0 + qq) * 256 + lane * 4, h0[qq]);
        }
        SF_TC(2 + 5 * t);
        c_barrier();
        // F2: a1 = tanh(b1 + W1 h0)
        float4 wff = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has) {
#pragma unroll
          for (int it = 0; it < 4; ++it) w2f[it] = c_frag(tp + c.o_w2, NT, mt, it <= ke ? it : ke, lane);
          wff = c_frag(tp + c.o_wf, NT, 0, mt, lane);
          const f32x4 b1v = c_ld4(cb + NT * 16 + (mt * 4 + g4) * 4);
          f32x4 acc[2] = {b1v, b1v};
#pragma unroll
          for (int it = 0; it < 4; ++it)
            if (it <= ke) {
#pragma unroll
              for (int qq = 0; qq < 2; ++qq)
                acc[qq] = c_mma(w1f[it], c_ld4(XBa + (it * 4 + q0 + qq) * 256 + lane * 4), acc[qq]);
            }
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[qq][r] = sf_tanh(acc[qq][r]);
            a1s[t][qq] = acc[qq];
            c_st4(XBb + (mt * 4 + q0 + qq) * 256 + lane * 4, acc[qq]);
          }
        }
        SF_TC(3 + 5 * t);
        c_barrier();
        // F3: a2 = tanh(b2 + W2 a1); head partial sums over my 16 hidden rows
        if (has) {
          {  // the next transform's first fragments (the last transform reloads its own: no branch around a load)
            const float* tpn = tp + (t + 1 < a.T ? c.t_stride : 0);
#pragma unroll
            for (int it = 0; it < NI; ++it) pwin[it] = c_frag(tpn + c.o_win, NI, mt, it, lane);
          }
          const f32x4 b2v = c_ld4(cb + 2 * NT * 16 + (mt * 4 + g4) * 4);
          f32x4 acc[2] = {b2v, b2v};
#pragma unroll
          for (int it = 0; it < 4; ++it)
            if (it <= ke) {
#pragma unroll
              for (int qq = 0; qq < 2; ++qq)
                acc[qq] = c_mma(w2f[it], c_ld4(XBb + (it * 4 + q0 + qq) * 256 + lane * 4), acc[qq]);
            }
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[qq][r] = sf_tanh(acc[qq][r]);
            a2s[t][qq] = acc[qq];
            c_st4(PBf + (mt * 4 + q0 + qq) * 256 + lane * 4, c_mma(wff, acc[qq], c_zero()));
          }
        }
        SF_TC(4 + 5 * t);
        c_barrier();
        SF_TC(5 + 5 * t);
        // F4 (every wave, replicated): head = bf + sum of the partial sums; affine update of the two slots of this lane
        {
          const f32x4 bfv = c_ld4(cb + 3 * NT * 16 + g4 * 4);
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            f32x4 fin = bfv;
#pragma unroll
            for (int it = 0; it < 4; ++it)
              if (it < NT) {
                const f32x4 pv = c_ld4(PBf + (it * 4 + q0 + qq) * 256 + lane * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) fin[r] += pv[r];
              }
            if (j == 0) {  // one wave per group keeps what the backward sweep needs
              const int sidx = ((t * 64) + (q0 + qq) * 16 + s) * 8 + 2 * g4;
              *reinterpret_cast<float2*>(USt + sidx) = make_float2(u0[qq], u1[qq]);
              *reinterpret_cast<float2*>(ASt + sidx) = make_float2(fin[0], fin[1]);
            }
            const float sc0 = c_scale(a.scale_fn, fin[0], a.eps), sc1 = c_scale(a.scale_fn, fin[1], a.eps);
            u0[qq] = s0on ? sc0 * u0[qq] + fin[2] : 0.f;
            u1[qq] = s1on ? sc1 * u1[qq] + fin[3] : 0.f;
            ld[qq] += (s0on ? sf_log(sc0) : 0.f) + (s1on ? sf_log(sc1) : 0.f);
          }
        }
      }
    }
    // first fragments of the backward sweep (top transform): in flight behind the loss
    float4 pwfT, pw2T[4], pwinB[NI];
    {
      const SfTrcArgs& a = c_args();
      const SfTrcDev& c = a.c;
      const float* tpl = a.img + (size_t)(a.T - 1) * c.t_stride + sf_opaque_zero();
      pwfT = c_frag(tpl + c.o_wfT, 1, mt, 0, lane);
#pragma unroll
      for (int it = 0; it < NI; ++it) pwinB[it] = c_frag(tpl + c.o_win, NI, mt, it, lane);
#pragma unroll
      for (int ot = 0; ot < 4; ++ot) pw2T[ot] = c_frag(tpl + c.o_w2T, NT, mt, (ot >= kb && ot < NT) ? ot : kb, lane);
    }
    // ------------------------------------------------------------------ loss, dL/du_T
    float G0[2], G1[2];
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      float ss = u0[qq] * u0[qq] + u1[qq] * u1[qq];
      float lds_ = ld[qq];
      ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);
      lds_ += __shfl_xor(lds_, 16, 64); lds_ += __shfl_xor(lds_, 32, 64);
      const float nll = 0.5f * ss + 0.5f * (float)D * 1.8378770664093453f - (a.logdet0 + lds_);
      if (j == 0) {
        if (a.loss && valid[qq] && g4 == 0) a.loss[chunk * 64 + (q0 + qq) * 16 + s] = nll;
        if (a.loss_sum) {
          float tsum = (valid[qq] && g4 == 0) ? nll : 0.f;
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) tsum += __shfl_xor(tsum, o, 64);
          // values on a 2^-20 grid add exactly in double: the sum does not depend on the order of the atomics
          if (lane == 0) atomicAdd(a.loss_sum, (double)rintf(tsum * 1048576.0f) * (1.0 / 1048576.0));
        }
      }
      G0[qq] = wgt[qq] * u0[qq];
      G1[qq] = wgt[qq] * u1[qq];
    }
    SF_TC(39);
    c_barrier();  // the last head sums have been read (PBf is TD2's buffer), the u / a stash is complete

    // ------------------------------------------------------------------ backward
    float* gpart = a.gpart + (size_t)blockIdx.x * a.gpart_stride;
#pragma unroll
    for (int tt = 0; tt < TS; ++tt) {
      const int t = TS - 1 - tt;
      if (t < a.T) {
        const SfTrcArgs& a = c_args();
        const SfTrcDev& c = a.c;
        const float* tp = a.img + (size_t)t * c.t_stride + sf_opaque_zero();
        const float* cb = CBb + t * NBIAS;
        float* gp = gpart + (size_t)t * c.g_stride;
        // B1: head backward, delta of block 2, h0 recomputed
        SF_TC(40 + 12 * tt);
        float Gd0[2], Gd1[2];
        {
          const f32x4 b0v = c_ld4(cb + (mt * 4 + g4) * 4);
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            const int sidx = ((t * 64) + (q0 + qq) * 16 + s) * 8 + 2 * g4;
            const float2 uu = *reinterpret_cast<const float2*>(USt + sidx);
            const float2 aa = *reinterpret_cast<const float2*>(ASt + sidx);
            const float sc0 = c_scale(a.scale_fn, aa.x, a.eps), sc1 = c_scale(a.scale_fn, aa.y, a.eps);
            f32x4 dfin;
            dfin[0] = s0on ? (G0[qq] * uu.x - sf_div(wgt[qq], sc0)) * c_dscale(a.scale_fn, aa.x) : 0.f;
            dfin[1] = s1on ? (G1[qq] * uu.y - sf_div(wgt[qq], sc1)) * c_dscale(a.scale_fn, aa.y) : 0.f;
            dfin[2] = s0on ? G0[qq] : 0.f;
            dfin[3] = s1on ? G1[qq] : 0.f;
            Gd0[qq] = s0on ? G0[qq] * sc0 : 0.f;
            Gd1[qq] = s1on ? G1[qq] * sc1 : 0.f;
            if (j == 0) c_put_T(TDF + (q0 + qq) * 256, dfin, s, g4);
            // input tiles of this transform (slot rows from the stash)
            f32x4 in0 = inx[0][qq];
            in0[0] = s0on ? uu.x : in0[0];
            in0[1] = s1on ? uu.y : in0[1];
            if (j == 0) {
              c_put_T(TIN + (q0 + qq) * 256, in0, s, g4);
              if constexpr (NI > 1) c_put_T(TIN + (4 + q0 + qq) * 256, inx[NI - 1][qq], s, g4);
            }
            if (has) {
              f32x4 dp2 = c_mma(pwfT, dfin, c_zero());
#pragma unroll
              for (int r = 0; r < 4; ++r) dp2[r] *= 1.0f - a2s[t][qq][r] * a2s[t][qq][r];
              c_st4(XBa + (mt * 4 + q0 + qq) * 256 + lane * 4, dp2);
              c_put_T(TD2 + (mt * 4 + q0 + qq) * 256, dp2, s, g4);
              c_put_T(TA2 + (mt * 4 + q0 + qq) * 256, a2s[t][qq], s, g4);
              f32x4 h0 = c_mma(pwinB[0], in0, b0v);
              if constexpr (NI > 1) h0 = c_mma(pwinB[NI - 1], inx[NI - 1][qq], h0);
              c_put_T(TH0 + (mt * 4 + q0 + qq) * 256, h0, s, g4);
            }
          }
        }
        SF_TC(41 + 12 * tt);
        c_barrier();
        SF_TC(42 + 12 * tt);
        // B2: delta of block 1; weight gradients of the head
        float4 w1T[4], wiT = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has) {
#pragma unroll
          for (int ot = 0; ot < 4; ++ot) w1T[ot] = c_frag(tp + c.o_w1T, NT, mt, (ot >= kb && ot < NT) ? ot : kb, lane);
          wiT = c_frag(tp + c.o_winT, NT, 0, mt, lane);
          f32x4 acc[2] = {c_zero(), c_zero()};
#pragma unroll
          for (int ot = 0; ot < 4; ++ot)
            if (ot >= kb && ot < NT) {
#pragma unroll
              for (int qq = 0; qq < 2; ++qq)
                acc[qq] = c_mma(pw2T[ot], c_ld4(XBa + (ot * 4 + q0 + qq) * 256 + lane * 4), acc[qq]);
            }
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[qq][r] *= 1.0f - a1s[t][qq][r] * a1s[t][qq][r];
            c_st4(XBb + (mt * 4 + q0 + qq) * 256 + lane * 4, acc[qq]);
            c_put_T(TD1 + (mt * 4 + q0 + qq) * 256, acc[qq], s, g4);
            c_put_T(TA1 + (mt * 4 + q0 + qq) * 256, a1s[t][qq], s, g4);
          }
        }
        SF_TC(43 + 12 * tt);
        if (wave < NT) {  // dWf[head tile][hidden tile]; bias with the first block
          const CJob J = {TDF, TA2, gp + c.g_wf + wave * 256, wave == 0 ? gp + c.g_bf : nullptr, 0, wave};
          c_dw_jobs(J, J, false, accumulate, lane);
        }
        SF_TC(44 + 12 * tt);
        c_barrier();
        SF_TC(45 + 12 * tt);
        // B3: delta of the initial layer, partial sums of W_in^T delta; weight gradients of block 2
        if (has) {
          f32x4 acc[2] = {c_zero(), c_zero()};
#pragma unroll
          for (int ot = 0; ot < 4; ++ot)
            if (ot >= kb && ot < NT) {
#pragma unroll
              for (int qq = 0; qq < 2; ++qq)
                acc[qq] = c_mma(w1T[ot], c_ld4(XBb + (ot * 4 + q0 + qq) * 256 + lane * 4), acc[qq]);
            }
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            c_put_T(TD0 + (mt * 4 + q0 + qq) * 256, acc[qq], s, g4);
            c_st4(PBb + (mt * 4 + q0 + qq) * 256 + lane * 4, c_mma(wiT, acc[qq], c_zero()));
          }
        }
        SF_TC(46 + 12 * tt);
        {
          const int nj = c.n_jobs;
          if (wave < nj) {
            const int cA = CBj[wave], cB = CBj[wave + 8 < nj ? wave + 8 : wave];
            const CJob A = {TD2, TA1, gp + c.g_w2 + ((cA >> 2) * NT + (cA & 3)) * 256, (cA & 3) == 0 ? gp + c.g_b2 : nullptr, cA >> 2, cA & 3};
            const CJob B = {TD2, TA1, gp + c.g_w2 + ((cB >> 2) * NT + (cB & 3)) * 256, (cB & 3) == 0 ? gp + c.g_b2 : nullptr, cB >> 2, cB & 3};
            c_dw_jobs(A, B, wave + 8 < nj, accumulate, lane);
          }
        }
        SF_TC(47 + 12 * tt);
        c_barrier();
        SF_TC(48 + 12 * tt);
        // B4 (every wave, replicated): dL/du of the transform below; weight gradients of block 1 and the initial layer
        {  // first fragments of the transform below (the bottom transform reloads its own)
          const float* tpn = tp - (t >= 1 ? c.t_stride : 0);
          pwfT = c_frag(tpn + c.o_wfT, 1, mt, 0, lane);
#pragma unroll
          for (int it = 0; it < NI; ++it) pwinB[it] = c_frag(tpn + c.o_win, NI, mt, it, lane);
#pragma unroll
          for (int ot = 0; ot < 4; ++ot) pw2T[ot] = c_frag(tpn + c.o_w2T, NT, mt, (ot >= kb && ot < NT) ? ot : kb, lane);
        }
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          f32x4 du = c_zero();
#pragma unroll
          for (int it = 0; it < 4; ++it)
            if (it < NT) {
              const f32x4 pv = c_ld4(PBb + (it * 4 + q0 + qq) * 256 + lane * 4);
#pragma unroll
              for (int r = 0; r < 4; ++r) du[r] += pv[r];
            }
          G0[qq] = Gd0[qq] + (s0on ? du[0] : 0.f);
          G1[qq] = Gd1[qq] + (s1on ? du[1] : 0.f);
          if (a.dctx && j == 0 && valid[qq]) {  // context gradient of the rows of input tile 0 that hold features
            const long brow = chunk * 64 + (q0 + qq) * 16 + s;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float* cbx = CBx + (4 * g4 + r) * 3;
              const int f = (int)cbx[0];
              const bool slot_row = (r == 0 && s0on) || (r == 1 && s1on);
              if (f >= 0 && !slot_row) a.dctx[brow * a.C + f] += du[r] * cbx[2];
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
              return CJob{TD1, TH0, gp + c.g_w1 + (ot * NT + it) * 256, it == 0 ? gp + c.g_b1 : nullptr, ot, it};
            }
            const int m = n - n1;
            const int ot = m / NI, it = m - ot * NI;
            return CJob{TD0, TIN, gp + c.g_win + (ot * NI + it) * 256, it == 0 ? gp + c.g_b0 : nullptr, ot, it};
          };
          for (int n = wave; n < ntot; n += 16) {
            const bool two = n + 8 < ntot;
            const CJob A = mk(n), B = mk(two ? n + 8 : n);
            c_dw_jobs(A, B, two, accumulate, lane);
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

// sum of the workgroups' gradient partials, in workgroup order: block = 64 parameters x 4 quarters of the partials
__global__ __launch_bounds__(256) void k_gather_c(const float* __restrict__ gpart, long stride, int nwg,
                                                   const int32_t* __restrict__ gdst, float* __restrict__ grad, long n) {
  __shared__ float part[4][64];
  const int px = threadIdx.x & 63, qy = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + px;
  const int g = i < n ? gdst[i] : -1;
  float v = 0.f;
  if (g >= 0) {
    const int per = (nwg + 3) / 4;
    const int lo = qy * per, hi = min(nwg, lo + per);
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    int w = lo;
    for (; w + 4 <= hi; w += 4) {
      v0 += gpart[(size_t)w * stride + g];
      v1 += gpart[(size_t)(w + 1) * stride + g];
      v2 += gpart[(size_t)(w + 2) * stride + g];
      v3 += gpart[(size_t)(w + 3) * stride + g];
    }
    for (; w < hi; ++w) v0 += gpart[(size_t)w * stride + g];
    v = (v0 + v1) + (v2 + v3);
  }
  part[qy][px] = v;
  c_barrier();
  if (qy == 0 && i < n) grad[i] = (part[0][px] + part[1][px]) + (part[2][px] + part[3][px]);
}

size_t sf_trainc_lds_bytes(const SfTrcDev& c, int TS) {
  return ((size_t)c.NT * 1024 * 7 + 1024 + (size_t)c.NI * 1024 + (size_t)TS * 1024 + (size_t)sf_trc_cb_floats(c.NT, c.NI, TS)) * sizeof(float);
}

bool sf_trainc_eligible(const SfLayout& L, bool want_dctx) {
  const SfTrcDev& c = L.trc;
  static int env = -1;
  if (env < 0) { const char* e = std::getenv("SF_TRAINC"); env = e ? std::atoi(e) : 1; }
  if (!env || !c.ok) return false;
  if (L.dev.T > SF_TRC_TS || c.NI > 2 || c.NT < 1 || c.NT > 4) return false;
  if (want_dctx && c.NI > 1) return false;
  return sf_trainc_lds_bytes(c, SF_TRC_TS) <= (size_t)160 * 1024;
}

int sf_trainc_grid(long B) {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t pr;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
              ? pr.multiProcessorCount : 256;
  }
  const long chunks = (B + 63) / 64;
  return (int)(chunks < cus ? chunks : cus);
}

template <int NI, int NT>
static hipError_t c_launch(const SfTrcArgs& a, int grid, hipStream_t st) {
  static SfAttrCache attr;
  int attr_dev;
  if (attr.need(attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)k_maf_trainc<SF_TRC_TS, NI, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr.set(attr_dev);
  }
  const size_t sh = sf_trainc_lds_bytes(a.c, SF_TRC_TS);
#ifdef SF_TRC_TRACE
  {
    static unsigned long long* d_tr = nullptr;
    if (!d_tr && hipMalloc(&d_tr, 8 * 256 * 8) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipMemsetAsync(d_tr, 0, 8 * 256 * 8, st);
    SfTrcArgs b = a;
    b.trace = d_tr;
    hipLaunchKernelGGL((k_maf_trainc<SF_TRC_TS, NI, NT>), dim3((unsigned)grid), dim3(512), sh, st, b);
    (void)hipStreamSynchronize(st);
    static unsigned long long h[8 * 256];
    (void)hipMemcpy(h, d_tr, sizeof(h), hipMemcpyDeviceToHost);
    static int calls = 0;
    if (++calls % 8 == 0) {
      fprintf(stderr, "[trainc trace] B=%ld grid=%d (units of 100 cycles since stamp 0 of wave 0)\n", a.B, grid);
      for (int w = 0; w < 8; ++w) {
        fprintf(stderr, "  wave %d:", w);
        for (int i = 0; i < 256; ++i)
          if (h[w * 256 + i]) fprintf(stderr, " %d:%.1f", i, (double)(long long)(h[w * 256 + i] - h[0]) * 0.01);
        fprintf(stderr, "\n");
      }
    }
    return hipGetLastError();
  }
#endif
  hipLaunchKernelGGL((k_maf_trainc<SF_TRC_TS, NI, NT>), dim3((unsigned)grid), dim3(512), sh, st, a);
  return hipGetLastError();
}

hipError_t sf_launch_maf_trainc(const SfTrcArgs& a, int grid, hipStream_t st) {
  const int key = a.c.NI * 10 + a.c.NT;
  switch (key) {
    case 11: return c_launch<1, 1>(a, grid, st);
    case 12: return c_launch<1, 2>(a, grid, st);
    case 13: return c_launch<1, 3>(a, grid, st);
    case 14: return c_launch<1, 4>(a, grid, st);
    case 21: return c_launch<2, 1>(a, grid, st);
    case 22: return c_launch<2, 2>(a, grid, st);
    case 23: return c_launch<2, 3>(a, grid, st);
    case 24: return c_launch<2, 4>(a, grid, st);
  }
  return hipErrorInvalidValue;
}

hipError_t sf_launch_gather_c(const float* gpart, long stride, int nwg, const int32_t* gdst, float* grad, long n, hipStream_t st) {
  hipLaunchKernelGGL(k_gather_c, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, gpart, stride, nwg, gdst, grad, n);
  return hipGetLastError();
}
