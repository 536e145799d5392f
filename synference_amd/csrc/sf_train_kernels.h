// sf_train_kernels.h -- forward+backward kernels of -log_prob (see the header comment in sf_train.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "sf_flows.h"
#include "sf_internal.h"
#include "sf_train_args.h"

// Transposed tile in LDS: 32 rows x 36 floats; sample s of a row sits at (s & 1) * 16 + (s >> 1), so that the 16 values an
// MFMA lane needs over the 16 k-steps (samples 2k + h) are contiguous: four ds_read_b128 instead of 16 ds_read_b32.
#define SF_TLR 36
#define SF_TL (32 * SF_TLR)


template <bool RELU = false>
__device__ __forceinline__ void sf_tile_to_lds(float* __restrict__ dst, const f32x16& t, int c, int h) {
#pragma unroll
  for (int r = 0; r < 16; ++r) dst[sf_row(r, h) * SF_TLR + (c & 1) * 16 + (c >> 1)] = RELU ? fmaxf(t[r], 0.f) : t[r];
}
// the 16 k-step operands of lane (c, h) from a transposed tile
__device__ __forceinline__ void sf_tile_row16(const float* __restrict__ tile, int c, int h, float (&v)[16]) {
  const float4* p = reinterpret_cast<const float4*>(tile + c * SF_TLR + h * 16);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 t = p[q];
    v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
  }
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

// ---------------------------------------------------------------------------------------------
// Weight gradients: one producer wave, NC consumer waves.
//   gw block [mt][kg][j][lane] += sum_s in[i][s] * delta[o][s]
// A workgroup owns one 32-sample tile.  Wave 0 (producer) runs forward + backward; whenever a layer's delta tile is
// ready it drops (in tiles, delta tiles, job descriptor) transposed into the next of NBUF LDS buffers.  Waves 1..NC
// (consumers; job i belongs to consumer i % NC) read the descriptor, do the MFMAs over the 32 samples and the f32
// atomics into the gradient image.  The consumers interpret descriptors, so the waves cannot disagree on the job
// sequence; the producer ends with one stop descriptor per consumer.  Effect: the ~50 % of a tile's MFMA work that
// is weight gradients leaves the latency chain of the data path and runs on other SIMDs (the waves of a workgroup
// are spread over the SIMDs of a CU).
//   hand-over: two sequence words per buffer in LDS -- ready[b] = i + 1 once job i is complete in buffer b (release
//   store after the wave's LDS writes), done[b] = i + 1 once its consumer has read everything it needs; the producer
//   re-uses buffer b for job i + NBUF after done[b] == i + 1.  No workgroup barrier: round 1's two-buffer barrier
//   protocol made the producer wait at EVERY job for the consumer to finish the job before last (the consumer is
//   busy 75 % of the backward sweep), and with a barrier a second consumer cannot work across hand-overs.
// ---------------------------------------------------------------------------------------------
#define SF_JOB_HDR 16  // floats reserved for the descriptor in front of the tiles
#define SF_PIPE_CTL 32  // ints in front of the buffers: ready[0..7], done[8..15]
struct SfGradPipe {
  float* lds;   // NBUF buffers of `stride` floats (behind the control words)
  int stride;
  int i;
  int det;      // 1: the consumer stores into this tile's own replica instead of atomically adding to a shared one
  int* ctl;
};
// The sequence words and the data they guard are all LDS, and a wave's LDS instructions are executed in order: the
// hand-over needs "my LDS accesses are done" (s_waitcnt lgkmcnt(0)) and compiler barriers, NOT a workgroup-scope
// fence -- that one would also wait for the consumer's outstanding global atomics (vmcnt(0), microseconds).
__device__ __forceinline__ int sf_pipe_load(const int* p) {
  const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
  return v;
}
__device__ __forceinline__ void sf_pipe_store(int* p, int v, bool writer) {  // called by the whole wave; `writer` lane stores
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (writer) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// producer: buffer of job P.i, free to be written
template <int NBUF>
__device__ __forceinline__ float* sf_pipe_acquire(SfGradPipe& P) {
  const int b = P.i & (NBUF - 1);
  if (P.i >= NBUF)
    while (sf_pipe_load(P.ctl + 8 + b) != P.i - NBUF + 1) __builtin_amdgcn_s_sleep(1);
  return P.lds + b * P.stride;
}
template <int NBUF>
__device__ __forceinline__ void sf_pipe_publish(SfGradPipe& P, int lane) {
  sf_pipe_store(P.ctl + (P.i & (NBUF - 1)), P.i + 1, lane == 0);
  ++P.i;
}

template <int NBUF, int OT, int IT, bool RELU_IN = false>
__device__ __forceinline__ void sf_grad_w(SfGradPipe& P, const f32x16 (&delta)[OT][1], const f32x16 (&in)[IT][1],
                                          float* __restrict__ gw, float* __restrict__ gb, int nGtot, int kg0, int ng,
                                          int lane, SfKLim lim = SfKLim{{1 << 20, 1 << 20, 1 << 20, 1 << 20}}) {
  const int c = lane & 31, h = lane >> 5;
  float* buf = sf_pipe_acquire<NBUF>(P);
  float* tiles = buf + SF_JOB_HDR;
#pragma unroll
  for (int kt = 0; kt < IT; ++kt) sf_tile_to_lds<RELU_IN>(tiles + kt * SF_TL, in[kt][0], c, h);
#pragma unroll
  for (int mt = 0; mt < OT; ++mt) sf_tile_to_lds<false>(tiles + (IT + mt) * SF_TL, delta[mt][0], c, h);
  if (lane == 0) {
    int* d = reinterpret_cast<int*>(buf);
    d[0] = OT; d[1] = IT; d[2] = nGtot; d[3] = kg0; d[4] = ng; d[5] = 0;  // d[5]: stop flag
    const unsigned long long pw = (unsigned long long)gw, pb = (unsigned long long)gb;
    d[6] = (int)(unsigned)pw; d[7] = (int)(unsigned)(pw >> 32);
    d[8] = (int)(unsigned)pb; d[9] = (int)(unsigned)(pb >> 32);
    d[10] = lim.v[0]; d[11] = lim.v[1]; d[12] = lim.v[2]; d[13] = lim.v[3];  // per-output-tile group limit (masked layers)
    d[14] = P.det;
  }
  sf_pipe_publish<NBUF>(P, lane);
}

template <int NBUF, int NC>
__device__ __forceinline__ void sf_grad_stop(SfGradPipe& P, int lane) {
#pragma unroll
  for (int q = 0; q < NC; ++q) {  // consecutive jobs go to different consumers: one stop descriptor each
    float* buf = sf_pipe_acquire<NBUF>(P);
    if (lane == 0) reinterpret_cast<int*>(buf)[5] = 1;
    sf_pipe_publish<NBUF>(P, lane);
  }
}

// the consumer wave: runs until the stop descriptor
template <int NBUF, int NC>
__device__ __forceinline__ void sf_grad_consumer(float* __restrict__ lds, int* __restrict__ ctl, int stride, int lane, int me,
                                                 const SfTrainArgs& a) {
  const int c = lane & 31, h = lane >> 5;
  for (int i = me;; i += NC) {
    const int bsel = i & (NBUF - 1);
    SF_TR(2 * (i < 60 ? i : 60) + 1);
    while (sf_pipe_load(ctl + bsel) != i + 1) __builtin_amdgcn_s_sleep(1);
    SF_TR(2 * (i < 60 ? i : 60) + 2);
    const float* buf = lds + bsel * stride;
    const int* d = reinterpret_cast<const int*>(buf);
    if (__builtin_amdgcn_readfirstlane(d[5])) break;
    const int OT = __builtin_amdgcn_readfirstlane(d[0]), IT = __builtin_amdgcn_readfirstlane(d[1]);
    const int nGtot = __builtin_amdgcn_readfirstlane(d[2]), kg0 = __builtin_amdgcn_readfirstlane(d[3]);
    const int ng = __builtin_amdgcn_readfirstlane(d[4]);
    const bool det = __builtin_amdgcn_readfirstlane(d[14]) != 0;
    float* gw = reinterpret_cast<float*>((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(d[6]) |
                                         ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(d[7]) << 32));
    float* gb = reinterpret_cast<float*>((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(d[8]) |
                                         ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(d[9]) << 32));
    const float* tiles = buf + SF_JOB_HDR;
    for (int mt = 0; mt < OT; ++mt) {
      float bsum = 0.f;
      float bv[16];
      sf_tile_row16(tiles + (IT + mt) * SF_TL, c, h, bv);
      if (gb) {
#pragma unroll
        for (int k = 0; k < 16; ++k) bsum += bv[k];
      }
      // groups past the limit are structurally masked weights (block-triangular MADE layers): no gradient needed
      const int ngm = min(ng, __builtin_amdgcn_readfirstlane(d[10 + mt]) - kg0);
      for (int kt = 0; kt < IT; ++kt) {
        if (kt * 4 < ngm) {
          float av[16];
          sf_tile_row16(tiles + kt * SF_TL, c, h, av);
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
          for (int k = 0; k < 16; ++k) acc = SF_MFMA(av[k], bv[k], acc);
#pragma unroll
          for (int g = 0; g < 4; ++g)
            if (kt * 4 + g < ngm) {
              float* dst = gw + (((size_t)mt * nGtot + kg0 + kt * 4 + g) * 4) * 64 + lane;
              if (det) {  // this tile owns the replica (pre-zeroed): plain stores, summed later in tile order
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[j * 64] = acc[4 * g + j];
              } else if ((acc[4 * g] != 0.f) | (acc[4 * g + 1] != 0.f) | (acc[4 * g + 2] != 0.f) | (acc[4 * g + 3] != 0.f)) {
                // padded output rows are exact zeros in all four values: no atomic traffic for those lanes
#pragma unroll
                for (int j = 0; j < 4; ++j) atomicAdd(dst + j * 64, acc[4 * g + j]);
              }
            }
        }
      }
      if (gb) {
        bsum += sf_xhalf(bsum);
        if (h == 0) {
          if (det) gb[mt * 32 + c] = bsum;
          else atomicAdd(gb + mt * 32 + c, bsum);
        }
      }
    }
    sf_pipe_store(ctl + 8 + bsel, i + 1, lane == 0);  // every operand of the job has been read: the buffer is free
  }
}

// single-wave form (used by the embedding MLP backward, sf_mlp.hip): the same wave transposes and consumes.
// gradient of one linear layer's weights (and optionally bias):
//   gw block [mt][kg][j][lane] += sum_s in[i][s] * delta[o][s]
// lds: (IT + OT) transposed tiles; in tiles first.
template <int OT, int IT, bool RELU_IN = false>
__device__ __forceinline__ void sf_grad_w_local(float* __restrict__ lds, const f32x16 (&delta)[OT][1],
                                          const f32x16 (&in)[IT][1], float* __restrict__ gw,
                                          float* __restrict__ gb, int nGtot, int kg0, int ng, int lane) {
  const int c = lane & 31, h = lane >> 5;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int kt = 0; kt < IT; ++kt) sf_tile_to_lds<RELU_IN>(lds + kt * SF_TL, in[kt][0], c, h);
#pragma unroll
  for (int mt = 0; mt < OT; ++mt) sf_tile_to_lds<false>(lds + (IT + mt) * SF_TL, delta[mt][0], c, h);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int mt = 0; mt < OT; ++mt) {
    float bsum = 0.f;
    float bv[16];
    sf_tile_row16(lds + (IT + mt) * SF_TL, c, h, bv);
#pragma unroll
    for (int k = 0; k < 16; ++k) bsum += bv[k];
#pragma unroll
    for (int kt = 0; kt < IT; ++kt) {
      if (kt * 4 < ng) {
        float av[16];
        sf_tile_row16(lds + kt * SF_TL, c, h, av);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = SF_MFMA(av[k], bv[k], acc);
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

// context gradient: dctx[row, f] += (W^T delta)[f] / x_std[f] for every context feature f (the lane that
// holds (sample, feature) owns that address: plain read-modify-write, no atomics)
template <int HT>
__device__ __forceinline__ void sf_ctx_grad(const SfDev& m, const f32x16 (&delta)[HT][1], const float* __restrict__ wT,
                                            float* __restrict__ dctx_row, bool valid, int lane) {
  const int h = lane >> 5;
  for (int kt = 0; kt * 32 < m.C; ++kt) {
    f32x16 de[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) de[0][0][r] = 0.f;
    sf_mm_acc<1, 1, HT, false, false, true>(de, delta, wT + (size_t)kt * m.nGh * 256, m.nGh, 0, m.nGh, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = kt * 32 + sf_row(r, h);
      if (valid && f < m.C) dctx_row[f] += de[0][0][r] / m.cst[m.c_xstd + f];
    }
  }
}

// DM: static bound of the theta loops (8 when D <= 8: the upper half of the u / G register arrays is then dead)
template <int HT, int NBUF, int NC, int DM>
__global__ __launch_bounds__(64 * (1 + NC), (HT <= 2 ? 2 : 1)) void k_maf_train(SfDev m0, SfTrainArgs a) {
  const SfDev& m = m0;
  extern __shared__ float lds_all[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const long wid = (long)blockIdx.x;  // one 32-sample tile per workgroup: wave 0 producer, wave 1 weight-gradient consumer
  const long base = wid * 32;
  if (base >= a.B) return;
  int* ctl = reinterpret_cast<int*>(lds_all);
  SfGradPipe lds = {lds_all + SF_PIPE_CTL, SF_JOB_HDR + (2 * HT) * SF_TL, 0, 0, ctl};
  if (threadIdx.x < SF_PIPE_CTL) ctl[threadIdx.x] = 0;
  __syncthreads();  // (the only workgroup barrier of the kernel)
  SF_TR(0);
  if (wave >= 1) {
    sf_grad_consumer<NBUF, NC>(lds.lds, ctl, lds.stride, lane, wave - 1, a);
    return;
  }
  // gradient-image replica of this XCD: f32 atomics from different XCDs then never meet on an address
  int xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float* gimg_x = a.gimg + (size_t)(a.det ? wid : (long)(xcc & (SF_GCOPIES - 1))) * a.gimg_stride;
  lds.det = a.det;
  float4* stash = a.act + wid * a.act_per_wave;
  const int TPT = (m.NB + 1) * HT + 1;  // stash tiles per transform: u, h0, a_1..a_NB

  const long row = base + c;
  const bool valid = row < a.B;
  const long ii = valid ? row : a.B - 1;
  const long src = a.idx ? (long)a.idx[ii] : ii;  // library row behind batch row ii
  const float* xr[1] = {a.x + src * m.C};
  f32x16 ct0[1][1];  // first standardised context tile, reused by every context product and weight gradient
  sf_build_ctx_tile<1>(ct0, xr, m, 0, lane >> 5);
  float u[1][SF_DMAX];
  float logdet[1] = {m.logdet0};
#pragma unroll
  for (int p = DM; p < SF_DMAX; ++p) u[0][p] = 0.f;
#pragma unroll
  for (int p = 0; p < DM; ++p) {
    u[0][p] = 0.f;
    if (p < m.D) {
      const int td = (int)m.cst[m.c_tdim + p];
      u[0][p] = a.theta[src * m.D + td] * m.cst[m.c_pscale + p] + m.cst[m.c_pshift + p];
    }
  }
  using Ops = MafOps<HT, 1>;
  SF_TR(1);

  // ------------------------------------------------------------------ forward (with stash)
  for (int t = 0; t < m0.T; ++t) {
    const SfDev m = sf_iter_view(m0);  // loop bounds opaque per iteration: predicates are not hoisted and spilled
    const float* tp = m.packed + (size_t)t * m.t_stride;
    {
      f32x16 ut;
#pragma unroll
      for (int p = 0; p < DM; ++p) ut[p] = u[0][p];
      sf_stash_store(stash, t * TPT, ut, lane);
    }
    f32x16 act[HT][1];
    sf_init_bias<HT, 1>(act, tp + m.o_b0, h);
    {
      f32x16 ut[1][1];
      sf_build_u_tile<1>(ut, u, h);
      sf_mm_acc<HT, 1, 1, false, false, true>(act, ut, tp + m.o_w0, m.nGu, 0, m.nGu, lane);
    }
    sf_ctx_mm<HT, 1>(act, xr, m, tp + m.o_wc, lane, &ct0);
#pragma unroll
    for (int mt = 0; mt < HT; ++mt) sf_stash_store(stash, t * TPT + 1 + mt, act[mt][0], lane);
#pragma unroll
    for (int k = 0; k < SF_NBMAX; ++k) {
      if (k < m.NB) {
        f32x16 b[HT][1];
        sf_init_bias<HT, 1>(b, tp + m.o_bk[k], h);
        sf_mm_acc<HT, 1, HT, false, false, true>(b, act, tp + m.o_wk[k], m.nGh, 0, m.nGh, lane);
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
    sf_mm_acc<1, 1, HT, false, false, true>(fin, act, tp + m.o_wf, m.nGh, 0, m.nGh, lane);
    float ld = 0.f;
#pragma unroll
    for (int p = 0; p < DM; ++p) {
      if (p < m.D) {
        const float s = Ops::scale(m, fin[0][0][2 * (p >> 1)]);
        const float val = s * u[0][p] + fin[0][0][2 * (p >> 1) + 1];
        const bool mine = (h == (p & 1));
        const float oth = sf_xhalf(val);
        u[0][p] = mine ? val : oth;
        ld += mine ? sf_log(s) : 0.f;
      }
    }
    logdet[0] += ld + sf_xhalf(ld);
    SF_TR(2 + t);
  }
  float G[SF_DMAX];  // dL/d(output of the current transform), replicated in both halves
  const float w = valid ? (a.wts ? a.w * a.wts[row] : a.w) : 0.f;
  {
    float ss = 0.f;
#pragma unroll
    for (int p = 0; p < DM; ++p) {
      G[p] = 0.f;
      if (p < m.D) {
        ss += u[0][p] * u[0][p];
        G[p] = w * u[0][p];
      }
    }
    const float nll = 0.5f * ss + 0.5f * (float)m.D * 1.8378770664093453f - logdet[0];
    if (a.loss && valid && h == 0) a.loss[row] = nll;
    if (a.loss_sum) {
      float t = (valid && h == 0) ? nll : 0.f;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
      // values on a 2^-20 grid add exactly in double: the sum does not depend on the order of the atomics
      if (lane == 0) atomicAdd(a.loss_sum, (double)rintf(t * 1048576.0f) * (1.0 / 1048576.0));
    }
  }

  // ------------------------------------------------------------------ backward
  // Activations come back from the stash one layer AHEAD of their use: a_j lives in register set (NB - j) & 1, so the
  // load of a layer's input is issued while the previous layer's data-gradient MFMAs run, and the first two loads of a
  // transform (a_NB and u) are issued before the last product of the transform above it.  (An HBM / L2 round trip per
  // layer used to sit in the dependent chain: four per transform.)
  f32x16 aset[2][HT][1];
  f32x16 utile;
  {
    const int t = m0.T - 1;
#pragma unroll
    for (int mt = 0; mt < HT; ++mt) sf_stash_load(stash, t * TPT + 1 + m.NB * HT + mt, aset[0][mt][0], lane);
    sf_stash_load(stash, t * TPT, utile, lane);
  }
  for (int t = m0.T - 1; t >= 0; --t) {
    const SfDev m = sf_iter_view(m0);
    const float* tp = m.packed + (size_t)t * m.t_stride;
    const float* tpT = m.packedT + (size_t)t * m.tT_stride;
    float* gp = gimg_x + (size_t)t * m.t_stride;
    float uin[1][SF_DMAX];
#pragma unroll
    for (int p = 0; p < SF_DMAX; ++p) uin[0][p] = p < DM ? utile[p] : 0.f;
    // input of the top block, needed after the head: in flight behind the head recomputation
    if (m.NB >= 1) {
#pragma unroll
      for (int mt = 0; mt < HT; ++mt) sf_stash_load(stash, t * TPT + 1 + (m.NB - 1) * HT + mt, aset[1][mt][0], lane);
    }
    // recompute the head
    f32x16 fin[1][1];
    sf_init_bias<1, 1>(fin, tp + m.o_bf, h);
    sf_mm_acc<1, 1, HT, false, false, true>(fin, aset[0], tp + m.o_wf, m.nGh, 0, m.nGh, lane);
    SF_TR(20 + 10 * (m0.T - 1 - t) + 0);
    f32x16 dfin[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) dfin[0][0][r] = 0.f;
    float Gd[SF_DMAX];
#pragma unroll
    for (int p = 0; p < DM; ++p) {
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
    SF_TR(20 + 10 * (m0.T - 1 - t) + 1);
    sf_grad_w<NBUF, 1, HT>(lds, dfin, aset[0], gp + m.o_wf, gp + m.o_bf, m.nGh, 0, m.nGh, lane);
    SF_TR(20 + 10 * (m0.T - 1 - t) + 2);
    f32x16 dh[HT][1];
#pragma unroll
    for (int mt = 0; mt < HT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dh[mt][0][r] = 0.f;
    sf_mm_acc<HT, 1, 1, false, false, true>(dh, dfin, tpT + m.oT_wf, m.nGf, 0, m.nGf, lane);
    SF_TR(20 + 10 * (m0.T - 1 - t) + 3);
#pragma unroll
    for (int d = 0; d < SF_NBMAX; ++d) {  // d = distance from the top block: block k = NB - 1 - d
      if (d < m.NB) {
        const int k = m.NB - 1 - d;
        // the descriptor arrays are indexed with select chains (a dynamic index would put the descriptor in scratch)
        int o_wk = m.o_wk[0], o_bk = m.o_bk[0], oT_wk = m.oT_wk[0];
#pragma unroll
        for (int q = 1; q < SF_NBMAX; ++q) {
          o_wk = (k == q) ? m.o_wk[q] : o_wk;
          o_bk = (k == q) ? m.o_bk[q] : o_bk;
          oT_wk = (k == q) ? m.oT_wk[q] : oT_wk;
        }
        f32x16 (&aout)[HT][1] = aset[d & 1];       // a_{k+1}: output of block k
        f32x16 (&ain)[HT][1] = aset[(d + 1) & 1];  // a_k: its input (loaded one step ago)
        f32x16 dpre[HT][1];
#pragma unroll
        for (int mt = 0; mt < HT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) dpre[mt][0][r] = dh[mt][0][r] * (1.0f - aout[mt][0][r] * aout[mt][0][r]);
        sf_grad_w<NBUF, HT, HT>(lds, dpre, ain, gp + o_wk, gp + o_bk, m.nGh, 0, m.nGh, lane,
                                SfKLim{{m.mt_kend[0], m.mt_kend[1], m.mt_kend[2], m.mt_kend[3]}});
        // a_{k+1} is dead now: its registers take the input of the block below (or, after the bottom block, the top
        // activation of the next transform), in flight during this block's data-gradient product
        if (k >= 1) {
#pragma unroll
          for (int mt = 0; mt < HT; ++mt) sf_stash_load(stash, t * TPT + 1 + (k - 1) * HT + mt, aout[mt][0], lane);
        }
        SF_TR(20 + 10 * (m0.T - 1 - t) + 4);
#pragma unroll
        for (int mt = 0; mt < HT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) dh[mt][0][r] = 0.f;
        sf_mm_acc<HT, 1, HT, false, false, true>(dh, dpre, tpT + oT_wk, m.nGh, 0, m.nGh, lane);
        SF_TR(20 + 10 * (m0.T - 1 - t) + 5);
      }
    }
    // initial layer: dW0 (u tile), dWc (context tiles), d(b0+bc)
    {
      f32x16 ut[1][1];
      sf_build_u_tile<1>(ut, uin, h);
      sf_grad_w<NBUF, HT, 1>(lds, dh, ut, gp + m.o_w0, gp + m.o_b0, m.nGu, 0, m.nGu, lane);
    }
    for (int kt = 0; kt * 4 < m.nGc; ++kt) {
      f32x16 ct[1][1];
      if (kt == 0) ct[0][0] = ct0[0][0];
      else sf_build_ctx_tile<1>(ct, xr, m, kt, h);
      sf_grad_w<NBUF, HT, 1>(lds, dh, ct, gp + m.o_wc, nullptr, m.nGc, kt * 4, min(4, m.nGc - kt * 4), lane);
    }
    // first loads of the transform above: in flight during the last products of this one
    if (t >= 1) {
#pragma unroll
      for (int mt = 0; mt < HT; ++mt) sf_stash_load(stash, (t - 1) * TPT + 1 + m.NB * HT + mt, aset[0][mt][0], lane);
      sf_stash_load(stash, (t - 1) * TPT, utile, lane);
    }
    SF_TR(20 + 10 * (m0.T - 1 - t) + 8);
    if (a.dctx) sf_ctx_grad<HT>(m, dh, tpT + m.oT_wc, a.dctx + ii * m.C, valid, lane);
    // delta_u = W0^T delta_h0
    f32x16 du[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) du[0][0][r] = 0.f;
    sf_mm_acc<1, 1, HT, false, false, true>(du, dh, tpT + m.oT_w0, m.nGh, 0, m.nGh, lane);
#pragma unroll
    for (int p = 0; p < DM; ++p) {
      if (p < m.D) {
        const float v = du[0][0][(p & 3) + 4 * (p >> 3)];
        const float oth = sf_xhalf(v);
        G[p] = Gd[p] + ((h == ((p >> 2) & 1)) ? v : oth);
      }
    }
    SF_TR(20 + 10 * (m0.T - 1 - t) + 9);
  }
  sf_grad_stop<NBUF, NC>(lds, lane);
  SF_TR(8);
}


// =============================================================================================
// NSF forward + backward
// =============================================================================================
// sum over the 32 samples of one row half (both halves hold identical values)
__device__ __forceinline__ float sf_half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int HT, int PT>
struct SfNsfLds {  // transposed tiles needed at once by sf_grad_w
  static constexpr int tiles = (2 * HT > HT + PT) ? 2 * HT : HT + PT;
};

template <int HT, int PT>
__global__ __launch_bounds__(128, (HT <= 2 ? 2 : 1)) void k_nsf_train(SfDev m0, SfTrainArgs a) {
  const SfDev& m = m0;
  extern __shared__ float lds_all[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const long wid = (long)blockIdx.x;  // one 32-sample tile per workgroup: wave 0 producer, wave 1 weight-gradient consumer
  const long base = wid * 32;
  if (base >= a.B) return;
  constexpr int NBUF = 2, NC = 1;
  int* ctl = reinterpret_cast<int*>(lds_all);
  SfGradPipe lds = {lds_all + SF_PIPE_CTL, SF_JOB_HDR + SfNsfLds<HT, PT>::tiles * SF_TL, 0, 0, ctl};
  if (threadIdx.x < SF_PIPE_CTL) ctl[threadIdx.x] = 0;
  __syncthreads();
  if (wave >= 1) {
    sf_grad_consumer<NBUF, NC>(lds.lds, ctl, lds.stride, lane, wave - 1, a);
    return;
  }
  // gradient-image replica of this XCD: f32 atomics from different XCDs then never meet on an address
  int xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float* gimg_x = a.gimg + (size_t)(a.det ? wid : (long)(xcc & (SF_GCOPIES - 1))) * a.gimg_stride;
  lds.det = a.det;
  float4* stash = a.act + wid * a.act_per_wave;
  // stash tiles per transform: [0] u_in, [1..HT] h_0, per block k: t1, t2, h_{k+1} (HT each), last: u'
  const int TPT = 2 + (3 * m.NB + 1) * HT;
  using Ops = NsfOps<HT, PT, 1>;

  const long row = base + c;
  const bool valid = row < a.B;
  const long ii = valid ? row : a.B - 1;
  const long src = a.idx ? (long)a.idx[ii] : ii;  // library row behind batch row ii
  const float* xr[1] = {a.x + src * m.C};
  f32x16 ct0[1][1];  // first standardised context tile, reused by every context product and weight gradient
  sf_build_ctx_tile<1>(ct0, xr, m, 0, lane >> 5);
  float u[1][SF_DMAX];
  float logdet[1] = {m.logdet0};
#pragma unroll
  for (int p = 0; p < SF_DMAX; ++p) {
    u[0][p] = 0.f;
    if (p < m.D) u[0][p] = a.theta[src * m.D + p] * m.cst[m.c_pscale + p] + m.cst[m.c_pshift + p];
  }
  auto store_u = [&](int tile) {
    f32x16 ut;
#pragma unroll
    for (int p = 0; p < SF_DMAX; ++p) ut[p] = u[0][p];
    sf_stash_store(stash, tile, ut, lane);
  };

  // ------------------------------------------------------------------ forward (with stash)
  for (int t = 0; t < m0.T; ++t) {
    const SfDev m = sf_iter_view(m0);  // loop bounds opaque per iteration: predicates are not hoisted and spilled
    const float* tp = m.packed + (size_t)t * m.t_stride;
    const int sb = t * TPT;
    store_u(sb);
    f32x16 hid[HT][1];
    sf_init_bias<HT, 1>(hid, tp + m.o_bin, h);
    {
      f32x16 ut[1][1];
      sf_build_u_tile<1>(ut, u, h);
      sf_mm_acc<HT, 1, 1, false, false, true>(hid, ut, tp + m.o_winu, m.nGu, 0, m.nGu, lane);
    }
    sf_ctx_mm<HT, 1>(hid, xr, m, tp + m.o_winc, lane, &ct0);
#pragma unroll
    for (int mt = 0; mt < HT; ++mt) sf_stash_store(stash, sb + 1 + mt, hid[mt][0], lane);
#pragma unroll
    for (int k = 0; k < SF_NBMAX; ++k) {
      if (k < m.NB) {
        const int bb = sb + 1 + HT + k * 3 * HT;
        f32x16 t2[HT][1];
        {
          f32x16 t1[HT][1];
          sf_init_bias<HT, 1>(t1, tp + m.o_b1[k], h);
          sf_mm_acc<HT, 1, HT, true, false, true>(t1, hid, tp + m.o_w1[k], m.nGh, 0, m.nGh, lane);
#pragma unroll
          for (int mt = 0; mt < HT; ++mt) sf_stash_store(stash, bb + mt, t1[mt][0], lane);
          sf_init_bias<HT, 1>(t2, tp + m.o_b2[k], h);
          sf_mm_acc<HT, 1, HT, true, false, true>(t2, t1, tp + m.o_w2[k], m.nGh, 0, m.nGh, lane);
        }
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) {
          sf_stash_store(stash, bb + HT + mt, t2[mt][0], lane);
          f32x16 g[1][1];
          sf_init_bias<1, 1>(g, tp + m.o_bg[k] + mt * 32, h);
          sf_ctx_mm<1, 1>(g, xr, m, tp + m.o_wg[k] + mt * m.nGc * 256, lane, &ct0);
#pragma unroll
          for (int r = 0; r < 16; ++r) hid[mt][0][r] += t2[mt][0][r] * sf_sigmoid(g[0][0][r]);
          sf_stash_store(stash, bb + 2 * HT + mt, hid[mt][0], lane);
        }
      }
    }
    Ops::spline_apply(m, tp, t, hid, u, logdet, false, lane);
    store_u(sb + TPT - 1);
    if (m.D > 1) Ops::lu_forward(m, tp + m.o_lu, u, logdet);
  }
  float G[SF_DMAX];
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
    const float nll = 0.5f * ss + 0.5f * (float)m.D * 1.8378770664093453f - logdet[0];
    if (a.loss && valid && h == 0) a.loss[row] = nll;
    if (a.loss_sum) {
      float t = (valid && h == 0) ? nll : 0.f;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
      // values on a 2^-20 grid add exactly in double: the sum does not depend on the order of the atomics
      if (lane == 0) atomicAdd(a.loss_sum, (double)rintf(t * 1048576.0f) * (1.0 / 1048576.0));
    }
  }

  // ------------------------------------------------------------------ backward
  for (int t = m0.T - 1; t >= 0; --t) {
    const SfDev m = sf_iter_view(m0);
    const float* tp = m.packed + (size_t)t * m.t_stride;
    const float* tpT = m.packedT + (size_t)t * m.tT_stride;
    float* gp = gimg_x + (size_t)t * m.t_stride;
    const int sb = t * TPT;
    const int D = m.D;
    float up[SF_DMAX], uin[1][SF_DMAX];
    {
      f32x16 ut;
      sf_stash_load(stash, sb + TPT - 1, ut, lane);
#pragma unroll
      for (int p = 0; p < SF_DMAX; ++p) up[p] = ut[p];
      sf_stash_load(stash, sb, ut, lane);
#pragma unroll
      for (int p = 0; p < SF_DMAX; ++p) uin[0][p] = ut[p];
    }
    // ---- LULinear backward:  y = L t + b, t = U u'
    if (D > 1) {
      const float* lp = tp + m.o_lu;
      float* gl = gp + m.o_lu;
      const float* Lm = lp;
      const float* Um = lp + D * D;
      const float* ud = lp + 2 * D * D;
      float tt[SF_DMAX], dt[SF_DMAX], dg[SF_DMAX];
#pragma unroll
      for (int i = 0; i < SF_DMAX; ++i) {
        tt[i] = 0.f; dt[i] = 0.f; dg[i] = 1.f;
        if (i < D) {
          dg[i] = sf_softplus(ud[i]) + m.lu_eps;
          tt[i] = dg[i] * up[i];
#pragma unroll
          for (int j = 0; j < SF_DMAX; ++j)
            if (j > i && j < D) tt[i] += Um[i * D + j] * up[j];
        }
      }
      // dt = L^T G ; dL_ij += G_i t_j ; dbias += G
#pragma unroll
      for (int j = 0; j < SF_DMAX; ++j)
        if (j < D) {
          dt[j] = G[j];
#pragma unroll
          for (int i = 0; i < SF_DMAX; ++i)
            if (i > j && i < D) dt[j] += Lm[i * D + j] * G[i];
        }
#pragma unroll
      for (int i = 0; i < SF_DMAX; ++i)
        if (i < D) {
          const float sb_ = sf_half_sum(G[i]);
          if (lane == 0) atomicAdd(gl + 2 * D * D + D + i, sb_);
#pragma unroll
          for (int j = 0; j < SF_DMAX; ++j)
            if (j < i) {
              const float sv = sf_half_sum(G[i] * tt[j]);
              if (lane == 0) atomicAdd(gl + i * D + j, sv);
            }
        }
      // du' = U^T dt ; dU_ij += dt_i u'_j ; d udiag
      float Gn[SF_DMAX];
#pragma unroll
      for (int j = 0; j < SF_DMAX; ++j) {
        Gn[j] = 0.f;
        if (j < D) {
          Gn[j] = dg[j] * dt[j];
#pragma unroll
          for (int i = 0; i < SF_DMAX; ++i)
            if (i < j) Gn[j] += Um[i * D + j] * dt[i];
        }
      }
#pragma unroll
      for (int i = 0; i < SF_DMAX; ++i)
        if (i < D) {
          const float ddiag = dt[i] * up[i] - w / dg[i];
          const float sd = sf_half_sum(ddiag * sf_sigmoid(ud[i]));
          if (lane == 0) atomicAdd(gl + 2 * D * D + i, sd);
#pragma unroll
          for (int j = 0; j < SF_DMAX; ++j)
            if (j > i && j < D) {
              const float sv = sf_half_sum(dt[i] * up[j]);
              if (lane == 0) atomicAdd(gl + D * D + i * D + j, sv);
            }
        }
#pragma unroll
      for (int p = 0; p < SF_DMAX; ++p) G[p] = Gn[p];
    }
    // ---- spline head + spline backward
    f32x16 hN[HT][1];
#pragma unroll
    for (int mt = 0; mt < HT; ++mt) sf_stash_load(stash, sb + 1 + HT + (m.NB - 1) * 3 * HT + 2 * HT + mt, hN[mt][0], lane);
    f32x16 dh[HT][1];
#pragma unroll
    for (int mt = 0; mt < HT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dh[mt][0][r] = 0.f;
    {
      const int start = t & 1;
      const int d_tr = (D - start + 1) / 2;
      for (int jp = 0; jp * 2 < d_tr; ++jp) {
        f32x16 q[PT][1];
        sf_init_bias<PT, 1>(q, tp + m.o_bout + jp * PT * 32, h);
        sf_mm_acc<PT, 1, HT, false, false, true>(q, hN, tp + m.o_wout + jp * PT * m.nGh * 256, m.nGh, 0, m.nGh, lane);
        const int kdim = 2 * jp + h;
        const bool have = kdim < d_tr;
        const int tgt = start + 2 * kdim;
        const int tgt_o = start + 2 * (2 * jp + (1 - h));
        const bool have_o = (2 * jp + (1 - h)) < d_tr;
        float vin = 0.f, Go = 0.f;
#pragma unroll
        for (int p = 0; p < SF_DMAX; ++p) {
          vin = (p == tgt) ? uin[0][p] : vin;
          Go = (p == tgt) ? G[p] : Go;
        }
        f32x16 dq[PT][1];
        float vout, lad, dv;
        SfSplineBwd<PT>::template eval<1>(m, q, 0, vin, have ? Go : 0.f, have ? -w : 0.f, vout, lad, dv, dq);
        dv = have ? dv : 0.f;
        const float dvo = sf_xhalf(dv);
#pragma unroll
        for (int p = 0; p < SF_DMAX; ++p) {
          G[p] = (have && p == tgt) ? dv : G[p];
          G[p] = (have_o && p == tgt_o) ? dvo : G[p];
        }
        sf_grad_w<NBUF, PT, HT>(lds, dq, hN, gp + m.o_wout + jp * PT * m.nGh * 256, gp + m.o_bout + jp * PT * 32,
                          m.nGh, 0, m.nGh, lane);
        sf_mm_acc<HT, 1, PT, false, false, true>(dh, dq, tpT + m.oT_wout + jp * HT * (PT * 4) * 256, PT * 4, 0, PT * 4, lane);
      }
    }
    // ---- ResidualNet backward
#pragma unroll
    for (int kk = 0; kk < SF_NBMAX; ++kk) {
      const int k = SF_NBMAX - 1 - kk;
      if (k < m.NB) {
        const int bb = sb + 1 + HT + k * 3 * HT;
        f32x16 dt2[HT][1];
        {
          f32x16 dgate[HT][1];
#pragma unroll
          for (int mt = 0; mt < HT; ++mt) {
            f32x16 t2;
            sf_stash_load(stash, bb + HT + mt, t2, lane);
            f32x16 g[1][1];
            sf_init_bias<1, 1>(g, tp + m.o_bg[k] + mt * 32, h);
            sf_ctx_mm<1, 1>(g, xr, m, tp + m.o_wg[k] + mt * m.nGc * 256, lane, &ct0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float sg = sf_sigmoid(g[0][0][r]);
              dt2[mt][0][r] = dh[mt][0][r] * sg;
              dgate[mt][0][r] = dh[mt][0][r] * t2[r] * sg * (1.f - sg);
            }
          }
          for (int kt = 0; kt * 4 < m.nGc; ++kt) {
            f32x16 ct[1][1];
            if (kt == 0) ct[0][0] = ct0[0][0];
            else sf_build_ctx_tile<1>(ct, xr, m, kt, h);
            sf_grad_w<NBUF, HT, 1>(lds, dgate, ct, gp + m.o_wg[k], kt == 0 ? gp + m.o_bg[k] : nullptr, m.nGc, kt * 4,
                             min(4, m.nGc - kt * 4), lane);
          }
          if (a.dctx) sf_ctx_grad<HT>(m, dgate, tpT + m.oT_wg[k], a.dctx + ii * m.C, valid, lane);
        }
        f32x16 t1[HT][1];
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) sf_stash_load(stash, bb + mt, t1[mt][0], lane);
        sf_grad_w<NBUF, HT, HT, true>(lds, dt2, t1, gp + m.o_w2[k], gp + m.o_b2[k], m.nGh, 0, m.nGh, lane);
        f32x16 dt1[HT][1];
#pragma unroll
        for (int mt = 0; mt < HT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) dt1[mt][0][r] = 0.f;
        sf_mm_acc<HT, 1, HT, false, false, true>(dt1, dt2, tpT + m.oT_w2[k], m.nGh, 0, m.nGh, lane);
#pragma unroll
        for (int mt = 0; mt < HT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) dt1[mt][0][r] = t1[mt][0][r] > 0.f ? dt1[mt][0][r] : 0.f;
        // h_k = input of this block
        f32x16 hk[HT][1];
#pragma unroll
        for (int mt = 0; mt < HT; ++mt)
          sf_stash_load(stash, k == 0 ? sb + 1 + mt : sb + 1 + HT + (k - 1) * 3 * HT + 2 * HT + mt, hk[mt][0], lane);
        sf_grad_w<NBUF, HT, HT, true>(lds, dt1, hk, gp + m.o_w1[k], gp + m.o_b1[k], m.nGh, 0, m.nGh, lane);
        f32x16 dr0[HT][1];
#pragma unroll
        for (int mt = 0; mt < HT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) dr0[mt][0][r] = 0.f;
        sf_mm_acc<HT, 1, HT, false, false, true>(dr0, dt1, tpT + m.oT_w1[k], m.nGh, 0, m.nGh, lane);
#pragma unroll
        for (int mt = 0; mt < HT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) dh[mt][0][r] += hk[mt][0][r] > 0.f ? dr0[mt][0][r] : 0.f;
      }
    }
    // ---- initial layer
    {
      f32x16 ut[1][1];
      sf_build_u_tile<1>(ut, uin, h);
      sf_grad_w<NBUF, HT, 1>(lds, dh, ut, gp + m.o_winu, gp + m.o_bin, m.nGu, 0, m.nGu, lane);
    }
    for (int kt = 0; kt * 4 < m.nGc; ++kt) {
      f32x16 ct[1][1];
      if (kt == 0) ct[0][0] = ct0[0][0];
      else sf_build_ctx_tile<1>(ct, xr, m, kt, h);
      sf_grad_w<NBUF, HT, 1>(lds, dh, ct, gp + m.o_winc, nullptr, m.nGc, kt * 4, min(4, m.nGc - kt * 4), lane);
    }
    if (a.dctx) sf_ctx_grad<HT>(m, dh, tpT + m.oT_wc, a.dctx + ii * m.C, valid, lane);
    f32x16 du[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) du[0][0][r] = 0.f;
    sf_mm_acc<1, 1, HT, false, false, true>(du, dh, tpT + m.oT_winu, m.nGh, 0, m.nGh, lane);
#pragma unroll
    for (int p = 0; p < SF_DMAX; ++p) {
      if (p < D) {
        const float v = du[0][0][(p & 3) + 4 * (p >> 3)];
        const float oth = sf_xhalf(v);
        G[p] += (h == ((p >> 2) & 1)) ? v : oth;
      }
    }
  }
  sf_grad_stop<NBUF, NC>(lds, lane);
}
