// sf_device.h -- register-tile MLP engine on v_mfma_f32_32x32x2_f32 (gfx950 only).
// Layout conventions: sf_layout.h.  Everything here is __device__ __forceinline__ and
// statically indexed so that activation tiles live in VGPRs for a whole flow.
#pragma once
#include <hip/hip_runtime.h>

#include "sf_layout.h"

// Developer build -DSF_FUZZ_SCHED (scripts/fuzz_sched.sh): behind EVERY workgroup barrier of the library -- __syncthreads() and the
// s_barrier helpers of the cooperative kernels -- a wave sleeps for a time that depends on the wave and on the clock, so that the order
// in which the waves reach the next phase changes from barrier to barrier: a missing barrier then loses its race sooner or later
// (round 5: one in k_ar_train had never lost it with one workgroup per CU; this build shows it at once).
#ifdef SF_FUZZ_SCHED
__device__ __forceinline__ void sf_fuzz_delay() {
  const unsigned int w = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int t = (unsigned int)__builtin_amdgcn_s_memtime();
  const unsigned int n = ((w * 2654435761u) ^ (t >> 3)) % 13u;
  for (unsigned int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(20);
}
__device__ __forceinline__ void sf_syncthreads_fuzz() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  sf_fuzz_delay();
}
#define __syncthreads() sf_syncthreads_fuzz()
#define SF_FUZZ() sf_fuzz_delay()
#else
#define SF_FUZZ() do { } while (0)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ constexpr int sf_row(int r, int h) {
  return (r & 3) + 8 * (r >> 2) + 4 * h;
}

// ---- scalar math -------------------------------------------------------------------------
__device__ __forceinline__ float sf_tanh(float x) {
  // 1 - 2/(1+e^{2x}); abs error ~1e-7, saturates cleanly at +-1
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float sf_sigmoid(float x) {
  const float e = __builtin_amdgcn_exp2f(-x * 1.4426950408889634f);
  return __builtin_amdgcn_rcpf(1.0f + e);
}
// hardware exp2/log2 forms (v_exp_f32 / v_log_f32, ~1 ulp): used for every in-flow transcendental
__device__ __forceinline__ float sf_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float sf_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float sf_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float sf_softplus(float x) {  // torch: threshold 20
  return x > 20.0f ? x : sf_log(1.0f + sf_exp(x));
}
// attempts-per-galaxy counters (n_drawn): int32, pinned at INT32_MAX once they would wrap (no attempt ceiling + S slots of
// a galaxy that accepts one draw in a million: S x attempts passes 2^31).  An add that finds the counter too full to take it
// writes the maximum back; whatever order concurrent adds arrive in, the last write after an overflow is such a write.
__device__ __forceinline__ void sf_sat_add(int32_t* p, int v) {
  const int old = atomicAdd(p, v);
  if (v > 0 && old > 0x7fffffff - v) atomicExch(p, 0x7fffffff);
}
__device__ __forceinline__ float sf_xhalf(float v) {  // value held by the other row-half
  return __shfl_xor(v, 32, 64);
}

// ---- transform base pointer ------------------------------------------------------------------
// LDSW: the whole operand image of transform t is copied into the workgroup's LDS once and every
// wave of the (512-thread) workgroup reads its MFMA A operands from there with ds_read_b128;
// otherwise weights stream from L2.  Must be called by every thread of the workgroup.
// A zero the optimiser cannot see through.  Added to the per-transform operand base so that the hundreds of
// weight-fragment addresses derived from it are recomputed (cheap SALU adds) inside the transform loop instead of
// being hoisted out of it as loop invariants and spilled (v_writelane / v_readlane) for lack of SGPRs.
__device__ __forceinline__ int sf_opaque_zero() {
  int z;
  asm volatile("s_mov_b32 %0, 0" : "=s"(z));
  return z;
}
// Per-iteration view of the model descriptor: the loop bounds carry an opaque zero, so the many predicates
// derived from them (k < K, kg < nGh, p < D ...) are evaluated where they are used (one s_cmp each) instead of
// being hoisted out of the transform loop as 64-bit masks that then live in spilled SGPRs.
__device__ __forceinline__ SfDev sf_iter_view(const SfDev& m) {
  SfDev v = m;
  const int z = sf_opaque_zero();
  v.K += z; v.D += z; v.C += z; v.NB += z; v.nGu += z; v.nGc += z; v.nGh += z;
  return v;
}
template <bool LDSW>
__device__ __forceinline__ const float* sf_stage_part(const SfDev& m, int t, int part, float* lds) {
  const float* src = m.packed + (size_t)t * m.t_stride + sf_opaque_zero();
  if (!LDSW) return src;
  int lo = 0, hi = 0;  // part_off[part], part_off[part + 1] without a runtime-indexed load (keeps the descriptor in SGPRs)
  int blo = 0, bhi = 0;  // ... and the part's slice of the bf16 image
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    lo = (q == part) ? m.part_off[q] : lo;
    hi = (q == part) ? m.part_off[q + 1] : hi;
    blo = (q == part) ? m.partB_off[q] : blo;
    bhi = (q == part) ? m.partB_off[q + 1] : bhi;
  }
  __syncthreads();  // previous image no longer in use
  const float4* __restrict__ s4 = reinterpret_cast<const float4*>(src + lo);
  float4* __restrict__ d4 = reinterpret_cast<float4*>(lds);
  const int n4 = (hi - lo) >> 2;
  // direct global -> LDS copies (global_load_lds_dwordx4): no staging registers, no ds_write; the LDS destination of
  // a wave-instruction is its (wave-uniform) base + lane * 16 bytes
  const int lane_ = threadIdx.x & 63;
  for (int i = threadIdx.x; i < n4; i += blockDim.x)
    __builtin_amdgcn_global_load_lds((const void*)(s4 + i), (void __attribute__((address_space(3)))*)(d4 + (i - lane_)), 16, 0, 0);
  if (m.hidden_bf16) {  // the part's bf16 hidden operands right behind its fp32 slice
    const uint4* __restrict__ sb = reinterpret_cast<const uint4*>(m.packedB + (size_t)t * m.tB_stride + blo);
    uint4* __restrict__ db = reinterpret_cast<uint4*>(lds + (hi - lo));
    const int nb = (bhi - blo) >> 3;
    for (int i = threadIdx.x; i < nb; i += blockDim.x)
      __builtin_amdgcn_global_load_lds((const void*)(sb + i), (void __attribute__((address_space(3)))*)(db + (i - lane_)), 16, 0, 0);
  }
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the copies have landed
  __syncthreads();
  return lds - lo + sf_opaque_zero();  // so that (returned + block offset) lands inside the staged part
}
template <bool LDSW>
__device__ __forceinline__ const float* sf_stage(const SfDev& m, int t, float* lds) {
  return sf_stage_part<LDSW>(m, t, 0, lds);
}

// ---- accumulator init from the bias image [mt][h][16] ------------------------------------
template <int OT, int NS>
__device__ __forceinline__ void sf_init_bias(f32x16 (&acc)[OT][NS], const float* __restrict__ bp, int h) {
#pragma unroll
  for (int mt = 0; mt < OT; ++mt) {
    const float4* p = reinterpret_cast<const float4*>(bp + (mt * 2 + h) * 16);
    const float4 b0 = p[0], b1 = p[1], b2 = p[2], b3 = p[3];
    f32x16 v;
    v[0] = b0.x; v[1] = b0.y; v[2] = b0.z; v[3] = b0.w;
    v[4] = b1.x; v[5] = b1.y; v[6] = b1.z; v[7] = b1.w;
    v[8] = b2.x; v[9] = b2.y; v[10] = b2.z; v[11] = b2.w;
    v[12] = b3.x; v[13] = b3.y; v[14] = b3.z; v[15] = b3.w;
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) acc[mt][ns] = v;
  }
}

// ---- tiles from the per-galaxy context table (rows in tile order; lane's rows 8j + 4h + 0..3) --------------
template <int OT, int NS>
__device__ __forceinline__ void sf_ctab_load(f32x16 (&acc)[OT][NS], const float* const (&cg)[NS], int off, int h) {
#pragma unroll
  for (int mt = 0; mt < OT; ++mt)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(cg[ns] + off + mt * 32 + 8 * j + 4 * h);
        acc[mt][ns][4 * j] = v.x; acc[mt][ns][4 * j + 1] = v.y; acc[mt][ns][4 * j + 2] = v.z; acc[mt][ns][4 * j + 3] = v.w;
      }
}
template <int OT>
__device__ __forceinline__ void sf_ctab_store(const f32x16 (&acc)[OT][1], float* dst, int h) {
#pragma unroll
  for (int mt = 0; mt < OT; ++mt)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4*>(dst + mt * 32 + 8 * j + 4 * h) =
          make_float4(acc[mt][0][4 * j], acc[mt][0][4 * j + 1], acc[mt][0][4 * j + 2], acc[mt][0][4 * j + 3]);
}

// ---- acc[mt] += W[mt, groups kg0..kg0+ng) . in  -------------------------------------------
// in[IT][NS]: IT input tiles (static); ng active groups (runtime, wave-uniform, <= 4*IT);
// nGtot: group stride of the weight block.  One float4 load feeds 4*NS MFMAs.
// lim (optional): per-output-tile group limit (block-triangular masked layers).
struct SfKLim {
  int v[4];
};
template <int OT, int NS, int IT, bool RELU, bool LIM = false, bool PF = false>
__device__ __forceinline__ void sf_mm_acc(f32x16 (&acc)[OT][NS], const f32x16 (&in)[IT][NS],
                                          const float* __restrict__ wp, int nGtot, int kg0, int ng,
                                          int lane, SfKLim lim = SfKLim{{0, 0, 0, 0}}) {
  const float4* __restrict__ w4 = reinterpret_cast<const float4*>(wp);
#pragma unroll
  for (int mt = 0; mt < OT; ++mt) {
    const int ngm = LIM ? min(ng, lim.v[mt]) : ng;
    float4 wpre[PF ? IT * 4 : 1];
    if (PF) {
      // issue every fragment load of this output tile before the first MFMA (indices past the
      // active range are clamped: duplicate, cache-resident loads) so their latencies overlap
      const int last = ngm > 0 ? ngm - 1 : 0;
#pragma unroll
      for (int g = 0; g < IT * 4; ++g) wpre[g] = w4[(mt * nGtot + kg0 + min(g, last)) * 64 + lane];
    }
#pragma unroll
    for (int g = 0; g < IT * 4; ++g) {
      if (g < ngm) {
        const float4 w = PF ? wpre[PF ? g : 0] : w4[(mt * nGtot + kg0 + g) * 64 + lane];
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          float b0 = in[g >> 2][ns][(g & 3) * 4 + 0];
          float b1 = in[g >> 2][ns][(g & 3) * 4 + 1];
          float b2 = in[g >> 2][ns][(g & 3) * 4 + 2];
          float b3 = in[g >> 2][ns][(g & 3) * 4 + 3];
          if (RELU) {
            b0 = fmaxf(b0, 0.f); b1 = fmaxf(b1, 0.f); b2 = fmaxf(b2, 0.f); b3 = fmaxf(b3, 0.f);
          }
          acc[mt][ns] = SF_MFMA(w.x, b0, acc[mt][ns]);
          acc[mt][ns] = SF_MFMA(w.y, b1, acc[mt][ns]);
          acc[mt][ns] = SF_MFMA(w.z, b2, acc[mt][ns]);
          acc[mt][ns] = SF_MFMA(w.w, b3, acc[mt][ns]);
        }
      }
    }
  }
}

// ---- bf16 operands for the hidden H x H layers (opt-in, inference only) -----------------------
// v_mfma_f32_32x32x16_bf16 consumes registers 8s..8s+7 of an accumulator-layout tile as the B fragment of
// the 16-row k-step s (element j of row-half h = row 16s + 8(j>>2) + 4h + (j&3)); the weight image stores
// the A fragment in the same k order (sf_layout.cpp emit_bf16).  fp32 accumulation.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NS, int IT, bool RELU>
__device__ __forceinline__ bf16x8 sf_bfrag(const f32x16 (&in)[IT][NS], int ns, int ks) {
  bf16x8 b;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = in[ks >> 1][ns][8 * (ks & 1) + j];
    if (RELU) v = fmaxf(v, 0.f);
    b[j] = (__bf16)v;
  }
  return b;
}

template <int OT, int NS, int IT, bool RELU, bool LIM = false>
__device__ __forceinline__ void sf_mm_acc_bf16(f32x16 (&acc)[OT][NS], const f32x16 (&in)[IT][NS],
                                               const unsigned short* __restrict__ wB, int nKStot, int nks,
                                               int lane, SfKLim lim = SfKLim{{0, 0, 0, 0}}) {
  const uint4* __restrict__ w4 = reinterpret_cast<const uint4*>(wB);
#pragma unroll
  for (int mt = 0; mt < OT; ++mt) {
    const int nk = LIM ? min(nks, lim.v[mt]) : nks;
#pragma unroll
    for (int ks = 0; ks < IT * 2; ++ks) {
      if (ks < nk) {
        const uint4 wv = w4[(mt * nKStot + ks) * 64 + lane];
        const bf16x8 a = __builtin_bit_cast(bf16x8, wv);
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
          acc[mt][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, sf_bfrag<NS, IT, RELU>(in, ns, ks), acc[mt][ns], 0, 0, 0);
      }
    }
  }
}

// Split-bf16 form (NSF sampler image, hidden_bf16 == 2): weights stored as hi = bf16(w) and lo = bf16(w - hi) fragments
// ([mt][ks][hi | lo][64 lanes][8]), activations split the same way on the fly; acc += hi.hi + hi.lo + lo.hi on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation: ~2^-17 relative per product at a fifth of the fp32-MFMA time.
template <int NS, int IT, bool RELU>
__device__ __forceinline__ void sf_bfrag_split(const f32x16 (&in)[IT][NS], int ns, int ks, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = in[ks >> 1][ns][8 * (ks & 1) + j];
    if (RELU) v = fmaxf(v, 0.f);
    const __bf16 h = (__bf16)v;
    hi[j] = h;
    lo[j] = (__bf16)(v - (float)h);
  }
}
template <int OT, int NS, int IT, bool RELU>
__device__ __forceinline__ void sf_mm_acc_bf16_split(f32x16 (&acc)[OT][NS], const f32x16 (&in)[IT][NS],
                                                     const unsigned short* __restrict__ wB, int nKStot, int nks, int lane) {
  const uint4* __restrict__ w4 = reinterpret_cast<const uint4*>(wB);
#pragma unroll
  for (int ks = 0; ks < IT * 2; ++ks) {
    if (ks < nks) {
      bf16x8 bh[NS], bl[NS];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) sf_bfrag_split<NS, IT, RELU>(in, ns, ks, bh[ns], bl[ns]);
#pragma unroll
      for (int mt = 0; mt < OT; ++mt) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, w4[((mt * nKStot + ks) * 2 + 0) * 64 + lane]);
        const bf16x8 al = __builtin_bit_cast(bf16x8, w4[((mt * nKStot + ks) * 2 + 1) * 64 + lane]);
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          acc[mt][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[ns], acc[mt][ns], 0, 0, 0);
          acc[mt][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[ns], acc[mt][ns], 0, 0, 0);
          acc[mt][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[ns], acc[mt][ns], 0, 0, 0);
        }
      }
    }
  }
}

template <int NS, int IT, bool RELU>
__device__ __forceinline__ void sf_mm_acc_bf16_tile(f32x16 (&acc)[NS], const f32x16 (&in)[IT][NS],
                                                    const unsigned short* __restrict__ wB, int nKStot, int mt,
                                                    int nks, int lane) {
  const uint4* __restrict__ w4 = reinterpret_cast<const uint4*>(wB);
#pragma unroll
  for (int ks = 0; ks < IT * 2; ++ks) {
    if (ks < nks) {
      const uint4 wv = w4[(mt * nKStot + ks) * 64 + lane];
      const bf16x8 a = __builtin_bit_cast(bf16x8, wv);
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
        acc[ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, sf_bfrag<NS, IT, RELU>(in, ns, ks), acc[ns], 0, 0, 0);
    }
  }
}

// bf16 image base of transform t (LDS copy right behind the fp32 image, or global)
// (LDS: such that base + an element offset of the bf16 image lands inside the slice staged with `part`)
template <bool LDSW>
__device__ __forceinline__ const unsigned short* sf_bf16_base(const SfDev& m, int t, float* lds, int part = 0) {
  if (LDSW) {
    int lo = 0, hi = 0, blo = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      lo = (q == part) ? m.part_off[q] : lo;
      hi = (q == part) ? m.part_off[q + 1] : hi;
      blo = (q == part) ? m.partB_off[q] : blo;
    }
    return reinterpret_cast<const unsigned short*>(lds + (hi - lo)) - blo;
  }
  return m.packedB + (size_t)t * m.tB_stride;
}

// single output tile `mt` of a multi-tile block (static mt), all other arguments as above
template <int NS, int IT, bool RELU>
__device__ __forceinline__ void sf_mm_acc_tile(f32x16 (&acc)[NS], const f32x16 (&in)[IT][NS],
                                               const float* __restrict__ wp, int nGtot, int mt, int ng,
                                               int lane) {
  const float4* __restrict__ w4 = reinterpret_cast<const float4*>(wp);
#pragma unroll
  for (int g = 0; g < IT * 4; ++g) {
    if (g < ng) {
      const float4 w = w4[(mt * nGtot + g) * 64 + lane];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        float b0 = in[g >> 2][ns][(g & 3) * 4 + 0];
        float b1 = in[g >> 2][ns][(g & 3) * 4 + 1];
        float b2 = in[g >> 2][ns][(g & 3) * 4 + 2];
        float b3 = in[g >> 2][ns][(g & 3) * 4 + 3];
        if (RELU) {
          b0 = fmaxf(b0, 0.f); b1 = fmaxf(b1, 0.f); b2 = fmaxf(b2, 0.f); b3 = fmaxf(b3, 0.f);
        }
        acc[ns] = SF_MFMA(w.x, b0, acc[ns]);
        acc[ns] = SF_MFMA(w.y, b1, acc[ns]);
        acc[ns] = SF_MFMA(w.z, b2, acc[ns]);
        acc[ns] = SF_MFMA(w.w, b3, acc[ns]);
      }
    }
  }
}

// bias image of ONE output tile mt -> acc[ns]
template <int NS>
__device__ __forceinline__ void sf_init_bias_tile(f32x16 (&acc)[NS], const float* __restrict__ bp, int mt, int h) {
  const float4* p = reinterpret_cast<const float4*>(bp + (mt * 2 + h) * 16);
  const float4 b0 = p[0], b1 = p[1], b2 = p[2], b3 = p[3];
  f32x16 v;
  v[0] = b0.x; v[1] = b0.y; v[2] = b0.z; v[3] = b0.w;
  v[4] = b1.x; v[5] = b1.y; v[6] = b1.z; v[7] = b1.w;
  v[8] = b2.x; v[9] = b2.y; v[10] = b2.z; v[11] = b2.w;
  v[12] = b3.x; v[13] = b3.y; v[14] = b3.z; v[15] = b3.w;
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) acc[ns] = v;
}

// ---- input tiles ----------------------------------------------------------------------------
// u tile: row rho = physical slot rho (D <= 16 -> registers 0..7 only)
template <int NS>
__device__ __forceinline__ void sf_build_u_tile(f32x16 (&ut)[1][NS], const float (&u)[NS][SF_DMAX], int h) {
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
#pragma unroll
    for (int r = 0; r < 8; ++r) ut[0][ns][r] = h ? u[ns][sf_row(r, 1)] : u[ns][sf_row(r, 0)];
#pragma unroll
    for (int r = 8; r < 16; ++r) ut[0][ns][r] = 0.f;
  }
}

// context tile kt: row rho = standardised context feature kt*32+rho of the lane's sample
template <int NS>
__device__ __forceinline__ void sf_build_ctx_tile(f32x16 (&ct)[1][NS], const float* const (&xr)[NS],
                                                  const SfDev& m, int kt, int h) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rho = kt * 32 + sf_row(r, h);
    const bool ok = rho < m.C;
    const int rr = ok ? rho : 0;
    const float mu = m.cst[m.c_xmean + rr], sd = m.cst[m.c_xstd + rr];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      const float v = sf_div(xr[ns][rr] - mu, sd);
      ct[0][ns][r] = ok ? v : 0.f;
    }
  }
}

// acc += Wc . e(x) over all context tiles.  pre: the first standardised context tile built once by the caller
// (hoisted out of the transform loop), or nullptr to build it here.
template <int OT, int NS>
__device__ __forceinline__ void sf_ctx_mm(f32x16 (&acc)[OT][NS], const float* const (&xr)[NS],
                                          const SfDev& m, const float* __restrict__ wp, int lane,
                                          const f32x16 (*pre)[1][NS] = nullptr) {
  for (int kt = 0; kt * 4 < m.nGc; ++kt) {
    const int ng = min(4, m.nGc - kt * 4);
    if (kt == 0 && pre) {
      sf_mm_acc<OT, NS, 1, false>(acc, *pre, wp, m.nGc, 0, ng, lane);
    } else {
      f32x16 ct[1][NS];
      sf_build_ctx_tile<NS>(ct, xr, m, kt, lane >> 5);
      sf_mm_acc<OT, NS, 1, false>(acc, ct, wp, m.nGc, kt * 4, ng, lane);
    }
  }
}
