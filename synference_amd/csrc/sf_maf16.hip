// sf_maf16.hip -- incremental MAF inverse / sampler on v_mfma_f32_16x16x4_f32 (16-row granularity).
//
// Same algorithm and the same C-ABI entry points as k_inverse<MafOps> (sf_flows.h, sf_inst_templates.h); only
// the tile geometry differs so that a 12-13 unit MADE degree group costs one 16-row tile instead of a 32-row one:
//   lane l: sample s = l & 15 of a 16-sample tile, row group g4 = l >> 4; a tile is 4 VGPRs, a[r] = row 4*g4 + r
//   C/D layout of the 16x16x4 MFMA == B-operand order of the next layer (k-step r <-> register r), so the
//   activations stay in registers exactly as in the 32-row engine.
//   weights: float4[(ot*IT + it)*64 + l] = W[ot*16 + (l&15)][it*16 + 4*(l>>4) + r], r = 0..3 (sf_layout.cpp)
// One wave = one tile of 16 draws, 4 waves per workgroup; the transform's 16-row image is staged in LDS.  The
// draw itself also lives in tile layout (lane (s, g4) owns physical slots 4*g4..4*g4+3), so Philox, the box test
// and the output writes are split over the four row groups instead of being repeated by them.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "sf_device.h"
#include "sf_internal.h"
#include "sf_rng.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define SF_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float sf_sum4groups(float v) {  // sum over the 4 row groups of a sample
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ f32x4 sf_mma16(float4 w, const f32x4& in, f32x4 acc) {
  acc = SF_MFMA16(w.x, in[0], acc);
  acc = SF_MFMA16(w.y, in[1], acc);
  acc = SF_MFMA16(w.z, in[2], acc);
  acc = SF_MFMA16(w.w, in[3], acc);
  return acc;
}
// head rows of one tile for this lane: (a0,m0,a1,m1), (a2,m2,a3,m3); acc = (sum a_r v_r, sum m_r v_r) as packed FMAs
__device__ __forceinline__ f32x2 sf_head_acc(f32x2 acc, const float4& h01, const float4& h23, const f32x4& v) {
  acc += f32x2{h01.x, h01.y} * f32x2{v[0], v[0]};
  acc += f32x2{h01.z, h01.w} * f32x2{v[1], v[1]};
  acc += f32x2{h23.x, h23.y} * f32x2{v[2], v[2]};
  acc += f32x2{h23.z, h23.w} * f32x2{v[3], v[3]};
  return acc;
}
__device__ __forceinline__ f32x4 sf_ld4(const float* p) {
  const float4 b = *reinterpret_cast<const float4*>(p);
  f32x4 r;
  r[0] = b.x; r[1] = b.y; r[2] = b.z; r[3] = b.w;
  return r;
}
// tanh of a tile's four values with the plain arithmetic on packed-f32 instructions (v_pk_add / v_pk_fma: two
// values per instruction at the full rate) -- the kernel is bound by vector ISSUE, and the four exp2 / four rcp cannot be
// packed.
// The argument is PRE-SCALED: the packer multiplies the hidden blocks' weights and biases of the 16-row images by
// 2 log2(e) (SF_PACK_TANH_SCALE, sf_layout.h), so tanh(x) = 1 - 2 / (1 + 2^b) with b = 2 log2(e) x starts at the exp2.
__device__ __forceinline__ float sf_tanh_pre(float b) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(b));
}
__device__ __forceinline__ f32x4 sf_tanh4(const f32x4& b) {
  const f32x2 one = {1.0f, 1.0f}, m2 = {-2.0f, -2.0f};
  const f32x2 e01 = f32x2{__builtin_amdgcn_exp2f(b[0]), __builtin_amdgcn_exp2f(b[1])} + one;
  const f32x2 e23 = f32x2{__builtin_amdgcn_exp2f(b[2]), __builtin_amdgcn_exp2f(b[3])} + one;
  const f32x2 r01 = {__builtin_amdgcn_rcpf(e01[0]), __builtin_amdgcn_rcpf(e01[1])};
  const f32x2 r23 = {__builtin_amdgcn_rcpf(e23[0]), __builtin_amdgcn_rcpf(e23[1])};
  const f32x2 o01 = __builtin_elementwise_fma(r01, m2, one), o23 = __builtin_elementwise_fma(r23, m2, one);
  return f32x4{o01[0], o01[1], o23[0], o23[1]};
}
// weight fragment of (out tile ot, in tile it) of a block with IT input tiles
__device__ __forceinline__ float4 sf_w16(const float* wp, int IT, int ot, int it, int lane) {
  return reinterpret_cast<const float4*>(wp)[(ot * IT + it) * 64 + lane];
}

extern __shared__ float sf_lds16[];

struct SfPass16 {
  f32x4 act[3][4];  // act[0] = initial layer, act[k+1] = output of block k; [tile]
  f32x4 c0[4];      // b0 + bc + Wc e(x), per tile
  f32x4 ut;         // finished dimensions of this transform, tile layout: slot 4*g4 + r
  float ldl;
};

// One autoregressive pass with the degree group in (static) tile OT: recompute that tile of every hidden
// layer from the finished dimensions, then the (a, m) head rows of physical slot sl as per-lane dot products.
template <int OT, int NB, bool LD = true>
__device__ __forceinline__ void sf_pass16(const SfDev& m, const float* tp, SfPass16& S, int NT, int sl, float u_sl,
                                          int lane, int g4) {
  // everything that does not depend on this pass's new dimension first: weight fragments, partial sums
  const float* hv = tp + m.o16_hv + sl * 128 + g4 * 32;
  float4 w0 = sf_w16(tp + m.o16_w0, 1, OT, 0, lane);
  float4 wk[2][OT + 1];
  f32x4 bk[2];
  float4 h01[OT + 1], h23[OT + 1];
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    bk[k] = sf_ld4(tp + m.o16_bk[k] + (OT * 4 + g4) * 4);
#pragma unroll
    for (int it = 0; it <= OT; ++it) wk[k][it] = sf_w16(tp + m.o16_wk[k], NT, OT, it, lane);
  }
#pragma unroll
  for (int tl = 0; tl <= OT; ++tl) {
    h01[tl] = *reinterpret_cast<const float4*>(hv + tl * 8);
    h23[tl] = *reinterpret_cast<const float4*>(hv + tl * 8 + 4);
  }
  const float ba = tp[m.o16_hvb + 2 * sl], bm = tp[m.o16_hvb + 2 * sl + 1];
  f32x2 pam = {0.f, 0.f};
#pragma unroll
  for (int tl = 0; tl < OT; ++tl) pam = sf_head_acc(pam, h01[tl], h23[tl], S.act[NB][tl]);
#pragma unroll
  for (int k = 0; k < NB; ++k)
#pragma unroll
    for (int it = 0; it < OT; ++it) bk[k] = sf_mma16(wk[k][it], S.act[k][it], bk[k]);
  // the dependent chain
  S.act[0][OT] = sf_mma16(w0, S.ut, S.c0[OT]);
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const f32x4 b = sf_mma16(wk[k][OT], S.act[k][OT], bk[k]);
#pragma unroll
    for (int r = 0; r < 4; ++r) S.act[k + 1][OT][r] = sf_tanh_pre(b[r]);
  }
  pam = sf_head_acc(pam, h01[OT], h23[OT], S.act[NB][OT]);
  const float av = ba + sf_sum4groups(pam[0]);
  const float mv = bm + sf_sum4groups(pam[1]);
  const float sc = (m.scale_fn == 0 ? sf_softplus(av) : sf_sigmoid(av + 2.0f)) + m.eps;
  const float wv = sf_div(u_sl - mv, sc);
  if (LD) S.ldl += sf_log(sc);  // (the sampler does not need the log-determinant)
#pragma unroll
  for (int r = 0; r < 4; ++r) S.ut[r] = (g4 == (sl >> 2) && r == (sl & 3)) ? wv : S.ut[r];
}

// Same pass when the degree group straddles tiles LO..HI (contiguous packing, sf_layout.cpp): every layer is
// recomputed for all of those tiles before the next layer starts (units of one degree feed each other), over the
// input tiles 0..HI; rows of later groups inside these tiles get provisional values that nothing unmasked reads
// and that their own pass overwrites.
template <int LO, int HI, int NB, bool LD = true>
__device__ __forceinline__ void sf_pass16_span(const SfDev& m, const float* tp, SfPass16& S, int NT, int sl, float u_sl,
                                               int lane, int g4) {
  const float* hv = tp + m.o16_hv + sl * 128 + g4 * 32;
#pragma unroll
  for (int ot = LO; ot <= HI; ++ot) S.act[0][ot] = sf_mma16(sf_w16(tp + m.o16_w0, 1, ot, 0, lane), S.ut, S.c0[ot]);
#pragma unroll
  for (int k = 0; k < NB; ++k) {
#pragma unroll
    for (int ot = LO; ot <= HI; ++ot) {
      f32x4 b = sf_ld4(tp + m.o16_bk[k] + (ot * 4 + g4) * 4);
#pragma unroll
      for (int it = 0; it <= HI; ++it) b = sf_mma16(sf_w16(tp + m.o16_wk[k], NT, ot, it, lane), S.act[k][it], b);
#pragma unroll
      for (int r = 0; r < 4; ++r) S.act[k + 1][ot][r] = sf_tanh_pre(b[r]);
    }
  }
  f32x2 pam = {0.f, 0.f};
#pragma unroll
  for (int tl = 0; tl <= HI; ++tl)
    pam = sf_head_acc(pam, *reinterpret_cast<const float4*>(hv + tl * 8), *reinterpret_cast<const float4*>(hv + tl * 8 + 4),
                      S.act[NB][tl]);
  const float av = tp[m.o16_hvb + 2 * sl] + sf_sum4groups(pam[0]);
  const float mv = tp[m.o16_hvb + 2 * sl + 1] + sf_sum4groups(pam[1]);
  const float sc = (m.scale_fn == 0 ? sf_softplus(av) : sf_sigmoid(av + 2.0f)) + m.eps;
  const float wv = sf_div(u_sl - mv, sc);
  if (LD) S.ldl += sf_log(sc);  // (the sampler does not need the log-determinant)
#pragma unroll
  for (int r = 0; r < 4; ++r) S.ut[r] = (g4 == (sl >> 2) && r == (sl & 3)) ? wv : S.ut[r];
}

// value of physical slot sl from a tile-layout register quad, broadcast to every row group
__device__ __forceinline__ float sf_slot16(const f32x4& t, int sl, int lane) {
  const int r = sl & 3;
  const float v = r == 0 ? t[0] : (r == 1 ? t[1] : (r == 2 ? t[2] : t[3]));
  return __shfl(v, (lane & 15) + 16 * (sl >> 2), 64);
}

// the same value where it is needed only in the row group that owns the slot (lanes of group sl >> 2): no cross-lane move
__device__ __forceinline__ float sf_slot16_own(const f32x4& t, int sl) {
  const int r = sl & 3;
  return r == 0 ? t[0] : (r == 1 ? t[1] : (r == 2 ? t[2] : t[3]));
}

template <int NB, bool SPAN>
__global__ __launch_bounds__(256, 3) void k_maf_inv16(SfDev m, SfSampleArgsHost a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = lane & 15, g4 = lane >> 4;
  // (no early exit: every wave takes part in the staging barriers)
  const long item = ((long)blockIdx.x * 4 + wave) * 16 + s;
  const bool valid = item < a.n_items;
  const long it = valid ? item : a.n_items - 1;
  // tile layout: this lane holds physical slots 4*g4 .. 4*g4+3 of sample s
  f32x4 u;
  uint64_t slot;
  long gal, ps_idx = 0;
  uint32_t att_mine = 0;
  if (a.z_in) {
    slot = (uint64_t)it;
    gal = it;
#pragma unroll
    for (int r = 0; r < 4; ++r) u[r] = (4 * g4 + r < m.D) ? a.z_in[it * m.D + 4 * g4 + r] : 0.f;
  } else {
    const long ps = it >> a.log2_attempts;  // listed slot; A (a power of two) consecutive items share it
    ps_idx = ps;
    slot = a.slots ? (uint64_t)a.slots[ps] : (uint64_t)(a.slot_base + ps);
    gal = (long)((uint32_t)slot / (uint32_t)a.S);  // slot ids fit 32 bits (checked by the API)
    const uint32_t att = a.att_list ? a.att_list[ps] : a.attempt + (uint32_t)(it & ((1L << a.log2_attempts) - 1));
    att_mine = att;
    float z4[4];
    sf_normal4(a.k0, a.k1, slot + a.rng_slot_offset, att, (uint32_t)g4, z4);  // Philox block g4 = dimensions 4*g4 .. 4*g4+3
#pragma unroll
    for (int r = 0; r < 4; ++r) u[r] = (4 * g4 + r < m.D) ? z4[r] : 0.f;
  }
  const float* xr = a.x + gal * m.C;
  float logdet = 0.f;

  // standardised context, tile ic: rows 16*ic + 4*g4 + r  (tile 0 is kept in registers across the transforms)
  auto ctx_tile = [&](int ic) {
    f32x4 ct;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rho = ic * 16 + 4 * g4 + r;
      const bool ok = rho < m.C;
      const int rr = ok ? rho : 0;
      const float v = sf_div(xr[rr] - m.cst[m.c_xmean + rr], m.cst[m.c_xstd + rr]);
      ct[r] = ok ? v : 0.f;
    }
    return ct;
  };
  const float* ctg = (m.ctab && !a.z_in) ? m.ctab + (size_t)gal * m.T * m.ctab_R : nullptr;  // wave-uniform choice
  f32x4 ct0;
  if (!ctg) ct0 = ctx_tile(0);

  const int NT = m.nT16;
  // activation tiles of the three layers: a pass only reads tiles that an earlier pass of the SAME transform has
  // written (tile index <= its own), so they are cleared once, not per transform
  SfPass16 S;
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int ot = 0; ot < 4; ++ot)
#pragma unroll
      for (int r = 0; r < 4; ++r) S.act[k][ot][r] = 0.f;
  uint32_t tile_bits = 0, lo_bits = 0;  // g16_tile / g16_lo packed 2 bits per degree (static indexing keeps them in SGPRs)
#pragma unroll
  for (int q = 0; q < SF_DMAX; ++q) {
    tile_bits |= (uint32_t)(m.g16_tile[q] & 3) << (2 * q);
    lo_bits |= (uint32_t)(m.g16_lo[q] & 3) << (2 * q);
  }
  for (int t = m.T - 1; t >= 0; --t) {
    // ---- stage this transform's 16-row image (direct-to-LDS loads)
    __syncthreads();
    {
      const float4* __restrict__ s4 = reinterpret_cast<const float4*>(m.packed16 + (size_t)t * m.t16_stride);
      float4* __restrict__ d4 = reinterpret_cast<float4*>(sf_lds16);
      const int n4 = m.t16_stride >> 2;
      // direct global -> LDS copies (global_load_lds_dwordx4): no staging registers, no ds_write; the LDS
      // destination of a wave-instruction is its (wave-uniform) base + lane * 16 bytes.  The image is padded to whole
      // 4 KiB groups (sf_layout.cpp): a wave copies a group with ONE address and four immediate offsets
      // (the immediate applies to the global and to the LDS address alike).
      const int lane_ = threadIdx.x & 63;
      const int ngroups = n4 >> 8;
      for (int gi = __builtin_amdgcn_readfirstlane(wave); gi < ngroups; gi += 4) {
        const float4* g = s4 + gi * 256 + lane_;
        float4* l = d4 + gi * 256;
        __builtin_amdgcn_global_load_lds((const void*)g, (void __attribute__((address_space(3)))*)l, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)g, (void __attribute__((address_space(3)))*)l, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds((const void*)g, (void __attribute__((address_space(3)))*)l, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds((const void*)g, (void __attribute__((address_space(3)))*)l, 16, 3072, 0);
      }
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0), other counters untouched: the copies have landed
    }
    __syncthreads();
    const float* tp = sf_lds16;
    // context product hoisted out of the passes: c0 = b0 + bc + Wc e
    if (ctg) {  // per-galaxy table (sf_flow_prepare_context): same values, computed once per galaxy
#pragma unroll
      for (int ot = 0; ot < 4; ++ot)
        if (ot < NT) S.c0[ot] = *reinterpret_cast<const f32x4*>(ctg + (size_t)t * m.ctab_R + ot * 16 + 4 * g4);
    } else {
#pragma unroll
      for (int ot = 0; ot < 4; ++ot)
        if (ot < NT) {
          S.c0[ot] = sf_ld4(tp + m.o16_b0 + (ot * 4 + g4) * 4);
          S.c0[ot] = sf_mma16(sf_w16(tp + m.o16_wc, m.nC16, ot, 0, lane), ct0, S.c0[ot]);
        }
      for (int ic = 1; ic < m.nC16; ++ic) {
        const f32x4 ct = ctx_tile(ic);
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
          if (ot < NT) S.c0[ot] = sf_mma16(sf_w16(tp + m.o16_wc, m.nC16, ot, ic, lane), ct, S.c0[ot]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) S.ut[r] = 0.f;
    S.ldl = 0.f;
    // physical slot of the dimension with MADE degree p: lane q holds entry q, read back with v_readlane
    const int dsl = (int)m.cst[m.c_dslot + t * SF_DMAX + s];
    {
      // pass 1: the dimension of degree 1 depends on the context only (head bias)
      const int sl = __builtin_amdgcn_readlane(dsl, 0);
      const float av = tp[m.o16_hvb + 2 * sl], mv = tp[m.o16_hvb + 2 * sl + 1];
      const float sc = (m.scale_fn == 0 ? sf_softplus(av) : sf_sigmoid(av + 2.0f)) + m.eps;
      const float wv = sf_div(sf_slot16(u, sl, lane) - mv, sc);
      S.ldl += sf_log(sc);
#pragma unroll
      for (int r = 0; r < 4; ++r) S.ut[r] = (g4 == (sl >> 2) && r == (sl & 3)) ? wv : S.ut[r];
    }
    for (int p = 2; p <= m.D; ++p) {
      const int sl = __builtin_amdgcn_readlane(dsl, p - 1);
      const float u_sl = sf_slot16(u, sl, lane);
      const uint32_t hi_t = (tile_bits >> (2 * (p - 1))) & 3u;
      const uint32_t lo_t = SPAN ? (lo_bits >> (2 * (p - 1))) & 3u : hi_t;  // (the aligned-packing kernel has no span code)
      switch (lo_t * 4 + hi_t) {
        case 0: sf_pass16<0, NB>(m, tp, S, NT, sl, u_sl, lane, g4); break;
        case 5: sf_pass16<1, NB>(m, tp, S, NT, sl, u_sl, lane, g4); break;
        case 10: sf_pass16<2, NB>(m, tp, S, NT, sl, u_sl, lane, g4); break;
        case 15: sf_pass16<3, NB>(m, tp, S, NT, sl, u_sl, lane, g4); break;
        // degree groups that straddle tiles (contiguous packing)
        case 1: if (SPAN) sf_pass16_span<0, 1, NB>(m, tp, S, NT, sl, u_sl, lane, g4); break;
        case 2: if (SPAN) sf_pass16_span<0, 2, NB>(m, tp, S, NT, sl, u_sl, lane, g4); break;
        case 3: if (SPAN) sf_pass16_span<0, 3, NB>(m, tp, S, NT, sl, u_sl, lane, g4); break;
        case 6: if (SPAN) sf_pass16_span<1, 2, NB>(m, tp, S, NT, sl, u_sl, lane, g4); break;
        case 7: if (SPAN) sf_pass16_span<1, 3, NB>(m, tp, S, NT, sl, u_sl, lane, g4); break;
        default: if (SPAN) sf_pass16_span<2, 3, NB>(m, tp, S, NT, sl, u_sl, lane, g4); break;
      }
    }
    logdet -= S.ldl;
    u = S.ut;
  }

  // ---------------------------------------------------------------- un-standardise, box test, outputs
  // each lane owns 4 physical slots of its sample; a draw is accepted when all 4 row groups agree
  float th[4];
  bool ok = true;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int p = 4 * g4 + r;
    th[r] = 0.f;
    if (p < m.D) {
      const int td = (int)m.cst[m.c_tdim + p];
      th[r] = sf_div(u[r] - m.cst[m.c_pshift + p], m.cst[m.c_pscale + p]);
      ok = ok && (fabsf(th[r]) <= 3.0e38f);  // finite (NaN compares false)
      if (a.lo) ok = ok && (th[r] >= a.lo[td]) && (th[r] <= a.hi[td]);
    }
  }
  if (a.att_list && att_mine == 0xffffffffu) ok = false;  // no attempt to resolve: straight to the rejected list
  const unsigned long long okb = __ballot(ok);
  const uint32_t acc16 = (uint32_t)(okb & (okb >> 16) & (okb >> 32) & (okb >> 48) & 0xffffull) &
                         (uint32_t)(__ballot(valid) & 0xffffull);
  const bool accepted = (acc16 >> s) & 1u;
  if (a.z_in) {
    if (valid) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * g4 + r < m.D) a.out[item * m.D + (int)m.cst[m.c_tdim + 4 * g4 + r]] = th[r];
      if (a.logdet_out && g4 == 0) a.logdet_out[item] = logdet - m.logdet0;
    }
  } else if (a.best) {
    if (accepted && g4 == 0) atomicMin(&a.best[ps_idx], att_mine);
  } else if (a.count) {
    const long g_first = __shfl(gal, 0, 64);
    const long g_last = __shfl(gal, 15, 64);
    if (g_first == g_last) {
      if (lane == 0 && acc16) atomicAdd(&a.count[g_first], (int)__popc(acc16));
    } else if (accepted && g4 == 0) {
      atomicAdd(&a.count[gal], 1);
    }
  } else {
    // A <= 16 consecutive items hold attempts att..att+A-1 of one slot: the lowest accepted one wins
    const int A = a.attempts_per_slot;
    const int grp0 = (s / A) * A;
    const uint32_t gmask = (acc16 >> grp0) & ((1u << A) - 1u);
    const int first = gmask ? (int)__builtin_ctz(gmask) : -1;
    const int me = s - grp0;
    if (valid) {
      if (g4 == 0 && me == 0 && a.n_drawn && a.attempt > 0) sf_sat_add(&a.n_drawn[gal], first >= 0 ? first + 1 : A);
      if (me == first) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * g4 + r < m.D) sf_out_store(a, (size_t)slot * m.D + (int)m.cst[m.c_tdim + 4 * g4 + r], th[r]);
      } else if (first < 0 && me == 0 && g4 == 0) {
        const uint32_t pos = atomicAdd(a.n_rejected, 1u);
        a.rejected[pos] = (uint32_t)slot;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Split-bf16 hidden blocks (k_maf_samp16 only).  The H x H blocks carry ~2/3 of a pass's MACs; on fp32 MFMA
// (v_mfma_f32_16x16x4_f32: 32 cycles for K = 4) they take as long as the same MACs on the vector pipe.  Here every
// operand is split into hi = bf16(v) and lo = bf16(v - hi) and the product is hi.hi + hi.lo + lo.hi on
// v_mfma_f32_16x16x32_bf16 (16 cycles for K = 32, fp32 accumulation): ~2^-17 relative per product -- the draws still
// meet the oracle at the fp32 tolerance of the parity tests -- at a fifth of the matrix-pipe time.
// Operand order: two 16-row activation tiles (4 registers per lane each, lane = sample + 16 * row group) ARE the B
// operand of one K = 32 step: element j of lane l is row 16*(2*pair + (j>>2)) + 4*(l>>4) + (j&3); the weight image
// (sf_layout.cpp, src16B) stores the A operand in the same k order, hi and lo parts as separate 16-byte fragments.
// ---------------------------------------------------------------------------------------------------------------
typedef __bf16 sf_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sf_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define SF_MFMA16B(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

struct SfSplit2 {  // two activation values' worth of split operands for one 16-row tile: hi/lo packed bf16 pairs
  unsigned int hi[2], lo[2];
};
__device__ __forceinline__ SfSplit2 sf_split16(const f32x4& v) {
  SfSplit2 t;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    // ONE packed conversion per pair (element-wise casts make the compiler convert the pair once packed and its first
    // element once more on its own); round-to-nearest-even, the same values
    const f32x2 pv = {v[2 * q], v[2 * q + 1]};
    const unsigned int hw = __builtin_bit_cast(unsigned int, __builtin_convertvector(pv, sf_bf16x2));
    const f32x2 rv = {v[2 * q] - __builtin_bit_cast(float, hw << 16), v[2 * q + 1] - __builtin_bit_cast(float, hw & 0xffff0000u)};
    t.hi[q] = hw;
    t.lo[q] = __builtin_bit_cast(unsigned int, __builtin_convertvector(rv, sf_bf16x2));
  }
  return t;
}
// acc += W[ot, pair] . (the pair's two tiles)   (three bf16 products; bh / bl ARE the B operands, no moves)
__device__ __forceinline__ f32x4 sf_mma16x3(const u32x4& w_hi, const u32x4& w_lo, const u32x4& bh, const u32x4& bl, f32x4 acc) {
  const sf_bf16x8 Ah = __builtin_bit_cast(sf_bf16x8, w_hi), Al = __builtin_bit_cast(sf_bf16x8, w_lo);
  const sf_bf16x8 Bh = __builtin_bit_cast(sf_bf16x8, bh), Bl = __builtin_bit_cast(sf_bf16x8, bl);
  acc = SF_MFMA16B(Al, Bh, acc);
  acc = SF_MFMA16B(Ah, Bl, acc);
  acc = SF_MFMA16B(Ah, Bh, acc);
  return acc;
}
// fragment of block k: (out tile ot, in-tile pair pr, part 0 = hi / 1 = lo); wB = base of the block in 32-bit words
// CP: aligned placement stores only the pairs a tile can read (sf_layout.cpp): entry ot + (ot == 3) + pr
template <bool CP>
__device__ __forceinline__ u32x4 sf_w16b(const unsigned int* wB, int NP, int ot, int pr, int part, int lane) {
  const int e = CP ? ot + (ot == 3 ? 1 : 0) + pr : ot * NP + pr;
  return reinterpret_cast<const u32x4*>(wB)[(e * 2 + part) * 64 + lane];
}

struct SfPass16B {
  // split inputs of hidden block k ([0] = initial layer, [1] = output of block 0), held per PAIR of tiles exactly as
  // the MFMA wants its B operand: components 0,1 = tile 2p (rows 0,1 | rows 2,3), components 2,3 = tile 2p+1
  u32x4 ph[2][2], pl[2][2];
  f32x4 head[4];       // output of the last block (fp32: the head rows are per-lane dot products); [tile]   (HM = false)
  f32x4 hdone;         // HM: head biases + the head rows' products with every FINISHED hidden tile, as one MFMA output
                       // tile: lane (s, g4), register r = row 4*g4 + r = (a | m) of physical slot (4*g4 + r) >> 1
  f32x4 ut;            // finished dimensions of this transform, tile layout: slot 4*g4 + r
  const float* c0p;    // this draw's context-table row for the transform (b0 + bc + Wc e(x), tile order), or nullptr
  const float* xr;     // the draw's context row (no-table path: c0 is evaluated where it is needed)
  f32x4 c0n;           // table path, aligned placement: c0 of the NEXT pass's tile, requested one pass ahead so that the
                       // L2 round trip is over before the pass that starts its dependent chain with it
  bool tab;            // wave-uniform: the table exists (c0p is per lane, the decision is not)
};
// The same state with the hidden blocks' inputs in fp32 (PREC = 1: v_mfma_f32_16x16x4_f32 everywhere, see sf_pass16f)
struct SfPass16F {
  f32x4 act[2][4];     // inputs of hidden block k ([0] = initial layer, [1] = output of block 0); [tile]: C/D layout of the MFMA
                       // that wrote them = B-operand order of the one that reads them
  f32x4 head[4];       // output of the last block; [tile]   (HM = false)
  f32x4 hdone;         // HM: see SfPass16B
  f32x4 ut;
  const float* c0p;
  const float* xr;
  f32x4 c0n;
  bool tab;
};
// request c0 of tile `ot` of the current transform (table path only)
template <typename ST>
__device__ __forceinline__ void sf_c0_prefetch(ST& S, int ot, int g4) {
  S.c0n = *reinterpret_cast<const f32x4*>(S.c0p + ot * 16 + 4 * g4);
}
// c0 of tile ot = b0 + bc + Wc e(x): from the per-galaxy table, else evaluated on the spot (rare: tables above the
// size cap); either way it is not kept in registers across the passes
template <typename ST>
__device__ __forceinline__ f32x4 sf_c0_16(const SfDev& m, const float* tp, const ST& S, int ot, int lane, int g4) {
  if (S.c0p) return *reinterpret_cast<const f32x4*>(S.c0p + ot * 16 + 4 * g4);
  f32x4 c = sf_ld4(tp + m.o16_b0 + (ot * 4 + g4) * 4);
  for (int ic = 0; ic < m.nC16; ++ic) {
    f32x4 ct;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rho = ic * 16 + 4 * g4 + r;
      const bool ok = rho < m.C;
      const int rr = ok ? rho : 0;
      const float v = sf_div(S.xr[rr] - m.cst[m.c_xmean + rr], m.cst[m.c_xstd + rr]);
      ct[r] = ok ? v : 0.f;
    }
    c = sf_mma16(sf_w16(tp + m.o16_wc, m.nC16, ot, ic, lane), ct, c);
  }
  return c;
}
template <int TILE>
__device__ __forceinline__ void sf_put16b(SfPass16B& S, int k, const f32x4& v) {  // k is a static loop index at every call
  const SfSplit2 t = sf_split16(v);
  S.ph[k][TILE >> 1][(TILE & 1) * 2] = t.hi[0];
  S.ph[k][TILE >> 1][(TILE & 1) * 2 + 1] = t.hi[1];
  S.pl[k][TILE >> 1][(TILE & 1) * 2] = t.lo[0];
  S.pl[k][TILE >> 1][(TILE & 1) * 2 + 1] = t.lo[1];
}

// One autoregressive pass with the degree group in (static) tile OT (see sf_pass16); hidden blocks on split bf16.
// HM (aligned placement, D <= 8): the head rows of ALL slots are one 16-row MFMA output tile (sf_layout.cpp, o16_wh); a
// pass adds its tile's product to the running tile S.hdone and reads its own (a, m) out of the result -- 4 MFMAs
// instead of 4 (OT + 1) packed FMAs, 2 (OT + 1) LDS reads and a two-step cross-row-group sum, and 4 registers of state
// instead of 16.
template <int OT, int NB, bool CP, bool HM = false>
__device__ __forceinline__ void sf_pass16b(const SfDev& m, const float* tp, const unsigned int* tpB, SfPass16B& S, int NT, int sl,
                                           float u_sl, int lane, int g4, int next_ot = -1) {
  constexpr int PR = OT >> 1;  // the pair that holds tile OT; pairs below it are complete
  const int NP = m.nP16;
  f32x4 c0;
  if (CP && HM && S.tab) {  // (aligned placement: one tile per pass, so the caller knows the next one; HM: the registers for it)
    c0 = S.c0n;
    if (next_ot >= 0) sf_c0_prefetch(S, next_ot, g4);
  } else {
    c0 = sf_c0_16(m, tp, S, OT, lane, g4);
  }
  const float* hv = tp + m.o16_hv + sl * 128 + g4 * 32;
  const float4 w0 = sf_w16(tp + m.o16_w0, 1, OT, 0, lane);
  // head rows of the tiles finished in earlier passes: partial sums first (nothing here depends on this pass)
  f32x2 pam = {0.f, 0.f};
  if (!HM) {
#pragma unroll
    for (int tl = 0; tl < OT; ++tl)
      pam = sf_head_acc(pam, *reinterpret_cast<const float4*>(hv + tl * 8), *reinterpret_cast<const float4*>(hv + tl * 8 + 4), S.head[tl]);
  }
  f32x4 last;
  float4 wh;
  if (HM) {
    // operand fragments are requested one stage ahead of the MFMAs that read them (block 0 under the initial layer's
    // chain, block k + 1 / the head rows under block k's tanh and split), so that an LDS round trip per fragment does
    // not sit in the dependent chain; the scheduling barriers pin that order
    u32x4 fh[PR + 1], fl[PR + 1];
    f32x4 b = sf_ld4(tp + m.o16_bk[0] + (OT * 4 + g4) * 4);
#pragma unroll
    for (int pr = 0; pr <= PR; ++pr) {
      fh[pr] = sf_w16b<CP>(tpB + m.o16B_wk[0], NP, OT, pr, 0, lane);
      fl[pr] = sf_w16b<CP>(tpB + m.o16B_wk[0], NP, OT, pr, 1, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    sf_put16b<OT>(S, 0, sf_mma16(w0, S.ut, c0));
#pragma unroll
    for (int k = 0; k < NB; ++k) {
#pragma unroll
      for (int pr = 0; pr <= PR; ++pr) b = sf_mma16x3(fh[pr], fl[pr], S.ph[k][pr], S.pl[k][pr], b);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 bn;
      if (k + 1 < NB) {
        bn = sf_ld4(tp + m.o16_bk[k + 1 < NB ? k + 1 : k] + (OT * 4 + g4) * 4);
#pragma unroll
        for (int pr = 0; pr <= PR; ++pr) {
          fh[pr] = sf_w16b<CP>(tpB + m.o16B_wk[k + 1 < NB ? k + 1 : k], NP, OT, pr, 0, lane);
          fl[pr] = sf_w16b<CP>(tpB + m.o16B_wk[k + 1 < NB ? k + 1 : k], NP, OT, pr, 1, lane);
        }
      } else {
        wh = sf_w16(tp + m.o16_wh, NT, 0, OT, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 th = sf_tanh4(b);
      if (k + 1 < NB) { sf_put16b<OT>(S, k + 1, th); b = bn; }
      else last = th;
    }
  } else {
  // initial layer of tile OT (the start of the dependent chain)
  sf_put16b<OT>(S, 0, sf_mma16(w0, S.ut, c0));
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    __builtin_amdgcn_sched_barrier(0);  // keep one block's fragments in flight at a time (registers)
    f32x4 b = sf_ld4(tp + m.o16_bk[k] + (OT * 4 + g4) * 4);
    // complete pairs, then the pair of tile OT: its other tile is either final (OT odd) or one that masked weights
    // never read (OT even)
#pragma unroll
    for (int pr = 0; pr <= PR; ++pr)
      b = sf_mma16x3(sf_w16b<CP>(tpB + m.o16B_wk[k], NP, OT, pr, 0, lane), sf_w16b<CP>(tpB + m.o16B_wk[k], NP, OT, pr, 1, lane),
                     S.ph[k][pr], S.pl[k][pr], b);
    f32x4 th;
#pragma unroll
    for (int r = 0; r < 4; ++r) th[r] = sf_tanh_pre(b[r]);
    if (k + 1 < NB) sf_put16b<OT>(S, k + 1, th);
    else S.head[OT] = th;
  }
  }
  float av, mv;
  if (HM) {
    const f32x4 fresh = sf_mma16(wh, last, S.hdone);
    // the tile is final once the next pass works on another one (several small degree groups may share a tile: each of
    // their passes recomputes it, the running tile takes it once)
    if (next_ot != OT) S.hdone = fresh;
    const bool odd = (sl & 1) != 0;  // rows 2 sl, 2 sl + 1 sit in row group sl >> 1, registers 0,1 or 2,3
    const int src = (lane & 15) + 16 * (sl >> 1);
    av = __shfl(odd ? fresh[2] : fresh[0], src, 64);
    mv = __shfl(odd ? fresh[3] : fresh[1], src, 64);
  } else {
    pam = sf_head_acc(pam, *reinterpret_cast<const float4*>(hv + OT * 8), *reinterpret_cast<const float4*>(hv + OT * 8 + 4), S.head[OT]);
    av = tp[m.o16_hvb + 2 * sl] + sf_sum4groups(pam[0]);
    mv = tp[m.o16_hvb + 2 * sl + 1] + sf_sum4groups(pam[1]);
  }
  const float sc = (m.scale_fn == 0 ? sf_softplus(av) : sf_sigmoid(av + 2.0f)) + m.eps;
  const float wv = sf_div(u_sl - mv, sc);
#pragma unroll
  for (int r = 0; r < 4; ++r) S.ut[r] = (g4 == (sl >> 2) && r == (sl & 3)) ? wv : S.ut[r];
}

// The same pass when the degree group straddles tiles LO..HI (contiguous packing; see sf_pass16_span).
template <int LO, int HI, int NB>
__device__ __forceinline__ void sf_pass16b_span(const SfDev& m, const float* tp, const unsigned int* tpB, SfPass16B& S, int NT,
                                                int sl, float u_sl, int lane, int g4) {
  constexpr int PH = HI >> 1;
  const int NP = m.nP16;
  const float* hv = tp + m.o16_hv + sl * 128 + g4 * 32;
  auto put = [&](int k, int ot, const f32x4& v) {  // (k, ot are unrolled loop indices: static after unrolling)
    const SfSplit2 t = sf_split16(v);
    S.ph[k][ot >> 1][(ot & 1) * 2] = t.hi[0];
    S.ph[k][ot >> 1][(ot & 1) * 2 + 1] = t.hi[1];
    S.pl[k][ot >> 1][(ot & 1) * 2] = t.lo[0];
    S.pl[k][ot >> 1][(ot & 1) * 2 + 1] = t.lo[1];
  };
#pragma unroll
  for (int ot = LO; ot <= HI; ++ot) put(0, ot, sf_mma16(sf_w16(tp + m.o16_w0, 1, ot, 0, lane), S.ut, sf_c0_16(m, tp, S, ot, lane, g4)));
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    f32x4 nb[HI - LO + 1];
#pragma unroll
    for (int ot = LO; ot <= HI; ++ot) {
      f32x4 b = sf_ld4(tp + m.o16_bk[k] + (ot * 4 + g4) * 4);
#pragma unroll
      for (int pr = 0; pr <= PH; ++pr)
        b = sf_mma16x3(sf_w16b<false>(tpB + m.o16B_wk[k], NP, ot, pr, 0, lane), sf_w16b<false>(tpB + m.o16B_wk[k], NP, ot, pr, 1, lane),
                       S.ph[k][pr], S.pl[k][pr], b);
#pragma unroll
      for (int r = 0; r < 4; ++r) nb[ot - LO][r] = sf_tanh_pre(b[r]);
    }
#pragma unroll
    for (int ot = LO; ot <= HI; ++ot) {
      if (k + 1 < NB) put(k + 1, ot, nb[ot - LO]);
      else S.head[ot] = nb[ot - LO];
    }
  }
  f32x2 pam = {0.f, 0.f};
#pragma unroll
  for (int tl = 0; tl <= HI; ++tl)
    pam = sf_head_acc(pam, *reinterpret_cast<const float4*>(hv + tl * 8), *reinterpret_cast<const float4*>(hv + tl * 8 + 4),
                      S.head[tl]);
  const float av = tp[m.o16_hvb + 2 * sl] + sf_sum4groups(pam[0]);
  const float mv = tp[m.o16_hvb + 2 * sl + 1] + sf_sum4groups(pam[1]);
  const float sc = (m.scale_fn == 0 ? sf_softplus(av) : sf_sigmoid(av + 2.0f)) + m.eps;
  const float wv = sf_div(u_sl - mv, sc);
#pragma unroll
  for (int r = 0; r < 4; ++r) S.ut[r] = (g4 == (sl >> 2) && r == (sl & 3)) ? wv : S.ut[r];
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 hidden blocks (PREC = 1): the passes of the persistent sampler with EVERY product on v_mfma_f32_16x16x4_f32 -- the
// arithmetic of BASELINE configs[1] ("fp32"), draw for draw the fmaf chains of the density kernel.  Same incremental
// inverse, same tiles, same queue; what changes is the operand form of the H x H blocks: 16 x 16 fp32 fragments
// (float4[block * 64 + lane], the layout of part A) instead of split-bf16 pairs, and the activation state is the
// accumulator tile itself (no conversion between layers).  The blocks are copied into LDS behind part A from the full
// fp32 image (packed16 + o16_wk): aligned placement (CP) only the blocks on and below the diagonal -- a tile never reads
// tiles above its own -- as entries ot (ot + 1) / 2 + it; contiguous placement all NT x NT.
// Cost: (OT + 1) x 4 MFMAs of 32 cycles per block row instead of (OT / 2 + 1) x 3 of 16: the matrix pipe, not the
// vector issue port, bounds this kernel's dense phase ("fp32 sampler").
// ---------------------------------------------------------------------------------------------------------------
template <bool CP>
__device__ __forceinline__ float4 sf_w16f(const float* wF, int NT, int ot, int it, int lane) {
  const int e = CP ? (ot * (ot + 1)) / 2 + it : ot * NT + it;
  return reinterpret_cast<const float4*>(wF)[e * 64 + lane];
}
template <bool CP>
__device__ __forceinline__ int sf_f16_block_floats(int NT) { return (CP ? NT * (NT + 1) / 2 : NT * NT) * 256; }

template <int OT, int NB, bool CP, bool HM = false>
__device__ __forceinline__ void sf_pass16f(const SfDev& m, const float* tp, const float* tpF, SfPass16F& S, int NT, int sl,
                                           float u_sl, int lane, int g4, int next_ot = -1) {
  const int BF = sf_f16_block_floats<CP>(NT);
  f32x4 c0;
  if (CP && HM && S.tab) {
    c0 = S.c0n;
    if (next_ot >= 0) sf_c0_prefetch(S, next_ot, g4);
  } else {
    c0 = sf_c0_16(m, tp, S, OT, lane, g4);
  }
  const float* hv = tp + m.o16_hv + sl * 128 + g4 * 32;
  const float4 w0 = sf_w16(tp + m.o16_w0, 1, OT, 0, lane);
  f32x2 pam = {0.f, 0.f};
  if (!HM) {
#pragma unroll
    for (int tl = 0; tl < OT; ++tl)
      pam = sf_head_acc(pam, *reinterpret_cast<const float4*>(hv + tl * 8), *reinterpret_cast<const float4*>(hv + tl * 8 + 4), S.head[tl]);
  }
  f32x4 last;
  float4 wh;
  // One block row's fragments are in flight at a time.  The products with the tiles finished in earlier passes (it < OT)
  // do not depend on this pass: they are issued under the initial layer's chain (block 0) and under the tanh of block k
  // (block k + 1), so that the dependent chain of a pass is w0 -> W0[OT, OT] -> tanh -> W1[OT, OT] -> tanh -> head.
  float4 fw[OT + 1];
  f32x4 b = sf_ld4(tp + m.o16_bk[0] + (OT * 4 + g4) * 4);
#pragma unroll
  for (int it = 0; it <= OT; ++it) fw[it] = sf_w16f<CP>(tpF, NT, OT, it, lane);
  __builtin_amdgcn_sched_barrier(0);
  S.act[0][OT] = sf_mma16(w0, S.ut, c0);
#pragma unroll
  for (int it = 0; it < OT; ++it) b = sf_mma16(fw[it], S.act[0][it], b);
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    b = sf_mma16(fw[OT], S.act[k][OT], b);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 bn;
    if (k + 1 < NB) {
      bn = sf_ld4(tp + m.o16_bk[k + 1 < NB ? k + 1 : k] + (OT * 4 + g4) * 4);
#pragma unroll
      for (int it = 0; it <= OT; ++it) fw[it] = sf_w16f<CP>(tpF + (k + 1 < NB ? k + 1 : k) * BF, NT, OT, it, lane);
    } else if (HM) {
      wh = sf_w16(tp + m.o16_wh, NT, 0, OT, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    const f32x4 th = sf_tanh4(b);
    if (k + 1 < NB) {
#pragma unroll
      for (int it = 0; it < OT; ++it) bn = sf_mma16(fw[it], S.act[k + 1][it], bn);
      S.act[k + 1][OT] = th;
      b = bn;
    } else {
      last = th;
    }
  }
  float av, mv;
  if (HM) {
    const f32x4 fresh = sf_mma16(wh, last, S.hdone);
    if (next_ot != OT) S.hdone = fresh;
    const bool odd = (sl & 1) != 0;
    const int src = (lane & 15) + 16 * (sl >> 1);
    av = __shfl(odd ? fresh[2] : fresh[0], src, 64);
    mv = __shfl(odd ? fresh[3] : fresh[1], src, 64);
  } else {
    S.head[OT] = last;
    pam = sf_head_acc(pam, *reinterpret_cast<const float4*>(hv + OT * 8), *reinterpret_cast<const float4*>(hv + OT * 8 + 4), S.head[OT]);
    av = tp[m.o16_hvb + 2 * sl] + sf_sum4groups(pam[0]);
    mv = tp[m.o16_hvb + 2 * sl + 1] + sf_sum4groups(pam[1]);
  }
  const float sc = (m.scale_fn == 0 ? sf_softplus(av) : sf_sigmoid(av + 2.0f)) + m.eps;
  const float wv = sf_div(u_sl - mv, sc);
#pragma unroll
  for (int r = 0; r < 4; ++r) S.ut[r] = (g4 == (sl >> 2) && r == (sl & 3)) ? wv : S.ut[r];
}

// The fp32 pass when the degree group straddles tiles LO..HI (contiguous packing; see sf_pass16_span).
template <int LO, int HI, int NB>
__device__ __forceinline__ void sf_pass16f_span(const SfDev& m, const float* tp, const float* tpF, SfPass16F& S, int NT, int sl,
                                                float u_sl, int lane, int g4) {
  const int BF = sf_f16_block_floats<false>(NT);
  const float* hv = tp + m.o16_hv + sl * 128 + g4 * 32;
#pragma unroll
  for (int ot = LO; ot <= HI; ++ot) S.act[0][ot] = sf_mma16(sf_w16(tp + m.o16_w0, 1, ot, 0, lane), S.ut, sf_c0_16(m, tp, S, ot, lane, g4));
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    f32x4 nb[HI - LO + 1];
#pragma unroll
    for (int ot = LO; ot <= HI; ++ot) {
      f32x4 b = sf_ld4(tp + m.o16_bk[k] + (ot * 4 + g4) * 4);
#pragma unroll
      for (int it = 0; it <= HI; ++it) b = sf_mma16(sf_w16f<false>(tpF + k * BF, NT, ot, it, lane), S.act[k][it], b);
      nb[ot - LO] = sf_tanh4(b);
    }
#pragma unroll
    for (int ot = LO; ot <= HI; ++ot) {
      if (k + 1 < NB) S.act[k + 1][ot] = nb[ot - LO];
      else S.head[ot] = nb[ot - LO];
    }
  }
  f32x2 pam = {0.f, 0.f};
#pragma unroll
  for (int tl = 0; tl <= HI; ++tl)
    pam = sf_head_acc(pam, *reinterpret_cast<const float4*>(hv + tl * 8), *reinterpret_cast<const float4*>(hv + tl * 8 + 4),
                      S.head[tl]);
  const float av = tp[m.o16_hvb + 2 * sl] + sf_sum4groups(pam[0]);
  const float mv = tp[m.o16_hvb + 2 * sl + 1] + sf_sum4groups(pam[1]);
  const float sc = (m.scale_fn == 0 ? sf_softplus(av) : sf_sigmoid(av + 2.0f)) + m.eps;
  const float wv = sf_div(u_sl - mv, sc);
#pragma unroll
  for (int r = 0; r < 4; ++r) S.ut[r] = (g4 == (sl >> 2) && r == (sl & 3)) ? wv : S.ut[r];
}

// ---- what the kernels below see of the two operand forms (PREC: 0 = split bf16 x3, 1 = fp32)
template <int PREC> struct SfHid16;
template <> struct SfHid16<0> {
  using State = SfPass16B;
  template <int OT, int NB, bool CP, bool HM>
  static __device__ __forceinline__ void pass(const SfDev& m, const float* tp, const void* tpH, State& S, int NT, int sl, float u_sl,
                                              int lane, int g4, int next_ot) {
    sf_pass16b<OT, NB, CP, HM>(m, tp, static_cast<const unsigned int*>(tpH), S, NT, sl, u_sl, lane, g4, next_ot);
  }
  template <int LO, int HI, int NB>
  static __device__ __forceinline__ void span(const SfDev& m, const float* tp, const void* tpH, State& S, int NT, int sl, float u_sl,
                                              int lane, int g4) {
    sf_pass16b_span<LO, HI, NB>(m, tp, static_cast<const unsigned int*>(tpH), S, NT, sl, u_sl, lane, g4);
  }
  // cleared per tile and transform: a non-finite value left behind by one draw must not reach another one through a
  // structural zero (a pass only reads tiles that an earlier pass of the SAME tile and transform wrote, or zeros).
  // SEQ (unrolled kernels: passes in tile order 0, 1, 2, 3): the only operands read before this tile and transform wrote
  // them are the odd tiles (the second half of a pair, read with all-zero weights by the pass of the even tile)
  template <bool SEQ, bool HM>
  static __device__ __forceinline__ void clear(State& S) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int c = SEQ ? 2 : 0; c < 4; ++c) { S.ph[k][pr][c] = 0u; S.pl[k][pr][c] = 0u; }
    if (!HM) {
#pragma unroll
      for (int ot = 0; ot < 4; ++ot)
#pragma unroll
        for (int r = 0; r < 4; ++r) S.head[ot][r] = 0.f;
    }
  }
  // LDS floats behind part A
  static __host__ __device__ int lds_floats(const SfDev& m, bool /*cp*/) { return m.t16B_stride; }
};
template <> struct SfHid16<1> {
  using State = SfPass16F;
  template <int OT, int NB, bool CP, bool HM>
  static __device__ __forceinline__ void pass(const SfDev& m, const float* tp, const void* tpH, State& S, int NT, int sl, float u_sl,
                                              int lane, int g4, int next_ot) {
    sf_pass16f<OT, NB, CP, HM>(m, tp, static_cast<const float*>(tpH), S, NT, sl, u_sl, lane, g4, next_ot);
  }
  template <int LO, int HI, int NB>
  static __device__ __forceinline__ void span(const SfDev& m, const float* tp, const void* tpH, State& S, int NT, int sl, float u_sl,
                                              int lane, int g4) {
    sf_pass16f_span<LO, HI, NB>(m, tp, static_cast<const float*>(tpH), S, NT, sl, u_sl, lane, g4);
  }
  // SEQ: a pass reads tiles 0 .. OT of its own tile and transform only, all written by then -- nothing to clear
  template <bool SEQ, bool HM>
  static __device__ __forceinline__ void clear(State& S) {
    if (!SEQ) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
          for (int r = 0; r < 4; ++r) S.act[k][ot][r] = 0.f;
    }
    if (!HM) {
#pragma unroll
      for (int ot = 0; ot < 4; ++ot)
#pragma unroll
        for (int r = 0; r < 4; ++r) S.head[ot][r] = 0.f;
    }
  }
  static __host__ __device__ int lds_floats(const SfDev& m, bool cp) {
    const int nb = m.NB < 2 ? m.NB : 2;
    return nb * (cp ? m.nT16 * (m.nT16 + 1) / 2 : m.nT16 * m.nT16) * 256;
  }
};

// ---------------------------------------------------------------------------------------------------------------
// Fused first layer (PREC = 2; fp32, unrolled kernels with the context table).  nflows' MADE has NO activation between the
// initial layer and the first block's linear (oracle/flows.py::_made: h = W0 u + b0 + Wc e + bc; then h = tanh(W1 h + b1), ...),
// so the two are one affine map of the finished dimensions:
//     W1 (W0 u + c0) + b1  =  W' u + c0',     W' = (W1 o M)(W0 o M0)  [H x D],     c0' = b1 + (W1 o M) c0  [per galaxy and transform]
// W' is computed once per parameter update (k_maf_fuse16, in fp64, rounded to fp32: image block o16_wp), c0' once per galaxy
// behind c0 in the context table (k_maf_ctab16).  A pass then costs 4 + 4 (OT + 1) + 4 fp32 MFMAs instead of 4 + 8 (OT + 1) + 4
// (72 instead of 112 per tile and transform for cfg1), the first block's fragments and the state of its inputs disappear
// (LDS 22 KB instead of 32 KB per transform, 16 registers), and the dependent chain of a pass is one layer shorter.  Same
// function of the same parameters as the two-layer form to fp32 rounding (parity rows: given noise, draw for draw).
// ---------------------------------------------------------------------------------------------------------------
struct SfPass16G {
  f32x4 act[4];        // output of block 0 = input of block 1; [tile]
  f32x4 hdone, ut;
  const float* c0p;    // this draw's c0' rows of the transform
  const float* xr;
  f32x4 c0n;
  bool tab;
};
template <int OT, int NB>
__device__ __forceinline__ void sf_pass16g(const SfDev& m, const float* tp, const float* tpF, SfPass16G& S, int NT, int sl, float u_sl,
                                           int lane, int g4, int next_ot) {
  const f32x4 c0 = S.c0n;
  if (next_ot >= 0) sf_c0_prefetch(S, next_ot, g4);
  const float4 wp = sf_w16(tp + m.o16_wp, 1, OT, 0, lane);
  const float4 wh = sf_w16(tp + m.o16_wh, NT, 0, OT, lane);
  float4 fw[OT + 1];
  f32x4 b1;
  if (NB == 2) {
    b1 = sf_ld4(tp + m.o16_bk[1] + (OT * 4 + g4) * 4);
#pragma unroll
    for (int it = 0; it <= OT; ++it) fw[it] = sf_w16f<true>(tpF, NT, OT, it, lane);
  }
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 b = sf_mma16(wp, S.ut, c0);   // the first block's pre-activation (tanh pre-scale folded in like every hidden block)
  if (NB == 2) {
#pragma unroll
    for (int it = 0; it < OT; ++it) b1 = sf_mma16(fw[it], S.act[it], b1);   // tiles finished in earlier passes: not on the chain
  }
  f32x4 last = sf_tanh4(b);
  if (NB == 2) {
    S.act[OT] = last;
    b1 = sf_mma16(fw[OT], last, b1);
    last = sf_tanh4(b1);
  }
  const f32x4 fresh = sf_mma16(wh, last, S.hdone);
  if (next_ot != OT) S.hdone = fresh;
  const bool odd = (sl & 1) != 0;
  const int src = (lane & 15) + 16 * (sl >> 1);
  const float av = __shfl(odd ? fresh[2] : fresh[0], src, 64);
  const float mv = __shfl(odd ? fresh[3] : fresh[1], src, 64);
  const float sc = (m.scale_fn == 0 ? sf_softplus(av) : sf_sigmoid(av + 2.0f)) + m.eps;
  const float wv = sf_div(u_sl - mv, sc);
#pragma unroll
  for (int r = 0; r < 4; ++r) S.ut[r] = (g4 == (sl >> 2) && r == (sl & 3)) ? wv : S.ut[r];
}
template <> struct SfHid16<2> {
  using State = SfPass16G;
  template <int OT, int NB, bool CP, bool HM>
  static __device__ __forceinline__ void pass(const SfDev& m, const float* tp, const void* tpH, State& S, int NT, int sl, float u_sl,
                                              int lane, int g4, int next_ot) {
    static_assert(CP && HM, "fused first layer: unrolled kernels only (aligned placement, head tile)");
    sf_pass16g<OT, NB>(m, tp, static_cast<const float*>(tpH), S, NT, sl, u_sl, lane, g4, next_ot);
  }
  template <bool SEQ, bool HM>
  static __device__ __forceinline__ void clear(State&) {}   // (a pass reads tiles 0 .. OT of its own tile and transform only)
  // LDS floats behind part A: the SECOND block's fragments on and below the diagonal (the first block lives in W')
  static __host__ __device__ int lds_floats(const SfDev& m, bool /*cp*/) { return m.NB >= 2 ? m.nT16 * (m.nT16 + 1) / 2 * 256 : 0; }
};

// Staging of one transform's operands (all four waves; the caller brackets it with barriers): part A of the fp32 image
// (`a_floats` floats: input layer, biases, head rows; + the context block without a table) in 4 KiB groups -- ONE address, four
// immediate offsets -- and behind it the hidden blocks: PREC 0 the split-bf16 image (4 KiB groups), PREC 1 the fp32 blocks
// of the full image, 1 KiB (one 16 x 16 block) per wave-instruction.  Direct global -> LDS copies.
template <int PREC, bool CP>
__device__ __forceinline__ void sf_stage16(const SfDev& m, int t, int a_floats, int wave) {
  const int lane_ = threadIdx.x & 63;
  const int ga = a_floats >> 10;
  const float4* __restrict__ sa = reinterpret_cast<const float4*>(m.packed16 + (size_t)t * m.t16_stride);
  float4* __restrict__ d4 = reinterpret_cast<float4*>(sf_lds16);
  if constexpr (PREC == 0) {
    const int gb = m.t16B_stride >> 10;
    const float4* __restrict__ sb = reinterpret_cast<const float4*>(m.packed16B + (size_t)t * m.t16B_stride);
    for (int gi = __builtin_amdgcn_readfirstlane(wave); gi < ga + gb; gi += 4) {
      const float4* g = (gi < ga ? sa + gi * 256 : sb + (gi - ga) * 256) + lane_;
      float4* l = d4 + gi * 256;
      __builtin_amdgcn_global_load_lds((const void*)g, (void __attribute__((address_space(3)))*)l, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const void*)g, (void __attribute__((address_space(3)))*)l, 16, 1024, 0);
      __builtin_amdgcn_global_load_lds((const void*)g, (void __attribute__((address_space(3)))*)l, 16, 2048, 0);
      __builtin_amdgcn_global_load_lds((const void*)g, (void __attribute__((address_space(3)))*)l, 16, 3072, 0);
    }
  } else {
    for (int gi = __builtin_amdgcn_readfirstlane(wave); gi < ga; gi += 4) {
      const float4* g = sa + gi * 256 + lane_;
      float4* l = d4 + gi * 256;
      __builtin_amdgcn_global_load_lds((const void*)g, (void __attribute__((address_space(3)))*)l, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const void*)g, (void __attribute__((address_space(3)))*)l, 16, 1024, 0);
      __builtin_amdgcn_global_load_lds((const void*)g, (void __attribute__((address_space(3)))*)l, 16, 2048, 0);
      __builtin_amdgcn_global_load_lds((const void*)g, (void __attribute__((address_space(3)))*)l, 16, 3072, 0);
    }
    const int NT = m.nT16, nb = m.NB < 2 ? m.NB : 2;
    const int per = CP ? NT * (NT + 1) / 2 : NT * NT;
    constexpr int K0 = PREC == 2 ? 1 : 0;   // (fused first layer: the first block is not staged)
    float4* __restrict__ dF = d4 + (a_floats >> 2);
    for (int e = __builtin_amdgcn_readfirstlane(wave); e < (nb - K0) * per; e += 4) {
      const int k = K0 + (e >= per ? 1 : 0), ee = e - (k - K0) * per;
      int src_blk = ee;
      if (CP) {  // entry ot (ot + 1) / 2 + it  ->  block ot * NT + it   (NT <= 4)
        const int ot = ee >= 6 ? 3 : (ee >= 3 ? 2 : (ee >= 1 ? 1 : 0));
        src_blk = ot * NT + (ee - ot * (ot + 1) / 2);
      }
      const int owk = k ? m.o16_wk[1] : m.o16_wk[0];  // (no dynamic index into the descriptor)
      const float4* g = sa + ((owk >> 2) + src_blk * 64) + lane_;
      __builtin_amdgcn_global_load_lds((const void*)g, (void __attribute__((address_space(3)))*)(dF + e * 64), 16, 0, 0);
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the copies have landed
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent sampler: the same tile pipeline as k_maf_inv16, driven by the device work queue of sf_queue.h.  One
// launch resolves every slot of the dense list (first attempts AND retries); sampler only (no parity hook, no
// acceptance mode, no log-determinant).
// Hidden H x H blocks run on split-bf16 MFMA (sf_pass16b); the rest of the arithmetic is fp32 as in k_maf_inv16.
// The two argument blocks are read through the kernarg segment pointer, laundered once per iteration: descriptor
// fields and table entries are then loaded where they are used (scalar-cache hits) instead of being hoisted out of
// the persistent loop and kept in registers for the life of the kernel -- with them live the pass functions spill.
// ---------------------------------------------------------------------------------------------------------------
struct SfSamp16Args {
  SfDev m;
  SfSampleArgsHost a;
};

// TPW = draw tiles per wave and iteration (1 or 2): with 2 a workgroup takes 128 items per iteration and every wave walks
// TWO tiles of 16 draws through each staged transform, one after the other -- the queue fetch, the image copy and their
// barriers are paid once per 128 draws instead of once per 64 (together ~20 % of a workgroup's time at TPW = 1); between
// transforms a tile is just its 4 registers of u and its galaxy index.
// Offsets of the 16-row sampler image for the shapes of the unrolled kernels (D = DD, DD - 1 hidden tiles of one degree
// group each, NB blocks), as sf_layout.cpp emits them: compile-time constants in those kernels (LDS reads with immediate
// offsets, no descriptor loads inside a pass); the host checks them against the packer's before it picks such a kernel.
template <int NB, int DD>
struct SfFix16 {
  static constexpr int NT = DD - 1;
  static constexpr int o_w0 = 0, o_b0 = NT * 256, o_bk0 = o_b0 + NT * 16, o_bk1 = o_bk0 + NT * 16;
  static constexpr int o_hv = o_b0 + NT * 16 * (1 + NB), o_hvb = o_hv + DD * 128;
  static constexpr int o_wh = (o_hvb + 2 * DD + 3) / 4 * 4, o_bh = o_wh + NT * 256;
  static constexpr int o_wp = (o_bh + 16 + 3) / 4 * 4;
  static constexpr int a_tab = (o_wp + NT * 256 + 1023) / 1024 * 1024;
  static constexpr int entries = NT == 4 ? 6 : (NT == 3 ? 4 : 2);   // (ot, pair) fragments a block keeps: sum of ot / 2 + 1
  static constexpr int oB_wk1 = entries * 512, B_stride = (NB * entries * 512 + 1023) / 1024 * 1024;
  static bool matches(const SfDev& m) {
    return m.nT16 == NT && m.o16_w0 == o_w0 && m.o16_b0 == o_b0 && m.o16_bk[0] == o_bk0 && (NB < 2 || m.o16_bk[1] == o_bk1) &&
           m.o16_hv == o_hv && m.o16_hvb == o_hvb && m.o16_wh == o_wh && m.o16_bh == o_bh && m.o16_wp == o_wp && m.t16_a_tab == a_tab &&
           m.o16B_wk[0] == 0 && (NB < 2 || m.o16B_wk[1] == oB_wk1) && m.t16B_stride == B_stride;
  }
  static __device__ __forceinline__ void apply(SfDev& m) {
    m.nT16 = NT; m.o16_w0 = o_w0; m.o16_b0 = o_b0; m.o16_bk[0] = o_bk0; m.o16_bk[1] = o_bk1; m.o16_hv = o_hv; m.o16_hvb = o_hvb;
    m.o16_wh = o_wh; m.o16_bh = o_bh; m.o16_wp = o_wp; m.t16_a_tab = a_tab; m.o16B_wk[0] = 0; m.o16B_wk[1] = oB_wk1; m.t16B_stride = B_stride;
  }
};

// DD > 0 (HM, aligned placement with ONE degree group per tile, D == DD <= 5): the passes of a transform are unrolled
// with the tile of each known at compile time (pass p works on tile p - 2) -- no dispatch, and the per-tile state is
// updated in place instead of being copied into the registers every arm of the switch has to agree on.
// PREC: operand form of the hidden H x H blocks (SfHid16): 0 = split bf16 x3, 1 = fp32.
// Workgroups per CU of the fused-first-layer kernel (PREC 2: 25 KB of LDS, 106 VGPRs).  Five fit once the compiler is held
// to 96 VGPRs (94 used, no scratch), and measured SLOWER on the headline workload: 2.11-2.15 ms against 2.06 ms with four.
#ifndef SF_SAMP16_WG_FUSED
#define SF_SAMP16_WG_FUSED 4
#endif
template <int NB, bool SPAN, bool HM, int TPW, int DD = 0, int PREC = 0>
__global__ __launch_bounds__(256, (SPAN ? 3 : (PREC == 2 ? SF_SAMP16_WG_FUSED : 4))) void k_maf_samp16(SfSamp16Args args_in) {
  static_assert(DD == 0 || (HM && !SPAN && DD >= 2 && DD <= 5), "unrolled passes: head tile, aligned placement, D <= 5");
  using HID = SfHid16<PREC>;
  constexpr int IPW = 64 * TPW;
  const int wave = threadIdx.x >> 6;
  // with the per-galaxy context table the context block Wc is never read: only the prefix of part A before it is staged
  unsigned int* ctrl = reinterpret_cast<unsigned int*>(
      sf_lds16 + (args_in.m.ctab ? args_in.m.t16_a_tab : args_in.m.t16_a) + HID::lds_floats(args_in.m, !SPAN));
  unsigned int pf;
  sf_q_begin<IPW>(args_in.a, ctrl, pf);
  // per-slot constants of the epilogue, once per workgroup: {shift, 1 / scale, lo, hi, theta column} of physical slot p.
  // (Read from global memory where they are used they cost every iteration two dependent L2 round trips.)
  float* ecb = reinterpret_cast<float*>(ctrl + SF_Q_WORDS(IPW));
  if (threadIdx.x < 16) {
    const int p = threadIdx.x;
    const bool on = p < args_in.m.D;
    const int td = on ? (int)args_in.m.cst[args_in.m.c_tdim + p] : 0;
    ecb[p * 5 + 0] = on ? args_in.m.cst[args_in.m.c_pshift + p] : 0.f;
    ecb[p * 5 + 1] = on ? __builtin_amdgcn_rcpf(args_in.m.cst[args_in.m.c_pscale + p]) : 0.f;
    ecb[p * 5 + 2] = (on && args_in.a.lo) ? args_in.a.lo[td] : -3.4e38f;
    ecb[p * 5 + 3] = (on && args_in.a.lo) ? args_in.a.hi[td] : 3.4e38f;
    reinterpret_cast<int*>(ecb)[p * 5 + 4] = td;
    if (on) reinterpret_cast<int*>(ecb)[80 + td] = p;   // theta column -> physical slot (the row-linear stores of the epilogue)
  }
  // (the first sf_q_fetch begins with a barrier: the block is visible to every wave before its first epilogue)
#ifdef SF_Q_STATS
  const unsigned long long qs_k0 = __builtin_amdgcn_s_memtime();
  unsigned long long qs_iters = 0, qs_last_work = 0;
  unsigned long long qs_ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // [5..9]: the same phases in tail mode
  if (threadIdx.x == 0) atomicMin(&args_in.a.q->stats[10], __builtin_amdgcn_s_memrealtime());  // first start (100 MHz)
#endif
  for (;;) {
    const SfSamp16Args* ap;
    {
      auto kp = __builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(kp));
      ap = (const SfSamp16Args*)kp;
    }
    const SfDev& m_arg = ap->m;
    SfDev m_fix;  // (DD > 0: the descriptor with the image offsets replaced by their compile-time values)
    if constexpr (DD > 0) { m_fix = m_arg; SfFix16<NB, DD>::apply(m_fix); }
    const SfDev& m = DD > 0 ? m_fix : m_arg;
    const SfSampleArgsHost& a = ap->a;
    const int lane = (threadIdx.x & 63) + sf_opaque_zero();  // lane-derived addresses are recomputed per iteration
    const int s = lane & 15, g4 = lane >> 4;
#ifdef SF_Q_STATS
    const unsigned long long qs_t_top = __builtin_amdgcn_s_memrealtime();
#endif
    if (!sf_q_fetch<IPW, 64>(a, ctrl, pf)) {
#ifdef SF_Q_STATS
      if (threadIdx.x == 0) {
        for (int i = 0; i < 10; ++i) atomicAdd(&a.q->stats[14 + i], qs_ph[i]);  // fetch | prologue | staging | passes | epilogue (10 ns), dense then tail
        atomicAdd(&a.q->stats[9], __builtin_amdgcn_s_memtime() - qs_k0);  // workgroup lifetime
        atomicMax(&a.q->stats[11], qs_iters);                              // most iterations of one workgroup
        atomicMax(&a.q->stats[12], qs_last_work);                          // end of the last flow evaluation (100 MHz)
        atomicMax(&a.q->stats[13], __builtin_amdgcn_s_memrealtime());      // last exit
      }
#endif
      break;
    }
#ifdef SF_Q_STATS
    if (a.qtrace && threadIdx.x == 0 && qs_iters < (2048u * 256u) / gridDim.x) {
      uint32_t* tr = a.qtrace + ((size_t)blockIdx.x * ((2048u * 256u) / gridDim.x) + qs_iters) * 4;
      tr[0] = (uint32_t)__builtin_amdgcn_s_memrealtime(); tr[1] = ctrl[0]; tr[2] = ctrl[1] | (ctrl[9] << 8);
    }
    ++qs_iters;
    const unsigned long long qs_t_fetch = __builtin_amdgcn_s_memrealtime();
    const int qs_o = ctrl[9] ? 5 : 0;
    qs_ph[qs_o + 0] += qs_t_fetch - qs_t_top;
    unsigned long long qs_t_mark = qs_t_fetch;
#endif
    const int NT = m.nT16;
    const unsigned int n_items = ctrl[0] << ctrl[1];  // items of this iteration (entries x attempts per entry)
    // the wave's tiles: tile j covers items (j * 4 + wave) * 16 .. + 15.  Between transforms a tile is its u registers
    // and its galaxy; `cur` is the tile being worked on, `oth` the other one (TPW = 2), swapped after every tile.
    // (the tile being worked on is always entry 0: the entries rotate after every tile, TPW turns restore the order)
    f32x4 u_t[TPW];
    unsigned int g_t[TPW];
#define u_cur u_t[0]
#define gal_cur g_t[0]
#pragma unroll
    for (int j = TPW - 1; j >= 0; --j) {  // (j = 0 last: it is the first `cur`)
      const int wi = (j * 4 + wave) * 16 + s;
      const unsigned int n_ent = ctrl[0];
      const int lgA = (int)ctrl[1];
      const unsigned int e = (unsigned)wi >> lgA;
      const unsigned int ee = e < n_ent ? e : 0u;
      const uint32_t slot = ctrl[SF_Q_HDR + ee];
      const uint32_t att = ctrl[SF_Q_HDR + IPW + ee] + ((unsigned)wi & ((1u << lgA) - 1u));
      float z4[4];
      sf_normal4(a.k0, a.k1, (uint64_t)slot + a.rng_slot_offset, att, (uint32_t)g4, z4);  // Philox block g4 = dimensions 4*g4 .. 4*g4+3
#pragma unroll
      for (int r = 0; r < 4; ++r) u_t[j][r] = (4 * g4 + r < m.D) ? z4[r] : 0.f;
      g_t[j] = slot / (uint32_t)a.S;
    }
    uint32_t tile_bits = 0, lo_bits = 0;  // g16_tile / g16_lo packed 2 bits per degree
    if constexpr (DD == 0) {                // (the unrolled kernels know the tile of every pass: p - 2)
#pragma unroll
      for (int q = 0; q < SF_DMAX; ++q) {
        tile_bits |= (uint32_t)(m.g16_tile[q] & 3) << (2 * q);
        lo_bits |= (uint32_t)(m.g16_lo[q] & 3) << (2 * q);
      }
    }
    typename HID::State S;
    S.tab = DD > 0 ? true : m.ctab != nullptr;  // (the unrolled kernels are only launched with the context table)
    for (int t = m.T - 1; t >= 0; --t) {
      // requested before the staging barriers so that their round trips overlap with the image copy: the degree ->
      // slot table of the transform and (table path) c0 of the first tile's first MFMA pass
      const int dsl = (int)m.cst[m.c_dslot + t * SF_DMAX + s];
      S.c0p = S.tab ? m.ctab + ((size_t)gal_cur * m.T + t) * m.ctab_R + (PREC == 2 ? m.nT16 * 16 : 0) : nullptr;
      if (HM && S.tab) sf_c0_prefetch(S, (int)((tile_bits >> 2) & 3u), g4);
#ifdef SF_Q_STATS
      { const unsigned long long n = __builtin_amdgcn_s_memrealtime(); qs_ph[qs_o + (t == m.T - 1 ? 1 : 3)] += n - qs_t_mark; qs_t_mark = n; }
#endif
      __syncthreads();
      // part A of the fp32 image and the hidden blocks, one behind the other in LDS
      sf_stage16<PREC, !SPAN>(m, t, S.tab ? m.t16_a_tab : m.t16_a, wave);
      __syncthreads();
#ifdef SF_Q_STATS
      { const unsigned long long n = __builtin_amdgcn_s_memrealtime(); qs_ph[qs_o + 2] += n - qs_t_mark; qs_t_mark = n; }
#endif
      const float* tp = sf_lds16;
      const void* tpB = sf_lds16 + (S.tab ? m.t16_a_tab : m.t16_a);
#pragma unroll 1
      for (int j = 0; j < TPW; ++j) {
        // a wave whose tile holds no item (tail iterations with few entries) skips the flow: its issue slots go to the
        // other workgroups of the CU
        const bool tile_has_work = (unsigned)((j * 4 + wave) * 16) < n_items;
        if (tile_has_work) {
          if (j > 0) {  // (tile 0's requests went out before the staging barriers)
            S.c0p = S.tab ? m.ctab + ((size_t)gal_cur * m.T + t) * m.ctab_R + (PREC == 2 ? m.nT16 * 16 : 0) : nullptr;
            if (HM && S.tab) sf_c0_prefetch(S, (int)((tile_bits >> 2) & 3u), g4);
          }
          S.xr = a.x + gal_cur * m.C;
          HID::template clear<(DD > 0), HM>(S);
#pragma unroll
          for (int r = 0; r < 4; ++r) S.ut[r] = 0.f;
          if (HM) S.hdone = sf_ld4(tp + m.o16_bh + g4 * 4);
          {
            const int sl = __builtin_amdgcn_readlane(dsl, 0);
            const float av = tp[m.o16_hvb + 2 * sl], mv = tp[m.o16_hvb + 2 * sl + 1];
            const float sc = (m.scale_fn == 0 ? sf_softplus(av) : sf_sigmoid(av + 2.0f)) + m.eps;
            const float wv = sf_div(sf_slot16_own(u_cur, sl) - mv, sc);
#pragma unroll
            for (int r = 0; r < 4; ++r) S.ut[r] = (g4 == (sl >> 2) && r == (sl & 3)) ? wv : S.ut[r];
          }
          if constexpr (DD > 0) {
            auto seq_pass = [&](auto otc) {
              constexpr int OT = decltype(otc)::value;
              const int sl = __builtin_amdgcn_readlane(dsl, OT + 1);
              HID::template pass<OT, NB, true, true>(m, tp, tpB, S, NT, sl, sf_slot16_own(u_cur, sl), lane, g4, OT + 2 < DD ? OT + 1 : -1);
            };
            seq_pass(std::integral_constant<int, 0>{});
            if constexpr (DD >= 3) seq_pass(std::integral_constant<int, 1>{});
            if constexpr (DD >= 4) seq_pass(std::integral_constant<int, 2>{});
            if constexpr (DD >= 5) seq_pass(std::integral_constant<int, 3>{});
          } else
          for (int p = 2; p <= m.D; ++p) {
            const int sl = __builtin_amdgcn_readlane(dsl, p - 1);
            const float u_sl = sf_slot16_own(u_cur, sl);  // (only the owning row group keeps what is computed from it)
            const uint32_t hi_t = (tile_bits >> (2 * (p - 1))) & 3u;
            const uint32_t lo_t = SPAN ? (lo_bits >> (2 * (p - 1))) & 3u : hi_t;
            const int nx = p < m.D ? (int)((tile_bits >> (2 * p)) & 3u) : -1;  // tile of the next pass (aligned placement)
            switch (lo_t * 4 + hi_t) {
              case 0: HID::template pass<0, NB, !SPAN, HM>(m, tp, tpB, S, NT, sl, u_sl, lane, g4, nx); break;
              case 5: HID::template pass<1, NB, !SPAN, HM>(m, tp, tpB, S, NT, sl, u_sl, lane, g4, nx); break;
              case 10: HID::template pass<2, NB, !SPAN, HM>(m, tp, tpB, S, NT, sl, u_sl, lane, g4, nx); break;
              case 15: HID::template pass<3, NB, !SPAN, HM>(m, tp, tpB, S, NT, sl, u_sl, lane, g4, nx); break;
              case 1: if constexpr (SPAN) HID::template span<0, 1, NB>(m, tp, tpB, S, NT, sl, u_sl, lane, g4); break;
              case 2: if constexpr (SPAN) HID::template span<0, 2, NB>(m, tp, tpB, S, NT, sl, u_sl, lane, g4); break;
              case 3: if constexpr (SPAN) HID::template span<0, 3, NB>(m, tp, tpB, S, NT, sl, u_sl, lane, g4); break;
              case 6: if constexpr (SPAN) HID::template span<1, 2, NB>(m, tp, tpB, S, NT, sl, u_sl, lane, g4); break;
              case 7: if constexpr (SPAN) HID::template span<1, 3, NB>(m, tp, tpB, S, NT, sl, u_sl, lane, g4); break;
              default: if constexpr (SPAN) HID::template span<2, 3, NB>(m, tp, tpB, S, NT, sl, u_sl, lane, g4); break;
            }
          }
          u_cur = S.ut;
        }
        if (TPW > 1) {  // the next tile's turn (after TPW turns every tile is entry 0 under its own index again)
          const f32x4 tu = u_t[0];
          const unsigned int tg = g_t[0];
#pragma unroll
          for (int q = 0; q + 1 < TPW; ++q) { u_t[q] = u_t[q + 1]; g_t[q] = g_t[q + 1]; }
          u_t[TPW - 1] = tu;
          g_t[TPW - 1] = tg;
        }
      }
    }
#ifdef SF_Q_STATS
    { const unsigned long long n = __builtin_amdgcn_s_memrealtime(); qs_ph[qs_o + 3] += n - qs_t_mark; qs_t_mark = n; }
#endif
    // ---------------------------------------------------------------- un-standardise, box test, outputs (per tile)
#pragma unroll 1
    for (int j = 0; j < TPW; ++j) {
      const int wi = (j * 4 + wave) * 16 + s;
      float th[4];
      int tdc[4];
      bool ok = true;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = 4 * g4 + r;
        const float* e = ecb + p * 5;
        th[r] = (u_cur[r] - e[0]) * e[1];
        tdc[r] = reinterpret_cast<const int*>(e)[4];
        // finite (NaN compares false) and inside the box; slots >= D: u = 0, shift = 0, 1 / scale = 0 -> 0, always inside
        ok = ok && (fabsf(th[r]) <= 3.0e38f) && (th[r] >= e[2]) && (th[r] <= e[3]);
      }
      // the work words are still in LDS: nothing about the item had to stay in registers through the flow
      const unsigned int n_ent = ctrl[0];
      const int lgA = (int)ctrl[1];
      const unsigned int e = (unsigned)wi >> lgA;
      const bool entry_ok = e < n_ent;
      const unsigned int ee = entry_ok ? e : 0u;
      const uint32_t slot = ctrl[SF_Q_HDR + ee];
      const uint32_t att_base = ctrl[SF_Q_HDR + IPW + ee];
      const uint32_t att = att_base + ((unsigned)wi & ((1u << lgA) - 1u));
      const bool valid = entry_ok && att < a.attempt_limit;
      const unsigned long long okb = __ballot(ok);
      const uint32_t acc16 = (uint32_t)(okb & (okb >> 16) & (okb >> 32) & (okb >> 48) & 0xffffull) &
                             (uint32_t)(__ballot(valid) & 0xffffull);
      // A consecutive items hold attempts att_base .. att_base+A-1 of one slot: the lowest accepted one wins.
      // A <= 16: the group sits inside this wave's tile.  A = 32 / 64 (the last few slots of a catalogue, each tried by
      // half of / the whole workgroup at once): the group is tiles j*4 + gw0 .. of the SAME j, one per wave; the waves
      // combine through four LDS words.
      const int A = 1 << lgA;
      int first, me;
      if (lgA <= 4) {
        const int grp0 = (s / A) * A;
        const uint32_t gmask = (acc16 >> grp0) & ((1u << A) - 1u);
        first = gmask ? (int)__builtin_ctz(gmask) : -1;
        me = s - grp0;
      } else {
        if ((threadIdx.x & 63) == 0) ctrl[20 + wave] = acc16 ? (unsigned)__builtin_ctz(acc16) : 16u;
        __syncthreads();  // (lgA is the same for the whole workgroup)
        const int gw0 = (int)(((e << lgA) >> 4) & 3u), gwn = A >> 4;  // first wave of the group, waves per group
        first = -1;
#pragma unroll
        for (int w = 3; w >= 0; --w) {
          const unsigned int cw = ctrl[20 + w];
          if (w >= gw0 && w < gw0 + gwn && cw < 16u) first = (w - gw0) * 16 + (int)cw;
        }
        me = wi - (int)(e << lgA);
        if (TPW > 1) __syncthreads();  // the four words are rewritten for the next tile
      }
      // One attempt per entry (every dense iteration): the tile's 16 x D block of draws is handed to the lanes LINEARLY -- lane L
      // stores element L, L + 64, ... = (draw se = e / D, column c = e % D) at out[slot(se) D + c] -- so that a wave-instruction
      // covers the rows of consecutive draws side by side (the dense order deals RUNS of consecutive draws of a galaxy:
      // dense_run) instead of five 4-byte pieces per draw from two row groups: whole 64-byte segments in HBM, and fewer PCIe
      // packets when `out` is the host's float64 array.  The value sits in lane se + 16 (p >> 2), register p & 3, p = the
      // column's physical slot; the slot id in lane se.
      if (lgA == 0) {
        const int Dn = DD > 0 ? DD : m.D;
        const int* c2s = reinterpret_cast<const int*>(ecb) + 80;
        for (int e0 = 0; e0 < 16 * Dn; e0 += 64) {
          const int e = e0 + (lane & 63);
          const int se = (e / Dn) & 15, c = e - (e / Dn) * Dn;
          const int p = c2s[c];
          const int src = se + 16 * (p >> 2), rr = p & 3;
          const float v0 = __shfl(th[0], src, 64), v1 = __shfl(th[1], src, 64), v2 = __shfl(th[2], src, 64), v3 = __shfl(th[3], src, 64);
          const float v = rr == 0 ? v0 : (rr == 1 ? v1 : (rr == 2 ? v2 : v3));
          const uint32_t sl_e = (uint32_t)__shfl((int)slot, se, 64);
          if (e < 16 * Dn && ((acc16 >> se) & 1u)) sf_out_store(a, (size_t)sl_e * Dn + c, v);
        }
      } else if (valid && me == first) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * g4 + r < m.D) sf_out_store(a, (size_t)slot * m.D + tdc[r], th[r]);
      }
      // accepted -> resolved; rejected -> staged for the retry ring, or for the survivor list once the launch's attempt
      // limit is reached; wave 0 publishes everything the workgroup staged at the top of the next sf_q_fetch
      const bool leader = entry_ok && g4 == 0 && me == 0;
      const uint32_t room = a.attempt_limit > att_base ? a.attempt_limit - att_base : 0u;
      const uint32_t tried = room < (uint32_t)A ? room : (uint32_t)A;  // attempts of this entry evaluated here
      const bool hit = leader && first >= 0;
      const bool retry = leader && first < 0 && att_base + (uint32_t)A < a.attempt_limit;
      const bool surv = leader && first < 0 && !retry;
      if (leader && (a.n_drawn || a.gal_acc)) {
        const long gal = (long)(slot / (uint32_t)a.S);
        // (the caller pre-counts ONE attempt per slot; a first attempt that ran with speculation may have used more)
        const int used = (first >= 0 ? first + 1 : (int)tried) - (att_base == 0u ? 1 : 0);
        if (a.n_drawn && used > 0) sf_sat_add(&a.n_drawn[gal], used);
        if (hit && a.gal_acc && att_base >= 64u) atomicAdd(&a.gal_acc[gal], 1);  // progress past the 64th attempt
      }
      if (retry) {
        const unsigned int pos = atomicAdd(&ctrl[2], 1u);
        ctrl[SF_Q_HDR + 2 * IPW + pos] = slot;
        ctrl[SF_Q_HDR + 3 * IPW + pos] = att_base + (uint32_t)A;
      }
      if (surv) {
        const unsigned int pos = atomicAdd(&ctrl[3], 1u);
        ctrl[SF_Q_HDR + 4 * IPW + pos] = slot;
      }
      const unsigned int n_res = (unsigned)__popcll(__ballot(hit || surv));
      const unsigned int n_ev = (unsigned)__popcll(__ballot(valid && g4 == 0));
      const unsigned int n_r0 = (unsigned)__popcll(__ballot(leader && first < 0 && att_base == 0u));
      if ((threadIdx.x & 63) == 0) {
        if (n_res) atomicAdd(&ctrl[4], n_res);
        if (n_ev) atomicAdd(&ctrl[5], n_ev);
        if (n_r0) atomicAdd(&ctrl[6], n_r0);
      }
      if (TPW > 1) {
        const f32x4 tu = u_t[0];
#pragma unroll
        for (int q = 0; q + 1 < TPW; ++q) u_t[q] = u_t[q + 1];
        u_t[TPW - 1] = tu;
      }
    }
#ifdef SF_Q_STATS
    qs_last_work = __builtin_amdgcn_s_memrealtime();
    qs_ph[qs_o + 4] += qs_last_work - qs_t_mark;
    if (a.qtrace && threadIdx.x == 0 && qs_iters <= (2048u * 256u) / gridDim.x)
      a.qtrace[((size_t)blockIdx.x * ((2048u * 256u) / gridDim.x) + qs_iters - 1) * 4 + 3] = (uint32_t)qs_last_work;
#endif
  }
}
#undef u_cur
#undef gal_cur

// Find / resolve launches of the deep tail (sf_api.hip: the slots that used up the persistent windows) on the sampler's OWN
// arithmetic and cost per evaluation: the unrolled split-bf16 pass sequence of k_maf_samp16<.., DD> without the queue.
//   find    (a.best):     item i = attempt a.attempt + (i & (A - 1)) of listed slot i >> log2 A; an accepted attempt only lowers
//                         best[slot index] (atomic min)
//   resolve (a.att_list): item i = attempt att_list[i] of listed slot i (0xffffffff: none): writes the draw, or lists the slot
//                         as still open
//   count   (a.count):    item i = first attempt of slot i: += 1 per accepted draw of its galaxy (sf_flow_acceptance)
// A catalogue of 1e5 galaxies spends two thirds of its evaluations here (a few galaxies of acceptance ~1e-4 x 1 000 slots x
// ~1e4 attempts); on the fp32 kernel k_maf_inv16 those ran at 0.6 of the sampler's rate.  Table path, aligned placement with
// one degree group per tile (the shapes of the unrolled sampler); two tiles of 16 items per wave and staged transform.
template <int NB, int DD, int PREC = 0>
__global__ __launch_bounds__(256, 4) void k_maf_find16s(SfDev m, SfSampleArgsHost a) {
  using HID = SfHid16<PREC>;
  SfFix16<NB, DD>::apply(m);  // (the launcher only picks this kernel when the packer's offsets are these)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = lane & 15, g4 = lane >> 4;
  float* ecb = sf_lds16 + m.t16_a_tab + HID::lds_floats(m, true);
  if (threadIdx.x < 16) {  // per-slot constants of the epilogue (see k_maf_samp16)
    const int p = threadIdx.x;
    const bool on = p < m.D;
    const int td = on ? (int)m.cst[m.c_tdim + p] : 0;
    ecb[p * 5 + 0] = on ? m.cst[m.c_pshift + p] : 0.f;
    ecb[p * 5 + 1] = on ? __builtin_amdgcn_rcpf(m.cst[m.c_pscale + p]) : 0.f;
    ecb[p * 5 + 2] = (on && a.lo) ? a.lo[td] : -3.4e38f;
    ecb[p * 5 + 3] = (on && a.lo) ? a.hi[td] : 3.4e38f;
    reinterpret_cast<int*>(ecb)[p * 5 + 4] = td;
  }
  const int NT = m.nT16;
  f32x4 u_cur, u_oth;
  long gal_cur = 0, gal_oth = 0;
#pragma unroll
  for (int j = 1; j >= 0; --j) {
    const long item = ((long)blockIdx.x * 8 + j * 4 + wave) * 16 + s;
    const long it = item < a.n_items ? item : a.n_items - 1;
    const long ps = it >> a.log2_attempts;
    const uint32_t slot = a.z_in ? (uint32_t)it : (a.slots ? a.slots[ps] : (uint32_t)(a.slot_base + ps));
    const uint32_t att = a.att_list ? a.att_list[ps] : a.attempt + (uint32_t)(it & ((1L << a.log2_attempts) - 1));
    float z4[4];
    if (a.z_in) {   // parity hook: given noise, item = row of z / x (the context table was built for those rows)
#pragma unroll
      for (int r = 0; r < 4; ++r) z4[r] = (4 * g4 + r < m.D) ? a.z_in[it * m.D + 4 * g4 + r] : 0.f;
    } else {
      sf_normal4(a.k0, a.k1, (uint64_t)slot + a.rng_slot_offset, att, (uint32_t)g4, z4);
    }
    f32x4 u;
#pragma unroll
    for (int r = 0; r < 4; ++r) u[r] = (4 * g4 + r < m.D) ? z4[r] : 0.f;
    if (j == 1) { u_oth = u; gal_oth = a.z_in ? it : (long)(slot / (uint32_t)a.S); }
    else { u_cur = u; gal_cur = a.z_in ? it : (long)(slot / (uint32_t)a.S); }
  }
  typename HID::State S;
  S.tab = true;
  S.xr = nullptr;
  for (int t = m.T - 1; t >= 0; --t) {
    const int dsl = (int)m.cst[m.c_dslot + t * SF_DMAX + s];
    S.c0p = m.ctab + ((size_t)gal_cur * m.T + t) * m.ctab_R + (PREC == 2 ? m.nT16 * 16 : 0);
    sf_c0_prefetch(S, 0, g4);
    __syncthreads();
    sf_stage16<PREC, true>(m, t, m.t16_a_tab, wave);
    __syncthreads();
    const float* tp = sf_lds16;
    const void* tpB = sf_lds16 + m.t16_a_tab;
#pragma unroll 1
    for (int j = 0; j < 2; ++j) {
      if (((long)blockIdx.x * 8 + j * 4 + wave) * 16 < a.n_items) {  // (wave-uniform: the tile holds an item)
        if (j > 0) {
          S.c0p = m.ctab + ((size_t)gal_cur * m.T + t) * m.ctab_R + (PREC == 2 ? m.nT16 * 16 : 0);
          sf_c0_prefetch(S, 0, g4);
        }
        HID::template clear<true, true>(S);
#pragma unroll
        for (int r = 0; r < 4; ++r) S.ut[r] = 0.f;
        S.hdone = sf_ld4(tp + m.o16_bh + g4 * 4);
        {
          const int sl = __builtin_amdgcn_readlane(dsl, 0);
          const float av = tp[m.o16_hvb + 2 * sl], mv = tp[m.o16_hvb + 2 * sl + 1];
          const float sc = (m.scale_fn == 0 ? sf_softplus(av) : sf_sigmoid(av + 2.0f)) + m.eps;
          const float wv = sf_div(sf_slot16_own(u_cur, sl) - mv, sc);
#pragma unroll
          for (int r = 0; r < 4; ++r) S.ut[r] = (g4 == (sl >> 2) && r == (sl & 3)) ? wv : S.ut[r];
        }
        auto seq_pass = [&](auto otc) {
          constexpr int OT = decltype(otc)::value;
          const int sl = __builtin_amdgcn_readlane(dsl, OT + 1);
          HID::template pass<OT, NB, true, true>(m, tp, tpB, S, NT, sl, sf_slot16_own(u_cur, sl), lane, g4, OT + 2 < DD ? OT + 1 : -1);
        };
        seq_pass(std::integral_constant<int, 0>{});
        if constexpr (DD >= 3) seq_pass(std::integral_constant<int, 1>{});
        if constexpr (DD >= 4) seq_pass(std::integral_constant<int, 2>{});
        if constexpr (DD >= 5) seq_pass(std::integral_constant<int, 3>{});
        u_cur = S.ut;
      }
      { const f32x4 tu = u_cur; u_cur = u_oth; u_oth = tu; }
      { const long tg = gal_cur; gal_cur = gal_oth; gal_oth = tg; }
    }
  }
#pragma unroll 1
  for (int j = 0; j < 2; ++j) {
    const long item = ((long)blockIdx.x * 8 + j * 4 + wave) * 16 + s;
    const bool valid = item < a.n_items;
    const long it = valid ? item : a.n_items - 1;
    const long ps = it >> a.log2_attempts;
    const uint32_t att = a.att_list ? a.att_list[ps] : a.attempt + (uint32_t)(it & ((1L << a.log2_attempts) - 1));
    float th[4];
    int tdc[4];
    bool ok = true;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* e = ecb + (4 * g4 + r) * 5;
      th[r] = (u_cur[r] - e[0]) * e[1];
      tdc[r] = reinterpret_cast<const int*>(e)[4];
      ok = ok && (fabsf(th[r]) <= 3.0e38f) && (th[r] >= e[2]) && (th[r] <= e[3]);
    }
    if (a.att_list && att == 0xffffffffu) ok = false;  // no attempt to resolve: straight to the open list
    const unsigned long long okb = __ballot(ok);
    const uint32_t acc16 = (uint32_t)(okb & (okb >> 16) & (okb >> 32) & (okb >> 48) & 0xffffull) & (uint32_t)(__ballot(valid) & 0xffffull);
    const bool accepted = (acc16 >> s) & 1u;
    if (a.z_in) {   // every row is written (no box: lo / hi are null)
      if (valid) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * g4 + r < m.D) a.out[item * m.D + tdc[r]] = th[r];
      }
    } else if (a.best) {
      if (accepted && g4 == 0) atomicMin(&a.best[ps], att);
    } else if (a.count) {  // acceptance counts (leakage correction): item = draw item % S of galaxy item / S
      const long gal = (long)((uint32_t)(a.slots ? a.slots[ps] : (uint32_t)(a.slot_base + ps)) / (uint32_t)a.S);
      const long g_first = __shfl(gal, 0, 64), g_last = __shfl(gal, 15, 64);
      if (g_first == g_last) {
        if (lane == 0 && acc16) atomicAdd(&a.count[g_first], (int)__popc(acc16));
      } else if (accepted && g4 == 0) {
        atomicAdd(&a.count[gal], 1);
      }
    } else if (valid) {  // resolve: one item per listed slot
      const uint32_t slot = a.slots ? a.slots[ps] : (uint32_t)(a.slot_base + ps);
      if (accepted) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * g4 + r < m.D) sf_out_store(a, (size_t)slot * m.D + tdc[r], th[r]);
      } else if (g4 == 0) {
        const uint32_t pos = atomicAdd(a.n_rejected, 1u);
        a.rejected[pos] = slot;
      }
    }
    { const f32x4 tu = u_cur; u_cur = u_oth; u_oth = tu; }
  }
}

template <int NB, int DD, int PREC>
static hipError_t sf_launch_find16s(const SfDev& m, const SfSampleArgsHost& a, hipStream_t st) {
  static SfAttrCache attr;
  const size_t sh = ((size_t)m.t16_a_tab + (size_t)SfHid16<PREC>::lds_floats(m, true)) * sizeof(float) + 80 * sizeof(float);
  int attr_dev;
  if (attr.need(attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)k_maf_find16s<NB, DD, PREC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr.set(attr_dev);
  }
  hipLaunchKernelGGL((k_maf_find16s<NB, DD, PREC>), dim3((unsigned)((a.n_items + 127) / 128)), dim3(256), sh, st, m, a);
  return hipGetLastError();
}

// Parity hook of the persistent sampler's ARITHMETIC: theta = inverse(z | x) from GIVEN noise through exactly the pass
// functions k_maf_samp16 runs (sf_pass16b / sf_pass16b_span: hidden H x H blocks as split-bf16 x3, everything else
// fp32), so that the sampler's precision can be asserted against the fp64 oracle draw for draw, with no Philox and no
// rejection in between (sf_flow_inverse_from_noise_sampler; tests/test_gpu_parity.py).  One wave = 16 rows of z.
template <int NB, bool SPAN, bool HM, int PREC = 0>
__global__ __launch_bounds__(256, 3) void k_maf_inv16b(SfDev m, const float* __restrict__ z, const float* __restrict__ x,
                                                       long n, float* __restrict__ out) {
  using HID = SfHid16<PREC>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = lane & 15, g4 = lane >> 4;
  const long item = ((long)blockIdx.x * 4 + wave) * 16 + s;
  const bool valid = item < n;
  const long it = valid ? item : n - 1;
  f32x4 u;
#pragma unroll
  for (int r = 0; r < 4; ++r) u[r] = (4 * g4 + r < m.D) ? z[it * m.D + 4 * g4 + r] : 0.f;
  const int NT = m.nT16;
  uint32_t tile_bits = 0, lo_bits = 0;
#pragma unroll
  for (int q = 0; q < SF_DMAX; ++q) {
    tile_bits |= (uint32_t)(m.g16_tile[q] & 3) << (2 * q);
    lo_bits |= (uint32_t)(m.g16_lo[q] & 3) << (2 * q);
  }
  typename HID::State S;
  HID::template clear<false, false>(S);
  for (int t = m.T - 1; t >= 0; --t) {
    __syncthreads();
    sf_stage16<PREC, !SPAN>(m, t, m.t16_a, wave);
    __syncthreads();
    const float* tp = sf_lds16;
    const void* tpB = sf_lds16 + m.t16_a;
    S.c0p = nullptr;
    S.tab = false;
    S.xr = x + it * m.C;
    if (HM) S.hdone = sf_ld4(tp + m.o16_bh + g4 * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) S.ut[r] = 0.f;
    const int dsl = (int)m.cst[m.c_dslot + t * SF_DMAX + s];
    {
      const int sl = __builtin_amdgcn_readlane(dsl, 0);
      const float av = tp[m.o16_hvb + 2 * sl], mv = tp[m.o16_hvb + 2 * sl + 1];
      const float sc = (m.scale_fn == 0 ? sf_softplus(av) : sf_sigmoid(av + 2.0f)) + m.eps;
      const float wv = sf_div(sf_slot16(u, sl, lane) - mv, sc);
#pragma unroll
      for (int r = 0; r < 4; ++r) S.ut[r] = (g4 == (sl >> 2) && r == (sl & 3)) ? wv : S.ut[r];
    }
    for (int p = 2; p <= m.D; ++p) {
      const int sl = __builtin_amdgcn_readlane(dsl, p - 1);
      const float u_sl = sf_slot16(u, sl, lane);
      const uint32_t hi_t = (tile_bits >> (2 * (p - 1))) & 3u;
      const uint32_t lo_t = SPAN ? (lo_bits >> (2 * (p - 1))) & 3u : hi_t;
      const int nx = p < m.D ? (int)((tile_bits >> (2 * p)) & 3u) : -1;
      switch (lo_t * 4 + hi_t) {
        case 0: HID::template pass<0, NB, !SPAN, HM>(m, tp, tpB, S, NT, sl, u_sl, lane, g4, nx); break;
        case 5: HID::template pass<1, NB, !SPAN, HM>(m, tp, tpB, S, NT, sl, u_sl, lane, g4, nx); break;
        case 10: HID::template pass<2, NB, !SPAN, HM>(m, tp, tpB, S, NT, sl, u_sl, lane, g4, nx); break;
        case 15: HID::template pass<3, NB, !SPAN, HM>(m, tp, tpB, S, NT, sl, u_sl, lane, g4, nx); break;
        case 1: if constexpr (SPAN) HID::template span<0, 1, NB>(m, tp, tpB, S, NT, sl, u_sl, lane, g4); break;
        case 2: if constexpr (SPAN) HID::template span<0, 2, NB>(m, tp, tpB, S, NT, sl, u_sl, lane, g4); break;
        case 3: if constexpr (SPAN) HID::template span<0, 3, NB>(m, tp, tpB, S, NT, sl, u_sl, lane, g4); break;
        case 6: if constexpr (SPAN) HID::template span<1, 2, NB>(m, tp, tpB, S, NT, sl, u_sl, lane, g4); break;
        case 7: if constexpr (SPAN) HID::template span<1, 3, NB>(m, tp, tpB, S, NT, sl, u_sl, lane, g4); break;
        default: if constexpr (SPAN) HID::template span<2, 3, NB>(m, tp, tpB, S, NT, sl, u_sl, lane, g4); break;
      }
    }
    u = S.ut;
  }
  if (valid) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = 4 * g4 + r;
      if (p < m.D) out[item * m.D + (int)m.cst[m.c_tdim + p]] = sf_div(u[r] - m.cst[m.c_pshift + p], m.cst[m.c_pscale + p]);
    }
  }
}

template <int NB, bool SPAN, bool HM, int PREC>
static hipError_t sf_launch16b_hook_p(const SfDev& m, const float* z, const float* x, long n, float* out, hipStream_t st) {
  static SfAttrCache attr;
  const size_t sh = ((size_t)m.t16_a + (size_t)SfHid16<PREC>::lds_floats(m, !SPAN)) * sizeof(float);
  int attr_dev;
  if (attr.need(attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)k_maf_inv16b<NB, SPAN, HM, PREC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr.set(attr_dev);
  }
  hipLaunchKernelGGL((k_maf_inv16b<NB, SPAN, HM, PREC>), dim3((unsigned)((n + 63) / 64)), dim3(256), sh, st, m, z, x, n, out);
  return hipGetLastError();
}
template <int NB, bool SPAN, bool HM>
static hipError_t sf_launch16b_hook(const SfDev& m, const float* z, const float* x, long n, float* out, hipStream_t st) {
  return sf_sampler_fp32_for(SF_MAF) ? sf_launch16b_hook_p<NB, SPAN, HM, 1>(m, z, x, n, out, st)
                                     : sf_launch16b_hook_p<NB, SPAN, HM, 0>(m, z, x, n, out, st);
}
// head rows on the matrix pipe (aligned placement with the head tile in the image: D <= 8); SF_HEAD_MFMA=0 keeps the
// per-lane dot products (A-B runs)
static bool sf_maf16_head_mfma(const SfDev& m) {
  static int env = -1;
  if (env < 0) { const char* e = std::getenv("SF_HEAD_MFMA"); env = e ? std::atoi(e) : 1; }
  return env != 0 && !m.m16_span && m.o16_wh >= 0;
}
// false when the flow has no 16-row persistent sampler (then the sampler IS the 32-row fp32 path and sf_flow_inverse_from_noise
// covers it)
bool sf_maf16b_available(const SfDev& m) {
  return m.kind == SF_MAF && m.m16_ok && !m.hidden_bf16 && m.packed16 != nullptr &&
         (m.packed16B != nullptr || sf_sampler_fp32_for(SF_MAF));
}
hipError_t sf_launch_maf_inv16b_hook(const SfDev& m, const float* z, const float* x, long n, float* out, hipStream_t st) {
  if (m.m16_span) return m.NB == 1 ? sf_launch16b_hook<1, true, false>(m, z, x, n, out, st) : sf_launch16b_hook<2, true, false>(m, z, x, n, out, st);
  if (sf_maf16_head_mfma(m)) return m.NB == 1 ? sf_launch16b_hook<1, false, true>(m, z, x, n, out, st) : sf_launch16b_hook<2, false, true>(m, z, x, n, out, st);
  return m.NB == 1 ? sf_launch16b_hook<1, false, false>(m, z, x, n, out, st) : sf_launch16b_hook<2, false, false>(m, z, x, n, out, st);
}

// Per-galaxy context table of the 16-row path: tab[gal][t][row] = b0 + bc + Wc e(x_gal), rows in tile order.
// One wave = 16 galaxies; same MFMA sequence as the in-kernel evaluation, so the sampler's draws do not change.
__global__ __launch_bounds__(256) void k_maf_ctab16(SfDev m, const float* __restrict__ x, long M, float* __restrict__ tab) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = lane & 15, g4 = lane >> 4;
  const long gal = ((long)blockIdx.x * 4 + wave) * 16 + s;
  if (((long)blockIdx.x * 4 + wave) * 16 >= M) return;
  const bool valid = gal < M;
  const float* xr = x + (valid ? gal : M - 1) * m.C;
  const int NT = m.nT16;
  for (int t = 0; t < m.T; ++t) {
    const float* tp = m.packed16 + (size_t)t * m.t16_stride;
    f32x4 c0[4];
#pragma unroll
    for (int ot = 0; ot < 4; ++ot)
      if (ot < NT) c0[ot] = sf_ld4(tp + m.o16_b0 + (ot * 4 + g4) * 4);
    for (int ic = 0; ic < m.nC16; ++ic) {
      f32x4 ct;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rho = ic * 16 + 4 * g4 + r;
        const bool ok = rho < m.C;
        const int rr = ok ? rho : 0;
        const float v = sf_div(xr[rr] - m.cst[m.c_xmean + rr], m.cst[m.c_xstd + rr]);
        ct[r] = ok ? v : 0.f;
      }
#pragma unroll
      for (int ot = 0; ot < 4; ++ot)
        if (ot < NT) c0[ot] = sf_mma16(sf_w16(tp + m.o16_wc, m.nC16, ot, ic, lane), ct, c0[ot]);
    }
    if (valid) {
#pragma unroll
      for (int ot = 0; ot < 4; ++ot)
        if (ot < NT)
          *reinterpret_cast<f32x4*>(tab + ((size_t)gal * m.T + t) * m.ctab_R + ot * 16 + 4 * g4) = c0[ot];
    }
    if (m.o16_wp >= 0) {   // fused first layer (sf_pass16g): c0' = b1 + (W1 o M) c0, behind c0 in the row
      f32x4 c1[4];
#pragma unroll
      for (int ot = 0; ot < 4; ++ot)
        if (ot < NT) {
          c1[ot] = sf_ld4(tp + m.o16_bk[0] + (ot * 4 + g4) * 4);
#pragma unroll
          for (int it = 0; it < 4; ++it)
            if (it < NT) c1[ot] = sf_mma16(sf_w16(tp + m.o16_wk[0], NT, ot, it, lane), c0[it], c1[ot]);
        }
      if (valid) {
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
          if (ot < NT)
            *reinterpret_cast<f32x4*>(tab + ((size_t)gal * m.T + t) * m.ctab_R + NT * 16 + ot * 16 + 4 * g4) = c1[ot];
      }
    }
  }
}
// W' = (W1 o M)(W0 o M0) of every transform, from the packed fp32 image into its o16_wp block (fp64 sums, one rounding): lane l of
// tile ot holds W'[ot*16 + (l & 15)][4 (l >> 4) + r], the fragment layout of o16_w0.  Masked weights are structural zeros of the
// image, and the packed first block carries the tanh pre-scale (SF_PACK_TANH_SCALE): W' and c0' inherit both.
__global__ __launch_bounds__(256) void k_maf_fuse16(SfDev m) {
  const int t = blockIdx.x, ot = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int NT = m.nT16;
  if (ot >= NT) return;
  float* tp = const_cast<float*>(m.packed16) + (size_t)t * m.t16_stride;
  const int row = l & 15, sg = l >> 4;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int it = 0; it < NT; ++it)
    for (int kk = 0; kk < 16; ++kk) {
      const float wk = tp[m.o16_wk[0] + ((ot * NT + it) * 64 + row + 16 * (kk >> 2)) * 4 + (kk & 3)];   // W1[ot*16 + row][it*16 + kk]
      const float* w0 = tp + m.o16_w0 + (it * 64 + kk + 16 * sg) * 4;                                    // W0[it*16 + kk][4 sg + r]
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] += (double)wk * (double)w0[r];
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) tp[m.o16_wp + (ot * 64 + l) * 4 + r] = (float)acc[r];
}
hipError_t sf_launch_maf_fuse16(const SfDev& m, hipStream_t st) {
  hipLaunchKernelGGL(k_maf_fuse16, dim3((unsigned)m.T), dim3(256), 0, st, m);
  return hipGetLastError();
}
hipError_t sf_launch_maf_ctab16(const SfDev& m, const float* x, long M, float* tab, hipStream_t st) {
  hipLaunchKernelGGL(k_maf_ctab16, dim3((unsigned)((M + 63) / 64)), dim3(256), 0, st, m, x, M, tab);
  return hipGetLastError();
}

// SF_MAF16=0 disables the path (diagnostics / A-B runs).  A = 32 retry rounds stay on the 32-row kernel.
// Arithmetic of the samplers' hidden blocks (process-wide; sf_set_sampler_fp32 / environment SF_SAMPLER_FP32):
//   1  fp32 everywhere: MAF k_maf_samp16<.., PREC = 1> (v_mfma_f32_16x16x4_f32), NSF the fp32 image
//   0  split bf16 x3 where the flow has such an image (the opt-in fast mode of round 2-4)
//  -1  (unset) per flow kind: MAF fp32 -- BASELINE configs[1] says fp32, and the split products move log p(draw) by up to
//      1.3e-3 against the north-star tolerance of 1e-4 -- NSF split (its draws meet the fp64 oracle as closely as the
//      all-fp32 kernels do: tests/test_gpu_parity.py::test_sampler_arithmetic_from_given_noise)
static int g_sampler_fp32 = -2;
void sf_sampler_fp32_set(int on) { g_sampler_fp32 = on < 0 ? -1 : (on ? 1 : 0); }
int sf_sampler_fp32_get() {
  if (g_sampler_fp32 == -2) { const char* e = std::getenv("SF_SAMPLER_FP32"); g_sampler_fp32 = (e && *e) ? (std::atoi(e) != 0 ? 1 : 0) : -1; }
  return g_sampler_fp32;
}
int sf_sampler_fp32_for(int kind) {
  const int g = sf_sampler_fp32_get();
  return g < 0 ? (kind == SF_MAF ? 1 : 0) : g;
}
bool sf_maf16_enabled(const SfDev& m, const SfSampleArgsHost& a) {
  static int env = -1;
  if (env < 0) {
    const char* e = std::getenv("SF_MAF16");
    env = e ? std::atoi(e) : 1;
  }
  return env != 0 && m.kind == SF_MAF && m.m16_ok && !m.hidden_bf16 && m.packed16 != nullptr &&
         (a.attempts_per_slot <= 16 || a.best != nullptr);  // (find mode has no in-tile attempt groups)
}

// workgroups that fit the chip at once (persistent launches): `cap` per CU by registers and LDS (4 with the compact
// image of the aligned placement at 128 VGPRs, 3 for the contiguous one)
static int sf_resident_blocks16(const void* fn, size_t sh, int cap) {
  int dev = 0, cus = 256, per = cap;
  if (hipGetDevice(&dev) == hipSuccess) {
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount;
  }
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, 256, sh) == hipSuccess && occ > 0) per = occ < cap ? occ : cap;
  return cus * per;
}

// persistent sampler: no more workgroups than the chip holds (more would only queue behind the spinning ones)
template <int NB, bool SPAN, bool HM, int TPW, int DD, int PREC>
static hipError_t sf_launch16q_p(const SfDev& m, const SfSampleArgsHost& a, hipStream_t st) {
  static SfAttrCache attr;
  static SfResidentCache rcache;
  const size_t sh = ((size_t)(m.ctab ? m.t16_a_tab : m.t16_a) + (size_t)SfHid16<PREC>::lds_floats(m, !SPAN)) * sizeof(float) +
                    (SF_Q_WORDS(64 * TPW) + 96) * sizeof(unsigned int);
  if (sh > 160 * 1024) return hipErrorInvalidValue;
  int attr_dev;
  if (attr.need(attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)k_maf_samp16<NB, SPAN, HM, TPW, DD, PREC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr.set(attr_dev);
  }
  int resident = 0, cur_dev = 0;
  (void)hipGetDevice(&cur_dev);
  if (!rcache.get(cur_dev, sh, resident)) {
    resident = sf_resident_blocks16((const void*)k_maf_samp16<NB, SPAN, HM, TPW, DD, PREC>, sh, SPAN ? 3 : (PREC == 2 ? SF_SAMP16_WG_FUSED : 4));
    rcache.put(cur_dev, sh, resident);
  }
  long grid = (a.n_items + 64 * TPW - 1) / (64 * TPW);
  if (grid > resident) grid = resident;
  SfSamp16Args args;
  args.m = m;
  args.a = a;
  hipLaunchKernelGGL((k_maf_samp16<NB, SPAN, HM, TPW, DD, PREC>), dim3((unsigned)grid), dim3(256), sh, st, args);
  return hipGetLastError();
}
template <int NB, bool SPAN, bool HM, int TPW, int DD = 0>
static hipError_t sf_launch16q(const SfDev& m, const SfSampleArgsHost& a, hipStream_t st) {
  if (sf_sampler_fp32_for(SF_MAF)) return sf_launch16q_p<NB, SPAN, HM, TPW, DD, 1>(m, a, st);
  if (!m.packed16B) return hipErrorInvalidValue;
  return sf_launch16q_p<NB, SPAN, HM, TPW, DD, 0>(m, a, st);
}
template <int NB, bool SPAN>
static hipError_t sf_launch16(const SfDev& m, const SfSampleArgsHost& a, hipStream_t st) {
  static SfAttrCache attr;
  const size_t sh = (size_t)m.t16_stride * sizeof(float);
  int attr_dev;
  if (attr.need(attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)k_maf_inv16<NB, SPAN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr.set(attr_dev);
  }
  const long per_block = 4L * 16;
  hipLaunchKernelGGL((k_maf_inv16<NB, SPAN>), dim3((unsigned)((a.n_items + per_block - 1) / per_block)), dim3(256), sh, st, m, a);
  return hipGetLastError();
}
// SF_TPW=1: one draw tile per wave and iteration (64 items per workgroup iteration) instead of two (A-B runs)
static int sf_maf16_tpw() {
  static int env = -1;
  if (env < 0) { const char* e = std::getenv("SF_TPW"); env = (e && std::atoi(e) == 1) ? 1 : 2; }
  return env;
}
// the unrolled-pass kernels apply when degree p - 1 sits alone in tile p - 2 (SF_SEQ=0: the dispatching kernel, A-B runs)
static int sf_maf16_seq_d(const SfDev& m) {
  static int env = -1;
  if (env < 0) { const char* e = std::getenv("SF_SEQ"); env = e ? std::atoi(e) : 1; }
  if (!env || !m.ctab || m.m16_span || m.D < 3 || m.D > 5 || m.nT16 != m.D - 1) return 0;
  for (int p = 2; p <= m.D; ++p)
    if (m.g16_tile[p - 1] != p - 2) return 0;
  bool fits = false;  // the image offsets the unrolled kernels hard-wire
  if (m.NB == 1) fits = m.D == 3 ? SfFix16<1, 3>::matches(m) : (m.D == 4 ? SfFix16<1, 4>::matches(m) : SfFix16<1, 5>::matches(m));
  else if (m.NB == 2) fits = m.D == 3 ? SfFix16<2, 3>::matches(m) : (m.D == 4 ? SfFix16<2, 4>::matches(m) : SfFix16<2, 5>::matches(m));
  return fits ? m.D : 0;
}
// The fused first layer (PREC = 2, sf_pass16g) applies where the fp32 unrolled kernels do and the context table carries the c0'
// rows (SF_FUSE=0: the two-layer fp32 form, A-B runs).  Returns D (3..5) or 0.
int sf_maf16_fused_d(const SfDev& m) {
  static int env = -1;
  if (env < 0) { const char* e = std::getenv("SF_FUSE"); env = e ? std::atoi(e) : 1; }
  if (!env || !sf_sampler_fp32_for(SF_MAF) || !m.ctab || m.o16_wp < 0 || m.ctab_R != 2 * m.nT16 * 16 || !sf_maf16_head_mfma(m)) return 0;
  return sf_maf16_seq_d(m);
}
template <int NB, bool SPAN, bool HM>
static hipError_t sf_launch16q_t(const SfDev& m, const SfSampleArgsHost& a, hipStream_t st) {
  if (sf_maf16_tpw() == 1) return sf_launch16q<NB, SPAN, HM, 1>(m, a, st);
  if constexpr (HM && !SPAN) {
    switch (sf_maf16_fused_d(m)) {
      case 3: return sf_launch16q_p<NB, SPAN, HM, 2, 3, 2>(m, a, st);
      case 4: return sf_launch16q_p<NB, SPAN, HM, 2, 4, 2>(m, a, st);
      case 5: return sf_launch16q_p<NB, SPAN, HM, 2, 5, 2>(m, a, st);
      default: break;
    }
    // (round 5, fp32 kernels: FOUR tiles per wave and staged transform -- fetch, staging and prologue once per 256 draws -- measured
    //  2.74 ms per catalogue against 2.62 with two: the coarser iterations cost the tail more than the dense phase saves)
    switch (sf_maf16_seq_d(m)) {
      case 3: return sf_launch16q<NB, SPAN, HM, 2, 3>(m, a, st);
      case 4: return sf_launch16q<NB, SPAN, HM, 2, 4>(m, a, st);
      case 5: return sf_launch16q<NB, SPAN, HM, 2, 5>(m, a, st);
      default: break;
    }
  }
  return sf_launch16q<NB, SPAN, HM, 2>(m, a, st);
}
// parity hook of the fused pass functions: theta = inverse(z | x) through k_maf_find16s<.., PREC = 2> in its given-noise mode (the
// context table must have been built for the rows of x: item i reads table row i)
hipError_t sf_launch_maf_find16_zin(const SfDev& m, const SfSampleArgsHost& a, hipStream_t st) {
  const int dd = sf_maf16_fused_d(m);
#define SF_ZIN_CASE(NBV, DDV) if (m.NB == NBV && dd == DDV) return sf_launch_find16s<NBV, DDV, 2>(m, a, st);
  SF_ZIN_CASE(1, 3) SF_ZIN_CASE(1, 4) SF_ZIN_CASE(1, 5) SF_ZIN_CASE(2, 3) SF_ZIN_CASE(2, 4) SF_ZIN_CASE(2, 5)
#undef SF_ZIN_CASE
  return hipErrorInvalidValue;
}
hipError_t sf_launch_maf_inv16(const SfDev& m, const SfSampleArgsHost& a, hipStream_t st) {
  if (a.q) {
    if (m.m16_span) return m.NB == 1 ? sf_launch16q_t<1, true, false>(m, a, st) : sf_launch16q_t<2, true, false>(m, a, st);
    if (sf_maf16_head_mfma(m)) return m.NB == 1 ? sf_launch16q_t<1, false, true>(m, a, st) : sf_launch16q_t<2, false, true>(m, a, st);
    return m.NB == 1 ? sf_launch16q_t<1, false, false>(m, a, st) : sf_launch16q_t<2, false, false>(m, a, st);
  }
  // find / resolve launches of the deep tail: the unrolled split-bf16 kernel where the sampler itself runs one
  // (SF_FIND16S=0: the fp32 kernel, A-B runs)
  const bool f32 = sf_sampler_fp32_for(SF_MAF) != 0;
  if ((a.best || a.att_list || a.count) && !a.z_in && m.ctab && (f32 || m.packed16B) && sf_maf16_head_mfma(m)) {
    static int env = -1;
    if (env < 0) { const char* e = std::getenv("SF_FIND16S"); env = e ? std::atoi(e) : 1; }
    const int dd = env ? sf_maf16_seq_d(m) : 0;
    const bool fused = env && sf_maf16_fused_d(m) > 0;
#define SF_FIND_CASE(NBV, DDV) \
    if (m.NB == NBV && dd == DDV) return fused ? sf_launch_find16s<NBV, DDV, 2>(m, a, st) : (f32 ? sf_launch_find16s<NBV, DDV, 1>(m, a, st) : sf_launch_find16s<NBV, DDV, 0>(m, a, st));
    SF_FIND_CASE(1, 3) SF_FIND_CASE(1, 4) SF_FIND_CASE(1, 5)
    SF_FIND_CASE(2, 3) SF_FIND_CASE(2, 4) SF_FIND_CASE(2, 5)
#undef SF_FIND_CASE
  }
  if (m.m16_span) return m.NB == 1 ? sf_launch16<1, true>(m, a, st) : sf_launch16<2, true>(m, a, st);
  return m.NB == 1 ? sf_launch16<1, false>(m, a, st) : sf_launch16<2, false>(m, a, st);
}
