// sf_inst_templates.h -- gfx950 kernel templates of the inference direction on the 32-row register-tile engine
// (log_prob, inverse, sampler, NSF context table).  One wave handles NS tiles of 32 samples end to end.
// LDSW = true: the workgroup stages one transform's operand image in LDS per transform (two barriers) and the
// waves read their MFMA A fragments from there; LDSW = false: waves are independent and weights stream from L2.
// The default MAF sampler is the 16-row kernel in sf_maf16.hip; k_inverse<MafOps> is its fallback.
#pragma once
#include <hip/hip_runtime.h>

#include "sf_flows.h"
#include "sf_internal.h"
#include "sf_rng.h"

#define SF_LOG_2PI 1.8378770664093453f

// ---------------------------------------------------------------------------------------------
// log_prob:  out[i] = log N(z;0,I) + logdet          ref: custom_runner.py:604 (via [UPSTREAM] Flow.log_prob)
// ---------------------------------------------------------------------------------------------
extern __shared__ float sf_lds_image[];

template <class Ops, int NS, bool LDSW>
__global__ __launch_bounds__(LDSW ? 512 : 256) void k_logprob(SfDev m, const float* __restrict__ theta,
                                                              const float* __restrict__ x, long B,
                                                              float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const long base = ((long)blockIdx.x * (blockDim.x >> 6) + wave) * (32 * NS);
  if (!LDSW && base >= B) return;  // LDSW: every wave takes part in the staging barriers
  float u[NS][SF_DMAX];
  const float* xr[NS];
  float logdet[NS];
  long row[NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    row[ns] = base + ns * 32 + c;
    const long ii = row[ns] < B ? row[ns] : B - 1;
    xr[ns] = x + ii * m.C;
    logdet[ns] = m.logdet0;
#pragma unroll
    for (int p = 0; p < SF_DMAX; ++p) {
      u[ns][p] = 0.f;
      if (p < m.D) {
        const int td = (int)m.cst[m.c_tdim + p];
        u[ns][p] = theta[ii * m.D + td] * m.cst[m.c_pscale + p] + m.cst[m.c_pshift + p];
      }
    }
  }
  Ops::forward(m, u, xr, logdet, lane, sf_lds_image);
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < SF_DMAX; ++p)
      if (p < m.D) s += u[ns][p] * u[ns][p];
    const float lp = -0.5f * s - 0.5f * (float)m.D * SF_LOG_2PI + logdet[ns];
    if (h == 0 && row[ns] < B) out[row[ns]] = lp;
  }
}

// ---------------------------------------------------------------------------------------------
// inverse / sampler round / acceptance count
//   z_in != NULL : parity hook, item i uses z_in[i,:], context row i, writes theta[i,:], logdet[i]
//   else         : item i -> slot (slots[i] or slot_base+i), galaxy g = slot / S, noise from Philox;
//                  box test; accepted -> out[slot,:], rejected -> appended to rejected[]
//                  count != NULL: acceptance mode, no writes, count[g] += accepted
// ref: sbi_runner.py:6442 -> [UPSTREAM] DirectPosterior.sample / accept_reject_sample;
//      box predicate custom_runner.py:982-987
// ---------------------------------------------------------------------------------------------

template <class Ops, int NS, bool LDSW>
__global__ __launch_bounds__(LDSW ? 512 : 256) void k_inverse(SfDev m, SfSampleArgsHost a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const long base = ((long)blockIdx.x * (blockDim.x >> 6) + wave) * (32 * NS);
  if (!LDSW && base >= a.n_items) return;
  float u[NS][SF_DMAX];
  const float* xr[NS];
  float logdet[NS];
  long item[NS];
  uint64_t slot[NS];
  long gal[NS], ps_idx[NS];
  uint32_t att_mine[NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    ps_idx[ns] = 0; att_mine[ns] = 0;
    item[ns] = base + ns * 32 + c;
    const long it = item[ns] < a.n_items ? item[ns] : a.n_items - 1;
    logdet[ns] = 0.f;
#pragma unroll
    for (int p = 0; p < SF_DMAX; ++p) u[ns][p] = 0.f;
    if (a.z_in) {
      slot[ns] = (uint64_t)it;
      gal[ns] = it;
#pragma unroll
      for (int p = 0; p < SF_DMAX; ++p)
        if (p < m.D) u[ns][p] = a.z_in[it * m.D + p];
    } else {
      const long ps = it >> a.log2_attempts;  // listed slot; A (a power of two) consecutive items share it
      ps_idx[ns] = ps;
      slot[ns] = a.slots ? (uint64_t)a.slots[ps] : (uint64_t)(a.slot_base + ps);
      gal[ns] = (long)((uint32_t)slot[ns] / (uint32_t)a.S);  // slot ids fit 32 bits (checked by the API)
      const uint32_t att = a.att_list ? a.att_list[ps] : a.attempt + (uint32_t)(it & ((1L << a.log2_attempts) - 1));
      att_mine[ns] = att;
#pragma unroll
      for (int blk = 0; blk < SF_DMAX / 4; ++blk)
        if (blk * 4 < m.D) {
          float z4[4];
          sf_normal4(a.k0, a.k1, slot[ns] + a.rng_slot_offset, att, (uint32_t)blk, z4);
#pragma unroll
          for (int j = 0; j < 4; ++j) u[ns][blk * 4 + j] = (blk * 4 + j < m.D) ? z4[j] : 0.f;
        }
    }
    xr[ns] = a.x + gal[ns] * m.C;
  }
  // per-galaxy context table (sampling rounds only; wave-uniform choice)
  const bool use_tab = m.ctab != nullptr && a.z_in == nullptr;
  const float* cg[NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) cg[ns] = use_tab ? m.ctab + (size_t)gal[ns] * m.T * m.ctab_NV * m.ctab_R : nullptr;
  Ops::inverse(m, u, xr, logdet, lane, sf_lds_image, use_tab ? &cg : nullptr);
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const bool valid = item[ns] < a.n_items;
    float th[SF_DMAX];
    bool ok = true;
#pragma unroll
    for (int p = 0; p < SF_DMAX; ++p)
      if (p < m.D) {
        const int td = (int)m.cst[m.c_tdim + p];
        th[p] = (u[ns][p] - m.cst[m.c_pshift + p]) / m.cst[m.c_pscale + p];
        ok = ok && (fabsf(th[p]) <= 3.0e38f);  // finite (NaN compares false)
        if (a.lo) ok = ok && (th[p] >= a.lo[td]) && (th[p] <= a.hi[td]);
      }
    if (a.att_list && att_mine[ns] == 0xffffffffu) ok = false;  // no attempt to resolve: straight to the rejected list
    if (a.z_in) {
      if (valid && h == 0) {
#pragma unroll
        for (int p = 0; p < SF_DMAX; ++p)
          if (p < m.D) a.out[item[ns] * m.D + (int)m.cst[m.c_tdim + p]] = th[p];
        if (a.logdet_out) a.logdet_out[item[ns]] = logdet[ns] - m.logdet0;
      }
    } else if (a.best) {
      if (valid && h == 0 && ok) atomicMin(&a.best[ps_idx[ns]], att_mine[ns]);
    } else if (a.count) {
      const bool hit = valid && h == 0 && ok;
      const unsigned long long bal = __ballot(hit);
      const long g_first = __shfl(gal[ns], 0, 64);
      const long g_last = __shfl(gal[ns], 31, 64);
      if (g_first == g_last) {
        if (lane == 0 && bal) atomicAdd(&a.count[g_first], (int)__popcll(bal));
      } else if (hit) {
        atomicAdd(&a.count[gal[ns]], 1);
      }
    } else {
      // A consecutive lanes hold attempts att..att+A-1 of one slot: the lowest accepted one wins
      const int A = a.attempts_per_slot;
      const unsigned long long bal = __ballot(valid && h == 0 && ok);
      const int grp0 = (c / A) * A;
      const uint32_t gmask = (uint32_t)((bal >> grp0) & ((A >= 32) ? 0xffffffffull : ((1ull << A) - 1ull)));
      const int first = gmask ? (int)__builtin_ctz(gmask) : -1;
      const int me = c - grp0;
      if (valid && h == 0) {
        if (me == 0 && a.n_drawn && a.attempt > 0) atomicAdd(&a.n_drawn[gal[ns]], first >= 0 ? first + 1 : A);
        if (ok && me == first) {
#pragma unroll
          for (int p = 0; p < SF_DMAX; ++p)
            if (p < m.D) sf_out_store(a, (size_t)slot[ns] * m.D + (int)m.cst[m.c_tdim + p], th[p]);
        } else if (first < 0 && me == 0) {
          const uint32_t pos = atomicAdd(a.n_rejected, 1u);
          a.rejected[pos] = (uint32_t)slot[ns];
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Persistent sampler on the 32-row engine (NSF; MAF shapes the 16-row kernel does not take): the tile pipeline of
// k_inverse driven by the device work queue (sf_queue.h) -- ONE launch resolves every slot, first attempts and
// retries.  Sampler only.  Argument blocks are read through the laundered kernarg pointer (see k_maf_samp16).
// ---------------------------------------------------------------------------------------------
struct SfSampArgs {
  SfDev m;
  SfSampleArgsHost a;
};

template <class Ops, int NS, bool LDSW, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_sample_persist(SfSampArgs args_in, int ctrl_off) {
  constexpr int IPW = WPB * 32 * NS;
  const int wave = threadIdx.x >> 6;
  unsigned int* ctrl = reinterpret_cast<unsigned int*>(sf_lds_image + ctrl_off);
  unsigned int pf;
  sf_q_begin<IPW>(args_in.a, ctrl, pf);
#ifdef SF_Q_STATS
  unsigned int qs_iters = 0;
#endif
  for (;;) {
    const SfSampArgs* ap;
    {
      auto kp = __builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(kp));
      ap = (const SfSampArgs*)kp;
    }
    const SfDev& m = ap->m;
    const SfSampleArgsHost& a = ap->a;
    const int lane = (threadIdx.x & 63) + sf_opaque_zero();
    const int c = lane & 31, h = lane >> 5;
    if (!sf_q_fetch<IPW, 32>(a, ctrl, pf)) break;
#ifdef SF_Q_STATS
    if (a.qtrace && threadIdx.x == 0 && qs_iters < (2048u * 256u) / gridDim.x) {
      uint32_t* tr = a.qtrace + ((size_t)blockIdx.x * ((2048u * 256u) / gridDim.x) + qs_iters) * 4;
      tr[0] = (uint32_t)__builtin_amdgcn_s_memrealtime(); tr[1] = ctrl[0]; tr[2] = ctrl[1] | (ctrl[9] << 8);
    }
    ++qs_iters;
#endif
    float u[NS][SF_DMAX];
    const float* xr[NS];
    float logdet[NS];
    const float* cg[NS];
    const bool use_tab = m.ctab != nullptr;
    {
      const unsigned int n_ent = ctrl[0];
      const int lgA = (int)ctrl[1];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        const int wi = (wave * NS + ns) * 32 + c;
        const unsigned int e = (unsigned)wi >> lgA;
        const unsigned int ee = e < n_ent ? e : 0u;
        const uint32_t slot = ctrl[SF_Q_HDR + ee];
        const uint32_t att = ctrl[SF_Q_HDR + IPW + ee] + ((unsigned)wi & ((1u << lgA) - 1u));
        const long gal = (long)(slot / (uint32_t)a.S);
        logdet[ns] = 0.f;
#pragma unroll
        for (int p = 0; p < SF_DMAX; ++p) u[ns][p] = 0.f;
#pragma unroll
        for (int blk = 0; blk < SF_DMAX / 4; ++blk)
          if (blk * 4 < m.D) {
            float z4[4];
            sf_normal4(a.k0, a.k1, (uint64_t)slot + a.rng_slot_offset, att, (uint32_t)blk, z4);
#pragma unroll
            for (int j = 0; j < 4; ++j) u[ns][blk * 4 + j] = (blk * 4 + j < m.D) ? z4[j] : 0.f;
          }
        xr[ns] = a.x + gal * m.C;
        cg[ns] = use_tab ? m.ctab + (size_t)gal * m.T * m.ctab_NV * m.ctab_R : nullptr;
      }
    }
    // a wave whose tiles hold no item (sparse iterations of the tail) only takes part in the staging barriers
    const bool wave_active = __builtin_amdgcn_readfirstlane((unsigned)(wave * NS * 32) < (ctrl[0] << ctrl[1]) ? 1 : 0) != 0;
    Ops::inverse(m, u, xr, logdet, lane, sf_lds_image, use_tab ? &cg : nullptr, wave_active);
    const unsigned int n_ent = ctrl[0];
    const int lgA = (int)ctrl[1];
    const int A = 1 << lgA;
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      float th[SF_DMAX];
      bool ok = true;
#pragma unroll
      for (int p = 0; p < SF_DMAX; ++p)
        if (p < m.D) {
          const int td = (int)m.cst[m.c_tdim + p];
          th[p] = (u[ns][p] - m.cst[m.c_pshift + p]) / m.cst[m.c_pscale + p];
          ok = ok && (fabsf(th[p]) <= 3.0e38f);  // finite (NaN compares false)
          if (a.lo) ok = ok && (th[p] >= a.lo[td]) && (th[p] <= a.hi[td]);
        }
      const int wi = (wave * NS + ns) * 32 + c;
      const unsigned int e = (unsigned)wi >> lgA;
      const bool entry_ok = e < n_ent;
      const unsigned int ee = entry_ok ? e : 0u;
      const uint32_t slot = ctrl[SF_Q_HDR + ee];
      const uint32_t att_base = ctrl[SF_Q_HDR + IPW + ee];
      const uint32_t att = att_base + ((unsigned)wi & ((1u << lgA) - 1u));
      const bool valid = entry_ok && att < a.attempt_limit;
      // A consecutive lanes hold attempts att_base .. att_base+A-1 of one slot: the lowest accepted one wins
      const unsigned long long bal = __ballot(valid && h == 0 && ok);
      const int grp0 = (c / A) * A;
      const uint32_t gmask = (uint32_t)((bal >> grp0) & ((A >= 32) ? 0xffffffffull : ((1ull << A) - 1ull)));
      const int first = gmask ? (int)__builtin_ctz(gmask) : -1;
      const int me = c - grp0;
      if (valid && h == 0 && ok && me == first) {
        if (m.kind == SF_NSF && (m.D & 3) == 0) {
          // a coupling NSF keeps theta's column order (no permutations): the draw's row leaves as 16-byte pieces -- with the
          // dense order's runs of consecutive draws on consecutive lanes, whole 64-byte segments per wave-instruction
          if (a.out_f64) {
            double2* o2 = reinterpret_cast<double2*>(reinterpret_cast<double*>(a.out) + (size_t)slot * m.D);
#pragma unroll
            for (int p = 0; p < SF_DMAX; p += 2)
              if (p < m.D) o2[p >> 1] = make_double2((double)th[p], (double)th[p + 1]);
          } else {
            float4* o4 = reinterpret_cast<float4*>(a.out + (size_t)slot * m.D);
#pragma unroll
            for (int p = 0; p < SF_DMAX; p += 4)
              if (p < m.D) o4[p >> 2] = make_float4(th[p], th[p + 1], th[p + 2], th[p + 3]);
          }
        } else {
#pragma unroll
          for (int p = 0; p < SF_DMAX; ++p)
            if (p < m.D) sf_out_store(a, (size_t)slot * m.D + (int)m.cst[m.c_tdim + p], th[p]);
        }
      }
      const bool leader = entry_ok && h == 0 && me == 0;
      const uint32_t room = a.attempt_limit > att_base ? a.attempt_limit - att_base : 0u;
      const uint32_t tried = room < (uint32_t)A ? room : (uint32_t)A;
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
      const unsigned int n_ev = (unsigned)__popcll(__ballot(valid && h == 0));
      const unsigned int n_r0 = (unsigned)__popcll(__ballot(leader && first < 0 && att_base == 0u));
      if ((threadIdx.x & 63) == 0) {
        if (n_res) atomicAdd(&ctrl[4], n_res);
        if (n_ev) atomicAdd(&ctrl[5], n_ev);
        if (n_r0) atomicAdd(&ctrl[6], n_r0);
      }
    }
#ifdef SF_Q_STATS
    if (a.qtrace && threadIdx.x == 0 && qs_iters <= (2048u * 256u) / gridDim.x)
      a.qtrace[((size_t)blockIdx.x * ((2048u * 256u) / gridDim.x) + qs_iters - 1) * 4 + 3] = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
  }
}

// ---------------------------------------------------------------------------------------------
// NSF per-galaxy context table (sf_flow_prepare_context): tab[gal][t][v][row], rows in tile order
//   v = 0: bin + Win_c e(x);  v = 1+k: bg_k + Wg_k e(x)     (SfDev::ctab)
// One wave = 32 galaxies; weights stream from the global operand image (tiny kernel).
// ---------------------------------------------------------------------------------------------
template <int HT>
__global__ __launch_bounds__(256) void k_nsf_ctab(SfDev m, const float* __restrict__ x, long M, float* __restrict__ tab) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const long base = ((long)blockIdx.x * 4 + wave) * 32;
  if (base >= M) return;
  const long gal = base + c;
  const bool valid = gal < M;
  const float* xr[1] = {x + (valid ? gal : M - 1) * m.C};
  f32x16 ct0[1][1];
  sf_build_ctx_tile<1>(ct0, xr, m, 0, h);
  for (int t = 0; t < m.T; ++t) {
    const float* tp = m.packed + (size_t)t * m.t_stride;
    float* dst = tab + ((size_t)gal * m.T + t) * m.ctab_NV * m.ctab_R;
    {
      f32x16 hid[HT][1];
      sf_init_bias<HT, 1>(hid, tp + m.o_bin, h);
      sf_ctx_mm<HT, 1>(hid, xr, m, tp + m.o_winc, lane, &ct0);
      if (valid) sf_ctab_store<HT>(hid, dst, h);
    }
#pragma unroll
    for (int k = 0; k < SF_NBMAX; ++k)
      if (k < m.NB) {
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) {
          f32x16 g[1][1];
          sf_init_bias<1, 1>(g, tp + m.o_bg[k] + mt * 32, h);
          sf_ctx_mm<1, 1>(g, xr, m, tp + m.o_wg[k] + mt * m.nGc * 256, lane, &ct0);
          if (valid) sf_ctab_store<1>(g, dst + (1 + k) * m.ctab_R + mt * 32, h);
        }
      }
  }
}
