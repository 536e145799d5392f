// sf_queue.h -- device-side work queue of the persistent sampler kernels (k_maf_inv16, k_inverse).
//
// Replaces the host-driven rejection rounds (launch + D2H counter + stream sync per round) of [UPSTREAM] sbi
// accept_reject_sample (reached from ref: src/synference/sbi_runner.py:6442): ONE launch of persistent workgroups
// works until every output slot is resolved.  Work comes from two sources:
//   dense list  : item i -> slot (slots[i] or slot_base + i), first attempt `a.attempt`; handed out in chunks of IPW
//                 items by one returning atomic add per workgroup iteration (no static assignment: a workgroup that
//                 is not resident yet owns nothing, so resident ones never wait for it)
//   retry ring  : 8-byte entries {slot, attempt + 1} (0 = empty) pushed for every rejected slot; multi-producer /
//                 multi-consumer: producers reserve positions with one atomic add per workgroup iteration and then
//                 store the entries (agent-scope, write-through); consumers claim a range by compare-and-swap on
//                 `head` and wait for each claimed entry to become non-zero, then clear it.
// A slot keeps the LOWEST accepted attempt of its Philox stream (slot, attempt), so the draws do not depend on the
// schedule (which workgroup retried what, with how much speculation) -- the property the parity tests rely on.
// Termination: `resolved` counts slots that were accepted or handed to the survivor list (attempt_limit reached);
// a workgroup leaves when it finds no work and resolved == n_total.  Every spin is bounded (SF_Q_SPIN_TICKS of the
// 100 MHz wall clock): on expiry the `error` word is set, every workgroup drains, and the host reports SF_ERR_STATE.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct SfQueue {  // device memory owned by the handle; zeroed before each persistent launch; one 128-B line per hot word
  unsigned int dense_next; unsigned int _p0[31];
  unsigned int head;       unsigned int _p1[31];
  unsigned int reserve;    unsigned int _p2[31];
  unsigned int resolved;   unsigned int _p3[31];
  unsigned int n_surv;     unsigned int dropped;  // (dropped: survivors turned into NaN rows by the progress rule)
  unsigned int _p4[30];
  unsigned int error;      unsigned int _p5[31];
  unsigned int rej0;       unsigned int _p6[31];  // slots whose FIRST attempt was rejected (acceptance statistic)
  unsigned long long evals; unsigned int _p7[30]; // flow evaluations (items with a valid attempt)
};

#define SF_Q_SPIN_TICKS 1500000000ull  // 15 s of s_memrealtime (100 MHz)

// control block of a workgroup in LDS (uint32 words), IPW = items per workgroup iteration
//   [0] entries fetched (0 = exit)  [1] log2(attempts per entry)  [2] retry entries staged  [3] survivors staged
//   [4] resolved  [5] evaluations  [6] first-attempt rejections  [7] unused
//   then slot[IPW] att[IPW] push_slot[IPW] push_att[IPW] surv_slot[IPW]
#define SF_Q_HDR 8
#define SF_Q_WORDS(IPW) (SF_Q_HDR + 5 * (IPW))

__device__ __forceinline__ unsigned int sf_q_ld(const unsigned int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifdef __HIPCC__
// Flushes the previous iteration's staged pushes / tallies and fetches the next batch of work for the workgroup.
// Must be called by every thread; returns false when the workgroup is done.  AMAX: largest speculation width the
// calling kernel resolves inside one tile (16 or 32).
template <int IPW, int AMAX, class Args>
__device__ __forceinline__ bool sf_q_fetch(const Args& a, unsigned int* ctrl) {
  __syncthreads();  // every wave has staged its results; the previous work words are no longer read
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    SfQueue* q = a.q;
    unsigned int* w_slot = ctrl + SF_Q_HDR;
    unsigned int* w_att = w_slot + IPW;
    const unsigned int* p_slot = w_att + IPW;
    const unsigned int* p_att = p_slot + IPW;
    const unsigned int* s_slot = p_att + IPW;
    // ---- flush: staged retry entries -> ring, survivors -> list, tallies -> counters
    const unsigned int pc = ctrl[2], sc = ctrl[3];
    if (pc) {
      unsigned int base = 0;
      if (lane == 0) base = atomicAdd(&q->reserve, pc);
      base = __builtin_amdgcn_readfirstlane(base);
      for (unsigned int i = lane; i < pc; i += 64) {
        const unsigned long long e = (unsigned long long)p_slot[i] | ((unsigned long long)(p_att[i] + 1u) << 32);
        __hip_atomic_store(a.ring + ((base + i) & a.ring_mask), e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (sc) {
      unsigned int pos = 0;
      if (lane == 0) pos = atomicAdd(&q->n_surv, sc);
      pos = __builtin_amdgcn_readfirstlane(pos);
      for (unsigned int i = lane; i < sc; i += 64) a.rejected[pos + i] = s_slot[i];
    }
    unsigned int n = 0, src = 0, base = 0;
    if (lane == 0) {
      if (ctrl[4]) atomicAdd(&q->resolved, ctrl[4]);
      if (ctrl[5]) atomicAdd(&q->evals, (unsigned long long)ctrl[5]);
      if (ctrl[6]) atomicAdd(&q->rej0, ctrl[6]);
      ctrl[2] = 0; ctrl[3] = 0; ctrl[4] = 0; ctrl[5] = 0; ctrl[6] = 0;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        const unsigned int head = sf_q_ld(&q->head), res = sf_q_ld(&q->reserve);
        const unsigned int avail = res - head;
        const bool dense_left = sf_q_ld(&q->dense_next) < a.n_total;
        if (avail >= (unsigned)IPW || (avail > 0 && !dense_left)) {
          // a full batch of retries goes first (the ring drains while the dense list is still being worked);
          // once the dense list is exhausted the remaining entries are shared out thinly, so that every claim
          // can speculate on several attempts per slot and the tail takes few sequential passes
          unsigned int take = IPW;
          if (avail < (unsigned)IPW) {
            const unsigned int share = (avail + gridDim.x - 1) / gridDim.x;
            const unsigned int lo = IPW / AMAX;
            take = share > lo ? share : lo;
            take = take < avail ? take : avail;
          }
          unsigned int expect = head;
          if (__hip_atomic_compare_exchange_strong(&q->head, &expect, head + take, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT)) {
            n = take; src = 2; base = head;
            break;
          }
          continue;  // another workgroup claimed first: look again
        }
        if (dense_left) {
          const unsigned int d = atomicAdd(&q->dense_next, (unsigned)IPW);
          if (d < a.n_total) {
            n = a.n_total - d < (unsigned)IPW ? a.n_total - d : (unsigned)IPW; src = 1; base = d;
            break;
          }
          continue;
        }
        if (sf_q_ld(&q->resolved) >= a.n_total || sf_q_ld(&q->error)) break;  // done (n = 0)
        __builtin_amdgcn_s_sleep(32);
        if (__builtin_amdgcn_s_memrealtime() - t0 > SF_Q_SPIN_TICKS) {
          atomicExch(&q->error, 1u);
          break;
        }
      }
    }
    n = __builtin_amdgcn_readfirstlane(n);
    src = __builtin_amdgcn_readfirstlane(src);
    base = __builtin_amdgcn_readfirstlane(base);
    unsigned int lg = 0;
    if (src == 1) {
      for (unsigned int i = lane; i < n; i += 64) {
        w_slot[i] = a.slots ? a.slots[base + i] : (unsigned int)a.slot_base + base + i;
        w_att[i] = a.attempt;
      }
    } else if (src == 2) {
      for (unsigned int i = lane; i < n; i += 64) {
        unsigned long long* rp = a.ring + ((base + i) & a.ring_mask);
        unsigned long long e = 0;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {  // the producer reserved this position before we could claim it: its store is on the way
          e = __hip_atomic_load(rp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (e != 0ull) break;
          __builtin_amdgcn_s_sleep(2);
          if (__builtin_amdgcn_s_memrealtime() - t0 > SF_Q_SPIN_TICKS) {
            atomicExch(&q->error, 1u);
            break;
          }
        }
        __hip_atomic_store(rp, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w_slot[i] = (unsigned int)e;
        w_att[i] = e ? (unsigned int)(e >> 32) - 1u : a.attempt_limit;  // (lost entry: evaluates nothing)
      }
      while ((2u << lg) * n <= (unsigned)IPW && (2u << lg) <= (unsigned)AMAX) ++lg;
    }
    if (lane == 0) { ctrl[0] = n; ctrl[1] = lg; }
  }
  __syncthreads();
  return ctrl[0] != 0u;
}
#endif
