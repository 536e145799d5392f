// sf_queue.h -- device-side work queue of the persistent sampler kernels (k_maf_inv16, k_inverse).
//
// Replaces the host-driven rejection rounds (launch + D2H counter + stream sync per round) of [UPSTREAM] sbi
// accept_reject_sample (reached from ref: src/synference/sbi_runner.py:6442): ONE launch of persistent workgroups
// works until every output slot is resolved.  Work comes from two sources:
//   dense list  : item i -> slot (slots[i] or slot_base + i), first attempt `a.attempt`; handed out by one returning
//                 atomic add per workgroup iteration (no static assignment: a workgroup that is not resident yet owns
//                 nothing, so resident ones never wait for it)
//   own retries : a workgroup's rejected slots stay in its LDS control block and lead its next iteration
//   retry ring  : only once the dense list is exhausted: 8-byte entries {slot, attempt + 1} (0 = empty) donated by
//                 workgroups that hold more entries than they can speculate on, claimed by idle ones (compare-and-swap
//                 on `head`, then a wait for each claimed entry to become non-zero; the claimer clears it).
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
  unsigned long long stats[24];  // SF_Q_STATS builds: cycles / counts per phase, summed over workgroups (diagnostics)
};

#define SF_Q_SPIN_TICKS 1500000000ull  // 15 s of s_memrealtime (100 MHz)
#define SF_Q_SPIN_ITERS 4000000u        // and an iteration bound that does not depend on any clock (a few seconds)

// control block of a workgroup in LDS (uint32 words), IPW = items per workgroup iteration
//   [0] entries fetched (0 = exit)  [1] log2(attempts per entry)  [2] retry entries staged  [3] survivors staged
//   [4] resolved  [5] evaluations  [6] first-attempt rejections  [7],[8] the workgroup's current range of the dense
//   list  [9] dense list exhausted  [10] tickets held  [12..19] their ring positions  [20..23] per-wave candidates of
//   a speculation group that spans waves
//   then slot[IPW] att[IPW] push_slot[IPW] push_att[IPW] surv_slot[IPW]
#define SF_Q_HDR 24
#define SF_Q_WORDS(IPW) (SF_Q_HDR + 5 * (IPW))

__device__ __forceinline__ unsigned int sf_q_ld(const unsigned int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifdef __HIPCC__
#ifdef SF_Q_STATS
#define SF_QS_T() __builtin_amdgcn_s_memtime()
#define SF_QS_ADD(i, v) do { if (threadIdx.x == 0) atomicAdd(&a.q->stats[i], (unsigned long long)(v)); } while (0)
#else
#define SF_QS_T() 0ull
#define SF_QS_ADD(i, v) do { } while (0)
#endif
// Publishes what the previous iteration staged and assembles the next batch of work for the workgroup.
// Must be called by every thread; returns false when the workgroup is done.  AMAX: largest speculation width the
// calling kernel resolves inside one tile (16 or 32).  `pf` (one register, meaningful on thread 0) carries the
// returning atomic that reserved the workgroup's NEXT range of the dense list: it is issued one iteration ahead, so
// its latency passes behind the flow evaluation instead of in front of it.
//
// Work of an iteration = the workgroup's OWN rejected entries of the previous iteration (they never leave the LDS
// control block) topped up with fresh items of the dense list (ranges of IPW items, one returning atomic add on
// `dense_next` per range -- the only global atomic while the dense list lasts).  Only when the dense list is
// exhausted ("tail mode") does work move between workgroups, through the retry ring, as a TICKET queue:
//   * a workgroup left without entries flushes its tallies and draws KEEP = IPW / AMAX tickets (one fetch-and-add on
//     `head`: no compare-and-swap, so hundreds of idle workgroups never fight over one word); ticket p is ring
//     position p; the workgroup then polls ITS OWN positions (with back-off) until one is filled, or until `resolved`
//     shows that every slot is done.  Tickets that are still empty when work arrives stay with the workgroup and are
//     looked at again on its later iterations;
//   * `head - reserve` is the number of tickets waiting for an entry: a workgroup that holds more than KEEP entries
//     donates up to that many to the ring (one fetch-and-add on `reserve`, then the entry stores), and runs the rest
//     with as many speculative attempts per slot as fit its items (a power of two <= AMAX).
template <int IPW, int AMAX, class Args>
__device__ __forceinline__ bool sf_q_fetch(const Args& a, unsigned int* ctrl, unsigned int& pf) {
  const unsigned long long qs_t0 = SF_QS_T();
  __syncthreads();  // every wave has staged its results; the previous work words are no longer read
  const unsigned long long qs_t1 = SF_QS_T();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    SfQueue* q = a.q;
    unsigned int* w_slot = ctrl + SF_Q_HDR;
    unsigned int* w_att = w_slot + IPW;
    const unsigned int* p_slot = w_att + IPW;
    const unsigned int* p_att = p_slot + IPW;
    const unsigned int* s_slot = p_att + IPW;
    constexpr unsigned int KEEP = IPW / AMAX;  // entries a workgroup can run at full speculation width (<= 8)
    // ---- survivors of the previous iteration -> global list
    const unsigned int sc = ctrl[3];
    if (sc) {
      unsigned int pos = 0;
      if (lane == 0) pos = atomicAdd(&q->n_surv, sc);
      pos = __builtin_amdgcn_readfirstlane(pos);
      for (unsigned int i = lane; i < sc; i += 64) {
        if (pos + i < a.n_total) a.rejected[pos + i] = s_slot[i];
        else atomicExch(&q->error, 3u);  // more survivors than slots: a slot was resolved twice
      }
    }
    // ---- own retries of the previous iteration are the first entries of this one
    unsigned int n = ctrl[2];
    for (unsigned int i = lane; i < n; i += 64) {
      w_slot[i] = p_slot[i];
      w_att[i] = p_att[i];
    }
    // ---- top up from the workgroup's range of the dense list; ranges are reserved one iteration ahead.  A top-up
    // may end one range and begin the next (two segments at most: a range holds IPW items)
    unsigned int dcur = ctrl[7], dend = ctrl[8], dense_done = ctrl[9];
    if (!dense_done && n >= (unsigned)IPW && dcur >= dend) {
      // own retries fill the whole iteration: still find out when the dense list has run dry (tail mode shares work)
      if (__shfl(pf, 0, 64) >= a.n_total) dense_done = 1u;
    }
#pragma unroll 1
    for (int seg = 0; seg < 2; ++seg) {
      if (dense_done || n >= (unsigned)IPW) break;  // (a full iteration never touches `pf`: it may have just been issued)
      if (dcur >= dend) {  // range used up: adopt the reserved one (its atomic was issued an iteration ago)
        const unsigned int nb = __shfl(pf, 0, 64);
        if (nb >= a.n_total) {
          dense_done = 1u;
          break;
        }
        dcur = nb;
        dend = a.n_total - nb < (unsigned)IPW ? a.n_total : nb + (unsigned)IPW;
        if (lane == 0) pf = atomicAdd(&q->dense_next, (unsigned)IPW);  // for a later iteration: not waited for here
      }
      const unsigned int want = (unsigned)IPW - n;
      const unsigned int nd = dend - dcur < want ? dend - dcur : want;
      for (unsigned int i = lane; i < nd; i += 64) {
        unsigned int sl;
        if (a.slots) {
          unsigned int pos = dcur + i;
          if (a.list_mul) {  // strided walk through a sorted list (sf_internal.h)
            const unsigned int msk = (1u << a.list_log2) - 1u;
            do { pos = (pos * a.list_mul) & msk; } while (pos >= a.n_total);
          }
          sl = a.slots[pos];
        } else if (a.dense_G) {  // block-interleaved order of a whole catalogue (sf_internal.h)
          const unsigned int idx = dcur + i, S_ = (unsigned int)a.S, M_ = a.n_total / S_;
          const unsigned int per_block = a.dense_G * S_, b = idx / per_block, j = idx - b * per_block;
          const unsigned int left = M_ - b * a.dense_G, Gb = left < a.dense_G ? left : a.dense_G;
          const unsigned int run = a.dense_run ? a.dense_run : 1u;
          const unsigned int q = j / run, ln = j - q * run;    // run number inside the block, position inside the run
          const unsigned int rr = q / Gb, g = q - rr * Gb;      // which run of the galaxy, which galaxy of the block
          sl = (b * a.dense_G + g) * S_ + rr * run + ln;
        } else {
          sl = (unsigned int)a.slot_base + dcur + i;
        }
        if (sl >= a.out_slots) { atomicExch(&q->error, 4u); sl = 0u; }  // a listed slot outside out[M*S]
        w_slot[n + i] = sl;
        w_att[n + i] = a.attempt;
      }
      n += nd;
      dcur += nd;
    }
    unsigned int lg = 0;
    if (dense_done) {  // ---- tail mode
      if (lane == 0) {
        if (ctrl[4]) atomicAdd(&q->resolved, ctrl[4]);
        if (ctrl[5]) atomicAdd(&q->evals, (unsigned long long)ctrl[5]);
        if (ctrl[6]) atomicAdd(&q->rej0, ctrl[6]);
        ctrl[4] = 0; ctrl[5] = 0; ctrl[6] = 0;
      }
      // tickets this workgroup holds (ring positions still waiting for an entry): lane i < nt looks after ticket i
      unsigned int nt = ctrl[10];
      unsigned int tpos = (unsigned)lane < nt ? ctrl[12 + lane] : 0u;
      const unsigned long long qs_i0 = SF_QS_T();
      bool waited = false;
      if (n == 0 && nt == 0) {  // nothing to do: draw tickets
        unsigned int t = 0;
        if (lane == 0) t = atomicAdd(&q->head, KEEP);
        t = __builtin_amdgcn_readfirstlane(t);
        nt = KEEP;
        tpos = t + (unsigned)lane;
      }
      unsigned long long e = 0ull;
      if (n < (unsigned)IPW && nt > 0) {
        unsigned int spins = 0, nap = 1;
        for (;;) {
          if ((unsigned)lane < nt) e = __hip_atomic_load(a.ring + (tpos & a.ring_mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__ballot(e != 0ull) != 0ull || n > 0) break;  // an entry arrived (or there is own work: no waiting)
          waited = true;
          unsigned int stop = 0;
          if (lane == 0 && (spins & 3u) == 0u) stop = (sf_q_ld(&q->resolved) >= a.n_total || sf_q_ld(&q->error)) ? 1u : 0u;
          if (__builtin_amdgcn_readfirstlane(stop)) break;  // every slot is resolved: leave (n stays 0)
          for (unsigned int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(127);  // ~3.4 us each
          if (nap < 6u) ++nap;
          if (++spins > SF_Q_SPIN_ITERS / 4u) {
            if (lane == 0) atomicExch(&q->error, 1u);
            break;
          }
        }
        // take the filled tickets, keep the empty ones
        const bool got = (unsigned)lane < nt && e != 0ull;
        if (got) {
          __hip_atomic_store(a.ring + (tpos & a.ring_mask), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((unsigned int)e >= a.out_slots || (unsigned int)(e >> 32) == 0u) {  // not an entry any donor wrote
            atomicExch(&q->error, 2u);
            e = (unsigned long long)a.attempt_limit << 32 | 1ull << 32;  // evaluates nothing, resolves nothing
          }
        }
        const unsigned long long gm = __ballot(got), km = __ballot((unsigned)lane < nt && !got);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (got) {
          const unsigned int r = n + (unsigned)__popcll(gm & below);
          if (r < (unsigned)IPW) {
            w_slot[r] = (unsigned int)e;
            w_att[r] = (unsigned int)(e >> 32) - 1u;
          }
        }
        if ((unsigned)lane < nt && !got) ctrl[12 + __popcll(km & below)] = tpos;
        n += (unsigned)__popcll(gm);
        nt = (unsigned)__popcll(km);
        if (n > (unsigned)IPW) n = IPW;  // (cannot happen: n + tickets <= IPW when tickets are looked at)
      }
      if (waited) {
        SF_QS_ADD(3, SF_QS_T() - qs_i0);  // cycles waiting as an idle workgroup
        SF_QS_ADD(4, 1);                  // idle episodes
      }
      if (n > KEEP) {
        // share the surplus out to the tickets that wait for an entry: entries KEEP .. n-1 go to the ring
        unsigned int give = 0, base = 0;
        if (lane == 0) {
          const unsigned int res = sf_q_ld(&q->reserve), head = sf_q_ld(&q->head);
          const unsigned int waiting = head - res;
          if (waiting > 0 && waiting < 0x80000000u) {
            give = n - KEEP < waiting ? n - KEEP : waiting;
            base = atomicAdd(&q->reserve, give);
          }
        }
        give = __builtin_amdgcn_readfirstlane(give);
        base = __builtin_amdgcn_readfirstlane(base);
        if (give) {
          for (unsigned int i = lane; i < give; i += 64) {
            const unsigned int j = n - give + i;
            const unsigned long long en = (unsigned long long)w_slot[j] | ((unsigned long long)(w_att[j] + 1u) << 32);
            __hip_atomic_store(a.ring + ((base + i) & a.ring_mask), en, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          n -= give;
          SF_QS_ADD(5, 1);                // donations
        }
      }
      if (n) {
        // Speculation fills lanes that would idle, but the waves it keeps busy share their SIMDs with three other
        // workgroups: entries that have only just been rejected (a first retry is accepted 9 times out of 10 on a
        // trained posterior) get A = 2, and A grows with the attempt number the entries have reached.
        unsigned int mn = 0xffffffffu;
        for (unsigned int i = lane; i < n; i += 64) mn = w_att[i] < mn ? w_att[i] : mn;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const unsigned int ot = (unsigned int)__shfl_xor((int)mn, o, 64);
          mn = ot < mn ? ot : mn;
        }
        unsigned int amax_now = mn < 4u ? 2u : (mn < (unsigned)AMAX ? mn : (unsigned)AMAX);
        if (a.spec_full_after && mn >= a.spec_full_after) amax_now = (unsigned)AMAX;
        const unsigned int cap = (a.tail_cap && a.tail_cap < (unsigned)IPW) ? a.tail_cap : (unsigned)IPW;
        while ((2u << lg) * n <= cap && (2u << lg) <= amax_now) ++lg;
      }
      if (lane == 0) ctrl[10] = nt;
    }
    if (lane == 0) {
      ctrl[0] = n; ctrl[1] = lg; ctrl[2] = 0; ctrl[3] = 0;
      ctrl[7] = dcur; ctrl[8] = dend; ctrl[9] = dense_done;
    }
    SF_QS_ADD(dense_done ? 7 : 6, 1);     // iterations in tail / dense mode
    SF_QS_ADD(8, n);                      // entries handed out
  }
  const unsigned long long qs_t2 = SF_QS_T();
  __syncthreads();
  SF_QS_ADD(0, qs_t1 - qs_t0);            // wait at the entry barrier (thread 0: skew between the waves)
  SF_QS_ADD(1, qs_t2 - qs_t1);            // wave 0's serial section
  SF_QS_ADD(2, SF_QS_T() - qs_t2);        // exit barrier
  return ctrl[0] != 0u;
}

// first reservation of a workgroup (before its first sf_q_fetch): thread 0 issues it, nobody waits for it yet
template <int IPW, class Args>
__device__ __forceinline__ void sf_q_begin(const Args& a, unsigned int* ctrl, unsigned int& pf) {
  if (threadIdx.x < SF_Q_HDR) ctrl[threadIdx.x] = 0u;
  pf = 0u;
  if (threadIdx.x == 0) pf = atomicAdd(&a.q->dense_next, (unsigned)IPW);
}
#endif
