// edge_queue_kernel.hpp -- `fk_edge_queue`: the edge bisection (VoxelEnvironment::voxelize_valid_backbone_motion,
// motion-planning/VoxelEnvironment.cpp:207-444, behind AbstractVoxelMotionValidator::checkMotion,
// AbstractVoxelMotionValidator.h:143-151) as ONE persistent launch over a device work queue, with a barrier per EDGE instead of
// one per level of all edges (edge_kernel.hpp: "the edge queue" states the scheme and why the results are those of the
// level-synchronous schedule).
//
// A wave's round: take up to 64 pool slots at the queue's head -> integrate their states with the verdict-only body and its
// per-point sweep (verdict_kernel.hpp: fk_uniform_body + PointSweep<.., SIG>, the signature rows go to the slots' rows) -> fold
// the verdicts into the edges -> for every edge whose level this wave completed: should_subdivide on both halves of the level's
// intervals (signatures_differ, the whole wave on one pair of rows), the survivors become the edge's next level at the queue's
// tail -> publish -> count the round's samples as done.
//
// Hand-offs between waves (MI355X: per-XCD L2s, a CU's vector L1 is never refreshed by another CU's stores): every record a wave
// leaves for others -- signature rows, edge_ok, the level records, the pushed slots' intervals and states -- is written with plain
// stores, then `s_waitcnt vmcnt(0)` -> agent-scope release fence -> `s_waitcnt vmcnt(0)` -> the agent-scope atomic that
// hands it over (the edge's `remaining` counter, the slot's ready flag); the wave that receives it -- its decrement returned 1, its
// poll of the ready flag matched -- runs an agent-scope acquire fence and `s_waitcnt vmcnt(0)` before its first load.
// No wave ever waits for a wave that has not started: a wave holds a ticket for slots, waits only while slots it holds are being
// written by a wave inside its push, and leaves when done == tail.  Every wait is bounded (EQF_STUCK ends the launch).
#pragma once
#include "verdict_kernel.hpp"
#define TRK_EDGE_DEVICE_ONLY
#include "edge_kernel.hpp"

namespace trk {

__device__ __forceinline__ uint32_t eq_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void eq_wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void eq_release() {
  eq_wait_vm();
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  eq_wait_vm();                         // (inline asm: the compiler may not drop the wait behind the write-back)
}
__device__ __forceinline__ void eq_acquire() {
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  eq_wait_vm();
}
__device__ __forceinline__ int eq_bcast(int v) { return __builtin_amdgcn_readfirstlane(v); }
// phase clock of a wave: the time since the last call goes to the 64-bit counter at ctl[word] (lane 0, one atomic per phase and round)
struct EqClock {
  unsigned long long t;
  __device__ __forceinline__ void start() { t = wall_clock64(); }
  __device__ __forceinline__ void lap(uint32_t *ctl, int word) {
    const unsigned long long n = wall_clock64();
    if (threadIdx.x == 0) __hip_atomic_fetch_add((unsigned long long *)(ctl + word), n - t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = n;
  }
};

// LDS scratch of the finishing stage: the head of the wave's image (state of the per-point sweep, dead between two
// integrations): the survivors' mask (one bit per candidate of a group), then per owner lane: its first candidate, edge, level base,
// survivor count, domain flag, the new level's first slot and fill count, 1 / validSegmentCount
constexpr int EQ_GROUP_CAND = 2 * EQ_MAX_CAND;                 // candidates of one group of finished levels
#define EQ_MASK(i) (((unsigned long long *)vlds)[(i)])
#define EQ_I(i) (((int32_t *)vlds)[(i)])
#define EQ_U(i) (((uint32_t *)vlds)[(i)])
constexpr int EQ_PRE = EQ_GROUP_CAND / 32, EQ_EOF = EQ_PRE + 64, EQ_BASEOF = EQ_EOF + 64, EQ_CNT = EQ_BASEOF + 64, EQ_DOM = EQ_CNT + 64,
              EQ_NBASE = EQ_DOM + 64, EQ_FILL = EQ_NBASE + 64, EQ_RELW = EQ_FILL + 64;     // (word offsets)
#define EQ_REL(i) (((double *)vlds)[EQ_RELW / 2 + (i)])

// signatures_differ (edge_kernel.hpp) by ONE lane: the lane's own pair of rows from the tip down, four points per 16-byte load,
// 24 loads in flight; the first event in tip-first order decides, a domain error at a point before a difference at that point
__device__ __forceinline__ int signatures_differ_lane(const uint32_t *__restrict__ ra, const uint32_t *__restrict__ rb, int P) {
  constexpr int U = 12;                 // 24 loads in flight per lane: a 129-point row in three rounds of the memory latency (4: nine)
  for (int k = (P - 1) >> 2; k >= 0; k -= U) {
    uint4 va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int kk = k - u >= 0 ? k - u : 0;
      va[u] = *reinterpret_cast<const uint4 *>(ra + 4 * kk);
      vb[u] = *reinterpret_cast<const uint4 *>(rb + 4 * kk);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (k - u < 0) break;
      const uint32_t wa[4] = {va[u].x, va[u].y, va[u].z, va[u].w}, wb[4] = {vb[u].x, vb[u].y, vb[u].z, vb[u].w};
#pragma unroll
      for (int w = 3; w >= 0; w--) {
        if (4 * (k - u) + w >= P) continue;                    // (words of the row's padding)
        const uint32_t a = wa[w], b = wb[w];
        if ((a | b) & SIG_BAD) return 2;
        const int dx = (int)(a & 1023u) - (int)(b & 1023u), dy = (int)((a >> 10) & 1023u) - (int)((b >> 10) & 1023u),
                  dz = (int)((a >> 20) & 1023u) - (int)((b >> 20) & 1023u);
        if (dx > 1 || dx < -1 || dy > 1 || dy < -1 || dz > 1 || dz < -1) return 1;
      }
    }
  }
  return 0;
}

// Take the wave's next samples: slots [h, h + cnt) of the pool.  own_lo / own_hi: the ticket the wave holds.  false: leave.
__device__ __forceinline__ bool eq_claim(const EdgeQueueArgs &q, int &own_lo, int &own_hi, int &h, int &cnt) {
  const int lane = threadIdx.x;
  uint32_t *ctl = q.ctl;
  uint32_t seen = 0xffffffffu;          // done + tail at the last look: a wait that sees them unchanged for too long gives up
  unsigned idle = 0;
  for (;;) {
    uint32_t fl = 0, d = 0, t = 0;
    if (lane == 0) {
      fl = eq_load(ctl + EQ_FLAGS);
      d = eq_load(ctl + EQ_DONE);
      eq_wait_vm();                     // `done` is read BEFORE `tail`: done == tail then means that nothing was in flight at that moment
      t = eq_load(ctl + EQ_TAIL);
    }
    fl = (uint32_t)eq_bcast((int)fl); d = (uint32_t)eq_bcast((int)d); t = (uint32_t)eq_bcast((int)t);
    if (fl) return false;
    if (own_lo == own_hi) {
      // a ticket sized by the queue's depth: whole waves while it is deep, single samples when it runs dry (an FK costs a wave
      // the same 128 serial steps whatever its lane count: the last samples finish soonest one per idle wave)
      int lo = 0, size = 0;
      if (lane == 0) {
        const int depth = (int)(t - eq_load(ctl + EQ_HEAD));
        size = depth >= 4096 ? 64 : (depth < 128 ? 1 : depth >> 6);
        lo = (int)__hip_atomic_fetch_add(ctl + EQ_HEAD, (uint32_t)size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      own_lo = eq_bcast(lo); own_hi = own_lo + eq_bcast(size);
    }
    if ((int)t > own_lo) {
      h = own_lo;
      cnt = ((int)t < own_hi ? (int)t : own_hi) - own_lo;
      own_lo += cnt;
      return true;
    }
    if (d == t) return false;
    if (d + t != seen) { seen = d + t; idle = 0; }
    else if (++idle > (1u << 19)) {     // ~ seconds without a sample finishing anywhere
      if (lane == 0) __hip_atomic_fetch_or(ctl + EQ_FLAGS, EQF_STUCK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(127);
    if (idle > 16) __builtin_amdgcn_s_sleep(127);
    if (idle > 256) { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); }
  }
}

template <int N, bool ROT>
__global__ __launch_bounds__(64, (N <= TRK_VERDICT_TWO_WAVE_MAXN ? 2 : 1)) void fk_edge_queue(
    RobotK K, const double *__restrict__ tab, const StepK *__restrict__ steps, int nsteps,
    const VerdictArgs *__restrict__ va, const EdgeQueueArgs *__restrict__ qa, const FusedSweepArgs *__restrict__ sa) {
  const int lane = threadIdx.x;
  int own_lo = 0, own_hi = 0;
  EqClock clk;
  clk.start();
  for (;;) {
    int zero = 0;
    asm volatile("" : "+s"(zero));      // (as PointSweep::args: the queue's arguments are re-read when needed, not held across the RK4 loop)
    const EdgeQueueArgs &q = qa[zero];
    int h = 0, cnt = 0;
    if (!eq_claim(q, own_lo, own_hi, h, cnt)) break;
    clk.lap(q.ctl, EQ_T_CLAIM);
    const bool live = lane < cnt;
    // ---- the slots' records: written by a wave that is inside its push right now, or long ago ----
    int e_lane = -1;
    {
      bool bail = false;
      if (live) {
        const uint32_t *flag = (const uint32_t *)q.sample_edge + h + lane;
        unsigned spins = 0;
        while ((e_lane = (int)eq_load(flag)) < 0) {
          if (eq_load(q.ctl + EQ_FLAGS) != 0u) { bail = true; break; }
          if (++spins > (1u << 22)) {
            __hip_atomic_fetch_or(q.ctl + EQ_FLAGS, EQF_STUCK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bail = true; break;
          }
          __builtin_amdgcn_s_sleep(8);
        }
      }
      if (__any(bail)) break;
    }
    eq_acquire();
    clk.lap(q.ctl, EQ_T_READY);

    // ---- integrate: fk_verdict's body on the claimed slots (lane l: slot h + l) ----
    PointSweep<false, true> ps;
    ps.va = va;
    ps.dn_prev = 0.0f; ps.sph_state = 0u;
    ps.qhead = 0; ps.qcount = 0; ps.active = false;
    ps.P = va->P; ps.CH = va->CH; ps.NM = va->NM; ps.Kl = (ps.P - 1 + ps.CH - 1) / ps.CH; ps.ms_next = 0; ps.ms_k = 0;
    ps.sigst.init(); ps.sig_row_of = -1; ps.sig_first_row = 0; ps.sig_row0 = h; ps.sig_cnt = cnt;
    __syncthreads();                    // (the previous round's LDS scratch is dead)
    VL_U(VL_HIT + lane) = 0u; VL_U(VL_INPREV + lane) = 0u; VL_F(VL_DIST + lane) = 0.0f;
    __syncthreads();
    // fk_uniform_body takes lane l of block b as configuration 64 b + l of `states`, live below n: shift both so that it is slot h + l
    const int S = K.state_size;
    const double *st_eff = q.states + ((int64_t)h - (int64_t)blockIdx.x * 64) * S;
    const int64_t n_eff = (int64_t)blockIdx.x * 64 + cnt;
    FkLane<N> fl_;
    {
      FkOut out{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
      fk_uniform_body<N, ROT, false, false>(st_eff, n_eff, 0, K, tab, steps, nsteps, out, ps, nullptr, &fl_);
    }
    ps.template sig_finish<false>();
    ps.finish();
    while (ps.qcount > 0) ps.flush();
    __syncthreads();
    LaneVerdict lv;
    {
      const VerdictArgs a = *va;
      lv = verdict_decide<N, false>(a, fl_, live);
    }
    int zero2 = 0;
    asm volatile("" : "+s"(zero2));
    const EdgeQueueArgs &q2 = qa[zero2];
    clk.lap(q2.ctl, EQ_T_FK);
    if (__any(lv.pending)) {
      // the exact pairwise self-collision sweep needs every backbone point at once: this wave integrates its round again,
      // storing the points in its own columns of the workspace, and takes sweep_body's verdict for the pending lanes
      // (what fk_sweep_fused_list does for the level-synchronous launches; rare: tight curls only)
      __syncthreads();
      fk_uniform_body<N, ROT, false, false>(st_eff, n_eff, q2.fb_ld, K, tab, steps, nsteps, q2.fb_out, NoPointHook(), nullptr, nullptr);
      __syncthreads();
      const FusedSweepArgs fa = *sa;
      bool exact = false;
      sweep_body<false>(q2.fb_in, n_eff, q2.fb_ld, fa.P, fa.CH, fa.NM, K, fa.g, fa.grid, fa.near_grid, 1, fa.debug, nullptr, nullptr, nullptr, &exact);
      __syncthreads();
      if (lane == 0) __hip_atomic_fetch_add(q2.ctl + EQ_PENDING, (uint32_t)__popcll(__ballot(lv.pending)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (lv.pending) lv.valid = exact;
      clk.lap(q2.ctl, EQ_T_EXACT);
    }

    // ---- fold ----
    if (live && !lv.valid) q2.edge_ok[e_lane] = 0u;
    eq_release();                       // the round's signature rows and edge_ok stores, before the counters say so
    int old = 0;
    if (live) old = __hip_atomic_fetch_add(q2.remaining + e_lane, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool fin = live && old == 1;        // this lane folded the last outstanding sample of its edge's level
    bool stop = false;
    EqClock sub = clk;
    sub.lap(q2.ctl, EQ_T_F0);
    if (__any(fin)) {
      eq_acquire();                     // the other samples of these levels: rows, edge_ok, the level records
      {
        const uint32_t nfin = (uint32_t)__popcll(__ballot(fin));
        if (lane == 0) __hip_atomic_fetch_add(q2.ctl + EQ_FINISHED, nfin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // ---- finish the levels: should_subdivide on both halves of every interval, ONE CANDIDATE PER LANE ----
      // candidates 2 i, 2 i + 1 of an edge: the distal (:387-390) and the proximal (:393-396) half of its interval i, whose midpoint
      // is pool slot base + i.  A finisher lane owns its edge's 2 cnt candidates; the wave deals all of them out 64 at a time.
      int base_l = 0, ncand_l = 0;
      double rel_l = 0.0;
      if (fin && q2.edge_ok[e_lane] != 0u) {                  // (an invalid sample decides the edge: nothing to open)
        base_l = q2.lvl_base[e_lane]; ncand_l = 2 * q2.lvl_cnt[e_lane]; rel_l = q2.rel[e_lane];
        if (ncand_l > EQ_MAX_CAND) {
          __hip_atomic_fetch_or(q2.ctl + EQ_FLAGS, EQF_DEEP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ncand_l = 0; stop = true;
        }
      }
      stop = __any(stop);
      sub.lap(q2.ctl, EQ_T_F1);
      unsigned long long todo = __ballot(ncand_l > 0);
      const int P = q2.P;
      const int S2 = q2.sk.S;
      while (todo && !stop) {
        // a group of owner lanes, in lane order, whose candidates fit the emit mask (EQ_GROUP_CAND bits; one level is at most half of it)
        const bool mine = (todo >> lane) & 1ull;
        int pre = mine ? ncand_l : 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(pre, o, 64); if (lane >= o) pre += v; }     // inclusive prefix
        const bool in_group = mine && pre <= EQ_GROUP_CAND;
        const unsigned long long gm = __ballot(in_group);
        todo &= ~gm;
        const int n_g = in_group ? ncand_l : 0;
        int ex = n_g;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(ex, o, 64); if (lane >= o) ex += v; }
        const int T = __shfl(ex, 63, 64);                      // the group's candidates
        ex -= n_g;                                             // exclusive prefix: the lane's first candidate
        __syncthreads();
        EQ_I(EQ_PRE + lane) = ex; EQ_I(EQ_EOF + lane) = e_lane; EQ_I(EQ_BASEOF + lane) = base_l;
        EQ_I(EQ_CNT + lane) = 0; EQ_I(EQ_DOM + lane) = 0; EQ_I(EQ_FILL + lane) = 0;
        EQ_REL(lane) = rel_l;
        __syncthreads();
        if (lane == 0) __hip_atomic_fetch_add(q2.ctl + EQ_CAND, (uint32_t)T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // candidate g of the group -> its owner lane (the last lane whose first candidate is <= g), the edge's candidate index
        auto owner_of = [&](int g) {
          int lo = 0;
#pragma unroll
          for (int st = 32; st > 0; st >>= 1) if (EQ_I(EQ_PRE + lo + st) <= g) lo += st;
          return lo;
        };
        auto candidate = [&](int ow, int c, int &sa, int &sb, double &ta, double &tb) {
#pragma clang fp contract(off)
          const int sm = EQ_I(EQ_BASEOF + ow) + (c >> 1);
          const EdgeIv iv = q2.iv[sm];
          const double tm = (iv.ta + iv.tb) / 2;               // :382
          if (c & 1) { sa = iv.sa; sb = sm; ta = iv.ta; tb = tm; } else { sa = sm; sb = iv.sb; ta = tm; tb = iv.tb; }
        };
        // pass 1: the verdict of every candidate, survivors counted per owner
        for (int g0 = 0; g0 < T; g0 += 64) {
          const int g = g0 + lane;
          bool emit = false;
          if (g < T) {
#pragma clang fp contract(off)
            const int ow = owner_of(g);
            int sa, sb; double ta, tb;
            candidate(ow, g - EQ_I(EQ_PRE + ow), sa, sb, ta, tb);
            const int f = signatures_differ_lane(q2.sig + (int64_t)sa * q2.sig_stride, q2.sig + (int64_t)sb * q2.sig_stride, P);
            if (f == 2) atomicOr(&EQ_U(EQ_DOM + ow), 1u);      // std::domain_error in the reference: the edge is invalid
            emit = f == 1 && (tb - ta) > EQ_REL(ow);           // width rule of :369-372, applied at push time
            if (emit) atomicAdd(&EQ_U(EQ_CNT + ow), 1u);
          }
          const unsigned long long m = __ballot(emit);
          if (lane == 0) EQ_MASK(g0 >> 6) = m;
        }
        __syncthreads();
        sub.lap(q2.ctl, EQ_T_F2);
        // the owners' next levels: one allocation at the queue's tail for the whole group
        int total_l = 0;
        if (in_group) {
          if (EQ_I(EQ_DOM + lane)) {
            q2.edge_ok[e_lane] = 0u;
            __hip_atomic_fetch_add(q2.ctl + EQ_DOMAIN, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else total_l = EQ_I(EQ_CNT + lane);
        }
        int nex = total_l;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(nex, o, 64); if (lane >= o) nex += v; }
        const int Tnew = __shfl(nex, 63, 64);
        nex -= total_l;
        if (Tnew == 0) continue;                               // nothing left to subdivide (or domain errors): these edges are decided
        int nb = 0;
        if (lane == 0) nb = (int)__hip_atomic_fetch_add(q2.ctl + EQ_TAIL, (uint32_t)Tnew, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nb = eq_bcast(nb);
        if (nb + Tnew > q2.slot_hi) {
          if (lane == 0) __hip_atomic_fetch_or(q2.ctl + EQ_FLAGS, EQF_OVERFLOW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          stop = true; break;
        }
        EQ_I(EQ_NBASE + lane) = nb + nex;
        if (total_l > 0) {
          q2.lvl_base[e_lane] = nb + nex; q2.lvl_cnt[e_lane] = total_l; q2.remaining[e_lane] = total_l;
          q2.nfk[e_lane] += total_l;                           // every sample of an opened level is evaluated
        }
        __syncthreads();
        // pass 2: the survivors' records at their slots
        for (int g0 = 0; g0 < T; g0 += 64) {
          const unsigned long long m = EQ_MASK(g0 >> 6);
          if ((m >> lane) & 1ull) {
#pragma clang fp contract(off)
            const int g = g0 + lane;
            const int ow = owner_of(g);
            if (!EQ_I(EQ_DOM + ow)) {
              int sa, sb; double ta, tb;
              candidate(ow, g - EQ_I(EQ_PRE + ow), sa, sb, ta, tb);
              const int slot = EQ_I(EQ_NBASE + ow) + (int)atomicAdd(&EQ_U(EQ_FILL + ow), 1u);
              const int e = EQ_I(EQ_EOF + ow);
              q2.iv[slot] = EdgeIv{e, sa, sb, 0, ta, tb};
              const double tm = (ta + tb) / 2;                 // edge_open's expression
              interpolate_state_dev(q2.sk, q2.A + (int64_t)e * S2, q2.B + (int64_t)e * S2, tm, q2.states + (int64_t)slot * S2);
            }
          }
        }
        sub.lap(q2.ctl, EQ_T_F3);
        eq_release();                   // the pushed records, before their ready flags
        for (int g = lane; g < Tnew; g += 64) {
          const int slot = nb + g;
          __hip_atomic_store((uint32_t *)q2.sample_edge + slot, (uint32_t)q2.iv[slot].e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        sub.lap(q2.ctl, EQ_T_F4);
      }
    }
    if (stop) break;
    // the round's samples are done -- after its pushes have moved the tail (the adds above have returned)
    if (lane == 0) {
      __hip_atomic_fetch_add(q2.ctl + EQ_DONE, (uint32_t)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(q2.ctl + EQ_BATCHES, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(q2.ctl + 124 + (cnt == 64 ? 0 : (cnt >= 32 ? 1 : (cnt >= 2 ? 2 : 3))), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    clk.lap(q2.ctl, EQ_T_FOLD);
  }
}

}  // namespace trk
