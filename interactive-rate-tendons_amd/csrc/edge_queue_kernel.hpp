// edge_queue_kernel.hpp -- `fk_edge_queue`: the edge bisection (VoxelEnvironment::voxelize_valid_backbone_motion,
// motion-planning/VoxelEnvironment.cpp:207-444, behind AbstractVoxelMotionValidator::checkMotion,
// AbstractVoxelMotionValidator.h:143-151) as ONE persistent launch over a device work queue, with a barrier per EDGE instead of
// one per level of all edges (edge_kernel.hpp: "the edge queue" states the scheme and why the results are those of the
// level-synchronous schedule).
//
// A wave's round: take up to 64 pool slots at the queue's head (push order: shallow levels first -- the samples with the longest
// chains of levels still ahead of them; taking pushed samples before the seeds was measured: 10 % longer) -> integrate their
// states with the verdict-only body and its per-point sweep (verdict_kernel.hpp: fk_uniform_body + PointSweep<.., SIG>, the
// signature rows go to the slots' rows) -> fold the verdicts into the edges -> for every edge whose level this wave completed:
// should_subdivide on both halves of the level's intervals, one candidate per lane (signatures_differ_lane), the survivors
// become the edge's next level at the queue's tail -> publish -> count the round's samples as done.
//
// Hand-offs between waves (MI355X: per-XCD L2s, a CU's vector L1 is never refreshed by another CU's stores): every record a wave
// leaves for others -- signature rows, edge_ok, the level records, the pushed slots' intervals and states -- is written with plain
// stores, then `s_waitcnt vmcnt(0)` -> agent-scope release fence -> `s_waitcnt vmcnt(0)` -> the agent-scope atomic that
// hands it over (the edge's `remaining` counter, the slot's ready flag); the wave that receives it -- its decrement returned 1, its
// poll of the ready flag matched -- runs an agent-scope acquire fence and `s_waitcnt vmcnt(0)` before its first load.
// No wave ever waits for a wave that has not started: a wave takes only samples that are published (the semaphore) or seeded, waits
// only while a slot it has taken is being written by a wave inside its push, and leaves when done == tail.  Every wait is
// bounded (EQF_STUCK ends the launch).
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
// integrations): the survivors' masks (two words per 64 intervals of a group: proximal, distal), then per owner lane: its first interval, edge, level base,
// survivor count, domain flag, the new level's first slot and fill count, 1 / validSegmentCount
constexpr int EQ_GROUP_IV = EQ_MAX_CAND;                       // intervals (two candidates each) of one group of finished levels
#define EQ_MASK(i) (((unsigned long long *)vlds)[(i)])
#define EQ_I(i) (((int32_t *)vlds)[(i)])
#define EQ_U(i) (((uint32_t *)vlds)[(i)])
constexpr int EQ_PRE = 2 * EQ_GROUP_IV / 32, EQ_EOF = EQ_PRE + 64, EQ_BASEOF = EQ_EOF + 64, EQ_CNT = EQ_BASEOF + 64, EQ_DOM = EQ_CNT + 64,
              EQ_NBASE = EQ_DOM + 64, EQ_FILL = EQ_NBASE + 64, EQ_RELW = EQ_FILL + 64;     // (word offsets)
#define EQ_REL(i) (((double *)vlds)[EQ_RELW / 2 + (i)])

// What a wave must remember across an integration -- its lanes' edges, the claimed slots, the phase clock -- waits in 272 bytes of
// LDS behind the verdict body's image (words [base, base + 68), base = the image's size in words): held in registers it cost the RK4
// loop 28 more scratch loads per step than fk_verdict's own (measured: the queue's rounds 3 - 5 % longer than the kernel's).
__device__ __forceinline__ int eq_stash_base(int NM) { return VLay<true>::MS + 4 * NM * 64 + SIG_LDS_WORDS; }

// signatures_differ (edge_kernel.hpp) for BOTH halves of one interval by ONE lane: the rows of the interval's ends (ra, rb) and of its
// midpoint (rm) from the tip down, four points per 16-byte load, 18 loads in flight; for each half the first event in tip-first
// order decides, a domain error at a point before a difference at that point.  fp: proximal half (ra, rm), fd: distal half (rm, rb).
__device__ __forceinline__ void signatures_differ_lane2(const uint32_t *__restrict__ ra, const uint32_t *__restrict__ rm,
                                                        const uint32_t *__restrict__ rb, int P, int &fp, int &fd) {
  constexpr int U = 6;
  fp = 0; fd = 0;
  bool open_p = true, open_d = true;
  for (int k = (P - 1) >> 2; k >= 0 && (open_p || open_d); k -= U) {
    uint4 va[U], vm[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int kk = k - u >= 0 ? k - u : 0;
      va[u] = *reinterpret_cast<const uint4 *>(ra + 4 * kk);
      vm[u] = *reinterpret_cast<const uint4 *>(rm + 4 * kk);
      vb[u] = *reinterpret_cast<const uint4 *>(rb + 4 * kk);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (k - u < 0) break;
      const uint32_t wa[4] = {va[u].x, va[u].y, va[u].z, va[u].w}, wm[4] = {vm[u].x, vm[u].y, vm[u].z, vm[u].w},
                     wb[4] = {vb[u].x, vb[u].y, vb[u].z, vb[u].w};
#pragma unroll
      for (int w = 3; w >= 0; w--) {
        if (4 * (k - u) + w >= P) continue;                    // (words of the row's padding)
        const uint32_t a = wa[w], m = wm[w], b = wb[w];
        const uint32_t mx = m & 1023u, my = (m >> 10) & 1023u, mz = (m >> 20) & 1023u;
        // d + 1 in {0, 1, 2} <=> the cells are at most one apart
        const uint32_t px = (a & 1023u) + 1u - mx, py = ((a >> 10) & 1023u) + 1u - my, pz = ((a >> 20) & 1023u) + 1u - mz;
        const uint32_t dx = mx + 1u - (b & 1023u), dy = my + 1u - ((b >> 10) & 1023u), dz = mz + 1u - ((b >> 20) & 1023u);
        const int ep = ((a | m) & SIG_BAD) ? 2 : ((px > 2u || py > 2u || pz > 2u) ? 1 : 0);
        const int ed = ((m | b) & SIG_BAD) ? 2 : ((dx > 2u || dy > 2u || dz > 2u) ? 1 : 0);
        if (open_p && ep) { fp = ep; open_p = false; }
        if (open_d && ed) { fd = ed; open_d = false; }
      }
    }
  }
}

// Take the wave's next samples: pool slots [pos, pos + n).  false: leave.
__device__ __forceinline__ bool eq_claim(const EdgeQueueArgs &q, int &pos, int &n) {
  const int lane = threadIdx.x;
  uint32_t *ctl = q.ctl;
  uint32_t seen = 0xffffffffu;          // done + tail at the last look: a wait that sees them unchanged for too long gives up
  unsigned idle = 0;
  int patience = 0;                     // (lane 0's)
  for (;;) {
    int r_fl = 0, r_pos = 0, r_n = 0, r_end = 0, r_wait = 0;
    uint32_t d = 0, t = 0;
    if (lane == 0) {
      r_fl = (int)eq_load(ctl + EQ_FLAGS);
      d = eq_load(ctl + EQ_DONE);
      eq_wait_vm();                     // `done` is read BEFORE `tail`: done == tail then means that nothing was in flight at that moment
      t = eq_load(ctl + EQ_TAIL);
      const int avail = (int)eq_load(ctl + EQ_AVAIL);
      // whole waves while work is plenty, single samples when it runs dry (an FK costs a wave the same 128 serial steps whatever
      // its lane count: the last samples finish soonest one per idle wave).  "Plenty" counts what is outstanding anywhere -- waiting
      // or being integrated (its children are pushed within a round): a wave that finds less than it wants waits a few
      // microseconds for the pushes in progress instead of running half empty
      const int out = (int)(t - d);
      const int want = out >= 4096 ? 64 : (out < 128 ? 1 : out >> 6);
      if (!r_fl && avail > 0 && avail < want && patience < 24) { patience++; r_wait = 1; }
      else if (!r_fl && avail > 0) {
        const int old = (int)__hip_atomic_fetch_add((int *)(ctl + EQ_AVAIL), -want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int got = old >= want ? want : (old > 0 ? old : 0);
        if (got < want) __hip_atomic_fetch_add((int *)(ctl + EQ_AVAIL), want - got, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (got > 0) { r_pos = (int)__hip_atomic_fetch_add(ctl + EQ_HEAD, (uint32_t)got, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); r_n = got; }
      } else if (!r_fl && d == t) r_end = 1;
    }
    if (eq_bcast(r_fl) || eq_bcast(r_end)) return false;
    if (eq_bcast(r_wait)) { __builtin_amdgcn_s_sleep(16); continue; }
    n = eq_bcast(r_n);
    if (n > 0) { pos = eq_bcast(r_pos); return true; }
    d = (uint32_t)eq_bcast((int)d); t = (uint32_t)eq_bcast((int)t);
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
  EqClock clk;
  clk.start();
#ifdef TRK_EQ_TRACE
  int tr_round = 0;
#endif
  for (;;) {
    int zero = 0;
    asm volatile("" : "+s"(zero));      // (as PointSweep::args: the queue's arguments are re-read when needed, not held across the RK4 loop)
    const EdgeQueueArgs &q = qa[zero];
    int h = 0, cnt = 0;
    if (!eq_claim(q, h, cnt)) break;
    clk.lap(q.ctl, EQ_T_CLAIM);
    const bool live = lane < cnt;
    const int slot = h + lane;
    // ---- the slots' records: written by a wave that is inside its push right now, or long ago ----
    int e_lane = -1;
    {
      bool bail = false;
      if (live) {
        const uint32_t *flag = (const uint32_t *)q.sample_edge + slot;
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
    VL_U(VLay<true>::HIT + lane) = 0u; VL_U(VLay<true>::INPREV + lane) = 0u; VL_F(VLay<true>::DIST + lane) = 0.0f;
    __syncthreads();
    // fk_uniform_body takes lane l of block b as configuration 64 b + l of `states`, live below n: shift both so that it is slot h + l
    const int S = K.state_size;
    const double *st_eff = q.states + ((int64_t)h - (int64_t)blockIdx.x * 64) * S;
    const int64_t n_eff = (int64_t)blockIdx.x * 64 + cnt;
    {
      const int sb = eq_stash_base(ps.NM);
      VL_I(sb + lane) = e_lane;
      if (lane == 0) { VL_I(sb + 64) = h; VL_I(sb + 65) = cnt; VL_U(sb + 66) = (uint32_t)clk.t; VL_U(sb + 67) = (uint32_t)(clk.t >> 32); }
    }
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
    int e2, h2, cnt2;                    // (the stash: see eq_stash_base)
    {
      const VerdictArgs a = *va;
      const int sb = eq_stash_base(a.NM);
      e2 = VL_I(sb + lane); h2 = VL_I(sb + 64); cnt2 = VL_I(sb + 65);
      clk.t = (unsigned long long)VL_U(sb + 66) | ((unsigned long long)VL_U(sb + 67) << 32);
      lv = verdict_decide<N, false, true>(a, fl_, lane < cnt2);
    }
    const bool live2 = lane < cnt2;
    int zero2 = 0;
    asm volatile("" : "+s"(zero2));
    const EdgeQueueArgs &q2 = qa[zero2];
#ifdef TRK_EQ_TRACE
    const uint32_t tr_start = (uint32_t)clk.t;
    const unsigned long long tr_w0 = wall_clock64(), tr_c0 = (unsigned long long)clock64();
    // (the shader clock against the 100 MHz wall clock over a fixed stretch of ALU work: the frequency the chip runs at right now)
    { float x = (float)lane; for (int i = 0; i < 2000; i++) x = __builtin_fmaf(x, 1.0000001f, 0.5f); if (x == 12345.678f) asm volatile("s_nop 0"); }
    const unsigned long long tr_w1 = wall_clock64(), tr_c1 = (unsigned long long)clock64();
    const uint32_t tr_mhz = tr_w1 > tr_w0 ? (uint32_t)((tr_c1 - tr_c0) * 100ull / (tr_w1 - tr_w0)) : 0u;
#endif
    clk.lap(q2.ctl, EQ_T_FK);
#ifdef TRK_EQ_TRACE
    const uint32_t tr_fk = (uint32_t)clk.t;
#endif
    if (__any(lv.pending)) {
      // the exact pairwise self-collision sweep needs every backbone point at once: this wave integrates its round again,
      // storing the points in its own columns of the workspace, and takes sweep_body's verdict for the pending lanes
      // (what fk_sweep_fused_list does for the level-synchronous launches; rare: tight curls only)
      __syncthreads();
      const double *st2 = q2.states + ((int64_t)h2 - (int64_t)blockIdx.x * 64) * K.state_size;
      const int64_t n2 = (int64_t)blockIdx.x * 64 + cnt2;
      fk_uniform_body<N, ROT, false, false>(st2, n2, q2.fb_ld, K, tab, steps, nsteps, q2.fb_out, NoPointHook(), nullptr, nullptr);
      __syncthreads();
      const FusedSweepArgs fa = *sa;
      bool exact = false;
      sweep_body<false>(q2.fb_in, n2, q2.fb_ld, fa.P, fa.CH, fa.NM, K, fa.g, fa.grid, fa.near_grid, 1, fa.debug, nullptr, nullptr, nullptr, &exact);
      __syncthreads();
      const uint32_t npend = (uint32_t)__popcll(__ballot(lv.pending));
      if (lane == 0) __hip_atomic_fetch_add(q2.ctl + EQ_PENDING, npend, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (lv.pending) lv.valid = exact;
      clk.lap(q2.ctl, EQ_T_EXACT);
    }

    // ---- fold ----
    if (live2 && !lv.valid) q2.edge_ok[e2] = 0u;
    eq_release();                       // the round's signature rows and edge_ok stores, before the counters say so
    int old = 0;
    if (live2) old = __hip_atomic_fetch_add(q2.remaining + e2, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool fin = live2 && old == 1;        // this lane folded the last outstanding sample of its edge's level
    bool stop = false;
    EqClock sub = clk;
    sub.lap(q2.ctl, EQ_T_F0);
    if (__any(fin)) {
      eq_acquire();                     // the other samples of these levels: rows, edge_ok, the level records
      {
        const uint32_t nfin = (uint32_t)__popcll(__ballot(fin));
        if (lane == 0) __hip_atomic_fetch_add(q2.ctl + EQ_FINISHED, nfin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // ---- finish the levels: should_subdivide on both halves of every interval, ONE INTERVAL PER LANE ----
      // interval i of an edge's level has its midpoint in pool slot base + i; its distal half (:387-390) and its proximal half
      // (:393-396) are the candidates for the next level.  A finisher lane owns its edge's cnt intervals; the wave deals all of
      // them out 64 at a time.
      int base_l = 0, ncand_l = 0;
      double rel_l = 0.0;
      if (fin && q2.edge_ok[e2] != 0u) {                  // (an invalid sample decides the edge: nothing to open)
        base_l = q2.lvl_base[e2]; ncand_l = q2.lvl_cnt[e2]; rel_l = q2.rel[e2];
        if (2 * ncand_l > EQ_MAX_CAND) {
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
        // a group of owner lanes, in lane order, whose intervals fit the emit masks (EQ_GROUP_IV intervals; one level is at most half of it)
        const bool mine = (todo >> lane) & 1ull;
        int pre = mine ? ncand_l : 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(pre, o, 64); if (lane >= o) pre += v; }     // inclusive prefix
        const bool in_group = mine && pre <= EQ_GROUP_IV;
        const unsigned long long gm = __ballot(in_group);
        todo &= ~gm;
        const int n_g = in_group ? ncand_l : 0;
        int ex = n_g;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(ex, o, 64); if (lane >= o) ex += v; }
        const int T = __shfl(ex, 63, 64);                      // the group's intervals
        ex -= n_g;                                             // exclusive prefix: the lane's first interval
        __syncthreads();
        EQ_I(EQ_PRE + lane) = ex; EQ_I(EQ_EOF + lane) = e2; EQ_I(EQ_BASEOF + lane) = base_l;
        EQ_I(EQ_CNT + lane) = 0; EQ_I(EQ_DOM + lane) = 0; EQ_I(EQ_FILL + lane) = 0;
        EQ_REL(lane) = rel_l;
        __syncthreads();
        if (lane == 0) __hip_atomic_fetch_add(q2.ctl + EQ_CAND, (uint32_t)(2 * T), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // interval g of the group -> its owner lane: the last lane whose first interval is <= g
        auto owner_of = [&](int g) {
          int lo = 0;
#pragma unroll
          for (int st = 32; st > 0; st >>= 1) if (EQ_I(EQ_PRE + lo + st) <= g) lo += st;
          return lo;
        };
        // pass 1: the verdicts of every interval's two halves, survivors counted per owner
        for (int g0 = 0; g0 < T; g0 += 64) {
          const int g = g0 + lane;
          bool emit_p = false, emit_d = false;
          if (g < T) {
#pragma clang fp contract(off)
            const int ow = owner_of(g);
            const int sm = EQ_I(EQ_BASEOF + ow) + (g - EQ_I(EQ_PRE + ow));
            const EdgeIv iv = q2.iv[sm];
            const double tm = (iv.ta + iv.tb) / 2;             // :382
            int fp, fd;
            signatures_differ_lane2(q2.sig + (int64_t)iv.sa * q2.sig_stride, q2.sig + (int64_t)sm * q2.sig_stride,
                                    q2.sig + (int64_t)iv.sb * q2.sig_stride, P, fp, fd);
            if (fp == 2 || fd == 2) atomicOr(&EQ_U(EQ_DOM + ow), 1u);    // std::domain_error in the reference: the edge is invalid
            const double relv = EQ_REL(ow);
            emit_p = fp == 1 && (tm - iv.ta) > relv;           // width rule of :369-372, applied at push time
            emit_d = fd == 1 && (iv.tb - tm) > relv;
            const uint32_t k = (emit_p ? 1u : 0u) + (emit_d ? 1u : 0u);
            if (k) atomicAdd(&EQ_U(EQ_CNT + ow), k);
          }
          const unsigned long long mp = __ballot(emit_p), md = __ballot(emit_d);
          if (lane == 0) { EQ_MASK(2 * (g0 >> 6)) = mp; EQ_MASK(2 * (g0 >> 6) + 1) = md; }
        }
        __syncthreads();
        sub.lap(q2.ctl, EQ_T_F2);
        // the owners' next levels: one allocation at the queue's tail for the whole group
        int total_l = 0;
        if (in_group) {
          if (EQ_I(EQ_DOM + lane)) {
            q2.edge_ok[e2] = 0u;
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
          q2.lvl_base[e2] = nb + nex; q2.lvl_cnt[e2] = total_l; q2.remaining[e2] = total_l;
          q2.nfk[e2] += total_l;                           // every sample of an opened level is evaluated
        }
        __syncthreads();
        // pass 2: the survivors' records at their slots
        for (int g0 = 0; g0 < T; g0 += 64) {
          const unsigned long long mp = EQ_MASK(2 * (g0 >> 6)), md = EQ_MASK(2 * (g0 >> 6) + 1);
          const bool ep = (mp >> lane) & 1ull, ed = (md >> lane) & 1ull;
          if (ep || ed) {
#pragma clang fp contract(off)
            const int g = g0 + lane;
            const int ow = owner_of(g);
            if (!EQ_I(EQ_DOM + ow)) {
              const int sm = EQ_I(EQ_BASEOF + ow) + (g - EQ_I(EQ_PRE + ow));
              const EdgeIv iv = q2.iv[sm];
              const double tm = (iv.ta + iv.tb) / 2;           // :382
              const int e = EQ_I(EQ_EOF + ow);
              int ns = EQ_I(EQ_NBASE + ow) + (int)atomicAdd(&EQ_U(EQ_FILL + ow), (ep ? 1u : 0u) + (ed ? 1u : 0u));
              if (ed) {                                        // the distal half [tm, tb] between the midpoint and the far end
                q2.iv[ns] = EdgeIv{e, sm, iv.sb, 0, tm, iv.tb};
                const double t2 = (tm + iv.tb) / 2;            // edge_open's expression
                interpolate_state_dev(q2.sk, q2.A + (int64_t)e * S2, q2.B + (int64_t)e * S2, t2, q2.states + (int64_t)ns * S2);
                ns++;
              }
              if (ep) {                                        // the proximal half [ta, tm]
                q2.iv[ns] = EdgeIv{e, iv.sa, sm, 0, iv.ta, tm};
                const double t2 = (iv.ta + tm) / 2;
                interpolate_state_dev(q2.sk, q2.A + (int64_t)e * S2, q2.B + (int64_t)e * S2, t2, q2.states + (int64_t)ns * S2);
              }
            }
          }
        }
        sub.lap(q2.ctl, EQ_T_F3);
        eq_release();                   // the pushed records, before their ready flags
        for (int g = lane; g < Tnew; g += 64) {
          const int ns = nb + g;
          __hip_atomic_store((uint32_t *)q2.sample_edge + ns, (uint32_t)q2.iv[ns].e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        eq_wait_vm();                   // the flags are out: the samples may be taken
        if (lane == 0) __hip_atomic_fetch_add((int *)(q2.ctl + EQ_AVAIL), Tnew, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sub.lap(q2.ctl, EQ_T_F4);
      }
    }
    if (stop) break;
    // the round's samples are done -- after its pushes have moved the tail (the adds above have returned)
    if (lane == 0) {
      __hip_atomic_fetch_add(q2.ctl + EQ_DONE, (uint32_t)cnt2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(q2.ctl + EQ_BATCHES, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(q2.ctl + EQ_SIZES + (cnt2 == 64 ? 0 : (cnt2 >= 32 ? 1 : (cnt2 >= 2 ? 2 : 3))), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    clk.lap(q2.ctl, EQ_T_FOLD);
#ifdef TRK_EQ_TRACE
    if (lane == 0 && (blockIdx.x & 255u) == 0u && (blockIdx.x >> 8) < 8u && tr_round < 64) {
      uint32_t *w = q2.ctl + EQ_TRACE + ((blockIdx.x >> 8) * 64 + tr_round) * 4;
      w[0] = tr_start; w[1] = tr_fk; w[2] = (uint32_t)clk.t; w[3] = tr_mhz ? tr_mhz : 1u;
    }
    tr_round++;
#endif
  }
}

}  // namespace trk
