// search_kernel.hpp -- `roadmap_astar`: the graph searches of the lazy query loop (VoxelCachedLazyPRM::solveWithRoadmap ->
// constructSolution -> astarSearch, motion-planning/VoxelCachedLazyPRM.cpp:1977-2096, 2689-2771, 2950-2976) on the device, ONE WAVE PER
// QUERY, thousands of queries in flight.  It is roadmap.hip's host `astar` statement for statement -- the same heuristic (state-space
// distance, sharpened by the landmark bounds), the same relaxation rule (a vertex whose cost improves is opened again), the same
// stopping rule (the goal leaves the open list) -- with the one thing a wave does better than a core: the ~12 arcs of an expanded
// vertex are relaxed by as many lanes at once, each with its own dependent chain of loads (arc -> validity bytes, node record,
// state, landmark row), and the chip hides those latencies behind the other waves.
//
// The open list is what a GPU has no good answer for; here it is split by a threshold T on the key f = g + h:
//   near  (LDS, SR_CAP entries): every entry with f < T, unsorted; the minimum is a wave-wide scan + reduction (a few hundred cycles);
//   far   (global, per wave):    every entry with f >= T, unsorted, append-only between refills.
// near full -> T drops halfway towards near's minimum and the entries above it move to far; near empty -> T rises to a value that
// lets about half a list's worth of far's entries in (found by counting) and they move to near.  Entries are never updated in place:
// a vertex reached again with a better cost gets a new entry, the old one is skipped when it surfaces (its vertex is closed), as on
// the host.
// Every loop is bounded; a query that exceeds a bound (far list, expansions, path length, path buffer) is flagged SR_FALLBACK and the
// host search answers it.  Ties between exactly equal keys may be broken differently than on the host (which breaks them by heap
// order): equal-cost alternative paths, possible only between paths whose fp64 cost sums agree in every bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace trk {

struct SArc { int32_t v, e; double w; };                      // roadmap.hip: Arc
struct SNode { double g, h; int32_t parent, parent_edge; uint32_t stamp, closed; };   // roadmap.hip: Node
constexpr int SR_CAP = 1024;                                  // near-list entries per wave (LDS)
constexpr int SR_MAXS = 12, SR_MAXL = 64;                     // state coordinates, landmarks
constexpr int SR_PATH_MAX = 4096;                             // vertices of a path (per-wave staging)
enum : uint8_t { SR_NO_PATH = 0, SR_FOUND = 1, SR_FALLBACK = 2 };
constexpr uint8_t SR_INVALID = 2;                             // roadmap.hip: V_INVALID
__host__ __device__ inline size_t search_lds_bytes() { return (size_t)SR_CAP * 12 + SR_MAXS * 8 + SR_MAXL * 4; }

struct SearchArgs {
  const int64_t *adj_off; const SArc *adj;                    // CSR adjacency, both directions
  const double *states; const float *lm;                      // [V][S]; [V][L] landmark distances or null
  int32_t S, NT, rot, ret, L;
  double w_rot, w_ret, lm_slack;
  const uint8_t *vstat, *estat;
  int64_t V, E, n_arcs;
  const int32_t *qs, *qg; int64_t nq;                         // the round's queries
  uint32_t *next;                                             // query ticket
  SNode *nodes; uint32_t *gens;                               // [slots][V], [slots]
  double *far_f; int32_t *far_v; int32_t far_cap;             // [slots][far_cap]
  int32_t *stage;                                             // [slots][2 SR_PATH_MAX] path staging
  uint8_t *found; int32_t *poff, *plen;                       // [nq]
  int32_t *pbuf; uint32_t pbuf_cap; uint32_t *pbuf_used;      // packed paths: vertices goal .. start, then their edges
  unsigned long long *expanded;
  int64_t max_pops;
};

// Values that are the same in every lane are told so to the compiler (readfirstlane): counters, thresholds and the popped vertex
// live in scalar registers and the loops around them branch on the scalar unit instead of being predicated lane by lane.
__device__ __forceinline__ int sr_u(int x) { return __builtin_amdgcn_readfirstlane(x); }
// The lane index as the loops below see it: re-read through an empty asm once per iteration, so that the optimiser cannot prove the
// `lane == 0` tests of consecutive iterations equal.  Without this it threads the back edge of the query loop for the 63 lanes that
// do not draw the ticket straight into the loop body -- a second, inner loop that lane 0 is not part of -- and the cross-lane
// operations (readfirstlane, ballot, shuffles) of the body then run without lane 0, who alone writes the list heads (hipcc 7.2:
// faults on garbage indices; found in the listing as a Depth-2 copy of the query loop with the ticket's register set to zero).
__device__ __forceinline__ int sr_opaque(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ double sr_u(double x) {
  const long long b = __double_as_longlong(x);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ int64_t sr_u(int64_t x) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)x), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(x >> 32));
  return (int64_t)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double sr_wave_min(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const double y = __shfl_xor(x, o, 64); x = y < x ? y : x; }
  return sr_u(x);
}
__device__ __forceinline__ double sr_wave_max(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const double y = __shfl_xor(x, o, 64); x = y > x ? y : x; }
  return sr_u(x);
}
__device__ __forceinline__ int sr_wave_sum(int c) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  return sr_u(c);
}

// -DTRK_SEARCH_CHECKS: every index is checked before it is used; the first violation is recorded in the control words (code, lane,
// value) and the wave leaves (debugging aid: the product build carries none of it)
#ifdef TRK_SEARCH_CHECKS
#define SR_CHK(cond, code, val)                                                                                          \
  do {                                                                                                                   \
    if (!(cond)) { a.next[8] = (uint32_t)(code); a.next[9] = (uint32_t)threadIdx.x; a.next[10] = (uint32_t)(val); a.next[11] = (uint32_t)((int64_t)(val) >> 32); return; } \
  } while (0)
#elif defined(TRK_SEARCH_CLAMPS)
#define SR_CHK(cond, code, val)                                                                                          \
  do {                                                                                                                   \
    if (!(cond)) { if (atomicCAS(&a.next[8], 0u, (uint32_t)(code)) == 0u) { a.next[9] = (uint32_t)threadIdx.x; a.next[10] = (uint32_t)(val); a.next[11] = (uint32_t)((int64_t)(val) >> 32); a.next[12] = qi; a.next[13] = (uint32_t)pops_dbg; } val = 0; } \
  } while (0)
#else
#define SR_CHK(cond, code, val) do { } while (0)
#endif
// (clamped build only) a loop that runs past any count it can legitimately reach is recorded and left
#ifdef TRK_SEARCH_CLAMPS
#define SR_LOOP_GUARD(counter, limit, code)                                                                              \
  if (++(counter) > (limit)) { atomicCAS(&a.next[14], 0u, (uint32_t)(code)); a.next[15] = qi; break; }
#else
#define SR_LOOP_GUARD(counter, limit, code)
#endif

// -DTRK_SEARCH_CLOCKS: the 100 MHz clock read at the phase boundaries of an expansion, summed per phase into control words 16.. (profiling aid)
#ifdef TRK_SEARCH_CLOCKS
#define SR_CLK(i) do { const unsigned long long t_ = wall_clock64(); clk[i] += t_ - t_last; t_last = t_; } while (0)
#else
#define SR_CLK(i) do { } while (0)
#endif

__global__ __launch_bounds__(64) void roadmap_astar(SearchArgs a) {
#pragma clang fp contract(off)
  extern __shared__ double sr_lds[];
  double *nf = sr_lds;                                        // [SR_CAP]
  int32_t *nv = (int32_t *)(nf + SR_CAP);                     // [SR_CAP]
  double *gst = (double *)(nv + SR_CAP);                      // [SR_MAXS] the goal's state
  float *glm = (float *)(gst + SR_MAXS);                      // [SR_MAXL] the goal's landmark row
  const int lane_id = threadIdx.x;
  const int64_t slot = blockIdx.x;
  SNode *__restrict__ node = a.nodes + slot * a.V;
  double *__restrict__ ff = a.far_f + slot * (int64_t)a.far_cap;
  int32_t *__restrict__ fv = a.far_v + slot * (int64_t)a.far_cap;
  int32_t *__restrict__ stage = a.stage + slot * (int64_t)(2 * SR_PATH_MAX);
  const double inf = __longlong_as_double(0x7ff0000000000000ll);
  const int S = a.S, L = a.L;

  // roadmap.hip: state_distance + the landmark bounds; every lane for its own vertex.  The loads -- the state row, then the landmark row
  // four float4 at a time (rows are padded to a multiple of four with zeros, which bound nothing) -- are all requested before the
  // first is used: a lane's expansion is a chain of dependent memory round trips, and this keeps it at one for the heuristic.
  const int NT = a.NT, L4 = (L + 3) >> 2;
  const bool rot = a.rot != 0, ret = a.ret != 0;
  const float slack = (float)a.lm_slack;
  auto heuristic = [&](int32_t v) -> double {
    const double *sv = a.states + (int64_t)v * S;
    double x[SR_MAXS];
#pragma unroll
    for (int i = 0; i < SR_MAXS; i++) x[i] = i < S ? sv[i] : 0.0;
    const float4 *lv = (const float4 *)(a.lm + (int64_t)v * (4 * L4));
    float4 y[4];
#pragma unroll
    for (int j = 0; j < 4; j++) y[j] = j < L4 ? lv[j] : float4{0.f, 0.f, 0.f, 0.f};
    double s = 0, t_rot = 0, t_ret = 0;
#pragma unroll
    for (int i = 0; i < SR_MAXS; i++) {
      if (i < S) {
        const double d = x[i] - gst[i];
        if (i < NT) s += d * d;
        else if (i == NT && rot) { double t = fabs(d); t = (t > M_PI) ? 2.0 * M_PI - t : t; t_rot = a.w_rot * t; }
        else t_ret = a.w_ret * sqrt(d * d);
      }
    }
    double h = sqrt(s);
    if (rot) h += t_rot;
    if (ret) h += t_ret;
    if (L4) {
      const float finf = __int_as_float(0x7f800000);
      float best = 0.0f;
      bool cut = false;
      auto bound = [&](float xv, float yv) {
        const float hi = xv > yv ? xv : yv, lo = xv > yv ? yv : xv;
        if (hi == finf) { cut |= lo != hi; return; }
        const float t = (hi - lo) - slack * hi;
        best = t > best ? t : best;
      };
      for (int c = 0; c < L4; c += 4) {
        if (c) {
#pragma unroll
          for (int j = 0; j < 4; j++) y[j] = c + j < L4 ? lv[c + j] : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (c + j < L4) {
            const float4 g = ((const float4 *)glm)[c + j];
            bound(y[j].x, g.x); bound(y[j].y, g.y); bound(y[j].z, g.z); bound(y[j].w, g.w);
          }
        }
      }
      if (cut) return inf;
      if ((double)best > h) h = (double)best;
    }
    return h;
  };

  int lg_ticket = 0;
  (void)lg_ticket;
  for (;;) {
    int lane = sr_opaque(lane_id);
    uint32_t qi = 0, gen = 0;
    SR_LOOP_GUARD(lg_ticket, 100000, 20)
    if (lane == 0) { qi = atomicAdd(a.next, 1u); }
    qi = (uint32_t)__builtin_amdgcn_readfirstlane((int)qi);
    if ((int64_t)qi >= a.nq) break;
    if (lane == 0) { gen = a.gens[slot] + 1u; a.gens[slot] = gen; }
    gen = (uint32_t)__builtin_amdgcn_readfirstlane((int)gen);
    int32_t start = a.qs[qi];
    const int32_t goal = a.qg[qi];
    int64_t pops_dbg = -1;
    (void)pops_dbg;
    SR_CHK(start >= 0 && start < a.V && goal >= 0 && goal < a.V, 1, start);
    __syncthreads();
    if (lane < S) gst[lane] = a.states[(int64_t)goal * S + lane];
    if (lane < 4 * L4) glm[lane] = a.lm[(int64_t)goal * (4 * L4) + lane];
    __syncthreads();

    int n_near = 0, n_far = 0;
    double T = inf;
    int result = SR_NO_PATH;
    unsigned long long exp_q = 0;
    unsigned moves = 0;                                         // times the threshold moved (near full / near empty)
    const double h0 = sr_u(heuristic(start));
    if (h0 != inf) {
      if (lane == 0) { node[start] = SNode{0.0, h0, start, -1, gen, 0u}; nf[0] = h0; nv[0] = start; }
      n_near = 1;
      __syncthreads();
#ifdef TRK_SEARCH_CLOCKS
      unsigned long long clk[6] = {0, 0, 0, 0, 0, 0}, t_last = wall_clock64();
#endif
      for (int64_t pops = 0;; pops++) {
        lane = sr_opaque(lane);
        SR_CLK(4);
        if (pops >= a.max_pops) { result = SR_FALLBACK; break; }
        if (n_near == 0) {
          if (n_far == 0) break;                               // the open list is empty: no path
          // ---- refill: raise T so that about half a list's worth of far's entries come in ----
          double mn = inf, mx = -inf;
          int lg = 0;
          (void)lg;
          for (int i = lane; i < n_far; i += 64) { SR_LOOP_GUARD(lg, 2048, 21) const double f = ff[i]; mn = f < mn ? f : mn; mx = f > mx ? f : mx; }
          mn = sr_wave_min(mn); mx = sr_wave_max(mx);
          double Tn = inf;
          if (n_far > SR_CAP / 2) {
            Tn = mn + (mx - mn) * ((double)(SR_CAP / 2) / (double)n_far);
            if (!(Tn > mn)) Tn = mn + (mx - mn) * 0.5;
            if (!(Tn > mn)) Tn = inf;                          // every key the same: they all qualify (and must fit: checked below)
            for (int tries = 0; tries < 64; tries++) {
              int c = 0;
              lg = 0;
              for (int i = lane; i < n_far; i += 64) { SR_LOOP_GUARD(lg, 2048, 22) c += ff[i] < Tn ? 1 : 0; }
              c = sr_wave_sum(c);
              if (c <= SR_CAP) break;
              const double Th = mn + (Tn - mn) * 0.5;
              if (!(Th > mn) || !(Th < Tn)) { Tn = -inf; break; }   // more equal keys than the list holds
              Tn = Th;
            }
            if (Tn == -inf) { result = SR_FALLBACK; break; }
          }
          // partition far in place: keys below Tn to near, the rest compacted to the front
          int keep = 0;
          lg = 0;
          for (int c0 = 0; c0 < n_far; c0 += 64) {
            SR_LOOP_GUARD(lg, 2048, 23)
            const int i = c0 + lane;
            const bool on = i < n_far;
            const double f = on ? ff[i] : 0.0;
            const int32_t v = on ? fv[i] : 0;
            const bool in = on && f < Tn, stay = on && !in;
            const unsigned long long mi = __ballot(in), ms = __ballot(stay);
            const unsigned long long below = ((unsigned long long)1 << lane) - 1;
            if (in) { const int p = n_near + __popcll(mi & below); if (p < SR_CAP) { nf[p] = f; nv[p] = v; } }
            if (stay) { const int p = keep + __popcll(ms & below); ff[p] = f; fv[p] = v; }
            n_near += __popcll(mi); keep += __popcll(ms);
          }
          if (n_near > SR_CAP) { result = SR_FALLBACK; break; }
          n_far = keep;
          T = Tn;
          moves++;
          __syncthreads();
          if (n_near == 0) { result = SR_FALLBACK; break; }    // (cannot happen: the minimum qualifies)
        }
        SR_CLK(0);
        // ---- pop: the smallest key of near ----
        double best = inf;
        int bi = -1;
        int lg2 = 0;
        (void)lg2;
        for (int i0 = 0; i0 < n_near; i0 += 256) {                // (four reads in flight per lane; the order of the comparisons is the index order)
          SR_LOOP_GUARD(lg2, 32, 24)
          double f4[4];
#pragma unroll
          for (int j = 0; j < 4; j++) { const int i = i0 + 64 * j + lane; f4[j] = i < n_near ? nf[i] : inf; }
#pragma unroll
          for (int j = 0; j < 4; j++) { const int i = i0 + 64 * j + lane; if (i < n_near && (f4[j] < best || bi < 0)) { best = f4[j]; bi = i; } }
        }
        const double fmin = sr_wave_min(best);
        const unsigned long long who = __ballot(bi >= 0 && best == fmin);
        pops_dbg = pops;
        int idx = sr_u(__shfl(bi, __ffsll((long long)who) - 1, 64));
        SR_CHK(idx >= 0 && idx < n_near, 2, idx);
        int32_t u = sr_u(nv[idx]);
        SR_CHK(u >= 0 && u < a.V, 3, u);
        __syncthreads();
        if (lane == 0) { nf[idx] = nf[n_near - 1]; nv[idx] = nv[n_near - 1]; }
        n_near--;
        __syncthreads();
        SR_CLK(1);
        const SNode nu_ = node[u];
        const int64_t a0_ = a.adj_off[u], a1_ = a.adj_off[u + 1];       // (requested with the record: both hang on u alone)
        const double nu_g = sr_u(nu_.g);
        if (sr_u((int)nu_.closed)) continue;                               // a stale entry of a vertex already expanded with a better cost
        if (lane == 0) node[u].closed = 1u;
        exp_q++;
        if (u == goal) { result = SR_FOUND; break; }
        const int64_t a0 = sr_u(a0_);
        int64_t a1 = sr_u(a1_);
        SR_CHK(a0 >= 0 && a0 <= a1 && a1 <= a.n_arcs, 4, a1);
        SR_CLK(2);
        bool failed = false;
        lg2 = 0;
        for (int64_t base = a0; base < a1 && !failed; base += 64) {
          SR_LOOP_GUARD(lg2, 4096, 25)
          const int64_t k = base + lane;
          bool push = false;
          double fp = 0.0;
          int32_t vp = 0;
          if (k < a1) {
            SArc arc = a.adj[k];
            SR_CHK(arc.v >= 0 && arc.v < a.V, 5, arc.v);
            SR_CHK(arc.e >= 0 && arc.e < a.E, 6, arc.e);
            // everything the relaxation can need is requested at once, whether or not it turns out to be needed: validity bytes,
            // the neighbour's record, and the rows of its heuristic (one memory round trip instead of three)
            const uint8_t es = a.estat[arc.e], vs = a.vstat[arc.v];
            const SNode nn = node[arc.v];
            const double hv = heuristic(arc.v);
            if (es != SR_INVALID && vs != SR_INVALID) {
              const double gv = nu_g + arc.w;
              const bool first = nn.stamp != gen;
              if (first || gv < nn.g) {
                const double h = first ? hv : nn.h;                  // h(v) is fixed for the query: computed when v is first reached
                if (h == inf) node[arc.v] = SNode{gv, h, first ? -1 : nn.parent, first ? -1 : nn.parent_edge, gen, 1u};
                else { node[arc.v] = SNode{gv, h, u, arc.e, gen, 0u}; push = true; fp = gv + h; vp = arc.v; }
              }
            }
          }
          // ---- append: keys below T to near, the others to far ----
          unsigned long long mn_ = __ballot(push && fp < T);
          SR_CLK(3);
          int lg3 = 0;
          (void)lg3;
          while (n_near + __popcll(mn_) > SR_CAP) {
            SR_LOOP_GUARD(lg3, 4096, 26)
            // near is full: T drops halfway towards its smallest key, what lies above moves to far
            double lo = inf, hi = -inf;
            int lg4 = 0;
            (void)lg4;
            for (int i = lane; i < n_near; i += 64) { SR_LOOP_GUARD(lg4, 32, 27) const double f = nf[i]; lo = f < lo ? f : lo; hi = f > hi ? f : hi; }
            lo = sr_wave_min(lo); hi = sr_wave_max(hi);
            const double top = T < inf ? T : hi;
            const double Tn = lo + (top - lo) * 0.5;
            if (!(Tn > lo) || !(Tn < top)) { failed = true; break; }
            int keep = 0;
            lg4 = 0;
            for (int c0 = 0; c0 < n_near; c0 += 64) {
              SR_LOOP_GUARD(lg4, 32, 28)
              const int i = c0 + lane;
              const bool on = i < n_near;
              const double f = on ? nf[i] : 0.0;
              const int32_t v = on ? nv[i] : 0;
              __syncthreads();
              const bool stay = on && f < Tn, out = on && !stay;
              const unsigned long long ms = __ballot(stay), mo = __ballot(out);
              const unsigned long long below = ((unsigned long long)1 << lane) - 1;
              if (stay) { const int p = keep + __popcll(ms & below); nf[p] = f; nv[p] = v; }
              if (out) { const int p = n_far + __popcll(mo & below); if (p < a.far_cap) { ff[p] = f; fv[p] = v; } }
              keep += __popcll(ms); n_far += __popcll(mo);
              __syncthreads();
            }
            if (n_far > a.far_cap) { failed = true; break; }
            n_near = keep;
            T = Tn;
            moves++;
            mn_ = __ballot(push && fp < T);
          }
          if (failed) break;
          const unsigned long long mf_ = __ballot(push && !(fp < T));
          if (n_far + __popcll(mf_) > a.far_cap) { failed = true; break; }
          const unsigned long long below = ((unsigned long long)1 << lane) - 1;
          if (push && fp < T) { int p = n_near + __popcll(mn_ & below); SR_CHK(p >= 0 && p < SR_CAP, 9, p); nf[p] = fp; nv[p] = vp; }
          else if (push) { int p = n_far + __popcll(mf_ & below); SR_CHK(p >= 0 && p < a.far_cap, 8, p); ff[p] = fp; fv[p] = vp; }
          n_near += __popcll(mn_); n_far += __popcll(mf_);
          __syncthreads();
        }
        if (failed) { result = SR_FALLBACK; break; }
      }
#ifdef TRK_SEARCH_CLOCKS
      if (lane == 0) {
        for (int i = 0; i < 6; i++) atomicAdd((unsigned long long *)(a.next + 16) + i, clk[i]);
        atomicMax((unsigned long long *)(a.next + 16) + 6, clk[0] + clk[1] + clk[2] + clk[3] + clk[4]);
      }
#endif
    }
    // ---- the path, goal ... start, and its edges ----
    int nvert = 0;
    if (result == SR_FOUND) {
      if (lane == 0) {
        int32_t v = goal;
        for (;;) {
          if (nvert >= SR_PATH_MAX) { nvert = -1; break; }
          SR_CHK(v >= 0 && v < a.V, 7, v);
          stage[nvert] = v;
          if (v == start) { nvert++; break; }
          const SNode nd = node[v];
          stage[SR_PATH_MAX + nvert] = nd.parent_edge;
          v = nd.parent;
          nvert++;
        }
      }
      nvert = __builtin_amdgcn_readfirstlane(nvert);
      if (nvert <= 0) result = SR_FALLBACK;
    }
    if (result == SR_FOUND) {
      uint32_t off = 0;
      const uint32_t need = (uint32_t)(2 * nvert - 1);
      if (lane == 0) off = atomicAdd(a.pbuf_used, need);
      off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
      if ((uint64_t)off + need > a.pbuf_cap) result = SR_FALLBACK;
      else {
        __threadfence_block();
        int lg5 = 0;
        (void)lg5;
        for (int i = lane; i < nvert; i += 64) { SR_LOOP_GUARD(lg5, 128, 29) a.pbuf[off + i] = stage[i]; }
        lg5 = 0;
        for (int i = lane; i < nvert - 1; i += 64) { SR_LOOP_GUARD(lg5, 128, 30) a.pbuf[off + nvert + i] = stage[SR_PATH_MAX + i]; }
        if (lane == 0) { a.poff[qi] = (int32_t)off; a.plen[qi] = nvert; }
      }
    }
    if (lane == 0) {
      a.found[qi] = (uint8_t)result;
      atomicAdd(a.expanded, exp_q);
      if (moves) atomicAdd(a.next + 4, moves);
    }
  }
}

}  // namespace trk
