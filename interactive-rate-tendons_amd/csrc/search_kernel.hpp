// search_kernel.hpp -- `roadmap_astar`: the graph searches of the lazy query loop (VoxelCachedLazyPRM::solveWithRoadmap ->
// constructSolution -> astarSearch, motion-planning/VoxelCachedLazyPRM.cpp:1977-2096, 2689-2771, 2950-2976) on the device, ONE WAVE PER
// QUERY, thousands of queries in flight.  It is roadmap.hip's host `astar` -- the same heuristic (state-space distance, sharpened by the
// landmark bounds), the same relaxation rule (a vertex whose cost improves is opened again), the same stopping rule (the goal leaves
// the open list as its minimum) -- shaped for a wave:
//   * a step takes up to SR_K = 6 vertices off the open list at once (the minimum, and the smallest of the other lanes' minima) -- as
//     many as their arcs fill the wave's 64 lanes, one arc per lane: an open-list entry carries its vertex's arc count next to the
//     vertex (5 bits of the word), so the lanes are dealt out before anything is loaded (~5 vertices of a 10-nearest roadmap).  Arcs
//     live in ADJACENCY ROWS at a fixed stride of SR_D = 16 per vertex (a vertex with more chains further rows through its last slot,
//     1 - 3 % of a k-nearest roadmap's vertices), so their address follows from the vertex alone and their load leaves together with
//     the vertex's own record.  A step is then a chain of TWO dependent memory round trips -- record + arcs of the popped vertices;
//     per arc the validity bytes, the neighbour's record and arc count and the rows of its heuristic, all requested at once --
//     whatever the number of lanes busy (round 4: three, through CSR offsets).
//     Expanding a vertex that is not the minimum is what any best-first search with re-opening may do: the stopping rule alone makes
//     the returned cost optimal, and with it the path (the optimum is unique unless two paths' fp64 cost sums agree in every bit).
//     One vertex per step (kbest = 1) is the host's order of expansions exactly (same count).
//   * two lanes of a step may reach the same neighbour from different parents; the better one must win, whole: the lanes agree through
//     three small LDS tables (owner by hash of the vertex, smallest cost, lowest lane among equals) before anyone writes.
//   * the search's per-vertex state (g, h, parent, closed) lives in a HASH TABLE sized to the search, not to the roadmap: 32-byte
//     records {g, h, parent, parent edge, vertex, generation << 1 | closed}, two to a 64-byte line, linear probing from the vertex's
//     line; a record of another generation is free, so nothing is ever cleared.  Every wave slot owns a table of 2^lc0 records (4 096:
//     a search of ~1 000 expansions touches ~3 000 vertices); a search that fills three quarters of its table moves -- rehashing what it
//     has -- into one four times the size claimed from a shared pool (bitmaps, one atomic), and returns it when it ends; no free table:
//     the search is handed back to the host.  Two lanes wanting the same free record in one step settle it through an LDS word (the
//     loser probes on): plain loads and stores only, as for the dense arrays this replaces (round 4: 32 B x V per slot, 9.8 GB at 10^5
//     vertices; now ~0.2 MB per slot + the pool, whatever V is).
//
// The open list is what a GPU has no good answer for; here it is split by a threshold T on the key f = g + h:
//   near  (LDS, SR_CAP entries): every entry with f < T, unsorted; every lane scans its share, a wave reduction finds the minimum;
//   far   (global, beside the table): every entry with f >= T, unsorted, append-only between refills.
// near full -> T drops halfway towards near's minimum and the entries above it move to far; near empty -> T rises to a value that
// lets about half a list's worth of far's entries in (found by counting) and they move to near.  Entries are never updated in place:
// a vertex reached again with a better cost gets a new entry, the old one is skipped when it surfaces (its vertex is closed), as on
// the host.
// Every loop is bounded; a query that exceeds a bound (expansion budget, no larger table free, path length, path buffer) is flagged
// SR_FALLBACK and the host search answers it.
//
// The query loop draws its ticket with an atomic EVERY lane executes (lane 0 adds one, the others zero): no `if (lane == 0)` stands
// between the loop head and the cross-lane operations, so there is no lane-dependent branch for the optimiser to thread the back edge
// through (hipcc 7.2 did that to round 4's `if (lane == 0) ticket = atomicAdd(..)`: the 63 other lanes got an inner copy of the loop
// that lane 0 was not part of, and the kernel faulted; tests/test_kernel_resources.py checks the listing for a single ticket site).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace trk {

struct SArc { int32_t v, e; double w; };                      // roadmap.hip: Arc; v = SR_ARC_NONE: unused slot, SR_ARC_MORE: e = the next row;
                                                              // in the kernel's rows v carries, above SR_VBITS, the lanes the neighbour's own row needs
struct SRec { double g, h; int32_t parent, parent_edge; uint32_t key, tag; };   // tag = generation << 1 | closed
#ifndef TRK_SR_CAP
#define TRK_SR_CAP 640
#endif
constexpr int SR_CAP = TRK_SR_CAP;                            // near-list entries per wave (LDS)
constexpr int SR_MAXS = 12, SR_MAXL = 64;                     // state coordinates, landmarks
constexpr int SR_K = 12;                                      // vertices expanded per step, at most (as many as their arcs fill two passes of 64 lanes: ~8)
constexpr int SR_VBITS = 26;                                  // an open-list word: vertex | lanes its first row needs << SR_VBITS
constexpr int SR_D = 16;                                      // arcs per adjacency row = lanes per expanded vertex
constexpr int SR_TAB = 128;                                   // slots of the conflict tables
constexpr int SR_CLASSES = 4;                                 // table sizes: 2^lc0 records, then x 4 per class
constexpr int SR_CTL_WORDS = 136;                              // control words ahead of the pool bitmaps
constexpr int32_t SR_ARC_NONE = -1, SR_ARC_MORE = -2;
constexpr float SR_LM_FAR = 1e30f;                            // a landmark distance of +inf as the vertices' rows store it
enum : uint8_t { SR_NO_PATH = 0, SR_FOUND = 1, SR_FALLBACK = 2 };
constexpr uint8_t SR_INVALID = 2;                             // roadmap.hip: V_INVALID
// near keys | near vertices | goal state | goal landmark row | conflict tables (owner, cost, lane): 16 waves per CU
// ... | the step's lane owners (a byte per lane of its two passes)
__host__ __device__ inline size_t search_lds_bytes() { return (size_t)SR_CAP * 12 + SR_MAXS * 8 + SR_MAXL * 4 + (size_t)SR_TAB * 16 + 128; }
// a table of C records with its far list beside it: records | far keys | far vertices
__host__ __device__ inline size_t search_chunk_bytes(int lc) { return ((size_t)1 << lc) * 44; }
// a vertex's row: state, then landmark distances from a 16-byte boundary; rows never straddle a 128-byte line they could share
__host__ __device__ inline int search_lm_offset(int S) { return (8 * S + 15) & ~15; }
__host__ __device__ inline int search_row_bytes(int S, int L) {
  const int size = search_lm_offset(S) + 16 * ((L + 3) >> 2);
  return size <= 32 ? 32 : (size <= 64 ? 64 : ((size + 127) & ~127));
}

struct SearchArgs {
  const SArc *rows;                                           // [V + continuation rows][SR_D]
  const double *states; const float *lm;                      // a vertex's state (S doubles) and its landmark distances (L rounded up to a
  int32_t row_bytes;                                          // multiple of 4 floats; null: none) sit in ONE row of row_bytes bytes (a 128-byte
                                                              // line at 4 coordinates + 16 landmarks): `lm` = `states` + the row's landmark offset
  int32_t S, NT, rot, ret, L;
  double w_rot, w_ret, lm_slack;
  const uint8_t *vstat, *estat;
  const uint8_t *deg;                                         // [V] lanes a vertex's first row needs: its arcs, 1 at least, SR_D at most
  int64_t V, E;
  const int32_t *qs, *qg; int64_t nq;                         // the round's queries
  uint32_t *next;                                             // control words: [0] query ticket, [1] path words used, [2..3] expansions, [4] list
                                                              // moves, [5] table growths, [6] most records a search held, [16..] clocks;
                                                              // from SR_CTL_WORDS on the pool bitmaps (bit set = table taken)
  char *base;                                                 // [slots] tables of class 0
  char *pool[SR_CLASSES];                                     // [pool_n[c]] tables of class c >= 1
  int32_t pool_n[SR_CLASSES], pool_word[SR_CLASSES];          // tables per class; first bitmap word of the class (index into next)
  int32_t lc0;                                                // log2 of the records of a class-0 table
  uint32_t gen_base;                                          // query i of the launch searches under generation gen_base + i + 1 (< 2^31)
  uint8_t *found; int32_t *poff, *plen;                       // [nq]
  uint32_t *handback;                                         // [nq] in pinned host memory, or null: set the moment a search is handed back, so
                                                              // that the host threads start on it while the kernel still runs
  int32_t *pbuf; uint32_t pbuf_cap; uint32_t *pbuf_used;      // packed paths: vertices goal .. start, then their edges
  unsigned long long *expanded;
  int64_t max_pops;                                           // expansions a search may spend before it is handed back
  int32_t kbest;                                              // 1 .. SR_K
};

// Values that are the same in every lane are told so to the compiler (readfirstlane): counters, thresholds and the popped vertices
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
// Wave reductions on the DPP path (no LDS traffic: a ds_bpermute butterfly is twelve LDS round trips for a double): rotations by
// 8, 4, 2, 1 inside each row of 16 lanes leave the row's result in all of its lanes, row_bcast15 / row_bcast31 fold the rows into
// lane 63, which is read back as a scalar.
#define SR_DPP32(v, ctrl, rmask) __builtin_amdgcn_update_dpp((v), (v), (ctrl), (rmask), 0xf, false)
#define SR_DPP_STEP64(x, ctrl, rmask, OP)                                                                                \
  do {                                                                                                                   \
    const long long b_ = __double_as_longlong(x);                                                                        \
    const int lo_ = SR_DPP32((int)(unsigned)b_, ctrl, rmask), hi_ = SR_DPP32((int)(unsigned)(b_ >> 32), ctrl, rmask);     \
    const double y_ = __longlong_as_double((long long)(((unsigned long long)(unsigned)hi_ << 32) | (unsigned)lo_));      \
    x = (y_ OP x) ? y_ : x;                                                                                              \
  } while (0)
__device__ __forceinline__ double sr_lane63(double x) {
  const long long b = __double_as_longlong(x);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <class T> __device__ __forceinline__ T sr_lane(T x, int l) {        // lane l's 64-bit value as a scalar (v_readlane, no LDS)
  static_assert(sizeof(T) == 8, "two dwords");
  unsigned long long b;
  __builtin_memcpy(&b, &x, 8);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
  b = ((unsigned long long)hi << 32) | lo;
  T y;
  __builtin_memcpy(&y, &b, 8);
  return y;
}
__device__ __forceinline__ double sr_wave_min(double x) {
  SR_DPP_STEP64(x, 0x128, 0xf, <); SR_DPP_STEP64(x, 0x124, 0xf, <); SR_DPP_STEP64(x, 0x122, 0xf, <); SR_DPP_STEP64(x, 0x121, 0xf, <);
  SR_DPP_STEP64(x, 0x142, 0xa, <); SR_DPP_STEP64(x, 0x143, 0xc, <);
  return sr_lane63(x);
}
__device__ __forceinline__ double sr_wave_max(double x) {
  SR_DPP_STEP64(x, 0x128, 0xf, >); SR_DPP_STEP64(x, 0x124, 0xf, >); SR_DPP_STEP64(x, 0x122, 0xf, >); SR_DPP_STEP64(x, 0x121, 0xf, >);
  SR_DPP_STEP64(x, 0x142, 0xa, >); SR_DPP_STEP64(x, 0x143, 0xc, >);
  return sr_lane63(x);
}
// ... of a 32-bit key: one v_min_u32 with a DPP operand per stage.  The minimum of 64 non-negative doubles, exactly, is two of these
// (the high words, then the low words among the lanes that hold the smallest high word): ~20 instructions where the 64-bit
// compare-and-select chain takes ~100, and a step runs it once per vertex it takes.
__device__ __forceinline__ uint32_t sr_wave_min_u32(uint32_t x) {
#define SR_MIN_STEP(ctrl, rmask) do { const uint32_t y_ = (uint32_t)SR_DPP32((int)x, ctrl, rmask); x = y_ < x ? y_ : x; } while (0)
  SR_MIN_STEP(0x128, 0xf); SR_MIN_STEP(0x124, 0xf); SR_MIN_STEP(0x122, 0xf); SR_MIN_STEP(0x121, 0xf);
  SR_MIN_STEP(0x142, 0xa); SR_MIN_STEP(0x143, 0xc);
#undef SR_MIN_STEP
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
// Inclusive prefix sum / running maximum over the wave's lanes (rows of 16 by row_shr 1, 2, 4, 8 -- a lane without a source keeps the
// identity --, then the row ends broadcast into the following rows).
__device__ __forceinline__ int sr_wave_scan_add(int x) {
#define SR_SCAN_STEP(ctrl, rmask) x += __builtin_amdgcn_update_dpp(0, x, ctrl, rmask, 0xf, false)
  SR_SCAN_STEP(0x111, 0xf); SR_SCAN_STEP(0x112, 0xf); SR_SCAN_STEP(0x114, 0xf); SR_SCAN_STEP(0x118, 0xf);
  SR_SCAN_STEP(0x142, 0xa); SR_SCAN_STEP(0x143, 0xc);
#undef SR_SCAN_STEP
  return x;
}
__device__ __forceinline__ uint32_t sr_wave_scan_max(uint32_t x) {
#define SR_SCAN_STEP(ctrl, rmask) do { const uint32_t y_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rmask, 0xf, false); x = y_ > x ? y_ : x; } while (0)
  SR_SCAN_STEP(0x111, 0xf); SR_SCAN_STEP(0x112, 0xf); SR_SCAN_STEP(0x114, 0xf); SR_SCAN_STEP(0x118, 0xf);
  SR_SCAN_STEP(0x142, 0xa); SR_SCAN_STEP(0x143, 0xc);
#undef SR_SCAN_STEP
  return x;
}
__device__ __forceinline__ int sr_wave_sum(int c) {
  c += SR_DPP32(c, 0x128, 0xf); c += SR_DPP32(c, 0x124, 0xf); c += SR_DPP32(c, 0x122, 0xf); c += SR_DPP32(c, 0x121, 0xf);
  // (the row sums are in every lane of their rows: fold the four rows through scalars)
  return __builtin_amdgcn_readlane(c, 0) + __builtin_amdgcn_readlane(c, 16) + __builtin_amdgcn_readlane(c, 32) + __builtin_amdgcn_readlane(c, 48);
}


// -DTRK_SEARCH_CLOCKS: the 100 MHz clock read at the phase boundaries of a step, summed per phase into control words 16.. (profiling aid)
#ifdef TRK_SEARCH_CLOCKS
#define SR_CLK(i) do { const unsigned long long t_ = wall_clock64(); clk[i] += t_ - t_last; t_last = t_; } while (0)
#else
#define SR_CLK(i) do { } while (0)
#endif

// a free table of class c from the shared pool (one lane): -1 when none is free
__device__ inline int sr_pool_claim(const SearchArgs &a, int c) {
  const int n = a.pool_n[c];
  uint32_t *bits = a.next + a.pool_word[c];
  for (int w = 0; w * 32 < n; w++) {
    const int left = n - w * 32;
    const uint32_t all = left >= 32 ? 0xffffffffu : ((1u << left) - 1u);
    for (int tries = 0; tries < 32; tries++) {
      const uint32_t cur = __hip_atomic_load(&bits[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t fr = ~cur & all;
      if (!fr) break;
      const uint32_t bit = fr & (0u - fr);
      if (!(atomicOr(&bits[w], bit) & bit)) return w * 32 + (__ffs((int)bit) - 1);
    }
  }
  return -1;
}
// A pooled table passes between waves on different XCDs inside one launch.  What its next user reads of the previous one's records is
// harmless whatever it sees (a record of another generation is free, stale or not); what must not happen is the previous user's dirty
// lines leaving ITS XCD's L2 after the next user has written the same lines through another L2.  So the wave that returns a table
// first writes its XCD's dirty lines back (agent-scope release: the caller's fence), and only then clears the bit.
__device__ inline void sr_pool_release(const SearchArgs &a, int c, int idx) { atomicAnd(a.next + a.pool_word[c] + (idx >> 5), ~(1u << (idx & 31))); }

// SX: the state coordinates the heuristic keeps in registers (4, 8 or SR_MAXS; S <= SX)
template <int SX> __global__ __launch_bounds__(64) void roadmap_astar(SearchArgs a) {
#pragma clang fp contract(off)
  extern __shared__ double sr_lds[];
  double *nf = sr_lds;                                        // [SR_CAP]
  int32_t *nv = (int32_t *)(nf + SR_CAP);                     // [SR_CAP]
  double *gst = (double *)(nv + SR_CAP);                      // [SR_MAXS] the goal's state
  float *glm = (float *)(gst + SR_MAXS);                      // [SR_MAXL] the goal's landmark row
  unsigned long long *tab_key = (unsigned long long *)(glm + SR_MAXL);   // [SR_TAB] smallest cost offered to the slot's vertex
  uint32_t *tab_owner = (uint32_t *)(tab_key + SR_TAB);       // [SR_TAB] a lane that claimed the slot
  uint32_t *tab_low = tab_owner + SR_TAB;                     // [SR_TAB] lowest lane among those offering the smallest cost
  uint8_t *own8 = (uint8_t *)(tab_low + SR_TAB);              // [128] per lane of a step's two passes: the candidate lane (+ 1) whose group starts here
  const int lane = threadIdx.x;
  const int64_t slot = blockIdx.x;
  const double inf = __longlong_as_double(0x7ff0000000000000ll);
  const int S = a.S, L = a.L;
  const int kbest = a.kbest < 1 ? 1 : (a.kbest > SR_K ? SR_K : a.kbest);
  const unsigned long long below = ((unsigned long long)1 << lane) - 1;
  const SArc no_arc = SArc{SR_ARC_NONE, -1, 0.0};

  // roadmap.hip: state_distance + the landmark bounds; every lane for its own vertex.  The loads -- the state row, then the landmark row
  // four float4 at a time (rows are padded to a multiple of four with zeros, which bound nothing) -- are all requested before the
  // first is used: a lane's relaxation is a chain of dependent memory round trips, and this keeps it at one for the heuristic.
  const int NT = a.NT, L4 = (L + 3) >> 2;
  const bool rot = a.rot != 0, ret = a.ret != 0;
  const float slack = (float)a.lm_slack;
  auto heuristic = [&](int32_t v) -> double {
    const double *sv = (const double *)((const char *)a.states + (int64_t)v * a.row_bytes);
    double x[SX];
#pragma unroll
    for (int i = 0; i < SX; i++) x[i] = i < S ? sv[i] : 0.0;
    const float4 *lv = (const float4 *)((const char *)a.lm + (int64_t)v * a.row_bytes);
    float4 y[4];
#pragma unroll
    for (int j = 0; j < 4; j++) y[j] = j < L4 ? lv[j] : float4{0.f, 0.f, 0.f, 0.f};
    double s = 0, t_rot = 0, t_ret = 0;
#pragma unroll
    for (int i = 0; i < SX; i++) {
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
      // (the rows hold SR_LM_FAR where the host's table holds +inf: "not connected to this landmark".  Two far entries bound nothing --
      // their term is hugely negative --, one far entry makes the term huge: the vertex and the goal lie in different components, as
      // the host's explicit test says; no comparison with infinity per landmark)
      float best = 0.0f;
      auto bound = [&](float xv, float yv) {
        const float hi = fmaxf(xv, yv), lo = fminf(xv, yv);
        const float t = (hi - lo) - slack * hi;
        best = fmaxf(best, t);
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
      if (best >= 0.5f * SR_LM_FAR) return inf;
      if ((double)best > h) h = (double)best;
    }
    return h;
  };

#ifdef TRK_SEARCH_CLOCKS
  if (lane == 0) atomicMin((unsigned long long *)(a.next + 40), wall_clock64());
#endif
  for (;;) {
    // the ticket: one atomic that every lane executes (see the header)
    uint32_t qi = atomicAdd(a.next, lane == 0 ? 1u : 0u);
    qi = (uint32_t)__builtin_amdgcn_readlane((int)qi, 0);
    if ((int64_t)qi >= a.nq) break;
    const uint32_t gen = a.gen_base + qi + 1u;
    const int32_t start = a.qs[qi], goal = a.qg[qi];
    __syncthreads();
    if (lane < S) gst[lane] = ((const double *)((const char *)a.states + (int64_t)goal * a.row_bytes))[lane];
    if (lane < 4 * L4) glm[lane] = ((const float *)((const char *)a.lm + (int64_t)goal * a.row_bytes))[lane];
    __syncthreads();

    // ---- the search's table: class, index in the class, records, far list; all wave-uniform ----
    int cls = 0, chunk = (int)slot, lc = a.lc0;
    char *cb = a.base + slot * (int64_t)search_chunk_bytes(lc);
    SRec *tb = (SRec *)cb;
    double *ff = (double *)(cb + ((size_t)32 << lc));
    int32_t *fv = (int32_t *)(cb + ((size_t)40 << lc));
    int count = 0, n_near = 0, n_far = 0, n_dead = 0;
    unsigned grows = 0;

    // v's record: linear probing from the first record of v's line; a record of another generation ends the probe (free).  The
    // first line is loaded by the CALLER (slot0 -> r0, r1) together with whatever else it needs of v, so that a lookup adds no
    // round trip of its own; resolve() goes on from there -- to further lines only when both records belong to other vertices.
    // found: rec / p are the record and its index; not found: p is the free record the probe stopped at.
    auto slot0 = [&](int32_t v) -> uint32_t { return (((uint32_t)v * 2654435761u) >> (33 - lc)) << 1; };
    auto resolve = [&](int32_t v, SRec r0, SRec r1, SRec &rec, uint32_t &p) -> bool {
      const uint32_t m = (1u << lc) - 1u;
      for (uint32_t t = 0; t <= m; t += 2) {
        if ((r0.tag >> 1) != gen) return false;
        if (r0.key == (uint32_t)v) { rec = r0; return true; }
        if ((r1.tag >> 1) != gen) { p = p + 1; return false; }
        if (r1.key == (uint32_t)v) { rec = r1; p = p + 1; return true; }
        p = (p + 2) & m;
        r0 = tb[p]; r1 = tb[p + 1];
      }
      return false;
    };
    auto lookup = [&](int32_t v, bool on, SRec &rec, uint32_t &p) -> bool {
      p = slot0(v);
      if (!on) return false;
      const SRec r0 = tb[p], r1 = tb[p + 1];
      return resolve(v, r0, r1, rec, p);
    };
    // a free record of table `t` (2^tlc records) for vertex v, on its probe path from p on: lanes that want the same record in the
    // same call settle it through an LDS word (last writer wins; the others probe on).  Wave-uniform call; the winner marks the
    // record taken (key, generation) at once, the caller fills in the rest.  known_free: p is where a lookup of this step ended, so
    // the first round need not look at it again (a dependent load saved per pass).
    auto claim = [&](SRec *t, int tlc, bool want, int32_t v, uint32_t p, bool known_free) -> uint32_t {
      const uint32_t m = (1u << tlc) - 1u;
      bool pending = want;
      for (int round = 0; round < 1024; round++) {
        if (!__ballot(pending)) break;
        if (pending) {
          if (round > 0 || !known_free)
            for (uint32_t k = 0; k <= m; k++) { if ((t[p].tag >> 1) != gen) break; p = (p + 1) & m; }
          tab_owner[p & (SR_TAB - 1)] = (uint32_t)lane;
        }
        __syncthreads();
        const bool won = pending && tab_owner[p & (SR_TAB - 1)] == (uint32_t)lane;
        if (won) { t[p].key = (uint32_t)v; t[p].tag = gen << 1; pending = false; }
        __syncthreads();
      }
      return p;
    };
    // into a table four times the size (or the next size that has one free): what the search has is rehashed, the far list copied,
    // the old table returned to the pool.  false: no table free (the search is handed back).
    auto grow = [&]() -> bool {
      int nc = cls, idx = -1;
      while (idx < 0 && nc + 1 < SR_CLASSES) {
        nc++;
        if (lane == 0) idx = sr_pool_claim(a, nc);
        idx = __builtin_amdgcn_readfirstlane(idx);
      }
      if (idx < 0) return false;
      const int nlc = a.lc0 + 2 * nc;
      char *nb = a.pool[nc] + (int64_t)idx * (int64_t)search_chunk_bytes(nlc);
      SRec *nt = (SRec *)nb;
      double *nff = (double *)(nb + ((size_t)32 << nlc));
      int32_t *nfv = (int32_t *)(nb + ((size_t)40 << nlc));
      for (int i0 = 0; i0 < (1 << lc); i0 += 64) {
        const SRec rc = tb[i0 + lane];
        const bool mine = (rc.tag >> 1) == gen;
        const uint32_t p0 = (((uint32_t)rc.key * 2654435761u) >> (33 - nlc)) << 1;
        const uint32_t p = claim(nt, nlc, mine, (int32_t)rc.key, p0, false);
        if (mine) { nt[p].g = rc.g; nt[p].h = rc.h; nt[p].parent = rc.parent; nt[p].parent_edge = rc.parent_edge; nt[p].tag = rc.tag; }
      }
      for (int i = lane; i < n_far; i += 64) { nff[i] = ff[i]; nfv[i] = fv[i]; }
      __syncthreads();
      if (cls > 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");     // (see sr_pool_release)
        if (lane == 0) sr_pool_release(a, cls, chunk);
      }
      cls = nc; chunk = idx; lc = nlc; tb = nt; ff = nff; fv = nfv;
      grows++;
      return true;
    };

    double T = inf;
    double slack = 0.0;                                         // how far above the minimum a step's other vertices may lie (follows the lanes' use)
    int result = SR_NO_PATH;
    bool over_budget = false;
    unsigned long long exp_q = 0;
    unsigned moves = 0;                                         // times the threshold moved (near full / near empty)
    const double h0 = sr_u(heuristic(start));
    slack = 0.01 * h0;
    if (h0 != inf) {
      {
        SRec none;
        uint32_t p0 = 0;
        (void)lookup(start, lane == 0, none, p0);
        if (lane == 0) { tb[p0] = SRec{0.0, h0, start, -1, (uint32_t)start, gen << 1}; nf[0] = h0; nv[0] = start | ((int32_t)a.deg[start] << SR_VBITS); }
      }
      n_near = 1; count = 1;
      __syncthreads();
#ifdef TRK_SEARCH_CLOCKS
      unsigned long long clk[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = wall_clock64(), n_steps = 0, n_passes = 0;
#endif
      for (;;) {
        SR_CLK(4);
#ifdef TRK_SEARCH_CLOCKS
        n_steps++;
#endif
        if ((int64_t)exp_q >= a.max_pops) { result = SR_FALLBACK; over_budget = true; break; }
        if (n_near == n_dead) {                                // (entries taken off the list stay as dead slots until the list is next rewritten)
          n_near = 0; n_dead = 0;
          if (n_far == 0) break;                               // the open list is empty: no path
          // ---- refill: raise T so that about half a list's worth of far's entries come in ----
          double mn = inf, mx = -inf;
          for (int i = lane; i < n_far; i += 64) { const double f = ff[i]; mn = f < mn ? f : mn; mx = f > mx ? f : mx; }
          mn = sr_wave_min(mn); mx = sr_wave_max(mx);
          double Tn = inf;
          if (n_far > SR_CAP / 2) {
            Tn = mn + (mx - mn) * ((double)(SR_CAP / 2) / (double)n_far);
            if (!(Tn > mn)) Tn = mn + (mx - mn) * 0.5;
            if (!(Tn > mn)) Tn = inf;                          // every key the same: they all qualify (and must fit: checked below)
            for (int tries = 0; tries < 64; tries++) {
              int c = 0;
              for (int i = lane; i < n_far; i += 64) c += ff[i] < Tn ? 1 : 0;
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
          for (int c0 = 0; c0 < n_far; c0 += 64) {
            const int i = c0 + lane;
            const bool on = i < n_far;
            const double f = on ? ff[i] : 0.0;
            const int32_t v = on ? fv[i] : 0;
            const bool in = on && f < Tn, stay = on && !in;
            const unsigned long long mi = __ballot(in), ms = __ballot(stay);
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
        // ---- select: every lane the smallest key of its share of near (entries lane, lane + 64, ...; four reads in flight, compared
        // in index order), then up to kbest rounds of a wave minimum over the lanes not taken yet: the first is near's minimum ----
        double best = inf;
        int bi = -1;
        for (int i0 = 0; i0 < n_near; i0 += 256) {
          double f4[4];
#pragma unroll
          for (int j = 0; j < 4; j++) { const int i = i0 + 64 * j + lane; f4[j] = i < n_near ? nf[i] : inf; }
#pragma unroll
          for (int j = 0; j < 4; j++) { const int i = i0 + 64 * j + lane; if (f4[j] < best) { best = f4[j]; bi = i; } }      // (a dead slot holds +inf)
        }
        // A step takes the minimum and, with it, every other lane's candidate within `slack` of it -- expanding a vertex that is not
        // the minimum is what any best-first search with re-opening may do -- as many as their arcs fill two passes of the wave's 64
        // lanes.  All of it in lane-parallel steps: no loop over the taken vertices (round 5's first version ran one selection round
        // per vertex: 35 instructions each).  The goal is only ever taken as the minimum.
        const bool has_c = bi >= 0;
        const int bw = has_c ? nv[bi] : 0;                      // the lane's candidate word: vertex | lanes its row needs
        const int bu = bw & ((1 << SR_VBITS) - 1), bd = (int)((unsigned)bw >> SR_VBITS);
        const unsigned long long bbits = (unsigned long long)__double_as_longlong(best);   // (keys are >= 0: ordered as integers)
        const uint32_t bhi = (uint32_t)(bbits >> 32), blo = (uint32_t)bbits;
        const uint32_t mh = sr_wave_min_u32(has_c ? bhi : 0xffffffffu);
        const bool top = has_c && bhi == mh;
        const uint32_t ml = sr_wave_min_u32(top ? blo : 0xffffffffu);
        const int src0 = __ffsll((long long)__ballot(top && blo == ml)) - 1;     // (the list holds a live entry: n_near > n_dead)
        const double fmin = sr_lane(best, src0);
        const int d0 = __builtin_amdgcn_readlane(bd, src0);
        // (a search widens its steps as it goes -- one more vertex per eight expansions so far -- so that a search of a few dozen
        // expansions does not spend most of them beside the path: on a 2 500-vertex roadmap full-width steps from the start expanded
        // four times what the host does; a search of a thousand is at full width after its first tenth)
        const int kcap = (int)(exp_q >> 3) + 1 < kbest ? (int)(exp_q >> 3) + 1 : kbest;
        bool cnd = has_c && (lane == src0 || (kcap > 1 && best <= fmin + slack && bu != goal));
        {
          const int rko = __popcll(__ballot(cnd) & below) - (lane > src0 ? 1 : 0);       // rank among the others
          cnd = cnd && (lane == src0 || rko < kcap - 1);
        }
        // lanes: the minimum's arcs first, the others' behind them in lane order; no group straddles the two passes
        const int x_ = (cnd && lane != src0) ? bd : 0;
        const int incl = sr_wave_scan_add(x_);
        int cstart = lane == src0 ? 0 : d0 + incl - x_;
        {
          const unsigned long long mstr = __ballot(cnd && cstart < 64 && cstart + bd > 64);
          if (mstr) {
            const int ss = __builtin_amdgcn_readlane(cstart, __ffsll((long long)mstr) - 1);
            if (cstart >= ss) cstart += 64 - ss;
          }
        }
        const unsigned long long mc_all = __ballot(cnd);
        cnd = cnd && cstart + bd <= 128;
        const unsigned long long mc = __ballot(cnd);
        // (the slack follows the lanes: widened while a step leaves a quarter of them idle, narrowed when candidates had to stay)
        if (mc != mc_all) slack *= 0.75;
        else if (d0 + __builtin_amdgcn_readlane(incl, 63) < 96 && __popcll(mc) < kcap) slack = slack * 1.4 + 1e-6 * fmin;
        own8[lane] = 0; own8[64 + lane] = 0;
        __syncthreads();
        // the entries leave the list: their slots are marked dead (+inf) where they are -- no entry moves -- and are dropped when the
        // list is next rewritten (near full: see the append)
        if (cnd) { own8[cstart] = (uint8_t)(lane + 1); nf[bi] = inf; }
        n_dead += __popcll(mc);
        __syncthreads();
        SR_CLK(1);
        // ---- the popped vertices' records and first adjacency rows, for both passes at once: a group of lanes per vertex (as many as
        // its row has arcs) finds its owner by a running maximum over the markers; one round trip; closed ones (stale entries) drop out ----
        int u_[2], sub_[2], gb_[2];
        bool on_[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const uint32_t mk = own8[64 * q + lane];
          const uint32_t v_ = sr_wave_scan_max(mk ? (((uint32_t)lane << 8) | mk) : 0u);
          const int s_ = (int)(v_ & 0xffu) - 1;
          const int cs = __shfl(cstart, s_ >= 0 ? s_ : 0, 64), w_ = __shfl(bw, s_ >= 0 ? s_ : 0, 64);
          const int t_ = 64 * q + lane;
          on_[q] = s_ >= 0 && t_ < cs + (int)((unsigned)w_ >> SR_VBITS);
          u_[q] = on_[q] ? (w_ & ((1 << SR_VBITS) - 1)) : -1;
          sub_[q] = t_ - cs; gb_[q] = cs - 64 * q;
        }
        SArc arc_[2];
        SRec urec_[2];
        uint32_t pu_[2];
        bool live_[2];
        {
          SRec r0_[2], r1_[2];
#pragma unroll
          for (int q = 0; q < 2; q++) {
            arc_[q] = on_[q] ? a.rows[(int64_t)u_[q] * SR_D + sub_[q]] : no_arc;
            pu_[q] = slot0(on_[q] ? u_[q] : 0);
            if (on_[q]) { r0_[q] = tb[pu_[q]]; r1_[q] = tb[pu_[q] + 1]; }
          }
#pragma unroll
          for (int q = 0; q < 2; q++) {
            urec_[q] = SRec{0.0, 0.0, -1, -1, 0u, 1u};
            live_[q] = on_[q] && resolve(u_[q], r0_[q], r1_[q], urec_[q], pu_[q]) && (urec_[q].tag & 1u) == 0u;
          }
        }
#pragma unroll
        for (int q = 0; q < 2; q++) if (live_[q] && sub_[q] == 0) tb[pu_[q]].tag = (gen << 1) | 1u;
        const unsigned long long mlive0 = __ballot(live_[0]);
        const int n_live = __popcll(__ballot(live_[0] && sub_[0] == 0)) + __popcll(__ballot(live_[1] && sub_[1] == 0));
        exp_q += (unsigned long long)n_live;
        if ((mlive0 & 1ull) && __builtin_amdgcn_readfirstlane(u_[0]) == goal) { result = SR_FOUND; break; }   // (lane 0 works for the minimum)
        // the heuristics of BOTH passes' neighbours now (read-only rows: nothing a pass writes can change them), so that the second
        // pass's rows travel while the first pass works; the neighbours' RECORDS are fetched pass by pass, after the previous pass's writes
        double hv_[2];
#pragma unroll
        for (int q = 0; q < 2; q++) hv_[q] = (live_[q] && arc_[q].v >= 0) ? heuristic(arc_[q].v & ((1 << SR_VBITS) - 1)) : 0.0;
        SR_CLK(2);
        bool failed = false;
        for (int q = 0; q < 2 && !failed; q++) {
        SArc arc = q ? arc_[1] : arc_[0];
        bool act = q ? live_[1] : live_[0];
        if (!__ballot(act)) continue;
        const int my_u = q ? u_[1] : u_[0], sub = q ? sub_[1] : sub_[0], gbase = q ? gb_[1] : gb_[0];
        const double ug = q ? urec_[1].g : urec_[0].g;
        for (int pass = 0; pass < (1 << 20) && !failed; pass++) {
          // a step may add a record per lane: the table moves before it could pass three quarters (no lane holds a position here)
          if (count + 64 > (3 << lc) / 4 && !grow()) { failed = true; break; }
#ifdef TRK_SEARCH_CLOCKS
          n_passes++;
#endif
          const bool has = act && arc.v >= 0;
          const bool more = act && arc.v == SR_ARC_MORE;
          const int32_t av = arc.v & ((1 << SR_VBITS) - 1);           // the neighbour; above it: the lanes ITS row needs (an open-list word's top bits)
          bool cand = false, push = false, fresh = false;
          double fp = 0.0, gv = 0.0, hh = 0.0;
          int32_t vp = 0, pe = -1;
          uint32_t pv = 0, vd = 0;
          if (has) {
            // everything the relaxation can need is requested at once, whether or not it turns out to be needed: validity bytes,
            // the neighbour's record (the line its probe starts at), and the rows of its heuristic
            pv = slot0(av);
            const SRec r0 = tb[pv], r1 = tb[pv + 1];
            const uint8_t es = a.estat[arc.e], vs = a.vstat[av];
            vd = (uint32_t)arc.v >> SR_VBITS;
            const double hv = pass == 0 ? (q ? hv_[1] : hv_[0]) : heuristic(av);     // (a continued row's neighbours: here)
            SRec nn = SRec{0.0, 0.0, -1, -1, 0u, 0u};
            const bool seen = resolve(av, r0, r1, nn, pv);
            if (es != SR_INVALID && vs != SR_INVALID) {
              gv = ug + arc.w;
              if (!seen || gv < nn.g) {
                hh = seen ? nn.h : hv;                           // h(v) is fixed for the query: computed when v is first reached
                cand = true; fresh = !seen; vp = av; pe = arc.e;
              }
            }
          }
          // ---- two lanes with the same neighbour: the smaller cost wins (the lower lane among equals), the others stand down ----
          SR_CLK(5);
          // (first the cheap question "do any two of them name the same neighbour at all?": every candidate lane leaves its number in a
          // byte slot of its neighbour's hash -- 1 024 slots in the cost table's place -- and looks whether it is still there; two lanes
          // with one neighbour share the slot, so one of them finds the other's number.  Four times in five nobody does, and the
          // protocol below -- four barriers a round -- is skipped.)
          bool contested = false;
          if (n_live > 1) {
            uint8_t *quick = (uint8_t *)tab_key;
            const unsigned hq = ((unsigned)vp * 2654435761u) >> 22;
            if (cand) quick[hq] = (uint8_t)lane;
            __syncthreads();
            contested = __ballot(cand && quick[hq] != (uint8_t)lane) != 0ull;
            __syncthreads();
          }
          if (contested) {
            bool open = cand;
            const unsigned long long key = (unsigned long long)__double_as_longlong(gv);      // (costs are >= 0: ordered as integers)
            const unsigned hs = ((unsigned)vp * 2654435761u) >> 25;
            for (int round = 0; round < 16; round++) {
              if (!__ballot(open)) break;
              if (open) { tab_owner[hs] = (uint32_t)lane; tab_key[hs] = ~0ull; tab_low[hs] = 64u; }
              __syncthreads();
              const int w = open ? (int)tab_owner[hs] : lane;
              const int32_t vw = __shfl(vp, w, 64);
              const bool same = open && vw == vp;                // the slot is this vertex's for the round (another vertex's lanes wait)
              if (same) atomicMin(&tab_key[hs], key);
              __syncthreads();
              const bool least = same && tab_key[hs] == key;
              if (least) atomicMin(&tab_low[hs], (uint32_t)lane);
              __syncthreads();
              if (same) { open = false; if (!(least && tab_low[hs] == (uint32_t)lane)) cand = false; }
              __syncthreads();
            }
            if (__ballot(open)) { failed = true; break; }
          }
          SR_CLK(6);
          // ---- the winners write: a vertex reached before in place, a new one into a free record of its probe path ----
          {
            const unsigned long long mnew = __ballot(cand && fresh);
            if (mnew) pv = claim(tb, lc, cand && fresh, vp, pv, true);
            count += __popcll(mnew);
            if (cand) {
              SRec *rp = tb + pv;
              rp->g = gv; rp->h = hh; rp->parent = my_u; rp->parent_edge = pe; rp->key = (uint32_t)vp;
              rp->tag = (gen << 1) | (hh == inf ? 1u : 0u);
              if (hh != inf) { push = true; fp = gv + hh; }
            }
          }
          // ---- append: keys below T to near, the others to far ----
          unsigned long long mn_ = __ballot(push && fp < T);
          SR_CLK(7);
          while (n_near + __popcll(mn_) > SR_CAP) {
            // near is full.  With dead slots in it: they are squeezed out, nothing else changes.  Without: T drops halfway towards
            // its smallest key, what lies above moves to far
            const bool spill = n_dead == 0;
            double Tn = inf;
            if (spill) {
              double lo = inf, hi = -inf;
              for (int i = lane; i < n_near; i += 64) { const double f = nf[i]; lo = f < lo ? f : lo; hi = f > hi ? f : hi; }
              lo = sr_wave_min(lo); hi = sr_wave_max(hi);
              const double top = T < inf ? T : hi;
              Tn = lo + (top - lo) * 0.5;
              if (!(Tn > lo) || !(Tn < top)) { failed = true; break; }
              if (n_far + n_near > (1 << lc) && !grow()) { failed = true; break; }
            }
            int keep = 0;
            for (int c0 = 0; c0 < n_near; c0 += 64) {
              const int i = c0 + lane;
              const bool on = i < n_near;
              const double f = on ? nf[i] : inf;
              const int32_t v = on ? nv[i] : 0;
              __syncthreads();
              const bool stay = f < Tn, out = !stay && f < inf;
              const unsigned long long ms = __ballot(stay), mo = __ballot(out);
              if (stay) { const int p = keep + __popcll(ms & below); nf[p] = f; nv[p] = v; }
              if (out) { const int p = n_far + __popcll(mo & below); ff[p] = f; fv[p] = v; }
              keep += __popcll(ms); n_far += __popcll(mo);
              __syncthreads();
            }
            n_near = keep; n_dead = 0;
            if (spill) {
              T = Tn;
              moves++;
              mn_ = __ballot(push && fp < T);
            }
          }
          if (failed) break;
          const unsigned long long mf_ = __ballot(push && !(fp < T));
          if (n_far + __popcll(mf_) > (1 << lc) && !grow()) { failed = true; break; }
          const int32_t wp = vp | (int32_t)(vd << SR_VBITS);
          if (push && fp < T) { const int p = n_near + __popcll(mn_ & below); nf[p] = fp; nv[p] = wp; }
          else if (push) { const int p = n_far + __popcll(mf_ & below); ff[p] = fp; fv[p] = wp; }
          n_near += __popcll(mn_); n_far += __popcll(mf_);
          __syncthreads();
          // ---- a vertex with more arcs than a row holds: its group goes on to the row its last slot names ----
          if (!__ballot(more)) break;
          const int nrow = __shfl(more ? arc.e : -1, gbase + (SR_D - 1), 64);
          act = act && nrow >= 0;
          arc = act ? a.rows[(int64_t)nrow * SR_D + sub] : no_arc;
        }
        }
        if (failed) { result = SR_FALLBACK; break; }
      }
#ifdef TRK_SEARCH_CLOCKS
      if (lane == 0) {
        clk[3] = clk[5] + clk[6] + clk[7];                         // "arcs + rows + relax" = loads / heuristic / lookup + conflicts + writes
        for (int i = 0; i < 6; i++) atomicAdd((unsigned long long *)(a.next + 16) + i, clk[i]);
        atomicMax((unsigned long long *)(a.next + 16) + 6, clk[0] + clk[1] + clk[2] + clk[3] + clk[4]);
        for (int i = 5; i < 8; i++) atomicAdd((unsigned long long *)(a.next + 124 - 2 * 5) + i, clk[i]);   // words 124 .. 129
        // when searches end, in 2 ms buckets since the first wave's start (words 40 .. 41: that start; 44 .. : 40 buckets of counts,
        // 84 .. : 40 buckets of the expansions of the searches that ended there)
        atomicAdd((unsigned long long *)(a.next + 32), n_steps);
        atomicAdd((unsigned long long *)(a.next + 32) + 1, n_passes);
      }
#endif
    }
    // ---- the path, goal ... start, and its edges, staged in the far list's place (the open list is dead) ----
    int nvert = 0;
    const int path_max = 1 << (lc - 1);
    int32_t *stage = fv;
    __syncthreads();
    if (result == SR_FOUND) {
      if (lane == 0) {
        int32_t v = goal;
        for (;;) {
          if (nvert >= path_max) { nvert = -1; break; }
          stage[nvert] = v;
          if (v == start) { nvert++; break; }
          SRec nd = SRec{0.0, 0.0, -1, -1, 0u, 0u};
          uint32_t p = 0;
          if (!lookup(v, true, nd, p)) { nvert = -1; break; }
          stage[path_max + nvert] = nd.parent_edge;
          v = nd.parent;
          nvert++;
        }
      }
      nvert = __builtin_amdgcn_readfirstlane(nvert);
      if (nvert <= 0) result = SR_FALLBACK;
    }
    if (result == SR_FOUND) {
      uint32_t off = atomicAdd(a.pbuf_used, lane == 0 ? (uint32_t)(2 * nvert - 1) : 0u);
      off = (uint32_t)__builtin_amdgcn_readlane((int)off, 0);
      const uint32_t need = (uint32_t)(2 * nvert - 1);
      if ((uint64_t)off + need > a.pbuf_cap) result = SR_FALLBACK;
      else {
        __syncthreads();
        for (int i = lane; i < nvert; i += 64) a.pbuf[off + i] = stage[i];
        for (int i = lane; i < nvert - 1; i += 64) a.pbuf[off + nvert + i] = stage[path_max + i];
        if (lane == 0) { a.poff[qi] = (int32_t)off; a.plen[qi] = nvert; }
      }
    }
    __syncthreads();
    if (cls > 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) {
      a.found[qi] = (uint8_t)result;
      // (1: over its budget of expansions -- the host's budget rule counts these --, 3: no table, list or path buffer left for it)
      if (result == SR_FALLBACK && a.handback) __hip_atomic_store(a.handback + qi, over_budget ? 1u : 3u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      atomicAdd(a.expanded, exp_q);
      if (moves) atomicAdd(a.next + 4, moves);
      if (grows) atomicAdd(a.next + 5, grows);
      atomicMax(a.next + 6, (uint32_t)count);
      if (cls > 0) sr_pool_release(a, cls, chunk);
#ifdef TRK_SEARCH_CLOCKS
      {
        const unsigned long long t0_ = __hip_atomic_load((unsigned long long *)(a.next + 40), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned b_ = (unsigned)((wall_clock64() - t0_) / 200000ull);
        if (b_ > 39u) b_ = 39u;
        atomicAdd(a.next + 44 + b_, 1u);
        atomicAdd(a.next + 84 + b_, (uint32_t)exp_q);
      }
#endif
    }
  }
}

}  // namespace trk
