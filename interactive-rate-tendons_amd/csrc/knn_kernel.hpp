// knn_kernel.hpp -- K6 `knn_bruteforce`: exact k nearest neighbours of every roadmap vertex among all
// vertices, in the reference's state-space metric -- what connectionStrategy_(v) (KBoundedStrategy /
// KStarStrategy over the GNAT nn_, motion-planning/VoxelCachedLazyPRM.cpp:1339,1352,1491-1502) returns
// for each v.  Metric = OMPL CompoundStateSpace::distance as wired by Problem.cpp:101-163:
//   |d tau|_2  +  (extent / 4 pi) * arc(d theta)  +  (2 extent / L) * |d s_start|.
// One lane per query vertex; the candidate index is wave-uniform, so candidates arrive through scalar
// loads and feed the fp64 FMAs as SGPR operands; each lane keeps its k best in an LDS column (sorted
// insertion; insertions become rare once the list has warmed up).  Exact, and no tree: the states are ordered by the cells
// of a uniform 2-D grid over two key coordinates (one rocPRIM radix pass on the cell ids, cache_merge.hip:
// sort_states_by_cells) and a wave -- 64 queries that are neighbours in that order, i.e. of one or two cells -- only visits
// the cells its queries' search radii reach in BOTH keys (every term of the metric is non-negative, so |d key| <= distance on
// each): the radius is the SEED, the k-th distance to the candidates nearest in sorted order (a first, short pass).
// Round 2 windowed one sorted coordinate (~1/6 of the candidates at 10^5 uniform 4-D states, ~1/12 at 10^6); the second key
// squares that.
// Like nearestK on a structure that already holds v, the result includes v itself (distance 0), which connectVertices
// then skips (:2848).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "tr_types.hpp"
#include "cache_merge.hpp"

namespace trk {

struct KnnMetric {
  int32_t n_tension, has_rot, has_ret, S;
  double w_rot, w_ret;
};

constexpr int KNN_SMAX = TRK_MAX_TENDONS + 2;


// One wave = 64 queries x one slice [j0, j1) of the candidates.  A roadmap of 10^5 vertices is only ~1 600 query waves --
// fewer than two per SIMD, so every scalar candidate load would be paid in full; slicing the candidate range over
// blockIdx.y puts 8+ waves on every SIMD (r02 profile: 17 % VALU issue utilisation, 57 % of wave cycles waiting, before).
// Each slice keeps its k best per query in LDS, ordered by (distance, index); knn_merge then merges the slices' lists.
// Ordering is by the DISTANCE as CompoundStateSpace::distance returns it -- for a tension-only space sqrt(sum d^2), so two
// candidates whose squared distances differ but whose square roots round to the same double tie and stay in index order,
// exactly like a stable sort of the distances.  The square root is only taken for candidates that pass a conservative
// test on the squared distance (rare once the list has warmed up).
// NT tension dimensions and the presence of the rotation / retraction coordinates are compile-time: a chunk is then
// straight-line code and its scalar loads are issued back to back.
// cand / perm: the states in cell order and the original index of each; cg / cellstart: the grid and the first sorted position of
// every cell.  qlist (optional):
// sorted positions of the queries (ascending; null = every state is a query), nq of them; row_first: original index of
// the first output row (tr_knn_range).  half_window > 0 marks the seeding pass: candidates = the half_window sorted
// neighbours either side of the wave's queries, only the k-th distance is written (seed_out, indexed like qlist).
// KCAP > 0: the list lives in REGISTERS (k <= KCAP entries; the rest are inert), an insertion is straight-line selects with no
// LDS round trip; KCAP = 0: in the lane's LDS column, any k.
template <int NT, bool ROT, bool RET, int KCAP = 0>
__global__ __launch_bounds__(64) void knn_bruteforce(const double *__restrict__ cand, KnnCells cg, const int32_t *__restrict__ cellstart,
                                                     const int32_t *__restrict__ perm, const int32_t *__restrict__ qlist, int64_t nq,
                                                     int64_t n_cand, KnnMetric m, int k, double max_dist, int64_t half_window,
                                                     const double *__restrict__ seed, double *__restrict__ seed_out,
                                                     int64_t row_first, int32_t *__restrict__ out_idx, double *__restrict__ out_dist) {
#pragma clang fp contract(off)
  extern __shared__ unsigned char knn_lds[];
  double *bd = reinterpret_cast<double *>(knn_lds) + threadIdx.x;               // [k][64]
  int32_t *bi = reinterpret_cast<int32_t *>(knn_lds + (size_t)k * 64 * 8) + threadIdx.x;
  const int64_t q = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const bool live = q < nq;
  const int64_t qc = live ? q : nq - 1;
  const int64_t js = qlist ? (int64_t)qlist[qc] : qc;                           // the query's position in sorted order
  const int S = m.S;
  double x[KNN_SMAX];
#pragma unroll
  for (int d = 0; d < KNN_SMAX; d++) x[d] = d < S ? cand[js * S + d] : 0.0;
  constexpr int RC = KCAP > 0 ? KCAP : 1;
  double rd[RC];
  int32_t ri[RC];
  if constexpr (KCAP > 0) {
    // entries beyond k hold (-inf, 0): never the largest, never replaced
#pragma unroll
    for (int e = 0; e < KCAP; e++) { rd[e] = e < k ? 1.0 / 0.0 : -1.0 / 0.0; ri[e] = e < k ? -1 : 0; }
  } else {
    for (int p = 0; p < k; p++) { bd[p * 64] = 1.0 / 0.0; bi[p * 64] = -1; }
  }
  // `worst`: a candidate enters the list when it sorts before this by (distance, index) -- the lane's current k-th entry,
  // or, while the list is not full, the SEED (any index): the k nearest of all candidates are at most that far, so a slice
  // starts by accepting exactly the candidates with distance <= seed instead of filling its list with whatever comes first
  // and shifting it ~k ln(slice / k) times (those insertions, not the distances, were 80 % of the kernel's time).
  constexpr int SS_KEY = NT + (ROT ? 1 : 0);                 // position of the retraction coordinate, when there is one
  double worst = 1.0 / 0.0;
  if (seed) worst = seed[qc];                                  // (an optional upper bound of the k-th distance; the expanding search needs none)
  int32_t worst_i = -1;                                        // original index of the list's largest entry (-1: the list is not full yet)
  int wpos = 0;                                                // ... and its position: the entry the next accepted candidate replaces
  // ---- the wave's candidates: an EXPANDING search over the cells of the 2-D grid ----
  // The states are ordered by the cells of a uniform grid over two key coordinates, each a TERM of the metric -- the first two
  // tensions, or the weighted retraction and the first tension -- so |key difference| <= distance on either: once a lane holds k
  // candidates, nothing outside [key - worst, key + worst] in either key can improve its list.  The wave scans the cells its own
  // queries lie in, then keeps growing that box of cells by one ring towards the window its lanes' CURRENT thresholds still ask
  // for, until the box covers the window (thresholds only shrink, so the window only shrinks).  Near candidates come first:
  // the thresholds are almost final after the first ring and few later candidates enter a list -- list insertions, not
  // distances, were what the one-coordinate window of round 2 spent most of its time on (its seeding pass started from empty
  // lists: 30 of 80 ms at 6 x 10^5 states).  r is widened by a relative 1e-12: sqrt(fl(t0^2 + ...)) may round an ulp below
  // |t0|.  Cells are a monotone function of the key (cache_merge.hpp: knn_cell_of), so the cells of an interval's end points
  // bracket the cells of everything inside it.
  constexpr int K1 = RET ? 0 : (NT >= 2 ? 1 : -1);            // the second key's coordinate (none for a single tension)
  const double k0 = RET ? m.w_ret * x[SS_KEY] : x[0];         // the products sort_states_by_cells formed (cg.scale x coordinate)
  const double k1 = K1 >= 0 ? x[K1 >= 0 ? K1 : 0] : 0.0;
  auto wave_min = [](double v) { for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64)); return v; };
  auto wave_max = [](double v) { for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64)); return v; };
  int64_t j1 = 0;                                             // end of the run being scanned (do_chunk masks a tail chunk against it)
  double gate2 = worst * worst * (1.0 + 4.5e-16);   // plain metric: squared distances above this cannot reach `worst`
  constexpr bool plain = !ROT && !RET;
  // Candidates are taken a chunk at a time: their scalar loads are issued together and the wave leaves the chunk at once
  // unless some lane can improve its list -- rare after the first few hundred candidates.  (One candidate per iteration
  // exposed a full scalar-load round trip each time: 300 cycles per candidate against ~40 of arithmetic.)
  constexpr int SS = NT + (ROT ? 1 : 0) + (RET ? 1 : 0);            // doubles per candidate
  constexpr int CH = SS <= 4 ? 8 : 4;                               // <= 64 SGPRs of candidate data in flight
  // FULL chunks address their candidates at compile-time offsets from one pointer (S == SS for this instantiation) and
  // need no tail masking: ~12 scalar instructions per chunk instead of ~110 (the scalar unit is shared by the CU's four
  // SIMDs, and 64-bit index arithmetic per candidate cost as many issue cycles as the distances themselves).
  auto do_chunk = [&](auto full_tag, int64_t jb) {
    constexpr bool FULL = decltype(full_tag)::value;
    double cc[CH][SS];
#pragma unroll
    for (int u = 0; u < CH; u++) {
      const int64_t j = (FULL || jb + u < j1) ? jb + u : j1 - 1;  // wave-uniform; a tail chunk repeats the last candidate, masked below
      const double *__restrict__ c = FULL ? cand + jb * SS + u * SS : cand + j * SS;
#pragma unroll
      for (int d = 0; d < SS; d++) cc[u][d] = c[d];
    }
    __builtin_amdgcn_sched_barrier(0);                              // all of the chunk's scalar loads are issued before any of its arithmetic
    double dd[CH];                                  // plain metric: squared distance; otherwise the distance itself
#pragma unroll
    for (int u = 0; u < CH; u++) {
      double s2 = 0.0;
#pragma unroll
      for (int d = 0; d < NT; d++) { const double t = x[d] - cc[u][d]; s2 += t * t; }
      double dist = s2;
      if constexpr (!plain) {
        dist = sqrt(s2);
        if constexpr (ROT) {                                  // SO2StateSpace::distance
          double a = fabs(x[NT] - cc[u][NT]);
          a = (a > 3.14159265358979323846) ? 2.0 * 3.14159265358979323846 - a : a;
          dist += m.w_rot * a;
        }
        if constexpr (RET) {
          const double t = x[SS - 1] - cc[u][SS - 1];
          dist += m.w_ret * sqrt(t * t);
        }
      }
      dd[u] = (FULL || jb + u < j1) ? dist : 1.0 / 0.0;
    }
    double mn = dd[0];
#pragma unroll
    for (int u = 1; u < CH; u++) mn = fmin(mn, dd[u]);
    if (!__any(plain ? mn <= gate2 : mn <= worst)) return;
#pragma unroll
    for (int u = 0; u < CH; u++) {
      if (!FULL && !(jb + u < j1)) continue;         // a tail chunk's padding
      double dist = dd[u];
      if (plain) {
        if (!(dist <= gate2)) continue;        // (<=: a zero threshold -- coincident states -- still admits distance 0)
        dist = sqrt(dist);
      }
      if (dist <= worst) {
        // The list is kept UNSORTED with its largest entry tracked (value, original index, position): a candidate that sorts
        // before that entry by (distance, original index) replaces it -- one LDS write pair at the lane's own position -- and
        // the new largest entry is found by one pass over the k entries.  (Round 2 kept the list sorted and shifted it on
        // every insertion: 4 k LDS operations and ~10 k VALU instructions per event, and with 64 lanes sharing the branch a
        // wave runs an event whenever ANY lane inserts; the search was spending most of its time there: 9.5 ms at k = 2
        // against 34 ms at k = 11 and 285 ms at k = 41 for 6 x 10^5 states.)  The list is sorted once, at the end.
        // Equal distances are ordered by the original index -- the order of a stable sort of the query's distance row; empty
        // entries (index < 0) sort last.
        const int32_t cj = perm[jb + u];
        if (dist < worst || worst_i < 0 || cj < worst_i) {
          double md;
          int32_t mi;
          int mp = 0;
          if constexpr (KCAP > 0) {
#pragma unroll
            for (int e = 0; e < KCAP; e++) { const bool hit = e == wpos; rd[e] = hit ? dist : rd[e]; ri[e] = hit ? cj : ri[e]; }
            md = rd[0]; mi = ri[0];
#pragma unroll
            for (int e = 1; e < KCAP; e++) {
              const double d = rd[e];
              const int32_t i = ri[e];
              const bool later = mi >= 0 && (i < 0 || d > md || (d == md && i > mi));     // entry e sorts after the largest so far
              md = later ? d : md; mi = later ? i : mi; mp = later ? e : mp;
            }
          } else {
            bd[wpos * 64] = dist; bi[wpos * 64] = cj;
            md = bd[0]; mi = bi[0];
            for (int e = 1; e < k; e++) {
              const double d = bd[e * 64];
              const int32_t i = bi[e * 64];
              const bool later = mi >= 0 && (i < 0 || d > md || (d == md && i > mi));
              md = later ? d : md; mi = later ? i : mi; mp = later ? e : mp;
            }
          }
          worst_i = mi; wpos = mp;
          if (mi >= 0) worst = md;                    // (an unfilled list keeps its threshold: infinity, or the seed)
        }
        // sqrt(s2) <= worst needs s2 <= worst^2 (1 + 2^-51): beyond that the correctly rounded root is > worst
        gate2 = worst * worst * (1.0 + 4.5e-16);
      }
    }
  };
  // a run [a, b) of sorted positions
  auto scan = [&](int64_t a, int64_t b) {
    int64_t jb = a;
    j1 = b;
    for (; jb + CH <= j1; jb += CH) do_chunk(std::true_type{}, jb);
    if (jb < j1) do_chunk(std::false_type{}, jb);
  };
  // rows [y0, y1] of column cx: consecutive in sorted order
  auto scan_cells = [&](int cx, int y0, int y1) {
    if (y0 <= y1) scan((int64_t)cellstart[cx * cg.B + y0], (int64_t)cellstart[cx * cg.B + y1 + 1]);
  };
  {
    // the box of cells the wave's own queries lie in (NaN keys: cell 0)
    const double q0lo = wave_min(k0), q0hi = wave_max(k0), q1lo = wave_min(k1), q1hi = wave_max(k1);
    int sx0 = __builtin_amdgcn_readfirstlane(knn_cell_of(q0lo, cg.lo0, cg.inv0, cg.C)), sx1 = __builtin_amdgcn_readfirstlane(knn_cell_of(q0hi, cg.lo0, cg.inv0, cg.C));
    int sy0 = __builtin_amdgcn_readfirstlane(knn_cell_of(q1lo, cg.lo1, cg.inv1, cg.B)), sy1 = __builtin_amdgcn_readfirstlane(knn_cell_of(q1hi, cg.lo1, cg.inv1, cg.B));
    if (sx1 < sx0) sx1 = sx0;
    if (sy1 < sy0) sy1 = sy0;
    for (int cx = sx0; cx <= sx1; cx++) scan_cells(cx, sy0, sy1);
    for (;;) {
      // the window the lanes' thresholds still ask for (an unfilled list: everything)
      double r = worst < max_dist ? worst : max_dist;
      r = r + r * 1e-12;
      const int nx0 = __builtin_amdgcn_readfirstlane(knn_cell_of(wave_min(k0 - r), cg.lo0, cg.inv0, cg.C));
      const int nx1 = __builtin_amdgcn_readfirstlane(knn_cell_of(wave_max(k0 + r), cg.lo0, cg.inv0, cg.C));
      const int ny0 = __builtin_amdgcn_readfirstlane(knn_cell_of(wave_min(k1 - r), cg.lo1, cg.inv1, cg.B));
      const int ny1 = __builtin_amdgcn_readfirstlane(knn_cell_of(wave_max(k1 + r), cg.lo1, cg.inv1, cg.B));
      if (nx0 >= sx0 && nx1 <= sx1 && ny0 >= sy0 && ny1 <= sy1) break;
      // one ring towards it: new rows of the old columns first, then the new columns over all rows of the grown box
      const int tx0 = nx0 < sx0 ? sx0 - 1 : sx0, tx1 = nx1 > sx1 ? sx1 + 1 : sx1;
      const int ty0 = ny0 < sy0 ? sy0 - 1 : sy0, ty1 = ny1 > sy1 ? sy1 + 1 : sy1;
      for (int cx = sx0; cx <= sx1; cx++) { scan_cells(cx, ty0, sy0 - 1); scan_cells(cx, sy1 + 1, ty1); }
      if (tx0 < sx0) scan_cells(tx0, ty0, ty1);
      if (tx1 > sx1) scan_cells(tx1, ty0, ty1);
      sx0 = tx0; sx1 = tx1; sy0 = ty0; sy1 = ty1;
    }
  }
  if constexpr (KCAP > 0) {
#pragma unroll
    for (int e = 0; e < KCAP; e++) if (e < k) { bd[e * 64] = rd[e]; bi[e * 64] = ri[e]; }
  }
  // order the list by (distance, original index), empty entries last: a selection sort over the lane's LDS column, once
  for (int p = 0; p + 1 < k; p++) {
    double md = bd[p * 64];
    int32_t mi = bi[p * 64];
    int mp = p;
    for (int e = p + 1; e < k; e++) {
      const double d = bd[e * 64];
      const int32_t i = bi[e * 64];
      const bool earlier = i >= 0 && (mi < 0 || d < md || (d == md && i < mi));
      md = earlier ? d : md; mi = earlier ? i : mi; mp = earlier ? e : mp;
    }
    const double d0 = bd[p * 64];
    const int32_t i0 = bi[p * 64];
    bd[mp * 64] = d0; bi[mp * 64] = i0;
    bd[p * 64] = md; bi[p * 64] = mi;
  }
  if (seed_out) {                                   // seeding pass: only the k-th distance is wanted
    if (live) seed_out[q] = bd[(k - 1) * 64];
    return;
  }
  if (live) {
    // slice lists go to out_* laid out [row][slice][k], row = the query's original index; with one slice that is the final result
    const int64_t o = (((int64_t)perm[js] - row_first) * gridDim.y + blockIdx.y) * k;
    const bool final_ = gridDim.y == 1;
    for (int p = 0; p < k; p++) {
      const double d = bd[p * 64];
      const bool ok = bi[p * 64] >= 0 && (!final_ || !(d > max_dist));
      out_idx[o + p] = ok ? bi[p * 64] : -1;
      out_dist[o + p] = ok ? d : 1.0 / 0.0;
    }
  }
}

// K6, wave-per-query form (k <= 64): a WAVE serves Q query vertices that are neighbours in sorted order, its 64 lanes each taking
// one candidate of a tile (a coalesced read of 64 consecutive sorted states).  No divergence between queries: the lane-per-query
// form above spends most of its time in list insertions that every lane of the wave sits through whenever ANY of its 64 queries
// inserts (issue utilisation 0.10, r03).  Per query the k best live one per lane, in order (lanes >= k hold an inert entry), the
// threshold (the list's last entry by (distance, original index)) is wave-uniform, a tile's candidates that pass it are taken
// one by one (ballot), each shifting the entries behind it one lane up.  Q queries per
// tile: a candidate fetched once is measured against Q queries -- with one query per wave the 6 x 10^5-vertex table re-read its
// 19 MB of states ~5 000 times (94 GB, 16.5 ms); the search box is the union of the Q queries' own.  Same arithmetic for the
// distances, same acceptance rule, same expanding cell search, same output order as knn_bruteforce: identical tables.
template <int NT, bool ROT, bool RET, int Q, int G>
__global__ __launch_bounds__(256) void knn_wave_query(const double *__restrict__ cand, KnnCells cg, const int32_t *__restrict__ cellstart,
                                                      const int32_t *__restrict__ perm, const int32_t *__restrict__ qlist, int64_t nq,
                                                      KnnMetric m, int k, double max_dist, int64_t row_first,
                                                      int32_t *__restrict__ out_idx, double *__restrict__ out_dist) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63;
  const int64_t q0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * Q;
  if (q0 >= nq) return;                                                 // wave-uniform
  constexpr int SS = NT + (ROT ? 1 : 0) + (RET ? 1 : 0);
  constexpr int SS_KEY = NT + (ROT ? 1 : 0);
  constexpr bool plain = !ROT && !RET;
  constexpr int K1 = RET ? 0 : (NT >= 2 ? 1 : -1);
  auto uniform = [](double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
  };
  auto lane_value = [](double v, int src) {                             // src wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
  };
  int64_t js[Q];                                                        // the queries' positions in sorted order (a query beyond nq repeats the last one)
  double x[Q][SS];
  // the lists: lane e < k holds one entry per query, +inf / -1 while empty; lanes >= k hold (-inf, 0): never the largest, never replaced
  double ld[Q], worst[Q], gate2[Q];
  int32_t li[Q], worst_i[Q];
#pragma unroll
  for (int q = 0; q < Q; q++) {
    const int64_t qi = q0 + q < nq ? q0 + q : nq - 1;
    js[q] = qlist ? (int64_t)qlist[qi] : qi;
#pragma unroll
    for (int d = 0; d < SS; d++) x[q][d] = uniform(cand[js[q] * SS + d]);
    ld[q] = lane < k ? 1.0 / 0.0 : -1.0 / 0.0;
    li[q] = lane < k ? -1 : 0;
    worst[q] = 1.0 / 0.0;                                               // wave-uniform: the threshold
    worst_i[q] = -1;                                                    // original index of the largest entry (-1: the list is not full)
    gate2[q] = worst[q] * worst[q] * (1.0 + 4.5e-16);
  }
  // The list is kept ORDERED across the lanes (lane e: the e-th entry by (distance, index), empty entries -- +inf, -1 -- last):
  // an accepted candidate shifts the entries that sort after it one lane up, the last one falls out, and the threshold is
  // whatever lane k - 1 then holds (+inf / -1 while the list is not full).  ~25 instructions per insertion; finding the new
  // largest entry of an unordered list by a butterfly reduction was ~100, and with ~80 insertions per query most of the kernel.
  auto insert = [&](double &ldq, int32_t &liq, double cd, int32_t ci, double &worstq, int32_t &worst_iq, double &gate2q) {
    const bool after = lane < k && (liq < 0 || cd < ldq || (cd == ldq && ci < liq));       // my entry sorts after the new one
    const int prev_after = __shfl_up((int)after, 1, 64);
    const double pd = __shfl_up(ldq, 1, 64);
    const int32_t pi = __shfl_up(liq, 1, 64);
    if (after) {
      const bool from_prev = lane > 0 && prev_after != 0;
      ldq = from_prev ? pd : cd;
      liq = from_prev ? pi : ci;
    }
    worstq = lane_value(ldq, k - 1);
    worst_iq = __builtin_amdgcn_readlane(liq, k - 1);
    gate2q = worstq * worstq * (1.0 + 4.5e-16);
  };
  // one tile: the lane's candidate c (position jl, `have`: it exists) against the Q queries
  auto tile = [&](const double (&c)[SS], int64_t jl, bool have) {
    double dist[Q];
    unsigned long long mask[Q], any = 0;
#pragma unroll
    for (int q = 0; q < Q; q++) {
      double s2 = 0.0;
#pragma unroll
      for (int d = 0; d < NT; d++) { const double t = x[q][d] - c[d]; s2 += t * t; }
      bool pass = have;
      if constexpr (plain) {
        // the root only of a squared distance that can still matter (the IEEE square root is ~20 instructions; a wave leaves
        // the branch unless one of its 64 candidates passes)
        pass = pass && s2 <= gate2[q];
        dist[q] = 1.0 / 0.0;
        if (pass) dist[q] = sqrt(s2);
      } else {
        dist[q] = sqrt(s2);
        if constexpr (ROT) {                                  // SO2StateSpace::distance
          double a_ = fabs(x[q][NT] - c[NT]);
          a_ = (a_ > 3.14159265358979323846) ? 2.0 * 3.14159265358979323846 - a_ : a_;
          dist[q] += m.w_rot * a_;
        }
        if constexpr (RET) {
          const double t = x[q][SS - 1] - c[SS - 1];
          dist[q] += m.w_ret * sqrt(t * t);
        }
      }
      pass = pass && dist[q] <= worst[q];
      mask[q] = __ballot(pass);
      any |= mask[q];
    }
    if (!any) return;
    const int32_t cj = perm[jl];
#pragma unroll
    for (int q = 0; q < Q; q++) {
      unsigned long long mk = mask[q];
      while (mk) {
        const int src = __builtin_ctzll(mk);
        mk &= mk - 1;
        const double cd = lane_value(dist[q], src);
        const int32_t ci = __builtin_amdgcn_readlane(cj, src);
        // the acceptance rule of knn_bruteforce, against the CURRENT threshold (it moves with every replacement)
        if (cd <= worst[q] && (cd < worst[q] || worst_i[q] < 0 || ci < worst_i[q])) insert(ld[q], li[q], cd, ci, worst[q], worst_i[q], gate2[q]);
      }
    }
  };
  // a run [a, b) of sorted positions, G tiles at a time: their reads are issued together (a wave with one tile in flight spent
  // ~3 us per tile waiting for it: 16.5 ms per 6 x 10^5 queries, whatever Q)
  auto scan = [&](int64_t a, int64_t b) {
    for (int64_t jt = a; jt < b; jt += 64 * G) {
      double c[G][SS];
      int64_t jl[G];
      bool have[G];
#pragma unroll
      for (int g = 0; g < G; g++) {
        const int64_t j = jt + 64 * g + lane;
        have[g] = j < b;
        jl[g] = have[g] ? j : b - 1;
#pragma unroll
        for (int d = 0; d < SS; d++) c[g][d] = cand[jl[g] * SS + d];
      }
#pragma unroll
      for (int g = 0; g < G; g++) {
        if (jt + 64 * g >= b) break;                                    // wave-uniform
        tile(c[g], jl[g], have[g]);
      }
    }
  };
  // cells (cx, cy, z0 .. z1): consecutive in sorted order
  auto scan_cells = [&](int cx, int cy, int z0, int z1) {
    if (z0 <= z1) { const int base = (cx * cg.B + cy) * cg.A; scan((int64_t)cellstart[base + z0], (int64_t)cellstart[base + z1 + 1]); }
  };
  {
    // Three keys, each a TERM of the metric (|key difference| <= distance): a box of cells around the queries' own, grown ring by
    // ring towards the window the current thresholds still ask for, until it covers that window.  (Two keys left ~5 000 of 6 x 10^5
    // uniform 4-D states inside the window of a 10-neighbour search, three leave ~900.)
    constexpr int K2 = RET ? (NT >= 2 ? 1 : -1) : (NT >= 3 ? 2 : -1);   // the third key's coordinate (cache_merge.hpp: KnnCells::col2)
    double k0[Q], k1[Q], k2[Q];
    int sx0 = 1 << 30, sx1 = -1, sy0 = 1 << 30, sy1 = -1, sz0 = 1 << 30, sz1 = -1;
#pragma unroll
    for (int q = 0; q < Q; q++) {
      k0[q] = RET ? m.w_ret * x[q][SS_KEY] : x[q][0];                   // the products sort_states_by_cells formed (cg.scale x coordinate)
      k1[q] = K1 >= 0 ? x[q][K1 >= 0 ? K1 : 0] : 0.0;
      k2[q] = (K2 >= 0 && cg.col2 >= 0) ? x[q][K2 >= 0 ? K2 : 0] : 0.0;
      const int cx = knn_cell_of(k0[q], cg.lo0, cg.inv0, cg.C), cy = knn_cell_of(k1[q], cg.lo1, cg.inv1, cg.B), cz = knn_cell_of(k2[q], cg.lo2, cg.inv2, cg.A);
      sx0 = cx < sx0 ? cx : sx0; sx1 = cx > sx1 ? cx : sx1; sy0 = cy < sy0 ? cy : sy0; sy1 = cy > sy1 ? cy : sy1; sz0 = cz < sz0 ? cz : sz0; sz1 = cz > sz1 ? cz : sz1;
    }
    for (int cx = sx0; cx <= sx1; cx++)
      for (int cy = sy0; cy <= sy1; cy++) scan_cells(cx, cy, sz0, sz1);
    for (;;) {
      // the window the queries' thresholds still ask for (an unfilled list: everything)
      int nx0 = 1 << 30, nx1 = -1, ny0 = 1 << 30, ny1 = -1, nz0 = 1 << 30, nz1 = -1;
#pragma unroll
      for (int q = 0; q < Q; q++) {
        double r = worst[q] < max_dist ? worst[q] : max_dist;
        r = r + r * 1e-12;
        const int a0 = knn_cell_of(k0[q] - r, cg.lo0, cg.inv0, cg.C), a1 = knn_cell_of(k0[q] + r, cg.lo0, cg.inv0, cg.C);
        const int b0 = knn_cell_of(k1[q] - r, cg.lo1, cg.inv1, cg.B), b1 = knn_cell_of(k1[q] + r, cg.lo1, cg.inv1, cg.B);
        const int c0 = knn_cell_of(k2[q] - r, cg.lo2, cg.inv2, cg.A), c1 = knn_cell_of(k2[q] + r, cg.lo2, cg.inv2, cg.A);
        nx0 = a0 < nx0 ? a0 : nx0; nx1 = a1 > nx1 ? a1 : nx1; ny0 = b0 < ny0 ? b0 : ny0; ny1 = b1 > ny1 ? b1 : ny1;
        nz0 = c0 < nz0 ? c0 : nz0; nz1 = c1 > nz1 ? c1 : nz1;
      }
      if (nx0 >= sx0 && nx1 <= sx1 && ny0 >= sy0 && ny1 <= sy1 && nz0 >= sz0 && nz1 <= sz1) break;
      // one ring towards it: every column (cx, cy) of the grown box takes the rows it has not been through
      const int tx0 = nx0 < sx0 ? sx0 - 1 : sx0, tx1 = nx1 > sx1 ? sx1 + 1 : sx1;
      const int ty0 = ny0 < sy0 ? sy0 - 1 : sy0, ty1 = ny1 > sy1 ? sy1 + 1 : sy1;
      const int tz0 = nz0 < sz0 ? sz0 - 1 : sz0, tz1 = nz1 > sz1 ? sz1 + 1 : sz1;
      for (int cx = tx0; cx <= tx1; cx++)
        for (int cy = ty0; cy <= ty1; cy++) {
          if (cx >= sx0 && cx <= sx1 && cy >= sy0 && cy <= sy1) { scan_cells(cx, cy, tz0, sz0 - 1); scan_cells(cx, cy, sz1 + 1, tz1); }
          else scan_cells(cx, cy, tz0, tz1);
        }
      sx0 = tx0; sx1 = tx1; sy0 = ty0; sy1 = ty1; sz0 = tz0; sz1 = tz1;
    }
  }
#pragma unroll
  for (int q = 0; q < Q; q++) {
    if (q0 + q >= nq) break;                                            // (a repeated last query: nothing to write)
    if (lane < k) {
      const int64_t o = ((int64_t)perm[js[q]] - row_first) * k + lane;       // (the list is in order)
      const bool ok = li[q] >= 0 && !(ld[q] > max_dist);
      out_idx[o] = ok ? li[q] : -1;
      out_dist[o] = ok ? ld[q] : 1.0 / 0.0;
    }
  }
}

// Merge the per-slice lists of a query (each ordered by (distance, original index)) into its k nearest: smallest head
// first, ties to the lower index -- the order of a stable sort of all distances.
__global__ __launch_bounds__(64) void knn_merge(const int32_t *__restrict__ part_idx, const double *__restrict__ part_dist, int64_t n,
                                                int nslice, int k, double max_dist, int32_t *__restrict__ out_idx,
                                                double *__restrict__ out_dist) {
  extern __shared__ unsigned char knn_lds[];
  uint8_t *head = knn_lds + threadIdx.x;           // [nslice][64]
  const int64_t q = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (q >= n) return;
  for (int s = 0; s < nslice; s++) head[s * 64] = 0;
  const int64_t base = q * nslice * k;
  for (int p = 0; p < k; p++) {
    double best = 1.0 / 0.0;
    int bs = -1;
    int32_t besti = -1;
    for (int s = 0; s < nslice; s++) {
      const int h = head[s * 64];
      if (h >= k) continue;
      const int64_t o = base + (int64_t)s * k + h;
      if (part_idx[o] < 0) continue;
      const double d = part_dist[o];
      if (bs < 0 || d < best || (d == best && part_idx[o] < besti)) { best = d; bs = s; besti = part_idx[o]; }
    }
    if (bs < 0 || best > max_dist) { out_idx[q * k + p] = -1; out_dist[q * k + p] = 1.0 / 0.0; continue; }
    out_idx[q * k + p] = part_idx[base + (int64_t)bs * k + head[bs * 64]];
    out_dist[q * k + p] = best;
    head[bs * 64]++;
  }
}

// A range of queries (tr_knn_range*: a rank's rows of the neighbour table): the positions j of the sorted order whose state is one of
// the queries -- perm[j] in [q0, q0 + nq) -- in ASCENDING j (neighbouring waves then search neighbouring cells), by a count per
// block of 1024 positions, a one-block scan of the counts and an ordered write.  (Round 3 brought the whole permutation to the
// host for this: 2.4 MB down and a 6 x 10^5-step loop per rank and build, a constant ~1.3 ms that no number of ranks divides.)
__device__ __forceinline__ uint32_t range_flags(const int32_t *__restrict__ perm, int64_t n, int64_t q0, int64_t nq, int64_t first, uint32_t (&cnt)[4]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t mine = 0;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int64_t j = first + r * 256 + wave * 64 + lane;
    const bool on = j < n && (int64_t)perm[j] >= q0 && (int64_t)perm[j] < q0 + nq;
    const unsigned long long m = __ballot(on);
    cnt[r] = (uint32_t)__popcll(m);
    if (on) mine |= 1u << r;
    mine |= (uint32_t)__popcll(m & (((unsigned long long)1 << lane) - 1)) << (8 + 6 * r);     // the lane's rank inside the wave's 64, per round
  }
  return mine;
}
__global__ __launch_bounds__(256) void range_positions_count(const int32_t *__restrict__ perm, int64_t n, int64_t q0, int64_t nq, uint32_t *__restrict__ bsum) {
  __shared__ uint32_t part[16];
  uint32_t cnt[4];
  (void)range_flags(perm, n, q0, nq, (int64_t)blockIdx.x * 1024, cnt);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) for (int r = 0; r < 4; r++) part[r * 4 + wave] = cnt[r];
  __syncthreads();
  if (threadIdx.x == 0) { uint32_t t = 0; for (int i = 0; i < 16; i++) t += part[i]; bsum[blockIdx.x] = t; }
}
__global__ __launch_bounds__(1024) void range_positions_scan(uint32_t *__restrict__ bsum, int64_t nb) {      // exclusive, in place, one block
  __shared__ uint32_t sh[1024];
  uint32_t carry = 0;
  for (int64_t b0 = 0; b0 < nb; b0 += 1024) {
    const int64_t i = b0 + threadIdx.x;
    const uint32_t v = i < nb ? bsum[i] : 0u;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const uint32_t a = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0u;
      __syncthreads();
      sh[threadIdx.x] += a;
      __syncthreads();
    }
    if (i < nb) bsum[i] = carry + sh[threadIdx.x] - v;
    carry += sh[1023];
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void range_positions_write(const int32_t *__restrict__ perm, int64_t n, int64_t q0, int64_t nq, const uint32_t *__restrict__ bsum,
                                                             int32_t *__restrict__ out) {
  __shared__ uint32_t part[16];
  uint32_t cnt[4];
  const uint32_t mine = range_flags(perm, n, q0, nq, (int64_t)blockIdx.x * 1024, cnt);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) for (int r = 0; r < 4; r++) part[r * 4 + wave] = cnt[r];
  __syncthreads();
  const uint32_t base = bsum[blockIdx.x];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    if (!((mine >> r) & 1u)) continue;
    uint32_t before = 0;
    for (int i = 0; i < r * 4 + wave; i++) before += part[i];                 // (round-major, wave-minor: ascending positions)
    const int64_t j = (int64_t)blockIdx.x * 1024 + r * 256 + wave * 64 + lane;
    const int64_t o = (int64_t)base + before + ((mine >> (8 + 6 * r)) & 63u);
    if (o < nq) out[o] = (int32_t)j;
  }
}

}  // namespace trk
