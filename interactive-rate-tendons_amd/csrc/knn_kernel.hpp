// knn_kernel.hpp -- K6 `knn_bruteforce`: exact k nearest neighbours of every roadmap vertex among all
// vertices, in the reference's state-space metric -- what connectionStrategy_(v) (KBoundedStrategy /
// KStarStrategy over the GNAT nn_, motion-planning/VoxelCachedLazyPRM.cpp:1339,1352,1491-1502) returns
// for each v.  Metric = OMPL CompoundStateSpace::distance as wired by Problem.cpp:101-163:
//   |d tau|_2  +  (extent / 4 pi) * arc(d theta)  +  (2 extent / L) * |d s_start|.
// One lane per query vertex; the candidate index is wave-uniform, so candidates arrive through scalar
// loads and feed the fp64 FMAs as SGPR operands; each lane keeps its k best in an LDS column (sorted
// insertion; insertions become rare once the list has warmed up).  Exact, and no tree: the states are sorted by their
// first coordinate (rocPRIM radix sort, cache_merge.hip) and a wave -- 64 queries that are neighbours in that order --
// only visits the candidates whose first coordinate lies within its queries' search radius of theirs (every term of the
// metric is non-negative, so |d tau_0| <= distance): the radius is the SEED, the k-th distance to the candidates nearest
// in sorted order (a first, short pass).  At 10^5 uniform 4-D states that is ~1/6 of the candidates, at 10^6 ~1/12.
// Like nearestK on a structure that already holds v, the result includes v itself (distance 0), which connectVertices
// then skips (:2848).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "tr_types.hpp"

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
// cand / xs / perm: the states in sorted order, their sort keys (first tension, or w_ret s_start), and the original index of each.  qlist (optional):
// sorted positions of the queries (ascending; null = every state is a query), nq of them; row_first: original index of
// the first output row (tr_knn_range).  half_window > 0 marks the seeding pass: candidates = the half_window sorted
// neighbours either side of the wave's queries, only the k-th distance is written (seed_out, indexed like qlist).
template <int NT, bool ROT, bool RET>
__global__ __launch_bounds__(64) void knn_bruteforce(const double *__restrict__ cand, const double *__restrict__ xs,
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
  for (int p = 0; p < k; p++) { bd[p * 64] = 1.0 / 0.0; bi[p * 64] = -1; }
  // `worst`: a candidate enters the list when it sorts before this by (distance, index) -- the lane's current k-th entry,
  // or, while the list is not full, the SEED (any index): the k nearest of all candidates are at most that far, so a slice
  // starts by accepting exactly the candidates with distance <= seed instead of filling its list with whatever comes first
  // and shifting it ~k ln(slice / k) times (those insertions, not the distances, were 80 % of the kernel's time).
  constexpr int SS_KEY = NT + (ROT ? 1 : 0);                 // position of the retraction coordinate, when there is one
  double worst = 1.0 / 0.0;
  if (seed) worst = seed[qc];
  // ---- the wave's candidate range [j0, j1) in sorted order ----
  int64_t j0, j1;
  {
    auto wave_min = [](double v) { for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64)); return v; };
    auto wave_max = [](double v) { for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64)); return v; };
    int64_t lo, hi;
    if (half_window > 0) {
      const int64_t a = (int64_t)wave_min((double)js), b = (int64_t)wave_max((double)js);      // exact: positions < 2^53
      lo = a - half_window; hi = b + 1 + half_window;
      lo = lo < 0 ? 0 : lo; hi = hi > n_cand ? n_cand : hi;
    } else {
      // |key(x) - key(c)| <= distance(x, c) (the key is one term of the metric: the first tension, or w_ret s_start when
      // the robot retracts): a candidate outside [key - r, key + r] cannot be among the k nearest when k candidates
      // within r exist (the seed).  r is widened by a relative 1e-12: sqrt(fl(t0^2 + ...)) may round an ulp below |t0|.
      double r = worst < max_dist ? worst : max_dist;
      r = r + r * 1e-12;
      const double xk = RET ? m.w_ret * x[SS_KEY] : x[0];      // the key the states were sorted by (tr_knn: sort_states_by_key)
      const double xlo = wave_min(xk - r), xhi = wave_max(xk + r);
      lo = 0; hi = n_cand;
      if (xlo > -1.0 / 0.0 && xlo == xlo) {                       // lower bound: first candidate with xs >= xlo
        int64_t a = 0, b = n_cand;
        while (a < b) { const int64_t mid = (a + b) >> 1; if (xs[mid] < xlo) a = mid + 1; else b = mid; }
        lo = a;
      }
      if (xhi < 1.0 / 0.0 && xhi == xhi) {                        // upper bound: first candidate with xs > xhi
        int64_t a = lo, b = n_cand;
        while (a < b) { const int64_t mid = (a + b) >> 1; if (xs[mid] <= xhi) a = mid + 1; else b = mid; }
        hi = a;
      }
    }
    lo = __builtin_amdgcn_readfirstlane((int)lo); hi = __builtin_amdgcn_readfirstlane((int)hi);     // wave-uniform by construction; n_cand < 2^31
    const int64_t len = hi - lo, ny = gridDim.y;
    const int64_t per = (len + ny - 1) / ny;
    j0 = lo + (int64_t)blockIdx.y * per;
    j1 = j0 + per < hi ? j0 + per : hi;
    if (j0 > hi) j0 = hi;
  }
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
        // Sorted insertion by (distance, original index) without a data-dependent loop: entry e becomes its left
        // neighbour when that one sorts after the candidate (shift), the candidate when the entry itself is the first one
        // that does, else it stays.  Every entry is a function of two OLD entries, so the LDS reads do not wait for one
        // another (the shifting while-loop paid a full LDS round trip per step: ~900 cycles per insertion against ~500
        // for a chunk's distances).  Candidates arrive in sorted-coordinate order, so equal distances are ordered here by
        // the original index -- the order of a stable sort of the query's distance row.  A candidate that ties with the
        // k-th entry and has the larger index changes nothing.
        const int32_t cj = perm[jb + u];
        auto after = [&](double d, int32_t i) { return d > dist || (d == dist && (i > cj || i < 0)); };
        double right = bd[(k - 1) * 64];
        int32_t righti = bi[(k - 1) * 64];
        for (int e = k - 1; e > 0; e--) {
          const double left = bd[(e - 1) * 64];
          const int32_t lefti = bi[(e - 1) * 64];
          const bool shift = after(left, lefti), take = after(right, righti);
          bd[e * 64] = shift ? left : (take ? dist : right);
          bi[e * 64] = shift ? lefti : (take ? cj : righti);
          right = left; righti = lefti;
        }
        if (after(right, righti)) { bd[0] = dist; bi[0] = cj; }
        const double kth = bd[(k - 1) * 64];
        if (kth < worst) worst = kth;               // (an unfilled list keeps the seed as its threshold)
        // sqrt(s2) <= worst needs s2 <= worst^2 (1 + 2^-51): beyond that the correctly rounded root is > worst
        gate2 = worst * worst * (1.0 + 4.5e-16);
      }
    }
  };
  int64_t jb = j0;
  for (; jb + CH <= j1; jb += CH) do_chunk(std::true_type{}, jb);
  if (jb < j1) do_chunk(std::false_type{}, jb);
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

}  // namespace trk
