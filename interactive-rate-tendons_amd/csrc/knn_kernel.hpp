// knn_kernel.hpp -- K6 `knn_bruteforce`: exact k nearest neighbours of every roadmap vertex among all
// vertices, in the reference's state-space metric -- what connectionStrategy_(v) (KBoundedStrategy /
// KStarStrategy over the GNAT nn_, motion-planning/VoxelCachedLazyPRM.cpp:1339,1352,1491-1502) returns
// for each v.  Metric = OMPL CompoundStateSpace::distance as wired by Problem.cpp:101-163:
//   |d tau|_2  +  (extent / 4 pi) * arc(d theta)  +  (2 extent / L) * |d s_start|.
// One lane per query vertex; the candidate index is wave-uniform, so candidates arrive through scalar
// loads and feed the fp64 FMAs as SGPR operands; each lane keeps its k best in an LDS column (sorted
// insertion; insertions become rare once the list has warmed up).  O(n^2) by design: at roadmap sizes
// (1e5 - 1e6 vertices, <= 10 dimensions) the exact brute force is milliseconds to a second on this
// chip and needs no tree build.  Like nearestK on a structure that already holds v, the result
// includes v itself (distance 0), which connectVertices then skips (:2848).
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
template <int NT, bool ROT, bool RET>
__global__ __launch_bounds__(64) void knn_bruteforce(const double *__restrict__ states, const double *__restrict__ queries, int64_t n,
                                                     KnnMetric m, int k,
                                                     double max_dist, int64_t slice, int64_t n_cand, const double *__restrict__ seed,
                                                     double *__restrict__ seed_out, int32_t *__restrict__ out_idx,
                                                     double *__restrict__ out_dist) {
#pragma clang fp contract(off)
  extern __shared__ unsigned char knn_lds[];
  double *bd = reinterpret_cast<double *>(knn_lds) + threadIdx.x;               // [k][64]
  int32_t *bi = reinterpret_cast<int32_t *>(knn_lds + (size_t)k * 64 * 8) + threadIdx.x;
  const int64_t q = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const bool live = q < n;
  const int64_t qc = live ? q : n - 1;
  const int S = m.S;
  double x[KNN_SMAX];
#pragma unroll
  for (int d = 0; d < KNN_SMAX; d++) x[d] = d < S ? queries[qc * S + d] : 0.0;     // the n queries: all of the states, or a range of them
  for (int p = 0; p < k; p++) { bd[p * 64] = 1.0 / 0.0; bi[p * 64] = -1; }
  // `worst`: a candidate must be strictly closer than this to enter the list -- the lane's current k-th entry, or, while
  // the list is not full, the SEED: the k-th smallest distance of the query to a sample of the candidates (a first,
  // short launch of this kernel with seed_out set).  The k nearest of all candidates are at most that far, so a slice
  // starts by accepting exactly the candidates with distance <= seed instead of filling its list with whatever comes
  // first and shifting it ~k ln(slice / k) times (those insertions, not the distances, were 80 % of the kernel's time).
  double worst = 1.0 / 0.0;
  if (seed) {
    const double t = seed[qc];
    worst = (t < 1.0 / 0.0) ? __longlong_as_double(__double_as_longlong(t) + 1) : t;     // next double above: "<= seed"
  }
  double gate2 = worst * worst * (1.0 + 4.5e-16);   // plain metric: squared distances at or above this cannot beat `worst`
  constexpr bool plain = !ROT && !RET;
  const int64_t j0 = (int64_t)blockIdx.y * slice, j1 = (j0 + slice < n_cand) ? j0 + slice : n_cand;
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
      const double *__restrict__ c = FULL ? states + jb * SS + u * SS : states + j * SS;
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
    if (!__any(mn < (plain ? gate2 : worst))) return;
#pragma unroll
    for (int u = 0; u < CH; u++) {                  // in index order
      double dist = dd[u];
      if (plain) {
        if (!(dist < gate2)) continue;
        dist = sqrt(dist);
      }
      if (dist < worst) {                           // strict: on exact ties the lower index stays
        // Sorted insertion without a data-dependent loop: entry e becomes its left neighbour when that one is farther
        // than the candidate (shift), the candidate when the entry itself is the first one farther, else it stays.  Every
        // entry is a function of two OLD entries, so the LDS reads do not wait for one another (the shifting while-loop
        // paid a full LDS round trip per step: ~900 cycles per insertion against ~500 for a chunk's distances).
        const int32_t cj = (int32_t)(jb + u);
        double right = bd[(k - 1) * 64];
        int32_t righti = bi[(k - 1) * 64];
        for (int e = k - 1; e > 0; e--) {
          const double left = bd[(e - 1) * 64];
          const int32_t lefti = bi[(e - 1) * 64];
          const bool shift = left > dist, take = right > dist;
          bd[e * 64] = shift ? left : (take ? dist : right);
          bi[e * 64] = shift ? lefti : (take ? cj : righti);
          right = left; righti = lefti;
        }
        if (right > dist) { bd[0] = dist; bi[0] = cj; }
        const double kth = bd[(k - 1) * 64];
        if (kth < worst) worst = kth;               // (an unfilled list keeps the seed as its threshold)
        // sqrt(s2) < worst needs s2 < worst^2 (1 + 2^-51): beyond that the correctly rounded root is >= worst
        gate2 = worst * worst * (1.0 + 4.5e-16);
      }
    }
  };
  int64_t jb = j0;
  for (; jb + CH <= j1; jb += CH) do_chunk(std::true_type{}, jb);
  if (jb < j1) do_chunk(std::false_type{}, jb);
  if (seed_out) {                                   // sampling pass: only the k-th distance is wanted
    if (live) seed_out[q] = bd[(k - 1) * 64];
    return;
  }
  if (live) {
    // slice lists go to out_* laid out [query][slice][k]; with one slice that is the final result
    const int64_t o = (q * gridDim.y + blockIdx.y) * k;
    const bool final_ = gridDim.y == 1;
    for (int p = 0; p < k; p++) {
      const double d = bd[p * 64];
      const bool ok = bi[p * 64] >= 0 && (!final_ || !(d > max_dist));
      out_idx[o + p] = ok ? bi[p * 64] : -1;
      out_dist[o + p] = ok ? d : 1.0 / 0.0;
    }
  }
}

// Merge the per-slice lists of a query (each ordered by (distance, index), slices in index order) into its k nearest:
// smallest head first, ties to the lower slice -- the order of a stable sort of all distances.
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
    for (int s = 0; s < nslice; s++) {
      const int h = head[s * 64];
      if (h >= k) continue;
      const int64_t o = base + (int64_t)s * k + h;
      if (part_idx[o] < 0) continue;
      const double d = part_dist[o];
      if (d < best || bs < 0) { best = d; bs = s; }
    }
    if (bs < 0 || best > max_dist) { out_idx[q * k + p] = -1; out_dist[q * k + p] = 1.0 / 0.0; continue; }
    out_idx[q * k + p] = part_idx[base + (int64_t)bs * k + head[bs * 64]];
    out_dist[q * k + p] = best;
    head[bs * 64]++;
  }
}

}  // namespace trk
