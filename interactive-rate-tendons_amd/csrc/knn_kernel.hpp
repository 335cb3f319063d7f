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
template <int NT>            // tension dimensions, compile time: the inner product is 3 NT straight-line fp64 operations
__global__ __launch_bounds__(64) void knn_bruteforce(const double *__restrict__ states, int64_t n, KnnMetric m, int k,
                                                     double max_dist, int64_t slice, int32_t *__restrict__ out_idx,
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
  for (int d = 0; d < KNN_SMAX; d++) x[d] = d < S ? states[qc * S + d] : 0.0;
  for (int p = 0; p < k; p++) { bd[p * 64] = 1.0 / 0.0; bi[p * 64] = -1; }
  double worst = 1.0 / 0.0;                         // distance of the lane's current k-th entry
  double gate2 = 1.0 / 0.0;                         // plain metric: squared distances at or above this cannot beat `worst`
  const bool plain = !m.has_rot && !m.has_ret;
  const int64_t j0 = (int64_t)blockIdx.y * slice, j1 = (j0 + slice < n) ? j0 + slice : n;
#pragma unroll 8
  for (int64_t j = j0; j < j1; j++) {
    const double *__restrict__ c = states + j * S;  // wave-uniform
    double s2 = 0.0;
#pragma unroll
    for (int d = 0; d < NT; d++) { const double t = x[d] - c[d]; s2 += t * t; }
    double dist;
    if (plain) {
      if (!(s2 < gate2)) continue;
      dist = sqrt(s2);
    } else {
      dist = sqrt(s2);
      int col = NT;
      if (m.has_rot) {                              // SO2StateSpace::distance
        double a = fabs(x[NT] - c[NT]);
        a = (a > 3.14159265358979323846) ? 2.0 * 3.14159265358979323846 - a : a;
        dist += m.w_rot * a;
        col++;
      }
      if (m.has_ret) {
        const double xs = m.has_rot ? x[NT + 1] : x[NT];
        const double t = xs - c[col];
        dist += m.w_ret * sqrt(t * t);
      }
    }
    if (dist < worst) {                             // strict: on exact ties the lower index stays
      int p = k - 1;
      while (p > 0 && bd[(p - 1) * 64] > dist) { bd[p * 64] = bd[(p - 1) * 64]; bi[p * 64] = bi[(p - 1) * 64]; p--; }
      bd[p * 64] = dist; bi[p * 64] = (int32_t)j;
      worst = bd[(k - 1) * 64];
      // sqrt(s2) < worst needs s2 < worst^2 (1 + 2^-51): beyond that the correctly rounded root is >= worst
      gate2 = worst * worst * (1.0 + 4.5e-16);
    }
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
