// edge_kernel.hpp -- K3 helper `pair_cells_differ`: the `should_subdivide` test of
// VoxelEnvironment::voxelize_valid_backbone_motion (motion-planning/VoxelEnvironment.cpp:304-341)
// for a batch of sample pairs: do two backbone shapes differ by more than one voxel, on any axis,
// at any backbone point?  Points are rotated into the voxel frame first (the reference stores the
// rotated shapes, :262-272) and located with find_cell (collision/VoxelOctree.cpp:309-317: closed
// domain check, then size_t((x - min) / d) -- a DIVISION, unlike add_line's reciprocal multiply).
// IEEE fp64, no contraction: the flags are integer-exact functions of the points.
#pragma once
#include <hip/hip_runtime.h>
#include "tr_types.hpp"
#include "sweep_kernel.hpp"

namespace trk {

struct PairK { int32_t a, b; };

// out[i]: 0 = do not subdivide, 1 = subdivide, 2 = a point lies outside the voxel domain
// (std::domain_error in the reference).  Points are visited from the tip down, as the reference
// does, so an early "subdivide" wins over a later out-of-domain point exactly as it does there.
__global__ __launch_bounds__(256) void pair_cells_differ(
    const double *__restrict__ px, const double *__restrict__ py, const double *__restrict__ pz, int64_t ld, int P,
    const int32_t *__restrict__ n_points /* per sample, or null = P */, const PairK *__restrict__ pairs, int64_t n_pairs, GridK g, uint8_t *__restrict__ out) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pairs) return;
  const int64_t sa = pairs[i].a, sb = pairs[i].b;
  uint8_t res = 0;
  int Pm = P;
  if (n_points) {
    // retraction: backbones of different lengths (VoxelEnvironment.cpp:317-326): more than one link
    // apart -> subdivide; otherwise compare the common prefix of points
    const int na = n_points[sa], nb = n_points[sb];
    if (na + 1 < nb || na > nb + 1) { out[i] = 1; return; }
    Pm = na < nb ? na : nb;
  }
  for (int j = Pm - 1; j >= 0; j--) {
    const int64_t oa = (int64_t)j * ld + sa, ob = (int64_t)j * ld + sb;
    V3 A = {px[oa], py[oa], pz[oa]}, B = {px[ob], py[ob], pz[ob]};
    if (!g.rot_is_identity) {
      const V3 a0 = A, b0 = B;
      A.x = g.inv_rot[0] * a0.x + g.inv_rot[1] * a0.y + g.inv_rot[2] * a0.z;
      A.y = g.inv_rot[3] * a0.x + g.inv_rot[4] * a0.y + g.inv_rot[5] * a0.z;
      A.z = g.inv_rot[6] * a0.x + g.inv_rot[7] * a0.y + g.inv_rot[8] * a0.z;
      B.x = g.inv_rot[0] * b0.x + g.inv_rot[1] * b0.y + g.inv_rot[2] * b0.z;
      B.y = g.inv_rot[3] * b0.x + g.inv_rot[4] * b0.y + g.inv_rot[5] * b0.z;
      B.z = g.inv_rot[6] * b0.x + g.inv_rot[7] * b0.y + g.inv_rot[8] * b0.z;
    }
    const bool in_a = !(A.x < g.xmin || g.xmax < A.x || A.y < g.ymin || g.ymax < A.y || A.z < g.zmin || g.zmax < A.z);
    const bool in_b = !(B.x < g.xmin || g.xmax < B.x || B.y < g.ymin || g.ymax < B.y || B.z < g.zmin || g.zmax < B.z);
    // NaN compares false everywhere above, i.e. "inside"; treat non-finite as a domain error too
    if (!in_a || !in_b || !(fabs(A.x) < 1e300) || !(fabs(B.x) < 1e300) || !(fabs(A.y) < 1e300) || !(fabs(B.y) < 1e300) ||
        !(fabs(A.z) < 1e300) || !(fabs(B.z) < 1e300)) { res = 2; break; }
    const long ax = (long)((A.x - g.xmin) / g.dx), ay = (long)((A.y - g.ymin) / g.dy), az = (long)((A.z - g.zmin) / g.dz);
    const long bx = (long)((B.x - g.xmin) / g.dx), by = (long)((B.y - g.ymin) / g.dy), bz = (long)((B.z - g.zmin) / g.dz);
    const long dx = ax > bx ? ax - bx : bx - ax, dy = ay > by ? ay - by : by - ay, dz = az > bz ? az - bz : bz - az;
    if (dx > 1 || dy > 1 || dz > 1) { res = 1; break; }
  }
  out[i] = res;
}

}  // namespace trk
