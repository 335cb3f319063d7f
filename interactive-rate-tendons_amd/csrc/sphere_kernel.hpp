// sphere_kernel.hpp -- K8: the voxel test of motion_planning::VoxelValidityChecker
// (motion-planning/VoxelValidityChecker.h:18-26): the robot is voxelised as a sphere of its radius at
// every (rotated) backbone point -- VoxelOctree::add_sphere (collision/VoxelOctree.cpp:434-469): the
// point's own cell plus every cell whose CENTRE lies within r -- and tested against the obstacles
// (AbstractVoxelValidityChecker.h:55-57).  No robot voxel set is built: a configuration collides iff
// some occupied obstacle cell has its centre within r of some backbone point, or a backbone point
// lies in an occupied cell.  (add_sphere's block range by nearest_block_idx always contains every cell
// whose centre is within r, so enumerating a superset of it and applying the same IEEE distance test
// gives the same set.)
//
// One lane per configuration decides which of its points lie near obstacles at all: the obstacle
// grid dilated by ceil(r/d) + 1 cells per axis (Chebyshev; `sphere_near`, built by three
// `cheb_dilate_axis` passes whenever the grid changes) holds every cell whose sphere could touch an
// occupied cell -- a point outside the domain is looked up at its projection onto the domain, which
// is closer to every cell than the point itself.  Flagged (lane, point) pairs are then served by the
// whole wave: each lane takes blocks of the point's block range, skips empty ones and tests the set
// bits' centres with the reference's arithmetic (fp64, no contraction).
#pragma once
#include <hip/hip_runtime.h>
#include "tr_types.hpp"

namespace trk {

// out(c) = OR of in(c + k e_axis), |k| <= R, inside the grid.  One wave per block, lane = cell.
__global__ __launch_bounds__(64) void cheb_dilate_axis(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, int Nb, int axis, int R) {
  const int b = blockIdx.x;
  const int bz = b % Nb, by = (b / Nb) % Nb, bx = b / (Nb * Nb);
  const int lane = threadIdx.x, N = 4 * Nb;
  const int X = 4 * bx + (lane >> 4), Y = 4 * by + ((lane >> 2) & 3), Z = 4 * bz + (lane & 3);
  bool on = false;
  for (int k = -R; k <= R && !on; k++) {
    const int x = X + (axis == 0 ? k : 0), y = Y + (axis == 1 ? k : 0), z = Z + (axis == 2 ? k : 0);
    if (x < 0 || x >= N || y < 0 || y >= N || z < 0 || z >= N) continue;
    on = (in[((size_t)(x >> 2) * Nb + (y >> 2)) * Nb + (z >> 2)] >> (((x & 3) << 4) | ((y & 3) << 2) | (z & 3))) & 1ull;
  }
  const unsigned long long m = __ballot(on);
  if (lane == 0) out[b] = m;
}

// valid_bits (in/out): bit i = configuration i passed is_valid_shape (K2 with check_voxels = 0); cleared
// here when its sphere-swept voxel set meets an obstacle.  flags (optional) gain TR_FLAG_NO_VOXCOL (8) for
// the survivors.
__global__ __launch_bounds__(64) void spheres_vs_grid(
    const double *__restrict__ px, const double *__restrict__ py, const double *__restrict__ pz,
    const int32_t *__restrict__ n_points, int64_t n, int64_t ld, int P, double radius, GridK g,
    const uint64_t *__restrict__ grid, const uint64_t *__restrict__ sphere_near,
    uint64_t *__restrict__ valid_bits, uint8_t *__restrict__ flags) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  const bool live = i < n;
  const int64_t ic = live ? i : n - 1;
  const uint64_t word = valid_bits[(int64_t)blockIdx.x];
  bool alive = live && ((word >> lane) & 1ull);
  bool hit = false;
  const int np = n_points ? n_points[ic] : P;
  const int64_t ib = ic + (int64_t)(P - np) * ld;        // retraction robots: the lane's point j is in row j + (P - np)
  const double rr = radius * radius;
  const int Nb = g.Nb, N = g.N;
  for (int j = 0; j < P; j++) {
    if (!__any(alive && !hit && j < np)) break;
    // this lane's point j, rotated into the voxel frame
    double x = 0, y = 0, z = 0;
    bool want = alive && !hit && j < np;
    if (want) {
      const int64_t o = (int64_t)j * ld + ib;
      const double x0 = px[o], y0 = py[o], z0 = pz[o];
      if (g.rot_is_identity) { x = x0; y = y0; z = z0; }
      else {
        x = g.inv_rot[0] * x0 + g.inv_rot[1] * y0 + g.inv_rot[2] * z0;
        y = g.inv_rot[3] * x0 + g.inv_rot[4] * y0 + g.inv_rot[5] * z0;
        z = g.inv_rot[6] * x0 + g.inv_rot[7] * y0 + g.inv_rot[8] * z0;
      }
      if (!(fabs(x) < 1e300) || !(fabs(y) < 1e300) || !(fabs(z) < 1e300)) { hit = true; want = false; }   // non-finite shape: never valid
    }
    if (want) {
      // cell of the projection of the point onto the domain (nearest_cell, VoxelOctree.cpp:295-307)
      const double cx = fmin(fmax(x, g.xmin), g.xmax), cy = fmin(fmax(y, g.ymin), g.ymax), cz = fmin(fmax(z, g.zmin), g.zmax);
      int ix = (int)((cx - g.xmin) / g.dx), iy = (int)((cy - g.ymin) / g.dy), iz = (int)((cz - g.zmin) / g.dz);
      ix = ix < 0 ? 0 : (ix > N - 1 ? N - 1 : ix); iy = iy < 0 ? 0 : (iy > N - 1 ? N - 1 : iy); iz = iz < 0 ? 0 : (iz > N - 1 ? N - 1 : iz);
      want = (sphere_near[((size_t)(ix >> 2) * Nb + (iy >> 2)) * Nb + (iz >> 2)] >> (((ix & 3) << 4) | ((iy & 3) << 2) | (iz & 3))) & 1ull;
      // add_point (:319-323): the point's own cell, if the point is inside the closed domain
      if (want && cx == x && cy == y && cz == z &&
          ((grid[((size_t)(ix >> 2) * Nb + (iy >> 2)) * Nb + (iz >> 2)] >> (((ix & 3) << 4) | ((iy & 3) << 2) | (iz & 3))) & 1ull)) {
        hit = true; want = false;
      }
    }
    // serve the flagged lanes one at a time with the whole wave
    unsigned long long todo = __ballot(want);
    while (todo) {
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const double sx = __shfl(x, src), sy = __shfl(y, src), sz = __shfl(z, src);
      // block range of add_sphere: nearest_block_idx(c - r) .. nearest_block_idx(c + r) (:272-283, :446-449)
      auto blk = [&](double v, double lo, double dd) { int q = (int)((v - lo) / dd); q = q / 4; return q < 0 ? 0 : (q > Nb - 1 ? Nb - 1 : q); };
      const int lx = blk(sx - radius, g.xmin, g.dx), hx = blk(sx + radius, g.xmin, g.dx);
      const int ly = blk(sy - radius, g.ymin, g.dy), hy = blk(sy + radius, g.ymin, g.dy);
      const int lz = blk(sz - radius, g.zmin, g.dz), hz = blk(sz + radius, g.zmin, g.dz);
      const int ny = hy - ly + 1, nz = hz - lz + 1, total = (hx - lx + 1) * ny * nz;
      bool found = false;
      for (int t0 = 0; t0 < total && !__any(found); t0 += 64) {
        const int t = t0 + lane;
        if (t < total) {
          const int bx = lx + t / (ny * nz), by = ly + (t / nz) % ny, bz = lz + t % nz;
          unsigned long long w = grid[((size_t)bx * Nb + by) * Nb + bz];
          while (w && !found) {
            const int bit = __ffsll((long long)w) - 1;
            w &= w - 1;
            const double vx = g.xmin + g.dx * ((double)((bx << 2) + (bit >> 4)) + 0.5);
            const double vy = g.ymin + g.dy * ((double)((by << 2) + ((bit >> 2) & 3)) + 0.5);
            const double vz = g.zmin + g.dz * ((double)((bz << 2) + (bit & 3)) + 0.5);
            const double d0 = sx - vx, d1 = sy - vy, d2 = sz - vz;
            found = d0 * d0 + d1 * d1 + d2 * d2 <= rr;
          }
        }
      }
      if (__any(found) && lane == src) hit = true;
    }
  }
  const bool ok = alive && !hit;
  const unsigned long long m = __ballot(ok);
  if (lane == 0) valid_bits[(int64_t)blockIdx.x] = m;
  if (flags && ok) flags[i] |= 8u;
}

}  // namespace trk
