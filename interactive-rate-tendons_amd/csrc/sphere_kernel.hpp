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
// One lane per configuration classifies each of its points with ONE gather from a distance field: the
// distance from the centre of the point's cell to the nearest occupied cell centre (`obstacle_distance_*`,
// an exact windowed separable transform rebuilt whenever the grid changes; float, used only with margins).
// With hd the half diagonal of a cell: field - hd > r means no occupied centre can be within r of the point
// (a point outside the domain is looked up at its projection onto it, which is closer to every cell than the
// point itself); field + hd < r means one certainly is.  Only the thin shell in between -- and the points
// outside the domain -- are served exactly, by the whole wave: each lane takes blocks of the point's block
// range, skips empty ones and tests the set bits' centres with the reference's arithmetic (fp64, no contraction).
#pragma once
#include <hip/hip_runtime.h>
#include "tr_types.hpp"

namespace trk {

#define TRK_EDT_FAR 1e30f

// The exact test for ONE point (wave-uniform sx, sy, sz, already in the voxel frame), served by the whole wave: does an
// occupied cell have its centre within `radius` of the point?  Each lane takes blocks of add_sphere's block range
// (nearest_block_idx(c - r) .. nearest_block_idx(c + r), VoxelOctree.cpp:272-283, :446-449), skips empty ones and tests the
// set bits' centres with the reference's arithmetic.  Returns the same answer in every lane.
__device__ __forceinline__ bool sphere_scan_wave(double sx, double sy, double sz, double radius, const GridK &g,
                                                 const uint64_t *__restrict__ grid, int lane) {
#pragma clang fp contract(off)
  const int Nb = g.Nb;
  const double rr = radius * radius;
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
  return __any(found) != 0;
}

#ifndef TRK_DEVICE_BODIES_ONLY      // verdict_kernel.hpp takes sphere_scan_wave only

// pass 1: squared distance (metres^2) along x to the nearest occupied cell within +-R cells.  One wave per
// block, lane = cell.
__global__ __launch_bounds__(64) void obstacle_distance_x(const uint64_t *__restrict__ grid, float *__restrict__ out, int Nb, int R, float dx) {
  const int b = blockIdx.x;
  const int bz = b % Nb, by = (b / Nb) % Nb, bx = b / (Nb * Nb);
  const int lane = threadIdx.x, N = 4 * Nb;
  const int X = 4 * bx + (lane >> 4), Y = 4 * by + ((lane >> 2) & 3), Z = 4 * bz + (lane & 3);
  float best = TRK_EDT_FAR;
  for (int k = -R; k <= R; k++) {
    const int x = X + k;
    if (x < 0 || x >= N) continue;
    if ((grid[((size_t)(x >> 2) * Nb + (Y >> 2)) * Nb + (Z >> 2)] >> (((x & 3) << 4) | ((Y & 3) << 2) | (Z & 3))) & 1ull) {
      const float dd = (float)k * dx;
      best = fminf(best, dd * dd);
    }
  }
  out[((size_t)X * N + Y) * N + Z] = best;
}

// passes 2 and 3: out(c) = min over |k| <= R of in(c + k e_axis) + (k d)^2; the last pass stores the root.
__global__ __launch_bounds__(256) void obstacle_distance_axis(const float *__restrict__ in, float *__restrict__ out, int N, int axis, int R,
                                                              float dd, int take_root) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)N * N * N) return;
  const int Z = (int)(i % N), Y = (int)((i / N) % N);
  const int c = axis == 1 ? Y : Z;
  const int64_t stride = axis == 1 ? N : 1;
  float best = TRK_EDT_FAR;
  for (int k = -R; k <= R; k++) {
    if (c + k < 0 || c + k >= N) continue;
    const float e = (float)k * dd;
    best = fminf(best, in[i + k * stride] + e * e);
  }
  out[i] = take_root ? sqrtf(best) : best;
}

// valid_bits (in/out): bit i = configuration i passed is_valid_shape (K2 with check_voxels = 0); cleared
// here when its sphere-swept voxel set meets an obstacle.  flags (optional) gain TR_FLAG_NO_VOXCOL (8) for
// the survivors.
__global__ __launch_bounds__(64) void spheres_vs_grid(
    const double *__restrict__ px, const double *__restrict__ py, const double *__restrict__ pz,
    const int32_t *__restrict__ n_points, int64_t n, int64_t ld, int P, double radius, GridK g,
    const uint64_t *__restrict__ grid, const float *__restrict__ field /* [N][N][N] distance to the nearest occupied centre */,
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
  const int Nb = g.Nb, N = g.N;
  const float hd = 0.5f * sqrtf((float)(g.dx * g.dx + g.dy * g.dy + g.dz * g.dz));
  const float r_lo = (float)radius - hd - 1e-6f, r_hi = (float)radius + hd + 1e-6f;
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
      const float dn = field[((size_t)ix * N + iy) * N + iz];
      const bool inside = cx == x && cy == y && cz == z;
      if (dn > r_hi) want = false;                             // no occupied centre within r of any point of this cell
      else if (inside && dn < r_lo) { hit = true; want = false; }   // one certainly is (also add_point's own cell: dn = 0)
      // add_point (:319-323): the point's own cell, if the point is inside the closed domain
      else if (inside &&
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
      if (sphere_scan_wave(sx, sy, sz, radius, g, grid, lane) && lane == src) hit = true;
    }
  }
  const bool ok = alive && !hit;
  const unsigned long long m = __ballot(ok);
  if (lane == 0) valid_bits[(int64_t)blockIdx.x] = m;
  if (flags && ok) flags[i] |= 8u;
}

#endif  // TRK_DEVICE_BODIES_ONLY

}  // namespace trk
