// env_kernel.hpp -- K7: environment preparation on the dense bit-packed grid, resident in HBM:
//   grid_add_spheres      VoxelOctree::add_sphere            (collision/VoxelOctree.cpp:434-469)
//   grid_add_capsules     VoxelOctree::add_capsule           (:471-515)
//   grid_remove_interior  remove_interior_6/27neighbor       (:533-689)
//   grid_dilate_step      one step of dilate_6/27neighbor    (:693-818), dilate_sphere (:950-952)
// One wave per 4x4x4 block, lane = cell (bit x*16 + y*4 + z), the 64 verdicts packed with a ballot.
// Integer / bit work on a 2 MiB grid that lives in L2: the kernels are launch- and latency-bound
// (microseconds); what matters is that the obstacle set never leaves the device between edits.
// IEEE fp64 without contraction in add_sphere: the voxel-centre test is bit-exact.
#pragma once
#include <hip/hip_runtime.h>
#include "tr_types.hpp"

namespace trk {

// Host-prepared sphere: bounding block range by nearest_block_idx (VoxelOctree.cpp:272-283) and the
// cell add_point(centre) sets (:319-323; -1 when the centre is outside the closed domain).
struct SphereK { double cx, cy, cz, rr; int32_t lo[3], hi[3], pc[3], pad_; };

__global__ __launch_bounds__(64) void grid_add_spheres(uint64_t *__restrict__ blocks, GridK g, const SphereK *__restrict__ sp, int n) {
#pragma clang fp contract(off)
  const int b = blockIdx.x;
  const int bz = b % g.Nb, by = (b / g.Nb) % g.Nb, bx = b / (g.Nb * g.Nb);
  const int lane = threadIdx.x, i = lane >> 4, j = (lane >> 2) & 3, k = lane & 3;
  const double x = g.xmin + g.dx * ((double)((bx << 2) + i) + 0.5);
  const double y = g.ymin + g.dy * ((double)((by << 2) + j) + 0.5);
  const double z = g.zmin + g.dz * ((double)((bz << 2) + k) + 0.5);
  bool on = false;
  for (int s = 0; s < n; s++) {
    const SphereK q = sp[s];                                   // wave-uniform
    if (q.pc[0] >= 0 && (q.pc[0] >> 2) == bx && (q.pc[1] >> 2) == by && (q.pc[2] >> 2) == bz)
      on = on || ((q.pc[0] & 3) == i && (q.pc[1] & 3) == j && (q.pc[2] & 3) == k);
    if (bx < q.lo[0] || bx > q.hi[0] || by < q.lo[1] || by > q.hi[1] || bz < q.lo[2] || bz > q.hi[2]) continue;
    const double d0 = q.cx - x, d1 = q.cy - y, d2 = q.cz - z;
    on = on || (d0 * d0 + d1 * d1 + d2 * d2 <= q.rr);
  }
  const unsigned long long m = __ballot(on);
  if (lane == 0 && m) blocks[b] |= m;
}

// Host-prepared capsule: end points, r^2, block range of the bounding box, and the cells add_point(a), add_point(b) set.
struct CapsuleK { double a[3], b[3], rr; int32_t lo[3], hi[3], pa[3], pb[3]; };

// VoxelOctree::add_capsule (:471-515): a voxel is set when its centre lies within r of the segment a-b --
// collides(Capsule, Point) (collision/collision.hxx:83-87): t = closest_t_segment (collision_primitives.h:33-49), the
// point a + (b - a) t, then |closest - p|^2 <= r^2 -- plus the cells of the two end points.
__global__ __launch_bounds__(64) void grid_add_capsules(uint64_t *__restrict__ blocks, GridK g, const CapsuleK *__restrict__ cp, int n) {
#pragma clang fp contract(off)
  const int b = blockIdx.x;
  const int bz = b % g.Nb, by = (b / g.Nb) % g.Nb, bx = b / (g.Nb * g.Nb);
  const int lane = threadIdx.x, i = lane >> 4, j = (lane >> 2) & 3, k = lane & 3;
  const double x = g.xmin + g.dx * ((double)((bx << 2) + i) + 0.5);
  const double y = g.ymin + g.dy * ((double)((by << 2) + j) + 0.5);
  const double z = g.zmin + g.dz * ((double)((bz << 2) + k) + 0.5);
  const double eps = 2.220446049250313e-16;
  bool on = false;
  for (int s = 0; s < n; s++) {
    const CapsuleK q = cp[s];                                  // wave-uniform
    if (q.pa[0] >= 0 && (q.pa[0] >> 2) == bx && (q.pa[1] >> 2) == by && (q.pa[2] >> 2) == bz)
      on = on || ((q.pa[0] & 3) == i && (q.pa[1] & 3) == j && (q.pa[2] & 3) == k);
    if (q.pb[0] >= 0 && (q.pb[0] >> 2) == bx && (q.pb[1] >> 2) == by && (q.pb[2] >> 2) == bz)
      on = on || ((q.pb[0] & 3) == i && (q.pb[1] & 3) == j && (q.pb[2] & 3) == k);
    if (bx < q.lo[0] || bx > q.hi[0] || by < q.lo[1] || by > q.hi[1] || bz < q.lo[2] || bz > q.hi[2]) continue;
    const double d0 = q.b[0] - q.a[0], d1 = q.b[1] - q.a[1], d2 = q.b[2] - q.a[2];
    const double dsq = d0 * d0 + d1 * d1 + d2 * d2;
    double t = 0.0;
    if (!(dsq <= eps * eps)) t = (d0 * (x - q.a[0]) + d1 * (y - q.a[1]) + d2 * (z - q.a[2])) / dsq;
    t = fmax(0.0, fmin(1.0, t));
    const double c0 = q.a[0] + d0 * t, c1 = q.a[1] + d1 * t, c2 = q.a[2] + d2 * t;
    const double e0 = c0 - x, e1 = c1 - y, e2 = c2 - z;
    on = on || (e0 * e0 + e1 * e1 + e2 * e2 <= q.rr);
  }
  const unsigned long long m = __ballot(on);
  if (lane == 0 && m) blocks[b] |= m;
}

// cell (X, Y, Z) of a dense grid of Nb^3 blocks; outside the grid reads as `outside`
__device__ inline bool grid_bit(const uint64_t *__restrict__ grid, int Nb, int X, int Y, int Z, bool outside) {
  const int N = 4 * Nb;
  if (X < 0 || X >= N || Y < 0 || Y >= N || Z < 0 || Z >= N) return outside;
  return (grid[((size_t)(X >> 2) * Nb + (Y >> 2)) * Nb + (Z >> 2)] >> (((X & 3) << 4) | ((Y & 3) << 2) | (Z & 3))) & 1ull;
}

// A cell is interior when it and its neighbours (6 face neighbours, or the whole 3x3x3) are all
// occupied, cells beyond the grid counting as occupied (:545-551, :627-633); interior cells are cleared.
__global__ __launch_bounds__(64) void grid_remove_interior(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, int Nb, int keep_diagonal) {
  const int b = blockIdx.x;
  const uint64_t old = in[b];
  if (old == 0) { if (threadIdx.x == 0) out[b] = 0; return; }
  const int bz = b % Nb, by = (b / Nb) % Nb, bx = b / (Nb * Nb);
  const int lane = threadIdx.x;
  const int X = 4 * bx + (lane >> 4), Y = 4 * by + ((lane >> 2) & 3), Z = 4 * bz + (lane & 3);
  bool interior = (old >> lane) & 1ull;
  if (interior) {
    if (keep_diagonal) {
      for (int dx = -1; dx <= 1 && interior; dx++)
        for (int dy = -1; dy <= 1 && interior; dy++)
          for (int dz = -1; dz <= 1; dz++)
            if (!grid_bit(in, Nb, X + dx, Y + dy, Z + dz, true)) { interior = false; break; }
    } else {
      interior = grid_bit(in, Nb, X - 1, Y, Z, true) && grid_bit(in, Nb, X + 1, Y, Z, true) && grid_bit(in, Nb, X, Y - 1, Z, true) &&
                 grid_bit(in, Nb, X, Y + 1, Z, true) && grid_bit(in, Nb, X, Y, Z - 1, true) && grid_bit(in, Nb, X, Y, Z + 1, true);
    }
  }
  const unsigned long long m = __ballot(interior);
  if (lane == 0) out[b] = old & ~m;
}

// Dilation works on a grid with a one-block apron (NbW = Nb + 2 blocks per axis), because the
// reference expands up to four steps inside a 12^3 neighbourhood before clipping to the grid (:700-760).
__global__ __launch_bounds__(256) void grid_embed(const uint64_t *__restrict__ grid, uint64_t *__restrict__ work, int Nb) {
  const int NbW = Nb + 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)NbW * NbW * NbW) return;
  const int bz = (int)(i % NbW) - 1, by = (int)((i / NbW) % NbW) - 1, bx = (int)(i / ((int64_t)NbW * NbW)) - 1;
  const bool in = bx >= 0 && bx < Nb && by >= 0 && by < Nb && bz >= 0 && bz < Nb;
  work[i] = in ? grid[((size_t)bx * Nb + by) * Nb + bz] : 0;
}
__global__ __launch_bounds__(256) void grid_extract(const uint64_t *__restrict__ work, uint64_t *__restrict__ grid, int Nb) {
  const int NbW = Nb + 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)Nb * Nb * Nb) return;
  const int bz = (int)(i % Nb), by = (int)((i / Nb) % Nb), bx = (int)(i / ((int64_t)Nb * Nb));
  grid[i] |= work[((size_t)(bx + 1) * NbW + (by + 1)) * NbW + (bz + 1)];
}

// One expansion step on the apron grid: out(t) = in(t) | OR over moves m of in(t - m).
// 6-neighbour moves: the face neighbours (:764-771).  27-neighbour moves: the list at :776-804, which
// names (x+1, y+1, z+1) twice and (x-1, y+1, z+1) never -- reproduced, not repaired.
__global__ __launch_bounds__(64) void grid_dilate_step(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, int NbW, int use_diagonal) {
  const int b = blockIdx.x;
  const int bz = b % NbW, by = (b / NbW) % NbW, bx = b / (NbW * NbW);
  const int lane = threadIdx.x;
  bool nz = false;                                             // quick reject: the 27 surrounding blocks are empty
  if (lane < 27) {
    const int qx = bx + lane / 9 - 1, qy = by + (lane / 3) % 3 - 1, qz = bz + lane % 3 - 1;
    if (qx >= 0 && qx < NbW && qy >= 0 && qy < NbW && qz >= 0 && qz < NbW) nz = in[((size_t)qx * NbW + qy) * NbW + qz] != 0;
  }
  if (!__any(nz)) { if (lane == 0) out[b] = 0; return; }
  const int X = 4 * bx + (lane >> 4), Y = 4 * by + ((lane >> 2) & 3), Z = 4 * bz + (lane & 3);
  bool on = grid_bit(in, NbW, X, Y, Z, false);
  if (!on) {
    if (use_diagonal) {
      for (int mx = -1; mx <= 1 && !on; mx++)
        for (int my = -1; my <= 1 && !on; my++)
          for (int mz = -1; mz <= 1; mz++) {
            if (mx == -1 && my == 1 && mz == 1) continue;      // the move the reference's list lacks
            if (grid_bit(in, NbW, X - mx, Y - my, Z - mz, false)) { on = true; break; }
          }
    } else {
      on = grid_bit(in, NbW, X - 1, Y, Z, false) || grid_bit(in, NbW, X + 1, Y, Z, false) || grid_bit(in, NbW, X, Y - 1, Z, false) ||
           grid_bit(in, NbW, X, Y + 1, Z, false) || grid_bit(in, NbW, X, Y, Z - 1, false) || grid_bit(in, NbW, X, Y, Z + 1, false);
    }
  }
  const unsigned long long m = __ballot(on);
  if (lane == 0) out[b] = m;
}

}  // namespace trk
