// sweep_kernel.hpp -- K2 `backbone_voxel_sweep`: the validity predicate on computed backbone
// shapes, one wavefront lane per configuration, one verdict bit per lane gathered with a
// wavefront ballot.
//
// Predicate (motion-planning/AbstractValidityChecker.cpp:99-133, short-circuit order kept):
//   converged  &&  min_length <= L_home_i - L_i <= max_length  (tendon/TendonRobot.h:247-278)
//   && !collides_self (collision/collision.cpp:6-46)
//   && !obstacles.collides(voxelize(backbone))
//      (VoxelBackboneValidityChecker.h:49-57 -> VoxelEnvironment.cpp:129-131 rotate_points ->
//       VoxelOctree::add_piecewise_line/add_line, collision/VoxelOctree.cpp:325-432 ->
//       VoxelOctree::collides :973-978)
//
// Everything in this file must give the SAME verdict as a plain IEEE fp64 evaluation of the
// reference's statements on the same points: no FMA contraction (pragma below), IEEE division and
// square root, the reference's evaluation order.  The robot's own voxel set is never materialised:
// "obstacles.collides(robot_voxels)" is true iff any cell add_line would set is occupied, so each
// visited cell is tested against the bit-packed obstacle grid as the DDA produces it (64-bit
// blocks, 4x4x4 voxels, bit x*16+y*4+z -- VoxelOctree.cpp:1501-1503; 2 MiB for 256^3, resident in
// every XCD's L2).
#pragma once
#include <hip/hip_runtime.h>
#include "tr_types.hpp"

#ifndef TRK_K2_PF
#define TRK_K2_PF 4      // points prefetched ahead per lane in backbone_voxel_sweep (A/B-tunable)
#endif

namespace trk {

struct V3 { double x, y, z; };

__device__ __forceinline__ double dot3(const V3 &a, const V3 &b) {
#pragma clang fp contract(off)
  return a.x * b.x + a.y * b.y + a.z * b.z;
}

// collision/collision_primitives.h:62-85 (segment_aabox_intersect), C = lower-left, D = upper-right
__device__ __forceinline__ bool segment_aabox_intersect(const V3 &A, const V3 &B, const GridK &g) {
#pragma clang fp contract(off)
  const V3 AB = {B.x - A.x, B.y - A.y, B.z - A.z};
  const double len = sqrt(dot3(AB, AB)) / 2;
  const V3 U = {AB.x / (2 * len), AB.y / (2 * len), AB.z / (2 * len)};
  const V3 Ua = {fabs(U.x), fabs(U.y), fabs(U.z)};
  const V3 P = {(A.x + B.x) / 2 - (g.xmax + g.xmin) / 2, (A.y + B.y) / 2 - (g.ymax + g.ymin) / 2,
                (A.z + B.z) / 2 - (g.zmax + g.zmin) / 2};
  const V3 ext = {fabs(g.xmax - g.xmin) / 2, fabs(g.ymax - g.ymin) / 2, fabs(g.zmax - g.zmin) / 2};
  const V3 UxP = {fabs(U.y * P.z - U.z * P.y), fabs(U.z * P.x - U.x * P.z), fabs(U.x * P.y - U.y * P.x)};
  const V3 Pa = {fabs(P.x), fabs(P.y), fabs(P.z)};
  const bool separated = Pa.x > ext.x + len * Ua.x || Pa.y > ext.y + len * Ua.y || Pa.z > ext.z + len * Ua.z ||
                         UxP.x > ext.y * Ua.z + ext.z * Ua.y || UxP.y > ext.z * Ua.x + ext.x * Ua.z ||
                         UxP.z > ext.x * Ua.y + ext.y * Ua.x;
  return !separated;
}

// Per-lane cursor into the obstacle grid with a one-block register cache.
struct GridCursor {
  const uint64_t *__restrict__ blocks;
  int Nb;
  int cached_id;
  uint64_t cached;
  __device__ __forceinline__ bool occupied(int x, int y, int z) {
    const int id = ((x >> 2) * Nb + (y >> 2)) * Nb + (z >> 2);
    if (id != cached_id) { cached = blocks[id]; cached_id = id; }
    return (cached >> (((x & 3) << 4) | ((y & 3) << 2) | (z & 3))) & 1ull;
  }
};

// VoxelOctree::add_line (collision/VoxelOctree.cpp:325-426) with set_cell replaced by a callback
// `on_cell(x, y, z) -> bool` (true = stop early).  Visits, like the reference, A's cell and B's cell
// (when inside the grid) and then the cells of the walk.  Returns true iff a callback asked to stop.
// `bad` is raised when an endpoint is non-finite or farther than 4N voxels outside the domain
// (reference: undefined behaviour / unbounded walk); the caller then forces the configuration invalid.
// `A`, `B` are the endpoints already in voxel coordinates ((a - ll) * (1/d), VoxelOctree.cpp:338-340).
template <class OnCell>
__device__ __forceinline__ bool walk_cells(const V3 &A, const V3 &B, const GridK &g, OnCell &&on_cell) {
#pragma clang fp contract(off)
  const int N = g.N;
  const int Axi = (int)A.x - (A.x < 0), Ayi = (int)A.y - (A.y < 0), Azi = (int)A.z - (A.z < 0);
  const int Bxi = (int)B.x - (B.x < 0), Byi = (int)B.y - (B.y < 0), Bzi = (int)B.z - (B.z < 0);
  auto idx_in = [N](int q) { return 0 <= q && q < N; };
  auto vox_in = [&](int x, int y, int z) { return idx_in(x) && idx_in(y) && idx_in(z); };
  bool entered = vox_in(Axi, Ayi, Azi);
  if (entered && on_cell(Axi, Ayi, Azi)) return true;
  if (vox_in(Bxi, Byi, Bzi) && on_cell(Bxi, Byi, Bzi)) return true;

  V3 U = {B.x - A.x, B.y - A.y, B.z - A.z};
  {
    const double z = dot3(U, U);                   // Eigen normalized(): n / sqrt(z) if z > 0
    if (z > 0.0) { const double s = sqrt(z); U.x = U.x / s; U.y = U.y / s; U.z = U.z / s; }
  }
  const int step_x = 1 - 2 * (U.x < 0), step_y = 1 - 2 * (U.y < 0), step_z = 1 - 2 * (U.z < 0);
  const double ex = fabs(A.x - (Axi + step_x) * g.dx);
  const double ey = fabs(A.y - (Ayi + step_y) * g.dy);
  const double ez = fabs(A.z - (Azi + step_z) * g.dz);
  const double uax = fabs(U.x), uay = fabs(U.y), uaz = fabs(U.z);
  const double threshold = 1e-10;
  const double tx_delta = (uax > threshold) ? 1 / uax : 1 / threshold;
  const double ty_delta = (uay > threshold) ? 1 / uay : 1 / threshold;
  const double tz_delta = (uaz > threshold) ? 1 / uaz : 1 / threshold;
  double tx = fabs(ex * tx_delta), ty = fabs(ey * ty_delta), tz = fabs(ez * tz_delta);
  int xi = Axi, yi = Ayi, zi = Azi;
  // the walk visits at most |dBx|+|dBy|+|dBz|+1 cells; the cap only guards against corrupt input
  for (int guard = 0; guard < 32 * N + 64; ++guard) {
    if (!(step_x * (Bxi - xi) >= 0 && step_y * (Byi - yi) >= 0 && step_z * (Bzi - zi) >= 0)) break;
    const bool tx_is_min = (tx < ty) && (tx < tz);
    const bool ty_is_min = !(tx < ty) && (ty < tz);
    if (tx_is_min) {
      xi += step_x;
      if (entered && !idx_in(xi)) break;
      tx += tx_delta;
    } else if (ty_is_min) {
      yi += step_y;
      if (entered && !idx_in(yi)) break;
      ty += ty_delta;
    } else {
      zi += step_z;
      if (entered && !idx_in(zi)) break;
      tz += tz_delta;
    }
    if (!entered && vox_in(xi, yi, zi)) entered = true;
    if (entered && on_cell(xi, yi, zi)) return true;
  }
  return false;
}

// Pre-test + voxel coordinates shared by every use of the walk.  Returns false when the segment
// misses the domain (add_line's early return).  `inside`: both endpoints at least 1e-6 of the box
// size inside the domain, in which case segment_aabox_intersect is true by a margin far larger than
// its rounding error and is not evaluated.
__device__ __forceinline__ bool line_setup(const V3 &a, const V3 &b, const GridK &g, V3 &A, V3 &B, bool &inside, bool &bad) {
#pragma clang fp contract(off)
  const double mx = 1e-6 * (g.xmax - g.xmin), my = 1e-6 * (g.ymax - g.ymin), mz = 1e-6 * (g.zmax - g.zmin);
  inside = a.x > g.xmin + mx && a.x < g.xmax - mx && b.x > g.xmin + mx && b.x < g.xmax - mx &&
           a.y > g.ymin + my && a.y < g.ymax - my && b.y > g.ymin + my && b.y < g.ymax - my &&
           a.z > g.zmin + mz && a.z < g.zmax - mz && b.z > g.zmin + mz && b.z < g.zmax - mz;
  if (!inside) {
    if (!segment_aabox_intersect(a, b, g)) return false;
  }
  A = V3{(a.x - g.xmin) * g.inv_dx, (a.y - g.ymin) * g.inv_dy, (a.z - g.zmin) * g.inv_dz};
  B = V3{(b.x - g.xmin) * g.inv_dx, (b.y - g.ymin) * g.inv_dy, (b.z - g.zmin) * g.inv_dz};
  const double lim = 5.0 * g.N;
  if (!(fabs(A.x) < lim && fabs(A.y) < lim && fabs(A.z) < lim && fabs(B.x) < lim && fabs(B.y) < lim && fabs(B.z) < lim)) {
    bad = true;
    return false;
  }
  return true;
}

// Does add_line(a, b) set a cell that is occupied in the obstacle grid?
//
// Fast path: `near` is the obstacle grid dilated by 2 cells (Chebyshev).  Every cell add_line can
// visit lies in [min(A,B)-1, max(A,B)+1] per axis (the walker moves from A's cell towards B's and
// stops once any axis has passed B's index by one), i.e. within Chebyshev distance |B-A|_inf + 1 of
// A's cell.  So when the end cells differ by at most one per axis and A's cell is free in `near`,
// no visited cell can be occupied and the whole fp64 set-up of the walk is skipped.
__device__ __forceinline__ bool line_hits(const V3 &a, const V3 &b, const GridK &g, GridCursor &gc, GridCursor &near,
                                          bool &bad) {
#pragma clang fp contract(off)
  V3 A, B;
  bool inside;
  if (!line_setup(a, b, g, A, B, inside, bad)) return false;
  if (inside && near.blocks) {
    const int Axi = (int)A.x - (A.x < 0), Ayi = (int)A.y - (A.y < 0), Azi = (int)A.z - (A.z < 0);
    const int Bxi = (int)B.x - (B.x < 0), Byi = (int)B.y - (B.y < 0), Bzi = (int)B.z - (B.z < 0);
    const int ddx = Bxi - Axi, ddy = Byi - Ayi, ddz = Bzi - Azi;
    if (ddx >= -1 && ddx <= 1 && ddy >= -1 && ddy <= 1 && ddz >= -1 && ddz <= 1 && !near.occupied(Axi, Ayi, Azi)) return false;
  }
  return walk_cells(A, B, g, [&](int x, int y, int z) { return gc.occupied(x, y, z); });
}

// collision/collision_primitives.cpp:10-102 (closest_st_segment) + collision.hxx:102-108
// capsule-capsule with radius sum rsum: distance^2 between closest points <= rsum^2.
__device__ __forceinline__ bool capsules_collide(const V3 &A, const V3 &B, const V3 &C, const V3 &D, double rsum) {
#pragma clang fp contract(off)
  const double eps = 2.220446049250313e-16, eps2 = eps * eps;
  double s = 0.0, t = 0.0;
  const V3 AB = {B.x - A.x, B.y - A.y, B.z - A.z};
  const V3 CD = {D.x - C.x, D.y - C.y, D.z - C.z};
  const double a = dot3(AB, AB), c = dot3(CD, CD);
  auto bound = [](double q) { return fmax(0.0, fmin(1.0, q)); };
  auto closest_AB_s = [&](const V3 &P) {
    if (a <= eps2) return 0.0;
    const V3 d = {P.x - A.x, P.y - A.y, P.z - A.z};
    return dot3(AB, d) / a;
  };
  auto closest_CD_t = [&](const V3 &P) {
    if (c <= eps2) return 0.0;
    const V3 d = {P.x - C.x, P.y - C.y, P.z - C.z};
    return dot3(CD, d) / c;
  };
  if (a <= eps2) { s = 0.0; t = bound(closest_CD_t(A)); }
  else if (c <= eps2) { s = bound(closest_AB_s(C)); t = 0.0; }
  else {
    const V3 AC = {C.x - A.x, C.y - A.y, C.z - A.z};
    const double b = dot3(AB, CD), d = dot3(AC, AB), e = dot3(AC, CD);
    const double denom = fmax(0.0, a * c - b * b);
    if (denom <= eps2) {
      bool found = false;
      t = closest_CD_t(A);
      if (0.0 <= t && t <= 1.0) { s = 0.0; found = true; }
      if (!found) { t = closest_CD_t(B); if (0.0 <= t && t <= 1.0) { s = 1.0; found = true; } }
      if (!found) { s = closest_AB_s(C); if (0.0 <= s && s <= 1.0) { t = 0.0; found = true; } }
      if (!found) {
        const V3 AD = {D.x - A.x, D.y - A.y, D.z - A.z};
        const V3 BC = {C.x - B.x, C.y - B.y, C.z - B.z};
        const V3 BD = {D.x - B.x, D.y - B.y, D.z - B.z};
        const double ac2 = dot3(AC, AC), ad2 = dot3(AD, AD), bc2 = dot3(BC, BC), bd2 = dot3(BD, BD);
        if (ac2 <= ad2 && ac2 <= bc2 && ac2 <= bd2) { s = 0.0; t = 0.0; }
        else if (ad2 <= bc2 && ad2 <= bd2) { s = 0.0; t = 1.0; }
        else if (bc2 <= bd2) { s = 1.0; t = 0.0; }
        else { s = 1.0; t = 1.0; }
      }
    } else {
      s = (c * d - b * e) / denom;
      t = (b * d - a * e) / denom;
      if (0.0 <= t && t <= 1.0) { s = bound(s); }
      else if (t < 0.0) { s = bound(-c / a); t = 0.0; }
      else { s = bound((b - c) / a); t = 1.0; }
    }
  }
  const V3 c1 = {A.x + (B.x - A.x) * s, A.y + (B.y - A.y) * s, A.z + (B.z - A.z) * s};
  const V3 c2 = {C.x + (D.x - C.x) * t, C.y + (D.y - C.y) * t, C.z + (D.z - C.z) * t};
  const V3 df = {c1.x - c2.x, c1.y - c2.y, c1.z - c2.z};
  return dot3(df, df) <= (rsum * rsum);
}

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}

struct SweepIn {
  const double *__restrict__ px, *__restrict__ py, *__restrict__ pz;   // [P][ld]
  const int32_t *__restrict__ n_points;                                // [n] or null (= P)
  const double *__restrict__ Li;                                       // [N][ld]
  const uint8_t *__restrict__ converged;                               // [n]
  const double *__restrict__ home_Li;                                  // [N][ld] per-config home lengths or null (use K.home_Li)
  double *__restrict__ acc;                                            // [P][ld] scratch: accumulated chord lengths
};

// Arguments of the sweep when it runs fused behind K1 (fused_kernel.hpp): they travel through memory,
// so they hold no SGPRs during the RK4 loop.
struct FusedSweepArgs {
  SweepIn in;
  int P, CH, NM, check_voxels;
  uint32_t debug, pad_;
  GridK g;
  const uint64_t *grid, *near_grid;
  uint64_t *valid_bits;
  uint8_t *flags;
  uint32_t *sig;                 // optional: per-sample cell signatures for the edge bisection (below), [n][sig_stride]
  int64_t sig_stride;
};

// Cell signature of a backbone point for `should_subdivide` (VoxelEnvironment.cpp:304-341): the point rotated into the voxel
// frame and located as find_cell does (collision/VoxelOctree.cpp:309-317: closed domain check, then size_t((x - min) / d)
// -- a DIVISION, unlike add_line's reciprocal multiply): 10 bits per axis, bit 30 = outside the domain or not finite
// (std::domain_error in the reference).  Two shapes "differ by more than a voxel" at a point iff their cells there differ
// by more than 1 on some axis: the bisection compares signatures instead of re-reading 24-byte points (r02 profile:
// the point-reading edge_filter moved 5.7x its algorithmic bytes, one cache line per 8-byte word).
constexpr uint32_t SIG_BAD = 1u << 30;
__device__ __forceinline__ uint32_t cell_signature(double x, double y, double z, const GridK &g) {
#pragma clang fp contract(off)
  V3 A = {x, y, z};
  if (!g.rot_is_identity) {
    A.x = g.inv_rot[0] * x + g.inv_rot[1] * y + g.inv_rot[2] * z;
    A.y = g.inv_rot[3] * x + g.inv_rot[4] * y + g.inv_rot[5] * z;
    A.z = g.inv_rot[6] * x + g.inv_rot[7] * y + g.inv_rot[8] * z;
  }
  const bool in = !(A.x < g.xmin || g.xmax < A.x || A.y < g.ymin || g.ymax < A.y || A.z < g.zmin || g.zmax < A.z);
  // NaN compares false everywhere above, i.e. "inside"; non-finite counts as a domain error too
  if (!in || !(fabs(A.x) < 1e300) || !(fabs(A.y) < 1e300) || !(fabs(A.z) < 1e300)) return SIG_BAD;
  // find_cell truncates the IEEE QUOTIENT (x - min) / d.  The product with the rounded reciprocal is within 3 ulp of it (< 4e-13
  // at up to 1024 cells per axis), so both truncate to the same cell unless the quotient lies within 1e-9 of an integer -- only
  // then is the division carried out (three fp64 divisions per point were a tenth of the edge samples' instructions)
  const double tx = A.x - g.xmin, ty = A.y - g.ymin, tz = A.z - g.zmin;
  const double qx = tx * g.inv_dx, qy = ty * g.inv_dy, qz = tz * g.inv_dz;
  int cx = (int)qx, cy = (int)qy, cz = (int)qz;                    // inside the closed domain: 0 <= q <= N
  const double fx = qx - (double)cx, fy = qy - (double)cy, fz = qz - (double)cz;
  const double lo = 1e-9, hi = 1.0 - 1e-9;
  if (!(fx > lo && fx < hi && fy > lo && fy < hi && fz > lo && fz < hi)) {
    cx = (int)(long)(tx / g.dx); cy = (int)(long)(ty / g.dy); cz = (int)(long)(tz / g.dz);
  }
  return ((uint32_t)cx & 1023u) | (((uint32_t)cy & 1023u) << 10) | (((uint32_t)cz & 1023u) << 20);
}

// How the signature rows leave the FK kernels.  The rows are sample-major ([sample][sig_stride]: edge_filter reads one pair
// of rows with the whole wave), but a wave produces ONE point of 64 different samples at a time: stored as they come, every
// 4-byte word opened a 32/64-byte write of its own (measured r03: 4.9 KB written per edge sample for a 516-byte row,
// profiles/r03/traffic_split_v1.json).  So a wave collects SIG_T consecutive points of its 64 samples in an LDS tile
// ([SIG_T][SIG_LDS_STRIDE] words, written lane-contiguous, conflict-free) and flushes it transposed: one store instruction =
// 8 rows x 32 contiguous, 32-byte-aligned bytes (rows are 64-byte aligned: sig_stride is a multiple of 16).
// `W(i)` gives the i-th 32-bit word of the wave's tile in LDS (a lambda over the kernel's own __shared__ symbol, so the
// accesses stay ds_read / ds_write).  row_of: the sample whose row this lane's points belong to (-1: none); first_row: the
// first point index this lane contributes through the tile (0, or a retraction robot's third point: its first two arrive in
// rows of their own and are stored directly).  cur / mask are wave-uniform.
// (the flush loop stays rolled: unrolled eight times inside the RK4 loop's hook it cost fk_verdict<4, .., SIG> 13 more scratch loads per step)
#ifndef TRK_SIG_FLUSH_UNROLL
#define TRK_SIG_FLUSH_UNROLL 1
#endif
constexpr int SIG_T = 8, SIG_LDS_STRIDE = 72, SIG_LDS_WORDS = SIG_T * SIG_LDS_STRIDE;
struct SigStage {
  int cur, mask;                 // wave-uniform: the tile being filled (point index / SIG_T, -1 = none), which of its words are in
  __device__ __forceinline__ void init() { cur = -1; mask = 0; }
  // row_of: the sample whose row this lane's points belong to (-1: none); first_row: the first point index this lane
  // contributes through the tile
  template <class Word>
  __device__ __forceinline__ void flush(Word &&W, uint32_t *__restrict__ sig, int64_t stride, int row_of, int first_row) {
    if (cur < 0) return;
    __syncthreads();                                   // one wave per workgroup: orders the tile's writes before the reads below
    const int lane = threadIdx.x, w = lane & (SIG_T - 1), sub = lane >> 3;
    const int j = cur * SIG_T + w;
#pragma unroll TRK_SIG_FLUSH_UNROLL
    for (int it = 0; it < 8; it++) {
      const int r = it * 8 + sub;                      // the lane whose sample's row this store serves
      const int ro = __shfl(row_of, r, 64), fr = __shfl(first_row, r, 64);
      const uint32_t v = W(w * SIG_LDS_STRIDE + r);
      if (((mask >> w) & 1) && ro >= 0 && j >= fr) sig[(int64_t)ro * stride + j] = v;
    }
    __syncthreads();
    cur = -1; mask = 0;
  }
  // point `row` (wave-uniform) of every lane that has one (`on`)
  template <class Word>
  __device__ __forceinline__ void put(Word &&W, uint32_t *__restrict__ sig, int64_t stride, int row_of, int first_row, int row, bool on,
                                      uint32_t value) {
    const int tile = row >> 3;
    if (tile != cur) { flush(W, sig, stride, row_of, first_row); cur = tile; }
    if (on) W((row & (SIG_T - 1)) * SIG_LDS_STRIDE + (int)threadIdx.x) = value;
    mask |= 1 << (row & (SIG_T - 1));
  }
};

// fk_uniform_body's point hook of the stored-point fused kernel: collects the signature of every observed point when the
// launch asks for them (edge samples); the tile sits at the start of the sweep's LDS image, which nothing else uses before
// sweep_body starts.  The arguments are re-read per point through an index the optimiser cannot hoist (as in
// verdict_kernel.hpp), so they hold no SGPRs across the RK4 loop.
struct SignatureHook {
  static constexpr int kLiInLdsFrom = 3;     // fk_kernel.hpp: li_in_lds (the fused kernels' LDS image leaves room: 10.3 of 20 KiB per wave)
  const FusedSweepArgs *sa;
  int64_t n;                     // configurations of the launch
  bool any;                      // wave-uniform: this launch wants signatures
  SigStage st;
  __device__ __forceinline__ static uint32_t &word(int i) { extern __shared__ float lds[]; return ((uint32_t *)lds)[i]; }
  __device__ __forceinline__ int row_of() const { const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x; return i < n ? (int)i : -1; }
  __device__ __forceinline__ void begin(bool) const {}
  __device__ __forceinline__ void operator()(int j, double x, double y, double z) {
    if (!any) return;
    int zero = 0;
    asm volatile("" : "+s"(zero));
    const FusedSweepArgs &a = sa[zero];
    st.put([](int i) -> uint32_t & { return word(i); }, a.sig, a.sig_stride, row_of(), 0, j, true, cell_signature(x, y, z, a.g));
  }
  __device__ __forceinline__ void finish() {
    if (!any) return;
    st.flush([](int i) -> uint32_t & { return word(i); }, sa->sig, sa->sig_stride, row_of(), 0);
  }
};

// Exact self-collision sweep for one lane set (collision/collision.cpp:6-46), reading the lane's
// points and accumulated chord lengths from global memory.  The (a, b) loops are wave-uniform; a lane
// skips ahead over pairs that provably cannot collide (distance bound) or are provably gated out
// (arc-length bound); the wave advances by the minimum skip.  `brute` disables skipping.
__device__ __forceinline__ bool exact_self_collision(const SweepIn &in, int64_t ic, int64_t ld, int P, int np,
                                                     bool act, double hmax, double r, bool brute) {
#pragma clang fp contract(off)
  const double consider = 3.0 * r;
  const double rsum = r + r;
  // conservative bounds for the skip sweep (all slack is >= 1e-9 relative, far above rounding)
  const double h_eff = hmax * (1.0 + 1e-9) + 1e-300;
  const double inv_h = 1.0 / h_eff;
  bool selfhit = false;
  act = act && np > 2;
  for (int a = 0; a < P - 3; ++a) {
    if (!__any(act && a < np - 3)) break;
    const bool act_a = act && a < np - 3;
    const int64_t oa = (int64_t)a * ld + ic;
    V3 pa = {0, 0, 0}, pa1 = {0, 0, 0};
    double acc_a1 = 0;
    if (act_a) {
      pa = V3{in.px[oa], in.py[oa], in.pz[oa]};
      pa1 = V3{in.px[oa + ld], in.py[oa + ld], in.pz[oa + ld]};
      acc_a1 = in.acc[oa + ld];
    }
    int b = a + 2;
    while (b < P - 1) {
      const bool act_b = act_a && !selfhit && b < np - 1;
      if (!__any(act_b)) break;
      int skip = 0x7fffffff;
      if (act_b) {
        const int64_t ob = (int64_t)b * ld + ic;
        const V3 pb = {in.px[ob], in.py[ob], in.pz[ob]};
        skip = 1;
        bool need_exact = true;
        if (!brute) {
          // any pair (a, b+k) has segment distance >= |pa - pb| - (k + 2) * hmax
          const V3 d = {pa.x - pb.x, pa.y - pb.y, pa.z - pb.z};
          const double D = sqrt(dot3(d, d));
          const double slack = D - rsum - 2.0 * h_eff - 1e-9 * (1.0 + D);
          if (slack > 0.0) {
            need_exact = false;
            const double m = slack * inv_h * (1.0 - 1e-9);
            skip = m > 1.0 ? (m < 1e6 ? (int)m : 1000000) : 1;
          }
        }
        if (need_exact) {
          const double gd = in.acc[ob] - acc_a1;
          if (gd < consider) {
            // gated out (collision.cpp:38-40); (a, b+k) stays gated while gd + k*hmax < 3r
            if (!brute) {
              const double m = (consider - gd) * inv_h * (1.0 - 1e-9) - 1e-9;
              skip = m > 1.0 ? (m < 1e6 ? (int)m : 1000000) : 1;
            }
          } else {
            const V3 pb1 = {in.px[ob + ld], in.py[ob + ld], in.pz[ob + ld]};
            if (capsules_collide(pa, pa1, pb, pb1, rsum)) selfhit = true;
          }
        }
      }
      b += wave_min_i32(skip);
    }
    if (selfhit) act = false;
  }
  return selfhit;
}

// Pass 2 of the sweep: proves the absence of self collision from the LDS milestones alone whenever it can (the three
// tests are described above sweep_body).  mx .. ma: the lane's milestone columns ([k * 64]); Kl: the lane's last milestone
// index; on: the lane takes part.  Returns true for a lane some span of which could not be cleared.
__device__ __forceinline__ bool milestones_unresolved(const float *__restrict__ mx, const float *__restrict__ my, const float *__restrict__ mz,
                                                      const float *__restrict__ ma, int NM, int Kl, bool on, float r) {
#pragma clang fp contract(off)
  const float mrg = 1e-6f;
  const float gate = 3.0f * r - mrg, aslack = r - mrg, bclear = 2.0f * r + mrg;
  bool unresolved = false;
  for (int a = 0; a < NM - 1; ++a) {
    const bool on_a = on && !unresolved && a < Kl;
    if (!__any(on_a)) break;
    if (on_a) {
      const float ax = mx[a * 64], ay = my[a * 64], az = mz[a * 64], aa = ma[a * 64];
      // whole remaining span first: s - c is monotone in the span, so this clears the row
      {
        const float ex = mx[Kl * 64] - ax, ey = my[Kl * 64] - ay, ez = mz[Kl * 64] - az;
        const float s = ma[Kl * 64] - aa;
        if (s < gate || s - sqrtf(ex * ex + ey * ey + ez * ez) < aslack) continue;
      }
      const float a1x = mx[(a + 1) * 64], a1y = my[(a + 1) * 64], a1z = mz[(a + 1) * 64], a1a = ma[(a + 1) * 64];
      const float Max = 0.5f * (ax + a1x), May = 0.5f * (ay + a1y), Maz = 0.5f * (az + a1z);
      const float rho_a = 0.5f * (a1a - aa);
      for (int b = a + 1; b <= Kl; ++b) {
        const float bx = mx[b * 64], by = my[b * 64], bz = mz[b * 64], ba = ma[b * 64];
        const float s = ba - aa;
        if (s < gate) continue;
        const float ex = bx - ax, ey = by - ay, ez = bz - az;
        if (s - sqrtf(ex * ex + ey * ey + ez * ez) < aslack) continue;
        // (B) first chunk [a, a+1] vs last chunk [b-1, b] of the span
        const float cx = mx[(b - 1) * 64], cy = my[(b - 1) * 64], cz = mz[(b - 1) * 64], ca = ma[(b - 1) * 64];
        const float fx = 0.5f * (bx + cx) - Max, fy = 0.5f * (by + cy) - May, fz = 0.5f * (bz + cz) - Maz;
        if (sqrtf(fx * fx + fy * fy + fz * fz) - rho_a - 0.5f * (ba - ca) > bclear) continue;
        unresolved = true;
        break;
      }
    }
  }
  return unresolved;
}

// K2.  One wave per block; dynamic LDS = 4 * NM * 64 floats (milestone x, y, z, arc per lane).
//
// Pass 1 streams every backbone point ONCE (coalesced, lane-contiguous): accumulated chord length,
// DDA of the segment against the obstacle grid, and a float copy of every CH-th point ("milestone")
// with its arc position into LDS.
// Pass 2 proves the absence of self collision from the milestones alone whenever it can.  For
// milestones i < j (span = the polyline between them) with arc length s and chord c:
//   (C) s < 3r               -> every capsule pair inside the span is gated out (collision.cpp:38-40);
//   (A) s - c < r            -> any gate-passing pair (a, b) inside the span has segment distance
//                               >= c - s + (arc(b) - arc(a+1)) >= 3r - (s - c) > 2r;
//   (B) |M_i - M_j'| - rho_i - rho_j' > 2r for the first / last chunk of the span (chunk midpoint M,
//       half arc length rho) -> those two chunks cannot touch.
// Each test keeps >= 1e-6 m of slack over the float rounding (~1e-7 m), so a cleared configuration
// gets exactly the reference's verdict ("no self collision").  A lane some span of which is not
// cleared falls back to pass 3, the exact pairwise sweep -- rare (tight curls only).
//   debug bit0: brute-force pairs in pass 3;  bit1: skip pass 2 (every lane takes pass 3);
//         bit2: disable the dilated-grid fast path of the voxel walk.
// out_map (optional): see the end of the function.  lane_valid (optional): the lane's verdict is returned there instead of written.
// TIPROWS (retraction robots, in.n_points set): K1r stores a lane's point j in row j + (P - n_points), i.e. rows
// are aligned at the tip like K1r's iterations, so that its stores -- and the loads here -- stay coalesced.
template <bool TIPROWS>
__device__ __forceinline__ void sweep_body(
    const SweepIn &in, int64_t n, int64_t ld, int P, int CH, int NM, const RobotK &K, const GridK &g, const uint64_t *__restrict__ grid,
    const uint64_t *__restrict__ near_grid, int check_voxels, uint32_t debug, uint64_t *__restrict__ valid_bits,
    uint8_t *__restrict__ flags, const int32_t *__restrict__ out_map = nullptr, bool *lane_valid = nullptr) {
#pragma clang fp contract(off)
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  float *__restrict__ mx = lds + lane;                   // [k][64] layout: conflict-free
  float *__restrict__ my = mx + (size_t)NM * 64;
  float *__restrict__ mz = my + (size_t)NM * 64;
  float *__restrict__ ma = mz + (size_t)NM * 64;
  // deferred voxel walks (see pass 1): a ring of owner lanes with the segments' rotated end points (six doubles each: the flush of
  // rounds 1 - 3 read them back from the point planes, six words from six lines per segment -- 0.03 x 384 B = 11.5 B on top of a
  // point's 24: 1.49x the algorithmic traffic) and one hit flag per lane
  uint32_t *__restrict__ wq = (uint32_t *)(lds + (size_t)4 * NM * 64);
  uint32_t *__restrict__ wflag = wq + 128;
  double *__restrict__ wpts = (double *)(wflag + 64);            // [6][128]: a.x, a.y, a.z, b.x, b.y, b.z (8-byte aligned: 4 NM 64 + 192 words)
  wflag[lane] = 0;

  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  const bool live = i < n;
  const int64_t ic = live ? i : (n - 1);
  const int NT = K.n_tendons;
  bool alive = live;
  bool conv_ok = false, len_ok = false;

  // 1. converged (the home shape always converges: TendonRobot.cpp:249-314 never clears the flag)
  if (alive) { conv_ok = in.converged[ic] != 0; alive = conv_ok; }
  // 2. tendon length limits: dl = L_home - L_fk in [min_length, max_length]
  if (alive) {
    bool ok = true;
    for (int j = 0; j < NT; j++) {
      const double home = in.home_Li ? in.home_Li[(int64_t)j * ld + ic] : K.home_Li[j];
      const double dl = home - in.Li[(int64_t)j * ld + ic];
      if (dl < K.min_len[j] || K.max_len[j] < dl) ok = false;
    }
    len_ok = ok; alive = ok;
  }
  const int np = TIPROWS ? in.n_points[ic] : P;
  const int shift = TIPROWS ? P - np : 0;                // row of the lane's point j is j + shift
  const int Kl = (np - 1 + CH - 1) / CH;                 // this lane's last milestone index

  // Pass 1: one streaming read of the points.  Arc positions of the milestones are accumulated in
  // float: they only feed the conservative tests of pass 2 (error ~2e-8 m against 1e-6 m of slack).
  float dist = 0.0f;
  bool hit = false, bad = false;
  if (__any(alive)) {
    GridCursor gc{grid, g.Nb, -1, 0ull};
    GridCursor near{(debug & 4u) ? nullptr : near_grid, g.Nb, -1, 0ull};
    V3 prev = {0, 0, 0}, prevr = {0, 0, 0};
    // The loads of point j+PF are issued before point j is processed (register ring, statically
    // indexed by unrolling): the walk below is a long dependent chain, and without the prefetch each
    // iteration exposes a full HBM round trip (r01: 49 % of K2's wave cycles were s_waitcnt).
    constexpr int PF = TRK_K2_PF;
    V3 ring[PF];
#pragma unroll
    for (int u = 0; u < PF; u++) {
      const int64_t o = (int64_t)(u < P ? u : P - 1) * ld + ic;
      ring[u] = V3{in.px[o], in.py[o], in.pz[o]};
    }
    // Per-point quantities are computed once and carried to the next segment (a point is the end of one
    // segment and the start of the next): "strictly inside the domain by the 1e-6 margin", the voxel
    // coordinates (a - ll) * (1/d) exactly as add_line forms them, and the cell index.
    const double bx0 = g.xmin + 1e-6 * (g.xmax - g.xmin), bx1 = g.xmax - 1e-6 * (g.xmax - g.xmin);
    const double by0 = g.ymin + 1e-6 * (g.ymax - g.ymin), by1 = g.ymax - 1e-6 * (g.ymax - g.ymin);
    const double bz0 = g.zmin + 1e-6 * (g.zmax - g.zmin), bz1 = g.zmax - 1e-6 * (g.zmax - g.zmin);
    const bool use_near = near.blocks != nullptr;
    bool in_prev = false;
    int cpx = 0, cpy = 0, cpz = 0;
    uint64_t near_prev = 0;                               // dilated-grid word of the previous point's block
    V3 seg_a = {0, 0, 0}, seg_b = {0, 0, 0};              // the deferred segment of the point being visited (rotated end points)
    auto visit = [&](int j, const V3 &q) -> bool {
      bool need = false;
      if (j == 0) prev = q;
      const float dx = (float)(q.x - prev.x), dy = (float)(q.y - prev.y), dz = (float)(q.z - prev.z);
      dist += sqrtf(dx * dx + dy * dy + dz * dz);
      prev = q;
      const bool last = (j == np - 1);
      if (last || (j % CH) == 0) {
        const int k = last ? Kl : j / CH;
        mx[k * 64] = (float)q.x; my[k * 64] = (float)q.y; mz[k * 64] = (float)q.z; ma[k * 64] = dist;
      }
      if (check_voxels == 1 && !hit && !bad) {
        V3 qr;
        if (g.rot_is_identity) { qr = q; }
        else {
          qr.x = g.inv_rot[0] * q.x + g.inv_rot[1] * q.y + g.inv_rot[2] * q.z;
          qr.y = g.inv_rot[3] * q.x + g.inv_rot[4] * q.y + g.inv_rot[5] * q.z;
          qr.z = g.inv_rot[6] * q.x + g.inv_rot[7] * q.y + g.inv_rot[8] * q.z;
        }
        const bool in_q = qr.x > bx0 && qr.x < bx1 && qr.y > by0 && qr.y < by1 && qr.z > bz0 && qr.z < bz1;
        const V3 Bq = {(qr.x - g.xmin) * g.inv_dx, (qr.y - g.ymin) * g.inv_dy, (qr.z - g.zmin) * g.inv_dz};
        // inside the margin box the voxel coordinates are in (0, N): floor == truncation, no range checks
        const int cqx = (int)Bq.x, cqy = (int)Bq.y, cqz = (int)Bq.z;
        if (j > 0) {
          if (in_prev && in_q) {
            const int ddx = cqx - cpx, ddy = cqy - cpy, ddz = cqz - cpz;
            const bool nearby = ddx >= -1 && ddx <= 1 && ddy >= -1 && ddy <= 1 && ddz >= -1 && ddz <= 1;
            // start cell free in the dilated grid?  Its word was requested when that point was visited (below),
            // so the answer does not wait for a dependent load here
            const bool start_free = use_near && !((near_prev >> (((cpx & 3) << 4) | ((cpy & 3) << 2) | (cpz & 3))) & 1ull);
            need = !(nearby && start_free);                    // the walk itself is deferred (below)
            if (need) { seg_a = prevr; seg_b = qr; }
          } else {
            hit = line_hits(prevr, qr, g, gc, near, bad);      // near or outside the domain boundary: full reference path
          }
        }
        prevr = qr; in_prev = in_q; cpx = cqx; cpy = cqy; cpz = cqz;
        if (use_near && in_q) near_prev = near.blocks[((size_t)(cqx >> 2) * g.Nb + (cqy >> 2)) * g.Nb + (cqz >> 2)];
      }
      return need;
    };
    // Only ~3 % of the segments need the DDA walk, but with 64 independent configurations per wave some lane
    // needs one at almost every second point, and the wave would run the (long) walk code each time for a
    // lane or two.  The walks are queued instead -- (owner lane, end row) in LDS -- and run 64 at a time, one
    // per lane, on the owners' points re-read from memory and turned into voxel coordinates by the same
    // expressions; a hit comes back through the owner's LDS flag.  Verdicts do not depend on the order.
    int qhead = 0, qcount = 0;
    auto flush = [&]() {
      __syncthreads();
      const int cnt = qcount < 64 ? qcount : 64;
      if (lane < cnt) {
        const int slot = (qhead + lane) & 127;
        const int owner = (int)wq[slot];
        const V3 a = {wpts[slot], wpts[128 + slot], wpts[256 + slot]}, b = {wpts[384 + slot], wpts[512 + slot], wpts[640 + slot]};
        const V3 A = {(a.x - g.xmin) * g.inv_dx, (a.y - g.ymin) * g.inv_dy, (a.z - g.zmin) * g.inv_dz};
        const V3 B = {(b.x - g.xmin) * g.inv_dx, (b.y - g.ymin) * g.inv_dy, (b.z - g.zmin) * g.inv_dz};
        GridCursor wc{grid, g.Nb, -1, 0ull};
        if (walk_cells(A, B, g, [&](int x, int y, int z) { return wc.occupied(x, y, z); })) wflag[owner] = 1u;
      }
      qhead = (qhead + cnt) & 127;
      qcount -= cnt;
      __syncthreads();
      if (wflag[lane]) hit = true;
    };
    for (int j0 = 0; j0 < P; j0 += PF) {                 // rows; the lane's point index is row - shift
      if (!__any(alive && j0 < np + shift)) break;
#pragma unroll
      for (int u = 0; u < PF; u++) {
        const int j = j0 + u;
        const V3 q = ring[u];
        {
          const int jn = j + PF;
          const int64_t o = (int64_t)(jn < P ? jn : P - 1) * ld + ic;
          ring[u] = V3{in.px[o], in.py[o], in.pz[o]};
        }
        bool need = false;
        if (alive && j >= shift && j < np + shift) need = visit(j - shift, q);
        const unsigned long long wm = __ballot(need);
        if (wm) {
          if (need) {
            const int slot = (qhead + qcount + __popcll(wm & (((unsigned long long)1 << lane) - 1))) & 127;
            wq[slot] = (uint32_t)lane;
            wpts[slot] = seg_a.x; wpts[128 + slot] = seg_a.y; wpts[256 + slot] = seg_a.z;
            wpts[384 + slot] = seg_b.x; wpts[512 + slot] = seg_b.y; wpts[640 + slot] = seg_b.z;
          }
          qcount += __popcll(wm);
          if (qcount >= 64) flush();
        }
      }
    }
    while (qcount > 0) flush();
  }
  if (alive && !(dist < 1e30f)) { alive = false; bad = true; }   // NaN / inf points

  // Pass 2: milestone proof of "no self collision" (milestones_unresolved).
  bool need_exact = alive && np > 2;
  if (!(debug & 2u) && __any(need_exact)) need_exact = milestones_unresolved(mx, my, mz, ma, NM, Kl, need_exact, (float)K.radius);

  // Pass 3 (rare): exact pairwise sweep for the lanes pass 2 could not clear: accumulated chord
  // lengths in fp64 exactly as collision.cpp:21-30 forms them, then the pair loops.
  bool selfhit = false;
  if (__any(need_exact)) {
    double hmax = 0.0;
    if (need_exact) {
      double dd = 0.0;
      const int64_t ib = ic + (int64_t)shift * ld;       // the lane's rows start at `shift`
      V3 prev = {in.px[ib], in.py[ib], in.pz[ib]};
      for (int j = 0; j < np; j++) {
        const int64_t o = (int64_t)j * ld + ib;
        const V3 q = {in.px[o], in.py[o], in.pz[o]};
        const V3 d = {q.x - prev.x, q.y - prev.y, q.z - prev.z};
        const double h = sqrt(dot3(d, d));
        dd += h;
        hmax = fmax(hmax, h);
        in.acc[o] = dd;
        prev = q;
      }
    }
    selfhit = exact_self_collision(in, ic + (int64_t)shift * ld, ld, P, np, need_exact, hmax, K.radius, (debug & 1u) != 0);
  }

  uint32_t fl = 0;
  if (conv_ok) fl |= 1u;
  if (conv_ok && len_ok) fl |= 2u;
  bool valid = conv_ok && len_ok;
  if (valid && bad) { fl |= 16u; valid = false; }
  if (valid) { if (!selfhit) fl |= 4u; else valid = false; }
  if (valid && check_voxels == 1) { if (!hit) fl |= 8u; else valid = false; }
  if (valid && check_voxels == 2) {
    // list mode behind fk_verdict<.., SPH>: the sphere test has been made there; its answer is bit 8 of the flags it wrote
    // (without flags only configurations that passed it are listed)
    const bool passed = (out_map && flags && live) ? (flags[out_map[i]] & 8u) != 0 : true;
    if (passed) fl |= 8u; else valid = false;
  }

  if (lane_valid) { *lane_valid = valid && live; return; }   // the caller folds the verdict itself (edge_queue_kernel.hpp): nothing is written
  if (out_map) {
    // compacted list (the fallback pass of the verdict path): column i is configuration out_map[i]; its verdict bit is
    // still 0 in the mask, its flags are overwritten
    if (live) {
      const int64_t c = out_map[i];
      if (valid) atomicOr((unsigned long long *)&valid_bits[c >> 6], 1ull << (c & 63));
      if (flags) flags[c] = (uint8_t)fl;
    }
    return;
  }
  const uint64_t bits = __ballot(valid && live);
  if (lane == 0 && i < n) valid_bits[i >> 6] = bits;
  if (flags && live) flags[i] = (uint8_t)fl;
}

// The kernels below are defined once (tendon_hip.hip); the fused K1 + K2 objects only take the bodies above.
#ifndef TRK_DEVICE_BODIES_ONLY
__global__ __launch_bounds__(64) void backbone_voxel_sweep(
    SweepIn in, int64_t n, int64_t ld, int P, int CH, int NM, RobotK K, GridK g, const uint64_t *__restrict__ grid,
    const uint64_t *__restrict__ near_grid, int check_voxels, uint32_t debug, uint64_t *__restrict__ valid_bits,
    uint8_t *__restrict__ flags) {
  if (in.n_points) sweep_body<true>(in, n, ld, P, CH, NM, K, g, grid, near_grid, check_voxels, debug, valid_bits, flags);
  else sweep_body<false>(in, n, ld, P, CH, NM, K, g, grid, near_grid, check_voxels, debug, valid_bits, flags);
}

// K5 `backbone_voxelize`: the robot's own voxel set (what voxelize_impl returns,
// VoxelBackboneValidityChecker.h:49-57) as a list of (block id, 64-bit mask) per configuration -- the form roadmap
// voxel caches are stored in (VoxelCachedLazyPRM.cpp:2816-2823).  One lane per configuration; cells arrive mostly block
// by block, so the lane keeps the current AND the previous block in registers (a walk that hops back and forth over a
// block face costs nothing) and appends a block to its list (a column of ids / masks, [maxB][ld]) when the walk has left
// it for a third one.  A block the backbone RETURNS to later is appended again: the lists may hold a block more than
// once, and every consumer goes through the sort + reduce-by-key of cache_merge.hip (which ORs duplicates), so the
// delivered sets are duplicate-free and ordered by block id.  (The first version searched the lane's whole list in
// global memory on every append: 74 % of the kernel's wave cycles were waits, r02 profile.)
// counts[i] = entries written, or -1 when the list overflowed maxB or a point left the representable range.
__global__ __launch_bounds__(64) void backbone_voxelize(
    const double *__restrict__ px, const double *__restrict__ py, const double *__restrict__ pz,
    const int32_t *__restrict__ n_points, const uint64_t *__restrict__ shape_valid_bits, int64_t n, int64_t ld, int P,
    GridK g, int maxB, uint32_t *__restrict__ ids, uint64_t *__restrict__ masks, int32_t *__restrict__ counts) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  if (!((shape_valid_bits[i >> 6] >> (i & 63)) & 1ull)) { counts[i] = 0; return; }
  const int np = n_points ? n_points[i] : P;
  const int64_t ib = i + (int64_t)(P - np) * ld;         // retraction robots: the lane's point j is in row j + (P - np)
  int cnt = 0;
  bool overflow = false, bad = false;
  int cur_id = -1, prev_id = -1;
  uint64_t cur_mask = 0, prev_mask = 0;
  auto append = [&](int id, uint64_t m) {
    if (id < 0) return;
    if (cnt >= maxB) { overflow = true; return; }
    ids[(int64_t)cnt * ld + i] = (uint32_t)id;
    masks[(int64_t)cnt * ld + i] = m;
    cnt++;
  };
  auto set_cell = [&](int x, int y, int z) {
    const int id = ((x >> 2) * g.Nb + (y >> 2)) * g.Nb + (z >> 2);
    const uint64_t bit = 1ull << (((x & 3) << 4) | ((y & 3) << 2) | (z & 3));
    if (id != cur_id) {
      if (id == prev_id) { const int t = cur_id; cur_id = prev_id; prev_id = t; const uint64_t tm = cur_mask; cur_mask = prev_mask; prev_mask = tm; }
      else { append(prev_id, prev_mask); prev_id = cur_id; prev_mask = cur_mask; cur_id = id; cur_mask = 0; }
    }
    cur_mask |= bit;
    return false;
  };
  V3 prev = {0, 0, 0};
  for (int j = 0; j < np; j++) {
    const int64_t o = (int64_t)j * ld + ib;
    const double x = px[o], y = py[o], z = pz[o];
    V3 q;
    if (g.rot_is_identity) { q = V3{x, y, z}; }
    else {
      q.x = g.inv_rot[0] * x + g.inv_rot[1] * y + g.inv_rot[2] * z;
      q.y = g.inv_rot[3] * x + g.inv_rot[4] * y + g.inv_rot[5] * z;
      q.z = g.inv_rot[6] * x + g.inv_rot[7] * y + g.inv_rot[8] * z;
    }
    if (j > 0) {
      V3 A, B;
      bool inside;
      if (line_setup(prev, q, g, A, B, inside, bad)) walk_cells(A, B, g, set_cell);
    }
    prev = q;
  }
  append(prev_id, prev_mask);
  append(cur_id, cur_mask);
  counts[i] = (overflow || bad) ? -1 : cnt;
}

// n 8-byte words from `src` to `dst`: upload_staged's copy out of pinned host memory (tendon_hip.hip), grid-stride
__global__ __launch_bounds__(256) void copy_words(uint64_t *__restrict__ dst, const uint64_t *__restrict__ src, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void iota_i32(int32_t *__restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (int32_t)i;
}

// Obstacle grid dilated by 2 cells in the Chebyshev metric (a cell is set iff some occupied cell
// lies within +-2 along every axis), same block / bit layout; feeds the fast path of line_hits.
// One wave per block, lane = bit index (x*16 + y*4 + z); the 64 verdicts are packed with a ballot.
__global__ __launch_bounds__(64) void dilate2_blocks(const uint64_t *__restrict__ grid, uint64_t *__restrict__ out, int Nb) {
  const int b = blockIdx.x;
  const int bz = b % Nb, by = (b / Nb) % Nb, bx = b / (Nb * Nb);
  const int lane = threadIdx.x;
  // quick reject: all 27 neighbouring blocks empty
  bool nz = false;
  if (lane < 27) {
    const int qx = bx + lane / 9 - 1, qy = by + (lane / 3) % 3 - 1, qz = bz + lane % 3 - 1;
    if (qx >= 0 && qx < Nb && qy >= 0 && qy < Nb && qz >= 0 && qz < Nb) nz = grid[((size_t)qx * Nb + qy) * Nb + qz] != 0;
  }
  if (!__any(nz)) { if (lane == 0) out[b] = 0; return; }
  const int N = 4 * Nb;
  const int X = 4 * bx + (lane >> 4), Y = 4 * by + ((lane >> 2) & 3), Z = 4 * bz + (lane & 3);
  bool hit = false;
  for (int dx = -2; dx <= 2 && !hit; dx++) {
    const int x = X + dx;
    if (x < 0 || x >= N) continue;
    for (int dy = -2; dy <= 2 && !hit; dy++) {
      const int y = Y + dy;
      if (y < 0 || y >= N) continue;
      for (int dz = -2; dz <= 2; dz++) {
        const int z = Z + dz;
        if (z < 0 || z >= N) continue;
        const uint64_t v = grid[((size_t)(x >> 2) * Nb + (y >> 2)) * Nb + (z >> 2)];
        if ((v >> (((x & 3) << 4) | ((y & 3) << 2) | (z & 3))) & 1ull) { hit = true; break; }
      }
    }
  }
  const uint64_t m = __ballot(hit);
  if (lane == 0) out[b] = m;
}

// K4 `cached_blocks_vs_grid`: sparse cached voxel sets (CSR of (block id, mask)) vs the dense grid:
// `obstacles.collides(*cached_voxels)` of VoxelCachedLazyPRM.cpp:2397-2411 for every roadmap item.
// HBM-streaming kernel, bound by the loads it keeps in flight: a roadmap item holds ~40 - 55 blocks, less than one
// wave-wide iteration, and testing it is a chain of dependent round trips (offsets -> entries -> grid word).  A wave takes
// FOUR items at once, 16 lanes each, four entries per lane in flight (a typical item's whole list in one round of loads),
// and is a workgroup of its own: with 16 waves sharing one output word and one offsets fetch the waves of a CU moved
// through their three phases in step (0.42 of the HBM peak); independent waves spread over them (0.50).  A hit sets the
// item's bit with an atomic (hit_bits is zeroed by the host).  Measured alternatives, all slower on real caches: one item
// per wave (0.27), a flat stream of entries with deferred item lookup, and persistent waves that prefetch the
// next group's entries and the offsets of the one after (0.47: the early exit at an item's first hit wastes the prefetch).
#ifndef K4_INFLIGHT
#define K4_INFLIGHT 4
#endif
__global__ __launch_bounds__(64) void cached_blocks_vs_grid(
    const uint32_t *__restrict__ ids, const uint64_t *__restrict__ masks, const int64_t *__restrict__ offsets,
    int64_t n_items, const uint64_t *__restrict__ grid, uint32_t n_blocks, uint64_t *__restrict__ hit_bits) {
  const int lane = threadIdx.x, grp = lane >> 4, sub = lane & 15;
  const int64_t item = (int64_t)blockIdx.x * 4 + grp;
  bool hit = false;
  if (item < n_items) {
    const int64_t e = offsets[item + 1];
    for (int64_t k = offsets[item] + sub; k < e && !hit; k += 16 * K4_INFLIGHT) {
      uint32_t id[K4_INFLIGHT];
      uint64_t m[K4_INFLIGHT], g[K4_INFLIGHT];
#pragma unroll
      for (int u = 0; u < K4_INFLIGHT; u++) {
        const int64_t ku = k + 16 * u;
        const bool in = ku < e;
        id[u] = in ? ids[ku] : 0xffffffffu;
        m[u] = in ? masks[ku] : 0ull;
      }
#pragma unroll
      for (int u = 0; u < K4_INFLIGHT; u++) g[u] = id[u] < n_blocks ? grid[id[u]] : 0ull;
      uint64_t any = 0;
#pragma unroll
      for (int u = 0; u < K4_INFLIGHT; u++) any |= g[u] & m[u];
      hit = any != 0;
    }
  }
  const unsigned long long bal = __ballot(hit);
  if (sub == 0 && ((bal >> (grp * 16)) & 0xffffull)) atomicOr((unsigned long long *)&hit_bits[item >> 6], 1ull << (item & 63));
}

// K4 on a SUBSET of the cached items: list[q] names an item of the CSR; its verdict goes to hit[q] (one byte).  The lazy query loop (roadmap.hip) validates the unknown vertices / edges of all
// candidate paths of a round with one launch of this.
__global__ __launch_bounds__(256) void cached_subset_vs_grid(
    const uint32_t *__restrict__ ids, const uint64_t *__restrict__ masks, const int64_t *__restrict__ offsets,
    const int32_t *__restrict__ list, int64_t n_list, int64_t n_items, const uint64_t *__restrict__ grid, uint32_t n_blocks,
    uint8_t *__restrict__ hit_out) {
  const int lane = threadIdx.x & 63, grp = lane >> 4, sub = lane & 15;
  const int64_t q = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + grp;       // four listed items per wave, 16 lanes each
  bool hit = false;
  if (q < n_list) {
    const int64_t item = list[q];
    if (item >= 0 && item < n_items) {
      const int64_t e = offsets[item + 1];
      for (int64_t k = offsets[item] + sub; k < e && !hit; k += 16) {
        const uint32_t id = ids[k];
        hit = id < n_blocks && (grid[id] & masks[k]) != 0;
      }
    }
  }
  const unsigned long long bal = __ballot(hit);
  if (sub == 0 && q < n_list) hit_out[q] = ((bal >> (grp * 16)) & 0xffffull) ? 1 : 0;
}

#endif  // TRK_DEVICE_BODIES_ONLY

}  // namespace trk
