// verdict_kernel.hpp -- `fk_verdict`: the whole validity predicate of a configuration in ONE pass, without ever storing
// its backbone: every point leaves the RK4 loop straight into the sweep (what K2 `backbone_voxel_sweep` does on stored
// points, sweep_kernel.hpp) and only the verdict bit (+ the tip, + optional flags) reaches memory -- 24 B in, ~24 B
// out per configuration instead of the 6 274 B the K1 -> K2 pair moves through HBM, and no 4 GB point workspace.
//
// Per observed point, between two RK4 steps (the integrator's temporaries are dead there, so this costs the hot loop
// two live registers -- the prefetched dilated-grid word -- and nothing else; all other per-lane sweep state sits in LDS):
//   * chord length in float and, every CH-th point, a milestone (x, y, z, arc) for the self-collision proof;
//   * the point's cell; a segment whose end cells differ by at most one per axis and whose START cell is free in the
//     2-cell-dilated grid cannot touch an occupied cell (see line_hits) and is done;
//   * every other segment (~3 %: larger cell steps, near obstacles, at the domain boundary) is queued in LDS with both
//     of its end points, and the queue is worked off 64 at a time, one add_line walk per lane (walk_cells: the
//     reference's DDA, bit-exact), a hit returning through the owner lane's LDS flag.
// After the loop: the milestone proof of "no self collision" (milestones_unresolved).  The few configurations it
// cannot clear need the exact pairwise sweep over ALL their points, which no longer exist: their indices go to a
// compacted list and `fk_sweep_fused_list` (fused_kernel.hpp) integrates just those again with stored points and ORs
// their bits into the mask.  Every decision is formed by the same expressions as in sweep_body, so the verdict bits and
// flags are identical to those of the separate kernels.
#pragma once
#include "fk_kernel.hpp"
#define TRK_DEVICE_BODIES_ONLY
#include "sweep_kernel.hpp"
#include "sphere_kernel.hpp"

namespace trk {

// What EVERY observed point reads, in one 128-byte block at the head of the argument block: two wide scalar loads per
// point instead of a dozen narrow ones with a wait each (finish_hot() fills it from the fields below, on the host).
struct VerdictHot {
  double box[6];                  // VerdictArgs::box
  double org[3], inv_d[3];        // g.xmin, g.ymin, g.zmin; g.inv_dx, g.inv_dy, g.inv_dz
  const uint64_t *near_grid;      // always a readable grid of Nb^3 words (the obstacle grid itself when use_near is 0)
  int32_t Nb, rot_is_identity, P, CH;
  int32_t use_near;               // 0: no dilated grid, or debug bit 2 (the fast path of the voxel walk is off)
  int32_t pad_;
};
static_assert(sizeof(VerdictHot) == 128, "one 128-byte block");

struct alignas(128) VerdictArgs {
  VerdictHot hot;
  int P, CH, NM, pad0_;
  uint32_t debug, pad1_;
  double box[6];                  // the margin box of sweep_body: {x0, x1, y0, y1, z0, z1} = limits -+ 1e-6 of the extent (same expressions, formed on the host)
  GridK g;
  const uint64_t *grid, *near_grid;
  uint64_t *valid_bits;
  uint8_t *flags;                 // optional
  int32_t *fb_list;               // configurations that need the exact self-collision sweep
  uint32_t *fb_count;
  uint32_t *sig;                  // optional (edge samples): cell signatures [n][sig_stride], see sweep_kernel.hpp
  int64_t sig_stride;
  // sphere-swept checker (fk_verdict<.., SPH = true>, sphere_kernel.hpp): distance field of the obstacle cells and the two
  // thresholds of its classification: radius -+ 2e-6 m of slack for the float arithmetic; the distance of the point from
  // its cell's centre, which K8 bounds by half a cell diagonal, is taken per point here (a thinner undecided shell)
  const float *field;
  float r_lo, r_hi;
  double radius;
  // retraction robots (fk_verdict_retract): lane i of the launch takes configuration perm[i] -- the batch ordered by
  // backbone length (cache_merge.hpp: retraction_order) -- and its outputs go to that configuration's places: the verdict
  // bit by an atomic OR into the zeroed mask; null = arrival order, bits by ballot
  const int32_t *perm;
  int32_t *np_out;                // optional (edge samples): the configurations' point counts, for the comparison of tip-aligned signature rows
  // with perm: per wave of the ordered batch, a step index before which none of its 64 backbones has begun (from the longest
  // backbone its retraction level allows, retraction_order): the wave enters the tip-aligned loop there instead of stepping
  // over ~100 table entries when it was dealt short backbones
  const int32_t *wave_k_begin;
  // the robot's length limits and home lengths (RobotK's): read here after the loop, so that the kernel's RobotK argument
  // keeps nothing but the stiffness constants live across it
  double home_Li[TRK_MAX_TENDONS], min_len[TRK_MAX_TENDONS], max_len[TRK_MAX_TENDONS];

  __host__ void finish_hot() {
    for (int q = 0; q < 6; q++) hot.box[q] = box[q];
    hot.org[0] = g.xmin; hot.org[1] = g.ymin; hot.org[2] = g.zmin;
    hot.inv_d[0] = g.inv_dx; hot.inv_d[1] = g.inv_dy; hot.inv_d[2] = g.inv_dz;
    hot.use_near = (near_grid != nullptr && !(debug & 4u)) ? 1 : 0;
    hot.near_grid = hot.use_near ? near_grid : grid;
    hot.Nb = g.Nb; hot.rot_is_identity = g.rot_is_identity; hot.P = P; hot.CH = CH;
  }
};

// Dynamic LDS of one wave.  Everything is addressed as vlds[constant + lane] (never through a stored pointer: hipcc then
// keeps the accesses in the LDS address space -- ds_read / ds_write -- instead of falling back to flat accesses).
//   doubles [0, 192)           prev[3][64]     previous point (as produced, before the environment rotation)
//   doubles [192, 192 + 6 VQ)  qe[6][VQ]       deferred segments: rotated end points a, b
//   words from 2 * (192 + 6 VQ):  qowner[VQ] | cell[3][64] | inprev[64] | hitflag[64] | dist[64] | delta[64] (SPH) / nearw[64] | milestones float[4][NM][64]
// VQ, the ring of deferred segments, is flushed 64 at a time as soon as VQ - 64 are queued (a point can queue 64 more): 128 entries and
// a flush per 64 for the plain kernels.  With signatures (edge samples) the image also holds the signature tile of SigStage
// (sweep_kernel.hpp) behind the milestones, and at 128 entries it was exactly 20 KiB at 8 milestones -- an eighth of a CU's LDS,
// with no room for the 2 KiB of the 4-tendon robots' tendon-length quadratures (fk_kernel.hpp: li_in_lds), which stayed in 38
// spilled registers (104 B of scratch per lane, 2.1x the algorithmic HBM traffic).  Round 4: 88 entries there (a flush per 24
// deferred segments: ~10 instead of ~4 per backbone, each a DDA walk of the queued lanes, < 0.5 % of the instructions) and the
// owners as bytes: 18 136 B + 2 048 B of quadratures = 19.7 KiB.
extern __shared__ double vlds[];
#ifndef TRK_SIG_VQ
#define TRK_SIG_VQ 88            // (A/B: 128 = round 3's layout, the quadratures of the signature kernels then stay in registers)
#endif
template <bool SIG>
struct VLay {
  static constexpr int VQ = SIG ? TRK_SIG_VQ : 128;
  static constexpr int QE = 3 * 64;                                         // doubles
  static constexpr int W0 = 2 * (QE + 6 * VQ);                              // words from here on
  static constexpr int QOWNER = W0, OWNER_WORDS = SIG ? (VQ + 3) / 4 : VQ;  // owner lanes: bytes with signatures, words without
  static constexpr int CELL = QOWNER + OWNER_WORDS, INPREV = CELL + 3 * 64, HIT = INPREV + 64, DIST = HIT + 64, DELTA = DIST + 64, MS = DELTA + 64;
  // voxel checker: delta[] is unused and holds the requested half of the dilated-grid word of the previous point's block instead
  static constexpr int NEARW = DELTA;
  __device__ static __forceinline__ int wrap(int i) {                       // ring index, i < 2 VQ
    if constexpr ((VQ & (VQ - 1)) == 0) return i & (VQ - 1); else return i >= VQ ? i - VQ : i;
  }
  __device__ static __forceinline__ void set_owner(int slot, int lane) {
    if constexpr (SIG) ((uint8_t *)vlds)[4 * QOWNER + slot] = (uint8_t)lane; else ((uint32_t *)vlds)[QOWNER + slot] = (uint32_t)lane;
  }
  __device__ static __forceinline__ int owner(int slot) {
    if constexpr (SIG) return (int)((const uint8_t *)vlds)[4 * QOWNER + slot]; else return (int)((const uint32_t *)vlds)[QOWNER + slot];
  }
};
__host__ __device__ inline size_t verdict_lds_bytes(int NM, bool with_sig = false) {
  return (size_t)(with_sig ? VLay<true>::MS : VLay<false>::MS) * 4 + (size_t)4 * NM * 64 * sizeof(float) + (with_sig ? (size_t)SIG_LDS_WORDS * 4 : 0);
}

// (plain accesses: hipcc does not move `volatile` ones into the LDS address space; the compiler barriers at both ends of
// the per-point code keep it from promoting this state to registers across the RK4 loop)
#define VL_D(i) (((double *)vlds)[(i)])
#define VL_U(i) (((uint32_t *)vlds)[(i)])
#define VL_I(i) (((int32_t *)vlds)[(i)])
#define VL_F(i) (((float *)vlds)[(i)])
#define VL_BARRIER() asm volatile("" ::: "memory")

// The per-point sweep: what sweep_body's pass 1 does with a stored point, done when the point is produced.
// SIG: the launch also writes the samples' cell signatures (edge samples); a compile-time switch so that the plain
// verdict kernels carry none of it.
template <bool SPH, bool SIG = false>
struct PointSweep {
  using Lay = VLay<SIG>;
  static constexpr int VQ = Lay::VQ;
  // fk_uniform_body keeps the 4-tendon robots' tendon-length quadratures in LDS when this hook sits in its loop (fk_kernel.hpp:
  // li_in_lds); with signatures too since round 4 made the room (VLay).  (3 tendons: 2 spilled dwords, nothing to gain)
  static constexpr int kLiInLdsFrom = (SIG && TRK_SIG_VQ > 88) ? (1 << 30) : 4;
  const VerdictArgs *va;
  float dn_prev;                  // SPH: distance-field value at the previous point's cell (requested one point ahead)
  uint32_t sph_state;             // SPH: bit 0 = the previous point awaits its classification, bit 1 = it lies inside the closed domain
  int qhead, qcount;              // wave-uniform
  int P, CH, NM, Kl, ms_next, ms_k;   // wave-uniform: point count, milestone spacing / count, last milestone, next milestone row / index
  SigStage sigst;                 // ... through this LDS tile (words [Lay::MS + 4 NM 64, + SIG_LDS_WORDS) of the wave's image)
  // the sample whose row this lane writes (-1: none) and its first point that goes through the tile: lane-private for the
  // retraction kernel (a permuted batch, the first two points in rows of the lane's own); the shared-grid kernel forms them
  // from sig_n on the fly
  int sig_row_of, sig_first_row;
  int sig_row0, sig_cnt;          // wave-uniform (shared-grid kernels): lane l < sig_cnt writes signature row sig_row0 + l -- the wave's 64
                                  // configurations of a launch, or the pool slots a wave of the edge queue has claimed (edge_queue_kernel.hpp)
  bool active;                    // live && converged: only these lanes test voxels

  __device__ __forceinline__ int sig_own_row() const { const int l = (int)threadIdx.x; return l < sig_cnt ? sig_row0 + l : -1; }
  // the rows of the wave's 64 lanes in a launch over n configurations, one lane each
  __device__ __forceinline__ void sig_rows_of_launch(int64_t n) {
    const int64_t r0 = (int64_t)blockIdx.x * 64, left = n - r0;
    sig_row0 = (int)r0; sig_cnt = left >= 64 ? 64 : (left > 0 ? (int)left : 0);
  }
  __device__ __forceinline__ void sig_put(const VerdictArgs &a, int row, bool on, uint32_t value, int row_of, int first_row) {
    const int base = Lay::MS + 4 * NM * 64;
    sigst.put([base](int i) -> uint32_t & { return VL_U(base + i); }, a.sig, a.sig_stride, row_of, first_row, row, on, value);
  }
  template <bool RETRACT>
  __device__ __forceinline__ void sig_finish() {
    if constexpr (SIG) {
      const VerdictArgs &a = args();
      const int base = Lay::MS + 4 * NM * 64;
      sigst.flush([base](int i) -> uint32_t & { return VL_U(base + i); }, a.sig, a.sig_stride, RETRACT ? sig_row_of : sig_own_row(),
                  RETRACT ? sig_first_row : 0);
    }
  }

  __device__ __forceinline__ void begin(bool on) { active = on; }

  __device__ __forceinline__ const VerdictArgs &args() const {
    // index the argument block with a zero the optimiser cannot see through: the address is then not loop-invariant, so
    // the arguments are re-read through the scalar cache when a point is swept (s_load: the pointer still is the kernel's
    // read-only argument) instead of occupying ~60 SGPRs across the RK4 loop
    int zero = 0;
    asm volatile("" : "+s"(zero));
    return va[zero];
  }

  __device__ __forceinline__ void flush() {
#pragma clang fp contract(off)
    const VerdictArgs &a = args();
    const int lane = threadIdx.x;
    __syncthreads();
    const int cnt = qcount < 64 ? qcount : 64;
    if constexpr (SPH) {
      // one queued point after the other, each served by the whole wave (as K8 serves its flagged lanes)
      const GridK &g = a.g;
      const double radius = a.radius;
      for (int e = 0; e < cnt; e++) {
        const int slot = Lay::wrap(qhead + e);
        const double sx = VL_D(Lay::QE + 0 * VQ + slot), sy = VL_D(Lay::QE + 1 * VQ + slot), sz = VL_D(Lay::QE + 2 * VQ + slot);
        const int owner = Lay::owner(slot);
        bool found = false;
        {
          // add_point (VoxelOctree.cpp:319-323): the point's own cell, if the point is inside the closed domain
          const double cx = fmin(fmax(sx, g.xmin), g.xmax), cy = fmin(fmax(sy, g.ymin), g.ymax), cz = fmin(fmax(sz, g.zmin), g.zmax);
          if (cx == sx && cy == sy && cz == sz) {
            const int N = g.N;
            int ix = (int)((cx - g.xmin) / g.dx), iy = (int)((cy - g.ymin) / g.dy), iz = (int)((cz - g.zmin) / g.dz);
            ix = ix < 0 ? 0 : (ix > N - 1 ? N - 1 : ix); iy = iy < 0 ? 0 : (iy > N - 1 ? N - 1 : iy); iz = iz < 0 ? 0 : (iz > N - 1 ? N - 1 : iz);
            found = ((a.grid[((size_t)(ix >> 2) * g.Nb + (iy >> 2)) * g.Nb + (iz >> 2)] >> (((ix & 3) << 4) | ((iy & 3) << 2) | (iz & 3))) & 1ull) != 0;
          }
        }
        if (!found) found = sphere_scan_wave(sx, sy, sz, radius, g, a.grid, lane);
        if (found && lane == 0) atomicOr(&((uint32_t *)vlds)[Lay::HIT + owner], 1u);
      }
    } else
    if (lane < cnt) {
      const int slot = Lay::wrap(qhead + lane);
      const V3 pa = {VL_D(Lay::QE + 0 * VQ + slot), VL_D(Lay::QE + 1 * VQ + slot), VL_D(Lay::QE + 2 * VQ + slot)};
      const V3 pb = {VL_D(Lay::QE + 3 * VQ + slot), VL_D(Lay::QE + 4 * VQ + slot), VL_D(Lay::QE + 5 * VQ + slot)};
      const int owner = Lay::owner(slot);
      V3 A, B;
      bool inside, bad = false, h = false;
      if (line_setup(pa, pb, a.g, A, B, inside, bad)) {
        GridCursor wc{a.grid, a.g.Nb, -1, 0ull};
        h = walk_cells(A, B, a.g, [&](int x, int y, int z) { return wc.occupied(x, y, z); });
      }
      if (h || bad) atomicOr(&((uint32_t *)vlds)[Lay::HIT + owner], (h ? 1u : 0u) | (bad ? 2u : 0u));
    }
    qhead = Lay::wrap(qhead + cnt);
    qcount -= cnt;
    __syncthreads();
  }

  // SPH: ask the distance field about point q (as produced; rotated into the voxel frame here): the cell of its projection
  // onto the domain (nearest_cell, VoxelOctree.cpp:295-307), as K8 forms it
  __device__ __forceinline__ void request(const VerdictArgs &a, const V3 &q) {
#pragma clang fp contract(off)
    const GridK &g = a.g;
    V3 r = q;
    if (!g.rot_is_identity) {
      r.x = g.inv_rot[0] * q.x + g.inv_rot[1] * q.y + g.inv_rot[2] * q.z;
      r.y = g.inv_rot[3] * q.x + g.inv_rot[4] * q.y + g.inv_rot[5] * q.z;
      r.z = g.inv_rot[6] * q.x + g.inv_rot[7] * q.y + g.inv_rot[8] * q.z;
    }
    if (!(fabs(r.x) < 1e300) || !(fabs(r.y) < 1e300) || !(fabs(r.z) < 1e300)) return;     // non-finite: the chord length settles it
    const int N = g.N;
    const double cx = fmin(fmax(r.x, g.xmin), g.xmax), cy = fmin(fmax(r.y, g.ymin), g.ymax), cz = fmin(fmax(r.z, g.zmin), g.zmax);
    int ix = (int)((cx - g.xmin) / g.dx), iy = (int)((cy - g.ymin) / g.dy), iz = (int)((cz - g.zmin) / g.dz);
    ix = ix < 0 ? 0 : (ix > N - 1 ? N - 1 : ix); iy = iy < 0 ? 0 : (iy > N - 1 ? N - 1 : iy); iz = iz < 0 ? 0 : (iz > N - 1 ? N - 1 : iz);
    dn_prev = a.field[((size_t)ix * N + iy) * N + iz];
    sph_state = 1u | ((cx == r.x && cy == r.y && cz == r.z) ? 2u : 0u);
    // how far the looked-up position (the point, or its projection onto the domain) is from the centre of its cell: an
    // occupied centre is at least field - delta and -- the nearest one -- at most field + delta away from it
    const float ex = (float)(cx - (g.xmin + g.dx * ((double)ix + 0.5))), ey = (float)(cy - (g.ymin + g.dy * ((double)iy + 0.5))),
                ez = (float)(cz - (g.zmin + g.dz * ((double)iz + 0.5)));
    VL_F(Lay::DELTA + threadIdx.x) = sqrtf(ex * ex + ey * ey + ez * ez) * 1.000001f;
  }

  // after the RK4 loop.  SPH: the last point's classification (its field value was requested when it was produced)
  __device__ __forceinline__ void finish() {
#pragma clang fp contract(off)
    if constexpr (SPH) {
      VL_BARRIER();
      const VerdictArgs &a = args();
      const GridK &g = a.g;
      const int lane = threadIdx.x;
      bool need = false;
      V3 pr = {VL_D(lane), VL_D(64 + lane), VL_D(128 + lane)};
      if (active && !VL_U(Lay::HIT + lane) && (sph_state & 1u)) {
        if (!g.rot_is_identity) {
          const V3 pv = pr;
          pr.x = g.inv_rot[0] * pv.x + g.inv_rot[1] * pv.y + g.inv_rot[2] * pv.z;
          pr.y = g.inv_rot[3] * pv.x + g.inv_rot[4] * pv.y + g.inv_rot[5] * pv.z;
          pr.z = g.inv_rot[6] * pv.x + g.inv_rot[7] * pv.y + g.inv_rot[8] * pv.z;
        }
        const float dl = VL_F(Lay::DELTA + lane);
        if (dn_prev - dl > a.r_hi) {}
        else if ((sph_state & 2u) && dn_prev + dl < a.r_lo) VL_U(Lay::HIT + lane) = 1u;
        else need = true;
      }
      sph_state = 0;
      const unsigned long long wm = __ballot(need);
      if (wm) {
        if (qcount + __popcll(wm) > VQ) flush();
        if (need) {
          const int slot = Lay::wrap(qhead + qcount + __popcll(wm & (((unsigned long long)1 << lane) - 1)));
          VL_D(Lay::QE + 0 * VQ + slot) = pr.x; VL_D(Lay::QE + 1 * VQ + slot) = pr.y; VL_D(Lay::QE + 2 * VQ + slot) = pr.z;
          Lay::set_owner(slot, lane);
        }
        qcount += __popcll(wm);
      }
      VL_BARRIER();
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the last point's request (voxel_test) must have landed in LDS before the wave moves on
    }
  }

  // Point q in the voxel frame (qr), whether it lies strictly inside the margin box, its voxel coordinates t and its cell c by
  // truncation (inside the box the coordinates are in (0, N): floor == truncation, no range checks).  Straight-line code from
  // the hot block -- no short-circuit chains, whose every link costs a scalar load and a wait.  For ALL lanes: the cell also
  // serves the point's signature.
  __device__ __forceinline__ void point_cell(const VerdictArgs &a, const V3 &q, V3 &qr, bool &in_q, int (&c)[3], double (&t)[3]) const {
#pragma clang fp contract(off)
    const VerdictHot &h = a.hot;
    qr = q;
    if (!h.rot_is_identity) {
      const GridK &g = a.g;
      qr.x = g.inv_rot[0] * q.x + g.inv_rot[1] * q.y + g.inv_rot[2] * q.z;
      qr.y = g.inv_rot[3] * q.x + g.inv_rot[4] * q.y + g.inv_rot[5] * q.z;
      qr.z = g.inv_rot[6] * q.x + g.inv_rot[7] * q.y + g.inv_rot[8] * q.z;
    }
    in_q = (qr.x > h.box[0]) & (qr.x < h.box[1]) & (qr.y > h.box[2]) & (qr.y < h.box[3]) & (qr.z > h.box[4]) & (qr.z < h.box[5]);
    t[0] = (qr.x - h.org[0]) * h.inv_d[0]; t[1] = (qr.y - h.org[1]) * h.inv_d[1]; t[2] = (qr.z - h.org[2]) * h.inv_d[2];
    c[0] = (int)t[0]; c[1] = (int)t[1]; c[2] = (int)t[2];
  }

  // cell_signature(x, y, z) from what point_cell has formed: inside the margin box the point is inside the closed domain and
  // finite, t is cell_signature's product with the rounded reciprocal and c its truncation -- the same fast path, decided by
  // the same test (no voxel coordinate within 1e-9 of an integer); every other point takes the function itself.
  __device__ __forceinline__ uint32_t signature_of(const VerdictArgs &a, bool in_q, const int (&c)[3], const double (&t)[3],
                                                   double x, double y, double z) const {
#pragma clang fp contract(off)
    const double fx = t[0] - (double)c[0], fy = t[1] - (double)c[1], fz = t[2] - (double)c[2];
    const double lo = 1e-9, hi = 1.0 - 1e-9;
    const bool plain = in_q & (fx > lo) & (fx < hi) & (fy > lo) & (fy < hi) & (fz > lo) & (fz < hi);
    uint32_t sg = ((uint32_t)c[0] & 1023u) | (((uint32_t)c[1] & 1023u) << 10) | (((uint32_t)c[2] & 1023u) << 20);
    if (!plain) sg = cell_signature(x, y, z, a.g);
    return sg;
  }

  // The voxel test of the segment that ends in the point of cell c: "the end cells differ by at most one per axis and the
  // START cell is free in the dilated grid" -- then the segment cannot touch an occupied cell -- and the request for the
  // dilated word of c's block (consumed one point later).  Returns true when the segment needs the full reference walk
  // (queued by the caller).  `first`: the lane's point 0 (nothing to test; the LDS words read are then stale and unused).
  __device__ __forceinline__ bool voxel_test(const VerdictHot &h, bool first, bool in_q, const int (&c)[3]) {
    const int lane = threadIdx.x;
    const int cqx = c[0], cqy = c[1], cqz = c[2];
    const bool use_near = h.use_near != 0;
    const uint32_t was_in = VL_U(Lay::INPREV + lane);
    const int cpx = VL_I(Lay::CELL + lane), cpy = VL_I(Lay::CELL + 64 + lane), cpz = VL_I(Lay::CELL + 128 + lane);
    const uint32_t ex = (uint32_t)(cqx - cpx + 1), ey = (uint32_t)(cqy - cpy + 1), ez = (uint32_t)(cqz - cpz + 1);   // 0, 1, 2: neighbours
    const uint32_t emax = ex > ey ? (ex > ez ? ex : ez) : (ey > ez ? ey : ez);
    // the half of the dilated-grid word of the previous point's block that holds its cell's bit: requested when that point
    // was produced, delivered straight into LDS (no register lives across the RK4 step for it)
    const uint32_t bit = (uint32_t)(((cpx & 1) << 4) | ((cpy & 3) << 2) | (cpz & 3));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (the compiler does not order LDS reads after loads into LDS; a whole RK4 step has passed)
    uint32_t near_bit = (VL_U(Lay::NEARW + lane) >> bit) & 1u;
    const bool start_free = use_near & (near_bit == 0u);
    // otherwise (also at the domain boundary): the full reference path, in the flush
    const bool need = !first & !((was_in != 0u) & in_q & (emax <= 2u) & start_free);
    VL_I(Lay::CELL + lane) = cqx; VL_I(Lay::CELL + 64 + lane) = cqy; VL_I(Lay::CELL + 128 + lane) = cqz;
    VL_U(Lay::INPREV + lane) = in_q ? 1u : 0u;
    {
      // request the half word of q's block for the next point: a global load whose destination is LDS word nearw[lane]
      // (global_load_lds_dword; counted by vmcnt, no destination register), half word 0 for lanes outside the box, whose
      // next segment takes the full path anyway.  The empty asm orders it after the read of the old word above.
      const uint32_t nb = (uint32_t)h.Nb;
      uint32_t w = (((uint32_t)cqx >> 2) * nb + ((uint32_t)cqy >> 2)) * nb + ((uint32_t)cqz >> 2);
      w = in_q ? 2u * w + (((uint32_t)cqx >> 1) & 1u) : 0u;
      asm volatile("" : "+v"(w) : "v"(near_bit));
      typedef __attribute__((address_space(3))) uint32_t *lds_u32_ptr;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) uint32_t *)h.near_grid + w,
                                       (lds_u32_ptr)((uint32_t *)vlds + Lay::NEARW), 4, 0, 0);
    }
    return need;
  }

  // the previous point in the voxel frame, for a segment that is queued (rare: the rotation is not spent on every point)
  __device__ __forceinline__ V3 to_voxel_frame(const VerdictArgs &a, const V3 &p) const {
#pragma clang fp contract(off)
    if (a.hot.rot_is_identity) return p;
    const GridK &g = a.g;
    return V3{g.inv_rot[0] * p.x + g.inv_rot[1] * p.y + g.inv_rot[2] * p.z, g.inv_rot[3] * p.x + g.inv_rot[4] * p.y + g.inv_rot[5] * p.z,
              g.inv_rot[6] * p.x + g.inv_rot[7] * p.y + g.inv_rot[8] * p.z};
  }

  // point j of the lane's backbone (j = 0 .. P-1, in order)
  __device__ __forceinline__ void operator()(int j, double x, double y, double z) {
#pragma clang fp contract(off)
    VL_BARRIER();
    const VerdictArgs &a = args();
    const GridK &g = a.g;
    const int lane = threadIdx.x;
    const V3 q = {x, y, z};
    V3 pv = q;
    float d = 0.0f;
    if (j > 0) { pv = V3{VL_D(lane), VL_D(64 + lane), VL_D(128 + lane)}; d = VL_F(Lay::DIST + lane); }
    {
      const float dx = (float)(q.x - pv.x), dy = (float)(q.y - pv.y), dz = (float)(q.z - pv.z);
      d += __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz);   // v_sqrt_f32 (1 ulp): the proof keeps 1e-6 m of slack over ~1e-8 m of rounding
      VL_F(Lay::DIST + lane) = d;
    }
    if (j == ms_next) {                                     // wave-uniform: every CH-th point and the tip (ms_next = min(k CH, P - 1))
      const int CH_ = a.hot.CH, P_ = a.hot.P;               // (from the argument block: they hold no registers between milestones)
      const int o = Lay::MS + ms_k * 64 + lane, pl = a.NM * 64;     // the tip's slot: ceil((P - 1) / CH) = the count of milestones before it
      VL_F(o) = (float)q.x; VL_F(o + pl) = (float)q.y; VL_F(o + 2 * pl) = (float)q.z; VL_F(o + 3 * pl) = d;
      ms_next = (ms_next + CH_ < P_ - 1) ? ms_next + CH_ : P_ - 1;
      ms_k++;
    }
    VL_D(lane) = q.x; VL_D(64 + lane) = q.y; VL_D(128 + lane) = q.z;
    bool need = false;
    V3 pr = pv, qr = q;
    bool in_q = false;
    int cq[3] = {0, 0, 0};
    if constexpr (!SPH) {
      double tq[3];
      point_cell(a, q, qr, in_q, cq, tq);
      if constexpr (SIG) sig_put(a, j, true, signature_of(a, in_q, cq, tq, x, y, z), sig_own_row(), 0);
    } else {
      if constexpr (SIG) sig_put(a, j, true, cell_signature(x, y, z, g), sig_own_row(), 0);
    }
    if constexpr (SPH) {
      if (active && !VL_U(Lay::HIT + lane)) {
        // the previous point's field value has arrived: far from every occupied centre, certainly within r of one, or
        // in the shell between -- then the point is queued for the exact scan
        if (sph_state & 1u) {
          if (!g.rot_is_identity) {
            pr.x = g.inv_rot[0] * pv.x + g.inv_rot[1] * pv.y + g.inv_rot[2] * pv.z;
            pr.y = g.inv_rot[3] * pv.x + g.inv_rot[4] * pv.y + g.inv_rot[5] * pv.z;
            pr.z = g.inv_rot[6] * pv.x + g.inv_rot[7] * pv.y + g.inv_rot[8] * pv.z;
          }
          const float dl = VL_F(Lay::DELTA + lane);
          if (dn_prev - dl > a.r_hi) {}
          else if ((sph_state & 2u) && dn_prev + dl < a.r_lo) VL_U(Lay::HIT + lane) = 1u;
          else need = true;
        }
        sph_state = 0;
        if (!VL_U(Lay::HIT + lane)) request(a, q);
      }
    } else
    if (active && !VL_U(Lay::HIT + lane)) need = voxel_test(a.hot, j == 0, in_q, cq);
    const unsigned long long wm = __ballot(need);
    if (wm) {
      if (need) {
        if constexpr (!SPH) pr = to_voxel_frame(a, pv);
        const int slot = Lay::wrap(qhead + qcount + __popcll(wm & (((unsigned long long)1 << lane) - 1)));
        VL_D(Lay::QE + 0 * VQ + slot) = pr.x; VL_D(Lay::QE + 1 * VQ + slot) = pr.y; VL_D(Lay::QE + 2 * VQ + slot) = pr.z;
        if constexpr (!SPH) { VL_D(Lay::QE + 3 * VQ + slot) = qr.x; VL_D(Lay::QE + 4 * VQ + slot) = qr.y; VL_D(Lay::QE + 5 * VQ + slot) = qr.z; }
        Lay::set_owner(slot, lane);
      }
      qcount += __popcll(wm);
      if (qcount >= VQ - 64) flush();
    }
    VL_BARRIER();
  }

  // Retraction robots (fk_retract_kernel.hpp): the wave runs tip-aligned, a lane's point j arrives in row j + (P - P_lane).
  // `row` is the lane's own for its first two points and wave-uniform afterwards; `on`: the lane has a point in this call;
  // `first`: it is the lane's point 0.  Same per-point work as operator() -- only the milestones of the self-collision
  // proof are laid out from the TIP: slot k holds the point k CH rows before the tip row P - 1 (slot 0 = the tip), the
  // lane's base point closes the list in slot ceil((P - 1 - row) / CH), and the arc positions are stored negated, so they
  // still grow with the slot index.  milestones_unresolved only looks at spans between slots, chord lengths and arc
  // differences, none of which depends on the direction the backbone is traversed in.
  __device__ __forceinline__ void tip_point(int row, bool on, bool first, double x, double y, double z, bool wave_row = true) {
#pragma clang fp contract(off)
    VL_BARRIER();
    const VerdictArgs &a = args();
    const GridK &g = a.g;
    const int lane = threadIdx.x;
    const V3 q = {x, y, z};
    V3 pv = q;
    bool need = false;
    V3 pr = pv, qr = q;
    bool in_q = false;
    int cq[3] = {0, 0, 0};
    double tq[3] = {0.0, 0.0, 0.0};
    if constexpr (!SPH) point_cell(a, q, qr, in_q, cq, tq);          // (idle lanes: whatever their registers hold -- every use is under `on`)
    if constexpr (SIG) {
      // signatures sit in tip-aligned rows like the rows of K1r's stored points.  A lane's first two points come in rows of
      // its own (wave_row false): stored directly, two words per lane; from its third point on the row is the wave's and
      // the words go through the tile (SigStage::first_row = the lane's base row + 2 keeps stale tile words of earlier rows
      // from reaching this lane's row)
      uint32_t sg = 0u;
      if constexpr (!SPH) sg = on ? signature_of(a, in_q, cq, tq, x, y, z) : 0u;
      else sg = on ? cell_signature(x, y, z, g) : 0u;
      if (!wave_row) {
        if (on && sig_row_of >= 0) a.sig[(int64_t)sig_row_of * a.sig_stride + row] = sg;
        if (first) sig_first_row = row + 2;
      } else sig_put(a, row, on, sg, sig_row_of, sig_first_row);
    }
    if (on) {
      float d = 0.0f;
      if (!first) { pv = V3{VL_D(lane), VL_D(64 + lane), VL_D(128 + lane)}; d = VL_F(Lay::DIST + lane); }
      pr = pv;
      {
        const float dx = (float)(q.x - pv.x), dy = (float)(q.y - pv.y), dz = (float)(q.z - pv.z);
        d += __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz);   // v_sqrt_f32 (1 ulp): the proof keeps 1e-6 m of slack over ~1e-8 m of rounding
        VL_F(Lay::DIST + lane) = d;
      }
      {
        // back = rows to the tip; k = back / CH through a float quotient (back < 2^16, CH <= 48: (back + 1/2) / CH is at
        // least 1 / (2 CH) away from an integer, far beyond the rounding of the product)
        const int back = P - 1 - row;
        const int k = (int)(((float)back + 0.5f) * (1.0f / (float)CH));
        const bool ms_row = (back - k * CH) == 0;
        if (first || ms_row) {
          const int slot = (first && !ms_row) ? k + 1 : k;
          const int o = Lay::MS + slot * 64 + lane, pl = NM * 64;
          VL_F(o) = (float)q.x; VL_F(o + pl) = (float)q.y; VL_F(o + 2 * pl) = (float)q.z; VL_F(o + 3 * pl) = -d;
        }
      }
      VL_D(lane) = q.x; VL_D(64 + lane) = q.y; VL_D(128 + lane) = q.z;
      if constexpr (SPH) {
        if (active && !VL_U(Lay::HIT + lane)) {
          if (sph_state & 1u) {
            if (!g.rot_is_identity) {
              pr.x = g.inv_rot[0] * pv.x + g.inv_rot[1] * pv.y + g.inv_rot[2] * pv.z;
              pr.y = g.inv_rot[3] * pv.x + g.inv_rot[4] * pv.y + g.inv_rot[5] * pv.z;
              pr.z = g.inv_rot[6] * pv.x + g.inv_rot[7] * pv.y + g.inv_rot[8] * pv.z;
            }
            const float dl = VL_F(Lay::DELTA + lane);
            if (dn_prev - dl > a.r_hi) {}
            else if ((sph_state & 2u) && dn_prev + dl < a.r_lo) VL_U(Lay::HIT + lane) = 1u;
            else need = true;
          }
          sph_state = 0;
          if (!VL_U(Lay::HIT + lane)) request(a, q);
        }
      } else
      if (active && !VL_U(Lay::HIT + lane)) need = voxel_test(a.hot, first, in_q, cq);
    }
    const unsigned long long wm = __ballot(need);
    if (wm) {
      if (need) {
        if constexpr (!SPH) pr = to_voxel_frame(a, pv);
        const int slot = Lay::wrap(qhead + qcount + __popcll(wm & (((unsigned long long)1 << lane) - 1)));
        VL_D(Lay::QE + 0 * VQ + slot) = pr.x; VL_D(Lay::QE + 1 * VQ + slot) = pr.y; VL_D(Lay::QE + 2 * VQ + slot) = pr.z;
        if constexpr (!SPH) { VL_D(Lay::QE + 3 * VQ + slot) = qr.x; VL_D(Lay::QE + 4 * VQ + slot) = qr.y; VL_D(Lay::QE + 5 * VQ + slot) = qr.z; }
        Lay::set_owner(slot, lane);
      }
      qcount += __popcll(wm);
      if (qcount >= VQ - 64) flush();
    }
    VL_BARRIER();
  }
};

// What a lane of the shared-grid verdict kernels concludes when its integration and the per-point sweep have ended (what sweep_body
// does after its pass 1; comparisons and one subtraction: nothing here can contract).  pending: only the exact pairwise
// self-collision sweep can decide the configuration (its bit is then left 0 and the fallback pass ORs it in).
struct LaneVerdict { bool valid, pending; uint32_t fl; };
template <int N, bool SPH, bool SIG>
__device__ __forceinline__ LaneVerdict verdict_decide(const VerdictArgs &a, const FkLane<N> &fl_, bool live) {
  using Lay = VLay<SIG>;
  const int lane = threadIdx.x;
  const int P = a.P;
  const int Kl = (P - 1 + a.CH - 1) / a.CH;
  const bool conv_ok = live && fl_.converged;
  bool len_ok = false;
  if (conv_ok) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; j++) {
      const double dl = a.home_Li[j] - fl_.Li[j];
      if (dl < a.min_len[j] || a.max_len[j] < dl) ok = false;
    }
    len_ok = ok;
  }
  bool alive = conv_ok && len_ok;
  const uint32_t hf = VL_U(Lay::HIT + lane);
  const bool hit = (hf & 1u) != 0;
  bool bad = (hf & 2u) != 0;
  if (alive && !(VL_F(Lay::DIST + lane) < 1e30f)) { alive = false; bad = true; }   // NaN / inf points
  bool need_exact = alive && P > 2;
  if (!(a.debug & 2u) && __any(need_exact)) {
    const float *mx = (const float *)vlds + Lay::MS + lane, *my = mx + (size_t)a.NM * 64, *mz = my + (size_t)a.NM * 64, *ma = mz + (size_t)a.NM * 64;
    need_exact = milestones_unresolved(mx, my, mz, ma, a.NM, Kl, need_exact, (float)a.radius);
  }

  uint32_t fl = 0;
  if (conv_ok) fl |= 1u;
  if (conv_ok && len_ok) fl |= 2u;
  bool valid = conv_ok && len_ok;
  if (valid && bad) { fl |= 16u; valid = false; }
  // unresolved self collision: decided by the fallback pass -- unless an obstacle hit already settles the verdict and
  // nobody asked for the flags (which record the self-collision result on their own)
  const bool pending = valid && need_exact && (a.flags != nullptr || !hit);
  if (valid && !pending) {
    if (!need_exact) { fl |= 4u; if (!hit) fl |= 8u; else valid = false; }
    else valid = false;                                       // hit, flags not wanted
  }
  if (pending) valid = false;                                 // its bit is ORed in by fk_sweep_fused_list
  if (SPH && pending && !hit) fl |= 8u;                       // ... which takes the sphere test's answer from this bit (sweep_body, check_voxels == 2)
  return LaneVerdict{valid, pending, fl};
}

// Waves per SIMD: K1 with stored points holds two up to 6 tendons (tr_types.hpp); with the sweep in the loop the spills at that
// budget cost more than the second wave hides from 5 tendons on (measured, profiles/r02/verdict_widths_v1.txt, ms per 2^18
// configurations, two waves | one: N=4 3.11 | 3.13, N=5 4.01 | 3.52, N=6 5.11 | 4.02; N=7 5.96 | 4.61, N=8 7.08 | 4.92).
#ifndef TRK_VERDICT_TWO_WAVE_MAXN
#define TRK_VERDICT_TWO_WAVE_MAXN 4
#endif
template <int N, bool ROT, bool SPH, bool SIG = false>
__global__ __launch_bounds__(64, (N <= TRK_VERDICT_TWO_WAVE_MAXN ? 2 : 1)) void fk_verdict(
    const double *__restrict__ states, int64_t n, RobotK K, const double *__restrict__ tab, const StepK *__restrict__ steps,
    int nsteps, double *__restrict__ tips, const VerdictArgs *__restrict__ va) {
  using Lay = VLay<SIG>;
  const int lane = threadIdx.x;
  PointSweep<SPH, SIG> ps;
  ps.va = va;
  ps.dn_prev = 0.0f; ps.sph_state = 0u;
  ps.qhead = 0; ps.qcount = 0; ps.active = false;
  ps.P = va->P; ps.CH = va->CH; ps.NM = va->NM; ps.Kl = (ps.P - 1 + ps.CH - 1) / ps.CH; ps.ms_next = 0; ps.ms_k = 0;
  VL_U(Lay::HIT + lane) = 0u; VL_U(Lay::INPREV + lane) = 0u; VL_F(Lay::DIST + lane) = 0.0f;
  {
    const int64_t i0 = (int64_t)blockIdx.x * 64 + lane;
    ps.sigst.init(); ps.sig_row_of = -1; ps.sig_first_row = 0; ps.sig_rows_of_launch(n);
  }
  __syncthreads();

  FkLane<N> fl_;
  FkOut out{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  fk_uniform_body<N, ROT, false, false>(states, n, 0, K, tab, steps, nsteps, out, ps, nullptr, &fl_);

  // ---- what sweep_body does after its pass 1 (comparisons and one subtraction: nothing here can contract) ----
  ps.template sig_finish<false>();
  ps.finish();
  while (ps.qcount > 0) ps.flush();
  __syncthreads();
  if (tips) {
    // the wave's 64 tips are 1 536 contiguous bytes: through LDS (the previous-point slots, free now) so that each of the
    // three store instructions writes 512 contiguous bytes -- lane by lane (x, y, z at a stride of 24 bytes) every
    // instruction touched every 32-byte sector of the block and the block was written three times over (measured: 70 B per
    // check against the 24 the tips are)
    VL_D(lane) = fl_.tip[0]; VL_D(64 + lane) = fl_.tip[1]; VL_D(128 + lane) = fl_.tip[2];
    __syncthreads();
    const int64_t e0 = (int64_t)blockIdx.x * 192, etot = 3 * n;
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int e = q * 64 + lane;                 // element e of the block = coordinate e % 3 of configuration e / 3
      if (e0 + e < etot) tips[e0 + e] = VL_D((e % 3) * 64 + e / 3);
    }
    __syncthreads();
  }
  const VerdictArgs a = *va;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  const bool live = i < n;
  const LaneVerdict lv = verdict_decide<N, SPH, SIG>(a, fl_, live);
  const bool valid = lv.valid, pending = lv.pending;
  const uint32_t fl = lv.fl;
  const uint64_t bits = __ballot(valid && live);
  if (lane == 0 && live) a.valid_bits[i >> 6] = bits;
  if (a.flags && live) a.flags[i] = (uint8_t)fl;
  const unsigned long long pm = __ballot(pending);
  if (pm) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(a.fb_count, (uint32_t)__popcll(pm));
    base = __shfl(base, 0, 64);
    if (pending) a.fb_list[base + __popcll(pm & (((unsigned long long)1 << lane) - 1))] = (int32_t)i;
  }
}

#ifdef TRK_WITH_RETRACT_VERDICT
// The same for retraction-enabled robots: K1r's body (fk_retract_kernel.hpp: per-lane arc-length grid, tip-aligned
// iterations) with the sweep in its point hook (PointSweep::tip_point).  What differs after the loop is per lane: the
// number of points, the last milestone slot and the home-shape tendon lengths.
// The lane's own first interval -- fixed-point solve of the base strains, per-lane routing, up to two RK4 steps that evaluate the
// routing polynomials per stage -- needs more than the 256 registers the two-wave budget of the kernel below allows: inside it the
// first interval spilled 450 - 650 B per lane, and with 8 waves x 32 CUs x 64 lanes of that per XCD the frames did not stay in
// the 4 MiB L2 (1.3 - 2.5 KB of HBM traffic per check for 68 algorithmic bytes, profiles/r03/traffic_split_v2.json).  So it runs
// here, ONE wave per SIMD with the whole register file, and hands the integrator's state over (18 + N + 1 doubles per
// configuration, written and read coalesced).
// (fk_retract_prologue: fk_retract_kernel.hpp -- the stored-point form of large batches takes the same two launches)
template <int N, bool ROT, bool SPH, bool SIG = false>
__global__ __launch_bounds__(64, (N <= TRK_VR_TWO_WAVE_MAXN ? 2 : 1)) void fk_verdict_retract(
    const double *__restrict__ states, int64_t n, RobotK K, const PolyK *__restrict__ pk, const double *__restrict__ tab,
    const StepK *__restrict__ steps, int nsteps, int k_first, const double *__restrict__ tgrid, const double *__restrict__ hl,
    double *__restrict__ tips, const VerdictArgs *__restrict__ va, RetractHandoff ho) {
  using Lay = VLay<SIG>;
  const int lane = threadIdx.x;
  PointSweep<SPH, SIG> ps;
  ps.va = va;
  ps.dn_prev = 0.0f; ps.sph_state = 0u;
  ps.qhead = 0; ps.qcount = 0; ps.active = false;
  ps.P = va->P; ps.CH = va->CH; ps.NM = va->NM; ps.Kl = 0; ps.ms_next = 0; ps.ms_k = 0;
  const int32_t *__restrict__ perm = va->perm;
  {
    const int64_t i0 = (int64_t)blockIdx.x * 64 + lane;
    ps.sigst.init(); ps.sig_row_of = i0 < n ? (perm ? perm[i0] : (int)i0) : -1; ps.sig_first_row = 0; ps.sig_row0 = 0; ps.sig_cnt = 0;
  }
  VL_U(Lay::HIT + lane) = 0u; VL_U(Lay::INPREV + lane) = 0u; VL_F(Lay::DIST + lane) = 0.0f;
  __syncthreads();

  FkLaneR<N> fl_;
  FkOut out{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, tips, nullptr, nullptr, nullptr};
  const int32_t *__restrict__ wkb = va->wave_k_begin;
  fk_retract_body<N, ROT, false, PointSweep<SPH, SIG> &, 2>(states, n, 0, K, pk, tab, steps, nsteps, k_first, tgrid, hl, out, ps, perm, &fl_,
                                                            wkb ? wkb[blockIdx.x] : 0, &ho);

  ps.template sig_finish<true>();
  ps.finish();
  while (ps.qcount > 0) ps.flush();
  __syncthreads();
  const VerdictArgs a = *va;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  const bool live = i < n;
  const int64_t c = (perm && live) ? (int64_t)perm[i] : i;      // the configuration this lane has integrated
  const int np = fl_.np;
  const int Kl = (np - 1 + a.CH - 1) / a.CH;              // this lane's last milestone slot: its base point
  const bool conv_ok = live && fl_.converged;
  bool len_ok = false;
  if (conv_ok) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; j++) {
      const double dl = fl_.home_Li[j] - fl_.Li[j];
      if (dl < a.min_len[j] || a.max_len[j] < dl) ok = false;
    }
    len_ok = ok;
  }
  bool alive = conv_ok && len_ok;
  const uint32_t hf = VL_U(Lay::HIT + lane);
  const bool hit = (hf & 1u) != 0;
  bool bad = (hf & 2u) != 0;
  if (alive && !(VL_F(Lay::DIST + lane) < 1e30f)) { alive = false; bad = true; }   // NaN / inf points
  bool need_exact = alive && np > 2;
  if (!(a.debug & 2u) && __any(need_exact)) {
    const float *mx = (const float *)vlds + Lay::MS + lane, *my = mx + (size_t)a.NM * 64, *mz = my + (size_t)a.NM * 64, *ma = mz + (size_t)a.NM * 64;
    need_exact = milestones_unresolved(mx, my, mz, ma, a.NM, Kl, need_exact, (float)a.radius);
  }

  uint32_t fl = 0;
  if (conv_ok) fl |= 1u;
  if (conv_ok && len_ok) fl |= 2u;
  bool valid = conv_ok && len_ok;
  if (valid && bad) { fl |= 16u; valid = false; }
  const bool pending = valid && need_exact && (a.flags != nullptr || !hit);
  if (valid && !pending) {
    if (!need_exact) { fl |= 4u; if (!hit) fl |= 8u; else valid = false; }
    else valid = false;
  }
  if (pending) valid = false;                                 // its bit is ORed in by fk_sweep_retract_list
  if (SPH && pending && !hit) fl |= 8u;
  if (perm) {
    if (valid && live) atomicOr((unsigned long long *)&a.valid_bits[c >> 6], 1ull << (c & 63));
  } else {
    const uint64_t bits = __ballot(valid && live);
    if (lane == 0 && live) a.valid_bits[i >> 6] = bits;
  }
  if (a.flags && live) a.flags[c] = (uint8_t)fl;
  if (a.np_out && live) a.np_out[c] = np;
  const unsigned long long pm = __ballot(pending);
  if (pm) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(a.fb_count, (uint32_t)__popcll(pm));
    base = __shfl(base, 0, 64);
    if (pending) a.fb_list[base + __popcll(pm & (((unsigned long long)1 << lane) - 1))] = (int32_t)c;
  }
}

// Fallback pass for the configurations whose self-collision test needs every backbone point at once (as
// fk_sweep_fused_list, fused_kernel.hpp): K1r with stored points on the compacted list, then K2's body on the tip-aligned rows.
template <int N, bool ROT>
__global__ __launch_bounds__(64, (N <= TRK_K1R_TWO_WAVE_MAXN ? 2 : 1)) void fk_sweep_retract_list(
    const double *__restrict__ states, int64_t cap, int64_t ld, RobotK K, const PolyK *__restrict__ pk, const double *__restrict__ tab,
    const StepK *__restrict__ steps, int nsteps, int k_first, const double *__restrict__ tgrid, const double *__restrict__ hl,
    FkOut out, const FusedSweepArgs *__restrict__ sa, const int32_t *__restrict__ list, const uint32_t *__restrict__ count) {
  const int64_t total = (int64_t)*count;
  for (int64_t offset = 0; offset < total; offset += cap) {
    const int64_t left = total - offset;
    const int64_t m = left < cap ? left : cap;
    if ((int64_t)blockIdx.x * 64 >= m) break;                     // wave-uniform
    fk_retract_body<N, ROT, false>(states, m, ld, K, pk, tab, steps, nsteps, k_first, tgrid, hl, out, NoPointHook(), list + offset);
    __syncthreads();
    const FusedSweepArgs a = *sa;
    sweep_body<true>(a.in, m, ld, a.P, a.CH, a.NM, K, a.g, a.grid, a.near_grid, a.check_voxels, a.debug, a.valid_bits, a.flags, list + offset);
    __syncthreads();
  }
}
#endif

}  // namespace trk
