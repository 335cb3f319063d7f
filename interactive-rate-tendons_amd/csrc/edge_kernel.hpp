// edge_kernel.hpp -- K3: the device-resident frontier of the batched swept-volume edge check
// (VoxelEnvironment::voxelize_valid_backbone_motion, motion-planning/VoxelEnvironment.cpp:304-424,
// driven by AbstractVoxelMotionValidator::checkMotion, AbstractVoxelMotionValidator.h:143-151).
//
// One bisection level of all edges at a time (edge_host.inc explains why that equals the reference's
// depth-first order): `edge_open` turns the frontier intervals that are still undecided into new FK
// samples (OMPL interpolate), K1 + K2 evaluate them, `edge_fold` folds the verdicts into per-edge
// state, `edge_filter` runs the reference's `should_subdivide` on both halves of every interval and
// emits the next frontier.  The host only reads one counter per level.
//
// `cells_differ` is `should_subdivide` (:304-341): do two backbone shapes differ by more than one
// voxel, on any axis, at any backbone point?  Points are rotated into the voxel frame first (the
// reference stores the rotated shapes, :262-272) and located with find_cell
// (collision/VoxelOctree.cpp:309-317: closed domain check, then size_t((x - min) / d) -- a DIVISION,
// unlike add_line's reciprocal multiply).  IEEE fp64, no contraction: the flags are integer-exact
// functions of the points, the interpolated states bit-equal to the host / oracle arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include "tr_types.hpp"
#include "sweep_kernel.hpp"

namespace trk {

struct EdgeIv { int32_t e, sa, sb, pad; double ta, tb; };   // interval [ta, tb] of edge e between pool samples sa, sb

// OMPL 1.5.0 state-space constants as wired by motion-planning/Problem.cpp:101-163 (host-computed
// so that device and host use the same bits): longest valid segment length per subspace.
struct EdgeSpaceK { int N, rot, retr, S; double lvs_tension, lvs_rot, lvs_retr; };

enum { EC_FRONT = 0, EC_OPEN = 1, EC_DOMAIN = 2, EC_COUNT = 4 };

struct EdgeState {
  const double *A, *B;          // [E][S] end states of the chunk's edges
  double *rel;                  // [E] 1 / validSegmentCount
  uint32_t *edge_ok;            // [E]
  int32_t *nfk;                 // [E] FK samples evaluated
  unsigned long long *first_inv;// [E] bits of the smallest invalid t (10.0 = none); "until invalid" mode only
  unsigned long long *last_t;   // [E] bits of the largest sampled t below first_inv
  int32_t *sample_edge;         // [cap] pool slot -> edge (-1 = padding)
  double *sample_t;             // [cap] pool slot -> t
  const uint64_t *bits;         // [cap/64] pool slot -> K2 verdict
  uint32_t *counters;           // [EC_COUNT]
};

// 0 = do not subdivide, 1 = subdivide, 2 = a point lies outside the voxel domain (std::domain_error
// in the reference).  Points are visited from the tip down, as the reference does, so an early
// "subdivide" wins over a later out-of-domain point exactly as it does there.
__device__ inline int cells_differ(const double *__restrict__ px, const double *__restrict__ py, const double *__restrict__ pz,
                                   int64_t ld, int P, const int32_t *__restrict__ n_points, int64_t sa, int64_t sb, const GridK &g) {
#pragma clang fp contract(off)
  int Pm = P;
  if (n_points) {
    // retraction: backbones of different lengths (VoxelEnvironment.cpp:317-326): more than one link
    // apart -> subdivide; otherwise compare the common prefix of points
    const int na = n_points[sa], nb = n_points[sb];
    if (na + 1 < nb || na > nb + 1) return 1;
    Pm = na < nb ? na : nb;
    sa += (int64_t)(P - na) * ld; sb += (int64_t)(P - nb) * ld;    // K1r's rows are aligned at the tip: point j is in row j + (P - n)
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
        !(fabs(A.z) < 1e300) || !(fabs(B.z) < 1e300)) return 2;
    const long ax = (long)((A.x - g.xmin) / g.dx), ay = (long)((A.y - g.ymin) / g.dy), az = (long)((A.z - g.zmin) / g.dz);
    const long bx = (long)((B.x - g.xmin) / g.dx), by = (long)((B.y - g.ymin) / g.dy), bz = (long)((B.z - g.zmin) / g.dz);
    const long dx = ax > bx ? ax - bx : bx - ax, dy = ay > by ? ay - by : by - ay, dz = az > bz ? az - bz : bz - az;
    if (dx > 1 || dy > 1 || dz > 1) return 1;
  }
  return 0;
}

// cells_differ on cell signatures (sweep_kernel.hpp: cell_signature): the whole wave tests ONE pair of samples, lanes
// over backbone points from the tip down, 64 at a time (two coalesced 256-byte reads); the first event in tip-first
// order decides, a domain error at a point before a difference at that point -- as the sequential loop does.
// n_points (retraction robots): the samples' point counts; their rows are aligned at the tip (point j of a backbone of n
// points sits in row j + P - n), backbones more than one link apart subdivide, otherwise the common prefix of points is
// compared index by index from the base (VoxelEnvironment.cpp:317-326), as cells_differ does on stored points.
__device__ inline int signatures_differ(const uint32_t *__restrict__ sig, int64_t stride, int P, const int32_t *__restrict__ n_points,
                                        int64_t sa, int64_t sb) {
  const int lane = (int)(threadIdx.x & 63);
  const uint32_t *__restrict__ ra = sig + sa * stride, *__restrict__ rb = sig + sb * stride;
  int Pm = P;
  if (n_points) {
    const int na = n_points[sa], nb = n_points[sb];
    if (na + 1 < nb || na > nb + 1) return 1;
    Pm = na < nb ? na : nb;
    ra += P - na; rb += P - nb;
  }
  for (int j0 = Pm - 1; j0 >= 0; j0 -= 64) {
    const int j = j0 - lane;
    uint32_t va = 0, vb = 0;
    if (j >= 0) { va = ra[j]; vb = rb[j]; }
    const bool bad = ((va | vb) & SIG_BAD) != 0;
    const int dx = (int)(va & 1023u) - (int)(vb & 1023u), dy = (int)((va >> 10) & 1023u) - (int)((vb >> 10) & 1023u),
              dz = (int)((va >> 20) & 1023u) - (int)((vb >> 20) & 1023u);
    const bool diff = !bad && (dx > 1 || dx < -1 || dy > 1 || dy < -1 || dz > 1 || dz < -1);
    const unsigned long long mb = __ballot(bad), md = __ballot(diff);
    if (mb | md) {
      const int fb = mb ? __ffsll((long long)mb) : 65, fd = md ? __ffsll((long long)md) : 65;
      return fb <= fd ? 2 : 1;
    }
  }
  return 0;
}

// One slot per lane with pred set, one atomic per wave.  Every lane of the wave must call it.
__device__ inline uint32_t wave_alloc(bool pred, uint32_t *counter) {
  const unsigned long long mask = __ballot(pred);
  const int lane = (int)(threadIdx.x & 63);
  uint32_t base = 0;
  if (mask != 0) {
    const int leader = __ffsll((long long)mask) - 1;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = (uint32_t)__shfl((int)base, leader);
  }
  return base + (uint32_t)__popcll(mask & (((unsigned long long)1 << lane) - 1));
}

// StateSpace::validSegmentCount of the compound space: max over subspaces of ceil(distance / lvs).
__device__ inline unsigned valid_segment_count_dev(const EdgeSpaceK &sk, const double *a, const double *b) {
#pragma clang fp contract(off)
  unsigned sc = 0;
  {
    double s = 0;
    for (int i = 0; i < sk.N; i++) { const double d = a[i] - b[i]; s += d * d; }
    const unsigned v = (unsigned)ceil(sqrt(s) / sk.lvs_tension);
    sc = v > sc ? v : sc;
  }
  if (sk.rot) {
    double d = fabs(a[sk.N] - b[sk.N]);
    d = (d > M_PI) ? 2.0 * M_PI - d : d;
    const unsigned v = (unsigned)ceil(d / sk.lvs_rot);
    sc = v > sc ? v : sc;
  }
  if (sk.retr) {
    const int k = sk.N + sk.rot;
    const double d = a[k] - b[k];
    const unsigned v = (unsigned)ceil(sqrt(d * d) / sk.lvs_retr);
    sc = v > sc ? v : sc;
  }
  return sc;
}

// CompoundStateSpace::interpolate: linear per tension / retraction, shortest arc on SO2.
__device__ inline void interpolate_state_dev(const EdgeSpaceK &sk, const double *a, const double *b, double t, double *out) {
#pragma clang fp contract(off)
  for (int i = 0; i < sk.N; i++) out[i] = a[i] + (b[i] - a[i]) * t;
  if (sk.rot) {
    const int N = sk.N;
    double diff = b[N] - a[N];
    if (fabs(diff) <= M_PI) out[N] = a[N] + diff * t;
    else {
      if (diff > 0.0) diff = 2.0 * M_PI - diff; else diff = -2.0 * M_PI - diff;
      double v = a[N] - diff * t;
      if (v > M_PI) v -= 2.0 * M_PI; else if (v < -M_PI) v += 2.0 * M_PI;
      out[N] = v;
    }
  }
  if (sk.retr) {
    const int k = sk.N + sk.rot;
    out[k] = a[k] + (b[k] - a[k]) * t;
  }
}

// ---- the edge queue (edge_queue_kernel.hpp: fk_edge_queue) --------------------------------------------------------------
// The bisection without level barriers: the pool is a queue of FK samples in push order.  Persistent waves take samples
// from its head, integrate them (the verdict-only body with cell signatures), fold the verdicts into their edges, and the wave
// that folds the LAST outstanding sample of an edge's level runs should_subdivide on that level's signature rows and pushes the
// edge's next level at the tail.  An edge waits for nothing but its own samples: first_invalid_t is per edge
// (VoxelEnvironment.cpp:357-398), so the sample set of every edge -- and with it the verdict AND the count of FK calls -- is what
// the level-synchronous schedule (edge_host.inc) evaluates.
// The queue is the pool in push order: slots [first, tail), the edges' first midpoints seeded before the launch in the order of
// the edge list (neighbouring edges share end vertices: their rows stay close in memory; ordering the seeds by edge length, longest
// first, was measured -- it shortens the tail and costs more than that in scattered row reads), then everything the waves push.  EQ_AVAIL counts the published samples nobody has taken yet (a
// semaphore: a wave subtracts what it wants and gives back what was not there), EQ_HEAD hands out their positions.  The launch
// ends when done == tail: `done` counts folded samples from `first`, and it is read BEFORE `tail`, so equality means that nothing
// was in flight at that moment.
// Control words (uint32, zeroed by the host; the hot ones on 128-byte lines of their own).
enum { EQ_HEAD = 0, EQ_TAIL = 32, EQ_DONE = 64, EQ_FLAGS = 96, EQ_DOMAIN = 97, EQ_PENDING = 98, EQ_FIRST = 99, EQ_BATCHES = 100, EQ_FINISHED = 101,
       EQ_CAND = 102,
       // where the waves' time went, summed over all rounds in ticks of the 100 MHz wall clock (64-bit words; TENDON_HIP_EDGE_TIMING prints them):
       // claiming a batch (incl. idling), waiting for its records, integrating, the exact sweep, folding + finishing levels + publishing
       EQ_T_CLAIM = 104, EQ_T_READY = 106, EQ_T_FK = 108, EQ_T_EXACT = 110, EQ_T_FOLD = 112,
       // ... and inside "folding": release + decrement | acquire + level records | candidates' verdicts | allocation + records | release + publish
       EQ_T_F0 = 114, EQ_T_F1 = 116, EQ_T_F2 = 118, EQ_T_F3 = 120, EQ_T_F4 = 122,
       EQ_SIZES = 124,              // rounds of 64 | 32..63 | 2..31 | 1 samples
       EQ_AVAIL = 128,
#ifdef TRK_EQ_TRACE
       EQ_TRACE = 160, EQ_WORDS = 160 + 8 * 64 * 4 };   // (a tuning build, profiles/build_ab.py fk_q4,tendon_hip -DTRK_EQ_TRACE: eight waves write start / end of
                                                        // integration / end of fold and the shader clock of their first 64 rounds; TENDON_HIP_EDGE_TIMING prints them)
#else
       EQ_WORDS = 160 };
#endif
enum { EQF_OVERFLOW = 1u, EQF_STUCK = 2u, EQF_DEEP = 4u };      // pool too small | a wait made no progress | a level of more than 2048 intervals
constexpr int EQ_STASH_WORDS = 68;        // LDS words a wave keeps behind the verdict body's image across an integration (edge_queue_kernel.hpp)
constexpr int EQ_MAX_CAND = 4096;        // candidates (two per interval) of one edge level the finishing wave can hold

struct EdgeQueueArgs {
  uint32_t *ctl;                  // [EQ_WORDS]
  int32_t slot_hi;                // the run's pool slots end here
  int32_t P;                      // backbone points = words of a signature row in use
  EdgeSpaceK sk;
  const double *A, *B;            // [E][S] end states
  const double *rel;              // [E] 1 / validSegmentCount
  uint32_t *edge_ok;              // [E]
  int32_t *nfk;                   // [E]
  int32_t *remaining;             // [E] samples of the edge's current level not folded yet
  int32_t *lvl_base, *lvl_cnt;    // [E] the current level's samples: pool slots [base, base + cnt)
  EdgeIv *iv;                     // [cap] per pool slot: the interval whose midpoint the sample is
  double *states;                 // [cap][S] per pool slot: the sample's state
  int32_t *sample_edge;           // [cap] per pool slot: its edge; -1 until the slot's record is complete (the ready flag)
  const uint32_t *sig;            // [cap][sig_stride] signature rows (written by the queue's own waves through VerdictArgs::sig)
  int64_t sig_stride;
  // the wave's own columns of the point workspace (column = blockIdx.x * 64 + lane, fb_ld >= 64 gridDim.x): a sample whose
  // self-collision test needs the exact pairwise sweep is integrated again by the wave that found it, with stored points
  FkOut fb_out; SweepIn fb_in; int64_t fb_ld;
};

#ifndef TRK_EDGE_DEVICE_ONLY
// After level 0 (edge_init* / edge_filter<true> / edge_open into per-slot positions): every open interval is a
// whole edge whose midpoint is pool sample s0 + q; seeds the per-edge level records and the queue's control words.
__global__ __launch_bounds__(256) void edge_queue_seed(EdgeState st, const EdgeIv *__restrict__ iv /* [slot] */, int64_t s0, int64_t slot_hi,
                                                       int32_t *__restrict__ remaining, int32_t *__restrict__ lvl_base, int32_t *__restrict__ lvl_cnt,
                                                       uint32_t *__restrict__ ctl) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t m = st.counters[EC_OPEN];
  const bool fits = s0 + m <= slot_hi;
  if (q == 0) {
    ctl[EQ_FIRST] = (uint32_t)s0; ctl[EQ_DONE] = (uint32_t)s0; ctl[EQ_HEAD] = (uint32_t)s0;
    ctl[EQ_TAIL] = (uint32_t)(fits ? s0 + m : s0); ctl[EQ_AVAIL] = (uint32_t)(fits ? m : 0);
    if (!fits) ctl[EQ_FLAGS] = EQF_OVERFLOW;
  }
  if (q >= m || !fits) return;
  const int32_t e = iv[s0 + q].e;
  remaining[e] = 1; lvl_base[e] = (int32_t)(s0 + q); lvl_cnt[e] = 1;
  st.nfk[e] += 1;
}

// level 0: both end states of every edge become pool samples 2k, 2k + 1
__global__ __launch_bounds__(256) void edge_init(EdgeState st, EdgeSpaceK sk, int64_t E, double *__restrict__ lvl_states) {
#pragma clang fp contract(off)
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= E) return;
  const double *a = st.A + k * sk.S, *b = st.B + k * sk.S;
  for (int i = 0; i < sk.S; i++) { lvl_states[(2 * k) * sk.S + i] = a[i]; lvl_states[(2 * k + 1) * sk.S + i] = b[i]; }
  st.sample_edge[2 * k] = st.sample_edge[2 * k + 1] = (int32_t)k;
  st.sample_t[2 * k] = 0.0; st.sample_t[2 * k + 1] = 1.0;
  st.rel[k] = 1.0 / (double)valid_segment_count_dev(sk, a, b);
  st.edge_ok[k] = 1; st.nfk[k] = 0;
  st.first_inv[k] = (unsigned long long)__double_as_longlong(10.0);
  st.last_t[k] = 0;
}

// Indexed edges (tr_validate_edges_indexed): the end states are roadmap vertices evaluated ONCE for all their
// edges (pool slots = vertex indices).  Gathers the chunk's end states for interpolation and seeds the per-edge
// state from the vertices' verdict bits; the reference's count of FK calls per edge starts at the two ends.
__global__ __launch_bounds__(256) void edge_init_indexed(EdgeState st, EdgeSpaceK sk, int64_t E, const double *__restrict__ states,
                                                         const int32_t *__restrict__ idx, double *__restrict__ A, double *__restrict__ B) {
#pragma clang fp contract(off)
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= E) return;
  const int64_t sa = idx[2 * k], sb = idx[2 * k + 1];
  for (int i = 0; i < sk.S; i++) { A[k * sk.S + i] = states[sa * sk.S + i]; B[k * sk.S + i] = states[sb * sk.S + i]; }
  st.rel[k] = 1.0 / (double)valid_segment_count_dev(sk, states + sa * sk.S, states + sb * sk.S);
  const bool va = (st.bits[sa >> 6] >> (sa & 63)) & 1ull, vb = (st.bits[sb >> 6] >> (sb & 63)) & 1ull;
  st.edge_ok[k] = (va && vb) ? 1u : 0u;
  st.nfk[k] = 2;
  st.first_inv[k] = (unsigned long long)__double_as_longlong(10.0);
  st.last_t[k] = 0;
}

// fold the verdicts of pool samples [s0, s0 + m) into their edges
__global__ __launch_bounds__(256) void edge_fold(EdgeState st, int64_t s0, int64_t m, int until_invalid) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= m) return;
  const int64_t s = s0 + q;
  const int32_t e = st.sample_edge[s];
  if (e < 0) return;
  atomicAdd(&st.nfk[e], 1);
  const bool ok = (st.bits[s >> 6] >> (s & 63)) & 1ull;
  if (!ok) {
    st.edge_ok[e] = 0;
    // t >= 0, so the bit patterns order like the values
    if (until_invalid) atomicMin(&st.first_inv[e], (unsigned long long)__double_as_longlong(st.sample_t[s]));
  }
}

// should_subdivide on the candidate intervals of this level; survivors form the next frontier.
// LEVEL0: candidate i is the whole edge i.  Otherwise candidates 2q, 2q + 1 are the distal
// (:387-390) and proximal (:393-396) halves of open interval q, whose midpoint is pool sample s0 + q.
template <bool LEVEL0>
__global__ __launch_bounds__(256) void edge_filter(EdgeState st, const EdgeIv *__restrict__ open, const int32_t *__restrict__ idx /* LEVEL0: end-state pool slots per edge, or null */,
                                                   int64_t n_cand, int64_t s0,
                                                   const double *__restrict__ px, const double *__restrict__ py, const double *__restrict__ pz,
                                                   int64_t ld, int P, const int32_t *__restrict__ n_points, GridK g, int until_invalid,
                                                   EdgeIv *__restrict__ frontier, const uint32_t *__restrict__ sig, int64_t sig_stride) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  bool emit = false, keep = false;
  EdgeIv c{};
  if (i < n_cand) {
    if (LEVEL0) c = idx ? EdgeIv{(int32_t)i, idx[2 * i], idx[2 * i + 1], 0, 0.0, 1.0} : EdgeIv{(int32_t)i, (int32_t)(2 * i), (int32_t)(2 * i + 1), 0, 0.0, 1.0};
    else {
      const EdgeIv iv = open[i >> 1];
      const double tm = (iv.ta + iv.tb) / 2;                   // :382
      const int32_t sm = (int32_t)(s0 + (i >> 1));
      c = (i & 1) ? EdgeIv{iv.e, iv.sa, sm, 0, iv.ta, tm} : EdgeIv{iv.e, sm, iv.sb, 0, tm, iv.tb};
    }
    // should_subdivide is false when the first shape is invalid (:307-310); intervals of decided edges are dropped
    if (until_invalid) {
      const bool va = (st.bits[c.sa >> 6] >> (c.sa & 63)) & 1ull;
      keep = va && !(__longlong_as_double((long long)st.first_inv[c.e]) <= c.ta);
    } else keep = st.edge_ok[c.e] != 0;
  }
  int f = 0;
  if (sig) {
    // the wave takes its lanes' surviving candidates one after the other, all 64 lanes on one pair of signature rows
    unsigned long long todo = __ballot(keep);
    while (todo) {
      const int l = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const int r = signatures_differ(sig, sig_stride, P, n_points, __shfl(c.sa, l), __shfl(c.sb, l));
      if ((int)(threadIdx.x & 63) == l) f = r;
    }
  } else if (keep) {
    f = cells_differ(px, py, pz, ld, P, n_points, c.sa, c.sb, g);
  }
  if (keep) {
    if (f == 2) {                                              // std::domain_error in the reference
      if (atomicExch(&st.edge_ok[c.e], 0u) != 0u) atomicAdd(&st.counters[EC_DOMAIN], 1u);
      if (until_invalid) st.first_inv[c.e] = 0;                // t = 0.0: nothing of this edge is usable
    } else if (f == 1 && (c.tb - c.ta) > st.rel[c.e]) emit = true;   // width rule of :369-372, applied at push time
  }
  const uint32_t slot = wave_alloc(emit, &st.counters[EC_FRONT]);
  if (emit) frontier[slot] = c;
}

// pop time: intervals of edges decided since they were pushed are skipped (:373-376); the rest get
// their midpoint state and a pool slot s0 + slot.  Slots beyond the pool are counted, not written.
__global__ __launch_bounds__(256) void edge_open(EdgeState st, EdgeSpaceK sk, const EdgeIv *__restrict__ frontier, int64_t n_bound,
                                                 int64_t s0, int64_t cap, int until_invalid, EdgeIv *__restrict__ open,
                                                 double *__restrict__ lvl_states) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = st.counters[EC_FRONT] < n_bound ? (int64_t)st.counters[EC_FRONT] : n_bound;
  bool take = false;
  EdgeIv iv{};
  if (i < n) {
    iv = frontier[i];
    take = until_invalid ? !(__longlong_as_double((long long)st.first_inv[iv.e]) <= iv.ta) : (st.edge_ok[iv.e] != 0);
  }
  const uint32_t slot = wave_alloc(take, &st.counters[EC_OPEN]);
  if (take && s0 + (int64_t)slot < cap) {
    open[slot] = iv;
    const double tm = (iv.ta + iv.tb) / 2;
    interpolate_state_dev(sk, st.A + (int64_t)iv.e * sk.S, st.B + (int64_t)iv.e * sk.S, tm, lvl_states + (int64_t)slot * sk.S);
    st.sample_edge[s0 + slot] = iv.e;
    st.sample_t[s0 + slot] = tm;
  }
}

// A level's samples re-dealt in the order `perm` (retraction robots, stored-point forms: cache_merge.hpp: retraction_order sorts the
// level by backbone length, so that a wave of K1r holds backbones of one length).  Slot s0 + j now belongs to what edge_open had
// put in slot s0 + perm[j]: the interval, its midpoint state, and the slot's edge / t entries (the midpoint t by edge_open's expression).
__global__ __launch_bounds__(256) void edge_level_gather(EdgeState st, const int32_t *__restrict__ perm, int64_t m, int S, int64_t s0,
                                                         const EdgeIv *__restrict__ open_in, const double *__restrict__ lvl_in,
                                                         EdgeIv *__restrict__ open_out, double *__restrict__ lvl_out) {
#pragma clang fp contract(off)
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  const int64_t i = perm[j];
  const EdgeIv iv = open_in[i];
  open_out[j] = iv;
  for (int d = 0; d < S; d++) lvl_out[j * S + d] = lvl_in[i * S + d];
  st.sample_edge[s0 + j] = iv.e;
  st.sample_t[s0 + j] = (iv.ta + iv.tb) / 2;
}

// partial.t of checkMotion(s1, s2, last_valid): the largest sampled t below the first invalid one
// (VoxelEnvironment.cpp:403-424)
__global__ __launch_bounds__(256) void edge_last_valid_t(EdgeState st, int64_t slot_lo, int64_t pool) {     // the run's own slots [slot_lo, pool)
  const int64_t s = slot_lo + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= pool) return;
  const int32_t e = st.sample_edge[s];
  if (e < 0) return;
  const double t = st.sample_t[s];
  if (t < __longlong_as_double((long long)st.first_inv[e])) atomicMax(&st.last_t[e], (unsigned long long)__double_as_longlong(t));
}

// *flag != 0 afterwards: some index is outside [0, limit)
// (below = 1: -1, "none", is allowed as well)
__global__ __launch_bounds__(256) void index_range_check(const int32_t *__restrict__ idx, int64_t n, int32_t limit, uint32_t *__restrict__ flag, int below = 0) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n && (uint32_t)(idx[i] + below) >= (uint32_t)limit + (uint32_t)below) *flag = 1u;
}

// rows of a batch's signature array picked by candidate index (the vertex sampler: accepted candidates' rows, in acceptance order)
__global__ __launch_bounds__(256) void gather_signature_rows(const uint32_t *__restrict__ sig, int64_t words, const int64_t *__restrict__ index, int64_t rows,
                                                             int64_t index_base, uint32_t *__restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= rows * words) return;
  const int64_t r = t / words, w = t - r * words;
  out[t] = sig[(index[r] - index_base) * words + w];
}

// bit e = edge e of the run is valid: the run's verdicts as the words of the caller's mask (E bits from bit 0 of out[0])
__global__ __launch_bounds__(256) void edge_ok_bits(const uint32_t *__restrict__ edge_ok, int64_t E, uint64_t *__restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const unsigned long long m = __ballot(e < E && edge_ok[e] != 0u);
  if ((threadIdx.x & 63) == 0 && e < E) out[e >> 6] = m;
}

// bit s = pool sample s belongs to a fully valid edge (its backbone joins the edge's voxel cache)
__global__ __launch_bounds__(256) void edge_sample_bits(EdgeState st, int64_t pool, uint64_t *__restrict__ out) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  bool on = false;
  if (s < pool) { const int32_t e = st.sample_edge[s]; on = e >= 0 && st.edge_ok[e] != 0; }
  const unsigned long long m = __ballot(on);
  if ((threadIdx.x & 63) == 0 && s < pool) out[s >> 6] = m;
}

// Indexed voxel caches (tr_voxelize_edges_indexed): the items of the cache merge.  Items [0, pool) are the pool samples
// with their edges (vertex slots carry -1 there); items pool + 2k, pool + 2k + 1 are the two END vertices of edge k --
// their block lists were built once for all edges -- taken as part of edge k when the edge is fully valid.
__global__ __launch_bounds__(256) void edge_cache_items(EdgeState st, int64_t pool, int64_t E, const int32_t *__restrict__ idx,
                                                        int32_t *__restrict__ item_src, int32_t *__restrict__ item_edge) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= pool + 2 * E) return;
  if (t < pool) {
    const int32_t e = st.sample_edge[t];
    item_src[t] = (int32_t)t;
    item_edge[t] = (e >= 0 && st.edge_ok[e] != 0) ? e : -1;
  } else {
    const int64_t k = (t - pool) >> 1;
    item_src[t] = idx[t - pool];
    item_edge[t] = st.edge_ok[k] != 0 ? (int32_t)k : -1;
  }
}

// ---- discrete variant (VoxelBackboneDiscreteMotionValidator.cpp:9-79; the loop of
// ompl::base::DiscreteMotionValidator::checkMotion): samples a, interpolate(i / nd) for i = 1..nd-1, b.
// Edge e owns pool samples [offs[e], offs[e+1]); all of them are evaluated in one K1 + K2 pass.

// per-edge validSegmentCount and sample count (a and b are always sampled)
__global__ __launch_bounds__(256) void discrete_count(EdgeState st, EdgeSpaceK sk, int64_t E, uint32_t *__restrict__ nd, int64_t *__restrict__ cnt) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= E) return;
  const unsigned n = valid_segment_count_dev(sk, st.A + k * sk.S, st.B + k * sk.S);
  nd[k] = n;
  cnt[k] = (int64_t)(n > 1 ? n : 1) + 1;
}

__device__ inline int64_t edge_of_sample(const int64_t *__restrict__ offs, int64_t E, int64_t q) {
  int64_t lo = 0, hi = E;                   // offs[lo] <= q < offs[hi]
  while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (offs[mid] <= q) lo = mid; else hi = mid; }
  return lo;
}

// e0 = first edge of this pass, q0 = offs[e0]; pool slot of sample q is q - q0
__global__ __launch_bounds__(256) void discrete_samples(EdgeState st, EdgeSpaceK sk, const uint32_t *__restrict__ nd, const int64_t *__restrict__ offs,
                                                        int64_t E, int64_t q0, int64_t m, double *__restrict__ lvl_states) {
#pragma clang fp contract(off)
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= m) return;
  const int64_t e = edge_of_sample(offs, E, q0 + s);
  const int64_t i = q0 + s - offs[e], cnt = offs[e + 1] - offs[e];
  const double *a = st.A + e * sk.S, *b = st.B + e * sk.S;
  double *out = lvl_states + s * sk.S;
  double t;
  if (i == 0) { t = 0.0; for (int k = 0; k < sk.S; k++) out[k] = a[k]; }
  else if (i == cnt - 1) { t = 1.0; for (int k = 0; k < sk.S; k++) out[k] = b[k]; }
  else { t = (double)i / (double)nd[e]; interpolate_state_dev(sk, a, b, t, out); }
  st.sample_edge[s] = (int32_t)e;
  st.sample_t[s] = t;
}

// first invalid sample index per edge (st.nfk doubles as that minimum, initialised to INT32_MAX)
__global__ __launch_bounds__(256) void discrete_fold(EdgeState st, const int64_t *__restrict__ offs, int64_t q0, int64_t m) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= m) return;
  if ((st.bits[s >> 6] >> (s & 63)) & 1ull) return;
  const int32_t e = st.sample_edge[s];
  atomicMin(&st.nfk[e], (int32_t)(q0 + s - offs[e]));
}

// verdict, samples the sequential loop would have evaluated, and v.t (the last valid sample's t)
__global__ __launch_bounds__(256) void discrete_finish(EdgeState st, const uint32_t *__restrict__ nd, const int64_t *__restrict__ offs, int64_t e0, int64_t e1) {
  const int64_t e = e0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= e1) return;
  const int32_t fb = st.nfk[e];
  const int64_t cnt = offs[e + 1] - offs[e];
  const bool ok = fb == 0x7fffffff;
  st.edge_ok[e] = ok ? 1u : 0u;
  st.nfk[e] = ok ? (int32_t)cnt : fb + 1;
  const double t = ok ? 1.0 : (fb <= 1 ? 0.0 : (double)(fb - 1) / (double)nd[e]);
  st.last_t[e] = (unsigned long long)__double_as_longlong(t);
}

#endif  // TRK_EDGE_DEVICE_ONLY

}  // namespace trk
