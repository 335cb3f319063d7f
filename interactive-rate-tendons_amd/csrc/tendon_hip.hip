// tendon_hip.hip -- libtendon_hip.so: C ABI (include/tendon_hip.h) over the HIP kernels in
// fk_kernel.hpp (K1 fk_rk4_batch) and sweep_kernel.hpp (K2 backbone_voxel_sweep,
// K4 cached_blocks_vs_grid).  gfx950 only.
//
// Host-side responsibilities (all per context, none per configuration):
//   * robot constants (get_stiffness_matrices, tendon/TendonRobot.cpp:105-148),
//   * the shared arc-length grid and its RK4 step list (t_range, TendonRobot.cpp:69-84;
//     Boost.odeint integrate_times stepping, call site :458-462),
//   * the tendon-routing table r, r', r'' at every stage abscissa (get_r_info2,
//     tendon/get_r_info.cpp:105-144),
//   * home-shape tendon lengths (home_shape, TendonRobot.cpp:249-314),
//   * device workspace, launches, optional HIP-event timing per kernel.
#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>
#include <ctime>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/tendon_hip.h"
#include "tr_types.hpp"
#include "fk_launch.hpp"
#include "sweep_kernel.hpp"
#include "sphere_kernel.hpp"
#include "verdict_kernel.hpp"
#include "edge_kernel.hpp"
#include "knn_kernel.hpp"
#include "cache_merge.hpp"
#include "env_kernel.hpp"
#include "sample.hpp"

void tr_dev_cache_trim();          // roadmap.hip: frees the idle device buffers of the query objects' cache

namespace {

std::string g_create_error;

struct Workspace {
  int64_t ld = 0;                 // capacity in configurations (multiple of 64)
  double *px = nullptr, *py = nullptr, *pz = nullptr, *acc = nullptr;   // [P][ld]
  double *Li = nullptr;           // [N][ld]
  double *homeLi = nullptr;       // [N][ld]  per-configuration home lengths (retraction only)
  int32_t *np = nullptr;          // [ld]     per-configuration point counts (retraction only)
  uint8_t *conv = nullptr;        // [ld]
  // staging for the host-pointer entry points
  int64_t st_cap = 0;
  double *states = nullptr;       // [st_cap][S]
  uint64_t *bits = nullptr;       // [st_cap/64]
  double *tips = nullptr;         // [st_cap][3]
  uint8_t *flags = nullptr;       // [st_cap]
  double *L = nullptr;            // [st_cap]
  int32_t *npts = nullptr;        // [st_cap]
};

struct EventPair { hipEvent_t a, b; };

constexpr int64_t kRetractSortMin = 8192;   // below this a launch is a few waves per CU at most: ordering them buys nothing
constexpr int64_t kSmallBatch = 4096;   // tr_validate_batch: up to here the single-stream path without a device-wide sync

// device state of the edge frontier (edge_kernel.hpp / edge_host.inc)
struct EdgeDev {
  int64_t cap = 0;                // pool capacity this was sized for; edges per chunk <= cap / 2
  double *lvl_states = nullptr;   // [cap][S]  states of the level being evaluated
  uint64_t *bits = nullptr;       // [cap/64]  K2 verdict per pool sample
  int32_t *sample_edge = nullptr; // [cap]
  double *sample_t = nullptr;     // [cap]
  trk::EdgeIv *open = nullptr, *frontier = nullptr;   // [cap], [2 cap]
  double *A = nullptr, *B = nullptr, *rel = nullptr;  // [cap/2][S] x 2, [cap/2]
  uint32_t *edge_ok = nullptr; int32_t *nfk = nullptr;
  unsigned long long *first_inv = nullptr, *last_t = nullptr;
  uint32_t *counters = nullptr;
  uint32_t *nd = nullptr; int64_t *cnt = nullptr;      // discrete variant: validSegmentCount, sample offsets [cap/2 + 2]
  uint32_t *sig = nullptr; int64_t sig_stride = 0;     // [cap][sig_stride] cell signatures of the pool samples (sweep_kernel.hpp)
  // [cap] point counts of the pool samples whose signatures fk_verdict_retract wrote (tip-aligned rows).  NOT the
  // workspace's n_points: the verdict path's fallback pass re-integrates its few configurations in workspace columns
  // [0, fb_cap) and would overwrite the counts of the pool's first samples
  int32_t *sig_np = nullptr;
  // second copies of a level's intervals and states: a retraction robot's stored-point levels are re-dealt in the order of their
  // backbone lengths (edge_level_gather)
  trk::EdgeIv *open2 = nullptr; double *lvl_states2 = nullptr;
  // the edge queue (edge_queue_kernel.hpp): per-edge level records [cap/2 + 1], control words, the arguments' device copy and pinned image
  int32_t *q_remaining = nullptr, *q_lvl_base = nullptr, *q_lvl_cnt = nullptr;
  uint32_t *q_ctl = nullptr, *q_hctl = nullptr;
  trk::EdgeQueueArgs *q_args = nullptr, *q_hargs = nullptr;
  // indexed forms: the roadmap's vertex states and the edges' index pairs (grow-only)
  double *ix_states = nullptr; int64_t ix_states_cap = 0;
  int32_t *ix_idx = nullptr; int64_t ix_idx_cap = 0;
};

}  // namespace

struct tr_ctx {
  // One workspace per context: the entry points serialise on this mutex, so concurrent callers (the
  // reference calls isValid from OpenMP threads, VoxelCachedLazyPRM.cpp:1448-1455) are safe, one at a time.
  std::recursive_mutex mu;
  int device = 0;
  std::string err;
  RobotK K{};
  std::vector<double> C, D, max_tension;
  // shared arc-length grid
  std::vector<double> t;          // backbone abscissae (s_start = 0)
  std::vector<StepK> steps;
  double *d_tab = nullptr;
  StepK *d_steps = nullptr;
  PolyK *d_poly = nullptr;        // routing polynomials (retraction kernel)
  double *d_tgrid = nullptr;      // [P] abscissae of the s_start = 0 grid (retraction kernel: shared, tip-anchored grid)
  double *d_hl = nullptr;         // [P][N] home-length integrand at those abscissae
  int k_first = 0;                // first step after the grid's own first interval
  // obstacle grid
  bool has_grid = false;
  GridK G{};
  uint64_t *d_grid = nullptr;
  uint64_t *d_near = nullptr;     // obstacle grid dilated by 2 cells (Chebyshev), same layout
  uint32_t n_blocks = 0;
  // VoxelValidityChecker mode (sphere-swept robot, sphere_kernel.hpp): distance to the nearest occupied cell centre
  int checker = TR_CHECKER_BACKBONE;
  float *d_sph_near = nullptr, *d_sph_tmp = nullptr;   // [N^3] each
  bool sph_near_valid = false;
  uint64_t *d_envw[2] = {nullptr, nullptr};   // environment-preparation scratch: two (Nb+2)^3 apron grids
  uint32_t envw_blocks = 0;
  Workspace ws;
  EdgeDev edge;
  trk::MergeScratch merge;       // device-side union of edge voxel caches (cache_merge.hip)
  int64_t max_chunk = 1 << 20;
  int64_t edge_pool_max = 1 << 22; // samples held at once by tr_validate_edges / tr_voxelize_edges
  // tr_validate_edges_indexed through the verdict-only kernels keeps no backbone points: its pool is the per-sample arrays of
  // EdgeDev alone (~0.7 KB per sample against 4.1 KB with the point planes) and may be larger than the FK workspace
  int64_t edge_slots_max = 1 << 24;
  int64_t edge_slots_now = 0;      // slots of the running indexed call (0: the pool is the workspace, ws.ld)
  double ch_scale = 2.0;          // milestone spacing of K2 in robot radii (env TENDON_HIP_CH_SCALE, tuning only)
  int64_t k1_round = 1 << 17;     // configurations in one resident round of K1 waves (CUs x 4 SIMDs x waves/SIMD x 64)
  // tr_validate_batch*: 2 (default) = verdict-only kernel (verdict_kernel.hpp: no point storage) for the backbone checker
  // on robots without retraction, 1 = K1 + K2 as one kernel over stored points (fused_kernel.hpp), 0 = separate launches.
  // env TENDON_HIP_FUSED selects; the edge calls, the sphere checker and the voxel caches need the points and use 1 / 0.
  int fuse = 2;
  int64_t fb_cap = 1 << 17;       // columns of the fallback pass's point workspace: one resident round of waves (tr_create), env TENDON_HIP_FB_CAP
  int32_t *d_fb_list = nullptr; uint32_t *d_fb_count = nullptr; int64_t fb_list_cap = 0;
  // second lane of the edge bisection (edge_host.inc: EdgeLane): its own fallback list, stream, counters
  static constexpr int kMaxLanes = 4;             // lanes of an edge bisection (edge_host.inc): streams, counters, fallback lists, ordering buffers
  int32_t *d_fb_list1[kMaxLanes - 1] = {}; uint32_t *d_fb_count1[kMaxLanes - 1] = {}; int64_t fb_list1_cap[kMaxLanes - 1] = {};   // lanes 1 ..
  hipStream_t edge_stream[kMaxLanes] = {};
  uint32_t *edge_hc[kMaxLanes] = {};              // pinned host images of the lanes' counters
  uint32_t *d_edge_counters1[kMaxLanes - 1] = {}; // lanes 1 .. (lane 0: EdgeDev::counters)
  int edge_lanes = kMaxLanes;                     // TENDON_HIP_EDGE_LANES=1 .. 4: exactly that many lanes (1: one lane only); default: by the edge count
  bool edge_lanes_fixed = false;
  bool edge_kernels_loaded = false;               // tr_reserve_edges has launched every kernel of the indexed edge path once
  // The indexed edge check as ONE persistent launch over a device work queue (edge_queue_kernel.hpp) instead of level-synchronous
  // launches on lanes: the default where it applies (backbone checker, no retraction, verdict-only schedule); TENDON_HIP_EDGE_QUEUE=0
  // keeps the lanes (A/B, tests).  edge_queue_waves: workgroups of that launch (0 = what the device holds at once)
  bool edge_queue = true;
  bool edge_queue_forced = false;                 // TENDON_HIP_EDGE_QUEUE=1: also for the host-array form (default there: the lanes, see validate_edges_indexed_impl)
  int edge_queue_waves = 0;
  uint32_t edge_queue_last[4] = {0, 0, 0, 0};     // the last queue run: samples, rounds (wave batches), samples through the exact sweep, flags
  double edge_rate_seen = 0.0;                    // own samples per edge of this context's last indexed edge call (0 = none yet): sizes
                                                  // the next call's chunks and lanes (a rotating robot's edges take ~10, not ~4)
  bool edge_lane_guess_forced = false;            // TENDON_HIP_EDGE_LANE_GUESS was given: it overrides the rate this context has seen
  double edge_lane_guess = 6.0;                   // own samples per edge assumed when a half is given its share of the pool
                                                  // (TENDON_HIP_EDGE_LANE_GUESS: testing, a small value provokes the overflow path)
  // retraction robots: the batch ordered by backbone length for the verdict-only kernel (cache_merge.hpp: retraction_order);
  // for batches of at least retract_sort_min configurations (TENDON_HIP_RETRACT_SORT=<n>; 0 = never, keep arrival order)
  // (one set of buffers per lane of the edge bisection, launch_verdict's `lane`)
  struct RetractOrder {
    uint32_t *keys[2] = {nullptr, nullptr}; int32_t *vals[2] = {nullptr, nullptr}; int64_t cap = 0;
    int32_t *kbegin = nullptr;          // [cap / 64] per wave of the ordered batch: the step its tip-aligned loop may start at
    double *handoff = nullptr; int64_t handoff_cap = 0;   // [19 + N + S][cap]: fk_retract_prologue -> fk_verdict_retract
    trk::MergeScratch ms;               // radix-sort scratch of lanes 1 .. (lane 0 uses tr_ctx::merge)
  } ro[kMaxLanes];
  int64_t retract_sort_min = kRetractSortMin;
  bool rows_one_step = false;           // behind the grid's own first interval every RK4 step ends in the next row
  bool retract_wave_start = true;       // TENDON_HIP_RETRACT_KBEGIN_OFF (A/B switch of profiles/probe_retract.py): +1 - 2 %
  struct VerdictRing {
    static constexpr int kSlots = 16;
    trk::VerdictArgs *d_slots = nullptr, *h_slots = nullptr;
    hipEvent_t ev[kSlots];
    bool used[kSlots] = {};
    int next = 0;
  } vring;
  struct FusedRing {
    static constexpr int kSlots = 16;
    trk::FusedSweepArgs *d_slots = nullptr, *h_slots = nullptr;   // device ring and its pinned host image
    hipEvent_t ev[kSlots];
    bool used[kSlots] = {};
    int next = 0;
  } fused;
  // scratch of the neighbour search (knn_impl), kept between calls: hipMalloc / hipFree of a dozen buffers per call cost more
  // than the search itself inside a process that holds large allocations (33 against 10 ms at 10^5 states)
  struct KnnScratch { void *p[13] = {}; size_t cap[13] = {}; } knn;
  // block lists of the last tr_voxelize_* call, resident on the device (tr_voxelize_fetch / tr_voxelize_fetch_dev copy
  // them out), and the scratch of the kernels that produce them
  struct VoxStore { uint32_t *ids = nullptr; uint64_t *masks = nullptr; int64_t cap = 0, n = 0; } vstore;
  uint32_t *d_vids = nullptr; uint64_t *d_vmasks = nullptr; int32_t *d_vcounts = nullptr;
  uint64_t *d_vbits = nullptr;
  int64_t vox_cap = 0;
  int32_t *d_item_src = nullptr, *d_item_edge = nullptr; int64_t vox_items_cap = 0;   // items of the indexed edge-cache merge
  // host-buffer pipeline of tr_validate_batch: pinned staging, copy/compute streams
  struct Pipe {
    bool ready = false;
    int64_t chunk = 0;
    hipStream_t s_up = nullptr, s_comp = nullptr, s_down = nullptr;
    double *h_states[2] = {nullptr, nullptr}; double *h_tips[2] = {nullptr, nullptr};
    uint64_t *h_bits[2] = {nullptr, nullptr}; uint8_t *h_flags[2] = {nullptr, nullptr};
    hipEvent_t up[2], done[2], down[2];
  } pipe;
  // last work enqueued by a *_dev entry point on a caller's stream: the small-batch host path waits for it on its own
  // stream instead of synchronising the whole device
  hipEvent_t last_dev_ev = nullptr;
  bool last_dev_used = false;
  hipStream_t last_dev_stream = nullptr;   // stream of that work: a *_dev call on ANOTHER stream first waits for it (begin_dev_work)
  // createRoadmap's vertex phase on the device (sample_host.inc): one batch of candidates (states, verdict bits, tips), the
  // compaction's scratch and counters, the accepted set when the caller wants host arrays
  struct Sampler {
    int64_t cap = 0;
    double *states = nullptr; uint64_t *bits = nullptr; double *tips = nullptr; uint32_t *wprefix = nullptr;
    uint32_t *wprefix2 = nullptr; int64_t wp2_cap = 0;       // tr_compact_rows_dev's own scan scratch
    trk::SampleCounters *d_ctr = nullptr, *h_ctr = nullptr;  // device block and its pinned host image
    int64_t out_cap = 0;
    double *out_states = nullptr, *out_tips = nullptr; int64_t *out_index = nullptr;
    double rate_seen = 0.0;                                  // acceptance rate of this context's last run (sizes the first batch)
    uint32_t *sig = nullptr; int64_t sig_cap = 0;            // [batch][tr_signature_words]: the candidates' signature rows of one batch
    int64_t *sig_index = nullptr; int64_t sig_index_cap = 0; // accepted candidates' indices when the caller keeps none
  } samp;
  // pinned staging for host arrays that are uploaded by a synchronous call (upload_staged)
  void *h_stage = nullptr; size_t h_stage_cap = 0;
  // instrumentation
  bool profiling = false;
  std::vector<EventPair> events[TR_PROFILE_SLOTS];
  uint32_t debug = 0;
};

namespace {

int fail(tr_ctx *ctx, int code, const std::string &msg) {
  if (ctx) ctx->err = msg; else g_create_error = msg;
  return code;
}

#define HIP_TRY(ctx, expr)                                                                   \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(ctx, TR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
  } while (0)

// util/vector_ops.h:67-75 (range) + tendon/TendonRobot.cpp:69-84 (t_range)
std::vector<double> t_range(double start, double end, double dt) {
  std::vector<double> vals;
  for (double p = start; p <= end - (dt / 2); p += dt) vals.push_back(p);
  vals.push_back(end);
  for (auto &v : vals) v = end - (v - start);
  std::reverse(vals.begin(), vals.end());
  return vals;
}

// tendon/get_r_info.cpp:17-40,105-144: routing r, r', r'' of every tendon at arc length t.
// out: N x 6 {rx, ry, rdx, rdy, rddx, rddy} (z components are identically zero).
void routing_at(const tr_ctx *c, double t, double *out) {
  const int N_a = c->K.n_a, N_m = c->K.n_m, N_s = std::max(N_a, N_m);
  double S[TRK_MAX_COEF], Sd[TRK_MAX_COEF], Sdd[TRK_MAX_COEF];
  S[0] = 1; Sd[0] = 0; Sdd[0] = 0;
  if (N_s >= 2) { S[1] = t; Sd[1] = 1; Sdd[1] = 0; }
  for (int i = 2; i < N_s; i++) { S[i] = t * S[i - 1]; Sd[i] = i * S[i - 1]; Sdd[i] = i * (i - 1) * S[i - 2]; }
  for (int j = 0; j < c->K.n_tendons; j++) {
    const double *Cj = &c->C[(size_t)j * N_a], *Dj = &c->D[(size_t)j * N_m];
    double C_a = 0, C_ad = 0, C_add = 0, D_m = 0, D_md = 0, D_mdd = 0;
    for (int i = 0; i < N_a; i++) { C_a += Cj[i] * S[i]; C_ad += Cj[i] * Sd[i]; C_add += Cj[i] * Sdd[i]; }
    for (int i = 0; i < N_m; i++) { D_m += Dj[i] * S[i]; D_md += Dj[i] * Sd[i]; D_mdd += Dj[i] * Sdd[i]; }
    const double sa = std::sin(C_a), ca = std::cos(C_a);
    double *o = out + 6 * j;
    o[0] = D_m * sa;
    o[1] = D_m * ca;
    o[2] = D_md * sa + D_m * (ca * C_ad);
    o[3] = D_md * ca + D_m * (-sa * C_ad);
    o[4] = D_mdd * sa + 2 * D_md * (ca * C_ad) - D_m * (sa * C_ad * C_ad) + D_m * (ca * C_add);
    o[5] = D_mdd * ca + 2 * D_md * (-sa * C_ad) - D_m * (ca * C_ad * C_ad) + D_m * (-sa * C_add);
  }
}

int poly_degree(const double *c, int n) {
  for (int i = n - 1; i > 0; i--) if (std::fabs(c[i]) > 0.0) return i;
  return 0;
}
double poly_at(const double *c, int n, double t) {     // util/poly.h:9-17
  double val = 0.0, tpow = 1;
  for (int i = 0; i < n; i++) { val += c[i] * tpow; tpow *= t; }
  return val;
}

// Home-shape tendon lengths (tendon/TendonRobot.cpp:281-311).  The reference's simpsons()
// (:160-178) reads one element past the end of its input; this uses the rule its comment
// describes (composite Simpson, trapezoid for a trailing odd interval) -- see DESIGN.md.
void home_lengths(const tr_ctx *c, const std::vector<double> &t, double s_start, double *Li) {
  const double Lh = c->K.L - s_start;
  const int N_a = c->K.n_a, N_m = c->K.n_m;
  for (int j = 0; j < c->K.n_tendons; j++) {
    const double *Cj = &c->C[(size_t)j * N_a], *Dj = &c->D[(size_t)j * N_m];
    const int rdeg = poly_degree(Dj, N_m), tdeg = poly_degree(Cj, N_a);
    if (rdeg == 0 && tdeg == 0) {
      Li[j] = Lh;
    } else if (rdeg == 0 && tdeg == 1) {
      const double d0 = Dj[0], c1 = Cj[1];
      Li[j] = Lh * std::sqrt(1 + d0 * d0 * c1 * c1);
    } else {
      double Cdot[TRK_MAX_COEF] = {0}, Ddot[TRK_MAX_COEF] = {0};
      for (int k = 1; k < N_a; k++) Cdot[k - 1] = k * Cj[k];
      for (int k = 1; k < N_m; k++) Ddot[k - 1] = k * Dj[k];
      const int n = (int)t.size();
      std::vector<double> vals(n);
      for (int q = 0; q < n; q++) {
        const double dd = poly_at(Ddot, N_m, t[q]), d = poly_at(Dj, N_m, t[q]), cd = poly_at(Cdot, N_a, t[q]);
        vals[q] = std::sqrt(dd * dd + (d * d) * (cd * cd) + 1);
      }
      const double dx = c->K.dL;
      double res = 0.0;
      if (n >= 2) {
        int nint = n - 1;
        double odd = 0.0;
        if (nint % 2 != 0) { odd = 0.5 * dx * (vals[n - 2] + vals[n - 1]); nint--; }
        if (nint == 0) res = odd;
        else {
          double integral = vals[0] + vals[nint];
          for (int i = 1; i < nint; i++) integral += ((i % 2) ? 4 : 2) * vals[i];
          res = odd + (integral * dx / 3.0);
        }
      }
      Li[j] = res;
    }
  }
}

int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

template <typename T>
int dev_alloc(tr_ctx *ctx, T **p, size_t count) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (count == 0) count = 1;
  hipError_t e = hipMalloc((void **)p, count * sizeof(T));
  if (e == hipErrorOutOfMemory) {        // the query objects' buffer cache (roadmap.hip) may hold idle memory: give it back, try once more
    (void)hipGetLastError();
    tr_dev_cache_trim();
    e = hipMalloc((void **)p, count * sizeof(T));
  }
  HIP_TRY(ctx, e);
  return TR_OK;
}

// Append `count` merged (block id, mask) entries that lie in device memory to the context's block-list store (null stream:
// ordered after the kernels that wrote them).  The store grows by doubling; what it already holds moves device to device.
int vstore_append(tr_ctx *c, const uint32_t *d_ids, const uint64_t *d_masks, int64_t count) {
  tr_ctx::VoxStore &v = c->vstore;
  if (count <= 0) return TR_OK;
  if (v.n + count > v.cap) {
    const int64_t ncap = std::max<int64_t>({v.n + count, 2 * v.cap, (int64_t)1 << 22});
    uint32_t *ni = nullptr; uint64_t *nm = nullptr;
    HIP_TRY(c, hipMalloc((void **)&ni, (size_t)ncap * sizeof(uint32_t)));
    if (hipMalloc((void **)&nm, (size_t)ncap * sizeof(uint64_t)) != hipSuccess) { (void)hipFree(ni); return fail(c, TR_ERR_HIP, "hipMalloc failed (block-list store)"); }
    if (v.n > 0) {
      HIP_TRY(c, hipMemcpyAsync(ni, v.ids, (size_t)v.n * sizeof(uint32_t), hipMemcpyDeviceToDevice, nullptr));
      HIP_TRY(c, hipMemcpyAsync(nm, v.masks, (size_t)v.n * sizeof(uint64_t), hipMemcpyDeviceToDevice, nullptr));
    }
    HIP_TRY(c, hipStreamSynchronize(nullptr));
    if (v.ids) (void)hipFree(v.ids);
    if (v.masks) (void)hipFree(v.masks);
    v.ids = ni; v.masks = nm; v.cap = ncap;
  }
  HIP_TRY(c, hipMemcpyAsync(v.ids + v.n, d_ids, (size_t)count * sizeof(uint32_t), hipMemcpyDeviceToDevice, nullptr));
  HIP_TRY(c, hipMemcpyAsync(v.masks + v.n, d_masks, (size_t)count * sizeof(uint64_t), hipMemcpyDeviceToDevice, nullptr));
  v.n += count;
  return TR_OK;
}

int ensure_workspace(tr_ctx *ctx, int64_t n) {
  Workspace &w = ctx->ws;
  const int64_t want = round_up(std::min<int64_t>(std::max<int64_t>(n, 64), ctx->max_chunk), 64);
  if (w.ld >= want) return TR_OK;
  HIP_TRY(ctx, hipDeviceSynchronize());
  const size_t P = (size_t)ctx->K.n_points, N = (size_t)ctx->K.n_tendons;
  int rc;
  if ((rc = dev_alloc(ctx, &w.px, P * want))) return rc;
  if ((rc = dev_alloc(ctx, &w.py, P * want))) return rc;
  if ((rc = dev_alloc(ctx, &w.pz, P * want))) return rc;
  if ((rc = dev_alloc(ctx, &w.acc, P * want))) return rc;
  if ((rc = dev_alloc(ctx, &w.Li, N * want))) return rc;
  if ((rc = dev_alloc(ctx, &w.conv, (size_t)want))) return rc;
  if (ctx->K.enable_retraction) {
    if ((rc = dev_alloc(ctx, &w.homeLi, N * want))) return rc;
    if ((rc = dev_alloc(ctx, &w.np, (size_t)want))) return rc;
  }
  w.ld = want;
  return TR_OK;
}

int ensure_staging(tr_ctx *ctx, int64_t n) {
  Workspace &w = ctx->ws;
  const int64_t want = round_up(std::max<int64_t>(n, 64), 64);
  if (w.st_cap >= want) return TR_OK;
  HIP_TRY(ctx, hipDeviceSynchronize());
  int rc;
  if ((rc = dev_alloc(ctx, &w.states, (size_t)want * ctx->K.state_size))) return rc;
  if ((rc = dev_alloc(ctx, &w.bits, (size_t)want / 64))) return rc;
  if ((rc = dev_alloc(ctx, &w.tips, (size_t)want * 3))) return rc;
  if ((rc = dev_alloc(ctx, &w.flags, (size_t)want))) return rc;
  if ((rc = dev_alloc(ctx, &w.L, (size_t)want))) return rc;
  if ((rc = dev_alloc(ctx, &w.npts, (size_t)want))) return rc;
  w.st_cap = want;
  return TR_OK;
}

// Host array -> device through a pinned buffer of the context (grow-only): memcpy, then a KERNEL that reads the pinned memory
// over the bus (hipHostMalloc memory is mapped into the device's address space): the same cost as the runtime's staged copy
// without its choice between staging and pinning the caller's pages.  `bytes` must be a multiple of 8 (doubles / int64).
// (How this came about: inside create_roadmap the upload of the 3.2 MB of vertices sometimes completed only after 10 - 35 ms.
// It is not the copy: the FIRST submission of any kind after some calls -- a 2 KB device-to-device kernel on any stream -- can
// take that long, a second one 0.02 ms, and after a 40 ms sleep the first one is fast too: something outside this library keeps
// the device from taking the process's work for a while.  It happens in ~3 - 13 % of the calls of a loop with or without query
// objects, allocation churn, munmap or user-page pinning (profiles/probe_prm_churn.py, probe_after_connect.py, r03/stalls_v1.txt)
// on this pool's shared hosts; medians and minima are what the tables quote.)
int upload_staged(tr_ctx *c, void *d_dst, const void *h_src, size_t bytes, hipStream_t s) {
  if (bytes == 0) return TR_OK;
  if (bytes % 8) return fail(c, TR_ERR_INVALID_ARG, "upload_staged: whole 8-byte words only");
  if (c->h_stage_cap < bytes) {
    if (c->h_stage) { HIP_TRY(c, hipDeviceSynchronize()); (void)hipHostFree(c->h_stage); c->h_stage = nullptr; c->h_stage_cap = 0; }
    const size_t want = bytes + bytes / 4;
    HIP_TRY(c, hipHostMalloc(&c->h_stage, want, hipHostMallocDefault));
    c->h_stage_cap = want;
  }
  std::memcpy(c->h_stage, h_src, bytes);
  const int64_t words = (int64_t)(bytes / 8);
  hipLaunchKernelGGL(trk::copy_words, dim3((unsigned)std::min<int64_t>((words + 255) / 256, 2048)), dim3(256), 0, s, (uint64_t *)d_dst,
                     (const uint64_t *)c->h_stage, words);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(s));                  // the buffer is free for the next call
  return TR_OK;
}

// Device array -> host array through the same pinned buffer: a copy KERNEL writes the mapped pinned memory, then a memcpy into the
// caller's (pageable) array -- 2.3 MB of FK counts took the runtime's staged copy ~1 ms after the edge queue's launch, this ~0.2.
// Synchronises the stream.  `bytes` is rounded up to whole words on the device side (the arrays here are allocated with room).
int download_staged(tr_ctx *c, void *h_dst, const void *d_src, size_t bytes, hipStream_t s) {
  if (bytes == 0) return TR_OK;
  const size_t padded = (bytes + 7) / 8 * 8;
  if (c->h_stage_cap < padded) {
    if (c->h_stage) { HIP_TRY(c, hipDeviceSynchronize()); (void)hipHostFree(c->h_stage); c->h_stage = nullptr; c->h_stage_cap = 0; }
    const size_t want = padded + padded / 4;
    HIP_TRY(c, hipHostMalloc(&c->h_stage, want, hipHostMallocDefault));
    c->h_stage_cap = want;
  }
  const int64_t words = (int64_t)(padded / 8);
  hipLaunchKernelGGL(trk::copy_words, dim3((unsigned)std::min<int64_t>((words + 255) / 256, 2048)), dim3(256), 0, s, (uint64_t *)c->h_stage,
                     (const uint64_t *)d_src, words);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(s));
  std::memcpy(h_dst, c->h_stage, bytes);
  return TR_OK;
}

struct ProfScope {
  tr_ctx *ctx; int slot; hipStream_t s; EventPair ev{};
  bool on;
  ProfScope(tr_ctx *c, int slot_, hipStream_t s_) : ctx(c), slot(slot_), s(s_), on(c->profiling) {
    if (on) {
      (void)hipEventCreate(&ev.a); (void)hipEventCreate(&ev.b);
      (void)hipEventRecord(ev.a, s);
    }
  }
  ~ProfScope() {
    if (on) { (void)hipEventRecord(ev.b, s); ctx->events[slot].push_back(ev); }
  }
};

// The *_dev entry points that use per-context scratch (fallback list and counter, workspace columns, ordering buffers, argument
// rings, the sampler's batch) bracket their work with these two.  Calls on ONE stream are ordered by the stream; a call that
// arrives on a DIFFERENT stream than the previous one first makes its stream wait for that call's work (a device-side
// dependency, no host synchronisation), so two caller streams can never run on the shared scratch at once -- calls on one
// context execute in the order they were issued, whichever streams they name.
int begin_dev_work(tr_ctx *ctx, hipStream_t s) {
  if (ctx->last_dev_used && ctx->last_dev_stream != s) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->last_dev_ev, 0));
  return TR_OK;
}
// ... and this after enqueueing their work
int note_dev_work(tr_ctx *ctx, hipStream_t s) {
  if (!ctx->last_dev_ev) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->last_dev_ev, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventRecord(ctx->last_dev_ev, s));
  ctx->last_dev_used = true;
  ctx->last_dev_stream = s;
  return TR_OK;
}

// ---- K1 launch (instantiations live in fk_inst.hip objects) ------------------------------------
int launch_fk(tr_ctx *ctx, const double *d_states, int64_t n, int64_t ld, const trk::FkOut &out, hipStream_t s, int lane = 0) {
  if (n <= 0) return TR_OK;
  ProfScope ps(ctx, 0, s);
  const bool ret = ctx->K.enable_retraction;
  trk::FkLaunch a{d_states, n, ld, ctx->K, (bool)ctx->K.enable_rotation, out.R != nullptr, ctx->d_tab, ctx->d_steps,
                  (int)ctx->steps.size(), ctx->d_poly, ctx->k_first, ctx->d_tgrid, ctx->d_hl, out, s};
  if (ret && ctx->retract_sort_min > 0 && n >= ctx->retract_sort_min) {
    // A large batch of a retraction robot is integrated in the order of its backbone lengths, as the verdict-only kernels do
    // (a wave runs from its LONGEST backbone's base to the tip, the shorter ones idle: in arrival order half of the lane-steps
    // are masked), and every stored output goes to its configuration's own column: same planes, bit for bit.
    tr_ctx::RetractOrder &ro = ctx->ro[lane];           // (a lane of the edge bisection: its own buffers and sort scratch, as launch_verdict's)
    int rc;
    if (ro.cap < n) {
      HIP_TRY(ctx, hipDeviceSynchronize());
      const int64_t want = round_up(n, 64);
      for (int q = 0; q < 2; q++) {
        if ((rc = dev_alloc(ctx, &ro.keys[q], (size_t)want))) return rc;
        if ((rc = dev_alloc(ctx, &ro.vals[q], (size_t)want))) return rc;
      }
      if ((rc = dev_alloc(ctx, &ro.kbegin, (size_t)want / 64))) return rc;
      ro.cap = want;
    }
    const int32_t *perm = nullptr;
    const hipError_t e = trk::retraction_order(lane ? ro.ms : ctx->merge, d_states, n, ctx->K.state_size, ctx->K.L, ro.keys, ro.vals, &perm, s,
                                               ctx->K.dL, ctx->k_first, ctx->rows_one_step, ro.kbegin);
    if (e != hipSuccess) return fail(ctx, TR_ERR_HIP, std::string("retraction order: ") + hipGetErrorString(e));
    a.d_perm = perm;
    a.d_wave_k_begin = ctx->retract_wave_start ? ro.kbegin : nullptr;
  }
  switch (ctx->K.n_tendons) {
#define TRK_CASE(N) case N: if (ret) trk::launch_fk_retract<N>(a); else trk::launch_fk_uniform<N>(a); break;
    TRK_CASE(1) TRK_CASE(2) TRK_CASE(3) TRK_CASE(4) TRK_CASE(5) TRK_CASE(6) TRK_CASE(7) TRK_CASE(8)
#undef TRK_CASE
    default: return fail(ctx, TR_ERR_OUT_OF_RANGE, "n_tendons out of range");
  }
  HIP_TRY(ctx, hipGetLastError());
  return TR_OK;
}

// milestone spacing for the LDS self-collision proof: about two robot radii of arc per chunk,
// coarser if needed to keep the per-wave LDS image (4 * NM * 64 floats) within 48 KiB
void sweep_geometry(const tr_ctx *ctx, int &CH, int &NM, size_t &lds) {
  const int P = ctx->K.n_points;
  CH = (int)std::lround(ctx->ch_scale * ctx->K.radius / ctx->K.dL);
  if (CH < 1) CH = 1;
  while ((P - 1 + CH - 1) / CH + 1 > 48) CH++;
  NM = (P - 1 + CH - 1) / CH + 1;
  lds = (size_t)4 * NM * 64 * sizeof(float) + (128 + 64) * sizeof(uint32_t) + (size_t)6 * 128 * sizeof(double);   // milestones + deferred-walk ring, flags and the ring's end points
}

int launch_sweep(tr_ctx *ctx, const trk::SweepIn &in, int64_t n, int64_t ld, int check_voxels,
                 uint64_t *d_bits, uint8_t *d_flags, hipStream_t s) {
  if (n <= 0) return TR_OK;
  if (check_voxels && !ctx->has_grid) return fail(ctx, TR_ERR_INVALID_ARG, "no obstacle grid set (tr_set_grid)");
  ProfScope ps(ctx, 1, s);
  int CH, NM; size_t lds;
  sweep_geometry(ctx, CH, NM, lds);
  const unsigned grid = (unsigned)((n + 63) / 64);
  hipLaunchKernelGGL(trk::backbone_voxel_sweep, dim3(grid), dim3(64), lds, s, in, n, ld, ctx->K.n_points, CH, NM, ctx->K,
                     ctx->G, ctx->d_grid, ctx->d_near, check_voxels, ctx->debug, d_bits, d_flags);
  HIP_TRY(ctx, hipGetLastError());
  return TR_OK;
}

// A device copy of the sweep's arguments for one fused launch: a small ring of device slots filled from pinned host
// memory (an asynchronous copy on the launch stream), each guarded by an event so a slot is not rewritten while a
// launch may still read it.  *slot_out receives the slot (record fr.ev[slot] on the stream after the launch).
int fused_args_slot(tr_ctx *ctx, const trk::SweepIn &in, int check_voxels, uint64_t *d_bits, uint8_t *d_flags, hipStream_t s,
                    const trk::FusedSweepArgs **d_args, size_t *lds, int *slot_out, uint32_t *sig = nullptr, int64_t sig_stride = 0) {
  tr_ctx::FusedRing &fr = ctx->fused;
  if (!fr.d_slots) {
    HIP_TRY(ctx, hipMalloc((void **)&fr.d_slots, sizeof(trk::FusedSweepArgs) * tr_ctx::FusedRing::kSlots));
    HIP_TRY(ctx, hipHostMalloc((void **)&fr.h_slots, sizeof(trk::FusedSweepArgs) * tr_ctx::FusedRing::kSlots, hipHostMallocDefault));
    for (auto &e : fr.ev) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  const int slot = fr.next;
  fr.next = (fr.next + 1) % tr_ctx::FusedRing::kSlots;
  // the slot's previous launch (kSlots launches ago, possibly on another stream) must be done before its
  // host image is rewritten and its device copy replaced
  if (fr.used[slot]) HIP_TRY(ctx, hipEventSynchronize(fr.ev[slot]));
  trk::FusedSweepArgs &a = fr.h_slots[slot];
  a = trk::FusedSweepArgs{};
  a.in = in;
  sweep_geometry(ctx, a.CH, a.NM, *lds);
  if (sig) *lds = std::max(*lds, (size_t)trk::SIG_LDS_WORDS * 4);       // the signature tile shares the sweep's image (sweep_kernel.hpp: SigStage)
  a.P = ctx->K.n_points; a.check_voxels = check_voxels; a.debug = ctx->debug;
  a.g = ctx->G; a.grid = ctx->d_grid; a.near_grid = ctx->d_near; a.valid_bits = d_bits; a.flags = d_flags;
  a.sig = sig; a.sig_stride = sig_stride;
  HIP_TRY(ctx, hipMemcpyAsync(fr.d_slots + slot, &a, sizeof(a), hipMemcpyHostToDevice, s));   // pinned source: asynchronous
  *d_args = fr.d_slots + slot;
  *slot_out = slot;
  return TR_OK;
}

// K1 + K2 in one launch (fused_kernel.hpp; shared arc-length grid only).
int launch_fused(tr_ctx *ctx, const double *d_states, int64_t n, int64_t ld, const trk::FkOut &out, const trk::SweepIn &in,
                 int check_voxels, uint64_t *d_bits, uint8_t *d_flags, hipStream_t s, uint32_t *sig = nullptr, int64_t sig_stride = 0) {
  if (n <= 0) return TR_OK;
  if (check_voxels && !ctx->has_grid) return fail(ctx, TR_ERR_INVALID_ARG, "no obstacle grid set (tr_set_grid)");
  const trk::FusedSweepArgs *d_args; size_t lds; int slot, rc;
  if ((rc = fused_args_slot(ctx, in, check_voxels, d_bits, d_flags, s, &d_args, &lds, &slot, sig, sig_stride))) return rc;
  {
    ProfScope ps(ctx, 4, s);
    const trk::FkLaunch fl{d_states, n, ld, ctx->K, (bool)ctx->K.enable_rotation, false, ctx->d_tab, ctx->d_steps,
                           (int)ctx->steps.size(), ctx->d_poly, ctx->k_first, ctx->d_tgrid, ctx->d_hl, out, s};
    switch (ctx->K.n_tendons) {
#define TRK_CASE(N) case N: trk::launch_fk_sweep_fused<N>(fl, d_args, lds); break;
      TRK_CASE(1) TRK_CASE(2) TRK_CASE(3) TRK_CASE(4) TRK_CASE(5) TRK_CASE(6) TRK_CASE(7) TRK_CASE(8)
#undef TRK_CASE
      default: return fail(ctx, TR_ERR_OUT_OF_RANGE, "n_tendons out of range");
    }
    HIP_TRY(ctx, hipGetLastError());
  }
  HIP_TRY(ctx, hipEventRecord(ctx->fused.ev[slot], s));
  ctx->fused.used[slot] = true;
  return TR_OK;
}

// The verdict path of tr_validate_batch* (verdict_kernel.hpp): fk_verdict over the n configurations -- no backbone point
// is stored -- then ONE launch of fk_sweep_fused_list, which integrates again, with stored points, the configurations whose
// self-collision test needs the exact pairwise sweep (their indices and count stay on the device; with an empty list its
// blocks return at once).  d_bits / d_flags / d_tips as in tr_validate_batch_dev; n <= 2^31.
int ensure_sphere_near(tr_ctx *c, hipStream_t s);
int launch_verdict(tr_ctx *ctx, const double *d_states, int64_t n, uint64_t *d_bits, double *d_tips, uint8_t *d_flags, hipStream_t s,
                   uint32_t *sig = nullptr, int64_t sig_stride = 0, bool spheres = false, int32_t *np_out = nullptr, int lane = 0) {
  if (n <= 0) return TR_OK;
  if (!ctx->has_grid) return fail(ctx, TR_ERR_INVALID_ARG, "no obstacle grid set (tr_set_grid)");
  int rc;
  if (spheres && (rc = ensure_sphere_near(ctx, s))) return rc;
  Workspace &w = ctx->ws;
  // lanes 1 .. (the further lanes of an edge bisection, running concurrently on their own streams): their own list and counter, and
  // the workspace columns [lane fb_cap, (lane + 1) fb_cap) for their fallback pass
  if (lane < 0 || lane >= tr_ctx::kMaxLanes) return fail(ctx, TR_ERR_RUNTIME, "bad lane");
  int32_t *&fb_list = lane ? ctx->d_fb_list1[lane - 1] : ctx->d_fb_list;
  uint32_t *&fb_count = lane ? ctx->d_fb_count1[lane - 1] : ctx->d_fb_count;
  int64_t &fb_list_cap = lane ? ctx->fb_list1_cap[lane - 1] : ctx->fb_list_cap;
  if (fb_list_cap < n) {
    HIP_TRY(ctx, hipDeviceSynchronize());
    if ((rc = dev_alloc(ctx, &fb_list, (size_t)round_up(n, 64)))) return rc;
    if (!fb_count && (rc = dev_alloc(ctx, &fb_count, 1))) return rc;
    fb_list_cap = round_up(n, 64);
  }
  const int64_t cap = std::min<int64_t>(ctx->fb_cap, round_up(n, 64));
  if ((rc = ensure_workspace(ctx, lane ? (lane + 1) * ctx->fb_cap : cap))) return rc;
  const int64_t fcol = lane * ctx->fb_cap;                        // first workspace column of this lane's fallback pass
  tr_ctx::VerdictRing &vr = ctx->vring;
  if (!vr.d_slots) {
    HIP_TRY(ctx, hipMalloc((void **)&vr.d_slots, sizeof(trk::VerdictArgs) * tr_ctx::VerdictRing::kSlots));
    HIP_TRY(ctx, hipHostMalloc((void **)&vr.h_slots, sizeof(trk::VerdictArgs) * tr_ctx::VerdictRing::kSlots, hipHostMallocDefault));
    for (auto &e : vr.ev) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  const int vslot = vr.next;
  vr.next = (vr.next + 1) % tr_ctx::VerdictRing::kSlots;
  if (vr.used[vslot]) HIP_TRY(ctx, hipEventSynchronize(vr.ev[vslot]));
  trk::VerdictArgs &a = vr.h_slots[vslot];
  a = trk::VerdictArgs{};
  size_t lds_k2;
  sweep_geometry(ctx, a.CH, a.NM, lds_k2);
  a.P = ctx->K.n_points; a.debug = ctx->debug;
  a.g = ctx->G; a.grid = ctx->d_grid; a.near_grid = ctx->d_near; a.valid_bits = d_bits; a.flags = d_flags;
  {
    const GridK &g = ctx->G;            // sweep_body's margin box, same expressions (this file is compiled without contraction)
    a.box[0] = g.xmin + 1e-6 * (g.xmax - g.xmin); a.box[1] = g.xmax - 1e-6 * (g.xmax - g.xmin);
    a.box[2] = g.ymin + 1e-6 * (g.ymax - g.ymin); a.box[3] = g.ymax - 1e-6 * (g.ymax - g.ymin);
    a.box[4] = g.zmin + 1e-6 * (g.zmax - g.zmin); a.box[5] = g.zmax - 1e-6 * (g.zmax - g.zmin);
  }
  a.fb_list = fb_list; a.fb_count = fb_count;
  a.sig = sig; a.sig_stride = sig_stride; a.np_out = np_out;
  a.radius = ctx->K.radius;
  for (int j = 0; j < TRK_MAX_TENDONS; j++) { a.home_Li[j] = ctx->K.home_Li[j]; a.min_len[j] = ctx->K.min_len[j]; a.max_len[j] = ctx->K.max_len[j]; }
  if (spheres) {                       // classification thresholds (verdict_kernel.hpp: PointSweep<true>)
    a.field = ctx->d_sph_near; a.radius = ctx->K.radius;
    a.r_lo = (float)ctx->K.radius - 2e-6f; a.r_hi = (float)ctx->K.radius + 2e-6f;
  }
  if (ctx->K.enable_retraction && ctx->retract_sort_min > 0 && n >= ctx->retract_sort_min) {
    // waves of one backbone length: see retraction_order.  The mask is filled by atomic ORs, so it starts from zero.
    tr_ctx::RetractOrder &ro = ctx->ro[lane];
    if (ro.cap < n) {
      HIP_TRY(ctx, hipDeviceSynchronize());
      const int64_t want = round_up(n, 64);
      for (int q = 0; q < 2; q++) {
        if ((rc = dev_alloc(ctx, &ro.keys[q], (size_t)want))) return rc;
        if ((rc = dev_alloc(ctx, &ro.vals[q], (size_t)want))) return rc;
      }
      if ((rc = dev_alloc(ctx, &ro.kbegin, (size_t)want / 64))) return rc;
      ro.cap = want;
    }
    const hipError_t e = trk::retraction_order(lane ? ro.ms : ctx->merge, d_states, n, ctx->K.state_size, ctx->K.L, ro.keys, ro.vals, &a.perm, s,
                                               ctx->K.dL, ctx->k_first, ctx->rows_one_step, ro.kbegin);
    a.wave_k_begin = ctx->retract_wave_start ? ro.kbegin : nullptr;
    if (e != hipSuccess) return fail(ctx, TR_ERR_HIP, std::string("retraction order: ") + hipGetErrorString(e));
    HIP_TRY(ctx, hipMemsetAsync(d_bits, 0, (size_t)((n + 63) / 64) * sizeof(uint64_t), s));
  }
  a.finish_hot();
  HIP_TRY(ctx, hipMemcpyAsync(vr.d_slots + vslot, &a, sizeof(a), hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipMemsetAsync(fb_count, 0, sizeof(uint32_t), s));
  // the fallback pass sweeps columns of the small point workspace
  const bool ret = ctx->K.enable_retraction;      // K1r's body in both launches, tip-aligned rows in the fallback workspace
  trk::SweepIn in{w.px + fcol, w.py + fcol, w.pz + fcol, ret ? w.np + fcol : nullptr, w.Li + fcol, w.conv + fcol, ret ? w.homeLi + fcol : nullptr,
                  w.acc + fcol};
  const trk::FusedSweepArgs *d_fargs; size_t lds_f; int fslot;
  if ((rc = fused_args_slot(ctx, in, spheres ? 2 : 1, d_bits, d_flags, s, &d_fargs, &lds_f, &fslot))) return rc;
  const trk::FkOut vout{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, d_tips, nullptr, nullptr, nullptr};
  trk::FkLaunch vl{d_states, n, 0, ctx->K, (bool)ctx->K.enable_rotation, false, ctx->d_tab, ctx->d_steps,
                   (int)ctx->steps.size(), ctx->d_poly, ctx->k_first, ctx->d_tgrid, ctx->d_hl, vout, s};
  if (ctx->K.enable_retraction) {
    // the prologue kernel's hand-over planes (fk_retract_prologue -> fk_verdict_retract), one set per lane of the edge bisection
    tr_ctx::RetractOrder &ro = ctx->ro[lane];
    const int64_t hld = round_up(n, 64);
    if (ro.handoff_cap < hld) {
      HIP_TRY(ctx, hipDeviceSynchronize());
      const int64_t want = hld + hld / 8;
      if ((rc = dev_alloc(ctx, &ro.handoff, (size_t)want * (19 + 2 * TRK_MAX_TENDONS + 3)))) return rc;     // R, v, u, p | L_i | converged | state | L
      ro.handoff_cap = want;
    }
    vl.d_handoff = ro.handoff; vl.handoff_ld = hld; vl.d_perm = a.perm;
  }
  const trk::FkOut fout{w.px + fcol, w.py + fcol, w.pz + fcol, nullptr, nullptr, w.Li + fcol, nullptr, w.conv + fcol, ret ? w.np + fcol : nullptr,
                        ret ? w.homeLi + fcol : nullptr};
  const trk::FkLaunch fl{d_states, cap, w.ld, ctx->K, (bool)ctx->K.enable_rotation, false, ctx->d_tab, ctx->d_steps,
                         (int)ctx->steps.size(), ctx->d_poly, ctx->k_first, ctx->d_tgrid, ctx->d_hl, fout, s};
  const size_t lds_v = trk::verdict_lds_bytes(a.NM, sig != nullptr);
  {
    ProfScope ps(ctx, 5, s);
    switch (ctx->K.n_tendons) {
#define TRK_CASE(N) case N: if (ret) trk::launch_fk_verdict_retract<N>(vl, vr.d_slots + vslot, lds_v, spheres, sig != nullptr); \
                            else trk::launch_fk_verdict<N>(vl, vr.d_slots + vslot, lds_v, spheres, sig != nullptr); break;
      TRK_CASE(1) TRK_CASE(2) TRK_CASE(3) TRK_CASE(4) TRK_CASE(5) TRK_CASE(6) TRK_CASE(7) TRK_CASE(8)
#undef TRK_CASE
      default: return fail(ctx, TR_ERR_OUT_OF_RANGE, "n_tendons out of range");
    }
    HIP_TRY(ctx, hipGetLastError());
  }
  {
    ProfScope ps(ctx, 4, s);
    switch (ctx->K.n_tendons) {
#define TRK_CASE(N) case N: if (ret) trk::launch_fk_sweep_retract_list<N>(fl, d_fargs, lds_f, fb_list, fb_count); \
                            else trk::launch_fk_sweep_fused_list<N>(fl, d_fargs, lds_f, fb_list, fb_count); break;
      TRK_CASE(1) TRK_CASE(2) TRK_CASE(3) TRK_CASE(4) TRK_CASE(5) TRK_CASE(6) TRK_CASE(7) TRK_CASE(8)
#undef TRK_CASE
      default: break;
    }
    HIP_TRY(ctx, hipGetLastError());
  }
  HIP_TRY(ctx, hipEventRecord(vr.ev[vslot], s));
  vr.used[vslot] = true;
  HIP_TRY(ctx, hipEventRecord(ctx->fused.ev[fslot], s));
  ctx->fused.used[fslot] = true;
  return TR_OK;
}

// The edge queue's launch (edge_queue_kernel.hpp: fk_edge_queue): the verdict-only body's arguments as launch_verdict forms them
// (backbone checker, signature rows into `sig`), the in-wave fallback's sweep arguments over the waves' own workspace columns, and
// `waves` persistent workgroups.  qa: the queue's arguments, already on the device.  The workspace must hold 64 x waves columns.
int launch_edge_queue(tr_ctx *ctx, hipStream_t s, uint32_t *sig, int64_t sig_stride, const trk::EdgeQueueArgs *qa, unsigned waves) {
  if (!ctx->has_grid) return fail(ctx, TR_ERR_INVALID_ARG, "no obstacle grid set (tr_set_grid)");
  Workspace &w = ctx->ws;
  if (w.ld < (int64_t)waves * 64) return fail(ctx, TR_ERR_RUNTIME, "edge queue: workspace smaller than the launch");
  tr_ctx::VerdictRing &vr = ctx->vring;
  if (!vr.d_slots) {
    HIP_TRY(ctx, hipMalloc((void **)&vr.d_slots, sizeof(trk::VerdictArgs) * tr_ctx::VerdictRing::kSlots));
    HIP_TRY(ctx, hipHostMalloc((void **)&vr.h_slots, sizeof(trk::VerdictArgs) * tr_ctx::VerdictRing::kSlots, hipHostMallocDefault));
    for (auto &e : vr.ev) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  const int vslot = vr.next;
  vr.next = (vr.next + 1) % tr_ctx::VerdictRing::kSlots;
  if (vr.used[vslot]) HIP_TRY(ctx, hipEventSynchronize(vr.ev[vslot]));
  trk::VerdictArgs &a = vr.h_slots[vslot];
  a = trk::VerdictArgs{};
  size_t lds_k2;
  sweep_geometry(ctx, a.CH, a.NM, lds_k2);
  a.P = ctx->K.n_points; a.debug = ctx->debug;
  a.g = ctx->G; a.grid = ctx->d_grid; a.near_grid = ctx->d_near;
  {
    const GridK &g = ctx->G;            // sweep_body's margin box, same expressions (this file is compiled without contraction)
    a.box[0] = g.xmin + 1e-6 * (g.xmax - g.xmin); a.box[1] = g.xmax - 1e-6 * (g.xmax - g.xmin);
    a.box[2] = g.ymin + 1e-6 * (g.ymax - g.ymin); a.box[3] = g.ymax - 1e-6 * (g.ymax - g.ymin);
    a.box[4] = g.zmin + 1e-6 * (g.zmax - g.zmin); a.box[5] = g.zmax - 1e-6 * (g.zmax - g.zmin);
  }
  a.sig = sig; a.sig_stride = sig_stride;
  a.radius = ctx->K.radius;
  for (int j = 0; j < TRK_MAX_TENDONS; j++) { a.home_Li[j] = ctx->K.home_Li[j]; a.min_len[j] = ctx->K.min_len[j]; a.max_len[j] = ctx->K.max_len[j]; }
  a.finish_hot();
  HIP_TRY(ctx, hipMemcpyAsync(vr.d_slots + vslot, &a, sizeof(a), hipMemcpyHostToDevice, s));
  int rc;
  trk::SweepIn in{w.px, w.py, w.pz, nullptr, w.Li, w.conv, nullptr, w.acc};
  const trk::FusedSweepArgs *d_fargs; size_t lds_f; int fslot;
  if ((rc = fused_args_slot(ctx, in, 1, nullptr, nullptr, s, &d_fargs, &lds_f, &fslot))) return rc;
  const size_t lds_v = std::max(trk::verdict_lds_bytes(a.NM, true) + (size_t)trk::EQ_STASH_WORDS * 4, lds_f);   // (+ the waves' stash: edge_queue_kernel.hpp)
  const trk::FkOut none{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const trk::FkLaunch fl{nullptr, 0, 0, ctx->K, (bool)ctx->K.enable_rotation, false, ctx->d_tab, ctx->d_steps,
                         (int)ctx->steps.size(), ctx->d_poly, ctx->k_first, ctx->d_tgrid, ctx->d_hl, none, s};
  {
    ProfScope ps(ctx, 5, s);
    switch (ctx->K.n_tendons) {
#define TRK_CASE(N) case N: trk::launch_fk_edge_queue<N>(fl, vr.d_slots + vslot, lds_v, qa, d_fargs, waves); break;
      TRK_CASE(1) TRK_CASE(2) TRK_CASE(3) TRK_CASE(4) TRK_CASE(5) TRK_CASE(6) TRK_CASE(7) TRK_CASE(8)
#undef TRK_CASE
      default: return fail(ctx, TR_ERR_OUT_OF_RANGE, "n_tendons out of range");
    }
    HIP_TRY(ctx, hipGetLastError());
  }
  HIP_TRY(ctx, hipEventRecord(vr.ev[vslot], s));
  vr.used[vslot] = true;
  HIP_TRY(ctx, hipEventRecord(ctx->fused.ev[fslot], s));
  ctx->fused.used[fslot] = true;
  return TR_OK;
}

// persistent workgroups of the edge queue's launch: what the device holds at once (occupancy x CUs)
int edge_queue_waves(tr_ctx *ctx) {
  if (ctx->edge_queue_waves > 0) return ctx->edge_queue_waves;
  int CH, NM; size_t lds_k2;
  sweep_geometry(ctx, CH, NM, lds_k2);
  const size_t lds_v = std::max(trk::verdict_lds_bytes(NM, true) + (size_t)trk::EQ_STASH_WORDS * 4, lds_k2);
  int per_cu = 0;
  switch (ctx->K.n_tendons) {
#define TRK_CASE(N) case N: per_cu = trk::fk_edge_queue_waves_per_cu<N>((bool)ctx->K.enable_rotation, lds_v); break;
    TRK_CASE(1) TRK_CASE(2) TRK_CASE(3) TRK_CASE(4) TRK_CASE(5) TRK_CASE(6) TRK_CASE(7) TRK_CASE(8)
#undef TRK_CASE
    default: break;
  }
  hipDeviceProp_t prop;
  if (per_cu <= 0 || hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) return 0;
  ctx->edge_queue_waves = per_cu * prop.multiProcessorCount;
  return ctx->edge_queue_waves;
}

// Distance from every cell centre to the nearest occupied cell centre, exact within the window that matters
// (r + a cell diagonal), TRK_EDT_FAR beyond: three separable min-plus passes (x on the bit grid, then y, z).
int ensure_sphere_near(tr_ctx *c, hipStream_t s) {
  if (c->sph_near_valid) return TR_OK;
  HIP_TRY(c, hipDeviceSynchronize());
  const size_t ncell = (size_t)c->G.N * c->G.N * c->G.N;
  if (c->d_sph_near) { (void)hipFree(c->d_sph_near); c->d_sph_near = nullptr; }
  if (c->d_sph_tmp) { (void)hipFree(c->d_sph_tmp); c->d_sph_tmp = nullptr; }
  HIP_TRY(c, hipMalloc((void **)&c->d_sph_near, ncell * sizeof(float)));
  HIP_TRY(c, hipMalloc((void **)&c->d_sph_tmp, ncell * sizeof(float)));
  const double reach = c->K.radius + std::sqrt(c->G.dx * c->G.dx + c->G.dy * c->G.dy + c->G.dz * c->G.dz);
  const int R[3] = {(int)std::ceil(reach / c->G.dx) + 1, (int)std::ceil(reach / c->G.dy) + 1, (int)std::ceil(reach / c->G.dz) + 1};
  const unsigned gcell = (unsigned)((ncell + 255) / 256);
  hipLaunchKernelGGL(trk::obstacle_distance_x, dim3(c->n_blocks), dim3(64), 0, s, c->d_grid, c->d_sph_near, c->G.Nb, R[0], (float)c->G.dx);
  hipLaunchKernelGGL(trk::obstacle_distance_axis, dim3(gcell), dim3(256), 0, s, c->d_sph_near, c->d_sph_tmp, c->G.N, 1, R[1], (float)c->G.dy, 0);
  hipLaunchKernelGGL(trk::obstacle_distance_axis, dim3(gcell), dim3(256), 0, s, c->d_sph_tmp, c->d_sph_near, c->G.N, 2, R[2], (float)c->G.dz, 1);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(s));
  c->sph_near_valid = true;
  return TR_OK;
}

// K1 then K2 on one stream: a single fused launch when the robot uses the shared arc-length grid.
// voxel_test: 0 = is_valid_shape only, 1 = backbone voxels (VoxelBackboneValidityChecker), 2 = sphere-swept
// robot (VoxelValidityChecker: K2 without the voxel test, then K8 on the survivors).
// sig (optional, edge samples): the fused launch also writes the samples' cell signatures; use edge_signatures(ctx, ..) to
// learn whether this context's fused path is active (the separate kernels write none).  points_unused: the caller will not
// read the points of this launch (out.px .. may then stay unwritten).
// with_points: the caller also needs the samples' stored points (voxel caches).  Retraction robots then run K1r -> K2 as
// separate launches, which write no signatures (the interval test reads the points); without points their samples go
// through fk_verdict_retract, which does.
bool edge_signatures(const tr_ctx *ctx, bool with_points) {
  return ctx->K.enable_retraction ? (ctx->fuse == 2 && !with_points) : ctx->fuse != 0;
}

int launch_fk_sweep(tr_ctx *ctx, const double *d_states, int64_t n, int64_t ld, const trk::FkOut &out, const trk::SweepIn &in,
                    int voxel_test, uint64_t *d_bits, uint8_t *d_flags, hipStream_t s, uint32_t *sig = nullptr, int64_t sig_stride = 0,
                    bool points_unused = false, int32_t *sig_np = nullptr /* retraction: the samples' point counts, next to sig */,
                    int lane = 0) {
  const int check_voxels = voxel_test == 1 ? 1 : 0;
  int rc;
  if (voxel_test == 2) {
    if (!ctx->has_grid) return fail(ctx, TR_ERR_INVALID_ARG, "no obstacle grid set (tr_set_grid)");
    if ((rc = ensure_sphere_near(ctx, s))) return rc;
  }
  if (ctx->fuse == 2 && voxel_test != 0 && points_unused && !out.R && !out.L && !out.tips) {
    // nobody reads this launch's backbone points (edge samples of the checkMotion forms: the bisection compares cell
    // signatures -- of tip-aligned rows for retraction robots, hence the point counts): the verdict-only kernel, which stores none
    return launch_verdict(ctx, d_states, n, d_bits, nullptr, d_flags, s, sig, sig_stride, voxel_test == 2, sig ? sig_np : nullptr, lane);
  }
  if (ctx->fuse != 0 && !ctx->K.enable_retraction && !out.R && !out.L) {      // the fused kernel integrates neither R output nor L
    if ((rc = launch_fused(ctx, d_states, n, ld, out, in, check_voxels, d_bits, d_flags, s, sig, sig_stride))) return rc;
  } else {
    if ((rc = launch_fk(ctx, d_states, n, ld, out, s, lane))) return rc;
    if ((rc = launch_sweep(ctx, in, n, ld, check_voxels, d_bits, d_flags, s))) return rc;
  }
  if (voxel_test == 2) {
    ProfScope ps(ctx, 3, s);
    hipLaunchKernelGGL(trk::spheres_vs_grid, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, in.px, in.py, in.pz, in.n_points, n, ld,
                       (int)ctx->K.n_points, ctx->K.radius, ctx->G, ctx->d_grid, ctx->d_sph_near, d_bits, d_flags);
    HIP_TRY(ctx, hipGetLastError());
  }
  return TR_OK;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char *tr_last_error(const tr_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int tr_create(const tr_robot_desc *rb, int device, tr_ctx **out) {
  if (!rb || !out) return fail(nullptr, TR_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  if (rb->n_tendons < 1 || rb->n_tendons > TR_MAX_TENDONS)
    return fail(nullptr, TR_ERR_OUT_OF_RANGE, "n_tendons must be in 1..8");
  if (rb->n_a < 1 || rb->n_a > TR_MAX_COEF || rb->n_m < 1 || rb->n_m > TR_MAX_COEF)
    return fail(nullptr, TR_ERR_OUT_OF_RANGE, "polynomial sizes must be in 1..8");
  if (!rb->C || !rb->D || !rb->max_tension || !rb->min_length || !rb->max_length)
    return fail(nullptr, TR_ERR_INVALID_ARG, "null tendon arrays");
  if (!(rb->L > 0) || !(rb->dL > 0) || !(rb->ro > rb->ri) || !(rb->E > 0))
    return fail(nullptr, TR_ERR_INVALID_ARG, "backbone specs must be positive (L, dL, E) with ro > ri");
  if (rb->L / rb->dL > 65536)
    return fail(nullptr, TR_ERR_INVALID_ARG, "more than 65536 backbone points");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, TR_ERR_HIP, "no HIP device available (libtendon_hip has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(nullptr, TR_ERR_INVALID_ARG, "bad device ordinal");
  tr_ctx *c = new tr_ctx();
  c->device = device;
  if (const char *e = std::getenv("TENDON_HIP_FUSED")) { const int v = std::atoi(e); c->fuse = v < 0 ? 0 : (v > 2 ? 2 : v); }
  if (const char *e = std::getenv("TENDON_HIP_RETRACT_SORT")) c->retract_sort_min = std::atoll(e);
  if (std::getenv("TENDON_HIP_RETRACT_KBEGIN_OFF")) c->retract_wave_start = false;
  if (const char *e = std::getenv("TENDON_HIP_EDGE_LANES")) { c->edge_lanes = std::max(1, std::min(tr_ctx::kMaxLanes, std::atoi(e))); c->edge_lanes_fixed = true; }
  if (const char *e = std::getenv("TENDON_HIP_EDGE_QUEUE")) { c->edge_queue = std::atoi(e) != 0; c->edge_queue_forced = c->edge_queue; }
  if (const char *e = std::getenv("TENDON_HIP_EDGE_QUEUE_WAVES")) { const int v = std::atoi(e); if (v >= 1 && v <= 8192) c->edge_queue_waves = v; }
  if (const char *e = std::getenv("TENDON_HIP_EDGE_LANE_GUESS")) { const double v = std::atof(e); if (v >= 0.5 && v <= 64.0) { c->edge_lane_guess = v; c->edge_lane_guess_forced = true; } }
  if (const char *e = std::getenv("TENDON_HIP_EDGE_POOL")) {      // testing only: a small pool forces the chunk-halving path
    const long long v = std::atoll(e);
    if (v >= 256 && v <= (1ll << 24)) c->edge_pool_max = c->edge_slots_max = (int64_t)round_up(v, 64);
  }
  if (const char *e = std::getenv("TENDON_HIP_CH_SCALE")) { const double v = std::atof(e); if (v > 0.05 && v < 50) c->ch_scale = v; }
  if (hipSetDevice(device) != hipSuccess) { delete c; return fail(nullptr, TR_ERR_HIP, "hipSetDevice failed"); }
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
      c->k1_round = (int64_t)cus * 4 * (rb->n_tendons <= TRK_K1_TWO_WAVE_MAXN ? 2 : 1) * 64;
    c->fb_cap = c->k1_round;
  }
  if (const char *e = std::getenv("TENDON_HIP_FB_CAP")) { const long long v = std::atoll(e); if (v >= 64 && v <= (1ll << 20)) c->fb_cap = (int64_t)round_up(v, 64); }

  RobotK &K = c->K;
  const int N = rb->n_tendons;
  K.n_tendons = N; K.n_a = rb->n_a; K.n_m = rb->n_m;
  K.enable_rotation = rb->enable_rotation ? 1 : 0;
  K.enable_retraction = rb->enable_retraction ? 1 : 0;
  K.state_size = N + K.enable_rotation + K.enable_retraction;
  K.residual_threshold = rb->residual_threshold;
  K.radius = rb->r; K.L = rb->L; K.dL = rb->dL;
  c->C.assign(rb->C, rb->C + (size_t)N * rb->n_a);
  c->D.assign(rb->D, rb->D + (size_t)N * rb->n_m);
  c->max_tension.assign(rb->max_tension, rb->max_tension + N);
  for (int j = 0; j < N; j++) { K.min_len[j] = rb->min_length[j]; K.max_len[j] = rb->max_length[j]; }
  {  // get_stiffness_matrices, tendon/TendonRobot.cpp:105-148
    const double ro2 = rb->ro * rb->ro, ri2 = rb->ri * rb->ri;
    const double I = (1.0 / 4.0) * M_PI * (ro2 * ro2 - ri2 * ri2);
    const double Ar = M_PI * (ro2 - ri2);
    const double J = 2 * I;
    const double Gmod = rb->E / (2 * (1 + rb->nu));
    K.kb0 = rb->E * I; K.kb2 = J * Gmod;
    K.ikb0 = 1 / (rb->E * I); K.ikb2 = 1 / (J * Gmod);
    K.ks0 = Gmod * Ar; K.ks2 = rb->E * Ar;
    K.iks0 = 1 / (Gmod * Ar); K.iks2 = 1 / (rb->E * Ar);
  }
  // shared arc-length grid, RK4 step list (integrate_times: steps of min(dL, t[j+1]-cur) while
  // t[j+1]-cur > eps; every interval restarts at exactly t[j]) and the routing table
  c->t = t_range(0.0, rb->L, rb->dL);
  K.n_points = (int)c->t.size();
  std::vector<double> step_t;
  for (size_t j = 0; j + 1 < c->t.size(); j++) {
    double cur = c->t[j];
    const double tn = c->t[j + 1];
    bool any = false;
    while (tn - cur > DBL_EPSILON) {
      const double h = std::min(rb->dL, tn - cur);
      c->steps.push_back(StepK{h, -1, 0});
      step_t.push_back(cur);
      cur += h;
      any = true;
    }
    if (!any) { c->steps.push_back(StepK{0.0, -1, 0}); step_t.push_back(cur); }
    c->steps.back().obs = (int)j + 1;
  }
  const size_t ent = (size_t)N * 6;
  std::vector<double> tab((1 + 3 * c->steps.size()) * ent);
  routing_at(c, 0.0, tab.data());
  for (size_t k = 0; k < c->steps.size(); k++) {
    const double tk = step_t[k], h = c->steps[k].h;
    routing_at(c, tk, &tab[(1 + 3 * k + 0) * ent]);
    routing_at(c, tk + h * 0.5, &tab[(1 + 3 * k + 1) * ent]);
    routing_at(c, tk + h, &tab[(1 + 3 * k + 2) * ent]);
  }
  home_lengths(c, c->t, 0.0, K.home_Li);
  // retraction kernel: the step after the grid's own first interval, and the home-length integrand
  // sqrt(rho'^2 + rho^2 theta'^2 + 1) (TendonRobot.cpp:300-307) at every shared abscissa
  for (size_t k = 0; k < c->steps.size(); k++) if (c->steps[k].obs == 1) c->k_first = (int)k + 1;
  c->rows_one_step = (int)c->steps.size() - c->k_first == K.n_points - 2;
  for (size_t k = (size_t)c->k_first; k < c->steps.size() && c->rows_one_step; k++)
    if (c->steps[k].obs != (int)k - c->k_first + 2) c->rows_one_step = false;
  std::vector<double> hl(c->t.size() * (size_t)N);
  for (size_t q = 0; q < c->t.size(); q++)
    for (int j = 0; j < N; j++) {
      const double *Cj = &c->C[(size_t)j * rb->n_a], *Dj = &c->D[(size_t)j * rb->n_m];
      double Cdot[TRK_MAX_COEF] = {0}, Ddot[TRK_MAX_COEF] = {0};
      for (int k = 1; k < rb->n_a; k++) Cdot[k - 1] = k * Cj[k];
      for (int k = 1; k < rb->n_m; k++) Ddot[k - 1] = k * Dj[k];
      const double dd = poly_at(Ddot, rb->n_m, c->t[q]), dv = poly_at(Dj, rb->n_m, c->t[q]), cd = poly_at(Cdot, rb->n_a, c->t[q]);
      hl[q * N + j] = std::sqrt(dd * dd + (dv * dv) * (cd * cd) + 1);
    }
  PolyK poly{};
  for (int j = 0; j < N; j++) {
    for (int i = 0; i < rb->n_a; i++) poly.C[j][i] = c->C[(size_t)j * rb->n_a + i];
    for (int i = 0; i < rb->n_m; i++) poly.D[j][i] = c->D[(size_t)j * rb->n_m + i];
    const int rdeg = poly_degree(&c->D[(size_t)j * rb->n_m], rb->n_m), tdeg = poly_degree(&c->C[(size_t)j * rb->n_a], rb->n_a);
    poly.home_kind[j] = (rdeg == 0 && tdeg == 0) ? 0 : ((rdeg == 0 && tdeg == 1) ? 1 : 2);
    const double d0 = poly.D[j][0], c1 = rb->n_a > 1 ? poly.C[j][1] : 0.0;
    poly.helix_scale[j] = std::sqrt(1 + d0 * d0 * c1 * c1);
  }

  auto bail = [&](const char *what) { std::string m = what; tr_destroy(c); return fail(nullptr, TR_ERR_HIP, m); };
  if (hipMalloc((void **)&c->d_poly, sizeof(PolyK)) != hipSuccess) return bail("hipMalloc(poly)");
  if (hipMemcpy(c->d_poly, &poly, sizeof(PolyK), hipMemcpyHostToDevice) != hipSuccess) return bail("hipMemcpy(poly)");
  if (hipMalloc((void **)&c->d_tgrid, c->t.size() * sizeof(double)) != hipSuccess ||
      hipMalloc((void **)&c->d_hl, hl.size() * sizeof(double)) != hipSuccess) return bail("hipMalloc(tgrid)");
  if (hipMemcpy(c->d_tgrid, c->t.data(), c->t.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(c->d_hl, hl.data(), hl.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return bail("hipMemcpy(tgrid)");
  if (hipMalloc((void **)&c->d_tab, tab.size() * sizeof(double)) != hipSuccess) return bail("hipMalloc(tab)");
  if (hipMemcpy(c->d_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return bail("hipMemcpy(tab)");
  if (hipMalloc((void **)&c->d_steps, std::max<size_t>(1, c->steps.size()) * sizeof(StepK)) != hipSuccess) return bail("hipMalloc(steps)");
  if (!c->steps.empty() &&
      hipMemcpy(c->d_steps, c->steps.data(), c->steps.size() * sizeof(StepK), hipMemcpyHostToDevice) != hipSuccess) return bail("hipMemcpy(steps)");
  *out = c;
  return TR_OK;
}

void tr_destroy(tr_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (auto &v : c->events) for (auto &e : v) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  if (c->pipe.ready) {
    for (int b = 0; b < 2; b++) {
      (void)hipHostFree(c->pipe.h_states[b]); (void)hipHostFree(c->pipe.h_tips[b]); (void)hipHostFree(c->pipe.h_bits[b]); (void)hipHostFree(c->pipe.h_flags[b]);
      (void)hipEventDestroy(c->pipe.up[b]); (void)hipEventDestroy(c->pipe.done[b]); (void)hipEventDestroy(c->pipe.down[b]);
    }
    (void)hipStreamDestroy(c->pipe.s_up); (void)hipStreamDestroy(c->pipe.s_comp); (void)hipStreamDestroy(c->pipe.s_down);
  }
  Workspace &w = c->ws;
  void *ptrs[] = {c->d_tab, c->d_steps, c->d_poly, c->d_tgrid, c->d_hl, c->d_grid, c->d_near, w.homeLi, w.np, c->d_vids, c->d_vmasks, c->d_vcounts,
                  c->d_vbits, w.px, w.py, w.pz, w.acc, w.Li, w.conv,
                  w.states, w.bits, w.tips, w.flags, w.L, w.npts,
                  c->edge.lvl_states, c->edge.bits, c->edge.sample_edge, c->edge.sample_t, c->edge.open, c->edge.frontier,
                  c->edge.A, c->edge.B, c->edge.rel, c->edge.edge_ok, c->edge.nfk, c->edge.first_inv, c->edge.last_t, c->edge.counters, c->edge.nd, c->edge.cnt, c->edge.sig, c->edge.sig_np, c->edge.ix_states, c->edge.ix_idx, c->edge.open2, c->edge.lvl_states2, c->edge.q_remaining, c->edge.q_lvl_base, c->edge.q_lvl_cnt, c->edge.q_ctl, c->edge.q_args, c->d_envw[0], c->d_envw[1], c->d_sph_near, c->d_sph_tmp};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  trk::merge_free(c->merge);
  if (c->fused.d_slots) { (void)hipFree(c->fused.d_slots); (void)hipHostFree(c->fused.h_slots); for (auto &e : c->fused.ev) (void)hipEventDestroy(e); }
  if (c->vring.d_slots) { (void)hipFree(c->vring.d_slots); (void)hipHostFree(c->vring.h_slots); for (auto &e : c->vring.ev) (void)hipEventDestroy(e); }
  if (c->last_dev_ev) (void)hipEventDestroy(c->last_dev_ev);
  {
    tr_ctx::Sampler &sm = c->samp;
    void *sp[] = {sm.states, sm.bits, sm.tips, sm.wprefix, sm.wprefix2, sm.d_ctr, sm.out_states, sm.out_tips, sm.out_index, sm.sig, sm.sig_index};
    for (void *q : sp) if (q) (void)hipFree(q);
    if (sm.h_ctr) (void)hipHostFree(sm.h_ctr);
  }
  if (c->d_item_src) (void)hipFree(c->d_item_src);
  if (c->d_item_edge) (void)hipFree(c->d_item_edge);
  for (void *q : c->knn.p) if (q) (void)hipFree(q);
  if (c->vstore.ids) (void)hipFree(c->vstore.ids);
  if (c->vstore.masks) (void)hipFree(c->vstore.masks);
  if (c->d_fb_list) (void)hipFree(c->d_fb_list);
  for (int q = 0; q < tr_ctx::kMaxLanes - 1; q++) {
    if (c->d_fb_list1[q]) (void)hipFree(c->d_fb_list1[q]);
    if (c->d_fb_count1[q]) (void)hipFree(c->d_fb_count1[q]);
    if (c->d_edge_counters1[q]) (void)hipFree(c->d_edge_counters1[q]);
  }
  for (int q = 0; q < tr_ctx::kMaxLanes; q++) { if (c->edge_stream[q]) (void)hipStreamDestroy(c->edge_stream[q]); if (c->edge_hc[q]) (void)hipHostFree(c->edge_hc[q]); }
  if (c->edge.q_hctl) (void)hipHostFree(c->edge.q_hctl);
  if (c->edge.q_hargs) (void)hipHostFree(c->edge.q_hargs);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  for (auto &ro : c->ro) {
    for (int q = 0; q < 2; q++) { if (ro.keys[q]) (void)hipFree(ro.keys[q]); if (ro.vals[q]) (void)hipFree(ro.vals[q]); }
    if (ro.kbegin) (void)hipFree(ro.kbegin);
    if (ro.handoff) (void)hipFree(ro.handoff);
    trk::merge_free(ro.ms);
  }
  if (c->d_fb_count) (void)hipFree(c->d_fb_count);
  delete c;
}

int tr_state_size(const tr_ctx *c) { return c ? c->K.state_size : -1; }
int tr_num_points(const tr_ctx *c) { return c ? c->K.n_points : -1; }
int tr_device(const tr_ctx *c) { return c ? c->device : -1; }

int tr_home_lengths(const tr_ctx *c, double *L_i) {
  if (!c || !L_i) return TR_ERR_INVALID_ARG;
  for (int j = 0; j < c->K.n_tendons; j++) L_i[j] = c->K.home_Li[j];
  return TR_OK;
}

int tr_set_checker(tr_ctx *c, int32_t checker) {
  if (!c) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  if (checker != TR_CHECKER_BACKBONE && checker != TR_CHECKER_SPHERES) return fail(c, TR_ERR_INVALID_ARG, "unknown checker");
  if (checker == TR_CHECKER_BACKBONE && c->has_grid && c->K.dL > std::max({c->G.dx, c->G.dy, c->G.dz}))
    return fail(c, TR_ERR_INVALID_ARG, "robot.specs.dL is larger than expected by VoxelBackboneValidityChecker");
  c->checker = checker;
  return TR_OK;
}

int tr_set_debug(tr_ctx *c, uint32_t bits) { if (!c) return TR_ERR_INVALID_ARG; c->debug = bits; return TR_OK; }
int tr_edge_schedule_last(const tr_ctx *c, uint32_t stats[4]) {
  if (!c || !stats) return TR_ERR_INVALID_ARG;
  for (int q = 0; q < 4; q++) stats[q] = c->edge_queue_last[q];
  return TR_OK;
}

int tr_set_grid(tr_ctx *c, uint32_t N, const double lim[6], const uint64_t *blocks, const double inv_rot[9]) {
  if (!c || !lim || !blocks) return fail(c, TR_ERR_INVALID_ARG, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  if (N < 4 || N > 512 || (N & (N - 1)))       // collision/VoxelOctree.cpp:98-116
    return fail(c, TR_ERR_INVALID_ARG, "unsupported voxel dimension: " + std::to_string(N));
  if (!(lim[0] < lim[1]) || !(lim[2] < lim[3]) || !(lim[4] < lim[5]))   // VoxelOctree.cpp:152-177
    return fail(c, TR_ERR_LENGTH, "limits must be positive in size");
  GridK g{};
  g.N = (int)N; g.Nb = (int)N / 4;
  g.xmin = lim[0]; g.xmax = lim[1]; g.ymin = lim[2]; g.ymax = lim[3]; g.zmin = lim[4]; g.zmax = lim[5];
  g.dx = (g.xmax - g.xmin) / N; g.dy = (g.ymax - g.ymin) / N; g.dz = (g.zmax - g.zmin) / N;
  g.inv_dx = 1 / g.dx; g.inv_dy = 1 / g.dy; g.inv_dz = 1 / g.dz;
  // VoxelBackboneValidityChecker.h:37-45
  const double max_dim = std::max({g.dx, g.dy, g.dz});
  if (c->checker == TR_CHECKER_BACKBONE && c->K.dL > max_dim) {    // the sphere checker has no such constructor check
    char buf[160];
    snprintf(buf, sizeof buf, "robot.specs.dL is larger than expected by VoxelBackboneValidityChecker (%g > %g)", c->K.dL, max_dim);
    return fail(c, TR_ERR_INVALID_ARG, buf);
  }
  static const double I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const double *r = inv_rot ? inv_rot : I9;
  g.rot_is_identity = 1;
  for (int i = 0; i < 9; i++) { g.inv_rot[i] = r[i]; if (r[i] != I9[i]) g.rot_is_identity = 0; }
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipDeviceSynchronize());          // launches that still read the previous grid (any stream) finish first
  const size_t nb = (size_t)g.Nb * g.Nb * g.Nb;
  if (c->n_blocks != nb) {
    if (c->d_grid) { (void)hipFree(c->d_grid); c->d_grid = nullptr; }
    if (c->d_near) { (void)hipFree(c->d_near); c->d_near = nullptr; }
    HIP_TRY(c, hipMalloc((void **)&c->d_grid, nb * sizeof(uint64_t)));
    HIP_TRY(c, hipMalloc((void **)&c->d_near, nb * sizeof(uint64_t)));
    c->n_blocks = (uint32_t)nb;
  }
  HIP_TRY(c, hipMemcpy(c->d_grid, blocks, nb * sizeof(uint64_t), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(trk::dilate2_blocks, dim3((unsigned)nb), dim3(64), 0, nullptr, c->d_grid, c->d_near, g.Nb);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipDeviceSynchronize());
  c->G = g;
  c->has_grid = true;
  c->sph_near_valid = false;
  return TR_OK;
}

// ---- environment preparation on the resident grid (env_kernel.hpp) ---------------------------
namespace {
int env_begin(tr_ctx *c) {
  if (!c->has_grid) return fail(c, TR_ERR_INVALID_ARG, "no obstacle grid set (tr_set_grid)");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipDeviceSynchronize());          // launches that still read the grid (any stream) finish before it is edited
  const uint32_t nbw = (uint32_t)((c->G.Nb + 2) * (c->G.Nb + 2) * (c->G.Nb + 2));
  if (c->envw_blocks < nbw) {
    HIP_TRY(c, hipDeviceSynchronize());
    for (int k = 0; k < 2; k++) { int rc = dev_alloc(c, &c->d_envw[k], (size_t)nbw); if (rc) return rc; }
    c->envw_blocks = nbw;
  }
  return TR_OK;
}
// the fast-skip grid of K2 follows every edit of the obstacle grid
int env_end(tr_ctx *c) {
  c->sph_near_valid = false;
  hipLaunchKernelGGL(trk::dilate2_blocks, dim3(c->n_blocks), dim3(64), 0, nullptr, c->d_grid, c->d_near, c->G.Nb);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipDeviceSynchronize());
  return TR_OK;
}
}  // namespace

int tr_get_grid(tr_ctx *c, uint64_t *blocks) {
  if (!c || !blocks) return fail(c, TR_ERR_INVALID_ARG, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  if (!c->has_grid) return fail(c, TR_ERR_INVALID_ARG, "no obstacle grid set (tr_set_grid)");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpy(blocks, c->d_grid, (size_t)c->n_blocks * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return TR_OK;
}

int tr_grid_add_spheres(tr_ctx *c, const double *spheres, int64_t n) {
  if (!c || n < 0 || (n > 0 && !spheres)) return fail(c, TR_ERR_INVALID_ARG, "bad argument");
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  int rc;
  if ((rc = env_begin(c))) return rc;
  if (n == 0) return TR_OK;
  const GridK &g = c->G;
  // nearest_block_idx (collision/VoxelOctree.cpp:272-283): int((x - min) / d), / 4 toward zero, clamped
  auto blk = [&](double v, double lo, double dd) {
    const int i = (int)((v - lo) / dd);
    return std::min(g.Nb - 1, std::max(0, i / 4));
  };
  std::vector<trk::SphereK> hs((size_t)n);
  for (int64_t k = 0; k < n; k++) {
    const double cx = spheres[4 * k], cy = spheres[4 * k + 1], cz = spheres[4 * k + 2], r = spheres[4 * k + 3];
    trk::SphereK &q = hs[(size_t)k];
    q.cx = cx; q.cy = cy; q.cz = cz; q.rr = r * r;
    q.lo[0] = blk(cx - r, g.xmin, g.dx); q.lo[1] = blk(cy - r, g.ymin, g.dy); q.lo[2] = blk(cz - r, g.zmin, g.dz);
    q.hi[0] = blk(cx + r, g.xmin, g.dx); q.hi[1] = blk(cy + r, g.ymin, g.dy); q.hi[2] = blk(cz + r, g.zmin, g.dz);
    // add_point (:319-323): closed domain test, then nearest_cell = int((x - min) / d) clamped (:295-307)
    const bool in = !(cx < g.xmin || g.xmax < cx || cy < g.ymin || g.ymax < cy || cz < g.zmin || g.zmax < cz);
    auto cell = [&](double v, double lo, double dd) { return std::min(g.N - 1, std::max(0, (int)((v - lo) / dd))); };
    q.pc[0] = in ? cell(cx, g.xmin, g.dx) : -1; q.pc[1] = in ? cell(cy, g.ymin, g.dy) : -1; q.pc[2] = in ? cell(cz, g.zmin, g.dz) : -1;
    q.pad_ = 0;
  }
  trk::SphereK *d_s = nullptr;
  HIP_TRY(c, hipMalloc((void **)&d_s, hs.size() * sizeof(trk::SphereK)));
  hipError_t e = hipMemcpy(d_s, hs.data(), hs.size() * sizeof(trk::SphereK), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    const int per = 4096;                                     // spheres per launch (a wave loops over them)
    for (int64_t k = 0; k < n && e == hipSuccess; k += per) {
      hipLaunchKernelGGL(trk::grid_add_spheres, dim3(c->n_blocks), dim3(64), 0, nullptr, c->d_grid, g, d_s + k, (int)std::min<int64_t>(per, n - k));
      e = hipGetLastError();
    }
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  (void)hipFree(d_s);
  HIP_TRY(c, e);
  return env_end(c);
}

int tr_grid_add_capsules(tr_ctx *c, const double *capsules, int64_t n) {
  if (!c || n < 0 || (n > 0 && !capsules)) return fail(c, TR_ERR_INVALID_ARG, "bad argument");
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  int rc;
  if ((rc = env_begin(c))) return rc;
  if (n == 0) return TR_OK;
  const GridK &g = c->G;
  auto blk = [&](double v, double lo, double dd) {              // nearest_block_idx (VoxelOctree.cpp:272-283)
    const int i = (int)((v - lo) / dd);
    return std::min(g.Nb - 1, std::max(0, i / 4));
  };
  auto cell = [&](double v, double lo, double dd) { return std::min(g.N - 1, std::max(0, (int)((v - lo) / dd))); };
  std::vector<trk::CapsuleK> hc((size_t)n);
  for (int64_t k = 0; k < n; k++) {
    const double *p = capsules + 7 * k, r = p[6];
    trk::CapsuleK &q = hc[(size_t)k];
    for (int d = 0; d < 3; d++) { q.a[d] = p[d]; q.b[d] = p[3 + d]; }
    q.rr = r * r;
    const double lo3[3] = {g.xmin, g.ymin, g.zmin}, dd3[3] = {g.dx, g.dy, g.dz};
    for (int d = 0; d < 3; d++) {
      q.lo[d] = blk(std::min(p[d], p[3 + d]) - r, lo3[d], dd3[d]);
      q.hi[d] = blk(std::max(p[d], p[3 + d]) + r, lo3[d], dd3[d]);
    }
    // add_point (:319-323): closed domain test, then nearest_cell (:295-307)
    auto inside = [&](const double *v) { return !(v[0] < g.xmin || g.xmax < v[0] || v[1] < g.ymin || g.ymax < v[1] || v[2] < g.zmin || g.zmax < v[2]); };
    const bool ia = inside(p), ib = inside(p + 3);
    for (int d = 0; d < 3; d++) { q.pa[d] = ia ? cell(p[d], lo3[d], dd3[d]) : -1; q.pb[d] = ib ? cell(p[3 + d], lo3[d], dd3[d]) : -1; }
  }
  trk::CapsuleK *d_c = nullptr;
  HIP_TRY(c, hipMalloc((void **)&d_c, hc.size() * sizeof(trk::CapsuleK)));
  hipError_t e = hipMemcpy(d_c, hc.data(), hc.size() * sizeof(trk::CapsuleK), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    const int per = 4096;
    for (int64_t k = 0; k < n && e == hipSuccess; k += per) {
      hipLaunchKernelGGL(trk::grid_add_capsules, dim3(c->n_blocks), dim3(64), 0, nullptr, c->d_grid, g, d_c + k, (int)std::min<int64_t>(per, n - k));
      e = hipGetLastError();
    }
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  (void)hipFree(d_c);
  HIP_TRY(c, e);
  return env_end(c);
}

int tr_grid_remove_interior(tr_ctx *c, int32_t keep_diagonal) {
  if (!c) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  int rc;
  if ((rc = env_begin(c))) return rc;
  hipLaunchKernelGGL(trk::grid_remove_interior, dim3(c->n_blocks), dim3(64), 0, nullptr, c->d_grid, c->d_envw[0], c->G.Nb, (int)(keep_diagonal != 0));
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpy(c->d_grid, c->d_envw[0], (size_t)c->n_blocks * sizeof(uint64_t), hipMemcpyDeviceToDevice));
  return env_end(c);
}

int tr_grid_dilate(tr_ctx *c, int32_t num, int32_t use_diagonal) {
  if (!c) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  int rc;
  if ((rc = env_begin(c))) return rc;
  const int Nb = c->G.Nb, NbW = Nb + 2;
  const unsigned nbw = (unsigned)(NbW * NbW * NbW);
  // four dilations at a time, clipped to the grid after each group (dilate_one_impl, VoxelOctree.cpp:693-762)
  for (; num > 0; num -= 4) {
    const int n = std::min(num, 4);
    hipLaunchKernelGGL(trk::grid_embed, dim3((nbw + 255) / 256), dim3(256), 0, nullptr, c->d_grid, c->d_envw[0], Nb);
    int cur = 0;
    for (int s = 0; s < n; s++, cur ^= 1)
      hipLaunchKernelGGL(trk::grid_dilate_step, dim3(nbw), dim3(64), 0, nullptr, c->d_envw[cur], c->d_envw[cur ^ 1], NbW, (int)(use_diagonal != 0));
    hipLaunchKernelGGL(trk::grid_extract, dim3((c->n_blocks + 255) / 256), dim3(256), 0, nullptr, c->d_envw[cur], c->d_grid, Nb);
    HIP_TRY(c, hipGetLastError());
  }
  return env_end(c);
}

int tr_grid_dilate_sphere(tr_ctx *c, double r) {
  if (!c) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  if (!c->has_grid) return fail(c, TR_ERR_INVALID_ARG, "no obstacle grid set (tr_set_grid)");
  // "over-approximation with multiple dilations" (VoxelOctree.cpp:950-952)
  return tr_grid_dilate(c, (int)std::round(r / std::min(c->G.dx, std::min(c->G.dy, c->G.dz))), 0);
}

int tr_reserve(tr_ctx *c, int64_t n) {
  if (!c || n < 0) return fail(c, TR_ERR_INVALID_ARG, "bad argument");
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;
  if (c->fuse == 2) {
    // the verdict path keeps no points: a list of fallback candidates and the fallback pass's small workspace
    if (c->fb_list_cap < n) {
      HIP_TRY(c, hipDeviceSynchronize());
      if ((rc = dev_alloc(c, &c->d_fb_list, (size_t)round_up(std::max<int64_t>(n, 64), 64)))) return rc;
      if (!c->d_fb_count && (rc = dev_alloc(c, &c->d_fb_count, 1))) return rc;
      c->fb_list_cap = round_up(std::max<int64_t>(n, 64), 64);
    }
    rc = ensure_workspace(c, std::min<int64_t>(c->fb_cap, std::max<int64_t>(n, 64)));
  } else {
    rc = ensure_workspace(c, n);
  }
  if (rc) return rc;
  return ensure_staging(c, n);
}

// ---- FK ------------------------------------------------------------------------------------
int tr_fk_batch_dev(tr_ctx *c, const double *d_states, int64_t n, int64_t ld, double *d_px, double *d_py,
                    double *d_pz, double *d_R, double *d_L, double *d_Li, uint8_t *d_converged,
                    int32_t *d_n_points, void *stream) {
  if (!c) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  if (n < 0 || ld < n || (ld & 63)) return fail(c, TR_ERR_INVALID_ARG, "ld must be a multiple of 64 and >= n");
  if (n == 0) return TR_OK;
  if (!d_states || !d_px || !d_py || !d_pz) return fail(c, TR_ERR_INVALID_ARG, "null device pointer");
  trk::FkOut out{d_px, d_py, d_pz, d_R, d_L, d_Li, nullptr, d_converged, d_n_points, nullptr};
  return launch_fk(c, d_states, n, ld, out, (hipStream_t)stream);
}

int tr_fk_batch_retraction_dev(tr_ctx *c, const double *d_states, int64_t n, int64_t ld, double *d_px, double *d_py,
                               double *d_pz, double *d_R, double *d_L, double *d_Li, uint8_t *d_converged,
                               int32_t *d_n_points, double *d_home_Li, void *stream) {
  if (!c) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  if (!c->K.enable_retraction) return fail(c, TR_ERR_INVALID_ARG, "robot has no retraction: use tr_fk_batch_dev (home lengths: tr_home_lengths)");
  if (n < 0 || ld < n || (ld & 63)) return fail(c, TR_ERR_INVALID_ARG, "ld must be a multiple of 64 and >= n");
  if (n == 0) return TR_OK;
  if (!d_states || !d_px || !d_py || !d_pz || !d_n_points || !d_home_Li) return fail(c, TR_ERR_INVALID_ARG, "null device pointer");
  trk::FkOut out{d_px, d_py, d_pz, d_R, d_L, d_Li, nullptr, d_converged, d_n_points, d_home_Li};
  return launch_fk(c, d_states, n, ld, out, (hipStream_t)stream);
}

int tr_fk_batch(tr_ctx *c, const double *states, int64_t n, double *p, double *R, double *L, double *L_i,
                uint8_t *converged, int32_t *n_points) {
  if (!c) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  if (n < 0 || (n > 0 && !states)) return fail(c, TR_ERR_INVALID_ARG, "bad argument");
  if (n == 0) return TR_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipDeviceSynchronize());          // the workspace is shared with *_dev calls that may still run on other streams
  const int S = c->K.state_size, P = c->K.n_points, N = c->K.n_tendons;
  const int64_t chunk_max = std::min<int64_t>(c->max_chunk, 1 << 16);
  int rc;
  if ((rc = ensure_workspace(c, std::min(n, chunk_max)))) return rc;
  if ((rc = ensure_staging(c, std::min(n, chunk_max)))) return rc;
  Workspace &w = c->ws;
  struct DevBuf {                               // freed on every return path
    double *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
  } dR;
  std::vector<double> hx, hy, hz, hR, hLi;
  std::vector<int32_t> hn;
  for (int64_t off = 0; off < n; off += chunk_max) {
    // the chunk uses its own leading dimension (the workspace may have been grown to 2^22 columns by an edge call:
    // an R buffer of that width would be 39 GB at P = 129)
    const int64_t m = std::min(chunk_max, n - off), ld = round_up(m, 64);
    HIP_TRY(c, hipMemcpy(w.states, states + off * S, (size_t)m * S * sizeof(double), hipMemcpyHostToDevice));
    if (R && !dR.p) HIP_TRY(c, hipMalloc((void **)&dR.p, (size_t)9 * P * round_up(std::min(chunk_max, n), 64) * sizeof(double)));
    trk::FkOut out{w.px, w.py, w.pz, dR.p, w.L, w.Li, nullptr, w.conv, w.npts, nullptr};
    if ((rc = launch_fk(c, w.states, m, ld, out, nullptr))) return rc;
    HIP_TRY(c, hipDeviceSynchronize());
    if (p || R) {                               // point counts of THIS chunk (retraction robots: rows are aligned at the tip)
      hn.resize((size_t)m);
      HIP_TRY(c, hipMemcpy(hn.data(), w.npts, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    // device planes are [P][ld]; only the first m columns are copied (2-D copies, host pitch m)
    if (p) {
      hx.resize((size_t)P * m); hy.resize((size_t)P * m); hz.resize((size_t)P * m);
      HIP_TRY(c, hipMemcpy2D(hx.data(), (size_t)m * 8, w.px, (size_t)ld * 8, (size_t)m * 8, (size_t)P, hipMemcpyDeviceToHost));
      HIP_TRY(c, hipMemcpy2D(hy.data(), (size_t)m * 8, w.py, (size_t)ld * 8, (size_t)m * 8, (size_t)P, hipMemcpyDeviceToHost));
      HIP_TRY(c, hipMemcpy2D(hz.data(), (size_t)m * 8, w.pz, (size_t)ld * 8, (size_t)m * 8, (size_t)P, hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < m; i++) {
        const int sh = P - hn[(size_t)i];               // device rows are aligned at the tip: point j is in row j + (P - n_points)
        for (int j = 0; j < P; j++) {
          double *o = p + ((size_t)(off + i) * P + j) * 3;
          if (j < hn[(size_t)i]) { o[0] = hx[(size_t)(j + sh) * m + i]; o[1] = hy[(size_t)(j + sh) * m + i]; o[2] = hz[(size_t)(j + sh) * m + i]; }
          else o[0] = o[1] = o[2] = std::numeric_limits<double>::quiet_NaN();
        }
      }
    }
    if (R) {
      hR.resize((size_t)9 * P * m);
      HIP_TRY(c, hipMemcpy2D(hR.data(), (size_t)m * 8, dR.p, (size_t)ld * 8, (size_t)m * 8, (size_t)9 * P, hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < m; i++) {
        const int np_i = hn[(size_t)i], sh = P - np_i;
        for (int j = 0; j < P; j++)
          for (int q = 0; q < 9; q++)
            R[((size_t)(off + i) * P + j) * 9 + q] = j < np_i ? hR[((size_t)q * P + j + sh) * m + i] : std::numeric_limits<double>::quiet_NaN();
      }
    }
    if (L) HIP_TRY(c, hipMemcpy(L + off, w.L, (size_t)m * sizeof(double), hipMemcpyDeviceToHost));
    if (L_i) {
      hLi.resize((size_t)N * m);
      HIP_TRY(c, hipMemcpy2D(hLi.data(), (size_t)m * 8, w.Li, (size_t)ld * 8, (size_t)m * 8, (size_t)N, hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < m; i++) for (int j = 0; j < N; j++) L_i[(size_t)(off + i) * N + j] = hLi[(size_t)j * m + i];
    }
    if (converged) HIP_TRY(c, hipMemcpy(converged + off, w.conv, (size_t)m, hipMemcpyDeviceToHost));
    if (n_points) HIP_TRY(c, hipMemcpy(n_points + off, w.npts, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost));
  }
  return TR_OK;
}

// ---- state validity ------------------------------------------------------------------------
int tr_validate_shapes_dev(tr_ctx *c, int64_t n, int64_t ld, const double *d_px, const double *d_py,
                           const double *d_pz, const int32_t *d_n_points, const double *d_Li,
                           const uint8_t *d_converged, int check_voxels, uint64_t *d_valid_bits,
                           uint8_t *d_flags, void *stream) {
  if (!c) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  if (n < 0 || ld < n || (ld & 63)) return fail(c, TR_ERR_INVALID_ARG, "ld must be a multiple of 64 and >= n");
  if (n == 0) return TR_OK;
  if (!d_px || !d_py || !d_pz || !d_Li || !d_converged || !d_valid_bits) return fail(c, TR_ERR_INVALID_ARG, "null device pointer");
  if (c->K.enable_retraction)
    return fail(c, TR_ERR_UNSUPPORTED, "with retraction the home lengths are per configuration: use tr_validate_shapes_retraction_dev");
  int rc;
  // the accumulated-chord scratch has the caller's leading dimension
  if (c->ws.ld < ld) {
    if (ld > c->max_chunk) return fail(c, TR_ERR_INVALID_ARG, "ld exceeds the workspace chunk size; call in smaller batches");
    if ((rc = ensure_workspace(c, ld))) return rc;
  }
  (void)d_n_points;                     // every configuration of a robot without retraction has all P points
  if ((rc = begin_dev_work(c, (hipStream_t)stream))) return rc;
  trk::SweepIn in{d_px, d_py, d_pz, nullptr, d_Li, d_converged, nullptr, c->ws.acc};
  // acc scratch is laid out [P][ws.ld]; the kernel indexes it with ld, which is <= ws.ld: fine as
  // long as P*ld <= P*ws.ld (it only needs P*ld doubles).
  if ((rc = launch_sweep(c, in, n, ld, check_voxels, d_valid_bits, d_flags, (hipStream_t)stream))) return rc;
  return note_dev_work(c, (hipStream_t)stream);
}

int tr_validate_shapes_retraction_dev(tr_ctx *c, int64_t n, int64_t ld, const double *d_px, const double *d_py,
                                      const double *d_pz, const int32_t *d_n_points, const double *d_Li,
                                      const double *d_home_Li, const uint8_t *d_converged, int check_voxels,
                                      uint64_t *d_valid_bits, uint8_t *d_flags, void *stream) {
  if (!c) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  if (!c->K.enable_retraction) return fail(c, TR_ERR_INVALID_ARG, "robot has no retraction: use tr_validate_shapes_dev");
  if (n < 0 || ld < n || (ld & 63)) return fail(c, TR_ERR_INVALID_ARG, "ld must be a multiple of 64 and >= n");
  if (n == 0) return TR_OK;
  if (!d_px || !d_py || !d_pz || !d_n_points || !d_Li || !d_home_Li || !d_converged || !d_valid_bits)
    return fail(c, TR_ERR_INVALID_ARG, "null device pointer");
  int rc;
  if (c->ws.ld < ld) {
    if (ld > c->max_chunk) return fail(c, TR_ERR_INVALID_ARG, "ld exceeds the workspace chunk size; call in smaller batches");
    if ((rc = ensure_workspace(c, ld))) return rc;
  }
  trk::SweepIn in{d_px, d_py, d_pz, d_n_points, d_Li, d_converged, d_home_Li, c->ws.acc};
  if ((rc = begin_dev_work(c, (hipStream_t)stream))) return rc;
  if ((rc = launch_sweep(c, in, n, ld, check_voxels, d_valid_bits, d_flags, (hipStream_t)stream))) return rc;
  return note_dev_work(c, (hipStream_t)stream);
}

}  // extern "C"
namespace {
int validate_batch_dev_impl(tr_ctx *c, const double *d_states, int64_t n, uint64_t *d_valid_bits,
                            double *d_tips, uint8_t *d_flags, void *stream, uint32_t *d_sig = nullptr);
}
extern "C" {
int tr_validate_batch_dev(tr_ctx *c, const double *d_states, int64_t n, uint64_t *d_valid_bits,
                          double *d_tips, uint8_t *d_flags, void *stream) {
  if (!c) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  int rc;
  if ((rc = begin_dev_work(c, (hipStream_t)stream))) return rc;
  rc = validate_batch_dev_impl(c, d_states, n, d_valid_bits, d_tips, d_flags, stream);
  if (rc == TR_OK && n > 0) return note_dev_work(c, (hipStream_t)stream);
  return rc;
}
}  // extern "C"
namespace {
// d_sig: [n][tr_signature_words] cell signatures of the backbones (the rows tr_validate_edges_indexed_sig_dev takes for the vertices)
int validate_batch_dev_impl(tr_ctx *c, const double *d_states, int64_t n, uint64_t *d_valid_bits,
                            double *d_tips, uint8_t *d_flags, void *stream, uint32_t *d_sig) {
  if (n < 0) return fail(c, TR_ERR_INVALID_ARG, "negative batch size");
  if (n == 0) return TR_OK;
  if (!d_states || !d_valid_bits) return fail(c, TR_ERR_INVALID_ARG, "null device pointer");
  if (!c->has_grid) return fail(c, TR_ERR_INVALID_ARG, "no obstacle grid set (tr_set_grid)");
  int rc;
  hipStream_t s = (hipStream_t)stream;
  const int S = c->K.state_size;
  const bool ret = c->K.enable_retraction;
  const int64_t sig_words = round_up(c->K.n_points, 16);
  if (d_sig && (c->fuse != 2 || ret || c->checker == TR_CHECKER_SPHERES)) return fail(c, TR_ERR_UNSUPPORTED, "signatures: backbone checker, no retraction, verdict-only schedule");
  if (c->fuse == 2) {
    // verdict-only kernel (both checkers, with or without retraction): nothing but the verdict (and the tips) leaves the chip, no point workspace to size
    const int64_t chunk = (int64_t)1 << 26;                 // list indices are 32-bit; per-launch grid stays far below 2^31 blocks
    for (int64_t off = 0; off < n; off += chunk) {
      const int64_t m = std::min<int64_t>(chunk, n - off);
      if ((rc = launch_verdict(c, d_states + off * S, m, d_valid_bits + off / 64, d_tips ? d_tips + 3 * off : nullptr,
                               d_flags ? d_flags + off : nullptr, s, d_sig ? d_sig + off * sig_words : nullptr, d_sig ? sig_words : 0,
                               c->checker == TR_CHECKER_SPHERES))) return rc;
    }
    return TR_OK;
  }
  if ((rc = ensure_workspace(c, n))) return rc;
  Workspace &w = c->ws;
  for (int64_t off = 0; off < n; off += w.ld) {
    const int64_t m = std::min<int64_t>(w.ld, n - off);
    trk::FkOut out{w.px, w.py, w.pz, nullptr, nullptr, w.Li, d_tips ? d_tips + 3 * off : nullptr, w.conv,
                   ret ? w.np : nullptr, ret ? w.homeLi : nullptr};
    trk::SweepIn in{w.px, w.py, w.pz, ret ? w.np : nullptr, w.Li, w.conv, ret ? w.homeLi : nullptr, w.acc};
    if ((rc = launch_fk_sweep(c, d_states + off * S, m, w.ld, out, in, c->checker == TR_CHECKER_SPHERES ? 2 : 1, d_valid_bits + off / 64,
                              d_flags ? d_flags + off : nullptr, s))) return rc;
  }
  return TR_OK;
}
}  // namespace
extern "C" {

namespace {
// Host-buffer batches are pipelined in chunks: the caller's (pageable) arrays are staged through two
// pinned buffers, uploads / kernels / downloads run on three streams chained by events, so PCIe
// transfers and the host-side staging copies of chunk i+1 overlap the kernels of chunk i.  Kernels of
// successive chunks share the FK workspace and stay ordered on the one compute stream.
int ensure_pipe(tr_ctx *c) {
  tr_ctx::Pipe &p = c->pipe;
  if (p.ready) return TR_OK;
  // chunk = one resident round of K1 waves (2^17 configurations on MI355X): measured 8.1e7 checks/s end to end
  // against 6.7e7 with 2^18 and 7.6e7 with 2^19 (a chunk's last round runs with a tail; shorter chunks start the
  // pipeline earlier)
  int64_t CH = std::max<int64_t>(1 << 14, c->k1_round);
  if (const char *e = std::getenv("TENDON_HIP_PIPE_LOG2")) { const int v = std::atoi(e); if (v >= 12 && v <= 22) CH = (int64_t)1 << v; }   // tuning only
  const int S = c->K.state_size;
  HIP_TRY(c, hipStreamCreateWithFlags(&p.s_up, hipStreamNonBlocking));
  HIP_TRY(c, hipStreamCreateWithFlags(&p.s_comp, hipStreamNonBlocking));
  HIP_TRY(c, hipStreamCreateWithFlags(&p.s_down, hipStreamNonBlocking));
  for (int b = 0; b < 2; b++) {
    HIP_TRY(c, hipHostMalloc((void **)&p.h_states[b], (size_t)CH * S * sizeof(double), hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc((void **)&p.h_tips[b], (size_t)CH * 3 * sizeof(double), hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc((void **)&p.h_bits[b], (size_t)(CH / 64) * sizeof(uint64_t), hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc((void **)&p.h_flags[b], (size_t)CH, hipHostMallocDefault));
    HIP_TRY(c, hipEventCreateWithFlags(&p.up[b], hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&p.done[b], hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&p.down[b], hipEventDisableTiming));
  }
  p.chunk = CH;
  p.ready = true;
  return TR_OK;
}
}  // namespace

int tr_validate_batch(tr_ctx *c, const double *states, int64_t n, uint64_t *valid_bits, double *tips, uint8_t *flags) {
  if (!c) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  if (n < 0 || (n > 0 && (!states || !valid_bits))) return fail(c, TR_ERR_INVALID_ARG, "bad argument");
  if (n == 0) return TR_OK;
  if (!c->has_grid) return fail(c, TR_ERR_INVALID_ARG, "no obstacle grid set (tr_set_grid)");
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure_pipe(c))) return rc;
  tr_ctx::Pipe &p = c->pipe;
  const int64_t CH = p.chunk;
  if (n <= kSmallBatch && n <= CH) {
    // A single state (isValid from a serial planner) or a handful: one stream, one pinned buffer, no device-wide
    // synchronisation -- the stream only waits for the last *_dev call that may still use this context's workspace.
    // What remains is the kernel itself: one lane integrates its configuration serially (INTEGRATION.md has the numbers).
    const int S = c->K.state_size;
    const bool verdict = c->fuse == 2;
    if ((rc = ensure_staging(c, n))) return rc;
    if (!verdict && (rc = ensure_workspace(c, n))) return rc;
    Workspace &w = c->ws;
    if ((rc = begin_dev_work(c, p.s_comp))) return rc;
    std::memcpy(p.h_states[0], states, (size_t)n * S * sizeof(double));
    HIP_TRY(c, hipMemcpyAsync(w.states, p.h_states[0], (size_t)n * S * sizeof(double), hipMemcpyHostToDevice, p.s_comp));
    if ((rc = validate_batch_dev_impl(c, w.states, n, w.bits, tips ? w.tips : nullptr, flags ? w.flags : nullptr, p.s_comp))) return rc;
    HIP_TRY(c, hipMemcpyAsync(p.h_bits[0], w.bits, (size_t)((n + 63) / 64) * sizeof(uint64_t), hipMemcpyDeviceToHost, p.s_comp));
    if (tips) HIP_TRY(c, hipMemcpyAsync(p.h_tips[0], w.tips, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, p.s_comp));
    if (flags) HIP_TRY(c, hipMemcpyAsync(p.h_flags[0], w.flags, (size_t)n, hipMemcpyDeviceToHost, p.s_comp));
    HIP_TRY(c, hipStreamSynchronize(p.s_comp));
    c->last_dev_used = false;                      // everything enqueued on this context so far has finished
    std::memcpy(valid_bits, p.h_bits[0], (size_t)((n + 63) / 64) * sizeof(uint64_t));
    if (tips) std::memcpy(tips, p.h_tips[0], (size_t)n * 3 * sizeof(double));
    if (flags) std::memcpy(flags, p.h_flags[0], (size_t)n);
    return TR_OK;
  }
  if ((rc = ensure_staging(c, std::min(n, 2 * CH)))) return rc;
  if (c->fuse != 2 && (rc = ensure_workspace(c, std::min(n, CH)))) return rc;
  HIP_TRY(c, hipDeviceSynchronize());          // earlier work on other streams (e.g. a *_dev call) is finished
  Workspace &w = c->ws;
  const int S = c->K.state_size;
  const int64_t nchunks = (n + CH - 1) / CH;
  auto drain = [&](int64_t i) -> int {         // copy chunk i's results out of its pinned buffers
    const int b = (int)(i & 1);
    const int64_t off = i * CH, m = std::min(CH, n - off);
    HIP_TRY(c, hipEventSynchronize(p.down[b]));
    std::memcpy(valid_bits + off / 64, p.h_bits[b], (size_t)((m + 63) / 64) * sizeof(uint64_t));
    if (tips) std::memcpy(tips + 3 * off, p.h_tips[b], (size_t)m * 3 * sizeof(double));
    if (flags) std::memcpy(flags + off, p.h_flags[b], (size_t)m);
    return TR_OK;
  };
  for (int64_t i = 0; i < nchunks; i++) {
    const int b = (int)(i & 1);
    const int64_t off = i * CH, m = std::min(CH, n - off);
    if (i >= 2 && (rc = drain(i - 2))) return rc;             // pinned buffers b are free again
    // device staging slots alternate with the pinned buffers
    double *d_st = w.states + (size_t)b * CH * S;
    uint64_t *d_bits = w.bits + (size_t)b * (CH / 64);
    double *d_tips = w.tips + (size_t)b * CH * 3;
    uint8_t *d_flags = w.flags + (size_t)b * CH;
    std::memcpy(p.h_states[b], states + off * S, (size_t)m * S * sizeof(double));
    HIP_TRY(c, hipMemcpyAsync(d_st, p.h_states[b], (size_t)m * S * sizeof(double), hipMemcpyHostToDevice, p.s_up));
    HIP_TRY(c, hipEventRecord(p.up[b], p.s_up));
    HIP_TRY(c, hipStreamWaitEvent(p.s_comp, p.up[b], 0));
    if (i >= 2) HIP_TRY(c, hipStreamWaitEvent(p.s_comp, p.down[b], 0));   // slot b's previous results have left the device
    if ((rc = validate_batch_dev_impl(c, d_st, m, d_bits, tips ? d_tips : nullptr, flags ? d_flags : nullptr, p.s_comp))) return rc;
    HIP_TRY(c, hipEventRecord(p.done[b], p.s_comp));
    HIP_TRY(c, hipStreamWaitEvent(p.s_down, p.done[b], 0));
    HIP_TRY(c, hipMemcpyAsync(p.h_bits[b], d_bits, (size_t)((m + 63) / 64) * sizeof(uint64_t), hipMemcpyDeviceToHost, p.s_down));
    if (tips) HIP_TRY(c, hipMemcpyAsync(p.h_tips[b], d_tips, (size_t)m * 3 * sizeof(double), hipMemcpyDeviceToHost, p.s_down));
    if (flags) HIP_TRY(c, hipMemcpyAsync(p.h_flags[b], d_flags, (size_t)m, hipMemcpyDeviceToHost, p.s_down));
    HIP_TRY(c, hipEventRecord(p.down[b], p.s_down));
  }
  for (int64_t i = std::max<int64_t>(0, nchunks - 2); i < nchunks; i++)
    if ((rc = drain(i))) return rc;
  return TR_OK;
}

// ---- cached voxel sets ---------------------------------------------------------------------
int tr_check_cached_dev(tr_ctx *c, const uint32_t *d_ids, const uint64_t *d_masks, const int64_t *d_offsets,
                        int64_t n_items, uint64_t *d_hit_bits, void *stream) {
  if (!c) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  if (n_items < 0) return fail(c, TR_ERR_INVALID_ARG, "negative item count");
  if (n_items == 0) return TR_OK;
  if (!c->has_grid) return fail(c, TR_ERR_INVALID_ARG, "no obstacle grid set (tr_set_grid)");
  if (!d_ids || !d_masks || !d_offsets || !d_hit_bits) return fail(c, TR_ERR_INVALID_ARG, "null device pointer");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(c, 2, s);
  const unsigned grid = (unsigned)((n_items + 3) / 4);
  HIP_TRY(c, hipMemsetAsync(d_hit_bits, 0, (size_t)((n_items + 63) / 64) * sizeof(uint64_t), s));
  hipLaunchKernelGGL(trk::cached_blocks_vs_grid, dim3(grid), dim3(64), 0, s, d_ids, d_masks, d_offsets, n_items,
                     c->d_grid, c->n_blocks, d_hit_bits);
  HIP_TRY(c, hipGetLastError());
  return TR_OK;
}

int tr_check_cached_subset_dev(tr_ctx *c, const uint32_t *d_ids, const uint64_t *d_masks, const int64_t *d_offsets,
                               int64_t n_items, const int32_t *d_list, int64_t n_list, uint8_t *d_hit, void *stream) {
  if (!c) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  if (n_items < 0 || n_list < 0) return fail(c, TR_ERR_INVALID_ARG, "negative count");
  if (n_list == 0) return TR_OK;
  if (!c->has_grid) return fail(c, TR_ERR_INVALID_ARG, "no obstacle grid set (tr_set_grid)");
  if (!d_ids || !d_masks || !d_offsets || !d_list || !d_hit) return fail(c, TR_ERR_INVALID_ARG, "null device pointer");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(c, 2, s);
  hipLaunchKernelGGL(trk::cached_subset_vs_grid, dim3((unsigned)((n_list + 15) / 16)), dim3(256), 0, s, d_ids, d_masks, d_offsets,
                     d_list, n_list, n_items, c->d_grid, c->n_blocks, d_hit);
  HIP_TRY(c, hipGetLastError());
  return TR_OK;
}

int tr_state_layout(const tr_ctx *c, int32_t *n_tendons, int32_t *has_rotation, int32_t *has_retraction) {
  if (!c) return TR_ERR_INVALID_ARG;
  if (n_tendons) *n_tendons = c->K.n_tendons;
  if (has_rotation) *has_rotation = c->K.enable_rotation;
  if (has_retraction) *has_retraction = c->K.enable_retraction;
  return TR_OK;
}

int tr_space_weights(const tr_ctx *c, double *w_rotation, double *w_retraction) {
  if (!c) return TR_ERR_INVALID_ARG;
  double ext2 = 0;
  for (int i = 0; i < c->K.n_tendons; i++) ext2 += c->max_tension[i] * c->max_tension[i];
  const double ext = std::sqrt(ext2);                         // RealVectorStateSpace::getMaximumExtent, bounds [0, max_tension]
  if (w_rotation) *w_rotation = ext / (4.0 * M_PI);           // Problem.cpp:131-137
  if (w_retraction) *w_retraction = 2.0 * ext / c->K.L;       // Problem.cpp:144-152
  return TR_OK;
}

int tr_check_cached(tr_ctx *c, const uint32_t *ids, const uint64_t *masks, const int64_t *offsets,
                    int64_t n_items, uint64_t *hit_bits) {
  if (!c) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  if (n_items < 0 || (n_items > 0 && (!offsets || !hit_bits))) return fail(c, TR_ERR_INVALID_ARG, "bad argument");
  if (n_items == 0) return TR_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  const int64_t nnz = offsets[n_items];
  if (nnz < 0 || (nnz > 0 && (!ids || !masks))) return fail(c, TR_ERR_INVALID_ARG, "bad CSR arrays");
  for (int64_t i = 0; i < n_items; i++)
    if (offsets[i] > offsets[i + 1]) return fail(c, TR_ERR_INVALID_ARG, "offsets must be non-decreasing");
  for (int64_t k = 0; k < nnz; k++)
    if (ids[k] >= c->n_blocks) return fail(c, TR_ERR_INVALID_ARG, "voxel dimension mismatch (block id out of range)");
  uint32_t *d_ids = nullptr; uint64_t *d_masks = nullptr, *d_bits = nullptr; int64_t *d_off = nullptr;
  const size_t words = (size_t)((n_items + 63) / 64);
  int rc = TR_OK;
  do {
    if (hipMalloc((void **)&d_ids, std::max<size_t>(1, nnz) * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void **)&d_masks, std::max<size_t>(1, nnz) * sizeof(uint64_t)) != hipSuccess ||
        hipMalloc((void **)&d_off, (size_t)(n_items + 1) * sizeof(int64_t)) != hipSuccess ||
        hipMalloc((void **)&d_bits, words * sizeof(uint64_t)) != hipSuccess) { rc = fail(c, TR_ERR_HIP, "hipMalloc failed"); break; }
    if ((nnz && (hipMemcpy(d_ids, ids, nnz * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess ||
                 hipMemcpy(d_masks, masks, nnz * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess)) ||
        hipMemcpy(d_off, offsets, (size_t)(n_items + 1) * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess) { rc = fail(c, TR_ERR_HIP, "hipMemcpy failed"); break; }
    if ((rc = tr_check_cached_dev(c, d_ids, d_masks, d_off, n_items, d_bits, nullptr))) break;
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(hit_bits, d_bits, words * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(c, TR_ERR_HIP, "copy back failed"); break; }
  } while (0);
  if (d_ids) (void)hipFree(d_ids);
  if (d_masks) (void)hipFree(d_masks);
  if (d_off) (void)hipFree(d_off);
  if (d_bits) (void)hipFree(d_bits);
  return rc;
}

// ---- robot voxel sets -----------------------------------------------------------------------
namespace {

// entries of a sample's block list: a backbone of P points enters at most a few hundred blocks even counting re-entries
int vox_max_blocks(const tr_ctx *c) { return 2 * c->K.n_points + 64; }

int ensure_vox_scratch(tr_ctx *c, int64_t cap) {
  if (c->vox_cap >= cap) return TR_OK;
  HIP_TRY(c, hipDeviceSynchronize());
  const size_t mb = (size_t)vox_max_blocks(c);
  int rc;
  if ((rc = dev_alloc(c, &c->d_vids, mb * cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_vmasks, mb * cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_vcounts, (size_t)cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_vbits, (size_t)cap / 64 + 1))) return rc;
  c->vox_cap = cap;
  return TR_OK;
}

// Voxelise samples [0, m) of the workspace (points already there, leading dimension ld) whose bit
// in d_bits is set; appends the per-sample lists to the context's block-list store (device) and writes counts (host).
int voxelize_samples(tr_ctx *c, int64_t m, int64_t ld, const int32_t *d_np, const uint64_t *d_bits,
                     std::vector<int32_t> &counts, std::vector<int64_t> &offs) {
  Workspace &w = c->ws;
  int rc;
  if ((rc = ensure_vox_scratch(c, ld))) return rc;
  const int mb = vox_max_blocks(c);
  {
    ProfScope ps(c, 3, nullptr);
    hipLaunchKernelGGL(trk::backbone_voxelize, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, nullptr, w.px, w.py, w.pz, d_np,
                       d_bits, m, ld, (int)c->K.n_points, c->G, mb, c->d_vids, c->d_vmasks, c->d_vcounts);
    HIP_TRY(c, hipGetLastError());
  }
  // duplicate-free lists ordered by block id: the sort + reduce-by-key of the edge caches with one "edge" per sample
  if (c->vox_items_cap < m) {
    HIP_TRY(c, hipDeviceSynchronize());
    if ((rc = dev_alloc(c, &c->d_item_src, (size_t)m + m / 4))) return rc;
    if ((rc = dev_alloc(c, &c->d_item_edge, (size_t)m + m / 4))) return rc;
    c->vox_items_cap = m + m / 4;
  }
  int64_t nu = 0;
  int ovf = 0;
  {
    ProfScope ps(c, 3, nullptr);
    hipLaunchKernelGGL(trk::iota_i32, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, nullptr, c->d_item_edge, m);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, trk::merge_edge_caches(c->merge, c->d_vids, c->d_vmasks, c->d_vcounts, nullptr, c->d_item_edge, m, ld, c->n_blocks, m, &nu, &ovf, nullptr));
  }
  if (ovf) return fail(c, TR_ERR_RUNTIME, "voxel set of a configuration exceeds the block-list capacity or leaves the domain");
  counts.resize((size_t)m);
  HIP_TRY(c, hipMemcpy(counts.data(), c->merge.ecount, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost));
  offs.assign((size_t)m + 1, 0);
  for (int64_t i = 0; i < m; i++) offs[(size_t)i + 1] = offs[(size_t)i] + counts[(size_t)i];
  if (offs[(size_t)m] != nu) return fail(c, TR_ERR_RUNTIME, "voxel cache merge: counts do not add up");
  return vstore_append(c, c->merge.uids, c->merge.uvals, nu);
}

}  // namespace

int tr_voxelize_batch(tr_ctx *c, const double *states, int64_t n, int64_t *offsets, uint64_t *shape_valid_bits, double *tips) {
  if (!c) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  if (n < 0 || (n > 0 && (!states || !offsets || !shape_valid_bits))) return fail(c, TR_ERR_INVALID_ARG, "bad argument");
  c->vstore.n = 0;
  if (offsets) offsets[0] = 0;
  if (n == 0) return TR_OK;
  if (!c->has_grid) return fail(c, TR_ERR_INVALID_ARG, "no obstacle grid set (tr_set_grid)");
  // TENDON_HIP_VOX_TIMING=1: where this call's time goes (stderr; tuning only -- every lap synchronises the device)
  const bool vt = std::getenv("TENDON_HIP_VOX_TIMING") != nullptr;
  auto vt0 = std::chrono::steady_clock::now();
  auto vlap = [&](const char *what) {
    if (!vt) return;
    (void)hipDeviceSynchronize();
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[voxelize_batch] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - vt0).count());
    vt0 = t1;
  };
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipDeviceSynchronize());
  vlap("entry (device drained)");
  const int64_t chunk = std::min<int64_t>(c->max_chunk, 1 << 18);   // block-list scratch: 12 B x (2P + 16) per configuration
  int rc;
  if ((rc = ensure_workspace(c, std::min(n, chunk)))) return rc;
  vlap("ensure_workspace");
  if ((rc = ensure_staging(c, std::min(n, chunk)))) return rc;
  vlap("ensure_staging");
  Workspace &w = c->ws;
  const int S = c->K.state_size;
  const bool ret = c->K.enable_retraction;
  std::vector<int32_t> counts; std::vector<int64_t> offs;
  for (int64_t off = 0; off < n; off += chunk) {
    const int64_t m = std::min(chunk, n - off);
    if ((rc = upload_staged(c, w.states, states + off * S, (size_t)m * S * sizeof(double), nullptr))) return rc;
    trk::FkOut out{w.px, w.py, w.pz, nullptr, nullptr, w.Li, tips ? w.tips : nullptr, w.conv, ret ? w.np : nullptr,
                   ret ? w.homeLi : nullptr};
    trk::SweepIn in{w.px, w.py, w.pz, ret ? w.np : nullptr, w.Li, w.conv, ret ? w.homeLi : nullptr, w.acc};
    vlap("upload");
    if ((rc = launch_fk_sweep(c, w.states, m, w.ld, out, in, 0, w.bits, nullptr, nullptr))) return rc;     // is_valid_shape only
    vlap("FK + shape test");
    const int64_t base = c->vstore.n;
    if ((rc = voxelize_samples(c, m, w.ld, ret ? w.np : nullptr, w.bits, counts, offs))) return rc;
    vlap("voxelise + merge + store");
    HIP_TRY(c, hipMemcpy(shape_valid_bits + off / 64, w.bits, (size_t)((m + 63) / 64) * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (tips) HIP_TRY(c, hipMemcpy(tips + 3 * off, w.tips, (size_t)m * 3 * sizeof(double), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < m; i++) offsets[off + i + 1] = base + offs[(size_t)i + 1];
    vlap("downloads + offsets");
  }
  return TR_OK;
}

int tr_voxelize_fetch(tr_ctx *c, uint32_t *block_ids, uint64_t *masks, int64_t capacity) {
  if (!c) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  const int64_t nnz = c->vstore.n;
  if (capacity < nnz) return fail(c, TR_ERR_INVALID_ARG, "capacity smaller than the stored block lists");
  if (nnz > 0) {
    if (!block_ids || !masks) return fail(c, TR_ERR_INVALID_ARG, "null output");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpy(block_ids, c->vstore.ids, (size_t)nnz * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(masks, c->vstore.masks, (size_t)nnz * sizeof(uint64_t), hipMemcpyDeviceToHost));
  }
  return TR_OK;
}

int tr_voxelize_fetch_dev(tr_ctx *c, uint32_t *d_block_ids, uint64_t *d_masks, int64_t capacity, void *stream) {
  if (!c) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  const int64_t nnz = c->vstore.n;
  if (capacity < nnz) return fail(c, TR_ERR_INVALID_ARG, "capacity smaller than the stored block lists");
  if (nnz > 0) {
    if (!d_block_ids || !d_masks) return fail(c, TR_ERR_INVALID_ARG, "null output");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(nullptr));            // the lists were written on the null stream
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(c, hipMemcpyAsync(d_block_ids, c->vstore.ids, (size_t)nnz * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(d_masks, c->vstore.masks, (size_t)nnz * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, hipStreamSynchronize(s));                  // the store may be overwritten by the next tr_voxelize_* call
  }
  return TR_OK;
}

int64_t tr_voxelize_count(const tr_ctx *c) { return c ? c->vstore.n : -1; }

// ---- nearest neighbours ---------------------------------------------------------------------
}  // extern "C"
namespace {
int knn_impl(tr_ctx *c, const double *states, int64_t n, int32_t k, double max_distance, int32_t *idx, double *dist,
             int32_t *edges, int64_t edge_capacity, int64_t *n_edges, int64_t q0 = 0, int64_t nq = -1, bool dev = false);
}
extern "C" {
int tr_knn(tr_ctx *c, const double *states, int64_t n, int32_t k, double max_distance, int32_t *idx, double *dist) {
  if (!c) return TR_ERR_INVALID_ARG;
  if (n > 0 && (!idx || !dist)) return fail(c, TR_ERR_INVALID_ARG, "bad argument");
  return knn_impl(c, states, n, k, max_distance, idx, dist, nullptr, 0, nullptr);
}

int tr_knn_range(tr_ctx *c, const double *states, int64_t n, int64_t first_query, int64_t n_queries, int32_t k, double max_distance,
                 int32_t *idx, double *dist) {
  if (!c) return TR_ERR_INVALID_ARG;
  if (first_query < 0 || n_queries < 0 || first_query + n_queries > n) return fail(c, TR_ERR_OUT_OF_RANGE, "query range outside the states");
  if (n_queries > 0 && (!idx || !dist)) return fail(c, TR_ERR_INVALID_ARG, "bad argument");
  if (n_queries == 0) return TR_OK;
  return knn_impl(c, states, n, k, max_distance, idx, dist, nullptr, 0, nullptr, first_query, n_queries);
}

static int knn_table_edges_impl(tr_ctx *c, const int32_t *idx, int64_t n, int32_t k, int32_t *edges, int64_t capacity, int64_t *n_edges, bool dev) {
  if (!c) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  if (!n_edges || n < 0 || k < 1 || capacity < 0 || (capacity > 0 && !edges) || (n > 0 && !idx)) return fail(c, TR_ERR_INVALID_ARG, "bad argument");
  *n_edges = 0;
  if (n == 0) return TR_OK;
  if (n > (int64_t)1 << 31) return fail(c, TR_ERR_INVALID_ARG, "too many states");
  if (!dev)
    for (int64_t i = 0; i < n * k; i++)
      if (idx[i] < -1 || idx[i] >= n) return fail(c, TR_ERR_OUT_OF_RANGE, "neighbour index outside the table");
  HIP_TRY(c, hipSetDevice(c->device));
  if (dev) HIP_TRY(c, hipDeviceSynchronize());                                 // the table may have been written on any stream of the caller
  if (c->last_dev_used) HIP_TRY(c, hipEventSynchronize(c->last_dev_ev));      // shared sort scratch, see knn_impl
  // the table and the list in the grow-only scratch of the neighbour search (a hipMalloc / hipFree pair per call costs 15 - 30 ms in a
  // process that holds the edge pool); device callers' arrays are used where they lie
  tr_ctx::KnnScratch &ks = c->knn;
  auto scratch = [&](int slot, size_t bytes) -> void * {
    if (ks.cap[slot] < bytes) {
      if (ks.p[slot]) { (void)hipFree(ks.p[slot]); ks.p[slot] = nullptr; ks.cap[slot] = 0; }
      const size_t want = bytes + bytes / 8;
      if (hipMalloc(&ks.p[slot], want) != hipSuccess) return nullptr;
      ks.cap[slot] = want;
    }
    return ks.p[slot];
  };
  const int64_t cap_e = std::min<int64_t>(capacity, n * (int64_t)k);
  const int32_t *d_i = idx;
  int32_t *d_e = edges;
  hipError_t e = hipSuccess;
  if (!dev) {
    int32_t *up = (int32_t *)scratch(7, (size_t)n * k * sizeof(int32_t));
    d_e = cap_e > 0 ? (int32_t *)scratch(11, (size_t)cap_e * 2 * sizeof(int32_t)) : nullptr;
    if (!up || (cap_e > 0 && !d_e)) return fail(c, TR_ERR_HIP, "hipMalloc failed (neighbour table)");
    e = hipMemcpy(up, idx, (size_t)n * k * sizeof(int32_t), hipMemcpyHostToDevice);
    d_i = up;
  } else {
    uint32_t *flag = (uint32_t *)scratch(12, sizeof(uint32_t));
    if (!flag) return fail(c, TR_ERR_HIP, "hipMalloc failed (neighbour table)");
    uint32_t bad = 0;
    e = hipMemsetAsync(flag, 0, sizeof(uint32_t), nullptr);
    if (e == hipSuccess) {
      // -1 (no neighbour) .. n - 1
      hipLaunchKernelGGL(trk::index_range_check, dim3((unsigned)((n * k + 255) / 256)), dim3(256), 0, nullptr, d_i, n * (int64_t)k, (int32_t)n, flag, 1);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(&bad, flag, sizeof(bad), hipMemcpyDeviceToHost);
    if (e == hipSuccess && bad) return fail(c, TR_ERR_OUT_OF_RANGE, "neighbour index outside the table");
  }
  if (e == hipSuccess) e = trk::knn_edge_list(c->merge, d_i, n, (int)k, d_e, cap_e, n_edges, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  const int64_t m = std::min<int64_t>(*n_edges, cap_e);
  if (e == hipSuccess && m > 0 && !dev) e = hipMemcpy(edges, d_e, (size_t)m * 2 * sizeof(int32_t), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail(c, TR_ERR_HIP, std::string("knn edge list: ") + hipGetErrorString(e));
  return TR_OK;
}

int tr_knn_table_edges(tr_ctx *c, const int32_t *idx, int64_t n, int32_t k, int32_t *edges, int64_t capacity, int64_t *n_edges) {
  return knn_table_edges_impl(c, idx, n, k, edges, capacity, n_edges, false);
}

// Device-resident forms of the sharded connection loop: a rank's rows of the table from vertices in HBM into a device array (to be
// all-gathered there), and the edge list of a gathered table written straight into the caller's device array.
int tr_knn_range_dev(tr_ctx *c, const double *d_states, int64_t n, int64_t first_query, int64_t n_queries, int32_t k, double max_distance,
                     int32_t *d_idx) {
  if (!c) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);         // (before the first fail(): it writes the context's error text)
  if (first_query < 0 || n_queries < 0 || first_query + n_queries > n) return fail(c, TR_ERR_OUT_OF_RANGE, "query range outside the states");
  if (n_queries > 0 && !d_idx) return fail(c, TR_ERR_INVALID_ARG, "bad argument");
  if (n_queries == 0) return TR_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipDeviceSynchronize());                  // d_states may have been written on any stream of the caller
  return knn_impl(c, d_states, n, k, max_distance, d_idx, nullptr, nullptr, 0, nullptr, first_query, n_queries, true);
}
int tr_knn_table_edges_dev(tr_ctx *c, const int32_t *d_idx, int64_t n, int32_t k, int32_t *d_edges, int64_t capacity, int64_t *n_edges) {
  return knn_table_edges_impl(c, d_idx, n, k, d_edges, capacity, n_edges, true);
}

int tr_knn_edges(tr_ctx *c, const double *states, int64_t n, int32_t k, double max_distance, int32_t *edges, int64_t capacity,
                 int64_t *n_edges) {
  if (!c) return TR_ERR_INVALID_ARG;
  if (!n_edges || capacity < 0 || (capacity > 0 && !edges)) return fail(c, TR_ERR_INVALID_ARG, "bad argument");
  *n_edges = 0;
  return knn_impl(c, states, n, k, max_distance, nullptr, nullptr, edges, capacity, n_edges);
}

// The same with the states in HBM and the edge list left there (for tr_validate_edges_indexed_dev): only *n_edges comes back.
int tr_knn_edges_dev(tr_ctx *c, const double *d_states, int64_t n, int32_t k, double max_distance, int32_t *d_edges, int64_t capacity,
                     int64_t *n_edges) {
  if (!c) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);         // (before the first fail(): it writes the context's error text)
  if (!n_edges || capacity < 0 || (capacity > 0 && !d_edges)) return fail(c, TR_ERR_INVALID_ARG, "bad argument");
  *n_edges = 0;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipDeviceSynchronize());                  // d_states may have been written on any stream of the caller
  return knn_impl(c, d_states, n, k, max_distance, nullptr, nullptr, d_edges, capacity, n_edges, 0, -1, true);
}
}  // extern "C"
namespace {
// queries [q0, q0 + nq) of the n states (nq < 0: all of them) against all n states as candidates
int knn_impl(tr_ctx *c, const double *states, int64_t n, int32_t k, double max_distance, int32_t *idx, double *dist,
             int32_t *edges, int64_t edge_capacity, int64_t *n_edges, int64_t q0, int64_t nq, bool dev) {
  // dev: `states`, `idx`, `dist` and `edges` are device arrays (tr_knn_edges_dev, tr_knn_range_dev)
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  if (n < 0 || k < 1 || (n > 0 && !states)) return fail(c, TR_ERR_INVALID_ARG, "bad argument");
  if (n == 0) return TR_OK;
  if (nq < 0) { q0 = 0; nq = n; }
  if ((size_t)k * 64 * 12 > 60 * 1024) return fail(c, TR_ERR_INVALID_ARG, "k too large (at most 80)");
  if (n >= (int64_t)1 << 31) return fail(c, TR_ERR_INVALID_ARG, "too many states");
  HIP_TRY(c, hipSetDevice(c->device));
  // the sort scratch (tr_ctx::merge) is shared with the ordering pass of a *_dev validity call that may still run on the
  // caller's stream
  if (c->last_dev_used) HIP_TRY(c, hipEventSynchronize(c->last_dev_ev));
  const int S = c->K.state_size, N = c->K.n_tendons;
  double ext2 = 0;
  for (int i = 0; i < N; i++) ext2 += c->max_tension[i] * c->max_tension[i];
  const double ext = std::sqrt(ext2);                         // RealVectorStateSpace::getMaximumExtent
  trk::KnnMetric m{N, c->K.enable_rotation, c->K.enable_retraction, S, ext / (4.0 * M_PI), 2.0 * ext / c->K.L};
  // candidate slices: enough waves to keep ~8 on every SIMD (see knn_kernel.hpp); at most 32 slices
  const int64_t qblocks = (nq + 63) / 64;
  // (round 2 sliced the candidate range over blockIdx.y and merged the slices' lists, and ran a seeding pass first; the
  // expanding cell search of round 3 needs neither: a wave's candidates are a few thousand, nearest first)
  const int nslice = 1;
  const int64_t half_window = 0;                 // (no seeding pass)
  double *d_s = nullptr, *d_ss = nullptr, *d_d = nullptr, *d_pd = nullptr, *d_seed = nullptr;
  uint32_t *d_ck[2] = {nullptr, nullptr};
  int32_t *d_perm = nullptr, *d_pt = nullptr, *d_ql = nullptr, *d_i = nullptr, *d_pi = nullptr, *d_cs = nullptr;
  trk::KnnCells cg{};
  constexpr int kMaxCells = 64;
  // k <= 64: a wave per query over a THREE-key cell grid (knn_kernel.hpp: knn_wave_query); larger k (or TENDON_HIP_KNN=lanes, read
  // per call: tests and A/B timing): a lane per query over two keys with the list in an LDS column
  const char *knn_mode = std::getenv("TENDON_HIP_KNN");
  const bool wave_form = k <= 64 && !(knn_mode && std::strcmp(knn_mode, "lanes") == 0);
  int rc = TR_OK;
  auto scratch = [&](int slot, size_t bytes, auto **out) -> bool {          // grow-only buffers kept in the context
    tr_ctx::KnnScratch &ks = c->knn;
    if (ks.cap[slot] < bytes) {
      if (ks.p[slot]) { (void)hipFree(ks.p[slot]); ks.p[slot] = nullptr; ks.cap[slot] = 0; }
      const size_t want = bytes + bytes / 8;
      if (hipMalloc(&ks.p[slot], want) != hipSuccess) return false;
      ks.cap[slot] = want;
    }
    *out = reinterpret_cast<std::remove_reference_t<decltype(**out)> *>(ks.p[slot]);
    return true;
  };
  do {
    const size_t nn = (size_t)n, qq = (size_t)nq;
    // (slot 2: the two cell-id arrays of the sort; slot 3: the cells' first positions)
    uint32_t *ck = nullptr;
    if (!scratch(0, nn * S * sizeof(double), &d_s) || !scratch(1, nn * S * sizeof(double), &d_ss) || !scratch(2, 2 * nn * sizeof(uint32_t), &ck) ||
        !scratch(3, ((size_t)kMaxCells * kMaxCells * kMaxCells + 1) * sizeof(int32_t), &d_cs) || !scratch(4, nn * sizeof(int32_t), &d_perm) || !scratch(5, nn * sizeof(int32_t), &d_pt) ||
        !scratch(6, qq * k * sizeof(double), &d_d) || !scratch(7, qq * k * sizeof(int32_t), &d_i) ||
        (half_window && !scratch(8, qq * sizeof(double), &d_seed)) ||
        (nslice > 1 && (!scratch(9, qq * nslice * k * sizeof(double), &d_pd) || !scratch(10, qq * nslice * k * sizeof(int32_t), &d_pi)))) {
      rc = fail(c, TR_ERR_HIP, "hipMalloc failed (neighbour search scratch)"); break;
    }
    d_ck[0] = ck; d_ck[1] = ck + nn;
    if (hipMemcpy(d_s, states, (size_t)n * S * sizeof(double), dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice) != hipSuccess) { rc = fail(c, TR_ERR_HIP, "hipMemcpy failed"); break; }
    {
      ProfScope ps(c, 3, nullptr);
      // the two coordinates the search is ordered and windowed by (knn_kernel.hpp): the retraction term of the metric when there
      // is one (its spread, 2 extent, is several times a tension's) and the first tension, else the first two tensions.  Cells
      // about a quarter of the expected k-th neighbour distance wide: for n points spread over the d metric dimensions with ranges R_i
      // that distance is ~ (k / (n V_d))^(1/d) (prod R_i)^(1/d), V_d^(1/d) ~ 1.5 for d = 2 .. 6.  The estimate only sizes the
      // grid: any cell width gives the same neighbours.
      const bool by_ret = c->K.enable_retraction != 0;
      cg.col0 = by_ret ? S - 1 : 0; cg.scale0 = by_ret ? m.w_ret : 1.0;
      cg.col1 = by_ret ? 0 : (N >= 2 ? 1 : -1); cg.scale1 = 1.0;
      cg.col2 = wave_form ? (by_ret ? (N >= 2 ? 1 : -1) : (N >= 3 ? 2 : -1)) : -1; cg.scale2 = 1.0; cg.A = 1;
      double logprod = 0.0;
      for (int i = 0; i < N; i++) logprod += std::log(std::max(c->max_tension[i], 1e-12));
      if (c->K.enable_rotation) logprod += std::log(m.w_rot * 2.0 * M_PI);
      if (c->K.enable_retraction) logprod += std::log(m.w_ret * c->K.L);
      const double r_est = std::pow((double)k / (double)n, 1.0 / S) * std::exp(logprod / S) / 1.5;
      double width = 0.25 * r_est;                  // measured at 10^5 and 6 x 10^5 states: 0.25 r (or the 64-cell cap) 6.2 / 27 ms, 0.5 r 7.3 / 31 ms, r 13 / 51 ms
      if (const char *e = std::getenv("TENDON_HIP_KNN_CELL")) { const double v = std::atof(e); if (v > 0) width *= v; }      // tuning only
      int max_cells = n >= 4096 ? (int)std::min<int64_t>(kMaxCells, std::max<int64_t>(1, (int64_t)std::sqrt((double)n / 64.0))) : 1;
      if (cg.col2 >= 0) {
        // three keys, one query per wave: ~12 states per cell (a run of a column's cells inside the search box is then about one
        // 64-candidate tile), cells about half the expected k-th distance wide
        max_cells = n >= 512 ? (int)std::min<int64_t>(kMaxCells, std::max<int64_t>(1, (int64_t)std::cbrt((double)n / 12.0))) : 1;
        width = 2.0 * width;
      }
      const hipError_t e = trk::sort_states_by_cells(c->merge, d_s, n, S, cg, width, max_cells, d_ss, d_perm, d_cs, d_ck, d_pt, nullptr);
      if (e != hipSuccess) { rc = fail(c, TR_ERR_HIP, std::string("knn sort: ") + hipGetErrorString(e)); break; }
    }
    if (nq != n) {
      // a range of queries: their positions in sorted order, ascending (neighbouring waves search neighbouring cells), on the device
      const int64_t nb = (n + 1023) / 1024;
      uint32_t *d_bs = nullptr;
      if (!scratch(11, (size_t)nq * sizeof(int32_t), &d_ql) || !scratch(12, (size_t)nb * sizeof(uint32_t), &d_bs)) { rc = fail(c, TR_ERR_HIP, "hipMalloc failed"); break; }
      ProfScope ps(c, 3, nullptr);
      hipLaunchKernelGGL(trk::range_positions_count, dim3((unsigned)nb), dim3(256), 0, nullptr, (const int32_t *)d_perm, n, q0, nq, d_bs);
      hipLaunchKernelGGL(trk::range_positions_scan, dim3(1), dim3(1024), 0, nullptr, d_bs, nb);
      hipLaunchKernelGGL(trk::range_positions_write, dim3((unsigned)nb), dim3(256), 0, nullptr, (const int32_t *)d_perm, n, q0, nq, (const uint32_t *)d_bs, d_ql);
      if (hipGetLastError() != hipSuccess) { rc = fail(c, TR_ERR_HIP, "query positions: launch failed"); break; }
    }
    {
      ProfScope ps(c, 3, nullptr);
      const dim3 grid((unsigned)qblocks, (unsigned)nslice);
      const size_t lds = (size_t)k * 64 * 12;
      int32_t *oi = nslice > 1 ? d_pi : d_i;
      double *od = nslice > 1 ? d_pd : d_d;
      const int variant = (c->K.enable_rotation ? 1 : 0) | (c->K.enable_retraction ? 2 : 0);
      constexpr int kWaveQ = 1, kWaveG = 1;                       // queries per wave, tiles in flight (measured: knn_kernel.hpp)
      const dim3 wgrid((unsigned)((nq + 4 * kWaveQ - 1) / (4 * kWaveQ)));
#define TRK_KNN_K(NT, R, T, KC) do { if (wave_form) hipLaunchKernelGGL((trk::knn_wave_query<NT, R, T, kWaveQ, kWaveG>), wgrid, dim3(256), 0, nullptr, d_ss, cg, d_cs, d_perm, d_ql, nq, m, \
                                                  (int)k, max_distance, q0, oi, od); \
                                     else hipLaunchKernelGGL((trk::knn_bruteforce<NT, R, T, KC>), grid, dim3(64), lds, nullptr, d_ss, cg, d_cs, d_perm, d_ql, nq, n, m, (int)k, \
                                                  max_distance, (int64_t)0, (const double *)nullptr, (double *)nullptr, q0, oi, od); } while (0)
      // (KCAP = 16 / 32 -- the list in registers -- measured SLOWER than the LDS column at every k: 29.9 against 27.4 ms at k = 11,
      // 416 against 57 ms at k = 21 for 6 x 10^5 states; the instantiations are not built)
#define TRK_KNN(NT, R, T) TRK_KNN_K(NT, R, T, 0)
      switch (N) {
#define TRK_CASE(NT) case NT: if (variant == 0) TRK_KNN(NT, false, false); else if (variant == 1) TRK_KNN(NT, true, false); \
                              else if (variant == 2) TRK_KNN(NT, false, true); else TRK_KNN(NT, true, true); break;
        TRK_CASE(1) TRK_CASE(2) TRK_CASE(3) TRK_CASE(4) TRK_CASE(5) TRK_CASE(6) TRK_CASE(7) TRK_CASE(8)
#undef TRK_CASE
#undef TRK_KNN
#undef TRK_KNN_K
        default: break;
      }
      if (nslice > 1)
        hipLaunchKernelGGL(trk::knn_merge, dim3((unsigned)qblocks), dim3(64), (size_t)nslice * 64, nullptr, d_pi, d_pd, nq, nslice, (int)k, max_distance, d_i, d_d);
    }
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { rc = fail(c, TR_ERR_HIP, "knn launch failed"); break; }
    const hipMemcpyKind down = dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (idx && hipMemcpy(idx, d_i, (size_t)nq * k * sizeof(int32_t), down) != hipSuccess) { rc = fail(c, TR_ERR_HIP, "copy back failed"); break; }
    if (dist && hipMemcpy(dist, d_d, (size_t)nq * k * sizeof(double), down) != hipSuccess) { rc = fail(c, TR_ERR_HIP, "copy back failed"); break; }
    if (n_edges) {
      // the undirected edge set of the table, deduplicated and ordered on the device (cache_merge.hip)
      int32_t *d_e = nullptr;                 // kept with the rest of the search's scratch: a hipMalloc / hipFree pair per call costs
                                              // 15 - 30 ms in a process that holds the edge pool (profiles/r02/create_roadmap_v1.txt)
      const int64_t cap_e = std::min<int64_t>(edge_capacity, n * (int64_t)k);
      if (cap_e > 0 && !scratch(11, (size_t)cap_e * 2 * sizeof(int32_t), &d_e)) { rc = fail(c, TR_ERR_HIP, "hipMalloc failed"); break; }
      hipError_t e = trk::knn_edge_list(c->merge, d_i, n, (int)k, d_e, cap_e, n_edges, nullptr);
      if (e == hipSuccess) e = hipDeviceSynchronize();
      const int64_t mm = std::min<int64_t>(*n_edges, cap_e);
      if (e == hipSuccess && mm > 0) e = hipMemcpy(edges, d_e, (size_t)mm * 2 * sizeof(int32_t), dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost);
      if (e != hipSuccess) { rc = fail(c, TR_ERR_HIP, std::string("knn edge list: ") + hipGetErrorString(e)); break; }
    }
  } while (0);
  return rc;
}
}  // namespace
extern "C" {

// KStarStrategy (OMPL 1.5.0 ompl/geometric/planners/prm/ConnectionStrategy.h, installed by setStarConnectionStrategy,
// motion-planning/VoxelCachedLazyPRM.cpp:1346-1356): k = ceil((e + e / dim) * ln(n)) with n = milestoneCount().  In
// createRoadmap every vertex is in the nearest-neighbour structure before the connection loop starts (:1463-1502), so
// one k serves the whole batch.
int tr_kstar_k(const tr_ctx *c, int64_t n_milestones) {
  if (!c || n_milestones < 1) return -1;
  const double e = 2.718281828459045235360287471352662498;   // boost::math::constants::e<double>()
  const double kc = e + e / (double)c->K.state_size;
  return (int)std::ceil(kc * std::log((double)n_milestones));
}

// ---- instrumentation -----------------------------------------------------------------------
int tr_profile_begin(tr_ctx *c) {
  if (!c) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  for (auto &v : c->events) { for (auto &e : v) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); } v.clear(); }
  c->profiling = true;
  return TR_OK;
}
int tr_profile_read(tr_ctx *c, int64_t launches[TR_PROFILE_SLOTS], double total_ms[TR_PROFILE_SLOTS]) {
  if (!c || !launches || !total_ms) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  for (int s = 0; s < TR_PROFILE_SLOTS; s++) {
    launches[s] = (int64_t)c->events[s].size();
    double tot = 0;
    for (auto &e : c->events[s]) {
      HIP_TRY(c, hipEventSynchronize(e.b));
      float ms = 0;
      HIP_TRY(c, hipEventElapsedTime(&ms, e.a, e.b));
      tot += ms;
    }
    total_ms[s] = tot;
  }
  return TR_OK;
}
int tr_profile_end(tr_ctx *c) {
  if (!c) return TR_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock_(c->mu);
  c->profiling = false;
  for (auto &v : c->events) { for (auto &e : v) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); } v.clear(); }
  return TR_OK;
}

}  // extern "C"

#include "edge_host.inc"
#include "sample_host.inc"
