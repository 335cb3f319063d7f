// roadmap.hip -- tr_roadmap_*: the interactive query loop of motion_planning::VoxelCachedLazyPRM on a cached
// roadmap (BASELINE config 5): solveWithRoadmap / constructSolution
// (motion-planning/VoxelCachedLazyPRM.cpp:1977-2096, :2689-2771), astarSearch (:2950-2976),
// computeVertexValidity / computeEdgeValidity on cached voxel sets (:2607-2631), clearValidity (:1656-1663).
//
// The reference answers one (start, goal) pair at a time: A* over the Boost graph, then every interior vertex of the
// candidate path is tested (cached voxel set AND obstacle octree), ALL invalid ones are removed, else the path's edges are
// tested from the goal side and the FIRST invalid one is removed; repeat until a path survives or start and goal fall
// into different components.  Each test is a tiny octree intersection -- latency-bound on the CPU and far too small for a
// GPU launch of its own.  Here a whole batch of queries advances in rounds:
//   round = [A* for every unresolved query on the graph minus what is known invalid: large rounds on the device, one wave per
//            query (search_kernel.hpp: roadmap_astar), shared with the host threads; small rounds on the host threads alone]
//           -> the union of the still-unknown vertices and edges on all candidate paths -> ONE K4 launch on that subset
//              (cached_subset_vs_grid; the caches live in HBM as one CSR) -> validity recorded, invalid items leave the graph
//   queries whose candidate path turned out all valid are done; the others search again next round.
//   Two shortcuts that change no answer (round 4): queries whose end points lie in different components of what is left of the
//   graph are answered from component labels computed on the device (component_labels; the reference's solutionComponent test),
//   and when queries are still open while ONE launch over every cached set is cheaper than another round of searches, everything
//   is tested and the next round is the last (the loop turns eager; TENDON_HIP_LAZY_ONLY=1 forbids it).
// tr_roadmap_revalidate is the eager form: one K4 pass over every cached set (well under a millisecond for 10^5..10^6
// items), after which every query resolves in its first round.
// Returned paths are the reference's: a path is accepted only when all its items are valid, and it is the shortest path
// of the graph minus the invalid items discovered so far -- which, all of its own items being valid, is also the
// shortest path of the graph minus ALL invalid items, whatever subset has been discovered (equal costs; equal vertex
// sequences unless two paths tie exactly).  What differs is bookkeeping only: every unknown edge of a vertex-clean path
// is tested in the round (the reference stops at the first invalid one), so the set of DISCOVERED invalid edges is a superset.
// A search is latency-bound pointer chasing: one search is ~8x faster on a host core than on a wave, but the device runs three
// thousand of them at once -- hence the shared schedule.  The heuristic is the reference's state-space distance, sharpened by landmark
// lower bounds (tr_roadmap_prepare): distances from a few extremal vertices over the FULL graph; |d(l, v) - d(l, goal)| never
// exceeds the distance from v to the goal, and stays a lower bound when invalid items leave the graph (distances only
// grow).  The path A* returns is still the shortest one -- an admissible heuristic only changes how few vertices are expanded.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/tendon_hip.h"
#include "search_kernel.hpp"

namespace {

// TENDON_HIP_ROADMAP_TIMING=1: the host phases of tr_roadmap_create / tr_roadmap_prepare on stderr (profiles/probe_query_object.py)
struct Laps {
  bool on = std::getenv("TENDON_HIP_ROADMAP_TIMING") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  const char *what;
  explicit Laps(const char *w) : what(w) { if (on) std::fprintf(stderr, "[%s]", what); }
  void lap(const char *name) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, " %s %.2f ms", name, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
  ~Laps() { if (on) std::fprintf(stderr, "\n"); }
};

// The library's host threads: started once per process and parked on a condition variable between jobs -- a tr_roadmap_solve runs six
// or seven parallel sections, and starting fifteen threads for each was 0.3 - 0.5 ms a time (2 - 3 ms of a 25 ms batch of queries).
// One job at a time; a caller that finds the team busy (another roadmap's call on another host thread) starts threads of its own as
// before.  Never destroyed: its threads wait inside it when the process ends.
class HostTeam {
  std::mutex mu_, job_mu_;
  std::condition_variable work_, done_;
  std::vector<std::thread> th_;
  const std::function<void(int)> *fn_ = nullptr;
  int T_ = 0, running_ = 0;
  uint64_t epoch_ = 0;
  void worker(int t) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(int)> *f = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        work_.wait(lk, [&] { return epoch_ != seen; });
        seen = epoch_;
        if (t < T_) f = fn_;
      }
      if (f) {
        inside() = true;
        (*f)(t);
        inside() = false;
        std::lock_guard<std::mutex> lk(mu_);
        if (--running_ == 0) done_.notify_all();
      }
    }
  }
  static bool &inside() { static thread_local bool in = false; return in; }   // this thread is running a section of the team's job
 public:
  static HostTeam &get() { static HostTeam *team = new HostTeam(); return *team; }
  // fn(1) .. fn(T - 1) on the team's threads, fn(0) on the caller's; false (nothing run) when the team is busy
  bool run(int T, const std::function<void(int)> &fn) {
    if (inside()) return false;                 // (a section started from inside a section: threads of its own, as when the team is busy)
    std::unique_lock<std::mutex> job(job_mu_, std::try_to_lock);
    if (!job.owns_lock()) return false;
    struct Mark { Mark() { inside() = true; } ~Mark() { inside() = false; } } mark;
    {
      std::lock_guard<std::mutex> lk(mu_);
      while ((int)th_.size() < T - 1) { const int t = (int)th_.size() + 1; th_.emplace_back([this, t] { worker(t); }); th_.back().detach(); }
      fn_ = &fn; T_ = T; running_ = T - 1; epoch_++;
    }
    work_.notify_all();
    fn(0);
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [&] { return running_ == 0; });
    fn_ = nullptr; T_ = 0;
    return true;
  }
};

// fn(t) for t = 0 .. T-1 on T host threads (the caller's included)
template <class F> void on_threads(int T, F &&fn) {
  if (T <= 1) { fn(0); return; }
  static const bool pooled = !(std::getenv("TENDON_HIP_HOST_TEAM") && std::atoi(std::getenv("TENDON_HIP_HOST_TEAM")) == 0);   // (A/B: 0 = threads per section)
  if (pooled) {
    const std::function<void(int)> f = [&fn](int t) { fn(t); };
    if (HostTeam::get().run(T, f)) return;
  }
  std::vector<std::thread> th;
  th.reserve((size_t)T - 1);
  for (int t = 1; t < T; t++) th.emplace_back([&fn, t] { fn(t); });
  fn(0);
  for (auto &x : th) x.join();
}

// Device buffers of the query objects come from a small per-process cache instead of hipMalloc / hipFree: a roadmap build attaches
// and later drops ~0.45 GB of them (block lists, offsets, the landmark arena), six allocations and frees of 0.1 - 1 ms each
// (set_caches 2.8 -> 2.2 ms, prepare 6.1 -> 5.7 ms).  A freed buffer is kept (up to kCacheMaxBytes / kCacheMaxEntries per
// process) and handed to the next request of at least half its size; a failed hipMalloc empties the cache and tries again.
struct DevCache {
  struct Buf { void *p; size_t bytes; int dev; };
  std::mutex mu;
  std::vector<Buf> idle;
  std::unordered_map<void *, Buf> live;
  size_t idle_bytes = 0;
  static constexpr size_t kCacheMaxBytes = (size_t)1 << 30, kCacheMaxEntries = 32;
  hipError_t alloc(int dev, void **out, size_t bytes) {
    bytes = std::max<size_t>(bytes, 256);
    std::lock_guard<std::mutex> lock(mu);
    size_t best = idle.size();
    for (size_t i = 0; i < idle.size(); i++)
      if (idle[i].dev == dev && idle[i].bytes >= bytes && idle[i].bytes <= 2 * bytes + ((size_t)1 << 20) &&
          (best == idle.size() || idle[i].bytes < idle[best].bytes)) best = i;
    if (best < idle.size()) {
      const Buf b = idle[best];
      idle.erase(idle.begin() + (long)best);
      idle_bytes -= b.bytes;
      live[b.p] = b;
      *out = b.p;
      return hipSuccess;
    }
    const size_t cap = (bytes + ((size_t)1 << 16) - 1) & ~(((size_t)1 << 16) - 1);
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, cap);
    if (e != hipSuccess && !idle.empty()) {                     // out of memory with buffers parked here: give them back and try again
      for (const Buf &b : idle) (void)hipFree(b.p);
      idle.clear(); idle_bytes = 0;
      e = hipMalloc(&p, cap);
    }
    if (e != hipSuccess) return e;
    live[p] = Buf{p, cap, dev};
    *out = p;
    return hipSuccess;
  }
  // everything parked goes back to the device (an allocation elsewhere in the library ran out of memory: tr_dev_cache_trim)
  void trim() {
    std::lock_guard<std::mutex> lock(mu);
    for (const Buf &b : idle) (void)hipFree(b.p);
    idle.clear(); idle_bytes = 0;
  }
  void release(void *p) {
    if (!p) return;
    // hipFree synchronised with the device; a parked buffer may be handed to its next owner at once, so work that still uses it
    // (a roadmap destroyed with launches in flight) is waited for here
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lock(mu);
    auto it = live.find(p);
    if (it == live.end()) { (void)hipFree(p); return; }
    const Buf b = it->second;
    live.erase(it);
    if (idle.size() < kCacheMaxEntries && idle_bytes + b.bytes <= kCacheMaxBytes) { idle.push_back(b); idle_bytes += b.bytes; }
    else (void)hipFree(b.p);
  }
};
DevCache &dev_cache() { static DevCache c; return c; }
}  // namespace
void release_idle_search_tables();   // (below: the search tables of the roadmaps that are not in a call right now)
void tr_dev_cache_trim() { release_idle_search_tables(); dev_cache().trim(); }
namespace {

enum : uint8_t { V_UNKNOWN = 0, V_VALID = 1, V_INVALID = 2 };   // VALIDITY_UNKNOWN / VALIDITY_TRUE / removed from the graph

int host_threads(int want) {
  if (want > 0) return want;
  unsigned n = std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = (unsigned)CPU_COUNT(&set);
  if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {          // a container's CPU quota
    long long quota = 0, period = 0;
    char q[32] = {0};
    if (std::fscanf(f, "%31s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) {
      quota = std::atoll(q);
      if (quota > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, quota / period));
    }
    std::fclose(f);
  }
  return (int)std::max(1u, std::min(n, 64u));
}

struct Node {                          // A* state of one vertex in one search: 32 B, one cache line touched per visit
  double g, h;
  int32_t parent, parent_edge;
  uint32_t stamp, closed;
};
struct Scratch {                       // per host thread, reused across queries: generation-stamped A* state
  std::vector<Node> node;
  std::vector<std::pair<double, int32_t>> heap;
  uint32_t gen = 0;
  double trace_f[4] = {0, 0, 0, 0};    // (TENDON_HIP_SEARCH_HIST=<file>: h(start), and the key taken off the list at the 2 000th / 3 000th / 4 000th expansion)
};

struct Arc { int32_t v, e; double w; };   // one adjacency entry: neighbour, edge id, edge weight (16 B: four per cache line)

}  // namespace

// A heap array that is NOT zeroed when sized (std::vector would touch all of it on the calling thread: for the 19 MB of arcs of a
// 100 k-vertex roadmap the page faults of that pass cost more than filling them): the pages are first touched by the threads that fill them.
template <class T> struct RawArray {
  std::unique_ptr<T[]> p;
  size_t n = 0;
  void resize_uninit(size_t m) { p.reset(new T[m]); n = m; }
  size_t size() const { return n; }
  T *data() { return p.get(); }
  const T *data() const { return p.get(); }
  T &operator[](size_t i) { return p[i]; }
  const T &operator[](size_t i) const { return p[i]; }
};

struct tr_roadmap {
  std::mutex mu;
  std::atomic<bool> busy{false};       // a call holds `mu` (RmLock): the out-of-memory trim leaves this roadmap alone
  tr_ctx *ctx = nullptr;
  std::string err;
  int S = 0, NT = 0;
  bool rot = false, ret = false;
  double w_rot = 0, w_ret = 0;
  int64_t V = 0, E = 0;
  std::vector<double> states;
  RawArray<int32_t> eu, ev;
  RawArray<double> w;
  std::vector<int64_t> adj_off;        // CSR adjacency, both directions
  RawArray<Arc> adj;
  // landmark lower bounds (tr_roadmap_prepare): lm_d[v * lm_n + l] = graph distance landmark l -> v over all edges, as float; SR_LM_FAR
  // (not +inf) where v is not connected to l: what the branch-free bounds of the host A* and of the kernel's rows read
  // (+inf = not connected); lm_n = 0: none, -1: not built yet (built with the default count by the first solve)
  int lm_n = -1;
  std::vector<float> lm_d;
  std::vector<int32_t> lm_v;
  bool lm_mismatch = false;            // TENDON_HIP_LANDMARKS=check: the device's table differed from the host's
  std::vector<uint8_t> vstat, estat;   // V_*
  std::vector<uint8_t> vpresent, epresent;
  // cached voxel sets in HBM: one CSR, items [0, V) = vertices, [V, V + E) = edges
  bool has_caches = false;
  uint32_t *d_ids = nullptr; uint64_t *d_masks = nullptr; int64_t *d_off = nullptr;
  int32_t *d_list = nullptr; uint8_t *d_hit = nullptr; uint64_t *d_bits = nullptr;
  uint64_t *h_bits = nullptr;          // pinned image of d_bits (tr_roadmap_revalidate)
  std::vector<uint64_t> absent;        // per word of the combined item index: bit set = the item has no cache (invalid for good)
  int64_t list_cap = 0;
  int64_t nnz = 0;
  // results of the last tr_roadmap_solve
  std::vector<int64_t> path_off;
  std::vector<int32_t> path_v;
  std::vector<std::vector<int32_t>> paths_buf, paths_e_buf;   // per query of the last solve: vertices goal ... start, their edges (capacity kept)
  // statistics of the last solve
  int64_t st_rounds = 0, st_items_checked = 0, st_astar_runs = 0, st_expanded = 0;
  std::vector<Scratch> scratch;
  // the graph searches on the device (search_kernel.hpp): everything below lives in HBM for the life of the roadmap
  struct DevSearch {
    int state = 0;                       // 0: not set up yet, 1: ready, -1: not available for this roadmap (reason in `why`)
    std::string why;
    int64_t slots = 0, nq_cap = 0;
    int64_t max_slots = 0;               // what the device holds at once (or TENDON_HIP_SEARCH_SLOTS); `slots` follows the rounds' sizes up to it
    int32_t lc0 = 12;                    // log2 of the records of a slot's own table
    int32_t pool_n[trk::SR_CLASSES] = {0, 0, 0, 0}, pool_word[trk::SR_CLASSES] = {0, 0, 0, 0};
    size_t ctl_bytes = 0, table_bytes = 0;
    uint32_t pbuf_cap = 0;
    uint64_t gens_issued = 0;
    bool lm_current = false;
    char *arena = nullptr;               // adjacency rows | states | landmark table | validity bytes | control words + pool bitmaps
    char *tables = nullptr;              // the slots' tables, then the pool's, class by class
    char *pool[trk::SR_CLASSES] = {nullptr, nullptr, nullptr, nullptr};
    char *qarena = nullptr;              // per-round arrays (queries, results, packed paths)
    trk::SArc *d_rows = nullptr; char *d_vrows = nullptr;        // adjacency rows; per vertex: state | landmark distances
    int32_t row_bytes = 0;
    uint8_t *d_vstat = nullptr, *d_estat = nullptr, *d_deg = nullptr;
    uint32_t *d_ctl = nullptr;
    int32_t *d_qs = nullptr, *d_qg = nullptr, *d_poff = nullptr, *d_plen = nullptr, *d_pbuf = nullptr;
    uint8_t *d_found = nullptr;
    uint32_t *h_handback = nullptr, *d_handback = nullptr;   // pinned: the kernel marks a search here the moment it hands it back
    int64_t handback_cap = 0;
    double kernel_ms = 0, host_after_ms = 0;                   // of the last shared round: the kernel's span, the host threads' work after it
    hipEvent_t ev[2] = {nullptr, nullptr};                     // around every roadmap_astar launch (tr_roadmap_profile)
    bool ev_pending = false;
    double st_kernel_ms = 0; int64_t st_launches = 0;          // of the last tr_roadmap_solve
    int64_t st_queries = 0, st_fallbacks = 0, st_host_share = 0, st_moves = 0, st_expanded = 0, st_grows = 0, st_max_records = 0;   // of the last tr_roadmap_solve
    int64_t in_flight = 0;               // queries of the launch that has not been collected yet
    bool budget_from_env = false;
    double share = -1.0;                 // the host threads' share of a shared round (< 0: not chosen yet); follows the two sides' times
    int64_t budget = 0;                  // expansions per search before the kernel hands it back (0: not chosen yet); doubles when
                                         // more than a twentieth of a round came back -- a larger roadmap has longer searches
    // sweep_search: a single-source sweep over the valid arcs for the few searches that expand a large part of the graph
    std::mutex sweep_mu;
    hipStream_t sweep_stream = nullptr;
    char *sweep_arena = nullptr;         // distances (ordered bit patterns) | validity bytes of this round | flags
    unsigned long long *sweep_h_dist = nullptr;   // pinned: the distances come back here
    int64_t sweep_round = -1;            // the round whose validity bytes the arena holds
    int64_t st_sweeps = 0;               // searches answered this way in the last tr_roadmap_solve
    std::vector<int32_t> h_qs, h_qg;     // host images of what the pending copies read
    std::vector<char> h_vrows;
  } ds;
  // connected components of the roadmap minus what is known invalid (component_labels): the edge list and the labels in HBM
  struct DevComp {
    int state = 0;                       // 0: not set up yet, 1: ready, -1: not available
    char *arena = nullptr;
    int32_t *d_eu = nullptr, *d_ev = nullptr, *d_parent = nullptr, *d_label = nullptr;
    uint8_t *d_vstat = nullptr, *d_estat = nullptr;
    bool status_current = false;         // d_vstat / d_estat hold this round's validity bytes (the search kernel reads them too)
    bool wanted = false;                 // a search on this roadmap has walked a component in vain: label every large round from now on
    std::vector<int32_t> label;          // per vertex: the smallest vertex of its component
    int64_t st_cut = 0;                  // searches of the last solve answered by the labels alone
  } dc;
};

namespace {

// Every entry point that works on a roadmap holds its mutex through this; `busy` lets the trim path (which may run on the very
// thread that holds the mutex: an allocation inside a call ran out of memory) tell without touching the mutex.
struct RmLock {
  tr_roadmap *r;
  explicit RmLock(tr_roadmap *r_) : r(r_) { r->mu.lock(); r->busy.store(true, std::memory_order_release); }
  ~RmLock() { r->busy.store(false, std::memory_order_release); r->mu.unlock(); }
  RmLock(const RmLock &) = delete;
  RmLock &operator=(const RmLock &) = delete;
};
// the live roadmaps (tr_roadmap_create .. tr_roadmap_destroy), for release_idle_search_tables
std::mutex g_roadmaps_mu;
std::vector<tr_roadmap *> g_roadmaps;

int rfail(tr_roadmap *r, int code, const std::string &m) { if (r) r->err = m; return code; }

#define RM_HIP(r, expr)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) return rfail(r, TR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// CompoundStateSpace::distance with the subspace weights of motion-planning/Problem.cpp:112-152
inline double state_distance(const tr_roadmap *r, const double *a, const double *b) {
  double s = 0;
  for (int i = 0; i < r->NT; i++) { const double d = a[i] - b[i]; s += d * d; }
  double dist = std::sqrt(s);
  int k = r->NT;
  if (r->rot) {
    double d = std::fabs(a[k] - b[k]);
    d = (d > M_PI) ? 2.0 * M_PI - d : d;
    dist += r->w_rot * d;
    k++;
  }
  if (r->ret) { const double d = a[k] - b[k]; dist += r->w_ret * std::sqrt(d * d); }
  return dist;
}

// relative slack that keeps the float-stored landmark distances on the safe side of the true ones (rounding to float is
// 2^-24 relative per distance; the fp64 path sums behind them differ from A*'s own sums by ~1e-16 per hop)
constexpr double kLmSlack = 1.0 / (1 << 21);

// astarSearch (:2950-2976): A* with the state-space distance to the goal as heuristic (costHeuristic :2773-2775 ->
// motionCostHeuristic), edge weights as given; stops when the goal is taken off the open list (AStarGoalVisitor).
// Vertices / edges known invalid are not part of the graph (the reference has removed them).  Returns false when the
// goal cannot be reached.  path: goal ... start (vertex ids), path_e: the edges between them.
// With landmark tables the heuristic is the larger of that distance and the landmark bounds: admissible, so the goal
// leaves the open list with the same (optimal) cost and, ties apart, the same parents; a vertex whose cost improves after
// it was expanded is opened again (with the consistent state-space distance alone that never happens).
// `cap` > 0: the search gives up after that many expansions (*abandoned = true, false returned): tr_roadmap_solve then answers it with a
// parallel sweep on the device (sweep_search below) -- a search that expands a large part of the graph is a poor fit for one core.
bool astar(const tr_roadmap *r, Scratch &sc, int32_t start, int32_t goal, std::vector<int32_t> &path, std::vector<int32_t> &path_e,
           int64_t &expanded, int64_t cap = 0, bool *abandoned = nullptr) {
  if (sc.node.size() != (size_t)r->V) { sc.node.assign((size_t)r->V, Node{0.0, 0.0, -1, -1, 0u, 0u}); sc.gen = 0; }
  if (++sc.gen == 0) { for (Node &nd : sc.node) nd.stamp = 0u; sc.gen = 1; }
  Node *node = sc.node.data();
  const uint32_t gen = sc.gen;
  auto &heap = sc.heap;
  heap.clear();
  const double *sg = &r->states[(size_t)goal * r->S];
  const int L = r->lm_n > 0 ? r->lm_n : 0;
  const float *lg = L ? &r->lm_d[(size_t)goal * L] : nullptr;
  const double inf = std::numeric_limits<double>::infinity();
  auto heuristic = [&](int32_t v) -> double {
    double h = state_distance(r, &r->states[(size_t)v * r->S], sg);
    if (L) {
      // (SR_LM_FAR where a vertex is not connected to the landmark: two far entries bound nothing -- their term is hugely negative --, one
      // makes the term huge: different components; no comparison with infinity, no branch: the loop vectorises, and every term is the
      // kernel's (search_kernel.hpp: heuristic), float operation for float operation)
      const float *lv = &r->lm_d[(size_t)v * L];
      const float slack = (float)kLmSlack;
      float b8[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};   // (eight running maxima side by side: element-wise work the compiler turns into vector code)
      int l = 0;
      for (; l + 8 <= L; l += 8)
        for (int j = 0; j < 8; j++) {
          const float a = lv[l + j], b = lg[l + j];
          const float hi = a > b ? a : b, lo = a > b ? b : a;
          const float t = (hi - lo) - slack * hi;                // float subtraction of nearby values: error <= 2^-24 hi, inside the slack
          b8[j] = t > b8[j] ? t : b8[j];
        }
      float best = 0.0f;
      for (; l < L; l++) {
        const float a = lv[l], b = lg[l];
        const float hi = a > b ? a : b, lo = a > b ? b : a;
        const float t = (hi - lo) - slack * hi;
        best = t > best ? t : best;
      }
      for (int j = 0; j < 8; j++) best = b8[j] > best ? b8[j] : best;
      if (best >= 0.5f * trk::SR_LM_FAR) return inf;
      if ((double)best > h) h = (double)best;
    }
    return h;
  };
  auto cmp = [](const std::pair<double, int32_t> &a, const std::pair<double, int32_t> &b) { return a.first > b.first; };
  node[start] = Node{0.0, heuristic(start), start, -1, gen, 0u};
  if (node[start].h == inf) return false;
  static const bool trace = [] { const char *e = std::getenv("TENDON_HIP_SEARCH_HIST"); return e && e[0] == '/'; }();
  const int64_t expanded0 = expanded;
  if (trace) { sc.trace_f[0] = node[start].h; sc.trace_f[1] = sc.trace_f[2] = sc.trace_f[3] = 0.0; }
  heap.emplace_back(node[start].h, start);
  bool found = false;
  constexpr bool node_is_new_hint = true;       // (the state / landmark rows are only read for a vertex met for the first time; most are)
  while (!heap.empty()) {
    std::pop_heap(heap.begin(), heap.end(), cmp);
    const int32_t u = heap.back().second;
    heap.pop_back();
    if (!heap.empty()) {                        // the likely next vertex: its record and its arcs on their way while this one is expanded
      const int32_t nx = heap.front().second;   // (the top's two children as well: measured, no gain)
      __builtin_prefetch(&node[nx]);
      __builtin_prefetch(r->adj.data() + r->adj_off[nx]);
    }
    if (node[u].closed) continue;               // a stale entry of a vertex already expanded with a better cost
    node[u].closed = 1u;
    expanded++;
    if (trace) {
      const int64_t n_ = expanded - expanded0;
      if (n_ == 2000) sc.trace_f[1] = node[u].g + node[u].h;
      else if (n_ == 3000) sc.trace_f[2] = node[u].g + node[u].h;
      else if (n_ == 4000) sc.trace_f[3] = node[u].g + node[u].h;
    }
    if (cap > 0 && expanded - expanded0 > cap) { if (abandoned) *abandoned = true; return false; }
    if (u == goal) { found = true; break; }
    const double gu = node[u].g;
    const Arc *arc = r->adj.data() + r->adj_off[u], *end = r->adj.data() + r->adj_off[u + 1];
    // (every neighbour is four lines somewhere in a few hundred megabytes -- its record, its validity byte, its state, its landmark
    // row: asked for together before the first is used, the misses overlap instead of queueing behind one another)
    for (const Arc *a = arc; a != end; ++a) {
      const int32_t v = a->v;
      __builtin_prefetch(&node[v], 1);
      __builtin_prefetch(&r->vstat[v]);
      __builtin_prefetch(&r->estat[a->e]);
      if (node_is_new_hint) {
        __builtin_prefetch(&r->states[(size_t)v * r->S]);
        if (L) __builtin_prefetch(&r->lm_d[(size_t)v * L]);
      }
    }
    for (; arc != end; ++arc) {
      const int32_t e = arc->e, v = arc->v;
      if (r->estat[e] == V_INVALID || r->vstat[v] == V_INVALID) continue;
      const double gv = gu + arc->w;
      Node &nv = node[v];
      if (nv.stamp != gen) { nv.stamp = gen; nv.h = heuristic(v); }     // h(v) is fixed for the query: computed when v is first reached
      else if (!(gv < nv.g)) continue;
      nv.g = gv;
      if (nv.h == inf) { nv.closed = 1u; continue; }
      nv.parent = u; nv.parent_edge = e; nv.closed = 0u;
      heap.emplace_back(gv + nv.h, v);
      std::push_heap(heap.begin(), heap.end(), cmp);
    }
  }
  if (!found) return false;
  path.clear(); path_e.clear();
  for (int32_t v = goal;; v = node[v].parent) {
    path.push_back(v);
    if (v == start) break;
    path_e.push_back(node[v].parent_edge);
  }
  return true;
}

// One Dijkstra per landmark over ALL edges on the host threads: lm_d[v * L + l] = (float) graph distance landmark l -> v.
void landmark_distances_host(tr_roadmap *r, int T) {
  const int64_t V = r->V;
  const int L = (int)r->lm_v.size();
  std::atomic<int> next{0};
  auto worker = [&]() {
    std::vector<double> dist((size_t)V);
    std::vector<std::pair<double, int32_t>> heap;
    auto cmp = [](const std::pair<double, int32_t> &a, const std::pair<double, int32_t> &b) { return a.first > b.first; };
    for (;;) {
      const int l = next.fetch_add(1);
      if (l >= L) break;
      std::fill(dist.begin(), dist.end(), std::numeric_limits<double>::infinity());
      heap.clear();
      dist[(size_t)r->lm_v[(size_t)l]] = 0.0;
      heap.emplace_back(0.0, r->lm_v[(size_t)l]);
      while (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), cmp);
        const double du = heap.back().first;
        const int32_t u = heap.back().second;
        heap.pop_back();
        if (du > dist[(size_t)u]) continue;
        for (int64_t k = r->adj_off[u]; k < r->adj_off[u + 1]; k++) {
          const Arc &a = r->adj[(size_t)k];
          const double dv = du + a.w;
          if (dv < dist[(size_t)a.v]) { dist[(size_t)a.v] = dv; heap.emplace_back(dv, a.v); std::push_heap(heap.begin(), heap.end(), cmp); }
        }
      }
      for (int64_t v = 0; v < V; v++) r->lm_d[(size_t)v * L + l] = (float)dist[(size_t)v];
    }
  };
  on_threads(std::max(1, std::min(T, L)), [&](int) { worker(); });
}

// The same distances on the device: every sweep relaxes all arcs for all landmarks at once -- thread (u, l) offers dist[u][l] + w(u, v)
// to every neighbour v (atomicMin on the bit patterns of the non-negative doubles) -- until a sweep changes nothing.  With
// non-negative weights and a monotone rounded addition this fixed point is Dijkstra's result bit for bit: both are the minimum over
// all paths of the left-to-right rounded sums of their weights.  A 100 k-vertex 10-NN roadmap converges in a few dozen sweeps of
// ~30 us; 16 Dijkstras on 16 host threads take 40 - 80 ms.
__global__ __launch_bounds__(256) void landmark_relax(const int64_t *__restrict__ adj_off, const Arc *__restrict__ adj, int64_t V, int L,
                                                      unsigned long long *__restrict__ dist, uint32_t *__restrict__ changed) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= V * L) return;
  const int64_t u = t / L;
  const int l = (int)(t - u * L);
  const double du = __longlong_as_double((long long)dist[u * L + l]);
  if (!(du < 1e300)) return;                                   // not reached yet
  bool any = false;
  for (int64_t k = adj_off[u]; k < adj_off[u + 1]; k++) {
    const Arc a = adj[k];
    const double cand = du + a.w;
    const unsigned long long cb = (unsigned long long)__double_as_longlong(cand);
    unsigned long long *p = &dist[(int64_t)a.v * L + l];
    if (cb < *p) { if (atomicMin(p, cb) > cb) any = true; }
  }
  if (any) *changed = 1u;
}
__global__ __launch_bounds__(256) void landmark_to_float(const unsigned long long *__restrict__ dist, int64_t n, float *__restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < n) out[t] = (float)__longlong_as_double((long long)dist[t]);
}

__global__ __launch_bounds__(256) void landmark_init(unsigned long long *__restrict__ dist, int64_t n, const int32_t *__restrict__ lm_v, int L) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const int64_t u = t / L;
  const int l = (int)(t - u * L);
  dist[t] = (lm_v[l] == (int32_t)u) ? 0ull : 0x7FF0000000000000ull;          // 0 at the landmark itself, +inf elsewhere
}

bool landmark_distances_device(tr_roadmap *r) {
  const int64_t V = r->V;
  const int L = (int)r->lm_v.size();
  if (hipSetDevice(tr_device(r->ctx)) != hipSuccess) return false;
  constexpr int BATCH = 8;                                     // sweeps between two looks at the flags
  // one allocation: offsets | arcs | distances (as ordered bit patterns) | float table | landmark vertices | flags
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t b_off = up((size_t)(V + 1) * sizeof(int64_t)), b_adj = up(std::max<size_t>(1, r->adj.size()) * sizeof(Arc)),
               b_dist = up((size_t)V * L * sizeof(unsigned long long)), b_out = up((size_t)V * L * sizeof(float)),
               b_lm = up((size_t)L * sizeof(int32_t)), b_flags = up(BATCH * sizeof(uint32_t));
  char *arena = nullptr;
  if (dev_cache().alloc(tr_device(r->ctx), (void **)&arena, b_off + b_adj + b_dist + b_out + b_lm + b_flags) != hipSuccess) return false;
  int64_t *d_off = (int64_t *)arena;
  Arc *d_adj = (Arc *)(arena + b_off);
  unsigned long long *d_dist = (unsigned long long *)(arena + b_off + b_adj);
  float *d_out = (float *)(arena + b_off + b_adj + b_dist);
  int32_t *d_lm = (int32_t *)(arena + b_off + b_adj + b_dist + b_out);
  uint32_t *d_changed = (uint32_t *)(arena + b_off + b_adj + b_dist + b_out + b_lm);
  const unsigned grid = (unsigned)((V * L + 255) / 256);
  bool ok = hipMemcpyAsync(d_lm, r->lm_v.data(), (size_t)L * sizeof(int32_t), hipMemcpyHostToDevice, nullptr) == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(landmark_init, dim3(grid), dim3(256), 0, nullptr, d_dist, V * L, d_lm, L);
    ok = hipGetLastError() == hipSuccess;
  }
  ok = ok && hipMemcpyAsync(d_off, r->adj_off.data(), (size_t)(V + 1) * sizeof(int64_t), hipMemcpyHostToDevice, nullptr) == hipSuccess &&
       hipMemcpyAsync(d_adj, r->adj.data(), r->adj.size() * sizeof(Arc), hipMemcpyHostToDevice, nullptr) == hipSuccess;
  bool converged = false;
  for (int64_t sweeps = 0; ok && !converged && sweeps < 4 * V + BATCH; sweeps += BATCH) {     // (V - 1 sweeps always suffice)
    uint32_t flags[BATCH];
    ok = hipMemsetAsync(d_changed, 0, BATCH * sizeof(uint32_t), nullptr) == hipSuccess;
    for (int b = 0; ok && b < BATCH; b++) {
      hipLaunchKernelGGL(landmark_relax, dim3(grid), dim3(256), 0, nullptr, d_off, d_adj, V, L, d_dist, d_changed + b);
      ok = hipGetLastError() == hipSuccess;
    }
    ok = ok && hipMemcpy(flags, d_changed, sizeof(flags), hipMemcpyDeviceToHost) == hipSuccess;
    for (int b = 0; ok && b < BATCH; b++) if (!flags[b]) converged = true;            // a sweep without a change: the fixed point
  }
  if (ok && converged) {
    hipLaunchKernelGGL(landmark_to_float, dim3(grid), dim3(256), 0, nullptr, d_dist, V * L, d_out);
    ok = hipGetLastError() == hipSuccess &&
         hipMemcpy(r->lm_d.data(), d_out, (size_t)V * L * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess;
  }
  dev_cache().release(arena);
  return ok && converged;
}

// Landmark tables: n extremal vertices of the largest component (the corners of the sampled state box first, then fixed
// pseudo-random directions), one Dijkstra each over ALL edges -- validity plays no part, see the header comment.
void build_landmarks(tr_roadmap *r, int n, int T) {
  r->lm_d.clear(); r->lm_v.clear(); r->lm_n = 0; r->lm_mismatch = false;
  r->ds.lm_current = false;
  const int64_t V = r->V;
  const int S = r->S;
  if (n <= 0 || V < 2 || r->E == 0) return;
  Laps laps("build_landmarks");
  // largest connected component: union-find over the edge list (sequential reads; the parent array stays in cache -- a graph
  // traversal would take one cache miss per vertex into the 16-byte arcs)
  std::vector<int32_t> comp((size_t)V);
  int32_t big = -1;
  int64_t big_n = 0;
  {
    std::vector<int32_t> &par = comp;
    for (int64_t v = 0; v < V; v++) par[(size_t)v] = (int32_t)v;
    auto root = [&par](int32_t x) {
      while (par[(size_t)x] != x) { par[(size_t)x] = par[(size_t)par[(size_t)x]]; x = par[(size_t)x]; }     // path halving
      return x;
    };
    for (int64_t e = 0; e < r->E; e++) {
      const int32_t a = root(r->eu[(size_t)e]), b = root(r->ev[(size_t)e]);
      if (a != b) par[(size_t)std::max(a, b)] = std::min(a, b);          // the smaller index becomes the root
    }
    std::vector<int32_t> cnt((size_t)V, 0);
    for (int64_t v = 0; v < V; v++) { const int32_t c = root((int32_t)v); par[(size_t)v] = c; cnt[(size_t)c]++; }   // comp[v] = its root
    for (int64_t v = 0; v < V; v++) if (cnt[(size_t)v] > big_n) { big_n = cnt[(size_t)v]; big = (int32_t)v; }       // ties: the smallest root
  }
  if (big_n < 2) return;
  laps.lap("components");
  std::vector<double> lo((size_t)S, std::numeric_limits<double>::infinity()), hi((size_t)S, -std::numeric_limits<double>::infinity());
  for (int64_t v = 0; v < V; v++)
    for (int i = 0; i < S; i++) { const double x = r->states[(size_t)v * S + i]; lo[(size_t)i] = std::min(lo[(size_t)i], x); hi[(size_t)i] = std::max(hi[(size_t)i], x); }
  // the directions, in their fixed order; the vertex furthest along each of the first n on the host threads, any further one
  // (needed only when two directions pick the same vertex) when its turn comes
  std::vector<double> dirs((size_t)4 * n * S);
  {
    uint64_t lcg = 0x9E3779B97F4A7C15ull;
    for (int l = 0; l < 4 * n; l++)
      for (int i = 0; i < S; i++) {
        double c;
        if (S <= 16 && l < (1 << S)) c = ((l >> i) & 1) ? 1.0 : -1.0;
        else { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; c = (double)(int64_t)(lcg >> 11) / (double)(1ll << 52) - 1.0; }
        const double ext = hi[(size_t)i] - lo[(size_t)i];
        dirs[(size_t)l * S + i] = ext > 0 ? c / ext : 0.0;
      }
  }
  auto furthest = [&](int l) {
    const double *dir = &dirs[(size_t)l * S];
    int32_t arg = -1;
    double best = -std::numeric_limits<double>::infinity();
    for (int64_t v = 0; v < V; v++) {
      if (comp[(size_t)v] != big) continue;
      double d = 0;
      for (int i = 0; i < S; i++) d += dir[i] * r->states[(size_t)v * S + i];
      if (d > best) { best = d; arg = (int32_t)v; }
    }
    return arg;
  };
  std::vector<int32_t> first((size_t)n, -1);
  {
    const int Tn = std::max(1, std::min(T, n));
    on_threads(Tn, [&](int t) { for (int l = t; l < n; l += Tn) first[(size_t)l] = furthest(l); });
  }
  for (int l = 0; l < 4 * n && (int)r->lm_v.size() < n; l++) {
    const int32_t arg = l < n ? first[(size_t)l] : furthest(l);
    if (arg >= 0 && std::find(r->lm_v.begin(), r->lm_v.end(), arg) == r->lm_v.end()) r->lm_v.push_back(arg);
  }
  const int L = (int)r->lm_v.size();
  if (L == 0) return;
  laps.lap("extremal vertices");
  r->lm_d.assign((size_t)V * L, std::numeric_limits<float>::infinity());
  // the distances: on the device (landmark_distances_device), or L Dijkstras on the host threads (TENDON_HIP_LANDMARKS=host, or when
  // the device path fails); TENDON_HIP_LANDMARKS=check builds both and keeps the host's if they differ in any bit
  const char *mode = std::getenv("TENDON_HIP_LANDMARKS");
  const bool host_only = mode && std::strcmp(mode, "host") == 0, check = mode && std::strcmp(mode, "check") == 0;
  bool done = false;
  if (!host_only) done = landmark_distances_device(r);
  laps.lap("distances");
  if (!done || check) {
    std::vector<float> dev;
    if (done) dev = r->lm_d;
    landmark_distances_host(r, T);
    if (done && check && std::memcmp(dev.data(), r->lm_d.data(), dev.size() * sizeof(float)) != 0) r->lm_mismatch = true;
  }
  for (float &x_ : r->lm_d) if (!(x_ < trk::SR_LM_FAR)) x_ = trk::SR_LM_FAR;
  r->lm_n = L;
}

void free_dev(tr_roadmap *r) {
  void *p[] = {r->d_ids, r->d_masks, r->d_off, r->d_list, r->d_hit, r->d_bits};
  for (void *q : p) if (q) dev_cache().release(q);
  r->d_ids = nullptr; r->d_masks = nullptr; r->d_off = nullptr; r->d_list = nullptr; r->d_hit = nullptr; r->d_bits = nullptr;
  if (r->h_bits) { (void)hipHostFree(r->h_bits); r->h_bits = nullptr; }
  r->list_cap = 0; r->has_caches = false;
}

void free_comp(tr_roadmap *r) {
  if (r->dc.arena) dev_cache().release(r->dc.arena);
  r->dc = tr_roadmap::DevComp{};
}

// The searches' tables and per-round arrays -- all of the search state that does not depend on the roadmap -- go back to the buffer
// cache; the adjacency rows stay, and the next large round allocates tables again (search_tables).  Caller holds r->mu.
size_t release_search_tables(tr_roadmap *r) {
  auto &d = r->ds;
  if (d.in_flight) return 0;
  size_t b = 0;
  if (d.tables) {
    dev_cache().release(d.tables);
    b += d.table_bytes;
    d.tables = nullptr; d.table_bytes = 0; d.slots = 0; d.gens_issued = 0;
    for (int c = 0; c < trk::SR_CLASSES; c++) { d.pool[c] = nullptr; d.pool_n[c] = 0; }
  }
  if (d.qarena) {
    dev_cache().release(d.qarena);
    b += (size_t)d.nq_cap * 17 + (size_t)d.pbuf_cap * 4;
    d.qarena = nullptr; d.nq_cap = 0; d.pbuf_cap = 0;
  }
  return b;
}

void free_search(tr_roadmap *r) {
  auto &d = r->ds;
  if (d.arena) dev_cache().release(d.arena);
  if (d.tables) dev_cache().release(d.tables);
  if (d.qarena) dev_cache().release(d.qarena);
  if (d.h_handback) (void)hipHostFree(d.h_handback);
  if (d.sweep_arena) dev_cache().release(d.sweep_arena);
  if (d.sweep_h_dist) (void)hipHostFree(d.sweep_h_dist);
  if (d.sweep_stream) (void)hipStreamDestroy(d.sweep_stream);
  for (hipEvent_t e : d.ev) if (e) (void)hipEventDestroy(e);
  d.~DevSearch();
  new (&d) tr_roadmap::DevSearch();
}

// ---- connected components of the roadmap minus the items known invalid ----
// The reference gives up on a query whose start and goal lie in different components before it searches (solutionComponent /
// sameComponent, VoxelCachedLazyPRM.cpp:2015-2044; LazyPRM renumbers the components when it removes items).  Without that a search
// for an unreachable goal walks the start's whole component before it reports "no path": 10^5 expansions, 37 ms on a core, 250 ms at
// a wave's pace.  Here the labels are recomputed per round on the device: union-find over the edge list with atomic hooks of the
// larger root under the smaller, then one pass that points every vertex at its root.  ~0.1 ms of kernels + the validity bytes up and the labels down.
// (Plain loads and stores except for the hooks: another XCD's L2 may show an older parent, which is an ancestor all the same; a
// vertex passed on the way is pointed at its grandparent -- path halving.  Only the compare-and-swap that turns a root into a
// child has to see the truth, and it does: it is an agent-scope atomic, and its return value is where a failed attempt goes on.)
__device__ __forceinline__ int32_t cc_root(int32_t *parent, int32_t x) {
  int32_t p = parent[x];
  for (int guard = 0; p != x && guard < (1 << 24); guard++) {
    const int32_t gp = parent[p];
    if (gp != p) parent[x] = gp;
    x = p; p = gp;
  }
  return x;
}
__global__ __launch_bounds__(256) void cc_init(int32_t *__restrict__ parent, int64_t V) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v < V) parent[v] = (int32_t)v;
}
// first pass: every vertex under its smallest smaller neighbour (one atomicMin per edge on mostly distinct words) -- a forest of
// short trees whose roots are the local minima, so that the hooks below contend for many words instead of one
__global__ __launch_bounds__(256) void cc_seed(const int32_t *__restrict__ eu, const int32_t *__restrict__ ev, const uint8_t *__restrict__ estat,
                                               const uint8_t *__restrict__ vstat, int64_t E, int32_t *parent) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E || estat[e] == V_INVALID) return;
  const int32_t a = eu[e], b = ev[e];
  if (a == b || vstat[a] == V_INVALID || vstat[b] == V_INVALID) return;
  atomicMin(&parent[a > b ? a : b], a > b ? b : a);
}
__global__ __launch_bounds__(256) void cc_hook(const int32_t *__restrict__ eu, const int32_t *__restrict__ ev, const uint8_t *__restrict__ estat,
                                               const uint8_t *__restrict__ vstat, int64_t E, int32_t *parent) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E || estat[e] == V_INVALID) return;
  const int32_t a = eu[e], b = ev[e];
  if (vstat[a] == V_INVALID || vstat[b] == V_INVALID) return;
  int32_t ra = cc_root(parent, a), rb = cc_root(parent, b);
  for (int guard = 0; ra != rb && guard < (1 << 24); guard++) {
    if (ra < rb) { const int32_t t = ra; ra = rb; rb = t; }                 // the larger root goes under the smaller
    const int32_t old = atomicCAS(&parent[ra], ra, rb);
    if (old == ra) break;                                                   // hooked
    ra = cc_root(parent, old);                                              // someone else hooked it first: follow and try again
    rb = cc_root(parent, rb);
  }
}
// (reads only: a halving store of one thread here could replace the root another thread has just written for the same vertex by a
// mere ancestor -- seen as connected pairs with different labels)
__global__ __launch_bounds__(256) void cc_flatten(const int32_t *__restrict__ parent, int32_t *__restrict__ label, int64_t V) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= V) return;
  int32_t x = (int32_t)v, p = parent[x];
  for (int guard = 0; p != x && guard < (1 << 24); guard++) { x = p; p = parent[x]; }
  label[v] = x;
}

// The same labels by union-find on one host thread (a few ms at 6 x 10^5 edges): for the searches the kernel hands back WHILE it runs,
// when the device cannot be asked (its stream is busy with the searches).  Only equality of two labels is ever used.
void host_component_labels(tr_roadmap *r) {
  const int64_t V = r->V, E = r->E;
  std::vector<int32_t> &parent = r->dc.label;
  parent.resize((size_t)V);
  for (int64_t v = 0; v < V; v++) parent[(size_t)v] = (int32_t)v;
  auto find = [&](int32_t x) { while (parent[(size_t)x] != x) { parent[(size_t)x] = parent[(size_t)parent[(size_t)x]]; x = parent[(size_t)x]; } return x; };
  for (int64_t e = 0; e < E; e++) {
    if (r->estat[(size_t)e] == V_INVALID) continue;
    const int32_t a = r->eu[(size_t)e], b = r->ev[(size_t)e];
    if (r->vstat[(size_t)a] == V_INVALID || r->vstat[(size_t)b] == V_INVALID) continue;
    const int32_t ra = find(a), rb = find(b);
    if (ra != rb) parent[(size_t)std::max(ra, rb)] = std::min(ra, rb);
  }
  for (int64_t v = 0; v < V; v++) parent[(size_t)v] = find((int32_t)v);
}

// r->dc.label[v] = the smallest vertex of v's component in the graph minus the items known invalid.  false: not available (no
// edges, out of memory): the caller searches as before.
bool component_labels(tr_roadmap *r) {
  auto &c = r->dc;
  c.status_current = false;
  if (c.state < 0 || r->E == 0 || r->V < 2) return false;
  const int dev = tr_device(r->ctx);
  const int64_t V = r->V, E = r->E;
  if (hipSetDevice(dev) != hipSuccess) return false;
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  if (c.state == 0) {
    c.state = -1;
    const size_t b_e = up((size_t)E * 4), b_v = up((size_t)V * 4), b_vs = up((size_t)V), b_es = up((size_t)E);
    if (dev_cache().alloc(dev, (void **)&c.arena, 2 * b_e + 2 * b_v + b_vs + b_es) != hipSuccess) return false;
    char *p = c.arena;
    c.d_eu = (int32_t *)p; p += b_e;
    c.d_ev = (int32_t *)p; p += b_e;
    c.d_parent = (int32_t *)p; p += b_v;
    c.d_label = (int32_t *)p; p += b_v;
    c.d_vstat = (uint8_t *)p; p += b_vs;
    c.d_estat = (uint8_t *)p;
    if (hipMemcpyAsync(c.d_eu, r->eu.data(), (size_t)E * 4, hipMemcpyHostToDevice, nullptr) != hipSuccess ||
        hipMemcpyAsync(c.d_ev, r->ev.data(), (size_t)E * 4, hipMemcpyHostToDevice, nullptr) != hipSuccess) { free_comp(r); r->dc.state = -1; return false; }
    c.label.resize((size_t)V);
    c.state = 1;
  }
  bool ok = hipMemcpyAsync(c.d_vstat, r->vstat.data(), (size_t)V, hipMemcpyHostToDevice, nullptr) == hipSuccess &&
            hipMemcpyAsync(c.d_estat, r->estat.data(), (size_t)E, hipMemcpyHostToDevice, nullptr) == hipSuccess;
  if (!ok) return false;
  const bool timing = std::getenv("TENDON_HIP_SEARCH_STATS") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  if (timing) (void)hipStreamSynchronize(nullptr);
  const auto t1 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(cc_init, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, nullptr, c.d_parent, V);
  hipLaunchKernelGGL(cc_seed, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, nullptr, c.d_eu, c.d_ev, c.d_estat, c.d_vstat, E, c.d_parent);
  hipLaunchKernelGGL(cc_hook, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, nullptr, c.d_eu, c.d_ev, c.d_estat, c.d_vstat, E, c.d_parent);
  hipLaunchKernelGGL(cc_flatten, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, nullptr, c.d_parent, c.d_label, V);
  if (hipGetLastError() != hipSuccess) return false;
  if (timing) (void)hipStreamSynchronize(nullptr);
  const auto t2 = std::chrono::steady_clock::now();
  if (hipMemcpy(c.label.data(), c.d_label, (size_t)V * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
  if (timing) {
    const auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::fprintf(stderr, "[tendon_hip] component labels: validity bytes up %.3f ms, kernels %.3f ms, labels down %.3f ms\n", ms(t0, t1), ms(t1, t2),
                 ms(t2, std::chrono::steady_clock::now()));
  }
  c.status_current = true;
  return true;
}

// ---- the graph searches on the device (search_kernel.hpp) ----
static_assert(sizeof(trk::SArc) == sizeof(Arc) && sizeof(trk::SRec) == 32, "the device's arcs are the host's; two records to a 64-byte line");

// 0 = the host threads, 1 = the device for rounds of at least kSearchMinQueries queries, 2 = the device always (tests)
constexpr int64_t kSearchMinQueries = 512;
constexpr int64_t kComponentMinQueries = 64;                     // rounds smaller than this are searched without component labels
constexpr int64_t kComponentTrigger = 2000;                      // expansions of a search that ends without a path, from which on labels pay
// 0: never; 1: when they have paid before on this roadmap (a search walked kComponentTrigger vertices and found no path) or the kernel
// hands searches back (default); 2: every round of kComponentMinQueries or more (TENDON_HIP_COMPONENTS=0 | unset | 1)
int components_mode() {
  const char *e = std::getenv("TENDON_HIP_COMPONENTS");
  if (!e) return 1;
  if (std::strcmp(e, "0") == 0 || std::strcmp(e, "off") == 0) return 0;
  return 2;
}
int search_mode() {
  const char *e = std::getenv("TENDON_HIP_SEARCH");
  if (!e) return 1;
  if (std::strcmp(e, "host") == 0 || std::strcmp(e, "0") == 0) return 0;
  if (std::strcmp(e, "device") == 0 || std::strcmp(e, "force") == 0) return 2;
  return 1;
}

// the share of a round's searches (the ones expected to be longest) that the host threads take while the kernel runs, and the
// kernel's pop budget per search; TENDON_HIP_SEARCH_HOST_SHARE (per cent) / TENDON_HIP_SEARCH_BUDGET override
double search_host_share() {
  const char *e = std::getenv("TENDON_HIP_SEARCH_HOST_SHARE");
  const double p = e ? std::atof(e) : 1.0;
  return std::min(100.0, std::max(0.0, p)) / 100.0;
}
int64_t search_budget() {
  const char *e = std::getenv("TENDON_HIP_SEARCH_BUDGET");
  const long long b = e ? std::atoll(e) : 6500;
  return b > 0 ? (int64_t)b : 0;
}

// vertices a step of the kernel takes off a search's open list (1: the host's order of expansions exactly); TENDON_HIP_SEARCH_K overrides
int search_kbest() {
  const char *e = std::getenv("TENDON_HIP_SEARCH_K");
  const int k = e ? std::atoi(e) : trk::SR_K;
  return std::max(1, std::min(k, (int)trk::SR_K));
}

// The kernel for the roadmap's state size (the heuristic keeps SX coordinates in registers)
using SearchKernel = void (*)(trk::SearchArgs);
SearchKernel search_kernel_for(int S) { return S <= 4 ? trk::roadmap_astar<4> : S <= 8 ? trk::roadmap_astar<8> : trk::roadmap_astar<trk::SR_MAXS>; }

// The resident part: adjacency rows, states, landmark table, validity bytes -- what depends on the roadmap -- and the searches' own state,
// which does not: per wave slot a table of 2^lc0 records with its far list (44 B per record: 176 KiB at lc0 = 12), and a pool of larger
// tables (x 4 per class) that long searches move into.  The slot count is what the chip holds of this kernel (LDS: 9.8 KiB per wave).
//   TENDON_HIP_SEARCH_SLOTS=n     searches in flight (default: what the device holds)
//   TENDON_HIP_SEARCH_LC0=8..14   log2 of a slot's own table (default 12; tests: a small value makes every search grow)
//   TENDON_HIP_SEARCH_POOL=a,b,c  tables of the three larger classes (default slots, slots / 4, slots / 64 -- 5.3 GB with the slots' own at
//                                 3 072 slots: a 6 x 10^5-vertex roadmap's searches touch 10^4 - 10^5 vertices each; 0,0,0: every search that outgrows
//                                 its table is handed back to the host threads)
// pool tables per class for `slots` searches in flight (class 0: the slots' own)
void search_pool_counts(int64_t slots, int64_t pn[trk::SR_CLASSES]) {
  pn[0] = 0; pn[1] = std::max<int64_t>(64, slots); pn[2] = std::max<int64_t>(16, slots / 4); pn[3] = std::max<int64_t>(8, slots / 64);
  if (const char *e = std::getenv("TENDON_HIP_SEARCH_POOL")) {
    long long x1 = 0, x2 = 0, x3 = 0;
    if (std::sscanf(e, "%lld,%lld,%lld", &x1, &x2, &x3) >= 1) {
      int64_t most[trk::SR_CLASSES];
      for (int c = 0; c < trk::SR_CLASSES; c++) most[c] = std::max<int64_t>(pn[c], (int64_t)1 << 16);
      pn[1] = std::min<int64_t>(most[1], std::max(0ll, x1)); pn[2] = std::min<int64_t>(most[2], std::max(0ll, x2)); pn[3] = std::min<int64_t>(most[3], std::max(0ll, x3));
    }
  }
}
bool search_setup(tr_roadmap *r) {
  auto &d = r->ds;
  if (d.state != 0) return d.state > 0;
  d.state = -1;
  Laps laps("search_setup");
  const int64_t V = r->V;
  if (r->S > trk::SR_MAXS) { d.why = "state size above the kernel's"; return false; }
  if (V < 2 || r->adj.size() == 0) { d.why = "no graph"; return false; }
  if (V >= ((int64_t)1 << trk::SR_VBITS)) { d.why = "more vertices than an open-list word names"; return false; }
  // two arcs between the same pair of vertices would make two lanes relax the same record in one step: such roadmaps stay on the host
  {
    std::vector<int32_t> nb;
    for (int64_t v = 0; v < V; v++) {
      const int64_t a0 = r->adj_off[(size_t)v], a1 = r->adj_off[(size_t)v + 1];
      if (a1 - a0 < 2) continue;
      nb.clear();
      for (int64_t k = a0; k < a1; k++) nb.push_back(r->adj[(size_t)k].v);
      std::sort(nb.begin(), nb.end());
      if (std::adjacent_find(nb.begin(), nb.end()) != nb.end()) { d.why = "parallel edges"; return false; }
    }
  }
  laps.lap("parallel-edge check");
  // adjacency at a fixed stride: row v holds v's arcs (at most SR_D; unused slots marked); a vertex with more keeps SR_D - 1 in a
  // row whose last slot names its next row (rows V, V + 1, ... in vertex order)
  constexpr int D = trk::SR_D;
  int64_t n_rows = V;
  for (int64_t v = 0; v < V; v++) {
    int64_t deg = r->adj_off[(size_t)v + 1] - r->adj_off[(size_t)v];
    while (deg > D) { deg -= D - 1; n_rows++; }
  }
  if (n_rows > std::numeric_limits<int32_t>::max() / D) { d.why = "roadmap too large for the row index"; return false; }
  RawArray<trk::SArc> rows;
  rows.resize_uninit((size_t)n_rows * D);
  std::vector<uint8_t> lanes((size_t)V);                          // lanes a vertex's first row needs (an open-list word carries it)
  {
    int64_t next_row = V;
    for (int64_t v = 0; v < V; v++) {
      const Arc *arc = r->adj.data() + r->adj_off[(size_t)v];
      int64_t deg = r->adj_off[(size_t)v + 1] - r->adj_off[(size_t)v], row = v;
      lanes[(size_t)v] = (uint8_t)std::max<int64_t>(1, std::min<int64_t>(deg, D));
      for (;;) {
        trk::SArc *out = rows.data() + (size_t)row * D;
        const int take = deg > D ? D - 1 : (int)deg;
        for (int j = 0; j < take; j++) out[j] = trk::SArc{arc[j].v, arc[j].e, arc[j].w};        // (the neighbours' lane counts: second pass below)
        for (int j = take; j < D; j++) out[j] = trk::SArc{trk::SR_ARC_NONE, -1, 0.0};
        arc += take; deg -= take;
        if (deg == 0) break;
        out[D - 1] = trk::SArc{trk::SR_ARC_MORE, (int32_t)next_row, 0.0};
        row = next_row++;
      }
    }
  }
  for (size_t t = 0; t < (size_t)n_rows * D; t++) {                 // an arc's vertex word carries the lanes its neighbour's own row needs
    trk::SArc &x = rows[t];
    if (x.v >= 0) x.v |= (int32_t)lanes[(size_t)x.v] << trk::SR_VBITS;
  }
  laps.lap("adjacency rows");
  const int dev = tr_device(r->ctx);
  if (hipSetDevice(dev) != hipSuccess) { d.why = "hipSetDevice"; return false; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { d.why = "hipGetDeviceProperties"; return false; }
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, search_kernel_for(r->S), 64, trk::search_lds_bytes()) != hipSuccess || per_cu < 1) {
    d.why = "occupancy query"; return false;
  }
  int64_t slots = (int64_t)per_cu * prop.multiProcessorCount;
  if (const char *e = std::getenv("TENDON_HIP_SEARCH_SLOTS")) slots = std::max<int64_t>(1, std::min<int64_t>(slots, std::atoll(e)));
  d.max_slots = slots;
  d.lc0 = 12;
  if (const char *e = std::getenv("TENDON_HIP_SEARCH_LC0")) d.lc0 = std::max(8, std::min(14, std::atoi(e)));
  {
    // control words: the counters, then the pool's claim bitmaps at their largest (search_tables lays them out per table set)
    int64_t pn[trk::SR_CLASSES];
    search_pool_counts(slots, pn);
    int64_t word = trk::SR_CTL_WORDS;
    for (int c = 0; c < trk::SR_CLASSES; c++) word += (pn[c] + 31) / 32;
    d.ctl_bytes = ((size_t)word * 4 + 255) & ~(size_t)255;
  }
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const int Lmax = trk::SR_MAXL;
  const size_t b_rows = up((size_t)n_rows * D * sizeof(trk::SArc)), b_vr = up((size_t)V * trk::search_row_bytes(r->S, Lmax)),
               b_vs = up((size_t)V), b_es = up((size_t)std::max<int64_t>(r->E, 1));
  if (dev_cache().alloc(dev, (void **)&d.arena, b_rows + b_vr + 2 * b_vs + b_es + d.ctl_bytes) != hipSuccess) {
    d.why = "out of device memory"; return false;
  }
  char *p = d.arena;
  d.d_rows = (trk::SArc *)p; p += b_rows;
  d.d_vrows = p; p += b_vr;
  d.d_vstat = (uint8_t *)p; p += b_vs;
  d.d_estat = (uint8_t *)p; p += b_es;
  d.d_deg = (uint8_t *)p; p += b_vs;
  d.d_ctl = (uint32_t *)p;
  laps.lap("device properties + arena");
  const bool ok = hipMemcpyAsync(d.d_rows, rows.data(), (size_t)n_rows * D * sizeof(trk::SArc), hipMemcpyHostToDevice, nullptr) == hipSuccess &&
                  hipMemcpyAsync(d.d_deg, lanes.data(), (size_t)V, hipMemcpyHostToDevice, nullptr) == hipSuccess &&
                  hipStreamSynchronize(nullptr) == hipSuccess;
  laps.lap("graph uploaded");
  if (!ok) { free_search(r); r->ds.state = -1; r->ds.why = "out of device memory"; return false; }
  if (std::getenv("TENDON_HIP_SEARCH_STATS"))
    std::fprintf(stderr, "[tendon_hip] search graph: %lld rows of %d arcs (%lld continued), %.1f MiB\n", (long long)n_rows, D, (long long)(n_rows - V),
                 (double)(b_rows + b_vr + 2 * b_vs + b_es) / 1048576.0);
  d.lm_current = false;
  d.state = 1;
  return true;
}

// The tables of the searches in flight, sized by the round: `want` queries need min(want, max_slots) slots (in steps of 256, and at
// least twice what a smaller round left, so a caller whose rounds grow re-allocates a handful of times) and a pool in proportion.
// A 512-query round on a fresh roadmap holds ~0.9 GB, a 10 000-query round the device's full 3 072 slots (~5.3 GB); the state stays
// with the roadmap until tr_roadmap_release_search_state, the out-of-memory trim (release_idle_search_tables) or tr_roadmap_destroy.
bool search_tables(tr_roadmap *r, int64_t want) {
  auto &d = r->ds;
  int64_t slots = std::min<int64_t>(d.max_slots, std::max<int64_t>(256, (want + 255) & ~(int64_t)255));
  if (d.tables && d.slots >= slots) return true;
  if (d.tables) slots = std::max(slots, std::min<int64_t>(d.max_slots, 2 * d.slots));
  Laps laps("search_tables");
  const int dev = tr_device(r->ctx);
  if (d.tables) { dev_cache().release(d.tables); d.tables = nullptr; d.slots = 0; }
  int64_t pn[trk::SR_CLASSES];
  search_pool_counts(slots, pn);
  // within a third of what is free: the pool shrinks first, then the slots
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { d.why = "hipMemGetInfo"; return false; }
  auto tables_bytes = [&](int64_t s_) {
    size_t b = (size_t)s_ * trk::search_chunk_bytes(d.lc0);
    for (int c = 1; c < trk::SR_CLASSES; c++) b += (size_t)pn[c] * trk::search_chunk_bytes(d.lc0 + 2 * c);
    return b;
  };
  for (int c = trk::SR_CLASSES - 1; c >= 1; c--)
    while (pn[c] > 0 && tables_bytes(slots) > free_b / 3) pn[c] /= 2;
  while (slots > 64 && tables_bytes(slots) > free_b / 3) slots /= 2;
  if (tables_bytes(slots) > free_b / 3) { d.why = "out of device memory"; return false; }
  int64_t word = trk::SR_CTL_WORDS;
  for (int c = 0; c < trk::SR_CLASSES; c++) { d.pool_n[c] = (int32_t)pn[c]; d.pool_word[c] = (int32_t)word; word += (pn[c] + 31) / 32; }
  if ((size_t)word * 4 > d.ctl_bytes) { d.why = "pool larger than the control words"; return false; }
  d.table_bytes = tables_bytes(slots);
  if (dev_cache().alloc(dev, (void **)&d.tables, d.table_bytes) != hipSuccess) { d.tables = nullptr; d.table_bytes = 0; d.why = "out of device memory"; return false; }
  char *q = d.tables + (size_t)slots * trk::search_chunk_bytes(d.lc0);
  for (int c = 1; c < trk::SR_CLASSES; c++) { d.pool[c] = q; q += (size_t)pn[c] * trk::search_chunk_bytes(d.lc0 + 2 * c); }
  laps.lap("tables allocated");
  // (generation 0 is nobody's: cleared once, never again until the generation counter would wrap)
  if (hipMemsetAsync(d.tables, 0, d.table_bytes, nullptr) != hipSuccess) {
    dev_cache().release(d.tables); d.tables = nullptr; d.table_bytes = 0; d.why = "hipMemsetAsync"; return false;
  }
  d.gens_issued = 0;
  d.slots = slots;
  laps.lap("tables cleared");
  if (std::getenv("TENDON_HIP_SEARCH_STATS"))
    std::fprintf(stderr, "[tendon_hip] search state: %lld slots x %zu KiB + pool %d / %d / %d tables = %.1f MiB (whatever the roadmap's size)\n",
                 (long long)slots, trk::search_chunk_bytes(d.lc0) >> 10, d.pool_n[1], d.pool_n[2], d.pool_n[3], (double)d.table_bytes / 1048576.0);
  return true;
}

// ---- a search as a parallel sweep ----
// A query whose path detours far round new obstacles makes A* expand a large part of the graph (on a 6 x 10^5-vertex roadmap: two of
// 10 000 queries with 3 - 4 x 10^5 expansions each, 200 ms on a core, ten times that on a wave -- they WERE the round).  Such a search
// is answered by relaxing every reached vertex's valid arcs at once, sweep after sweep (the scheme of landmark_relax: atomicMin on the
// bit patterns of the non-negative distances), until a sweep changes nothing; vertices at or beyond the goal's distance are not
// relaxed (weights are non-negative: nothing through them improves the goal).  The fixed point is Dijkstra's -- and A*'s -- cost bit for
// bit: the minimum over all paths of the left-to-right rounded sums of their weights.  The path is walked back on the host from the
// goal along arcs with dist[u] + w == dist[v] exactly (the first such arc of a row: A* keeps the first parent that reaches the
// final cost, so the two can differ only where two routes tie to the last bit).
__global__ __launch_bounds__(256) void sweep_relax(const trk::SArc *__restrict__ rows, int D, const uint8_t *__restrict__ vstat,
                                                   const uint8_t *__restrict__ estat, int64_t V, int32_t goal,
                                                   unsigned long long *__restrict__ dist, uint32_t *__restrict__ changed) {
  const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (u >= V) return;
  const unsigned long long bu = dist[u], bg = dist[goal];
  if (!(bu < bg)) return;                                      // not reached yet (+inf), or no closer than the goal is already
  const double du = __longlong_as_double((long long)bu);
  bool any = false;
  int64_t row = u;
  for (int guard = 0; guard < 4096; guard++) {                 // (a vertex's rows: 16 arcs each, chained through the last slot)
    int64_t next = -1;
    for (int j = 0; j < D; j++) {
      const trk::SArc a = rows[row * D + j];
      if (a.v == trk::SR_ARC_NONE) continue;
      if (a.v == trk::SR_ARC_MORE) { next = a.e; break; }
      const int32_t v = a.v & (int32_t)((1u << trk::SR_VBITS) - 1u);
      if (estat[a.e] == V_INVALID || vstat[v] == V_INVALID) continue;
      const unsigned long long cb = (unsigned long long)__double_as_longlong(du + a.w);
      if (cb < dist[v]) { if (atomicMin(&dist[v], cb) > cb) any = true; }
    }
    if (next < 0) break;
    row = next;
  }
  if (any) *changed = 1u;
}
__global__ __launch_bounds__(256) void sweep_init(unsigned long long *__restrict__ dist, int64_t V, int32_t start) {
  const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (u < V) dist[u] = u == start ? 0ull : 0x7FF0000000000000ull;
}

// 1: path found (path: goal ... start, path_e: the edges between them, as astar leaves them), 0: no path, -1: not available (the caller
// searches on).  Serialised per roadmap (one arena); runs on a stream of its own beside the searches' kernel.
int sweep_search(tr_roadmap *r, int32_t start, int32_t goal, std::vector<int32_t> &path, std::vector<int32_t> &path_e) {
  auto &d = r->ds;
  std::lock_guard<std::mutex> lk(d.sweep_mu);
  if (d.state != 1 || !d.d_rows) return -1;
  const int64_t V = r->V, E = r->E;
  if (hipSetDevice(tr_device(r->ctx)) != hipSuccess) return -1;
  constexpr int BATCH = 8;
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t b_dist = up((size_t)V * 8), b_vs = up((size_t)V), b_es = up((size_t)std::max<int64_t>(E, 1));
  if (!d.sweep_stream && hipStreamCreateWithFlags(&d.sweep_stream, hipStreamNonBlocking) != hipSuccess) { d.sweep_stream = nullptr; return -1; }
  if (!d.sweep_arena && dev_cache().alloc(tr_device(r->ctx), (void **)&d.sweep_arena, b_dist + b_vs + b_es + 256) != hipSuccess) { d.sweep_arena = nullptr; return -1; }
  if (!d.sweep_h_dist && hipHostMalloc((void **)&d.sweep_h_dist, (size_t)V * 8, hipHostMallocDefault) != hipSuccess) { d.sweep_h_dist = nullptr; return -1; }
  unsigned long long *d_dist = (unsigned long long *)d.sweep_arena;
  uint8_t *d_vs = (uint8_t *)(d.sweep_arena + b_dist), *d_es = d_vs + b_vs;
  uint32_t *d_changed = (uint32_t *)(d.sweep_arena + b_dist + b_vs + b_es);
  hipStream_t st = d.sweep_stream;
  bool ok = true;
  if (d.sweep_round != r->st_rounds) {                          // the round's validity bytes (they do not change inside a round)
    ok = hipMemcpyAsync(d_vs, r->vstat.data(), (size_t)V, hipMemcpyHostToDevice, st) == hipSuccess &&
         (E == 0 || hipMemcpyAsync(d_es, r->estat.data(), (size_t)E, hipMemcpyHostToDevice, st) == hipSuccess);
    if (ok) d.sweep_round = r->st_rounds;
  }
  const unsigned grid = (unsigned)((V + 255) / 256);
  if (ok) { hipLaunchKernelGGL(sweep_init, dim3(grid), dim3(256), 0, st, d_dist, V, start); ok = hipGetLastError() == hipSuccess; }
  bool converged = false;
  for (int64_t sweeps = 0; ok && !converged && sweeps < V + BATCH; sweeps += BATCH) {
    uint32_t flags[BATCH];
    ok = hipMemsetAsync(d_changed, 0, BATCH * sizeof(uint32_t), st) == hipSuccess;
    for (int b = 0; ok && b < BATCH; b++) {
      hipLaunchKernelGGL(sweep_relax, dim3(grid), dim3(256), 0, st, (const trk::SArc *)d.d_rows, (int)trk::SR_D, (const uint8_t *)d_vs, (const uint8_t *)d_es, V, goal,
                         d_dist, d_changed + b);
      ok = hipGetLastError() == hipSuccess;
    }
    ok = ok && hipMemcpyAsync(flags, d_changed, sizeof(flags), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    for (int b = 0; ok && b < BATCH; b++) if (!flags[b]) converged = true;
  }
  ok = ok && converged && hipMemcpyAsync(d.sweep_h_dist, d_dist, (size_t)V * 8, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
  if (!ok) { d.sweep_round = -1; return -1; }
  const unsigned long long *dist = d.sweep_h_dist;
  const unsigned long long inf_b = 0x7FF0000000000000ull;
  if (dist[goal] >= inf_b) return 0;
  path.clear(); path_e.clear();
  int32_t v = goal;
  for (int64_t guard = 0; guard <= V; guard++) {
    path.push_back(v);
    if (v == start) { d.st_sweeps++; return 1; }
    const double dv = __builtin_bit_cast(double, dist[v]);
    const Arc *arc = r->adj.data() + r->adj_off[(size_t)v], *end = r->adj.data() + r->adj_off[(size_t)v + 1];
    const Arc *pick = nullptr;
    for (; arc != end; ++arc) {
      if (r->estat[(size_t)arc->e] == V_INVALID || r->vstat[(size_t)arc->v] == V_INVALID) continue;
      const unsigned long long bu = dist[arc->v];
      if (bu >= inf_b) continue;
      if (__builtin_bit_cast(double, bu) + arc->w == dv && bu < dist[v]) { pick = arc; break; }
    }
    if (!pick) break;                                           // (cannot happen at a fixed point; the caller searches on)
    path_e.push_back(pick->e);
    v = pick->v;
  }
  path.clear(); path_e.clear();
  return -1;
}

// expansions after which a host search is given to sweep_search (0: never): a sixth of the graph, 50 000 at least (well above the kernel's
// budget: what comes back over that is still a core's work); TENDON_HIP_SEARCH_SWEEP=n
// overrides (tests: a small n sends most searches that way), =0 switches it off
int64_t sweep_cap(const tr_roadmap *r) {
  if (const char *e = std::getenv("TENDON_HIP_SEARCH_SWEEP")) return std::max<long long>(0, std::atoll(e));
  return r->ds.state == 1 ? std::max<int64_t>(50000, r->V / 6) : 0;
}

// One round's searches in two halves, so that the host threads can search their share while the kernel runs.
// device_search_launch: the queries active[klist[.]], in that order (the caller puts the ones it expects to be long first), are sent
// to the kernel; nothing is waited for.  `budget` caps the pops of one search (0: no cap): a search that reaches it is handed back.
// Returns false when the device cannot take the round (the caller then searches everything on the host).
// device_search_collect: waits for the kernel and leaves found[k] / paths / paths_e as the host search would; the queries the
// kernel gave up on (SR_FALLBACK) are listed in `redo`.
bool device_search_launch(tr_roadmap *r, const int32_t *starts, const int32_t *goals, const std::vector<int64_t> &active,
                          const std::vector<size_t> &klist, int64_t budget) {
  if (!search_setup(r)) return false;
  auto &d = r->ds;
  const int dev = tr_device(r->ctx);
  const int64_t nq = (int64_t)klist.size(), V = r->V;
  if (nq == 0) return false;
  if (!search_tables(r, nq)) return false;
  const int L = r->lm_n > 0 ? r->lm_n : 0;
  if (L > trk::SR_MAXL) return false;
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  if (nq > d.nq_cap) {
    if (d.qarena) dev_cache().release(d.qarena);
    d.qarena = nullptr;
    d.nq_cap = std::max<int64_t>(nq, 1024);
    d.pbuf_cap = (uint32_t)std::min<int64_t>((int64_t)d.nq_cap * 256, (int64_t)1 << 28);
    const size_t bq = up((size_t)d.nq_cap * 4);
    if (dev_cache().alloc(dev, (void **)&d.qarena, 4 * bq + up((size_t)d.nq_cap) + (size_t)d.pbuf_cap * 4) != hipSuccess) { d.nq_cap = 0; return false; }
    char *p = d.qarena;
    d.d_qs = (int32_t *)p; p += bq;
    d.d_qg = (int32_t *)p; p += bq;
    d.d_poff = (int32_t *)p; p += bq;
    d.d_plen = (int32_t *)p; p += bq;
    d.d_found = (uint8_t *)p; p += up((size_t)d.nq_cap);
    d.d_pbuf = (int32_t *)p;
  }
  if (nq > d.handback_cap) {
    if (d.h_handback) (void)hipHostFree(d.h_handback);
    d.h_handback = nullptr; d.d_handback = nullptr; d.handback_cap = 0;
    const int64_t cap = std::max<int64_t>(nq, 4096);
    if (hipHostMalloc((void **)&d.h_handback, (size_t)cap * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
        hipHostGetDevicePointer((void **)&d.d_handback, d.h_handback, 0) == hipSuccess) d.handback_cap = cap;
    else { if (d.h_handback) (void)hipHostFree(d.h_handback); d.h_handback = nullptr; d.d_handback = nullptr; }
  }
  if (d.h_handback) std::memset(d.h_handback, 0, (size_t)nq * sizeof(uint32_t));
  if (d.gens_issued + (uint64_t)nq >= ((uint64_t)1 << 31) - 2) {     // (a generation may not come round again while its records could be met)
    if (hipMemsetAsync(d.tables, 0, d.table_bytes, nullptr) != hipSuccess) return false;
    d.gens_issued = 0;
  }
  const uint32_t gen_base = (uint32_t)d.gens_issued;
  d.gens_issued += (uint64_t)nq;
  std::vector<int32_t> &qs = d.h_qs, &qg = d.h_qg;                  // (members: the copies below may still be reading them when this returns)
  qs.resize((size_t)nq); qg.resize((size_t)nq);
  for (int64_t j = 0; j < nq; j++) { qs[(size_t)j] = starts[active[klist[(size_t)j]]]; qg[(size_t)j] = goals[active[klist[(size_t)j]]]; }
  bool ok = true;
  if (!d.lm_current) {
    // the vertices' rows: state | landmark distances (padded to a multiple of four floats with zeros, which bound nothing)
    d.row_bytes = trk::search_row_bytes(r->S, L);
    const int lm_off = trk::search_lm_offset(r->S);
    std::vector<char> &rows = d.h_vrows;                          // (a member: the copy below may still be reading it when this returns)
    rows.assign((size_t)V * d.row_bytes, 0);
    for (int64_t v = 0; v < V; v++) {
      char *row = rows.data() + (size_t)v * d.row_bytes;
      std::memcpy(row, &r->states[(size_t)v * r->S], (size_t)r->S * 8);
      if (L) {
        float *lm = (float *)(row + lm_off);
        for (int l = 0; l < L; l++) { const float x = r->lm_d[(size_t)v * L + l]; lm[l] = x < trk::SR_LM_FAR ? x : trk::SR_LM_FAR; }   // (+inf: see the kernel's heuristic)
      }
    }
    ok = hipMemcpyAsync(d.d_vrows, rows.data(), rows.size(), hipMemcpyHostToDevice, nullptr) == hipSuccess;
    if (!ok) return false;
    d.lm_current = true;
  }
  const bool shared_status = r->dc.status_current;               // this round's validity bytes are in HBM already (component_labels)
  ok = ok && (shared_status || (hipMemcpyAsync(d.d_vstat, r->vstat.data(), (size_t)V, hipMemcpyHostToDevice, nullptr) == hipSuccess &&
       (r->E == 0 || hipMemcpyAsync(d.d_estat, r->estat.data(), (size_t)r->E, hipMemcpyHostToDevice, nullptr) == hipSuccess))) &&
       hipMemcpyAsync(d.d_qs, qs.data(), (size_t)nq * 4, hipMemcpyHostToDevice, nullptr) == hipSuccess &&
       hipMemcpyAsync(d.d_qg, qg.data(), (size_t)nq * 4, hipMemcpyHostToDevice, nullptr) == hipSuccess &&
       hipMemsetAsync(d.d_ctl, 0, d.ctl_bytes, nullptr) == hipSuccess &&
       hipMemsetAsync(d.d_ctl + 40, 0xff, 8, nullptr) == hipSuccess;       // (a -DTRK_SEARCH_CLOCKS build keeps the first wave's start there)
  if (!ok) return false;
  trk::SearchArgs a{};
  a.rows = d.d_rows; a.states = (const double *)d.d_vrows; a.lm = L ? (const float *)(d.d_vrows + trk::search_lm_offset(r->S)) : nullptr;
  a.row_bytes = d.row_bytes;
  a.S = r->S; a.NT = r->NT; a.rot = r->rot; a.ret = r->ret; a.L = L;
  a.w_rot = r->w_rot; a.w_ret = r->w_ret; a.lm_slack = kLmSlack;
  a.vstat = shared_status ? r->dc.d_vstat : d.d_vstat; a.estat = shared_status ? r->dc.d_estat : d.d_estat; a.deg = d.d_deg; a.V = V; a.E = r->E;
  a.qs = d.d_qs; a.qg = d.d_qg; a.nq = nq;
  a.next = d.d_ctl; a.pbuf_used = d.d_ctl + 1; a.expanded = (unsigned long long *)(d.d_ctl + 2);
  a.base = d.tables; a.lc0 = d.lc0; a.gen_base = gen_base;
  for (int c = 0; c < trk::SR_CLASSES; c++) { a.pool[c] = d.pool[c]; a.pool_n[c] = d.pool_n[c]; a.pool_word[c] = d.pool_word[c]; }
  a.found = d.d_found; a.poff = d.d_poff; a.plen = d.d_plen; a.pbuf = d.d_pbuf; a.pbuf_cap = d.pbuf_cap;
  a.handback = d.handback_cap >= nq ? d.d_handback : nullptr;
  a.max_pops = budget > 0 ? budget : 16 * V + 1024;             // (uncapped: every vertex reopened a few times, far beyond what a search does)
  a.kbest = search_kbest();
  const unsigned grid = (unsigned)std::min<int64_t>(d.slots, nq);
  if (!d.ev[0] && (hipEventCreate(&d.ev[0]) != hipSuccess || hipEventCreate(&d.ev[1]) != hipSuccess)) { d.ev[0] = d.ev[1] = nullptr; }
  if (d.ev[0]) (void)hipEventRecord(d.ev[0], nullptr);
  hipLaunchKernelGGL(search_kernel_for(r->S), dim3(grid), dim3(64), trk::search_lds_bytes(), nullptr, a);
  if (hipGetLastError() != hipSuccess) return false;
  if (d.ev[0]) { (void)hipEventRecord(d.ev[1], nullptr); d.ev_pending = true; }
  d.in_flight = nq;
  return true;
}

void device_search_collect(tr_roadmap *r, const std::vector<int64_t> &active, const std::vector<size_t> &klist,
                           std::vector<uint8_t> &found, std::vector<std::vector<int32_t>> &paths,
                           std::vector<std::vector<int32_t>> &paths_e, std::vector<size_t> &redo, int64_t &expanded, int T,
                           const std::vector<uint8_t> *handled = nullptr) {
  auto &d = r->ds;
  const int64_t nq = d.in_flight;
  d.in_flight = 0;
  // (any failure: the whole list goes back to the host threads)
  // (`handled`: positions the host threads have searched already -- handed back while the kernel ran: their answers stand)
  auto give_back = [&]() { redo.clear(); for (size_t k : klist) if (!handled || !(*handled)[k]) { redo.push_back(k); found[k] = 0; } };
  redo.clear();
  if (nq != (int64_t)klist.size()) { give_back(); return; }
  bool ok = true;
  std::vector<uint8_t> res((size_t)nq);
  std::vector<int32_t> poff((size_t)nq), plen((size_t)nq);
  uint32_t ctl[136] = {0};
  ok = hipMemcpy(ctl, d.d_ctl, sizeof(ctl), hipMemcpyDeviceToHost) == hipSuccess &&
       hipMemcpy(res.data(), d.d_found, (size_t)nq, hipMemcpyDeviceToHost) == hipSuccess &&
       hipMemcpy(poff.data(), d.d_poff, (size_t)nq * 4, hipMemcpyDeviceToHost) == hipSuccess &&
       hipMemcpy(plen.data(), d.d_plen, (size_t)nq * 4, hipMemcpyDeviceToHost) == hipSuccess;
  if (!ok) { give_back(); return; }
  if (d.ev_pending) {
    float ms_ = 0.0f;
    if (hipEventElapsedTime(&ms_, d.ev[0], d.ev[1]) == hipSuccess) { d.st_kernel_ms += (double)ms_; d.st_launches++; }
    d.ev_pending = false;
  }
  if (std::getenv("TENDON_HIP_SEARCH_STATS")) {                     // (non-zero only in a -DTRK_SEARCH_CLOCKS build)
    unsigned long long c[7];
    std::memcpy(c, &ctl[16], sizeof(c));
    const double tot = (double)(c[0] + c[1] + c[2] + c[3] + c[4]);
    unsigned long long sp[2];
    std::memcpy(sp, &ctl[32], sizeof(sp));
    if (tot > 0) std::fprintf(stderr, "[tendon_hip] search steps: %llu steps, %llu passes, %.2f us per step\n", sp[0], sp[1], sp[0] ? tot * 1e-2 / (double)sp[0] : 0.0);
    if (tot > 0) {
      unsigned long long rx[3];
      std::memcpy(rx, &ctl[124], sizeof(rx));
      std::fprintf(stderr, "[tendon_hip] inside arcs + rows + relax: loads + heuristic + lookup %.1f%%, conflicts %.1f%%, claims + writes %.1f%% (of all)\n",
                   100.0 * rx[0] / tot, 100.0 * rx[1] / tot, 100.0 * rx[2] / tot);
      std::fprintf(stderr, "[tendon_hip] searches ended per 2 ms (count/expansions):");
      for (int b = 0; b < 40; b++) if (ctl[44 + b]) std::fprintf(stderr, " %d:%u/%u", 2 * b, ctl[44 + b], ctl[84 + b]);
      std::fprintf(stderr, "\n");
    }
    if (tot > 0)
      std::fprintf(stderr, "[tendon_hip] search clocks: %.1f wave-ms in all (longest search %.2f ms): refill %.1f%%, pop %.1f%%, record + offsets %.1f%%, arcs + rows + relax %.1f%%, append %.1f%%\n",
                   tot * 1e-5, (double)c[6] * 1e-5, 100.0 * c[0] / tot, 100.0 * c[1] / tot, 100.0 * c[2] / tot, 100.0 * c[3] / tot, 100.0 * c[4] / tot);
  }
  const uint32_t used = std::min(ctl[1], d.pbuf_cap);
  std::vector<int32_t> pbuf((size_t)used);
  if (used && hipMemcpy(pbuf.data(), d.d_pbuf, (size_t)used * 4, hipMemcpyDeviceToHost) != hipSuccess) { give_back(); return; }
  unsigned long long ex = 0;
  std::memcpy(&ex, &ctl[2], sizeof(ex));
  expanded += (int64_t)ex;
  d.st_expanded += (int64_t)ex;
  const int Tb = nq >= 2048 ? std::max(1, std::min(T, 16)) : 1;            // (ten thousand small vectors: by ranges on the host threads)
  std::vector<std::vector<size_t>> part((size_t)Tb);
  std::vector<int64_t> nfb((size_t)Tb, 0);
  on_threads(Tb, [&](int t) {
    const int64_t j0 = nq * t / Tb, j1 = nq * (t + 1) / Tb;
    for (int64_t j = j0; j < j1; j++) {
      const size_t k = klist[(size_t)j];
      const int64_t q = active[k];
      if (res[(size_t)j] == trk::SR_FALLBACK) nfb[(size_t)t]++;
      if (handled && (*handled)[k]) continue;
      found[k] = 0;
      if (res[(size_t)j] == trk::SR_FALLBACK) { part[(size_t)t].push_back(k); continue; }
      if (res[(size_t)j] != trk::SR_FOUND) continue;
      const int32_t n = plen[(size_t)j], o = poff[(size_t)j];
      if (n < 1 || o < 0 || (uint64_t)o + (uint64_t)(2 * n - 1) > used) { part[(size_t)t].push_back(k); continue; }
      paths[(size_t)q].assign(pbuf.begin() + o, pbuf.begin() + o + n);
      paths_e[(size_t)q].assign(pbuf.begin() + o + n, pbuf.begin() + o + 2 * n - 1);
      found[k] = 1;
    }
  });
  for (const auto &p : part) redo.insert(redo.end(), p.begin(), p.end());
  int64_t n_fb = 0;
  for (int64_t x : nfb) n_fb += x;
  n_fb = std::max<int64_t>(n_fb, (int64_t)redo.size());          // (a path that did not fit its buffer comes back too)
  d.st_queries += nq - n_fb; d.st_fallbacks += n_fb; d.st_moves += (int64_t)ctl[4];
  d.st_grows += (int64_t)ctl[5]; d.st_max_records = std::max<int64_t>(d.st_max_records, (int64_t)ctl[6]);
}

// validity of the listed combined items (vertex v -> v, edge e -> V + e) against the current obstacle grid: one K4 launch
int check_items(tr_roadmap *r, const std::vector<int32_t> &list, std::vector<uint8_t> &hit) {
  hit.assign(list.size(), 0);
  if (list.empty()) return TR_OK;
  if (!r->has_caches) return rfail(r, TR_ERR_INVALID_ARG, "no voxel caches attached (tr_roadmap_set_caches)");
  const int64_t n = (int64_t)list.size();
  if (n > r->list_cap) {
    if (r->d_list) dev_cache().release(r->d_list);
    if (r->d_hit) dev_cache().release(r->d_hit);
    r->d_list = nullptr; r->d_hit = nullptr;
    r->list_cap = std::max<int64_t>(n + n / 2, 1 << 14);
    RM_HIP(r, dev_cache().alloc(tr_device(r->ctx), (void **)&r->d_list, (size_t)r->list_cap * sizeof(int32_t)));
    RM_HIP(r, dev_cache().alloc(tr_device(r->ctx), (void **)&r->d_hit, (size_t)r->list_cap));
  }
  RM_HIP(r, hipMemcpy(r->d_list, list.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
  const int rc = tr_check_cached_subset_dev(r->ctx, r->d_ids, r->d_masks, r->d_off, r->V + r->E, r->d_list, n, r->d_hit, nullptr);
  if (rc) return rfail(r, rc, tr_last_error(r->ctx));
  RM_HIP(r, hipMemcpy(hit.data(), r->d_hit, (size_t)n, hipMemcpyDeviceToHost));     // synchronises with the launch
  return TR_OK;
}

}  // namespace

extern "C" {

const char *tr_roadmap_last_error(const tr_roadmap *r) { return r ? r->err.c_str() : "null roadmap"; }

int tr_roadmap_create(tr_ctx *ctx, const double *states, int64_t n_vertices, const int32_t *edges, const double *weights,
                      int64_t n_edges, tr_roadmap **out) {
  if (!ctx || !out || n_vertices < 0 || n_edges < 0 || (n_vertices > 0 && !states) || (n_edges > 0 && !edges)) return TR_ERR_INVALID_ARG;
  *out = nullptr;
  if (n_vertices > std::numeric_limits<int32_t>::max() / 2 || n_edges > std::numeric_limits<int32_t>::max() / 2) return TR_ERR_INVALID_ARG;
  tr_roadmap *r = new tr_roadmap();
  r->ctx = ctx;
  r->S = tr_state_size(ctx);
  r->V = n_vertices; r->E = n_edges;
  tr_space_weights(ctx, &r->w_rot, &r->w_ret);
  {
    int rot = 0, ret = 0, nt = 0;
    tr_state_layout(ctx, &nt, &rot, &ret);
    r->NT = nt; r->rot = rot != 0; r->ret = ret != 0;
  }
  Laps laps("tr_roadmap_create");
  r->states.assign(states, states + (size_t)n_vertices * r->S);
  r->eu.resize_uninit((size_t)n_edges); r->ev.resize_uninit((size_t)n_edges); r->w.resize_uninit((size_t)n_edges);
  // One team of T threads, three phases with a barrier between them (a team per phase cost ~1 ms each in thread starts):
  //   1. the edge arrays by edge ranges (end points, weights);  2. degrees by edge ranges (atomic increments), then thread 0 turns
  //   them into offsets and sizes the arc array;  3. the arcs by VERTEX ranges: every thread scans all edges in order and takes the
  //   arcs that leave its vertices, so a vertex's arcs keep the edge order whatever T is (atomic cursors + a per-vertex sort were
  //   slower: 5.6 against 3.0 ms, scattered first touches of the 19 MB).
  const int T = n_edges >= (1 << 16) ? std::min(host_threads(0), 16) : 1;
  std::atomic<int> bad{TR_OK};
  r->adj_off.assign((size_t)n_vertices + 1, 0);
  struct Barrier {
    const int n; std::atomic<int> arrived{0}, phase{0};
    explicit Barrier(int n_) : n(n_) {}
    void wait() {
      const int ph = phase.load(std::memory_order_acquire);
      if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == n) { arrived.store(0, std::memory_order_relaxed); phase.store(ph + 1, std::memory_order_release); }
      else while (phase.load(std::memory_order_acquire) == ph) std::this_thread::yield();
    }
  } barrier(T);
  on_threads(T, [&](int t) {
    const int64_t lo = n_edges * t / T, hi = n_edges * (t + 1) / T;
    for (int64_t e = lo; e < hi; e++) {
      const int32_t a = edges[2 * e], b = edges[2 * e + 1];
      if (a < 0 || a >= n_vertices || b < 0 || b >= n_vertices) { bad = TR_ERR_OUT_OF_RANGE; break; }
      r->eu[(size_t)e] = a; r->ev[(size_t)e] = b;
      // edge cost = opt_->motionCost = si->distance(a, b) unless the file supplies one (weightProperty_, :2598-2603)
      r->w[(size_t)e] = weights ? weights[e] : state_distance(r, &r->states[(size_t)a * r->S], &r->states[(size_t)b * r->S]);
      if (!(r->w[(size_t)e] >= 0)) { int want = TR_OK; bad.compare_exchange_strong(want, TR_ERR_INVALID_ARG); break; }
    }
    barrier.wait();
    if (bad != TR_OK) return;                                   // (every thread sees the same value after the barrier)
    for (int64_t e = lo; e < hi; e++) {
      __atomic_fetch_add(&r->adj_off[(size_t)r->eu[(size_t)e] + 1], (int64_t)1, __ATOMIC_RELAXED);
      __atomic_fetch_add(&r->adj_off[(size_t)r->ev[(size_t)e] + 1], (int64_t)1, __ATOMIC_RELAXED);
    }
    barrier.wait();
    if (t == 0) {
      for (int64_t v = 0; v < n_vertices; v++) r->adj_off[(size_t)v + 1] += r->adj_off[(size_t)v];
      r->adj.resize_uninit((size_t)r->adj_off[(size_t)n_vertices]);
    }
    barrier.wait();
    const int32_t vlo = (int32_t)(n_vertices * t / T), vhi = (int32_t)(n_vertices * (t + 1) / T);
    if (vlo == vhi) return;
    std::vector<int64_t> fill(r->adj_off.begin() + vlo, r->adj_off.begin() + vhi);
    for (int64_t e = 0; e < n_edges; e++) {
      const int32_t a = r->eu[(size_t)e], b = r->ev[(size_t)e];
      if (a >= vlo && a < vhi) r->adj[(size_t)fill[(size_t)(a - vlo)]++] = Arc{b, (int32_t)e, r->w[(size_t)e]};
      if (b >= vlo && b < vhi) r->adj[(size_t)fill[(size_t)(b - vlo)]++] = Arc{a, (int32_t)e, r->w[(size_t)e]};
    }
  });
  if (bad != TR_OK) { const int rc = bad; delete r; return rc; }
  laps.lap("edges + degrees + adjacency");
  r->vstat.assign((size_t)n_vertices, V_UNKNOWN); r->estat.assign((size_t)n_edges, V_UNKNOWN);
  r->vpresent.assign((size_t)n_vertices, 1); r->epresent.assign((size_t)n_edges, 1);
  { std::lock_guard<std::mutex> g(g_roadmaps_mu); g_roadmaps.push_back(r); }
  *out = r;
  return TR_OK;
}

void tr_roadmap_destroy(tr_roadmap *r) {
  if (!r) return;
  { std::lock_guard<std::mutex> g(g_roadmaps_mu); g_roadmaps.erase(std::remove(g_roadmaps.begin(), g_roadmaps.end(), r), g_roadmaps.end()); }
  (void)hipSetDevice(tr_device(r->ctx));
  free_dev(r);
  free_search(r);
  free_comp(r);
  delete r;
}

}  // extern "C"
namespace {
int set_caches_impl(tr_roadmap *r, const int64_t *v_offsets, const uint32_t *v_ids, const uint64_t *v_masks,
                    const uint64_t *v_present_bits, const int64_t *e_offsets, const uint32_t *e_ids,
                    const uint64_t *e_masks, const uint64_t *e_present_bits, hipMemcpyKind kind) {
  if (!r) return TR_ERR_INVALID_ARG;
  RmLock lock_(r);
  if (!v_offsets || !e_offsets) return rfail(r, TR_ERR_INVALID_ARG, "null offsets");
  const int64_t nv = v_offsets[r->V], ne = e_offsets[r->E];
  if (nv < 0 || ne < 0 || (nv > 0 && (!v_ids || !v_masks)) || (ne > 0 && (!e_ids || !e_masks))) return rfail(r, TR_ERR_INVALID_ARG, "bad CSR arrays");
  for (int64_t i = 0; i < r->V; i++) if (v_offsets[i] > v_offsets[i + 1]) return rfail(r, TR_ERR_INVALID_ARG, "offsets must be non-decreasing");
  for (int64_t i = 0; i < r->E; i++) if (e_offsets[i] > e_offsets[i + 1]) return rfail(r, TR_ERR_INVALID_ARG, "offsets must be non-decreasing");
  RM_HIP(r, hipSetDevice(tr_device(r->ctx)));
  free_dev(r);
  const int64_t items = r->V + r->E;
  r->nnz = nv + ne;
  std::vector<int64_t> off((size_t)items + 1);
  for (int64_t i = 0; i <= r->V; i++) off[(size_t)i] = v_offsets[i];
  for (int64_t i = 0; i <= r->E; i++) off[(size_t)(r->V + i)] = nv + e_offsets[i];
  const int dev = tr_device(r->ctx);
  RM_HIP(r, dev_cache().alloc(dev, (void **)&r->d_ids, std::max<size_t>(1, (size_t)r->nnz) * sizeof(uint32_t)));
  RM_HIP(r, dev_cache().alloc(dev, (void **)&r->d_masks, std::max<size_t>(1, (size_t)r->nnz) * sizeof(uint64_t)));
  RM_HIP(r, dev_cache().alloc(dev, (void **)&r->d_off, off.size() * sizeof(int64_t)));
  RM_HIP(r, dev_cache().alloc(dev, (void **)&r->d_bits, ((size_t)items / 64 + 1) * sizeof(uint64_t)));
  if (nv) {
    RM_HIP(r, hipMemcpy(r->d_ids, v_ids, (size_t)nv * sizeof(uint32_t), kind));
    RM_HIP(r, hipMemcpy(r->d_masks, v_masks, (size_t)nv * sizeof(uint64_t), kind));
  }
  if (ne) {
    RM_HIP(r, hipMemcpy(r->d_ids + nv, e_ids, (size_t)ne * sizeof(uint32_t), kind));
    RM_HIP(r, hipMemcpy(r->d_masks + nv, e_masks, (size_t)ne * sizeof(uint64_t), kind));
  }
  RM_HIP(r, hipMemcpy(r->d_off, off.data(), off.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  for (int64_t i = 0; i < r->V; i++) r->vpresent[(size_t)i] = v_present_bits ? (uint8_t)((v_present_bits[i >> 6] >> (i & 63)) & 1) : 1;
  for (int64_t i = 0; i < r->E; i++) r->epresent[(size_t)i] = e_present_bits ? (uint8_t)((e_present_bits[i >> 6] >> (i & 63)) & 1) : 1;
  // what tr_roadmap_revalidate needs on the host: a pinned image of the hit words and, per word, the items without a cache
  RM_HIP(r, hipHostMalloc((void **)&r->h_bits, ((size_t)items / 64 + 1) * sizeof(uint64_t), hipHostMallocDefault));
  r->absent.assign((size_t)items / 64 + 1, 0);
  for (int64_t i = 0; i < r->V; i++) if (!r->vpresent[(size_t)i]) r->absent[(size_t)i >> 6] |= (uint64_t)1 << (i & 63);
  for (int64_t i = 0; i < r->E; i++) if (!r->epresent[(size_t)i]) { const int64_t q = r->V + i; r->absent[(size_t)q >> 6] |= (uint64_t)1 << (q & 63); }
  r->has_caches = true;
  return TR_OK;
}
}  // namespace
extern "C" {

int tr_roadmap_set_caches(tr_roadmap *r, const int64_t *v_offsets, const uint32_t *v_ids, const uint64_t *v_masks,
                          const uint64_t *v_present_bits, const int64_t *e_offsets, const uint32_t *e_ids,
                          const uint64_t *e_masks, const uint64_t *e_present_bits) {
  return set_caches_impl(r, v_offsets, v_ids, v_masks, v_present_bits, e_offsets, e_ids, e_masks, e_present_bits, hipMemcpyHostToDevice);
}

int tr_roadmap_set_caches_dev(tr_roadmap *r, const int64_t *v_offsets, const uint32_t *d_v_ids, const uint64_t *d_v_masks,
                              const uint64_t *v_present_bits, const int64_t *e_offsets, const uint32_t *d_e_ids,
                              const uint64_t *d_e_masks, const uint64_t *e_present_bits) {
  return set_caches_impl(r, v_offsets, d_v_ids, d_v_masks, v_present_bits, e_offsets, d_e_ids, d_e_masks, e_present_bits, hipMemcpyDeviceToDevice);
}

int tr_roadmap_prepare(tr_roadmap *r, int32_t n_landmarks, int32_t n_threads) {
  if (!r) return TR_ERR_INVALID_ARG;
  RmLock lock_(r);
  if (n_landmarks < 0 || n_landmarks > 64) return rfail(r, TR_ERR_INVALID_ARG, "landmark count must be in [0, 64]");
  build_landmarks(r, n_landmarks, host_threads(n_threads));
  if (r->lm_mismatch) return rfail(r, TR_ERR_RUNTIME, "landmark distances: the device's table differs from the host's (TENDON_HIP_LANDMARKS=check)");
  return TR_OK;
}

int tr_roadmap_clear_validity(tr_roadmap *r) {
  if (!r) return TR_ERR_INVALID_ARG;
  RmLock lock_(r);
  std::fill(r->vstat.begin(), r->vstat.end(), (uint8_t)V_UNKNOWN);
  std::fill(r->estat.begin(), r->estat.end(), (uint8_t)V_UNKNOWN);
  return TR_OK;
}

}  // extern "C"
namespace {
// (the caller holds r->mu)
int revalidate_locked(tr_roadmap *r, int64_t *n_invalid_vertices, int64_t *n_invalid_edges) {
  if (!r->has_caches) return rfail(r, TR_ERR_INVALID_ARG, "no voxel caches attached (tr_roadmap_set_caches)");
  RM_HIP(r, hipSetDevice(tr_device(r->ctx)));
  const int64_t items = r->V + r->E;
  int64_t nv = 0, ne = 0;
  if (items > 0) {
    const int rc = tr_check_cached_dev(r->ctx, r->d_ids, r->d_masks, r->d_off, items, r->d_bits, nullptr);
    if (rc) return rfail(r, rc, tr_last_error(r->ctx));
    // one kernel, one copy of the hit words into pinned memory, and a pass over WORDS, not items: a word's 64 items are marked
    // valid at once, then its set bits -- hits and missing caches, a few per cent of the items -- invalid one by one
    // (the per-item pass of round 3 was 0.4 of the call's 0.58 ms at 6.8 x 10^5 items, the kernel 0.115)
    const size_t nw = (size_t)(items + 63) / 64;
    RM_HIP(r, hipMemcpyAsync(r->h_bits, r->d_bits, nw * sizeof(uint64_t), hipMemcpyDeviceToHost, nullptr));
    RM_HIP(r, hipStreamSynchronize(nullptr));
    uint8_t *vs = r->vstat.data(), *es = r->estat.data();
    const int64_t V = r->V;
    for (size_t w = 0; w < nw; w++) {
      const int64_t q0 = (int64_t)w * 64, q1 = std::min<int64_t>(q0 + 64, items);
      uint64_t bad = r->h_bits[w] | r->absent[w];
      if (q1 - q0 < 64) bad &= ((uint64_t)1 << (q1 - q0)) - 1;
      if (q1 <= V || q0 >= V) {                             // the word's items are all vertices or all edges: valid, except the set bits
        uint8_t *st = q1 <= V ? vs + q0 : es + (q0 - V);
        std::memset(st, V_VALID, (size_t)(q1 - q0));
        (q1 <= V ? nv : ne) += __builtin_popcountll(bad);
        for (; bad; bad &= bad - 1) st[__builtin_ctzll(bad)] = V_INVALID;
        continue;
      }
      for (int64_t q = q0; q < q1; q++) {                   // (the one word that holds the last vertices and the first edges)
        const bool b = (bad >> (q - q0)) & 1;
        if (q < V) { vs[q] = b ? V_INVALID : V_VALID; nv += b; }
        else { es[q - V] = b ? V_INVALID : V_VALID; ne += b; }
      }
    }
  }
  if (n_invalid_vertices) *n_invalid_vertices = nv;
  if (n_invalid_edges) *n_invalid_edges = ne;
  return TR_OK;
}
}  // namespace
extern "C" {

int tr_roadmap_revalidate(tr_roadmap *r, int64_t *n_invalid_vertices, int64_t *n_invalid_edges) {
  if (!r) return TR_ERR_INVALID_ARG;
  RmLock lock_(r);
  return revalidate_locked(r, n_invalid_vertices, n_invalid_edges);
}

int tr_roadmap_get_validity(tr_roadmap *r, uint8_t *vertex_status, uint8_t *edge_status) {
  if (!r) return TR_ERR_INVALID_ARG;
  RmLock lock_(r);
  if (vertex_status) std::memcpy(vertex_status, r->vstat.data(), r->vstat.size());
  if (edge_status) std::memcpy(edge_status, r->estat.data(), r->estat.size());
  return TR_OK;
}

int tr_roadmap_set_validity(tr_roadmap *r, const uint8_t *vertex_status, const uint8_t *edge_status) {
  if (!r) return TR_ERR_INVALID_ARG;
  RmLock lock_(r);
  for (int64_t i = 0; vertex_status && i < r->V; i++) if (vertex_status[i] > V_INVALID) return rfail(r, TR_ERR_INVALID_ARG, "vertex status must be 0, 1 or 2");
  for (int64_t i = 0; edge_status && i < r->E; i++) if (edge_status[i] > V_INVALID) return rfail(r, TR_ERR_INVALID_ARG, "edge status must be 0, 1 or 2");
  if (vertex_status) std::memcpy(r->vstat.data(), vertex_status, r->vstat.size());
  if (edge_status) std::memcpy(r->estat.data(), edge_status, r->estat.size());
  return TR_OK;
}

int tr_roadmap_solve(tr_roadmap *r, const int32_t *starts, const int32_t *goals, int64_t n_queries, int32_t n_threads,
                     int32_t *status, double *cost, int64_t *path_offsets, tr_roadmap_stats *stats) {
  if (!r) return TR_ERR_INVALID_ARG;
  RmLock lock_(r);
  if (n_queries < 0 || (n_queries > 0 && (!starts || !goals || !status || !path_offsets))) return rfail(r, TR_ERR_INVALID_ARG, "bad argument");
  r->path_off.assign((size_t)n_queries + 1, 0); r->path_v.clear();
  r->st_rounds = r->st_items_checked = r->st_astar_runs = r->st_expanded = 0;
  r->ds.st_queries = r->ds.st_fallbacks = r->ds.st_host_share = r->ds.st_moves = r->ds.st_expanded = r->ds.st_grows = r->ds.st_max_records = 0;
  r->ds.st_kernel_ms = 0; r->ds.st_launches = 0;
  r->ds.st_sweeps = 0; r->ds.sweep_round = -1;
  r->dc.st_cut = 0;
  if (path_offsets) path_offsets[0] = 0;
  if (n_queries == 0) { if (stats) *stats = tr_roadmap_stats{0, 0, 0, 0}; return TR_OK; }
  for (int64_t q = 0; q < n_queries; q++)
    if (starts[q] < 0 || starts[q] >= r->V || goals[q] < 0 || goals[q] >= r->V) return rfail(r, TR_ERR_OUT_OF_RANGE, "query vertex outside the roadmap");
  RM_HIP(r, hipSetDevice(tr_device(r->ctx)));
  Laps laps("tr_roadmap_solve");
  const int T = host_threads(n_threads);
  if ((int)r->scratch.size() < T) r->scratch.resize((size_t)T);
  if (r->lm_n < 0 && n_queries >= 64) build_landmarks(r, 16, T);         // a handful of queries does not repay 16 graph sweeps
  // (the queries' path vectors live with the roadmap: ten thousand small vectors cost 1 - 2 ms to allocate and to free per call otherwise)
  std::vector<std::vector<int32_t>> &paths = r->paths_buf, &paths_e = r->paths_e_buf;
  if ((int64_t)paths.size() < n_queries) { paths.resize((size_t)n_queries); paths_e.resize((size_t)n_queries); }
  for (int64_t q = 0; q < n_queries; q++) { paths[(size_t)q].clear(); paths_e[(size_t)q].clear(); }
  std::vector<int32_t> list;
  std::vector<uint8_t> hit;
  std::vector<uint8_t> vmark((size_t)r->V, 0), emark((size_t)r->E, 0);
  int rc;

  // items become known: a missing cache (the voxelisation found the shape invalid when the cache was built) is invalid for good
  auto record = [&](const std::vector<int32_t> &lst, const std::vector<uint8_t> &h) {
    for (size_t k = 0; k < lst.size(); k++) {
      const int64_t it = lst[k];
      if (it < r->V) r->vstat[(size_t)it] = (h[k] || !r->vpresent[(size_t)it]) ? V_INVALID : V_VALID;
      else r->estat[(size_t)(it - r->V)] = (h[k] || !r->epresent[(size_t)(it - r->V)]) ? V_INVALID : V_VALID;
    }
    r->st_items_checked += (int64_t)lst.size();
  };

  // the query end points first (solvePrep :2978-3010 only admits valid start / goal states)
  for (int64_t q = 0; q < n_queries; q++) {
    for (int32_t v : {starts[q], goals[q]})
      if (r->vstat[(size_t)v] == V_UNKNOWN && !vmark[(size_t)v]) { vmark[(size_t)v] = 1; list.push_back(v); }
  }
  if (!list.empty()) {
    if ((rc = check_items(r, list, hit))) return rc;
    record(list, hit);
    for (int32_t v : list) vmark[(size_t)v] = 0;
  }
  std::vector<int64_t> active;
  for (int64_t q = 0; q < n_queries; q++) {
    status[q] = TR_QUERY_SOLVED;
    if (cost) cost[q] = std::numeric_limits<double>::infinity();
    if (r->vstat[(size_t)starts[q]] == V_INVALID) status[q] = TR_QUERY_INVALID_START;
    else if (r->vstat[(size_t)goals[q]] == V_INVALID) status[q] = TR_QUERY_INVALID_GOAL;
    else if (starts[q] == goals[q]) { paths[(size_t)q] = {starts[q]}; if (cost) cost[q] = 0.0; }   // constructSolution :2696-2701
    else active.push_back(q);
  }

  std::vector<uint8_t> found;
  const int smode = search_mode();
  bool went_eager = false;
  // A batch so large that its candidate paths would hold a quarter of the cached sets anyway (at ~64 items a path) does not start
  // lazily: one launch tests every cached set (0.25 ms at 6.8 x 10^5 sets) and the searches run on known validity -- one round
  // instead of two or more.  A count, not a clock; TENDON_HIP_LAZY_ONLY=1: never (the reference's loop item by item).
  if (r->has_caches && (int64_t)active.size() * 256 >= r->V + r->E && !std::getenv("TENDON_HIP_LAZY_ONLY")) {
    int64_t unknown = 0;
    for (uint8_t x : r->vstat) unknown += x == V_UNKNOWN;
    for (uint8_t x : r->estat) unknown += x == V_UNKNOWN;
    if (unknown > 0) {
      if ((rc = revalidate_locked(r, nullptr, nullptr))) return rc;
      r->st_items_checked += unknown;
      went_eager = true;
      // (end points found invalid by that test: their queries end here, as they would have before the first search)
      std::vector<int64_t> keep;
      for (int64_t q : active) {
        if (r->vstat[(size_t)starts[q]] == V_INVALID) status[q] = TR_QUERY_INVALID_START;
        else if (r->vstat[(size_t)goals[q]] == V_INVALID) status[q] = TR_QUERY_INVALID_GOAL;
        else keep.push_back(q);
      }
      active.swap(keep);
    }
  }
  laps.lap("end points + set-up");
  while (!active.empty()) {
    r->st_rounds++;
    // A* for every unresolved query, on the host cores
    found.assign(active.size(), 0);
    std::atomic<int64_t> expanded{0};
    // queries whose end points lie in different components of what is left of the graph have no path: the labels answer them
    // (rounds of kComponentMinQueries or more; TENDON_HIP_COMPONENTS=0 searches them as before, to the same answer)
    r->dc.status_current = false;
    std::vector<size_t> todo;                                   // positions in `active` that need a search
    todo.reserve(active.size());
    const int cmode = components_mode();
    bool labels_now = false;
    auto ensure_labels = [&]() {
      if (labels_now) return true;
      const auto t_cc = std::chrono::steady_clock::now();
      if (cmode == 0 || !component_labels(r)) return false;
      labels_now = true;
      if (std::getenv("TENDON_HIP_SEARCH_STATS"))
        std::fprintf(stderr, "[tendon_hip] round %lld: component labels %.3f ms\n", (long long)r->st_rounds,
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_cc).count());
      return true;
    };
    if ((int64_t)active.size() >= kComponentMinQueries && (cmode == 2 || (cmode == 1 && r->dc.wanted)) && ensure_labels()) {
      const int32_t *lab = r->dc.label.data();
      for (size_t k = 0; k < active.size(); k++) {
        const int64_t q = active[k];
        if (lab[starts[q]] == lab[goals[q]]) todo.push_back(k);
      }
      r->dc.st_cut += (int64_t)(active.size() - todo.size());
    } else {
      for (size_t k = 0; k < active.size(); k++) todo.push_back(k);
    }
    // ... on the device when the round is large enough to fill it (search_kernel.hpp).  The searches are ordered by the state-space
    // distance between their end points, longest first: the host threads take the head of that order (a core expands a vertex in
    // a fraction of the time a wave does, so the searches expected to be longest are theirs) while the kernel works through the
    // rest, longest first; what the kernel hands back (over its pop budget, or a list full) the host threads search afterwards.
    std::vector<size_t> host_list, dev_list, redo;
    bool on_device = false;
    const auto t_round = std::chrono::steady_clock::now();
    if (!todo.empty() && (smode == 2 || (smode == 1 && (int64_t)todo.size() >= kSearchMinQueries))) {
      std::vector<std::pair<double, size_t>> key(todo.size());
      for (size_t j = 0; j < todo.size(); j++) {
        const size_t k = todo[j];
        const int64_t q = active[k];
        key[j] = {state_distance(r, &r->states[(size_t)starts[q] * r->S], &r->states[(size_t)goals[q] * r->S]), k};
      }
      std::sort(key.begin(), key.end(), [](const std::pair<double, size_t> &x, const std::pair<double, size_t> &y) { return x.first > y.first || (x.first == y.first && x.second < y.second); });
      const bool share_from_env = std::getenv("TENDON_HIP_SEARCH_HOST_SHARE") != nullptr;
      if (r->ds.share < 0 || share_from_env) r->ds.share = search_host_share();
      const size_t n_h = smode == 2 ? 0 : (size_t)((double)todo.size() * r->ds.share);
      for (size_t i = 0; i < key.size(); i++) (i < n_h ? host_list : dev_list).push_back(key[i].second);
      // (TENDON_HIP_SEARCH=device: no budget unless TENDON_HIP_SEARCH_BUDGET asks for one)
      {
        const bool from_env = std::getenv("TENDON_HIP_SEARCH_BUDGET") != nullptr;
        // (a roadmap's first shared round: 6 500 expansions, or a sixteenth of its vertices if that is more -- searches grow with the
        // graph; afterwards the budget doubles whenever more than one search in fifty came back: see below)
        if (r->ds.budget == 0 || from_env || r->ds.budget_from_env) r->ds.budget = from_env ? search_budget() : std::max<int64_t>(search_budget(), r->V / 16);
        r->ds.budget_from_env = from_env;
      }
      on_device = device_search_launch(r, starts, goals, active, dev_list, smode == 2 && !std::getenv("TENDON_HIP_SEARCH_BUDGET") ? 0 : r->ds.budget);
      if (!on_device) { host_list.clear(); dev_list.clear(); }
      else r->ds.st_host_share += (int64_t)host_list.size();
    }
    std::vector<int64_t> hist_v;
    if (std::getenv("TENDON_HIP_SEARCH_HIST")) hist_v.assign(active.size(), 0);
    int64_t *hist = hist_v.empty() ? nullptr : hist_v.data();
    std::vector<double> hist_f;                                   // (per search: Scratch::trace_f)
    if (hist) hist_f.assign(active.size() * 4, 0.0);
    // the host threads over a list of positions in `active` (null: all of them)
    std::atomic<bool> walked_in_vain{false};
    // one query on a host thread: A*, and past sweep_cap expansions the parallel sweep on the device (which failing, A* to the end)
    const int64_t cap_sweep = sweep_cap(r);
    auto one_search = [&](Scratch &sc, int64_t q, int64_t &ex) -> bool {
      bool abandoned = false;
      bool f = astar(r, sc, starts[q], goals[q], paths[(size_t)q], paths_e[(size_t)q], ex, cap_sweep, &abandoned);
      if (abandoned) {
        const int m = sweep_search(r, starts[q], goals[q], paths[(size_t)q], paths_e[(size_t)q]);
        f = m >= 0 ? m == 1 : astar(r, sc, starts[q], goals[q], paths[(size_t)q], paths_e[(size_t)q], ex);
      }
      return f;
    };
    auto host_search = [&](const std::vector<size_t> *list) {
      const int64_t n_host = list ? (int64_t)list->size() : (int64_t)active.size();
      if (n_host == 0) return;
      std::atomic<int64_t> next{0};
      auto worker = [&](int t) {
        Scratch &sc = r->scratch[(size_t)t];
        int64_t ex = 0;
        for (;;) {
          const int64_t j = next.fetch_add(1);
          if (j >= n_host) break;
          const size_t k = list ? (*list)[(size_t)j] : (size_t)j;
          const int64_t q = active[k];
          const int64_t ex0 = ex;
          found[k] = one_search(sc, q, ex) ? 1 : 0;
          if (!found[k] && ex - ex0 >= kComponentTrigger) walked_in_vain.store(true, std::memory_order_relaxed);
          if (hist) { hist[k] = ex - ex0; for (int i_ = 0; i_ < 4; i_++) hist_f[k * 4 + i_] = sc.trace_f[i_]; }
        }
        expanded += ex;
      };
      on_threads((int)std::min<int64_t>(T, n_host), worker);
    };
    if (!on_device) host_search(&todo);
    else {
      // One team of host threads for the whole shared round: first the host's own share (the searches expected to be longest), then --
      // while the kernel is still running -- whatever it hands back, the MOMENT it does: the kernel sets a word per query in pinned
      // memory when a search exceeds its budget (or finds no larger table, ...), a thread with nothing else to do polls those
      // words and the stream, and feeds the others.  The longest searches of a round -- which bound the launch when a wave has to
      // finish them at a tenth of a core's pace -- are thus finished by cores while the waves work through the rest, and the budget
      // can be small.  Unreachable goals among them are weeded out by component labels computed here on the host (the device's
      // stream is busy), once, when the first search comes back.
      const auto t0 = std::chrono::steady_clock::now();
      const int dev_id = tr_device(r->ctx);
      const uint32_t *flags = (r->ds.handback_cap >= (int64_t)dev_list.size()) ? r->ds.h_handback : nullptr;
      const int64_t n_dev = (int64_t)dev_list.size(), n_share = (int64_t)host_list.size();
      std::vector<uint8_t> handled(active.size(), 0), seen((size_t)n_dev, 0);
      std::vector<size_t> feed((size_t)n_dev);
      std::atomic<int64_t> feed_tail{0}, feed_head{0}, next_share{0}, share_left{n_share};
      std::atomic<bool> closed{flags == nullptr};
      std::mutex poll_mu;
      bool labels_host = labels_now;                              // (poller only)
      int64_t n_streamed = 0, n_cut_host = 0, n_over_budget = 0;   // (poller only)
      std::chrono::steady_clock::time_point t_share_done = t0, t_kernel_done = t0;
      std::atomic<bool> kernel_done{false};
      auto process = [&](int t, size_t k, int64_t &ex) {
        Scratch &sc = r->scratch[(size_t)t];
        const int64_t q = active[k];
        const int64_t ex0 = ex;
        found[k] = one_search(sc, q, ex) ? 1 : 0;
        if (!found[k] && ex - ex0 >= kComponentTrigger) walked_in_vain.store(true, std::memory_order_relaxed);
        if (hist) { hist[k] = ex - ex0; for (int i_ = 0; i_ < 4; i_++) hist_f[k * 4 + i_] = sc.trace_f[i_]; }
      };
      auto poll = [&]() {                                          // (under poll_mu)
        const bool done = hipStreamQuery(nullptr) == hipSuccess;   // read BEFORE the words: one set before the kernel ended is then seen below
        if (done && !kernel_done.load()) { t_kernel_done = std::chrono::steady_clock::now(); kernel_done.store(true); }
        int64_t t = feed_tail.load(std::memory_order_relaxed);
        for (int64_t j = 0; j < n_dev; j++) {
          const uint32_t why_ = seen[(size_t)j] ? 0u : __atomic_load_n(&flags[j], __ATOMIC_ACQUIRE);
          if (!why_) continue;
          seen[(size_t)j] = 1;
          if (why_ == 1u) n_over_budget++;
          const size_t k = dev_list[(size_t)j];
          handled[k] = 1;
          n_streamed++;
          if (!labels_host && cmode != 0) { host_component_labels(r); labels_host = true; }
          if (labels_host) {
            const int64_t q = active[k];
            if (r->dc.label[(size_t)starts[q]] != r->dc.label[(size_t)goals[q]]) { found[k] = 0; n_cut_host++; continue; }
          }
          feed[(size_t)t++] = k;
        }
        feed_tail.store(t, std::memory_order_release);
        if (done) closed.store(true, std::memory_order_release);
      };
      auto member = [&](int t) {
        (void)hipSetDevice(dev_id);
        int64_t ex = 0;
        for (;;) {
          const int64_t j = next_share.load(std::memory_order_relaxed) < n_share ? next_share.fetch_add(1) : n_share;
          if (j < n_share) {
            process(t, host_list[(size_t)j], ex);
            if (share_left.fetch_sub(1) == 1) t_share_done = std::chrono::steady_clock::now();
            continue;
          }
          int64_t h = feed_head.load(std::memory_order_relaxed);
          if (h < feed_tail.load(std::memory_order_acquire)) {
            if (feed_head.compare_exchange_weak(h, h + 1)) process(t, feed[(size_t)h], ex);
            continue;
          }
          if (closed.load(std::memory_order_acquire)) {
            if (feed_head.load() < feed_tail.load(std::memory_order_acquire)) continue;
            break;
          }
          if (poll_mu.try_lock()) { poll(); poll_mu.unlock(); }
          std::this_thread::sleep_for(std::chrono::microseconds(25));
        }
        expanded += ex;
      };
      on_threads(std::max(1, T), member);
      if (n_cut_host) { r->dc.st_cut += n_cut_host; r->dc.wanted = true; }
      const auto t1 = std::chrono::steady_clock::now();
      int64_t ex = 0;
      device_search_collect(r, active, dev_list, found, paths, paths_e, redo, ex, T, &handled);
      expanded += ex;
      if (!kernel_done.load()) t_kernel_done = std::chrono::steady_clock::now();     // (no flags to poll: collect waited for it)
      const auto t2 = std::chrono::steady_clock::now();
      // what is left (a path that did not fit its buffer; everything, without the pinned words): as before, after the kernel
      if (!redo.empty() && !labels_now && !labels_host && ensure_labels()) labels_host = true;
      if (!redo.empty() && (labels_now || labels_host)) {
        const int32_t *lab = r->dc.label.data();
        std::vector<size_t> keep;
        for (size_t k : redo) {
          const int64_t q = active[k];
          if (lab[starts[q]] == lab[goals[q]]) keep.push_back(k);
          else found[k] = 0;
        }
        r->dc.st_cut += (int64_t)(redo.size() - keep.size());
        if (keep.size() < redo.size()) r->dc.wanted = true;
        redo.swap(keep);
      }
      host_search(&redo);
      const auto t3 = std::chrono::steady_clock::now();
      const auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
      // The host's share follows the clock (answers do not depend on it): halved when the host's own share outlasted the kernel,
      // raised when it was done in a fraction of the kernel's span and nothing was left to do after it.
      const double t_kernel = ms(t_round, t_kernel_done), t_after = std::max(0.0, ms(t_kernel_done, t3)), t_share = n_share ? ms(t0, t_share_done) : 0.0;
      r->ds.kernel_ms = t_kernel; r->ds.host_after_ms = t_after;
      // (only the searches that came back OVER THE BUDGET count: one that found no table left for its records says nothing about the budget --
      // at 6 x 10^5 vertices those alone are 2 % of a round, and doubling on them let single searches run 600 ms on their wave)
      if (smode != 2 && !r->ds.budget_from_env && r->ds.budget > 0 && (flags ? n_over_budget : n_streamed + (int64_t)redo.size()) * 50 > n_dev && r->ds.budget < 16 * r->V) r->ds.budget *= 2;
      if (smode != 2 && !std::getenv("TENDON_HIP_SEARCH_HOST_SHARE") && n_share > 0) {
        if (t_share > t_kernel) r->ds.share = std::max(0.0025, r->ds.share * 0.5);
        else if (t_share < 0.4 * t_kernel && t_after < 0.1 * t_kernel) r->ds.share = std::min(0.08, r->ds.share * 1.5);
      }
      if (std::getenv("TENDON_HIP_SEARCH_STATS"))
        std::fprintf(stderr, "[tendon_hip] round %lld: order + launch %.2f ms; kernel done at %.2f ms (%zu searches); host threads: own share %zu done at %.2f ms, %lld handed back meanwhile (%lld answered by labels), all done at %.2f ms; collect %.2f ms, %zu afterwards %.2f ms; next budget %lld, share %.4f\n",
                     (long long)r->st_rounds, ms(t_round, t0), t_kernel, dev_list.size(), host_list.size(), ms(t_round, t_share_done), (long long)n_streamed,
                     (long long)n_cut_host, ms(t_round, t1), ms(t1, t2), redo.size(), ms(t2, t3), (long long)r->ds.budget, r->ds.share);
    }
    if (walked_in_vain.load()) r->dc.wanted = true;
    if (hist) {
      std::vector<int64_t> f, nf;
      for (size_t k = 0; k < active.size(); k++) (found[k] ? f : nf).push_back(hist[k]);
      auto show = [](const char *name, std::vector<int64_t> &v) {
        if (v.empty()) { std::fprintf(stderr, "  %s: none\n", name); return; }
        std::sort(v.begin(), v.end());
        int64_t sum = 0; for (int64_t x : v) sum += x;
        std::fprintf(stderr, "  %s: %zu searches, %lld expansions; median %lld, 90%% %lld, 99%% %lld, max %lld\n", name, v.size(), (long long)sum,
                     (long long)v[v.size() / 2], (long long)v[v.size() * 9 / 10], (long long)v[v.size() * 99 / 100], (long long)v.back());
      };
      std::fprintf(stderr, "[tendon_hip] round %lld:\n", (long long)r->st_rounds);
      show("found", f); show("not found", nf);
      const char *hp = std::getenv("TENDON_HIP_SEARCH_HIST");
      if (hp && hp[0] == '/') {                               // (a path: one line per search -- expansions against what could predict them)
        if (FILE *fh = std::fopen(hp, "a")) {
          const int L = r->lm_n > 0 ? r->lm_n : 0;
          for (size_t k = 0; k < active.size(); k++) {
            const int64_t q = active[k];
            const int32_t s_ = starts[q], g_ = goals[q];
            double lb = 0.0, sum_s = 1e300;
            for (int l = 0; l < L; l++) {
              const double a_ = r->lm_d[(size_t)s_ * L + l], b_ = r->lm_d[(size_t)g_ * L + l];
              lb = std::max(lb, std::fabs(a_ - b_)); sum_s = std::min(sum_s, a_ + b_);
            }
            std::fprintf(fh, "%lld %lld %d %lld %.6g %.6g %.6g %d %d %.6g %.6g %.6g %.6g\n", (long long)r->st_rounds, (long long)q, (int)found[k], (long long)hist[k],
                         state_distance(r, &r->states[(size_t)s_ * r->S], &r->states[(size_t)g_ * r->S]), lb, sum_s,
                         (int)(r->adj_off[(size_t)s_ + 1] - r->adj_off[(size_t)s_]), (int)(r->adj_off[(size_t)g_ + 1] - r->adj_off[(size_t)g_]),
                         hist_f[k * 4], hist_f[k * 4 + 1], hist_f[k * 4 + 2], hist_f[k * 4 + 3]);
          }
          std::fclose(fh);
        }
      }
    }
    laps.lap("searches");
    r->st_astar_runs += (int64_t)todo.size();
    r->st_expanded += expanded.load();
    // unknown items on the candidate paths: all interior vertices, and the edges of paths without an unknown vertex
    // are only worth testing once the vertices are clean -- but testing them in the same launch costs nothing, saves
    // a round, and removing more invalid items never changes an accepted path (see the header comment)
    list.clear();
    // (a large round from unknown validity: when the candidate paths hold a quarter as many items as there are cached sets, listing
    // the unknown ones, sending the list and fetching the verdicts costs several times the one launch that tests EVERY cached set
    // -- 0.25 ms at 6.8 x 10^5 sets -- so everything is tested at once; a count again, not a clock.  TENDON_HIP_LAZY_ONLY=1: never)
    bool all_tested = false;
    if (!went_eager && r->has_caches && !std::getenv("TENDON_HIP_LAZY_ONLY")) {
      int64_t on_paths = 0;
      for (size_t k = 0; k < active.size(); k++)
        if (found[k]) on_paths += (int64_t)(paths[(size_t)active[k]].size() + paths_e[(size_t)active[k]].size());
      if (on_paths * 4 >= r->V + r->E) {
        int64_t unknown = 0;
        for (uint8_t x : r->vstat) unknown += x == V_UNKNOWN;
        for (uint8_t x : r->estat) unknown += x == V_UNKNOWN;
        if ((rc = revalidate_locked(r, nullptr, nullptr))) return rc;
        r->st_items_checked += unknown;
        went_eager = true; all_tested = true;
      }
    }
    const auto t_items0 = std::chrono::steady_clock::now();
    // (large rounds: by ranges of queries on the host threads; an item goes to the list of the thread that marks it first -- the set
    // is the same whoever that is, and the order of the list decides nothing)
    const int Tb = active.size() >= 2048 ? std::min(T, 16) : 1;
    if (!all_tested) {
      std::vector<std::vector<int32_t>> part((size_t)Tb);
      on_threads(Tb, [&](int t) {
        std::vector<int32_t> &mine = part[(size_t)t];
        const size_t k0 = active.size() * (size_t)t / (size_t)Tb, k1 = active.size() * (size_t)(t + 1) / (size_t)Tb;
        for (size_t k = k0; k < k1; k++) {
          if (!found[k]) continue;
          const int64_t q = active[k];
          const auto &pv = paths[(size_t)q];
          const auto &pe = paths_e[(size_t)q];
          for (size_t i = 1; i + 1 < pv.size(); i++) {
            const int32_t v = pv[i];
            if (r->vstat[(size_t)v] == V_UNKNOWN && !__atomic_exchange_n(&vmark[(size_t)v], (uint8_t)1, __ATOMIC_RELAXED)) mine.push_back(v);
          }
          for (int32_t e : pe)
            if (r->estat[(size_t)e] == V_UNKNOWN && !__atomic_exchange_n(&emark[(size_t)e], (uint8_t)1, __ATOMIC_RELAXED)) mine.push_back((int32_t)(r->V + e));
        }
      });
      for (const auto &p : part) list.insert(list.end(), p.begin(), p.end());
    }
    const auto t_items1 = std::chrono::steady_clock::now();
    if (!list.empty()) {
      if ((rc = check_items(r, list, hit))) return rc;
      record(list, hit);
      for (int32_t it : list) { if (it < r->V) vmark[(size_t)it] = 0; else emark[(size_t)(it - r->V)] = 0; }
    }
    const auto t_items2 = std::chrono::steady_clock::now();
    std::vector<int64_t> still;
    {
      std::vector<std::vector<int64_t>> part((size_t)Tb);
      on_threads(Tb, [&](int t) {
        const size_t k0 = active.size() * (size_t)t / (size_t)Tb, k1 = active.size() * (size_t)(t + 1) / (size_t)Tb;
        for (size_t k = k0; k < k1; k++) {
          const int64_t q = active[k];
          if (!found[k]) { status[q] = TR_QUERY_NO_PATH; paths[(size_t)q].clear(); continue; }   // different components (:2026-2036)
          bool ok = true;
          for (size_t i = 1; i + 1 < paths[(size_t)q].size() && ok; i++) ok = r->vstat[(size_t)paths[(size_t)q][i]] == V_VALID;
          for (size_t i = 0; i < paths_e[(size_t)q].size() && ok; i++) ok = r->estat[(size_t)paths_e[(size_t)q][i]] == V_VALID;
          if (ok) {
            if (cost) { double c = 0; for (size_t i = paths_e[(size_t)q].size(); i-- > 0;) c += r->w[(size_t)paths_e[(size_t)q][i]]; cost[q] = c; }
          } else part[(size_t)t].push_back(q);
        }
      });
      for (const auto &p : part) still.insert(still.end(), p.begin(), p.end());    // (ranges in order: the queries keep their order)
    }
    // The lazy loop exists to save validity tests; here a test of EVERY cached set is one K4 launch (0.25 ms at 6.8 x 10^5 sets),
    // while every further round costs at least its longest search (milliseconds on a core) -- and in a cluttered environment the
    // open queries find new candidate paths through untested items round after round, hundreds of rounds in the worst case.  So
    // when enough queries are still open for another round to cost more than that launch, everything is tested and the next round
    // is the last.  The rule is a function of the round's counts alone (open queries against the number of cached sets: one open
    // query per 2^17 sets, four at least) -- not of clocks: rounds, items_checked and the validity a call leaves behind are the same
    // run after run (tests/test_gpu_search.py).  Answers are those of the lazy loop (validity is a function of the environment);
    // what changes is which items end up known.  TENDON_HIP_LAZY_ONLY=1 keeps the loop lazy to the end (the reference's
    // behaviour item by item; A/B, tests).
    const int64_t eager_from = std::max<int64_t>(4, (r->V + r->E) >> 17);
    if (!went_eager && r->has_caches && (int64_t)still.size() >= eager_from && !std::getenv("TENDON_HIP_LAZY_ONLY")) {
      int64_t unknown = 0;
      for (uint8_t x : r->vstat) unknown += x == V_UNKNOWN;
      for (uint8_t x : r->estat) unknown += x == V_UNKNOWN;
      if ((rc = revalidate_locked(r, nullptr, nullptr))) return rc;
      r->st_items_checked += unknown;
      went_eager = true;
    }
    active.swap(still);
    laps.lap("items + verdicts");
  }
  for (int64_t q = 0; q < n_queries; q++) {
    const auto &pv = paths[(size_t)q];
    if (status[q] == TR_QUERY_SOLVED) r->path_v.insert(r->path_v.end(), pv.rbegin(), pv.rend());     // start ... goal
    r->path_off[(size_t)q + 1] = (int64_t)r->path_v.size();
    path_offsets[q + 1] = r->path_off[(size_t)q + 1];
  }
  laps.lap("paths out");
  if (stats) *stats = tr_roadmap_stats{r->st_rounds, r->st_items_checked, r->st_astar_runs, r->st_expanded};
  if (std::getenv("TENDON_HIP_SEARCH_STATS"))
    std::fprintf(stderr, "[tendon_hip] searches: mode %d, device state %d%s%s, %lld slots, %lld searches finished on the device, %lld handed back to the host, %lld on the host meanwhile, %lld answered by a sweep\n",
                 smode, r->ds.state, r->ds.why.empty() ? "" : " -- ", r->ds.why.c_str(), (long long)r->ds.slots, (long long)r->ds.st_queries,
                 (long long)r->ds.st_fallbacks, (long long)r->ds.st_host_share, (long long)r->ds.st_sweeps);
  return TR_OK;
}

int tr_roadmap_search_stats(tr_roadmap *r, int64_t out[8]) {
  if (!r || !out) return TR_ERR_INVALID_ARG;
  RmLock lock_(r);
  out[0] = r->ds.st_queries; out[1] = r->ds.st_fallbacks; out[2] = r->ds.st_host_share; out[3] = r->ds.st_moves;
  out[4] = r->ds.st_expanded; out[5] = r->st_expanded - r->ds.st_expanded;
  out[6] = r->dc.st_cut; out[7] = r->ds.st_grows;
  return TR_OK;
}

int tr_roadmap_profile(tr_roadmap *r, double out[4]) {
  if (!r || !out) return TR_ERR_INVALID_ARG;
  RmLock lock_(r);
  const double deg = r->V > 0 ? (double)r->adj.size() / (double)r->V : 0.0;
  const int L = r->lm_n > 0 ? r->lm_n : 0;
  out[0] = r->ds.st_kernel_ms; out[1] = (double)r->ds.st_launches; out[2] = (double)r->ds.st_expanded;
  // per expansion: the vertex's record and row header, per arc the arc, two validity bytes, the arc count, the neighbour's record read
  // and written, its state and landmark rows
  out[3] = 48.0 + deg * (16.0 + 2.0 + 1.0 + 32.0 + 8.0 * r->S + 4.0 * L + 32.0);
  return TR_OK;
}

int tr_roadmap_release_search_state(tr_roadmap *r, int64_t *bytes_released) {
  if (!r) return TR_ERR_INVALID_ARG;
  RmLock lock_(r);
  RM_HIP(r, hipSetDevice(tr_device(r->ctx)));
  const size_t b = release_search_tables(r);
  dev_cache().trim();                                       // (not parked for the next owner: back to the device)
  if (bytes_released) *bytes_released = (int64_t)b;
  return TR_OK;
}

int tr_roadmap_search_sweeps(tr_roadmap *r, int64_t *n) {
  if (!r || !n) return TR_ERR_INVALID_ARG;
  RmLock lock_(r);
  *n = r->ds.st_sweeps;
  return TR_OK;
}

int tr_roadmap_reserve_search_state(tr_roadmap *r, int64_t n_queries) {
  if (!r || n_queries < 0) return TR_ERR_INVALID_ARG;
  RmLock lock_(r);
  RM_HIP(r, hipSetDevice(tr_device(r->ctx)));
  if (n_queries == 0) return TR_OK;
  if (!search_setup(r) || !search_tables(r, n_queries)) return rfail(r, TR_ERR_UNSUPPORTED, "device searches not available for this roadmap: " + r->ds.why);
  RM_HIP(r, hipStreamSynchronize(nullptr));
  return TR_OK;
}

int tr_roadmap_search_state_bytes(tr_roadmap *r, int64_t *bytes) {
  if (!r || !bytes) return TR_ERR_INVALID_ARG;
  RmLock lock_(r);
  const auto &d = r->ds;
  *bytes = (int64_t)((d.tables ? d.table_bytes : 0) + (d.qarena ? (size_t)d.nq_cap * 17 + (size_t)d.pbuf_cap * 4 : 0));
  return TR_OK;
}

int tr_roadmap_fetch_paths(tr_roadmap *r, int32_t *path_vertices, int64_t capacity) {
  if (!r) return TR_ERR_INVALID_ARG;
  RmLock lock_(r);
  const int64_t n = (int64_t)r->path_v.size();
  if (capacity < n) return rfail(r, TR_ERR_INVALID_ARG, "capacity smaller than the stored paths");
  if (n > 0) {
    if (!path_vertices) return rfail(r, TR_ERR_INVALID_ARG, "null output");
    std::memcpy(path_vertices, r->path_v.data(), (size_t)n * sizeof(int32_t));
  }
  return TR_OK;
}

}  // extern "C"

// An allocation somewhere in the library ran out of memory (tr_dev_cache_trim): the search tables of every roadmap that is not inside
// a call right now go back to the device; their next large round allocates them again.
void release_idle_search_tables() {
  std::lock_guard<std::mutex> g(g_roadmaps_mu);
  for (tr_roadmap *r : g_roadmaps) {
    if (r->busy.load(std::memory_order_acquire)) continue;  // (possibly by this very thread: the mutex is not asked)
    if (!r->mu.try_lock()) continue;
    (void)release_search_tables(r);
    r->mu.unlock();
  }
}
