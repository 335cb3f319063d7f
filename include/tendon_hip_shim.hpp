// tendon_hip_shim.hpp -- header-only C++17 host shim over the C ABI (tendon_hip.h) that keeps the
// reference's class and method names for the hot path, so reference-side code can switch with a
// namespace alias.  STL only (the reference's Eigen / OMPL types are adapted in INTEGRATION.md).
//
//   tendon::TendonRobot / BackboneSpecs / TendonSpecs / TendonResult   tendon/*.h
//   collision::VoxelOctree (dense view)                                 collision/VoxelOctree.h:68-330
//   motion_planning::VoxelEnvironment                                   motion-planning/VoxelEnvironment.h:31-49
//   motion_planning::VoxelBackboneValidityChecker                       motion-planning/VoxelBackboneValidityChecker.h:28-58
//   motion_planning::VoxelValidityChecker                               motion-planning/VoxelValidityChecker.h:18-26
//   motion_planning::VoxelBackboneMotionValidator                       motion-planning/VoxelBackboneMotionValidator.h
//   motion_planning::VoxelBackboneDiscreteMotionValidator               motion-planning/VoxelBackboneDiscreteMotionValidator.h
//   motion_planning::VoxelCaches, voxelize_states, caches_collide       the cache loops of VoxelCachedLazyPRM.cpp
//   motion_planning::VoxelCachedLazyPRM                                  createRoadmap / precompute* (the build) and solveWithRoadmap
//                                                                        for a batch of queries (motion-planning/VoxelCachedLazyPRM.h:469-514)
//
// Error behaviour: every tr_status is rethrown as the C++ exception type the reference throws at
// the same condition (std::invalid_argument, std::out_of_range, std::domain_error,
// std::length_error, std::runtime_error).
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "tendon_hip.h"

namespace tendon_hip {

inline void check(const tr_ctx *ctx, int status) {
  if (status == TR_OK) return;
  const std::string msg = tr_last_error(ctx);
  switch (status) {
    case TR_ERR_INVALID_ARG: throw std::invalid_argument(msg);
    case TR_ERR_OUT_OF_RANGE: throw std::out_of_range(msg);
    case TR_ERR_DOMAIN: throw std::domain_error(msg);
    case TR_ERR_LENGTH: throw std::length_error(msg);
    default: throw std::runtime_error(msg);
  }
}

namespace tendon {

struct BackboneSpecs {              // tendon/BackboneSpecs.h:14-20
  double L = 0.2, dL = 0.005, ro = 0.01, ri = 0.0, E = 2.1e6, nu = 0.3;
};

struct TendonSpecs {                // tendon/TendonSpecs.h:25-30
  std::vector<double> C{0.0}, D{0.01};
  double max_tension = 20.0, min_length = -0.015, max_length = 0.035;
};

struct TendonResult {               // tendon/TendonResult.h:17-40
  std::vector<double> t;
  std::vector<std::array<double, 3>> p;
  std::vector<std::array<double, 9>> R;   // column-major, as Eigen::Matrix3d stores it
  double L = 0.0;
  std::vector<double> L_i;
  bool converged = true;
};

class TendonRobot {                 // tendon/TendonRobot.h:52-355
 public:
  double r = 0.015;
  BackboneSpecs specs{};
  std::vector<TendonSpecs> tendons;
  bool enable_rotation = false, enable_retraction = false;
  double residual_threshold = 5e-6;

  TendonRobot() = default;
  /// value semantics as in the reference; a copy gets its own GPU context on first use
  TendonRobot(const TendonRobot &o)
      : r(o.r), specs(o.specs), tendons(o.tendons), enable_rotation(o.enable_rotation), enable_retraction(o.enable_retraction),
        residual_threshold(o.residual_threshold) {}
  TendonRobot &operator=(const TendonRobot &o) {
    r = o.r; specs = o.specs; tendons = o.tendons; enable_rotation = o.enable_rotation; enable_retraction = o.enable_retraction;
    residual_threshold = o.residual_threshold; ctx_.reset();
    return *this;
  }

  size_t state_size() const { return tendons.size() + (enable_rotation ? 1 : 0) + (enable_retraction ? 1 : 0); }

  /// The GPU context holding this robot's constants; (re)created on first use on `device`.
  tr_ctx *context(int device = 0) const {
    if (!ctx_) {
      const size_t n = tendons.size();
      if (n == 0) throw std::out_of_range("tendons and tau are not the same length");
      const size_t na = tendons[0].C.size(), nm = tendons[0].D.size();
      std::vector<double> C, D, mt, mn, mx;
      for (auto &t : tendons) {
        if (t.C.size() != na || t.D.size() != nm) throw std::invalid_argument("tendons must share C.size() and D.size()");
        C.insert(C.end(), t.C.begin(), t.C.end());
        D.insert(D.end(), t.D.begin(), t.D.end());
        mt.push_back(t.max_tension); mn.push_back(t.min_length); mx.push_back(t.max_length);
      }
      tr_robot_desc d{};
      d.r = r; d.L = specs.L; d.dL = specs.dL; d.ro = specs.ro; d.ri = specs.ri; d.E = specs.E; d.nu = specs.nu;
      d.n_tendons = (int32_t)n; d.n_a = (int32_t)na; d.n_m = (int32_t)nm;
      d.C = C.data(); d.D = D.data(); d.max_tension = mt.data(); d.min_length = mn.data(); d.max_length = mx.data();
      d.enable_rotation = enable_rotation; d.enable_retraction = enable_retraction;
      d.residual_threshold = residual_threshold;
      tr_ctx *c = nullptr;
      check(nullptr, tr_create(&d, device, &c));
      ctx_ = std::shared_ptr<tr_ctx>(c, tr_destroy);
    }
    return ctx_.get();
  }

  /// TendonRobot::shape(state), TendonRobot.h:105-115
  TendonResult shape(const std::vector<double> &state) const {
    if (state.size() != state_size()) throw std::invalid_argument("State is not the right size");
    return std::move(shape_batch(state, 1)[0]);
  }

  /// forward_kinematics(state), TendonRobot.h:68-72
  std::vector<std::array<double, 3>> forward_kinematics(const std::vector<double> &state) const { return shape(state).p; }

  /// n shapes in one launch (the omp loop of apps/estimate_length_discretization.cpp:62-71)
  std::vector<TendonResult> shape_batch(const std::vector<double> &states, size_t n) const {
    tr_ctx *c = context();
    const size_t S = state_size(), P = (size_t)tr_num_points(c), N = tendons.size();
    if (states.size() != n * S) throw std::invalid_argument("State is not the right size");
    std::vector<double> p(n * P * 3), R(n * P * 9), L(n), Li(n * N);
    std::vector<uint8_t> conv(n);
    std::vector<int32_t> np(n);
    check(c, tr_fk_batch(c, states.data(), (int64_t)n, p.data(), R.data(), L.data(), Li.data(), conv.data(), np.data()));
    const std::vector<double> tg = t_grid();
    std::vector<TendonResult> out(n);
    for (size_t i = 0; i < n; i++) {
      TendonResult &res = out[i];
      const size_t m = (size_t)np[i];
      res.t.assign(tg.begin(), tg.begin() + m);
      res.p.resize(m); res.R.resize(m);
      for (size_t j = 0; j < m; j++) {
        for (int k = 0; k < 3; k++) res.p[j][k] = p[(i * P + j) * 3 + k];
        for (int k = 0; k < 9; k++) res.R[j][k] = R[(i * P + j) * 9 + k];
      }
      res.L = L[i];
      res.L_i.assign(Li.begin() + i * N, Li.begin() + (i + 1) * N);
      res.converged = conv[i] != 0;
    }
    return out;
  }

  /// home_shape(0).L_i, TendonRobot.cpp:249-314
  std::vector<double> home_lengths() const {
    std::vector<double> Li(tendons.size());
    check(context(), tr_home_lengths(context(), Li.data()));
    return Li;
  }

  std::vector<double> calc_dl(const std::vector<double> &home_l, const std::vector<double> &other_l) const {   // :247-259
    if (home_l.size() != other_l.size()) throw std::out_of_range("vector size mismatch");
    std::vector<double> dl(home_l.size());
    for (size_t i = 0; i < dl.size(); i++) dl[i] = home_l[i] - other_l[i];
    return dl;
  }
  bool is_within_length_limits(const std::vector<double> &dl) const {                                          // :268-278
    if (dl.size() != tendons.size()) throw std::out_of_range("length mismatch");
    for (size_t i = 0; i < dl.size(); i++)
      if (dl[i] < tendons[i].min_length || tendons[i].max_length < dl[i]) return false;
    return true;
  }

  /// util::range + t_range (TendonRobot.cpp:69-84) for s_start = 0: only to fill TendonResult::t
  std::vector<double> t_grid() const {
    std::vector<double> v;
    for (double q = 0.0; q <= specs.L - specs.dL / 2; q += specs.dL) v.push_back(q);
    v.push_back(specs.L);
    std::vector<double> t(v.size());
    for (size_t i = 0; i < v.size(); i++) t[v.size() - 1 - i] = specs.L - (v[i] - 0.0);
    return t;
  }

 private:
  mutable std::shared_ptr<tr_ctx> ctx_;
};

}  // namespace tendon

namespace collision {

/// Dense view of collision::VoxelOctree: what the engine consumes as the obstacle set.
class VoxelOctree {
 public:
  explicit VoxelOctree(size_t Ndim = 4) : N_(Ndim) {
    if (Ndim < 4 || Ndim > 512 || (Ndim & (Ndim - 1)))
      throw std::invalid_argument("unsupported voxel dimension: " + std::to_string(Ndim));     // VoxelOctree.cpp:113-115
    blocks_.assign((Ndim / 4) * (Ndim / 4) * (Ndim / 4), 0);
  }
  size_t Nx() const { return N_; }
  size_t Nbx() const { return N_ / 4; }
  void set_xlim(double lo, double hi) { set(0, lo, hi, "x"); }
  void set_ylim(double lo, double hi) { set(1, lo, hi, "y"); }
  void set_zlim(double lo, double hi) { set(2, lo, hi, "z"); }
  double dx() const { return (lim_[1] - lim_[0]) / N_; }
  double dy() const { return (lim_[3] - lim_[2]) / N_; }
  double dz() const { return (lim_[5] - lim_[4]) / N_; }
  const double *limits() const { return lim_; }
  static uint64_t bitmask(unsigned x, unsigned y, unsigned z) { return uint64_t(1) << (x * 16 + y * 4 + z); }   // :1501-1503
  uint64_t block(size_t bx, size_t by, size_t bz) const { return blocks_[(bx * Nbx() + by) * Nbx() + bz]; }
  void set_block(size_t bx, size_t by, size_t bz, uint64_t v) { blocks_[(bx * Nbx() + by) * Nbx() + bz] = v; }
  bool cell(size_t ix, size_t iy, size_t iz) const { return block(ix / 4, iy / 4, iz / 4) & bitmask(ix % 4, iy % 4, iz % 4); }
  bool set_cell(size_t ix, size_t iy, size_t iz) {                                                               // :256-265
    uint64_t &b = blocks_[((ix / 4) * Nbx() + iy / 4) * Nbx() + iz / 4];
    const uint64_t m = bitmask(ix % 4, iy % 4, iz % 4), old = b;
    b |= m;
    return old & m;
  }
  const std::vector<uint64_t> &blocks() const { return blocks_; }
  std::vector<uint64_t> &blocks() { return blocks_; }

 private:
  void set(int a, double lo, double hi, const char *name) {
    if (lo >= hi) throw std::length_error(std::string(name) + "limits must be positive in size");              // :152-177
    lim_[2 * a] = lo; lim_[2 * a + 1] = hi;
  }
  size_t N_;
  double lim_[6] = {0, 1, 0, 1, 0, 1};
  std::vector<uint64_t> blocks_;
};

}  // namespace collision

namespace motion_planning {

struct VoxelEnvironment {           // motion-planning/VoxelEnvironment.h:46-49 (fields the hot path reads)
  double inv_rotation[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};   // row-major
};

/// VoxelBackboneValidityChecker: isValid(state) and its batched form.
class VoxelBackboneValidityChecker {
 public:
  VoxelBackboneValidityChecker(const tendon::TendonRobot &robot, const VoxelEnvironment &venv,
                               const collision::VoxelOctree &voxels, int device = 0, int checker = TR_CHECKER_BACKBONE)
      : robot_(robot), ctx_(robot.context(device)) {
    check(ctx_, tr_set_checker(ctx_, checker));
    // throws std::invalid_argument when robot.specs.dL exceeds the largest voxel edge (VoxelBackboneValidityChecker.h:37-45)
    check(ctx_, tr_set_grid(ctx_, (uint32_t)voxels.Nx(), voxels.limits(), voxels.blocks().data(), venv.inv_rotation));
  }
  const tendon::TendonRobot &robot() const { return robot_; }

  /// AbstractValidityChecker::isValid on a robot state (AbstractValidityChecker.cpp:124-133)
  bool isValid(const std::vector<double> &robot_state) const {
    if (robot_state.size() != robot_.state_size()) throw std::invalid_argument("State is not the right size");
    uint64_t bits = 0;
    check(ctx_, tr_validate_batch(ctx_, robot_state.data(), 1, &bits, nullptr, nullptr));
    return bits & 1u;
  }

  /// n states in one K1 + K2 pass; tips (optional) receive fk_shape.p.back()
  std::vector<bool> isValidBatch(const std::vector<double> &states, size_t n, std::vector<double> *tips = nullptr,
                                 std::vector<uint8_t> *flags = nullptr) const {
    if (states.size() != n * robot_.state_size()) throw std::invalid_argument("State is not the right size");
    std::vector<uint64_t> bits((n + 63) / 64);
    if (tips) tips->resize(3 * n);
    if (flags) flags->resize(n);
    check(ctx_, tr_validate_batch(ctx_, states.data(), (int64_t)n, bits.data(), tips ? tips->data() : nullptr,
                                  flags ? flags->data() : nullptr));
    std::vector<bool> out(n);
    for (size_t i = 0; i < n; i++) out[i] = (bits[i >> 6] >> (i & 63)) & 1u;
    return out;
  }
  tr_ctx *context() const { return ctx_; }

  // ---- edits of the obstacle set where it lives (collision::VoxelOctree's add_sphere / dilate* /
  // remove_interior, VoxelOctree.cpp:434-469, :533-952, as apps/prepare_voxel_env.cpp:269-315 applies them) ----
  /// VoxelOctree::add_capsule on the resident obstacle set (collision/VoxelOctree.cpp:471-515)
  void add_capsules(const std::vector<double> &capsules /* n x (ax, ay, az, bx, by, bz, r) */) const {
    check(ctx_, tr_grid_add_capsules(ctx_, capsules.data(), (int64_t)(capsules.size() / 7)));
  }
  void add_spheres(const std::vector<double> &spheres /* n x (cx, cy, cz, r) */) const {
    check(ctx_, tr_grid_add_spheres(ctx_, spheres.data(), (int64_t)(spheres.size() / 4)));
  }
  void dilate(int num = 1, bool use_diagonal = false) const { check(ctx_, tr_grid_dilate(ctx_, num, use_diagonal)); }
  void dilate_sphere(double r) const { check(ctx_, tr_grid_dilate_sphere(ctx_, r)); }
  void remove_interior(bool keep_diagonal = true) const { check(ctx_, tr_grid_remove_interior(ctx_, keep_diagonal)); }
  /// the obstacle set as it is on the device now, into a VoxelOctree of the same dimension
  void obstacles(collision::VoxelOctree &out) const { check(ctx_, tr_get_grid(ctx_, out.blocks().data())); }

  /// connectionStrategy_(v) for every state at once (VoxelCachedLazyPRM.cpp:1491-1502): the k nearest states
  /// (self included) in the compound state-space metric of Problem.cpp:112-152; idx is n x k, -1 = none in range
  void nearest_k(const std::vector<double> &states, size_t n, int k, std::vector<int32_t> &idx, std::vector<double> &dist,
                 double max_distance = 1e300) const {
    if (states.size() != n * robot_.state_size()) throw std::invalid_argument("State is not the right size");
    idx.resize(n * (size_t)k); dist.resize(n * (size_t)k);
    check(ctx_, tr_knn(ctx_, states.data(), (int64_t)n, k, max_distance, idx.data(), dist.data()));
  }

 private:
  const tendon::TendonRobot &robot_;
  tr_ctx *ctx_;
};

/// VoxelValidityChecker (motion-planning/VoxelValidityChecker.h:18-26): the same interface; the robot is
/// voxelised as a sphere of its radius at every backbone point and tested against the raw environment.
/// The GPU context belongs to the TendonRobot object: give this checker its own copy of the robot if a
/// backbone checker on the same robot is alive at the same time.
class VoxelValidityChecker : public VoxelBackboneValidityChecker {
 public:
  VoxelValidityChecker(const tendon::TendonRobot &robot, const VoxelEnvironment &venv, const collision::VoxelOctree &voxels,
                       int device = 0)
      : VoxelBackboneValidityChecker(robot, venv, voxels, device, TR_CHECKER_SPHERES) {}
};

/// Sparse voxel set as the roadmap caches store it (VoxelCachedLazyPRM.h:165-179; CSR over items).
struct VoxelCaches {
  std::vector<int64_t> offsets{0};          // [items + 1]
  std::vector<uint32_t> block_ids;          // ((bx * Nb) + by) * Nb + bz
  std::vector<uint64_t> masks;              // bit x*16 + y*4 + z
  std::vector<bool> usable;                 // shape-valid vertex / fully valid edge
  size_t items() const { return offsets.size() - 1; }
};

namespace detail {
inline std::vector<bool> unpack(const std::vector<uint64_t> &bits, size_t n) {
  std::vector<bool> out(n);
  for (size_t i = 0; i < n; i++) out[i] = (bits[i >> 6] >> (i & 63)) & 1u;
  return out;
}
inline void fetch(tr_ctx *c, VoxelCaches &vc) {
  const size_t nnz = (size_t)vc.offsets.back();
  vc.block_ids.resize(nnz); vc.masks.resize(nnz);
  check(c, tr_voxelize_fetch(c, vc.block_ids.data(), vc.masks.data(), (int64_t)nnz));
}
}  // namespace detail

/// AbstractVoxelValidityChecker::voxelize for a batch of states (VoxelCachedLazyPRM.cpp:2816-2823):
/// the backbone voxel set of every shape-valid state, plus fk_shape.p.back() as the vertex tip.
inline VoxelCaches voxelize_states(const VoxelBackboneValidityChecker &vc, const std::vector<double> &states, size_t n,
                                   std::vector<double> *tips = nullptr) {
  if (states.size() != n * vc.robot().state_size()) throw std::invalid_argument("State is not the right size");
  VoxelCaches out;
  out.offsets.assign(n + 1, 0);
  std::vector<uint64_t> bits((n + 63) / 64);
  if (tips) tips->resize(3 * n);
  check(vc.context(), tr_voxelize_batch(vc.context(), states.data(), (int64_t)n, out.offsets.data(), bits.data(),
                                        tips ? tips->data() : nullptr));
  out.usable = detail::unpack(bits, n);
  detail::fetch(vc.context(), out);
  return out;
}

/// obstacles.collides(*cached_voxels) for every cached item against the checker's current grid
/// (VoxelCachedLazyPRM.cpp:2397-2411, :2497-2509).
inline std::vector<bool> caches_collide(const VoxelBackboneValidityChecker &vc, const VoxelCaches &caches) {
  const size_t n = caches.items();
  std::vector<uint64_t> bits((n + 63) / 64);
  check(vc.context(), tr_check_cached(vc.context(), caches.block_ids.data(), caches.masks.data(), caches.offsets.data(),
                                      (int64_t)n, bits.data()));
  return detail::unpack(bits, n);
}

/// VoxelBackboneMotionValidator: checkMotion(s1, s2), checkMotion(s1, s2, last_valid), voxelize(a, b)
/// and their batched forms.
class VoxelBackboneMotionValidator {
 public:
  explicit VoxelBackboneMotionValidator(const VoxelBackboneValidityChecker &vc) : vc_(vc) {}
  virtual ~VoxelBackboneMotionValidator() = default;
  tr_space_params space{0.02, 0.01, 0.0001};      // Problem.h:59-62

  bool checkMotion(const std::vector<double> &a, const std::vector<double> &b) const {   // AbstractVoxelMotionValidator.h:143-151
    return checkMotionBatch(a, b, 1)[0];
  }
  /// AbstractVoxelMotionValidator.h:153-169: last_valid = (interpolate(a, b, t), t) with t = PartialVoxelization::t
  bool checkMotion(const std::vector<double> &a, const std::vector<double> &b,
                   std::pair<std::vector<double>, double> &last_valid) const {
    std::vector<double> t;
    const bool ok = checkMotionBatch(a, b, 1, nullptr, &t)[0];
    last_valid.second = t[0];
    last_valid.first = interpolate(a, b, t[0]);
    return ok;
  }
  std::vector<bool> checkMotionBatch(const std::vector<double> &a, const std::vector<double> &b, size_t n,
                                     std::vector<int32_t> *n_fk = nullptr, std::vector<double> *last_valid_t = nullptr) const {
    const size_t S = vc_.robot().state_size();
    if (a.size() != n * S || b.size() != n * S) throw std::invalid_argument("start and end are different sizes");
    std::vector<uint64_t> bits((n + 63) / 64);
    if (n_fk) n_fk->resize(n);
    if (last_valid_t) last_valid_t->resize(n);
    check(vc_.context(), run(a.data(), b.data(), (int64_t)n, bits.data(), n_fk ? n_fk->data() : nullptr,
                             last_valid_t ? last_valid_t->data() : nullptr));
    return detail::unpack(bits, n);
  }
  /// Roadmap form (the loops of VoxelCachedLazyPRM.cpp:1520-1542, :1621-1641): edge e joins rows edges[2e] and
  /// edges[2e + 1] of `states`; every vertex is evaluated once for all of its edges.
  std::vector<bool> checkMotionIndexed(const std::vector<double> &states, size_t n_states, const std::vector<int32_t> &edges,
                                       std::vector<int32_t> *n_fk = nullptr) const {
    if (states.size() != n_states * vc_.robot().state_size()) throw std::invalid_argument("State is not the right size");
    const size_t n = edges.size() / 2;
    std::vector<uint64_t> bits((n + 63) / 64);
    if (n_fk) n_fk->resize(n);
    check(vc_.context(), tr_validate_edges_indexed(vc_.context(), &space, states.data(), (int64_t)n_states, edges.data(), (int64_t)n,
                                                   bits.data(), n_fk ? n_fk->data() : nullptr, nullptr));
    return detail::unpack(bits, n);
  }
  /// AbstractVoxelMotionValidator::voxelize(a, b) for a batch of edges (VoxelCachedLazyPRM.cpp:2890-2898):
  /// the swept voxel set of every fully valid edge.
  VoxelCaches voxelizeBatch(const std::vector<double> &a, const std::vector<double> &b, size_t n) const {
    const size_t S = vc_.robot().state_size();
    if (a.size() != n * S || b.size() != n * S) throw std::invalid_argument("start and end are different sizes");
    VoxelCaches out;
    out.offsets.assign(n + 1, 0);
    std::vector<uint64_t> bits((n + 63) / 64);
    check(vc_.context(), tr_voxelize_edges(vc_.context(), &space, a.data(), b.data(), (int64_t)n, out.offsets.data(), bits.data(), nullptr));
    out.usable = detail::unpack(bits, n);
    detail::fetch(vc_.context(), out);
    return out;
  }
  /// The same for roadmap edges given as index pairs into one vertex array (VoxelCachedLazyPRM.cpp:1751-1775): every
  /// vertex is integrated and voxelised once for all of its edges.
  /// `validate`: also checkMotion on the same samples (tr_connect_edges_indexed): `usable` is then checkMotion's verdict and
  /// only accepted edges own a voxel set -- createRoadmap's connectVertices + voxelizeEdge in one traversal.
  VoxelCaches voxelizeIndexed(const std::vector<double> &states, size_t n_states, const std::vector<int32_t> &edges,
                              bool validate = false) const {
    if (states.size() != n_states * vc_.robot().state_size()) throw std::invalid_argument("State is not the right size");
    const size_t n = edges.size() / 2;
    VoxelCaches out;
    out.offsets.assign(n + 1, 0);
    std::vector<uint64_t> bits((n + 63) / 64);
    check(vc_.context(), (validate ? tr_connect_edges_indexed : tr_voxelize_edges_indexed)(
                             vc_.context(), &space, states.data(), (int64_t)n_states, edges.data(), (int64_t)n,
                             out.offsets.data(), bits.data(), nullptr));
    out.usable = detail::unpack(bits, n);
    detail::fetch(vc_.context(), out);
    return out;
  }
  /// CompoundStateSpace::interpolate as wired by Problem.cpp:101-163: linear, shortest arc on the SO2 rotation
  std::vector<double> interpolate(const std::vector<double> &a, const std::vector<double> &b, double t) const {
    const auto &rb = vc_.robot();
    const size_t N = rb.tendons.size();
    std::vector<double> out(a.size());
    for (size_t i = 0; i < a.size(); i++) out[i] = a[i] + (b[i] - a[i]) * t;
    if (rb.enable_rotation) {
      const double kPi = 3.14159265358979323846;
      double diff = b[N] - a[N];
      if (std::fabs(diff) > kPi) {
        diff = diff > 0.0 ? 2.0 * kPi - diff : -2.0 * kPi - diff;
        double v = a[N] - diff * t;
        if (v > kPi) v -= 2.0 * kPi; else if (v < -kPi) v += 2.0 * kPi;
        out[N] = v;
      }
    }
    return out;
  }

 protected:
  virtual int run(const double *a, const double *b, int64_t n, uint64_t *bits, int32_t *n_fk, double *t) const {
    if (t) return tr_validate_edges_last_valid(vc_.context(), &space, a, b, n, bits, t, n_fk);
    return tr_validate_edges(vc_.context(), &space, a, b, n, bits, n_fk, nullptr);
  }
  const VoxelBackboneValidityChecker &vc_;
};

/// VoxelBackboneDiscreteMotionValidator (motion-planning/VoxelBackboneDiscreteMotionValidator.cpp:9-79):
/// same interface, samples at a, i / validSegmentCount, b.
class VoxelBackboneDiscreteMotionValidator : public VoxelBackboneMotionValidator {
 public:
  using VoxelBackboneMotionValidator::VoxelBackboneMotionValidator;

 protected:
  int run(const double *a, const double *b, int64_t n, uint64_t *bits, int32_t *n_fk, double *t) const override {
    return tr_validate_edges_discrete(vc_.context(), &space, a, b, n, bits, t, n_fk);
  }
};

namespace detail {
inline std::vector<uint64_t> pack(const std::vector<bool> &b) {
  std::vector<uint64_t> w((b.size() + 63) / 64, 0);
  for (size_t i = 0; i < b.size(); i++) if (b[i]) w[i >> 6] |= (uint64_t)1 << (i & 63);
  return w;
}
/// n items without a cache
inline VoxelCaches empty_items(size_t n) {
  VoxelCaches c;
  c.offsets.assign(n + 1, 0);
  c.usable.assign(n, false);
  return c;
}
/// out item i = item keep[i] of c
inline VoxelCaches select_items(const VoxelCaches &c, const std::vector<size_t> &keep) {
  VoxelCaches out;
  out.offsets.assign(keep.size() + 1, 0);
  out.usable.resize(keep.size());
  for (size_t i = 0; i < keep.size(); i++) out.offsets[i + 1] = out.offsets[i] + (c.offsets[keep[i] + 1] - c.offsets[keep[i]]);
  out.block_ids.resize((size_t)out.offsets.back()); out.masks.resize((size_t)out.offsets.back());
  for (size_t i = 0; i < keep.size(); i++) {
    const size_t a = (size_t)c.offsets[keep[i]], b = (size_t)c.offsets[keep[i] + 1], o = (size_t)out.offsets[i];
    std::copy(c.block_ids.begin() + a, c.block_ids.begin() + b, out.block_ids.begin() + o);
    std::copy(c.masks.begin() + a, c.masks.begin() + b, out.masks.begin() + o);
    out.usable[i] = c.usable[keep[i]];
  }
  return out;
}
/// b's items after a's
inline void append_items(VoxelCaches &a, const VoxelCaches &b) {
  const int64_t base = a.offsets.back();
  for (size_t i = 0; i < b.items(); i++) a.offsets.push_back(base + b.offsets[i + 1]);
  a.block_ids.insert(a.block_ids.end(), b.block_ids.begin(), b.block_ids.begin() + b.offsets.back());
  a.masks.insert(a.masks.end(), b.masks.begin(), b.masks.begin() + b.offsets.back());
  a.usable.insert(a.usable.end(), b.usable.begin(), b.usable.end());
}
/// item idx[i] of dst becomes item i of src (idx ascending)
inline void replace_items(VoxelCaches &dst, const std::vector<size_t> &idx, const VoxelCaches &src) {
  if (idx.empty()) return;
  VoxelCaches out;
  out.offsets.assign(dst.items() + 1, 0);
  out.usable = dst.usable;
  std::vector<int64_t> from(dst.items(), -1);
  for (size_t i = 0; i < idx.size(); i++) from[idx[i]] = (int64_t)i;
  for (size_t v = 0; v < dst.items(); v++) {
    const VoxelCaches &c = from[v] < 0 ? dst : src;
    const size_t it = from[v] < 0 ? v : (size_t)from[v];
    out.offsets[v + 1] = out.offsets[v] + (c.offsets[it + 1] - c.offsets[it]);
    out.block_ids.insert(out.block_ids.end(), c.block_ids.begin() + c.offsets[it], c.block_ids.begin() + c.offsets[it + 1]);
    out.masks.insert(out.masks.end(), c.masks.begin() + c.offsets[it], c.masks.begin() + c.offsets[it + 1]);
    if (from[v] >= 0) out.usable[v] = src.usable[it];
  }
  dst = std::move(out);
}
}  // namespace detail

/// motion_planning::VoxelCachedLazyPRM (motion-planning/VoxelCachedLazyPRM.h:195-830) on the batched engine: the roadmap BUILD --
/// createRoadmap with its CreateRoadmapOption flags (.h:469-497, .cpp:1380-1561), precompute{Vertex,Edge}Validity (:1563-1648),
/// precompute{Vertex,Edge}VoxelCache (:1692-1789), clear*VoxelCache, clearDisconnectedVertices (:1665-1690), the connection
/// strategies (:1327-1364) -- as apps/create_roadmap.cpp:252-331 drives it, and the QUERY side: solveWithRoadmap (:1977-2096 ->
/// constructSolution :2689-2771) for a batch of (start, goal) roadmap vertices, clearValidity (:1656-1663), the eager
/// re-validation of every cached set.  The graph (states, edges, voxel sets, validity) lives in this object on the host and in
/// a tr_roadmap in HBM that is rebuilt when the graph was edited; every loop over vertices or edges is one batched call.
///
/// Differences a caller sees: vertices are numbered 0 .. milestoneCount() - 1 in creation order (the reference's indexProperty_);
/// new milestones come from ONE reproducible candidate sequence (seed, candidate index), not from thread-local OMPL samplers;
/// items found invalid by a query stay in the arrays with status 2 instead of leaving the graph (the searches skip them), and
/// clearValidity() makes them unknown again; queries name roadmap vertices (the reference adds start / goal states first).
class VoxelCachedLazyPRM {
 public:
  enum CreateRoadmapOption {             // VoxelCachedLazyPRM.h:469-482
    LazyRoadmap = 0x0, VoxelizeVertices = 0x1, ValidateVertices = 0x2, VoxelizeEdges = 0x4, ValidateEdges = 0x8
  };
  struct Solution { std::vector<int32_t> status; std::vector<double> cost; std::vector<std::vector<int32_t>> paths; tr_roadmap_stats stats; };
  /// what the last createRoadmap did: the candidate edges of the connection loop with checkMotion's verdict (or is_fully_valid)
  /// and the reference's count of FK evaluations per edge, the candidates consumed by the vertex phase
  struct BuildReport {
    std::vector<int32_t> candidate_edges; std::vector<bool> accepted; std::vector<int32_t> n_fk;
    int64_t candidates_tried = 0; std::vector<int64_t> candidate_index; int32_t k = 0; double max_distance = 0;
  };

  /// a planner with an empty roadmap: createRoadmap / precompute* build it (`mv` supplies checkMotion and the space resolution)
  VoxelCachedLazyPRM(const VoxelBackboneValidityChecker &vc, const VoxelBackboneMotionValidator &mv, uint64_t seed = 0)
      : vc_(vc), mv_(&mv), S_(vc.robot().state_size()), seed_(seed) {
    vcache_ = detail::empty_items(0); ecache_ = detail::empty_items(0);
  }
  /// a loaded roadmap (fromRoadmapParser :2357-2580): states, edge index pairs, optional weights (NULL = state-space distance)
  VoxelCachedLazyPRM(const VoxelBackboneValidityChecker &vc, const std::vector<double> &states, size_t n_states,
                     const std::vector<int32_t> &edges, const std::vector<double> *weights = nullptr,
                     const VoxelBackboneMotionValidator *mv = nullptr)
      : vc_(vc), mv_(mv), S_(vc.robot().state_size()) {
    if (states.size() != n_states * S_) throw std::invalid_argument("State is not the right size");
    if (weights && weights->size() != edges.size() / 2) throw std::invalid_argument("one weight per edge");
    states_ = states; edges_ = edges;
    if (weights) weights_ = *weights;
    tips_.assign(3 * n_states, std::nan("")); has_tip_.assign(n_states, 0);
    vstat_.assign(n_states, 0); estat_.assign(edges.size() / 2, 0);
    vcache_ = detail::empty_items(n_states); ecache_ = detail::empty_items(edges.size() / 2);
    create_device_graph();                               // (an edge outside the roadmap throws here, as before)
  }
  ~VoxelCachedLazyPRM() { tr_roadmap_destroy(rm_); }
  VoxelCachedLazyPRM(const VoxelCachedLazyPRM &) = delete;
  VoxelCachedLazyPRM &operator=(const VoxelCachedLazyPRM &) = delete;

  // ---- the graph ----
  size_t milestoneCount() const { return vstat_.size(); }                       // .h:449
  size_t edgeCount() const { return estat_.size(); }                            // .h:452
  const std::vector<double> &states() const { return states_; }                 // milestoneCount() x state_size
  const std::vector<double> &tipPositions() const { return tips_; }             // x 3; NaN where tipPositionProperty_ is empty
  const std::vector<int32_t> &edges() const { return edges_; }                  // edgeCount() x 2
  const VoxelCaches &vertexVoxels() const { return vcache_; }                   // vertexVoxelsProperty_; usable[v] = has a cache
  const VoxelCaches &edgeVoxels() const { return ecache_; }
  const BuildReport &lastBuild() const { return report_; }
  /// 0 unknown, 1 VALIDITY_TRUE, 2 found invalid
  void validity(std::vector<uint8_t> &vertex_status, std::vector<uint8_t> &edge_status) { absorb(); vertex_status = vstat_; edge_status = estat_; }

  // ---- connection strategy (:1316-1364) ----
  void setMaxNearestNeighbors(size_t k) {                                        // KBoundedStrategy(k, maxDistance_) :1327-1344
    if (star_) throw std::runtime_error("Cannot set the maximum nearest neighbors for VoxelCachedLazyPRM");
    k_ = k;
  }
  void setStarConnectionStrategy() { star_ = true; }                             // KStarStrategy :1346-1356
  void setRange(double distance) { max_distance_ = distance; }                   // :1319-1326
  /// maxDistance_: 0.2 of the space's maximum extent unless set (SelfConfig::configurePlannerRange, :1242)
  double getRange() const {
    if (max_distance_ > 0) return max_distance_;
    double w_rot = 0, w_ret = 0, ext2 = 0;
    tr_space_weights(vc_.context(), &w_rot, &w_ret);
    const auto &rb = vc_.robot();
    for (auto &t : rb.tendons) ext2 += t.max_tension * t.max_tension;
    return 0.2 * (std::sqrt(ext2) + (rb.enable_rotation ? w_rot * 3.14159265358979323846 : 0.0) + (rb.enable_retraction ? w_ret * rb.specs.L : 0.0));
  }
  /// the box new milestones are drawn from (default: the planner's state-space bounds, Problem.cpp:101-163)
  void setSamplingBounds(const std::vector<double> &lo, const std::vector<double> &hi) {
    if (lo.size() != S_ || hi.size() != S_) throw std::invalid_argument("State is not the right size");
    lo_ = lo; hi_ = hi;
  }

  // ---- createRoadmap (:1380-1561) ----
  /// Brings the roadmap up to N milestones.  New milestones: the next candidates of the sequence that pass what `opt` asks of a
  /// vertex (nothing | is_valid_shape | is_valid_shape and no collision, :1412-1440); every new milestone is connected to its
  /// connection-strategy neighbours among ALL milestones (:1491-1502: nearestK on a structure that already holds the milestone, so
  /// k counts the milestone itself); with VoxelizeEdges / ValidateEdges the new edges are voxelised / checked (voxelizeEdge :2879 /
  /// computeEdgeValidity :2621) in one batched pass and the failing ones removed (:1508-1551).
  void createRoadmap(size_t N, int opt = LazyRoadmap) {
    const size_t Nv = milestoneCount();
    if (N <= Nv) return;                                                         // :1387-1391
    need_validators("createRoadmap");
    absorb();
    tr_ctx *c = vc_.context();
    const bool validate_verts = opt & ValidateVertices, validate_edges = opt & ValidateEdges;
    const bool voxelize_verts = validate_verts || (opt & VoxelizeVertices), voxelize_edges = validate_edges || (opt & VoxelizeEdges);
    const size_t add = N - Nv;
    const double *lo = lo_.empty() ? nullptr : lo_.data(), *hi = hi_.empty() ? nullptr : hi_.data();
    report_ = BuildReport{};
    std::vector<double> st(add * S_), tips(3 * add, std::nan(""));
    report_.candidate_index.resize(add);
    const uint64_t first = next_candidate_;
    if (!voxelize_verts) {                                                       // sampleUniform, accepted as it is (:1417-1419)
      check(c, tr_candidate_states(c, seed_, first, (int64_t)add, lo, hi, st.data()));
      for (size_t i = 0; i < add; i++) report_.candidate_index[i] = (int64_t)(first + i);
      next_candidate_ += add;
    } else if (validate_verts) {                                                 // ... until valid shape and no collision (:1421-1436)
      int64_t acc = 0, tried = 0;
      check(c, tr_sample_valid_vertices(c, seed_, first, lo, hi, (int64_t)add, 0, st.data(), tips.data(), report_.candidate_index.data(), &acc, &tried));
      if ((size_t)acc < add) throw std::runtime_error("createRoadmap: only " + std::to_string(acc) + " valid milestones in " + std::to_string(tried) + " candidates");
      next_candidate_ += (uint64_t)tried;
    } else {                                                                     // ... until valid shape (:1421-1426)
      size_t have = 0;
      while (have < add) {
        const size_t m = std::max<size_t>(4096, (add - have) + (add - have) / 2);
        std::vector<double> cand(m * S_);
        std::vector<uint64_t> bits((m + 63) / 64);
        std::vector<uint8_t> flags(m);
        check(c, tr_candidate_states(c, seed_, next_candidate_, (int64_t)m, lo, hi, cand.data()));
        check(c, tr_validate_batch(c, cand.data(), (int64_t)m, bits.data(), nullptr, flags.data()));
        const unsigned shape_ok = TR_FLAG_CONVERGED | TR_FLAG_LENGTH_OK | TR_FLAG_NO_SELFCOL;
        size_t used = m;
        for (size_t i = 0; i < m; i++) {
          if ((flags[i] & (shape_ok | TR_FLAG_DOMAIN)) != shape_ok) continue;
          std::copy(cand.begin() + i * S_, cand.begin() + (i + 1) * S_, st.begin() + have * S_);
          report_.candidate_index[have] = (int64_t)(next_candidate_ + i);
          if (++have == add) { used = i + 1; break; }
        }
        next_candidate_ += used;
        if (next_candidate_ - first > 64 * (uint64_t)add + (1u << 20)) throw std::runtime_error("createRoadmap: too few shape-valid candidates");
      }
    }
    report_.candidates_tried = (int64_t)(next_candidate_ - first);
    // the milestones join the graph before any of them is connected (:1463-1483)
    VoxelCaches vnew = detail::empty_items(add);
    if (voxelize_verts) {
      vnew = voxelize_states(vc_, st, add, &tips);
      for (size_t i = 0; i < add; i++) if (!vnew.usable[i]) throw std::runtime_error("createRoadmap: an accepted milestone has no valid shape");
    }
    states_.insert(states_.end(), st.begin(), st.end());
    tips_.insert(tips_.end(), tips.begin(), tips.end());
    has_tip_.insert(has_tip_.end(), add, voxelize_verts ? 1 : 0);
    vstat_.insert(vstat_.end(), add, validate_verts ? 1 : 0);
    detail::append_items(vcache_, vnew);
    dirty_ = true;
    // connectionStrategy_(v) for every new v; an edge exists once (:1491-1502)
    const int64_t kk = std::min<int64_t>((int64_t)N, star_ ? (int64_t)tr_kstar_k(c, (int64_t)N) : (int64_t)k_);
    const double maxd = star_ ? HUGE_VAL : getRange();
    report_.k = (int32_t)kk; report_.max_distance = maxd;
    std::vector<int32_t> cand((size_t)(2 * (int64_t)N * kk));
    int64_t ne = 0;
    if (kk >= 1 && Nv == 0) {
      check(c, tr_knn_edges(c, states_.data(), (int64_t)N, (int32_t)kk, maxd, cand.data(), (int64_t)N * kk, &ne));
    } else if (kk >= 1) {
      std::vector<int32_t> table((size_t)((int64_t)N * kk), -1);                  // rows of the old milestones stay empty: they do not connect again
      std::vector<double> dist((size_t)((int64_t)add * kk));
      check(c, tr_knn_range(c, states_.data(), (int64_t)N, (int64_t)Nv, (int64_t)add, (int32_t)kk, maxd, table.data() + Nv * (size_t)kk, dist.data()));
      check(c, tr_knn_table_edges(c, table.data(), (int64_t)N, (int32_t)kk, cand.data(), (int64_t)N * kk, &ne));
    }
    cand.resize((size_t)(2 * ne));
    report_.candidate_edges = cand;
    report_.accepted.assign((size_t)ne, true);
    VoxelCaches enew = detail::empty_items((size_t)ne);
    std::vector<size_t> keep;
    if (voxelize_edges && ne > 0) {
      enew.offsets.assign((size_t)ne + 1, 0);
      std::vector<uint64_t> bits((size_t)(ne + 63) / 64);
      report_.n_fk.resize((size_t)ne);
      check(c, (validate_edges ? tr_connect_edges_indexed : tr_voxelize_edges_indexed)(
                   c, &mv_->space, states_.data(), (int64_t)N, cand.data(), ne, enew.offsets.data(), bits.data(), report_.n_fk.data()));
      enew.usable = detail::unpack(bits, (size_t)ne);
      detail::fetch(c, enew);
      report_.accepted = enew.usable;
      for (size_t e = 0; e < (size_t)ne; e++) if (enew.usable[e]) keep.push_back(e);
      enew = detail::select_items(enew, keep);
    } else {
      for (size_t e = 0; e < (size_t)ne; e++) keep.push_back(e);
    }
    for (size_t e : keep) {
      edges_.push_back(cand[2 * e]); edges_.push_back(cand[2 * e + 1]);
      if (!weights_.empty()) weights_.push_back(distance(&states_[(size_t)cand[2 * e] * S_], &states_[(size_t)cand[2 * e + 1] * S_]));
    }
    estat_.insert(estat_.end(), keep.size(), validate_edges ? 1 : 0);
    detail::append_items(ecache_, enew);
  }

  // ---- precompute* (:1563-1648, :1692-1789) ----
  /// voxelizeVertex for every vertex without a cache or a tip; vertices without a valid shape leave the roadmap
  void precomputeVertexVoxelCache() {
    need_validators("precomputeVertexVoxelCache");
    absorb();
    remove_vertices(voxelize_missing_vertices(nullptr));
  }
  /// voxelizeEdge for every edge without a cache; edges that are not fully valid leave the roadmap
  void precomputeEdgeVoxelCache() {
    need_validators("precomputeEdgeVoxelCache");
    absorb();
    std::vector<size_t> miss;
    for (size_t e = 0; e < edgeCount(); e++) if (!ecache_.usable[e]) miss.push_back(e);
    remove_edges(edge_pass(miss, false));
  }
  void precomputeVoxelCache() { precomputeVertexVoxelCache(); precomputeEdgeVoxelCache(); }
  /// computeVertexValidity for every vertex not known valid (voxelise if needed, then against the obstacles); the others leave
  void precomputeVertexValidity() {
    need_validators("precomputeVertexValidity");
    absorb();
    std::vector<bool> todo(milestoneCount());
    for (size_t v = 0; v < milestoneCount(); v++) todo[v] = vstat_[v] != 1;
    std::vector<size_t> gone = voxelize_missing_vertices(&todo);
    const std::vector<bool> hit = caches_collide(vc_, vcache_);
    std::vector<bool> out(milestoneCount(), false);
    for (size_t v : gone) out[v] = true;
    for (size_t v = 0; v < milestoneCount(); v++) {
      if (!todo[v] || out[v]) continue;
      if (hit[v]) out[v] = true; else vstat_[v] = 1;
    }
    gone.clear();
    for (size_t v = 0; v < milestoneCount(); v++) if (out[v]) gone.push_back(v);
    remove_vertices(gone);
  }
  /// computeEdgeValidity for every edge not known valid; the others leave
  void precomputeEdgeValidity() {
    need_validators("precomputeEdgeValidity");
    absorb();
    std::vector<size_t> miss, have;
    for (size_t e = 0; e < edgeCount(); e++) if (estat_[e] != 1) (ecache_.usable[e] ? have : miss).push_back(e);
    std::vector<size_t> gone = edge_pass(miss, true);                            // voxelise + collide in one traversal of the samples
    std::vector<bool> out(edgeCount(), false);
    for (size_t e : gone) out[e] = true;
    for (size_t e : miss) if (!out[e]) estat_[e] = 1;
    if (!have.empty()) {
      const std::vector<bool> hit = caches_collide(vc_, ecache_);
      for (size_t e : have) { if (hit[e]) out[e] = true; else estat_[e] = 1; }
    }
    gone.clear();
    for (size_t e = 0; e < edgeCount(); e++) if (out[e]) gone.push_back(e);
    remove_edges(gone);
  }
  void precomputeValidity() { precomputeVertexValidity(); precomputeEdgeValidity(); }
  void clearVertexVoxelCache() { absorb(); vcache_ = detail::empty_items(milestoneCount()); dirty_ = true; }    // :1791-1795
  void clearEdgeVoxelCache() { absorb(); ecache_ = detail::empty_items(edgeCount()); dirty_ = true; }           // :1797-1801
  void clearVoxelCache() { clearVertexVoxelCache(); clearEdgeVoxelCache(); }
  /// vertices outside the largest connected component leave the roadmap (:1665-1690)
  void clearDisconnectedVertices() {
    absorb();
    const size_t V = milestoneCount();
    std::vector<int32_t> parent(V);
    for (size_t v = 0; v < V; v++) parent[v] = (int32_t)v;
    auto find = [&](int32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
    for (size_t e = 0; e < edgeCount(); e++) {
      const int32_t a = find(edges_[2 * e]), b = find(edges_[2 * e + 1]);
      if (a != b) parent[std::max(a, b)] = std::min(a, b);
    }
    std::vector<size_t> size(V, 0);
    for (size_t v = 0; v < V; v++) size[(size_t)find((int32_t)v)]++;
    size_t best = 0;
    for (size_t v = 0; v < V; v++) if (size[v] > size[best]) best = v;
    std::vector<size_t> gone;
    for (size_t v = 0; v < V; v++) if ((size_t)find((int32_t)v) != best) gone.push_back(v);
    remove_vertices(gone);
  }

  // ---- queries ----
  /// vertexVoxelsProperty_ / edgeVoxelsProperty_ as CSR; `usable` = which items have a cache at all (empty = all of them)
  void setCaches(const VoxelCaches &vertices, const VoxelCaches &edges) {
    if (vertices.items() != milestoneCount() || edges.items() != edgeCount()) throw std::invalid_argument("cache offsets do not match the roadmap");
    absorb();
    vcache_ = vertices; ecache_ = edges;
    if (vcache_.usable.empty()) vcache_.usable.assign(milestoneCount(), true);
    if (ecache_.usable.empty()) ecache_.usable.assign(edgeCount(), true);
    caches_given_ = true;
    dirty_ = true;
  }
  /// landmark lower bounds for the searches (0 = the reference's heuristic alone); paths and costs do not depend on it
  void prepare(int n_landmarks = 16, int n_threads = 0) { lm_ = n_landmarks; lm_threads_ = n_threads; sync(); }
  void clearValidity() {                                                         // :1656-1663
    std::fill(vstat_.begin(), vstat_.end(), 0); std::fill(estat_.begin(), estat_.end(), 0);
    if (rm_ && !dirty_) rcheck(tr_roadmap_clear_validity(rm_));
  }
  /// every cached set against the checker's current obstacle grid -> (#invalid vertices, #invalid edges)
  std::pair<int64_t, int64_t> revalidate() {
    sync();
    int64_t nv = 0, ne = 0;
    rcheck(tr_roadmap_revalidate(rm_, &nv, &ne));
    return {nv, ne};
  }
  Solution solveWithRoadmap(const std::vector<int32_t> &starts, const std::vector<int32_t> &goals, int n_threads = 0) {
    if (starts.size() != goals.size()) throw std::invalid_argument("starts and goals differ in length");
    sync();
    const size_t n = starts.size();
    Solution s;
    s.status.resize(n); s.cost.resize(n);
    std::vector<int64_t> off(n + 1, 0);
    rcheck(tr_roadmap_solve(rm_, starts.data(), goals.data(), (int64_t)n, n_threads, s.status.data(), s.cost.data(), off.data(), &s.stats));
    std::vector<int32_t> pv((size_t)off[n]);
    rcheck(tr_roadmap_fetch_paths(rm_, pv.data(), (int64_t)pv.size()));
    s.paths.resize(n);
    for (size_t q = 0; q < n; q++) s.paths[q].assign(pv.begin() + off[q], pv.begin() + off[q + 1]);
    return s;
  }
  // Where the graph searches of the last solveWithRoadmap ran (tr_roadmap_search_stats): the kernel, the host threads, the component labels.
  struct SearchStats { int64_t on_device, handed_back, on_host_meanwhile, list_moves, expanded_on_device, expanded_on_host, answered_by_components; };
  SearchStats searchStats() const {
    int64_t o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (rm_) rcheck(tr_roadmap_search_stats(rm_, o));
    return SearchStats{o[0], o[1], o[2], o[3], o[4], o[5], o[6]};
  }
  /// Device memory the graph searches hold for this planner between calls, and handing it back (tr_roadmap_release_search_state;
  /// the next large batch of queries allocates it again).
  int64_t searchStateBytes() const { int64_t b = 0; if (rm_) rcheck(tr_roadmap_search_state_bytes(rm_, &b)); return b; }
  void reserveSearchState(int64_t n_queries) { sync(); rcheck(tr_roadmap_reserve_search_state(rm_, n_queries)); }
  int64_t releaseSearchState() { int64_t b = 0; if (rm_) rcheck(tr_roadmap_release_search_state(rm_, &b)); return b; }
  /// CompoundStateSpace::distance with the weights of Problem.cpp:112-152 (the edge cost connectVertices stores, :2857-2861)
  double distance(const double *a, const double *b) const {
    const auto &rb = vc_.robot();
    const size_t N = rb.tendons.size();
    double w_rot = 0, w_ret = 0, s = 0;
    tr_space_weights(vc_.context(), &w_rot, &w_ret);
    for (size_t i = 0; i < N; i++) s += (a[i] - b[i]) * (a[i] - b[i]);
    double d = std::sqrt(s);
    size_t k = N;
    if (rb.enable_rotation) { double r = std::fabs(a[k] - b[k]); if (r > 3.14159265358979323846) r = 2 * 3.14159265358979323846 - r; d += w_rot * r; k++; }
    if (rb.enable_retraction) d += w_ret * std::fabs(a[k] - b[k]);
    return d;
  }

 private:
  void rcheck(int st) const {
    if (st == TR_OK) return;
    const std::string m = tr_roadmap_last_error(rm_);
    if (st == TR_ERR_OUT_OF_RANGE) throw std::out_of_range(m);
    if (st == TR_ERR_INVALID_ARG) throw std::invalid_argument(m);
    throw std::runtime_error(m);
  }
  void need_validators(const char *what) const {
    if (!mv_) throw std::runtime_error(std::string(what) + ": missing voxel motion validator");      // the reference's setup_ == false (:1286-1298)
  }
  void create_device_graph() {
    tr_roadmap_destroy(rm_); rm_ = nullptr;
    const int st = tr_roadmap_create(vc_.context(), states_.data(), (int64_t)milestoneCount(), edges_.data(),
                                     weights_.empty() ? nullptr : weights_.data(), (int64_t)edgeCount(), &rm_);
    if (st == TR_ERR_OUT_OF_RANGE) throw std::out_of_range("edge refers to a state outside the roadmap");
    if (st != TR_OK) throw std::invalid_argument("tr_roadmap_create failed");
    dirty_ = false;
  }
  /// validity the query loop discovered since the last edit comes back into this object
  void absorb() {
    if (rm_ && !dirty_ && milestoneCount()) rcheck(tr_roadmap_get_validity(rm_, vstat_.data(), estat_.empty() ? nullptr : estat_.data()));
  }
  /// the device image follows the host graph: graph, caches (computed where a builder left them out: the reference voxelises such
  /// items when a query first meets them, computeVertexValidity :2607-2618), known validity, landmark tables
  void sync() {
    if (rm_ && !dirty_) return;
    if (mv_ && !caches_given_) {
      std::vector<size_t> gone = voxelize_missing_vertices(nullptr);
      for (size_t v : gone) vstat_[v] = 2;                                       // no valid shape: invalid in every environment
      std::vector<size_t> miss;
      for (size_t e = 0; e < edgeCount(); e++) if (!ecache_.usable[e] && estat_[e] != 2) miss.push_back(e);
      for (size_t e : edge_pass(miss, false)) estat_[e] = 2;
    }
    create_device_graph();
    const auto vp = detail::pack(vcache_.usable), ep = detail::pack(ecache_.usable);
    rcheck(tr_roadmap_set_caches(rm_, vcache_.offsets.data(), vcache_.block_ids.data(), vcache_.masks.data(), vp.data(),
                                 ecache_.offsets.data(), ecache_.block_ids.data(), ecache_.masks.data(), ep.data()));
    if (milestoneCount()) rcheck(tr_roadmap_set_validity(rm_, vstat_.data(), estat_.empty() ? nullptr : estat_.data()));
    if (lm_ >= 0) rcheck(tr_roadmap_prepare(rm_, lm_, lm_threads_));
  }
  /// voxelizeVertex (:2803-2837) for the vertices (of `only`, if given) without a cache or a tip -> those without a valid shape
  std::vector<size_t> voxelize_missing_vertices(const std::vector<bool> *only) {
    std::vector<size_t> miss, gone;
    for (size_t v = 0; v < milestoneCount(); v++)
      if ((!vcache_.usable[v] || !has_tip_[v]) && vstat_[v] != 2 && (!only || (*only)[v])) miss.push_back(v);
    if (miss.empty()) return gone;
    std::vector<double> st(miss.size() * S_), tips;
    for (size_t i = 0; i < miss.size(); i++) std::copy(states_.begin() + miss[i] * S_, states_.begin() + (miss[i] + 1) * S_, st.begin() + i * S_);
    VoxelCaches got = voxelize_states(vc_, st, miss.size(), &tips);
    detail::replace_items(vcache_, miss, got);
    for (size_t i = 0; i < miss.size(); i++) {
      if (!got.usable[i]) { gone.push_back(miss[i]); continue; }
      std::copy(tips.begin() + 3 * i, tips.begin() + 3 * i + 3, tips_.begin() + 3 * miss[i]);
      has_tip_[miss[i]] = 1;
    }
    dirty_ = true;
    return gone;
  }
  /// voxelizeEdge (collide = false, :2879-2902) or computeEdgeValidity (collide = true, :2621-2631) for the listed edges, which have
  /// no cache yet -> the ones that fail; the others own their voxel set afterwards
  std::vector<size_t> edge_pass(const std::vector<size_t> &list, bool collide) {
    std::vector<size_t> gone;
    if (list.empty()) return gone;
    tr_ctx *c = vc_.context();
    std::vector<int32_t> sub(2 * list.size());
    for (size_t i = 0; i < list.size(); i++) { sub[2 * i] = edges_[2 * list[i]]; sub[2 * i + 1] = edges_[2 * list[i] + 1]; }
    VoxelCaches got;
    got.offsets.assign(list.size() + 1, 0);
    std::vector<uint64_t> bits((list.size() + 63) / 64);
    check(c, (collide ? tr_connect_edges_indexed : tr_voxelize_edges_indexed)(c, &mv_->space, states_.data(), (int64_t)milestoneCount(), sub.data(),
                                                                              (int64_t)list.size(), got.offsets.data(), bits.data(), nullptr));
    got.usable = detail::unpack(bits, list.size());
    detail::fetch(c, got);
    detail::replace_items(ecache_, list, got);
    for (size_t i = 0; i < list.size(); i++) if (!got.usable[i]) gone.push_back(list[i]);
    dirty_ = true;
    return gone;
  }
  /// removeVertices (:2904-2948): the vertices and every edge at them leave; the others are renumbered in order
  void remove_vertices(const std::vector<size_t> &gone) {
    if (gone.empty()) return;
    const size_t V = milestoneCount();
    std::vector<int32_t> renum(V, 0);
    for (size_t v : gone) renum[v] = -1;
    std::vector<size_t> keep;
    for (size_t v = 0; v < V; v++) if (renum[v] == 0) { renum[v] = (int32_t)keep.size(); keep.push_back(v); }
    std::vector<size_t> egone;
    for (size_t e = 0; e < edgeCount(); e++) if (renum[(size_t)edges_[2 * e]] < 0 || renum[(size_t)edges_[2 * e + 1]] < 0) egone.push_back(e);
    remove_edges(egone);
    for (int32_t &x : edges_) x = renum[(size_t)x];
    std::vector<double> st(keep.size() * S_), tp(keep.size() * 3);
    std::vector<uint8_t> ht(keep.size()), vs(keep.size());
    for (size_t i = 0; i < keep.size(); i++) {
      std::copy(states_.begin() + keep[i] * S_, states_.begin() + (keep[i] + 1) * S_, st.begin() + i * S_);
      std::copy(tips_.begin() + keep[i] * 3, tips_.begin() + keep[i] * 3 + 3, tp.begin() + i * 3);
      ht[i] = has_tip_[keep[i]]; vs[i] = vstat_[keep[i]];
    }
    states_.swap(st); tips_.swap(tp); has_tip_.swap(ht); vstat_.swap(vs);
    vcache_ = detail::select_items(vcache_, keep);
    dirty_ = true;
  }
  void remove_edges(const std::vector<size_t> &gone) {
    if (gone.empty()) return;
    std::vector<bool> out(edgeCount(), false);
    for (size_t e : gone) out[e] = true;
    std::vector<size_t> keep;
    for (size_t e = 0; e < edgeCount(); e++) if (!out[e]) keep.push_back(e);
    std::vector<int32_t> ed(2 * keep.size());
    std::vector<double> w(weights_.empty() ? 0 : keep.size());
    std::vector<uint8_t> es(keep.size());
    for (size_t i = 0; i < keep.size(); i++) {
      ed[2 * i] = edges_[2 * keep[i]]; ed[2 * i + 1] = edges_[2 * keep[i] + 1];
      if (!weights_.empty()) w[i] = weights_[keep[i]];
      es[i] = estat_[keep[i]];
    }
    edges_.swap(ed); weights_.swap(w); estat_.swap(es);
    ecache_ = detail::select_items(ecache_, keep);
    dirty_ = true;
  }

  const VoxelBackboneValidityChecker &vc_;
  const VoxelBackboneMotionValidator *mv_ = nullptr;
  size_t S_ = 0;
  uint64_t seed_ = 0, next_candidate_ = 0;
  std::vector<double> lo_, hi_;
  std::vector<double> states_, tips_, weights_;
  std::vector<uint8_t> has_tip_, vstat_, estat_;
  std::vector<int32_t> edges_;
  VoxelCaches vcache_, ecache_;
  bool caches_given_ = false;
  bool star_ = false;
  size_t k_ = 5;                       // magic::DEFAULT_NEAREST_NEIGHBORS_LAZY (:125)
  double max_distance_ = 0.0;
  int lm_ = -1, lm_threads_ = 0;
  BuildReport report_;
  tr_roadmap *rm_ = nullptr;
  bool dirty_ = true;
};

}  // namespace motion_planning
}  // namespace tendon_hip
