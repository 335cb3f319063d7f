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
//   motion_planning::VoxelCachedLazyPRM                                  solveWithRoadmap for a batch of queries on a cached roadmap
//
// Error behaviour: every tr_status is rethrown as the C++ exception type the reference throws at
// the same condition (std::invalid_argument, std::out_of_range, std::domain_error,
// std::length_error, std::runtime_error).
#pragma once
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

/// The query side of motion_planning::VoxelCachedLazyPRM on a roadmap with voxel caches: solveWithRoadmap
/// (motion-planning/VoxelCachedLazyPRM.cpp:1977-2096 -> constructSolution :2689-2771) for a batch of (start, goal)
/// roadmap vertices, clearValidity (:1656-1663), and the eager re-validation of every cached set.
class VoxelCachedLazyPRM {
 public:
  struct Solution { std::vector<int32_t> status; std::vector<double> cost; std::vector<std::vector<int32_t>> paths; tr_roadmap_stats stats; };

  VoxelCachedLazyPRM(const VoxelBackboneValidityChecker &vc, const std::vector<double> &states, size_t n_states,
                     const std::vector<int32_t> &edges, const std::vector<double> *weights = nullptr)
      : vc_(vc) {
    if (states.size() != n_states * vc.robot().state_size()) throw std::invalid_argument("State is not the right size");
    const int st = tr_roadmap_create(vc.context(), states.data(), (int64_t)n_states, edges.data(), weights ? weights->data() : nullptr,
                                     (int64_t)(edges.size() / 2), &rm_);
    if (st == TR_ERR_OUT_OF_RANGE) throw std::out_of_range("edge refers to a state outside the roadmap");
    if (st != TR_OK) throw std::invalid_argument("tr_roadmap_create failed");
  }
  ~VoxelCachedLazyPRM() { tr_roadmap_destroy(rm_); }
  VoxelCachedLazyPRM(const VoxelCachedLazyPRM &) = delete;
  VoxelCachedLazyPRM &operator=(const VoxelCachedLazyPRM &) = delete;

  /// vertexVoxelsProperty_ / edgeVoxelsProperty_ as CSR; `usable` = which items have a cache at all
  void setCaches(const VoxelCaches &vertices, const VoxelCaches &edges) {
    auto pack = [](const std::vector<bool> &b) {
      std::vector<uint64_t> w((b.size() + 63) / 64, 0);
      for (size_t i = 0; i < b.size(); i++) if (b[i]) w[i >> 6] |= (uint64_t)1 << (i & 63);
      return w;
    };
    const auto vp = pack(vertices.usable), ep = pack(edges.usable);
    rcheck(tr_roadmap_set_caches(rm_, vertices.offsets.data(), vertices.block_ids.data(), vertices.masks.data(),
                                 vertices.usable.empty() ? nullptr : vp.data(), edges.offsets.data(), edges.block_ids.data(),
                                 edges.masks.data(), edges.usable.empty() ? nullptr : ep.data()));
  }
  /// landmark lower bounds for the searches (0 = the reference's heuristic alone); paths and costs do not depend on it
  void prepare(int n_landmarks = 16, int n_threads = 0) { rcheck(tr_roadmap_prepare(rm_, n_landmarks, n_threads)); }
  void clearValidity() { rcheck(tr_roadmap_clear_validity(rm_)); }
  /// every cached set against the checker's current obstacle grid -> (#invalid vertices, #invalid edges)
  std::pair<int64_t, int64_t> revalidate() {
    int64_t nv = 0, ne = 0;
    rcheck(tr_roadmap_revalidate(rm_, &nv, &ne));
    return {nv, ne};
  }
  Solution solveWithRoadmap(const std::vector<int32_t> &starts, const std::vector<int32_t> &goals, int n_threads = 0) {
    if (starts.size() != goals.size()) throw std::invalid_argument("starts and goals differ in length");
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
    rcheck(tr_roadmap_search_stats(rm_, o));
    return SearchStats{o[0], o[1], o[2], o[3], o[4], o[5], o[6]};
  }

 private:
  void rcheck(int st) const {
    if (st == TR_OK) return;
    const std::string m = tr_roadmap_last_error(rm_);
    if (st == TR_ERR_OUT_OF_RANGE) throw std::out_of_range(m);
    if (st == TR_ERR_INVALID_ARG) throw std::invalid_argument(m);
    throw std::runtime_error(m);
  }
  const VoxelBackboneValidityChecker &vc_;
  tr_roadmap *rm_ = nullptr;
};

}  // namespace motion_planning
}  // namespace tendon_hip
