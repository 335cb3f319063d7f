/* tendon_hip.h -- C ABI of libtendon_hip.so: the MI355X (gfx950) batched tendon-robot
 * forward-kinematics + voxel-collision engine.
 *
 * This is the drop-in boundary for the reference's hot path (paths relative to the reference's
 * cpp/src/).  Plain pointers and sizes only.  Every entry point cites the reference interface it
 * stands in for; INTEGRATION.md shows the C++ shim (tendon::TendonRobot, collision::VoxelOctree,
 * motion_planning::VoxelBackboneValidityChecker / VoxelBackboneMotionValidator) a maintainer
 * adds on the reference side.
 *
 * Conventions
 *   - every function returns a tr_status (0 = ok); tr_last_error(ctx) gives the message.  The
 *     statuses map 1:1 to the C++ exception types the reference throws at the same conditions.
 *   - "states" are the reference's robot states: n x S row-major doubles,
 *     [tau_1..tau_N, (theta if enable_rotation), (s_start if enable_retraction)]
 *     (tendon/TendonRobot.h:105-115, motion-planning/AbstractValidityChecker.cpp:50-78).
 *   - *_dev entry points take DEVICE pointers and a hipStream_t (as void*); they enqueue work and
 *     return without synchronising.  The host-pointer forms copy in/out and synchronise.
 *   - one context = one workspace (fallback list, point columns, ordering buffers, argument slots are
 *     per context).  Entry points serialise on a per-context mutex, and the *_dev calls that touch that
 *     workspace execute in ISSUE ORDER whichever streams they name: a call arriving on a different
 *     stream than the previous one first makes its stream wait (hipStreamWaitEvent) for that call's
 *     work.  So two caller streams never share the scratch concurrently, and there is no concurrency
 *     to gain from several streams on ONE context -- use one context per concurrent stream
 *     (contexts share nothing but the device).  Caller buffers are the caller's to order.
 *   - validity bitmasks: bit (i & 63) of word (i >> 6) is configuration i; padding bits are 0.
 *   - non-convergence / NaN in a lane is not an error: that configuration is simply invalid.
 */
#ifndef TENDON_HIP_H
#define TENDON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TR_MAX_TENDONS 8
#define TR_MAX_COEF    8

typedef enum {
  TR_OK                = 0,
  TR_ERR_INVALID_ARG   = 1,  /* std::invalid_argument: state size, dims, dL vs voxel size ...   */
  TR_ERR_OUT_OF_RANGE  = 2,  /* std::out_of_range: tendon / tau count mismatch                  */
  TR_ERR_DOMAIN        = 3,  /* std::domain_error: point outside the voxel domain (find_cell)   */
  TR_ERR_LENGTH        = 4,  /* std::length_error: non-positive voxel limits                    */
  TR_ERR_RUNTIME       = 5,  /* std::runtime_error                                              */
  TR_ERR_HIP           = 6,  /* HIP runtime failure (message carries hipGetErrorString)          */
  TR_ERR_UNSUPPORTED   = 7   /* feature of the reference not offered by this build              */
} tr_status;

/* tendon::TendonRobot + BackboneSpecs + TendonSpecs (tendon/TendonRobot.h:52-58,
 * tendon/BackboneSpecs.h:14-20, tendon/TendonSpecs.h:25-30).  All tendons share n_a = C.size()
 * and n_m = D.size(), as get_r_info assumes (tendon/get_r_info.cpp:112-115). */
typedef struct {
  double  r;                       /* robot radius (m)                                         */
  double  L, dL, ro, ri, E, nu;    /* backbone                                                  */
  int32_t n_tendons, n_a, n_m;
  const double *C;                 /* n_tendons x n_a row-major: theta_i(t) = sum C[i][k] t^k   */
  const double *D;                 /* n_tendons x n_m row-major: rho_i(t)   = sum D[i][k] t^k   */
  const double *max_tension;       /* [n_tendons]                                               */
  const double *min_length;        /* [n_tendons]                                               */
  const double *max_length;        /* [n_tendons]                                               */
  int32_t enable_rotation, enable_retraction;
  double  residual_threshold;
} tr_robot_desc;

typedef struct tr_ctx tr_ctx;

/* flag bits written per configuration by the validate calls (optional output) */
#define TR_FLAG_CONVERGED   1u   /* fk.converged && home.converged                               */
#define TR_FLAG_LENGTH_OK   2u   /* is_within_length_limits                                      */
#define TR_FLAG_NO_SELFCOL  4u   /* !collides_self                                               */
#define TR_FLAG_NO_VOXCOL   8u   /* !collides(voxelize(fk))                                      */
#define TR_FLAG_DOMAIN     16u   /* a backbone point was non-finite or absurdly far outside the
                                    voxel domain (reference: undefined behaviour); forced invalid */

/* ---- life cycle ------------------------------------------------------------------------- */

/* Replaces constructing a tendon::TendonRobot (TendonRobot::from_toml, tendon/TendonRobot.cpp:1033-1089)
 * plus the per-call constants of tension_shape (get_stiffness_matrices, t_range,
 * tendon/TendonRobot.cpp:69-84,105-148).  device = HIP device ordinal. */
int tr_create(const tr_robot_desc *robot, int device, tr_ctx **out);
void tr_destroy(tr_ctx *ctx);
const char *tr_last_error(const tr_ctx *ctx);   /* ctx may be NULL: last create error */

int tr_state_size(const tr_ctx *ctx);           /* TendonRobot::state_size, TendonRobot.h:60-64 */
int tr_num_points(const tr_ctx *ctx);           /* |t_range(0, L, dL)| = max backbone points    */
int tr_device(const tr_ctx *ctx);

/* Home shape tendon lengths at s_start = 0 (TendonRobot::home_shape, tendon/TendonRobot.cpp:249-314) */
int tr_home_lengths(const tr_ctx *ctx, double *L_i /*[n_tendons]*/);

/* Replaces the VoxelOctree obstacle member + VoxelEnvironment rotation held by the voxel
 * validity checkers (motion-planning/AbstractVoxelValidityChecker.h:63-64,
 * VoxelEnvironment.cpp:129-131).  blocks: HOST pointer, Nb^3 uint64 (Nb = N/4), index
 * ((bx*Nb)+by)*Nb+bz, bit x*16+y*4+z inside a block (collision/VoxelOctree.cpp:1501-1503).
 * lim = {xmin,xmax,ymin,ymax,zmin,zmax}.  inv_rot: row-major 3x3 (NULL = identity).
 * Fails with TR_ERR_INVALID_ARG when dL > max voxel edge, as VoxelBackboneValidityChecker's
 * constructor does (motion-planning/VoxelBackboneValidityChecker.h:37-45). */
int tr_set_grid(tr_ctx *ctx, uint32_t N, const double lim[6], const uint64_t *blocks,
                const double inv_rot[9]);

/* Which state validity checker is installed (si->setStateValidityChecker, motion-planning/Problem.h:175-197):
 * TR_CHECKER_BACKBONE = motion_planning::VoxelBackboneValidityChecker (the backbone polyline against a
 * pre-dilated environment, VoxelBackboneValidityChecker.h:28-58; the default),
 * TR_CHECKER_SPHERES  = motion_planning::VoxelValidityChecker (a sphere of the robot radius at every
 * backbone point against the raw environment, VoxelValidityChecker.h:18-26).
 * It decides tr_validate_batch* (isValid) and -- exactly where the reference's motion validators ask the installed
 * checker, `_vc->collides(shape)` in voxelize_until_invalid_impl (VoxelBackboneMotionValidator.cpp:83-91) -- the
 * per-sample test of the checkMotion(s1, s2, last_valid) forms: tr_validate_edges_last_valid and
 * tr_validate_edges_discrete with a last_valid_t output.  checkMotion(s1, s2) itself (tr_validate_edges,
 * tr_validate_edges_indexed, tr_validate_edges_discrete without last_valid_t) and the voxel caches (tr_voxelize_*)
 * sweep the BACKBONE against the voxels under either checker, as AbstractVoxelMotionValidator::checkMotion does
 * (AbstractVoxelMotionValidator.h:143-151: voxelize() = is_valid_shape only, then collides(partial.voxels)).
 * Call it before tr_set_grid: the dL <= voxel size check of tr_set_grid is the backbone checker's constructor check
 * and is skipped for TR_CHECKER_SPHERES. */
#define TR_CHECKER_BACKBONE 0
#define TR_CHECKER_SPHERES 1
int tr_set_checker(tr_ctx *ctx, int32_t checker);

/* ---- environment preparation on the resident obstacle grid ------------------------------
 * collision::VoxelOctree's editing operations, applied to the grid tr_set_grid uploaded, without it
 * leaving the device (apps/prepare_voxel_env.cpp:247-315 runs them on the host octree):
 *   tr_grid_add_spheres      add_sphere for n spheres, (cx, cy, cz, r) each (VoxelOctree.cpp:434-469)
 *   tr_grid_add_capsules     add_capsule for n capsules, (ax, ay, az, bx, by, bz, r) each (:471-515) -- with add_spheres
 *                            everything Environment::voxelize rasterises (motion-planning/Environment.cpp:62-100)
 *   tr_grid_remove_interior  remove_interior(keep_diagonal) (:533-689; VoxelOctree.h:207-210)
 *   tr_grid_dilate           dilate(num, use_diagonal) (:693-818; VoxelOctree.h:215-218)
 *   tr_grid_dilate_sphere    dilate_sphere(r) (:950-952)
 *   tr_get_grid              download the current blocks ((N/4)^3 words, layout as tr_set_grid) */
int tr_grid_add_spheres(tr_ctx *ctx, const double *spheres, int64_t n);
int tr_grid_add_capsules(tr_ctx *ctx, const double *capsules, int64_t n);
int tr_grid_remove_interior(tr_ctx *ctx, int32_t keep_diagonal);
int tr_grid_dilate(tr_ctx *ctx, int32_t num, int32_t use_diagonal);
int tr_grid_dilate_sphere(tr_ctx *ctx, double r);
int tr_get_grid(tr_ctx *ctx, uint64_t *blocks);

/* Pre-size the device workspace for batches of up to n configurations (optional; the batch
 * calls grow it on demand, which allocates and therefore synchronises). */
int tr_reserve(tr_ctx *ctx, int64_t n);

/* Pre-size the FK sample pool and the device frontier of the edge calls (tr_validate_edges*,
 * tr_voxelize_edges) for batches of up to n_edges edges (optional, as above). */
int tr_reserve_edges(tr_ctx *ctx, int64_t n_edges);

/* ---- forward kinematics: TendonRobot::shape / forward_kinematics ------------------------ */

/* Batched TendonRobot::shape(state) (tendon/TendonRobot.h:105-131 -> tension_shape,
 * tendon/TendonRobot.cpp:325-500), replacing the omp-parallel loops at
 * apps/estimate_length_discretization.cpp:62-71 and apps/roadmap2samples.cpp:65-78.
 * Host buffers; any output pointer may be NULL.
 *   p         n x P x 3   backbone points (rows past n_points[i] are NaN), P = tr_num_points
 *   R         n x P x 9   rotation matrices, column-major per matrix (Eigen layout)
 *   L, L_i    n, n x N    backbone / tendon lengths
 *   converged n           TendonResult::converged
 *   n_points  n           points of configuration i (P unless retraction shortens it)      */
int tr_fk_batch(tr_ctx *ctx, const double *states, int64_t n,
                double *p, double *R, double *L, double *L_i, uint8_t *converged, int32_t *n_points);

/* Device form.  Points are written structure-of-arrays, lane-contiguous:
 * d_px[j*ld + i] is x of point j of configuration i (ld >= n, multiple of 64).  d_R (optional)
 * is [9][P][ld].  d_Li is [N][ld].
 * With retraction enabled a configuration has d_n_points[i] <= P points and its rows are aligned at the
 * TIP: point j of configuration i is in row j + (P - d_n_points[i]) (the tip always in row P - 1). */
int tr_fk_batch_dev(tr_ctx *ctx, const double *d_states, int64_t n, int64_t ld,
                    double *d_px, double *d_py, double *d_pz, double *d_R,
                    double *d_L, double *d_Li, uint8_t *d_converged, int32_t *d_n_points,
                    void *stream);

/* The same for retraction-enabled robots, whose home-shape tendon lengths depend on the configuration
 * (home_shape(s_start), tendon/TendonRobot.cpp:249-314): d_n_points and d_home_Li ([N][ld]) are mandatory
 * outputs; rows are aligned at the tip as described above.  Feeds tr_validate_shapes_retraction_dev. */
int tr_fk_batch_retraction_dev(tr_ctx *ctx, const double *d_states, int64_t n, int64_t ld,
                               double *d_px, double *d_py, double *d_pz, double *d_R,
                               double *d_L, double *d_Li, uint8_t *d_converged, int32_t *d_n_points,
                               double *d_home_Li, void *stream);

/* ---- state validity: StateValidityChecker::isValid -------------------------------------- */

/* Batched AbstractValidityChecker::isValid (motion-planning/AbstractValidityChecker.cpp:124-133)
 * with VoxelBackboneValidityChecker::voxelize_impl + collides
 * (VoxelBackboneValidityChecker.h:49-57, AbstractVoxelValidityChecker.h:55-57): FK, converged,
 * tendon-length limits, self-collision, backbone voxelisation vs the obstacle grid.  Replaces the
 * loops at motion-planning/VoxelCachedLazyPRM.cpp:1448-1455 and :1584-1591.
 *   valid_bits  ceil(n/64) words
 *   tips        n x 3 (optional)  fk_shape.p.back() (VoxelCachedLazyPRM.cpp:1438)
 *   flags       n (optional)      TR_FLAG_* bits                                            */
int tr_validate_batch(tr_ctx *ctx, const double *states, int64_t n,
                      uint64_t *valid_bits, double *tips, uint8_t *flags);
int tr_validate_batch_dev(tr_ctx *ctx, const double *d_states, int64_t n,
                          uint64_t *d_valid_bits, double *d_tips, uint8_t *d_flags, void *stream);

/* ---- createRoadmap phase 1 on the device: rejection sampling of valid vertices ------------- */

/* The reference's vertex phase (motion-planning/VoxelCachedLazyPRM.cpp:1415-1455) draws every new milestone in a loop
 *   sampleUniform -> fk -> is_valid_shape -> voxelize -> collides   until a state is accepted (:1441-1444),
 * from thread-local OMPL generators (not reproducible).  Here the candidates form ONE sequence that is a pure function of
 * (seed, candidate index): Philox-4x32-10 with counter (index, coordinate pair) and key = seed gives 53-bit uniforms u,
 * state[d] = lo[d] + u * (hi[d] - lo[d]) (product and sum rounded separately).  lo / hi: S doubles each, or both NULL for the
 * planner's state-space bounds (Problem.cpp:101-163: tension [0, max_tension_i], rotation [-pi, pi), retraction [0, L]).
 * The same candidates come out for every batch size, every GPU count (rank g takes a contiguous index range) and from the
 * host mirror (interactive-rate-tendons_amd/distributed.py: candidate_states).  States are generated in HBM and never
 * uploaded. */
int tr_candidate_states(tr_ctx *ctx, uint64_t seed, uint64_t first, int64_t count, const double *lo, const double *hi,
                        double *states /* count x S, host */);
int tr_candidate_states_dev(tr_ctx *ctx, uint64_t seed, uint64_t first, int64_t count, const double *lo, const double *hi,
                            double *d_states /* count x S */, void *stream);

/* tr_validate_batch_dev on candidates [first, first + count) of the sequence, generated on the device (first % 64 == 0 so
 * that shards are whole mask words).  The shard form of config 4: the mask stays on the device for the all-gather. */
int tr_validate_candidates_dev(tr_ctx *ctx, uint64_t seed, uint64_t first, int64_t count, const double *lo, const double *hi,
                               uint64_t *d_valid_bits, double *d_tips, uint8_t *d_flags, void *stream);

/* Order-preserving compaction on the device: rows i of d_rows (row_doubles doubles each) whose mask bit is set go, in
 * ascending i, to d_rows_out (at most `capacity` rows); d_index_out (optional) receives their i.  *n_out = number of set
 * bits among the first `count` (rows written = min(*n_out, capacity)).  With the gathered vertex mask and the regenerated
 * candidates this yields the same vertex array on every rank.  Synchronises `stream` (one counter read-back). */
int tr_compact_rows_dev(tr_ctx *ctx, const uint64_t *d_mask, int64_t count, const double *d_rows, int32_t row_doubles,
                        int64_t capacity, double *d_rows_out, int64_t *d_index_out, int64_t *n_out, void *stream);

/* The whole loop: candidates first_candidate, first_candidate + 1, ... are validated batch by batch (fk_verdict; batch sizes
 * follow the acceptance rate seen so far) and the first n_want VALID ones are returned in candidate order -- the accepted set
 * does not depend on the batch sizes.  One 32-byte counter read-back per batch is the only host traffic of the _dev form.
 *   states  n_want x S     tips  n_want x 3 (optional)     index  n_want (optional): candidate index of each vertex
 *   *n_accepted  vertices returned (< n_want only when max_candidates candidates did not yield enough; 0 = the default
 *                64 n_want + 2^20)        *n_tried  candidates consumed up to and including the last accepted one         */
int tr_sample_valid_vertices(tr_ctx *ctx, uint64_t seed, uint64_t first_candidate, const double *lo, const double *hi,
                             int64_t n_want, int64_t max_candidates, double *states, double *tips, int64_t *index,
                             int64_t *n_accepted, int64_t *n_tried);
int tr_sample_valid_vertices_dev(tr_ctx *ctx, uint64_t seed, uint64_t first_candidate, const double *lo, const double *hi,
                                 int64_t n_want, int64_t max_candidates, double *d_states, double *d_tips, int64_t *d_index,
                                 int64_t *n_accepted, int64_t *n_tried, void *stream);

/* The second stage alone, on caller-supplied backbone shapes (is_valid_shape + voxelize +
 * collides on given TendonResults: AbstractValidityChecker.cpp:99-122).  Device pointers, SoA as
 * produced by tr_fk_batch_dev.  d_n_points is ignored (a robot without retraction has all P points in
 * every configuration; kept in the signature for symmetry).  check_voxels = 0 skips the
 * obstacle test (the "is_valid_shape only" predicate used while voxelising edges,
 * VoxelBackboneMotionValidator.cpp:29-36).  Retraction-enabled robots: the _retraction_ form below. */
int tr_validate_shapes_dev(tr_ctx *ctx, int64_t n, int64_t ld,
                           const double *d_px, const double *d_py, const double *d_pz,
                           const int32_t *d_n_points, const double *d_Li, const uint8_t *d_converged,
                           int check_voxels, uint64_t *d_valid_bits, uint8_t *d_flags, void *stream);
/* ... on the outputs of tr_fk_batch_retraction_dev (per-configuration point counts and home lengths). */
int tr_validate_shapes_retraction_dev(tr_ctx *ctx, int64_t n, int64_t ld,
                                      const double *d_px, const double *d_py, const double *d_pz,
                                      const int32_t *d_n_points, const double *d_Li, const double *d_home_Li,
                                      const uint8_t *d_converged, int check_voxels,
                                      uint64_t *d_valid_bits, uint8_t *d_flags, void *stream);

/* ---- motion validity: MotionValidator::checkMotion -------------------------------------- */

typedef struct {
  double min_tension_change;     /* motion-planning/Problem.h:59 (0.02)   */
  double min_rotation_change;    /* :61 (0.01)                            */
  double min_retraction_change;  /* :62 (0.0001)                          */
} tr_space_params;

/* Batched AbstractVoxelMotionValidator::checkMotion(s1, s2)
 * (motion-planning/AbstractVoxelMotionValidator.h:143-151 -> VoxelBackboneMotionValidator.cpp:41-81
 * -> VoxelEnvironment::voxelize_valid_backbone_motion, VoxelEnvironment.cpp:207-444), replacing
 * the loops at VoxelCachedLazyPRM.cpp:1520-1542 and :1621-1641.  a, b: n_edges x S host states.
 * n_fk (optional) receives the number of FK samples taken per edge.
 * An edge whose samples leave the voxel domain (std::domain_error in the reference) is reported
 * invalid and counted in *n_domain_errors (optional). */
int tr_validate_edges(tr_ctx *ctx, const tr_space_params *sp, const double *a, const double *b,
                      int64_t n_edges, uint64_t *valid_bits, int32_t *n_fk, int64_t *n_domain_errors);

/* The same for a roadmap: edge e joins states[edges[2e]] and states[edges[2e + 1]] (rows of one n_states x S
 * array, the graph's vertices).  Every vertex is evaluated once for all of its edges -- the reference's
 * checkMotion recomputes both end shapes for every edge (VoxelEnvironment.cpp:262-272) -- with identical
 * verdicts and n_fk (which keeps counting the two ends per edge, as the reference does). */
int tr_validate_edges_indexed(tr_ctx *ctx, const tr_space_params *sp, const double *states, int64_t n_states,
                              const int32_t *edges, int64_t n_edges, uint64_t *valid_bits, int32_t *n_fk,
                              int64_t *n_domain_errors);

/* Batched checkMotion(s1, s2, last_valid) (AbstractVoxelMotionValidator.h:153-169 ->
 * voxelize_until_invalid, VoxelBackboneMotionValidator.cpp:83-91): same verdict bits, plus per edge
 * last_valid_t = PartialVoxelization::t, the largest sampled interpolation parameter below the first
 * invalid sample (1.0 for a valid edge, 0.0 when already the start is invalid).  The caller obtains
 * last_valid.first with its own space->interpolate(s1, s2, t).  A sample is judged by the installed state
 * checker (tr_set_checker), as voxelize_until_invalid_impl does. */
int tr_validate_edges_last_valid(tr_ctx *ctx, const tr_space_params *sp, const double *a, const double *b,
                                 int64_t n_edges, uint64_t *valid_bits, double *last_valid_t, int32_t *n_fk);

/* Batched discrete edge check: VoxelBackboneDiscreteMotionValidator::generic_voxelize
 * (motion-planning/VoxelBackboneDiscreteMotionValidator.cpp:9-79, the loop of
 * ompl::base::DiscreteMotionValidator::checkMotion): samples a, interpolate(i / nd) for
 * i = 1 .. nd-1 with nd = validSegmentCount(a, b), then b; the edge is valid iff every sample is a
 * valid state.  last_valid_t (optional) = PartialVoxelization::t, n_fk (optional) = samples the
 * reference's sequential loop evaluates (it stops after the first invalid one).  With last_valid_t the call is
 * checkMotion(s1, s2, last_valid) and a sample is judged by the installed state checker; without it the call is
 * checkMotion(s1, s2): shape validity per sample plus the swept backbone volume (see tr_set_checker). */
int tr_validate_edges_discrete(tr_ctx *ctx, const tr_space_params *sp, const double *a, const double *b,
                               int64_t n_edges, uint64_t *valid_bits, double *last_valid_t, int32_t *n_fk);

/* ---- cached voxel sets vs obstacles: VoxelOctree::collides on roadmap caches ------------ */

/* Batched `obstacles.collides(*cached_voxels)` for roadmap vertices / edges
 * (motion-planning/VoxelCachedLazyPRM.cpp:2397-2411, :2497-2509, :2607-2631).  Items are sparse
 * block lists in CSR form: item i owns entries offsets[i]..offsets[i+1]-1 of (block_ids, masks);
 * block id = ((bx*Nb)+by)*Nb+bz.  hit_bits: ceil(n_items/64) words, bit set = collides. */
int tr_check_cached(tr_ctx *ctx, const uint32_t *block_ids, const uint64_t *masks,
                    const int64_t *offsets, int64_t n_items, uint64_t *hit_bits);
int tr_check_cached_dev(tr_ctx *ctx, const uint32_t *d_block_ids, const uint64_t *d_masks,
                        const int64_t *d_offsets, int64_t n_items, uint64_t *d_hit_bits, void *stream);

/* The same test for a SUBSET of the items: d_list[q] names an item, d_hit[q] (one byte) receives its verdict.
 * The lazy query loop (tr_roadmap_solve) validates the unknown items of all candidate paths of a round with one
 * launch of this -- the batched form of computeVertexValidity / computeEdgeValidity on cached sets
 * (motion-planning/VoxelCachedLazyPRM.cpp:2607-2631). */
int tr_check_cached_subset_dev(tr_ctx *ctx, const uint32_t *d_block_ids, const uint64_t *d_masks,
                               const int64_t *d_offsets, int64_t n_items, const int32_t *d_list, int64_t n_list,
                               uint8_t *d_hit, void *stream);

/* ---- robot voxel sets for roadmap caches: voxelizeVertex / voxelizeEdge -------------------------- */

/* Batched VoxelCachedLazyPRM::voxelizeVertex (motion-planning/VoxelCachedLazyPRM.cpp:2803-2837),
 * replacing the loop at :1704-1713: FK, is_valid_shape (no obstacle test), and for every valid shape
 * the voxel set voxelize_impl would return (VoxelBackboneValidityChecker.h:49-57) as a sparse list
 * of (block id, mask).  Results are CSR: offsets[n+1] is written here (item i owns
 * offsets[i]..offsets[i+1]-1, nothing for an invalid shape); the lists stay inside the context
 * until tr_voxelize_fetch copies them.  shape_valid_bits: ceil(n/64) words; tips optional n x 3.
 * Inside an item the blocks are distinct and ordered by block id. */
int tr_voxelize_batch(tr_ctx *ctx, const double *states, int64_t n, int64_t *offsets,
                      uint64_t *shape_valid_bits, double *tips);

/* Batched VoxelCachedLazyPRM::voxelizeEdge (:2879-2902 -> AbstractVoxelMotionValidator::voxelize,
 * AbstractVoxelMotionValidator.h:98-107), replacing the loop at :1751-1775: the swept-volume
 * voxelisation of every edge whose samples are all shape-valid (is_fully_valid); other edges get an
 * empty item and a 0 bit.  Same CSR + fetch protocol. */
int tr_voxelize_edges(tr_ctx *ctx, const tr_space_params *sp, const double *a, const double *b,
                      int64_t n_edges, int64_t *offsets, uint64_t *fully_valid_bits, int32_t *n_fk);

/* The same for a roadmap: edge e joins states[edges[2e]] and states[edges[2e + 1]]; every vertex is integrated and
 * voxelised ONCE for all of its edges (the loop at :1751-1775 recomputes both end shapes per edge).  Same sets. */
int tr_voxelize_edges_indexed(tr_ctx *ctx, const tr_space_params *sp, const double *states, int64_t n_states,
                              const int32_t *edges, int64_t n_edges, int64_t *offsets, uint64_t *fully_valid_bits,
                              int32_t *n_fk);

/* connectVertices and voxelizeEdge in one pass (createRoadmap adds an edge when checkMotion accepts it, :1491-1502, and
 * voxelises it, :1751-1775 -- two traversals of the same samples): the samples are tested against the obstacle grid as
 * tr_validate_edges_indexed tests them AND voxelised as tr_voxelize_edges_indexed voxelises them.  valid_bits are
 * checkMotion's verdicts; a valid edge owns the voxel set tr_voxelize_edges_indexed gives it, an invalid one nothing. */
int tr_connect_edges_indexed(tr_ctx *ctx, const tr_space_params *sp, const double *states, int64_t n_states,
                             const int32_t *edges, int64_t n_edges, int64_t *offsets, uint64_t *valid_bits, int32_t *n_fk);

/* The block lists of the last tr_voxelize_* call stay in device memory (they are produced there, and their consumers --
 * tr_check_cached_dev, tr_roadmap_set_caches_dev -- read them there); capacity must be >= its offsets[n] = tr_voxelize_count.
 * tr_voxelize_fetch copies them to host arrays; tr_voxelize_fetch_dev copies them device to device into arrays the caller
 * owns (complete when it returns; the next tr_voxelize_* call overwrites the store). */
int tr_voxelize_fetch(tr_ctx *ctx, uint32_t *block_ids, uint64_t *masks, int64_t capacity);
int tr_voxelize_fetch_dev(tr_ctx *ctx, uint32_t *d_block_ids, uint64_t *d_masks, int64_t capacity, void *stream);
int64_t tr_voxelize_count(const tr_ctx *ctx);

/* ---- nearest neighbours in state space ---------------------------- */

/* For every state its k nearest among the same n states in the reference's state-space metric
 * (CompoundStateSpace::distance with the subspace weights of motion-planning/Problem.cpp:112-152):
 * the neighbour lists connectionStrategy_(v) produces in createRoadmap phase 3
 * (motion-planning/VoxelCachedLazyPRM.cpp:1491-1502; KBoundedStrategy :1339 / KStarStrategy :1352).
 * Like nearestK on a structure that already contains v, row i starts with i itself at distance 0.
 * Entries farther than max_distance (KBoundedStrategy's bound; pass INFINITY for none) are -1 / inf.
 * idx, dist: n x k row-major, ascending distance.  Exact: every pair is either computed or excluded by a bound on one coordinate. */
int tr_knn(tr_ctx *ctx, const double *states, int64_t n, int32_t k, double max_distance,
           int32_t *idx, double *dist);

/* Layout and metric of the state space built by motion_planning::Problem::create_space_information
 * (motion-planning/Problem.cpp:101-163): number of tension dimensions, whether a rotation / retraction coordinate
 * follows, and the weights of those two subspaces in CompoundStateSpace::distance (tension weight 1). */
int tr_state_layout(const tr_ctx *ctx, int32_t *n_tendons, int32_t *has_rotation, int32_t *has_retraction);
int tr_space_weights(const tr_ctx *ctx, double *w_rotation, double *w_retraction);

/* ---- interactive queries on a cached roadmap (BASELINE config 5) -------------------------------------
 * motion_planning::VoxelCachedLazyPRM::solveWithRoadmap / constructSolution
 * (motion-planning/VoxelCachedLazyPRM.cpp:1977-2096, :2689-2771) for a BATCH of (start, goal) vertex pairs:
 * A* with the state-space distance as heuristic (astarSearch :2950-2976, costHeuristic :2773-2775) on the host
 * cores, validity of the candidate paths' vertices and edges from their cached voxel sets -- resident in HBM --
 * with one K4 launch per round over all queries' unknown items (computeVertexValidity / computeEdgeValidity
 * :2607-2631), invalid items leave the graph (removeVertices / removeEdge), repeat until every query has a valid
 * path or its start and goal are disconnected.  Accepted paths equal the reference's (equal cost; equal vertex
 * sequence unless two paths tie exactly).  The roadmap object borrows `ctx` (obstacle grid, device) and must be
 * destroyed before it. */
typedef struct tr_roadmap tr_roadmap;
typedef struct {
  int64_t rounds;          /* search / validate rounds of the last solve                        */
  int64_t items_checked;   /* cached sets tested against the obstacle grid (K4 work)            */
  int64_t astar_runs;      /* A* searches (>= queries: a query searches again after removals)    */
  int64_t expanded;        /* vertices taken off the open lists                                  */
} tr_roadmap_stats;
#define TR_QUERY_SOLVED        0
#define TR_QUERY_NO_PATH       1   /* start and goal are in different components (solveWithRoadmap :2026-2036) */
#define TR_QUERY_INVALID_START 2   /* ob::PlannerStatus::INVALID_START (solvePrep :2995-2998)                  */
#define TR_QUERY_INVALID_GOAL  3
/* Graph of a loaded roadmap (fromRoadmapParser :2357-2580): n_vertices x S states, n_edges index pairs,
 * edge weights (weightProperty_; NULL = state-space distance, what connectVertices stores :2857-2861). */
int tr_roadmap_create(tr_ctx *ctx, const double *states, int64_t n_vertices, const int32_t *edges,
                      const double *weights, int64_t n_edges, tr_roadmap **out);
/* Device buffers of a destroyed (or re-attached) roadmap are parked in a per-process cache for the next one -- at most 3 GiB /
 * 48 buffers; a failed allocation empties it first -- so free device memory does not return to its earlier level at once. */
void tr_roadmap_destroy(tr_roadmap *rm);
const char *tr_roadmap_last_error(const tr_roadmap *rm);
/* Upload the cached voxel sets (vertexVoxelsProperty_ / edgeVoxelsProperty_) as CSR block lists -- the output of
 * tr_voxelize_batch / tr_voxelize_edges* or of a .rmp file.  present bits (optional): a 0 bit = no cache, because
 * voxelizeVertex / voxelizeEdge found the shape invalid (:2803-2837, :2879-2902): such an item is invalid in every
 * environment. */
int tr_roadmap_set_caches(tr_roadmap *rm, const int64_t *v_offsets, const uint32_t *v_block_ids, const uint64_t *v_masks,
                          const uint64_t *v_present_bits, const int64_t *e_offsets, const uint32_t *e_block_ids,
                          const uint64_t *e_masks, const uint64_t *e_present_bits);
/* The same with the block ids / masks in device memory (the arrays tr_voxelize_fetch_dev filled): the caches of a roadmap
 * built on this GPU never cross PCIe.  Offsets and present bits are host arrays as above. */
int tr_roadmap_set_caches_dev(tr_roadmap *rm, const int64_t *v_offsets, const uint32_t *d_v_block_ids, const uint64_t *d_v_masks,
                              const uint64_t *v_present_bits, const int64_t *e_offsets, const uint32_t *d_e_block_ids,
                              const uint64_t *d_e_masks, const uint64_t *e_present_bits);
/* Landmark tables for the searches of tr_roadmap_solve: graph distances from n_landmarks extremal vertices over ALL edges
 * (one Dijkstra each, on n_threads host threads; 0 = the process's CPU share).  They sharpen A*'s heuristic -- the
 * reference's state-space distance (costHeuristic :2773-2775) -- by lower bounds that stay valid when invalid items leave
 * the graph, so the returned paths and costs are unchanged and far fewer vertices are expanded.  n_landmarks = 0 searches
 * with the reference's heuristic alone.  Without this call the first tr_roadmap_solve of >= 64 queries builds 16. */
int tr_roadmap_prepare(tr_roadmap *rm, int32_t n_landmarks, int32_t n_threads);
/* clearValidity (:1656-1663): everything unknown again, removed items back in the graph -- call it after the
 * obstacle grid of `ctx` changed (tr_set_grid / tr_grid_*). */
int tr_roadmap_clear_validity(tr_roadmap *rm);
/* Eager form of the loading loops (:2397-2411, :2486-2526): every cached set against the current grid in one K4
 * launch; afterwards no query finds an unknown item. */
int tr_roadmap_revalidate(tr_roadmap *rm, int64_t *n_invalid_vertices, int64_t *n_invalid_edges);
/* status per item: 0 unknown, 1 valid, 2 invalid (removed) */
int tr_roadmap_get_validity(tr_roadmap *rm, uint8_t *vertex_status /*[n_vertices]*/, uint8_t *edge_status /*[n_edges]*/);
/* The inverse of tr_roadmap_get_validity: what a builder already knows goes in (either array may be NULL = leave as it is).
 * createRoadmap with ValidateVertices / ValidateEdges leaves the items it accepted VALIDITY_TRUE
 * (vertexValidityProperty_ :1476, computeEdgeValidity :2621-2631), so the first query on such a roadmap tests nothing again. */
int tr_roadmap_set_validity(tr_roadmap *rm, const uint8_t *vertex_status /*[n_vertices]*/, const uint8_t *edge_status /*[n_edges]*/);
/* The batched query loop (lazy: only the items on candidate paths are tested, round by round -- until so many queries are still
 * open after a round that testing every cached set at once is cheaper than another round, see TENDON_HIP_LAZY_ONLY).  status[q] = TR_QUERY_*; cost[q] (optional) = path cost; path_offsets[n_queries + 1]:
 * query q's path (start ... goal) is entries path_offsets[q] .. path_offsets[q+1]-1 of the array
 * tr_roadmap_fetch_paths copies out.  n_threads = host threads for the A* searches (0 = the process's CPU share).
 * Validity discovered by a call is kept for the next one (as the reference's graph keeps it between queries). */
int tr_roadmap_solve(tr_roadmap *rm, const int32_t *starts, const int32_t *goals, int64_t n_queries, int32_t n_threads,
                     int32_t *status, double *cost, int64_t *path_offsets, tr_roadmap_stats *stats);
int tr_roadmap_fetch_paths(tr_roadmap *rm, int32_t *path_vertices, int64_t capacity);
/* Where the graph searches of the last tr_roadmap_solve ran (the A* of a round is done by the `roadmap_astar` kernel, one wave
 * per query, when the round has 512 queries or more -- see TENDON_HIP_SEARCH below -- and by the host threads otherwise; the
 * answers are the same):  out[0] searches finished by the kernel, out[1] searches the kernel handed back to the host threads
 * (over its pop budget, or a list full), out[2] searches the host threads took while the kernel ran (the ones expected to be
 * longest), out[3] times a search's open list moved entries between its LDS part and its HBM part, out[4] vertex expansions
 * by the kernel (those of searches it handed back included), out[5] vertex expansions by the host threads, out[6] queries answered
 * "no path" without a search because their end points lie in different components of the roadmap minus the items known invalid
 * (the reference's solutionComponent test, :2015-2044; labels recomputed on the device per round, see TENDON_HIP_COMPONENTS), out[7] times
 * a search on the device outgrew its table of per-vertex records and moved into a larger one from the shared pool. */
int tr_roadmap_search_stats(tr_roadmap *rm, int64_t out[8]);
/* The roadmap_astar launches of the last tr_roadmap_solve, timed with HIP events on the stream they ran on: out[0] their total
 * milliseconds, out[1] their number, out[2] the vertex expansions they did, out[3] the ALGORITHMIC bytes of one expansion on this
 * roadmap (the vertex's record and row header, and per arc: the arc, two validity bytes, the arc count, the neighbour's record read
 * and written, its state and landmark rows) -- bench.py turns them into the kernel's HBM roofline fraction. */
int tr_roadmap_profile(tr_roadmap *rm, double out[4]);
/* The device searches' state -- a table of per-vertex records per search in flight, a pool of larger ones, the per-round arrays: sized
 * by the largest round so far (a 512-query round ~0.9 GB, 3 072 searches in flight ~5.3 GB, whatever the roadmap's size), it stays with the
 * roadmap between calls.  tr_roadmap_release_search_state hands it back to the device (the adjacency rows stay; the next round of 512
 * queries or more allocates tables again, ~2 ms) and reports the bytes; tr_roadmap_search_state_bytes says what is held now.  When an
 * allocation anywhere in the library runs out of device memory, the tables of every roadmap that is not inside a call are released
 * the same way before the allocation is tried again. */
int tr_roadmap_release_search_state(tr_roadmap *rm, int64_t *bytes_released);
/* ... and setting it up ahead of the first batch (the adjacency rows and the tables for rounds of up to n_queries queries: a process's first
 * allocation of the full 5.3 GB takes ~0.15 s, which otherwise falls into the first tr_roadmap_solve of 512 queries or more).
 * TR_ERR_UNSUPPORTED when this roadmap's searches stay on the host threads (parallel edges, state size above the kernel's, no memory). */
int tr_roadmap_reserve_search_state(tr_roadmap *rm, int64_t n_queries);
/* Searches of the last tr_roadmap_solve that were answered by a parallel SWEEP on the device instead of A*: a host search that passes a
 * sixth of the graph in expansions (50 000 at least; TENDON_HIP_SEARCH_SWEEP=n overrides, 0 = never) is abandoned and its source's
 * distances are relaxed over all valid arcs at once until nothing changes -- the same cost bit for bit, the path walked back along
 * arcs that meet it exactly.  Only on roadmaps whose searches are set up on the device (a round of 512 queries or more has run, or
 * tr_roadmap_reserve_search_state). */
int tr_roadmap_search_sweeps(tr_roadmap *rm, int64_t *n);
int tr_roadmap_search_state_bytes(tr_roadmap *rm, int64_t *bytes);

/* The connection loop itself (motion-planning/VoxelCachedLazyPRM.cpp:1491-1502: for every vertex v and every neighbour n
 * of connectionStrategy_(v), `if (!getEdge(v, n)) connectVertices(v, n)`): the undirected edge set of the k-nearest
 * table -- pairs (lo, hi), lo < hi, each once, ordered by (lo, hi) -- built on the device (sort + unique of the pair
 * keys); only the edge list crosses PCIe.  k counts the vertex itself, as in tr_knn.  edges: capacity x 2 int32; *n_edges
 * receives the number of edges (if it exceeds capacity only the first `capacity` are written: at most n (k - 1) edges exist -- n k when k or more states coincide, a row then need not hold its own vertex). */
int tr_knn_edges(tr_ctx *ctx, const double *states, int64_t n, int32_t k, double max_distance, int32_t *edges,
                 int64_t capacity, int64_t *n_edges);
/* Device-resident form of the connection loop and of the edge checks after it: d_states (n x state_size doubles) and d_edges
 * (capacity x 2 int32) are device arrays on the context's GPU, so a roadmap whose vertices were produced there (sampled or
 * filtered on the device) is connected and validated without its states or its edge list crossing PCIe -- only *n_edges, and
 * from tr_validate_edges_indexed_dev the mask words (d_valid_bits: (n_edges + 63) / 64 words; d_n_fk: n_edges counts or NULL),
 * which are device arrays too.  Same results, order and error codes as tr_knn_edges / tr_validate_edges_indexed; both calls
 * synchronise with the device on entry (the inputs may have been written on any stream) and before they return.
 * Where the host form falls back to gathering the end states on the host (more vertices than half the sample pool: beyond 2^23
 * vertices, or under a bounded pool) tr_validate_edges_indexed_dev brings its inputs over and does the same: correct, not fast. */
int tr_knn_edges_dev(tr_ctx *ctx, const double *d_states, int64_t n, int32_t k, double max_distance, int32_t *d_edges,
                     int64_t capacity, int64_t *n_edges);
int tr_validate_edges_indexed_dev(tr_ctx *ctx, const tr_space_params *sp, const double *d_states, int64_t n_states,
                                  const int32_t *d_edges, int64_t n_edges, uint64_t *d_valid_bits, int32_t *d_n_fk,
                                  int64_t *n_domain_errors);
/* Vertices that have just passed the vertex phase need not be integrated again by the edge call (every indexed edge call otherwise
 * evaluates each vertex once for its signature: 1.4 ms per 10^5 vertices, and the same on every rank when the edge list is sharded):
 * tr_validate_candidates_sig_dev is tr_validate_candidates_dev that also writes each candidate's backbone cell signature, a row of
 * tr_signature_words(ctx) uint32 (an even number; meaningful where the candidate's bit is set); compact the rows of the accepted
 * candidates like their states (tr_compact_rows_dev with row_doubles = tr_signature_words / 2) and pass them as d_vertex_sig
 * (n_states rows) to tr_validate_edges_indexed_sig_dev, which then treats every vertex as valid and gives the verdicts, counts and
 * errors of tr_validate_edges_indexed_dev.  tr_signature_words returns 0 -- and the two calls TR_ERR_UNSUPPORTED -- for retraction
 * robots, under TR_CHECKER_SPHERES and on schedules other than the default verdict-only one. */
int tr_signature_words(const tr_ctx *ctx);
/* ... and tr_sample_valid_vertices_dev that also returns the accepted vertices' rows (d_sig: n_want x tr_signature_words uint32, row i
 * belongs to d_states row i): createRoadmap's "sample until N are valid" feeding the edge phase directly. */
int tr_sample_valid_vertices_sig_dev(tr_ctx *ctx, uint64_t seed, uint64_t first_candidate, const double *lo, const double *hi,
                                     int64_t n_want, int64_t max_candidates, double *d_states, double *d_tips, int64_t *d_index,
                                     uint32_t *d_sig, int64_t *n_accepted, int64_t *n_tried, void *stream);
int tr_validate_candidates_sig_dev(tr_ctx *ctx, uint64_t seed, uint64_t first, int64_t count, const double *lo, const double *hi,
                                   uint64_t *d_valid_bits, double *d_tips, uint32_t *d_sig, void *stream);
int tr_validate_edges_indexed_sig_dev(tr_ctx *ctx, const tr_space_params *sp, const double *d_states, int64_t n_states,
                                      const uint32_t *d_vertex_sig, const int32_t *d_edges, int64_t n_edges,
                                      uint64_t *d_valid_bits, int32_t *d_n_fk, int64_t *n_domain_errors);

/* The signature rows on the wire.  They are the one large collective of a sharded roadmap build (576 B per vertex at 129 backbone
 * points: 346 MB per build of 6 x 10^5 vertices), and they are redundant: consecutive backbone points lie at most dL <= one voxel edge
 * apart (the checker's constructor enforces it), so consecutive cells of a row differ by -1, 0 or +1 per axis.  A packed row =
 * the first point's word, then 6 bits per further point, padded to tr_signature_packed_words(ctx) uint32 (an even number: 26 = 104 B
 * at 129 points).  tr_pack_signatures_dev packs n_rows rows (d_sig: n_rows x tr_signature_words) and reports in *n_uncodable the rows
 * it could not code (a point outside the voxel domain, or a larger step: never for vertices that passed the vertex phase) -- the
 * caller then sends the rows as they are; it synchronises `stream` (one counter read-back).  tr_unpack_signatures_dev restores the rows
 * word for word (the padding words of a row beyond the backbone's points come back as zeros; nothing reads them). */
int tr_signature_packed_words(const tr_ctx *ctx);
int tr_pack_signatures_dev(tr_ctx *ctx, const uint32_t *d_sig, int64_t n_rows, uint32_t *d_packed, int64_t *n_uncodable, void *stream);
int tr_unpack_signatures_dev(tr_ctx *ctx, const uint32_t *d_packed, int64_t n_rows, uint32_t *d_sig, void *stream);

/* The same neighbour lists for a RANGE of the states as queries (all n states remain the candidates): rows
 * first_query .. first_query + n_queries - 1 of tr_knn's tables, whatever the range -- one rank's share when the connection
 * loop of a large roadmap is spread over several GPUs.  idx / dist: n_queries x k. */
int tr_knn_range(tr_ctx *ctx, const double *states, int64_t n, int64_t first_query, int64_t n_queries, int32_t k,
                 double max_distance, int32_t *idx, double *dist);
/* The undirected edge set of a k-nearest table given by the caller (n x k indices, -1 = none; e.g. the ranks' tr_knn_range
 * rows gathered): what tr_knn_edges builds from its own table, with the same order and capacity rule. */
int tr_knn_table_edges(tr_ctx *ctx, const int32_t *idx, int64_t n, int32_t k, int32_t *edges, int64_t capacity, int64_t *n_edges);
/* Device-resident forms of the two calls above, for the sharded connection loop of a roadmap whose vertices are in HBM: the rank's rows
 * go to a device array (all-gathered there), the gathered table's edge list is written straight into d_edges.  Same rows, edges,
 * order and error codes; both synchronise with the device on entry and before they return. */
int tr_knn_range_dev(tr_ctx *ctx, const double *d_states, int64_t n, int64_t first_query, int64_t n_queries, int32_t k,
                     double max_distance, int32_t *d_idx);
int tr_knn_table_edges_dev(tr_ctx *ctx, const int32_t *d_idx, int64_t n, int32_t k, int32_t *d_edges, int64_t capacity,
                           int64_t *n_edges);

/* k of the PRM* connection strategy for a roadmap of n_milestones vertices (og::KStarStrategy as installed by
 * setStarConnectionStrategy, motion-planning/VoxelCachedLazyPRM.cpp:1346-1356): ceil((e + e / dim) * ln(n)), dim =
 * state dimension.  createRoadmap connects after all vertices are in place, so this k applies to every vertex of the
 * batch: pass k + 1 to tr_knn (row i holds i itself first).  -1 on bad arguments. */
int tr_kstar_k(const tr_ctx *ctx, int64_t n_milestones);

/* ---- instrumentation ---------------------------------------------------------------------- */

/* Time the last `which` kernel launches with HIP events on the stream they ran on.
 * tr_profile_begin enables event recording around every kernel launched by this context;
 * tr_profile_read returns, per kernel slot, launches and total milliseconds since begin.
 * slots: 0 = fk_rk4_batch, 1 = backbone_voxel_sweep, 2 = cached_blocks_vs_grid, 3 = edge helpers,
 * 4 = fk_sweep_fused (K1 + K2 in one launch over stored points: edge samples, voxel caches, the sphere checker, and the
 *     fallback pass of the verdict path), 5 = fk_verdict (the verdict-only kernel of tr_validate_batch* and of the edge
 *     samples; fk_verdict_retract for retraction-enabled robots -- the ordering of its batch by backbone length runs
 *     outside the slot) */
#define TR_PROFILE_SLOTS 6
int tr_profile_begin(tr_ctx *ctx);
int tr_profile_read(tr_ctx *ctx, int64_t launches[TR_PROFILE_SLOTS], double total_ms[TR_PROFILE_SLOTS]);
int tr_profile_end(tr_ctx *ctx);

/* debug / A-B switches (tests): bit0 = brute-force O(P^2) self-collision instead of the
 * conservative-skip sweep (verdicts must be identical); bit1 = no milestone proof of "no self collision": every
 * configuration takes the exact pairwise sweep (the fallback pass of the verdict path, the in-wave sweep of the edge queue);
 * bit2 = no dilated-grid fast path in the voxel walk.  Verdicts, flags and FK counts are the same under every combination. */
int tr_set_debug(tr_ctx *ctx, uint32_t bits);

/* How the last tr_validate_edges_indexed* call of this context was scheduled (instrumentation; tests assert on it):
 *   stats[0]  FK samples the edge queue took (0: the call ran on the level-synchronous lanes)
 *   stats[1]  rounds: batches of up to 64 samples a persistent wave integrated
 *   stats[2]  samples whose self-collision test took the exact pairwise sweep inside the queue
 *   stats[3]  flags the queue ended with: 0 = complete; 1 = sample pool too small, 4 = an edge level of more than 2048 intervals
 *             (in both cases the lanes then took the call); 2 = a wait made no progress (the call returned TR_ERR_RUNTIME)
 * The edge queue (csrc/edge_queue_kernel.hpp) is ONE persistent launch over a device work queue with a barrier per edge instead of
 * one per level of all edges: same verdicts, FK counts and domain-error count as the lanes, bit for bit. */
int tr_edge_schedule_last(const tr_ctx *ctx, uint32_t stats[4]);

/* Environment switches, read ONCE by tr_create (a context keeps what it read).  They exist for A/B measurements and for
 * tests that must reach a rarely taken path; none of them changes a result, and production code sets none of them.
 *   TENDON_HIP_FUSED=0|1|2          schedule of tr_validate_batch*: 2 (default) verdict-only kernel, 1 K1 + K2 as one kernel over
 *                                   stored points, 0 separate launches (the edge queue, the signature hand-over and
 *                                   tr_sample_valid_vertices* need 2 and report TR_ERR_UNSUPPORTED or fall back otherwise)
 *   TENDON_HIP_EDGE_QUEUE=0|1       unset: the device-resident indexed edge checks (tr_validate_edges_indexed_dev / _sig_dev) go through
 *                                   the edge queue where it applies (no retraction, schedule 2), the host-array form through the
 *                                   level-synchronous lanes; 1: both through the queue; 0: both through the lanes
 *   TENDON_HIP_EDGE_QUEUE_WAVES=n   persistent workgroups of the edge queue (default: what the device holds at once)
 *   TENDON_HIP_EDGE_LANES=1..4      exactly that many lanes of the level-synchronous edge bisection (default: by the edge count)
 *   TENDON_HIP_EDGE_LANE_GUESS=x    samples per edge assumed when a lane is given its share of the pool (tests: a small value
 *                                   provokes the overflow path)
 *   TENDON_HIP_EDGE_POOL=n          upper bound of the FK sample pool (tests: forces chunking / the overflow paths)
 *   TENDON_HIP_FB_CAP=n             columns of the fallback pass's point workspace (default: one resident round of waves)
 *   TENDON_HIP_RETRACT_SORT=n       retraction robots: batches of at least n configurations are ordered by backbone length
 *                                   (0 = never); TENDON_HIP_RETRACT_KBEGIN_OFF: no per-wave loop start in that order
 *   TENDON_HIP_CH_SCALE=x           milestone spacing of the self-collision proof in robot radii (default 2)
 * Read per call (the tests switch between two paths inside one process):
 *   TENDON_HIP_KNN=lanes            neighbour search by the lane-per-query kernel for every k (default: wave per query up to k = 64);
 *                                   TENDON_HIP_KNN_CELL=x scales its cell width (tuning)
 *   TENDON_HIP_MERGE=sort           edge voxel sets merged by the segmented sort (default: the per-edge LDS table);
 *                                   TENDON_HIP_MERGE_MAXLOAD=n bounds that table's load (tests: reaches the overflow fallback)
 *   TENDON_HIP_LANDMARKS=...        landmark choice of tr_roadmap_prepare (csrc/roadmap.hip; tuning)
 *   TENDON_HIP_SEARCH=host|device   graph searches of tr_roadmap_solve: unset = a round of 512 or more queries is shared between the
 *                                   roadmap_astar kernel and the host threads (below), a smaller round is the host threads'; host =
 *                                   the host threads always; device = the kernel always, whole rounds, no budget (tests)
 *   TENDON_HIP_SEARCH_HOST_SHARE=p  per cent of a shared round's searches (the ones with the most distant end points) that the host
 *                                   threads take while the kernel runs (default: starts at 1 and follows the two sides' times per roadmap)
 *   TENDON_HIP_SEARCH_BUDGET=n      expansions after which the kernel hands a search back: the host threads start on it at once, while
 *                                   the kernel is still running (a word per query in pinned memory tells them) (default: starts at 6500,
 *                                   or a sixteenth of the roadmap's vertices if that is more, and doubles per roadmap while more than
 *                                   one search in fifty comes back; 0 none)
 *   TENDON_HIP_SEARCH_K=1..12       vertices the kernel takes off a search's open list per step, at most (default 12: as many as their
 *                                   arcs fill the 64 lanes of the step's two passes; 1 = the host's order)
 *   TENDON_HIP_SEARCH_SWEEP=n       expansions after which a host search is handed to the device sweep (tr_roadmap_search_sweeps; 0 never)
 *   TENDON_HIP_SEARCH_SLOTS=n       searches in flight on the device, at most (default: what it holds: 12 waves per CU; a round of
 *                                   fewer queries allocates tables for that many only)
 *   TENDON_HIP_SEARCH_LC0=8..14     log2 of the per-vertex records a search in flight owns (default 12: 176 KiB per slot with its far list;
 *                                   the searches' state does not depend on the roadmap's size); TENDON_HIP_SEARCH_POOL=a,b,c: shared tables of
 *                                   4 / 16 / 64 times that size for the searches that outgrow it (default slots, slots / 4, slots / 64: 5.3 GB in all at 3 072 slots; a search
 *                                   that finds none free is handed back to the host threads).  Both read when a roadmap's first large
 *                                   round sets the searches up (tests reach the growth and hand-back paths with small values)
 *   TENDON_HIP_LAZY_ONLY=1          tr_roadmap_solve never looks at items off the candidate paths (default: when at least
 *                                   max(4, cached sets / 2^17) queries are still open after a round -- or a round's candidate paths hold a
 *                                   quarter as many items as there are cached sets, or the batch has a 256th as many queries as there
 *                                   are cached sets to begin with --, every cached set is tested in one
 *                                   launch and the next round is the last; same answers; the rules read counts, not clocks, so
 *                                   rounds / items_checked / the validity left behind are reproducible; `expanded` depends on which
 *                                   side -- kernel or host threads -- ran a search)
 *   TENDON_HIP_COMPONENTS=0|1       component labels of the roadmap minus the invalid items (queries across components are answered "no
 *                                   path" without a search): unset = from the moment they would have paid on this roadmap (a search
 *                                   walked 2000 vertices in vain, or the kernel handed searches back), 1 = every round of 64 queries
 *                                   or more, 0 = never (a query with an unreachable goal is searched to exhaustion)
 *   TENDON_HIP_PIPE_LOG2=n          chunk size 2^n of tr_validate_batch's host-buffer pipeline (tuning)
 * Output on stderr only: TENDON_HIP_EDGE_TIMING, TENDON_HIP_VOX_TIMING, TENDON_HIP_ROADMAP_TIMING, TENDON_HIP_SEARCH_STATS (where a
 * round's searches ran and for how long), TENDON_HIP_SEARCH_HIST (expansions per search, host searches only). */

#ifdef __cplusplus
}
#endif
#endif /* TENDON_HIP_H */
