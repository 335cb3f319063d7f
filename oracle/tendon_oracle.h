/* tendon_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99, scalar fp64, no third-party code) of the reference's hot path:
 * Cosserat-rod tendon FK by RK4 -> validity predicate -> backbone voxelisation -> occupancy AND,
 * plus the swept-volume edge check.  Every function cites the reference file:line it follows
 * (paths relative to /root/reference/cpp/src).
 *
 * PARITY UNPINNED: the reference ships no tests / golden vectors and cannot be built in this
 * environment (Eigen, Boost.odeint, OMPL absent).  The oracle is pinned instead by analytic
 * known-answer tests, physics invariants and an independent high-order integrator (tests/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (interactive-rate-tendons_amd/) must never include, link or call anything here.
 */
#ifndef TENDON_ORACLE_H
#define TENDON_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_TENDONS 8
#define ORC_MAX_COEF    8

/* tendon/TendonRobot.h:52-58, tendon/BackboneSpecs.h:14-20, tendon/TendonSpecs.h:25-30 */
typedef struct {
  double r;                               /* robot radius (m)                        */
  double L, dL, ro, ri, E, nu;            /* BackboneSpecs                           */
  int    n_tendons;
  int    n_a, n_m;                        /* C.size(), D.size() of tendons[0]        */
  double C[ORC_MAX_TENDONS][ORC_MAX_COEF];
  double D[ORC_MAX_TENDONS][ORC_MAX_COEF];
  double max_tension[ORC_MAX_TENDONS];
  double min_length[ORC_MAX_TENDONS];
  double max_length[ORC_MAX_TENDONS];
  int    enable_rotation;
  int    enable_retraction;
  double residual_threshold;
} orc_robot;

/* tendon/TendonResult.h:17-40.  Caller provides arrays with capacity cap points. */
typedef struct {
  int     cap;          /* capacity (points) of t, p, R                               */
  int     n;            /* number of backbone points                                  */
  double *t;            /* [n]                                                        */
  double *p;            /* [n][3]                                                     */
  double *R;            /* [n][9]  column-major per matrix, as Eigen stores it        */
  double  L;
  double  L_i[ORC_MAX_TENDONS];
  double  u_i[3], u_f[3], v_i[3], v_f[3];
  int     converged;
  int     fp_iters;     /* iterations used by solve_initial_bending (diagnostic)      */
} orc_result;

/* Dense stand-in for collision::VoxelOctree (collision/VoxelOctree.h:68-330): same cell /
 * block / bit layout, blocks stored densely, block index ((bx*Nb)+by)*Nb+bz. */
typedef struct {
  int       N;          /* voxels per axis (4..512, power of two)                     */
  int       Nb;         /* blocks per axis = N/4                                      */
  double    xmin, xmax, ymin, ymax, zmin, zmax;
  double    dx, dy, dz;
  uint64_t *blocks;     /* [Nb*Nb*Nb]                                                 */
} orc_grid;

/* ---- kinematics ---- */
int  orc_state_size(const orc_robot *rb);
int  orc_t_range(double start, double end, double dt, double *out, int cap);
void orc_get_r_info(const orc_robot *rb, double t, double r[][3], double r_dot[][3], double r_ddot[][3]);
int  orc_solve_initial_bending(const orc_robot *rb, const double *tau, double s_start,
                               double v0[3], double u0[3]);
void orc_tendon_deriv(const orc_robot *rb, const double *tau, const double *x, double *dxdt, double t);
int  orc_tension_shape(const orc_robot *rb, const double *tau, double s_start, orc_result *res);
int  orc_home_shape(const orc_robot *rb, double s_start, orc_result *res);
int  orc_shape(const orc_robot *rb, const double *state, orc_result *res);
int  orc_home_shape_state(const orc_robot *rb, const double *state, orc_result *res);
void orc_rotate_z(orc_result *res, double theta);
double orc_base_residual(const orc_robot *rb, const double *tau, double s_start,
                         const double v0[3], const double u0[3]);

/* ---- validity predicate ---- */
int  orc_collides_self(const double *p, int n, double r);
int  orc_is_within_length_limits(const orc_robot *rb, const double *home_Li, const double *fk_Li);
int  orc_is_valid_shape(const orc_robot *rb, const orc_result *fk, const orc_result *home);
void orc_closest_st_segment(const double A[3], const double B[3], const double C[3], const double D[3],
                            double *s, double *t);

/* ---- voxels ---- */
orc_grid *orc_grid_create(int N);
orc_grid *orc_grid_empty_copy(const orc_grid *g);
void orc_grid_free(orc_grid *g);
void orc_grid_clear(orc_grid *g);
int  orc_grid_set_limits(orc_grid *g, double xmin, double xmax, double ymin, double ymax,
                         double zmin, double zmax);
uint64_t orc_bitmask(int x, int y, int z);
int  orc_grid_set_cell(orc_grid *g, int ix, int iy, int iz);
int  orc_grid_cell(const orc_grid *g, int ix, int iy, int iz);
int  orc_grid_is_in_domain(const orc_grid *g, double x, double y, double z);
void orc_grid_nearest_cell(const orc_grid *g, double x, double y, double z, int out[3]);
int  orc_grid_find_cell(const orc_grid *g, double x, double y, double z, int out[3]);
void orc_grid_add_point(orc_grid *g, double x, double y, double z);
void orc_grid_add_line(orc_grid *g, const double a[3], const double b[3]);
void orc_grid_add_piecewise_line(orc_grid *g, const double *pts, int n);
void orc_grid_add_sphere(orc_grid *g, const double c[3], double r);
int  orc_capsule_contains(const double a[3], const double b[3], double r, const double p[3]);
void orc_grid_add_capsule(orc_grid *g, const double a[3], const double b[3], double r);   /* VoxelOctree.cpp:471-515 */
void orc_grid_remove_interior(orc_grid *g, int keep_diagonal);
void orc_grid_dilate(orc_grid *g, int num, int use_diagonal);
void orc_grid_dilate_sphere(orc_grid *g, double r);
int  orc_grid_collides(const orc_grid *a, const orc_grid *b);
int  orc_grid_collides_point(const orc_grid *g, double x, double y, double z);
size_t orc_grid_nblocks(const orc_grid *g);
size_t orc_grid_ncells(const orc_grid *g);
int  orc_segment_aabox_intersect(const double A[3], const double B[3], const double C[3], const double D[3]);
void orc_rotate_points(const double inv_rot[9], double *pts, int n);

/* ---- one full state-validity check (AbstractValidityChecker::isValid) ----
 * flags out (optional): bit0 converged, bit1 length ok, bit2 no self collision, bit3 no voxel collision */
int  orc_is_valid_state(const orc_robot *rb, const orc_grid *obstacles, const double inv_rot[9],
                        const double *state, double tip[3], int *flags);

/* the same with VoxelValidityChecker's robot voxelisation: a sphere of radius rb->r at every backbone point */
int  orc_is_valid_state_spheres(const orc_robot *rb, const orc_grid *obstacles, const double inv_rot[9],
                                const double *state, double tip[3], int *flags);

/* Batch of state-validity checks, OpenMP over configurations when built with -fopenmp
 * (mirrors motion-planning/VoxelCachedLazyPRM.cpp:1448-1455). Returns threads used. */
int  orc_validate_batch(const orc_robot *rb, const orc_grid *obstacles, const double inv_rot[9],
                        const double *states, long n, uint8_t *valid, double *tips, int nthreads);
int  orc_fk_batch(const orc_robot *rb, const double *states, long n, double *p /*[n][P][3]*/,
                  double *L, double *L_i, uint8_t *converged, int P, int nthreads);

/* ---- edges (swept volume) ---- */
typedef struct {
  double min_tension_change;   /* motion-planning/Problem.h:59 */
  double min_rotation_change;  /* :61 */
  double min_retraction_change;/* :62 */
} orc_space_params;

unsigned orc_valid_segment_count(const orc_robot *rb, const orc_space_params *sp,
                                 const double *a, const double *b);
void orc_interpolate_state(const orc_robot *rb, const double *a, const double *b, double t, double *out);
double orc_state_distance(const orc_robot *rb, const double *a, const double *b);

/* VoxelEnvironment::voxelize_valid_backbone_motion + checkMotion.
 * swept (optional) receives the union voxelisation; n_fk (optional) the number of FK samples.
 * Returns 1 if the edge is valid (fully valid and swept volume free of obstacles). */
int  orc_check_motion(const orc_robot *rb, const orc_space_params *sp, const orc_grid *obstacles,
                      const double inv_rot[9], const double *a, const double *b,
                      orc_grid *swept, int *n_fk, int *is_fully_valid, double *last_valid_t);
int  orc_check_motion_until_invalid(const orc_robot *rb, const orc_space_params *sp, const orc_grid *obstacles,
                                    const double inv_rot[9], const double *a, const double *b,
                                    int *n_fk, double *last_valid_t);
int  orc_check_motion_discrete(const orc_robot *rb, const orc_space_params *sp, const orc_grid *obstacles,
                               const double inv_rot[9], const double *a, const double *b, int until_invalid,
                               int *n_fk, int *is_fully_valid, double *last_valid_t);
/* The last_valid forms with motion_planning::VoxelValidityChecker installed as the state checker (vc_spheres = 1):
 * `_vc->collides(shape)` of voxelize_until_invalid_impl (VoxelBackboneMotionValidator.cpp:83-91) then voxelises a
 * sphere of the robot radius at every backbone point (VoxelValidityChecker.h:18-26). */
int  orc_check_motion_until_invalid_vc(const orc_robot *rb, const orc_space_params *sp, const orc_grid *obstacles,
                                       const double inv_rot[9], const double *a, const double *b, int vc_spheres,
                                       int *n_fk, double *last_valid_t);
int  orc_check_motion_discrete_vc(const orc_robot *rb, const orc_space_params *sp, const orc_grid *obstacles,
                                  const double inv_rot[9], const double *a, const double *b, int until_invalid, int vc_spheres,
                                  int *n_fk, int *is_fully_valid, double *last_valid_t);
int  orc_check_motion_batch(const orc_robot *rb, const orc_space_params *sp, const orc_grid *obstacles,
                            const double inv_rot[9], const double *a, const double *b, long n,
                            uint8_t *valid, int32_t *n_fk, int nthreads);

/* cached-voxel re-validation: sparse (block id, mask) lists vs dense obstacle grid
 * (VoxelCachedLazyPRM.cpp:2397-2411 -> VoxelOctree::collides) */
/* Interactive queries on a cached roadmap: VoxelCachedLazyPRM::solveWithRoadmap / constructSolution, sequential,
 * one query at a time as the reference proceeds (see the .c file).  The CSR cache arrays are borrowed. */
typedef struct orc_roadmap orc_roadmap;
orc_roadmap *orc_roadmap_create(const orc_robot *rb, const double *states, long V, const int32_t *edges, const double *weights, long E,
                                const int64_t *v_off, const uint32_t *v_ids, const uint64_t *v_masks, const uint8_t *v_present,
                                const int64_t *e_off, const uint32_t *e_ids, const uint64_t *e_masks, const uint8_t *e_present);
void orc_roadmap_free(orc_roadmap *r);
void orc_roadmap_clear_validity(orc_roadmap *r);
void orc_roadmap_get_validity(const orc_roadmap *r, uint8_t *vstat, uint8_t *estat);
int  orc_roadmap_query(orc_roadmap *r, const orc_grid *obstacles, int start, int goal, int32_t *path_out, int cap,
                       double *cost_out, int *iterations_out, long *checked_out);
void orc_check_cached(const orc_grid *obstacles, const uint32_t *block_ids, const uint64_t *masks,
                      const int64_t *offsets, long n_items, uint8_t *hit);
/* export occupied blocks of g: returns count; ids/masks may be NULL to count only */
long orc_grid_export_blocks(const orc_grid *g, uint32_t *ids, uint64_t *masks, long cap);
long orc_grid_export_blocks_leaf_order(const orc_grid *g, uint32_t *ids, uint64_t *masks, long cap);   /* TreeNode.hxx:176-190 */
uint64_t orc_grid_block(const orc_grid *g, int bx, int by, int bz);
void orc_grid_set_block(orc_grid *g, int bx, int by, int bz, uint64_t value);
uint64_t orc_grid_union_block(orc_grid *g, int bx, int by, int bz, uint64_t value);

int  orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
