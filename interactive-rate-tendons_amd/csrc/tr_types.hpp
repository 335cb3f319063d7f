// tr_types.hpp -- plain structs shared by host and device code of libtendon_hip.so.
#pragma once
#include <stdint.h>

#define TRK_MAX_TENDONS 8
#define TRK_MAX_COEF    8

// Robot constants handed to kernels by value (lands in SGPRs via the kernarg segment).
// Reference: tendon/TendonRobot.h:52-58, get_stiffness_matrices tendon/TendonRobot.cpp:105-148.
struct RobotK {
  double ks0, ks2;            // K_se = diag(G*Ar, G*Ar, E*Ar)
  double kb0, kb2;            // K_bt = diag(E*I, E*I, J*G)
  double iks0, iks2;          // K_se_inv diagonal (computed as 1/x on the host, as the reference does)
  double ikb0, ikb2;          // K_bt_inv diagonal
  double residual_threshold;
  double radius;              // robot radius r
  double L, dL;
  double min_len[TRK_MAX_TENDONS];
  double max_len[TRK_MAX_TENDONS];
  double home_Li[TRK_MAX_TENDONS];   // home_shape(0).L_i  (used when retraction is disabled)
  int32_t n_tendons, n_a, n_m;
  int32_t enable_rotation, enable_retraction;
  int32_t state_size;
  int32_t n_points;           // P = |t_range(0, L, dL)|
  int32_t pad_;
};

// Tendon routing polynomials and home-length classification, read by the retraction kernel
// (per-lane arc-length grid) through wave-uniform scalar loads.
struct PolyK {
  double C[TRK_MAX_TENDONS][TRK_MAX_COEF];   // theta_i(t) coefficients (tendon/TendonSpecs.h:26)
  double D[TRK_MAX_TENDONS][TRK_MAX_COEF];   // rho_i(t) coefficients   (tendon/TendonSpecs.h:27)
  double helix_scale[TRK_MAX_TENDONS];       // sqrt(1 + d0^2 c1^2)      (tendon/TendonRobot.cpp:289-292)
  int32_t home_kind[TRK_MAX_TENDONS];        // 0 straight, 1 helix, 2 general (numerical integration)
};

// One RK4 step of the shared arc-length grid (retraction disabled): produced on the host by the
// integrate_times stepping rule (Boost.odeint, call site tendon/TendonRobot.cpp:458-462).
struct StepK {
  double h;                   // step size
  int32_t obs;                // backbone-point index observed after this step, or -1
  int32_t pad_;
};

// Obstacle grid + environment rotation (collision/VoxelOctree.h:68-330 dense form,
// motion-planning/VoxelEnvironment.h:46-49).
struct GridK {
  double xmin, xmax, ymin, ymax, zmin, zmax;
  double dx, dy, dz;
  double inv_dx, inv_dy, inv_dz;     // 1/dx() etc. exactly as VoxelOctree::add_line forms them (:338)
  double inv_rot[9];                 // row-major
  int32_t N, Nb;
  int32_t rot_is_identity;
  int32_t pad_;
};

// Up to this many tendons the shared-grid kernel is held to 256 registers (two waves per SIMD); wider
// robots spill too much at that budget and run one wave per SIMD with the full 512-register file.
// Measured, ms per 2^19 configurations (two waves | one wave): N=4 5.7 | 7.0, N=5 6.9 | 7.7, N=6 8.4 | 9.7,
// N=8 12.9 | 11.0.
#ifndef TRK_K1_TWO_WAVE_MAXN
#define TRK_K1_TWO_WAVE_MAXN 6
#endif

namespace trk {

// Outputs of K1 (both kernels); null = not wanted.
struct FkOut {
  double *__restrict__ px, *__restrict__ py, *__restrict__ pz;   // [P][ld]
  double *__restrict__ R;                                        // [9][P][ld] or null
  double *__restrict__ L;                                        // [n] or null
  double *__restrict__ Li;                                       // [N][ld] or null
  double *__restrict__ tips;                                     // [n][3] or null
  uint8_t *__restrict__ converged;                               // [n] or null
  int32_t *__restrict__ n_points;                                // [n] or null
  double *__restrict__ home_Li;                                  // [N][ld] or null (retraction kernel only)
};

}  // namespace trk
