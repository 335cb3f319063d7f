/* tendon_oracle.c -- TEST INFRASTRUCTURE ONLY (see tendon_oracle.h).
 *
 * Plain-C restatement of the reference hot path.  Written to follow the reference's
 * expression structure (matrix products formed where the reference forms them, divisions
 * where it divides) rather than to be fast.  Build with -ffp-contract=off for golden
 * vectors (see oracle/Makefile).  PARITY UNPINNED: see header.
 */
#include "tendon_oracle.h"

#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * small 3-vector / 3x3 helpers (row-major m[r][c])
 * ---------------------------------------------------------------------------------------- */
typedef double M3[3][3];

static void m3_zero(M3 A) { memset(A, 0, sizeof(M3)); }
static void m3_copy(M3 D, const M3 S) { memcpy(D, S, sizeof(M3)); }
static void m3_add(M3 D, const M3 A, const M3 B) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) D[i][j] = A[i][j] + B[i][j];
}
static void m3_sub(M3 D, const M3 A, const M3 B) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) D[i][j] = A[i][j] - B[i][j];
}
static void m3_neg(M3 D, const M3 A) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) D[i][j] = -A[i][j];
}
static void m3_scale(M3 D, double s, const M3 A) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) D[i][j] = s * A[i][j];
}
static void m3_div(M3 D, const M3 A, double s) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) D[i][j] = A[i][j] / s;
}
static void m3_mul(M3 D, const M3 A, const M3 B) {
  M3 T;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      T[i][j] = A[i][0] * B[0][j] + A[i][1] * B[1][j] + A[i][2] * B[2][j];
  m3_copy(D, T);
}
static void m3_vec(double d[3], const M3 A, const double v[3]) {
  double t[3];
  for (int i = 0; i < 3; i++) t[i] = A[i][0] * v[0] + A[i][1] * v[1] + A[i][2] * v[2];
  d[0] = t[0]; d[1] = t[1]; d[2] = t[2];
}
static double v3_dot(const double a[3], const double b[3]) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
static double v3_norm(const double a[3]) { return sqrt(v3_dot(a, a)); }
static void v3_cross(double d[3], const double a[3], const double b[3]) {
  double t0 = a[1] * b[2] - a[2] * b[1];
  double t1 = a[2] * b[0] - a[0] * b[2];
  double t2 = a[0] * b[1] - a[1] * b[0];
  d[0] = t0; d[1] = t1; d[2] = t2;
}
/* Eigen 3.3 MatrixBase::normalized(): z = squaredNorm; z > 0 ? n / sqrt(z) : n */
static void v3_normalized(double d[3], const double a[3]) {
  double z = v3_dot(a, a);
  if (z > 0.0) {
    double s = sqrt(z);
    d[0] = a[0] / s; d[1] = a[1] / s; d[2] = a[2] / s;
  } else {
    d[0] = a[0]; d[1] = a[1]; d[2] = a[2];
  }
}

/* util/vector_ops.h:53-59 */
static void hat(M3 H, const double u[3]) {
  H[0][0] = 0;     H[0][1] = -u[2]; H[0][2] = u[1];
  H[1][0] = u[2];  H[1][1] = 0;     H[1][2] = -u[0];
  H[2][0] = -u[1]; H[2][1] = u[0];  H[2][2] = 0;
}

/* Eigen Matrix3d::inverse() (cofactor / adjugate formula, compute_inverse_size3_helper) */
static double cof(const M3 m, int i, int j) {
  int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1];
}
static void m3_inverse(M3 inv, const M3 m) {
  double c0[3] = { cof(m, 0, 0), cof(m, 1, 0), cof(m, 2, 0) };
  double det = c0[0] * m[0][0] + c0[1] * m[1][0] + c0[2] * m[2][0];
  double invdet = 1.0 / det;
  M3 T;
  T[0][0] = c0[0] * invdet; T[0][1] = c0[1] * invdet; T[0][2] = c0[2] * invdet;
  T[1][0] = cof(m, 0, 1) * invdet; T[1][1] = cof(m, 1, 1) * invdet; T[1][2] = cof(m, 2, 1) * invdet;
  T[2][0] = cof(m, 0, 2) * invdet; T[2][1] = cof(m, 1, 2) * invdet; T[2][2] = cof(m, 2, 2) * invdet;
  m3_copy(inv, T);
}

/* ------------------------------------------------------------------------------------------
 * kinematics
 * ---------------------------------------------------------------------------------------- */

/* tendon/TendonRobot.h:60-64 */
int orc_state_size(const orc_robot *rb) {
  return rb->n_tendons + (rb->enable_rotation ? 1 : 0) + (rb->enable_retraction ? 1 : 0);
}

/* tendon/TendonRobot.cpp:69-84 (t_range) over util/vector_ops.h:67-75 (range).
 * Returns the number of samples; writes at most cap of them. */
int orc_t_range(double start, double end, double dt, double *out, int cap) {
  int n = 0;
  for (double p = start; p <= end - (dt / 2); p += dt) {
    if (out && n < cap) out[n] = p;
    n++;
  }
  if (out && n < cap) out[n] = end;
  n++;
  if (out) {
    int m = n < cap ? n : cap;
    for (int i = 0; i < m; i++) out[i] = end - (out[i] - start);
    for (int i = 0; i < m / 2; i++) { double t = out[i]; out[i] = out[m - 1 - i]; out[m - 1 - i] = t; }
  }
  return n;
}

typedef struct { M3 K_bt, K_se, K_bt_inv, K_se_inv; } stiffness;

/* tendon/TendonRobot.cpp:105-148 */
static void get_stiffness_matrices(const orc_robot *rb, stiffness *K) {
  double ro2 = rb->ro * rb->ro, ri2 = rb->ri * rb->ri;
  double I = (1.0 / 4.0) * M_PI * (ro2 * ro2 - ri2 * ri2);
  double Ar = M_PI * (ro2 - ri2);
  double J = 2 * I;
  double Gmod = rb->E / (2 * (1 + rb->nu));
  m3_zero(K->K_bt); m3_zero(K->K_se); m3_zero(K->K_bt_inv); m3_zero(K->K_se_inv);
  K->K_bt[0][0] = rb->E * I; K->K_bt[1][1] = rb->E * I; K->K_bt[2][2] = J * Gmod;
  K->K_bt_inv[0][0] = 1 / (rb->E * I); K->K_bt_inv[1][1] = 1 / (rb->E * I); K->K_bt_inv[2][2] = 1 / (J * Gmod);
  K->K_se[0][0] = Gmod * Ar; K->K_se[1][1] = Gmod * Ar; K->K_se[2][2] = rb->E * Ar;
  K->K_se_inv[0][0] = 1 / (Gmod * Ar); K->K_se_inv[1][1] = 1 / (Gmod * Ar); K->K_se_inv[2][2] = 1 / (rb->E * Ar);
}

/* tendon/get_r_info.cpp:17-40 (get_poly_vecs) + :105-144 (get_r_info2) */
void orc_get_r_info(const orc_robot *rb, double t, double r[][3], double r_dot[][3], double r_ddot[][3]) {
  const int N_a = rb->n_a, N_m = rb->n_m;
  const int N_s = N_a > N_m ? N_a : N_m;
  double S[ORC_MAX_COEF], Sd[ORC_MAX_COEF], Sdd[ORC_MAX_COEF];
  S[0] = 1; Sd[0] = 0; Sdd[0] = 0;
  if (N_s >= 2) { S[1] = t; Sd[1] = 1; Sdd[1] = 0; }
  for (int i = 2; i < N_s; i++) {
    S[i] = t * S[i - 1];
    Sd[i] = i * S[i - 1];
    Sdd[i] = i * (i - 1) * S[i - 2];
  }
  for (int j = 0; j < rb->n_tendons; j++) {
    double C_a = 0, C_ad = 0, C_add = 0, D_m = 0, D_md = 0, D_mdd = 0;
    for (int i = 0; i < N_a; i++) {
      C_a += rb->C[j][i] * S[i]; C_ad += rb->C[j][i] * Sd[i]; C_add += rb->C[j][i] * Sdd[i];
    }
    for (int i = 0; i < N_m; i++) {
      D_m += rb->D[j][i] * S[i]; D_md += rb->D[j][i] * Sd[i]; D_mdd += rb->D[j][i] * Sdd[i];
    }
    double sa = sin(C_a), ca = cos(C_a);
    /* r = D_m (sin, cos, 0): note x = sin, y = cos (get_r_info.cpp:136) */
    r[j][0] = D_m * sa; r[j][1] = D_m * ca; r[j][2] = 0;
    r_dot[j][0] = D_md * sa + D_m * (ca * C_ad);
    r_dot[j][1] = D_md * ca + D_m * (-sa * C_ad);
    r_dot[j][2] = 0;
    r_ddot[j][0] = D_mdd * sa + 2 * D_md * (ca * C_ad) - D_m * (sa * C_ad * C_ad) + D_m * (ca * C_add);
    r_ddot[j][1] = D_mdd * ca + 2 * D_md * (-sa * C_ad) - D_m * (ca * C_ad * C_ad) + D_m * (-sa * C_add);
    r_ddot[j][2] = 0;
  }
}

/* tendon/solve_initial_bending.cpp:14-73; thresholds from tendon/TendonRobot.cpp:401-408.
 * Returns the iteration count at exit. */
static int solve_initial_bending_K(const orc_robot *rb, const stiffness *K, const double *tau,
                                   double s_start, double v[3], double u[3]) {
  const int iter_max = 1000;
  const double dv_threshold = 1e-9, du_threshold = 1e-9;
  const int N_t = rb->n_tendons;
  double r[ORC_MAX_TENDONS][3], r_dot[ORC_MAX_TENDONS][3], r_ddot[ORC_MAX_TENDONS][3];
  M3 rhat[ORC_MAX_TENDONS];
  v[0] = 0; v[1] = 0; v[2] = 1;
  u[0] = 0; u[1] = 0; u[2] = 0;
  orc_get_r_info(rb, s_start, r, r_dot, r_ddot);
  for (int k = 0; k < N_t; k++) hat(rhat[k], r[k]);
  int iters;
  for (iters = 0; iters < iter_max; ++iters) {
    M3 uhat; hat(uhat, u);
    double Ft[3] = {0, 0, 0}, Lt[3] = {0, 0, 0};
    for (int k = 0; k < N_t; ++k) {
      double w[3], pdu[3], tmp[3];
      m3_vec(w, uhat, r[k]);
      for (int i = 0; i < 3; i++) w[i] = w[i] + r_dot[k][i] + v[i];
      v3_normalized(pdu, w);
      for (int i = 0; i < 3; i++) Ft[i] -= tau[k] * pdu[i];
      /* Lt -= tau[k] * rhat[k] * pi_dot_unit  == (tau*rhat)*unit in Eigen; same value up to rounding */
      M3 trh; m3_scale(trh, tau[k], rhat[k]);
      m3_vec(tmp, trh, pdu);
      for (int i = 0; i < 3; i++) Lt[i] -= tmp[i];
    }
    double vm[3] = { v[0], v[1], v[2] - 1 };
    double n[3], m[3];
    m3_vec(n, K->K_se, vm);
    m3_vec(m, K->K_bt, u);
    double e1[3] = { n[0] - Ft[0], n[1] - Ft[1], n[2] - Ft[2] };
    double e2[3] = { m[0] - Lt[0], m[1] - Lt[1], m[2] - Lt[2] };
    double residual = sqrt(v3_dot(e1, e1) + v3_dot(e2, e2));
    if (residual < rb->residual_threshold) break;
    double v_new[3], u_new[3];
    m3_vec(v_new, K->K_se_inv, Ft); v_new[2] += 1;
    m3_vec(u_new, K->K_bt_inv, Lt);
    double dv[3] = { v_new[0] - v[0], v_new[1] - v[1], v_new[2] - v[2] };
    double du[3] = { u_new[0] - u[0], u_new[1] - u[1], u_new[2] - u[2] };
    if (v3_norm(dv) < dv_threshold * v3_norm(v) && v3_norm(du) < du_threshold * v3_norm(u)) break;
    for (int i = 0; i < 3; i++) { v[i] = v_new[i]; u[i] = u_new[i]; }
  }
  return iters;
}

int orc_solve_initial_bending(const orc_robot *rb, const double *tau, double s_start,
                              double v0[3], double u0[3]) {
  stiffness K; get_stiffness_matrices(rb, &K);
  return solve_initial_bending_K(rb, &K, tau, s_start, v0, u0);
}

/* tendon/tendon_deriv.cpp:60-87 (linsubsolve2): x = inv([[A,B],[C,D]]) [a;b] by block inverse */
static void linsubsolve2(const M3 A, const M3 B, const M3 C, const M3 D,
                         const double a[3], const double b[3], double x[6]) {
  M3 Ai, G, Gi, AiB, CAi, T, T2;
  m3_inverse(Ai, A);
  m3_mul(T, C, Ai); m3_mul(T, T, B);       /* (C*Ai)*B */
  m3_sub(G, D, T);
  m3_inverse(Gi, G);
  m3_mul(AiB, Ai, B);
  m3_mul(CAi, C, Ai);
  double Mi[6][6];
  m3_mul(T, AiB, Gi); m3_mul(T, T, CAi);   /* (AiB*Gi)*CAi */
  m3_add(T2, Ai, T);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Mi[i][j] = T2[i][j];
  m3_neg(T, AiB); m3_mul(T, T, Gi);        /* (-AiB)*Gi */
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Mi[i][3 + j] = T[i][j];
  m3_neg(T, Gi); m3_mul(T, T, CAi);        /* (-Gi)*CAi */
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Mi[3 + i][j] = T[i][j];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Mi[3 + i][3 + j] = Gi[i][j];
  double vv[6] = { a[0], a[1], a[2], b[0], b[1], b[2] };
  for (int i = 0; i < 6; i++) {
    double s = 0;
    for (int j = 0; j < 6; j++) s += Mi[i][j] * vv[j];
    x[i] = s;
  }
}

/* tendon/tendon_deriv.cpp:95-178.  State x (19+N): p[0:3], R[3:12] column-major, v[12:15],
 * u[15:18], x[18] backbone length, x[19+i] tendon i length. */
static void tendon_deriv_K(const orc_robot *rb, const stiffness *K, const double *tau,
                           const double *x, double *dxdt, double t) {
  const int N_t = rb->n_tendons;
  M3 R;  /* R(data+3) column-major: R[r][c] = data[3 + c*3 + r] */
  for (int c = 0; c < 3; c++) for (int rr = 0; rr < 3; rr++) R[rr][c] = x[3 + c * 3 + rr];
  const double v[3] = { x[12], x[13], x[14] };
  const double u[3] = { x[15], x[16], x[17] };
  M3 vhat, uhat; hat(vhat, v); hat(uhat, u);
  double r[ORC_MAX_TENDONS][3], r_dot[ORC_MAX_TENDONS][3], r_ddot[ORC_MAX_TENDONS][3];
  orc_get_r_info(rb, t, r, r_dot, r_ddot);

  M3 A, B, G, H; m3_zero(A); m3_zero(B); m3_zero(G); m3_zero(H);
  double a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
  double si_dot[ORC_MAX_TENDONS];
  for (int j = 0; j < N_t; j++) {
    M3 rhat, pdhat, Ai, Bi, Gi, Hi, T;
    double pd[3], w[3], w2[3], ai[3], bi[3];
    hat(rhat, r[j]);
    m3_vec(pd, uhat, r[j]);
    for (int i = 0; i < 3; i++) pd[i] = pd[i] + r_dot[j][i] + v[i];
    hat(pdhat, pd);
    si_dot[j] = v3_norm(pd);
    /* Ai = -tau*pdhat*pdhat / s^3 */
    m3_scale(T, -tau[j], pdhat);
    m3_mul(T, T, pdhat);
    m3_div(Ai, T, si_dot[j] * si_dot[j] * si_dot[j]);
    m3_mul(Bi, rhat, Ai);
    m3_neg(T, Ai); m3_mul(Gi, T, rhat);
    m3_neg(T, Bi); m3_mul(Hi, T, rhat);
    /* ai = Ai*(uhat*pd + uhat*r_dot + r_ddot) */
    m3_vec(w, uhat, pd);
    m3_vec(w2, uhat, r_dot[j]);
    for (int i = 0; i < 3; i++) w[i] = w[i] + w2[i] + r_ddot[j][i];
    m3_vec(ai, Ai, w);
    m3_vec(bi, rhat, ai);
    m3_add(A, A, Ai); m3_add(B, B, Bi); m3_add(G, G, Gi); m3_add(H, H, Hi);
    for (int i = 0; i < 3; i++) { a[i] += ai[i]; b[i] += bi[i]; }
  }
  double vmv[3] = { v[0], v[1], v[2] - 1 };
  double c[3], d[3], t1[3], t2[3];
  M3 T;
  /* c = -uhat*K_bt*u - vhat*K_se*vmv - b */
  m3_neg(T, uhat); m3_mul(T, T, K->K_bt); m3_vec(t1, T, u);
  m3_mul(T, vhat, K->K_se); m3_vec(t2, T, vmv);
  for (int i = 0; i < 3; i++) c[i] = t1[i] - t2[i] - b[i];
  /* d = -uhat*K_se*vmv - a */
  m3_neg(T, uhat); m3_mul(T, T, K->K_se); m3_vec(t1, T, vmv);
  for (int i = 0; i < 3; i++) d[i] = t1[i] - a[i];

  M3 KA, KH; m3_add(KA, K->K_se, A); m3_add(KH, K->K_bt, H);
  double xi[6];
  linsubsolve2(KA, G, B, KH, d, c, xi);

  double p_dot[3]; m3_vec(p_dot, R, v);
  M3 R_dot; m3_mul(R_dot, R, uhat);
  dxdt[0] = p_dot[0]; dxdt[1] = p_dot[1]; dxdt[2] = p_dot[2];
  for (int cc = 0; cc < 3; cc++) for (int rr = 0; rr < 3; rr++) dxdt[3 + cc * 3 + rr] = R_dot[rr][cc];
  dxdt[12] = xi[0]; dxdt[13] = xi[1]; dxdt[14] = xi[2];
  dxdt[15] = xi[3]; dxdt[16] = xi[4]; dxdt[17] = xi[5];
  dxdt[18] = v3_norm(v);
  for (int j = 0; j < N_t; j++) dxdt[19 + j] = si_dot[j];
}

void orc_tendon_deriv(const orc_robot *rb, const double *tau, const double *x, double *dxdt, double t) {
  stiffness K; get_stiffness_matrices(rb, &K);
  tendon_deriv_K(rb, &K, tau, x, dxdt, t);
}

/* Boost.odeint 1.65 runge_kutta4 (generic explicit RK, Butcher tableau a={{1/2},{0,1/2},{0,0,1}},
 * b={1/6,1/3,1/3,1/6}, c={0,1/2,1/2,1}); zero-coefficient terms dropped (adding 0*k is exact for
 * finite k).  Call site tendon/TendonRobot.cpp:458. */
#define ORC_NS (19 + ORC_MAX_TENDONS)
static void rk4_step(const orc_robot *rb, const stiffness *K, const double *tau,
                     double *x, double t, double h) {
  const int ns = 19 + rb->n_tendons;
  double k1[ORC_NS], k2[ORC_NS], k3[ORC_NS], k4[ORC_NS], xt[ORC_NS];
  const double a21 = h * 0.5, a32 = h * 0.5, a43 = h * 1.0;
  const double b1 = h * (1.0 / 6.0), b2 = h * (1.0 / 3.0), b3 = h * (1.0 / 3.0), b4 = h * (1.0 / 6.0);
  tendon_deriv_K(rb, K, tau, x, k1, t);
  for (int i = 0; i < ns; i++) xt[i] = x[i] + a21 * k1[i];
  tendon_deriv_K(rb, K, tau, xt, k2, t + h * 0.5);
  for (int i = 0; i < ns; i++) xt[i] = x[i] + a32 * k2[i];
  tendon_deriv_K(rb, K, tau, xt, k3, t + h * 0.5);
  for (int i = 0; i < ns; i++) xt[i] = x[i] + a43 * k3[i];
  tendon_deriv_K(rb, K, tau, xt, k4, t + h);
  for (int i = 0; i < ns; i++) x[i] = x[i] + b1 * k1[i] + b2 * k2[i] + b3 * k3[i] + b4 * k4[i];
}

/* tendon/TendonRobot.cpp:188-217 (PointForces::calc_point_forces) + TendonRobot.h:47-49
 * (residual), evaluated at the base with R = I as at TendonRobot.cpp:470-474. */
static double base_residual_K(const orc_robot *rb, const stiffness *K, const double *tau,
                              double s_start, const double v0[3], const double u0[3]) {
  double r[ORC_MAX_TENDONS][3], r_dot[ORC_MAX_TENDONS][3], r_ddot[ORC_MAX_TENDONS][3];
  orc_get_r_info(rb, s_start, r, r_dot, r_ddot);
  double vm[3] = { v0[0], v0[1], v0[2] - 1 };
  double n[3], m[3];
  m3_vec(n, K->K_se, vm);      /* R = I */
  m3_vec(m, K->K_bt, u0);
  double Ft[3] = {0, 0, 0}, Lt[3] = {0, 0, 0};
  for (int i = 0; i < rb->n_tendons; i++) {
    double w[3], unit[3], Fti[3], Lti[3];
    v3_cross(w, u0, r[i]);
    for (int k = 0; k < 3; k++) w[k] = w[k] + r_dot[i][k] + v0[k];
    v3_normalized(unit, w);
    for (int k = 0; k < 3; k++) Fti[k] = -tau[i] * unit[k];
    v3_cross(Lti, r[i], Fti);
    for (int k = 0; k < 3; k++) { Ft[k] += Fti[k]; Lt[k] += Lti[k]; }
  }
  double Fe[3] = { n[0] - Ft[0], n[1] - Ft[1], n[2] - Ft[2] };
  double Le[3] = { m[0] - Lt[0], m[1] - Lt[1], m[2] - Lt[2] };
  return sqrt(v3_dot(Fe, Fe) + v3_dot(Le, Le));
}

double orc_base_residual(const orc_robot *rb, const double *tau, double s_start,
                         const double v0[3], const double u0[3]) {
  stiffness K; get_stiffness_matrices(rb, &K);
  return base_residual_K(rb, &K, tau, s_start, v0, u0);
}

static void result_single_point(const orc_robot *rb, orc_result *res, double s_start) {
  res->n = 1;
  if (res->cap >= 1) {
    res->t[0] = s_start;
    res->p[0] = res->p[1] = res->p[2] = 0;
    memset(res->R, 0, 9 * sizeof(double));
    res->R[0] = res->R[4] = res->R[8] = 1;
  }
  res->L = 0;
  for (int i = 0; i < rb->n_tendons; i++) res->L_i[i] = 0;
  res->u_i[0] = res->u_i[1] = res->u_i[2] = 0;
  res->u_f[0] = res->u_f[1] = res->u_f[2] = 0;
  res->v_i[0] = res->v_i[1] = 0; res->v_i[2] = 1;
  res->v_f[0] = res->v_f[1] = 0; res->v_f[2] = 1;
  res->converged = 1;
  res->fp_iters = 0;
}

/* tendon/TendonRobot.cpp:325-500 (tension_shape) with Boost.odeint integrate_times semantics
 * (plain stepper overload): observer at every t[j]; between t[j] and t[j+1] steps of
 * min(dL, t[j+1]-cur) while t[j+1]-cur > DBL_EPSILON; each interval restarts at exactly t[j].
 * Returns 0 on success, -1 if res->cap is too small. */
int orc_tension_shape(const orc_robot *rb, const double *tau, double s_start, orc_result *res) {
  const int N = rb->n_tendons;
  if (s_start > rb->L) s_start = rb->L;           /* :359 */
  if (s_start == rb->L) { result_single_point(rb, res, s_start); return 0; }   /* :361-372 */

  stiffness K; get_stiffness_matrices(rb, &K);
  double v0[3], u0[3];
  res->fp_iters = solve_initial_bending_K(rb, &K, tau, s_start, v0, u0);
  for (int i = 0; i < 3; i++) { res->u_i[i] = u0[i]; res->v_i[i] = v0[i]; }

  double x[ORC_NS];
  memset(x, 0, sizeof(x));
  x[3] = 1; x[7] = 1; x[11] = 1;
  for (int i = 0; i < 3; i++) { x[12 + i] = v0[i]; x[15 + i] = u0[i]; }

  int P = orc_t_range(s_start, rb->L, rb->dL, NULL, 0);
  res->n = P;
  if (P > res->cap) return -1;
  orc_t_range(s_start, rb->L, rb->dL, res->t, res->cap);

  for (int j = 0; j < P; j++) {
    /* observer */
    res->p[3 * j + 0] = x[0]; res->p[3 * j + 1] = x[1]; res->p[3 * j + 2] = x[2];
    memcpy(res->R + 9 * j, x + 3, 9 * sizeof(double));
    if (j == P - 1) break;
    double cur = res->t[j];
    const double tn = res->t[j + 1];
    while (tn - cur > DBL_EPSILON) {
      double h = (rb->dL < tn - cur) ? rb->dL : (tn - cur);
      rk4_step(rb, &K, tau, x, cur, h);
      cur += h;
    }
  }
  res->L = x[18];
  for (int i = 0; i < N; i++) res->L_i[i] = x[19 + i];
  for (int i = 0; i < 3; i++) { res->v_f[i] = x[12 + i]; res->u_f[i] = x[15 + i]; }
  res->converged = (base_residual_K(rb, &K, tau, s_start, v0, u0) <= rb->residual_threshold);
  return 0;
}

/* tendon/TendonSpecs.cpp degree helpers (highest index with |coef| > eps, eps = 0) */
static int poly_degree(const double *c, int n) {
  for (int i = n - 1; i > 0; i--) if (fabs(c[i]) > 0.0) return i;
  return 0;
}
static double poly_at(const double *c, int n, double t) {   /* util/poly.h:9-17 */
  double val = 0.0, tpow = 1;
  for (int i = 0; i < n; i++) { val += c[i] * tpow; tpow *= t; }
  return val;
}

/* Defined-behaviour replacement for simpsons() (tendon/TendonRobot.cpp:160-178), which reads
 * vals[N] past the end and skips one interior sample.  Intent per its comment: composite
 * Simpson on equally spaced samples, trapezoid for the last interval when the number of
 * intervals is odd.  DEVIATION documented in DESIGN.md. */
static double simpsons_defined(const double *vals, int n, double dx) {
  if (n < 2) return 0.0;
  int nint = n - 1;
  double odd = 0.0;
  if (nint % 2 != 0) { odd = 0.5 * dx * (vals[n - 2] + vals[n - 1]); nint--; }
  if (nint == 0) return odd;
  double integral = vals[0] + vals[nint];
  for (int i = 1; i < nint; i++) integral += ((i % 2) ? 4 : 2) * vals[i];
  return odd + (integral * dx / 3.0);
}

/* tendon/TendonRobot.cpp:249-314 */
int orc_home_shape(const orc_robot *rb, double s_start, orc_result *res) {
  if (s_start < 0.0) s_start = 0.0;
  if (s_start > rb->L) s_start = rb->L;
  if (s_start == rb->L) { result_single_point(rb, res, s_start); return 0; }
  int P = orc_t_range(s_start, rb->L, rb->dL, NULL, 0);
  res->n = P;
  if (P > res->cap) return -1;
  orc_t_range(s_start, rb->L, rb->dL, res->t, res->cap);
  for (int j = 0; j < P; j++) {
    res->p[3 * j] = 0; res->p[3 * j + 1] = 0; res->p[3 * j + 2] = res->t[j] - s_start;
    memset(res->R + 9 * j, 0, 9 * sizeof(double));
    res->R[9 * j] = res->R[9 * j + 4] = res->R[9 * j + 8] = 1;
  }
  res->L = rb->L - s_start;
  for (int k = 0; k < 3; k++) { res->u_i[k] = res->u_f[k] = 0; res->v_i[k] = res->v_f[k] = (k == 2); }
  res->converged = 1;
  res->fp_iters = 0;
  for (int i = 0; i < rb->n_tendons; i++) {
    int rdeg = poly_degree(rb->D[i], rb->n_m), tdeg = poly_degree(rb->C[i], rb->n_a);
    if (rdeg == 0 && tdeg == 0) {
      res->L_i[i] = res->L;
    } else if (rdeg == 0 && tdeg == 1) {
      double d0 = rb->D[i][0], c1 = rb->C[i][1];
      res->L_i[i] = res->L * sqrt(1 + d0 * d0 * c1 * c1);
    } else {
      double Cdot[ORC_MAX_COEF] = {0}, Ddot[ORC_MAX_COEF] = {0};
      for (int k = 1; k < rb->n_a; k++) Cdot[k - 1] = k * rb->C[i][k];
      for (int k = 1; k < rb->n_m; k++) Ddot[k - 1] = k * rb->D[i][k];
      double *vals = (double *)malloc(sizeof(double) * (size_t)P);
      for (int j = 0; j < P; j++) {
        double tt = res->t[j];
        double dd = poly_at(Ddot, rb->n_m, tt), d = poly_at(rb->D[i], rb->n_m, tt), cd = poly_at(Cdot, rb->n_a, tt);
        vals[j] = sqrt(dd * dd + (d * d) * (cd * cd) + 1);
      }
      res->L_i[i] = simpsons_defined(vals, P, rb->dL);
      free(vals);
    }
  }
  return 0;
}

/* tendon/TendonResult.cpp:13-18: rot = AngleAxisd(theta, UnitZ).toRotationMatrix() (Rodrigues
 * form in Eigen: diagonal = (1-c)*axis^2 + c, so rot(2,2) = (1-c)+c), p <- rot p, R <- rot R. */
void orc_rotate_z(orc_result *res, double theta) {
  double s = sin(theta), c = cos(theta);
  double c1 = 1 - c;
  M3 rot = { { c1 * 0 * 0 + c, 0 - s, 0 }, { 0 + s, c1 * 0 * 0 + c, 0 }, { 0, 0, c1 * 1 * 1 + c } };
  for (int j = 0; j < res->n && j < res->cap; j++) {
    double q[3];
    m3_vec(q, rot, res->p + 3 * j);
    res->p[3 * j] = q[0]; res->p[3 * j + 1] = q[1]; res->p[3 * j + 2] = q[2];
    M3 Rm, T;
    for (int cc = 0; cc < 3; cc++) for (int rr = 0; rr < 3; rr++) Rm[rr][cc] = res->R[9 * j + cc * 3 + rr];
    m3_mul(T, rot, Rm);
    for (int cc = 0; cc < 3; cc++) for (int rr = 0; rr < 3; rr++) res->R[9 * j + cc * 3 + rr] = T[rr][cc];
  }
}

/* tendon/TendonRobot.h:105-131 */
int orc_shape(const orc_robot *rb, const double *state, orc_result *res) {
  const int N = rb->n_tendons;
  double rotate = rb->enable_rotation ? state[N] : 0.0;
  double retract = rb->enable_retraction ? state[orc_state_size(rb) - 1] : 0.0;
  int rc = orc_tension_shape(rb, state, retract, res);
  if (rc) return rc;
  if (rb->enable_rotation) orc_rotate_z(res, rotate);
  return 0;
}

/* tendon/TendonRobot.h:92-95 */
int orc_home_shape_state(const orc_robot *rb, const double *state, orc_result *res) {
  double retract = rb->enable_retraction ? state[orc_state_size(rb) - 1] : 0.0;
  return orc_home_shape(rb, retract, res);
}

/* ------------------------------------------------------------------------------------------
 * validity predicate
 * ---------------------------------------------------------------------------------------- */

/* collision/collision_primitives.cpp:10-102 */
void orc_closest_st_segment(const double A[3], const double B[3], const double C[3], const double D[3],
                            double *so, double *to) {
  const double eps = DBL_EPSILON, eps2 = eps * eps;
  double s = 0.0, t = 0.0;
  double AB[3] = { B[0] - A[0], B[1] - A[1], B[2] - A[2] };
  double CD[3] = { D[0] - C[0], D[1] - C[1], D[2] - C[2] };
  const double a = v3_dot(AB, AB), c = v3_dot(CD, CD);
#define BOUND(x) fmax(0.0, fmin(1.0, (x)))
#define CLOSEST_AB_S(P) ((a <= eps2) ? 0.0 : \
    ((AB[0] * ((P)[0] - A[0]) + AB[1] * ((P)[1] - A[1]) + AB[2] * ((P)[2] - A[2])) / a))
#define CLOSEST_CD_T(P) ((c <= eps2) ? 0.0 : \
    ((CD[0] * ((P)[0] - C[0]) + CD[1] * ((P)[1] - C[1]) + CD[2] * ((P)[2] - C[2])) / c))
  if (a <= eps2) { *so = 0.0; *to = BOUND(CLOSEST_CD_T(A)); return; }
  if (c <= eps2) { *so = BOUND(CLOSEST_AB_S(C)); *to = 0.0; return; }
  double AC[3] = { C[0] - A[0], C[1] - A[1], C[2] - A[2] };
  const double b = v3_dot(AB, CD), d = v3_dot(AC, AB), e = v3_dot(AC, CD);
  const double denom = fmax(0.0, a * c - b * b);
  if (denom <= eps2) {
    t = CLOSEST_CD_T(A);
    if (0.0 <= t && t <= 1.0) { *so = 0.0; *to = t; return; }
    t = CLOSEST_CD_T(B);
    if (0.0 <= t && t <= 1.0) { *so = 1.0; *to = t; return; }
    s = CLOSEST_AB_S(C);
    if (0.0 <= s && s <= 1.0) { *so = s; *to = 0.0; return; }
    double AD[3] = { D[0] - A[0], D[1] - A[1], D[2] - A[2] };
    double BC[3] = { C[0] - B[0], C[1] - B[1], C[2] - B[2] };
    double BD[3] = { D[0] - B[0], D[1] - B[1], D[2] - B[2] };
    double ac2 = v3_dot(AC, AC), ad2 = v3_dot(AD, AD), bc2 = v3_dot(BC, BC), bd2 = v3_dot(BD, BD);
    if (ac2 <= ad2 && ac2 <= bc2 && ac2 <= bd2) { *so = 0.0; *to = 0.0; return; }
    if (ad2 <= bc2 && ad2 <= bd2) { *so = 0.0; *to = 1.0; return; }
    if (bc2 <= bd2) { *so = 1.0; *to = 0.0; return; }
    *so = 1.0; *to = 1.0; return;
  }
  s = (c * d - b * e) / denom;
  t = (b * d - a * e) / denom;
  if (0.0 <= t && t <= 1.0) { *so = BOUND(s); *to = t; return; }
  if (t < 0.0) { *so = BOUND(-c / a); *to = 0.0; return; }
  *so = BOUND((b - c) / a); *to = 1.0;
#undef BOUND
#undef CLOSEST_AB_S
#undef CLOSEST_CD_T
}

/* collision/collision.hxx:102-108 capsule-capsule (radius sum), via :65-68 sphere-point and
 * collision_primitives.h:17-19 interpolate */
static int capsules_collide(const double *a0, const double *a1, const double *b0, const double *b1, double rsum) {
  double s, t;
  orc_closest_st_segment(a0, a1, b0, b1, &s, &t);
  double c1[3], c2[3], diff[3];
  for (int i = 0; i < 3; i++) {
    c1[i] = a0[i] + (a1[i] - a0[i]) * s;
    c2[i] = b0[i] + (b1[i] - b0[i]) * t;
    diff[i] = c1[i] - c2[i];
  }
  return v3_dot(diff, diff) <= (rsum * rsum);
}

/* collision/collision.cpp:6-46 */
int orc_collides_self(const double *p, int n, double r) {
  double dist_to_consider = 3.0 * r;
  if (n <= 2) return 0;
  double *acc = (double *)malloc(sizeof(double) * (size_t)n);
  double dist = 0.0;
  const double *prev = p;
  for (int i = 0; i < n; i++) {
    double d[3] = { p[3 * i] - prev[0], p[3 * i + 1] - prev[1], p[3 * i + 2] - prev[2] };
    dist += v3_norm(d);
    acc[i] = dist;
    prev = p + 3 * i;
  }
  int hit = 0;
  for (int a = 0; a < n - 3 && !hit; ++a) {
    for (int b = a + 2; b < n - 1; ++b) {
      if (acc[b] - acc[a + 1] < dist_to_consider) continue;
      if (capsules_collide(p + 3 * a, p + 3 * (a + 1), p + 3 * b, p + 3 * (b + 1), r + r)) { hit = 1; break; }
    }
  }
  free(acc);
  return hit;
}

/* tendon/TendonRobot.h:247-278 */
int orc_is_within_length_limits(const orc_robot *rb, const double *home_Li, const double *fk_Li) {
  for (int i = 0; i < rb->n_tendons; i++) {
    double dl = home_Li[i] - fk_Li[i];
    if (dl < rb->min_length[i] || rb->max_length[i] < dl) return 0;
  }
  return 1;
}

/* motion-planning/AbstractValidityChecker.cpp:99-114 */
int orc_is_valid_shape(const orc_robot *rb, const orc_result *fk, const orc_result *home) {
  if (!fk->converged || !home->converged) return 0;
  if (!orc_is_within_length_limits(rb, home->L_i, fk->L_i)) return 0;
  return !orc_collides_self(fk->p, fk->n, rb->r);
}

/* ------------------------------------------------------------------------------------------
 * voxels (dense stand-in for collision::VoxelOctree)
 * ---------------------------------------------------------------------------------------- */

orc_grid *orc_grid_create(int N) {
  if (N < 4 || N > 512 || (N & (N - 1))) return NULL;     /* collision/VoxelOctree.cpp:98-116 */
  orc_grid *g = (orc_grid *)calloc(1, sizeof(orc_grid));
  g->N = N; g->Nb = N / 4;
  g->blocks = (uint64_t *)calloc((size_t)g->Nb * g->Nb * g->Nb, sizeof(uint64_t));
  orc_grid_set_limits(g, 0, 1, 0, 1, 0, 1);
  return g;
}
orc_grid *orc_grid_empty_copy(const orc_grid *s) {        /* VoxelOctree.cpp:134-150 */
  orc_grid *g = orc_grid_create(s->N);
  g->xmin = s->xmin; g->xmax = s->xmax; g->ymin = s->ymin; g->ymax = s->ymax;
  g->zmin = s->zmin; g->zmax = s->zmax; g->dx = s->dx; g->dy = s->dy; g->dz = s->dz;
  return g;
}
void orc_grid_free(orc_grid *g) { if (g) { free(g->blocks); free(g); } }
void orc_grid_clear(orc_grid *g) { memset(g->blocks, 0, sizeof(uint64_t) * (size_t)g->Nb * g->Nb * g->Nb); }

/* VoxelOctree.cpp:152-177 */
int orc_grid_set_limits(orc_grid *g, double xmin, double xmax, double ymin, double ymax,
                        double zmin, double zmax) {
  if (xmin >= xmax || ymin >= ymax || zmin >= zmax) return -1;
  g->xmin = xmin; g->xmax = xmax; g->dx = (xmax - xmin) / g->N;
  g->ymin = ymin; g->ymax = ymax; g->dy = (ymax - ymin) / g->N;
  g->zmin = zmin; g->zmax = zmax; g->dz = (zmax - zmin) / g->N;
  return 0;
}

/* VoxelOctree.cpp:1501-1503 */
uint64_t orc_bitmask(int x, int y, int z) { return (uint64_t)1 << (x * 16 + y * 4 + z); }

static inline size_t block_index(const orc_grid *g, int bx, int by, int bz) {
  return ((size_t)bx * g->Nb + by) * g->Nb + bz;
}

/* VoxelOctree.cpp:256-265 (value=true): returns whether the cell was already set */
int orc_grid_set_cell(orc_grid *g, int ix, int iy, int iz) {
  uint64_t mask = orc_bitmask(ix % 4, iy % 4, iz % 4);
  uint64_t *b = &g->blocks[block_index(g, ix / 4, iy / 4, iz / 4)];
  uint64_t old = *b;
  *b = old | mask;
  return (old & mask) != 0;
}
int orc_grid_cell(const orc_grid *g, int ix, int iy, int iz) {   /* :249-252 */
  uint64_t b = g->blocks[block_index(g, ix / 4, iy / 4, iz / 4)];
  return b && (b & orc_bitmask(ix % 4, iy % 4, iz % 4));
}
int orc_grid_is_in_domain(const orc_grid *g, double x, double y, double z) {  /* :1505-1509 */
  return (g->xmin <= x && x <= g->xmax) && (g->ymin <= y && y <= g->ymax) && (g->zmin <= z && z <= g->zmax);
}
/* :295-307 */
void orc_grid_nearest_cell(const orc_grid *g, double x, double y, double z, int out[3]) {
  int ix = (int)((x - g->xmin) / g->dx);
  int iy = (int)((y - g->ymin) / g->dy);
  int iz = (int)((z - g->zmin) / g->dz);
  int m = g->N - 1;
  out[0] = ix < 0 ? 0 : (ix > m ? m : ix);
  out[1] = iy < 0 ? 0 : (iy > m ? m : iy);
  out[2] = iz < 0 ? 0 : (iz > m ? m : iz);
}
/* :309-317; returns -1 where the reference throws std::domain_error.  NOTE no clamping: a
 * coordinate exactly on the upper limit yields index N (as in the reference). */
int orc_grid_find_cell(const orc_grid *g, double x, double y, double z, int out[3]) {
  if (x < g->xmin || g->xmax < x || y < g->ymin || g->ymax < y || z < g->zmin || g->zmax < z) return -1;
  out[0] = (int)(size_t)((x - g->xmin) / g->dx);
  out[1] = (int)(size_t)((y - g->ymin) / g->dy);
  out[2] = (int)(size_t)((z - g->zmin) / g->dz);
  return 0;
}
void orc_grid_add_point(orc_grid *g, double x, double y, double z) {   /* :319-323 */
  if (!orc_grid_is_in_domain(g, x, y, z)) return;
  int c[3]; orc_grid_nearest_cell(g, x, y, z, c);
  orc_grid_set_cell(g, c[0], c[1], c[2]);
}

/* collision/collision_primitives.h:62-85 */
int orc_segment_aabox_intersect(const double A[3], const double B[3], const double C[3], const double D[3]) {
  double AB[3] = { B[0] - A[0], B[1] - A[1], B[2] - A[2] };
  double len = v3_norm(AB) / 2;
  double U[3] = { AB[0] / (2 * len), AB[1] / (2 * len), AB[2] / (2 * len) };
  double Uabs[3] = { fabs(U[0]), fabs(U[1]), fabs(U[2]) };
  double P[3], ext[3], UxP[3], Pabs[3];
  for (int i = 0; i < 3; i++) {
    P[i] = (A[i] + B[i]) / 2 - (D[i] + C[i]) / 2;
    ext[i] = fabs(D[i] - C[i]) / 2;
  }
  v3_cross(UxP, U, P);
  for (int i = 0; i < 3; i++) { UxP[i] = fabs(UxP[i]); Pabs[i] = fabs(P[i]); }
  int separated =
         Pabs[0] > ext[0] + len * Uabs[0]
      || Pabs[1] > ext[1] + len * Uabs[1]
      || Pabs[2] > ext[2] + len * Uabs[2]
      || UxP[0] > ext[1] * Uabs[2] + ext[2] * Uabs[1]
      || UxP[1] > ext[2] * Uabs[0] + ext[0] * Uabs[2]
      || UxP[2] > ext[0] * Uabs[1] + ext[1] * Uabs[0];
  return !separated;
}

/* collision/VoxelOctree.cpp:325-426, including its two quirks (metre-scaled boundary term in the
 * initial t, and the walk continuing one cell past B's cell). */
void orc_grid_add_line(orc_grid *g, const double a[3], const double b[3]) {
  const double ll[3] = { g->xmin, g->ymin, g->zmin };
  const double ur[3] = { g->xmax, g->ymax, g->zmax };
  if (!orc_segment_aabox_intersect(a, b, ll, ur)) return;
  const double npm[3] = { 1 / g->dx, 1 / g->dy, 1 / g->dz };
  double A[3], B[3];
  for (int i = 0; i < 3; i++) { A[i] = (a[i] - ll[i]) * npm[i]; B[i] = (b[i] - ll[i]) * npm[i]; }
  const int Axi = (int)A[0] - (A[0] < 0), Ayi = (int)A[1] - (A[1] < 0), Azi = (int)A[2] - (A[2] < 0);
  const int Bxi = (int)B[0] - (B[0] < 0), Byi = (int)B[1] - (B[1] < 0), Bzi = (int)B[2] - (B[2] < 0);
  const int N = g->N;
#define IDX_IN(v) (0 <= (v) && (v) < N)
#define VOX_IN(x, y, z) (IDX_IN(x) && IDX_IN(y) && IDX_IN(z))
  int entered = VOX_IN(Axi, Ayi, Azi);
  if (entered) orc_grid_set_cell(g, Axi, Ayi, Azi);
  if (VOX_IN(Bxi, Byi, Bzi)) orc_grid_set_cell(g, Bxi, Byi, Bzi);

  double BA[3] = { B[0] - A[0], B[1] - A[1], B[2] - A[2] }, U[3];
  v3_normalized(U, BA);
  const int step_x = 1 - 2 * (U[0] < 0), step_y = 1 - 2 * (U[1] < 0), step_z = 1 - 2 * (U[2] < 0);
  const double ex = fabs(A[0] - (Axi + step_x) * g->dx);
  const double ey = fabs(A[1] - (Ayi + step_y) * g->dy);
  const double ez = fabs(A[2] - (Azi + step_z) * g->dz);
  const double Uabs[3] = { fabs(U[0]), fabs(U[1]), fabs(U[2]) };
  const double threshold = 1e-10;
  const double tx_delta = (Uabs[0] > threshold) ? 1 / Uabs[0] : 1 / threshold;
  const double ty_delta = (Uabs[1] > threshold) ? 1 / Uabs[1] : 1 / threshold;
  const double tz_delta = (Uabs[2] > threshold) ? 1 / Uabs[2] : 1 / threshold;
  double tx = fabs(ex * tx_delta), ty = fabs(ey * ty_delta), tz = fabs(ez * tz_delta);
  int xi = Axi, yi = Ayi, zi = Azi;
  while (step_x * (Bxi - xi) >= 0 && step_y * (Byi - yi) >= 0 && step_z * (Bzi - zi) >= 0) {
    const int tx_is_min = (tx < ty) && (tx < tz);
    const int ty_is_min = !(tx < ty) && (ty < tz);
    if (tx_is_min) {
      xi += step_x;
      if (entered && !IDX_IN(xi)) break;
      tx += tx_delta;
    } else if (ty_is_min) {
      yi += step_y;
      if (entered && !IDX_IN(yi)) break;
      ty += ty_delta;
    } else {
      zi += step_z;
      if (entered && !IDX_IN(zi)) break;
      tz += tz_delta;
    }
    if (!entered && VOX_IN(xi, yi, zi)) entered = 1;
    if (entered) orc_grid_set_cell(g, xi, yi, zi);
  }
#undef IDX_IN
#undef VOX_IN
}

void orc_grid_add_piecewise_line(orc_grid *g, const double *pts, int n) {   /* :428-432 */
  for (int i = 1; i < n; i++) orc_grid_add_line(g, pts + 3 * (i - 1), pts + 3 * i);
}

/* :281-293 */
static void nearest_block_idx(const orc_grid *g, double x, double y, double z, int out[3]) {
  int ix = (int)((x - g->xmin) / g->dx);
  int iy = (int)((y - g->ymin) / g->dy);
  int iz = (int)((z - g->zmin) / g->dz);
  int m = g->Nb - 1;
  int bx = ix / 4, by = iy / 4, bz = iz / 4;
  out[0] = bx < 0 ? 0 : (bx > m ? m : bx);
  out[1] = by < 0 ? 0 : (by > m ? m : by);
  out[2] = bz < 0 ? 0 : (bz > m ? m : bz);
}

/* :434-469, voxel centre inside sphere (collision.hxx:65-68: |c-p|^2 <= r^2) */
void orc_grid_add_sphere(orc_grid *g, const double c[3], double r) {
  orc_grid_add_point(g, c[0], c[1], c[2]);
  int lo[3], hi[3];
  nearest_block_idx(g, c[0] - r, c[1] - r, c[2] - r, lo);
  nearest_block_idx(g, c[0] + r, c[1] + r, c[2] + r, hi);
  for (int bx = lo[0]; bx <= hi[0]; bx++)
    for (int by = lo[1]; by <= hi[1]; by++)
      for (int bz = lo[2]; bz <= hi[2]; bz++) {
        uint64_t bm = 0;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) for (int k = 0; k < 4; k++) {
          double x = g->xmin + g->dx * ((bx << 2) + i + 0.5);
          double y = g->ymin + g->dy * ((by << 2) + j + 0.5);
          double z = g->zmin + g->dz * ((bz << 2) + k + 0.5);
          double d[3] = { c[0] - x, c[1] - y, c[2] - z };
          if (v3_dot(d, d) <= r * r) bm |= orc_bitmask(i, j, k);
        }
        if (bm) g->blocks[block_index(g, bx, by, bz)] |= bm;
      }
}

/* collision/collision.hxx:83-87 collides(Capsule, Point): closest point of the segment (closest_t_segment,
 * collision_primitives.h:33-49: t = diff.(p - a) / diff.diff clamped to [0, 1], 0 when a == b to eps^2), then the sphere
 * test around it (:65-68). */
int orc_capsule_contains(const double a[3], const double b[3], double r, const double p[3]) {
  const double eps = 2.220446049250313e-16;
  double diff[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] };
  double dsq = v3_dot(diff, diff);
  double t = 0.0;
  if (!(dsq <= eps * eps)) {
    double pa[3] = { p[0] - a[0], p[1] - a[1], p[2] - a[2] };
    t = v3_dot(diff, pa) / dsq;
  }
  t = fmax(0.0, fmin(1.0, t));
  double c[3] = { a[0] + diff[0] * t, a[1] + diff[1] * t, a[2] + diff[2] * t };   /* interpolate: a + (b - a) * t */
  double d[3] = { c[0] - p[0], c[1] - p[1], c[2] - p[2] };
  return v3_dot(d, d) <= r * r;
}

/* :471-515 VoxelOctree::add_capsule: both end points' cells, then every voxel centre inside the capsule within the
 * block range of its bounding box */
void orc_grid_add_capsule(orc_grid *g, const double a[3], const double b[3], double r) {
  orc_grid_add_point(g, a[0], a[1], a[2]);
  orc_grid_add_point(g, b[0], b[1], b[2]);
  int lo[3], hi[3];
  nearest_block_idx(g, fmin(a[0], b[0]) - r, fmin(a[1], b[1]) - r, fmin(a[2], b[2]) - r, lo);
  nearest_block_idx(g, fmax(a[0], b[0]) + r, fmax(a[1], b[1]) + r, fmax(a[2], b[2]) + r, hi);
  for (int bx = lo[0]; bx <= hi[0]; bx++)
    for (int by = lo[1]; by <= hi[1]; by++)
      for (int bz = lo[2]; bz <= hi[2]; bz++) {
        uint64_t bm = 0;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) for (int k = 0; k < 4; k++) {
          double p[3] = { g->xmin + g->dx * ((bx << 2) + i + 0.5), g->ymin + g->dy * ((by << 2) + j + 0.5),
                          g->zmin + g->dz * ((bz << 2) + k + 0.5) };
          if (orc_capsule_contains(a, b, r, p)) bm |= orc_bitmask(i, j, k);
        }
        if (bm) g->blocks[block_index(g, bx, by, bz)] |= bm;
      }
}

/* collision/VoxelOctree.cpp:533-689: remove_interior_6neighbor / remove_interior_27neighbor.  Works
 * from a copy; only non-empty blocks are visited (visit_leaves); blocks beyond the grid count as full. */
static uint64_t blk_or_full(const orc_grid *g, const uint64_t *src, long bx, long by, long bz) {
  if (bx < 0 || bx >= g->Nb || by < 0 || by >= g->Nb || bz < 0 || bz >= g->Nb) return ~(uint64_t)0;
  return src[block_index(g, (int)bx, (int)by, (int)bz)];
}
void orc_grid_remove_interior(orc_grid *g, int keep_diagonal) {
  const size_t nb = (size_t)g->Nb * g->Nb * g->Nb;
  uint64_t *copy = (uint64_t *)malloc(nb * sizeof(uint64_t));
  memcpy(copy, g->blocks, nb * sizeof(uint64_t));
  for (int bx = 0; bx < g->Nb; bx++) for (int by = 0; by < g->Nb; by++) for (int bz = 0; bz < g->Nb; bz++) {
    const uint64_t old_b = copy[block_index(g, bx, by, bz)];
    if (!old_b) continue;
    uint64_t nbh[3][3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++)
      nbh[i][j][k] = blk_or_full(g, copy, (long)bx - 1 + i, (long)by - 1 + j, (long)bz - 1 + k);
    uint64_t new_b = old_b;
    for (int ix = 0; ix < 4; ix++) for (int iy = 0; iy < 4; iy++) for (int iz = 0; iz < 4; iz++) {
      int interior = 1;
      for (int i = -1; i <= 1 && interior; i++) for (int j = -1; j <= 1 && interior; j++) for (int k = -1; k <= 1; k++) {
        if (!keep_diagonal && (abs(i) + abs(j) + abs(k) > 1)) continue;     /* 6-neighbour form: the cell and its face neighbours (:552-577) */
        int x = ix + i, y = iy + j, z = iz + k, nx = 1, ny = 1, nz = 1;      /* :646-657 */
        if (x == -1) { x = 3; nx = 0; } else if (x == 4) { x = 0; nx = 2; }
        if (y == -1) { y = 3; ny = 0; } else if (y == 4) { y = 0; ny = 2; }
        if (z == -1) { z = 3; nz = 0; } else if (z == 4) { z = 0; nz = 2; }
        if (!(nbh[nx][ny][nz] & orc_bitmask(x, y, z))) { interior = 0; break; }
      }
      if (interior) new_b &= ~orc_bitmask(ix, iy, iz);
    }
    g->blocks[block_index(g, bx, by, bz)] = new_b;
  }
  free(copy);
}

/* collision/VoxelOctree.cpp:693-818: dilate_6neighbor / dilate_27neighbor through dilate_one_impl --
 * per non-empty block of a copy, a depth-limited search from every occupied cell inside the 12^3
 * neighbourhood (at most four steps), the touched 3x3x3 blocks OR-ed into the live grid; repeated
 * four dilations at a time.  The 27-neighbour move list (:776-804) names (x+1,y+1,z+1) twice and
 * (x-1,y+1,z+1) never; that is reproduced here. */
typedef struct { uint8_t depths[12][12][12]; uint64_t voxels[3][3][3]; } depth_set;
static const int MOVES6[6][3] = {{-1,0,0},{1,0,0},{0,-1,0},{0,1,0},{0,0,-1},{0,0,1}};
static const int MOVES27[27][3] = {
  {0,0,0},{-1,0,0},{1,0,0},{0,-1,0},{-1,-1,0},{1,-1,0},{0,1,0},{-1,1,0},{1,1,0},
  {0,0,-1},{-1,0,-1},{1,0,-1},{0,-1,-1},{-1,-1,-1},{1,-1,-1},{0,1,-1},{-1,1,-1},{1,1,-1},
  {0,0,1},{-1,0,1},{1,0,1},{0,-1,1},{-1,-1,1},{1,-1,1},{0,1,1},{1,1,1},{1,1,1}};
static void dilate_dfs(depth_set *v, int x, int y, int z, int d, int diag) {
  if (d == 0 || d <= v->depths[x][y][z]) return;             /* DepthSet::in / add (:707-719) */
  v->depths[x][y][z] = (uint8_t)d;
  v->voxels[x / 4][y / 4][z / 4] |= orc_bitmask(x % 4, y % 4, z % 4);
  const int (*mv)[3] = diag ? MOVES27 : MOVES6;
  const int nm = diag ? 27 : 6;
  for (int m = 0; m < nm; m++) dilate_dfs(v, x + mv[m][0], y + mv[m][1], z + mv[m][2], d - 1, diag);
}
void orc_grid_dilate(orc_grid *g, int num, int use_diagonal) {
  const size_t nb = (size_t)g->Nb * g->Nb * g->Nb;
  uint64_t *copy = (uint64_t *)malloc(nb * sizeof(uint64_t));
  depth_set *v = (depth_set *)malloc(sizeof(depth_set));
  for (; num > 0; num -= 4) {
    const int n = num < 4 ? num : 4;
    memcpy(copy, g->blocks, nb * sizeof(uint64_t));
    for (int bx = 0; bx < g->Nb; bx++) for (int by = 0; by < g->Nb; by++) for (int bz = 0; bz < g->Nb; bz++) {
      const uint64_t old_b = copy[block_index(g, bx, by, bz)];
      if (!old_b) continue;
      memset(v, 0, sizeof(*v));
      for (int x = 0; x < 4; x++) for (int y = 0; y < 4; y++) for (int z = 0; z < 4; z++)
        if (old_b & orc_bitmask(x, y, z)) dilate_dfs(v, x + 4, y + 4, z + 4, n + 1, use_diagonal);
      for (int nx = 0; nx < 3; nx++) for (int ny = 0; ny < 3; ny++) for (int nz = 0; nz < 3; nz++) {
        const int qx = bx + nx - 1, qy = by + ny - 1, qz = bz + nz - 1;
        if (qx < 0 || qx >= g->Nb || qy < 0 || qy >= g->Nb || qz < 0 || qz >= g->Nb) continue;
        g->blocks[block_index(g, qx, qy, qz)] |= v->voxels[nx][ny][nz];
      }
    }
  }
  free(copy); free(v);
}
void orc_grid_dilate_sphere(orc_grid *g, double r) {           /* :950-952 */
  orc_grid_dilate(g, (int)round(r / fmin(g->dx, fmin(g->dy, g->dz))), 0);
}

/* :973-978 -> detail/TreeNode.hxx:164-174,268: any block with a & b != 0.
 * Dimension mismatch (std::invalid_argument in the reference) returns -1. */
int orc_grid_collides(const orc_grid *a, const orc_grid *b) {
  if (a->N != b->N) return -1;
  size_t nb = (size_t)a->Nb * a->Nb * a->Nb;
  for (size_t i = 0; i < nb; i++) if (a->blocks[i] & b->blocks[i]) return 1;
  return 0;
}
int orc_grid_collides_point(const orc_grid *g, double x, double y, double z) {   /* :967-971 */
  if (!orc_grid_is_in_domain(g, x, y, z)) return 0;
  int c[3]; orc_grid_nearest_cell(g, x, y, z, c);
  return orc_grid_cell(g, c[0], c[1], c[2]);
}
size_t orc_grid_nblocks(const orc_grid *g) {
  size_t nb = (size_t)g->Nb * g->Nb * g->Nb, c = 0;
  for (size_t i = 0; i < nb; i++) c += g->blocks[i] != 0;
  return c;
}
size_t orc_grid_ncells(const orc_grid *g) {
  size_t nb = (size_t)g->Nb * g->Nb * g->Nb, c = 0;
  for (size_t i = 0; i < nb; i++) c += (size_t)__builtin_popcountll(g->blocks[i]);
  return c;
}
long orc_grid_export_blocks(const orc_grid *g, uint32_t *ids, uint64_t *masks, long cap) {
  size_t nb = (size_t)g->Nb * g->Nb * g->Nb;
  long c = 0;
  for (size_t i = 0; i < nb; i++) if (g->blocks[i]) {
    if (ids && masks && c < cap) { ids[c] = (uint32_t)i; masks[c] = g->blocks[i]; }
    c++;
  }
  return c;
}

/* collision/detail/TreeNode.hxx:176-190 visit_leaves_impl: at every level the eight children are visited with bx outermost,
 * then by, then bz (child index 4*(bx half) + 2*(by half) + (bz half), TreeNode.h:68-70), absent children skipped, down to
 * the TreeNode<4> leaves (:271-273).  On a dense grid that is: non-zero blocks in ascending order of the key that takes one
 * bit of bx, by, bz per level from the top (bx the most significant of each triple).  This is the order in which
 * VoxelOctree::to_json (collision/VoxelOctree.cpp:1357-1363) and the roadmap files' serialize_inner
 * (motion-planning/VoxelCachedLazyPRM.cpp:633-642) write a voxel set.  Pinned against the reference's own TreeNode
 * (oracle/_ref, tests/golden/treenode_*.npz). */
static void leaf_order_rec(const orc_grid *g, int bx0, int by0, int bz0, int size,
                           uint32_t *ids, uint64_t *masks, long cap, long *c) {
  if (size == 1) {
    size_t i = block_index(g, bx0, by0, bz0);
    if (g->blocks[i]) {
      if (ids && masks && *c < cap) { ids[*c] = (uint32_t)i; masks[*c] = g->blocks[i]; }
      (*c)++;
    }
    return;
  }
  int h = size / 2;
  for (int x = 0; x < 2; x++)
    for (int y = 0; y < 2; y++)
      for (int z = 0; z < 2; z++)
        leaf_order_rec(g, bx0 + x * h, by0 + y * h, bz0 + z * h, h, ids, masks, cap, c);
}
long orc_grid_export_blocks_leaf_order(const orc_grid *g, uint32_t *ids, uint64_t *masks, long cap) {
  long c = 0;
  leaf_order_rec(g, 0, 0, 0, g->Nb, ids, masks, cap, &c);
  return c;
}

/* collision/VoxelOctree.cpp:211-242 block / set_block / union_block (-> TreeNode.hxx:74-95,140-148; leaf :255-259,267):
 * union_block returns the block's value BEFORE the OR. */
uint64_t orc_grid_block(const orc_grid *g, int bx, int by, int bz) { return g->blocks[block_index(g, bx, by, bz)]; }
void orc_grid_set_block(orc_grid *g, int bx, int by, int bz, uint64_t value) { g->blocks[block_index(g, bx, by, bz)] = value; }
uint64_t orc_grid_union_block(orc_grid *g, int bx, int by, int bz, uint64_t value) {
  size_t i = block_index(g, bx, by, bz);
  uint64_t prev = g->blocks[i];
  g->blocks[i] = prev | value;
  return prev;
}

/* motion-planning/VoxelEnvironment.cpp:129-131: p <- inv_rotation * p (inv_rot row-major 3x3) */
void orc_rotate_points(const double inv_rot[9], double *pts, int n) {
  for (int j = 0; j < n; j++) {
    double *p = pts + 3 * j, q[3];
    for (int i = 0; i < 3; i++) q[i] = inv_rot[3 * i] * p[0] + inv_rot[3 * i + 1] * p[1] + inv_rot[3 * i + 2] * p[2];
    p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
  }
}

/* ------------------------------------------------------------------------------------------
 * one state-validity check: motion-planning/AbstractValidityChecker.cpp:124-133 with
 * AbstractVoxelValidityChecker.h:55-57 and VoxelBackboneValidityChecker.h:49-57
 * ---------------------------------------------------------------------------------------- */
static int is_valid_state_ws2(const orc_robot *rb, const orc_grid *obstacles, const double inv_rot[9],
                              const double *state, double tip[3], int *flags,
                              orc_result *fk, orc_result *home, orc_grid *robot_vox, int spheres);
static int is_valid_state_ws(const orc_robot *rb, const orc_grid *obstacles, const double inv_rot[9],
                             const double *state, double tip[3], int *flags,
                             orc_result *fk, orc_result *home, orc_grid *robot_vox) {
  return is_valid_state_ws2(rb, obstacles, inv_rot, state, tip, flags, fk, home, robot_vox, 0);
}
/* spheres = 0: VoxelBackboneValidityChecker::voxelize_impl (VoxelBackboneValidityChecker.h:49-57);
 * spheres = 1: VoxelValidityChecker::voxelize_impl (VoxelValidityChecker.h:18-26): a sphere of the
 * robot radius at every rotated backbone point. */
static int is_valid_state_ws2(const orc_robot *rb, const orc_grid *obstacles, const double inv_rot[9],
                              const double *state, double tip[3], int *flags,
                              orc_result *fk, orc_result *home, orc_grid *robot_vox, int spheres) {
  int fl = 0;
  if (tip) tip[0] = tip[1] = tip[2] = NAN;
  orc_shape(rb, state, fk);
  orc_home_shape_state(rb, state, home);
  if (tip && fk->n > 0) { tip[0] = fk->p[3 * (fk->n - 1)]; tip[1] = fk->p[3 * (fk->n - 1) + 1]; tip[2] = fk->p[3 * (fk->n - 1) + 2]; }
  int valid = 0;
  do {
    if (!fk->converged || !home->converged) break;
    fl |= 1;
    if (!orc_is_within_length_limits(rb, home->L_i, fk->L_i)) break;
    fl |= 2;
    if (orc_collides_self(fk->p, fk->n, rb->r)) break;
    fl |= 4;
    /* voxelize_impl: copy points, rotate, empty_copy, add_piecewise_line */
    double *rot = (double *)malloc(sizeof(double) * 3 * (size_t)fk->n);
    memcpy(rot, fk->p, sizeof(double) * 3 * (size_t)fk->n);
    orc_rotate_points(inv_rot, rot, fk->n);
    orc_grid_clear(robot_vox);
    if (spheres) { for (int j = 0; j < fk->n; j++) orc_grid_add_sphere(robot_vox, rot + 3 * j, rb->r); }
    else orc_grid_add_piecewise_line(robot_vox, rot, fk->n);
    free(rot);
    if (orc_grid_collides(obstacles, robot_vox)) break;
    fl |= 8;
    valid = 1;
  } while (0);
  if (flags) *flags = fl;
  return valid;
}

static void result_alloc(orc_result *r, int cap) {
  memset(r, 0, sizeof(*r));
  r->cap = cap;
  r->t = (double *)malloc(sizeof(double) * (size_t)cap);
  r->p = (double *)malloc(sizeof(double) * 3 * (size_t)cap);
  r->R = (double *)malloc(sizeof(double) * 9 * (size_t)cap);
}
static void result_free(orc_result *r) { free(r->t); free(r->p); free(r->R); }
static int max_points(const orc_robot *rb) { return orc_t_range(0.0, rb->L, rb->dL, NULL, 0) + 2; }

int orc_is_valid_state(const orc_robot *rb, const orc_grid *obstacles, const double inv_rot[9],
                       const double *state, double tip[3], int *flags) {
  orc_result fk, home;
  int cap = max_points(rb);
  result_alloc(&fk, cap); result_alloc(&home, cap);
  orc_grid *rv = orc_grid_empty_copy(obstacles);
  int v = is_valid_state_ws(rb, obstacles, inv_rot, state, tip, flags, &fk, &home, rv);
  orc_grid_free(rv); result_free(&fk); result_free(&home);
  return v;
}

/* AbstractValidityChecker::isValid with VoxelValidityChecker (sphere-swept robot, VoxelValidityChecker.h:18-26) */
int orc_is_valid_state_spheres(const orc_robot *rb, const orc_grid *obstacles, const double inv_rot[9],
                               const double *state, double tip[3], int *flags) {
  orc_result fk, home;
  int cap = max_points(rb);
  result_alloc(&fk, cap); result_alloc(&home, cap);
  orc_grid *rv = orc_grid_empty_copy(obstacles);
  int v = is_valid_state_ws2(rb, obstacles, inv_rot, state, tip, flags, &fk, &home, rv, 1);
  orc_grid_free(rv); result_free(&fk); result_free(&home);
  return v;
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* Sparse per-thread robot voxel set: the dense clear of a 256^3 grid (2 MiB) per configuration
 * would dominate the CPU baseline unfairly, so the batch path voxelises into a small hash-free
 * list of (block, mask) pairs and ANDs them against the obstacle grid -- the same cells, the same
 * verdict as orc_grid_collides(obstacles, fresh robot grid). */
typedef struct { uint32_t *ids; uint64_t *masks; int n, cap; const orc_grid *ref; } sparse_set;

static void sparse_set_cell(sparse_set *s, int ix, int iy, int iz) {
  uint32_t id = (uint32_t)block_index(s->ref, ix / 4, iy / 4, iz / 4);
  uint64_t m = orc_bitmask(ix % 4, iy % 4, iz % 4);
  for (int i = s->n - 1; i >= 0; i--) if (s->ids[i] == id) { s->masks[i] |= m; return; }
  if (s->n == s->cap) {
    s->cap = s->cap ? 2 * s->cap : 256;
    s->ids = (uint32_t *)realloc(s->ids, sizeof(uint32_t) * (size_t)s->cap);
    s->masks = (uint64_t *)realloc(s->masks, sizeof(uint64_t) * (size_t)s->cap);
  }
  s->ids[s->n] = id; s->masks[s->n] = m; s->n++;
}

/* add_line again, emitting into a sparse set (same statement sequence as orc_grid_add_line) */
static void sparse_add_line(sparse_set *s, const double a[3], const double b[3]) {
  const orc_grid *g = s->ref;
  const double ll[3] = { g->xmin, g->ymin, g->zmin };
  const double ur[3] = { g->xmax, g->ymax, g->zmax };
  if (!orc_segment_aabox_intersect(a, b, ll, ur)) return;
  const double npm[3] = { 1 / g->dx, 1 / g->dy, 1 / g->dz };
  double A[3], B[3];
  for (int i = 0; i < 3; i++) { A[i] = (a[i] - ll[i]) * npm[i]; B[i] = (b[i] - ll[i]) * npm[i]; }
  const int Axi = (int)A[0] - (A[0] < 0), Ayi = (int)A[1] - (A[1] < 0), Azi = (int)A[2] - (A[2] < 0);
  const int Bxi = (int)B[0] - (B[0] < 0), Byi = (int)B[1] - (B[1] < 0), Bzi = (int)B[2] - (B[2] < 0);
  const int N = g->N;
#define IDX_IN(v) (0 <= (v) && (v) < N)
#define VOX_IN(x, y, z) (IDX_IN(x) && IDX_IN(y) && IDX_IN(z))
  int entered = VOX_IN(Axi, Ayi, Azi);
  if (entered) sparse_set_cell(s, Axi, Ayi, Azi);
  if (VOX_IN(Bxi, Byi, Bzi)) sparse_set_cell(s, Bxi, Byi, Bzi);
  double BA[3] = { B[0] - A[0], B[1] - A[1], B[2] - A[2] }, U[3];
  v3_normalized(U, BA);
  const int step_x = 1 - 2 * (U[0] < 0), step_y = 1 - 2 * (U[1] < 0), step_z = 1 - 2 * (U[2] < 0);
  const double ex = fabs(A[0] - (Axi + step_x) * g->dx);
  const double ey = fabs(A[1] - (Ayi + step_y) * g->dy);
  const double ez = fabs(A[2] - (Azi + step_z) * g->dz);
  const double Uabs[3] = { fabs(U[0]), fabs(U[1]), fabs(U[2]) };
  const double threshold = 1e-10;
  const double tx_delta = (Uabs[0] > threshold) ? 1 / Uabs[0] : 1 / threshold;
  const double ty_delta = (Uabs[1] > threshold) ? 1 / Uabs[1] : 1 / threshold;
  const double tz_delta = (Uabs[2] > threshold) ? 1 / Uabs[2] : 1 / threshold;
  double tx = fabs(ex * tx_delta), ty = fabs(ey * ty_delta), tz = fabs(ez * tz_delta);
  int xi = Axi, yi = Ayi, zi = Azi;
  while (step_x * (Bxi - xi) >= 0 && step_y * (Byi - yi) >= 0 && step_z * (Bzi - zi) >= 0) {
    const int tx_is_min = (tx < ty) && (tx < tz);
    const int ty_is_min = !(tx < ty) && (ty < tz);
    if (tx_is_min) { xi += step_x; if (entered && !IDX_IN(xi)) break; tx += tx_delta; }
    else if (ty_is_min) { yi += step_y; if (entered && !IDX_IN(yi)) break; ty += ty_delta; }
    else { zi += step_z; if (entered && !IDX_IN(zi)) break; tz += tz_delta; }
    if (!entered && VOX_IN(xi, yi, zi)) entered = 1;
    if (entered) sparse_set_cell(s, xi, yi, zi);
  }
#undef IDX_IN
#undef VOX_IN
}

static int sparse_collides(const sparse_set *s, const orc_grid *obstacles) {
  for (int i = 0; i < s->n; i++) if (obstacles->blocks[s->ids[i]] & s->masks[i]) return 1;
  return 0;
}

static int is_valid_state_sparse(const orc_robot *rb, const orc_grid *obstacles, const double inv_rot[9],
                                 const double *state, double tip[3],
                                 orc_result *fk, orc_result *home, sparse_set *ss, double *rot) {
  if (tip) tip[0] = tip[1] = tip[2] = NAN;
  orc_shape(rb, state, fk);
  orc_home_shape_state(rb, state, home);
  if (tip && fk->n > 0) { tip[0] = fk->p[3 * (fk->n - 1)]; tip[1] = fk->p[3 * (fk->n - 1) + 1]; tip[2] = fk->p[3 * (fk->n - 1) + 2]; }
  if (!orc_is_valid_shape(rb, fk, home)) return 0;
  memcpy(rot, fk->p, sizeof(double) * 3 * (size_t)fk->n);
  orc_rotate_points(inv_rot, rot, fk->n);
  ss->n = 0; ss->ref = obstacles;
  for (int i = 1; i < fk->n; i++) sparse_add_line(ss, rot + 3 * (i - 1), rot + 3 * i);
  return !sparse_collides(ss, obstacles);
}

int orc_validate_batch(const orc_robot *rb, const orc_grid *obstacles, const double inv_rot[9],
                       const double *states, long n, uint8_t *valid, double *tips, int nthreads) {
  const int S = orc_state_size(rb);
  const int cap = max_points(rb);
  int used = 1;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
  used = nthreads;
#pragma omp parallel num_threads(nthreads)
#endif
  {
    orc_result fk, home;
    result_alloc(&fk, cap); result_alloc(&home, cap);
    sparse_set ss = {0};
    double *rot = (double *)malloc(sizeof(double) * 3 * (size_t)cap);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (long i = 0; i < n; i++) {
      valid[i] = (uint8_t)is_valid_state_sparse(rb, obstacles, inv_rot, states + (size_t)i * S,
                                                tips ? tips + 3 * i : NULL, &fk, &home, &ss, rot);
    }
    free(rot); free(ss.ids); free(ss.masks);
    result_free(&fk); result_free(&home);
  }
  (void)nthreads;
  return used;
}

/* apps/estimate_length_discretization.cpp:62-71: omp parallel for over robot.forward_kinematics.
 * p receives n x P x 3 (row-major); rows beyond a configuration's point count are NaN. */
int orc_fk_batch(const orc_robot *rb, const double *states, long n, double *p,
                 double *L, double *L_i, uint8_t *converged, int P, int nthreads) {
  const int S = orc_state_size(rb);
  const int cap = max_points(rb);
  int used = 1;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
  used = nthreads;
#pragma omp parallel num_threads(nthreads)
#endif
  {
    orc_result fk; result_alloc(&fk, cap);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (long i = 0; i < n; i++) {
      orc_shape(rb, states + (size_t)i * S, &fk);
      for (int j = 0; j < P; j++)
        for (int k = 0; k < 3; k++)
          p[((size_t)i * P + j) * 3 + k] = (j < fk.n) ? fk.p[3 * j + k] : NAN;
      if (L) L[i] = fk.L;
      if (L_i) for (int k = 0; k < rb->n_tendons; k++) L_i[(size_t)i * rb->n_tendons + k] = fk.L_i[k];
      if (converged) converged[i] = (uint8_t)fk.converged;
    }
    result_free(&fk);
  }
  (void)nthreads;
  return used;
}

/* ------------------------------------------------------------------------------------------
 * edges: OMPL 1.5.0 state-space arithmetic as wired by motion-planning/Problem.cpp:101-163
 * (CompoundStateSpace { RealVector tension w=1, SO2 rotation w=extent/(4 pi), RealVector
 * retraction w=2 extent/L }).  OMPL is a third-party dependency absent from /root/reference;
 * these restate its published behaviour (ompl/base/StateSpace.cpp validSegmentCount,
 * RealVectorStateSpace / SO2StateSpace interpolate & distance).
 * ---------------------------------------------------------------------------------------- */
static double tension_extent(const orc_robot *rb) {
  double s = 0;
  for (int i = 0; i < rb->n_tendons; i++) { double d = rb->max_tension[i] - 0.0; s += d * d; }
  return sqrt(s);
}

double orc_state_distance(const orc_robot *rb, const double *a, const double *b) {
  const int N = rb->n_tendons;
  double ext = tension_extent(rb);
  double s = 0;
  for (int i = 0; i < N; i++) { double d = a[i] - b[i]; s += d * d; }
  double dist = 1.0 * sqrt(s);
  int k = N;
  if (rb->enable_rotation) {
    double d = fabs(a[k] - b[k]);
    d = (d > M_PI) ? 2.0 * M_PI - d : d;
    dist += (ext / (4.0 * M_PI)) * d;
    k++;
  }
  if (rb->enable_retraction) {
    double d = a[k] - b[k];
    dist += (2.0 * ext / rb->L) * sqrt(d * d);
  }
  return dist;
}

unsigned orc_valid_segment_count(const orc_robot *rb, const orc_space_params *sp,
                                 const double *a, const double *b) {
  const int N = rb->n_tendons;
  unsigned sc = 0;
  double ext = tension_extent(rb);
  {
    double frac = sp->min_tension_change / ext;         /* Problem.cpp:118-120 */
    double lvs = ext * frac;                            /* StateSpace::setup */
    double s = 0;
    for (int i = 0; i < N; i++) { double d = a[i] - b[i]; s += d * d; }
    unsigned c = (unsigned)ceil(sqrt(s) / lvs);
    if (c > sc) sc = c;
  }
  int k = N;
  if (rb->enable_rotation) {
    double frac = sp->min_rotation_change / (2 * M_PI); /* :131-132 */
    double lvs = M_PI * frac;                           /* SO2 max extent = pi */
    double d = fabs(a[k] - b[k]);
    d = (d > M_PI) ? 2.0 * M_PI - d : d;
    unsigned c = (unsigned)ceil(d / lvs);
    if (c > sc) sc = c;
    k++;
  }
  if (rb->enable_retraction) {
    double frac = fmin(0.01, sp->min_retraction_change / rb->L);   /* :144-145 */
    double lvs = rb->L * frac;                          /* extent of [0, L] */
    double d = a[k] - b[k];
    unsigned c = (unsigned)ceil(sqrt(d * d) / lvs);
    if (c > sc) sc = c;
  }
  return sc;
}

void orc_interpolate_state(const orc_robot *rb, const double *a, const double *b, double t, double *out) {
  const int N = rb->n_tendons;
  for (int i = 0; i < N; i++) out[i] = a[i] + (b[i] - a[i]) * t;
  int k = N;
  if (rb->enable_rotation) {
    double diff = b[k] - a[k];
    if (fabs(diff) <= M_PI) {
      out[k] = a[k] + diff * t;
    } else {
      if (diff > 0.0) diff = 2.0 * M_PI - diff; else diff = -2.0 * M_PI - diff;
      double v = a[k] - diff * t;
      if (v > M_PI) v -= 2.0 * M_PI; else if (v < -M_PI) v += 2.0 * M_PI;
      out[k] = v;
    }
    k++;
  }
  if (rb->enable_retraction) out[k] = a[k] + (b[k] - a[k]) * t;
}

typedef struct { double t; double *pts; int n; int is_valid; } fk_sample;

/* motion-planning/VoxelEnvironment.cpp:207-444 driven as VoxelBackboneMotionValidator.cpp:19-81
 * drives it (per-sample validity = is_valid_shape only; obstacles tested on the union) and
 * AbstractVoxelMotionValidator.h:143-151 (checkMotion). */
static int check_motion_impl(const orc_robot *rb, const orc_space_params *sp, const orc_grid *obstacles,
                             const double inv_rot[9], const double *a, const double *b, int until_invalid, int vc_spheres,
                             orc_grid *swept, int *n_fk_out, int *is_fully_valid_out, double *last_valid_t_out) {
  const int S = orc_state_size(rb);
  orc_grid *sample_vox = until_invalid ? orc_grid_empty_copy(obstacles) : NULL;
  const int cap = max_points(rb);
  unsigned nseg = orc_valid_segment_count(rb, sp, a, b);
  double rel_threshold = 1.0 / (double)nseg;

  orc_result fk, home; result_alloc(&fk, cap); result_alloc(&home, cap);
  int nfk = 0, fcap = 64;
  fk_sample *fks = (fk_sample *)malloc(sizeof(fk_sample) * (size_t)fcap);
  double first_invalid_t = 10.0;
  double *cur = (double *)malloc(sizeof(double) * (size_t)S);

#define ADD_FK(tt, cfg) do { \
    if (nfk == fcap) { fcap *= 2; fks = (fk_sample *)realloc(fks, sizeof(fk_sample) * (size_t)fcap); } \
    orc_shape(rb, (cfg), &fk); \
    orc_home_shape_state(rb, (cfg), &home); \
    int ok_ = orc_is_valid_shape(rb, &fk, &home); \
    if (!ok_ && (tt) < first_invalid_t) first_invalid_t = (tt); \
    fks[nfk].t = (tt); fks[nfk].n = fk.n; fks[nfk].is_valid = ok_; \
    fks[nfk].pts = (double *)malloc(sizeof(double) * 3 * (size_t)fk.n); \
    memcpy(fks[nfk].pts, fk.p, sizeof(double) * 3 * (size_t)fk.n); \
    orc_rotate_points(inv_rot, fks[nfk].pts, fk.n); \
    if (ok_ && until_invalid) { /* voxelize_until_invalid_impl: validity also needs !_vc->collides(shape), i.e. the \
                                   installed state checker's own voxelize_impl (backbone, or a sphere per point) */ \
      orc_grid_clear(sample_vox); \
      if (vc_spheres) { for (int j_ = 0; j_ < fk.n; j_++) orc_grid_add_sphere(sample_vox, fks[nfk].pts + 3 * j_, rb->r); } \
      else orc_grid_add_piecewise_line(sample_vox, fks[nfk].pts, fk.n); \
      if (orc_grid_collides(obstacles, sample_vox)) { \
        ok_ = 0; fks[nfk].is_valid = 0; \
        if ((tt) < first_invalid_t) first_invalid_t = (tt); \
      } \
    } \
    nfk++; } while (0)

  ADD_FK(0.0, a);
  ADD_FK(1.0, b);

  int domain_error = 0;
  /* should_subdivide :304-341 */
#define SHOULD_SUBDIVIDE(res, ia, ib) do { \
    const fk_sample *sa_ = &fks[ia], *sb_ = &fks[ib]; \
    (res) = 0; \
    if (!sa_->is_valid) break; \
    if (sa_->n + 1 < sb_->n || sa_->n > sb_->n + 1) { (res) = 1; break; } \
    int P_ = sa_->n < sb_->n ? sa_->n : sb_->n; \
    for (int i_ = P_ - 1; i_ >= 0; i_--) { \
      int ca_[3], cb_[3]; \
      if (orc_grid_find_cell(obstacles, sa_->pts[3*i_], sa_->pts[3*i_+1], sa_->pts[3*i_+2], ca_) || \
          orc_grid_find_cell(obstacles, sb_->pts[3*i_], sb_->pts[3*i_+1], sb_->pts[3*i_+2], cb_)) { domain_error = 1; break; } \
      long dx_ = labs((long)ca_[0] - (long)cb_[0]), dy_ = labs((long)ca_[1] - (long)cb_[1]), dz_ = labs((long)ca_[2] - (long)cb_[2]); \
      if (dx_ > 1 || dy_ > 1 || dz_ > 1) { (res) = 1; break; } \
    } } while (0)

  int scap = 64, sp_n = 0;
  int (*stack)[2] = (int (*)[2])malloc(sizeof(int[2]) * (size_t)scap);
#define PUSH(i0, i1) do { if (sp_n == scap) { scap *= 2; stack = (int (*)[2])realloc(stack, sizeof(int[2]) * (size_t)scap); } \
    stack[sp_n][0] = (i0); stack[sp_n][1] = (i1); sp_n++; } while (0)

  int sub;
  SHOULD_SUBDIVIDE(sub, 0, 1);
  if (sub) PUSH(0, 1);
  while (sp_n > 0 && !domain_error) {
    sp_n--;
    int ia = stack[sp_n][0], ib = stack[sp_n][1];
    double t_a = fks[ia].t, t_b = fks[ib].t;
    if ((t_b - t_a) <= rel_threshold) continue;
    if (first_invalid_t <= t_a) continue;
    double mid = (t_a + t_b) / 2;
    orc_interpolate_state(rb, a, b, mid, cur);
    int im = nfk;
    ADD_FK(mid, cur);
    SHOULD_SUBDIVIDE(sub, im, ib);
    if (sub) PUSH(im, ib);
    SHOULD_SUBDIVIDE(sub, ia, im);
    if (sub) PUSH(ia, im);
  }

  orc_grid *vox = swept ? swept : orc_grid_empty_copy(obstacles);
  if (swept) orc_grid_clear(swept);
  double last_valid_t = 0.0;
  for (int i = nfk; i-- > 0;) {
    if (fks[i].t < first_invalid_t) {
      orc_grid_add_piecewise_line(vox, fks[i].pts, fks[i].n);
      if (last_valid_t < fks[i].t) last_valid_t = fks[i].t;
    }
  }
  int fully = (5.0 < first_invalid_t);
  int valid = fully && !orc_grid_collides(obstacles, vox);
  if (domain_error) valid = -1;    /* reference would throw std::domain_error (find_cell) */
  if (n_fk_out) *n_fk_out = nfk;
  if (is_fully_valid_out) *is_fully_valid_out = fully;
  if (last_valid_t_out) *last_valid_t_out = last_valid_t;

  if (!swept) orc_grid_free(vox);
  if (sample_vox) orc_grid_free(sample_vox);
  for (int i = 0; i < nfk; i++) free(fks[i].pts);
  free(fks); free(stack); free(cur);
  result_free(&fk); result_free(&home);
  return valid;
#undef ADD_FK
#undef SHOULD_SUBDIVIDE
#undef PUSH
}

int orc_check_motion(const orc_robot *rb, const orc_space_params *sp, const orc_grid *obstacles,
                     const double inv_rot[9], const double *a, const double *b,
                     orc_grid *swept, int *n_fk_out, int *is_fully_valid_out, double *last_valid_t_out) {
  return check_motion_impl(rb, sp, obstacles, inv_rot, a, b, 0, 0, swept, n_fk_out, is_fully_valid_out, last_valid_t_out);
}

/* checkMotion(s1, s2, last_valid) (AbstractVoxelMotionValidator.h:153-169) over voxelize_until_invalid
 * (VoxelBackboneMotionValidator.cpp:83-91): per-sample validity = is_valid_shape && !collides(shape).
 * Returns is_fully_valid; *last_valid_t = partial.t. */
int orc_check_motion_until_invalid(const orc_robot *rb, const orc_space_params *sp, const orc_grid *obstacles,
                                   const double inv_rot[9], const double *a, const double *b,
                                   int *n_fk_out, double *last_valid_t_out) {
  int fully = 0;
  check_motion_impl(rb, sp, obstacles, inv_rot, a, b, 1, 0, NULL, n_fk_out, &fully, last_valid_t_out);
  return fully;
}

/* The same with motion_planning::VoxelValidityChecker installed as the state checker (Problem.h:175-180 next to
 * :203-210): `_vc->collides(shape)` voxelises the shape as a sphere of the robot radius at every backbone point
 * (VoxelValidityChecker.h:18-26). */
int orc_check_motion_until_invalid_vc(const orc_robot *rb, const orc_space_params *sp, const orc_grid *obstacles,
                                      const double inv_rot[9], const double *a, const double *b, int vc_spheres,
                                      int *n_fk_out, double *last_valid_t_out) {
  int fully = 0;
  check_motion_impl(rb, sp, obstacles, inv_rot, a, b, 1, vc_spheres, NULL, n_fk_out, &fully, last_valid_t_out);
  return fully;
}

/* motion-planning/VoxelBackboneDiscreteMotionValidator.cpp:9-79 (generic_voxelize) driven as
 * AbstractVoxelMotionValidator drives it: until_invalid = 0 is voxelize_impl (per-sample validity =
 * is_valid_shape, obstacles tested on the union, checkMotion(s1, s2), .h:143-151); until_invalid = 1 is
 * voxelize_until_invalid_impl (per-sample validity also needs !collides(sample), .h:153-169).
 * Returns the checkMotion verdict; *last_valid_t = PartialVoxelization::t. */
int orc_check_motion_discrete_vc(const orc_robot *rb, const orc_space_params *sp, const orc_grid *obstacles,
                                 const double inv_rot[9], const double *a, const double *b, int until_invalid, int vc_spheres,
                                 int *n_fk_out, int *is_fully_valid_out, double *last_valid_t_out);
int orc_check_motion_discrete(const orc_robot *rb, const orc_space_params *sp, const orc_grid *obstacles,
                              const double inv_rot[9], const double *a, const double *b, int until_invalid,
                              int *n_fk_out, int *is_fully_valid_out, double *last_valid_t_out) {
  return orc_check_motion_discrete_vc(rb, sp, obstacles, inv_rot, a, b, until_invalid, 0, n_fk_out, is_fully_valid_out, last_valid_t_out);
}
/* vc_spheres: the installed state checker is VoxelValidityChecker (matters for until_invalid only, see above) */
int orc_check_motion_discrete_vc(const orc_robot *rb, const orc_space_params *sp, const orc_grid *obstacles,
                                 const double inv_rot[9], const double *a, const double *b, int until_invalid, int vc_spheres,
                                 int *n_fk_out, int *is_fully_valid_out, double *last_valid_t_out) {
  const int S = orc_state_size(rb);
  const int cap = max_points(rb);
  orc_result fk, home; result_alloc(&fk, cap); result_alloc(&home, cap);
  orc_grid *vox = orc_grid_empty_copy(obstacles), *one = orc_grid_empty_copy(obstacles);
  double *test = (double *)malloc(sizeof(double) * (size_t)S), *pts = (double *)malloc(sizeof(double) * 3 * (size_t)cap);
  int nfk = 0, fully;
  double vt = 0.0;

  /* one sample: FK, validity, and (if valid) its backbone joins the union */
#define SAMPLE(cfg, ok_out) do { \
    orc_shape(rb, (cfg), &fk); orc_home_shape_state(rb, (cfg), &home); nfk++; \
    memcpy(pts, fk.p, sizeof(double) * 3 * (size_t)fk.n); \
    orc_rotate_points(inv_rot, pts, fk.n); \
    (ok_out) = orc_is_valid_shape(rb, &fk, &home); \
    if ((ok_out) && until_invalid) { \
      orc_grid_clear(one); \
      if (vc_spheres) { for (int j_ = 0; j_ < fk.n; j_++) orc_grid_add_sphere(one, pts + 3 * j_, rb->r); } \
      else orc_grid_add_piecewise_line(one, pts, fk.n); \
      if (orc_grid_collides(obstacles, one)) (ok_out) = 0; \
    } } while (0)

  SAMPLE(a, fully);                                             /* :24-28: a is voxelised whatever its validity */
  orc_grid_add_piecewise_line(vox, pts, fk.n);
  unsigned nd = orc_valid_segment_count(rb, sp, a, b);          /* :40 */
  if (nd > 1) {
    for (unsigned i = 1; fully && i < nd; ++i) {                /* :45-58 */
      double t = (double)i / (double)nd;
      orc_interpolate_state(rb, a, b, t, test);
      SAMPLE(test, fully);
      if (fully) { vt = t; orc_grid_add_piecewise_line(vox, pts, fk.n); }
    }
  }
  if (fully) {                                                   /* :62-72 */
    SAMPLE(b, fully);
    if (fully) { vt = 1; orc_grid_add_piecewise_line(vox, pts, fk.n); }
  }
#undef SAMPLE
  int valid = fully && !orc_grid_collides(obstacles, vox);
  if (n_fk_out) *n_fk_out = nfk;
  if (is_fully_valid_out) *is_fully_valid_out = fully;
  if (last_valid_t_out) *last_valid_t_out = vt;
  orc_grid_free(vox); orc_grid_free(one); free(test); free(pts);
  result_free(&fk); result_free(&home);
  return valid;
}

int orc_check_motion_batch(const orc_robot *rb, const orc_space_params *sp, const orc_grid *obstacles,
                           const double inv_rot[9], const double *a, const double *b, long n,
                           uint8_t *valid, int32_t *n_fk, int nthreads) {
  const int S = orc_state_size(rb);
  int used = 1;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
  used = nthreads;
#pragma omp parallel num_threads(nthreads)
#endif
  {
    orc_grid *swept = orc_grid_empty_copy(obstacles);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (long i = 0; i < n; i++) {
      int nf = 0;
      int v = orc_check_motion(rb, sp, obstacles, inv_rot, a + (size_t)i * S, b + (size_t)i * S,
                               swept, &nf, NULL, NULL);
      valid[i] = (uint8_t)(v == 1);
      if (n_fk) n_fk[i] = nf;
    }
    orc_grid_free(swept);
  }
  (void)nthreads;
  return used;
}

/* motion-planning/VoxelCachedLazyPRM.cpp:2397-2411,2497-2509: cached voxel set vs obstacles */
void orc_check_cached(const orc_grid *obstacles, const uint32_t *block_ids, const uint64_t *masks,
                      const int64_t *offsets, long n_items, uint8_t *hit) {
  for (long i = 0; i < n_items; i++) {
    uint8_t h = 0;
    for (int64_t k = offsets[i]; k < offsets[i + 1]; k++)
      if (obstacles->blocks[block_ids[k]] & masks[k]) { h = 1; break; }
    hit[i] = h;
  }
}

/* ------------------------------------------------------------------------------------------
 * The interactive query loop on a cached roadmap: VoxelCachedLazyPRM::solveWithRoadmap's inner loop
 * (motion-planning/VoxelCachedLazyPRM.cpp:2066-2071) over constructSolution (:2689-2771), astarSearch
 * (:2950-2976, Boost astar_search with costHeuristic = state-space distance, :2773-2775),
 * computeVertexValidity / computeEdgeValidity on cached voxel sets (:2607-2631), removeVertices / removeEdge,
 * clearValidity (:1656-1663).  One query at a time, sequentially, exactly as the reference proceeds: all invalid
 * interior vertices of the candidate path are removed, else the FIRST invalid edge from the goal side.
 * Boost.Graph is third-party (not under /root/reference): A* is restated from its published algorithm (best-first
 * on g + h, consistent heuristic, stop when the goal is examined).
 * ---------------------------------------------------------------------------------------- */
struct orc_roadmap {
  const orc_robot *rb;
  long V, E;
  int S;
  double *states; int32_t *eu, *ev; double *w;
  int64_t *adj_off; int32_t *adj_v, *adj_e;
  const int64_t *v_off, *e_off; const uint32_t *v_ids, *e_ids; const uint64_t *v_masks, *e_masks;
  const uint8_t *v_present, *e_present;
  uint8_t *vstat, *estat;            /* 0 unknown, 1 VALIDITY_TRUE, 2 removed from the graph */
  /* A* scratch */
  double *g; int32_t *parent, *parent_e; uint32_t *stamp; uint8_t *closed; uint32_t gen;
  double *hkey; int32_t *hval; long hn, hcap;
};

orc_roadmap *orc_roadmap_create(const orc_robot *rb, const double *states, long V, const int32_t *edges, const double *weights, long E,
                                const int64_t *v_off, const uint32_t *v_ids, const uint64_t *v_masks, const uint8_t *v_present,
                                const int64_t *e_off, const uint32_t *e_ids, const uint64_t *e_masks, const uint8_t *e_present) {
  orc_roadmap *r = (orc_roadmap *)calloc(1, sizeof(orc_roadmap));
  r->rb = rb; r->V = V; r->E = E; r->S = orc_state_size(rb);
  r->states = (double *)malloc(sizeof(double) * (size_t)(V * r->S + 1));
  memcpy(r->states, states, sizeof(double) * (size_t)(V * r->S));
  r->eu = (int32_t *)malloc(sizeof(int32_t) * (size_t)(E + 1)); r->ev = (int32_t *)malloc(sizeof(int32_t) * (size_t)(E + 1));
  r->w = (double *)malloc(sizeof(double) * (size_t)(E + 1));
  r->adj_off = (int64_t *)calloc((size_t)V + 2, sizeof(int64_t));
  for (long e = 0; e < E; e++) {
    r->eu[e] = edges[2 * e]; r->ev[e] = edges[2 * e + 1];
    r->w[e] = weights ? weights[e] : orc_state_distance(rb, states + (size_t)r->eu[e] * r->S, states + (size_t)r->ev[e] * r->S);
    r->adj_off[r->eu[e] + 1]++; r->adj_off[r->ev[e] + 1]++;
  }
  for (long v = 0; v < V; v++) r->adj_off[v + 1] += r->adj_off[v];
  r->adj_v = (int32_t *)malloc(sizeof(int32_t) * (size_t)(2 * E + 1)); r->adj_e = (int32_t *)malloc(sizeof(int32_t) * (size_t)(2 * E + 1));
  int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)(V + 1));
  memcpy(fill, r->adj_off, sizeof(int64_t) * (size_t)(V + 1));
  for (long e = 0; e < E; e++) {
    r->adj_v[fill[r->eu[e]]] = r->ev[e]; r->adj_e[fill[r->eu[e]]++] = (int32_t)e;
    r->adj_v[fill[r->ev[e]]] = r->eu[e]; r->adj_e[fill[r->ev[e]]++] = (int32_t)e;
  }
  free(fill);
  r->v_off = v_off; r->v_ids = v_ids; r->v_masks = v_masks; r->v_present = v_present;
  r->e_off = e_off; r->e_ids = e_ids; r->e_masks = e_masks; r->e_present = e_present;
  r->vstat = (uint8_t *)calloc((size_t)V + 1, 1); r->estat = (uint8_t *)calloc((size_t)E + 1, 1);
  r->g = (double *)malloc(sizeof(double) * (size_t)(V + 1));
  r->parent = (int32_t *)malloc(sizeof(int32_t) * (size_t)(V + 1)); r->parent_e = (int32_t *)malloc(sizeof(int32_t) * (size_t)(V + 1));
  r->stamp = (uint32_t *)calloc((size_t)V + 1, sizeof(uint32_t)); r->closed = (uint8_t *)calloc((size_t)V + 1, 1);
  r->hcap = 1024; r->hkey = (double *)malloc(sizeof(double) * (size_t)r->hcap); r->hval = (int32_t *)malloc(sizeof(int32_t) * (size_t)r->hcap);
  return r;
}

void orc_roadmap_free(orc_roadmap *r) {
  if (!r) return;
  free(r->states); free(r->eu); free(r->ev); free(r->w); free(r->adj_off); free(r->adj_v); free(r->adj_e);
  free(r->vstat); free(r->estat); free(r->g); free(r->parent); free(r->parent_e); free(r->stamp); free(r->closed);
  free(r->hkey); free(r->hval); free(r);
}

void orc_roadmap_clear_validity(orc_roadmap *r) {          /* :1656-1663 (and the graph as loaded) */
  memset(r->vstat, 0, (size_t)r->V); memset(r->estat, 0, (size_t)r->E);
}
void orc_roadmap_get_validity(const orc_roadmap *r, uint8_t *vstat, uint8_t *estat) {
  if (vstat) memcpy(vstat, r->vstat, (size_t)r->V);
  if (estat) memcpy(estat, r->estat, (size_t)r->E);
}

static void heap_push(orc_roadmap *r, double key, int32_t val) {
  if (r->hn == r->hcap) {
    r->hcap *= 2;
    r->hkey = (double *)realloc(r->hkey, sizeof(double) * (size_t)r->hcap); r->hval = (int32_t *)realloc(r->hval, sizeof(int32_t) * (size_t)r->hcap);
  }
  long i = r->hn++;
  while (i > 0) {
    long p = (i - 1) / 2;
    if (!(key < r->hkey[p])) break;
    r->hkey[i] = r->hkey[p]; r->hval[i] = r->hval[p]; i = p;
  }
  r->hkey[i] = key; r->hval[i] = val;
}
static int32_t heap_pop(orc_roadmap *r) {
  int32_t top = r->hval[0];
  double key = r->hkey[--r->hn]; int32_t val = r->hval[r->hn];
  long i = 0;
  for (;;) {
    long c = 2 * i + 1;
    if (c >= r->hn) break;
    if (c + 1 < r->hn && r->hkey[c + 1] < r->hkey[c]) c++;
    if (!(r->hkey[c] < key)) break;
    r->hkey[i] = r->hkey[c]; r->hval[i] = r->hval[c]; i = c;
  }
  r->hkey[i] = key; r->hval[i] = val;
  return top;
}

static int roadmap_astar(orc_roadmap *r, int start, int goal) {
  if (++r->gen == 0) { memset(r->stamp, 0, sizeof(uint32_t) * (size_t)r->V); r->gen = 1; }
  const uint32_t gen = r->gen;
  const double *sg = r->states + (size_t)goal * r->S;
  r->hn = 0;
  r->stamp[start] = gen; r->g[start] = 0.0; r->parent[start] = start; r->parent_e[start] = -1; r->closed[start] = 0;
  heap_push(r, orc_state_distance(r->rb, r->states + (size_t)start * r->S, sg), start);
  while (r->hn > 0) {
    int32_t u = heap_pop(r);
    if (r->closed[u]) continue;
    r->closed[u] = 1;
    if (u == goal) return 1;                                  /* AStarGoalVisitor::examine_vertex */
    for (int64_t k = r->adj_off[u]; k < r->adj_off[u + 1]; k++) {
      int32_t e = r->adj_e[k], v = r->adj_v[k];
      if (r->estat[e] == 2 || r->vstat[v] == 2) continue;      /* removed from the graph */
      double gv = r->g[u] + r->w[e];
      if (r->stamp[v] != gen) { r->stamp[v] = gen; r->closed[v] = 0; }
      else if (r->closed[v] || !(gv < r->g[v])) continue;
      r->g[v] = gv; r->parent[v] = u; r->parent_e[v] = e;
      heap_push(r, gv + orc_state_distance(r->rb, r->states + (size_t)v * r->S, sg), v);
    }
  }
  return 0;
}

static int cached_collides(const orc_grid *obstacles, const int64_t *off, const uint32_t *ids, const uint64_t *masks, long i) {
  for (int64_t k = off[i]; k < off[i + 1]; k++) if (obstacles->blocks[ids[k]] & masks[k]) return 1;
  return 0;
}
static int vertex_validity(orc_roadmap *r, const orc_grid *obstacles, int v, long *checked) {      /* computeVertexValidity */
  if (r->vstat[v] == 0) {
    (*checked)++;
    int ok = (!r->v_present || r->v_present[v]) && !cached_collides(obstacles, r->v_off, r->v_ids, r->v_masks, v);
    r->vstat[v] = ok ? 1 : 2;
  }
  return r->vstat[v] == 1;
}
static int edge_validity(orc_roadmap *r, const orc_grid *obstacles, int e, long *checked) {        /* computeEdgeValidity */
  if (r->estat[e] == 0) {
    (*checked)++;
    int ok = (!r->e_present || r->e_present[e]) && !cached_collides(obstacles, r->e_off, r->e_ids, r->e_masks, e);
    r->estat[e] = ok ? 1 : 2;
  }
  return r->estat[e] == 1;
}

/* Returns the number of path vertices written to path_out (start ... goal), 0 when start and goal are not connected,
 * -2 / -3 when the start / goal vertex is itself invalid in this environment. */
int orc_roadmap_query(orc_roadmap *r, const orc_grid *obstacles, int start, int goal, int32_t *path_out, int cap,
                      double *cost_out, int *iterations_out, long *checked_out) {
  long checked = 0;
  int iters = 0, n = 0;
  double cost = 0.0;
  if (!vertex_validity(r, obstacles, start, &checked)) { n = -2; goto done; }
  if (!vertex_validity(r, obstacles, goal, &checked)) { n = -3; goto done; }
  if (start == goal) { if (cap > 0) path_out[0] = start; n = 1; goto done; }               /* :2696-2701 */
  for (;;) {
    iters++;
    if (!roadmap_astar(r, start, goal)) { n = 0; break; }
    int removed = 0;
    for (int pos = r->parent[goal]; r->parent[pos] != pos; pos = r->parent[pos])             /* :2711-2714 */
      if (!vertex_validity(r, obstacles, pos, &checked)) removed++;                          /* marking 2 = removeVertices(:2726) */
    if (removed) continue;
    int bad_edge = 0;
    for (int v = goal; v != start; v = r->parent[v])                                         /* :2745-2762, from the goal side */
      if (!edge_validity(r, obstacles, r->parent_e[v], &checked)) { bad_edge = 1; break; }   /* first invalid edge only */
    if (bad_edge) continue;
    int len = 0;
    for (int v = goal;; v = r->parent[v]) { len++; if (v == start) break; }
    if (len <= cap) {
      int i = len - 1;
      for (int v = goal;; v = r->parent[v]) { path_out[i--] = v; if (v == start) break; }
    }
    {  /* solution->cost(opt_) (:2080): ompl::geometric::PathGeometric::cost accumulates from the start state on */
      int32_t *pe = (int32_t *)malloc(sizeof(int32_t) * (size_t)len);
      int k = 0;
      for (int v = goal; v != start; v = r->parent[v]) pe[k++] = r->parent_e[v];
      while (k-- > 0) cost += r->w[pe[k]];
      free(pe);
    }
    n = len;
    break;
  }
done:
  if (cost_out) *cost_out = cost;
  if (iterations_out) *iterations_out = iters;
  if (checked_out) *checked_out = checked;
  return n;
}
