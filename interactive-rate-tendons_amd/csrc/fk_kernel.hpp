// fk_kernel.hpp -- K1 `fk_rk4_batch`: one wavefront lane integrates one configuration's
// Cosserat-rod ODE with classical RK4 along arc length.
//
// What it computes is TendonRobot::tension_shape (tendon/TendonRobot.cpp:325-500) for a batch:
// solve_initial_bending (tendon/solve_initial_bending.cpp:14-73), then RK4 over the arc-length
// grid with tendon_deriv (tendon/tendon_deriv.cpp:95-178) as right-hand side, then the base
// residual test (TendonRobot.cpp:470-474), then rotate_z (tendon/TendonResult.cpp:13-18).
//
// How it computes it is specific to this engine:
//  * the state lives in VGPRs for the whole integration (no per-step memory traffic); only the
//    observed backbone points are stored, structure-of-arrays so a wavefront writes 512 contiguous
//    bytes per coordinate;
//  * with retraction disabled every configuration of the batch shares the arc-length grid, so the
//    tendon routing r(s), r'(s), r''(s) (polynomial + sin/cos, get_r_info.cpp:105-144) is
//    tabulated once on the host for the 3 distinct stage abscissae of every step and read through
//    wave-uniform scalar loads (SGPR operands) -- no per-lane transcendental in the hot loop;
//  * the right-hand side exploits structure the reference leaves to Eigen: r has no z component,
//    A_i is symmetric, G = B^T, H is symmetric, K_se/K_bt are diagonal, and the 6x6 solve
//    [v';u'] = M^-1 [d;c] with M = [[K_se+A, B^T],[B, K_bt+H]] is an unrolled symmetric L D L^T
//    factorisation (round 4; before: block elimination with adjugate 3x3 inverses);
//  * (p, L, L_i) are pure quadratures (nothing depends on them), so they keep no stage copy.
//  * __launch_bounds__(64, 2): two waves per SIMD (<= 256 registers) measured 16 % faster than the
//    365-register single-wave allocation the compiler picks when unconstrained.
// fp64 throughout.  Contraction to FMA is enabled here (parity with the oracle is by tolerance:
// tip position <= 1e-9 m, see tests/); the bit-exact integer/predicate stage lives in
// sweep_kernel.hpp and is compiled without contraction.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include "tr_types.hpp"

namespace trk {

// 1/x and 1/sqrt(x) from the hardware seed (v_rcp_f64 / v_rsq_f64: ~2^-23 relative) + Newton steps.  One step
// leaves ~2e-14 relative error in a handful of reciprocals per RK4 stage: backbone points move from 1.4e-16 m
// to 1.5e-15 m off the oracle -- the parity tolerance is 1e-9 m -- and the kernel gets 2.6 % faster (16
// reciprocal square roots and 8 reciprocals per step).  -DTRK_NEWTON_STEPS=2 restores full double precision.
#ifndef TRK_NEWTON_STEPS
#define TRK_NEWTON_STEPS 1
#endif
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(e, y, y);
#if TRK_NEWTON_STEPS > 1
  e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(e, y, y);
#endif
  return y;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  // y <- y * (1.5 - 0.5 x y^2)
  double hx = 0.5 * x;
  double t = __builtin_fma(-hx * y, y, 0.5);
  y = __builtin_fma(y, t, y);
#if TRK_NEWTON_STEPS > 1
  t = __builtin_fma(-hx * y, y, 0.5);
  y = __builtin_fma(y, t, y);
#endif
  return y;
}

// Right-hand side for the strain part of the state: (v', u') and the length rates.
// route(j, r6) yields tendon j's routing {rx, ry, rdx, rdy, rddx, rddy} when the tendon loop reaches it
// (a table row for the shared-grid kernel, an on-the-fly evaluation for the retraction kernel).
template <int N, class Route>
__device__ __forceinline__ void strain_rates_routed(const double v[3], const double u[3], const double (&tau)[N],
                                                    Route &&route, const RobotK &K,
                                                    double dv[3], double du[3], double (&sdot)[N]) {
#pragma clang fp contract(fast)
  // Sums over tendons, using A_i = c (pd pd^T - |pd|^2 I), c = -tau/|pd|^3, q = c pd, e = r x pd, g = c e:
  //   A = sum q pd^T - (sum c|pd|^2) I
  //   B = sum rhat A_i   = sum g pd^T - sum c|pd|^2 rhat
  //   H = -sum B_i rhat  = sum g e^T  + sum c|pd|^2 rhat^2
  //   a = sum q (pd.w) - c|pd|^2 w,   b = sum r x a_i,   w = u x (pd + r') + r''
  double Axx = 0, Axy = 0, Axz = 0, Ayy = 0, Ayz = 0, Azz = 0, Z = 0;
  double B00 = 0, B01 = 0, B02 = 0, B10 = 0, B11 = 0, B12 = 0, B20 = 0, B21 = 0, B22 = 0, Q1 = 0, Q2 = 0;
  double Hxx = 0, Hxy = 0, Hxz = 0, Hyy = 0, Hyz = 0, Hzz = 0, P1 = 0, P2 = 0, P3 = 0;
  double ax = 0, ay = 0, az = 0, bx = 0, by = 0, bz = 0;
#pragma unroll
  for (int j = 0; j < N; j++) {
    double r6[6];
    route(j, r6);
    const double rx = r6[0], ry = r6[1];
    const double rdx = r6[2], rdy = r6[3];
    const double rddx = r6[4], rddy = r6[5];
    // pd = u x r + r' + v
    const double pdx = (v[0] + rdx) - u[2] * ry;
    const double pdy = (v[1] + rdy) + u[2] * rx;
    const double pdz = __builtin_fma(u[0], ry, __builtin_fma(-u[1], rx, v[2]));
    const double s2 = pdx * pdx + pdy * pdy + pdz * pdz;
    const double rs = fast_rsqrt(s2);
    sdot[j] = s2 * rs;
    // c = -tau / |pd|^3 and c |pd|^2 = -tau / |pd| from the same reciprocal root (one multiply fewer than c * s2)
    const double cs2 = -tau[j] * rs;
    const double c = cs2 * (rs * rs);
    const double qx = c * pdx, qy = c * pdy, qz = c * pdz;
    Axx += qx * pdx; Axy += qx * pdy; Axz += qx * pdz; Ayy += qy * pdy; Ayz += qy * pdz; Azz += qz * pdz;
    Z += cs2;
    // e = r x pd, g = c e  (r_z = 0)
    const double ex = ry * pdz, ey = -rx * pdz, ez = rx * pdy - ry * pdx;
    const double gx = c * ex, gy = c * ey, gz = c * ez;
    B00 += gx * pdx; B01 += gx * pdy; B02 += gx * pdz;
    B10 += gy * pdx; B11 += gy * pdy; B12 += gy * pdz;
    B20 += gz * pdx; B21 += gz * pdy; B22 += gz * pdz;
    Hxx += gx * ex; Hxy += gx * ey; Hxz += gx * ez; Hyy += gy * ey; Hyz += gy * ez; Hzz += gz * ez;
    const double t1 = cs2 * rx, t2 = cs2 * ry;
    Q1 += t1; Q2 += t2;
    P1 += t1 * rx; P2 += t1 * ry; P3 += t2 * ry;
    // w = u x (pd + r') + r''
    const double hx = pdx + rdx, hy = pdy + rdy, hz = pdz;
    const double wx = __builtin_fma(u[1], hz, __builtin_fma(-u[2], hy, rddx));
    const double wy = __builtin_fma(u[2], hx, __builtin_fma(-u[0], hz, rddy));
    const double wz = u[0] * hy - u[1] * hx;
    const double pw = pdx * wx + pdy * wy + pdz * wz;
    const double aix = qx * pw - cs2 * wx, aiy = qy * pw - cs2 * wy, aiz = qz * pw - cs2 * wz;
    ax += aix; ay += aiy; az += aiz;
    bx += ry * aiz; by -= rx * aiz; bz += rx * aiy - ry * aix;
  }
  // The tail below is written as explicit FMA chains: without re-association the compiler keeps the a - (b*c + d*e)
  // shapes as written and spends an extra add / negate on each (~90 instructions per RK4 step).
#define TRK_FMA __builtin_fma
  // - sum c|pd|^2 rhat, rhat = [[0,0,ry],[0,0,-rx],[-ry,rx,0]]
  B02 -= Q2; B12 += Q1; B20 += Q2; B21 -= Q1;
  // + sum c|pd|^2 rhat^2, rhat^2 = [[-ry^2, rx ry, 0],[rx ry, -rx^2, 0],[0,0,-(rx^2+ry^2)]]
  Hxy += P2;
  // c = -u x (K_bt u) - v x (K_se (v - e3)) - b ;  d = -u x (K_se (v - e3)) - a
  const double kux = K.kb0 * u[0], kuy = K.kb0 * u[1], kuz = K.kb2 * u[2];
  const double svx = K.ks0 * v[0], svy = K.ks0 * v[1], svz = K.ks2 * (v[2] - 1.0);
  const double cx = TRK_FMA(u[2], kuy, TRK_FMA(-u[1], kuz, TRK_FMA(v[2], svy, TRK_FMA(-v[1], svz, -bx))));
  const double cy = TRK_FMA(u[0], kuz, TRK_FMA(-u[2], kux, TRK_FMA(v[0], svz, TRK_FMA(-v[2], svx, -by))));
  const double cz = TRK_FMA(u[1], kux, TRK_FMA(-u[0], kuy, TRK_FMA(v[1], svx, TRK_FMA(-v[0], svy, -bz))));
  const double dx = TRK_FMA(u[2], svy, TRK_FMA(-u[1], svz, -ax));
  const double dy = TRK_FMA(u[0], svz, TRK_FMA(-u[2], svx, -ay));
  const double dz = TRK_FMA(u[1], svx, TRK_FMA(-u[0], svy, -az));
  // M11 = K_se + A - Z I (symmetric): the leading block of the 6 x 6 system solved below
  const double ksz0 = K.ks0 - Z, ksz2 = K.ks2 - Z;
  const double m00 = ksz0 + Axx, m01 = Axy, m02 = Axz, m11 = ksz0 + Ayy, m12 = Ayz, m22 = ksz2 + Azz;
#ifndef TRK_SOLVE_SCHUR
  // The symmetric 6 x 6 system [[M11, B^T], [B, K_bt + H]] [v'; u'] = [d; c] by an unrolled L D L^T factorisation without pivoting:
  // 35 FMAs, 15 products and six reciprocals to factor, 36 operations for the two triangular solves (181 flops, ~105 instructions).
  // Round 4's A/B against the two-adjugate Schur form below (272 flops with its set-up, ~132 instructions, two reciprocals;
  // -DTRK_SOLVE_SCHUR): the bare RK4 loop 7.84 -> 7.56 ms per 2^20 (N = 3), 4.75 -> 4.61 ms per 2^19 (N = 4), fk_verdict
  // 8.27 -> 8.00 / 5.02 -> 4.93 ms (profiles/r04/kbench_ldlt_v1.txt); the six reciprocals' dependent chain is covered by the SIMD's
  // other wave.  Backbone points stay within 1e-14 m of the Schur form's (tests: 1e-9 m of the oracle).
  {
    // (no contraction beyond the FMAs written out: every kernel that holds this body then forms the same bits)
#pragma clang fp contract(off)
    const double n00 = (K.kb0 - P3) + Hxx, n01 = Hxy, n02 = Hxz, n11 = (K.kb0 - P1) + Hyy, n12 = Hyz, n22 = ((K.kb2 - P1) - P3) + Hzz;
    const double p0 = m00;
    const double r0 = fast_rcp(p0);
    const double w10 = m01, l10 = w10 * r0;
    const double w20 = m02, l20 = w20 * r0;
    const double w30 = B00, l30 = w30 * r0;
    const double w40 = B10, l40 = w40 * r0;
    const double w50 = B20, l50 = w50 * r0;
    const double p1 = TRK_FMA(-l10, w10, m11);
    const double r1 = fast_rcp(p1);
    const double w21 = TRK_FMA(-l20, w10, m12), l21 = w21 * r1;
    const double w31 = TRK_FMA(-l30, w10, B01), l31 = w31 * r1;
    const double w41 = TRK_FMA(-l40, w10, B11), l41 = w41 * r1;
    const double w51 = TRK_FMA(-l50, w10, B21), l51 = w51 * r1;
    const double p2 = TRK_FMA(-l21, w21, TRK_FMA(-l20, w20, m22));
    const double r2 = fast_rcp(p2);
    const double w32 = TRK_FMA(-l31, w21, TRK_FMA(-l30, w20, B02)), l32 = w32 * r2;
    const double w42 = TRK_FMA(-l41, w21, TRK_FMA(-l40, w20, B12)), l42 = w42 * r2;
    const double w52 = TRK_FMA(-l51, w21, TRK_FMA(-l50, w20, B22)), l52 = w52 * r2;
    const double p3 = TRK_FMA(-l32, w32, TRK_FMA(-l31, w31, TRK_FMA(-l30, w30, n00)));
    const double r3 = fast_rcp(p3);
    const double w43 = TRK_FMA(-l42, w32, TRK_FMA(-l41, w31, TRK_FMA(-l40, w30, n01))), l43 = w43 * r3;
    const double w53 = TRK_FMA(-l52, w32, TRK_FMA(-l51, w31, TRK_FMA(-l50, w30, n02))), l53 = w53 * r3;
    const double p4 = TRK_FMA(-l43, w43, TRK_FMA(-l42, w42, TRK_FMA(-l41, w41, TRK_FMA(-l40, w40, n11))));
    const double r4 = fast_rcp(p4);
    const double w54 = TRK_FMA(-l53, w43, TRK_FMA(-l52, w42, TRK_FMA(-l51, w41, TRK_FMA(-l50, w40, n12)))), l54 = w54 * r4;
    const double p5 = TRK_FMA(-l54, w54, TRK_FMA(-l53, w53, TRK_FMA(-l52, w52, TRK_FMA(-l51, w51, TRK_FMA(-l50, w50, n22)))));
    const double r5 = fast_rcp(p5);
    const double y0 = dx;
    const double y1 = TRK_FMA(-l10, y0, dy);
    const double y2 = TRK_FMA(-l21, y1, TRK_FMA(-l20, y0, dz));
    const double y3 = TRK_FMA(-l32, y2, TRK_FMA(-l31, y1, TRK_FMA(-l30, y0, cx)));
    const double y4 = TRK_FMA(-l43, y3, TRK_FMA(-l42, y2, TRK_FMA(-l41, y1, TRK_FMA(-l40, y0, cy))));
    const double y5 = TRK_FMA(-l54, y4, TRK_FMA(-l53, y3, TRK_FMA(-l52, y2, TRK_FMA(-l51, y1, TRK_FMA(-l50, y0, cz)))));
    const double x5 = y5 * r5;
    const double x4 = TRK_FMA(-l54, x5, y4 * r4);
    const double x3 = TRK_FMA(-l53, x5, TRK_FMA(-l43, x4, y3 * r3));
    const double x2 = TRK_FMA(-l52, x5, TRK_FMA(-l42, x4, TRK_FMA(-l32, x3, y2 * r2)));
    const double x1 = TRK_FMA(-l51, x5, TRK_FMA(-l41, x4, TRK_FMA(-l31, x3, TRK_FMA(-l21, x2, y1 * r1))));
    const double x0 = TRK_FMA(-l50, x5, TRK_FMA(-l40, x4, TRK_FMA(-l30, x3, TRK_FMA(-l20, x2, TRK_FMA(-l10, x1, y0 * r0)))));
    dv[0] = x0; dv[1] = x1; dv[2] = x2; du[0] = x3; du[1] = x4; du[2] = x5;
  }
#else
  const double c00 = TRK_FMA(m11, m22, -(m12 * m12)), c01 = TRK_FMA(m02, m12, -(m01 * m22)), c02 = TRK_FMA(m01, m12, -(m02 * m11));
  const double c11 = TRK_FMA(m00, m22, -(m02 * m02)), c12 = TRK_FMA(m01, m02, -(m00 * m12)), c22 = TRK_FMA(m00, m11, -(m01 * m01));
  const double idet = fast_rcp(TRK_FMA(m00, c00, TRK_FMA(m01, c01, m02 * c02)));
  const double i00 = c00 * idet, i01 = c01 * idet, i02 = c02 * idet, i11 = c11 * idet, i12 = c12 * idet, i22 = c22 * idet;
  // y = M11^-1 d
  const double yx = TRK_FMA(i00, dx, TRK_FMA(i01, dy, i02 * dz));
  const double yy_ = TRK_FMA(i01, dx, TRK_FMA(i11, dy, i12 * dz));
  const double yz = TRK_FMA(i02, dx, TRK_FMA(i12, dy, i22 * dz));
  // T = B M11^-1
  const double T00 = TRK_FMA(B00, i00, TRK_FMA(B01, i01, B02 * i02)), T01 = TRK_FMA(B00, i01, TRK_FMA(B01, i11, B02 * i12)),
               T02 = TRK_FMA(B00, i02, TRK_FMA(B01, i12, B02 * i22));
  const double T10 = TRK_FMA(B10, i00, TRK_FMA(B11, i01, B12 * i02)), T11 = TRK_FMA(B10, i01, TRK_FMA(B11, i11, B12 * i12)),
               T12 = TRK_FMA(B10, i02, TRK_FMA(B11, i12, B12 * i22));
  const double T20 = TRK_FMA(B20, i00, TRK_FMA(B21, i01, B22 * i02)), T21 = TRK_FMA(B20, i01, TRK_FMA(B21, i11, B22 * i12)),
               T22 = TRK_FMA(B20, i02, TRK_FMA(B21, i12, B22 * i22));
  // Schur complement S = (K_bt + H) - T B^T (symmetric), with H's rhat^2 terms folded into the first operand
  const double s00 = TRK_FMA(-T02, B02, TRK_FMA(-T01, B01, TRK_FMA(-T00, B00, (K.kb0 - P3) + Hxx)));
  const double s01 = TRK_FMA(-T02, B12, TRK_FMA(-T01, B11, TRK_FMA(-T00, B10, Hxy)));
  const double s02 = TRK_FMA(-T02, B22, TRK_FMA(-T01, B21, TRK_FMA(-T00, B20, Hxz)));
  const double s11 = TRK_FMA(-T12, B12, TRK_FMA(-T11, B11, TRK_FMA(-T10, B10, (K.kb0 - P1) + Hyy)));
  const double s12 = TRK_FMA(-T12, B22, TRK_FMA(-T11, B21, TRK_FMA(-T10, B20, Hyz)));
  const double s22 = TRK_FMA(-T22, B22, TRK_FMA(-T21, B21, TRK_FMA(-T20, B20, ((K.kb2 - P1) - P3) + Hzz)));
  // rhs = c - B y
  const double ex = TRK_FMA(-B02, yz, TRK_FMA(-B01, yy_, TRK_FMA(-B00, yx, cx)));
  const double ey = TRK_FMA(-B12, yz, TRK_FMA(-B11, yy_, TRK_FMA(-B10, yx, cy)));
  const double ez = TRK_FMA(-B22, yz, TRK_FMA(-B21, yy_, TRK_FMA(-B20, yx, cz)));
  // u' = S^-1 rhs (adjugate)
  const double g00 = TRK_FMA(s11, s22, -(s12 * s12)), g01 = TRK_FMA(s02, s12, -(s01 * s22)), g02 = TRK_FMA(s01, s12, -(s02 * s11));
  const double g11 = TRK_FMA(s00, s22, -(s02 * s02)), g12 = TRK_FMA(s01, s02, -(s00 * s12)), g22 = TRK_FMA(s00, s11, -(s01 * s01));
  const double isd = fast_rcp(TRK_FMA(s00, g00, TRK_FMA(s01, g01, s02 * g02)));
  du[0] = TRK_FMA(g00, ex, TRK_FMA(g01, ey, g02 * ez)) * isd;
  du[1] = TRK_FMA(g01, ex, TRK_FMA(g11, ey, g12 * ez)) * isd;
  du[2] = TRK_FMA(g02, ex, TRK_FMA(g12, ey, g22 * ez)) * isd;
  // v' = y - T^T u'
  dv[0] = TRK_FMA(-T20, du[2], TRK_FMA(-T10, du[1], TRK_FMA(-T00, du[0], yx)));
  dv[1] = TRK_FMA(-T21, du[2], TRK_FMA(-T11, du[1], TRK_FMA(-T01, du[0], yy_)));
  dv[2] = TRK_FMA(-T22, du[2], TRK_FMA(-T12, du[1], TRK_FMA(-T02, du[0], yz)));
#endif
#undef TRK_FMA
}

// ri: wave-uniform pointer to N x 6 doubles {rx, ry, rdx, rdy, rddx, rddy} per tendon.
template <int N>
__device__ __forceinline__ void strain_rates(const double v[3], const double u[3], const double (&tau)[N],
                                             const double *__restrict__ ri, const RobotK &K,
                                             double dv[3], double du[3], double (&sdot)[N]) {
  strain_rates_routed<N>(v, u, tau, [&](int j, double (&r6)[6]) {
#pragma unroll
    for (int q = 0; q < 6; q++) r6[q] = ri[6 * j + q];
  }, K, dv, du, sdot);
}

// solve_initial_bending + base residual.  Written without contraction and with IEEE div/sqrt so
// the data-dependent iteration count follows the same decisions as a plain fp64 evaluation.
// rb: wave-uniform pointer to N x 6 routing values at s_start (only the first 4 of each are used).
template <int N>
__device__ __forceinline__ void initial_bending(const double (&tau)[N], const double *__restrict__ rb,
                                                const RobotK &K, double v[3], double u[3], bool &converged) {
#pragma clang fp contract(off)
  v[0] = 0; v[1] = 0; v[2] = 1;
  u[0] = 0; u[1] = 0; u[2] = 0;
  bool done = false;
  for (int it = 0; it < 1000; ++it) {
    if (!__any(!done)) break;
    double Fx = 0, Fy = 0, Fz = 0, Lx = 0, Ly = 0, Lz = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
      const double rx = rb[6 * k + 0], ry = rb[6 * k + 1], rdx = rb[6 * k + 2], rdy = rb[6 * k + 3];
      double px = (-u[2] * ry) + rdx + v[0];
      double py = (u[2] * rx) + rdy + v[1];
      double pz = (u[0] * ry - u[1] * rx) + v[2];
      const double z = px * px + py * py + pz * pz;
      if (z > 0.0) { const double s = sqrt(z); px = px / s; py = py / s; pz = pz / s; }
      Fx -= tau[k] * px; Fy -= tau[k] * py; Fz -= tau[k] * pz;
      // (tau * rhat) * unit, rhat = [[0,0,ry],[0,0,-rx],[-ry,rx,0]]
      const double trx = tau[k] * rx, try_ = tau[k] * ry;
      Lx -= try_ * pz; Ly -= (-trx) * pz; Lz -= ((-try_) * px + trx * py);
    }
    const double nx = K.ks0 * v[0], ny = K.ks0 * v[1], nz = K.ks2 * (v[2] - 1);
    const double mx = K.kb0 * u[0], my = K.kb0 * u[1], mz = K.kb2 * u[2];
    const double e1 = (nx - Fx) * (nx - Fx) + (ny - Fy) * (ny - Fy) + (nz - Fz) * (nz - Fz);
    const double e2 = (mx - Lx) * (mx - Lx) + (my - Ly) * (my - Ly) + (mz - Lz) * (mz - Lz);
    const double residual = sqrt(e1 + e2);
    if (!done && residual < K.residual_threshold) done = true;
    const double vnx = K.iks0 * Fx, vny = K.iks0 * Fy, vnz = K.iks2 * Fz + 1;
    const double unx = K.ikb0 * Lx, uny = K.ikb0 * Ly, unz = K.ikb2 * Lz;
    if (!done) {
      const double dvn = sqrt((vnx - v[0]) * (vnx - v[0]) + (vny - v[1]) * (vny - v[1]) + (vnz - v[2]) * (vnz - v[2]));
      const double dun = sqrt((unx - u[0]) * (unx - u[0]) + (uny - u[1]) * (uny - u[1]) + (unz - u[2]) * (unz - u[2]));
      const double vn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      const double un = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
      if (dvn < 1e-9 * vn && dun < 1e-9 * un) done = true;
    }
    if (!done) { v[0] = vnx; v[1] = vny; v[2] = vnz; u[0] = unx; u[1] = uny; u[2] = unz; }
  }
  // base residual: PointForces::calc_point_forces with R = I (TendonRobot.cpp:188-217,470-474)
  {
    double Fx = 0, Fy = 0, Fz = 0, Lx = 0, Ly = 0, Lz = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
      const double rx = rb[6 * k + 0], ry = rb[6 * k + 1], rdx = rb[6 * k + 2], rdy = rb[6 * k + 3];
      double px = (u[1] * 0.0 - u[2] * ry) + rdx + v[0];
      double py = (u[2] * rx - u[0] * 0.0) + rdy + v[1];
      double pz = (u[0] * ry - u[1] * rx) + 0.0 + v[2];
      const double z = px * px + py * py + pz * pz;
      if (z > 0.0) { const double s = sqrt(z); px = px / s; py = py / s; pz = pz / s; }
      const double fx = -tau[k] * px, fy = -tau[k] * py, fz = -tau[k] * pz;
      Fx += fx; Fy += fy; Fz += fz;
      Lx += ry * fz - 0.0 * fy; Ly += 0.0 * fx - rx * fz; Lz += rx * fy - ry * fx;
    }
    const double nx = K.ks0 * v[0], ny = K.ks0 * v[1], nz = K.ks2 * (v[2] - 1);
    const double mx = K.kb0 * u[0], my = K.kb0 * u[1], mz = K.kb2 * u[2];
    const double e1 = (nx - Fx) * (nx - Fx) + (ny - Fy) * (ny - Fy) + (nz - Fz) * (nz - Fz);
    const double e2 = (mx - Lx) * (mx - Lx) + (my - Ly) * (my - Ly) + (mz - Lz) * (mz - Lz);
    converged = sqrt(e1 + e2) <= K.residual_threshold;
  }
}

// One classical RK4 step of size h (Boost.odeint runge_kutta4 tableau: a = {1/2},{0,1/2},{0,0,1};
// b = {1/6,1/3,1/3,1/6}; c = {0,1/2,1/2,1}) of the state (R, v, u | p, L, L_i), with the routing
// produced on demand, tendon by tendon, by `route(t, j, r6)` at t, t + h/2 (stages 2 and 3) and t + h.
// Used by the retraction kernel; the shared-grid kernel below carries the same statements inline
// (as a function taking three table rows it made hipcc hoist all scalar loads and spill ~100 VGPRs).
template <int N, class Route>
__device__ __forceinline__ void rk4_step_routed(double (&R)[9], double (&v)[3], double (&u)[3], double (&p)[3], double &Lb,
                                                double (&Li)[N], const double (&tau)[N], const RobotK &K, double t, double h,
                                                Route &&route) {
#pragma clang fp contract(fast)
  const double hh = h * 0.5;
  const double b1 = h * (1.0 / 6.0), b2 = h * (1.0 / 3.0);
  double aR[9], av[3], au[3];
#pragma unroll
  for (int q = 0; q < 9; q++) aR[q] = R[q];
#pragma unroll
  for (int q = 0; q < 3; q++) { av[q] = v[q]; au[q] = u[q]; }
  double sR[9], sv[3], su[3];
#pragma unroll
  for (int q = 0; q < 9; q++) sR[q] = R[q];
#pragma unroll
  for (int q = 0; q < 3; q++) { sv[q] = v[q]; su[q] = u[q]; }
#pragma unroll
  for (int st = 0; st < 4; st++) {
    // The routing depends on the abscissa only, so the scheduler would evaluate all four stages' routing
    // up front and keep it live (measured: +170 registers).  Tying the abscissa to the stage state keeps
    // each evaluation next to its use.
    double ts = (st == 0) ? t : ((st == 3) ? t + h : t + hh);
    asm volatile("" : "+v"(ts) : "v"(sv[0]), "v"(su[0]));
    const double bw = (st == 0 || st == 3) ? b1 : b2;
    const double aw = (st == 2) ? h : hh;
    double dv[3], du[3], sd[N];
    strain_rates_routed<N>(sv, su, tau, [&](int j, double (&r6)[6]) { route(ts, j, r6); }, K, dv, du, sd);
    p[0] += bw * (sR[0] * sv[0] + sR[3] * sv[1] + sR[6] * sv[2]);
    p[1] += bw * (sR[1] * sv[0] + sR[4] * sv[1] + sR[7] * sv[2]);
    p[2] += bw * (sR[2] * sv[0] + sR[5] * sv[1] + sR[8] * sv[2]);
    {
      const double v2 = sv[0] * sv[0] + sv[1] * sv[1] + sv[2] * sv[2];
      Lb += bw * (v2 * fast_rsqrt(v2));
    }
#pragma unroll
    for (int j = 0; j < N; j++) Li[j] += bw * sd[j];
    double dR[9];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const double r0 = sR[0 + r], r1 = sR[3 + r], r2 = sR[6 + r];
      dR[0 + r] = r1 * su[2] - r2 * su[1];
      dR[3 + r] = r2 * su[0] - r0 * su[2];
      dR[6 + r] = r0 * su[1] - r1 * su[0];
    }
#pragma unroll
    for (int q = 0; q < 9; q++) aR[q] += bw * dR[q];
#pragma unroll
    for (int q = 0; q < 3; q++) { av[q] += bw * dv[q]; au[q] += bw * du[q]; }
    if (st < 3) {
#pragma unroll
      for (int q = 0; q < 9; q++) sR[q] = R[q] + aw * dR[q];
#pragma unroll
      for (int q = 0; q < 3; q++) { sv[q] = v[q] + aw * dv[q]; su[q] = u[q] + aw * du[q]; }
    }
  }
#pragma unroll
  for (int q = 0; q < 9; q++) R[q] = aR[q];
#pragma unroll
  for (int q = 0; q < 3; q++) { v[q] = av[q]; u[q] = au[q]; }
}


// K1 for the shared arc-length grid (retraction disabled).
//   tab:   [(nsteps*3 + 1)][N][6] routing table; entry 0 = base (s_start), then 3 per step
//   steps: [nsteps]
// What a lane knows about its configuration when the integration ends (the verdict-only kernel continues from here).
template <int N>
struct FkLane {
  bool converged;
  double Li[N];
  double tip[3];              // the last backbone point (after rotate_z)
};

struct NoPointHook {
  __device__ __forceinline__ void begin(bool) const {}
  __device__ __forceinline__ void operator()(int, double, double, double) const {}
  __device__ __forceinline__ void tip_point(int, bool, bool, double, double, double, bool = true) const {}   // retraction kernel
};

// A hook may ask for the tendon-length quadratures in LDS from some tendon count on (`static constexpr int kLiInLdsFrom = 3`;
// verdict_kernel.hpp: PointSweep, sweep_kernel.hpp: SignatureHook); granted up to four tendons, the widest robot that runs two
// waves per SIMD with a sweep in (or behind) its loop.
template <class H, class = void> struct HookLiInLds { static constexpr int from = 1 << 30; };
template <class H> struct HookLiInLds<H, std::void_t<decltype(H::kLiInLdsFrom)>> { static constexpr int from = H::kLiInLdsFrom; };
template <class OnPoint> __host__ __device__ constexpr bool li_in_lds(int N) {
  return N >= HookLiInLds<std::remove_cv_t<std::remove_reference_t<OnPoint>>>::from && N <= 4;
}

// on_point(j, x, y, z): called for every observed backbone point (after rotate_z), in order j = 0 .. P-1 -- the
// verdict-only kernel sweeps the point there instead of storing it; on_point.begin(converged && live) precedes point 0.  row_map (optional): lane i integrates
// configuration row_map[i] of `states` (the fallback pass of the verdict path works on a compacted list).
template <int N, bool ROT, bool WRITE_R, bool WANT_L = true, class OnPoint = NoPointHook>
__device__ __forceinline__ void fk_uniform_body(
    const double *__restrict__ states, int64_t n, int64_t ld, const RobotK &K,
    const double *__restrict__ tab, const StepK *__restrict__ steps, int nsteps, const FkOut &out,
    OnPoint &&on_point = NoPointHook(), const int32_t *__restrict__ row_map = nullptr, FkLane<N> *lane_out = nullptr) {
#pragma clang fp contract(fast)
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const bool live = i < n;
  const int64_t il = live ? i : (n - 1);       // tail lanes recompute the last configuration, stores masked
  const int64_t ic = row_map ? (int64_t)row_map[il] : il;
  const int S = K.state_size;
  double tau[N];
#pragma unroll
  for (int j = 0; j < N; j++) tau[j] = states[ic * S + j];
  // Four tendons at two waves per SIMD with a per-point sweep in the loop: the N tendon-length quadratures -- written four times
  // a step, read once after the loop -- live in LDS (N x 512 B per wave) instead of 2 N registers.  Measured on fk_verdict<4>:
  // VGPR spills 25 -> 2 dwords (76 -> 12 B of scratch per lane, which every wave wrote back to HBM once: 2.4x -> 1.2x the
  // algorithmic traffic), 5.26 -> 5.19 ms per 2^19 (profiles/r03/kbench_v8_li_in_lds.txt; with 2.5 KB instead of 2 the wave count per CU drops to seven: -11 %, kbench_v7).  The tensions themselves in LDS changed nothing.
  constexpr bool kLiLds = li_in_lds<OnPoint>(N);
  double *li_lds = nullptr;
  if constexpr (kLiLds) {
    __shared__ double li_lds_store[N * 64];
    li_lds = li_lds_store;
#pragma unroll
    for (int j = 0; j < N; j++) li_lds[j * 64 + threadIdx.x] = 0.0;
  }
  double rc = 1.0, rs = 0.0, r22 = 1.0;
  if (ROT) {
    const double th = states[ic * S + N];
    rs = sin(th); rc = cos(th);
    r22 = (1.0 - rc) + rc;                     // Eigen AngleAxis::toRotationMatrix diagonal term
  }

  double v[3], u[3];
  bool conv;
  initial_bending<N>(tau, tab, K, v, u, conv);
  on_point.begin(conv && live);

  // state
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};   // column-major: R[c*3+r]
  double p[3] = {0, 0, 0};
  double Lb = 0;
  double Li[N];
#pragma unroll
  for (int j = 0; j < N; j++) Li[j] = 0;

  auto store_point = [&](int j) {
    const int64_t o = (int64_t)j * ld + i;
    double x = p[0], y = p[1], z = p[2];
    // rotate_z (tendon/TendonResult.cpp:13-18); explicit FMAs so that every kernel holding this body forms the same bits
    if (ROT) { const double x2 = __builtin_fma(rc, x, -(rs * y)), y2 = __builtin_fma(rs, x, rc * y); x = x2; y = y2; z = r22 * z; }
    on_point(j, x, y, z);
    if (!live) return;
    if (out.px) { out.px[o] = x; out.py[o] = y; out.pz[o] = z; }
    if (WRITE_R) {
      const int64_t PS = (int64_t)K.n_points * ld;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        double a = R[c * 3 + 0], b = R[c * 3 + 1], cc = R[c * 3 + 2];
        if (ROT) { const double a2 = rc * a - rs * b, b2 = rs * a + rc * b; a = a2; b = b2; cc = r22 * cc; }
        out.R[(c * 3 + 0) * PS + o] = a; out.R[(c * 3 + 1) * PS + o] = b; out.R[(c * 3 + 2) * PS + o] = cc;
      }
    }
  };
  store_point(0);

  for (int k = 0; k < nsteps; k++) {
    const double h = steps[k].h;
    const int obs = steps[k].obs;
    const double *__restrict__ rt = tab + (size_t)(1 + 3 * k) * (N * 6);
    const double hh = h * 0.5;
    const double b1 = h * (1.0 / 6.0), b2 = h * (1.0 / 3.0);
    // accumulators start at the current state
    double aR[9], av[3], au[3];
#pragma unroll
    for (int q = 0; q < 9; q++) aR[q] = R[q];
#pragma unroll
    for (int q = 0; q < 3; q++) { av[q] = v[q]; au[q] = u[q]; }
    double sR[9], sv[3], su[3];                // stage state
#pragma unroll
    for (int q = 0; q < 9; q++) sR[q] = R[q];
#pragma unroll
    for (int q = 0; q < 3; q++) { sv[q] = v[q]; su[q] = u[q]; }

#pragma unroll
    for (int st = 0; st < 4; st++) {
      const double *__restrict__ ri = rt + (st == 0 ? 0 : (st == 3 ? 2 : 1)) * (N * 6);
      const double bw = (st == 0 || st == 3) ? b1 : b2;      // weight of this stage in the update
      const double aw = (st == 2) ? h : hh;                  // coefficient towards the next stage
      double dv[3], du[3], sd[N];
      strain_rates<N>(sv, su, tau, ri, K, dv, du, sd);
      // quadratures: p' = R v, L' = |v|, L_i' = |pd_i|
      p[0] += bw * (sR[0] * sv[0] + sR[3] * sv[1] + sR[6] * sv[2]);
      p[1] += bw * (sR[1] * sv[0] + sR[4] * sv[1] + sR[7] * sv[2]);
      p[2] += bw * (sR[2] * sv[0] + sR[5] * sv[1] + sR[8] * sv[2]);
      if (WANT_L && out.L) {                               // wave-uniform
        const double v2 = sv[0] * sv[0] + sv[1] * sv[1] + sv[2] * sv[2];
        Lb += bw * (v2 * fast_rsqrt(v2));
      }
      if constexpr (kLiLds) {
#pragma unroll
        for (int j = 0; j < N; j++) li_lds[j * 64 + threadIdx.x] += bw * sd[j];
      } else {
#pragma unroll
        for (int j = 0; j < N; j++) Li[j] += bw * sd[j];
      }
      // R' = R uhat : col0 = R1*uz - R2*uy ; col1 = R2*ux - R0*uz ; col2 = R0*uy - R1*ux
      double dR[9];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const double r0 = sR[0 + r], r1 = sR[3 + r], r2 = sR[6 + r];
        dR[0 + r] = r1 * su[2] - r2 * su[1];
        dR[3 + r] = r2 * su[0] - r0 * su[2];
        dR[6 + r] = r0 * su[1] - r1 * su[0];
      }
#pragma unroll
      for (int q = 0; q < 9; q++) aR[q] += bw * dR[q];
#pragma unroll
      for (int q = 0; q < 3; q++) { av[q] += bw * dv[q]; au[q] += bw * du[q]; }
      if (st < 3) {
#pragma unroll
        for (int q = 0; q < 9; q++) sR[q] = R[q] + aw * dR[q];
#pragma unroll
        for (int q = 0; q < 3; q++) { sv[q] = v[q] + aw * dv[q]; su[q] = u[q] + aw * du[q]; }
      }
    }
#pragma unroll
    for (int q = 0; q < 9; q++) R[q] = aR[q];
#pragma unroll
    for (int q = 0; q < 3; q++) { v[q] = av[q]; u[q] = au[q]; }
    if (obs >= 0) store_point(obs);
  }

  if constexpr (kLiLds) {
#pragma unroll
    for (int j = 0; j < N; j++) Li[j] = li_lds[j * 64 + threadIdx.x];
  }
  if (live) {
    if (out.L) out.L[i] = Lb;
    if (out.Li) {
#pragma unroll
      for (int j = 0; j < N; j++) out.Li[(int64_t)j * ld + i] = Li[j];
    }
    if (out.converged) out.converged[i] = conv ? 1 : 0;
    if (out.n_points) out.n_points[i] = K.n_points;
    if (out.tips) {
      double x = p[0], y = p[1], z = p[2];
      if (ROT) { const double x2 = __builtin_fma(rc, x, -(rs * y)), y2 = __builtin_fma(rs, x, rc * y); x = x2; y = y2; z = r22 * z; }
      out.tips[3 * i + 0] = x; out.tips[3 * i + 1] = y; out.tips[3 * i + 2] = z;
    }
  }
  if (lane_out) {
    lane_out->converged = conv;
#pragma unroll
    for (int j = 0; j < N; j++) lane_out->Li[j] = Li[j];
    double x = p[0], y = p[1], z = p[2];
    if (ROT) { const double x2 = __builtin_fma(rc, x, -(rs * y)), y2 = __builtin_fma(rs, x, rc * y); x = x2; y = y2; z = r22 * z; }
    lane_out->tip[0] = x; lane_out->tip[1] = y; lane_out->tip[2] = z;
  }
}

template <int N, bool ROT, bool WRITE_R>
__global__ __launch_bounds__(64, (N <= TRK_K1_TWO_WAVE_MAXN ? 2 : 1)) void fk_rk4_batch_uniform(
    const double *__restrict__ states, int64_t n, int64_t ld, RobotK K,
    const double *__restrict__ tab, const StepK *__restrict__ steps, int nsteps, FkOut out) {
  fk_uniform_body<N, ROT, WRITE_R>(states, n, ld, K, tab, steps, nsteps, out);
}

}  // namespace trk
