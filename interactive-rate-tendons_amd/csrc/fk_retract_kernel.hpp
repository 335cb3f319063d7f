// fk_retract_kernel.hpp -- K1 for retraction-enabled robots (state ends with s_start): every lane has
// its own arc-length grid t_range(s_start, L, dL) (tendon/TendonRobot.cpp:69-84), its own number of
// backbone points, a first interval in [dL/2, 1.5 dL) that may need two RK4 steps
// (integrate_times, call site TendonRobot.cpp:458-462), and its own home-shape tendon lengths
// (home_shape(s_start), TendonRobot.cpp:249-314).  The routing r(t), r'(t), r''(t) is evaluated per
// lane, tendon by tendon, where the RK4 stage needs it (rk4_step_routed, route_tendon).
//
// Lanes run their intervals aligned at the base (interval j of every lane in iteration j); lanes
// with a retracted, shorter backbone idle at the end of the wave's loop.
#pragma once
#include "fk_kernel.hpp"

namespace trk {

// Per-lane routing (get_poly_vecs + get_r_info2, tendon/get_r_info.cpp:17-40,105-144).  A lane carries, per
// tendon, the routing angle theta at the abscissa it was last evaluated at together with its sine and
// cosine.  Evaluating at a new abscissa is Horner for theta, rho and their first two derivatives (scalar
// coefficient loads, wave-uniform degree) plus a rotation of (sin, cos) by dtheta = theta_new - theta_old
// with an 11th/12th-order Taylor pair -- |dtheta| is ~1e-2 for a half step; beyond 1/8 rad it falls back to
// sincos.  Re-evaluating at the same abscissa rotates by exactly zero.  (FK parity is a 1e-9 m tolerance:
// the rotations add ~1e-16 each.)
template <int N>
struct RouteCarry { double th[N], sn[N], cs[N]; };

// value, first and second derivative of sum_i c[i] t^i, c wave-uniform
__device__ __forceinline__ void horner3(const double *__restrict__ c, int n, double t, double &p, double &d1, double &d2) {
  p = 0.0; d1 = 0.0; d2 = 0.0;
  for (int i = n - 1; i >= 0; --i) { d2 = d2 * t + d1; d1 = d1 * t + p; p = p * t + c[i]; }
  d2 = d2 + d2;
}

template <int N>
__device__ __forceinline__ void route_anchor(const PolyK *__restrict__ pk, int n_a, double t, RouteCarry<N> &rc) {
#pragma unroll
  for (int j = 0; j < N; j++) {
    double th, a1, a2;
    horner3(pk->C[j], n_a, t, th, a1, a2);
    rc.th[j] = th;
    sincos(th, &rc.sn[j], &rc.cs[j]);
  }
}

template <int N>
__device__ __forceinline__ void route_tendon(const PolyK *__restrict__ pk, int n_a, int n_m, int j, double t, RouteCarry<N> &rc,
                                             double (&r6)[6]) {
#pragma clang fp contract(fast)
  double th, C_ad, C_add, D_m, D_md, D_mdd;
  horner3(pk->C[j], n_a, t, th, C_ad, C_add);
  horner3(pk->D[j], n_m, t, D_m, D_md, D_mdd);
  const double dth = th - rc.th[j];
  double sa, ca;
  if (fabs(dth) <= 0.125) {
    const double x2 = dth * dth;
    const double sd = dth * (1.0 + x2 * (-1.0 / 6 + x2 * (1.0 / 120 + x2 * (-1.0 / 5040 + x2 * (1.0 / 362880 + x2 * (-1.0 / 39916800))))));
    const double cd = 1.0 + x2 * (-0.5 + x2 * (1.0 / 24 + x2 * (-1.0 / 720 + x2 * (1.0 / 40320 + x2 * (-1.0 / 3628800 + x2 * (1.0 / 479001600))))));
    sa = rc.sn[j] * cd + rc.cs[j] * sd;
    ca = rc.cs[j] * cd - rc.sn[j] * sd;
  } else {
    sincos(th, &sa, &ca);
  }
  rc.th[j] = th; rc.sn[j] = sa; rc.cs[j] = ca;
  r6[0] = D_m * sa;
  r6[1] = D_m * ca;
  r6[2] = D_md * sa + D_m * (ca * C_ad);
  r6[3] = D_md * ca + D_m * (-sa * C_ad);
  r6[4] = D_mdd * sa + 2 * D_md * (ca * C_ad) - D_m * (sa * C_ad * C_ad) + D_m * (ca * C_add);
  r6[5] = D_mdd * ca + 2 * D_md * (-sa * C_ad) - D_m * (ca * C_ad * C_ad) + D_m * (-sa * C_add);
}

// integrand of the home tendon length: sqrt(rho'^2 + rho^2 theta'^2 + 1) (TendonRobot.cpp:300-307,
// util/poly.h:9-28)
__device__ __forceinline__ double home_ldot(const PolyK *__restrict__ pk, int j, int n_a, int n_m, double t) {
#pragma clang fp contract(off)
  double dd = 0, d = 0, cd = 0, tpow = 1;
#pragma unroll
  for (int i = 0; i < TRK_MAX_COEF; i++) {
    if (i < n_m) d += pk->D[j][i] * tpow;
    if (i + 1 < n_m) dd += ((i + 1) * pk->D[j][i + 1]) * tpow;
    if (i + 1 < n_a) cd += ((i + 1) * pk->C[j][i + 1]) * tpow;
    tpow *= t;
  }
  return sqrt(dd * dd + (d * d) * (cd * cd) + 1);
}

// One wave per SIMD: the per-lane routing state (carried angles, abscissae, home-length sums) adds ~70
// registers to K1's 255; held to 256 registers (two waves) the kernel spills ~290 and runs 50 % slower
// (measured: 12.7 vs 8.5 ms per 2^19 three-tendon configurations).
#ifndef TRK_K1R_WAVES
#define TRK_K1R_WAVES 1
#endif
template <int N, bool ROT, bool WRITE_R>
__global__ __launch_bounds__(64, TRK_K1R_WAVES) void fk_rk4_batch_retract(
    const double *__restrict__ states, int64_t n, int64_t ld, RobotK K, const PolyK *__restrict__ pk,
    double *__restrict__ pscr /* [P][ld] scratch for the range() abscissae */, FkOut out) {
#pragma clang fp contract(fast)
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const bool live = i < n;
  const int64_t ic = live ? i : (n - 1);
  const int S = K.state_size, Pmax = K.n_points;
  const double L = K.L, dL = K.dL;
  double tau[N];
#pragma unroll
  for (int j = 0; j < N; j++) tau[j] = states[ic * S + j];
  double rc = 1.0, rs = 0.0, r22 = 1.0;
  if (ROT) {
    const double th = states[ic * S + N];
    rs = sin(th); rc = cos(th);
    r22 = (1.0 - rc) + rc;
  }
  const double s_raw = states[ic * S + (S - 1)];
  // s_start below 0 lies outside the RetractionStateSpace bounds [0, L] (Problem.cpp:142); the
  // reference would integrate a longer backbone than any buffer here holds -> reported unconverged.
  const bool negative = s_raw < 0.0;
  double s = s_raw;
  if (s > L) s = L;                                  // TendonRobot.cpp:359
  const bool single = (s == L) || negative;          // :361-372

  // forward pass of util::range: p_0 = s, p_{k+1} = p_k + dL while p <= L - dL/2
  int m = 0;
  {
    double q = s;
    for (int k = 0; k < Pmax; k++) {
      const bool go = !single && (q <= L - (dL / 2)) && k < Pmax - 1;
      if (!__any(go)) break;
      if (go) { pscr[(int64_t)k * ld + ic] = q; m = k + 1; q += dL; }
    }
  }
  const int P_lane = single ? 1 : m + 1;

  RouteCarry<N> rcy;
  route_anchor<N>(pk, K.n_a, s, rcy);
  double rloc[N * 6];
#pragma unroll
  for (int j = 0; j < N; j++) {
    double r6[6];
    route_tendon<N>(pk, K.n_a, K.n_m, j, s, rcy, r6);
#pragma unroll
    for (int q = 0; q < 6; q++) rloc[6 * j + q] = r6[q];
  }
  double v[3], u[3];
  bool conv;
  initial_bending<N>(tau, rloc, K, v, u, conv);
  if (single) { conv = !negative; }                  // early return keeps the default converged = true

  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double p[3] = {0, 0, 0};
  double Lb = 0;
  double Li[N];
#pragma unroll
  for (int j = 0; j < N; j++) Li[j] = 0;

  auto store_point = [&](int j) {
    const int64_t o = (int64_t)j * ld + i;
    double x = p[0], y = p[1], z = p[2];
    if (ROT) { const double x2 = rc * x - rs * y, y2 = rs * x + rc * y; x = x2; y = y2; z = r22 * z; }
    out.px[o] = x; out.py[o] = y; out.pz[o] = z;
    if (WRITE_R) {
      const int64_t PS = (int64_t)Pmax * ld;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        double a = R[c * 3 + 0], b = R[c * 3 + 1], cc = R[c * 3 + 2];
        if (ROT) { const double a2 = rc * a - rs * b, b2 = rs * a + rc * b; a = a2; b = b2; cc = r22 * cc; }
        out.R[(c * 3 + 0) * PS + o] = a; out.R[(c * 3 + 1) * PS + o] = b; out.R[(c * 3 + 2) * PS + o] = cc;
      }
    }
  };
  if (live) store_point(0);

  // home-shape tendon lengths: home_shape clamps s_start into [0, L] (TendonRobot.cpp:257-258)
  bool any_general = false;
#pragma unroll
  for (int j = 0; j < N; j++) any_general = any_general || (pk->home_kind[j] == 2);
  const int nint = P_lane - 1;
  const int ne = (nint % 2 != 0) ? nint - 1 : nint;  // intervals covered by Simpson's rule
  double hsum[N], hodd[N];
#pragma unroll
  for (int j = 0; j < N; j++) { hsum[j] = 0; hodd[j] = 0; }
  auto home_visit = [&](int jpt, double t) {
    if (!any_general || single) return;
#pragma unroll
    for (int j = 0; j < N; j++) {
      if (pk->home_kind[j] != 2) continue;
      const double val = home_ldot(pk, j, K.n_a, K.n_m, t);
      if (ne > 0 && jpt <= ne) hsum[j] += ((jpt == 0 || jpt == ne) ? 1.0 : ((jpt & 1) ? 4.0 : 2.0)) * val;
      if (nint % 2 != 0 && jpt >= P_lane - 2) hodd[j] += val;
    }
  };

  double cur = L - (L - s);                          // t[0]: the mirrored `end` sample of t_range
  home_visit(0, cur);
  for (int j = 0; j < Pmax - 1; j++) {
    const bool act = j < P_lane - 1;
    if (!__any(act)) break;
    double tn = cur;
    if (act) tn = L - (pscr[(int64_t)(m - 1 - j) * ld + ic] - s);      // t[j+1]
    // integrate_times: steps of min(dL, t[j+1] - cur) while t[j+1] - cur > eps
    for (int sub = 0; sub < 4; sub++) {
      const bool go = act && (tn - cur > 2.220446049250313e-16);
      if (!__any(go)) break;
      if (go) {
        const double h = (dL < tn - cur) ? dL : (tn - cur);
        rk4_step_routed<N>(R, v, u, p, Lb, Li, tau, K, cur, h,
                           [&](double tt, int j, double (&r6)[6]) { route_tendon<N>(pk, K.n_a, K.n_m, j, tt, rcy, r6); });
        cur += h;
      }
    }
    if (act) {
      cur = tn;                                      // the next interval restarts at exactly t[j+1]
      if (live) store_point(j + 1);
      home_visit(j + 1, tn);
    }
  }

  if (live) {
    if (out.L) out.L[i] = Lb;
    if (out.Li) {
#pragma unroll
      for (int j = 0; j < N; j++) out.Li[(int64_t)j * ld + i] = Li[j];
    }
    if (out.converged) out.converged[i] = conv ? 1 : 0;
    if (out.n_points) out.n_points[i] = P_lane;
    if (out.tips) {
      double x = p[0], y = p[1], z = p[2];
      if (ROT) { const double x2 = rc * x - rs * y, y2 = rs * x + rc * y; x = x2; y = y2; z = r22 * z; }
      out.tips[3 * i + 0] = x; out.tips[3 * i + 1] = y; out.tips[3 * i + 2] = z;
    }
    if (out.home_Li) {
      double sh = s_raw;
      if (sh < 0.0) sh = 0.0;
      if (sh > L) sh = L;
      const double Lh = L - sh;
#pragma unroll
      for (int j = 0; j < N; j++) {
        double val;
        if (sh == L) val = 0.0;
        else if (pk->home_kind[j] == 0) val = Lh;
        else if (pk->home_kind[j] == 1) val = Lh * pk->helix_scale[j];
        else {
          const double odd = (nint % 2 != 0) ? 0.5 * dL * hodd[j] : 0.0;
          val = (P_lane < 2) ? 0.0 : ((ne == 0) ? odd : odd + (hsum[j] * dL / 3.0));
        }
        out.home_Li[(int64_t)j * ld + i] = val;
      }
    }
  }
}

}  // namespace trk
