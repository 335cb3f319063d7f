// fk_retract_kernel.hpp -- K1 for retraction-enabled robots (state ends with s_start): every lane has
// its own arc-length grid t_range(s_start, L, dL) (tendon/TendonRobot.cpp:69-84), its own number of
// backbone points, a first interval in [dL/2, 1.5 dL) that may need two RK4 steps
// (integrate_times, call site TendonRobot.cpp:458-462), and its own home-shape tendon lengths
// (home_shape(s_start), TendonRobot.cpp:249-314).  The routing r(t), r'(t), r''(t) is evaluated per
// lane (polynomials with scalar coefficients + sincos) on demand per RK4 stage (rk4_step_routed).
//
// Lanes run their intervals aligned at the base (interval j of every lane in iteration j); lanes
// with a retracted, shorter backbone idle at the end of the wave's loop.
#pragma once
#include "fk_kernel.hpp"

namespace trk {

// get_poly_vecs + get_r_info2 (tendon/get_r_info.cpp:17-40,105-144) for one lane's abscissa t.
template <int N>
__device__ __forceinline__ void routing_lane(const PolyK *__restrict__ pk, int n_a, int n_m, double t, double (&out)[N * 6]) {
#pragma clang fp contract(off)
  double S[TRK_MAX_COEF], Sd[TRK_MAX_COEF], Sdd[TRK_MAX_COEF];
  S[0] = 1; Sd[0] = 0; Sdd[0] = 0;
  S[1] = t; Sd[1] = 1; Sdd[1] = 0;
#pragma unroll
  for (int i = 2; i < TRK_MAX_COEF; i++) { S[i] = t * S[i - 1]; Sd[i] = i * S[i - 1]; Sdd[i] = i * (i - 1) * S[i - 2]; }
#pragma unroll
  for (int j = 0; j < N; j++) {
    double C_a = 0, C_ad = 0, C_add = 0, D_m = 0, D_md = 0, D_mdd = 0;
#pragma unroll
    for (int i = 0; i < TRK_MAX_COEF; i++) {
      if (i < n_a) { const double c = pk->C[j][i]; C_a += c * S[i]; C_ad += c * Sd[i]; C_add += c * Sdd[i]; }
      if (i < n_m) { const double d = pk->D[j][i]; D_m += d * S[i]; D_md += d * Sd[i]; D_mdd += d * Sdd[i]; }
    }
    double sa, ca;
    sincos(C_a, &sa, &ca);
    out[6 * j + 0] = D_m * sa;
    out[6 * j + 1] = D_m * ca;
    out[6 * j + 2] = D_md * sa + D_m * (ca * C_ad);
    out[6 * j + 3] = D_md * ca + D_m * (-sa * C_ad);
    out[6 * j + 4] = D_mdd * sa + 2 * D_md * (ca * C_ad) - D_m * (sa * C_ad * C_ad) + D_m * (ca * C_add);
    out[6 * j + 5] = D_mdd * ca + 2 * D_md * (-sa * C_ad) - D_m * (ca * C_ad * C_ad) + D_m * (-sa * C_add);
  }
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

template <int N, bool ROT, bool WRITE_R>
__global__ __launch_bounds__(64) void fk_rk4_batch_retract(
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

  double rloc[N * 6];
  routing_lane<N>(pk, K.n_a, K.n_m, s, rloc);
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
                           [&](double tt, double (&ri)[N * 6]) { routing_lane<N>(pk, K.n_a, K.n_m, tt, ri); });
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
