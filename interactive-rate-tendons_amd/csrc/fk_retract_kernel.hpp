// fk_retract_kernel.hpp -- K1 for retraction-enabled robots (state ends with s_start): every lane has
// its own arc-length grid t_range(s_start, L, dL) (tendon/TendonRobot.cpp:69-84), its own number of
// backbone points, a first interval in [dL/2, 1.5 dL) that may need two RK4 steps
// (integrate_times, call site TendonRobot.cpp:458-462), and its own home-shape tendon lengths
// (home_shape(s_start), TendonRobot.cpp:249-314).  The routing r(t), r'(t), r''(t) is evaluated per
// lane, tendon by tendon, where the RK4 stage needs it (rk4_step_routed, route_tendon).
//
// t_range spaces its abscissae uniformly FROM THE TIP: apart from the first interval (s_start to the first
// grid point, length in [dL/2, 1.5 dL)) every lane integrates over the same grid L - k dL as the
// unretracted robot, to an ulp.  So a lane first covers its own first interval with per-lane routing
// (route_tendon), and then the wave runs TIP-ALIGNED: iteration = interval of the shared grid, retracted
// lanes join late instead of finishing early, and the routing comes from K1's scalar table -- K1's
// inner loop and K1's two-wave register budget.  A lane's point i is the shared grid's point
// i + (P - P_lane) and is stored in THAT row, so the stores of a wave stay coalesced; every consumer of a
// retraction robot's points (K2, K5, K8, the edge pair test, the host unpack) adds the same row offset.
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

// Waves per SIMD (measured, profiles/r02/retract_widths_v1.txt: ms per 2^18 configurations, s_start ~ U[0, L), rotation and
// general routing; one wave | two waves).  K1r with stored points: two waves pay up to 4 tendons (N=4 3.31 | 3.15, N=5
// 3.68 | 3.77, N=6 4.16 | 4.31, N=8 5.93 | 6.37).  The verdict-only form (fk_verdict_retract: no stores, the sweep's loads
// between the RK4 steps want a second wave to hide behind) is faster with two waves at every width (N=4 1.97 | 1.82, N=5
// 2.34 | 2.16, N=6 2.58 | 2.36, N=7 3.02 | 2.77, N=8 3.53 | 3.49).
#ifndef TRK_K1R_TWO_WAVE_MAXN
#define TRK_K1R_TWO_WAVE_MAXN 4
#endif
#ifndef TRK_VR_TWO_WAVE_MAXN
#define TRK_VR_TWO_WAVE_MAXN 8
#endif

// What a lane knows about its configuration when the integration ends (the verdict-only kernel continues from here).
template <int N>
struct FkLaneR {
  bool converged;
  int np;                     // the lane's number of backbone points
  double Li[N], home_Li[N];
};

// What the prologue kernel hands to the tip-aligned kernel (PHASE 1 -> PHASE 2 of fk_retract_body): the integrator's state at
// the end of the lane's own first interval, plane by plane, [field][ld] with ld = the launch's lane count rounded to 64 -- lane i
// of both launches is the same configuration, so both sides are coalesced.  Fields: R 0-8, v 9-11, u 12-14, p 15-17, L_i 18 ..
// 18 + N - 1, converged (as 0.0 / 1.0), then the configuration's S state coordinates (so that the second kernel does not gather
// them once more through the batch order: 40 scattered bytes cost a 128-byte line each), then the backbone-length quadrature so far.
struct RetractHandoff {
  double *data;
  int64_t ld;
};

// The body of K1r.  on_point.tip_point(row, on, first, x, y, z) is called in wave-uniform control flow for every observed
// backbone point (after rotate_z): `row` is the point's tip-aligned row (the lane's point j sits in row j + P - P_lane; it
// is the lane's own row for its first two points and wave-uniform afterwards), `on` says whether this lane has a point in
// this call, `first` whether it is the lane's point 0; on_point.begin(converged && live) precedes the first call.
// row_map (optional): lane i integrates configuration row_map[i] of `states`.
// PHASE 0: everything.  PHASE 1: up to the end of the lane's own first interval (initial bending, per-lane routing) -- the part
// whose register demand spills -- then the state goes to `ho` and the body returns; no hook is called, nothing else is written.
// PHASE 2: the state comes from `ho`, the hook sees points 0 and 1, and the tip-aligned loop and the epilogue follow: this kernel
// holds neither the per-lane routing nor the fixed-point solve.
template <int N, bool ROT, bool WRITE_R, class OnPoint = NoPointHook, int PHASE = 0>
__device__ __forceinline__ void fk_retract_body(
    const double *__restrict__ states, int64_t n, int64_t ld, const RobotK &K, const PolyK *__restrict__ pk,
    const double *__restrict__ tab /* K1's routing table of the s_start = 0 grid */, const StepK *__restrict__ steps, int nsteps,
    int k_first /* first step after the grid's own first interval */, const double *__restrict__ tgrid /* [P] shared abscissae */,
    const double *__restrict__ hl /* [P][N] home-length integrand at the shared abscissae */, const FkOut &out,
    OnPoint &&on_point = NoPointHook(), const int32_t *__restrict__ row_map = nullptr, FkLaneR<N> *lane_out = nullptr,
    int k_begin = 0 /* wave-uniform: no lane of this wave has a point in a row that a step before k_begin ends in (0 = unknown) */,
    const RetractHandoff *ho = nullptr,
    bool by_config = false /* with row_map: the stored outputs (points, R, L, L_i, converged, n_points, home L_i) go to the
                              CONFIGURATION's column -- scattered 8-byte stores, 1 % of the kernel's instructions -- so that a batch
                              integrated in the order of its backbone lengths lands in the caller's order */) {
#pragma clang fp contract(fast)
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const bool live = i < n;
  const int64_t il = live ? i : (n - 1);
  const int64_t ic = row_map ? (int64_t)row_map[il] : il;
  const int64_t oc = by_config ? ic : i;              // the column this lane's stored outputs go to
  const int S = K.state_size, Pmax = K.n_points;
  const double L = K.L, dL = K.dL;
  // the state: from the caller's array (through the batch order), or -- PHASE 2 -- from the hand-over planes, coalesced
  auto state_at = [&](int d) -> double {
    if constexpr (PHASE == 2) return ho->data[(int64_t)(19 + N + d) * ho->ld + (live ? i : n - 1)];
    else return states[ic * S + d];
  };
  double tau[N];
#pragma unroll
  for (int j = 0; j < N; j++) tau[j] = state_at(j);
  double rc = 1.0, rs = 0.0, r22 = 1.0;
  if (ROT) {
    const double th = state_at(N);
    rs = sin(th); rc = cos(th);
    r22 = (1.0 - rc) + rc;
  }
  const double s_raw = state_at(S - 1);
  // s_start below 0 lies outside the RetractionStateSpace bounds [0, L] (Problem.cpp:142); the
  // reference would integrate a longer backbone than any buffer here holds -> reported unconverged.
  const bool negative = s_raw < 0.0;
  double s = s_raw;
  if (s > L) s = L;                                  // TendonRobot.cpp:359
  const bool single = (s == L) || negative;          // :361-372

  // forward pass of util::range: p_0 = s, p_{k+1} = p_k + dL while p <= L - dL/2 -> number of points
  int m = 0;
  {
    double q = s;
    for (int k = 0; k < Pmax; k++) {
      const bool go = !single && (q <= L - (dL / 2)) && k < Pmax - 1;
      if (!__any(go)) break;
      if (go) { m = k + 1; q += dL; }
    }
  }
  const int P_lane = single ? 1 : m + 1;
  const int shift = Pmax - P_lane;                   // lane point i = shared grid point i + shift (i >= 1)

  RouteCarry<N> rcy;
  double v[3], u[3];
  bool conv = true;
  if constexpr (PHASE != 2) {
    route_anchor<N>(pk, K.n_a, s, rcy);
    double rloc[N * 6];
#pragma unroll
    for (int j = 0; j < N; j++) {
      double r6[6];
      route_tendon<N>(pk, K.n_a, K.n_m, j, s, rcy, r6);
#pragma unroll
      for (int q = 0; q < 6; q++) rloc[6 * j + q] = r6[q];
    }
    initial_bending<N>(tau, rloc, K, v, u, conv);
    if (single) { conv = !negative; }                // early return keeps the default converged = true
  }

  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double p[3] = {0, 0, 0};
  double Lb = 0;
  double Li[N];
#pragma unroll
  for (int j = 0; j < N; j++) Li[j] = 0;
  double p1[3] = {0, 0, 0};                          // PHASE 2: the lane's point 1 (the end of its own first interval)
  double R1[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};        // ... and the frame there (point 0 keeps the identity: the stored-point form writes R)
  if constexpr (PHASE == 2) {
    const double *__restrict__ h = ho->data + (live ? i : n - 1);
    const int64_t hl = ho->ld;
#pragma unroll
    for (int q = 0; q < 9; q++) R1[q] = h[q * hl];
#pragma unroll
    for (int q = 0; q < 3; q++) { v[q] = h[(9 + q) * hl]; u[q] = h[(12 + q) * hl]; p1[q] = h[(15 + q) * hl]; }
#pragma unroll
    for (int j = 0; j < N; j++) Li[j] = h[(18 + j) * hl];
    conv = h[(18 + N) * hl] != 0.0;
    if (out.L) Lb = h[(19 + N + S) * hl];
  }

  if constexpr (PHASE != 1) on_point.begin(conv && live);
  // `row` may differ from lane to lane (the lane's first two points) or be wave-uniform (the tip-aligned loop); `on`: the
  // lane has a point here.  The hook runs for the whole wave (it holds ballots and workgroup barriers).
  auto store_point = [&](int row, bool on, bool is_first, bool wave_row = false) {
    double x = p[0], y = p[1], z = p[2];
    // rotate_z (tendon/TendonResult.cpp:13-18); explicit FMAs so that every kernel holding this body forms the same bits
    if (ROT) { const double x2 = __builtin_fma(rc, x, -(rs * y)), y2 = __builtin_fma(rs, x, rc * y); x = x2; y = y2; z = r22 * z; }
    on_point.tip_point(row, on, is_first, x, y, z, wave_row);
    if (!(on && live)) return;
    const int64_t o = (int64_t)row * ld + oc;
    if (out.px) { out.px[o] = x; out.py[o] = y; out.pz[o] = z; }
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
  if constexpr (PHASE != 1) store_point(shift, true, true);    // rows are aligned at the tip: the lane's point j goes to row j + shift

  // the lane's own first interval: s -> shared grid point shift + 1, steps of min(dL, remaining) while
  // remaining > eps (integrate_times), routing evaluated per lane
  // (s_start within dL/2 of L gives a one-point backbone that is not the `single` early return: no interval at all)
  const bool first = !single && P_lane >= 2;
  if (__any(first)) {
    if constexpr (PHASE != 2) {
      double cur = L - (L - s);                      // t[0]: the mirrored `end` sample of t_range
      double tn = cur;
      if (first) tn = tgrid[shift + 1];
      for (int sub = 0; sub < 4; sub++) {
        const bool go = first && (tn - cur > 2.220446049250313e-16);
        if (!__any(go)) break;
        if (go) {
          const double h = (dL < tn - cur) ? dL : (tn - cur);
          rk4_step_routed<N>(R, v, u, p, Lb, Li, tau, K, cur, h,
                             [&](double tt, int j, double (&r6)[6]) { route_tendon<N>(pk, K.n_a, K.n_m, j, tt, rcy, r6); });
          cur += h;
        }
      }
    } else {
      p[0] = p1[0]; p[1] = p1[1]; p[2] = p1[2];
#pragma unroll
      for (int q = 0; q < 9; q++) R[q] = R1[q];
    }
    if constexpr (PHASE != 1) store_point(shift + 1, first, false);
  }
  if constexpr (PHASE == 1) {
    // hand the state over and stop (the quadrature of the backbone length is not carried: no verdict path asks for L)
    if (live) {
      double *__restrict__ h = ho->data + i;
      const int64_t hl = ho->ld;
#pragma unroll
      for (int q = 0; q < 9; q++) h[q * hl] = R[q];
#pragma unroll
      for (int q = 0; q < 3; q++) { h[(9 + q) * hl] = v[q]; h[(12 + q) * hl] = u[q]; h[(15 + q) * hl] = p[q]; }
#pragma unroll
      for (int j = 0; j < N; j++) h[(18 + j) * hl] = Li[j];
      h[(18 + N) * hl] = conv ? 1.0 : 0.0;
#pragma unroll
      for (int j = 0; j < N; j++) h[(19 + N + j) * hl] = tau[j];
      if (ROT) h[(19 + N + N) * hl] = states[ic * S + N];
      h[(19 + N + S - 1) * hl] = s_raw;
      h[(19 + N + S) * hl] = Lb;                           // (the stored-point form returns the backbone length; the verdict forms never read it)
    }
    return;
  }

  // tip-aligned: step k of the shared grid ends at its point steps[k].obs = the lane's point obs - shift
  // (k_begin reaches this function through a pointer that was itself loaded from memory -- a flat pointer, whose loads the
  // compiler must treat as lane-dependent.  Left like that, the loop counter lived in a VGPR, the loop ran under an exec mask
  // and every step's three routing-table rows were fetched with per-lane vector loads into ~108 VGPRs, spilling the
  // integrator's state around them: 470 - 550 B of scratch per lane, whose write-backs showed up as 1 - 4 KB of HBM
  // traffic per check (profiles/r03/traffic_split_v1.json).  It IS wave-uniform: say so.)
  const int k_start = __builtin_amdgcn_readfirstlane(k_begin > k_first ? k_begin : k_first);
  for (int k = k_start; k < nsteps; k++) {
    const int obs = steps[k].obs;
    const int ipt = obs - shift;
    const bool act = !single && ipt >= 2;
    if (!__any(act)) continue;
    if (act) {
      const double h = steps[k].h;
      const double *__restrict__ rt = tab + (size_t)(1 + 3 * k) * (N * 6);
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
        const double *__restrict__ ri = rt + (st == 0 ? 0 : (st == 3 ? 2 : 1)) * (N * 6);
        const double bw = (st == 0 || st == 3) ? b1 : b2;
        const double aw = (st == 2) ? h : hh;
        double dv[3], du[3], sd[N];
        strain_rates<N>(sv, su, tau, ri, K, dv, du, sd);
        p[0] += bw * (sR[0] * sv[0] + sR[3] * sv[1] + sR[6] * sv[2]);
        p[1] += bw * (sR[1] * sv[0] + sR[4] * sv[1] + sR[7] * sv[2]);
        p[2] += bw * (sR[2] * sv[0] + sR[5] * sv[1] + sR[8] * sv[2]);
        if (out.L) {                                       // wave-uniform: the validity paths do not ask for the backbone length
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
    store_point(obs, act, false, true);
  }

  // home-shape tendon lengths: home_shape clamps s_start into [0, L] (TendonRobot.cpp:257-258); composite
  // Simpson over the lane's points with a trapezoid for a trailing odd interval (see tendon_hip.hip: home_lengths)
  bool any_general = false;
#pragma unroll
  for (int j = 0; j < N; j++) any_general = any_general || (pk->home_kind[j] == 2);
  const int nint = P_lane - 1;
  const int ne = (nint % 2 != 0) ? nint - 1 : nint;  // intervals covered by Simpson's rule
  double hsum[N], hodd[N];
#pragma unroll
  for (int j = 0; j < N; j++) { hsum[j] = 0; hodd[j] = 0; }
  auto home_add = [&](int jpt, int j, double val) {
    if (ne > 0 && jpt <= ne) hsum[j] += ((jpt == 0 || jpt == ne) ? 1.0 : ((jpt & 1) ? 4.0 : 2.0)) * val;
    if (nint % 2 != 0 && jpt >= P_lane - 2) hodd[j] += val;
  };
  if (any_general && __any(!single)) {
    // the integrand at the lane's points: its own base abscissa, then the shared grid (a pass of its own after
    // the RK4 loop, so that the sums are not live -- 2 N more registers -- across it)
    if (!single) {
#pragma unroll
      for (int j = 0; j < N; j++)
        if (pk->home_kind[j] == 2) home_add(0, j, home_ldot(pk, j, K.n_a, K.n_m, L - (L - s)));
    }
    for (int ipt = 1; ipt < Pmax; ipt++) {
      const bool on = !single && ipt < P_lane;
      if (!__any(on)) break;
      if (on) {
#pragma unroll
        for (int j = 0; j < N; j++)
          if (pk->home_kind[j] == 2) home_add(ipt, j, hl[(int64_t)(ipt + shift) * N + j]);
      }
    }
  }


  // home_shape(s_start).L_i (TendonRobot.cpp:249-314)
  double home[N];
  if (out.home_Li || lane_out) {
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
      home[j] = val;
    }
  }
  if (live) {
    if (out.L) out.L[oc] = Lb;
    if (out.Li) {
#pragma unroll
      for (int j = 0; j < N; j++) out.Li[(int64_t)j * ld + oc] = Li[j];
    }
    if (out.converged) out.converged[oc] = conv ? 1 : 0;
    if (out.n_points) out.n_points[oc] = P_lane;
    if (out.tips) {
      double x = p[0], y = p[1], z = p[2];
      if (ROT) { const double x2 = __builtin_fma(rc, x, -(rs * y)), y2 = __builtin_fma(rs, x, rc * y); x = x2; y = y2; z = r22 * z; }
      out.tips[3 * ic + 0] = x; out.tips[3 * ic + 1] = y; out.tips[3 * ic + 2] = z;   // (row_map: the configuration's own place)
    }
    if (out.home_Li) {
#pragma unroll
      for (int j = 0; j < N; j++) out.home_Li[(int64_t)j * ld + oc] = home[j];
    }
  }
  if (lane_out) {
    lane_out->converged = conv;
    lane_out->np = P_lane;
#pragma unroll
    for (int j = 0; j < N; j++) { lane_out->Li[j] = Li[j]; lane_out->home_Li[j] = home[j]; }
  }
}

template <int N, bool ROT, bool WRITE_R>
__global__ __launch_bounds__(64, (N <= TRK_K1R_TWO_WAVE_MAXN ? 2 : 1)) void fk_rk4_batch_retract(
    const double *__restrict__ states, int64_t n, int64_t ld, RobotK K, const PolyK *__restrict__ pk,
    const double *__restrict__ tab, const StepK *__restrict__ steps, int nsteps, int k_first, const double *__restrict__ tgrid,
    const double *__restrict__ hl, FkOut out,
    const int32_t *__restrict__ order /* null, or: lane i integrates configuration order[i] -- the batch by backbone length, so that a
                                         wave holds backbones of one length (retraction_order); the outputs land in the caller's order */,
    const int32_t *__restrict__ wave_k_begin /* null, or per wave of that order: the step its tip-aligned loop may start at */) {
  const int kb = wave_k_begin ? wave_k_begin[blockIdx.x] : 0;
  fk_retract_body<N, ROT, WRITE_R>(states, n, ld, K, pk, tab, steps, nsteps, k_first, tgrid, hl, out, NoPointHook(), order, nullptr, kb, nullptr,
                                   order != nullptr);
}

// The verdict form's first launch: every lane's own first interval (base strains by the fixed-point solve, routing polynomials
// evaluated per lane and stage, up to two RK4 steps) needs more than the 256 registers of a two-wave kernel, so it runs HERE, one
// wave per SIMD with the whole register file, and hands the integrator's state over in coalesced planes.  (The stored-point form
// was tried in the same two launches in round 5: 3.97 against 3.52 ms per 2^19 for the one kernel in length order -- the planes
// cost it more than the spills.)
template <int N, bool ROT>
__global__ __launch_bounds__(64, 1) void fk_retract_prologue(
    const double *__restrict__ states, int64_t n, RobotK K, const PolyK *__restrict__ pk, const double *__restrict__ tab,
    const StepK *__restrict__ steps, int nsteps, int k_first, const double *__restrict__ tgrid, const double *__restrict__ hl,
    const int32_t *__restrict__ perm, RetractHandoff ho) {
  FkOut out{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  fk_retract_body<N, ROT, false, NoPointHook, 1>(states, n, 0, K, pk, tab, steps, nsteps, k_first, tgrid, hl, out, NoPointHook(), perm, nullptr, 0, &ho);
}

}  // namespace trk
