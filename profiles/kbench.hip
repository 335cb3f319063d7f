// kbench.hip -- a standalone timer for the RK4 loop every FK kernel of the library holds (csrc/fk_kernel.hpp: fk_uniform_body),
// so that source variants of the right-hand side can be compiled and timed ON the GPU box in seconds each, without rebuilding
// the library:
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I interactive-rate-tendons_amd/csrc [-DVARIANT...] \
//         -DKB_N=3 profiles/kbench.hip -o /tmp/kb && /tmp/kb
//
// Robots: KB_N == 3 -> config 2's helical robot, KB_N == 4 -> config 3's quadratic routing; 129 points, tau ~ U[0, tmax)^N from a
// small LCG.  Prints the best and the median of KB_REPS launches over 2^KB_LOG2 configurations, the tips' checksum (so variants
// can be compared bit for bit) and the resources hipcc gave the kernel.  Test infrastructure: nothing here ships.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "tr_types.hpp"
#include "fk_kernel.hpp"
#ifdef KB_VERDICT
#include "verdict_kernel.hpp"     // -DKB_VERDICT: time fk_verdict<N> (the per-point sweep hook in the loop) against an EMPTY 256^3 grid
#ifdef KB_SIG                     // -DKB_VERDICT -DKB_SIG: the edge samples' form, which also writes every point's cell signature
constexpr bool kSig = true;
#else
constexpr bool kSig = false;
#endif
#endif

#ifndef KB_N
#define KB_N 3
#endif
#ifndef KB_LOG2
#define KB_LOG2 20
#endif
#ifndef KB_REPS
#define KB_REPS 7
#endif
#ifndef KB_WAVES
#define KB_WAVES 2
#endif

namespace trk {
template <int N> __global__ __launch_bounds__(64, KB_WAVES) void rk4_only(const double *__restrict__ states, int64_t n, RobotK K,
                                                                           const double *__restrict__ tab, const StepK *__restrict__ steps,
                                                                           int nsteps, double *__restrict__ tips) {
  FkOut out{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, tips, nullptr, nullptr, nullptr};
  fk_uniform_body<N, false, false, false>(states, n, 0, K, tab, steps, nsteps, out);
}
}  // namespace trk

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static void routing_at(int N, const double C[][3], const double D[][3], double t, double *out) {
  for (int j = 0; j < N; j++) {
    const double th = C[j][0] + C[j][1] * t + C[j][2] * t * t, th1 = C[j][1] + 2 * C[j][2] * t, th2 = 2 * C[j][2];
    const double rho = D[j][0] + D[j][1] * t + D[j][2] * t * t, rho1 = D[j][1] + 2 * D[j][2] * t, rho2 = 2 * D[j][2];
    const double sa = std::sin(th), ca = std::cos(th);
    double *o = out + 6 * j;
    o[0] = rho * sa; o[1] = rho * ca;
    o[2] = rho1 * sa + rho * (ca * th1);
    o[3] = rho1 * ca + rho * (-sa * th1);
    o[4] = rho2 * sa + 2 * rho1 * (ca * th1) - rho * (sa * th1 * th1) + rho * (ca * th2);
    o[5] = rho2 * ca + 2 * rho1 * (-sa * th1) - rho * (ca * th1 * th1) + rho * (-sa * th2);
  }
}

int main() {
  constexpr int N = KB_N;
  const double L = 0.2, dL = L / 128, ro = 0.01, ri = 0.0, E = 2.1e6, nu = 0.3;
  double C[8][3] = {}, D[8][3] = {};
  if (N == 3) for (int k = 0; k < 3; k++) { C[k][0] = 2 * M_PI * k / 3; C[k][1] = 5.0; D[k][0] = 0.01; }
  else {
    const double c1[4] = {3.0, -2.0, 4.0, -5.0}, c2[4] = {10.0, 15.0, -12.0, 8.0}, d1[4] = {-0.01, 0.005, 0.0, -0.005};
    for (int k = 0; k < N; k++) { C[k][0] = M_PI * k / 2; C[k][1] = c1[k % 4]; C[k][2] = c2[k % 4]; D[k][0] = 0.01; D[k][1] = d1[k % 4]; }
  }
  RobotK K{};
  {
    const double ro2 = ro * ro, ri2 = ri * ri, I = 0.25 * M_PI * (ro2 * ro2 - ri2 * ri2), Ar = M_PI * (ro2 - ri2), J = 2 * I, G = E / (2 * (1 + nu));
    K.kb0 = E * I; K.kb2 = J * G; K.ikb0 = 1 / K.kb0; K.ikb2 = 1 / K.kb2; K.ks0 = G * Ar; K.ks2 = E * Ar; K.iks0 = 1 / K.ks0; K.iks2 = 1 / K.ks2;
  }
  K.residual_threshold = 5e-6; K.radius = 0.015; K.L = L; K.dL = dL; K.n_tendons = N; K.n_a = 3; K.n_m = 3; K.state_size = N; K.n_points = 129;
  std::vector<StepK> steps;
  std::vector<double> tab((1 + 3 * 128) * N * 6);
  routing_at(N, C, D, 0.0, tab.data());
  for (int k = 0; k < 128; k++) {
    const double tk = k * dL;
    steps.push_back(StepK{dL, k + 1, 0});
    for (int q = 0; q < 3; q++) routing_at(N, C, D, tk + 0.5 * q * dL, &tab[(size_t)(1 + 3 * k + q) * N * 6]);
  }
#ifdef KB_COUNT                   // -DKB_COUNT=<n>: a launch that is not a whole number of rounds of the chip's 2048 wave slots
  const int64_t n = KB_COUNT;
#else
  const int64_t n = (int64_t)1 << KB_LOG2;
#endif
  std::vector<double> st((size_t)n * N);
  unsigned long long lcg = 88172645463325252ull;
  const double tmax = N == 3 ? 10.0 : 20.0;
  for (auto &x : st) { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; x = tmax * (double)(lcg >> 11) * (1.0 / 9007199254740992.0); }
  double *d_st, *d_tab, *d_tips; StepK *d_steps;
  CK(hipMalloc(&d_st, st.size() * 8)); CK(hipMalloc(&d_tab, tab.size() * 8)); CK(hipMalloc(&d_tips, (size_t)n * 24)); CK(hipMalloc(&d_steps, steps.size() * sizeof(StepK)));
  CK(hipMemcpy(d_st, st.data(), st.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_steps, steps.data(), steps.size() * sizeof(StepK), hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
#ifdef KB_VERDICT
  trk::VerdictArgs va{};
  va.P = 129; va.CH = (int)std::lround(2.0 * K.radius / dL); va.NM = (128 + va.CH - 1) / va.CH + 1;
  GridK &g = va.g;
  g.xmin = g.ymin = g.zmin = -0.25; g.xmax = g.ymax = g.zmax = 0.25; g.N = 256; g.Nb = 64; g.rot_is_identity = 1;
  g.dx = g.dy = g.dz = 0.5 / 256; g.inv_dx = g.inv_dy = g.inv_dz = 1 / g.dx;
  for (int q = 0; q < 9; q++) g.inv_rot[q] = (q % 4 == 0) ? 1.0 : 0.0;
  for (int q = 0; q < 3; q++) { va.box[2 * q] = -0.25 + 1e-6 * 0.5; va.box[2 * q + 1] = 0.25 - 1e-6 * 0.5; }
  uint64_t *d_grid, *d_bits; int32_t *d_fb; uint32_t *d_fbc; trk::VerdictArgs *d_va;
  CK(hipMalloc(&d_grid, (size_t)64 * 64 * 64 * 8 * 2)); CK(hipMemset(d_grid, 0, (size_t)64 * 64 * 64 * 8 * 2));
  CK(hipMalloc(&d_bits, (size_t)((n + 63) / 64) * 8)); CK(hipMalloc(&d_fb, (size_t)n * 4)); CK(hipMalloc(&d_fbc, 4)); CK(hipMemset(d_fbc, 0, 4));
  va.grid = d_grid; va.near_grid = d_grid + 64 * 64 * 64; va.valid_bits = d_bits; va.fb_list = d_fb; va.fb_count = d_fbc;
  va.radius = K.radius;
  uint32_t *d_sig = nullptr;
  if (kSig) { va.sig_stride = (129 + 15) / 16 * 16; CK(hipMalloc(&d_sig, (size_t)n * va.sig_stride * 4)); va.sig = d_sig; }
  for (int j = 0; j < N; j++) { va.min_len[j] = K.min_len[j] = -1.0; va.max_len[j] = K.max_len[j] = 1.0; va.home_Li[j] = K.home_Li[j] = L; }
  va.finish_hot();
  CK(hipMalloc(&d_va, sizeof(va))); CK(hipMemcpy(d_va, &va, sizeof(va), hipMemcpyHostToDevice));
  const size_t lds = trk::verdict_lds_bytes(va.NM, kSig);
#endif
  std::vector<float> ms;
  for (int r = 0; r < KB_REPS + 2; r++) {
    CK(hipEventRecord(a));
#ifdef KB_VERDICT
    hipLaunchKernelGGL((trk::fk_verdict<N, false, false, kSig>), dim3((unsigned)((n + 63) / 64)), dim3(64), lds, nullptr, d_st, n, K, d_tab, d_steps, 128, d_tips, d_va);
#else
    hipLaunchKernelGGL((trk::rk4_only<N>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, nullptr, d_st, n, K, d_tab, d_steps, 128, d_tips);
#endif
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float t; CK(hipEventElapsedTime(&t, a, b));
    if (r >= 2) ms.push_back(t);
  }
  std::vector<double> tips((size_t)n * 3);
  CK(hipMemcpy(tips.data(), d_tips, tips.size() * 8, hipMemcpyDeviceToHost));
  double sum = 0, asum = 0; for (double v : tips) { sum += v; asum += std::fabs(v); }
  std::sort(ms.begin(), ms.end());
  hipFuncAttributes fa;
#ifdef KB_VERDICT
  CK(hipFuncGetAttributes(&fa, (const void *)trk::fk_verdict<N, false, false, kSig>));
  {
    std::vector<uint64_t> bits((size_t)((n + 63) / 64));
    CK(hipMemcpy(bits.data(), d_bits, bits.size() * 8, hipMemcpyDeviceToHost));
    uint32_t fbc = 0; CK(hipMemcpy(&fbc, d_fbc, 4, hipMemcpyDeviceToHost));
    long valid = 0; for (uint64_t w : bits) valid += __builtin_popcountll(w);
    std::printf("fk_verdict: valid %ld of %ld, fallback entries %u (summed over launches)\n", valid, (long)n, fbc);
  }
#else
  CK(hipFuncGetAttributes(&fa, (const void *)trk::rk4_only<N>));
#endif
  std::printf("N=%d n=%ld (2^%d unless KB_COUNT) waves/SIMD<=%d: best %.3f ms  median %.3f ms  -> %.4g FK/s   regs %d  scratch %zu B   tips sum %.17g abs %.17g\n", N, (long)n, KB_LOG2, KB_WAVES,
              ms.front(), ms[ms.size() / 2], (double)n / (ms.front() * 1e-3), fa.numRegs, (size_t)fa.localSizeBytes, sum, asum);
  return 0;
}
