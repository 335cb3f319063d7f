// latency_floor.hip -- can a single-state isValid (AbstractValidityChecker.cpp:124-133, one state per call from a serial OMPL
// planner) be served faster than the 0.15 ms one host core takes?  Measures the ingredients of the answer on the GPU box:
//   1. the host round trip every single-state call pays whatever the kernel does: 32 B up, an EMPTY kernel, 32 B down, stream
//      synchronise -- with pageable and with pinned host buffers;
//   2. the latency of a DEPENDENT fp64 FMA, of a dependent reciprocal-square-root + Newton step, and of a dependent cross-lane
//      add (what a wave-per-configuration kernel would chain along the critical path of the right-hand side), in clock cycles
//      (s_memtime) and ns;
//   3. one wave of the CURRENT kernel's RK4 loop: profiles/kbench.hip with -DKB_LOG2=6 gives that.
//
//   hipcc --offload-arch=gfx950 -O3 profiles/latency_floor.hip -o /tmp/lf && /tmp/lf
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void empty_kernel(const double *in, double *out) { if (threadIdx.x == 0) out[0] = in[0]; }

__global__ void chain_fma(double *io, int n, unsigned long long *cycles) {
  double x = io[threadIdx.x], a = 1.0000001, b = 1e-9;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
  for (int i = 0; i < n; i++) x = __builtin_fma(x, a, b);
  const unsigned long long t1 = __builtin_readcyclecounter();
  io[threadIdx.x] = x;
  if (threadIdx.x == 0) *cycles = t1 - t0;
}

__global__ void chain_rsq(double *io, int n, unsigned long long *cycles) {
  double x = io[threadIdx.x] + 2.0;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 4
  for (int i = 0; i < n; i++) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    const double t = __builtin_fma(-hx * y, y, 0.5);
    y = __builtin_fma(y, t, y);
    x = y + 2.0;                               // the next seed depends on this result
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  io[threadIdx.x] = x;
  if (threadIdx.x == 0) *cycles = t1 - t0;
}

__global__ void chain_xlane(double *io, int n, unsigned long long *cycles) {
  double x = io[threadIdx.x];
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
  for (int i = 0; i < n; i++) x = x * 0.5 + __shfl_xor(x, 1 + (i & 3), 64);     // a dependent cross-lane exchange + add
  const unsigned long long t1 = __builtin_readcyclecounter();
  io[threadIdx.x] = x;
  if (threadIdx.x == 0) *cycles = t1 - t0;
}

template <class F> static double median_us(int reps, F &&f) {
  std::vector<double> t;
  for (int r = 0; r < reps; r++) {
    const auto a = std::chrono::steady_clock::now();
    f();
    const auto b = std::chrono::steady_clock::now();
    t.push_back(std::chrono::duration<double, std::micro>(b - a).count());
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main() {
  double *d_in, *d_out, *d_io; unsigned long long *d_cyc;
  CK(hipMalloc(&d_in, 64)); CK(hipMalloc(&d_out, 64)); CK(hipMalloc(&d_io, 64 * 8)); CK(hipMalloc(&d_cyc, 8));
  CK(hipMemset(d_io, 0, 64 * 8));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  double pageable_in[4] = {1, 2, 3, 4}, pageable_out[4];
  double *pin_in, *pin_out;
  CK(hipHostMalloc((void **)&pin_in, 64, hipHostMallocDefault)); CK(hipHostMalloc((void **)&pin_out, 64, hipHostMallocDefault));
  pin_in[0] = 1.0;
  auto trip = [&](double *hin, double *hout) {
    (void)hipMemcpyAsync(d_in, hin, 32, hipMemcpyHostToDevice, s);
    hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, d_in, d_out);
    (void)hipMemcpyAsync(hout, d_out, 32, hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
  };
  for (int i = 0; i < 50; i++) trip(pin_in, pin_out);
  std::printf("host round trip (32 B up, empty kernel, 32 B down, synchronise), median of 400: pinned %.1f us, pageable %.1f us\n",
              median_us(400, [&] { trip(pin_in, pin_out); }), median_us(400, [&] { trip(pageable_in, pageable_out); }));
  std::printf("launch + synchronise alone: %.1f us\n", median_us(400, [&] {
                hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, d_in, d_out); (void)hipStreamSynchronize(s); }));
  const int n = 200000;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto chain = [&](const char *name, void (*k)(double *, int, unsigned long long *), int ops_per_iter) -> int {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d_io, n, d_cyc);           // warm
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d_io, n, d_cyc);
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    unsigned long long cyc = 0; CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
    std::printf("%-34s %7.2f ns per link (%d dependent op(s) each), %6.1f counter ticks per link\n", name, 1e6 * ms / n, ops_per_iter, (double)cyc / n);
    return 0;
  };
  if (chain("dependent v_fma_f64", chain_fma, 1)) return 1;
  if (chain("dependent rsq + Newton step + add", chain_rsq, 5)) return 1;
  if (chain("dependent cross-lane shuffle + fma", chain_xlane, 2)) return 1;
  return 0;
}
