// fused_kernel.hpp -- K1 + K2 in one launch for the verdict path (tr_validate_batch*): every wave
// integrates its 64 configurations (fk_uniform_body), then sweeps the backbones it has just stored
// (sweep_body) -- same code, same arithmetic, same memory layout as the two separate kernels: the
// verdicts and flags are the same; the points agree to rounding (hipcc pairs the multiply-adds of a sum
// of products per kernel: one ulp in about a hundredth of the backbones).  What changes is the schedule: a SIMD holds two waves, and
// while one is in its sweep (latency-bound: loads, cell tests) the other is in its RK4 steps
// (fp64-VALU-bound), so the sweep runs in issue slots and memory latency the FK leaves idle, and
// there is no second launch with its own ramp and tail.  The points are read back by the lane that
// wrote them (program order within a thread; the workgroup barrier keeps the compiler from moving
// the loads), mostly from L2 / Infinity Cache.  (Starting the odd wave slot of every SIMD late, so that
// neighbours run unlike phases from the first round on, was measured and changes nothing: the phases
// drift apart on their own.)
#pragma once
#include "fk_kernel.hpp"
#define TRK_DEVICE_BODIES_ONLY
#ifndef TRK_K2_PF
#define TRK_K2_PF 8          // behind the FK the sweep has registers to spare: 8 points in flight per lane (4 standalone); 12 measured worse
#endif
#include "sweep_kernel.hpp"

namespace trk {

// Waves per SIMD of the stored-point fused kernels, measured per width like the verdict-only ones (profiles/r02/verdict_widths_v1.txt,
// second table; ms per 2^18 configurations, two waves | one): N=4 3.19 | 3.19, N=5 4.25 | 4.52, N=6 5.29 | 4.91.
#ifndef TRK_FUSED_TWO_WAVE_MAXN
#define TRK_FUSED_TWO_WAVE_MAXN 5
#endif

template <int N, bool ROT>
__global__ __launch_bounds__(64, (N <= TRK_FUSED_TWO_WAVE_MAXN ? 2 : 1)) void fk_sweep_fused(
    const double *__restrict__ states, int64_t n, int64_t ld, RobotK K, const double *__restrict__ tab,
    const StepK *__restrict__ steps, int nsteps, FkOut out, const FusedSweepArgs *__restrict__ sa) {
  // (the verdict paths never ask for the backbone length)  Edge samples also get their cell signatures, point by point
  SignatureHook hook{sa, n, sa->sig != nullptr, {}};
  hook.st.init();
  fk_uniform_body<N, ROT, false, false>(states, n, ld, K, tab, steps, nsteps, out, hook);
  hook.finish();
  __syncthreads();
  const FusedSweepArgs a = *sa;
  sweep_body<false>(a.in, n, ld, a.P, a.CH, a.NM, K, a.g, a.grid, a.near_grid, a.check_voxels, a.debug, a.valid_bits, a.flags);
}

// The same on a compacted list of configurations: this is the fallback pass of the verdict-only kernel
// (verdict_kernel.hpp) for the few configurations whose self-collision test needs the exact pairwise sweep, i.e. all of
// their backbone points at once: they are integrated again, this time storing the points in a small workspace of `cap`
// columns.  Block b owns columns [64 b, 64 b + 64) and walks the list in strides of `cap`: in round r its lane i takes
// configuration list[r cap + 64 b + i] (while that index is below *count) and ORs its verdict bit into the mask.  A block
// leaves as soon as its slice of the list is exhausted, so with an empty list the launch costs nothing.
template <int N, bool ROT>
__global__ __launch_bounds__(64, (N <= TRK_FUSED_TWO_WAVE_MAXN ? 2 : 1)) void fk_sweep_fused_list(
    const double *__restrict__ states, int64_t cap, int64_t ld, RobotK K, const double *__restrict__ tab,
    const StepK *__restrict__ steps, int nsteps, FkOut out, const FusedSweepArgs *__restrict__ sa,
    const int32_t *__restrict__ list, const uint32_t *__restrict__ count) {
  const int64_t total = (int64_t)*count;
  for (int64_t offset = 0; offset < total; offset += cap) {
    const int64_t left = total - offset;
    const int64_t m = left < cap ? left : cap;
    if ((int64_t)blockIdx.x * 64 >= m) break;                     // wave-uniform
    fk_uniform_body<N, ROT, false, false>(states, m, ld, K, tab, steps, nsteps, out, NoPointHook(), list + offset);
    __syncthreads();
    const FusedSweepArgs a = *sa;
    sweep_body<false>(a.in, m, ld, a.P, a.CH, a.NM, K, a.g, a.grid, a.near_grid, a.check_voxels, a.debug, a.valid_bits, a.flags, list + offset);
    __syncthreads();
  }
}

}  // namespace trk
