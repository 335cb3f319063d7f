// fused_kernel.hpp -- K1 + K2 in one launch for the verdict path (tr_validate_batch*): every wave
// integrates its 64 configurations (fk_uniform_body), then sweeps the backbones it has just stored
// (sweep_body) -- same code, same arithmetic, same memory layout as the two separate kernels, so
// results are identical bit for bit.  What changes is the schedule: a SIMD holds two waves, and
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

template <int N, bool ROT>
__global__ __launch_bounds__(64, (N <= TRK_K1_TWO_WAVE_MAXN ? 2 : 1)) void fk_sweep_fused(
    const double *__restrict__ states, int64_t n, int64_t ld, RobotK K, const double *__restrict__ tab,
    const StepK *__restrict__ steps, int nsteps, FkOut out, const FusedSweepArgs *__restrict__ sa) {
  fk_uniform_body<N, ROT, false, false>(states, n, ld, K, tab, steps, nsteps, out);   // the verdict paths never ask for the backbone length
  __syncthreads();
  const FusedSweepArgs a = *sa;
  sweep_body<false>(a.in, n, ld, a.P, a.CH, a.NM, K, a.g, a.grid, a.near_grid, a.check_voxels, a.debug, a.valid_bits, a.flags);
}

}  // namespace trk
