// fk_inst.hip -- one K1 instantiation set per object file: compiled once per tendon count and
// kernel with -DTRK_INST_N=<1..8> -DTRK_INST_KIND=<0 uniform | 1 retract | 2 fused with K2 | 3 verdict-only | 4 verdict-only, retraction | 5 edge queue>
// (see _lib.py: build()).
#include "fk_launch.hpp"
#include "fk_kernel.hpp"
#if TRK_INST_KIND == 1
#include "fk_retract_kernel.hpp"
#elif TRK_INST_KIND == 2
#include "fused_kernel.hpp"
#elif TRK_INST_KIND == 3
#include "verdict_kernel.hpp"
#elif TRK_INST_KIND == 4
#define TRK_WITH_RETRACT_VERDICT
#include "fk_retract_kernel.hpp"
#include "verdict_kernel.hpp"
#elif TRK_INST_KIND == 5
#include "edge_queue_kernel.hpp"
#endif

namespace trk {

#if TRK_INST_KIND == 5
template <> void launch_fk_edge_queue<TRK_INST_N>(const FkLaunch &a, const VerdictArgs *va, size_t lds, const EdgeQueueArgs *qa,
                                                  const FusedSweepArgs *sweep, unsigned waves) {
  if (a.rotation)
    hipLaunchKernelGGL((fk_edge_queue<TRK_INST_N, true>), dim3(waves), dim3(64), lds, a.stream, a.K, a.d_tab, a.d_steps, a.n_steps, va, qa, sweep);
  else
    hipLaunchKernelGGL((fk_edge_queue<TRK_INST_N, false>), dim3(waves), dim3(64), lds, a.stream, a.K, a.d_tab, a.d_steps, a.n_steps, va, qa, sweep);
}
template <> int fk_edge_queue_waves_per_cu<TRK_INST_N>(bool rotation, size_t lds) {
  int nb = 0;
  const hipError_t e = rotation ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fk_edge_queue<TRK_INST_N, true>, 64, lds)
                                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fk_edge_queue<TRK_INST_N, false>, 64, lds);
  return e == hipSuccess ? nb : 0;
}
#endif

#if TRK_INST_KIND == 5
#elif TRK_INST_KIND == 4
template <bool ROT, bool SPH, bool SIG>
static void go(const FkLaunch &a, const VerdictArgs *va, size_t lds) {
  const unsigned grid = (unsigned)((a.n + 63) / 64);
  const RetractHandoff ho{a.d_handoff, a.handoff_ld};
  hipLaunchKernelGGL((fk_retract_prologue<TRK_INST_N, ROT>), dim3(grid), dim3(64), 0, a.stream, a.d_states, a.n, a.K, a.d_poly, a.d_tab, a.d_steps,
                     a.n_steps, a.k_first, a.d_tgrid, a.d_hl, a.d_perm, ho);
  hipLaunchKernelGGL((fk_verdict_retract<TRK_INST_N, ROT, SPH, SIG>), dim3(grid), dim3(64), lds, a.stream, a.d_states, a.n, a.K, a.d_poly,
                     a.d_tab, a.d_steps, a.n_steps, a.k_first, a.d_tgrid, a.d_hl, a.out.tips, va, ho);
}
template <bool SPH, bool SIG>
static void go_rot(const FkLaunch &a, const VerdictArgs *va, size_t lds) {
  if (a.rotation) go<true, SPH, SIG>(a, va, lds); else go<false, SPH, SIG>(a, va, lds);
}
template <> void launch_fk_verdict_retract<TRK_INST_N>(const FkLaunch &a, const VerdictArgs *va, size_t lds, bool spheres, bool with_sig) {
  if (spheres) { if (with_sig) go_rot<true, true>(a, va, lds); else go_rot<true, false>(a, va, lds); }
  else         { if (with_sig) go_rot<false, true>(a, va, lds); else go_rot<false, false>(a, va, lds); }
}
template <bool ROT>
static void go_list(const FkLaunch &a, const FusedSweepArgs *sweep, size_t lds, const int32_t *list, const uint32_t *count) {
  const unsigned grid = (unsigned)((a.n + 63) / 64);                       // a.n = columns of the fallback workspace (multiple of 64)
  hipLaunchKernelGGL((fk_sweep_retract_list<TRK_INST_N, ROT>), dim3(grid), dim3(64), lds, a.stream, a.d_states, a.n, a.ld, a.K,
                     a.d_poly, a.d_tab, a.d_steps, a.n_steps, a.k_first, a.d_tgrid, a.d_hl, a.out, sweep, list, count);
}
template <> void launch_fk_sweep_retract_list<TRK_INST_N>(const FkLaunch &a, const FusedSweepArgs *sweep, size_t lds, const int32_t *list,
                                                          const uint32_t *count) {
  if (a.rotation) go_list<true>(a, sweep, lds, list, count); else go_list<false>(a, sweep, lds, list, count);
}
#elif TRK_INST_KIND == 3
template <bool ROT, bool SPH, bool SIG>
static void go(const FkLaunch &a, const VerdictArgs *va, size_t lds) {
  const unsigned grid = (unsigned)((a.n + 63) / 64);
  hipLaunchKernelGGL((fk_verdict<TRK_INST_N, ROT, SPH, SIG>), dim3(grid), dim3(64), lds, a.stream, a.d_states, a.n, a.K, a.d_tab, a.d_steps,
                     a.n_steps, a.out.tips, va);
}
template <bool SPH, bool SIG>
static void go_rot(const FkLaunch &a, const VerdictArgs *va, size_t lds) {
  if (a.rotation) go<true, SPH, SIG>(a, va, lds); else go<false, SPH, SIG>(a, va, lds);
}
template <> void launch_fk_verdict<TRK_INST_N>(const FkLaunch &a, const VerdictArgs *va, size_t lds, bool spheres, bool with_sig) {
  if (spheres) { if (with_sig) go_rot<true, true>(a, va, lds); else go_rot<true, false>(a, va, lds); }
  else         { if (with_sig) go_rot<false, true>(a, va, lds); else go_rot<false, false>(a, va, lds); }
}
#elif TRK_INST_KIND == 2
template <bool ROT>
static void go_list(const FkLaunch &a, const FusedSweepArgs *sweep, size_t lds, const int32_t *list, const uint32_t *count) {
  const unsigned grid = (unsigned)((a.n + 63) / 64);                       // a.n = columns of the fallback workspace (multiple of 64)
  hipLaunchKernelGGL((fk_sweep_fused_list<TRK_INST_N, ROT>), dim3(grid), dim3(64), lds, a.stream, a.d_states, a.n, a.ld, a.K,
                     a.d_tab, a.d_steps, a.n_steps, a.out, sweep, list, count);
}
template <> void launch_fk_sweep_fused_list<TRK_INST_N>(const FkLaunch &a, const FusedSweepArgs *sweep, size_t lds, const int32_t *list,
                                                        const uint32_t *count) {
  if (a.rotation) go_list<true>(a, sweep, lds, list, count); else go_list<false>(a, sweep, lds, list, count);
}
template <bool ROT>
static void go(const FkLaunch &a, const FusedSweepArgs *sweep, size_t lds) {
  const unsigned grid = (unsigned)((a.n + 63) / 64);
  hipLaunchKernelGGL((fk_sweep_fused<TRK_INST_N, ROT>), dim3(grid), dim3(64), lds, a.stream, a.d_states, a.n, a.ld, a.K,
                     a.d_tab, a.d_steps, a.n_steps, a.out, sweep);
}
template <> void launch_fk_sweep_fused<TRK_INST_N>(const FkLaunch &a, const FusedSweepArgs *sweep, size_t lds) {
  if (a.rotation) go<true>(a, sweep, lds); else go<false>(a, sweep, lds);
}
#else
#if TRK_INST_KIND == 1
template <bool ROT, bool WR>
static void go(const FkLaunch &a) {
  const unsigned grid = (unsigned)((a.n + 63) / 64);
  hipLaunchKernelGGL((fk_rk4_batch_retract<TRK_INST_N, ROT, WR>), dim3(grid), dim3(64), 0, a.stream, a.d_states, a.n, a.ld, a.K,
                     a.d_poly, a.d_tab, a.d_steps, a.n_steps, a.k_first, a.d_tgrid, a.d_hl, a.out, a.d_perm, a.d_wave_k_begin);
}
template <> void launch_fk_retract<TRK_INST_N>(const FkLaunch &a) {
#else
template <bool ROT, bool WR>
static void go(const FkLaunch &a) {
  const unsigned grid = (unsigned)((a.n + 63) / 64);
  hipLaunchKernelGGL((fk_rk4_batch_uniform<TRK_INST_N, ROT, WR>), dim3(grid), dim3(64), 0, a.stream, a.d_states, a.n, a.ld, a.K,
                     a.d_tab, a.d_steps, a.n_steps, a.out);
}
template <> void launch_fk_uniform<TRK_INST_N>(const FkLaunch &a) {
#endif
  if (a.rotation) { if (a.write_R) go<true, true>(a); else go<true, false>(a); }
  else            { if (a.write_R) go<false, true>(a); else go<false, false>(a); }
}
#endif

}  // namespace trk
