// fk_launch.hpp -- host-side launch interface of K1.  Each (tendon count, kernel) pair is its own
// translation unit (fk_inst.hip compiled with -DTRK_INST_N=.. -DTRK_INST_KIND=..) so the
// instantiations compile in parallel; tendon_hip.hip only sees these declarations.
#pragma once
#include <hip/hip_runtime.h>
#include "tr_types.hpp"

namespace trk {

struct FkLaunch {
  const double *d_states; int64_t n, ld;
  RobotK K;
  bool rotation, write_R;
  // shared arc-length grid (retraction disabled)
  const double *d_tab; const StepK *d_steps; int n_steps;
  // per-lane grid (retraction enabled)
  const PolyK *d_poly; int k_first; const double *d_tgrid, *d_hl;
  FkOut out;
  hipStream_t stream;
  // retraction robots' verdict path: the prologue kernel's hand-over planes ([19 + N][handoff_ld] doubles) and the batch order
  double *d_handoff = nullptr; int64_t handoff_ld = 0; const int32_t *d_perm = nullptr;
  const int32_t *d_wave_k_begin = nullptr;      // stored-point retraction kernel: per wave of the order d_perm, the first step of its loop
};

struct FusedSweepArgs;
struct VerdictArgs;
struct EdgeQueueArgs;
template <int N> void launch_fk_uniform(const FkLaunch &a);
// K1 + K2 fused (fused_kernel.hpp): `sweep` is a device pointer, `lds` the sweep's dynamic LDS bytes
template <int N> void launch_fk_sweep_fused(const FkLaunch &a, const FusedSweepArgs *sweep, size_t lds);
// the same on a compacted list of *count configurations, in rounds over an a.n-column workspace (fused_kernel.hpp)
template <int N> void launch_fk_sweep_fused_list(const FkLaunch &a, const FusedSweepArgs *sweep, size_t lds, const int32_t *list,
                                                 const uint32_t *count);
// verdict-only kernel (verdict_kernel.hpp): no point storage; `va` is a device pointer, `lds` its dynamic LDS bytes
// (with_sig: the launch writes cell signatures, va->sig != null; lds then includes the signature tile)
template <int N> void launch_fk_verdict(const FkLaunch &a, const VerdictArgs *va, size_t lds, bool spheres, bool with_sig);
template <int N> void launch_fk_retract(const FkLaunch &a);
// the verdict-only kernel and its fallback pass for retraction-enabled robots (verdict_kernel.hpp, TRK_WITH_RETRACT_VERDICT)
template <int N> void launch_fk_verdict_retract(const FkLaunch &a, const VerdictArgs *va, size_t lds, bool spheres, bool with_sig);
template <int N> void launch_fk_sweep_retract_list(const FkLaunch &a, const FusedSweepArgs *sweep, size_t lds, const int32_t *list,
                                                   const uint32_t *count);

// the edge bisection as one persistent launch over a device work queue (edge_queue_kernel.hpp; backbone checker, shared arc-length
// grid): `waves` workgroups of one wave; va / qa / sweep are device pointers, `lds` the verdict body's image with the signature tile
template <int N> void launch_fk_edge_queue(const FkLaunch &a, const VerdictArgs *va, size_t lds, const EdgeQueueArgs *qa,
                                           const FusedSweepArgs *sweep, unsigned waves);
// workgroups of that kernel one CU holds (hipOccupancyMaxActiveBlocksPerMultiprocessor; 0 on error)
template <int N> int fk_edge_queue_waves_per_cu(bool rotation, size_t lds);

#define TRK_DECL_FK(N) template <> void launch_fk_uniform<N>(const FkLaunch &); template <> void launch_fk_retract<N>(const FkLaunch &); \
  template <> void launch_fk_sweep_fused<N>(const FkLaunch &, const FusedSweepArgs *, size_t); \
  template <> void launch_fk_sweep_fused_list<N>(const FkLaunch &, const FusedSweepArgs *, size_t, const int32_t *, const uint32_t *); \
  template <> void launch_fk_verdict<N>(const FkLaunch &, const VerdictArgs *, size_t, bool, bool); \
  template <> void launch_fk_verdict_retract<N>(const FkLaunch &, const VerdictArgs *, size_t, bool, bool); \
  template <> void launch_fk_sweep_retract_list<N>(const FkLaunch &, const FusedSweepArgs *, size_t, const int32_t *, const uint32_t *); \
  template <> void launch_fk_edge_queue<N>(const FkLaunch &, const VerdictArgs *, size_t, const EdgeQueueArgs *, const FusedSweepArgs *, unsigned); \
  template <> int fk_edge_queue_waves_per_cu<N>(bool, size_t);
TRK_DECL_FK(1) TRK_DECL_FK(2) TRK_DECL_FK(3) TRK_DECL_FK(4) TRK_DECL_FK(5) TRK_DECL_FK(6) TRK_DECL_FK(7) TRK_DECL_FK(8)
#undef TRK_DECL_FK

}  // namespace trk
