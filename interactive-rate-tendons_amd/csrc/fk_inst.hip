// fk_inst.hip -- one K1 instantiation set per object file: compiled once per tendon count and
// kernel with -DTRK_INST_N=<1..8> -DTRK_INST_RETRACT=<0|1> (see _lib.py: build()).
#include "fk_launch.hpp"
#include "fk_kernel.hpp"
#if TRK_INST_RETRACT
#include "fk_retract_kernel.hpp"
#endif

namespace trk {

#if TRK_INST_RETRACT
template <bool ROT, bool WR>
static void go(const FkLaunch &a) {
  const unsigned grid = (unsigned)((a.n + 63) / 64);
  hipLaunchKernelGGL((fk_rk4_batch_retract<TRK_INST_N, ROT, WR>), dim3(grid), dim3(64), 0, a.stream, a.d_states, a.n, a.ld, a.K,
                     a.d_poly, a.pscr, a.out);
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

}  // namespace trk
