// mpc_planner_amd/csrc/tmpc_solve.hip -- the translation units that INSTANTIATE the solve kernels of libtmpc_hip.so (templates: tmpc_kernels.hpp,
// lists: tmpc_instances.hpp).  The same file is compiled several times, each time with one -D switch naming the group it instantiates, so that
// the groups build in parallel (__graft_entry__.build()):
//   -DTMPC_TU_FAST      fast kernels, production (MPCC + the other stage models) and the generic kernel
//   -DTMPC_TU_COMPACT   compact one-wave kernels and the parallel-in-time (latency mode 2) kernels
//   -DTMPC_TU_PROF      profiled twins of the fast kernels (tmpc_debug_profile) + the profiled latency-mode-2 kernel
//   -DTMPC_TU_CP2       compact two-wave kernels
//   -DTMPC_TU_SQRT      the square-root-Riccati instantiations (tmpc_dims.riccati_form = 1) + the Gaussian-row latency (mode 2) and one-wave compact kernels
//   -DTMPC_TU_QUAD      the four-wave tick kernels (latency mode 3)
//   -DTMPC_TU_QUADW     ... for 21 <= N <= 31
// The C-ABI -- dispatch tables, handle, entry points -- is tmpc_capi.hip; it declares every instantiation `extern`.
// Experiment builds (tools/kernel_probe.sh): -DTMPC_SINGLE_KERNEL=<fast template arguments> / -DTMPC_SINGLE_COMPACT=<compact template arguments>
// compile ONE instantiation and nothing else -- seconds instead of minutes when looking at one kernel's registers / ISA.
#include "tmpc_kernels.hpp"
#include "tmpc_instances.hpp"

#if defined(TMPC_SINGLE_COMPACT)
template __global__ void tmpc::tmpc_solve_compact_kernel<TMPC_SINGLE_COMPACT>(TMPC_KARGS);
#elif defined(TMPC_SINGLE_KERNEL)
template __global__ void tmpc::tmpc_solve_fast_kernel<TMPC_SINGLE_KERNEL>(TMPC_KARGS);
#elif defined(TMPC_GENERATED_STAGE)
#error "generated solvers are ONE translation unit: compile tmpc_capi.hip with -DTMPC_GENERATED_STAGE (it instantiates what it dispatches)"
#elif defined(TMPC_TU_FAST)
TMPC_FAST_SHAPES(TMPC_I_FAST_DEF)
TMPC_FAST_CM_SHAPES(TMPC_I_FASTCM_DEF)
TMPC_GENERIC_MODELS(TMPC_I_GEN_DEF)
#elif defined(TMPC_TU_COMPACT)
TMPC_COMPACT_SHAPES(TMPC_I_CP_DEF)
TMPC_SCAN_SHAPES(TMPC_I_SCAN_DEF)
#elif defined(TMPC_TU_PROF)
TMPC_FAST_SHAPES(TMPC_I_PROF_DEF)
template __global__ void tmpc::tmpc_solve_fast_kernel<8, 8, 6, 128, true, tmpc::ScanSolo>(TMPC_KARGS);      // profiled twin of latency mode 2 (cfg 2)
#elif defined(TMPC_TU_CP2)
TMPC_CP2_SHAPES(TMPC_I_CP2_DEF)
#elif defined(TMPC_TU_SQRT)
TMPC_SQRT_SHAPES(TMPC_I_SQRT_DEF)
TMPC_SCAN_G_SHAPES(TMPC_I_SCANG_DEF)
TMPC_COMPACT_G_SHAPES(TMPC_I_CPG_DEF)
#elif defined(TMPC_TU_QUAD)
TMPC_QUAD_SHAPES(TMPC_I_QUAD_DEF)
TMPC_QUAD_G_SHAPES(TMPC_I_QUADG_DEF)
#elif defined(TMPC_TU_QUADW)
TMPC_QUAD_W_SHAPES(TMPC_I_QUADW_DEF)
#else
#error "tmpc_solve.hip: name the group to instantiate (-DTMPC_TU_FAST / _COMPACT / _PROF / _CP2 / _SQRT / _QUAD / _QUADW) or one kernel (-DTMPC_SINGLE_KERNEL= / -DTMPC_SINGLE_COMPACT=)"
#endif
