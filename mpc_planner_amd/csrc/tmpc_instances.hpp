// mpc_planner_amd/csrc/tmpc_instances.hpp -- which instantiations of the solve-kernel templates (tmpc_kernels.hpp) libtmpc_hip.so contains.
// ONE list per group; tmpc_solve.hip instantiates a group in the translation unit its -D switch names, tmpc_capi.hip declares all of them
// `extern` and takes their addresses in the dispatch tables (pick_*_kernel).  A shape a dispatch table names but no list carries would be
// instantiated implicitly inside tmpc_capi.hip -- correct, but the build guard in __graft_entry__.build() refuses it (the C-ABI unit must
// stay free of solve kernels), so the lists and the tables cannot drift apart unnoticed.
#pragma once
#define TMPC_KARGS tmpc::Dims, int, const double *, const double *, const double *, double *, double *, double *, int *, int *, int *, double *, int *, long long *, tmpc::StateIO
// fast kernels, MPCC cost + ellipsoid rows (NLIN, MM, LPS, NTH): production (PROF = false) and profiled twins (PROF = true, tmpc_debug_profile)
#define TMPC_FAST_SHAPES(X) X(8, 8, 4, 128) X(12, 12, 4, 128) X(20, 8, 4, 128) X(-1, 6, 4, 128) X(-1, 9, 4, 128) X(-1, 12, 4, 128) X(0, 4, 3, 64) \
    X(8, 8, 3, 64) X(12, 12, 3, 64) X(24, 0, 3, 64) X(-1, 7, 3, 64) X(-1, 10, 3, 64) X(-1, 13, 3, 64) X(-1, 9, 6, 128) X(0, 4, 2, 64) X(8, 8, 6, 128)
// fast kernels with another stage model (NLIN, MM, LPS, NTH, CM): CM = 1 curvature-aware cost, CM = 2 Gaussian rows
#define TMPC_FAST_CM_SHAPES(X) X(20, 8, 4, 128, 1) X(-1, 13, 3, 64, 1) X(5, 5, 4, 128, 2) X(5, 5, 2, 64, 2) X(8, 8, 2, 64, 0) X(-1, 6, 4, 128, 2) X(-1, 12, 4, 128, 2) X(-1, 13, 3, 64, 2)
// latency mode 2: the parallel-in-time Newton solve (NLIN, MM, LPS, NTH, policy)
#define TMPC_SCAN_SHAPES(X) X(-1, 12, 4, 128, tmpc::ScanSoloT<2>) X(8, 8, 6, 128, tmpc::ScanSolo) X(8, 8, 3, 64, tmpc::ScanSolo) X(-1, 9, 6, 128, tmpc::ScanSolo)
// square-root form of the Riccati recursion (tmpc_dims.riccati_form = 1): run-time-shape fast kernels (NLIN, MM, LPS, NTH, CM)
#define TMPC_SQRT_SHAPES(X) X(-1, 13, 3, 64, 0) X(-1, 12, 4, 128, 0) X(20, 8, 4, 128, 1)
// latency mode 3 (round 6): four waves per trajectory, twelve lanes per stage, the wide phases of the parallel-in-time factorisation on all four waves
// (NLIN, MM, PROF, policy): the tuned cfg 2 shape + its profiled twin + its A/B twin whose factorisation stays on one wave; run-time shapes up to 48 rows
#define TMPC_QUAD_SHAPES(X) X(8, 8, false, tmpc::ScanQuad) X(8, 8, true, tmpc::ScanQuad) X(8, 8, false, tmpc::ScanSolo) X(-1, 4, false, tmpc::ScanQuad)
// Gaussian chance-constraint rows (CM = 2: mpc_planner_jackal's shipped default stack) on the latency variants and the one-wave compact kernel (round 6):
// latency mode 2 (NLIN, MM, LPS, NTH, policy), latency mode 3 (NLIN, MM, policy), compact one-wave (NLIN, MM, LPS) -- run-time shapes
#define TMPC_SCAN_G_SHAPES(X) X(-1, 12, 4, 128, tmpc::ScanSoloT<2>) X(-1, 9, 6, 128, tmpc::ScanSolo)
#define TMPC_QUAD_G_SHAPES(X) X(-1, 4, tmpc::ScanQuad)
// latency mode 3 for 21 <= N <= 31 (the shipped jackal / jackalsimulator horizon N = 30): eight lanes per stage, run-time row mix up to 34 rows (MM, CM)
#define TMPC_QUAD_W_SHAPES(X) X(6, 0) X(6, 1) X(6, 2) X(6, 3)
#define TMPC_COMPACT_G_SHAPES(X) X(-1, 10, 3) X(5, 5, 2)
// compact kernels, one wave per trajectory (NLIN, MM, LPS)
#define TMPC_COMPACT_SHAPES(X) X(8, 8, 3) X(0, 4, 3) X(12, 12, 3) X(24, 0, 3) X(-1, 7, 3) X(-1, 10, 3) X(8, 8, 2)
// compact kernels, two waves per trajectory (NLIN, MM, LPS, CM)
#define TMPC_CP2_SHAPES(X) X(20, 8, 4, 0) X(20, 8, 4, 1) X(12, 12, 4, 0) X(8, 8, 4, 0) X(-1, 6, 4, 0) X(-1, 9, 4, 0) X(-1, 6, 4, 2) X(5, 5, 4, 2)
// generic kernel (CM)
#define TMPC_GENERIC_MODELS(X) X(0) X(1) X(2) X(3)

#define TMPC_ALL_INSTANCES(KW)                                                                                                                     \
    TMPC_FAST_SHAPES(TMPC_I_FAST_##KW) TMPC_FAST_SHAPES(TMPC_I_PROF_##KW) TMPC_FAST_CM_SHAPES(TMPC_I_FASTCM_##KW) TMPC_SCAN_SHAPES(TMPC_I_SCAN_##KW) \
    TMPC_COMPACT_SHAPES(TMPC_I_CP_##KW) TMPC_CP2_SHAPES(TMPC_I_CP2_##KW) TMPC_GENERIC_MODELS(TMPC_I_GEN_##KW) TMPC_SQRT_SHAPES(TMPC_I_SQRT_##KW) TMPC_QUAD_SHAPES(TMPC_I_QUAD_##KW) \
    TMPC_SCAN_G_SHAPES(TMPC_I_SCANG_##KW) TMPC_QUAD_G_SHAPES(TMPC_I_QUADG_##KW) TMPC_QUAD_W_SHAPES(TMPC_I_QUADW_##KW) TMPC_COMPACT_G_SHAPES(TMPC_I_CPG_##KW)
// KW = DEF: explicit instantiation definition; KW = EXT: extern declaration
#define TMPC_I_FAST_DEF(a, b, c, e) template __global__ void tmpc::tmpc_solve_fast_kernel<a, b, c, e, false>(TMPC_KARGS);
#define TMPC_I_FAST_EXT(a, b, c, e) extern template __global__ void tmpc::tmpc_solve_fast_kernel<a, b, c, e, false>(TMPC_KARGS);
#define TMPC_I_PROF_DEF(a, b, c, e) template __global__ void tmpc::tmpc_solve_fast_kernel<a, b, c, e, true>(TMPC_KARGS);
#define TMPC_I_PROF_EXT(a, b, c, e) extern template __global__ void tmpc::tmpc_solve_fast_kernel<a, b, c, e, true>(TMPC_KARGS);
#define TMPC_I_FASTCM_DEF(a, b, c, e, m) template __global__ void tmpc::tmpc_solve_fast_kernel<a, b, c, e, false, tmpc::Solo, m>(TMPC_KARGS);
#define TMPC_I_FASTCM_EXT(a, b, c, e, m) extern template __global__ void tmpc::tmpc_solve_fast_kernel<a, b, c, e, false, tmpc::Solo, m>(TMPC_KARGS);
#define TMPC_I_SCAN_DEF(a, b, c, e, t) template __global__ void tmpc::tmpc_solve_fast_kernel<a, b, c, e, false, t>(TMPC_KARGS);
#define TMPC_I_SCAN_EXT(a, b, c, e, t) extern template __global__ void tmpc::tmpc_solve_fast_kernel<a, b, c, e, false, t>(TMPC_KARGS);
#define TMPC_I_SQRT_DEF(a, b, c, e, m) template __global__ void tmpc::tmpc_solve_fast_kernel<a, b, c, e, false, tmpc::SoloSqrt, m>(TMPC_KARGS);
#define TMPC_I_SQRT_EXT(a, b, c, e, m) extern template __global__ void tmpc::tmpc_solve_fast_kernel<a, b, c, e, false, tmpc::SoloSqrt, m>(TMPC_KARGS);
#define TMPC_I_QUAD_DEF(a, b, pf, t) template __global__ void tmpc::tmpc_solve_fast_kernel<a, b, 12, 256, pf, t>(TMPC_KARGS);
#define TMPC_I_QUAD_EXT(a, b, pf, t) extern template __global__ void tmpc::tmpc_solve_fast_kernel<a, b, 12, 256, pf, t>(TMPC_KARGS);
#define TMPC_I_SCANG_DEF(a, b, c, e, t) template __global__ void tmpc::tmpc_solve_fast_kernel<a, b, c, e, false, t, 2>(TMPC_KARGS);
#define TMPC_I_SCANG_EXT(a, b, c, e, t) extern template __global__ void tmpc::tmpc_solve_fast_kernel<a, b, c, e, false, t, 2>(TMPC_KARGS);
#define TMPC_I_QUADG_DEF(a, b, t) template __global__ void tmpc::tmpc_solve_fast_kernel<a, b, 12, 256, false, t, 2>(TMPC_KARGS);
#define TMPC_I_QUADG_EXT(a, b, t) extern template __global__ void tmpc::tmpc_solve_fast_kernel<a, b, 12, 256, false, t, 2>(TMPC_KARGS);
#define TMPC_I_QUADW_DEF(b, m) template __global__ void tmpc::tmpc_solve_fast_kernel<-1, b, 8, 256, false, tmpc::ScanQuadT<2>, m>(TMPC_KARGS);
#define TMPC_I_QUADW_EXT(b, m) extern template __global__ void tmpc::tmpc_solve_fast_kernel<-1, b, 8, 256, false, tmpc::ScanQuadT<2>, m>(TMPC_KARGS);
#define TMPC_I_CPG_DEF(a, b, c) template __global__ void tmpc::tmpc_solve_compact_kernel<a, b, c, false, 64, 2>(TMPC_KARGS);
#define TMPC_I_CPG_EXT(a, b, c) extern template __global__ void tmpc::tmpc_solve_compact_kernel<a, b, c, false, 64, 2>(TMPC_KARGS);
#define TMPC_I_CP_DEF(a, b, c) template __global__ void tmpc::tmpc_solve_compact_kernel<a, b, c, false>(TMPC_KARGS);
#define TMPC_I_CP_EXT(a, b, c) extern template __global__ void tmpc::tmpc_solve_compact_kernel<a, b, c, false>(TMPC_KARGS);
#define TMPC_I_CP2_DEF(a, b, c, m) template __global__ void tmpc::tmpc_solve_compact_kernel<a, b, c, false, 128, m>(TMPC_KARGS);
#define TMPC_I_CP2_EXT(a, b, c, m) extern template __global__ void tmpc::tmpc_solve_compact_kernel<a, b, c, false, 128, m>(TMPC_KARGS);
#define TMPC_I_GEN_DEF(m) template __global__ void tmpc::tmpc_solve_kernel<m>(TMPC_KARGS);
#define TMPC_I_GEN_EXT(m) extern template __global__ void tmpc::tmpc_solve_kernel<m>(TMPC_KARGS);
