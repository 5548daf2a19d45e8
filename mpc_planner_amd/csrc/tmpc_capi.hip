// mpc_planner_amd/csrc/tmpc_capi.hip -- the C-ABI translation unit of libtmpc_hip.so (include/tmpc_hip.h): kernel dispatch tables, the handle, every
// exported entry point, and the small kernels around the solve (tmpc_aux_kernels.hpp: selection, records, gather, f-1 / f-2 / f-3 helpers).
// The solve kernels themselves are templates (tmpc_kernels.hpp) instantiated in the units of tmpc_solve.hip; this unit declares them `extern`
// (tmpc_instances.hpp) and only takes their addresses.  A GENERATED solver (-DTMPC_GENERATED_STAGE, mpc_planner_amd/codegen/build.py) and the
// experiment builds that pass -DTMPC_SINGLE_TU compile this file ALONE: without the extern declarations every kernel the tables name is
// instantiated here.
#include <algorithm>
#include <cstring>
#include "tmpc_kernels.hpp"
#include "tmpc_instances.hpp"
#if !defined(TMPC_GENERATED_STAGE) && !defined(TMPC_SINGLE_TU)
TMPC_ALL_INSTANCES(EXT)
extern template __global__ void tmpc::tmpc_solve_fast_kernel<8, 8, 6, 128, true, tmpc::ScanSolo>(TMPC_KARGS);
#endif
#include "tmpc_aux_kernels.hpp"
// The lane-per-trajectory kernel family (tmpc_lanes.hip, tmpc_set_throughput_mode) is an OPTIONAL part of the library since round 5: it loses to
// the wave kernels on every shape measured and is kept for its persistent-state protocol on arbitrary shapes.  -DTMPC_WITH_LANES links
// it (__graft_entry__.build(with_lanes=True) / TMPC_BUILD_LANES=1); without it tmpc_set_throughput_mode reports that the build has no such kernels.
#ifdef TMPC_WITH_LANES
#include "tmpc_lanes_api.hpp"
#else
// Lab switches (kernel selection overrides for A/B measurements and for tests that have to reach one particular kernel family): environment variables
// TMPC_* read when a handle is created.  They exist ONLY in a library built with -DTMPC_LAB_SWITCHES (mpc_planner_amd/libtmpc_hip_lab.so, which
// __graft_entry__.build() links from the same kernel objects; tools/build_compact_variants.sh): the product library never reads the environment -- what a
// drop-in does depends on the C-ABI calls alone (round-5 verdict, next-8).  INTEGRATION.md section 7 lists them.
#ifdef TMPC_LAB_SWITCHES
static inline const char *lab_env(const char *name) { return getenv(name); }
#else
static inline const char *lab_env(const char *) { return nullptr; }
#endif
namespace tmpc {
namespace lanes {
struct Context;
static const char *const kAbsent = "this build of the library does not contain the lane-per-trajectory kernels (build with TMPC_BUILD_LANES=1 / build(with_lanes=True))";
static inline Context *create(const Dims &, int, std::string &err) { err = kAbsent; return nullptr; }
static inline void destroy(Context *) {}
static inline int stage_in(Context *, hipStream_t, int, const double *, const double *, const double *, bool, bool, std::string &err) { err = kAbsent; return 1; }
static inline int solve(Context *, hipStream_t, int, int, bool, bool, double *, double *, double *, int *, int *, int *, double *, int *, std::string &err) { err = kAbsent; return 1; }
static inline int reset_multipliers(Context *, hipStream_t, int, std::string &err) { err = kAbsent; return 1; }
static inline int clear_stopped(Context *, hipStream_t, int, std::string &err) { err = kAbsent; return 1; }
}  // namespace lanes
}  // namespace tmpc
#endif

// =================================================================================================
// C-ABI
// =================================================================================================
namespace tmpc {
typedef void (*SolveKernel)(Dims, int, const double *, const double *, const double *, double *, double *, double *, int *,
                            int *, int *, double *, int *, long long *, StateIO);
// Registered fast shapes (upper-bounded rows n_lin + n_slk, ellipsoids M) x lanes-per-stage; anything else runs the generic kernel.
// Only instantiations that compile WITHOUT scratch (zero VGPR spills) are registered: __graft_entry__.build() checks
// the compiler's resource remarks and fails otherwise.  Reason: with > ~100 spilled VGPRs this kernel was observed to
// return wrong iterates (spill/reload around partially-masked regions), see DESIGN.md section 5.  Shapes with more rows
// per lane ((8,8) at 2 lanes/stage for N > 21, (12,12) at 2 lanes/stage) therefore use the generic kernel for now.  The library is
// built with -mllvm -disable-machine-licm: hoisted constant materialisations were what pushed (12,12,3) into scratch.
// prof: the instrumented instantiation (tmpc_debug_profile) instead of the production one.
#define TMPC_FAST(...) (prof ? (SolveKernel)tmpc_solve_fast_kernel<__VA_ARGS__, true> : (SolveKernel)tmpc_solve_fast_kernel<__VA_ARGS__, false>)
static SolveKernel pick_fast_kernel(const Dims &d, int *threads, bool prof)
{
    *threads = NT;
    if (lab_env("TMPC_FORCE_GENERIC")) return nullptr;
    const int lps = (3 * d.N <= NT) ? 3 : ((2 * d.N <= NT) ? 2 : 0);
#ifndef TMPC_GENERATED_STAGE
    if (d.cost_model == 1 && d.row_model == 1) return nullptr;      // curvature-aware cost AND Gaussian rows (CM = 3, round 6): the generic kernel and the four-wave tick kernel of 21 <= N <= 31
    if (d.cost_model == 1) {
        // curvature-aware contouring (BASELINE configs[2]): the cfg-3 shape on the two-wave kernel, every other row mix of N <= 20 on the
        // runtime-shape one-wave kernel, anything else on the generic kernel -- all instantiated with CM = 1 (no profiled twins)
        if (prof) return nullptr;
        const int nrc = d.n_up + d.M + 14;
        if (lps != 3 && 4 * d.N <= 128 && d.n_up == 20 && d.M == 8 && !lab_env("TMPC_NO_TWO_WAVE")) {
            *threads = 128;
            return (SolveKernel)tmpc_solve_fast_kernel<20, 8, 4, 128, false, Solo, 1>;
        }
        if (lps == 3 && nrc <= 3 * 13) return (SolveKernel)tmpc_solve_fast_kernel<-1, 13, 3, 64, false, Solo, 1>;
        return nullptr;
    }
    if (d.row_model == 1) {
        // Gaussian chance-constraint rows instead of the ellipsoids (mpc_planner_jackal's default: N = 30, 5 topology + 5 Gaussian rows):
        // runtime-shape instantiations with CM = 2 -- two-wave for 22 <= N <= 32, one-wave for N <= 21, else the generic kernel
        if (prof) return nullptr;
        const int nrg = d.n_up + d.M + 14;
        // mpc_planner_jackal's default (generate_jackal_solver.py:53-73: N = 30, 5 + 5 rows) on ONE wave at two lanes per stage (round 6): twelve rows per lane
        // fit the registers, and eight one-wave trajectories per CU keep eight waves busy where four two-wave ones idle a wave through every Riccati sweep --
        // saturated +27 % (585 -> 741 k solves/s).  Its small-launch twin below, the compact kernel in pick_compact_kernel; ticks: latency modes 2 / 3.
        if (lps == 2 && d.n_up == 5 && d.M == 5 && !lab_env("TMPC_NO_ONE_WAVE_N30")) return (SolveKernel)tmpc_solve_fast_kernel<5, 5, 2, 64, false, Solo, 2>;
        if (lps != 3 && 4 * d.N <= 128 && !lab_env("TMPC_NO_TWO_WAVE")) {
            *threads = 128;
            if (d.n_up == 5 && d.M == 5) return (SolveKernel)tmpc_solve_fast_kernel<5, 5, 4, 128, false, Solo, 2>;     // mpc_planner_jackal's default (generate_jackal_solver.py:53-73), tuned
            if (nrg <= 4 * 6) return (SolveKernel)tmpc_solve_fast_kernel<-1, 6, 4, 128, false, Solo, 2>;
            if (nrg <= 4 * 12) return (SolveKernel)tmpc_solve_fast_kernel<-1, 12, 4, 128, false, Solo, 2>;
            *threads = NT;
        }
        if (lps == 3 && nrg <= 3 * 13) return (SolveKernel)tmpc_solve_fast_kernel<-1, 13, 3, 64, false, Solo, 2>;
        return nullptr;
    }
#endif
#ifdef TMPC_GENERATED_STAGE
    // generated solver: one row shape (tmpc_gen::NH upper-bounded rows); the fast instantiations are compiled only when the
    // generator's build found them free of scratch (TMPC_GEN_FAST / TMPC_GEN_FAST2 set by codegen/build.py)
#ifdef TMPC_GEN_FAST
    if (lps == 3) return TMPC_FAST(tmpc_gen::NH, 0, 3, 64);
#endif
#ifdef TMPC_GEN_FAST2
    if (lps != 3 && 4 * d.N <= 128) { *threads = 128; return TMPC_FAST(tmpc_gen::NH, 0, 4, 128); }
#endif
    return nullptr;
#else
    const int nr = d.n_up + d.M + 14;                    // interior-point rows per stage
    // the jackalsimulator T-MPC stack at the horizon it ships with (8 + 8 rows, N = 30; settings.yaml) on ONE wave at two lanes per stage (round 6, like
    // mpc_planner_jackal's default above: fifteen rows per lane still fit the registers); its compact twin in pick_compact_kernel
    if (lps == 2 && d.n_up == 8 && d.M == 8 && !prof && !lab_env("TMPC_NO_ONE_WAVE_N30")) return (SolveKernel)tmpc_solve_fast_kernel<8, 8, 2, 64, false, Solo, 0>;
    if (lps != 3 && 4 * d.N <= 128 && !lab_env("TMPC_NO_TWO_WAVE")) {
        // two waves per trajectory, 4 lanes per stage (22 <= N <= 32: the reference's default N = 30 and BASELINE cfg 3)
        SolveKernel k2 = nullptr;
        if (d.n_up == 8 && d.M == 8) k2 = TMPC_FAST(8, 8, 4, 128);
        else if (d.n_up == 12 && d.M == 12) k2 = TMPC_FAST(12, 12, 4, 128);  // mpc_planner_jackalsimulator defaults (N = 30, 12 obstacles)
        else if (d.n_up == 20 && d.M == 8) k2 = TMPC_FAST(20, 8, 4, 128);    // cfg 3: 8 topology + 12 decomp rows + 8 ellipsoids
        else if (nr <= 4 * 6) k2 = TMPC_FAST(-1, 6, 4, 128);                 // any other row mix: runtime-shape instantiations
        else if (nr <= 4 * 9) k2 = TMPC_FAST(-1, 9, 4, 128);                 //   (e.g. mpc_planner_jackal: N = 30, 5 obstacles)
        else if (nr <= 4 * 12) k2 = TMPC_FAST(-1, 12, 4, 128);
        if (k2) { *threads = 128; return k2; }
    }
    if (lps == 3) {
        if (d.n_up == 0 && d.M == 4) return TMPC_FAST(0, 4, 3, 64);
        if (d.n_up == 8 && d.M == 8) return TMPC_FAST(8, 8, 3, 64);
        if (d.n_up == 12 && d.M == 12) return TMPC_FAST(12, 12, 3, 64);      // zero scratch only with machine-LICM off (build flag)
        if (d.n_up == 24 && d.M == 0) return TMPC_FAST(24, 0, 3, 64);        // SH-MPC: 24 scenario halfspaces (cfg 5)
        if (nr <= 3 * 7) return TMPC_FAST(-1, 7, 3, 64);                     // runtime-shape instantiations
        if (nr <= 3 * 10) return TMPC_FAST(-1, 10, 3, 64);
        if (nr <= 3 * 13) return TMPC_FAST(-1, 13, 3, 64);
        if (d.N <= 2 * (64 / 6) && nr <= 6 * 9 && !lab_env("TMPC_NO_TWO_WAVE")) {   // more rows: two waves, 6 lanes per stage
            *threads = 128;                                                  //   (mpc_planner_rosnavigation T-MPC: 24 + 12 rows)
            return TMPC_FAST(-1, 9, 6, 128);
        }
    } else if (lps == 2) {
        if (d.n_up == 0 && d.M == 4) return TMPC_FAST(0, 4, 2, 64);
    }
    return nullptr;
#endif
}
// Square-root form of the Riccati recursion (tmpc_dims.riccati_form = TMPC_RICCATI_SQUARE_ROOT; csrc/tmpc_riccati.hpp SQ): run-time-shape fast kernels only
// -- a comparison aid (HPIPM's default recursion) with one instantiation per kernel family member that the BASELINE shapes need, not a throughput path.
static SolveKernel pick_sqrt_kernel(const Dims &d, int *threads)
{
    *threads = NT;
#ifndef TMPC_GENERATED_STAGE
    const int nr = d.n_up + d.M + 14, sm = stage_model(d);
    if (3 * d.N <= NT) {
        if (sm == 0 && nr <= 3 * 13) return (SolveKernel)tmpc_solve_fast_kernel<-1, 13, 3, 64, false, SoloSqrt>;
        return nullptr;
    }
    if (4 * d.N <= 128 && nr <= 4 * 12) {
        *threads = 128;
        if (sm == 0) return (SolveKernel)tmpc_solve_fast_kernel<-1, 12, 4, 128, false, SoloSqrt>;
        if (sm == 1 && d.n_up == 20 && d.M == 8) return (SolveKernel)tmpc_solve_fast_kernel<20, 8, 4, 128, false, SoloSqrt, 1>;      // cfg 3 as named (CA-MPC)
    }
#endif
    (void)d;
    return nullptr;
}
// Compact variant (tmpc_fast.hpp: tmpc_solve_compact_kernel): two waves per SIMD, eight trajectories per CU, persistent
// workgroups.  Bitwise the same results as the fast kernel of the shape (tools/ab_compare.py against TMPC_NO_COMPACT=1).
// Round 4: the shapes with 13 rows per lane ((12,12) and (24,0) at three lanes per stage: cfg 4, cfg 5) fit 256 registers too since the
// row passes are specialised by the compile-time kind of each row slot (FastCfg::KIND): 238 registers, zero scratch; their larger row tables
// allow 7 (cfg 4: 23.3 KB) and 6 (cfg 5: 25.2 KB) workgroups per CU.  The runtime-shape instantiation with 13 rows per lane still spills
// (168 B) and is not registered.
// *lay: the instantiation's Hh layout (compact_layout, tmpc_fast.hpp) -- evaluated on the template arguments
// where they are written, so that the host's LDS size and the kernel's layout cannot disagree
#define TMPC_CP(a, b, c) (*lay = compact_layout(a, b, 64), (SolveKernel)tmpc_solve_compact_kernel<a, b, c, false>)
#define TMPC_CP2(a, b, c, m) (*lay = compact_layout(a, b, 128), (SolveKernel)tmpc_solve_compact_kernel<a, b, c, false, 128, m>)
static SolveKernel pick_compact_kernel(const Dims &d, bool prof, int *lay)
{
    *lay = 1;
#ifndef TMPC_GENERATED_STAGE
    if (!lab_env("TMPC_FORCE_GENERIC") && !lab_env("TMPC_NO_COMPACT") && !lab_env("TMPC_NO_ONE_WAVE_N30") && !prof && 2 * d.N <= NT && 3 * d.N > NT) {
        // 22 <= N <= 32 on one wave, two lanes per stage (pick_fast_kernel): mpc_planner_jackal's default, the jackalsimulator stack
        if (stage_model(d) == 2 && d.n_up == 5 && d.M == 5) { *lay = compact_layout(5, 5, 64); return (SolveKernel)tmpc_solve_compact_kernel<5, 5, 2, false, 64, 2>; }
        if (stage_model(d) == 0 && d.n_up == 8 && d.M == 8) { *lay = compact_layout(8, 8, 64); return (SolveKernel)tmpc_solve_compact_kernel<8, 8, 2, false>; }
    }
    if (lab_env("TMPC_FORCE_GENERIC") || lab_env("TMPC_NO_COMPACT") || prof || d.N > 20 || (stage_model(d) != 0 && stage_model(d) != 2)) return nullptr;
    if (stage_model(d) == 2) {                       // Gaussian chance-constraint rows (round 6): the run-time-shape instantiation with up to ten rows per lane
        if (d.n_up + d.M + 14 <= 3 * 10) { *lay = compact_layout(-1, 10, 64); return (SolveKernel)tmpc_solve_compact_kernel<-1, 10, 3, false, 64, 2>; }
        return nullptr;
    }
    const int nr = d.n_up + d.M + 14;                    // interior-point rows per stage
    if (d.n_up == 8 && d.M == 8) return TMPC_CP(8, 8, 3);
    if (d.n_up == 0 && d.M == 4) return TMPC_CP(0, 4, 3);
    if (d.n_up == 12 && d.M == 12) return TMPC_CP(12, 12, 3);
    if (d.n_up == 24 && d.M == 0) return TMPC_CP(24, 0, 3);
    if (nr <= 3 * 7) return TMPC_CP(-1, 7, 3);       // runtime-shape instantiations
    if (nr <= 3 * 10) return TMPC_CP(-1, 10, 3);
#endif
    (void)d; (void)prof;
    return nullptr;
}
// Two-wave compact variant (round 4: 22 <= N <= 32, four lanes per stage -- the reference's N = 30 defaults, cfg 3): the same kernel with
// NTH = 128.  236 / 251 registers, zero scratch, 27-39 KB of LDS: FOUR trajectories per CU (two waves each, two waves per SIMD) where the
// fast two-wave kernel (57-70 KB of LDS) holds two.  Bitwise the same results; a trajectory takes longer on it (NLP data in the global
// workspace, the linearisation on one of the two waves), so launch_solve uses it only for launches that the fast kernel could not hold
// resident at once (more than two trajectories per CU).  The runtime-shape instantiation with 12 rows per lane spills (144 B): not registered.
static SolveKernel pick_compact2_kernel(const Dims &d, int *lay, int *threads)
{
    *lay = 1; *threads = 128;
#ifndef TMPC_GENERATED_STAGE

    if (lab_env("TMPC_FORCE_GENERIC") || lab_env("TMPC_NO_COMPACT") || lab_env("TMPC_NO_TWO_WAVE") || 3 * d.N <= NT || 4 * d.N > 128) return nullptr;
    const int nr = d.n_up + d.M + 14, sm = stage_model(d);
    if (sm == 1) return (d.n_up == 20 && d.M == 8) ? TMPC_CP2(20, 8, 4, 1) : nullptr;      // cfg 3 as named (CA-MPC)
    if (sm == 2) {                                                                                                                   // Gaussian chance-constraint rows
        if (d.n_up == 5 && d.M == 5) return TMPC_CP2(5, 5, 4, 2);                          // mpc_planner_jackal's default, tuned (round 5)
        return nr <= 4 * 6 ? TMPC_CP2(-1, 6, 4, 2) : nullptr;
    }
    if (sm != 0) return nullptr;
    if (d.n_up == 20 && d.M == 8) return TMPC_CP2(20, 8, 4, 0);
    if (d.n_up == 12 && d.M == 12) return TMPC_CP2(12, 12, 4, 0);
    if (d.n_up == 8 && d.M == 8) return TMPC_CP2(8, 8, 4, 0);
    if (nr <= 4 * 6) return TMPC_CP2(-1, 6, 4, 0);
    if (nr <= 4 * 9) return TMPC_CP2(-1, 9, 4, 0);
#endif
    (void)d;
    return nullptr;
}
// Latency variant (tmpc_set_latency_mode): two waves per trajectory at 6 lanes per stage, built for two waves per SIMD
// (<= 256 registers, so four trajectories per CU stay resident).  The stage-parallel phases run on twice the lanes:
// -8 % kernel time on a 64-trajectory control tick; on a saturated GPU the one-wave kernel is as fast or faster, which is
// why it stays the default.  The variant is chosen by the caller, never by the batch size: a trajectory's result does
// not depend on what else is in the launch.
static SolveKernel pick_latency_kernel(const Dims &d, bool prof)
{
#ifndef TMPC_GENERATED_STAGE
    if (lab_env("TMPC_FORCE_GENERIC") || lab_env("TMPC_NO_TWO_WAVE") || d.N > 2 * (64 / 6) || stage_model(d) != 0) return nullptr;
    if (d.n_up == 8 && d.M == 8) return TMPC_FAST(8, 8, 6, 128);
#endif
    (void)d; (void)prof;
    return nullptr;
}
// Latency variant 2 (tmpc_set_latency_mode(h, 2)): one wave per trajectory like the fast kernels, the interior-point Newton systems
// solved parallel in time (tmpc_scan.hpp) instead of by the sequential Riccati recursion.  One workgroup per CU is what a control
// tick gives it anyway: built for one wave per SIMD (all 512 registers, 73 KB of LDS).  Another factorisation of the same systems:
// steps agree with the recursion's to rounding (~1e-6 of a step on ill-conditioned late iterations, like the recursion itself
// against an exact solve), so iteration counts can differ by one where a residual sits at the tolerance -- the caller opts in.
static SolveKernel pick_scan_kernel(const Dims &d, int *threads, int *sl)
{
    *sl = 3;
#ifndef TMPC_GENERATED_STAGE
    if (lab_env("TMPC_FORCE_GENERIC") || d.N > 31 || d.N < 2 || (stage_model(d) != 0 && stage_model(d) != 2)) return nullptr;
    const bool gauss = stage_model(d) == 2;                      // Gaussian chance-constraint rows (mpc_planner_jackal's default stack): the run-time-shape instantiations, CM = 2
    if (d.N > 20) {                                              // 21 <= N <= 31 (cfg 3, the reference's N = 30 defaults): two lanes per stage in the
        if (d.n_up + d.M + 14 > 4 * 12) return nullptr;          // Newton solve, the runtime-shape two-wave kernel (4 lanes per stage, up to 34 rows) around it
        *threads = 128; *sl = 2;
        return gauss ? (SolveKernel)tmpc_solve_fast_kernel<-1, 12, 4, 128, false, ScanSoloT<2>, 2> : (SolveKernel)tmpc_solve_fast_kernel<-1, 12, 4, 128, false, ScanSoloT<2>>;
    }
    if (gauss) {
        if (d.N <= 2 * (64 / 6) && d.n_up + d.M + 14 <= 6 * 9) { *threads = 128; return (SolveKernel)tmpc_solve_fast_kernel<-1, 9, 6, 128, false, ScanSolo, 2>; }
        return nullptr;
    }
    const char *w = lab_env("TMPC_SCAN_WAVES");               // A/B: "1" = one wave per trajectory
    if (d.n_up == 8 && d.M == 8 && d.N <= 2 * (64 / 6) && !(w && atoi(w) == 1)) { *threads = 128; return (SolveKernel)tmpc_solve_fast_kernel<8, 8, 6, 128, false, ScanSolo>; }
    if (d.n_up == 8 && d.M == 8) { *threads = 64; return (SolveKernel)tmpc_solve_fast_kernel<8, 8, 3, 64, false, ScanSolo>; }
    if (d.N <= 2 * (64 / 6) && d.n_up + d.M + 14 <= 6 * 9) {     // every other row mix of the one-wave shapes (cfg 1, cfg 4, cfg 5, ...): runtime row counts, two waves
        *threads = 128;
        return (SolveKernel)tmpc_solve_fast_kernel<-1, 9, 6, 128, false, ScanSolo>;
    }
#endif
    (void)d; (void)threads;
    return nullptr;
}
// Latency variant 3 (tmpc_set_latency_mode(h, 3), round 6): FOUR waves per trajectory -- a control tick of a few planners leaves a whole CU (four SIMDs,
// 160 KB of LDS) to every trajectory.  The stage evaluation is split four ways by content (dynamics | cost + halfspace rows | obstacle rows on two waves),
// the interior-point row passes run at twelve lanes per stage (three rows per lane instead of five), and the wide phases of the parallel-in-time
// factorisation (the stage phase: one column per lane instead of four; level 0 of the cyclic reduction: one instead of two) use all 256 lanes
// (csrc/tmpc_scan.hpp factor4).  Same algorithm as variant 2 (sums associate differently: rounding level).  N <= 20, hand-written MPCC stages.
// `ab`: TMPC_QUAD_AB=1 in a lab build picks the twin whose factorisation stays on one wave (A/B of the factorisation split alone).
static SolveKernel pick_quad_kernel(const Dims &d, bool prof, bool ab, int *sl = nullptr)
{
    if (sl) *sl = 3;
#ifndef TMPC_GENERATED_STAGE
    if (d.N > 20 && d.N <= 31 && !prof && !ab && d.n_up + d.M + 14 <= 8 * 6) {
        // 21 <= N <= 31 (the horizon the reference ships for jackal / jackalsimulator: N = 30): eight lanes per stage around the two-lanes-per-stage Newton solve;
        // every stage model (the four-wave linearisation regularises a coupled W -- curvature-aware cost -- on wave 0)
        if (sl) *sl = 2;
        const int sm = stage_model(d);
        return sm == 0 ? (SolveKernel)tmpc_solve_fast_kernel<-1, 6, 8, 256, false, ScanQuadT<2>, 0>
             : sm == 1 ? (SolveKernel)tmpc_solve_fast_kernel<-1, 6, 8, 256, false, ScanQuadT<2>, 1>
             : sm == 2 ? (SolveKernel)tmpc_solve_fast_kernel<-1, 6, 8, 256, false, ScanQuadT<2>, 2>
             : sm == 3 ? (SolveKernel)tmpc_solve_fast_kernel<-1, 6, 8, 256, false, ScanQuadT<2>, 3> : nullptr;
    }
    if (d.N > 20 || d.N < 2 || (stage_model(d) != 0 && stage_model(d) != 2)) return nullptr;
    if (stage_model(d) == 2) return (!prof && d.n_up + d.M + 14 <= 12 * 4) ? (SolveKernel)tmpc_solve_fast_kernel<-1, 4, 12, 256, false, ScanQuad, 2> : nullptr;      // Gaussian rows
    if (d.n_up == 8 && d.M == 8) {
        if (prof) return (SolveKernel)tmpc_solve_fast_kernel<8, 8, 12, 256, true, ScanQuad>;
        return ab ? (SolveKernel)tmpc_solve_fast_kernel<8, 8, 12, 256, false, ScanSolo> : (SolveKernel)tmpc_solve_fast_kernel<8, 8, 12, 256, false, ScanQuad>;
    }
    if (!prof && d.n_up + d.M + 14 <= 12 * 4) return (SolveKernel)tmpc_solve_fast_kernel<-1, 4, 12, 256, false, ScanQuad>;      // cfg 1, cfg 4, cfg 5, any row mix up to 34 rows
#endif
    (void)d; (void)prof; (void)ab;
    return nullptr;
}
// ---- stage stride of the row Jacobians in LDS (Dims::dpad) --------------------------------------------------------------------
// LDS bank-conflict model of the row passes' coefficient loads (ipm_fast: coef()).  Lane (stage k, sub-lane c) of a wave owns the rows c, c + LPS, ...;
// per row slot the lanes read the row's three entries as three 8-byte accesses at  k * dstride + offset(row)  (rows without a Jacobian read the
// zero triple: one address, a broadcast).  64 banks of 4 bytes; an access costs as many passes as the most loaded bank has DISTINCT dwords.
// Returns the passes summed over the row slots and the three entries: with the bare strides (48 doubles for 8 + 8 rows, 40 packed) every second
// / fourth stage falls on the same banks -- 180 passes where 36 would do (fast (8,8) layout), 83 / 36 packed.  The model only ranks strides; the
// measured effect is in profiles/round5_p_dpad_ab.jsonl.
static int d_load_passes(int N, int n_pair, int nh, int threads, int dstride)
{
    const int lps = threads == 64 ? (3 * N <= 64 ? 3 : 2) : (N <= 21 ? 6 : 4);       // lanes per stage of the kernel families (pick_*_kernel)
    const int spw = 64 / lps, nk = N < spw ? N : spw;                                 // stages of one wave (two-wave kernels: each wave loads for its own)
    const int rpl = (nh + 14 + lps - 1) / lps;
    int total = 0;
    std::vector<int> dw;
    for (int s = 0; s < rpl && lps * s < nh; s++)
        for (int part = 0; part < 3; part++) {
            dw.clear();
            for (int k = 0; k < nk; k++)
                for (int c = 0; c < lps; c++) {
                    const int r = c + lps * s;
                    int a = N * dstride + part;                                       // zero triple
                    if (r < nh) {
                        const int off = r < n_pair ? 2 * r : 3 * r - n_pair;
                        if (part < 2 || r >= n_pair) a = k * dstride + off + part; else a = N * dstride + 2;
                    }
                    dw.push_back(2 * a); dw.push_back(2 * a + 1);
                }
            std::sort(dw.begin(), dw.end());
            dw.erase(std::unique(dw.begin(), dw.end()), dw.end());
            int load[64] = {0}, worst = 0;
            for (int x : dw) { const int b = ++load[x & 63]; if (b > worst) worst = b; }
            total += worst;
        }
    return total;
}
// the padding in 0 .. max_pad that the model likes best (ties: the smaller); `allowed(pad)`: the layout still fits what it has to fit
template <typename Allowed>
static int pick_d_pad(int N, int n_pair, int nh, int threads, int max_pad, Allowed allowed)
{
    if (const char *e = lab_env("TMPC_EXP_DPAD")) { const int v = atoi(e); return v >= 0 && v <= max_pad && allowed(v) ? v : 0; }   // experiments ("0": the bare strides)
    const int base = 2 * n_pair + 3 * (nh - n_pair);
    int best = 0, best_cost = d_load_passes(N, n_pair, nh, threads, base);
    for (int pad = 1; pad <= max_pad; pad++) {
        if (!allowed(pad)) continue;
        const int c = d_load_passes(N, n_pair, nh, threads, base + pad);
        if (c < best_cost) { best = pad; best_cost = c; }
    }
    return best;
}
}  // namespace tmpc

struct tmpc_handle {
    tmpc::Dims d{};
    int B_max = 0, B = 0, device = 0;
    hipStream_t stream = nullptr;
    // inputs: owned staging buffers (tmpc_set_batch) or borrowed device pointers (tmpc_set_batch_device)
    double *o_xinit = nullptr, *o_x0 = nullptr, *o_params = nullptr;
    const double *xinit = nullptr, *x0 = nullptr, *params = nullptr;
    double *xtraj = nullptr, *utraj = nullptr, *pobj = nullptr, *res_eq = nullptr, *d_weight = nullptr;
    int *exit_code = nullptr, *qp_status = nullptr, *sqp_iter = nullptr, *qp_iter = nullptr, *d_best = nullptr;
    uint8_t *d_disabled = nullptr;
    // Control-tick handles (inputs and outputs of B_max trajectories <= TICK_SLAB_MAX bytes each): the owned input buffers (+ the slot map) are ONE device
    // allocation, the outputs another, each mirrored in pinned host memory with the same layout -- tmpc_set_batch is one asynchronous H2D copy instead of
    // three staged ones, tmpc_set_slots needs no stream synchronisation, tmpc_get is one D2H copy instead of eight (round 6: the C++ BatchContext's tick
    // spent ~0.2 ms of 1.1 ms in those thirteen transfers).  Larger handles (the bench's 32768 trajectories) keep separate allocations and direct copies.
    char *slab_in = nullptr, *slab_out = nullptr, *pin_in = nullptr, *pin_out = nullptr;
    size_t slab_in_bytes = 0, slab_out_bytes = 0;
    hipEvent_t in_done = nullptr, slot_done = nullptr;    // the last H2D copies out of pin_in -- batch inputs / slot map: disjoint regions of the mirror, each rewritten only after ITS copy
    bool in_pending = false, slot_pending = false;
    size_t lds_bytes = 0;
    tmpc::SolveKernel kernel = nullptr;
    int threads = tmpc::NT;          // threads per trajectory (64, or 128 for the two-wave fast variant)
    tmpc::SolveKernel kernel_lat = nullptr;   // optional latency variant (128 threads), used when latency_mode is 1
    tmpc::SolveKernel kernel_scan = nullptr;  // optional latency variant 2 (parallel-in-time Newton solve, 64 threads)
    tmpc::SolveKernel kernel_quad = nullptr;  // optional latency variant 3 (four waves per trajectory, 256 threads; LDS = lds_bytes_scan3)
    size_t lds_bytes_quad = 0;
    size_t lds_bytes_scan = 0;
    int scan_threads = 64, scan_sl = 3;
    size_t lds_bytes_fast = 0;                // LDS of the fast-layout kernels (the profiled twin) when `kernel` is compact
    size_t lds_bytes_fast2 = 0;               // ... of their two-wave variants (kernel_lat): + the W shares parked during the linearisation
    bool compact = false;                     // `kernel` is a compact persistent kernel: grid = resident workgroups, needs ws + ticket
    int grid_max = 0;                         // resident workgroups of the compact kernel on this device
    double *ws = nullptr;                     // [grid_max][ws_doubles(N)] per-workgroup NLP workspace
    int *ticket = nullptr;
    tmpc::SolveKernel kernel_small = nullptr; // compact shapes (N <= 21): the fast one-wave kernel, for launches of at most cp_min_B trajectories
    size_t lds_bytes_small = 0;
    int cp_min_B = 0;                         // what the fast one-wave kernel holds resident at once (workgroups per CU x CUs)
    tmpc::SolveKernel kernel_cp2 = nullptr;   // optional two-wave compact variant (22 <= N <= 32): launches of more than cp2_min_B trajectories
    size_t lds_bytes_cp2 = 0;
    int cp2_min_B = 0;                        // what the fast two-wave kernel holds resident at once (workgroups per CU x CUs)
    int cp2_threads = 128;
    int dpad_cp = 0, dpad_cp2 = 0;            // Dims::dpad of the compact one-wave / two-wave kernel (pick_d_pad); the fast layouts do not pad
    int lay_cp = 1, lay_cp2 = 1;              // the compact kernels' Hh layout (pick_compact*_kernel)
    bool prio_cp = false, prio_cp2 = false;   // wave issue priorities (Dims::prio) for the compact one-wave / two-wave kernel: only when its residency puts two waves on
                                              // every SIMD (8 waves per CU) -- with an odd count the waves that share a SIMD starve and set the makespan (tmpc_riccati.hpp)
    int latency_mode = 0;                     // 0: throughput kernels, 1: two-wave variant, 2: parallel-in-time variant
    bool throughput_mode = false;             // lane-per-trajectory kernels (tmpc_lanes.hip) instead of one wave per trajectory
    tmpc::lanes::Context *lanes = nullptr;    // their HBM workspace, created when the mode is first enabled
    bool fast = false;
    // persistent per-slot solver state (tmpc_solve_iterations), allocated on first use
    double *st_z = nullptr, *st_pi = nullptr, *st_lamh = nullptr;
    int *st_stopped = nullptr;
    int *st_has = nullptr;           // [B_max] the slot holds state of an earlier tmpc_solve_iterations (set by the kernels' store)
    int *d_slot = nullptr;           // [B_max] state slot of every batch entry (tmpc_set_slots)
    bool slots_set = false;
    int slots_B = 0;                 // batch size the slot map was given for: a map of another size is refused, never read past its end
    int *d_share = nullptr;          // [B_max] tmpc_set_param_sharing
    int share_B = 0;                 // batch size the sharing map was given for (0: none)
    bool share_strict = false;       // tmpc_set_param_sharing_ex(.., TMPC_SHARE_COPIES_NOT_MAINTAINED): a solve that cannot honour the map is an error
    bool st_valid = false;           // lane kernels (state = their workspace, per launch): it holds the result of a previous call ...
    int st_B = 0;                    // ... for slots [0, st_B)
    // SH-MPC bookkeeping: the sample behind each scenario row of the last tmpc_scenario_halfspaces (i32 [B][N][scn_rows])
    unsigned char *scn_discard = nullptr;     // [B_max][scn_discard_S] scenarios discarded for each trajectory (tmpc_scenario_discard); applies to the next tmpc_scenario_halfspaces
    int scn_discard_S = 0, scn_discard_B = 0, scn_discard_n = 0;
    size_t scn_discard_cap = 0;
    int *scn_sample = nullptr;
    size_t scn_cap = 0;
    int scn_rows = 0, scn_B = 0;
    std::vector<hipEvent_t> ev;      // per-launch timing events (pairs)
    int ev_used = 0;
    bool timing = false;
    std::string err;
};

#define TMPC_HIP_CHECK(h, expr)                                                                     \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                           \
            return TMPC_ERR_HIP;                                                                    \
        }                                                                                           \
    } while (0)

namespace {
// Scratch device buffers / events of the diagnostic entry points: released on every return path.
struct DevBufs {
    std::vector<void *> p;
    ~DevBufs() { for (void *q : p) if (q) (void)hipFree(q); }
    hipError_t alloc(double **out, size_t bytes) { hipError_t e = hipMalloc(out, bytes ? bytes : 8); if (e == hipSuccess) p.push_back(*out); return e; }
};
struct Events {
    std::vector<hipEvent_t> ev;
    ~Events() { for (auto &e : ev) if (e) (void)hipEventDestroy(e); }
};
}  // namespace

extern "C" {

void tmpc_default_dims(tmpc_dims *d, int32_t N, int32_t S, int32_t n_lin, int32_t M) { tmpc_default_dims_ex(d, N, S, n_lin, M, 0, 0); }

void tmpc_default_dims_ex(tmpc_dims *d, int32_t N, int32_t S, int32_t n_lin, int32_t M, int32_t n_slk, int32_t slack)
{
    memset(d, 0, sizeof *d);
#ifdef TMPC_GENERATED_STAGE
    // generated solver: the row / parameter structure is fixed by the generated stage functions
    (void)n_lin; (void)M; (void)n_slk; (void)slack;
    n_lin = tmpc_gen::NH; M = 0; n_slk = 0; slack = tmpc_gen::SLACK;
#endif
    d->N = N; d->S = S; d->n_lin = n_lin; d->M = M; d->n_slk = n_slk; d->slack = slack ? 1 : 0;
    tmpc::Dims t{}; t.S = S; t.n_lin = n_lin; t.M = M; t.n_slk = n_slk; t.slack = d->slack;
    d->npar = tmpc::expected_npar(t);
    d->n_sqp = 10; d->qp_iter_max = 50; d->erk_steps = 3;
    d->dt = 0.2; d->qp_tol = 1e-5; d->reg_eps = 1e-4; d->ipm_mu0 = 0.01; d->ipm_thr0 = 0.01;
    const double lb[TMPC_NV] = {-2.0, -0.8, -2000.0, -2000.0, -M_PI * 4, -0.01, -1.0};
    const double ub[TMPC_NV] = {2.0, 0.8, 2000.0, 2000.0, M_PI * 4, 3.0, 10000.0};
    for (int i = 0; i < TMPC_NV; i++) { d->lb[i] = lb[i]; d->ub[i] = ub[i]; }
#ifdef TMPC_GENERATED_STAGE
    for (int i = 0; i < TMPC_NV; i++) { d->lb[i] = tmpc_gen::LB[i]; d->ub[i] = tmpc_gen::UB[i]; }     // the plugin model's own bounds (emit.py)
#endif
}

int tmpc_create_v2(tmpc_handle **out, const tmpc_dims *dims_in, uint32_t dims_size, int32_t B_max, int32_t device)
{
    // a caller built against an older header passes a SHORTER struct: the fields it does not know are the defaults (0), never garbage.  Only the
    // struct's revision boundaries are sizes a header of this library ever had (round-5 advisor: a size ending inside a field copied part of it)
    constexpr uint32_t kRev[] = {(uint32_t)offsetof(tmpc_dims, cost_model),         // rounds 1-3: up to n_slk / slack
                                 (uint32_t)offsetof(tmpc_dims, riccati_form),       // rounds 4-5: + cost_model, row_model
                                 (uint32_t)sizeof(tmpc_dims)};                      // round 6: + riccati_form
    if (!out || !dims_in || dims_size > 4096) return TMPC_ERR_INVALID;
    if (out) *out = nullptr;
    bool known = dims_size > sizeof(tmpc_dims);
    for (uint32_t r : kRev) known |= dims_size == r;
    if (!known) return TMPC_ERR_INVALID;
    if (dims_size > sizeof(tmpc_dims)) {                                           // a NEWER header: fields this library does not know must be unset
        const unsigned char *tail = (const unsigned char *)dims_in + sizeof(tmpc_dims);
        for (uint32_t i = 0; i < dims_size - (uint32_t)sizeof(tmpc_dims); i++) if (tail[i] != 0) return TMPC_ERR_INVALID;
    }
    tmpc_dims d;
    memset(&d, 0, sizeof d);
    memcpy(&d, dims_in, dims_size < sizeof d ? dims_size : sizeof d);
    return tmpc_create(out, &d, B_max, device);
}

int tmpc_create(tmpc_handle **out, const tmpc_dims *dims, int32_t B_max, int32_t device)
{
    if (!out || !dims || B_max <= 0) return TMPC_ERR_INVALID;
    *out = nullptr;
    {
        tmpc::Dims t{}; t.S = dims->S; t.n_lin = dims->n_lin; t.M = dims->M; t.n_slk = dims->n_slk; t.slack = dims->slack; t.row_model = dims->row_model;
        if (dims->row_model != 0 && dims->row_model != 1) return TMPC_ERR_INVALID;
        if (dims->N < 2 || dims->N > 62 || dims->S < 1 || dims->M < 0 || dims->n_lin < 0 || dims->n_slk < 0 ||
            (dims->slack != 0 && dims->slack != 1) || dims->npar != tmpc::expected_npar(t) || dims->erk_steps < 1 ||
            dims->n_sqp < 1 || dims->qp_iter_max < 1 || !(dims->dt > 0.0) || !(dims->qp_tol > 0.0) || !(dims->reg_eps > 0.0) ||
            !(dims->ipm_mu0 > 0.0) || !(dims->ipm_thr0 > 0.0) || (dims->cost_model != 0 && dims->cost_model != 1) ||
            (dims->riccati_form != TMPC_RICCATI_SCHUR && dims->riccati_form != TMPC_RICCATI_SQUARE_ROOT))
            return TMPC_ERR_INVALID;
#ifdef TMPC_GENERATED_STAGE
        if (dims->n_lin != tmpc_gen::NH || dims->M != 0 || dims->n_slk != 0 || dims->slack != tmpc_gen::SLACK || dims->cost_model != 0 || dims->row_model != 0) return TMPC_ERR_INVALID;
#endif
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return TMPC_ERR_NO_DEVICE;
    tmpc_handle *h = new tmpc_handle();
    h->device = device; h->B_max = B_max;
    tmpc::Dims &d = h->d;
    d.N = dims->N; d.S = dims->S; d.n_lin = dims->n_lin; d.M = dims->M; d.npar = dims->npar;
    d.n_slk = dims->n_slk; d.slack = dims->slack; d.cost_model = dims->cost_model; d.row_model = dims->row_model;
    d.n_sqp = dims->n_sqp; d.qp_iter_max = dims->qp_iter_max; d.erk_steps = dims->erk_steps;
    d.dt = dims->dt; d.qp_tol = dims->qp_tol; d.reg_eps = dims->reg_eps; d.mu0 = dims->ipm_mu0; d.thr0 = dims->ipm_thr0;
    for (int i = 0; i < TMPC_NV; i++) { d.lb[i] = dims->lb[i]; d.ub[i] = dims->ub[i]; }
#ifdef TMPC_GENERATED_STAGE
    d.model = tmpc_gen::MODEL;               // fixed by the module stack's model (emit.py): 1 = SecondOrderUnicycleModel, the fifth state slot inert
#else
    d.model = 0;
#endif
    d.riccati_form = dims->riccati_form;
    tmpc::derive_dims(d);
    d.split_rows = tmpc::split_rows_for(d.N, d.n_up + d.M) ? 1 : 0;
    h->kernel = d.riccati_form == TMPC_RICCATI_SQUARE_ROOT ? tmpc::pick_sqrt_kernel(d, &h->threads) : tmpc::pick_fast_kernel(d, &h->threads, false);
    if (d.riccati_form == TMPC_RICCATI_SQUARE_ROOT && !h->kernel) { delete h; return TMPC_ERR_INVALID; }       // (no square-root instantiation for this shape: never a silent other form)
    if (const char *lm = lab_env("TMPC_LATENCY_MODE")) {      // experiments: latency variant regardless of the caller ("0" .. "3"; anything else is ignored)
        if (lm[0] >= '0' && lm[0] <= '3' && lm[1] == '\0') h->latency_mode = lm[0] - '0';
    }
    h->fast = h->kernel != nullptr;
    if (h->fast) h->lds_bytes = sizeof(double) * (size_t)tmpc::lds_doubles_fast(d.N, d.n_up + d.M);
    else { const int sm = tmpc::stage_model(d); h->kernel = sm == 3 ? tmpc::tmpc_solve_kernel<3> : sm == 1 ? tmpc::tmpc_solve_kernel<1> : (sm == 2 ? tmpc::tmpc_solve_kernel<2> : tmpc::tmpc_solve_kernel<0>); h->lds_bytes = sizeof(double) * (size_t)tmpc::lds_doubles(d.N, d.n_up + d.M); }
    h->lds_bytes_fast = h->lds_bytes;
    // two-wave (128-thread) fast kernels park one share of W per stage behind the layout while they linearise (linearise<.., 128>)
    // (a shape whose default is the generic kernel may still have a four-wave tick kernel -- curvature-aware cost + Gaussian rows: the fast LAYOUT's size then)
    h->lds_bytes_fast2 = (h->fast ? h->lds_bytes_fast : sizeof(double) * (size_t)tmpc::lds_doubles_fast(d.N, d.n_up + d.M)) + sizeof(double) * (size_t)d.N * tmpc::NP28;
    if (h->fast && h->threads == 128) h->lds_bytes = h->lds_bytes_fast2;
    auto fail = [&](int code) { delete h; return code; };
    if (hipSetDevice(device) != hipSuccess) return fail(TMPC_ERR_HIP);
    if (h->lds_bytes > 160 * 1024) return fail(TMPC_ERR_INVALID);
    const bool schur = d.riccati_form == TMPC_RICCATI_SCHUR;      // the square-root form has its fast kernels only: no latency / compact variants
    if (schur && h->fast && h->threads == tmpc::NT && (h->kernel_lat = tmpc::pick_latency_kernel(d, false)) != nullptr) {
        if (hipFuncSetAttribute((const void *)h->kernel_lat, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes_fast2) != hipSuccess)
            h->kernel_lat = nullptr;
    }
    if (schur && h->fast && (h->threads == tmpc::NT || d.N > 20) && (h->kernel_scan = tmpc::pick_scan_kernel(d, &h->scan_threads, &h->scan_sl)) != nullptr) {
        h->lds_bytes_scan = h->lds_bytes_fast2 + sizeof(double) * (size_t)(h->scan_sl == 3 ? tmpc::scan::lds_doubles<3>(d.N) : tmpc::scan::lds_doubles<2>(d.N));
        if (h->lds_bytes_scan > 160 * 1024) h->kernel_scan = nullptr;
    }
    if (h->kernel_scan) {
        if (hipFuncSetAttribute((const void *)h->kernel_scan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes_scan) != hipSuccess)
            h->kernel_scan = nullptr;
    }
    int quad_sl = 3;
    if (schur && (h->fast || tmpc::stage_model(d) == 3) && (h->threads == tmpc::NT || d.N > 20) && (h->kernel_quad = tmpc::pick_quad_kernel(d, false, lab_env("TMPC_QUAD_AB") != nullptr, &quad_sl)) != nullptr) {
        // fast layout + the W shares of the split linearisation (wave 0's N x 28, the obstacle lanes' 36 N / 24 N: they lie inside the scan scratch, which is dead then) + the scan scratch
        h->lds_bytes_quad = h->lds_bytes_fast2 + sizeof(double) * (size_t)(quad_sl == 3 ? tmpc::scan::lds_doubles<3>(d.N) : tmpc::scan::lds_doubles<2>(d.N));
        if (h->lds_bytes_quad > 160 * 1024 ||
            hipFuncSetAttribute((const void *)h->kernel_quad, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes_quad) != hipSuccess)
            h->kernel_quad = nullptr;
    }
    if (tmpc::SolveKernel kc = (schur && h->fast && h->threads == tmpc::NT) ? tmpc::pick_compact_kernel(d, false, &h->lay_cp) : nullptr) {
        // the fast kernel of the shape (everything in LDS, four per CU) stays for launches it holds resident at once: bitwise the same results
        // (tests/test_gpu_compact2.py), a trajectory is ~10 % faster on it.  TMPC_COMPACT_MIN_B=0: the compact kernel for every launch (rounds 3-4)
        int fast_per_cu = 0, cus = 0;
        if (hipFuncSetAttribute((const void *)h->kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&fast_per_cu, (const void *)h->kernel, 64, h->lds_bytes) == hipSuccess && fast_per_cu > 0 &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) {
            h->kernel_small = h->kernel; h->lds_bytes_small = h->lds_bytes; h->cp_min_B = fast_per_cu * cus;
            if (const char *e = lab_env("TMPC_COMPACT_MIN_B")) h->cp_min_B = atoi(e);                                              // experiments
        }
        h->kernel = kc; h->compact = true;
        // padding of the packed rows' stage stride: only what keeps the residency (LDS is what bounds it: 8 x 20 KB at cfg 2)
        auto lds_cp = [&](int pad) { return sizeof(double) * (size_t)tmpc::lds_doubles_compact(d.N, d.n_lin, d.n_up + d.M, 64, pad, h->lay_cp); };
        auto per_cu_cp = [&](int pad) {
            int n = 0;
            if (hipFuncSetAttribute((const void *)kc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cp(pad)) != hipSuccess ||
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)kc, 64, lds_cp(pad)) != hipSuccess) return 0;
            return n;
        };
        const int per_cu0 = per_cu_cp(0);
        h->dpad_cp = tmpc::pick_d_pad(d.N, d.n_lin, d.n_up + d.M, 64, tmpc::DPAD_MAX, [&](int pad) { return pad == 0 || (per_cu0 > 0 && per_cu_cp(pad) == per_cu0); });
        h->lds_bytes = lds_cp(h->dpad_cp);
    }
    if (hipFuncSetAttribute((const void *)h->kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)h->lds_bytes) != hipSuccess)
        return fail(TMPC_ERR_NO_DEVICE);
    if (schur && h->fast && h->threads == 128 && !h->compact && (h->kernel_cp2 = tmpc::pick_compact2_kernel(d, &h->lay_cp2, &h->cp2_threads)) != nullptr) {
        {
            auto lds_cp2 = [&](int pad) { return sizeof(double) * (size_t)tmpc::lds_doubles_compact(d.N, d.n_lin, d.n_up + d.M, h->cp2_threads, pad, h->lay_cp2); };
            auto per_cu_cp2 = [&](int pad) {
                int n = 0;
                if (hipFuncSetAttribute((const void *)h->kernel_cp2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cp2(pad)) != hipSuccess ||
                    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)h->kernel_cp2, h->cp2_threads, lds_cp2(pad)) != hipSuccess) return 0;
                return n;
            };
            const int per_cu0 = per_cu_cp2(0);
            h->dpad_cp2 = tmpc::pick_d_pad(d.N, d.n_lin, d.n_up + d.M, h->cp2_threads, tmpc::DPAD_MAX, [&](int pad) { return pad == 0 || (per_cu0 > 0 && per_cu_cp2(pad) == per_cu0); });
            h->lds_bytes_cp2 = lds_cp2(h->dpad_cp2);
        }
        int per_cu = 0, fast_per_cu = 0, cus = 0;
        if (hipFuncSetAttribute((const void *)h->kernel_cp2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes_cp2) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)h->kernel_cp2, h->cp2_threads, h->lds_bytes_cp2) != hipSuccess || per_cu <= 0 ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&fast_per_cu, (const void *)h->kernel, 128, h->lds_bytes) != hipSuccess || fast_per_cu <= 0 ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0 || per_cu <= fast_per_cu)
            h->kernel_cp2 = nullptr;                         // (no gain in residency: the fast kernel stays alone)
        else {
            if (const char *e = lab_env("TMPC_COMPACT_PER_CU")) { const int v = atoi(e); if (v > 0 && v < per_cu) per_cu = v; }   // experiments
            h->grid_max = per_cu * cus; h->cp2_min_B = fast_per_cu * cus; h->prio_cp2 = per_cu * (h->cp2_threads / 64) == 8;
            if (const char *e = lab_env("TMPC_COMPACT2_MIN_B")) h->cp2_min_B = atoi(e);                                            // experiments
        }
    }
    if (h->compact) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)h->kernel, 64, h->lds_bytes) != hipSuccess || per_cu <= 0 ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0)
            return fail(TMPC_ERR_HIP);
        if (const char *e = lab_env("TMPC_COMPACT_PER_CU")) { const int v = atoi(e); if (v > 0 && v < per_cu) per_cu = v; }   // experiments
        h->grid_max = per_cu * cus; h->prio_cp = per_cu == 8;
    }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return fail(TMPC_ERR_HIP);
    const size_t N = d.N, B = B_max;
    bool ok = true;
    const size_t nxe = tmpc::ext_nx(d), nve = tmpc::ext_nv(d);
    {
        constexpr size_t TICK_SLAB_MAX = 2u << 20;
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t in_sz[4] = {B * nxe * 8, B * (N + 1) * nve * 8, B * N * d.npar * 8, B * 4};
        const size_t out_sz[8] = {B * (N + 1) * nxe * 8, B * N * tmpc::NU * 8, B * 8, B * 8, B * 4, B * 4, B * 4, B * 4};
        size_t in_off[5] = {0}, out_off[9] = {0};
        for (int i = 0; i < 4; i++) in_off[i + 1] = in_off[i] + up(in_sz[i]);
        for (int i = 0; i < 8; i++) out_off[i + 1] = out_off[i] + up(out_sz[i]);
        if (in_off[4] <= TICK_SLAB_MAX && out_off[8] <= TICK_SLAB_MAX) {
            h->slab_in_bytes = in_off[4]; h->slab_out_bytes = out_off[8];
            ok &= hipMalloc(&h->slab_in, in_off[4]) == hipSuccess && hipMalloc(&h->slab_out, out_off[8]) == hipSuccess;
            ok &= hipHostMalloc(&h->pin_in, in_off[4], hipHostMallocDefault) == hipSuccess && hipHostMalloc(&h->pin_out, out_off[8], hipHostMallocDefault) == hipSuccess;
            ok &= hipEventCreateWithFlags(&h->in_done, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&h->slot_done, hipEventDisableTiming) == hipSuccess;
            if (ok) {
                h->o_xinit = (double *)(h->slab_in + in_off[0]); h->o_x0 = (double *)(h->slab_in + in_off[1]); h->o_params = (double *)(h->slab_in + in_off[2]);
                h->d_slot = (int *)(h->slab_in + in_off[3]);
                h->xtraj = (double *)(h->slab_out + out_off[0]); h->utraj = (double *)(h->slab_out + out_off[1]); h->pobj = (double *)(h->slab_out + out_off[2]);
                h->res_eq = (double *)(h->slab_out + out_off[3]); h->exit_code = (int *)(h->slab_out + out_off[4]); h->qp_status = (int *)(h->slab_out + out_off[5]);
                h->sqp_iter = (int *)(h->slab_out + out_off[6]); h->qp_iter = (int *)(h->slab_out + out_off[7]);
            }
        } else {
            ok &= hipMalloc(&h->o_xinit, in_sz[0]) == hipSuccess;
            ok &= hipMalloc(&h->o_x0, in_sz[1]) == hipSuccess;
            ok &= hipMalloc(&h->o_params, in_sz[2]) == hipSuccess;
            ok &= hipMalloc(&h->xtraj, out_sz[0]) == hipSuccess;
            ok &= hipMalloc(&h->utraj, out_sz[1]) == hipSuccess;
            ok &= hipMalloc(&h->pobj, out_sz[2]) == hipSuccess;
            ok &= hipMalloc(&h->res_eq, out_sz[3]) == hipSuccess;
            ok &= hipMalloc(&h->exit_code, out_sz[4]) == hipSuccess;
            ok &= hipMalloc(&h->qp_status, out_sz[5]) == hipSuccess;
            ok &= hipMalloc(&h->sqp_iter, out_sz[6]) == hipSuccess;
            ok &= hipMalloc(&h->qp_iter, out_sz[7]) == hipSuccess;
        }
    }
    ok &= hipMalloc(&h->d_weight, B * 8) == hipSuccess;
    ok &= hipMalloc(&h->d_best, 4) == hipSuccess;
    ok &= hipMalloc(&h->d_disabled, B) == hipSuccess;
    if (h->compact || h->kernel_cp2) {
        ok &= hipMalloc(&h->ws, (size_t)h->grid_max * tmpc::ws_doubles(d.N, h->kernel_cp2 != nullptr && h->cp2_threads == 128) * 8) == hipSuccess;
        ok &= hipMalloc(&h->ticket, 8 * 4) == hipSuccess;         // one work counter per XCD (next_trajectory)
    }
    if (!ok) { tmpc_destroy(h); return TMPC_ERR_HIP; }
    *out = h;
    return TMPC_OK;
}

void tmpc_destroy(tmpc_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void *ptrs[] = {h->o_xinit, h->o_x0, h->o_params, h->xtraj, h->utraj, h->pobj, h->res_eq, h->d_weight,
                    h->exit_code, h->qp_status, h->sqp_iter, h->qp_iter, h->d_best, h->d_disabled,
                    h->st_z, h->st_pi, h->st_lamh, h->st_stopped, h->st_has, h->d_slot, h->d_share, h->scn_sample, h->scn_discard, h->ws, h->ticket};
    auto in_slab = [&](void *p) {
        return (h->slab_in && (char *)p >= h->slab_in && (char *)p < h->slab_in + h->slab_in_bytes) ||
               (h->slab_out && (char *)p >= h->slab_out && (char *)p < h->slab_out + h->slab_out_bytes);
    };
    for (void *p : ptrs) if (p && !in_slab(p)) (void)hipFree(p);
    if (h->slab_in) (void)hipFree(h->slab_in);
    if (h->slab_out) (void)hipFree(h->slab_out);
    if (h->pin_in) (void)hipHostFree(h->pin_in);
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    if (h->in_done) (void)hipEventDestroy(h->in_done);
    if (h->slot_done) (void)hipEventDestroy(h->slot_done);
    for (auto &e : h->ev) (void)hipEventDestroy(e);
    tmpc::lanes::destroy(h->lanes);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char *tmpc_last_error(const tmpc_handle *h) { return h ? h->err.c_str() : "null handle"; }

int tmpc_set_batch(tmpc_handle *h, int32_t B, const double *xinit, const double *x0, const double *params)
{
    if (!h || B <= 0 || B > h->B_max || !xinit || !x0 || !params) { if (h) h->err = "tmpc_set_batch: bad argument"; return TMPC_ERR_INVALID; }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const size_t N = h->d.N;
    const size_t in_bytes_B = (size_t)B * (tmpc::ext_nx(h->d) + (N + 1) * tmpc::ext_nv(h->d) + N * h->d.npar) * 8;
    if (h->slab_in && in_bytes_B <= (512u << 10)) {                // (measured: at 64 trajectories = 1.4 MB the runtime's own path for pageable memory is 12 us faster than a host copy into the mirror)
        // through the pinned mirror: one asynchronous copy of [xinit | x0 | params of the first B trajectories] (the regions of B_max trajectories lie in this order)
        if (h->in_pending) { TMPC_HIP_CHECK(h, hipEventSynchronize(h->in_done)); h->in_pending = false; }
        const size_t o0 = (char *)h->o_xinit - h->slab_in, o1 = (char *)h->o_x0 - h->slab_in, o2 = (char *)h->o_params - h->slab_in;
        const size_t n_par = (size_t)B * N * h->d.npar * 8;
        std::memcpy(h->pin_in + o0, xinit, (size_t)B * tmpc::ext_nx(h->d) * 8);
        std::memcpy(h->pin_in + o1, x0, (size_t)B * (N + 1) * tmpc::ext_nv(h->d) * 8);
        std::memcpy(h->pin_in + o2, params, n_par);
        if (o2 <= (256u << 10)) {                                   // (B_max close to B, as for a control tick's handle: the gaps cost less than two more copies)
            TMPC_HIP_CHECK(h, hipMemcpyAsync(h->slab_in, h->pin_in, o2 + n_par, hipMemcpyHostToDevice, h->stream));
        } else {
            TMPC_HIP_CHECK(h, hipMemcpyAsync(h->o_xinit, h->pin_in + o0, (size_t)B * tmpc::ext_nx(h->d) * 8, hipMemcpyHostToDevice, h->stream));
            TMPC_HIP_CHECK(h, hipMemcpyAsync(h->o_x0, h->pin_in + o1, (size_t)B * (N + 1) * tmpc::ext_nv(h->d) * 8, hipMemcpyHostToDevice, h->stream));
            TMPC_HIP_CHECK(h, hipMemcpyAsync(h->o_params, h->pin_in + o2, n_par, hipMemcpyHostToDevice, h->stream));
        }
        TMPC_HIP_CHECK(h, hipEventRecord(h->in_done, h->stream)); h->in_pending = true;
    } else {
    TMPC_HIP_CHECK(h, hipMemcpyAsync(h->o_xinit, xinit, (size_t)B * tmpc::ext_nx(h->d) * 8, hipMemcpyHostToDevice, h->stream));
    TMPC_HIP_CHECK(h, hipMemcpyAsync(h->o_x0, x0, (size_t)B * (N + 1) * tmpc::ext_nv(h->d) * 8, hipMemcpyHostToDevice, h->stream));
    TMPC_HIP_CHECK(h, hipMemcpyAsync(h->o_params, params, (size_t)B * N * h->d.npar * 8, hipMemcpyHostToDevice, h->stream));
    }
    h->xinit = h->o_xinit; h->x0 = h->o_x0; h->params = h->o_params; h->B = B;
    h->scn_B = 0;                      // new parameter rows: the scenario-row bookkeeping of the previous batch no longer describes them
    h->scn_discard_B = 0;
    h->share_B = 0;                    // ... nor does a parameter-sharing map given for them
    return TMPC_OK;
}

int tmpc_set_batch_device(tmpc_handle *h, int32_t B, const void *d_xinit, const void *d_x0, const void *d_params)
{
    if (!h || B <= 0 || B > h->B_max || !d_xinit || !d_x0 || !d_params) { if (h) h->err = "tmpc_set_batch_device: bad argument"; return TMPC_ERR_INVALID; }
    h->xinit = (const double *)d_xinit; h->x0 = (const double *)d_x0; h->params = (const double *)d_params; h->B = B;
    h->scn_B = 0; h->scn_discard_B = 0; h->share_B = 0;
    return TMPC_OK;
}

// One launch over the current batch: n_iter RTI iterations per trajectory + completeOneIteration.  st_flags: ST_* (0 = fresh
// solver instances from the batch's warm start, nothing kept or stored: Solver::solve() of a new capsule).
static int launch_solve(tmpc_handle *h, int n_iter, int st_flags)
{
    const bool rec = h->timing && h->ev_used + 2 <= (int)h->ev.size();
    if (rec) TMPC_HIP_CHECK(h, hipEventRecord(h->ev[h->ev_used], h->stream));
    if (h->throughput_mode) {
        // lane-per-trajectory variant: transpose the reference-layout inputs into the lane-major workspace, then one launch of
        // the scalar-per-lane SQP_RTI program; the workspace itself is the persistent state
        if (tmpc::lanes::stage_in(h->lanes, h->stream, h->B, h->xinit, h->x0, h->params, !(st_flags & tmpc::ST_KEEP_ITERATE),
                                  !(st_flags & tmpc::ST_KEEP_MULTIPLIERS), h->err)) return TMPC_ERR_HIP;
        if (tmpc::lanes::solve(h->lanes, h->stream, h->B, n_iter, (st_flags & tmpc::ST_STORE) != 0, (st_flags & tmpc::ST_COMPLETE) != 0,
                               h->xtraj, h->utraj, h->pobj,
                               h->exit_code, h->qp_status, h->sqp_iter, h->res_eq, h->qp_iter, h->err)) return TMPC_ERR_HIP;
    } else {
        tmpc::Dims dd = h->d;
        dd.n_sqp = n_iter;
        tmpc::StateIO io{h->st_z, h->st_pi, h->st_lamh, h->st_stopped, st_flags, h->ws, h->ticket, (h->slots_set && h->slots_B == h->B) ? h->d_slot : nullptr, h->st_has,
                         (h->share_B == h->B) ? h->d_share : nullptr};      // (a map given for another batch size is not applied)
        const bool lat3 = h->kernel_quad && h->latency_mode == 3;
        const bool lat2 = !lat3 && h->kernel_scan && h->latency_mode >= 2;        // (mode 3 without a four-wave variant runs as mode 2, ...
        const bool lat = !lat3 && !lat2 && h->kernel_lat && h->latency_mode != 0; //  ... mode 2 without a scan variant as the two-wave variant)
        // compact <-> fast kernels of a shape compute bit for bit the same, so the launch size may choose between them: the fast kernel while it
        // holds the whole launch resident (lower latency per trajectory), the compact one (twice the residency) above that
        const bool cp2 = h->kernel_cp2 && !lat && !lat2 && !lat3 && h->B > h->cp2_min_B;
        const bool small = h->compact && h->kernel_small && !lat && !lat2 && !lat3 && h->B <= h->cp_min_B;
        const bool cp = (h->compact && !lat && !lat2 && !lat3 && !small) || cp2;
        dd.prio = cp2 ? h->prio_cp2 : (cp ? h->prio_cp : false);
        dd.dpad = cp2 ? h->dpad_cp2 : (cp ? h->dpad_cp : 0);      // (layout only: results do not depend on it)
        if (cp) TMPC_HIP_CHECK(h, hipMemsetAsync(h->ticket, 0, 8 * 4, h->stream));    // the persistent launch's work counters (one per XCD)
        hipLaunchKernelGGL(lat3 ? h->kernel_quad : lat2 ? h->kernel_scan : lat ? h->kernel_lat : cp2 ? h->kernel_cp2 : small ? h->kernel_small : h->kernel,
                           dim3(cp ? (h->B < h->grid_max ? h->B : h->grid_max) : h->B),   // (persistent launch: at most the resident workgroups)
                           dim3(lat3 ? 256 : lat2 ? h->scan_threads : lat ? 128 : cp2 ? h->cp2_threads : ((cp || small) ? 64 : h->threads)),
                           lat3 ? h->lds_bytes_quad : lat2 ? h->lds_bytes_scan : lat ? h->lds_bytes_fast2 : cp2 ? h->lds_bytes_cp2 : small ? h->lds_bytes_small : h->lds_bytes, h->stream, dd, h->B,
                           h->xinit, h->x0, h->params, h->xtraj, h->utraj, h->pobj, h->exit_code, h->qp_status,
                           h->sqp_iter, h->res_eq, h->qp_iter, (long long *)nullptr, io);
        TMPC_HIP_CHECK(h, hipGetLastError());
        if (st_flags & tmpc::ST_COMPLETE) {
            // a failed solve resets the reference's capsule (Solver_acados_reset, acados_solver_interface.cpp:187-191): zero multipliers
            const int n_pi = (h->d.N + 1) * tmpc::NX, n_lam = h->d.N * (h->d.n_up + h->d.M);
            hipLaunchKernelGGL(tmpc::tmpc_state_finalize_kernel, dim3(h->B), dim3(64), 0, h->stream, n_pi, n_lam, h->exit_code, h->st_pi, h->st_lamh,
                               (h->slots_set && h->slots_B == h->B) ? h->d_slot : nullptr);
            TMPC_HIP_CHECK(h, hipGetLastError());
        }
    }
    if (rec) { TMPC_HIP_CHECK(h, hipEventRecord(h->ev[h->ev_used + 1], h->stream)); h->ev_used += 2; }
    return TMPC_OK;
}

// A caller that declared "the copies are not maintained" must never reach a kernel that reads them: the map has to be in force for the
// CURRENT batch (tmpc_set_batch* drops it) and the kernel family has to honour it.
static int share_check(tmpc_handle *h, const char *who)
{
    if (!h->share_strict) return TMPC_OK;
    if (h->share_B != h->B) { h->err = std::string(who) + ": the parameter-sharing map was declared with TMPC_SHARE_COPIES_NOT_MAINTAINED but is not in force for the "
                                       "current batch (tmpc_set_batch* drops it: give it again, or clear it with a null map)"; return TMPC_ERR_INVALID; }
    if (h->throughput_mode) { h->err = std::string(who) + ": the lane kernels read every entry's own parameter rows, the map says they are not maintained"; return TMPC_ERR_INVALID; }
    return TMPC_OK;
}

// the slots' persistent state no longer describes what the handle last solved
static int invalidate_state(tmpc_handle *h)
{
    h->st_valid = false; h->st_B = 0;
    if (h->st_has) TMPC_HIP_CHECK(h, hipMemsetAsync(h->st_has, 0, (size_t)h->B_max * 4, h->stream));
    return TMPC_OK;
}

int tmpc_solve(tmpc_handle *h)
{
    if (!h || h->B <= 0 || !h->xinit) { if (h) h->err = "tmpc_solve: no batch set"; return TMPC_ERR_INVALID; }
    if (int rc = share_check(h, "tmpc_solve")) return rc;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    if (int rc = invalidate_state(h)) return rc;
    return launch_solve(h, h->d.n_sqp, 0);
}

int tmpc_solve_iterations(tmpc_handle *h, int32_t n_iter, int32_t flags)
{
    if (!h || h->B <= 0 || !h->xinit || n_iter < 0 || (flags & ~15)) { if (h) h->err = "tmpc_solve_iterations: no batch set / bad argument"; return TMPC_ERR_INVALID; }
    if (int rc = share_check(h, "tmpc_solve_iterations")) return rc;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    if (h->throughput_mode && h->slots_set) { h->err = "tmpc_solve_iterations: slot maps (tmpc_set_slots) are not available with the lane kernels"; return TMPC_ERR_INVALID; }
    if (h->slots_set && h->slots_B != h->B) {
        // the map has one entry per batch entry: entries [slots_B, B) of a larger batch would be read from uninitialised memory
        h->err = "tmpc_solve_iterations: the slot map was given for a batch of " + std::to_string(h->slots_B) + " entries, the current batch has " +
                 std::to_string(h->B) + ": call tmpc_set_slots again after tmpc_set_batch (or clear it with a null map)";
        return TMPC_ERR_INVALID;
    }
    if (!h->throughput_mode && !h->st_z) {
        const size_t B = h->B_max, N = h->d.N, nh = h->d.n_up + h->d.M;
        const size_t sz[5] = {B * (N + 1) * tmpc::NV * 8, B * (N + 1) * tmpc::NX * 8, (B * N * nh + 1) * 8, B * 4, B * 4};
        void **dst[5] = {(void **)&h->st_z, (void **)&h->st_pi, (void **)&h->st_lamh, (void **)&h->st_stopped, (void **)&h->st_has};
        bool ok = true;
        for (int i = 0; i < 5 && ok; i++) ok = hipMalloc(dst[i], sz[i]) == hipSuccess && hipMemsetAsync(*dst[i], 0, sz[i], h->stream) == hipSuccess;
        if (!ok) {                          // all or nothing: a later call must not find half of the arrays
            for (int i = 0; i < 5; i++) { if (*dst[i]) (void)hipFree(*dst[i]); *dst[i] = nullptr; }
            h->err = "tmpc_solve_iterations: state allocation failed"; return TMPC_ERR_HIP;
        }
        h->st_valid = false; h->st_B = 0;
    }
    int st = tmpc::ST_STORE;
    if (h->throughput_mode) {
        // lane kernels keep their state per launch, not per slot: nothing to keep on the first call, and a grown batch starts fresh
        if (!h->st_valid) h->st_B = 0;
        if (h->B > h->st_B) h->st_valid = false;
        if (h->st_valid) {
            if (flags & TMPC_ITER_KEEP_ITERATE) st |= tmpc::ST_KEEP_ITERATE;
            if (flags & TMPC_ITER_KEEP_MULTIPLIERS) st |= tmpc::ST_KEEP_MULTIPLIERS;
        }
    } else {
        // wave kernels: the keep-flags apply per slot -- a slot without stored state (first call, grown batch, new slot of a map)
        // starts like a fresh capsule (slot_flags in the kernels)
        if (flags & TMPC_ITER_KEEP_ITERATE) st |= tmpc::ST_KEEP_ITERATE;
        if (flags & TMPC_ITER_KEEP_MULTIPLIERS) st |= tmpc::ST_KEEP_MULTIPLIERS;
    }
    if (flags & TMPC_ITER_COMPLETE) st |= tmpc::ST_COMPLETE;
    // a new solve() of the slots' Solvers: the "iteration loop has ended" marks belong to the previous solve (:105-106 is local to one solve())
    if ((flags & TMPC_ITER_NEW_SOLVE) && !h->throughput_mode && h->st_stopped) TMPC_HIP_CHECK(h, hipMemsetAsync(h->st_stopped, 0, (size_t)h->B_max * 4, h->stream));
    if ((flags & TMPC_ITER_NEW_SOLVE) && h->throughput_mode && h->lanes && tmpc::lanes::clear_stopped(h->lanes, h->stream, h->B_max, h->err)) return TMPC_ERR_HIP;
    const int rc = launch_solve(h, n_iter, st);
    if (rc == TMPC_OK && h->throughput_mode) { h->st_valid = true; if (h->B > h->st_B) h->st_B = h->B; }
    return rc;
}

int tmpc_reset_multipliers(tmpc_handle *h)
{
    if (!h) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    if (h->throughput_mode) {
        if (h->lanes && tmpc::lanes::reset_multipliers(h->lanes, h->stream, h->B_max, h->err)) return TMPC_ERR_HIP;
    } else if (h->st_pi) {
        const size_t B = h->B_max, N = h->d.N, nh = h->d.n_up + h->d.M;
        TMPC_HIP_CHECK(h, hipMemsetAsync(h->st_pi, 0, B * (N + 1) * tmpc::NX * 8, h->stream));
        TMPC_HIP_CHECK(h, hipMemsetAsync(h->st_lamh, 0, B * N * nh * 8, h->stream));
    }
    return TMPC_OK;
}

int tmpc_set_latency_mode(tmpc_handle *h, int32_t on)
{
    if (!h) return TMPC_ERR_INVALID;
    if (on < 0 || on > 3) return TMPC_ERR_INVALID;
    h->latency_mode = on;
    if (on == 3) return h->kernel_quad ? TMPC_OK : 1;              // 1: accepted, but this shape has no such variant (mode 3 then runs as mode 2, mode 2 as mode 1, if those exist)
    if (on == 2) return h->kernel_scan ? TMPC_OK : 1;
    return (on == 1 && !h->kernel_lat) ? 1 : TMPC_OK;
}

int tmpc_latency_mode_capacity(tmpc_handle *h, int32_t mode)
{
    if (!h || mode < 0 || mode > 3) return TMPC_ERR_INVALID;
    if (hipSetDevice(h->device) != hipSuccess) return TMPC_ERR_HIP;
    const void *k = nullptr; int threads = 64; size_t lds = 0;
    if (mode == 3) { if (!h->kernel_quad) return 0; k = (const void *)h->kernel_quad; threads = 256; lds = h->lds_bytes_quad; }
    else if (mode == 2) { if (!h->kernel_scan) return 0; k = (const void *)h->kernel_scan; threads = h->scan_threads; lds = h->lds_bytes_scan; }
    else if (mode == 1) { if (!h->kernel_lat) return 0; k = (const void *)h->kernel_lat; threads = 128; lds = h->lds_bytes_fast2; }
    else {
        // the throughput kernels of the handle: the resident set of the persistent (compact) launch, or of the plain kernel
        if (h->compact || h->kernel_cp2) return h->grid_max;
        k = (const void *)h->kernel; threads = h->threads; lds = h->lds_bytes;
    }
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, threads, lds) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess) { h->err = "tmpc_latency_mode_capacity: occupancy query failed"; return TMPC_ERR_HIP; }
    // variant 3 is built for ONE workgroup per CU (a wave on every SIMD): a second one fits (LDS, registers) but shares the SIMDs, and the launch is then
    // slower than variant 2's (measured, cfg 4's share of 8 = 512 trajectories: 1.79 ms against 1.31 ms, profiles/round6_cfg4_share8_*): its capacity is what
    // it serves well, not what the hardware would hold
    if (mode == 3 && per_cu > 1) per_cu = 1;
    return per_cu * cus;
}

int tmpc_set_slots(tmpc_handle *h, const int32_t *slots)
{
    if (!h) return TMPC_ERR_INVALID;
    if (!slots) { h->slots_set = false; h->slots_B = 0; return TMPC_OK; }
    if (h->B <= 0) { h->err = "tmpc_set_slots: set the batch first (the map has one entry per batch entry)"; return TMPC_ERR_INVALID; }
    std::vector<char> seen((size_t)h->B_max, 0);
    for (int b = 0; b < h->B; b++) {
        if (slots[b] < 0 || slots[b] >= h->B_max || seen[slots[b]]) { h->err = "tmpc_set_slots: slots must be distinct and in [0, B_max)"; return TMPC_ERR_INVALID; }
        seen[slots[b]] = 1;
    }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    if (h->slab_in) {                                               // pinned mirror: asynchronous, nothing to wait for (the caller's array is copied here)
        if (h->slot_pending) { TMPC_HIP_CHECK(h, hipEventSynchronize(h->slot_done)); h->slot_pending = false; }      // (not the batch copy just enqueued: it reads other bytes)
        const size_t o3 = (char *)h->d_slot - h->slab_in;
        std::memcpy(h->pin_in + o3, slots, (size_t)h->B * 4);
        TMPC_HIP_CHECK(h, hipMemcpyAsync(h->d_slot, h->pin_in + o3, (size_t)h->B * 4, hipMemcpyHostToDevice, h->stream));
        TMPC_HIP_CHECK(h, hipEventRecord(h->slot_done, h->stream)); h->slot_pending = true;
    } else {
    if (!h->d_slot) TMPC_HIP_CHECK(h, hipMalloc(&h->d_slot, (size_t)h->B_max * 4));
    TMPC_HIP_CHECK(h, hipMemcpyAsync(h->d_slot, slots, (size_t)h->B * 4, hipMemcpyHostToDevice, h->stream));
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));            // (the caller's array may go away)
    }
    h->slots_set = true; h->slots_B = h->B;
    return TMPC_OK;
}

int tmpc_set_param_sharing(tmpc_handle *h, const int32_t *base_of) { return tmpc_set_param_sharing_ex(h, base_of, 0); }

int tmpc_set_param_sharing_ex(tmpc_handle *h, const int32_t *base_of, int32_t flags)
{
    if (!h || (flags & ~TMPC_SHARE_COPIES_NOT_MAINTAINED)) return TMPC_ERR_INVALID;
    if (!base_of) { h->share_B = 0; h->share_strict = false; return TMPC_OK; }
#ifdef TMPC_GENERATED_STAGE
    // generated stage functions read every parameter -- halfspace rows included -- from ONE row block (tmpc_gen::rows has no notion of
    // "own" rows), so the hint cannot be honoured: as a pure hint it is accepted and ignored, as include/tmpc_hip.h says; a caller that
    // does not maintain the copies is refused
    if (flags & TMPC_SHARE_COPIES_NOT_MAINTAINED) { h->err = "tmpc_set_param_sharing_ex: generated solvers read every entry's own rows (copies must be maintained)"; return TMPC_ERR_INVALID; }
    h->share_B = 0;
    return TMPC_OK;
#endif
    if ((flags & TMPC_SHARE_COPIES_NOT_MAINTAINED) && h->throughput_mode) { h->err = "tmpc_set_param_sharing_ex: the lane kernels read every entry's own rows (copies must be maintained)"; return TMPC_ERR_INVALID; }
    if (h->B <= 0) { h->err = "tmpc_set_param_sharing: set the batch first (the map has one entry per batch entry)"; return TMPC_ERR_INVALID; }
    for (int b = 0; b < h->B; b++)
        if (base_of[b] < 0 || base_of[b] >= h->B) { h->err = "tmpc_set_param_sharing: entries must be batch indices in [0, B)"; return TMPC_ERR_INVALID; }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    if (!h->d_share) TMPC_HIP_CHECK(h, hipMalloc(&h->d_share, (size_t)h->B_max * 4));
    TMPC_HIP_CHECK(h, hipMemcpyAsync(h->d_share, base_of, (size_t)h->B * 4, hipMemcpyHostToDevice, h->stream));
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));            // (the caller's array may go away)
    h->share_B = h->B;
    h->share_strict = (flags & TMPC_SHARE_COPIES_NOT_MAINTAINED) != 0;
    return TMPC_OK;
}

int tmpc_copy_state(tmpc_handle *dst, tmpc_handle *src)
{
    if (!dst || !src || dst == src) return TMPC_ERR_INVALID;
    const tmpc::Dims &a = dst->d, &b = src->d;
    if (a.N != b.N || a.n_up != b.n_up || a.M != b.M || a.npar != b.npar || a.slack != b.slack || a.cost_model != b.cost_model || a.row_model != b.row_model ||
        dst->device != src->device || dst->throughput_mode || src->throughput_mode) {
        dst->err = "tmpc_copy_state: handles of different shape / device / kernel family"; return TMPC_ERR_INVALID;
    }
    if (!src->st_z) return TMPC_OK;                                 // nothing stored yet
    TMPC_HIP_CHECK(dst, hipSetDevice(dst->device));
    TMPC_HIP_CHECK(dst, hipStreamSynchronize(src->stream));
    if (!dst->st_z) {                                               // allocate through the regular path: an evaluation-only call on the (unset) batch is not possible, so inline it
        const size_t B = dst->B_max, N = a.N, nh = a.n_up + a.M;
        const size_t sz[5] = {B * (N + 1) * tmpc::NV * 8, B * (N + 1) * tmpc::NX * 8, (B * N * nh + 1) * 8, B * 4, B * 4};
        void **p[5] = {(void **)&dst->st_z, (void **)&dst->st_pi, (void **)&dst->st_lamh, (void **)&dst->st_stopped, (void **)&dst->st_has};
        bool ok = true;
        for (int i = 0; i < 5 && ok; i++) ok = hipMalloc(p[i], sz[i]) == hipSuccess && hipMemsetAsync(*p[i], 0, sz[i], dst->stream) == hipSuccess;
        if (!ok) { for (int i = 0; i < 5; i++) { if (*p[i]) (void)hipFree(*p[i]); *p[i] = nullptr; } dst->err = "tmpc_copy_state: allocation failed"; return TMPC_ERR_HIP; }
    }
    const size_t n = (size_t)(dst->B_max < src->B_max ? dst->B_max : src->B_max), N = a.N, nh = a.n_up + a.M;
    TMPC_HIP_CHECK(dst, hipMemcpyAsync(dst->st_z, src->st_z, n * (N + 1) * tmpc::NV * 8, hipMemcpyDeviceToDevice, dst->stream));
    TMPC_HIP_CHECK(dst, hipMemcpyAsync(dst->st_pi, src->st_pi, n * (N + 1) * tmpc::NX * 8, hipMemcpyDeviceToDevice, dst->stream));
    TMPC_HIP_CHECK(dst, hipMemcpyAsync(dst->st_lamh, src->st_lamh, n * N * nh * 8, hipMemcpyDeviceToDevice, dst->stream));
    TMPC_HIP_CHECK(dst, hipMemcpyAsync(dst->st_stopped, src->st_stopped, n * 4, hipMemcpyDeviceToDevice, dst->stream));
    TMPC_HIP_CHECK(dst, hipMemcpyAsync(dst->st_has, src->st_has, n * 4, hipMemcpyDeviceToDevice, dst->stream));
    TMPC_HIP_CHECK(dst, hipStreamSynchronize(dst->stream));
    return TMPC_OK;
}

int tmpc_clear_slot(tmpc_handle *h, int32_t slot)
{
    if (!h || slot < 0 || slot >= h->B_max) { if (h) h->err = "tmpc_clear_slot: slot out of range"; return TMPC_ERR_INVALID; }
    if (h->throughput_mode) { h->err = "tmpc_clear_slot: the lane kernels keep their state per launch, not per slot"; return TMPC_ERR_INVALID; }
    if (!h->st_has) return TMPC_OK;                                 // nothing stored yet: every slot is fresh
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    TMPC_HIP_CHECK(h, hipMemsetAsync(h->st_has + slot, 0, 4, h->stream));
    TMPC_HIP_CHECK(h, hipMemsetAsync(h->st_stopped + slot, 0, 4, h->stream));
    return TMPC_OK;
}

__global__ __launch_bounds__(256) void tmpc_poison_lds_kernel(int n_doubles)
{
    extern __shared__ __attribute__((aligned(16))) double poison_smem[];
    const unsigned long long pat = 0x7ff4dead0000beefull;                 // a signalling NaN
    for (int i = threadIdx.x; i < n_doubles; i += blockDim.x) poison_smem[i] = __builtin_bit_cast(double, pat);
    __syncthreads();
    if (poison_smem[(threadIdx.x * 97) % n_doubles] == 0.0) poison_smem[0] = 1.0;      // (keeps the stores alive)
}

int tmpc_has_lab_switches(void)
{
#ifdef TMPC_LAB_SWITCHES
    return 1;
#else
    return 0;
#endif
}

int tmpc_debug_poison_lds(tmpc_handle *h)
{
    if (!h) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    int cus = 0;
    TMPC_HIP_CHECK(h, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
    const int bytes = 160 * 1024;
    TMPC_HIP_CHECK(h, hipFuncSetAttribute((const void *)tmpc_poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    for (int round = 0; round < 4; round++) {                               // (one workgroup per CU fits at a time: a few rounds reach every CU whatever the dispatch order)
        hipLaunchKernelGGL(tmpc_poison_lds_kernel, dim3(cus * 2), dim3(256), bytes, h->stream, bytes / 8);
        TMPC_HIP_CHECK(h, hipGetLastError());
    }
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    return TMPC_OK;
}

int tmpc_debug_lds_passes(int32_t N, int32_t n_pair, int32_t nh, int32_t threads, int32_t dstride)
{
    if (N < 1 || N > 64 || nh < 0 || n_pair < 0 || n_pair > nh || (threads != 64 && threads != 128) || dstride < 2 * n_pair + 3 * (nh - n_pair)) return TMPC_ERR_INVALID;
    return tmpc::d_load_passes(N, n_pair, nh, threads, dstride);
}

int tmpc_has_lane_kernels(void)
{
#ifdef TMPC_WITH_LANES
    return 1;
#else
    return 0;
#endif
}

int tmpc_set_throughput_mode(tmpc_handle *h, int32_t on)
{
    if (!h) return TMPC_ERR_INVALID;
    if (on && tmpc::stage_model(h->d) != 0) { h->err = "tmpc_set_throughput_mode: the lane kernels have the MPCC contouring cost and ellipsoid rows only"; return TMPC_ERR_INVALID; }
    if (on && h->d.model != 0) { h->err = "tmpc_set_throughput_mode: the lane kernels integrate the contouring model (spline state) only"; return TMPC_ERR_INVALID; }
    if (on && h->share_strict) { h->err = "tmpc_set_throughput_mode: a parameter-sharing map with TMPC_SHARE_COPIES_NOT_MAINTAINED is registered and the lane kernels read every entry's own rows"; return TMPC_ERR_INVALID; }
    if (on && !h->lanes) {
        TMPC_HIP_CHECK(h, hipSetDevice(h->device));
        h->lanes = tmpc::lanes::create(h->d, h->B_max, h->err);
        if (!h->lanes) return TMPC_ERR_HIP;
    }
    if (h->throughput_mode != (on != 0)) { if (int rc = invalidate_state(h)) return rc; }      // the two kernel families keep their persistent state separately
    h->throughput_mode = on != 0;
    return TMPC_OK;
}

int tmpc_synchronize(tmpc_handle *h)
{
    if (!h) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    return TMPC_OK;
}

int tmpc_get(tmpc_handle *h, double *xtraj, double *utraj, double *pobj, int32_t *exit_code, int32_t *qp_status,
             int32_t *sqp_iter, double *res_eq, int32_t *qp_iter_total)
{
    if (!h || h->B <= 0) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const size_t N = h->d.N, B = h->B;
    if (h->slab_out) {                                              // one D2H copy of the output slab into its pinned mirror, then host copies of what was asked for
        if (h->slab_out_bytes <= (256u << 10)) {
            TMPC_HIP_CHECK(h, hipMemcpyAsync(h->pin_out, h->slab_out, h->slab_out_bytes, hipMemcpyDeviceToHost, h->stream));
        } else {                                                    // B_max well above a tick's size: the first B entries of every array the caller asked for
            auto dc = [&](const void *want, const void *dev, size_t n) {
                return want ? hipMemcpyAsync(h->pin_out + ((const char *)dev - h->slab_out), dev, n, hipMemcpyDeviceToHost, h->stream) : hipSuccess;
            };
            TMPC_HIP_CHECK(h, dc(xtraj, h->xtraj, B * (N + 1) * tmpc::ext_nx(h->d) * 8)); TMPC_HIP_CHECK(h, dc(utraj, h->utraj, B * N * tmpc::NU * 8));
            TMPC_HIP_CHECK(h, dc(pobj, h->pobj, B * 8)); TMPC_HIP_CHECK(h, dc(res_eq, h->res_eq, B * 8));
            TMPC_HIP_CHECK(h, dc(exit_code, h->exit_code, B * 4)); TMPC_HIP_CHECK(h, dc(qp_status, h->qp_status, B * 4));
            TMPC_HIP_CHECK(h, dc(sqp_iter, h->sqp_iter, B * 4)); TMPC_HIP_CHECK(h, dc(qp_iter_total, h->qp_iter, B * 4));
        }
        TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));
        auto hc = [&](void *dst, const void *dev, size_t n) { if (dst) std::memcpy(dst, h->pin_out + ((const char *)dev - h->slab_out), n); };
        hc(xtraj, h->xtraj, B * (N + 1) * tmpc::ext_nx(h->d) * 8); hc(utraj, h->utraj, B * N * tmpc::NU * 8);
        hc(pobj, h->pobj, B * 8); hc(res_eq, h->res_eq, B * 8);
        hc(exit_code, h->exit_code, B * 4); hc(qp_status, h->qp_status, B * 4); hc(sqp_iter, h->sqp_iter, B * 4); hc(qp_iter_total, h->qp_iter, B * 4);
        return TMPC_OK;
    }
    auto cp = [&](void *dst, const void *src, size_t n) { return dst ? hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, h->stream) : hipSuccess; };
    TMPC_HIP_CHECK(h, cp(xtraj, h->xtraj, B * (N + 1) * tmpc::ext_nx(h->d) * 8));
    TMPC_HIP_CHECK(h, cp(utraj, h->utraj, B * N * tmpc::NU * 8));
    TMPC_HIP_CHECK(h, cp(pobj, h->pobj, B * 8));
    TMPC_HIP_CHECK(h, cp(res_eq, h->res_eq, B * 8));
    TMPC_HIP_CHECK(h, cp(exit_code, h->exit_code, B * 4));
    TMPC_HIP_CHECK(h, cp(qp_status, h->qp_status, B * 4));
    TMPC_HIP_CHECK(h, cp(sqp_iter, h->sqp_iter, B * 4));
    TMPC_HIP_CHECK(h, cp(qp_iter_total, h->qp_iter, B * 4));
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    return TMPC_OK;
}

int tmpc_select_best(tmpc_handle *h, int32_t first, int32_t count, const double *weight, const uint8_t *disabled, int32_t *best)
{
    if (!h || !best || first < 0 || count <= 0 || first + count > h->B) { if (h) h->err = "tmpc_select_best: bad range"; return TMPC_ERR_INVALID; }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    if (weight) TMPC_HIP_CHECK(h, hipMemcpyAsync(h->d_weight, weight, (size_t)count * 8, hipMemcpyHostToDevice, h->stream));
    if (disabled) TMPC_HIP_CHECK(h, hipMemcpyAsync(h->d_disabled, disabled, (size_t)count, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(tmpc::tmpc_select_best_kernel, dim3(1), dim3(256), 0, h->stream, first, count, h->pobj, h->exit_code,
                       weight ? h->d_weight : nullptr, disabled ? h->d_disabled : nullptr, h->d_best);
    TMPC_HIP_CHECK(h, hipGetLastError());
    TMPC_HIP_CHECK(h, hipMemcpyAsync(best, h->d_best, 4, hipMemcpyDeviceToHost, h->stream));
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    return TMPC_OK;
}

int tmpc_get_stream(tmpc_handle *h, void **stream)
{
    if (!h || !stream) return TMPC_ERR_INVALID;
    *stream = (void *)h->stream;
    return TMPC_OK;
}

int tmpc_kernel_info(const tmpc_handle *h, char *buf, int32_t capacity)
{
    if (!h || !buf || capacity <= 0) return TMPC_ERR_INVALID;
    const char *family = h->throughput_mode ? "lanes (one lane per trajectory)"
                         : !h->fast        ? "generic (one wave per trajectory, rows in LDS)"
                         : !h->compact     ? (h->threads == 128 ? "fast, two waves per trajectory" : "fast (one wave per trajectory)")
                                           : "compact (one wave per trajectory, two waves per SIMD)";
    const std::string sm = (h->compact && h->kernel_small && h->cp_min_B > 0) ? "; launches of at most " + std::to_string(h->cp_min_B) + " trajectories: fast one-wave variant (LDS " +
                                                                                std::to_string(h->lds_bytes_small) + " B, one workgroup per trajectory)" : "";
    const std::string cp2 = h->kernel_cp2 ? "; launches of more than " + std::to_string(h->cp2_min_B) + " trajectories: compact two-wave variant (LDS " +
                                            std::to_string(h->lds_bytes_cp2) + " B, persistent launch, resident workgroups " + std::to_string(h->grid_max) + ")" : "";
    const int n = snprintf(buf, (size_t)capacity, "%s; trajectories per workgroup %d; LDS %zu B per workgroup; %s%s", family, 1,
                           h->lds_bytes, h->compact ? (std::string("persistent launch, resident workgroups ") + std::to_string(h->grid_max)).c_str()
                                                    : "one workgroup per trajectory", (cp2 + sm).c_str());
    return n < capacity ? n : capacity - 1;
}

int tmpc_result_device_ptrs(tmpc_handle *h, void **d_pobj, void **d_exit_code)
{
    if (!h) return TMPC_ERR_INVALID;
    if (d_pobj) *d_pobj = h->pobj;
    if (d_exit_code) *d_exit_code = h->exit_code;
    return TMPC_OK;
}

int tmpc_pack_records(tmpc_handle *h, void *d_records, const void *d_guidance_id, const void *d_weight)
{
    if (!h || h->B <= 0 || !d_records) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(tmpc::tmpc_pack_records_kernel, dim3((h->B + 255) / 256), dim3(256), 0, h->stream, h->B, h->pobj,
                       h->exit_code, (const int *)d_guidance_id, (const double *)d_weight, (tmpc_record *)d_records);
    TMPC_HIP_CHECK(h, hipGetLastError());
    return TMPC_OK;
}

int tmpc_select_best_records(tmpc_handle *h, const void *d_records, int32_t n_ranks, int32_t n_scenes, int32_t per_rank, void *d_best)
{
    if (!h || !d_records || !d_best || n_ranks <= 0 || n_scenes <= 0 || per_rank <= 0) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(tmpc::tmpc_select_best_records_kernel, dim3(n_scenes), dim3(64), 0, h->stream,
                       (const tmpc_record *)d_records, n_ranks, n_scenes, per_rank, (int *)d_best);
    TMPC_HIP_CHECK(h, hipGetLastError());
    return TMPC_OK;
}

int tmpc_gather_best(tmpc_handle *h, const void *d_best, int32_t n_sets, int32_t set_size, int32_t index_offset, void *d_xtraj, void *d_utraj)
{
    if (!h || !d_best || !d_xtraj || !d_utraj || n_sets <= 0 || set_size <= 0 || index_offset < 0 || (int64_t)n_sets * set_size > h->B) {
        if (h) h->err = "tmpc_gather_best: bad argument (n_sets x set_size entries of the current batch)";
        return TMPC_ERR_INVALID;
    }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(tmpc::tmpc_gather_best_kernel, dim3(n_sets), dim3(64), 0, h->stream, (const int *)d_best, set_size, index_offset,
                       (h->d.N + 1) * tmpc::ext_nx(h->d), h->d.N * tmpc::NU, h->xtraj, h->utraj, (double *)d_xtraj, (double *)d_utraj);
    TMPC_HIP_CHECK(h, hipGetLastError());
    return TMPC_OK;
}

int tmpc_linearize_topology_ex(tmpc_handle *h, const void *d_obstacle_pos, int32_t n_obstacles, const void *d_obstacle_radius,
                               const void *d_static_halfspaces, int32_t n_static, const void *d_scene_of, const void *d_state_x,
                               double robot_radius, const void *d_is_original)
{
#ifdef TMPC_GENERATED_STAGE
    if (h) h->err = "tmpc_linearize_topology: not available in a generated solver (its parameter layout is the module stack's)";
    return TMPC_ERR_INVALID;
#endif
    if (!h || h->B <= 0 || !h->params || !d_scene_of || !d_state_x || h->d.n_lin <= 0 || n_obstacles < 0 || n_static < 0 ||
        n_obstacles + n_static > h->d.n_lin || (n_obstacles > 0 && !d_obstacle_pos) || (n_static > 0 && !d_static_halfspaces)) {
        if (h) h->err = "tmpc_linearize_topology: bad argument / no batch / more obstacle + static rows than the problem's topology rows";
        return TMPC_ERR_INVALID;
    }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const int n = h->B * h->d.N;
    hipLaunchKernelGGL(tmpc::tmpc_linearize_topology_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d, h->B,
                       h->x0, const_cast<double *>(h->params), (const double *)d_obstacle_pos, (const int *)d_scene_of,
                       (const double *)d_state_x, robot_radius, (const uint8_t *)d_is_original, n_obstacles, (const double *)d_obstacle_radius,
                       (const double *)d_static_halfspaces, n_static);
    TMPC_HIP_CHECK(h, hipGetLastError());
    return TMPC_OK;
}

int tmpc_linearize_topology(tmpc_handle *h, const void *d_obstacle_pos, const void *d_scene_of, const void *d_state_x,
                            double robot_radius, const void *d_is_original)
{
    if (!h || !d_obstacle_pos) { if (h) h->err = "tmpc_linearize_topology: bad argument"; return TMPC_ERR_INVALID; }
    return tmpc_linearize_topology_ex(h, d_obstacle_pos, h->d.n_lin, nullptr, nullptr, 0, d_scene_of, d_state_x, robot_radius, d_is_original);
}

int tmpc_scenario_halfspaces(tmpc_handle *h, const void *d_samples, int32_t n_pts, int32_t n_rows, const void *d_scene_of,
                             const void *d_state_x, double radius, double disc_offset)
{
#ifdef TMPC_GENERATED_STAGE
    if (h) h->err = "tmpc_scenario_halfspaces: not available in a generated solver (its parameter layout is the module stack's)";
    return TMPC_ERR_INVALID;
#endif
    if (!h || h->B <= 0 || !h->params || !d_samples || !d_scene_of || !d_state_x || n_pts <= 0 || n_rows <= 0 || n_rows > 64 ||
        n_rows > h->d.n_slk) {
        if (h) h->err = "tmpc_scenario_halfspaces: bad argument / no batch / more rows than the problem's slack rows";
        return TMPC_ERR_INVALID;
    }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const size_t per_entry = 3 * sizeof(double) + sizeof(int);                        // a candidate: normal, margin, index word
    hipFuncAttributes fa;
    TMPC_HIP_CHECK(h, hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(tmpc::tmpc_scenario_halfspaces_kernel)));
    if ((size_t)n_pts * per_entry + fa.sharedSizeBytes > 160 * 1024) {      // the second pass's list (room for every sample) + the kernel's static tables
        h->err = "tmpc_scenario_halfspaces: that many samples per stage do not fit a workgroup's LDS (160 KiB minus the kernel's static tables: about 5480)";
        return TMPC_ERR_INVALID;
    }
    const size_t units = (size_t)h->B * h->d.N;
    const size_t need = units * n_rows + 1 + units + (size_t)h->B;  // rows' samples, the first pass's overflow list (count, units), empty-polygon stages per trajectory
    if (need > h->scn_cap) {
        if (h->scn_sample) { TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->scn_sample); h->scn_sample = nullptr; h->scn_cap = 0; }
        TMPC_HIP_CHECK(h, hipMalloc(&h->scn_sample, need * sizeof(int)));
        h->scn_cap = need;
    }
    h->scn_rows = n_rows; h->scn_B = h->B;
    // scenarios discarded for this batch (tmpc_scenario_discard after the batch was set) are left out of the polygons
    const bool use_discard = h->scn_discard && h->scn_discard_B == h->B && h->scn_discard_S > 0 && n_pts % h->scn_discard_S == 0;
    // first pass with a short candidate list (more workgroups per CU); second pass, with room for every sample, only for the
    // units the first pass recorded as not fitting
    int *overflow = h->scn_sample + units * n_rows;
    int *empty_stages = overflow + 1 + units;
    TMPC_HIP_CHECK(h, hipMemsetAsync(overflow, 0, sizeof(int), h->stream));
    TMPC_HIP_CHECK(h, hipMemsetAsync(empty_stages, 0, sizeof(int) * (size_t)h->B, h->stream));
    int list_cap = tmpc::POLY_LIST_CAP;
    if (const char *e = lab_env("TMPC_POLY_LIST_CAP")) { const int v = atoi(e); if (v >= 64 && v <= 4096) list_cap = v; }   // experiments
    const int cap1 = n_pts < list_cap ? n_pts : list_cap;
    for (int pass = 0; pass < (cap1 < n_pts ? 2 : 1); pass++) {
        const int cap = pass == 0 ? cap1 : n_pts;
        const size_t lds = (size_t)cap * per_entry;
        if (lds > 48 * 1024)
            TMPC_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(tmpc::tmpc_scenario_halfspaces_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(tmpc::tmpc_scenario_halfspaces_kernel, dim3(pass == 0 ? units : (units < 512 ? units : 512)), dim3(256), lds, h->stream, h->d, h->B, h->x0,
                           const_cast<double *>(h->params), (const double *)d_samples, n_pts, n_rows, (const int *)d_scene_of,
                           (const double *)d_state_x, radius, disc_offset, h->scn_sample, cap, overflow, pass, empty_stages,
                           use_discard ? h->scn_discard : nullptr, use_discard ? h->scn_discard_S : 1);
    }
    TMPC_HIP_CHECK(h, hipGetLastError());
    return TMPC_OK;
}

int tmpc_sample_scenarios(tmpc_handle *h, const void *d_pred, const void *d_prob, int32_t n_solvers, int32_t n_obstacles, int32_t n_modes,
                          int32_t n_scenarios, uint64_t seed, void *d_samples)
{
    if (!h || !d_pred || !d_prob || !d_samples || n_solvers <= 0 || n_obstacles <= 0 || n_modes <= 0 || n_scenarios <= 0) {
        if (h) h->err = "tmpc_sample_scenarios: bad argument";
        return TMPC_ERR_INVALID;
    }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const int n = n_solvers * n_obstacles * n_scenarios;
    hipLaunchKernelGGL(tmpc::tmpc_sample_scenarios_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d.N, n_solvers, n_obstacles, n_modes,
                       n_scenarios, (unsigned long long)seed, (const double *)d_pred, (const double *)d_prob, (double *)d_samples);
    TMPC_HIP_CHECK(h, hipGetLastError());
    return TMPC_OK;
}

int tmpc_scenario_discard(tmpc_handle *h, const void *d_samples, int32_t n_pts, int32_t n_scenarios, int32_t n_discard, const void *d_scene_of, double radius)
{
    if (!h || h->B <= 0 || !h->x0 || !d_samples || !d_scene_of || n_pts <= 0 || n_scenarios <= 0 || n_pts % n_scenarios != 0 || n_discard < 0 ||
        n_discard >= n_scenarios || n_scenarios > 16384) {
        if (h) h->err = "tmpc_scenario_discard: bad argument / no batch (n_pts = obstacles x n_scenarios, 0 <= n_discard < n_scenarios <= 16384)";
        return TMPC_ERR_INVALID;
    }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const size_t need = (size_t)h->B_max * n_scenarios;
    if (need > h->scn_discard_cap) {
        if (h->scn_discard) { TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->scn_discard); h->scn_discard = nullptr; h->scn_discard_cap = 0; }
        TMPC_HIP_CHECK(h, hipMalloc(&h->scn_discard, need));
        h->scn_discard_cap = need;
    }
    const size_t lds = (size_t)n_scenarios * sizeof(double);
    if (lds > 48 * 1024)
        TMPC_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(tmpc::tmpc_scenario_discard_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(tmpc::tmpc_scenario_discard_kernel, dim3(h->B), dim3(256), lds, h->stream, h->d, h->B, h->x0, (const double *)d_samples, n_pts,
                       n_scenarios, (const int *)d_scene_of, radius, n_discard, h->scn_discard);
    TMPC_HIP_CHECK(h, hipGetLastError());
    h->scn_discard_S = n_scenarios; h->scn_discard_B = h->B; h->scn_discard_n = n_discard;
    return TMPC_OK;
}

int tmpc_scenario_discarded(tmpc_handle *h, void *d_mask)
{
    if (!h || !d_mask || !h->scn_discard || h->scn_discard_B != h->B) { if (h) h->err = "tmpc_scenario_discarded: no discard set for the current batch"; return TMPC_ERR_INVALID; }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    TMPC_HIP_CHECK(h, hipMemcpyAsync(d_mask, h->scn_discard, (size_t)h->B * h->scn_discard_S, hipMemcpyDeviceToDevice, h->stream));
    return TMPC_OK;
}

int tmpc_scenario_empty_stages(tmpc_handle *h, void *d_count)
{
    if (!h || !d_count) return TMPC_ERR_INVALID;
    if (!h->scn_sample || h->scn_B != h->B || h->B <= 0) { h->err = "tmpc_scenario_empty_stages: the scenario rows of the current batch were not built by tmpc_scenario_halfspaces"; return TMPC_ERR_INVALID; }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const size_t units = (size_t)h->B * h->d.N;
    TMPC_HIP_CHECK(h, hipMemcpyAsync(d_count, h->scn_sample + units * h->scn_rows + 1 + units, sizeof(int) * (size_t)h->B, hipMemcpyDeviceToDevice, h->stream));
    return TMPC_OK;
}

int tmpc_scenario_support(tmpc_handle *h, int32_t n_scenarios, double tol, void *d_support, void *d_active_rows)
{
    if (!h || !d_support || n_scenarios <= 0 || n_scenarios > 8192 || !(tol >= 0.0)) {
        if (h) h->err = "tmpc_scenario_support: bad argument (1 <= n_scenarios <= 8192, tol >= 0)";
        return TMPC_ERR_INVALID;
    }
    if (!h->scn_sample || h->scn_B != h->B || h->B <= 0 || !h->params) {
        h->err = "tmpc_scenario_support: the scenario rows of the current batch were not built by tmpc_scenario_halfspaces (call it after tmpc_set_batch)";
        return TMPC_ERR_INVALID;
    }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(tmpc::tmpc_scenario_support_kernel, dim3(h->B), dim3(64), 0, h->stream, h->d, h->B, h->params, h->xtraj,
                       h->scn_sample, h->scn_rows, n_scenarios, tol, (int *)d_support, (int *)d_active_rows);
    TMPC_HIP_CHECK(h, hipGetLastError());
    return TMPC_OK;
}

int tmpc_warmstart(tmpc_handle *h, const void *d_state, const void *d_mode, const void *d_src, double deceleration)
{
    if (!h || h->B <= 0 || !h->x0 || !h->xinit || !d_state) { if (h) h->err = "tmpc_warmstart: bad argument / no batch"; return TMPC_ERR_INVALID; }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const int n = h->B * (h->d.N + 1);
    hipLaunchKernelGGL(tmpc::tmpc_warmstart_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d, h->B, (const double *)d_state,
                       (const int *)d_mode, (const int *)d_src, h->xtraj, h->utraj, const_cast<double *>(h->x0),
                       const_cast<double *>(h->xinit), deceleration);
    TMPC_HIP_CHECK(h, hipGetLastError());
    return TMPC_OK;
}

int tmpc_init_with_guidance(tmpc_handle *h, const void *d_gpos, const void *d_gvel, const void *d_enabled)
{
    if (!h || h->B <= 0 || !h->x0 || !d_gpos || !d_gvel) { if (h) h->err = "tmpc_init_with_guidance: bad argument / no batch"; return TMPC_ERR_INVALID; }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const int n = h->B * (h->d.N + 1);
    hipLaunchKernelGGL(tmpc::tmpc_init_with_guidance_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d, h->B,
                       (const double *)d_gpos, (const double *)d_gvel, (const uint8_t *)d_enabled, const_cast<double *>(h->x0));
    TMPC_HIP_CHECK(h, hipGetLastError());
    return TMPC_OK;
}

int tmpc_debug_get_x0(tmpc_handle *h, double *x0, double *xinit)
{
    if (!h || h->B <= 0 || !h->x0) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    if (x0) TMPC_HIP_CHECK(h, hipMemcpy(x0, h->x0, (size_t)h->B * (h->d.N + 1) * tmpc::ext_nv(h->d) * 8, hipMemcpyDeviceToHost));
    if (xinit) TMPC_HIP_CHECK(h, hipMemcpy(xinit, h->xinit, (size_t)h->B * tmpc::ext_nx(h->d) * 8, hipMemcpyDeviceToHost));
    return TMPC_OK;
}

int tmpc_enable_timing(tmpc_handle *h, int32_t max_records)
{
    if (!h || max_records < 0) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    for (auto &e : h->ev) (void)hipEventDestroy(e);
    h->ev.clear(); h->ev_used = 0; h->timing = max_records > 0;
    h->ev.resize(2 * (size_t)max_records);
    for (auto &e : h->ev) TMPC_HIP_CHECK(h, hipEventCreate(&e));
    return TMPC_OK;
}

int tmpc_get_timings(tmpc_handle *h, float *ms, int32_t capacity, int32_t *n_out)
{
    if (!h || !ms || !n_out) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    int n = h->ev_used / 2; if (n > capacity) n = capacity;
    for (int i = 0; i < n; i++) TMPC_HIP_CHECK(h, hipEventElapsedTime(&ms[i], h->ev[2 * i], h->ev[2 * i + 1]));
    *n_out = n; h->ev_used = 0;
    return TMPC_OK;
}

int tmpc_time_solve(tmpc_handle *h, int32_t reps, float *ms_each)
{
    if (!h || reps <= 0 || !ms_each) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    Events evs; evs.ev.assign(2 * (size_t)reps, nullptr);
    std::vector<hipEvent_t> &ev = evs.ev;
    for (auto &e : ev) TMPC_HIP_CHECK(h, hipEventCreate(&e));
    for (int i = 0; i < reps; i++) {
        TMPC_HIP_CHECK(h, hipEventRecord(ev[2 * i], h->stream));
        int rc = tmpc_solve(h);
        if (rc != TMPC_OK) return rc;
        TMPC_HIP_CHECK(h, hipEventRecord(ev[2 * i + 1], h->stream));
    }
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < reps; i++) TMPC_HIP_CHECK(h, hipEventElapsedTime(&ms_each[i], ev[2 * i], ev[2 * i + 1]));
    return TMPC_OK;
}

int tmpc_debug_get_params(tmpc_handle *h, double *params)
{
    if (!h || h->B <= 0 || !params || !h->params) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    TMPC_HIP_CHECK(h, hipMemcpy(params, h->params, (size_t)h->B * h->d.N * h->d.npar * 8, hipMemcpyDeviceToHost));
    return TMPC_OK;
}

int tmpc_debug_profile(tmpc_handle *h, int64_t *cycles, int32_t n_phases)
{
    if (!h || h->B <= 0 || !cycles || n_phases < tmpc::PH_COUNT) return TMPC_ERR_INVALID;
    if (h->share_strict) { h->err = "tmpc_debug_profile: the profiled twins read every entry's own parameter rows; a map with TMPC_SHARE_COPIES_NOT_MAINTAINED is registered"; return TMPC_ERR_INVALID; }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    DevBufs bufs;
    double *dp_ = nullptr;
    const size_t n = (size_t)h->B * tmpc::PH_COUNT;
    TMPC_HIP_CHECK(h, bufs.alloc(&dp_, n * 8));
    long long *dp = (long long *)dp_;
    TMPC_HIP_CHECK(h, hipMemset(dp, 0, n * 8));
    tmpc::SolveKernel pk = h->kernel;                   // the generic kernel profiles itself; fast shapes have an instrumented twin
    int thr = h->threads;
    size_t lds = h->fast ? h->lds_bytes_fast : h->lds_bytes;
    if (h->fast && h->threads == 128) lds = h->lds_bytes_fast2;
    if (h->fast && tmpc::stage_model(h->d) != 0) { h->err = "tmpc_debug_profile: no profiled twin for the curvature-aware cost / Gaussian rows"; return TMPC_ERR_INVALID; }
    if (h->fast) {
        pk = tmpc::pick_fast_kernel(h->d, &thr, true);
#ifndef TMPC_GENERATED_STAGE
        // the latency variants of cfg 2 are profiled as themselves (tmpc_set_latency_mode before the call)
        if (h->latency_mode == 3 && h->kernel_quad && tmpc::pick_quad_kernel(h->d, true, false)) {
            pk = tmpc::pick_quad_kernel(h->d, true, false); thr = 256; lds = h->lds_bytes_quad;
        } else if (h->latency_mode >= 2 && h->kernel_scan && h->scan_threads == 128 && h->scan_sl == 3 && h->d.n_up == 8 && h->d.M == 8) {
            pk = (tmpc::SolveKernel)tmpc::tmpc_solve_fast_kernel<8, 8, 6, 128, true, tmpc::ScanSolo>; thr = 128; lds = h->lds_bytes_scan;
        } else if (h->latency_mode != 0 && h->kernel_lat) {
            pk = tmpc::pick_latency_kernel(h->d, true); thr = 128; lds = h->lds_bytes_fast2;
        }
#endif
        TMPC_HIP_CHECK(h, hipFuncSetAttribute((const void *)pk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    // (a compact handle is profiled through the fast kernel of its shape: same phases and arithmetic, one wave per SIMD)
    hipLaunchKernelGGL(pk, dim3(h->B), dim3(thr), lds, h->stream, h->d, h->B,
                       h->xinit, h->x0, h->params, h->xtraj, h->utraj, h->pobj, h->exit_code, h->qp_status,
                       h->sqp_iter, h->res_eq, h->qp_iter, dp, tmpc::StateIO{nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr});
    TMPC_HIP_CHECK(h, hipGetLastError());
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    std::vector<long long> host(n);
    TMPC_HIP_CHECK(h, hipMemcpy(host.data(), dp, n * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < tmpc::PH_COUNT; i++) {
        double acc = 0.0;
        for (int b = 0; b < h->B; b++) acc += (double)host[(size_t)b * tmpc::PH_COUNT + i];
        cycles[i] = (int64_t)(acc / h->B);
    }
    return TMPC_OK;
}

int tmpc_debug_eval_stage(tmpc_handle *h, int32_t n, const double *z, const double *p, const double *pi, const double *lamh,
                          double *cost, double *cost_grad, double *cost_hess, double *hval, double *h_jac,
                          double *x_next, double *x_jac, double *lag_hess, double *mirror)
{
    if (!h || n <= 0 || !z || !p) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const int nh = h->d.n_up + h->d.M;
    const size_t sz_in[4] = {(size_t)n * tmpc::ext_nv(h->d) * 8, (size_t)n * h->d.npar * 8, (size_t)n * 5 * 8, (size_t)n * nh * 8};
    const void *src[4] = {z, p, pi, lamh};
    DevBufs bufs;
    double *din[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 4; i++) {
        if (!src[i]) continue;
        TMPC_HIP_CHECK(h, bufs.alloc(&din[i], sz_in[i]));
        TMPC_HIP_CHECK(h, hipMemcpy(din[i], src[i], sz_in[i], hipMemcpyHostToDevice));
    }
    const size_t sz_out[9] = {(size_t)n * 8, (size_t)n * 7 * 8, (size_t)n * 49 * 8, (size_t)n * nh * 8, (size_t)n * nh * 7 * 8,
                              (size_t)n * 5 * 8, (size_t)n * 35 * 8, (size_t)n * 49 * 8, (size_t)n * 49 * 8};
    double *dout[9]; void *dst[9] = {cost, cost_grad, cost_hess, hval, h_jac, x_next, x_jac, lag_hess, mirror};
    for (int i = 0; i < 9; i++) TMPC_HIP_CHECK(h, bufs.alloc(&dout[i], sz_out[i]));
    hipLaunchKernelGGL(tmpc::tmpc_debug_eval_kernel, dim3((n + 63) / 64), dim3(64), 0, h->stream, h->d, n, din[0], din[1], din[2], din[3],
                       dout[0], dout[1], dout[2], dout[3], dout[4], dout[5], dout[6], dout[7], dout[8]);
    TMPC_HIP_CHECK(h, hipGetLastError());
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < 9; i++)
        if (dst[i]) TMPC_HIP_CHECK(h, hipMemcpy(dst[i], dout[i], sz_out[i], hipMemcpyDeviceToHost));
    return TMPC_OK;
}

#ifdef TMPC_SWEEP_PROFILE
// Profiling build only (not declared in tmpc_hip.h): read and reset the sweep clock accumulators.
int tmpc_debug_sweep_profile(tmpc_handle *h, uint64_t *out, int32_t n)
{
    if (!h || !out || n < tmpc::SP_COUNT) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    unsigned long long host[tmpc::SP_COUNT];
    TMPC_HIP_CHECK(h, hipMemcpyFromSymbol(host, HIP_SYMBOL(tmpc::g_sweep_prof), sizeof host));
    for (int i = 0; i < tmpc::SP_COUNT; i++) out[i] = host[i];
    memset(host, 0, sizeof host);
    TMPC_HIP_CHECK(h, hipMemcpyToSymbol(HIP_SYMBOL(tmpc::g_sweep_prof), host, sizeof host));
    return TMPC_OK;
}
#endif

}  // extern "C"
