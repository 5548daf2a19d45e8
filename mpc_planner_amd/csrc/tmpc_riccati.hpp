// Square-root Riccati recursion of the interior-point QP solve: factorisation (matrix recursion) and vector solves
// (backward / forward sweeps), one wave per trajectory, operands in LDS (struct Lds, tmpc_solve.hip / tmpc_fast.hpp).
// Included by tmpc_solve.hip after the LDS layout and the cross-lane helpers; shared by the generic and the fast kernels.
#pragma once

namespace tmpc {

// ---- optional coarse clock split of the sweeps (tools/profile_sweeps.py builds a separate library with this macro;
// the product library never defines it) ---------------------------------------------------------------------------
#ifdef TMPC_SWEEP_PROFILE
enum { SP_FACTOR_TERM = 0, SP_FACTOR_LOOP, SP_SOLVE_PRE, SP_SOLVE_BWD, SP_SOLVE_FWD, SP_SOLVE_POST, SP_CALLS_FACTOR, SP_CALLS_SOLVE, SP_COUNT };
__device__ unsigned long long g_sweep_prof[SP_COUNT];
#define SWEEP_T0() long long sp_t = clock64()
#define SWEEP_T(i) do { const long long sp_n = clock64(); if (tid == 0) atomicAdd(&g_sweep_prof[i], (unsigned long long)(sp_n - sp_t)); sp_t = sp_n; } while (0)
#define SWEEP_COUNT(i) do { if (tid == 0) atomicAdd(&g_sweep_prof[i], 1ull); } while (0)
#else
#define SWEEP_T0()
#define SWEEP_T(i)
#define SWEEP_COUNT(i)
#endif

// Issue priority of the wave (s_setprio, round 5).  Two waves share a SIMD; when both have an instruction ready the one with the higher priority issues.
// A wave inside a SEQUENTIAL phase (the Riccati factorisation and sweeps: 8 of 64 lanes, every instruction on the trajectory's critical path) gets
// TMPC_PRIO_SEQ, the rest of an interior-point iteration TMPC_PRIO_IPM, the linearisation 0 -- measured on the cfg 2 bench launch
// (profiles/round5_g_setprio_ab{,2}.jsonl): no priorities 27.53 ms, (3, 0) 27.15, (3, 1) 27.07, **(3, 2) 27.01** (+1.9 %), (3, 3) 27.20, (1, 1) 27.19,
// the reverse assignment (0, 3) 27.63; at 2-4 rounds per launch the same shape gains 4-8 % (round5_k_prio_vs_launch_size.jsonl).  Results are
// unaffected (it only orders issue between the waves of a SIMD).  Only where every SIMD hosts two waves: at 7 workgroups per CU (cfg 4's shape) one
// SIMD hosts a single wave, the priorities make the launch 6 % SLOWER (round5_j_prio_other_configs.jsonl: the starved waves set the makespan) --
// the host sets Dims::prio per launch (launch_solve) and the macros test it (a scalar branch).
#ifndef TMPC_PRIO_SEQ
#define TMPC_PRIO_SEQ 3
#endif
#ifndef TMPC_PRIO_IPM
#define TMPC_PRIO_IPM 2
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define TMPC_PRIO_HIGH() do { if (d.prio) __builtin_amdgcn_s_setprio(TMPC_PRIO_SEQ); } while (0)
#define TMPC_PRIO_LOW() do { if (d.prio) __builtin_amdgcn_s_setprio(TMPC_PRIO_IPM); } while (0)
#define TMPC_PRIO_LINEARISE() do { if (d.prio) __builtin_amdgcn_s_setprio(0); } while (0)
#else
#define TMPC_PRIO_HIGH()
#define TMPC_PRIO_LOW()
#define TMPC_PRIO_LINEARISE()
#endif

// ---- Riccati recursion: factorisation --------------------------------------------------------------
// Lane i (< 7) owns ROW i of the stage matrix F_k = Hh_k + [B A]^T P_{k+1} [B A] in registers f[0..i].  The elimination runs entirely in
// registers: pivots and column entries are broadcast inside the 16-lane row (v_mov_b64_dpp row_newbcast), no LDS traffic and no barriers.
// Round 5: the elimination STOPS after the two input columns.  What is left in rows 2..6 is the Schur complement
// P_k = F_xx - Lxu Lxu^T -- the cost-to-go Hessian itself --, and the next stage uses it as it is (w = P [B A]_.,i per lane, F_ij = Hh_ij +
// sum_n w_n [B A]_nj) instead of re-factorising it into Lxx Lxx^T and forming G = Lxx^T [B A], F = Hh + G^T G (rounds 1-4: the square-root
// form, 7 pivots per stage).  The first two column eliminations are THE SAME operations in both forms, so P_k is the same matrix; what goes
// is the five state pivots (each: broadcast, v_rsq_f64 + Halley step, scale, column broadcasts, rank-1 update -- and v_rsq_f64 alone issues
// for 16 cycles, profiles/round5_chain_floor.json) and the triangular products: 227 -> ~150 VALU instructions per stage, two pivots on
// the chain instead of seven.  Numerics: tools/riccati_form_study.py ran both forms inside the oracle's interior-point method on the
// bench scenes -- at the reference's qp_tol = 1e-5 no exit code, SQP or interior-point iteration count changes on 2560 trajectories and the
// iterates agree to 4e-11 (the level of the kernels' rounding differences); profiles/round5_riccati_form_study.json.
// In place of Hh_k the "factor block" (28 doubles) is written for the vector solves:
//   [0..9]  Lxu (5x2, row-major)   [10] L10   [11] 1/L00   [12] 1/L11   [13..27] P_k (packed lower 5x5)
#ifdef TMPC_EXP_UNIFORM_PIVOTS
constexpr int FB_LXU = 0, FB_L10 = 10, FB_R0 = 11, FB_R1 = 12, FB_P = 13;
#else
// (round 6: [10] 1/L00  [11] L10  [12] 1/L11 -- the vector sweeps apply Luu on lanes 0 and 1 only and broadcast the result: lane 0 needs (1/L00, L10), lane 1
//  (L10, 1/L11), ONE two-double load with a per-lane base where every lane loaded all three, two instructions; the arithmetic is the same)
constexpr int FB_LXU = 0, FB_R0 = 10, FB_L10 = 11, FB_R1 = 12, FB_P = 13;
#endif

// Offset of stage k's block in Hh.  A stride of 28 doubles (56 dwords) puts the stages k, k + 8, k + 16 on the same LDS banks: the stage-parallel loops
// (one lane per stage: the node's 28 stores of Hh <- W, the 15 + 15 loads of P in the vector solves' prologue / epilogue, the row passes' ds_add_f64)
// ran every access in three passes; 29 is conflict-free and costs (N + 1) doubles of LDS.  The stride is a COMPILE-TIME fact of the instantiation --
// the layout parameter CP of the routines below: 0 fast / generic layouts (28), 1 compact (28), 2 compact with 29, 3 compact with 30 doubles of which the
// last two hold the stage's y (compact_layout(), tmpc_fast.hpp: the tuned one-wave shapes; 3 is (12,12)'s, whose seven-per-CU budget has 69 bytes to spare --
// 30 k is 2-way at worst for 21 stages, and dropping the y array pays for it) -- because a run-time stride, one more live SGPR in kernels that spill 250 of
// them, cost the two-wave compact kernels 3.7 %, and a bank swizzle k * 28 + (k >> 3) (no extra LDS) more than it saved (profiles/round5_q_*).  Measured: +2.4 % at cfg 2.
template <int CP> __host__ __device__ constexpr int hstride() { return CP == 2 ? NP28 + 1 : (CP == 3 ? NP28 + 2 : NP28); }
template <int CP> __device__ __forceinline__ int hoff(int k) { return k * hstride<CP>(); }
template <int CP> __device__ __forceinline__ int hoff_lane(int k) { return mul24(k, hstride<CP>()); }                             // (k per lane)
// y = Luu^-1 (...) of stage k, handed from the backward sweep (or the factorisation's extra row) to the forward sweep: its own array, or -- layout 3 --
// the two spare doubles of the stage's 30-double block
template <int CP> __device__ __forceinline__ double *ysl(const Lds &L, int k) { return CP == 3 ? L.Hh + hoff<CP>(k) + NP28 : L.y + k * NU; }

// (The sequential sweeps use lanes 0..7 of the wave and rows 1..3 run along on copies.  Switching those rows off for the sweeps -- the LDS unit is
// busy 70 % of the kernel time at eight trajectories per CU and the sweeps issue two thirds of its instructions -- was measured in round 5: 1.3 %
// SLOWER, profiles/round5_o_sweep_rows_ab.jsonl; an LDS instruction costs the same with 16 lanes as with 64 and the EXEC changes are not free.)
// The stage loop of the factorisation is unrolled by two in the one-wave-per-SIMD kernels and rolled in the two-waves-per-SIMD (compact) ones -- a
// property of the INSTANTIATION (its layout parameter CP: 0 = fast / generic layouts, >= 1 = compact), so that every translation unit -- the four
// instantiation units, a generated solver's single unit, experiment builds -- gives a kernel the same body (round-5 advisor: a per-unit macro made
// the same template differ between units).  -DTMPC_FACTOR_UNROLL=0/1 forces one form everywhere (A/B builds only).
template <int CP> __host__ __device__ constexpr bool factor_unrolled()
{
#ifdef TMPC_FACTOR_UNROLL
    return TMPC_FACTOR_UNROLL != 0;
#else
    return CP == 0;
#endif
}

// right-looking elimination of columns C0 .. C1-1 of the row-per-lane matrix
template <int C0, int C1 = NV>
__device__ __forceinline__ bool chol_rows(double (&f)[NV], int lane, double *r0, double *r1)
{
    bool bad = false;
    static_for<C0, C1>([&](auto c_) {
        constexpr int c = decltype(c_)::value;
        const double dpiv = bcast16<c>(f[c]);
        if (!(dpiv > 0.0)) bad = true;
        const double y = rsqrt_nr(dpiv);
        f[c] *= y;                                  // lane i >= c: L_ic (i == c: sqrt(d))
        if (c == 0 && r0) *r0 = y;
        if (c == 1 && r1) *r1 = y;
        static_for<c + 1, NV>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            const double ljc = bcast16<j>(f[c]);    // L_jc to the row's lanes (one v_mov_b64_dpp; folding it into the fmac as
            f[j] -= f[c] * ljc;                     //  v_fmac_f64_dpp by hand measured no gain: profiles/round3_e_fused_fmac_dpp_rejected.json)
        });
    });
    (void)lane;                                       // entries above the diagonal are never read
    return bad;
}

// dyn8[k] = (Xa, Xw, Xp, Xv, Ya, Yw, Yp, Yv): the only non-constant entries of [B A] for the unicycle
// (tmpc_stage.hpp dyn_jacobian); the rest is identity / dt / dt^2/2.
enum { D8_XA = 0, D8_XW, D8_XP, D8_XV, D8_YA, D8_YW, D8_YP, D8_YV };

// Compact layout (no dense [B A] in LDS): a lane's column / row of [B A] through the table `tab` (ba_off, tmpc_solve.hip).
// Column j: entries 0, 1 are stage-dependent for j in {a, w, psi, v} (offset o + 8 k) and constants otherwise (stride 0); entries
// 2..4 are constants for every column.  Row i (state index) in dyn8 column order (a, w, psi, v): stage-dependent for x, y.
struct BaLane { int o0, o1, st; };
__device__ __forceinline__ BaLane ba_column(int N, int j)
{
    BaLane c;
    c.o0 = ba_off(N, 0, 0, j); c.o1 = ba_off(N, 0, 1, j);
    c.st = c.o0 < 8 ? 8 : 0;
    return c;
}
__device__ __forceinline__ BaLane ba_row4(int N, int i5)          // o0: base of (b_a, b_w, a_psi, a_v) of row i5
{
    BaLane c;
    c.o0 = i5 < 2 ? 4 * i5 : N * 8 + BA_NGROUP0 + 4 * (i5 - 2); c.o1 = 0;
    c.st = i5 < 2 ? 8 : 0;
    return c;
}

// The sweeps below are strictly sequential over stages; to keep LDS latency off the critical path every
// operand of stage k-1 is (re)loaded into the same registers right after its last use in stage k, so the loads
// complete underneath the dependent Cholesky / readlane chain of stage k.
// NTH = threads per trajectory: 64, or 128 for the two-wave variant, in which the sweeps run on wave 0 alone (wave 1
// waits at the closing barrier) and only the stage-parallel loops use all threads.
// `sw`: which of the two waves sweeps (two-wave variant); the other one waits at the closing barrier.
// ---- the sequential parts work on 16-lane ROWS ---------------------------------------------------------------------------
// All cross-lane traffic of the sweeps is row-local (bcast16, row shifts), so a wave can run up to four trajectories' sweeps at
// once, one per row: `li` is the lane's index inside its row (li >= 16: idle lane), `L` the LDS view of the row's trajectory
// (per-lane pointers when rows differ) and `wr` enables the row's stores.  One trajectory per wave (the fast / compact kernels):
// li = lane, rows 1..3 idle.  Four per wave: the team kernels (tmpc_fast.hpp).
// VEC (the register-row kernels): the PREDICTOR's right-hand side rides through the factorisation as an extra row.  Lane 7 of the row
// holds the row [g_u g_x] of the bordered matrix [[F, g], [g^T, .]] (g = gh, the predictor right-hand side, complete before the
// factorisation starts): it takes part in every column update like the rows below the pivot and never becomes a pivot, so after the
// elimination of the two input columns it is  [y0 y1 | p_k]: y = Luu^-1 g_u is what the forward sweep needs, and the rest is the Schur
// complement of the border, i.e. the cost-to-go gradient p_k itself.  At the next stage the row's "own column" is the dynamics residual rb
// and p joins the product: w = P rb + p.  The separate backward vector sweep of the predictor (and its stage-parallel prologue P rb)
// disappear: one of the five sequential passes over the stages of an interior-point iteration.  Same algebra as the separate sweep; the
// operations associate differently (rounding-level differences).
// SQ (Dims::riccati_form = 1, tmpc_dims.riccati_form = TMPC_RICCATI_SQUARE_ROOT): the SQUARE-ROOT form of rounds 1-4 -- what acados' HPIPM runs in its
// default mode (square_root_alg = 1 [UPSTREAM]; solver_generator/generate_acados_solver.py:171 takes the defaults) -- kept as a selectable
// instantiation (round 6): all seven columns are eliminated, rows 2..6 hold Lxx with P_k = Lxx Lxx^T, the next stage forms G = Lxx^T [B A] and
// F = Hh + G^T G (no subtraction of large numbers), FB_P holds Lxx, and the extra row ends as [y0 y1 | lx] with p_k = Lxx lx.  At the
// reference's qp_tol = 1e-5 both forms give the same exit codes, iteration counts and iterates (profiles/round5_riccati_form_study.json); at
// qp_tol = 1e-9 a few solves per thousand end differently, which is why a run against a real acados at that tolerance (tools/acados_replay.py)
// should use this form.
template <int CP, bool VEC = false, bool SQ = false>
__device__ __forceinline__ bool riccati_factor_rows(const Lds &L, const Dims &d, int li, bool wr)
{
    const int N = d.N;
    const bool rowl = li < NV && wr;
    const bool vec = VEC && li == NV;                // the right-hand-side row
    const double vmask = vec ? 1.0 : 0.0;
    const int ls = li < NV ? li : 0;
    const int i5 = li - NU;                          // state index of lanes 2..6
    const double dt = d.dt, sdt = d.sdt, shdt2 = d.shdt2;      // (sdt, shdt2: the spline row of [B A]; zero for the model without a spline state)
    bool bad = false;
    // Operands of a stage.  The operands of stage k - 1 are loaded while stage k still needs some of its own (the scheduler computes rows 2..6 of
    // F after the pivots), so two register sets are live and the compiler copies the shadow set over at the back edge: 11 v_mov_b64 of the loop's
    // 138 instructions.  TMPC_FACTOR_UNROLL = 1 names the two sets and unrolls the stage loop by two (like the vector sweeps): no copies, 129
    // instructions per stage, 4 % fewer cycles per factorisation on a lone wave -- and 0.7 % LESS throughput on the saturated compact kernel
    // (profiles/round5_m_factor_unroll_rotation_ab.jsonl).  So the translation units of the one-wave-per-SIMD kernels (fast, profiled twins)
    // build the unrolled loop, those of the compact kernels (two waves per SIMD) the rolled one; the arithmetic is the same.
    struct Opnd { double hk[NV], ba[NX], dn[8]; };
    double f[NV];
    const BaLane bc = ba_column(N, ls);
    double bac[NX];
    if constexpr (CP && !VEC) {                          // constant entries of the own column of [B A]
#pragma unroll
        for (int m = 2; m < NX; m++) bac[m] = L.tab[ba_off(N, 0, m, ls)];
    }
    // VEC: every lane loads hk[] / ba[] through per-lane (pointer, stride) pairs, so that the extra row reads gh / rb with the same loads.
    // Running pointers, stepped back by the lane's stride once per stage (round 4: the k * stride multiplications and per-entry index
    // arithmetic were ~17 of the stage loop's ~250 instructions): a lane reads hk[j] = row[j] at IMMEDIATE offsets from the start of its row of
    // the packed Hh block -- the entries j > own row index belong to the next row and are never used (chol_rows reads the lower triangle only).
    // (the same for ba[] -- running offsets or pointers, five more values live across the stage loop -- pushed the compact kernels into scratch
    // twice: ba[] keeps its (base + k * stride) form, which the compiler rematerialises; mul24: tmpc_kernels.hpp)
    const double *hrow = nullptr;
    int hstep = 0;
    const double *bbase = CP ? L.tab : L.BA;
    int bo[NX], bst[NX];
    if constexpr (VEC) {
        hrow = vec ? L.gh : L.Hh + pidx(ls, 0);       // (row start inside a stage's block; the block's offset is added per stage: hoff<CP>)
        hstep = 0;
#pragma unroll
        for (int m = 0; m < NX; m++) {
            if constexpr (CP) { bo[m] = vec ? (int)(L.rb - L.tab) + m : ba_off(N, 0, m, ls); bst[m] = vec ? NX : (bo[m] < 8 ? 8 : 0); }
            else { bo[m] = vec ? (int)(L.rb - L.BA) + m : m * NV + ls; bst[m] = vec ? NX : NX * NV; }
        }
    }
    // (the running pointers are positioned by seek_stage and stepped by load_stage: stage k, then k - 1, ...; the last call re-loads stage 0)
    auto seek_stage = [&](int) {};
    auto load_stage = [&](Opnd &o, int k, bool step) {
        // unconditional loads (clamped indices): entries above the diagonal / of idle lanes are never used
        if constexpr (VEC) {
            const double *hr = hrow + (vec ? k * NV : hoff<CP>(k));          // (two wave-uniform offsets, one select per stage)
#pragma unroll
            for (int j = 0; j < NV; j++) o.hk[j] = hr[j];
#pragma unroll
            for (int m = 0; m < NX; m++) o.ba[m] = bbase[bo[m] + mul24(k, bst[m])];
#ifdef TMPC_EXP_DN_LOADS
#pragma unroll
            for (int q = 0; q < 8; q++) o.dn[q] = (CP ? L.tab : L.dyn8)[k * 8 + q];
#endif
            (void)step;
        } else {
            const double *Hk = L.Hh + hoff<CP>(k);
#pragma unroll
            for (int j = 0; j < NV; j++) o.hk[j] = Hk[pidx(ls, j <= ls ? j : ls)];
            if constexpr (CP) {
                o.ba[0] = L.tab[bc.o0 + mul24(k, bc.st)]; o.ba[1] = L.tab[bc.o1 + mul24(k, bc.st)];
#pragma unroll
                for (int m = 2; m < NX; m++) o.ba[m] = bac[m];
#ifdef TMPC_EXP_DN_LOADS
#pragma unroll
                for (int q = 0; q < 8; q++) o.dn[q] = L.tab[k * 8 + q];
#endif
            } else {
                const double *BA = L.BA + k * NX * NV;
#pragma unroll
                for (int m = 0; m < NX; m++) o.ba[m] = BA[m * NV + ls];
#ifdef TMPC_EXP_DN_LOADS
#pragma unroll
                for (int q = 0; q < 8; q++) o.dn[q] = L.dyn8[k * 8 + q];
#endif
            }
        }
    };
    // terminal node: P_N = the xx-block of Hh_N (rows/cols 2..6) as it is; the extra row starts as p_N = g_x of node N
#pragma unroll
    for (int j = 0; j < NV; j++) f[j] = (j >= NU) ? (vec ? L.gh[N * NV + j] : L.Hh[hoff<CP>(N) + pidx(ls, j <= ls ? j : ls)]) : 0.0;
    Opnd oa, ob;
    seek_stage(N - 1);
    load_stage(oa, N - 1, true);
    if constexpr (SQ) bad |= chol_rows<NU>(f, li, nullptr, nullptr);       // square-root form: Cholesky of the xx block of node N
#ifdef TMPC_EXP_NO_MERGE_FACTOR
    if (vec && wr) {
#pragma unroll
        for (int l = 0; l < NX; l++) L.pr[N * NX + l] = f[NU + l];          // p_N (square-root form: lx of node N)
    }
#endif
    // Store merging (round 6, A/B builds: -DTMPC_EXP_NO_MERGE_FACTOR / -DTMPC_EXP_MERGE_FWD / -DTMPC_EXP_MERGE_BWD).  Stores of disjoint lane sets that hold the same
    // registers at the same point of the loop can leave in ONE LDS instruction with a per-lane address.  In the FACTORISATION -- p_{k+1} of the extra row with the rows
    // of P_{k+1}, its [y0 y1] with the Lxu pairs: four LDS instructions per stage fewer, nothing else changes -- that is +1.4 % on the saturated cfg 2 launch and taken.
    // In the vector sweeps (du of lanes 0, 1 with dx of lanes 2..6; y with p) the merged store's value needs a select on the END of the stage's dependent chain,
    // which holds the store back: -3.4 % (forward) and -2.0 % (backward) -- not taken.  profiles/round6_saturated_levers_ab.jsonl; results are bitwise the same either way.
    // End of stage k: the Lxu pairs of lanes 2..6 and the extra row's [y0 y1] are the registers f[0], f[1] of their lanes: ONE store for both (per-lane
    // address); lane 1 adds L10 and the two reciprocal pivots.  (p_k follows at the top of stage k - 1; p_0 is never read.)
    auto store_pairs = [&](int k, double r0, double r1) {
        double *Fb = L.Hh + hoff<CP>(k);
#ifndef TMPC_EXP_NO_MERGE_FACTOR
        if ((rowl && li >= NU) || (vec && wr)) {
            double *pp = vec ? ysl<CP>(L, k) : Fb + FB_LXU + 2 * i5;
            pp[0] = f[0]; pp[1] = f[1];
        }
        if (rowl && li == 1) { Fb[FB_L10] = f[0]; Fb[FB_R0] = r0; Fb[FB_R1] = r1; }
#else
        if (rowl) {
            if (li >= NU) { Fb[FB_LXU + 2 * i5] = f[0]; Fb[FB_LXU + 2 * i5 + 1] = f[1]; }
            if (li == 1) { Fb[FB_L10] = f[0]; Fb[FB_R0] = r0; Fb[FB_R1] = r1; }
        }
        if (vec && wr) {                                           // [y0 y1 | p_k] of stage k (square-root form: lx)
            ysl<CP>(L, k)[0] = f[0]; ysl<CP>(L, k)[1] = f[1];
#pragma unroll
            for (int l = 0; l < NX; l++) L.pr[k * NX + l] = f[NU + l];
        }
#endif
    };
    auto stage = [&](const Opnd &o, Opnd &nx, int k) {
        // broadcast P (lower triangle of the 5x5 cost-to-go Hessian of stage k+1: rows 2..6 after the elimination) to every lane of the row
        double Pm[NX][NX];
        static_for<0, NX>([&](auto m_) {
            constexpr int m = decltype(m_)::value;
#pragma unroll
            for (int l = 0; l <= m; l++) Pm[m][l] = bcast16<NU + m>(f[NU + l]);
        });
        // P_{k+1} (own row of lanes 2..6) is kept for the vector solves -- and the extra row's p_{k+1} (square-root form: lx) leaves in the SAME five store
        // instructions (round 6: the LDS unit is the busy one at eight trajectories per CU, and an LDS instruction costs the same with one lane as with six):
        // lane 7 holds it in the same registers f[2..6] at the same point of the loop; only the address differs per lane
#ifndef TMPC_EXP_NO_MERGE_FACTOR
        if ((rowl && li >= NU) || (vec && wr)) {
            double *Ln = vec ? L.pr + (k + 1) * NX : L.Hh + hoff<CP>(k + 1) + FB_P + i5 * (i5 + 1) / 2;
            const int lim = vec ? NX : i5;
#pragma unroll
            for (int l = 0; l < NX; l++) *(l <= lim ? Ln + l : L.scr + (CP ? 0 : 56) + l) = f[NU + l];   // (entries above the diagonal go to a dummy slot: a select on the address instead of five masked stores)
        }
#else
        if (rowl && li >= NU) {
            double *Ln = L.Hh + hoff<CP>(k + 1) + FB_P;
#pragma unroll
            for (int l = 0; l < NX; l++) *(l <= i5 ? Ln + i5 * (i5 + 1) / 2 + l : L.scr + (CP ? 0 : 56) + l) = f[NU + l];   // (entries above the diagonal go to a dummy slot: a select on the address instead of five masked stores)
        }
#endif
        if constexpr (SQ) {
            // G = Lp^T [B A] (5 x 7), Lp = Pm (the lower triangle holds Lxx of stage k + 1).  Own column densely from ba[]; all columns (row-uniform)
            // from the sparse [B A]:  x: e0   y: e1   s: e4   psi: (Xp,Yp,1,0,0)   v: (Xv,Yv,0,1,sdt)   a: (Xa,Ya,0,dt,shdt2)   w: (Xw,Yw,dt,0,0)
            // F = Hh + G^T G keeps the square-root structure (a factor-level perturbation only).  The extra row: (Lp^T rb)_l + lx_l of stage k + 1.
            double Go[NX];
#pragma unroll
            for (int l = 0; l < NX; l++) {
                double acc = 0.0;
#pragma unroll
                for (int m = l; m < NX; m++) acc += Pm[m][l] * o.ba[m];
                Go[l] = VEC ? fma(vmask, f[NU + l], acc) : acc;
            }
#ifdef TMPC_EXP_DN_LOADS
            const double Xa = o.dn[D8_XA], Xw = o.dn[D8_XW], Xp = o.dn[D8_XP], Xv = o.dn[D8_XV];
            const double Ya = o.dn[D8_YA], Yw = o.dn[D8_YW], Yp = o.dn[D8_YP], Yv = o.dn[D8_YV];
#else
            const double Xa = bcast16<ZA>(o.ba[0]), Ya = bcast16<ZA>(o.ba[1]), Xw = bcast16<ZW>(o.ba[0]), Yw = bcast16<ZW>(o.ba[1]);
            const double Xp = bcast16<ZPSI>(o.ba[0]), Yp = bcast16<ZPSI>(o.ba[1]), Xv = bcast16<ZV>(o.ba[0]), Yv = bcast16<ZV>(o.ba[1]);
#endif
            double Ga[NX], Gw[NX], Gp[3], Gv[NX];
            Ga[0] = ((Pm[0][0] * Xa + Pm[1][0] * Ya) + Pm[3][0] * dt) + Pm[4][0] * shdt2;
            Ga[1] = (Pm[1][1] * Ya + Pm[3][1] * dt) + Pm[4][1] * shdt2;
            Ga[2] = Pm[3][2] * dt + Pm[4][2] * shdt2;
            Ga[3] = Pm[3][3] * dt + Pm[4][3] * shdt2;
            Ga[4] = Pm[4][4] * shdt2;
            Gw[0] = (Pm[0][0] * Xw + Pm[1][0] * Yw) + Pm[2][0] * dt;
            Gw[1] = Pm[1][1] * Yw + Pm[2][1] * dt;
            Gw[2] = Pm[2][2] * dt; Gw[3] = 0.0; Gw[4] = 0.0;
            Gp[0] = (Pm[0][0] * Xp + Pm[1][0] * Yp) + Pm[2][0];
            Gp[1] = Pm[1][1] * Yp + Pm[2][1];
            Gp[2] = Pm[2][2];
            Gv[0] = ((Pm[0][0] * Xv + Pm[1][0] * Yv) + Pm[3][0]) + Pm[4][0] * sdt;
            Gv[1] = (Pm[1][1] * Yv + Pm[3][1]) + Pm[4][1] * sdt;
            Gv[2] = Pm[3][2] + Pm[4][2] * sdt;
            Gv[3] = Pm[3][3] + Pm[4][3] * sdt;
            Gv[4] = Pm[4][4] * sdt;
            double a0 = o.hk[ZA], a1 = o.hk[ZW], a2 = o.hk[ZX], a3 = o.hk[ZY], a4 = o.hk[ZPSI], a5 = o.hk[ZV], a6 = o.hk[ZS];
#pragma unroll
            for (int l = 0; l < NX; l++) {                       // F row `li`: F_ij = Hh_ij + sum_l G_l,li G_l,j
                a0 += Go[l] * Ga[l];
                if (l < 3) a1 += Go[l] * Gw[l];
                if (l < 1) a2 += Go[l] * Pm[0][0];
                if (l < 2) a3 += Go[l] * Pm[1][l];
                if (l < 3) a4 += Go[l] * Gp[l];
                a5 += Go[l] * Gv[l];
                a6 += Go[l] * Pm[4][l];
            }
            f[ZA] = a0; f[ZW] = a1; f[ZX] = a2; f[ZY] = a3; f[ZPSI] = a4; f[ZV] = a5; f[ZS] = a6;
            load_stage(nx, k > 0 ? k - 1 : 0, k > 1);
            double r0 = 0.0, r1 = 0.0;
            bad |= chol_rows<0>(f, li, &r0, &r1);                 // all seven columns: rows 2..6 now hold Lxx of stage k (the extra row: lx)
            store_pairs(k, r0, r1);
            return;
        }
        // Schur-complement form (default): P_{k+1} enters the next stage's F unfactorised (its diagonal is tested after the loop, riccati_factor)
        // w = P [B A]_.,own (the lane's own column of [B A], densely from ba[]); the extra row: P rb + p_{k+1} -- as fma(1.0 or 0.0, f, acc):
        // exactly acc + f on the extra row and acc on the others (f is finite there), without the v_cndmask a select costs
        double w[NX];
#pragma unroll
        for (int n = 0; n < NX; n++) {
            double acc = Pm[n > 0 ? n : 0][0] * o.ba[0];
#pragma unroll
            for (int m = 1; m < NX; m++) acc = fma(Pm[m > n ? m : n][m > n ? n : m], o.ba[m], acc);
            w[n] = VEC ? fma(vmask, f[NU + n], acc) : acc;
        }
        // F row `li`: F_ij = Hh_ij + sum_n w_n [B A]_nj with the sparse columns of [B A] (row-uniform):
        //   x: e0   y: e1   s: e4   psi: (Xp,Yp,1,0,0)   v: (Xv,Yv,0,1,dt)   a: (Xa,Ya,0,dt,dt^2/2)   w: (Xw,Yw,dt,0,0)
        {
#ifdef TMPC_EXP_DN_LOADS
            const double Xa = o.dn[D8_XA], Xw = o.dn[D8_XW], Xp = o.dn[D8_XP], Xv = o.dn[D8_XV];
            const double Ya = o.dn[D8_YA], Yw = o.dn[D8_YW], Yp = o.dn[D8_YP], Yv = o.dn[D8_YV];
#else
            // (round 6) the eight stage-dependent entries of [B A] are rows x, y of the columns a, w, psi, v -- which the lanes of those columns hold as
            // ba[0], ba[1] of their OWN column: eight row broadcasts instead of four row-uniform LDS loads of the dyn8 block (the LDS unit is the busy one)
            const double Xa = bcast16<ZA>(o.ba[0]), Ya = bcast16<ZA>(o.ba[1]), Xw = bcast16<ZW>(o.ba[0]), Yw = bcast16<ZW>(o.ba[1]);
            const double Xp = bcast16<ZPSI>(o.ba[0]), Yp = bcast16<ZPSI>(o.ba[1]), Xv = bcast16<ZV>(o.ba[0]), Yv = bcast16<ZV>(o.ba[1]);
#endif
            f[ZA] = fma(w[4], shdt2, fma(w[3], dt, fma(w[1], Ya, fma(w[0], Xa, o.hk[ZA]))));
            f[ZW] = fma(w[2], dt, fma(w[1], Yw, fma(w[0], Xw, o.hk[ZW])));
            f[ZX] = o.hk[ZX] + w[0];
            f[ZY] = o.hk[ZY] + w[1];
            f[ZPSI] = fma(w[1], Yp, fma(w[0], Xp, o.hk[ZPSI])) + w[2];
            f[ZV] = fma(w[4], sdt, fma(w[1], Yv, fma(w[0], Xv, o.hk[ZV])) + w[3]);
            f[ZS] = o.hk[ZS] + w[4];
        }
        load_stage(nx, k > 0 ? k - 1 : 0, k > 1);     // operands of the next stage, hidden under the elimination (unconditional, clamped;
                                                      // the pointers stop at stage 0, which the last pass re-loads and discards)
        double r0 = 0.0, r1 = 0.0;
        bad |= chol_rows<0, NU>(f, li, &r0, &r1);     // the two input columns; rows 2..6 now hold P_k (lanes 2..6) / p_k (the extra row)
        store_pairs(k, r0, r1);
    };
    if constexpr (factor_unrolled<CP>()) {
        int k = N - 1;
        for (; k >= 1; k -= 2) {
            stage(oa, ob, k);
            stage(ob, oa, k - 1);
        }
        if (k == 0) stage(oa, ob, 0);
    } else {
        for (int k = N - 1; k >= 0; k--) stage(oa, oa, k);      // (one named set, re-loaded after its last use in source order)
    }
    return bad;
}

// NTH = threads per trajectory: 64, or 128 for the two-wave variant, in which the sweeps run on one wave (the other waits at the
// closing barrier) and only the stage-parallel loops use all threads.  `sw`: which of the two waves sweeps.
template <int NTH, int CP = 0, bool VEC = false, bool SQ = false>
__device__ __forceinline__ bool riccati_factor(const Lds &L, const Dims &d, int tid, int sw = 0)
{
    asm volatile("" : "+v"(tid));                    // opaque per call: lane-derived addresses are not shared with (kept live until) other phases
    bool anybad = false;
    SWEEP_T0(); SWEEP_COUNT(SP_CALLS_FACTOR);
    if (NTH == 64 || (tid >> 6) == sw) {
        const int lane = tid & 63;
        TMPC_PRIO_HIGH();
        const bool bad = riccati_factor_rows<CP, VEC, SQ>(L, d, lane, true);
        TMPC_PRIO_LOW();
        bool bad_p = false;
        if constexpr (!SQ) {
            // Schur-complement form: P_k = F_xx - Lxu Lxu^T enters the next stage unfactorised, so nothing downstream would notice a diagonal entry that
            // cancellation has driven NEGATIVE (the square-root form's state pivots did; round-5 advisor).  Tested here, off the sequential chain: lane
            // k reads the sign bits of diag(P_{k+1}) from the factor blocks the loop has just written (same wave: LDS operations execute in order) --
            // five ds_read_b32 and three integer operations per factorisation.  (Inside the stage loop the same test cost the compact kernels, at 254
            // of 256 registers, their zero-scratch build.)  The oracle tests the same bits (oracle/qp_ipm.c riccati_factor_classical).
            const unsigned *Pd = reinterpret_cast<const unsigned *>(L.Hh + hoff_lane<CP>(lane < d.N ? lane + 1 : d.N) + FB_P) + 1;      // (high dwords)
            bad_p = (int)(Pd[2 * 0] | Pd[2 * 2] | Pd[2 * 5] | Pd[2 * 9] | Pd[2 * 14]) < 0;
        }
        anybad = __any((bad && lane < 16) || bad_p);  // (rows 1..3 of the wave compute on copies: their pivots mean nothing)
        if (NTH > 64 && lane == 0) L.scr[63] = anybad ? 1.0 : 0.0;
    }
    __syncthreads();
    SWEEP_T(SP_FACTOR_LOOP);
    if (NTH > 64) anybad = L.scr[63] != 0.0;
    return anybad;
}

// ---- square-root Riccati: vector solve (backward + forward), rhs gh / rb -> dv, dpi ------------------
// Stage-parallel parts (one lane per stage, `nth` lanes of the trajectory's own wave(s)):
//   pre   q_k = P_{k+1} rb_k for all stages at once (off the sequential chain); parked in dpi[k+1]
//   post  dpi_k = P_k dx_k + p_k, k = 1..N
// SQ (square-root form): FB_P holds Lxx, P x = Lxx (Lxx^T x)
template <int CP, bool SQ = false>
__device__ __forceinline__ void riccati_solve_pre(const Lds &L, const Dims &d, int tid, int nth)
{
    const int N = d.N;
    for (int k = tid; k < N; k += nth) {
        const double *Pn = L.Hh + hoff_lane<CP>(k + 1) + FB_P;              // P_{k+1}, packed lower triangle
        const double *r = L.rb + mul24(k, NX);
        double pp[15], rr[NX];
#pragma unroll
        for (int e = 0; e < 15; e++) pp[e] = Pn[e];
#pragma unroll
        for (int m = 0; m < NX; m++) rr[m] = r[m];
        if constexpr (SQ) {
            double tl[NX];
#pragma unroll
            for (int l = 0; l < NX; l++) {
                double acc = 0.0;
#pragma unroll
                for (int m = l; m < NX; m++) acc += pp[m * (m + 1) / 2 + l] * rr[m];
                tl[l] = acc;
            }
#pragma unroll
            for (int i = 0; i < NX; i++) {
                double acc = 0.0;
#pragma unroll
                for (int l = 0; l <= i; l++) acc += pp[i * (i + 1) / 2 + l] * tl[l];
                L.dpi[mul24(k + 1, NX) + i] = acc;
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < NX; i++) {
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < NX; m++) acc += pp[(m > i ? m * (m + 1) / 2 + i : i * (i + 1) / 2 + m)] * rr[m];
            L.dpi[mul24(k + 1, NX) + i] = acc;
        }
    }
}
// dpi_k = P_k dx_k + p_k, k = 1..N (L.pr holds p_k: from the backward sweep, or from the extra row of a VEC factorisation)
// SQ: dpi_k = Lxx (Lxx^T dx_k) + p_k; LX (the fused predictor of the square-root form): L.pr holds lx with p_k = Lxx lx, so dpi_k = Lxx (Lxx^T dx_k + lx_k)
template <int CP, bool SQ = false, bool LX = false>
__device__ __forceinline__ void riccati_solve_post(const Lds &L, const Dims &d, int tid, int nth)
{
    const int N = d.N;
    for (int kk = tid; kk < N; kk += nth) {
        const int k = kk + 1;
        const double *Pk = L.Hh + hoff_lane<CP>(k) + FB_P;
        const double *dxk = L.dv + mul24(k, NV) + NU;
        double pp[15], rr[NX], pk[NX];
#pragma unroll
        for (int e = 0; e < 15; e++) pp[e] = Pk[e];
#pragma unroll
        for (int m = 0; m < NX; m++) { rr[m] = dxk[m]; pk[m] = L.pr[mul24(k, NX) + m]; }
        if constexpr (SQ) {
            double tl[NX];
#pragma unroll
            for (int l = 0; l < NX; l++) {
                double acc = 0.0;
#pragma unroll
                for (int m = l; m < NX; m++) acc += pp[m * (m + 1) / 2 + l] * rr[m];
                tl[l] = LX ? acc + pk[l] : acc;
            }
#pragma unroll
            for (int i = 0; i < NX; i++) {
                double acc = 0.0;
#pragma unroll
                for (int l = 0; l <= i; l++) acc += pp[i * (i + 1) / 2 + l] * tl[l];
                L.dpi[mul24(k, NX) + i] = LX ? acc : acc + pk[i];
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < NX; i++) {
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < NX; m++) acc += pp[(m > i ? m * (m + 1) / 2 + i : i * (i + 1) / 2 + m)] * rr[m];
            L.dpi[mul24(k, NX) + i] = acc + pk[i];
        }
    }
}

// The two sequential sweeps on rows (see riccati_factor_rows for li / L / wr).  Lane j (< 7) of a row = component j of the stage
// vector [u; x]; the cost-to-go gradient p lives in lanes 2..6.  MID: what separates the sweeps (the forward sweep reads the y the
// backward sweep stored): a workgroup barrier in the one-trajectory kernels, a wave-level fence when one wave sweeps for a team.
template <int CP, bool BWD = true, typename MID>
__device__ __forceinline__ void riccati_sweeps_rows(const Lds &L, const Dims &d, int li, bool wr, bool sweeper, MID mid)
{
    const int N = d.N;
    const bool rowl = li < NV && wr, xl = rowl && li >= NU;
    const int ls = li < NV ? li : 0;
    const int i5 = (li >= NU && li < NV) ? li - NU : 0;
    SWEEP_T0();
    if constexpr (BWD) {                                // (the fused predictor has done this part inside the factorisation)
    if (sweeper) {
    double p = L.gh[N * NV + ls];                       // p_N (lanes 2..6 meaningful)
    {
        // Two operand sets, filled alternately two stages ahead: a set is (re)loaded right after the stage that used it, so
        // its LDS latency passes underneath the other set's stage (with a single set the loads were issued at the end of
        // stage k and awaited at the top of stage k-1: ~120 exposed cycles per stage).
        // Compact layout: L.pr aliases L.dpi -- p_k takes the slot of q_{k-1} = (P rb)_{k-1}, which stage k - 1's operand load
        // (always issued before stage k runs: the sets are filled at least one stage ahead, and LDS operations of a wave
        // execute in order) has fetched by then; the closing loop adds P dx in place.
        struct Ops { double ghj, ba[NX], r0, l10, r1, lx0, lx1, q; };
        const int l01 = li == 1 ? 1 : 0;                                   // (lanes >= 2 follow lane 0: their values are not used)
        const BaLane bc = ba_column(N, ls);
        double bac[NX];
        if constexpr (CP) {
#pragma unroll
            for (int m = 2; m < NX; m++) bac[m] = L.tab[ba_off(N, 0, m, ls)];
        }
        auto load_stage = [&](Ops &o, int k) {
            const double *Fb = L.Hh + hoff<CP>(k);
            o.ghj = L.gh[uni(k * NV) + ls];
            if constexpr (CP) {
                o.ba[0] = L.tab[bc.o0 + mul24(k, bc.st)]; o.ba[1] = L.tab[bc.o1 + mul24(k, bc.st)];
#pragma unroll
                for (int l = 2; l < NX; l++) o.ba[l] = bac[l];
            } else {
                const double *BA = L.BA + k * NX * NV;
#pragma unroll
                for (int l = 0; l < NX; l++) o.ba[l] = BA[l * NV + ls];
            }
#ifdef TMPC_EXP_UNIFORM_PIVOTS
            o.r0 = Fb[FB_R0]; o.l10 = Fb[FB_L10]; o.r1 = Fb[FB_R1];
#else
            { const double *Pv = Fb + FB_R0 + l01; o.r0 = Pv[0]; o.r1 = Pv[1]; o.l10 = 0.0; }      // lane 0: (1/L00, L10); lane 1: (L10, 1/L11)
#endif
            o.lx0 = Fb[FB_LXU + 2 * i5]; o.lx1 = Fb[FB_LXU + 2 * i5 + 1];
            o.q = L.dpi[uni((k + 1) * NX) + i5];
        };
        auto stage = [&](const Ops &o, int k) {
            const double Pb = p + o.q;                                     // (P_{k+1} rb_k + p_{k+1}), lane 2+i
            double fj = o.ghj;
            static_for<0, NX>([&](auto l_) { constexpr int l = decltype(l_)::value; fj += o.ba[l] * bcast16<NU + l>(Pb); });
#ifdef TMPC_EXP_UNIFORM_PIVOTS
            const double y0 = bcast16<0>(fj) * o.r0;
            const double y1 = (bcast16<1>(fj) - o.l10 * y0) * o.r1;
#else
            const double y0 = bcast16<0>(fj * o.r0);                       // lane 0: fj 1/L00
            const double y1 = bcast16<1>((fj - o.r0 * y0) * o.r1);         // lane 1: (fj - L10 y0) 1/L11   (its o.r0 is L10)
#endif
            p = fj - o.lx0 * y0 - o.lx1 * y1;
#ifdef TMPC_EXP_MERGE_BWD
            if (rowl) *(li < NU ? ysl<CP>(L, k) + ls : L.pr + k * NX + i5) = li == 0 ? y0 : (li == 1 ? y1 : p);      // (one store: y from lanes 0, 1, p_k from lanes 2..6)
#else
            if (rowl && li == 0) { ysl<CP>(L, k)[0] = y0; ysl<CP>(L, k)[1] = y1; }
            if (xl) L.pr[k * NX + i5] = p;
#endif
        };
        Ops oa, ob;
        load_stage(oa, N - 1);
        if (xl) L.pr[N * NX + i5] = p;                                    // (after the load of q_{N-1}: same slot when aliased)
        int k = N - 1;
        for (; k >= 1; k -= 2) {
            load_stage(ob, k - 1);
            stage(oa, k);
            load_stage(oa, k >= 2 ? k - 2 : 0);                           // unconditional (clamped): a branch here makes the
                                                                          // compiler wait for ALL outstanding LDS loads at the join
            stage(ob, k - 1);
        }
        if (k == 0) stage(oa, 0);
    }
    }
    mid();
    }
    SWEEP_T(SP_SOLVE_BWD);
    // forward sweep; dx_0 = 0 (dx lives in lanes 2..6).  dx+ = A dx + B du + rb with A = I + E (E: columns psi, v).
    if (sweeper) {
        double dx = 0.0;
        // Lxu^T dx (two sums over the state lanes 2..6): every lane holds the stage's ten Lxu entries (row-uniform LDS reads) and gets the five
        // dx components by row broadcasts -- 5 broadcasts + 10 fma, where folding the two sums with DPP row shifts cost 12 32-bit DPP moves, 6
        // adds, 4 selects and 2 more broadcasts (round 4: -15 of the stage's 49 instructions); dpsi and dv are two of the five broadcasts
        // The ten Lxu entries have ONE register set, re-loaded for stage k + 1 right after their last use in stage k (two sets, like the other
        // operands, pushed the compact kernels into scratch); the loads pass underneath the rest of the stage.
        struct Ops { double y0, y1, r0, l10, r1, a_psi, a_v, b_a, b_w, rbi; };
        double lxu[2 * NX];
        auto load_lxu = [&](int k) {
#ifndef TMPC_EXP_LXU_ALL
            const double *Fb = L.Hh + hoff<CP>(k) + FB_LXU + 2 * i5;       // the lane's own pair
            lxu[0] = Fb[0]; lxu[1] = Fb[1];
#else
            const double *Fb = L.Hh + hoff<CP>(k) + FB_LXU;
#pragma unroll
            for (int e = 0; e < 2 * NX; e++) lxu[e] = Fb[e];
#endif
        };
        const double i_psi = li == ZPSI ? 1.0 : 0.0, i_v = li == ZV ? 1.0 : 0.0;
        const int l01f = li == 1 ? 1 : 0;
        double *dv_own = L.dv + ls;
        const BaLane br = ba_row4(N, i5);
        auto load_stage = [&](Ops &o, int k) {
            const double *Fb = L.Hh + hoff<CP>(k);
            o.y0 = ysl<CP>(L, k)[0]; o.y1 = ysl<CP>(L, k)[1];
#ifdef TMPC_EXP_UNIFORM_PIVOTS
            o.r0 = Fb[FB_R0]; o.l10 = Fb[FB_L10]; o.r1 = Fb[FB_R1];
#else
            { const double *Pv = Fb + FB_R0 + l01f; o.r0 = Pv[0]; o.r1 = Pv[1]; o.l10 = 0.0; }     // lane 0: (1/L00, L10); lane 1: (L10, 1/L11)
#endif
            if constexpr (CP) {
                const double *Tr = L.tab + br.o0 + mul24(k, br.st);              // own row of [B A] as (b_a, b_w, a_psi, a_v)
                o.a_psi = Tr[2]; o.a_v = Tr[3];
                o.b_a = Tr[0]; o.b_w = Tr[1];
            } else {
                const double *BAr = L.BA + k * NX * NV + i5 * NV;          // own row of [B A]
                o.a_psi = BAr[ZPSI]; o.a_v = BAr[ZV];                      // loads only: arithmetic here would wait for them
                o.b_a = BAr[ZA]; o.b_w = BAr[ZW];
            }
            o.rbi = L.rb[uni(k * NX) + i5];
        };
        auto stage = [&](const Ops &o, int k) {
            // du = -Luu^-T (Lxu^T dx + y)
#ifndef TMPC_EXP_LXU_ALL
            // Round 6 (round-5 verdict next-2 (a)): a lane keeps ITS OWN pair of Lxu -- one LDS load per stage instead of the five row-uniform ones that gave
            // every lane all ten entries --, multiplies it by its own dx, and the ten products are summed by row broadcasts: 22 VALU instructions where the
            // product form (round 4; -DTMPC_EXP_LXU_ALL rebuilds it) has 13, four LDS instructions fewer.  At eight trajectories per CU the LDS unit is the
            // busier one (65 % of the kernel time, most of it these sweeps): cfg 2 1.288 -> 1.343 M solves/s (+4.3 %, profiles/round6_saturated_levers_ab.jsonl).
            // Every kernel family takes it (the fast and the compact kernel of a shape stay bitwise equal); the sums associate differently: rounding level.
            const double xm = (li >= NU && li < NV) ? 1.0 : 0.0;
            const double p0 = xm * (lxu[0] * dx), p1 = xm * (lxu[1] * dx);
            double s0 = o.y0, s1 = o.y1;
            static_for<0, NX>([&](auto m_) { constexpr int m = decltype(m_)::value; s0 += bcast16<NU + m>(p0); s1 += bcast16<NU + m>(p1); });
            double dxs[NX];
            dxs[ZPSI - NU] = bcast16<ZPSI>(dx); dxs[ZV - NU] = bcast16<ZV>(dx);
            load_lxu(k + 1 < N ? k + 1 : N - 1);
#else
            double dxs[NX];
            static_for<0, NX>([&](auto m_) { constexpr int m = decltype(m_)::value; dxs[m] = bcast16<NU + m>(dx); });
            double s0 = o.y0, s1 = o.y1;
#pragma unroll
            for (int m = 0; m < NX; m++) { s0 = fma(lxu[2 * m], dxs[m], s0); s1 = fma(lxu[2 * m + 1], dxs[m], s1); }
            load_lxu(k + 1 < N ? k + 1 : N - 1);                           // (unconditional, clamped)
#endif
#ifdef TMPC_EXP_UNIFORM_PIVOTS
            const double u1 = -s1 * o.r1;
            const double u0 = (-s0 - o.l10 * u1) * o.r0;
#else
            const double u1 = bcast16<1>(-s1 * o.r1);                      // lane 1: -s1 1/L11
            const double u0 = bcast16<0>((-s0 - o.r1 * u1) * o.r0);        // lane 0: (-s0 - L10 u1) 1/L00   (its o.r1 is L10)
#endif
#ifdef TMPC_EXP_MERGE_FWD
            if (rowl) dv_own[k * NV] = li == 0 ? u0 : (li == 1 ? u1 : dx);      // (one store: lane j writes component j of dv_k -- du from lanes 0, 1, dx from lanes 2..6)
#else
            if (rowl && li == 0) { L.dv[k * NV] = u0; L.dv[k * NV + 1] = u1; }
            if (xl) dv_own[k * NV] = dx;
#endif
            const double dpsi = dxs[ZPSI - NU], dvv = dxs[ZV - NU];
            const double e_psi = o.a_psi - i_psi, e_v = o.a_v - i_v;
            dx = dx + e_psi * dpsi + e_v * dvv + o.b_a * u0 + o.b_w * u1 + o.rbi;   // lanes 2..6 meaningful
        };
        Ops oa, ob;
        load_stage(oa, 0);
        load_lxu(0);
        int k = 0;
        for (; k + 1 < N; k += 2) {
            load_stage(ob, k + 1);
            stage(oa, k);
            load_stage(oa, k + 2 < N ? k + 2 : N - 1);                    // unconditional (clamped), see the backward sweep
            stage(ob, k + 1);
        }
        if (k < N) stage(oa, k);
        if (rowl) L.dv[N * NV + li] = xl ? dx : 0.0;
    }
    SWEEP_T(SP_SOLVE_FWD);
}

template <int NTH, int CP = 0, bool SQ = false>
__device__ __forceinline__ void riccati_solve(const Lds &L, const Dims &d, int tid, int sw = 1)
{
    asm volatile("" : "+v"(tid));                    // opaque per call (see riccati_factor): the two solves of an iteration do not share address registers
    // two-wave variant: the vector sweeps run on the wave that did not factorise, so that with two trajectories' waves
    // sharing a SIMD pair the sequential work is spread over both SIMDs
    const bool sweeper = NTH == 64 || (tid >> 6) == sw;
    const int lane = NTH == 64 ? tid : (sweeper ? (tid & 63) : 64);
    SWEEP_COUNT(SP_CALLS_SOLVE);
    riccati_solve_pre<CP, SQ>(L, d, tid, NTH);
    __syncthreads();
    TMPC_PRIO_HIGH();
    riccati_sweeps_rows<CP, true>(L, d, lane, true, sweeper, [] { __syncthreads(); });
    TMPC_PRIO_LOW();
    __syncthreads();
    riccati_solve_post<CP, SQ, false>(L, d, tid, NTH);
    __syncthreads();
}

// The rest of the predictor solve after a VEC factorisation: forward sweep + stage-parallel closing loop.
template <int NTH, int CP = 0, bool SQ = false>
__device__ __forceinline__ void riccati_forward(const Lds &L, const Dims &d, int tid, int sw = 1)
{
    asm volatile("" : "+v"(tid));
    const bool sweeper = NTH == 64 || (tid >> 6) == sw;
    const int lane = NTH == 64 ? tid : (sweeper ? (tid & 63) : 64);
    SWEEP_COUNT(SP_CALLS_SOLVE);
    TMPC_PRIO_HIGH();
    riccati_sweeps_rows<CP, false>(L, d, lane, true, sweeper, [] {});
    TMPC_PRIO_LOW();
    __syncthreads();
#ifdef TMPC_EXP_PREDICTOR_POST
    riccati_solve_post<CP, SQ, SQ>(L, d, tid, NTH);      // (square-root form: the extra row left lx, not p)
    __syncthreads();
#endif
    // (round 6: no closing loop here.  dpi = P dx + p is the step of the dynamics multipliers, and the PREDICTOR's is never used -- the row passes between
    //  predictor and corrector read dv only, the update takes the corrector's dpi: one stage-parallel pass and one barrier per interior-point iteration less)
}

}  // namespace tmpc
