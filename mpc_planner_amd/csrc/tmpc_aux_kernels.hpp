// Kernels either side of the solve: FindBestPlanner and the multi-GPU records, the "next" rows f-1 (topology
// linearisation), f-2 (cross-tick warm start), f-3 (scenario -> halfspace reduction) and the stage-function debug kernel.
// Included by tmpc_solve.hip after the stage functions (tmpc_stage.hpp).
#pragma once

namespace tmpc {

// ---- FindBestPlanner on device (guidance_constraints.cpp:416-434) ---------------------------------
__global__ void tmpc_select_best_kernel(int first, int count, const double *pobj, const int *exit_code,
                                        const double *weight, const uint8_t *disabled, int *best_out)
{
    __shared__ double s_val[256];
    __shared__ int s_idx[256];
    double best = 1e10; int idx = -1;
    for (int i = threadIdx.x; i < count; i += blockDim.x) {
        const int g = first + i;
        if (disabled && disabled[i]) continue;
        if (exit_code[g] != 1) continue;
        const double o = weight ? pobj[g] * weight[i] : pobj[g];
        if (o < best) { best = o; idx = i; }       // ascending i per thread: strict '<' keeps the lowest index
    }
    s_val[threadIdx.x] = best; s_idx[threadIdx.x] = idx;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const double ov = s_val[threadIdx.x + s]; const int oi = s_idx[threadIdx.x + s];
            const double mv = s_val[threadIdx.x]; const int mi = s_idx[threadIdx.x];
            const bool take = (oi >= 0) && (mi < 0 || ov < mv || (ov == mv && oi < mi));
            if (take) { s_val[threadIdx.x] = ov; s_idx[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *best_out = s_idx[0];
}

// ---- multi-GPU records ------------------------------------------------------------
__global__ void tmpc_pack_records_kernel(int B, const double *pobj, const int *exit_code, const int *gid,
                                         const double *weight, tmpc_record *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    tmpc_record r;
    r.objective = weight ? pobj[i] * weight[i] : pobj[i];
    r.exit_code = exit_code[i];
    r.guidance_id = gid ? gid[i] : i;
    out[i] = r;
}

// one workgroup (64 lanes) per scene; records [n_ranks][n_scenes][per_rank]
__global__ void tmpc_select_best_records_kernel(const tmpc_record *rec, int n_ranks, int n_scenes, int per_rank, int *best_out)
{
    const int s = blockIdx.x;
    double best = 1e10; int idx = -1;
    const int total = n_ranks * per_rank;
    for (int g = threadIdx.x; g < total; g += 64) {           // ascending global index per lane
        const int rk = g / per_rank, t = g - rk * per_rank;
        const tmpc_record r = rec[((size_t)rk * n_scenes + s) * per_rank + t];
        if (r.exit_code == 1 && r.objective < best) { best = r.objective; idx = g; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const double ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(idx, o, 64);
        const bool take = (oi >= 0) && (idx < 0 || ov < best || (ov == best && oi < idx));
        if (take) { best = ov; idx = oi; }
    }
    if (threadIdx.x == 0) best_out[s] = idx;
}

// the winners' trajectories, one workgroup per set (guidance_constraints.cpp:382-384 for every set of the launch)
__global__ void tmpc_gather_best_kernel(const int *best, int set_size, int index_offset, int nx_doubles, int nu_doubles, const double *xtraj,
                                        const double *utraj, double *out_x, double *out_u)
{
    const int s = blockIdx.x;
    const int g = best[s], loc = g - index_offset;
    if (g >= 0 && (loc < 0 || loc >= set_size)) return;                 // another rank's winner
    const size_t b = (size_t)s * set_size + (g >= 0 ? loc : 0);
    const double nan = __builtin_nan("");
    for (int e = threadIdx.x; e < nx_doubles; e += blockDim.x) out_x[(size_t)s * nx_doubles + e] = g >= 0 ? xtraj[b * nx_doubles + e] : nan;
    for (int e = threadIdx.x; e < nu_doubles; e += blockDim.x) out_u[(size_t)s * nu_doubles + e] = g >= 0 ? utraj[b * nu_doubles + e] : nan;
}

// ---- f-1: LinearizedConstraints::update + setParameters on device (linearized_constraints.cpp:49-189) ----------
// one thread per (trajectory, stage)
// n_obs dynamic obstacles (rows 0 .. n_obs-1), then n_static static halfspaces per stage copied as they are (linearized_constraints.cpp:
// 107-123, `add_halfspaces`), then dummies up to n_lin.  obst_radius == nullptr: guidance mode, every disc has radius 1e-3 + robot_radius
// (:99, :140); otherwise the `_use_guidance == false` branch (:63-73): obstacle j's own radius + robot_radius, in the projection and in b.
__global__ void tmpc_linearize_topology_kernel(Dims d, int B, const double *x0, double *params, const double *obst,
                                               const int *scene_of, const double *state_x, double robot_radius,
                                               const uint8_t *is_original, int n_obs, const double *obst_radius, const double *stat, int n_static)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = d.N;
    if (e >= B * N) return;
    const int b = e / N, k = e - b * N;
    const int sc = scene_of[b];
    double *p = params + ((size_t)b * N + k) * d.npar;
    const double dummy_b = state_x[sc] + 100.0;                         // _dummy_b (:54)
    const bool dummy = (k == 0) || (is_original && is_original[b]);
    auto radius_of = [&](int j) { return (obst_radius ? obst_radius[(size_t)sc * n_obs + j] : 1e-3) + robot_radius; };   // (:99, :140)
    double px = x0[((size_t)b * (N + 1) + k) * ext_nv(d) + ZX], py = x0[((size_t)b * (N + 1) + k) * ext_nv(d) + ZY];
    const double *ob = obst + (size_t)sc * n_obs * N * 2;
    if (!dummy && n_obs > 0) {
        // projectToSafety (:130-148): at most 3 sweeps over the obstacles of ros_tools' Douglas-Rachford step with obstacle 0 as the
        // anchor.  ros_tools is not in the reference tree; restated from the published operator p <- (p + R_delta R_anchor p) / 2,
        // R = 2 P - I, P = nearest point outside the disc of radius r, applied when p is inside the obstacle's disc (same arithmetic
        // as modules.py::project_to_safety and the C++ DouglasRachford).  No FMA contraction: bit-equal to the host mirrors.
        const double ax0 = ob[(size_t)(k - 1) * 2], ay0 = ob[(size_t)(k - 1) * 2 + 1];          // anchor = obstacle 0 at step k-1
        auto outside = [&](double qx, double qy, double cx, double cy, double r, double &ox_, double &oy_) {
            const double dx = qx - cx, dy = qy - cy;
            const double dist = sqrt(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
            if (dist >= r) { ox_ = qx; oy_ = qy; }
            else if (dist > 1e-12) { const double s = r / dist; ox_ = __dadd_rn(cx, __dmul_rn(dx, s)); oy_ = __dadd_rn(cy, __dmul_rn(dy, s)); }
            else { ox_ = cx; oy_ = cy + r; }
        };
        for (int sweep = 0; sweep < 3; sweep++)
            for (int j = 0; j < n_obs; j++) {
                const double ox = ob[((size_t)j * N + (k - 1)) * 2], oy = ob[((size_t)j * N + (k - 1)) * 2 + 1];
                const double dx = px - ox, dy = py - oy;
                const double r = radius_of(j);
                if (sqrt(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy))) < r) {
                    double qx, qy, bx, by;
                    outside(px, py, ax0, ay0, r, qx, qy);
                    const double rx = __dmul_rn(2.0, qx) - px, ry = __dmul_rn(2.0, qy) - py;
                    outside(rx, ry, ox, oy, r, bx, by);
                    const double sx = __dmul_rn(2.0, bx) - rx, sy = __dmul_rn(2.0, by) - ry;
                    px = (px + sx) / 2.0; py = (py + sy) / 2.0;
                }
            }
    }
    for (int j = 0; j < d.n_lin; j++) {
        double a1 = 1.0, a2 = 0.0, bb = dummy_b;                        // _dummy_a1, _dummy_a2
        if (!dummy && j < n_obs) {
            const double ox = ob[((size_t)j * N + (k - 1)) * 2], oy = ob[((size_t)j * N + (k - 1)) * 2 + 1];
            const double dx = ox - px, dy = oy - py;
            const double dist = sqrt(dx * dx + dy * dy);
            a1 = dx / dist; a2 = dy / dist;
            bb = a1 * ox + a2 * oy - radius_of(j);
        } else if (!dummy && stat && j < n_obs + n_static) {            // static halfspaces of the stage, as given (:107-123)
            const double *hs = stat + (((size_t)sc * N + k) * n_static + (j - n_obs)) * 3;
            a1 = hs[0]; a2 = hs[1]; bb = hs[2];
        }
        p[ip_lin(d, j, 0)] = a1; p[ip_lin(d, j, 1)] = a2; p[ip_lin(d, j, 2)] = bb;
    }
}


// ---- f-3: scenario -> polygon on device (SH-MPC, BASELINE config 5) ------------------------------------------------
// The reference delegates this to the external scenario_module (scenario_constraints.cpp:47,76-79; source absent), so
// this restates the host mirror mpc_planner_amd/modules.py::scenario_halfspaces: for stage k >= 1 of trajectory b every
// sampled obstacle position o (n_pts = obstacles x scenarios of the trajectory's scene, prediction step k-1) gives the
// halfspace a = (o - p)/|o - p|, b = a.o - radius around the guess p = x0[b][k](x, y); the constraints of the stage are
// the halfspaces that form the boundary of the intersection polygon (all others are redundant), closest first, at most
// n_rows of them; unused rows and stage 0 get the dummy row (1, 0, x + 100) (decomp_constraints.cpp:153-169 pattern).
// One workgroup per (trajectory, stage), in two parts:
//   FILTER (free to be anything conservative: a halfspace whose boundary line misses the polygon of SOME of the
//      halfspaces -- the "seeds" -- cannot touch the smaller polygon of all of them, and dropping such halfspaces
//      changes neither the polygon nor which halfspaces are its edges).  On registers: the closest halfspace of each of
//      256 direction sectors is a seed (LDS table, atomicMin on an order-preserving key of the margin, then on the
//      index); a halfspace is clipped by the seed of its own sector and the two nearest seeds on either side only --
//      those decide, 5 clips instead of one per seed, the interval kept as two fractions compared by cross-multiplication
//      (no division); the survivors (~100 of 2048 on the SH-MPC scenes) are appended to a compact LDS list, on which a
//      second round with 2048 sectors and every seed clipping leaves ~11.
//   EDGE TEST (the definition, same per-pair arithmetic as the mirror): candidate i is an edge iff the piece of its line
//      inside all other candidates' halfspaces has positive length; its row index is its rank by (margin, sample index).
// The samples of a stage are contiguous ([scene][N][n_pts][2]): a coalesced stream shared through L2 by the scene's
// trajectories.  The edge test is written without FMA contraction, so the rows equal the mirror's bit for bit.
constexpr double POLY_EPS_PARALLEL = 1e-12, POLY_TOL_EDGE = 1e-9, POLY_SEED_MARGIN = 1e-6;
constexpr int POLY_IDX_MASK = 0x1fff, POLY_DROP_FLAG = 1 << 29;     // a candidate's index word during the second filter round
constexpr int POLY_SEC1 = 256;                        // direction sectors of round 1 (32 per octant, in angular order)
constexpr int POLY_NONE = 0x7fffffff;
constexpr int POLY_LIST_CAP = 1024;                 // candidate list of the first pass.  512 until round 4: on the cfg-5 scenes (8 x 256 samples per stage) enough units kept
                                                    // more than 512 candidates that the second pass cost another 250 us per tick; with 1024 none overflows there
                                                    // (second pass 4 us, the tick 2.06 -> 1.80 ms), and 28 KB of list still leave 4 workgroups per CU

__device__ __forceinline__ int poly_sector(double ax, double ay, int bins)        // octant x bins of min(|ax|,|ay|)/max: a partition of the directions
{
    const double u = fabs(ax), v = fabs(ay);
    const int oct = (ax < 0.0 ? 1 : 0) | (ay < 0.0 ? 2 : 0) | (v > u ? 4 : 0);
    const double t = (v > u ? u : v) / (v > u ? v : u);
    int sub = (int)(t * (double)bins);
    sub = sub > bins - 1 ? bins - 1 : sub;
    return oct * bins + sub;
}

__device__ __forceinline__ int poly_sector_angular(double ax, double ay)          // the same partition with 32 bins, numbered counter-clockwise
{
    const double u = fabs(ax), v = fabs(ay);
    const bool steep = v > u;
    const double t = (steep ? u : v) / (steep ? v : u);
    int bin = (int)(t * 32.0);
    bin = bin > 31 ? 31 : bin;
    const int quad = ax >= 0.0 ? (ay >= 0.0 ? 0 : 3) : (ay >= 0.0 ? 1 : 2);
    const bool second = ((quad & 1) == 0) == steep;           // second half (45..90 deg) of the quadrant
    const bool rising = ((quad & 1) == 0) != steep;           // t grows with the angle in this half
    return quad * 64 + (second ? 32 : 0) + (rising ? bin : 31 - bin);
}

__device__ __forceinline__ unsigned long long poly_key(double x)          // order-preserving map of a double to an unsigned integer
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// ---- SH-MPC scenario sampler (f-3; scenario_constraints.cpp:121-131: every solver draws its own scenarios,
// scenario_module.GetSampler().IntegrateAndTranslateToMeanAndVariance -- the scenario_module is absent from the reference tree, so this
// restates what the call's name and its inputs say: unit samples translated to the mean and the integrated variance of every prediction
// step).  One scenario = one joint draw for all obstacles over the horizon.  For obstacle m of solver (scene) q and scenario s:
// a mode of the obstacle's Gaussian mixture is drawn from its probabilities, then ONE standard-normal pair (xi1, xi2) places the obstacle
// on every step k of that mode:  o_k = mean_k + R(angle_k) (major_k xi1, minor_k xi2)  (PredictionStep: position, angle, major / minor
// radius = the integrated standard deviations, data_types.h:42-54).
// Counter-based and bit-reproducible: the random bits are a splitmix64 hash of (seed, solver, obstacle, scenario, draw), the normal
// deviates come from the inverse normal CDF (Acklam's rational approximation, |error| < 1.2e-9) evaluated with +, x, /, sqrt and a
// logarithm built from the same operations (det_log), no FMA contraction -- so modules.sample_scenarios reproduces every sample
// bit for bit and the polygon rows of a device-sampled tick can be checked exactly.  The rotation enters as (cos, sin) computed by
// the caller: trigonometric functions are not reproducible across libraries.
__host__ __device__ inline unsigned long long smp_mix(unsigned long long z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ inline double smp_uniform(unsigned long long key, unsigned long long ctr)      // in (0, 1), 53 bits
{
    const unsigned long long r = smp_mix(key + (ctr + 1ull) * 0x9E3779B97F4A7C15ull);
    return ((double)(r >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
__device__ inline double det_log(double x)                   // natural logarithm from +, x, / only (x > 0, normal): ~1e-14 relative
{
#pragma clang fp contract(off)
    int e;
    double m = frexp(x, &e);                                  // x = m 2^e, m in [0.5, 1)
    if (m < 0.70710678118654752) { m = m * 2.0; e = e - 1; }  // m in [sqrt(1/2), sqrt(2))
    const double f = (m - 1.0) / (m + 1.0), w = f * f;
    double p = 1.0 / 19.0;
    p = p * w + 1.0 / 17.0; p = p * w + 1.0 / 15.0; p = p * w + 1.0 / 13.0; p = p * w + 1.0 / 11.0; p = p * w + 1.0 / 9.0;
    p = p * w + 1.0 / 7.0; p = p * w + 1.0 / 5.0; p = p * w + 1.0 / 3.0; p = p * w + 1.0;
    return (double)e * 0.69314718055994531 + 2.0 * f * p;
}
__device__ inline double smp_normal(double u)                // inverse normal CDF (Acklam), u in (0, 1)
{
#pragma clang fp contract(off)
    const double a0 = -3.969683028665376e+01, a1 = 2.209460984245205e+02, a2 = -2.759285104469687e+02, a3 = 1.383577518672690e+02,
                 a4 = -3.066479806614716e+01, a5 = 2.506628277459239e+00;
    const double b0 = -5.447609879822406e+01, b1 = 1.615858368580409e+02, b2 = -1.556989798598866e+02, b3 = 6.680131188771972e+01,
                 b4 = -1.328068155288572e+01;
    const double c0 = -7.784894002430293e-03, c1 = -3.223964580411365e-01, c2 = -2.400758277161838e+00, c3 = -2.549732539343734e+00,
                 c4 = 4.374664141464968e+00, c5 = 2.938163982698783e+00;
    const double d0 = 7.784695709041462e-03, d1 = 3.224671290700398e-01, d2 = 2.445134137142996e+00, d3 = 3.754408661907416e+00;
    const double lo = 0.02425;
    if (u < lo) {
        const double q = sqrt(-2.0 * det_log(u));
        return (((((c0 * q + c1) * q + c2) * q + c3) * q + c4) * q + c5) / ((((d0 * q + d1) * q + d2) * q + d3) * q + 1.0);
    }
    if (u > 1.0 - lo) {
        const double q = sqrt(-2.0 * det_log(1.0 - u));
        return -(((((c0 * q + c1) * q + c2) * q + c3) * q + c4) * q + c5) / ((((d0 * q + d1) * q + d2) * q + d3) * q + 1.0);
    }
    const double q = u - 0.5, r = q * q;
    return (((((a0 * r + a1) * r + a2) * r + a3) * r + a4) * r + a5) * q / (((((b0 * r + b1) * r + b2) * r + b3) * r + b4) * r + 1.0);
}
// pred [n_solvers][M][n_modes][N][6] = (x, y, cos angle, sin angle, major, minor) per prediction step; prob [n_solvers][M][n_modes];
// out [n_solvers][N][M * S][2] (the layout tmpc_scenario_halfspaces reads: step k - 1 for stage k, sample m * S + s)
__global__ void tmpc_sample_scenarios_kernel(int N, int n_solvers, int M, int n_modes, int S, unsigned long long seed, const double *pred,
                                             const double *prob, double *out)
{
#pragma clang fp contract(off)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_solvers * M * S) return;
    const int s = idx % S, m = (idx / S) % M, q = idx / (S * M);
    const unsigned long long key = smp_mix(seed ^ smp_mix((unsigned long long)q + 0x51ED270B1ull));
    const unsigned long long ctr = ((unsigned long long)m * (unsigned long long)S + (unsigned long long)s) * 4ull;
    const double um = smp_uniform(key, ctr);
    const double *pr = prob + ((size_t)q * M + m) * n_modes;
    int mode = n_modes - 1;
    double cum = 0.0;
    for (int j = 0; j < n_modes; j++) { cum = cum + pr[j]; if (um < cum) { mode = j; break; } }
    const double xi1 = smp_normal(smp_uniform(key, ctr + 1ull)), xi2 = smp_normal(smp_uniform(key, ctr + 2ull));
    const double *ps = pred + (((size_t)q * M + m) * n_modes + mode) * N * 6;
    for (int k = 0; k < N; k++) {
        const double *e = ps + (size_t)k * 6;
        const double a = e[4] * xi1, b = e[5] * xi2;
        double *o = out + (((size_t)q * N + k) * ((size_t)M * S) + (size_t)m * S + s) * 2;
        o[0] = (e[0] + e[2] * a) - e[3] * b;
        o[1] = (e[1] + e[3] * a) + e[2] * b;
    }
}

// ---- scenario removal (SH-MPC discards a fixed number of scenarios before building the constraints and pays for it in the bound:
// the discarded scenarios count into the compression set, modules.scenario_risk(removed=...)).  The scenario_module's own policy is
// not in the reference tree; restated as the greedy rule of the method's paper: the scenarios that constrain the previous plan most --
// smallest clearance  min over obstacles m, stages k >= 1 of |o_{m,s,k-1} - p_k| - radius  -- go first (lowest scenario index on ties).
// One workgroup per trajectory: clearance per scenario in LDS, then n_discard rounds of a block-wide argmin.  discard [B][S] (1 = out).
__global__ __launch_bounds__(256) void tmpc_scenario_discard_kernel(Dims d, int B, const double *x0, const double *samples, int n_pts, int S,
                                                                 const int *scene_of, double radius, int n_discard, unsigned char *discard)
{
#pragma clang fp contract(off)
    extern __shared__ double s_clear[];                               // [S]
    __shared__ unsigned long long s_min;
    __shared__ int s_arg;
    const int b = blockIdx.x, tid = threadIdx.x, N = d.N;
    if (b >= B) return;
    const int sc = scene_of[b], M = n_pts / S;
    for (int s = tid; s < S; s += 256) {
        double best = 1e300;
        for (int k = 1; k < N; k++) {
            const double px = x0[((size_t)b * (N + 1) + k) * ext_nv(d) + ZX], py = x0[((size_t)b * (N + 1) + k) * ext_nv(d) + ZY];
            const double2 *o = reinterpret_cast<const double2 *>(samples) + ((size_t)sc * N + (k - 1)) * n_pts;
            for (int m = 0; m < M; m++) {
                const double2 q = o[m * S + s];
                const double dx = q.x - px, dy = q.y - py;
                const double c = sqrt(dx * dx + dy * dy) - radius;
                best = c < best ? c : best;
            }
        }
        s_clear[s] = best;
        discard[(size_t)b * S + s] = 0;
    }
    __syncthreads();
    for (int r = 0; r < n_discard && r < S; r++) {
        if (tid == 0) { s_min = ~0ull; s_arg = 0x7fffffff; }
        __syncthreads();
        for (int s = tid; s < S; s += 256) if (!discard[(size_t)b * S + s]) atomicMin(&s_min, poly_key(s_clear[s]));
        __syncthreads();
        for (int s = tid; s < S; s += 256) if (!discard[(size_t)b * S + s] && poly_key(s_clear[s]) == s_min) atomicMin(&s_arg, s);
        __syncthreads();
        if (tid == 0) discard[(size_t)b * S + s_arg] = 1;
        __syncthreads();
    }
}


// halfspace j = (aj1, aj2, dmj) clips the boundary line of halfspace i, q_i + t perp_i, to  c t <= rhs
struct PolyClip { double lo, hi; bool kill; };
__device__ __forceinline__ void poly_clip(PolyClip &w, double ai1, double ai2, double dmi, int i, double aj1, double aj2, double dmj, int j)
{
#pragma clang fp contract(off)
    const double c = aj1 * (-ai2) + aj2 * ai1;
    const double dot = aj1 * ai1 + aj2 * ai2;
    const double rhs = dmj - dmi * dot;
    // hi = min(hi, rhs / c), lo = max(lo, rhs / c); the division only when the bound may move: a quotient that is clearly (1e-9
    // relative) on the far side of the current bound leaves it as it is, so the result equals the plain min / max
    // (one code path for both signs of c -- round 6: as two branches the lanes of a wave diverged and every clip paid for two divisions; the expressions
    //  evaluated per case are the same: bit for bit the same bounds)
    const bool pos = c > POLY_EPS_PARALLEL;
    if (pos || c < -POLY_EPS_PARALLEL) {
        const double t = (pos ? w.hi : w.lo) * c;                 // c < 0:  rhs / c > lo  <=>  rhs < lo c
        if (!(rhs > t + 1e-9 * fabs(t))) {
            const double r = rhs / c;
            const double nhi = r < w.hi ? r : w.hi, nlo = r > w.lo ? r : w.lo;
            w.hi = pos ? nhi : w.hi; w.lo = pos ? w.lo : nlo;
        }
    } else if (dot > 0.0 ? (dmj < dmi || (dmj == dmi && j < i)) : rhs < 0.0) w.kill = true;   // parallel: same direction -> the closer one (lowest index) wins
}

// The filter's clip: the same interval kept as two fractions, hi = nh / dh and lo = nl / dl (dh, dl >= 0; 1/0 and -1/0 are the
// infinities), compared by cross-multiplication -- no division.  Nearly parallel seeds (|c| < 1e-6) are skipped, which keeps every
// quotient below ~1e8 m and the rounding of the products (1e-16 relative) far inside the filter's margin; skipping only loosens it.
struct PolyFrac { double nh, dh, nl, dl; };
__device__ __forceinline__ void poly_clip_fast(PolyFrac &w, double ai1, double ai2, double dmi, double aj1, double aj2, double dmj)
{
    const double c = aj1 * (-ai2) + aj2 * ai1;
    const double rhs = dmj - dmi * (aj1 * ai1 + aj2 * ai2);
    if (c > 1e-6) { if (rhs * w.dh < w.nh * c) { w.nh = rhs; w.dh = c; } }
    else if (c < -1e-6) { if (-rhs * w.dl > w.nl * -c) { w.nl = -rhs; w.dl = -c; } }
}
__device__ __forceinline__ bool poly_frac_alive(const PolyFrac &w) { return w.nh * w.dl - w.nl * w.dh >= -POLY_SEED_MARGIN * (w.dh * w.dl); }   // (>=: both sides are 0 while a bound is infinite)

#ifdef TMPC_POLY_PROFILE
#define POLY_T(n) do { __syncthreads(); if (tid == 0) { const long long t_ = wall_clock64(); if (t_ - t_prev > 2500) printf("poly unit %d seg %d: %lld (x10 ns) nk %d ns %d ne %d\n", unit, n, t_ - t_prev, s_nk, s_ns, s_ne); t_prev = wall_clock64(); } } while (0)
#else
#define POLY_T(n) do { } while (0)
#endif
// one (trajectory, stage) = `unit` by one workgroup of 256 threads
__device__ __forceinline__ void poly_stage(int unit, const Dims &d, int B, const double *x0, double *params, const double *samples, int n_pts,
                                           int n_rows, const int *scene_of, const double *state_x, double radius, double disc_offset,
                                           int *row_sample, int cap, int *overflow, int *empty_stages, const unsigned char *discard, int n_scen)
{
#pragma clang fp contract(off)
    extern __shared__ double s_dyn[];                                    // the candidates, compact: normal, margin, sample index (~index: not an edge)
    __shared__ unsigned long long s_best[POLY_SEC1];
    __shared__ double sd_ax[POLY_SEC1], sd_ay[POLY_SEC1], sd_dm[POLY_SEC1];
    __shared__ int s_seed[POLY_SEC1], s_pick[POLY_SEC1];
    __shared__ short s_next[POLY_SEC1], s_prev[POLY_SEC1];              // nearest sector with a seed, counter-clockwise / clockwise (-1: none)
    __shared__ int s_nk, s_ns, s_nw, s_ne;
    __shared__ unsigned long long s_mask[4];
    double *c_ax = s_dyn, *c_ay = s_dyn + cap, *c_dm = s_dyn + 2 * (size_t)cap;
    int *c_idx = reinterpret_cast<int *>(s_dyn + 3 * (size_t)cap);
    const int N = d.N;
    const int b = unit / N, k = unit - b * N;
    const int sc = scene_of[b];
    double *p = params + ((size_t)b * N + k) * d.npar;
    const int tid = threadIdx.x;
    int *which = row_sample + (size_t)unit * n_rows;                  // the sample behind each row (-1: dummy), for tmpc_scenario_support
    if (tid == 0) p[ip_disc_offset(d)] = disc_offset;
    if (k == 0) {
        if (tid < n_rows) { p[ip_slk(d, tid, 0)] = 1.0; p[ip_slk(d, tid, 1)] = 0.0; p[ip_slk(d, tid, 2)] = state_x[sc] + 100.0; which[tid] = -1; }
        return;
    }
    long long t_prev = 0; (void)t_prev;
#ifdef TMPC_POLY_PROFILE
    t_prev = wall_clock64();
#endif
    s_best[tid] = ~0ull; s_seed[tid] = POLY_NONE;
    if (tid == 0) { s_nk = 0; s_ne = 0; s_ns = 0; }
    __syncthreads();
    const double px = x0[((size_t)b * (N + 1) + k) * ext_nv(d) + ZX], py = x0[((size_t)b * (N + 1) + k) * ext_nv(d) + ZY];
    const double2 *o = reinterpret_cast<const double2 *>(samples) + ((size_t)sc * N + (k - 1)) * n_pts;
    auto halfspace = [&](int i, double &ax, double &ay, double &dm) {
        const double2 q = o[i];
        const double dx = q.x - px, dy = q.y - py;
        const double dist = sqrt(dx * dx + dy * dy);
        ax = dx / dist; ay = dy / dist; dm = dist - radius;
    };
    // ---- round 1 on registers: the first CACHE halfspaces of a thread are kept, the rest (n_pts > 2048) recomputed per pass
    constexpr int CACHE = 8;
    double r_ax[CACHE], r_ay[CACHE], r_dm[CACHE]; int r_sec[CACHE];
    // discarded scenarios (tmpc_scenario_discard; scenario of sample i = i % n_scen) do not exist for this trajectory's polygons
    auto alive = [&](int i) { return discard == nullptr || discard[(size_t)b * n_scen + i % n_scen] == 0; };
    unsigned r_alive = 0;
#pragma unroll
    for (int c = 0; c < CACHE; c++)
        if (tid + c * 256 < n_pts && alive(tid + c * 256)) { r_alive |= 1u << c; halfspace(tid + c * 256, r_ax[c], r_ay[c], r_dm[c]); r_sec[c] = poly_sector_angular(r_ax[c], r_ay[c]); }
    auto every_sample = [&](auto &&f) {
#pragma unroll
        for (int c = 0; c < CACHE; c++) if (r_alive >> c & 1) f(tid + c * 256, r_ax[c], r_ay[c], r_dm[c], r_sec[c]);
        for (int i = tid + CACHE * 256; i < n_pts; i += 256) if (alive(i)) { double ax, ay, dm; halfspace(i, ax, ay, dm); f(i, ax, ay, dm, poly_sector_angular(ax, ay)); }
    };
    POLY_T(0);
    every_sample([&](int, double, double, double dm, int sec) { atomicMin(&s_best[sec], poly_key(dm)); });
    __syncthreads();
    every_sample([&](int i, double, double, double dm, int sec) { if (poly_key(dm) == s_best[sec]) atomicMin(&s_seed[sec], i); });
    __syncthreads();
    every_sample([&](int i, double ax, double ay, double dm, int sec) { if (s_seed[sec] == i) { sd_ax[sec] = ax; sd_ay[sec] = ay; sd_dm[sec] = dm; } });
    {   // nearest sectors with a seed on either side of sector tid: from the four ballot words of the table (wave w holds sectors 64w..64w+63)
        const unsigned long long mine = __ballot(s_seed[tid] != POLY_NONE);
        if ((tid & 63) == 0) s_mask[tid >> 6] = mine;
        __syncthreads();
        const int w0 = tid >> 6, bit = tid & 63;
        int nx = -1, pv = -1;
#pragma unroll
        for (int step = 0; step <= 4; step++) {                   // own word above the bit, the three other words, own word below the bit
            const int wn = (w0 + step) & 3, wp = (w0 - step) & 3;
            unsigned long long xn = s_mask[wn], xp = s_mask[wp];
            if (step == 0) { xn &= bit == 63 ? 0ull : ~0ull << (bit + 1); xp &= (1ull << bit) - 1ull; }
            if (step == 4) { xn &= (1ull << bit) - 1ull; xp &= bit == 63 ? 0ull : ~0ull << (bit + 1); }
            if (nx < 0 && xn) nx = wn * 64 + __ffsll((long long)xn) - 1;
            if (pv < 0 && xp) pv = wp * 64 + 63 - __clzll((long long)xp);
        }
        s_next[tid] = (short)nx; s_prev[tid] = (short)pv;
    }
    __syncthreads();
    POLY_T(1);
    // a halfspace is clipped by the seed of its own sector and the two nearest seeds on either side; the interval only shrinks,
    // so once it is empty the halfspace is out, whatever other seeds would do
    every_sample([&](int i, double ax, double ay, double dm, int sec) {
        const int n1 = s_next[sec], p1 = s_prev[sec];
        const int at[5] = {sec, n1, p1, n1 >= 0 ? s_next[n1] : -1, p1 >= 0 ? s_prev[p1] : -1};
        PolyFrac w = {1.0, 0.0, -1.0, 0.0};
#pragma unroll
        for (int q = 0; q < 5; q++)
            if (at[q] >= 0 && poly_frac_alive(w) && s_seed[at[q]] != i) poly_clip_fast(w, ax, ay, dm, sd_ax[at[q]], sd_ay[at[q]], sd_dm[at[q]]);
        if (poly_frac_alive(w)) { const int e = atomicAdd(&s_nk, 1); if (e < cap) { c_ax[e] = ax; c_ay[e] = ay; c_dm[e] = dm; c_idx[e] = i; } }
    });
    __syncthreads();
    POLY_T(2);
    if (s_nk > cap) { if (tid == 0) overflow[1 + atomicAdd(&overflow[0], 1)] = unit; return; }          // (uniform: s_nk is final after the barrier; only in the first pass)
    // ---- filter, second round, on the list (~100 -> ~11): seeds = the closest candidate of each of 2048 sectors (pairwise comparison:
    //      the list is short; the sector rides in bits 13-23 of the index word), every seed clips every candidate -- G threads per
    //      candidate, each taking every G-th seed, the fractions combined over the G lanes --, then the list is compacted in place
    {
        const int nk1 = s_nk;
        for (int ci = tid; ci < nk1; ci += 256) c_idx[ci] |= poly_sector(c_ax[ci], c_ay[ci], 256) << 13;
        if (tid == 0) { s_ns = 0; s_nw = 0; }
        __syncthreads();
        // Seeds by table (round 6): one pass per octant over the 256 bins of the octant -- keys in s_best, the lowest sample index among equal keys in s_pick --
        // O(nk1) where the pairwise comparison of rounds 2-5 was O(nk1^2 / 256): the units that keep 300 - 600 candidates (stage 1 of a horizon: the 256
        // scenarios of an obstacle still nearly coincide) took 400 us and with them the whole launch, 27 % of cfg 5's step.  (Both forms behind a branch on nk1
        // cost the kernel 20 B of scratch more and the saturated launch 6 %: one form.)
        for (int oct = 0; oct < 8; oct++) {
            s_best[tid] = ~0ull; s_pick[tid] = POLY_NONE;
            __syncthreads();
            for (int ci = tid; ci < nk1; ci += 256) {
                const int sec = c_idx[ci] >> 13;
                if ((sec >> 8) == oct) atomicMin(&s_best[sec & 255], poly_key(c_dm[ci]));
            }
            __syncthreads();
            for (int ci = tid; ci < nk1; ci += 256) {
                const int me = c_idx[ci], sec = me >> 13;
                if ((sec >> 8) == oct && poly_key(c_dm[ci]) == s_best[sec & 255]) atomicMin(&s_pick[sec & 255], me & POLY_IDX_MASK);
            }
            __syncthreads();
            for (int ci = tid; ci < nk1; ci += 256) {
                const int me = c_idx[ci], i = me & POLY_IDX_MASK, sec = me >> 13;
                if ((sec >> 8) == oct && s_pick[sec & 255] == i) {   // (the round-1 seed table is free; seeds beyond its size are not used: a looser filter)
                    const int e = atomicAdd(&s_ns, 1);
                    if (e < POLY_SEC1) { sd_ax[e] = c_ax[ci]; sd_ay[e] = c_ay[ci]; sd_dm[e] = c_dm[ci]; s_seed[e] = i; }
                }
            }
            __syncthreads();
        }
        const int ns = s_ns < POLY_SEC1 ? s_ns : POLY_SEC1;
        int G = 1;
        while (G < 64 && 2 * G * nk1 <= 256) G *= 2;
        const int g = tid & (G - 1);
        for (int base = 0; base < nk1; base += 256 / G) {
            const int ci = base + tid / G;
            const bool valid = ci < nk1;
            const double ax = valid ? c_ax[ci] : 1.0, ay = valid ? c_ay[ci] : 0.0, dm = valid ? c_dm[ci] : 0.0;
            const int i = valid ? c_idx[ci] & POLY_IDX_MASK : -1;
            PolyFrac w = {1.0, 0.0, -1.0, 0.0};
            if (valid)
                for (int q = g; q < ns; q += G)
                    if (s_seed[q] != i) poly_clip_fast(w, ax, ay, dm, sd_ax[q], sd_ay[q], sd_dm[q]);
            for (int m = 1; m < G; m <<= 1) {                     // the lower upper bound, the higher lower bound
                const double onh = __shfl_xor(w.nh, m, 64), odh = __shfl_xor(w.dh, m, 64), onl = __shfl_xor(w.nl, m, 64), odl = __shfl_xor(w.dl, m, 64);
                if (onh * w.dh < w.nh * odh) { w.nh = onh; w.dh = odh; }
                if (onl * w.dl > w.nl * odl) { w.nl = onl; w.dl = odl; }
            }
            if (valid && g == 0 && !poly_frac_alive(w)) c_idx[ci] |= POLY_DROP_FLAG;
        }
        __syncthreads();
        POLY_T(4);
        for (int c0 = 0; c0 < nk1; c0 += 256) {                   // in-place compaction, chunk-wise: read, then append below the chunk
            const int ci = c0 + tid;
            double ax = 0.0, ay = 0.0, dm = 0.0; int me = POLY_DROP_FLAG;
            if (ci < nk1) { ax = c_ax[ci]; ay = c_ay[ci]; dm = c_dm[ci]; me = c_idx[ci]; }
            __syncthreads();
            if (!(me & POLY_DROP_FLAG)) { const int e = atomicAdd(&s_nw, 1); c_ax[e] = ax; c_ay[e] = ay; c_dm[e] = dm; c_idx[e] = me & POLY_IDX_MASK; }
            __syncthreads();
        }
        if (tid == 0) s_nk = s_nw;
        __syncthreads();
    }
    POLY_T(5);
    // ---- edge test among the candidates, G threads per candidate (G = 256 / candidates, rounded down to a power of two): each takes
    //      every G-th of the other candidates, then min / max / or across the G lanes.  Every candidate clips every other, like in
    //      the mirror, and a quotient is the same whichever thread computes it, so the split does not change the result.  A
    //      candidate found redundant stays in the list as ~index (readers decode it): it still clips the others.
    const int nk = s_nk;
    // G by cost (round 6): trips x (clips per thread + combination) = ceil(nk G / 256) (ceil(nk / G) + ...).  One thread per candidate (the rule until round 5 for nk > 128) leaves a long
    // list -- stage 1 of a horizon, where the 256 scenarios of an obstacle still nearly coincide and ~300 nearly parallel candidates pass the filter -- with two
    // trips of nk sequential clips, the second with 44 busy threads: the tail of the whole launch.  (A pre-test against the seeds alone was tried and prunes nothing
    // there: the near-duplicates of an edge are near-edges.)
    int G = 1;
    {
        int best = ((nk + 255) / 256) * nk * 100;                    // (a clip ~ 100 instructions, a combination step over the G lanes ~ 15)
        for (int cand = 2, lg = 1; cand <= 64; cand *= 2, lg++) {
            const int cost = ((nk * cand + 255) / 256) * (((nk + cand - 1) / cand) * 100 + 15 * lg);
            if (cost < best) { best = cost; G = cand; }
        }
    }
    const int g = tid & (G - 1);
    for (int base = 0; base < nk; base += 256 / G) {                 // (several trips when nk G > 256)
        const int ci = base + tid / G;
        const bool valid = ci < nk;
        const int i = valid ? c_idx[ci] : -1;                         // (entry ci is rewritten only below, by this group)
        const double ai1 = valid ? c_ax[ci] : 1.0, ai2 = valid ? c_ay[ci] : 0.0, dmi = valid ? c_dm[ci] : 0.0;
        PolyClip w = {-HUGE_VAL, HUGE_VAL, false};
        if (valid)
            for (int cj = g; cj < nk; cj += G) {
                int j = c_idx[cj];
                j = j < 0 ? ~j : j;
                if (cj != ci) poly_clip(w, ai1, ai2, dmi, i, c_ax[cj], c_ay[cj], c_dm[cj], j);
            }
        int kill = w.kill ? 1 : 0;
        for (int m = 1; m < G; m <<= 1) {
            const double hi = __shfl_xor(w.hi, m, 64), lo = __shfl_xor(w.lo, m, 64);
            w.hi = hi < w.hi ? hi : w.hi; w.lo = lo > w.lo ? lo : w.lo; kill |= __shfl_xor(kill, m, 64);
        }
        if (valid && g == 0) {
            if (!kill && w.hi - w.lo > POLY_TOL_EDGE) atomicAdd(&s_ne, 1);
            else c_idx[ci] = ~i;
        }
    }
    __syncthreads();
    POLY_T(6);
    // ---- rows: an edge's row is its rank by (margin, sample index)
    for (int ci = tid; ci < nk; ci += 256) {
        const int i = c_idx[ci];
        if (i < 0) continue;
        const double dmi = c_dm[ci];
        int rank = 0;
        for (int cj = 0; cj < nk; cj++) {
            const int j = c_idx[cj];
            if (j >= 0 && (c_dm[cj] < dmi || (c_dm[cj] == dmi && j < i))) rank++;
        }
        if (rank < n_rows) {
            const double2 q = o[i];
            const double a1 = c_ax[ci], a2 = c_ay[ci];
            p[ip_slk(d, rank, 0)] = a1; p[ip_slk(d, rank, 1)] = a2; p[ip_slk(d, rank, 2)] = a1 * q.x + a2 * q.y - radius;
            which[rank] = i;
        }
    }
    POLY_T(7);
    int n_real = s_ne;
    if (s_ne == 0 && n_pts > 0) {
        // EMPTY polygon: the halfspaces contradict each other (the guess sits in the overlap of inflated discs on opposite sides).
        // Leaving the stage unconstrained would certify the most dangerous geometry as safe, so the stage keeps the n_rows CLOSEST
        // halfspaces of all samples (lowest index on ties) -- contradictory rows: the QP is infeasible, or pays slack, and the solve
        // reports it -- and the unit is counted in empty_stages[b] (tmpc_scenario_empty_stages), which the callers turn into "not
        // eligible".  n_rows rounds of a block-wide argmin over (margin, index) above the previous pick; a rare path.
        unsigned long long prev_key = 0ull; int prev_idx = -1;
        const int rounds = n_rows < n_pts ? n_rows : n_pts;
        for (int r = 0; r < rounds; r++) {
            if (tid == 0) { s_best[0] = ~0ull; s_seed[0] = POLY_NONE; }
            __syncthreads();
            every_sample([&](int i, double, double, double dm, int) {
                const unsigned long long key = poly_key(dm);
                if (prev_idx < 0 || key > prev_key || (key == prev_key && i > prev_idx)) atomicMin(&s_best[0], key);
            });
            __syncthreads();
            every_sample([&](int i, double, double, double dm, int) {
                const unsigned long long key = poly_key(dm);
                if (key == s_best[0] && (prev_idx < 0 || key > prev_key || i > prev_idx)) atomicMin(&s_seed[0], i);
            });
            __syncthreads();
            every_sample([&](int i, double ax, double ay, double, int) {
                if (i == s_seed[0]) {
                    const double2 q = o[i];
                    p[ip_slk(d, r, 0)] = ax; p[ip_slk(d, r, 1)] = ay; p[ip_slk(d, r, 2)] = ax * q.x + ay * q.y - radius;
                    which[r] = i;
                }
            });
            prev_key = s_best[0]; prev_idx = s_seed[0];
            __syncthreads();
        }
        n_real = rounds;
        if (tid == 0 && empty_stages) atomicAdd(&empty_stages[b], 1);
    }
    if (tid < n_rows && tid >= n_real) { p[ip_slk(d, tid, 0)] = 1.0; p[ip_slk(d, tid, 1)] = 0.0; p[ip_slk(d, tid, 2)] = state_x[sc] + 100.0; which[tid] = -1; }
}


// The candidate list (`cap` entries of dynamic LDS) is what bounds the workgroups per CU, and the filter leaves ~100 candidates of 2048
// samples: the first pass (one workgroup per unit) runs with a short list (POLY_LIST_CAP) and records the rare unit whose candidates do
// not fit (overflow[0] = their number, overflow[1..] = the units); the second pass, a few workgroups with room for every sample, redoes
// only those.
__global__ __launch_bounds__(256, 4) void tmpc_scenario_halfspaces_kernel(Dims d, int B, const double *x0, double *params,
                                                                       const double *samples, int n_pts, int n_rows,
                                                                       const int *scene_of, const double *state_x,
                                                                       double radius, double disc_offset, int *row_sample, int cap, int *overflow, int second_pass,
                                                                       int *empty_stages, const unsigned char *discard, int n_scen)
{
    const int n_units = second_pass ? overflow[0] : B * d.N;
    for (int q = blockIdx.x; q < n_units; q += gridDim.x) {      // (first pass: one unit per workgroup)
        poly_stage(second_pass ? overflow[1 + q] : q, d, B, x0, params, samples, n_pts, n_rows, scene_of, state_x, radius, disc_offset, row_sample, cap, overflow, empty_stages, discard, n_scen);
        __syncthreads();                                          // (the LDS tables are reused by the next unit)
    }
}


// Support of a scenario program's solution (the scenarios whose constraints are active at it; SH-MPC bounds the collision
// probability of the plan through the size of this set -- ScenarioSolver::support / ::status, scenario_constraints.h:38-40; the
// scenario_module that fills them is absent, so this restates the definition of the method the reference cites, README.md:22):
// scenario s = sample index % n_scenarios (samples are [obstacle][scenario]: one scenario is one joint draw of all obstacles over
// the horizon); a row is active when  a.p_disc - (b + slack) >= -tol  at the solution.  One wave per trajectory; the set is a
// bitmask in LDS.  support[b] = number of distinct active scenarios, active_rows[b] = number of active rows.
__global__ __launch_bounds__(64) void tmpc_scenario_support_kernel(Dims d, int B, const double *params, const double *xtraj, const int *row_sample,
                                                                  int n_rows, int n_scenarios, double tol, int *support, int *active_rows)
{
#pragma clang fp contract(off)
    __shared__ unsigned int s_mask[256];                                // up to 8192 scenarios
    __shared__ int s_rows;
    const int b = blockIdx.x, tid = threadIdx.x, N = d.N, nxe = ext_nx(d);
    if (b >= B) return;
    for (int w = tid; w < 256; w += 64) s_mask[w] = 0u;
    if (tid == 0) s_rows = 0;
    __syncthreads();
    for (int e = tid; e < (N - 1) * n_rows; e += 64) {
        const int k = 1 + e / n_rows, r = e - (k - 1) * n_rows;
        const int smp = row_sample[((size_t)b * N + k) * n_rows + r];
        if (smp < 0) continue;
        const double *p = params + ((size_t)b * N + k) * d.npar;
        const double *x = xtraj + ((size_t)b * (N + 1) + k) * nxe;
        const double off = p[ip_disc_offset(d)], slack = d.slack ? x[NX] : 0.0;
        const double px = x[0] + off * cos(x[2]), py = x[1] + off * sin(x[2]);
        const double h = p[ip_slk(d, r, 0)] * px + p[ip_slk(d, r, 1)] * py - (p[ip_slk(d, r, 2)] + slack);
        if (h >= -tol) {
            const int s = smp % n_scenarios;
            atomicOr(&s_mask[s >> 5], 1u << (s & 31));
            atomicAdd(&s_rows, 1);
        }
    }
    __syncthreads();
    int cnt = 0;
    for (int w = tid; w < 256; w += 64) cnt += __popc(s_mask[w]);
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    if (tid == 0) { support[b] = cnt; if (active_rows) active_rows[b] = s_rows; }
}


// ---- f-2: cross-tick state on device ------------------------------------------------------------------------------
// Warm start of the next tick from the previous tick's solution, without a host round trip.  One thread per
// (trajectory, node).  mode[b]:
//   0  leave x0[b] alone (the caller loads a guidance trajectory, tmpc_init_with_guidance)
//   1  Solver::initializeWarmstart(state, shift = true)   (acados_solver_interface.cpp:344-364):
//        [state, out_2, ..., out_{N-1}, out_{N-1}, out_{N-1}]
//   2  Solver::initializeWarmstart(state, shift = false)  (:365-375): x0[k] = out_k for k < N
//   3  Solver::initializeWithBraking(state)               (:303-342): constant deceleration roll-out
// out_k is the previous solution of trajectory src[b] (src = NULL: b itself -- a planner re-uses its own last output,
// guidance_constraints.cpp:310-311).  xinit[b] <- state[b] (Solver::setXinit).
// In mode 1 the reference fills the INPUT entries of node 0 from State::get(<input name>), which indexes the state
// vector at -2 / -1 (state.cpp:21-24: index - nu) -- an out-of-bounds read; this restatement writes 0 there.
__global__ void tmpc_warmstart_kernel(Dims d, int B, const double *state, const int *mode, const int *src, const double *xtraj,
                                      const double *utraj, double *x0, double *xinit, double decel)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = d.N, nxe = ext_nx(d), nve = ext_nv(d);
    if (e >= B * (N + 1)) return;
    const int b = e / (N + 1), k = e - b * (N + 1);
    const double *st = state + (size_t)b * nxe;
    if (k == 0) for (int i = 0; i < nxe; i++) xinit[(size_t)b * nxe + i] = st[i];
    const int m = mode ? mode[b] : 1;
    if (m == 0) return;
    double *z = x0 + ((size_t)b * (N + 1) + k) * nve;
    const int sb = src ? src[b] : b;
    const double *xo = xtraj + (size_t)sb * (N + 1) * nxe, *uo = utraj + (size_t)sb * N * NU;
    if (m == 1) {
        if (k == 0) {
            for (int i = 0; i < NU; i++) z[i] = 0.0;
            for (int i = 0; i < nxe; i++) z[NU + i] = st[i];
        } else {
            const int ko = (k >= N - 1) ? N - 1 : k + 1;
            for (int i = 0; i < NU; i++) z[i] = uo[ko * NU + i];
            for (int i = 0; i < nxe; i++) z[NU + i] = xo[ko * nxe + i];
        }
    } else if (m == 2) {
        if (k < N) {
            for (int i = 0; i < NU; i++) z[i] = uo[k * NU + i];
            for (int i = 0; i < nxe; i++) z[NU + i] = xo[k * nxe + i];
        }
    } else if (m == 3) {
        double x = st[0], y = st[1], v = st[3], spline = st[4];
        const double psi = st[2], a = -fabs(decel);
        double sn, cs;
        sincos(psi, &sn, &cs);
        for (int j = 1; j <= k; j++) {                       // same recursion order as the reference's loop (:322-331)
            x += v * d.dt * cs; y += v * d.dt * sn; spline += v * d.sdt;      // (sdt = 0: the model without a spline state keeps its padded slot)
            v += a * d.dt; v = fmax(v, 0.0);
        }
        z[ZA] = a; z[ZW] = 0.0; z[ZX] = x; z[ZY] = y; z[ZPSI] = psi; z[ZV] = v; z[ZS] = spline;
        for (int i = NX; i < nxe; i++) z[NU + i] = st[i];    // initializeWithState: remaining states = initial state
    }
}

// GuidanceConstraints::initializeSolverWithGuidance (guidance_constraints.cpp:390-414): k = 1..N-1: x, y from the guidance
// trajectory at t = k dt, psi = atan2(vy, vx), v = |vel|.  gpos / gvel: [B][N+1][2]; enabled[b] == 0 skips b.
__global__ void tmpc_init_with_guidance_kernel(Dims d, int B, const double *gpos, const double *gvel, const uint8_t *enabled, double *x0)
{
#pragma clang fp contract(off)
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = d.N;
    if (e >= B * (N + 1)) return;
    const int b = e / (N + 1), k = e - b * (N + 1);
    if (k < 1 || k > N - 1 || (enabled && !enabled[b])) return;
    double *z = x0 + ((size_t)b * (N + 1) + k) * ext_nv(d);
    const double vx = gvel[(size_t)e * 2], vy = gvel[(size_t)e * 2 + 1];
    z[ZX] = gpos[(size_t)e * 2]; z[ZY] = gpos[(size_t)e * 2 + 1];
    z[ZPSI] = atan2(vy, vx);
    z[ZV] = sqrt(vx * vx + vy * vy);
}

// ---- persistent state: a solve that did not succeed resets the reference's capsule (Solver_acados_reset + reset_qp_memory,
// acados_solver_interface.cpp:187-191): the slot's multipliers go back to zero (the primal iterate is overwritten by the next
// loadWarmstart anyway) ----
__global__ void tmpc_state_finalize_kernel(int n_pi, int n_lam, const int *__restrict__ exit_code, double *__restrict__ pi,
                                           double *__restrict__ lamh, const int *__restrict__ slot)
{
    const int b = blockIdx.x;
    if (exit_code[b] == 1) return;
    const int sb = slot ? slot[b] : b;                      // the batch entry's state slot (tmpc_set_slots)
    for (int e = threadIdx.x; e < n_pi; e += blockDim.x) pi[(size_t)sb * n_pi + e] = 0.0;
    for (int e = threadIdx.x; e < n_lam; e += blockDim.x) lamh[(size_t)sb * n_lam + e] = 0.0;
}

// ---- debug: stage functions on device -----------------------------------------------------------
__global__ void tmpc_debug_eval_kernel(Dims d, int n, const double *z, const double *p, const double *pi, const double *lamh,
                                       double *cost, double *cgrad, double *chess, double *hval, double *hjac,
                                       double *xnext, double *xjac, double *lag, double *mir)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int nh = d.n_up + d.M;
    const double *ze = z + (size_t)e * ext_nv(d); const double *pe = p + (size_t)e * d.npar;
    double zz[NV];
    for (int i = 0; i < NV; i++) zz[i] = ze[i];
    const double slack = d.slack ? ze[NV] : 0.0;
    // rows are reported in the reference's order [topology | ellipsoids | slack rows] (module order)
    auto ext = [&](int r) { return r < d.n_lin ? r : (r < d.n_up ? d.M + r : r - d.n_slk); };
    double Wc[NV][NV], cg[NV];
    for (int i = 0; i < NV; i++) for (int j = 0; j < NV; j++) Wc[i][j] = 0.0;
#ifndef TMPC_GENERATED_STAGE
    if (d.cost_model == 1) {
        CostOutCA co;
        cost_eval_ca(d, zz, pe, 1, co, true, slack);
        cost[e] = co.val;
        cost_add_hessian_ca(co, 1.0, Wc);
        for (int i = 0; i < NV; i++) cg[i] = co.g[i];
    } else
#endif
    {
        CostOut co;
        cost_eval(d, zz, pe, 1, co, true, slack);
        cost[e] = co.val;
        cost_add_hessian(co, 1.0, Wc);
        for (int i = 0; i < NV; i++) cg[i] = co.g[i];
    }
    for (int i = 0; i < NV; i++) {
        cgrad[(size_t)e * NV + i] = cg[i];
        for (int j = 0; j < NV; j++) chess[(size_t)e * NV * NV + i * NV + j] = Wc[i][j];
    }
    double W[NV][NV], g[NV], BA[NX * NV], xn[NX];
    auto lam = [&](int r) { return lamh ? lamh[(size_t)e * nh + ext(r)] : 0.0; };
    auto sink = [&](int r, const RowOut &ro) {
        hval[(size_t)e * nh + ext(r)] = ro.h;
        double *J = hjac + ((size_t)e * nh + ext(r)) * NV;
        for (int i = 0; i < NV; i++) J[i] = 0.0;
        J[ZX] = ro.gx; J[ZY] = ro.gy; J[ZPSI] = ro.gp;
    };
#ifndef TMPC_GENERATED_STAGE
    if (d.cost_model == 1 && d.row_model == 1) stage_linearise<3>(d, zz, pe, 1, pi ? pi[(size_t)e * NX] : 0.0, pi ? pi[(size_t)e * NX + 1] : 0.0, lam, sink, W, g, BA, xn, slack);
    else if (d.cost_model == 1) stage_linearise<1>(d, zz, pe, 1, pi ? pi[(size_t)e * NX] : 0.0, pi ? pi[(size_t)e * NX + 1] : 0.0, lam, sink, W, g, BA, xn, slack);
    else if (d.row_model == 1) stage_linearise<2>(d, zz, pe, 1, pi ? pi[(size_t)e * NX] : 0.0, pi ? pi[(size_t)e * NX + 1] : 0.0, lam, sink, W, g, BA, xn, slack);
    else
#endif
    stage_linearise(d, zz, pe, 1, pi ? pi[(size_t)e * NX] : 0.0, pi ? pi[(size_t)e * NX + 1] : 0.0, lam, sink, W, g, BA, xn, slack);
    for (int i = 0; i < NX; i++) xnext[(size_t)e * NX + i] = xn[i];
    for (int i = 0; i < NX * NV; i++) xjac[(size_t)e * NX * NV + i] = BA[i];
    for (int i = 0; i < NV; i++) for (int j = 0; j < NV; j++) lag[(size_t)e * NV * NV + i * NV + j] = W[i][j];
    mirror7(W, d.reg_eps);
    for (int i = 0; i < NV; i++) for (int j = 0; j < NV; j++) mir[(size_t)e * NV * NV + i * NV + j] = W[i][j];
}

}  // namespace tmpc
