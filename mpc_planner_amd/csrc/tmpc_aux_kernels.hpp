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

// ---- f-1: LinearizedConstraints::update + setParameters on device (linearized_constraints.cpp:49-189) ----------
// one thread per (trajectory, stage)
__global__ void tmpc_linearize_topology_kernel(Dims d, int B, const double *x0, double *params, const double *obst,
                                               const int *scene_of, const double *state_x, double robot_radius,
                                               const uint8_t *is_original)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = d.N;
    if (e >= B * N) return;
    const int b = e / N, k = e - b * N;
    const int sc = scene_of[b];
    double *p = params + ((size_t)b * N + k) * d.npar;
    const double dummy_b = state_x[sc] + 100.0;                         // _dummy_b (:54)
    const bool dummy = (k == 0) || (is_original && is_original[b]);
    const double r = 1e-3 + robot_radius;                               // guidance mode radius (:99)
    double px = x0[((size_t)b * (N + 1) + k) * ext_nv(d) + ZX], py = x0[((size_t)b * (N + 1) + k) * ext_nv(d) + ZY];
    const double *ob = obst + (size_t)sc * d.n_lin * N * 2;
    if (!dummy) {
        // projectToSafety (:130-148): at most 3 sweeps over the obstacles of ros_tools' Douglas-Rachford step with obstacle 0 as the
        // anchor.  ros_tools is not in the reference tree; restated from the published operator p <- (p + R_delta R_anchor p) / 2,
        // R = 2 P - I, P = nearest point outside the disc of radius r, applied when p is inside the obstacle's disc (same arithmetic
        // as modules.py::project_to_safety and the C++ DouglasRachford).  No FMA contraction: bit-equal to the host mirrors.
        const double ax0 = ob[(size_t)(k - 1) * 2], ay0 = ob[(size_t)(k - 1) * 2 + 1];          // anchor = obstacle 0 at step k-1
        auto outside = [&](double qx, double qy, double cx, double cy, double &ox_, double &oy_) {
            const double dx = qx - cx, dy = qy - cy;
            const double dist = sqrt(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
            if (dist >= r) { ox_ = qx; oy_ = qy; }
            else if (dist > 1e-12) { const double s = r / dist; ox_ = __dadd_rn(cx, __dmul_rn(dx, s)); oy_ = __dadd_rn(cy, __dmul_rn(dy, s)); }
            else { ox_ = cx; oy_ = cy + r; }
        };
        for (int sweep = 0; sweep < 3; sweep++)
            for (int j = 0; j < d.n_lin; j++) {
                const double ox = ob[((size_t)j * N + (k - 1)) * 2], oy = ob[((size_t)j * N + (k - 1)) * 2 + 1];
                const double dx = px - ox, dy = py - oy;
                if (sqrt(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy))) < r) {
                    double qx, qy, bx, by;
                    outside(px, py, ax0, ay0, qx, qy);
                    const double rx = __dmul_rn(2.0, qx) - px, ry = __dmul_rn(2.0, qy) - py;
                    outside(rx, ry, ox, oy, bx, by);
                    const double sx = __dmul_rn(2.0, bx) - rx, sy = __dmul_rn(2.0, by) - ry;
                    px = (px + sx) / 2.0; py = (py + sy) / 2.0;
                }
            }
    }
    for (int j = 0; j < d.n_lin; j++) {
        double a1 = 1.0, a2 = 0.0, bb = dummy_b;                        // _dummy_a1, _dummy_a2
        if (!dummy) {
            const double ox = ob[((size_t)j * N + (k - 1)) * 2], oy = ob[((size_t)j * N + (k - 1)) * 2 + 1];
            const double dx = ox - px, dy = oy - py;
            const double dist = sqrt(dx * dx + dy * dy);
            a1 = dx / dist; a2 = dy / dist;
            bb = a1 * ox + a2 * oy - r;
        }
        p[ip_lin(d, j, 0)] = a1; p[ip_lin(d, j, 1)] = a2; p[ip_lin(d, j, 2)] = bb;
    }
}


// ---- f-3: scenario -> halfspace reduction on device (SH-MPC, BASELINE config 5) -----------------------------------
// The reference delegates this to the external scenario_module (scenario_constraints.cpp:47,76-79; source absent), so
// this restates the host mirror mpc_planner_amd/modules.py::scenario_halfspaces: for stage k >= 1 of trajectory b every
// sampled obstacle position o (n_pts = obstacles x scenarios of the trajectory's scene, prediction step k-1) gives the
// halfspace a = (o - p)/|o - p|, b = a.o - radius around the guess p = x0[b][k](x, y); the n_rows angular sectors of a
// each keep their closest sample (lowest sample index on ties); empty sectors and stage 0 get the dummy row
// (1, 0, x + 100) (decomp_constraints.cpp:153-169 pattern).  One workgroup per (trajectory, stage); the samples of a
// stage are contiguous ([scene][N][n_pts][2]) so the scan is a coalesced HBM/L2 stream shared by the scene's
// trajectories.  Arithmetic is written without FMA contraction so the rows equal the host mirror's bit for bit.
__global__ __launch_bounds__(256) void tmpc_scenario_halfspaces_kernel(Dims d, int B, const double *x0, double *params,
                                                                       const double *samples, int n_pts, int n_rows,
                                                                       const int *scene_of, const double *state_x,
                                                                       double radius, double disc_offset)
{
#pragma clang fp contract(off)
    __shared__ unsigned long long s_best[64];
    __shared__ int s_idx[64];
    const int N = d.N;
    const int b = blockIdx.x / N, k = blockIdx.x - b * N;
    if (b >= B) return;
    const int sc = scene_of[b];
    double *p = params + ((size_t)b * N + k) * d.npar;
    const int tid = threadIdx.x;
    if (tid == 0) p[ip_disc_offset(d)] = disc_offset;
    if (k == 0) {
        if (tid < n_rows) { p[ip_slk(d, tid, 0)] = 1.0; p[ip_slk(d, tid, 1)] = 0.0; p[ip_slk(d, tid, 2)] = state_x[sc] + 100.0; }
        return;
    }
    if (tid < n_rows) { s_best[tid] = ~0ull; s_idx[tid] = 0x7fffffff; }
    __syncthreads();
    const double px = x0[((size_t)b * (N + 1) + k) * ext_nv(d) + ZX], py = x0[((size_t)b * (N + 1) + k) * ext_nv(d) + ZY];
    const double2 *o = reinterpret_cast<const double2 *>(samples) + ((size_t)sc * N + (k - 1)) * n_pts;
    const double scale = (double)n_rows / (2.0 * M_PI);
    auto classify = [&](int i, double &dist, double &ax, double &ay) {
        const double2 q = o[i];
        const double dx = q.x - px, dy = q.y - py;
        dist = sqrt(dx * dx + dy * dy);
        ax = dx / dist; ay = dy / dist;
        int sec = (int)((atan2(ay, ax) + M_PI) * scale);
        return sec < n_rows - 1 ? sec : n_rows - 1;
    };
    // pass 1: per-sector minimum distance; the first CACHE classifications of a thread stay in registers for pass 2
    constexpr int CACHE = 8;
    double c_dist[CACHE]; int c_sec[CACHE];
#pragma unroll
    for (int c = 0; c < CACHE; c++) {
        const int i = tid + c * 256;
        c_sec[c] = -1; c_dist[c] = 0.0;
        if (i < n_pts) {
            double ax, ay;
            c_sec[c] = classify(i, c_dist[c], ax, ay);
            atomicMin(&s_best[c_sec[c]], (unsigned long long)__double_as_longlong(c_dist[c]));   // dist > 0: bit pattern is monotone
        }
    }
    for (int i = tid + CACHE * 256; i < n_pts; i += 256) {
        double dist, ax, ay;
        const int sec = classify(i, dist, ax, ay);
        atomicMin(&s_best[sec], (unsigned long long)__double_as_longlong(dist));
    }
    __syncthreads();
    // pass 2: lowest sample index among the samples at the minimum
#pragma unroll
    for (int c = 0; c < CACHE; c++)
        if (c_sec[c] >= 0 && (unsigned long long)__double_as_longlong(c_dist[c]) == s_best[c_sec[c]]) atomicMin(&s_idx[c_sec[c]], tid + c * 256);
    for (int i = tid + CACHE * 256; i < n_pts; i += 256) {
        double dist, ax, ay;
        const int sec = classify(i, dist, ax, ay);
        if ((unsigned long long)__double_as_longlong(dist) == s_best[sec]) atomicMin(&s_idx[sec], i);
    }
    __syncthreads();
    if (tid < n_rows) {
        double a1 = 1.0, a2 = 0.0, bb = state_x[sc] + 100.0;
        const int i = s_idx[tid];
        if (i != 0x7fffffff) {
            double dist;
            classify(i, dist, a1, a2);
            const double2 q = o[i];
            bb = a1 * q.x + a2 * q.y - radius;
        }
        p[ip_slk(d, tid, 0)] = a1; p[ip_slk(d, tid, 1)] = a2; p[ip_slk(d, tid, 2)] = bb;
    }
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
            x += v * d.dt * cs; y += v * d.dt * sn; spline += v * d.dt;
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
                                           double *__restrict__ lamh)
{
    const int b = blockIdx.x;
    if (exit_code[b] == 1) return;
    for (int e = threadIdx.x; e < n_pi; e += blockDim.x) pi[(size_t)b * n_pi + e] = 0.0;
    for (int e = threadIdx.x; e < n_lam; e += blockDim.x) lamh[(size_t)b * n_lam + e] = 0.0;
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
    CostOut co;
    cost_eval(d, zz, pe, 1, co, true, slack);
    cost[e] = co.val;
    double Wc[NV][NV];
    for (int i = 0; i < NV; i++) for (int j = 0; j < NV; j++) Wc[i][j] = 0.0;
    cost_add_hessian(co, 1.0, Wc);
    for (int i = 0; i < NV; i++) {
        cgrad[(size_t)e * NV + i] = co.g[i];
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
    stage_linearise(d, zz, pe, 1, pi ? pi[(size_t)e * NX] : 0.0, pi ? pi[(size_t)e * NX + 1] : 0.0, lam, sink, W, g, BA, xn, slack);
    for (int i = 0; i < NX; i++) xnext[(size_t)e * NX + i] = xn[i];
    for (int i = 0; i < NX * NV; i++) xjac[(size_t)e * NX * NV + i] = BA[i];
    for (int i = 0; i < NV; i++) for (int j = 0; j < NV; j++) lag[(size_t)e * NV * NV + i * NV + j] = W[i][j];
    mirror7(W, d.reg_eps);
    for (int i = 0; i < NV; i++) for (int j = 0; j < NV; j++) mir[(size_t)e * NV * NV + i * NV + j] = W[i][j];
}

}  // namespace tmpc
