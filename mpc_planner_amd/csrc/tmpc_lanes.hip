// mpc_planner_amd/csrc/tmpc_lanes.hip -- gfx950 kernels of the lane-per-trajectory throughput variant (see tmpc_lanes.hpp)
// and their host-side context.  Second translation unit of libtmpc_hip.so.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "tmpc_lanes.hpp"
#include "tmpc_lanes_api.hpp"

namespace tmpc {
namespace lanes {

// ---- inputs: reference layout [B][n] (one trajectory = n contiguous doubles) -> workspace [block][field][lane] ------------
// 64 x 64 tiles through LDS: reads are contiguous along the trajectory's own data, writes contiguous along the lanes.
enum { IN_PARAMS = 0, IN_X0 = 1, IN_XINIT = 2 };
template <int KIND>
__global__ __launch_bounds__(256) void lanes_transpose_in_kernel(Layout L, int B, int n, int nve, const double *__restrict__ src,
                                                                 double *__restrict__ ws)
{
    __shared__ double tile[LW][LW + 1];
    const int blk = blockIdx.y, e0 = blockIdx.x * LW;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < LW; r += 4) {
        const int b = blk * LW + r, e = e0 + tx;
        tile[r][tx] = (b < B && e < n) ? src[(size_t)b * n + e] : 0.0;
    }
    __syncthreads();
    double *wb = ws + (size_t)blk * block_doubles(L);
    for (int c = ty; c < LW; c += 4) {
        const int e = e0 + c;
        if (e >= n) continue;
        int f;
        if (KIND == IN_PARAMS) f = L.o_par + e;
        else if (KIND == IN_XINIT) f = L.o_xinit + e;
        else {
            const int k = e / nve, i = e - k * nve;
            if (i >= NV) continue;                                  // the slack column of the warm start is ignored (DESIGN U9)
            f = k * L.sd + L.o_z + i;
        }
        wb[(size_t)f * LW + tx] = tile[tx][c];
    }
}

// start of a call: fresh multipliers (pi of every node, lam of every general row) and / or a newly loaded iterate
__global__ __launch_bounds__(LW) void lanes_prepare_kernel(Layout L, int load_iterate, int fresh_multipliers, double *__restrict__ ws)
{
    double *w = ws + (size_t)blockIdx.x * block_doubles(L) + threadIdx.x;
    if (load_iterate < 0) { w[((size_t)L.o_xinit + 7) * LW] = 0.0; return; }      // clear_stopped: a new solve() of the slots' Solvers
    if (fresh_multipliers) {
        for (int k = 0; k <= L.N; k++) {
            for (int i = 0; i < NX; i++) w[((size_t)k * L.sd + L.o_pi + i) * LW] = 0.0;
            if (k < L.N)
                for (int r = 0; r < L.nh; r++) w[((size_t)k * L.sd + L.o_rows + 6 * r + 5) * LW] = 0.0;
        }
    }
    if (load_iterate) {
        w[((size_t)L.N * L.sd + L.o_z + 0) * LW] = 0.0;            // inputs of the terminal node are not variables
        w[((size_t)L.N * L.sd + L.o_z + 1) * LW] = 0.0;
        w[((size_t)L.o_xinit + 7) * LW] = 0.0;                      // the slot's RTI loop is open again
    }
}

#if defined(TMPC_LANES_PROF)
__device__ long long g_lanes_prof[65536 * PF_COUNT];      // profiling build only: per-trajectory phase cycles (first 65536 trajectories)
#endif

__global__ __launch_bounds__(LW) void lanes_solve_kernel(Dims d, Layout L, int B, int n_iter, int persistent, int complete, double *__restrict__ ws,
                                                         double *__restrict__ xtraj, double *__restrict__ utraj,
                                                         double *__restrict__ pobj, int *__restrict__ exit_code,
                                                         int *__restrict__ qp_status, int *__restrict__ sqp_iter,
                                                         double *__restrict__ res_eq, int *__restrict__ qp_iter)
{
    const int lane = threadIdx.x;
    const int b = blockIdx.x * LW + lane;
    if (b >= B) return;
    #if defined(TMPC_LANES_PROF)
    long long *prof = g_lanes_prof + (size_t)b * PF_COUNT;
    for (int i = 0; i < PF_COUNT; i++) prof[i] = 0;
#else
    long long *prof = nullptr;
#endif
    const Lane ln{d, L, ws + (size_t)blockIdx.x * block_doubles(L), (unsigned)lane, prof};
    if (persistent && ln.stopped()) return;                        // this solver's loop has ended: outputs of its last call stand
    const Result R = ln.solve(n_iter);
    if (persistent) ln.close_call(R, complete != 0);
    const int N = d.N, nxe = ext_nx(d);
    const double sl = ln.slack();
    for (int k = 0; k <= N; k++) {
        for (int i = 0; i < NX; i++) xtraj[((size_t)b * (N + 1) + k) * nxe + i] = ln.F(k, L.o_z + NU + i);
        if (nxe > NX) xtraj[((size_t)b * (N + 1) + k) * nxe + NX] = sl;          // the pinned slack state
    }
    for (int k = 0; k < N; k++)
        for (int i = 0; i < NU; i++) utraj[((size_t)b * N + k) * NU + i] = ln.F(k, L.o_z + i);
    pobj[b] = R.pobj; res_eq[b] = R.res_eq; exit_code[b] = R.exit_code;
    qp_status[b] = R.qp_status; sqp_iter[b] = R.sqp_iter; qp_iter[b] = R.qp_iter;
}

struct Context {
    Dims d{};
    Layout L;
    int B_max = 0, nblocks = 0;
    double *ws = nullptr;
    size_t bytes = 0;
};

Context *create(const Dims &d, int B_max, std::string &err)
{
    // Same rule as for the wave kernels (tmpc_solve.hip, pick_fast_kernel): a solve kernel that spills registers to scratch is not
    // dispatched.  The hand-written library is built with zero scratch (__graft_entry__.build() refuses anything else); a generated
    // solver's long emitted stage functions can push the lane kernel into scratch, and such a build was observed to return
    // non-finite linearisations on the device (its host twin is fine): refuse instead of returning wrong iterates.
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, (const void *)lanes_solve_kernel) != hipSuccess) { err = "lanes::create: hipFuncGetAttributes failed"; return nullptr; }
    if (attr.localSizeBytes > 0) {
        err = "throughput mode is not available in this library: its lane-per-trajectory kernel uses " + std::to_string(attr.localSizeBytes) +
              " B/lane of scratch (register spills of the generated stage functions); use the default kernels";
        return nullptr;
    }
    Context *c = new Context();
    c->d = d; c->L = make_layout(d); c->B_max = B_max; c->nblocks = (B_max + LW - 1) / LW;
    c->bytes = (size_t)c->nblocks * block_doubles(c->L) * sizeof(double);
    hipError_t e = hipMalloc(&c->ws, c->bytes);
    if (e != hipSuccess) {
        err = std::string("lanes workspace hipMalloc(") + std::to_string(c->bytes) + " B): " + hipGetErrorString(e);
        delete c;
        return nullptr;
    }
    return c;
}

void destroy(Context *c)
{
    if (!c) return;
    if (c->ws) (void)hipFree(c->ws);
    delete c;
}

size_t workspace_bytes(const Context *c) { return c ? c->bytes : 0; }

#define LANES_CHECK(expr)                                                            \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return -2; } \
    } while (0)

int stage_in(Context *c, hipStream_t stream, int B, const double *xinit, const double *x0, const double *params, bool load_iterate,
             bool fresh_multipliers, std::string &err)
{
    if (!c || B <= 0 || B > c->B_max) { err = "lanes::stage_in: bad batch size"; return -1; }
    const Layout &L = c->L;
    const int nb = (B + LW - 1) / LW, N = c->d.N, nve = ext_nv(c->d), nxe = ext_nx(c->d);
    const int n_par = N * c->d.npar, n_x0 = (N + 1) * nve;
    hipLaunchKernelGGL(lanes_transpose_in_kernel<IN_PARAMS>, dim3((n_par + LW - 1) / LW, nb), dim3(256), 0, stream, L, B, n_par, nve, params, c->ws);
    if (load_iterate)
        hipLaunchKernelGGL(lanes_transpose_in_kernel<IN_X0>, dim3((n_x0 + LW - 1) / LW, nb), dim3(256), 0, stream, L, B, n_x0, nve, x0, c->ws);
    hipLaunchKernelGGL(lanes_transpose_in_kernel<IN_XINIT>, dim3(1, nb), dim3(256), 0, stream, L, B, nxe, nve, xinit, c->ws);
    hipLaunchKernelGGL(lanes_prepare_kernel, dim3(nb), dim3(LW), 0, stream, L, (int)load_iterate, (int)fresh_multipliers, c->ws);
    LANES_CHECK(hipGetLastError());
    return 0;
}

int solve(Context *c, hipStream_t stream, int B, int n_iter, bool persistent, bool complete, double *xtraj, double *utraj, double *pobj, int *exit_code,
          int *qp_status, int *sqp_iter, double *res_eq, int *qp_iter, std::string &err)
{
    if (!c || B <= 0 || B > c->B_max || n_iter < 0) { err = "lanes::solve: bad argument"; return -1; }
    const int nb = (B + LW - 1) / LW;
    hipLaunchKernelGGL(lanes_solve_kernel, dim3(nb), dim3(LW), 0, stream, c->d, c->L, B, n_iter, (int)persistent, (int)complete, c->ws, xtraj, utraj, pobj,
                       exit_code, qp_status, sqp_iter, res_eq, qp_iter);
    LANES_CHECK(hipGetLastError());
    return 0;
}

int reset_multipliers(Context *c, hipStream_t stream, int B, std::string &err)
{
    if (!c || B <= 0 || B > c->B_max) { err = "lanes::reset_multipliers: bad batch size"; return -1; }
    hipLaunchKernelGGL(lanes_prepare_kernel, dim3((B + LW - 1) / LW), dim3(LW), 0, stream, c->L, 0, 1, c->ws);
    LANES_CHECK(hipGetLastError());
    return 0;
}

int clear_stopped(Context *c, hipStream_t stream, int B, std::string &err)
{
    if (!c || B <= 0 || B > c->B_max) { err = "lanes::clear_stopped: bad batch size"; return -1; }
    hipLaunchKernelGGL(lanes_prepare_kernel, dim3((B + LW - 1) / LW), dim3(LW), 0, stream, c->L, -1, 0, c->ws);
    LANES_CHECK(hipGetLastError());
    return 0;
}

}  // namespace lanes
}  // namespace tmpc

#if defined(TMPC_LANES_PROF)
// profiling build only (not part of the C-ABI): mean cycles per phase over the first n trajectories of the last launch
extern "C" int tmpc_lanes_debug_profile(int n, double *mean_cycles)
{
    std::vector<long long> host((size_t)n * tmpc::lanes::PF_COUNT);
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (hipMemcpyFromSymbol(host.data(), HIP_SYMBOL(tmpc::lanes::g_lanes_prof), host.size() * sizeof(long long)) != hipSuccess) return -2;
    for (int i = 0; i < tmpc::lanes::PF_COUNT; i++) {
        double acc = 0.0;
        for (int b = 0; b < n; b++) acc += (double)host[(size_t)b * tmpc::lanes::PF_COUNT + i];
        mean_cycles[i] = acc / n;
    }
    return 0;
}
#endif
