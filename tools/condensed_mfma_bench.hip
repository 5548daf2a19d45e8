// tools/condensed_mfma_bench.hip -- measurement for VERDICT r1 next-4 / north_star "MFMA only for the dense condensed-KKT GEMMs":
// the Newton system of ONE interior-point iteration of the cfg-2 QP (N = 20, nu = 2, nx = 5) solved by FULL CONDENSING on the
// 40 inputs with v_mfma_f64_16x16x4_f64, timed per phase, next to what the production kernels spend on the same system with the
// stage-wise square-root Riccati recursion (tmpc_debug_profile: factor + two vector solves).
//
// One workgroup (4 wavefronts) per trajectory, B = 64 (one control tick), everything in LDS:
//   1. G_k = d v_k / d u  (7 x 40 per node, k = 0..N): G_{k+1,x} = A_k G_{k,x} + B_k E_k          (sequential in k, parallel in columns)
//   2. M = sum_k G_k^T H_k G_k  (40 x 40, H_k = W_k + barrier terms, 7 x 7): f64 MFMA 16x16x4, K = 8 per node (7 padded), the
//      six lower 16 x 16 tiles, k-steps dealt to the four waves, partial tiles reduced through LDS; the operand H_k G_k is formed
//      on the fly (7 FMAs per lane per MFMA)
//   3. Cholesky of M (right-looking, all 256 threads on the trailing update)
//   4. two right-hand sides (predictor, corrector): forward + backward substitution each
//   5. v_k = G_k du (expansion back to the stage variables)
// The result is checked against a host solve of the same dense system.  Build: hipcc --offload-arch=gfx950 -O3 -o build/condensed_mfma_bench
// tools/condensed_mfma_bench.hip ; run on the GPU box: build/condensed_mfma_bench [reps]  -> one JSON line.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int N = 20, NU = 2, NX = 5, NV = 7, NUT = NU * N;      // 40 inputs
constexpr int NT = 256;
constexpr int MP = 48;                                             // 40 padded to 3 tiles of 16
typedef double double4_t __attribute__((ext_vector_type(4)));

enum { PH_G = 0, PH_M, PH_CHOL, PH_SOLVE, PH_EXPAND, PH_COUNT };

struct Problem {          // per trajectory, row-major
    double H[N + 1][NV][NV];
    double A[N][NX][NV];   // [B A]: columns ordered [u; x]
    double rhs[2][NUT];
};

__global__ __launch_bounds__(NT) void condensed_kernel(const Problem *__restrict__ pb, double *__restrict__ du_out, double *__restrict__ v_out,
                                                       long long *__restrict__ cycles, int reps)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *G = smem;                                   // [N+1][8][MP]  (row 7 of every node block and columns >= 40 are zero padding)
    double *part = G + (N + 1) * 8 * MP;                // [4 waves][6 tiles][256]
    double *M = part + 4 * 6 * 256;                     // [MP][MP]
    double *y = M + MP * MP;                            // [2][MP]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const Problem &P = pb[blockIdx.x];
    long long acc[PH_COUNT] = {0, 0, 0, 0, 0};
    for (int rep = 0; rep < reps; rep++) {
        long long t0 = clock64();
        // ---- 1. G ----
        for (int e = tid; e < (N + 1) * 8 * MP; e += NT) G[e] = 0.0;
        __syncthreads();
        for (int k = 0; k < N; k++) {
            if (tid < NU) G[(k * 8 + tid) * MP + NU * k + tid] = 1.0;                       // du_k block: identity
            __syncthreads();
            // x-part of node k+1: rows 2..6 of block k+1 = [B A]_k (5 x 7) * block k (7 x MP); only columns < 2 (k + 1) are non-zero
            for (int e = tid; e < NX * MP; e += NT) {
                const int i = e / MP, c = e - i * MP;
                double a = 0.0;
                if (c < NU * (k + 1)) {
#pragma unroll
                    for (int m = 0; m < NV; m++) a += P.A[k][i][m] * G[(k * 8 + m) * MP + c];
                }
                G[((k + 1) * 8 + NU + i) * MP + c] = a;
            }
            __syncthreads();
        }
        long long t1 = clock64(); acc[PH_G] += t1 - t0; t0 = t1;
        // ---- 2. M = sum_k G_k^T (H_k G_k) with f64 MFMA: tiles (ti, tj), ti >= tj ----
        {
            double4_t c[6];
#pragma unroll
            for (int t = 0; t < 6; t++) c[t] = double4_t{0.0, 0.0, 0.0, 0.0};
            const int l15 = lane & 15, l4 = lane >> 4;
            for (int ks = wave; ks < (N + 1) * 2; ks += 4) {                                // k-step = (node, half): 4 of the 8 padded rows
                const int k = ks >> 1, r = (ks & 1) * 4 + l4;                               // row of the node block this lane feeds
                double a_op[3], b_op[3];
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const int col = t * 16 + l15;
                    a_op[t] = G[(k * 8 + r) * MP + col];                                    // A[row = col of G^T][kk] = G_k[r][col]
                    double tb = 0.0;                                                        // B[kk][col] = (H_k G_k)[r][col]
                    if (r < NV) {
#pragma unroll
                        for (int m = 0; m < NV; m++) tb += P.H[k][r][m] * G[(k * 8 + m) * MP + col];
                    }
                    b_op[t] = tb;
                }
                c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op[0], b_op[0], c[0], 0, 0, 0);
                c[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op[1], b_op[0], c[1], 0, 0, 0);
                c[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op[1], b_op[1], c[2], 0, 0, 0);
                c[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op[2], b_op[0], c[3], 0, 0, 0);
                c[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op[2], b_op[1], c[4], 0, 0, 0);
                c[5] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op[2], b_op[2], c[5], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 6; t++)
#pragma unroll
                for (int q = 0; q < 4; q++) part[((wave * 6 + t) * 4 + q) * 64 + lane] = c[t][q];
            __syncthreads();
            // reduce the four partials; element (reg q, lane) of tile t is row = (lane >> 4) + 4 q, col = lane & 15
            constexpr int ti_[6] = {0, 1, 1, 2, 2, 2}, tj_[6] = {0, 0, 1, 0, 1, 2};
            for (int e = tid; e < 6 * 256; e += NT) {
                const int t = e >> 8, q = (e >> 6) & 3, ln = e & 63;
                const double s = part[((0 * 6 + t) * 4 + q) * 64 + ln] + part[((1 * 6 + t) * 4 + q) * 64 + ln] +
                                 part[((2 * 6 + t) * 4 + q) * 64 + ln] + part[((3 * 6 + t) * 4 + q) * 64 + ln];
                const int row = ti_[t] * 16 + (ln >> 4) + 4 * q, col = tj_[t] * 16 + (ln & 15);
                M[row * MP + col] = s;
                if (ti_[t] != tj_[t]) M[col * MP + row] = s;
            }
            __syncthreads();
        }
        t1 = clock64(); acc[PH_M] += t1 - t0; t0 = t1;
        // ---- 3. Cholesky (lower) of the leading 40 x 40 block, in place ----
        for (int j = 0; j < NUT; j++) {
            const double ljj = sqrt(M[j * MP + j]);
            __syncthreads();
            if (tid == 0) M[j * MP + j] = ljj;
            for (int i = j + 1 + tid; i < NUT; i += NT) M[i * MP + j] /= ljj;
            __syncthreads();
            const int n = NUT - 1 - j;                                                      // trailing block: rows/cols j+1 .. 39, lower part
            for (int e = tid; e < n * n; e += NT) {
                const int i = j + 1 + e / n, cc = j + 1 + e % n;
                if (cc <= i) M[i * MP + cc] -= M[i * MP + j] * M[cc * MP + j];
            }
            __syncthreads();
        }
        t1 = clock64(); acc[PH_CHOL] += t1 - t0; t0 = t1;
        // ---- 4. two right-hand sides: L y = r, L^T x = y (column-oriented, one wave per right-hand side) ----
        if (wave < 2) {
            volatile double *yy = y + wave * MP;               // wave-synchronous: LDS operations of one wave execute in order
            if (lane < NUT) yy[lane] = P.rhs[wave][lane];
            for (int j = 0; j < NUT; j++) {
                const double yj = yy[j] / M[j * MP + j];
                if (lane == j) yy[j] = yj;
                if (lane > j && lane < NUT) yy[lane] -= M[lane * MP + j] * yj;
            }
            for (int j = NUT - 1; j >= 0; j--) {
                const double xj = yy[j] / M[j * MP + j];
                if (lane == j) yy[j] = xj;
                if (lane < j) yy[lane] -= M[j * MP + lane] * xj;
            }
        }
        __syncthreads();
        t1 = clock64(); acc[PH_SOLVE] += t1 - t0; t0 = t1;
        // ---- 5. v_k = G_k du (both solutions) ----
        for (int e = tid; e < 2 * (N + 1) * NV; e += NT) {
            const int s = e / ((N + 1) * NV), rem = e - s * (N + 1) * NV, k = rem / NV, i = rem - k * NV;
            double a = 0.0;
            for (int c2 = 0; c2 < NUT; c2++) a += G[(k * 8 + i) * MP + c2] * y[s * MP + c2];
            v_out[((size_t)blockIdx.x * 2 + s) * (N + 1) * NV + rem] = a;
        }
        __syncthreads();
        t1 = clock64(); acc[PH_EXPAND] += t1 - t0;
    }
    if (tid < NUT) { du_out[(size_t)blockIdx.x * 2 * NUT + tid] = y[tid]; du_out[(size_t)blockIdx.x * 2 * NUT + NUT + tid] = y[MP + tid]; }
    if (tid == 0) for (int i = 0; i < PH_COUNT; i++) cycles[(size_t)blockIdx.x * PH_COUNT + i] = acc[i];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 200, B = 64;
    std::vector<Problem> h(B);
    srand(7);
    auto rnd = [] { return rand() / (double)RAND_MAX - 0.5; };
    for (auto &P : h) {
        for (int k = 0; k <= N; k++) {                                  // SPD stage Hessians: R R^T + I (barrier-augmented blocks look like this)
            double R[NV][NV];
            for (auto &row : R) for (auto &x : row) x = rnd();
            for (int i = 0; i < NV; i++) for (int j = 0; j < NV; j++) { double a = i == j ? 1.0 : 0.0; for (int m = 0; m < NV; m++) a += R[i][m] * R[j][m]; P.H[k][i][j] = a; }
        }
        for (int k = 0; k < N; k++) {                                   // unicycle-like [B A] (tmpc_stage.hpp dyn_jacobian pattern)
            for (auto &row : P.A[k]) for (auto &x : row) x = 0.0;
            const double dt = 0.2;
            P.A[k][0][0] = 0.02 * rnd(); P.A[k][0][1] = 0.05 * rnd(); P.A[k][0][2] = 1; P.A[k][0][4] = 0.3 * rnd(); P.A[k][0][5] = dt;
            P.A[k][1][0] = 0.02 * rnd(); P.A[k][1][1] = 0.05 * rnd(); P.A[k][1][3] = 1; P.A[k][1][4] = 0.3 * rnd(); P.A[k][1][5] = 0.1 * rnd();
            P.A[k][2][1] = dt; P.A[k][2][4] = 1; P.A[k][3][0] = dt; P.A[k][3][5] = 1; P.A[k][4][0] = 0.5 * dt * dt; P.A[k][4][5] = dt; P.A[k][4][6] = 1;
        }
        for (auto &r : P.rhs) for (auto &x : r) x = rnd();
    }
    // ---- host reference of the same dense system ----
    std::vector<double> ref((size_t)B * 2 * NUT);
    for (int b = 0; b < B; b++) {
        const Problem &P = h[b];
        static double G[N + 1][NV][NUT], M[NUT][NUT];
        for (auto &blk : G) for (auto &row : blk) for (auto &x : row) x = 0.0;
        for (int k = 0; k < N; k++) {
            for (int i = 0; i < NU; i++) G[k][i][NU * k + i] = 1.0;
            for (int i = 0; i < NX; i++) for (int c = 0; c < NUT; c++) { double a = 0.0; for (int m = 0; m < NV; m++) a += P.A[k][i][m] * G[k][m][c]; G[k + 1][NU + i][c] = a; }
        }
        for (int i = 0; i < NUT; i++) for (int j = 0; j < NUT; j++) {
            double a = 0.0;
            for (int k = 0; k <= N; k++) for (int r = 0; r < NV; r++) { double t = 0.0; for (int m = 0; m < NV; m++) t += P.H[k][r][m] * G[k][m][j]; a += G[k][r][i] * t; }
            M[i][j] = a;
        }
        for (int j = 0; j < NUT; j++) {
            for (int c = 0; c < j; c++) M[j][j] -= M[j][c] * M[j][c];
            M[j][j] = sqrt(M[j][j]);
            for (int i = j + 1; i < NUT; i++) { for (int c = 0; c < j; c++) M[i][j] -= M[i][c] * M[j][c]; M[i][j] /= M[j][j]; }
        }
        for (int s = 0; s < 2; s++) {
            double yv[NUT];
            for (int i = 0; i < NUT; i++) { double a = P.rhs[s][i]; for (int c = 0; c < i; c++) a -= M[i][c] * yv[c]; yv[i] = a / M[i][i]; }
            for (int i = NUT - 1; i >= 0; i--) { double a = yv[i]; for (int c = i + 1; c < NUT; c++) a -= M[c][i] * yv[c]; yv[i] = a / M[i][i]; }
            for (int i = 0; i < NUT; i++) ref[((size_t)b * 2 + s) * NUT + i] = yv[i];
        }
    }
    Problem *d_pb; double *d_du, *d_v; long long *d_cyc;
    CK(hipMalloc(&d_pb, sizeof(Problem) * B)); CK(hipMalloc(&d_du, sizeof(double) * B * 2 * NUT));
    CK(hipMalloc(&d_v, sizeof(double) * B * 2 * (N + 1) * NV)); CK(hipMalloc(&d_cyc, sizeof(long long) * B * PH_COUNT));
    CK(hipMemcpy(d_pb, h.data(), sizeof(Problem) * B, hipMemcpyHostToDevice));
    const size_t lds = sizeof(double) * ((N + 1) * 8 * MP + 4 * 6 * 256 + MP * MP + 2 * MP);
    CK(hipFuncSetAttribute((const void *)condensed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(condensed_kernel, dim3(B), dim3(NT), lds, 0, d_pb, d_du, d_v, d_cyc, 2);      // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(condensed_kernel, dim3(B), dim3(NT), lds, 0, d_pb, d_du, d_v, d_cyc, reps);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<double> du((size_t)B * 2 * NUT); std::vector<long long> cyc((size_t)B * PH_COUNT);
    CK(hipMemcpy(du.data(), d_du, du.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
    double err = 0.0, scale = 0.0;
    for (size_t i = 0; i < du.size(); i++) { err = fmax(err, fabs(du[i] - ref[i])); scale = fmax(scale, fabs(ref[i])); }
    double mean[PH_COUNT], total = 0.0;
    for (int i = 0; i < PH_COUNT; i++) { double a = 0; for (int b = 0; b < B; b++) a += (double)cyc[(size_t)b * PH_COUNT + i]; mean[i] = a / B / reps; total += mean[i]; }
    const double flops_mfma = 2.0 * 6 * 42 * 16 * 16 * 4;               // per system: 6 tiles x 42 k-steps x 16x16x4 FMAs x 2
    printf("{\"B\": %d, \"reps\": %d, \"lds_bytes\": %zu, \"max_abs_err_vs_host\": %.3e, \"solution_scale\": %.3e, \"us_per_newton_system\": %.3f, "
           "\"cycles_per_newton_system\": {\"build_G\": %.0f, \"mfma_M\": %.0f, \"cholesky_40\": %.0f, \"four_substitutions\": %.0f, \"expand\": %.0f, \"total\": %.0f}, "
           "\"mfma_flops_per_system\": %.0f, \"mfma_instructions_per_system\": %d}\n",
           B, reps, lds, err, scale, ms * 1e3 / reps, mean[0], mean[1], mean[2], mean[3], mean[4], total, flops_mfma, 6 * 42);
    return err <= 1e-9 * fmax(scale, 1.0) ? 0 : 3;
}
