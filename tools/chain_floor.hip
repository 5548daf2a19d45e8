// tools/chain_floor.hip -- gfx950 microbenchmark behind profiles/round5_chain_floor.json (round-4 verdict, next-1a): what does ONE dependent
// FP64 operation cost a wave, how fast can one wave / two waves per SIMD issue INDEPENDENT FP64 operations, and what do the cross-lane and LDS
// primitives of the Riccati sweeps (csrc/tmpc_riccati.hpp) add to a dependent chain?  From these the critical-path length of one stage step of
// riccati_factor_rows / riccati_sweeps_rows follows (tools/chain_floor_report.py), to be set against the cycles measured for the real kernel
// (tmpc_debug_profile, profiles/round*_phases.jsonl).
//
// Every test is a loop of ITER x 16 unrolled repetitions of a short pattern over K independent chains; cycles = s_memtime deltas of each
// wave (shader clock), also s_memrealtime (100 MHz) to calibrate the clock.  grid = 1 workgroup of 64 x 4 x W threads: W waves on each of the
// four SIMDs of one CU (one CU is enough: SIMDs do not share issue).  Build: hipcc --offload-arch=gfx950 -O3 -o chain_floor chain_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int LANE> __device__ __forceinline__ double bcast16(double x)
{
    const long long r = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(long long, x), 0x150 + LANE, 0xf, 0xf, false);
    return __builtin_bit_cast(double, r);
}
__device__ __forceinline__ double readlane_d(double x, int src)
{
    union { double d; int i[2]; } u; u.d = x;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], src); u.i[1] = __builtin_amdgcn_readlane(u.i[1], src);
    return u.d;
}
__device__ __forceinline__ double rsqrt_nr(double d)       // = csrc/tmpc_solve.hip::rsqrt_nr
{
    const double y = __builtin_amdgcn_rsq(d);
    const double e = fma(-d * y, y, 1.0);
    return fma(y, e * fma(0.375, e, 0.5), y);
}
__device__ __forceinline__ double rcp_nr(double d)
{
    const double y = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, y, 1.0);
    return fma(y, fma(e, e, e), y);
}

typedef __attribute__((address_space(3))) double lds_double;
typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(unsigned long long)(const lds_double *)p; }
__device__ __forceinline__ void ds_write64(unsigned addr, double v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void ds_add64(unsigned addr, double v) { asm volatile("ds_add_f64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ double ds_read64(unsigned addr) { double r; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory"); return r; }
__device__ __forceinline__ int ds_read32(unsigned addr) { int r; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory"); return r; }

enum Test { T_FMA = 0, T_MUL, T_ADD, T_FMA32, T_DPP_FMA, T_DPP_ONLY, T_READLANE_FMA, T_RSQ_ONLY, T_RSQRT_NR, T_RCP_NR, T_CHOL_COLUMN, T_LDS_RT, T_LDS_ADD_RT,
            T_LDS_READ_CHAIN, T_FMA_DPPMIX, T_COUNT };
static const char *kName[T_COUNT] = {"fma_f64", "mul_f64", "add_f64", "fma_f32", "dpp_bcast+fma_f64", "dpp_bcast_only", "readlane_pair+fma_f64", "rsq_f64_only",
                                     "rsqrt_nr (rsq + 5 ops)", "rcp_nr (rcp + 4 ops)", "chol_column (bcast, rsqrt_nr, mul, bcast, fma)", "lds write->read round trip + fma",
                                     "lds ds_add_f64 -> read + fma", "lds read, address from the value read", "fma_f64 with an INDEPENDENT dpp_bcast beside each"};
// dependent operations per repetition of the pattern (per chain), for "cycles per dependent operation"
static const int kOps[T_COUNT] = {1, 1, 1, 1, 2, 1, 2, 1, 6, 5, 10, 2, 2, 1, 1};

template <int TEST, int K>
__global__ __launch_bounds__(1024) void bench(double *out, long long *cyc, long long *rt, int iters, double a, double b, double seed)
{
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    double x[K];
#pragma unroll
    for (int c = 0; c < K; c++) x[c] = seed + 1e-3 * c + 1e-6 * (tid & 15);
    float xf[K];
#pragma unroll
    for (int c = 0; c < K; c++) xf[c] = (float)x[c];
    double *slot = lds + tid * K;
    int idx = tid * 2;
    const unsigned sa = lds_addr(slot), la = lds_addr(lds);
#pragma unroll
    for (int c = 0; c < K; c++) slot[c] = x[c];
    if (TEST == T_LDS_READ_CHAIN) { reinterpret_cast<int *>(lds)[tid * 2] = tid * 2; }
    __syncthreads();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int c = 0; c < K; c++) {
                if constexpr (TEST == T_FMA) x[c] = fma(x[c], a, b);
                else if constexpr (TEST == T_MUL) x[c] = x[c] * a;
                else if constexpr (TEST == T_ADD) x[c] = x[c] + b;
                else if constexpr (TEST == T_FMA32) xf[c] = fmaf(xf[c], (float)a, (float)b);
                else if constexpr (TEST == T_DPP_FMA) x[c] = fma(bcast16<3>(x[c]), a, b);
                else if constexpr (TEST == T_DPP_ONLY) x[c] = bcast16<3>(x[c]);
                else if constexpr (TEST == T_READLANE_FMA) x[c] = fma(readlane_d(x[c], 3), a, b);
                else if constexpr (TEST == T_RSQ_ONLY) x[c] = __builtin_amdgcn_rsq(x[c]);
                else if constexpr (TEST == T_RSQRT_NR) x[c] = rsqrt_nr(x[c]);
                else if constexpr (TEST == T_RCP_NR) x[c] = rcp_nr(x[c]);
                else if constexpr (TEST == T_CHOL_COLUMN) {
                    // one column of chol_rows (csrc/tmpc_riccati.hpp): pivot broadcast, rsqrt_nr, scale, column broadcast, rank-1 update of the next pivot
                    const double piv = bcast16<0>(x[c]);
                    const double y = rsqrt_nr(piv);
                    const double l = x[c] * y;
                    const double lj = bcast16<1>(l);
                    x[c] = fma(-l, lj, b);
                }
                else if constexpr (TEST == T_LDS_RT) { ds_write64(sa + 8 * c, x[c]); x[c] = fma(ds_read64(sa + 8 * c), a, b); }
                else if constexpr (TEST == T_LDS_ADD_RT) { ds_add64(sa + 8 * c, x[c]); x[c] = fma(ds_read64(sa + 8 * c), a, 0.25 * b); }
                else if constexpr (TEST == T_LDS_READ_CHAIN) { idx = ds_read32(la + 4 * idx); }
                else if constexpr (TEST == T_FMA_DPPMIX) { x[c] = fma(x[c], a, b); slot[c] = bcast16<2>(a + u); }
            }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < K; c++) s += x[c] + (double)xf[c];
    s += (double)idx;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::"v"(s));
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + tid] = s;
    if ((tid & 63) == 0) { cyc[blockIdx.x * (blockDim.x / 64) + tid / 64] = t1 - t0; rt[blockIdx.x * (blockDim.x / 64) + tid / 64] = r1 - r0; }
}

struct Result { std::string name; int K, W, ops; double cyc_per_rep, cyc_per_dep_op, ns_per_rep, mhz; };

template <int TEST, int K>
static Result run(int W, int iters, double *d_out, long long *d_cyc, long long *d_rt)
{
    const int threads = 64 * 4 * W;
    std::vector<long long> cyc(4 * W), rt(4 * W);
    // a = 0.5, b = 0.75: x -> 0.5 x + 0.75 converges to 1.5 (finite, positive: rsq / rcp stay well defined)
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL((bench<TEST, K>), dim3(1), dim3(threads), threads * K * 8 + 64, 0, d_out, d_cyc, d_rt, iters, 0.5, 0.75, 1.25);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(rt.data(), d_rt, rt.size() * 8, hipMemcpyDeviceToHost));
    long long cmax = 0, rmax = 0;
    for (size_t i = 0; i < cyc.size(); i++) { if (cyc[i] > cmax) cmax = cyc[i]; if (rt[i] > rmax) rmax = rt[i]; }
    Result r;
    r.name = kName[TEST]; r.K = K; r.W = W; r.ops = kOps[TEST];
    const double reps = (double)iters * 16;
    r.cyc_per_rep = cmax / reps;                          // cycles for one repetition of the pattern over ALL K chains, slowest wave
    r.cyc_per_dep_op = r.cyc_per_rep / kOps[TEST];        // with K = 1: latency of one dependent operation of the pattern (mean over its operations)
    r.ns_per_rep = rmax * 10.0 / reps;                    // s_memrealtime ticks at 100 MHz
    r.mhz = r.ns_per_rep > 0 ? r.cyc_per_rep / r.ns_per_rep * 1e3 : 0.0;
    return r;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4096;
    double *d_out; long long *d_cyc, *d_rt;
    CHECK(hipMalloc(&d_out, 1024 * 8)); CHECK(hipMalloc(&d_cyc, 64 * 8)); CHECK(hipMalloc(&d_rt, 64 * 8));
    std::vector<Result> res;
    for (int W = 1; W <= 2; W++) {
        res.push_back(run<T_FMA, 1>(W, iters, d_out, d_cyc, d_rt));
        res.push_back(run<T_FMA, 2>(W, iters, d_out, d_cyc, d_rt));
        res.push_back(run<T_FMA, 4>(W, iters, d_out, d_cyc, d_rt));
        res.push_back(run<T_FMA, 8>(W, iters, d_out, d_cyc, d_rt));
        res.push_back(run<T_MUL, 1>(W, iters, d_out, d_cyc, d_rt));
        res.push_back(run<T_ADD, 1>(W, iters, d_out, d_cyc, d_rt));
        res.push_back(run<T_FMA32, 1>(W, iters, d_out, d_cyc, d_rt));
        res.push_back(run<T_FMA32, 4>(W, iters, d_out, d_cyc, d_rt));
        res.push_back(run<T_DPP_FMA, 1>(W, iters, d_out, d_cyc, d_rt));
        res.push_back(run<T_DPP_FMA, 2>(W, iters, d_out, d_cyc, d_rt));
        res.push_back(run<T_DPP_FMA, 4>(W, iters, d_out, d_cyc, d_rt));
        res.push_back(run<T_DPP_ONLY, 1>(W, iters, d_out, d_cyc, d_rt));
        res.push_back(run<T_DPP_ONLY, 4>(W, iters, d_out, d_cyc, d_rt));
        res.push_back(run<T_READLANE_FMA, 1>(W, iters, d_out, d_cyc, d_rt));
        res.push_back(run<T_RSQ_ONLY, 1>(W, iters, d_out, d_cyc, d_rt));
        res.push_back(run<T_RSQ_ONLY, 4>(W, iters, d_out, d_cyc, d_rt));
        res.push_back(run<T_RSQRT_NR, 1>(W, iters, d_out, d_cyc, d_rt));
        res.push_back(run<T_RSQRT_NR, 2>(W, iters, d_out, d_cyc, d_rt));
        res.push_back(run<T_RCP_NR, 1>(W, iters, d_out, d_cyc, d_rt));
        res.push_back(run<T_CHOL_COLUMN, 1>(W, iters, d_out, d_cyc, d_rt));
        res.push_back(run<T_CHOL_COLUMN, 2>(W, iters, d_out, d_cyc, d_rt));
        res.push_back(run<T_CHOL_COLUMN, 4>(W, iters, d_out, d_cyc, d_rt));
        res.push_back(run<T_LDS_RT, 1>(W, iters / 4, d_out, d_cyc, d_rt));
        res.push_back(run<T_LDS_ADD_RT, 1>(W, iters / 4, d_out, d_cyc, d_rt));
        res.push_back(run<T_LDS_READ_CHAIN, 1>(W, iters / 4, d_out, d_cyc, d_rt));
        res.push_back(run<T_FMA_DPPMIX, 1>(W, iters, d_out, d_cyc, d_rt));
    }
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"clock_rate_khz\": %d, \"iters\": %d, \"what\": \"cycles = s_memtime delta of the slowest wave; one workgroup on one CU, W waves per SIMD; "
           "K independent chains per wave; cyc_per_rep = one repetition of the pattern over all K chains; cyc_per_dep_op (K = 1) = latency of one dependent operation\", \"tests\": [\n",
           p.name, p.gcnArchName, p.clockRate, iters);
    for (size_t i = 0; i < res.size(); i++) {
        const Result &r = res[i];
        printf("  {\"test\": \"%s\", \"chains_per_wave\": %d, \"waves_per_simd\": %d, \"dependent_ops_per_rep\": %d, \"cycles_per_rep\": %.3f, \"cycles_per_dependent_op\": %.3f, "
               "\"cycles_per_rep_per_chain\": %.3f, \"ns_per_rep\": %.4f, \"memtime_mhz\": %.1f}%s\n",
               r.name.c_str(), r.K, r.W, r.ops, r.cyc_per_rep, r.cyc_per_dep_op, r.cyc_per_rep / r.K, r.ns_per_rep, r.mhz, i + 1 < res.size() ? "," : "");
    }
    printf("]}\n");
    return 0;
}
