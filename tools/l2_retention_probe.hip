// Does the L2 retain a workgroup's private global scratch between its own writes and reads?  Every workgroup owns 8 KiB of a hipMalloc'ed
// buffer and does REPS rounds of { write it, barrier, read it back (other lanes) }.  If the L2 keeps the lines, FETCH_SIZE / WRITE_SIZE
// (rocprofv3 --pmc, separate passes) stay near the buffer size; if every round goes out to the fabric they scale with REPS.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2probe tools/l2_retention_probe.hip && rocprofv3 --pmc FETCH_SIZE -d out -- /tmp/l2probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(64) void probe(double *buf, int reps, double *sink)
{
    double *w = buf + (size_t)blockIdx.x * 1024;
    double acc = 0.0;
    for (int r = 0; r < reps; r++) {
        for (int i = threadIdx.x; i < 1024; i += 64) w[i] = (double)(r + i);
        __syncthreads();
        for (int i = threadIdx.x; i < 1024; i += 64) acc += w[(i + 517) & 1023];
        __syncthreads();
    }
    if (acc == -1.0) sink[0] = acc;
}
int main(int argc, char **argv)
{
    const int wgs = 2048, reps = argc > 1 ? atoi(argv[1]) : 200;
    double *buf, *sink;
    hipMalloc(&buf, (size_t)wgs * 8192); hipMalloc(&sink, 8);
    hipMemset(buf, 0, (size_t)wgs * 8192);
    hipLaunchKernelGGL(probe, dim3(wgs), dim3(64), 0, 0, buf, reps, sink);
    hipDeviceSynchronize();
    std::printf("wgs %d reps %d: buffer %.1f MB, written+read per side %.1f MB\n", wgs, reps, wgs * 8192 / 1e6, (double)wgs * 8192 * reps / 1e6);
    return 0;
}
