// tools/pmc_calibrate.hip -- known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern").
// Access pattern = the solve kernel's: 8-byte loads/stores per lane, consecutive lanes consecutive doubles.
//   calib_read : reads  n doubles once (grid-stride), writes one double per workgroup
//   calib_write: writes n doubles once
// Buffers are 1 GiB (past the 256 MiB Infinity Cache), each kernel launched 3 times.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void calib_read(const double *x, size_t n, double *out)
{
    double acc = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += x[i];
    if (acc == 123.456) out[blockIdx.x] = acc;      // never true: keeps the loads alive without a write stream
}
__global__ void calib_write(double *x, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = 1.0;
}

int main()
{
    const size_t n = (size_t)1 << 27;                // 2^27 doubles = 1 GiB
    double *x, *out;
    if (hipMalloc(&x, n * 8) != hipSuccess || hipMalloc(&out, 65536 * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(x, 0, n * 8);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(calib_write, dim3(4096), dim3(256), 0, 0, x, n);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(calib_read, dim3(4096), dim3(256), 0, 0, x, n, out);
    hipDeviceSynchronize();
    printf("calib bytes per launch %zu\n", n * 8);
    return 0;
}
