// Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950 for 8-byte-per-lane streaming accesses
// (the access width of the Taylor steppers): kernels with a known byte count, to be run under
//   rocprofv3 --pmc FETCH_SIZE -- ./stream8.bin     and     rocprofv3 --pmc WRITE_SIZE -- ./stream8.bin
// (MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of the bytes of a 16 B/lane stream; other widths uncalibrated).
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void read8(const double *__restrict__ x, double *out, size_t n)
{
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += x[i];
    if (acc == 12345.678) out[0] = acc;
}
__global__ void write8(double *__restrict__ y, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = (double)i;
}
__global__ void copy8(const double *__restrict__ x, double *__restrict__ y, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i];
}
__global__ void read16(const double2 *__restrict__ x, double *out, size_t n)
{
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 v = x[i];
        acc += v.x + v.y;
    }
    if (acc == 12345.678) out[0] = acc;
}

int main()
{
    const size_t n = (size_t)1 << 29; // 4 GiB of doubles: 16x the Infinity Cache
    double *x, *y;
    if (hipMalloc(&x, n * 8) != hipSuccess || hipMalloc(&y, n * 8) != hipSuccess) return 1;
    (void)hipMemset(x, 0, n * 8);
    (void)hipMemset(y, 0, n * 8);
    (void)hipDeviceSynchronize();
    const int grid = 256 * 16, bs = 256;
    for (int rep = 0; rep < 2; ++rep) {
        read8<<<grid, bs>>>(x, y, n);
        write8<<<grid, bs>>>(y, n);
        copy8<<<grid, bs>>>(x, y, n);
        read16<<<grid, bs>>>((const double2 *)x, y, n / 2);
    }
    (void)hipDeviceSynchronize();
    printf("bytes per kernel: read8 %zu, write8 %zu, copy8 %zu read + %zu written, read16 %zu\n", n * 8, n * 8, n * 8, n * 8, n * 8);
    return 0;
}
