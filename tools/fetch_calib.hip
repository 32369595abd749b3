// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS path's access shapes
// (MI355X_MICROARCH.md §HBM: FETCH_SIZE reads 1/2 of the bytes of a 16 B/lane stream; other widths are uncalibrated).
// Each kernel moves a known byte count, larger than the 256 MiB Infinity Cache, once.
//   hipcc -O3 --offload-arch=gfx950 -o tools/fetch_calib tools/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- tools/fetch_calib     (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void read8(const double* __restrict__ x, double* __restrict__ out, size_t n) {      // 8 B per lane, coalesced
    double acc = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += x[i];
    if (acc == 12345.678) out[0] = acc;
}
__global__ void read16(const double2* __restrict__ x, double* __restrict__ out, size_t n) {    // 16 B per lane
    double acc = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = x[i]; acc += v.x + v.y; }
    if (acc == 12345.678) out[0] = acc;
}
__global__ void read_rows(const double* __restrict__ x, double* __restrict__ out, size_t rows) {   // one 264-B row per 33 lanes of a wave (the series-row shape)
    double acc = 0.0;
    const int lane = threadIdx.x & 63;
    for (size_t r = blockIdx.x * (size_t)(blockDim.x / 64) + (threadIdx.x >> 6); r < rows; r += (size_t)gridDim.x * (blockDim.x / 64))
        if (lane < 33) acc += x[r * 72 + lane];
    if (acc == 12345.678) out[0] = acc;
}
__global__ void write8(double* __restrict__ y, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = 1.0;
}
__global__ void write_f2(float2* __restrict__ y, size_t n) {                                   // 8 B per lane fp32 pairs (the obs store shape)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = make_float2(1.f, 2.f);
}

int main() {
    const size_t bytes = 1ull << 30;                   // 1 GiB per pass
    double *x, *out;
    hipMalloc(&x, bytes); hipMalloc(&out, 64);
    hipMemset(x, 0, bytes);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(read8, dim3(4096), dim3(256), 0, 0, x, out, bytes / 8);
    hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, (const double2*)x, out, bytes / 16);
    hipLaunchKernelGGL(read_rows, dim3(4096), dim3(256), 0, 0, x, out, bytes / (72 * 8));
    hipLaunchKernelGGL(write8, dim3(4096), dim3(256), 0, 0, x, bytes / 8);
    hipLaunchKernelGGL(write_f2, dim3(4096), dim3(256), 0, 0, (float2*)x, bytes / 8);
    hipDeviceSynchronize();
    printf("moved per kernel: read8 %zu B, read16 %zu B, read_rows %zu B useful (%zu B of lines touched), write8 %zu B, write_f2 %zu B\n",
           bytes, bytes, (bytes / (72 * 8)) * 264, (bytes / (72 * 8)) * 320, bytes, bytes);
    return 0;
}
