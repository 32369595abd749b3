// Launch-cadence floor probe (diagnostic, not part of the product): how long does one launch of the step kernel's
// SHAPE take when it computes nothing?  (a) empty kernel, (b) the observation copy alone (ring -> obs, the 5.6 KB per
// env that dominate the step's traffic) with plain and with nontemporal stores.  Back-to-back launches on one
// stream, timed with events, like bench.py's loop.
//   hipcc -O3 --offload-arch=gfx950 -o tools/floor_probe tools/floor_probe.hip && tools/floor_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256, 2) void k_empty(int* sink) { if (sink && threadIdx.x == 9999) sink[0] = 1; }

template <bool NT>
__global__ __launch_bounds__(256, 2) void k_copy(const float2* __restrict__ ring, float2* __restrict__ obs, int n_envs, int units) {
    const int env = (blockIdx.x * 256 + threadIdx.x) >> 5, l = threadIdx.x & 31;
    if (env >= n_envs) return;
    const float2* r = ring + (size_t)env * units;
    float2* o = obs + (size_t)env * units;
    float2 buf[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) { const int u = l + 32 * j; buf[j] = r[u < units ? u : 0]; }
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const int u = l + 32 * j;
        if (u < units) {
            if (NT) { v2f t; t.x = buf[j].x; t.y = buf[j].y; __builtin_nontemporal_store(t, reinterpret_cast<v2f*>(o + u)); }
            else o[u] = buf[j];
        }
    }
}

template <typename F> static double time_launches(F launch, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 200; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / iters;
}

int main() {
    const int n_envs = 4096, units = 360;
    float2 *ring, *obs; int* sink;
    CK(hipMalloc(&ring, (size_t)n_envs * units * 8)); CK(hipMalloc(&obs, (size_t)n_envs * units * 8)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(ring, 0, (size_t)n_envs * units * 8));
    const dim3 grid(n_envs / 8), block(256);
    printf("empty kernel, %d blocks: %.2f us/launch\n", grid.x, time_launches([&] { hipLaunchKernelGGL(k_empty, grid, block, 0, 0, sink); }, 3000));
    printf("empty kernel, 1 block:   %.2f us/launch\n", time_launches([&] { hipLaunchKernelGGL(k_empty, dim3(1), block, 0, 0, sink); }, 3000));
    printf("obs copy (plain stores): %.2f us/launch\n", time_launches([&] { hipLaunchKernelGGL(k_copy<false>, grid, block, 0, 0, ring, obs, n_envs, units); }, 3000));
    printf("obs copy (nt stores):    %.2f us/launch\n", time_launches([&] { hipLaunchKernelGGL(k_copy<true>, grid, block, 0, 0, ring, obs, n_envs, units); }, 3000));
    printf("obs copy ping-pong plain:%.2f us/launch\n", time_launches([&] { hipLaunchKernelGGL(k_copy<false>, grid, block, 0, 0, ring, obs, n_envs, units); hipLaunchKernelGGL(k_copy<false>, grid, block, 0, 0, obs, ring, n_envs, units); }, 1500) / 2);
    printf("obs copy ping-pong nt:   %.2f us/launch\n", time_launches([&] { hipLaunchKernelGGL(k_copy<true>, grid, block, 0, 0, ring, obs, n_envs, units); hipLaunchKernelGGL(k_copy<true>, grid, block, 0, 0, obs, ring, n_envs, units); }, 1500) / 2);
    return 0;
}
