// flex_reduce.h — the second stage of every deterministic two-stage reduction of the learner kernels (critic.hip,
// wgrad.hip, lnrelu.hip): element e of `rows` partial rows (one per thread block of the first stage, `pitch` floats
// apart), summed in a FIXED order.  A thread block of 64 x FLEX_RED_G threads takes 64 consecutive elements: thread
// (ex, gy) walks rows gy, gy + G, gy + 2G, ... with four loads in flight into four accumulators, the G group sums are
// folded through LDS in index order.  The result is valid in the threads with gy == 0 (the others return false).
#ifndef FLEX_REDUCE_H
#define FLEX_REDUCE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FLEX_RED_G 16

// `src` already points at this thread's element (e = blockIdx.x * 64 + ex); `active` = e is a real element.
__device__ __forceinline__ bool flex_reduce_rows(const float* src, int64_t pitch, int rows, bool active, float& sum) {
    __shared__ float part[FLEX_RED_G][64];
    const int ex = threadIdx.x & 63, gy = threadIdx.x >> 6;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (active) {
        int b = gy;
        for (; b + 3 * FLEX_RED_G < rows; b += 4 * FLEX_RED_G) {
            s0 += src[(int64_t)b * pitch];
            s1 += src[(int64_t)(b + FLEX_RED_G) * pitch];
            s2 += src[(int64_t)(b + 2 * FLEX_RED_G) * pitch];
            s3 += src[(int64_t)(b + 3 * FLEX_RED_G) * pitch];
        }
        for (; b < rows; b += FLEX_RED_G) s0 += src[(int64_t)b * pitch];
    }
    part[gy][ex] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (gy != 0 || !active) return false;
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < FLEX_RED_G; ++k) t += part[k][ex];
    sum = t;
    return true;
}

#endif
