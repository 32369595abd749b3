// flex_launch.h — host-side helper shared by the learner kernels' entry points, and the wavefront sum they share.
#ifndef FLEX_LAUNCH_H
#define FLEX_LAUNCH_H
#include <hip/hip_runtime.h>

// Compute units of the current device (one process drives one GPU: cached after the first query); -1 on a HIP error.
static inline int flex_cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
            return -1;
        cus = n;
    }
    return cus;
}

// Sum over the 64 lanes, returned in every lane: DPP row shifts / broadcasts (no LDS traffic) and one v_readlane.
template <int CTRL, int ROW_MASK, bool BC>
__device__ __forceinline__ float flex_dpp_f32(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xF, BC));
}
__device__ __forceinline__ float flex_wave_sum(float v) {
    v += flex_dpp_f32<0x111, 0xF, true>(v);            // row_shr:1
    v += flex_dpp_f32<0x112, 0xF, true>(v);            // row_shr:2
    v += flex_dpp_f32<0x114, 0xF, true>(v);            // row_shr:4
    v += flex_dpp_f32<0x118, 0xF, true>(v);            // row_shr:8   -> lane 15 of each row holds the row sum
    v += flex_dpp_f32<0x142, 0xA, false>(v);           // row_bcast:15 into rows 1, 3
    v += flex_dpp_f32<0x143, 0xC, false>(v);           // row_bcast:31 into rows 2, 3 -> lane 63 holds the total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

#endif
