// optim.hip — gradient-norm clip + RMSprop step for a whole (small) network in ONE launch (gfx950).
// Boundary: include/flexnet.h (flexnet_clip_rmsprop).
//
// madrl/utils/trainer.py:86-90,103-107:  clip_grad_norm_(params, grad_clip_eps); optimizer.step()  with
// torch.optim.RMSprop(alpha=0.99, eps=1e-5) (trainer.py:34-35).  The actor has 34 948 parameters in 10 tensors, the
// critic 52 097 in 8: PyTorch spends some ten launches (norm per tensor, norm of norms, clip factor, scale, and five
// foreach kernels of the optimiser) of 5-13 us each on what is 350 KB of data.  Here two launches of 64 small blocks:
// sums of squares per block (fixed-order trees, folded in index order: the norm is bit-reproducible), then per element
//     g <- g * min(1, max_norm / (norm + 1e-6));  v <- alpha v + (1 - alpha) g^2;  p <- p - lr g / (sqrt(v) + eps)
// with the clipped gradient written back as clip_grad_norm_ does, and the optimiser's per-tensor step counters advanced.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flexnet.h"

#define OPT_THREADS 256
#define OPT_BLOCKS 64              // partial sums of squares, one per block (= one per lane of the wavefront that folds them)

// pass 1: this block's share of sum g^2 (fixed tree inside the block)
__global__ __launch_bounds__(OPT_THREADS) void clip_norm_kernel(FlexClipRmspropArgs a, float* partial) {
    __shared__ float part[OPT_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t gid = (int64_t)blockIdx.x * OPT_THREADS + tid, stride = (int64_t)OPT_BLOCKS * OPT_THREADS;
    float ss = 0.0f;
    for (int t = 0; t < a.n_tensors; ++t) {
        const float* g = a.grad[t];
        const int64_t n = a.numel[t];
        for (int64_t i = gid; i < n; i += stride) ss = fmaf(g[i], g[i], ss);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if (lane == 0) part[wave] = ss;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.0f;
#pragma unroll
        for (int w = 0; w < OPT_THREADS / 64; ++w) tot += part[w];
        partial[blockIdx.x] = tot;
    }
}

// pass 2: norm from the partials, clip factor, RMSprop step on this block's share
__global__ __launch_bounds__(OPT_THREADS) void clip_rmsprop_kernel(FlexClipRmspropArgs a, const float* partial) {
    const int tid = threadIdx.x;
    const int64_t gid = (int64_t)blockIdx.x * OPT_THREADS + tid, stride = (int64_t)OPT_BLOCKS * OPT_THREADS;
    __shared__ float tot_s;
    if (tid < 64) {                                                      // OPT_BLOCKS == 64: one partial per lane, fixed tree
        float t = partial[tid];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        if (tid == 0) tot_s = t;
    }
    __syncthreads();
    const float norm = sqrtf(tot_s);
    if (gid == 0 && a.total_norm) *a.total_norm = norm;
    float coef = 1.0f;
    if (a.max_norm > 0.0f) coef = fminf(a.max_norm / (norm + 1e-6f), 1.0f);
    const float alpha = a.alpha, beta = 1.0f - a.alpha, eps = a.eps, lr = a.lr;
    for (int t = 0; t < a.n_tensors; ++t) {
        float* g = a.grad[t];
        float* p = a.param[t];
        float* v = a.square_avg[t];
        const int64_t n = a.numel[t];
        for (int64_t i = gid; i < n; i += stride) {
            const float gi = g[i] * coef;
            const float vi = fmaf(beta * gi, gi, alpha * v[i]);
            g[i] = gi;
            v[i] = vi;
            p[i] = p[i] - lr * (gi / (sqrtf(vi) + eps));
        }
        if (gid == 0 && a.step[t]) *a.step[t] += 1.0f;
    }
}

extern "C" int flexnet_clip_rmsprop(const FlexClipRmspropArgs* a, void* stream) {
    if (!a || a->n_tensors < 0 || a->n_tensors > FLEXNET_OPT_MAX_TENSORS) return FLEXNET_EINVAL;
    int64_t total = 0;
    for (int t = 0; t < a->n_tensors; ++t) {
        if (!a->param[t] || !a->grad[t] || !a->square_avg[t] || a->numel[t] < 0) return FLEXNET_EINVAL;
        total += a->numel[t];
    }
    if (total > FLEXNET_OPT_MAX_ELEMENTS) return FLEXNET_EUNSUPPORTED;      // a fixed small grid: meant for the MADDPG networks
    if (!a->workspace) return FLEXNET_EINVAL;
    if (a->n_tensors == 0) return FLEXNET_OK;
    hipLaunchKernelGGL(clip_norm_kernel, dim3(OPT_BLOCKS), dim3(OPT_THREADS), 0, (hipStream_t)stream, *a, a->workspace);
    hipLaunchKernelGGL(clip_rmsprop_kernel, dim3(OPT_BLOCKS), dim3(OPT_THREADS), 0, (hipStream_t)stream, *a, a->workspace);
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}
