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
#include "window_refresh.h"

#define OPT_THREADS 256
#define OPT_BLOCKS 64              // partial sums of squares, one per block (= one per lane of the wavefront that folds them)

// Both passes walk the tensors as ONE flat index space (thread g takes flat elements g, g + T, g + 2 T, ..., T = 16 384
// threads) and handle four elements at a time with all their loads issued before the first is used.  As a loop over the
// tensors with a grid-stride loop inside (rounds 1-4) the step kernel was a chain of up to sixteen dependent memory round
// trips of 1-3 elements per thread: 12-14 us for 52 097 elements.
#define OPT_UNROLL 4
struct OptFlat {                       // prefix sums of numel in LDS: flat index -> (tensor, offset)
    int64_t pre[FLEXNET_OPT_MAX_TENSORS + 1];
};
__device__ __forceinline__ void opt_build(const FlexClipRmspropArgs& a, OptFlat& f) {
    if (threadIdx.x == 0) {
        int64_t acc = 0;
        for (int t = 0; t < a.n_tensors; ++t) { f.pre[t] = acc; acc += a.numel[t]; }
        for (int t = a.n_tensors; t <= FLEXNET_OPT_MAX_TENSORS; ++t) f.pre[t] = acc;
    }
}
__device__ __forceinline__ int opt_find(const OptFlat& f, int n_tensors, int64_t i) {
    int t = 0;
#pragma unroll
    for (int k = 1; k < FLEXNET_OPT_MAX_TENSORS; ++k) t += (k < n_tensors && i >= f.pre[k]) ? 1 : 0;
    return t;
}

// RIDER (flexnet_clip_rmsprop_refresh): inside an update event's graph the optimiser step that ends value sub-update j is
// followed by the refresh of the static batch for sub-update j + 1 (csrc/window_refresh.h) — which depends on nothing the step
// computes, and whose targets (the batch's small columns, the in-place windows' cells, the reward statistics) nobody reads
// any more once sub-update j's weight gradient and finish are done.  Its copy blocks ride behind pass 1's blocks, its
// statistics blocks behind pass 2's (on the rewards pass 1's riders have just copied): no launch of its own (12 us of a
// 335-us sub-update).
static_assert(OPT_THREADS == WINDOW_THREADS, "the refresh blocks ride in the optimiser's launches");
struct WindowRider {
    FlexWindowRefreshArgs a;
    WindowPlan p;
    FlexTdLossArgs td;
    int32_t has_td, copy_blocks;
};

// pass 1: this block's share of sum g^2 (fixed order per thread, fixed trees above it: bit-reproducible)
template <bool RIDER>
__global__ __launch_bounds__(OPT_THREADS) void clip_norm_kernel(FlexClipRmspropArgs a, float* partial, WindowRider r) {
    if constexpr (RIDER) {
        if (blockIdx.x >= OPT_BLOCKS) { window_refresh_copy_block(r.a, r.p, blockIdx.x - OPT_BLOCKS); return; }
    }
    __shared__ float part[OPT_THREADS / 64];
    __shared__ OptFlat f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    opt_build(a, f);
    __syncthreads();
    const int64_t n = f.pre[FLEXNET_OPT_MAX_TENSORS];
    const int64_t gid = (int64_t)blockIdx.x * OPT_THREADS + tid, T = (int64_t)OPT_BLOCKS * OPT_THREADS;
    float ss = 0.0f;
    for (int64_t base = gid; base < n; base += OPT_UNROLL * T) {
        float g[OPT_UNROLL];
#pragma unroll
        for (int u = 0; u < OPT_UNROLL; ++u) {
            const int64_t i = base + u * T;
            const bool live = i < n;
            const int t = opt_find(f, a.n_tensors, live ? i : 0);
            g[u] = live ? a.grad[t][i - f.pre[t]] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < OPT_UNROLL; ++u) ss = fmaf(g[u], g[u], ss);
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
template <bool RIDER>
__global__ __launch_bounds__(OPT_THREADS) void clip_rmsprop_kernel(FlexClipRmspropArgs a, const float* partial, WindowRider r) {
    if constexpr (RIDER) {
        if (blockIdx.x >= OPT_BLOCKS) {
            // the statistics of the rewards the norm launch's copy blocks have just written: read from the contiguous copy
            // (td.reward), not from the ring's 108-byte-apart rows — the same rows in the same partition, 3 us less of chain
            const TdRewardRows rr = {r.td.reward, r.td.reward, (int64_t)r.td.rows, r.td.n_agents, r.td.n_agents};
            td_stats_block(r.td, rr, blockIdx.x - OPT_BLOCKS);
            return;
        }
    }
    const int tid = threadIdx.x;
    __shared__ float tot_s;
    __shared__ OptFlat f;
    opt_build(a, f);
    if (tid < 64) {                                                      // OPT_BLOCKS == 64: one partial per lane, fixed tree
        float t = partial[tid];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        if (tid == 0) tot_s = t;
    }
    __syncthreads();
    const int64_t n = f.pre[FLEXNET_OPT_MAX_TENSORS];
    const int64_t gid = (int64_t)blockIdx.x * OPT_THREADS + tid, T = (int64_t)OPT_BLOCKS * OPT_THREADS;
    const float norm = sqrtf(tot_s);
    if (gid == 0 && a.total_norm) *a.total_norm = norm;
    float coef = 1.0f;
    if (a.max_norm > 0.0f) coef = fminf(a.max_norm / (norm + 1e-6f), 1.0f);
    const float alpha = a.alpha, beta = 1.0f - a.alpha, eps = a.eps, lr = a.lr;
    for (int64_t base = gid; base < n; base += OPT_UNROLL * T) {
        float *gp[OPT_UNROLL], *pp[OPT_UNROLL], *vp[OPT_UNROLL];
        float g[OPT_UNROLL], v[OPT_UNROLL], pv[OPT_UNROLL];
        bool live[OPT_UNROLL];
#pragma unroll
        for (int u = 0; u < OPT_UNROLL; ++u) {
            const int64_t i = base + u * T;
            live[u] = i < n;
            const int t = opt_find(f, a.n_tensors, live[u] ? i : 0);
            const int64_t o = live[u] ? i - f.pre[t] : 0;
            gp[u] = a.grad[t] + o; pp[u] = a.param[t] + o; vp[u] = a.square_avg[t] + o;
            g[u] = *gp[u]; v[u] = *vp[u]; pv[u] = *pp[u];                 // (index 0 of tensor 0 for a dead slot: valid, unused)
        }
#pragma unroll
        for (int u = 0; u < OPT_UNROLL; ++u) {
            const float gi = g[u] * coef;
            const float vi = fmaf(beta * gi, gi, alpha * v[u]);
            if (live[u]) {
                *gp[u] = gi;
                *vp[u] = vi;
                *pp[u] = pv[u] - lr * (gi / (sqrtf(vi) + eps));
            }
        }
    }
    if (gid < a.n_tensors && a.step[gid]) *a.step[gid] += 1.0f;
}

static int clip_rmsprop_run(const FlexClipRmspropArgs* a, const WindowRider* r, void* stream) {
    if (!a || a->n_tensors < 0 || a->n_tensors > FLEXNET_OPT_MAX_TENSORS) return FLEXNET_EINVAL;
    int64_t total = 0;
    for (int t = 0; t < a->n_tensors; ++t) {
        if (!a->param[t] || !a->grad[t] || !a->square_avg[t] || a->numel[t] < 0) return FLEXNET_EINVAL;
        total += a->numel[t];
    }
    if (total > FLEXNET_OPT_MAX_ELEMENTS) return FLEXNET_EUNSUPPORTED;      // a fixed small grid: meant for the MADDPG networks
    if (!a->workspace) return FLEXNET_EINVAL;
    if (a->n_tensors == 0) return r ? FLEXNET_EINVAL : FLEXNET_OK;
    hipStream_t s = (hipStream_t)stream;
    if (r) {
        hipLaunchKernelGGL(clip_norm_kernel<true>, dim3(OPT_BLOCKS + r->copy_blocks), dim3(OPT_THREADS), 0, s, *a, a->workspace, *r);
        hipLaunchKernelGGL(clip_rmsprop_kernel<true>, dim3(OPT_BLOCKS + (r->has_td ? TD_BLOCKS : 0)), dim3(OPT_THREADS), 0, s, *a, a->workspace, *r);
    } else {
        WindowRider none;
        none.has_td = none.copy_blocks = 0;                       // (never read without RIDER)
        hipLaunchKernelGGL(clip_norm_kernel<false>, dim3(OPT_BLOCKS), dim3(OPT_THREADS), 0, s, *a, a->workspace, none);
        hipLaunchKernelGGL(clip_rmsprop_kernel<false>, dim3(OPT_BLOCKS), dim3(OPT_THREADS), 0, s, *a, a->workspace, none);
    }
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}

extern "C" int flexnet_clip_rmsprop(const FlexClipRmspropArgs* a, void* stream) { return clip_rmsprop_run(a, nullptr, stream); }

// flexnet_clip_rmsprop(a) followed by flexnet_window_refresh(refresh, td) in the optimiser's two launches (include/flexnet.h):
// the caller guarantees that nothing still reads what the refresh writes (see WindowRider above).
extern "C" int flexnet_clip_rmsprop_refresh(const FlexClipRmspropArgs* a, const FlexWindowRefreshArgs* refresh, const FlexTdLossArgs* td,
                                            void* stream) {
    if (!a || !refresh) return FLEXNET_EINVAL;
    // (the statistics blocks read the copy of the rewards the copy blocks make: td->reward must BE that copy)
    if (td && (refresh->reward_job < 0 || refresh->reward_job >= refresh->n_jobs || td->reward != refresh->dst[refresh->reward_job]))
        return FLEXNET_EINVAL;
    WindowRider r;
    const int rc = window_refresh_prepare(refresh, td, &r.p, &r.copy_blocks);
    if (rc != FLEXNET_OK) return rc;
    r.a = *refresh;
    r.has_td = td ? 1 : 0;
    if (td) r.td = *td;
    return clip_rmsprop_run(a, &r, stream);
}
