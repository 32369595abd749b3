// tdloss.hip — the value loss of the DDPG family and its gradient in three small launches (gfx950).
// Boundary: include/flexnet.h (flexnet_td_loss).
//
// madrl/models/maddpg.py:100-123 with model.py:308-323 in front of it:
//     r      = BatchNorm1d(n_agents)(reward)                 (train mode: batch statistics; running stats updated)
//     ret    = r + gamma (1 - done) Q'(s', pi'(s'))          (no gradient)
//     loss   = mean((ret - Q(s, a))^2),   dLoss/dQ = -2 (ret - Q) / (B n)
// On [32 768, 5] tensors PyTorch runs this as some sixteen launches of 4-15 us each (statistics, normalisation, five
// pointwise steps, the mean, and their backward) — a tenth of a value sub-update for 650 KB of data.  Here: per-block
// column sums (fp64, fixed order) -> apply + per-block sums of squares -> a one-wavefront finish that writes the loss
// and moves the running statistics exactly as nn.BatchNorm1d does (biased variance for the output, unbiased for
// running_var, momentum weighting, num_batches_tracked += 1).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flexnet.h"
#include "flex_td.h"

__global__ __launch_bounds__(TD_THREADS) void td_stats_kernel(FlexTdLossArgs a) {
    const TdRewardRows rr = {a.reward, a.reward, (int64_t)a.rows, a.n_agents, a.n_agents};
    td_stats_block(a, rr, blockIdx.x);
}

__global__ __launch_bounds__(TD_THREADS) void td_apply_kernel(FlexTdLossArgs a) {
    __shared__ float mean_s[TD_NA], scale_s[TD_NA], shift_s[TD_NA];
    __shared__ double red[TD_THREADS];
    const int tid = threadIdx.x, n = a.n_agents;
    if (tid < TD_NA) {
        float m, sc, sh;
        td_column_affine(a, tid, m, sc, sh);
        mean_s[tid] = m; scale_s[tid] = sc; shift_s[tid] = sh;
    }
    __syncthreads();
    const int64_t total = (int64_t)a.rows * n;
    const float inv = 1.0f / (float)total;
    double sq = 0.0;
    for (int64_t idx = (int64_t)blockIdx.x * TD_THREADS + tid; idx < total; idx += (int64_t)TD_BLOCKS * TD_THREADS) {
        const int b = (int)(idx / n), j = (int)(idx - (int64_t)b * n);
        const float rn = (a.reward[idx] - mean_s[j]) * scale_s[j] + shift_s[j];
        const float ret = rn + a.gamma * (1.0f - a.done[b]) * a.next_q[idx];
        const float delta = ret - a.q[idx];
        a.dq[idx] = -2.0f * delta * inv;
        sq += (double)delta * (double)delta;
    }
    red[tid] = sq;
    __syncthreads();
    for (int sft = TD_THREADS / 2; sft > 0; sft >>= 1) {
        if (tid < sft) red[tid] += red[tid + sft];
        __syncthreads();
    }
    if (tid == 0) reinterpret_cast<double*>(a.workspace)[TD_WS_SQ + blockIdx.x] = red[0];
}

__global__ __launch_bounds__(64) void td_finish_kernel(FlexTdLossArgs a, int sq_blocks) {
    td_finish(a, sq_blocks, threadIdx.x);
}

void flex_td_launch_stats(const FlexTdLossArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(td_stats_kernel, dim3(TD_BLOCKS), dim3(TD_THREADS), 0, s, a);
}

void flex_td_launch_finish(const FlexTdLossArgs& a, int sq_blocks, hipStream_t s) {
    hipLaunchKernelGGL(td_finish_kernel, dim3(1), dim3(64), 0, s, a, sq_blocks);
}

extern "C" int flexnet_td_stats(const FlexTdLossArgs* a, void* stream) {
    if (!a || a->rows < 1 || a->n_agents < 1 || !a->reward || !a->workspace || a->workspace_floats < FLEXNET_TD_WS_FLOATS ||
        (reinterpret_cast<uintptr_t>(a->workspace) & 7) != 0)
        return FLEXNET_EINVAL;
    if (a->n_agents > TD_NA) return FLEXNET_EUNSUPPORTED;
    flex_td_launch_stats(*a, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}

extern "C" int flexnet_td_loss(const FlexTdLossArgs* a, void* stream) {
    if (!a || a->rows < 1 || a->n_agents < 1 || !a->reward || !a->workspace || a->workspace_floats < FLEXNET_TD_WS_FLOATS)
        return FLEXNET_EINVAL;
    // statistics only (q == NULL): the running statistics move as a training-mode forward of the BatchNorm would move
    // them, nothing else is computed — what a get_loss call that does not use the value loss still owes the module
    const bool stats_only = !a->q;
    if (stats_only && (!a->normalise || a->next_q || a->dq || a->loss)) return FLEXNET_EINVAL;
    if (!stats_only && (!a->done || !a->next_q || !a->dq || !a->loss)) return FLEXNET_EINVAL;
    if (a->n_agents > TD_NA) return FLEXNET_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(a->workspace) & 7) != 0) return FLEXNET_EINVAL;         // holds doubles
    hipStream_t s = (hipStream_t)stream;
    if (a->normalise && !a->stats_ready) flex_td_launch_stats(*a, s);
    if (!stats_only) hipLaunchKernelGGL(td_apply_kernel, dim3(TD_BLOCKS), dim3(TD_THREADS), 0, s, *a);
    flex_td_launch_finish(*a, stats_only ? 0 : TD_BLOCKS, s);
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}


// ---- fixed-order scalar sum (flexnet_scaled_sum): the reported means of the losses, graph-replay safe -------------------
__global__ __launch_bounds__(TD_THREADS) void sum_partial_kernel(FlexSumArgs a) {
    __shared__ double red[TD_THREADS];
    const int tid = threadIdx.x;
    double s0 = 0.0, s1 = 0.0;
    int64_t i = (int64_t)blockIdx.x * TD_THREADS + tid;
    const int64_t stride = (int64_t)TD_BLOCKS * TD_THREADS;
    for (; i + stride < a.n; i += 2 * stride) { s0 += (double)a.x[i]; s1 += (double)a.x[i + stride]; }
    if (i < a.n) s0 += (double)a.x[i];
    red[tid] = s0 + s1;
    __syncthreads();
    for (int sft = TD_THREADS / 2; sft > 0; sft >>= 1) {
        if (tid < sft) red[tid] += red[tid + sft];
        __syncthreads();
    }
    if (tid == 0) reinterpret_cast<double*>(a.workspace)[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(64) void sum_finish_kernel(FlexSumArgs a) {
    if (threadIdx.x != 0) return;
    const double* ws = reinterpret_cast<const double*>(a.workspace);
    double t = 0.0;
    for (int b = 0; b < TD_BLOCKS; ++b) t += ws[b];
    *a.out = (float)(t * (double)a.scale);
}

extern "C" int flexnet_scaled_sum(const FlexSumArgs* a, void* stream) {
    if (!a || a->n < 1 || !a->x || !a->out || !a->workspace || a->workspace_floats < FLEXNET_SUM_WS_FLOATS ||
        (reinterpret_cast<uintptr_t>(a->workspace) & 7) != 0)
        return FLEXNET_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sum_partial_kernel, dim3(TD_BLOCKS), dim3(TD_THREADS), 0, s, *a);
    hipLaunchKernelGGL(sum_finish_kernel, dim3(1), dim3(64), 0, s, *a);
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}
