// flex_td.h — what csrc/tdloss.hip (flexnet_td_loss) and the critic backward that forms the TD error itself
// (csrc/critic.hip, flexnet_critic_td_backward) share: the workspace layout of the value loss and its launch helpers.
#ifndef FLEX_TD_H
#define FLEX_TD_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flexnet.h"

#define TD_THREADS 256
#define TD_BLOCKS 64
#define TD_NA FLEXNET_MAX_AGENTS
#define TD_SQ_MAX 1024                // per-block sums of squared TD errors the finish kernel may be asked to add up

// workspace (doubles): [TD_BLOCKS][2 TD_NA] column sums and sums of squares of the reward | [<= TD_SQ_MAX] sums of squared TD errors
#define TD_WS_SQ (TD_BLOCKS * 2 * TD_NA)

// batch mean and biased variance of reward column j from the statistics pass's per-block partial sums (fixed order)
__device__ __forceinline__ void td_column_stats(const FlexTdLossArgs& a, int j, double& mean, double& var) {
    const double* ws = reinterpret_cast<const double*>(a.workspace);
    double s = 0.0, ss = 0.0;
    for (int b = 0; b < TD_BLOCKS; ++b) { s += ws[b * 2 * TD_NA + j]; ss += ws[b * 2 * TD_NA + TD_NA + j]; }
    const double rows = (double)(a.stat_rows > 0 ? a.stat_rows : (int64_t)a.rows);   // (sums over all ranks' rows: stats_ready)
    mean = s / rows;
    var = ss / rows - mean * mean;                               // biased (what the normalisation uses)
    if (var < 0.0) var = 0.0;
}

// reward normalisation of column j as (r - mean) * scale + shift (nn.BatchNorm1d in training mode), or the identity
__device__ __forceinline__ void td_column_affine(const FlexTdLossArgs& a, int j, float& m, float& sc, float& sh) {
    m = 0.0f; sc = 1.0f; sh = 0.0f;
    if (a.normalise && j < a.n_agents) {
        double mean, var;
        td_column_stats(a, j, mean, var);
        const float w = a.bn_weight ? a.bn_weight[j] : 1.0f;
        m = (float)mean;
        sc = (float)(1.0 / sqrt(var + (double)a.bn_eps)) * w;
        sh = a.bn_bias ? a.bn_bias[j] : 0.0f;
    }
}

// One wavefront (lane = 0..63): the loss from `sq_blocks` per-block sums of squared TD errors — lane l adds those of blocks
// l, l + 64, ..., then a fixed shuffle tree — and the BatchNorm's running statistics moved as nn.BatchNorm1d (training
// mode) moves them: momentum weighting, unbiased variance, num_batches_tracked += 1.
__device__ __forceinline__ void td_finish(const FlexTdLossArgs& a, int sq_blocks, int lane) {
    const double* ws = reinterpret_cast<const double*>(a.workspace);
    double t = 0.0;
    for (int b = lane; b < sq_blocks; b += 64) t += ws[TD_WS_SQ + b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    if (lane == 0) {
        if (a.loss) *a.loss = (float)(t / ((double)a.rows * a.n_agents));
        if (a.normalise && a.num_batches_tracked) *a.num_batches_tracked += 1;
    }
    if (a.normalise && lane < a.n_agents && a.running_mean && a.running_var) {
        double mean, var;
        td_column_stats(a, lane, mean, var);
        const double m = (double)a.bn_momentum;
        const double nr = (double)(a.stat_rows > 0 ? a.stat_rows : (int64_t)a.rows);
        const double unbiased = nr > 1.0 ? var * nr / (nr - 1.0) : var;
        a.running_mean[lane] = (float)((1.0 - m) * (double)a.running_mean[lane] + m * mean);
        a.running_var[lane] = (float)((1.0 - m) * (double)a.running_var[lane] + m * unbiased);
    }
}

// The statistics pass of block `block` of TD_BLOCKS (TD_THREADS threads): per-block column sums and sums of squares of the
// reward (fp64, fixed order) into the workspace.  The [rows, n_agents] reward rows may lie in up to TWO pieces (a window that
// wraps a ring's seam, read where the replay keeps it): rows [0, rows0) at src0 + b * stride0, the rest at src1 + (b - rows0)
// * stride1.  td_stats_kernel runs it on a.reward itself; as blocks riding in flexnet_gather_rows_td's launch it reads the
// rows the same launch copies into a.reward — same values, same partition of the rows over blocks and threads, same sums.
struct TdRewardRows { const float* src0; const float* src1; int64_t rows0; int32_t stride0, stride1; };
__device__ __forceinline__ void td_stats_block(const FlexTdLossArgs& a, const TdRewardRows& rr, int block) {
    const int tid = threadIdx.x, n = a.n_agents;
    double s[TD_NA], ss[TD_NA];
#pragma unroll
    for (int j = 0; j < TD_NA; ++j) { s[j] = 0.0; ss[j] = 0.0; }
    // four rows' loads in flight per thread, added in row order (the order of the plain loop: the sums keep their bits) — as a
    // rider the pass reads rows a ring keeps 108 bytes apart, and a 131 072-row batch is eight dependent round trips otherwise
    constexpr int STEP = TD_BLOCKS * TD_THREADS;
    for (int b0 = block * TD_THREADS + tid; b0 < a.rows; b0 += 4 * STEP) {
        float v[4][TD_NA];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = b0 + u * STEP < a.rows ? b0 + u * STEP : b0;
            const float* r = b < rr.rows0 ? rr.src0 + (int64_t)b * rr.stride0 : rr.src1 + ((int64_t)b - rr.rows0) * rr.stride1;
#pragma unroll
            for (int j = 0; j < TD_NA; ++j) v[u][j] = j < n ? r[j] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (b0 + u * STEP < a.rows) {
#pragma unroll
                for (int j = 0; j < TD_NA; ++j)
                    if (j < n) { const double x = (double)v[u][j]; s[j] += x; ss[j] += x * x; }
            }
        }
    }
    // wavefront sums by shuffles (fixed tree), then the block's four wavefronts in index order
    __shared__ double part[TD_THREADS / 64][2 * TD_NA];
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int q = 0; q < 2 * TD_NA; ++q) {
        double v = q < TD_NA ? s[q < TD_NA ? q : 0] : ss[q < TD_NA ? 0 : q - TD_NA];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) part[wave][q] = v;
    }
    __syncthreads();
    if (tid < 2 * TD_NA) {
        double t = part[0][tid];
#pragma unroll
        for (int w = 1; w < TD_THREADS / 64; ++w) t += part[w][tid];
        reinterpret_cast<double*>(a.workspace)[(int64_t)block * 2 * TD_NA + tid] = t;
    }
}

// csrc/tdloss.hip: the statistics pass (per-block column sums of the reward) and the one-wavefront finish (loss from
// `sq_blocks` partial sums of squared errors; running statistics moved as nn.BatchNorm1d moves them)
void flex_td_launch_stats(const FlexTdLossArgs& a, hipStream_t s);
void flex_td_launch_finish(const FlexTdLossArgs& a, int sq_blocks, hipStream_t s);

#endif
