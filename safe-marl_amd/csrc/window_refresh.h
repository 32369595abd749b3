// window_refresh.h — flexnet_window_refresh (include/flexnet.h): the refresh of a captured sub-update's static batch with the
// window's first slot read from device memory, as device functions — csrc/rollout.hip launches them as a kernel of their own,
// csrc/optim.hip carries them as riders of the optimiser step that ENDS the previous sub-update of an update event's graph
// (flexnet_clip_rmsprop_refresh).  Reference: utils/replay_buffer.py:17-21 (the window), model.py:308-323 (the statistics).
#ifndef FLEX_WINDOW_REFRESH_H
#define FLEX_WINDOW_REFRESH_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flexnet.h"
#include "flex_td.h"

#define WINDOW_THREADS 256
static_assert(WINDOW_THREADS == TD_THREADS, "the statistics blocks ride in the refresh launch");
struct WindowPlan { int first_block[FLEXNET_WINDOW_MAX_JOBS + 1]; };

// statistics block `bx` of TD_BLOCKS: the reward rows of job reward_job, read where the ring keeps them (two pieces at the seam)
__device__ __forceinline__ void window_refresh_td_block(const FlexWindowRefreshArgs& a, const FlexTdLossArgs& td, int bx) {
    const int64_t start = *a.start, cap = a.ring_rows;
    const int j = a.reward_job;
    const int64_t first = (start + a.row_off[j]) % cap;
    const int64_t rows0 = a.rows[j] < cap - first ? a.rows[j] : cap - first;
    const TdRewardRows rr = {a.base[j] + first * a.src_stride[j], a.base[j], rows0, a.src_stride[j], a.src_stride[j]};
    td_stats_block(td, rr, bx);
}

// copy block `bx` of p.first_block[n_jobs] (one block when there are cells only): block 0 also sets the cells
__device__ __forceinline__ void window_refresh_copy_block(const FlexWindowRefreshArgs& a, const WindowPlan& p, int bx) {
    const int64_t start = *a.start, cap = a.ring_rows;
    if (bx == 0 && threadIdx.x < a.n_cells) *a.cell[threadIdx.x] = start % a.cell_mod[threadIdx.x];
    if (a.n_jobs == 0) return;
    int j = 0;
    while (j + 1 < a.n_jobs && bx >= p.first_block[j + 1]) ++j;
    const int nb = p.first_block[j + 1] - p.first_block[j], b = bx - p.first_block[j];
    const float* __restrict__ src = a.base[j];
    float* __restrict__ dst = a.dst[j];
    const int w = a.width[j], ss = a.src_stride[j];
    const int rows = (int)a.rows[j];                          // (< 2^31 / width: checked by window_refresh_prepare)
    const int64_t first = (start + a.row_off[j]) % cap;
    const int until_seam = cap - first < rows ? (int)(cap - first) : rows;      // rows before the ring's seam
    if (w <= 8) {                                             // narrow columns: a thread per ROW (gather_rows_kernel)
        for (int r = b * WINDOW_THREADS + threadIdx.x; r < rows; r += nb * WINDOW_THREADS) {
            const int64_t pr = r < until_seam ? first + r : (int64_t)(r - until_seam);
            const float* sp = src + pr * ss;
            float* dp = dst + (int64_t)r * w;
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = c < w ? sp[c] : 0.0f;
#pragma unroll
            for (int c = 0; c < 8; ++c) if (c < w) dp[c] = v[c];
        }
    } else if (w <= 32) {
        // up to 32 columns (the action block: 20): a thread per ROW as well, four columns per round with their loads issued
        // together — no index division per element (one per element made this path a chain of sixteen dependent round trips
        // at 32 768 rows); columns past the row's end re-read its last one and are not stored
        for (int r = b * WINDOW_THREADS + threadIdx.x; r < rows; r += nb * WINDOW_THREADS) {
            const int64_t pr = r < until_seam ? first + r : (int64_t)(r - until_seam);
            const float* sp = src + pr * ss;
            float* dp = dst + (int64_t)r * w;
            for (int c0 = 0; c0 < w; c0 += 4) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = sp[c0 + u < w ? c0 + u : w - 1];
#pragma unroll
                for (int u = 0; u < 4; ++u) if (c0 + u < w) dp[c0 + u] = v[u];
            }
        }
    } else {
        const int total = rows * w;
        for (int i = b * WINDOW_THREADS + threadIdx.x; i < total; i += nb * WINDOW_THREADS) {
            const int r = i / w, c = i - r * w;
            const int64_t pr = r < until_seam ? first + r : (int64_t)(r - until_seam);
            dst[i] = src[pr * ss + c];
        }
    }
}

// argument checks and the block plan (host).  *copy_blocks >= 1.
static inline int window_refresh_prepare(const FlexWindowRefreshArgs* a, const FlexTdLossArgs* td, WindowPlan* p, int* copy_blocks) {
    if (!a || !a->start || a->n_jobs < 0 || a->n_jobs > FLEXNET_WINDOW_MAX_JOBS || a->n_cells < 0 ||
        a->n_cells > FLEXNET_WINDOW_MAX_CELLS || a->ring_rows < 1 || (a->n_jobs == 0 && a->n_cells == 0))
        return FLEXNET_EINVAL;
    for (int k = 0; k < a->n_cells; ++k) if (!a->cell[k] || a->cell_mod[k] < 1) return FLEXNET_EINVAL;
    int blocks = 0;
    for (int j = 0; j < a->n_jobs; ++j) {
        if (!a->base[j] || !a->dst[j] || a->rows[j] < 0 || a->rows[j] > a->ring_rows || a->row_off[j] < 0 || a->width[j] < 1 ||
            a->src_stride[j] < a->width[j] || a->rows[j] * (int64_t)a->width[j] >= 0x7fffffffll)
            return FLEXNET_EINVAL;
        p->first_block[j] = blocks;
        const int64_t bytes = a->rows[j] * (int64_t)a->width[j] * 4;
        int64_t nb = (bytes + 16383) / 16384;
        nb = nb < 1 ? 1 : (nb > 4096 ? 4096 : nb);
        blocks += (int)nb;
    }
    if (a->n_jobs == 0) blocks = 1;                                      // cells only: one block
    for (int j = a->n_jobs; j <= FLEXNET_WINDOW_MAX_JOBS; ++j) p->first_block[j] = blocks;
    if (a->n_jobs == 0) p->first_block[0] = 0;
    if (td) {
        const int j = a->reward_job;
        if (j < 0 || j >= a->n_jobs || a->width[j] != td->n_agents || a->rows[j] != td->rows || td->rows < 1 || td->n_agents < 1 ||
            td->n_agents > TD_NA || !td->workspace || td->workspace_floats < FLEXNET_TD_WS_FLOATS ||
            (reinterpret_cast<uintptr_t>(td->workspace) & 7) != 0)
            return FLEXNET_EINVAL;
    }
    *copy_blocks = blocks;
    return FLEXNET_OK;
}

#endif
