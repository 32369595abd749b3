// rollout.hip — one vector step's transition packing, hand-over and statistics in one launch (gfx950).
// Boundary: include/flexnet.h (FlexRolloutPackArgs).  Reference: madrl/models/model.py:230-262, utils/replay_buffer.py:23-27.
// Pure data movement: ~8.4 KB per environment (two observations, two hidden states, action, reward) -> 35 MB per step at
// 4096 envs, which the PyTorch path moved with some twenty pointwise launches.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flexnet.h"

#define PACK_THREADS 256
#define PACK_ENVS 2                // environments per copy block
#define PACK_STATS 10               // statistics blocks at the head of the grid

__global__ __launch_bounds__(PACK_THREADS) void rollout_pack_kernel(FlexRolloutPackArgs a) {
    if (blockIdx.x < PACK_STATS) {
        // the first ten blocks own one statistic each (info columns, reward, failures): a block reduction over all
        // environments and one plain += — ten atomics per copy block (20 k on ten addresses) cost more than the copies
        __shared__ double red[PACK_THREADS];
        const int q = blockIdx.x, t = threadIdx.x;
        const bool is_info = q < 8;
        if (is_info && (q >= a.info_w || !a.info || !a.info_sum)) return;
        if (q == 9 && (!a.failed || !a.fail_sum)) return;
        double v = 0.0;
        for (int e = t; e < a.n_envs; e += PACK_THREADS)
            v += is_info ? a.info[(int64_t)e * a.info_w + q] : (q == 8 ? a.reward[e] : (a.failed[e] ? 1.0 : 0.0));
        red[t] = v;
        __syncthreads();
        for (int sft = PACK_THREADS / 2; sft > 0; sft >>= 1) {
            if (t < sft) red[t] += red[t + sft];
            __syncthreads();
        }
        if (t == 0) {
            if (is_info) a.info_sum[q] += red[0];
            else if (q == 8) { *a.rew_sum += red[0]; if (a.rng_state) a.rng_state[1] += 1; }     // this block always runs
            else *a.fail_sum += red[0];
        }
        return;
    }
    const int tid = threadIdx.x;
    const int no = a.n_agents * a.obs_dim, na = a.n_agents * a.act_dim, nh = a.n_agents * FLEXNET_HID;
    const int e0 = (blockIdx.x - PACK_STATS) * PACK_ENVS;
    for (int k = 0; k < PACK_ENVS; ++k) {
        const int e = e0 + k;
        if (e >= a.n_envs) break;
        float* rec = a.rec + (int64_t)e * a.rec_stride;
        const float done = a.done[e] ? 1.0f : 0.0f;
        // all loads first, then the stores: obs_state / hid_state may BE obs_prev / hid_prev, so the compiler has to keep
        // every load-store pair in order and would otherwise expose one memory round trip per loop iteration
        constexpr int MO = (FLEXNET_MAX_AGENTS * FLEXNET_MAX_OBS + PACK_THREADS - 1) / PACK_THREADS;     // 5
        constexpr int MH = (FLEXNET_MAX_AGENTS * FLEXNET_HID + PACK_THREADS - 1) / PACK_THREADS;         // 2
        float prev[MO], next[MO], hp[MH], hn[MH];
#pragma unroll
        for (int j = 0; j < MO; ++j) {
            const int i = tid + PACK_THREADS * j;
            if (i < no) { prev[j] = a.obs_prev[(int64_t)e * no + i]; next[j] = a.obs_next[(int64_t)e * no + i]; }
        }
#pragma unroll
        for (int j = 0; j < MH; ++j) {
            const int i = tid + PACK_THREADS * j;
            if (i < nh) { hp[j] = a.hid_prev[(int64_t)e * nh + i]; hn[j] = a.hid_new[(int64_t)e * nh + i]; }
        }
#pragma unroll
        for (int j = 0; j < MO; ++j) {
            const int i = tid + PACK_THREADS * j;
            if (i < no) {
                rec[a.col_state + i] = prev[j];                        // model.py:230
                rec[a.col_next_state + i] = next[j];                   // model.py:236
                a.obs_state[(int64_t)e * no + i] = next[j];            // model.py:262: state = next_state
            }
        }
#pragma unroll
        for (int j = 0; j < MH; ++j) {
            const int i = tid + PACK_THREADS * j;
            if (i < nh) {
                rec[a.col_last_hid + i] = hp[j];
                rec[a.col_hid + i] = hn[j];
                a.hid_state[(int64_t)e * nh + i] = hn[j] * (1.0f - done);   // fresh hidden state for a new episode
            }
        }
        for (int i = tid; i < na; i += PACK_THREADS) rec[a.col_action + i] = a.action[(int64_t)e * na + i];
        if (tid < a.n_agents) rec[a.col_reward + tid] = (float)a.reward[e];
        if (tid == 0) { rec[a.col_done] = done; rec[a.col_last_step] = done; }
    }
}

extern "C" int flexnet_rollout_pack(const FlexRolloutPackArgs* a, void* stream) {
    if (!a || a->n_envs < 0) return FLEXNET_EINVAL;
    if (a->n_envs == 0) return FLEXNET_OK;
    if (!a->obs_prev || !a->action || !a->reward || !a->obs_next || !a->done || !a->hid_prev || !a->hid_new || !a->rec ||
        !a->obs_state || !a->hid_state || !a->rew_sum || a->info_w < 0 || a->info_w > 8 || a->n_agents < 1 ||
        a->n_agents > FLEXNET_MAX_AGENTS || a->obs_dim > FLEXNET_MAX_OBS)
        return FLEXNET_EINVAL;
    const int blocks = (a->n_envs + PACK_ENVS - 1) / PACK_ENVS + PACK_STATS;
    hipLaunchKernelGGL(rollout_pack_kernel, dim3(blocks), dim3(PACK_THREADS), 0, (hipStream_t)stream, *a);
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}
