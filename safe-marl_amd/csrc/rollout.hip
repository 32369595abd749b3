// rollout.hip — one vector step's transition packing, hand-over and statistics in one launch (gfx950).
// Boundary: include/flexnet.h (FlexRolloutPackArgs).  Reference: madrl/models/model.py:230-262, utils/replay_buffer.py:23-27.
// Pure data movement: ~8.4 KB per environment (two observations, two hidden states, action, reward) -> 35 MB per step at
// 4096 envs, which the PyTorch path moved with some twenty pointwise launches.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flexnet.h"

#define PACK_THREADS 256
#define PACK_ENVS 2                // environments per block (their statistics leave as one atomic per quantity and block)

__global__ __launch_bounds__(PACK_THREADS) void rollout_pack_kernel(FlexRolloutPackArgs a) {
    __shared__ double stat[PACK_ENVS][10];
    const int tid = threadIdx.x;
    const int no = a.n_agents * a.obs_dim, na = a.n_agents * a.act_dim, nh = a.n_agents * FLEXNET_HID;
    const int e0 = blockIdx.x * PACK_ENVS;
    for (int k = 0; k < PACK_ENVS; ++k) {
        const int e = e0 + k;
        if (e >= a.n_envs) break;
        float* rec = a.rec + (int64_t)e * a.rec_stride;
        const float done = a.done[e] ? 1.0f : 0.0f;
        for (int i = tid; i < no; i += PACK_THREADS) {
            const float prev = a.obs_prev[(int64_t)e * no + i], next = a.obs_next[(int64_t)e * no + i];
            rec[a.col_state + i] = prev;                               // model.py:230
            rec[a.col_next_state + i] = next;                          // model.py:236
            a.obs_state[(int64_t)e * no + i] = next;                   // model.py:262: state = next_state
        }
        for (int i = tid; i < nh; i += PACK_THREADS) {
            const float hp = a.hid_prev[(int64_t)e * nh + i], hn = a.hid_new[(int64_t)e * nh + i];
            rec[a.col_last_hid + i] = hp;
            rec[a.col_hid + i] = hn;
            a.hid_state[(int64_t)e * nh + i] = hn * (1.0f - done);     // fresh hidden state for a new episode
        }
        for (int i = tid; i < na; i += PACK_THREADS) rec[a.col_action + i] = a.action[(int64_t)e * na + i];
        if (tid < a.n_agents) rec[a.col_reward + tid] = (float)a.reward[e];
        if (tid == 0) { rec[a.col_done] = done; rec[a.col_last_step] = done; }
        if (tid < 10) {
            double v = 0.0;
            if (tid < a.info_w && tid < 8) v = a.info ? a.info[(int64_t)e * a.info_w + tid] : 0.0;
            else if (tid == 8) v = a.reward[e];
            else if (tid == 9) v = (a.failed && a.failed[e]) ? 1.0 : 0.0;
            stat[k][tid] = v;
        }
    }
    __syncthreads();
    if (tid < 10) {
        const int n = min(PACK_ENVS, a.n_envs - e0);
        double sum = 0.0;
        for (int k = 0; k < n; ++k) sum += stat[k][tid];
        if (tid < 8) { if (a.info_sum && tid < a.info_w) unsafeAtomicAdd(&a.info_sum[tid], sum); }
        else if (tid == 8) unsafeAtomicAdd(a.rew_sum, sum);
        else if (a.fail_sum) unsafeAtomicAdd(a.fail_sum, sum);
    }
}

extern "C" int flexnet_rollout_pack(const FlexRolloutPackArgs* a, void* stream) {
    if (!a || a->n_envs < 0) return FLEXNET_EINVAL;
    if (a->n_envs == 0) return FLEXNET_OK;
    if (!a->obs_prev || !a->action || !a->reward || !a->obs_next || !a->done || !a->hid_prev || !a->hid_new || !a->rec ||
        !a->obs_state || !a->hid_state || !a->rew_sum || a->info_w < 0 || a->info_w > 8 || a->n_agents < 1 ||
        a->n_agents > PACK_THREADS)
        return FLEXNET_EINVAL;
    const int blocks = (a->n_envs + PACK_ENVS - 1) / PACK_ENVS;
    hipLaunchKernelGGL(rollout_pack_kernel, dim3(blocks), dim3(PACK_THREADS), 0, (hipStream_t)stream, *a);
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}
