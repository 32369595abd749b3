// critic_finish.h — the second stage that ends flexnet_critic_td_backward (csrc/critic.hip): the parameter gradients'
// fixed-order sums, the id-column sums and the loss / running-statistics finish.  A header because these blocks may RIDE in
// another kernel's launch: they depend on the backward kernel alone, and the only launch between them and their consumer (the
// gradient clip) is the first layer's weight gradient with its own second stage (csrc/wgrad.hip) — as blocks behind that
// second stage's (flexnet_wgrad_critic_finish) they cost no launch of their own (6.1 us of a 346-us value sub-update: a
// replayed graph's small kernels are chains of dependent memory round trips, and each of them starts when the one before
// has drained).  Reference: madrl/models/maddpg.py:100-123, madrl/critics/mlp_critic.py:25-33.
#ifndef FLEX_CRITIC_FINISH_H
#define FLEX_CRITIC_FINISH_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flexnet.h"
#include "flex_reduce.h"
#include "flex_td.h"

#define CF_HID FLEXNET_HID
#define CRITIC_WS_PITCH 4416                // floats per block in the workspace (4353 used)
#define DZF_PITCH (FLEXNET_MAX_AGENTS * CF_HID)
#define CRITIC_RED_BLOCKS ((CF_HID * CF_HID + 4 * CF_HID + 1 + 63) / 64)

// What the finish blocks need (filled by critic_finish_prepare, csrc/critic.hip): `blocks` = how many of them there are
struct CriticFinishK {
    FlexCriticTailArgs a;
    FlexTdLossArgs td;
    int32_t nb, dz_blocks, blocks, pad;
    int64_t dz_off;
};

// second stage of the deterministic path: element e of every block's partial row, summed in a fixed order, ADDED to
// the caller's gradient tensor (or stored there: overwrite_grads).  64 elements x 16 block groups per thread block: each thread walks its group's rows
// with eight loads in flight, the 16 group sums are folded through LDS in index order.
__device__ __forceinline__ void critic_reduce(const FlexCriticTailArgs& a, int blocks, int chunk) {
    const int e = chunk * 64 + (threadIdx.x & 63);
    float sum;
    if (!flex_reduce_rows(a.workspace + e, CRITIC_WS_PITCH, blocks, e < CF_HID * CF_HID + 4 * CF_HID + 1, sum)) return;
    float* dst;
    if (e < CF_HID * CF_HID) dst = a.d_fc2_w + e;
    else if (e < CF_HID * CF_HID + CF_HID) dst = a.d_fc2_b + (e - CF_HID * CF_HID);
    else if (e < CF_HID * CF_HID + 2 * CF_HID) dst = a.d_fc3_w + (e - CF_HID * CF_HID - CF_HID);
    else if (e < CF_HID * CF_HID + 3 * CF_HID) { if (!a.layernorm) return; dst = a.d_ln_w + (e - CF_HID * CF_HID - 2 * CF_HID); }
    else if (e < CF_HID * CF_HID + 4 * CF_HID) { if (!a.layernorm) return; dst = a.d_ln_b + (e - CF_HID * CF_HID - 3 * CF_HID); }
    else dst = a.d_fc3_b;
    *dst = a.overwrite_grads ? sum : *dst + sum;
}

// agent `agent`'s id-column sums from the fold kernel's per-block partial rows (fixed order)
__device__ __forceinline__ void critic_dz_reduce(const FlexCriticTailArgs& a, const float* partials, int blocks, int agent) {
    const int e = agent * 64 + (threadIdx.x & 63);                        // unit ex
    float sum;
    if (!flex_reduce_rows(partials + e, DZF_PITCH, blocks, true, sum)) return;
    const int sa = a.d_z_id_agent_stride, su = a.d_z_id_unit_stride;
    if (sa == 0 && su == 0) a.d_z_id[e] = sum;
    else a.d_z_id[(int64_t)agent * sa + (int64_t)(threadIdx.x & 63) * su] = sum;
}

// finish block `bx` of k.blocks (thread block of 64 x FLEX_RED_G threads): blocks 0 .. CRITIC_RED_BLOCKS - 1 the parameter
// gradients, then one block per agent for the id-column sums (composed input), the last one the loss / running statistics
__device__ __forceinline__ void critic_finish_block(const CriticFinishK& k, int bx) {
    if (bx < CRITIC_RED_BLOCKS) { critic_reduce(k.a, k.nb, bx); return; }
    if (bx < CRITIC_RED_BLOCKS + (k.dz_blocks > 0 ? k.a.n_agents : 0)) {
        critic_dz_reduce(k.a, k.a.workspace + k.dz_off, k.dz_blocks, bx - CRITIC_RED_BLOCKS);
        return;
    }
    if (threadIdx.x < 64) td_finish(k.td, k.nb, threadIdx.x);
}

// csrc/critic.hip: the checks of flexnet_critic_td_backward_phases and the geometry of its finish launch, for a caller
// (csrc/wgrad.hip) that carries the finish blocks in its own launch.  Returns a FLEXNET_* code.
int critic_finish_prepare(const FlexCriticTailArgs* a, const FlexTdLossArgs* t, CriticFinishK* out);

#endif
