// gru.hip — pointwise backward of the GRUCell for update batches (gfx950).  Boundary: include/flexnet.h (FlexGruBwdArgs).
// Reference: madrl/agents/rnn_agent.py:30-32 under autograd (nn.GRUCell + fc2).  The forward pass is csrc/actor.hip with its
// `save_*` outputs; this kernel turns the gradient arriving at the action means into the gradients of the two gate
// pre-activations.  One lane per hidden unit, four rows per wavefront pass with every load issued before the arithmetic;
// HBM-bound: 324 floats read and 384 written per row (464 MB at 163 840 rows).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flexnet.h"

#define GRU_THREADS 256
#define GRU_ROWS 4                  // rows per wavefront pass

__global__ __launch_bounds__(GRU_THREADS) void gru_backward_kernel(FlexGruBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * GRU_THREADS + threadIdx.x) >> 6, n_waves = (gridDim.x * GRU_THREADS) >> 6;
    float w2[FLEXNET_MAX_ACT];
#pragma unroll
    for (int k = 0; k < FLEXNET_MAX_ACT; ++k) w2[k] = k < a.act_dim ? a.fc2_w[k * FLEXNET_HID + lane] : 0.0f;
    for (int64_t r0 = (int64_t)wave * GRU_ROWS; r0 < a.rows; r0 += (int64_t)n_waves * GRU_ROWS) {
        float rg[GRU_ROWS], zg[GRU_ROWS], ng[GRU_ROWS], hn[GRU_ROWS], hp[GRU_ROWS], dh[GRU_ROWS], dm[GRU_ROWS][FLEXNET_MAX_ACT];
#pragma unroll
        for (int j = 0; j < GRU_ROWS; ++j) {
            const int64_t row = r0 + j < a.rows ? r0 + j : a.rows - 1;              // clamped: loads stay in bounds
            const int64_t at = row * FLEXNET_HID + lane;
            rg[j] = a.r[at]; zg[j] = a.z[at]; ng[j] = a.n[at]; hn[j] = a.hn[at]; hp[j] = a.h_prev[at];
            dh[j] = a.d_hidden ? a.d_hidden[at] : 0.0f;
#pragma unroll
            for (int k = 0; k < FLEXNET_MAX_ACT; ++k) dm[j][k] = k < a.act_dim ? a.d_means[row * a.act_dim + k] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < GRU_ROWS; ++j) {
            if (r0 + j >= a.rows) break;
            float g = dh[j];
#pragma unroll
            for (int k = 0; k < FLEXNET_MAX_ACT; ++k) g = fmaf(dm[j][k], w2[k], g);      // dh' = d_means @ fc2_w
            const float dn = g * (1.0f - zg[j]) * (1.0f - ng[j] * ng[j]);
            const float dz = g * (hp[j] - ng[j]) * zg[j] * (1.0f - zg[j]);
            const float dr = dn * hn[j] * rg[j] * (1.0f - rg[j]);
            float* gi = a.d_gi + (r0 + j) * (3 * FLEXNET_HID) + lane;
            __builtin_nontemporal_store(dr, gi);
            __builtin_nontemporal_store(dz, gi + FLEXNET_HID);
            __builtin_nontemporal_store(dn, gi + 2 * FLEXNET_HID);
            float* gh = a.d_gh + (r0 + j) * (3 * FLEXNET_HID) + lane;
            __builtin_nontemporal_store(dr, gh);
            __builtin_nontemporal_store(dz, gh + FLEXNET_HID);
            __builtin_nontemporal_store(dn * rg[j], gh + 2 * FLEXNET_HID);
        }
    }
}

extern "C" int flexnet_gru_backward(const FlexGruBwdArgs* a, void* stream) {
    if (!a || a->rows < 0) return FLEXNET_EINVAL;
    if (a->rows == 0) return FLEXNET_OK;
    if (!a->d_means || !a->fc2_w || !a->r || !a->z || !a->n || !a->hn || !a->h_prev || !a->d_gi || !a->d_gh || a->act_dim < 1)
        return FLEXNET_EINVAL;
    if (a->act_dim > FLEXNET_MAX_ACT) return FLEXNET_EUNSUPPORTED;
    const int64_t passes = ((int64_t)a->rows + GRU_ROWS - 1) / GRU_ROWS;
    int64_t blocks = (passes + GRU_THREADS / 64 - 1) / (GRU_THREADS / 64);
    if (blocks > 8192) blocks = 8192;                                         // grid-stride beyond 32 rows per CU slot
    hipLaunchKernelGGL(gru_backward_kernel, dim3((unsigned)blocks), dim3(GRU_THREADS), 0, (hipStream_t)stream, *a);
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}
