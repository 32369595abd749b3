// lnrelu.hip — the actor's first-layer epilogue for update batches, forward and backward (gfx950).
// Boundary: include/flexnet.h (flexnet_lnrelu_forward / flexnet_lnrelu_backward).
//
// madrl/agents/rnn_agent.py:25-29:  x = fc1(inputs); x = LayerNorm(x); x = relu(x), where the last n columns of
// `inputs` are the one-hot agent id (model.py:105-108).  The caller forms z = obs @ W_obs^T with one GEMM; this file
// does the rest in one pass over z: + fc1.bias + the id column of row r % n (what the one-hot block of the GEMM would
// add), LayerNorm, ReLU — and the whole backward of that chain, including the four parameter gradients, which autograd
// spreads over a concat, a broadcast add, three LayerNorm kernels and several column reductions (some 450 us per
// policy sub-update at 163 840 rows).  One lane per hidden unit, eight rows in flight per wavefront; HBM-bound:
// forward reads z and writes the activation, backward reads z and the upstream gradient and writes dz.
// Parameter gradients: per-block partial sums, folded in a fixed order by a second launch (bit-reproducible).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flexnet.h"
#include "flex_reduce.h"
#include "flex_launch.h"

#define HID FLEXNET_HID
#define LW 4                       // wavefronts per block
#define LR 8                       // rows per wavefront and iteration
#define LN_VECS (3 + FLEXNET_MAX_AGENTS)   // d_ln_w, d_ln_b, d_bias, d_id[8]
#define LN_PITCH (LN_VECS * HID)

struct LnRow { float xhat, rstd, y; };

__device__ __forceinline__ LnRow ln_row(float x, bool layernorm, float eps, float g, float b) {
    LnRow o;
    if (layernorm) {
        const float mean = flex_wave_sum(x) * (1.0f / HID);
        const float d = x - mean;
        const float var = flex_wave_sum(d * d) * (1.0f / HID);
        o.rstd = rsqrtf(var + eps);
        o.xhat = d * o.rstd;
        o.y = o.xhat * g + b;
    } else {
        o.rstd = 1.0f; o.xhat = x; o.y = x;
    }
    return o;
}

// bias + id column per agent, staged once per block: addend[i][lane]
__device__ __forceinline__ void stage_addend(const FlexLnReluArgs& a, float* addend) {
    for (int idx = threadIdx.x; idx < FLEXNET_MAX_AGENTS * HID; idx += 64 * LW) {
        const int i = idx / HID, u = idx - i * HID;
        float v = a.bias ? a.bias[u] : 0.0f;
        if (a.id_cols && i < a.n_agents) v += a.id_cols[i * HID + u];
        addend[idx] = v;
    }
    __syncthreads();
}

__global__ __launch_bounds__(64 * LW) void lnrelu_fwd_kernel(FlexLnReluArgs a) {
    __shared__ float addend[FLEXNET_MAX_AGENTS * HID];
    stage_addend(a, addend);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float g = a.layernorm ? a.ln_w[lane] : 1.0f, be = a.layernorm ? a.ln_b[lane] : 0.0f;
    const int n_tiles = (a.rows + LR - 1) / LR;
    for (int tile = blockIdx.x * LW + wave; tile < n_tiles; tile += gridDim.x * LW) {
        const int r0 = tile * LR;
        float x[LR];
#pragma unroll
        for (int r = 0; r < LR; ++r) x[r] = a.z[(int64_t)min(r0 + r, a.rows - 1) * HID + lane];
        int agent = r0 % a.n_agents;
#pragma unroll
        for (int r = 0; r < LR; ++r) {
            const LnRow o = ln_row(x[r] + addend[agent * HID + lane], a.layernorm != 0, a.ln_eps, g, be);
            if (r0 + r < a.rows) a.out[(int64_t)(r0 + r) * HID + lane] = fmaxf(o.y, 0.0f);
            agent = agent + 1 == a.n_agents ? 0 : agent + 1;
        }
    }
}

__global__ __launch_bounds__(64 * LW) void lnrelu_bwd_kernel(FlexLnReluArgs a) {
    __shared__ float addend[FLEXNET_MAX_AGENTS * HID];
    __shared__ float fold[LW][LN_PITCH];
    stage_addend(a, addend);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float g = a.layernorm ? a.ln_w[lane] : 1.0f, be = a.layernorm ? a.ln_b[lane] : 0.0f;
    float acc_g = 0.0f, acc_b = 0.0f, acc_bias = 0.0f, acc_id[FLEXNET_MAX_AGENTS];
#pragma unroll
    for (int k = 0; k < FLEXNET_MAX_AGENTS; ++k) acc_id[k] = 0.0f;
    const int n_tiles = (a.rows + LR - 1) / LR;
    for (int tile = blockIdx.x * LW + wave; tile < n_tiles; tile += gridDim.x * LW) {
        const int r0 = tile * LR;
        float x[LR], d[LR];
#pragma unroll
        for (int r = 0; r < LR; ++r) {
            const int64_t off = (int64_t)min(r0 + r, a.rows - 1) * HID + lane;
            x[r] = a.z[off];
            d[r] = r0 + r < a.rows ? a.dout[off] : 0.0f;       // spare rows of the last tile contribute nothing
        }
        int agent = r0 % a.n_agents;
#pragma unroll
        for (int r = 0; r < LR; ++r) {
            const LnRow o = ln_row(x[r] + addend[agent * HID + lane], a.layernorm != 0, a.ln_eps, g, be);
            const float dy = o.y > 0.0f ? d[r] : 0.0f;
            float dz = dy;
            if (a.layernorm) {
                acc_g = fmaf(dy, o.xhat, acc_g);
                acc_b += dy;
                const float dxh = dy * g;
                const float m1 = flex_wave_sum(dxh) * (1.0f / HID);
                const float m2 = flex_wave_sum(dxh * o.xhat) * (1.0f / HID);
                dz = o.rstd * (dxh - m1 - o.xhat * m2);
            }
            if (r0 + r < a.rows) a.dz[(int64_t)(r0 + r) * HID + lane] = dz;
            acc_bias += dz;
#pragma unroll
            for (int k = 0; k < FLEXNET_MAX_AGENTS; ++k) acc_id[k] += agent == k ? dz : 0.0f;
            agent = agent + 1 == a.n_agents ? 0 : agent + 1;
        }
    }
    float* mine = fold[wave];
    mine[0 * HID + lane] = acc_g;
    mine[1 * HID + lane] = acc_b;
    mine[2 * HID + lane] = acc_bias;
#pragma unroll
    for (int k = 0; k < FLEXNET_MAX_AGENTS; ++k) mine[(3 + k) * HID + lane] = acc_id[k];
    __syncthreads();
    float* out = a.workspace + (int64_t)blockIdx.x * LN_PITCH;
    for (int e = threadIdx.x; e < LN_PITCH; e += 64 * LW) {
        float s = fold[0][e];
#pragma unroll
        for (int w = 1; w < LW; ++w) s += fold[w][e];
        out[e] = s;
    }
}

// element e of every block's partial row, summed in a fixed order, stored in the caller's gradient tensor
#define LRED FLEX_RED_G
__global__ __launch_bounds__(64 * LRED) void lnrelu_reduce_kernel(FlexLnReluArgs a, int blocks) {
    const int ex = threadIdx.x & 63;
    float sum;
    if (!flex_reduce_rows(a.workspace + blockIdx.x * 64 + ex, LN_PITCH, blocks, true, sum)) return;   // LN_PITCH % 64 == 0
    const int vec = blockIdx.x;                               // one 64-element vector per block
    if (vec == 0) { if (a.layernorm && a.d_ln_w) a.d_ln_w[ex] = sum; }
    else if (vec == 1) { if (a.layernorm && a.d_ln_b) a.d_ln_b[ex] = sum; }
    else if (vec == 2) { if (a.d_bias) a.d_bias[ex] = sum; }
    else if (a.d_id && vec - 3 < a.n_agents) a.d_id[(vec - 3) * HID + ex] = sum;
}

static int lnrelu_check(const FlexLnReluArgs* a, bool backward) {
    if (!a || a->rows < 0 || !a->z || a->n_agents < 1) return FLEXNET_EINVAL;
    if (a->n_agents > FLEXNET_MAX_AGENTS) return FLEXNET_EUNSUPPORTED;
    if (a->layernorm && (!a->ln_w || !a->ln_b)) return FLEXNET_EINVAL;
    if (!backward && !a->out) return FLEXNET_EINVAL;
    if (backward && (!a->dout || !a->dz || !a->workspace || a->workspace_floats < FLEXNET_LNRELU_WS_FLOATS)) return FLEXNET_EINVAL;
    return FLEXNET_OK;
}

static int lnrelu_grid(int rows, int per_cu, int cap) {
    const int cus = flex_cu_count();
    if (cus < 1) return -1;
    const int want = (rows + LW * LR - 1) / (LW * LR);
    int blocks = want < cus * per_cu ? want : cus * per_cu;
    return blocks < cap ? blocks : cap;
}

extern "C" int flexnet_lnrelu_forward(const FlexLnReluArgs* a, void* stream) {
    const int rc = lnrelu_check(a, false);
    if (rc != FLEXNET_OK) return rc;
    if (a->rows == 0) return FLEXNET_OK;
    const int blocks = lnrelu_grid(a->rows, 8, 1 << 20);
    if (blocks < 1) return FLEXNET_EHIP;
    hipLaunchKernelGGL(lnrelu_fwd_kernel, dim3(blocks), dim3(64 * LW), 0, (hipStream_t)stream, *a);
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}

extern "C" int flexnet_lnrelu_backward(const FlexLnReluArgs* a, void* stream) {
    const int rc = lnrelu_check(a, true);
    if (rc != FLEXNET_OK) return rc;
    int blocks = 1;
    if (a->rows > 0) {
        blocks = lnrelu_grid(a->rows, 4, 1024);
        if (blocks < 1) return FLEXNET_EHIP;
    }
    hipLaunchKernelGGL(lnrelu_bwd_kernel, dim3(blocks), dim3(64 * LW), 0, (hipStream_t)stream, *a);
    hipLaunchKernelGGL(lnrelu_reduce_kernel, dim3(LN_VECS), dim3(64 * LRED), 0, (hipStream_t)stream, *a, blocks);
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}
