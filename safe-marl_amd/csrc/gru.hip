// gru.hip — pointwise backward of the GRUCell for update batches (gfx950).  Boundary: include/flexnet.h (FlexGruBwdArgs).
// Reference: madrl/agents/rnn_agent.py:30-32 under autograd (nn.GRUCell + fc2).  The forward pass is csrc/actor.hip with its
// `save_*` outputs; this kernel turns the gradient arriving at the action means into the gradients of the two gate
// pre-activations.  One lane per hidden unit, four rows per wavefront pass with every load issued before the arithmetic;
// HBM-bound: 324 floats read and 384 written per row (464 MB at 163 840 rows).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flexnet.h"
#include "flex_reduce.h"
#include "flex_launch.h"
#include "actor_r16.h"             // the 16-row register maps: f32x4, MFMA16, r16_group_sum

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


// ------------------------------------------------------------------------------------------------------------------------
// Round 3: the same gate gradients AND the first layer's backward in one pass (FlexGruBwdArgs::dz non-NULL).
// Per 16-row tile and wavefront, in the register maps of csrc/actor_r16.h (lane = row j + 16 g, register (S, r) = unit
// 16 S + 4 g + r): the gate gradients are formed where the MFMA wants its B operand, dx = d_gi @ W_ih is 192 MFMAs against
// W_ih in LDS (read as it is stored: lane (m, k) takes W_ih[gate unit][16 T + m]), and the LayerNorm / ReLU / bias / id-column
// backward of rnn_agent.py:25-29 (csrc/lnrelu.hip's arithmetic on this layout) runs on the accumulators.  d_gi / d_gh are
// still written (two weight-gradient passes read them); dx is never stored, d_gi is not read back for it, and the launches
// of the dx GEMM, the transposed copy of the id columns, lnrelu_bwd_kernel and the copy of d_id into fc1's gradient are
// gone: 193 -> ~120 us of a policy sub-update at 163 840 rows.  Parameter gradients: per-lane sums (a wavefront's tiles are a
// multiple of n_agents tiles apart, so a lane's row always belongs to the same agent), folded per block in a fixed order,
// summed over blocks by gru_fused_reduce_kernel: bit-reproducible.
// ------------------------------------------------------------------------------------------------------------------------
#define GF_W 4
#define GF_P 68
#define GF_VECS (3 + FLEXNET_MAX_AGENTS)        // d_ln_w, d_ln_b, d_bias, d_id[8]  (csrc/lnrelu.hip's partial-row layout)
#define GF_PITCH (GF_VECS * HID)
#define GF_LB 17                                // pitch of a lane's 16 sums in the fold buffer

// KMAX: the action columns the instantiation carries (4 or FLEXNET_MAX_ACT)
template <int KMAX>
__global__ __launch_bounds__(64 * GF_W, 2) void gru_backward_fused_kernel(FlexGruBwdArgs a) {
    __shared__ float wih[3 * HID * GF_P];
    __shared__ float addend[FLEXNET_MAX_AGENTS * HID];
    __shared__ __attribute__((aligned(16))) float w2s[FLEXNET_MAX_ACT * HID];
    static_assert(KMAX <= FLEXNET_MAX_ACT, "");
    __shared__ __attribute__((aligned(16))) float lnw[HID], lnb[HID];
    __shared__ float lanebuf[GF_W][64][GF_LB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int n = a.n_agents, ad = a.act_dim;
    for (int idx = tid; idx < 3 * HID * HID; idx += 64 * GF_W) wih[(idx >> 6) * GF_P + (idx & 63)] = a.w_ih[idx];
    for (int idx = tid; idx < FLEXNET_MAX_AGENTS * HID; idx += 64 * GF_W) {
        const int i = idx / HID, u = idx - i * HID;
        float v = a.fc1_b ? a.fc1_b[u] : 0.0f;
        if (a.agent_id && i < n) v += a.fc1_w[(int64_t)u * a.fc1_ld + a.obs_dim + i];
        addend[idx] = v;
    }
    for (int idx = tid; idx < FLEXNET_MAX_ACT * HID; idx += 64 * GF_W) w2s[idx] = idx < ad * HID ? a.fc2_w[idx] : 0.0f;
    if (tid < HID) { lnw[tid] = a.layernorm ? a.ln_w[tid] : 1.0f; lnb[tid] = a.layernorm ? a.ln_b[tid] : 0.0f; }
    __syncthreads();

    const int n_tiles = (a.rows + 15) / 16;
    const int waves_total = gridDim.x * GF_W;                              // 16 * waves_total is a multiple of n_agents (host)
    const int tile0 = blockIdx.x * GF_W + wave;
    const int my_agent = (16 * tile0 + j) % n;
    const float* ad_l = addend + my_agent * HID + 4 * g;
    const float* wih_l = wih + (4 * g) * GF_P + j;
    f32x4 acc_g[4], acc_b[4], acc_d[4];
#pragma unroll
    for (int S = 0; S < 4; ++S) { acc_g[S] = f32x4{0, 0, 0, 0}; acc_b[S] = f32x4{0, 0, 0, 0}; acc_d[S] = f32x4{0, 0, 0, 0}; }

    // The gate arithmetic runs in a ROW-CONTIGUOUS map — lane (q, c) = (lane >> 4, lane & 15), register i: row 4 i + q of the
    // tile, units 4 c .. 4 c + 3 — so that every load of the five saved tensors and every store of d_gi / d_gh is a
    // wavefront-wide run of whole rows (with the MFMA map — 64 bytes of each of 16 rows per instruction — the kernel ran at
    // 3.8 TB/s instead of 4.7: 157 -> 125 us, measured by swapping the addresses alone).  d_gi then goes through this
    // wavefront's [16][68] LDS tile, one gate at a time, into the B-operand map of the dx chains (a wavefront's own LDS
    // operations execute in order: no barrier); only the first layer's z1 / x / dz stay in the MFMA map.
    float* tbuf = &lanebuf[wave][0][0];                                    // (the fold buffer: not used before the end)
    static_assert(64 * GF_LB == 16 * GF_P, "the fold buffer of a wavefront is its transpose tile");
    for (int tile = tile0; tile < n_tiles; tile += waves_total) {
        const int r0 = tile * 16;
        const bool live = r0 + j < a.rows;
        const int64_t row = live ? r0 + j : a.rows - 1;
        const int64_t at = row * HID + 4 * g;
        f32x4 dgi[3][4];                                                    // [gate][i]: row 4 i + q, units 4 c ..
        {
            f32x4 R[4], Z[4], N[4], HN[4], HP[4];
            float dm[4][KMAX];
            bool lv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lv[i] = r0 + 4 * i + g < a.rows;
                const int64_t rw = lv[i] ? r0 + 4 * i + g : a.rows - 1;
                const int64_t ar = rw * HID + 4 * j;
                R[i] = *reinterpret_cast<const f32x4*>(a.r + ar);
                Z[i] = *reinterpret_cast<const f32x4*>(a.z + ar);
                N[i] = *reinterpret_cast<const f32x4*>(a.n + ar);
                HN[i] = *reinterpret_cast<const f32x4*>(a.hn + ar);
                HP[i] = *reinterpret_cast<const f32x4*>(a.h_prev + ar);
#pragma unroll
                for (int k = 0; k < KMAX; ++k) {                            // (clamped index: no branch around a load)
                    const float v = a.d_means[rw * ad + (k < ad ? k : ad - 1)];
                    dm[i][k] = (k < ad && lv[i]) ? v : 0.0f;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t rw = lv[i] ? r0 + 4 * i + g : a.rows - 1;
                // dh' = d_means @ fc2_w (+ the gradient arriving at the new hidden state), then the gate gradients
                f32x4 dh = f32x4{0, 0, 0, 0};
                if (a.d_hidden) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(a.d_hidden + rw * HID + 4 * j);
                    if (lv[i]) dh = t;
                }
#pragma unroll
                for (int k = 0; k < KMAX; ++k) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(w2s + k * HID + 4 * j);    // (zero rows past act_dim)
#pragma unroll
                    for (int e = 0; e < 4; ++e) dh[e] = fmaf(dm[i][k], w[e], dh[e]);
                }
                f32x4 dnr;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dn = dh[e] * (1.0f - Z[i][e]) * (1.0f - N[i][e] * N[i][e]);
                    const float dz = dh[e] * (HP[i][e] - N[i][e]) * Z[i][e] * (1.0f - Z[i][e]);
                    const float dr = dn * HN[i][e] * R[i][e] * (1.0f - R[i][e]);
                    dgi[0][i][e] = dr; dgi[1][i][e] = dz; dgi[2][i][e] = dn;
                    dnr[e] = dn * R[i][e];
                }
                if (lv[i]) {
                    float* gi = a.d_gi + rw * (3 * HID) + 4 * j;
                    float* gh = a.d_gh + rw * (3 * HID) + 4 * j;
                    *reinterpret_cast<f32x4*>(gi) = dgi[0][i];
                    *reinterpret_cast<f32x4*>(gi + HID) = dgi[1][i];
                    *reinterpret_cast<f32x4*>(gi + 2 * HID) = dgi[2][i];
                    *reinterpret_cast<f32x4*>(gh) = dgi[0][i];
                    *reinterpret_cast<f32x4*>(gh + HID) = dgi[1][i];
                    *reinterpret_cast<f32x4*>(gh + 2 * HID) = dnr;
                }
            }
        }
        // the first layer's saved rows (MFMA map), requested now: they land underneath the dx chains
        f32x4 Z1[4], XS[4];
#pragma unroll
        for (int S = 0; S < 4; ++S) {
            // ReLU's mask from the forward's own output when the caller has it (save_x): the gradient of the function that was
            // evaluated — a recomputed LayerNorm output within an ulp of zero can land on the other side
            Z1[S] = *reinterpret_cast<const f32x4*>(a.z1 + at + 16 * S);
            XS[S] = *reinterpret_cast<const f32x4*>((a.x ? a.x : a.z1) + at + 16 * S);
        }
        // dx = d_gi @ W_ih: four output-unit tiles; per gate the tile goes through LDS into the B-operand map (lane (j, g):
        // units 16 S + 4 g + r of row j) and 16 k-steps follow, the next step's weights requested before this step's MFMAs
        f32x4 dx[4];
#pragma unroll
        for (int T = 0; T < 4; ++T) dx[T] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int G = 0; G < 3; ++G) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(tbuf + (4 * i + g) * GF_P + 4 * j) = dgi[G][i];
            f32x4 bop[4];
#pragma unroll
            for (int S = 0; S < 4; ++S) bop[S] = *reinterpret_cast<const f32x4*>(tbuf + j * GF_P + 16 * S + 4 * g);
            float w[4], wn[4];
#pragma unroll
            for (int T = 0; T < 4; ++T) w[T] = wih_l[(64 * G) * GF_P + 16 * T];
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                const int S = st >> 2, r = st & 3;
                if (st + 1 < 16) {
                    const int S1 = (st + 1) >> 2, r1 = (st + 1) & 3;
#pragma unroll
                    for (int T = 0; T < 4; ++T) wn[T] = wih_l[(64 * G + 16 * S1 + r1) * GF_P + 16 * T];
                }
                const float b = bop[S][r];
#pragma unroll
                for (int T = 0; T < 4; ++T) dx[T] = MFMA16(w[T], b, dx[T]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int T = 0; T < 4; ++T) w[T] = wn[T];
            }
        }
        // LayerNorm / ReLU backward on the row (csrc/lnrelu.hip: lnrelu_bwd_kernel), 16 of its 64 units in this lane
        f32x4 xh[4], dzv[4];
        float rstd = 1.0f;
        {
            f32x4 v[4];
            float sum = 0.0f;
#pragma unroll
            for (int S = 0; S < 4; ++S)
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[S][r] = Z1[S][r] + ad_l[16 * S + r]; sum += v[S][r]; }
            if (a.layernorm) {
                const float mean = r16_group_sum(sum) * (1.0f / HID);
                float var = 0.0f;
#pragma unroll
                for (int S = 0; S < 4; ++S)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[S][r] -= mean; var = fmaf(v[S][r], v[S][r], var); }
                rstd = rsqrtf(r16_group_sum(var) * (1.0f / HID) + a.ln_eps);
#pragma unroll
                for (int S = 0; S < 4; ++S)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xh[S][r] = v[S][r] * rstd;
            } else {
#pragma unroll
                for (int S = 0; S < 4; ++S) xh[S] = v[S];
            }
        }
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int S = 0; S < 4; ++S) {
            const f32x4 gw = *reinterpret_cast<const f32x4*>(lnw + 16 * S + 4 * g);
            const f32x4 gb = *reinterpret_cast<const f32x4*>(lnb + 16 * S + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y = a.x ? XS[S][r] : xh[S][r] * gw[r] + gb[r];
                const float dy = (y > 0.0f && live) ? dx[S][r] : 0.0f;
                acc_g[S][r] = fmaf(dy, xh[S][r], acc_g[S][r]);
                acc_b[S][r] += dy;
                const float dxh = dy * gw[r];
                dzv[S][r] = dxh;
                s1 += dxh;
                s2 = fmaf(dxh, xh[S][r], s2);
            }
        }
        if (a.layernorm) {
            const float m1 = r16_group_sum(s1) * (1.0f / HID), m2 = r16_group_sum(s2) * (1.0f / HID);
#pragma unroll
            for (int S = 0; S < 4; ++S)
#pragma unroll
                for (int r = 0; r < 4; ++r) dzv[S][r] = rstd * (dzv[S][r] - m1 - xh[S][r] * m2);
        }
#pragma unroll
        for (int S = 0; S < 4; ++S) {
            if (live) *reinterpret_cast<f32x4*>(a.dz + at + 16 * S) = dzv[S];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc_d[S][r] += live ? dzv[S][r] : 0.0f;
        }
    }

    // block fold, fixed order: wavefronts in index order, rows in index order.  Element (S, r) of lane (j, g) is unit
    // 16 S + 4 g + r of a row of agent (16 (block * GF_W + w) + j) % n.
    float* out = a.workspace + (int64_t)blockIdx.x * GF_PITCH;
    for (int q = 0; q < 3; ++q) {
        const f32x4* src = q == 0 ? acc_g : q == 1 ? acc_b : acc_d;
#pragma unroll
        for (int S = 0; S < 4; ++S)
#pragma unroll
            for (int r = 0; r < 4; ++r) lanebuf[wave][lane][4 * S + r] = src[S][r];
        __syncthreads();
        const int u = tid & 63, part = tid >> 6;
        const int ug = (u >> 2) & 3, ue = 4 * (u >> 4) + (u & 3);
        if (q < 2) {
            if (part == 0) {
                float t = 0.0f;
                for (int w = 0; w < GF_W; ++w)
                    for (int jj = 0; jj < 16; ++jj) t += lanebuf[w][16 * ug + jj][ue];
                out[q * HID + u] = t;
            }
        } else {
            // d_id[k][u] for k = part, part + 4; d_bias[u] (all rows) by the first 64 threads as well
            for (int k = part; k < FLEXNET_MAX_AGENTS; k += GF_W) {
                float t = 0.0f;
                if (k < n) {
                    for (int w = 0; w < GF_W; ++w)
                        for (int jj = 0; jj < 16; ++jj)
                            if ((16 * (blockIdx.x * GF_W + w) + jj) % n == k) t += lanebuf[w][16 * ug + jj][ue];
                }
                out[(3 + k) * HID + u] = t;
            }
            if (part == 0) {
                float t = 0.0f;
                for (int w = 0; w < GF_W; ++w)
                    for (int jj = 0; jj < 16; ++jj) t += lanebuf[w][16 * ug + jj][ue];
                out[2 * HID + u] = t;
            }
        }
        __syncthreads();
    }
}

// element e of every block's partial row, summed in a fixed order, stored in the caller's gradient tensors
__global__ __launch_bounds__(64 * FLEX_RED_G) void gru_fused_reduce_kernel(FlexGruBwdArgs a, int blocks) {
    const int ex = threadIdx.x & 63;
    float sum;
    if (!flex_reduce_rows(a.workspace + blockIdx.x * 64 + ex, GF_PITCH, blocks, true, sum)) return;
    const int vec = blockIdx.x;
    if (vec == 0) { if (a.layernorm && a.d_ln_w) a.d_ln_w[ex] = sum; }
    else if (vec == 1) { if (a.layernorm && a.d_ln_b) a.d_ln_b[ex] = sum; }
    else if (vec == 2) { if (a.d_fc1_b) a.d_fc1_b[ex] = sum; }
    else if (a.d_id && vec - 3 < a.n_agents) a.d_id[(int64_t)(vec - 3) * a.d_id_agent_stride + (int64_t)ex * a.d_id_unit_stride] = sum;
}

static int gru_backward_fused(const FlexGruBwdArgs* a, hipStream_t s) {
    if (!a->w_ih || !a->z1 || !a->workspace || a->n_agents < 1 || (a->layernorm && (!a->ln_w || !a->ln_b)) ||
        (a->agent_id && (!a->fc1_w || a->fc1_ld < a->obs_dim + a->n_agents || a->obs_dim < 0)))
        return FLEXNET_EINVAL;
    if (a->n_agents > FLEXNET_MAX_AGENTS) return FLEXNET_EUNSUPPORTED;
    if (a->d_id && (a->d_id_agent_stride == 0 || a->d_id_unit_stride == 0)) return FLEXNET_EINVAL;
    const int cus = flex_cu_count();
    if (cus < 1) return FLEXNET_EHIP;
    const int n = a->n_agents;
    const int tiles = (a->rows + 15) / 16;
    int blocks = (tiles + GF_W - 1) / GF_W;
    if (blocks > 2 * cus) blocks = 2 * cus;
    blocks = blocks >= n ? blocks / n * n : n;            // a wavefront's tiles: a multiple of n_agents tiles apart
    if ((int64_t)blocks * GF_PITCH > a->workspace_floats) return FLEXNET_EINVAL;
    if (a->act_dim <= 4) hipLaunchKernelGGL(gru_backward_fused_kernel<4>, dim3(blocks), dim3(64 * GF_W), 0, s, *a);
    else hipLaunchKernelGGL(gru_backward_fused_kernel<FLEXNET_MAX_ACT>, dim3(blocks), dim3(64 * GF_W), 0, s, *a);
    hipLaunchKernelGGL(gru_fused_reduce_kernel, dim3(GF_VECS), dim3(64 * FLEX_RED_G), 0, s, *a, blocks);
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}

extern "C" int flexnet_gru_backward(const FlexGruBwdArgs* a, void* stream) {
    if (!a || a->rows < 0) return FLEXNET_EINVAL;
    if (a->rows == 0) return FLEXNET_OK;
    if (!a->d_means || !a->fc2_w || !a->r || !a->z || !a->n || !a->hn || !a->h_prev || !a->d_gi || !a->d_gh || a->act_dim < 1)
        return FLEXNET_EINVAL;
    if (a->act_dim > FLEXNET_MAX_ACT) return FLEXNET_EUNSUPPORTED;
    if (a->dz) return gru_backward_fused(a, (hipStream_t)stream);
    const int64_t passes = ((int64_t)a->rows + GRU_ROWS - 1) / GRU_ROWS;
    int64_t blocks = (passes + GRU_THREADS / 64 - 1) / (GRU_THREADS / 64);
    if (blocks > 8192) blocks = 8192;                                         // grid-stride beyond 32 rows per CU slot
    hipLaunchKernelGGL(gru_backward_kernel, dim3((unsigned)blocks), dim3(GRU_THREADS), 0, (hipStream_t)stream, *a);
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}
