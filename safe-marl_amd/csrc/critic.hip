// critic.hip — the centralised critic after its first layer, forward and backward (gfx950).  Boundary: include/flexnet.h.
//
// madrl/critics/mlp_critic.py:25-33:  x = fc1(inputs); x = LayerNorm(x); h = relu(fc2(relu(x))); v = fc3(h).
// The caller forms z1 = fc1(inputs) from column blocks (learner.MADDPG.value); everything behind it is 64 wide, so
// one lane per hidden unit again: a wavefront takes four rows at a time, reads fc2.weight from LDS (transposed for
// z2 = W2 a1, as stored for da1 = W2^T dz2; odd row pitch, conflict-free both ways), and exchanges the rows'
// activations through a per-wavefront LDS staging area read back as broadcasts.  LayerNorm statistics are wavefront reductions.  The backward kernel
// recomputes the forward from z1 instead of reading saved activations (z1 is the only [rows, 64] tensor either pass
// reads), accumulates dW2 as 64 register accumulators per lane, folds the block's wavefronts in LDS and adds to the
// caller's gradient buffers with one atomic per element and block.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "flexnet.h"
#include "flex_reduce.h"
#include "flex_td.h"
#include "flex_launch.h"
#include "critic_finish.h"

#define HID FLEXNET_HID
#define CRT 4                      // rows per wavefront tile
#define CW 4                       // wavefronts per block


struct CriticRow { float xhat, rstd, y, a1; };

// fc1's output of row r for this lane's unit: stored, or composed from the per-sample part and the agent's id column
__device__ __forceinline__ float critic_z1(const FlexCriticTailArgs& a, const float* idt, int r, int lane) {
    if (a.z1) return a.z1[(int64_t)r * HID + lane];
    const int b = r / a.n_agents, i = r - b * a.n_agents;
    return a.z_shared[(int64_t)b * HID + lane] + idt[i * HID + lane];
}

// the id-column table [n_agents, 64] of the composed input, staged in LDS once per block from wherever it lives (dense, or
// the id columns of fc1.weight through the two strides); the caller's next __syncthreads() publishes it
#define CRITIC_IDT_FLOATS (FLEXNET_MAX_AGENTS * HID)
__device__ __forceinline__ void critic_stage_ids(const FlexCriticTailArgs& a, float* idt, int tid, int threads) {
    if (a.z1) return;
    const int sa = a.z_id_agent_stride ? a.z_id_agent_stride : HID, su = a.z_id_agent_stride ? a.z_id_unit_stride : 1;
    for (int e = tid; e < a.n_agents * HID; e += threads) {
        const int i = e / HID, u = e - i * HID;
        idt[e] = a.z_id[(int64_t)i * sa + (int64_t)u * su];
    }
}

// LayerNorm + ReLU of one row held one unit per lane (mlp_critic.py:27-29)
__device__ __forceinline__ CriticRow critic_ln_relu(float z1, bool layernorm, float eps, float g, float b) {
    CriticRow o;
    if (layernorm) {
        const float mean = flex_wave_sum(z1) * (1.0f / HID);
        const float d = z1 - mean;
        const float var = flex_wave_sum(d * d) * (1.0f / HID);
        o.rstd = rsqrtf(var + eps);
        o.xhat = d * o.rstd;
        o.y = o.xhat * g + b;
    } else {
        o.rstd = 1.0f; o.xhat = z1; o.y = z1;
    }
    o.a1 = fmaxf(o.y, 0.0f);
    return o;
}

// ---- forward: q = fc3(relu(fc2(relu(LayerNorm(z1)))))  (mlp_critic.py:27-31) ---------------------------------
__global__ __launch_bounds__(64 * CW, 2) void critic_tail_fwd_kernel(FlexCriticTailArgs a) {
    __shared__ float stage[CW][HID * CRT];               // per wavefront: a1 as [i][row]
    __shared__ float w2t[HID * (HID + 1)];               // w2t[i][j] = W2[j][i]: lane j reads its row of W2 along i
    __shared__ float idt[CRITIC_IDT_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid; idx < HID * HID; idx += 64 * CW) {
        const int j = idx / HID, i = idx - j * HID;
        w2t[i * (HID + 1) + j] = a.fc2_w[idx];
    }
    critic_stage_ids(a, idt, tid, 64 * CW);
    __syncthreads();
    float* sa = stage[wave];
    const float g = a.layernorm ? a.ln_w[lane] : 1.0f, be = a.layernorm ? a.ln_b[lane] : 0.0f;
    const float b2 = a.fc2_b[lane], w3 = a.fc3_w[lane], b3 = a.fc3_b[0];
    const int n_tiles = (a.rows + CRT - 1) / CRT;
    for (int tile = blockIdx.x * CW + wave; tile < n_tiles; tile += gridDim.x * CW) {
        const int r0 = tile * CRT;
        float a1v[CRT];
#pragma unroll
        for (int r = 0; r < CRT; ++r) {
            const int rr = min(r0 + r, a.rows - 1);
            a1v[r] = critic_ln_relu(critic_z1(a, idt, rr, lane), a.layernorm != 0, a.ln_eps, g, be).a1;
        }
        *reinterpret_cast<float4*>(sa + lane * CRT) = make_float4(a1v[0], a1v[1], a1v[2], a1v[3]);
        __builtin_amdgcn_wave_barrier();
        float z2[CRT] = {b2, b2, b2, b2};
#pragma unroll 8
        for (int i = 0; i < HID; ++i) {
            const float w = w2t[i * (HID + 1) + lane];
            const float4 v = *reinterpret_cast<const float4*>(sa + i * CRT);
            z2[0] = fmaf(w, v.x, z2[0]); z2[1] = fmaf(w, v.y, z2[1]);
            z2[2] = fmaf(w, v.z, z2[2]); z2[3] = fmaf(w, v.w, z2[3]);
        }
#pragma unroll
        for (int r = 0; r < CRT; ++r) {
            const float qv = flex_wave_sum(fmaxf(z2[r], 0.0f) * w3) + b3;
            if (lane == 0 && r0 + r < a.rows) a.q[r0 + r] = qv;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- backward ------------------------------------------------------------------------------------------------
// A block takes CW * CRT = 16 rows at a time.  Each wavefront recomputes the forward of its four rows, forms dz2,
// da1 = W2^T dz2, the ReLU / LayerNorm backward and writes dz1; the rows' a1 and dz2 sit in block-shared staging, and
// after a barrier wavefront w accumulates ITS 16 columns of dW2 (dW2[j][i] += sum_r dz2[r][j] a1[r][i], i in
// [16 w, 16 w + 16)) over all 16 rows — 16 register accumulators per lane instead of 64, no fold across wavefronts.
#define BT (CW * CRT)
// (CRITIC_WS_PITCH: floats per block in the workspace, 4353 used — csrc/critic_finish.h)
// PGRAD = false: dz1 only (the policy loss differentiates THROUGH the critic; its parameters take no step there).
template <bool PGRAD>
__global__ __launch_bounds__(64 * CW, 2) void critic_tail_bwd_kernel(FlexCriticTailArgs a) {
    __shared__ float sa[HID * BT];                       // a1  as [i][row of the block tile]
    __shared__ float sd[HID * BT];                       // dz2 as [j][row]
    __shared__ float w2t[HID * (HID + 1)];               // w2t[i][j] = W2[j][i]
    __shared__ float w2n[HID * (HID + 1)];               // w2n[j][i] = W2[j][i]: lane i reads its column along j
    __shared__ float idt[CRITIC_IDT_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid; idx < HID * HID; idx += 64 * CW) {
        const int j = idx / HID, i = idx - j * HID;
        const float w = a.fc2_w[idx];
        w2t[i * (HID + 1) + j] = w;
        w2n[j * (HID + 1) + i] = w;
    }
    critic_stage_ids(a, idt, tid, 64 * CW);
    __syncthreads();
    const float g = a.layernorm ? a.ln_w[lane] : 1.0f, be = a.layernorm ? a.ln_b[lane] : 0.0f;
    const float b2 = a.fc2_b[lane], w3 = a.fc3_w[lane];
    float acc_w2[BT];
#pragma unroll
    for (int c = 0; c < BT; ++c) acc_w2[c] = 0.0f;
    float acc_g = 0.0f, acc_b = 0.0f, acc_b2 = 0.0f, acc_w3 = 0.0f, acc_b3 = 0.0f;
    const int n_tiles = (a.rows + BT - 1) / BT;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {          // block-uniform trip count
        const int r0 = tile * BT + wave * CRT;
        CriticRow row[CRT];
        float a1v[CRT];
#pragma unroll
        for (int r = 0; r < CRT; ++r) {
            const int rr = min(r0 + r, a.rows - 1);
            row[r] = critic_ln_relu(critic_z1(a, idt, rr, lane), a.layernorm != 0, a.ln_eps, g, be);
            a1v[r] = row[r].a1;
        }
        float* my_a = sa + wave * CRT;                   // this wavefront's four columns of the [i][16] staging
        float* my_d = sd + wave * CRT;
        *reinterpret_cast<float4*>(my_a + lane * BT) = make_float4(a1v[0], a1v[1], a1v[2], a1v[3]);
        __builtin_amdgcn_wave_barrier();
        float z2[CRT] = {b2, b2, b2, b2};
#pragma unroll 8
        for (int i = 0; i < HID; ++i) {
            const float w = w2t[i * (HID + 1) + lane];
            const float4 v = *reinterpret_cast<const float4*>(my_a + i * BT);
            z2[0] = fmaf(w, v.x, z2[0]); z2[1] = fmaf(w, v.y, z2[1]);
            z2[2] = fmaf(w, v.z, z2[2]); z2[3] = fmaf(w, v.w, z2[3]);
        }
        float dz2[CRT];
#pragma unroll
        for (int r = 0; r < CRT; ++r) {
            const float dq = r0 + r < a.rows ? a.dq[r0 + r] : 0.0f;              // spare rows of the last tile contribute nothing
            dz2[r] = z2[r] > 0.0f ? dq * w3 : 0.0f;
            if (PGRAD) {
                acc_w3 = fmaf(dq, fmaxf(z2[r], 0.0f), acc_w3);
                acc_b3 += dq;
                acc_b2 += dz2[r];
            }
        }
        *reinterpret_cast<float4*>(my_d + lane * BT) = make_float4(dz2[0], dz2[1], dz2[2], dz2[3]);
        __builtin_amdgcn_wave_barrier();
        float da1[CRT] = {0.0f, 0.0f, 0.0f, 0.0f};      // da1_i = sum_j W2[j][i] dz2_j, this lane is unit i
#pragma unroll 8
        for (int j = 0; j < HID; ++j) {
            const float wc = w2n[j * (HID + 1) + lane];
            const float4 dv = *reinterpret_cast<const float4*>(my_d + j * BT);
            da1[0] = fmaf(wc, dv.x, da1[0]); da1[1] = fmaf(wc, dv.y, da1[1]);
            da1[2] = fmaf(wc, dv.z, da1[2]); da1[3] = fmaf(wc, dv.w, da1[3]);
        }
#pragma unroll
        for (int r = 0; r < CRT; ++r) {                  // ReLU and LayerNorm backward
            const float dy = row[r].y > 0.0f ? da1[r] : 0.0f;
            float dz1 = dy;
            if (a.layernorm) {
                if (PGRAD) {
                    acc_g = fmaf(dy, row[r].xhat, acc_g);
                    acc_b += dy;
                }
                const float dxh = dy * g;
                const float m1 = flex_wave_sum(dxh) * (1.0f / HID);
                const float m2 = flex_wave_sum(dxh * row[r].xhat) * (1.0f / HID);
                dz1 = row[r].rstd * (dxh - m1 - row[r].xhat * m2);
            }
            if (r0 + r < a.rows) a.dz1[(int64_t)(r0 + r) * HID + lane] = dz1;
        }
        if (!PGRAD) {                                    // staging is per wavefront on this path: no block barrier
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        __syncthreads();                                 // all 16 rows' a1 and dz2 are staged
        float dzr[BT];                                   // dz2 of THIS unit for the 16 rows
#pragma unroll
        for (int q = 0; q < BT / 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(sd + lane * BT + 4 * q);
            dzr[4 * q] = t.x; dzr[4 * q + 1] = t.y; dzr[4 * q + 2] = t.z; dzr[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int c = 0; c < BT; ++c) {
            const float* col = sa + (wave * BT + c) * BT;                       // a1 of unit i = 16 w + c, 16 rows (broadcast)
            float sum = acc_w2[c];
#pragma unroll
            for (int q = 0; q < BT / 4; ++q) {
                const float4 t = *reinterpret_cast<const float4*>(col + 4 * q);
                sum = fmaf(dzr[4 * q], t.x, fmaf(dzr[4 * q + 1], t.y, fmaf(dzr[4 * q + 2], t.z, fmaf(dzr[4 * q + 3], t.w, sum))));
            }
            acc_w2[c] = sum;
        }
        __syncthreads();                                 // staging may be overwritten
    }
    if (!PGRAD) return;
    // the small vectors: fold the block's four wavefronts through LDS (staging is free now)
    __syncthreads();
    float* vs = sa;                                      // [4 vectors + 1][CW][64]
    vs[(0 * CW + wave) * HID + lane] = acc_b2;
    vs[(1 * CW + wave) * HID + lane] = acc_w3;
    vs[(2 * CW + wave) * HID + lane] = acc_g;
    vs[(3 * CW + wave) * HID + lane] = acc_b;
    if (lane == 0) vs[4 * CW * HID + wave] = acc_b3;
    __syncthreads();
    float vec = 0.0f, b3sum = 0.0f;                       // wavefront w finishes vector w
#pragma unroll
    for (int k = 0; k < CW; ++k) { vec += vs[(wave * CW + k) * HID + lane]; b3sum += vs[4 * CW * HID + k]; }
    if (a.workspace) {
        // deterministic path: this block's partial sums as one row [dW2 4096 | db2 | dw3 | dg | db | db3], reduced over
        // blocks in a fixed order by critic_reduce_kernel
        float* out = a.workspace + (int64_t)blockIdx.x * CRITIC_WS_PITCH;
#pragma unroll
        for (int c = 0; c < BT; ++c) out[lane * HID + wave * BT + c] = acc_w2[c];
        out[HID * HID + wave * HID + lane] = vec;
        if (tid == 0) out[HID * HID + 4 * HID] = b3sum;
    } else {
        // one atomic per element and block (wavefront w owns columns 16 w .. 16 w + 15 of dW2)
#pragma unroll
        for (int c = 0; c < BT; ++c) unsafeAtomicAdd(&a.d_fc2_w[lane * HID + wave * BT + c], acc_w2[c]);
        float* dst = wave == 0 ? a.d_fc2_b : wave == 1 ? a.d_fc3_w : wave == 2 ? a.d_ln_w : a.d_ln_b;
        if (wave < 2 || a.layernorm) unsafeAtomicAdd(&dst[lane], vec);
        if (tid == 0) unsafeAtomicAdd(&a.d_fc3_b[0], b3sum);
    }
}

// second stage of the deterministic path (critic_reduce: csrc/critic_finish.h)
#define RED_G FLEX_RED_G
__global__ __launch_bounds__(64 * RED_G) void critic_reduce_kernel(FlexCriticTailArgs a, int blocks) {
    critic_reduce(a, blocks, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------
// Matrix-core forms of the forward pass and of the dz1-only backward (variant 0).  As in csrc/actor.hip every layer is
// evaluated transposed with v_mfma_f32_32x32x2_f32 (exact fp32): A = weights (rows = output units), B = activations
// (columns = the 32 batch rows of a wavefront's tile).  Lane (row rb, half hf) holds units 8q + 4hf + j of its row in
// accumulator register 4q + j — which is what the next product's B operand wants when MFMA step (q, j) takes the
// k-pair (8q + j, 8q + 4 + j) — so z1 -> LayerNorm/ReLU -> fc2 -> ReLU -> fc3 and back (dz2 -> W2^T dz2 -> ReLU and
// LayerNorm backward -> dz1) chain through registers; z1 is read and dz1 written as float4s of that unit pattern.
// LayerNorm sums and the fc3 dot product are in-lane sums over 32 units plus one exchange between the halves.
// 64 MFMAs per 32 rows forward, 128 backward.  (The backward WITH parameter gradients stays on the VALU kernel above:
// dW2 contracts over rows, i.e. needs the other operand layout.)
// ---------------------------------------------------------------------------------------------------------------
typedef float cf32x16 __attribute__((ext_vector_type(16)));
#define CMW 8                                            // wavefronts per block
#define CDU0(i) (8 * ((i) >> 2) + ((i) & 3))
#define CMFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x2f32((a_), (b_), (c_), 0, 0, 0)

// z1 of this lane's row in the accumulator layout: v[u][4q + j] = z1[row][32u + 8q + 4hf + j]
__device__ __forceinline__ void critic_load_z1(const FlexCriticTailArgs& a, const float* idt, int row, int hf, cf32x16* v) {
    const float* p0;
    const float* p1 = nullptr;
    if (a.z1) {
        p0 = a.z1 + (int64_t)row * HID + 4 * hf;
    } else {
        const int b = row / a.n_agents, i = row - b * a.n_agents;
        p0 = a.z_shared + (int64_t)b * HID + 4 * hf;
        p1 = idt + i * HID + 4 * hf;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 t = *reinterpret_cast<const float4*>(p0 + 32 * u + 8 * q);
            if (p1) {
                const float4 s = *reinterpret_cast<const float4*>(p1 + 32 * u + 8 * q);
                t.x += s.x; t.y += s.y; t.z += s.z; t.w += s.w;
            }
            v[u][4 * q] = t.x; v[u][4 * q + 1] = t.y; v[u][4 * q + 2] = t.z; v[u][4 * q + 3] = t.w;
        }
}

// LayerNorm of the row (statistics over this lane's 32 units and the other half's 32): v <- xhat, returns rstd
__device__ __forceinline__ float critic_ln_inplace(cf32x16* v, bool layernorm, float eps) {
    if (!layernorm) return 1.0f;
    float sum = 0.0f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += v[u][i];
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.0f / HID);
    float var = 0.0f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) { const float d = v[u][i] - mean; var = fmaf(d, d, var); }
    var += __shfl_xor(var, 32, 64);
    const float rstd = rsqrtf(var * (1.0f / HID) + eps);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[u][i] = (v[u][i] - mean) * rstd;
    return rstd;
}

// out[t] (32 output units 32t .. 32t+31, accumulator layout) = M[32t.., :] * in, M given as wl[k * HID + unit] with the
// lane part (4 hf rows down, rb units across) already in the pointer; weights of step s + 1 requested before step s
__device__ __forceinline__ cf32x16 critic_mfma_tile(const float* wl, int t, const cf32x16* in) {
    cf32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    float w = wl[CDU0(0) * HID + 32 * t];
#pragma unroll
    for (int st = 0; st < 32; ++st) {
        float wn = 0.0f;
        if (st + 1 < 32) wn = wl[(32 * ((st + 1) >> 4) + CDU0((st + 1) & 15)) * HID + 32 * t];
        acc = CMFMA(w, in[st >> 4][st & 15], acc);
        w = wn;
    }
    return acc;
}

template <bool BACKWARD, bool QMEAN = false>         // QMEAN (with BACKWARD): uniform dLoss/dq and the sum of q (the policy loss)
__global__ __launch_bounds__(64 * CMW, 2) void critic_tail_mfma_kernel(FlexCriticTailArgs a) {
    __shared__ float w2t[HID * HID];                     // w2t[k][j] = W2[j][k]: A operand of z2 = W2 a1
    __shared__ float w2n[BACKWARD ? HID * HID : 1];      // W2 as stored [j][i]:  A operand of da1 = W2^T dz2
    __shared__ float vec[5][HID];                        // ln_w, ln_b, b2, w3, (unused)
    __shared__ __attribute__((aligned(16))) float idt[CRITIC_IDT_FLOATS];
    __shared__ double qsw[QMEAN ? CMW : 1];
    float qsum = 0.0f;                                   // BACKWARD with q_mean_out: this lane's sum of q over its (two or three) rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rb = lane & 31, hf = lane >> 5;
    for (int idx = tid; idx < HID * HID; idx += 64 * CMW) {
        const int j = idx / HID, k = idx - j * HID;
        const float w = a.fc2_w[idx];
        w2t[k * HID + j] = w;
        if (BACKWARD) w2n[idx] = w;
    }
    critic_stage_ids(a, idt, tid, 64 * CMW);
    if (tid < HID) {
        vec[0][tid] = a.layernorm ? a.ln_w[tid] : 1.0f;
        vec[1][tid] = a.layernorm ? a.ln_b[tid] : 0.0f;
        vec[2][tid] = a.fc2_b[tid];
        vec[3][tid] = a.fc3_w[tid];
    }
    __syncthreads();
    const float b3 = a.fc3_b[0];
    const float* w2t_l = w2t + (4 * hf) * HID + rb;
    const float* w2n_l = w2n + (4 * hf) * HID + rb;
    const float* g_l = vec[0] + 4 * hf;
    const float* be_l = vec[1] + 4 * hf;
    const float* b2_l = vec[2] + 4 * hf;
    const float* w3_l = vec[3] + 4 * hf;
    const bool ln = a.layernorm != 0;
    const int n_tiles = (a.rows + 31) / 32;
    for (int tile = wave * gridDim.x + blockIdx.x; tile < n_tiles; tile += gridDim.x * CMW) {
        const int r0 = tile * 32;
        const int row = min(r0 + rb, a.rows - 1);
        const bool live = r0 + rb < a.rows;
        cf32x16 xh[2], a1[2];
        critic_load_z1(a, idt, row, hf, xh);
        const float rstd = critic_ln_inplace(xh, ln, a.ln_eps);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int cu = 32 * u + CDU0(i);
                a1[u][i] = fmaxf(ln ? fmaf(xh[u][i], g_l[cu], be_l[cu]) : xh[u][i], 0.0f);
            }
        cf32x16 z2[2];
        float part = 0.0f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            z2[t] = critic_mfma_tile(w2t_l, t, a1);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int cu = 32 * t + CDU0(i);
                z2[t][i] += b2_l[cu];
                part = fmaf(w3_l[cu], fmaxf(z2[t][i], 0.0f), part);
            }
        }
        if (!BACKWARD) {
            const float qv = part + __shfl_xor(part, 32, 64) + b3;
            if (hf == 0 && live) a.q[r0 + rb] = qv;
            continue;
        }
        float dq;
        if constexpr (QMEAN) {
            dq = live ? a.dq_value : 0.0f;
            const float qv = part + __shfl_xor(part, 32, 64) + b3;
            if (hf == 0 && live) qsum += qv;
        } else {
            dq = live ? a.dq[r0 + rb] : 0.0f;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) z2[t][i] = z2[t][i] > 0.0f ? dq * w3_l[32 * t + CDU0(i)] : 0.0f;     // dz2
        float m1 = 0.0f, m2 = 0.0f;
        cf32x16 d[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            d[u] = critic_mfma_tile(w2n_l, u, z2);                                                        // da1
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int cu = 32 * u + CDU0(i);
                const float dy = a1[u][i] > 0.0f ? d[u][i] : 0.0f;          // a1 > 0 <=> y > 0
                const float dxh = ln ? dy * g_l[cu] : dy;
                d[u][i] = dxh;
                m1 += dxh;
                m2 = fmaf(dxh, xh[u][i], m2);
            }
        }
        if (ln) {
            m1 = (m1 + __shfl_xor(m1, 32, 64)) * (1.0f / HID);
            m2 = (m2 + __shfl_xor(m2, 32, 64)) * (1.0f / HID);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) d[u][i] = rstd * (d[u][i] - m1 - xh[u][i] * m2);
        }
        if (live) {
            float* out = a.dz1 + (int64_t)(r0 + rb) * HID + 4 * hf;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(out + 32 * u + 8 * q) =
                        make_float4(d[u][4 * q], d[u][4 * q + 1], d[u][4 * q + 2], d[u][4 * q + 3]);
        }
    }
    if constexpr (QMEAN) {                               // lanes, then wavefronts in order -> the block's partial
        double v = (double)qsum;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) qsw[wave] = v;
        __syncthreads();
        if (tid == 0) {
            double t = qsw[0];
#pragma unroll
            for (int w = 1; w < CMW; ++w) t += qsw[w];
            reinterpret_cast<double*>(a.workspace)[blockIdx.x] = t;
        }
    }
}

// sum of q over the blocks' partials (lane-strided, then a fixed shuffle tree), scaled: the loss value of a mean-of-q loss
__global__ __launch_bounds__(64) void critic_qmean_finish_kernel(FlexCriticTailArgs a, int blocks) {
    const int lane = threadIdx.x;
    const double* ws = reinterpret_cast<const double*>(a.workspace);
    double t = 0.0;
    for (int b = lane; b < blocks; b += 64) t += ws[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    if (lane == 0) *a.q_mean_out = (float)(t * (double)a.q_mean_scale);
}

// ---------------------------------------------------------------------------------------------------------------
// Backward WITH parameter gradients on the matrix cores (variant 0, large batches).  Phase A per 32-row tile is the
// dz1-only chain above (forward recompute, dz2, da1 = W2^T dz2, ReLU / LayerNorm backward, dz1 stored).  Phase B:
// dW2[j][i] = sum_row dz2[row][j] a1[row][i] contracts over ROWS, the other operand layout, so the tile's a1 and dz2
// pass through a per-wavefront LDS transpose (written as float4s in the accumulator's unit pattern, read back with
// lanes = units, k = row pair) into 64 more MFMAs; db2 falls out of those A operands.  The three remaining vectors
// (d ln_w, d ln_b, d fc3_w) are sums over rows of values a lane already holds: they accumulate per lane over all of the
// wavefront's tiles and are transposed once, at the end.  ~330 registers per lane: ONE wavefront per SIMD (4 per CU),
// which the 192-MFMA chain per tile keeps busy by itself.  Per-block partial rows go to the same workspace layout and
// the same fixed-order second launch as the VALU kernel's.
// ---------------------------------------------------------------------------------------------------------------
#define CPW 4                                            // wavefronts per block
#define CTP 68                                           // transpose pitch (floats): float4-aligned rows
// TD: the kernel forms dLoss/dq itself (flexnet_critic_td_backward) — the forward recompute already holds fc2's output, so
// q = fc3(h2) is one more in-lane dot product; with the reward column's batch statistics (csrc/tdloss.hip's statistics pass),
// the bootstrap value and the done flag of the row: dq = -2 (BatchNorm(r) + gamma (1 - done) q' - q) / rows, maddpg.py:110-123.
// No forward launch of the tail, no q / dq round trip, no td_apply launch.
template <bool TD>
__global__ __launch_bounds__(64 * CPW, 1) void critic_tail_pgrad_mfma_kernel(FlexCriticTailArgs a, FlexTdLossArgs td) {
    __shared__ float w2t[HID * HID];
    __shared__ float w2n[HID * HID];
    __shared__ float vec[4][HID];
    __shared__ float td_m[TD_NA], td_sc[TD_NA], td_sh[TD_NA];
    __shared__ double td_sqw[CPW];
    __shared__ float tr[CPW][2][32 * CTP];               // per wavefront: a1 tile, dz2 tile as [row][unit]
    __shared__ __attribute__((aligned(16))) float idt[CRITIC_IDT_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rb = lane & 31, hf = lane >> 5;
    for (int idx = tid; idx < HID * HID; idx += 64 * CPW) {
        const int j = idx / HID, k = idx - j * HID;
        const float w = a.fc2_w[idx];
        w2t[k * HID + j] = w;
        w2n[idx] = w;
    }
    critic_stage_ids(a, idt, tid, 64 * CPW);
    if (TD && tid < TD_NA) td_column_affine(td, tid, td_m[tid], td_sc[tid], td_sh[tid]);
    if (tid < HID) {
        vec[0][tid] = a.layernorm ? a.ln_w[tid] : 1.0f;
        vec[1][tid] = a.layernorm ? a.ln_b[tid] : 0.0f;
        vec[2][tid] = a.fc2_b[tid];
        vec[3][tid] = a.fc3_w[tid];
    }
    __syncthreads();
    const float* w2t_l = w2t + (4 * hf) * HID + rb;
    const float* w2n_l = w2n + (4 * hf) * HID + rb;
    const float* g_l = vec[0] + 4 * hf;
    const float* be_l = vec[1] + 4 * hf;
    const float* b2_l = vec[2] + 4 * hf;
    const float* w3_l = vec[3] + 4 * hf;
    float* t1 = tr[wave][0];
    float* t2 = tr[wave][1];
    float* t1_w = t1 + rb * CTP + 4 * hf;                // this lane's row, its unit pattern (write side)
    float* t2_w = t2 + rb * CTP + 4 * hf;
    const float* t1_r = t1 + hf * CTP + rb;              // row pair 2s + hf, unit rb (+ 32): read side
    const float* t2_r = t2 + hf * CTP + rb;
    const bool ln = a.layernorm != 0;

    cf32x16 dW[2][2];                                    // [tj][ti]: dW2 rows 32 tj.., columns 32 ti..
    cf32x16 sg[2], sb[2], sw3[2];                        // per-lane (row) partial sums of dy xhat, dy, dq h2
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            dW[u][0][i] = 0.0f; dW[u][1][i] = 0.0f; sg[u][i] = 0.0f; sb[u][i] = 0.0f; sw3[u][i] = 0.0f;
        }
    float cs2[2] = {0.0f, 0.0f}, sb3 = 0.0f;
    double td_sq = 0.0;                                  // TD: this lane's sum of squared TD errors
    const float td_b3 = TD ? a.fc3_b[0] : 0.0f, td_inv = TD ? 1.0f / (float)a.rows : 0.0f;

    const int n_tiles = (a.rows + 31) / 32;
    for (int tile = wave * gridDim.x + blockIdx.x; tile < n_tiles; tile += gridDim.x * CPW) {
        const int r0 = tile * 32;
        const int row = min(r0 + rb, a.rows - 1);
        const bool live = r0 + rb < a.rows;
        // ---- phase A ------------------------------------------------------------------------------------------
        cf32x16 xh[2], a1[2];
        critic_load_z1(a, idt, row, hf, xh);
        const float rstd = critic_ln_inplace(xh, ln, a.ln_eps);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int cu = 32 * u + CDU0(i);
                a1[u][i] = fmaxf(ln ? fmaf(xh[u][i], g_l[cu], be_l[cu]) : xh[u][i], 0.0f);
            }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(t1_w + 32 * u + 8 * q) =
                    make_float4(a1[u][4 * q], a1[u][4 * q + 1], a1[u][4 * q + 2], a1[u][4 * q + 3]);
        float dq = 0.0f, td_r = 0.0f, td_nq = 0.0f, td_dn = 0.0f;
        int td_j = 0;
        if constexpr (TD) {                                  // the row's TD inputs, in flight under fc2
            const int tb = row / td.n_agents;
            td_j = row - tb * td.n_agents;
            td_r = td.reward[row]; td_nq = td.next_q[row]; td_dn = td.done[tb];
        } else {
            dq = live ? a.dq[r0 + rb] : 0.0f;                // spare rows of the last tile contribute nothing
        }
        cf32x16 z2[2];
        float qp = 0.0f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            z2[t] = critic_mfma_tile(w2t_l, t, a1);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int cu = 32 * t + CDU0(i);
                const float zz = z2[t][i] + b2_l[cu];
                z2[t][i] = zz;
                if constexpr (TD) qp = fmaf(w3_l[cu], fmaxf(zz, 0.0f), qp);
            }
        }
        if constexpr (TD) {
            const float q = qp + __shfl_xor(qp, 32, 64) + td_b3;
            const float rn = (td_r - td_m[td_j]) * td_sc[td_j] + td_sh[td_j];
            const float delta = rn + td.gamma * (1.0f - td_dn) * td_nq - q;
            dq = live ? -2.0f * delta * td_inv : 0.0f;
            if (live && hf == 0) {
                td_sq += (double)delta * (double)delta;
                if (td.q) const_cast<float*>(td.q)[r0 + rb] = q;          // (outputs here: the caller asked to see them)
                if (td.dq) td.dq[r0 + rb] = dq;
            }
        }
        if (hf == 0) sb3 += dq;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int cu = 32 * t + CDU0(i);
                const float zz = z2[t][i];
                sw3[t][i] = fmaf(dq, fmaxf(zz, 0.0f), sw3[t][i]);
                z2[t][i] = zz > 0.0f ? dq * w3_l[cu] : 0.0f;                                               // dz2
            }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(t2_w + 32 * t + 8 * q) =
                    make_float4(z2[t][4 * q], z2[t][4 * q + 1], z2[t][4 * q + 2], z2[t][4 * q + 3]);
        float m1 = 0.0f, m2 = 0.0f;
        cf32x16 d[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            d[u] = critic_mfma_tile(w2n_l, u, z2);                                                        // da1
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int cu = 32 * u + CDU0(i);
                const float dy = a1[u][i] > 0.0f ? d[u][i] : 0.0f;
                if (ln) {
                    sg[u][i] = fmaf(dy, xh[u][i], sg[u][i]);
                    sb[u][i] += dy;
                }
                const float dxh = ln ? dy * g_l[cu] : dy;
                d[u][i] = dxh;
                m1 += dxh;
                m2 = fmaf(dxh, xh[u][i], m2);
            }
        }
        if (ln) {
            m1 = (m1 + __shfl_xor(m1, 32, 64)) * (1.0f / HID);
            m2 = (m2 + __shfl_xor(m2, 32, 64)) * (1.0f / HID);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) d[u][i] = rstd * (d[u][i] - m1 - xh[u][i] * m2);
        }
        if (live) {
            float* out = a.dz1 + (int64_t)(r0 + rb) * HID + 4 * hf;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(out + 32 * u + 8 * q) =
                        make_float4(d[u][4 * q], d[u][4 * q + 1], d[u][4 * q + 2], d[u][4 * q + 3]);
        }
        // ---- phase B: dW2 += dz2^T a1 over the tile's 32 rows (16 row pairs), db2 from the A operands -----------
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);                                    // lgkmcnt(0): the transposes have landed
#pragma unroll
        for (int sp = 0; sp < 16; ++sp) {
            const float a0 = t2_r[(2 * sp) * CTP], a1v = t2_r[(2 * sp) * CTP + 32];
            const float b0 = t1_r[(2 * sp) * CTP], b1v = t1_r[(2 * sp) * CTP + 32];
            dW[0][0] = CMFMA(a0, b0, dW[0][0]);
            dW[0][1] = CMFMA(a0, b1v, dW[0][1]);
            dW[1][0] = CMFMA(a1v, b0, dW[1][0]);
            dW[1][1] = CMFMA(a1v, b1v, dW[1][1]);
            cs2[0] += a0; cs2[1] += a1v;
        }
        __builtin_amdgcn_wave_barrier();                                       // the next tile overwrites the transposes
    }

    // ---- the wavefront's sums -> the block's partial row [dW2 | db2 | dw3 | dg | db | db3] ---------------------------
    if constexpr (TD) {                                                        // squared TD errors: lanes, then wavefronts in order
        double v = td_sq;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) td_sqw[wave] = v;
    }
    __syncthreads();
    if constexpr (TD) {
        if (tid == 0) {
            double v = td_sqw[0];
#pragma unroll
            for (int w = 1; w < CPW; ++w) v += td_sqw[w];
            reinterpret_cast<double*>(td.workspace)[TD_WS_SQ + blockIdx.x] = v;
        }
    }
    float* fold = &tr[0][0][0];                                                // CRITIC_WS_PITCH floats, all tiles are done
    for (int w = 0; w < CPW; ++w) {
        if (wave == w) {
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int e = (32 * tj + CDU0(r) + 4 * hf) * HID + 32 * ti + rb;       // dW2[j][i]
                        fold[e] = (w == 0 ? 0.0f : fold[e]) + dW[tj][ti][r];
                    }
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                const float v = cs2[tj] + __shfl_xor(cs2[tj], 32, 64);
                if (hf == 0) { const int e = HID * HID + 32 * tj + rb; fold[e] = (w == 0 ? 0.0f : fold[e]) + v; }
            }
            const float b3 = sb3;                                              // rows live in the hf == 0 lanes
            float b3t = hf == 0 ? b3 : 0.0f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) b3t += __shfl_xor(b3t, off, 64);
            if (lane == 0) { const int e = HID * HID + 4 * HID; fold[e] = (w == 0 ? 0.0f : fold[e]) + b3t; }
        }
        __syncthreads();
    }
    // the row-lane vectors: one [32 rows][64 units] image per vector and wavefront, column sums with lanes = units
    float* img = &tr[0][0][0] + CRITIC_WS_PITCH;                                // behind the fold area
    for (int v = 0; v < 3; ++v) {                                              // 0: dw3, 1: dg, 2: db  (fold order)
        for (int w = 0; w < CPW; ++w) {
            if (wave == w) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float x = v == 0 ? sw3[u][i] : v == 1 ? sg[u][i] : sb[u][i];
                        img[rb * HID + 32 * u + CDU0(i) + 4 * hf] = x;
                    }
            }
            __syncthreads();
            if (wave == w) {
                float sum = 0.0f;
                for (int r = 0; r < 32; ++r) sum += img[r * HID + lane];
                const int e = HID * HID + (1 + v) * HID + lane;
                fold[e] = (w == 0 ? 0.0f : fold[e]) + sum;
            }
            __syncthreads();
        }
    }
    float* out = a.workspace + (int64_t)blockIdx.x * CRITIC_WS_PITCH;
    for (int e = tid; e < HID * HID + 4 * HID + 1; e += 64 * CPW) out[e] = fold[e];
}

// ---------------------------------------------------------------------------------------------------------------
// The same backward (phase A + phase B, TD or handed-in dq) on 16-ROW tiles with v_mfma_f32_16x16x4_f32, TWO wavefronts
// per SIMD (round 3).  The 32-row kernel above needs 371 registers — one wavefront per SIMD, so nobody issues VALU / LDS
// work while its MFMA chains run and nobody issues MFMAs while it does its LayerNorm / ReLU arithmetic (0.21 of the fp32
// matrix peak).  With 16x16 tiles a lane holds 16 units of ONE row (lane = row j + 16 g, register r of unit tile S =
// unit 16 S + 4 g + r: the instruction's C/D map), so every per-row vector is 16 registers instead of 32 and the row-sum
// accumulators (d ln_w, d ln_b, d fc3_w) 48 instead of 96; dW2 stays 64 (sixteen 16x16 accumulators).  Under 256
// registers: a 512-thread block per CU, two wavefronts per SIMD that cover each other's chains.  The accumulator layout
// chains into the next product exactly as with 32x32 tiles: MFMA step (S, r) contracts over units {16 S + 4 g + r}, the
// matching weights come from LDS (pitch 68: the four lane groups of an A-operand read land 16 banks apart).  Same MFMA
// time per row (192 x 32 cycles per 16 rows), same workspace row per block, same second-stage launches.
//
// SM ("sample-major", composed input with d_z_shared wanted — both callers of the update path): a tile is 16 SAMPLES of
// one agent, and a wavefront walks the n agents of its 16 samples before it moves on.  Then d_z_shared[b] = sum over the
// agents of dz1[b n + i] is an in-lane sum (16 registers, stored once per sample tile) and d_z_id[i] = sum over samples of
// dz1 is a DPP row sum per tile into a per-wavefront LDS row: dz1 itself — 42 MB at the update batch — is never written,
// and the fold launch that read it back (critic_dz_fold_kernel, 17 us) is gone; the block's id partial rows go where the
// fold kernel's went, for the same second stage.
// ---------------------------------------------------------------------------------------------------------------
typedef float cf32x4 __attribute__((ext_vector_type(4)));
#define C16W 8                                           // wavefronts per block: two per SIMD
#define C16P 68                                          // pitch (floats) of the two W2 images
#define C16Q 68                                          // pitch of the per-wavefront [row][unit] transposes: the float4 stores of
                                                         // eight consecutive rows cover all 32 banks (the reads of two row
                                                         // groups overlap in 12 of 16 banks: 32 two-way reads per tile, cheaper
                                                         // than four-way stores at pitch 80, and 12 KB less LDS)
#define C16MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x4f32((a_), (b_), (c_), 0, 0, 0)

// sum over the 16 lanes of a DPP row (the 16 rows of a tile for one lane group); valid in lane 15 of the row
__device__ __forceinline__ float critic16_row_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    return v;
}

// sum over the four lane groups holding one row (lanes j, j + 16, j + 32, j + 48), the same bits in all four:
// v_permlane16_swap / v_permlane32_swap (gfx950) exchange 16- and 32-lane halves in the VALU — no trip through the LDS
// crossbar as with ds_bpermute (five of these sit on every tile's dependent chain)
__device__ __forceinline__ float critic16_group_sum(float v) {
    const unsigned x = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);       // r[0] = rows 0,0,2,2 of x; r[1] = rows 1,1,3,3
    v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    const unsigned y = __builtin_bit_cast(unsigned, v);
    auto q = __builtin_amdgcn_permlane32_swap(y, y, false, false);       // q[0] = lower half twice; q[1] = upper half twice
    return __builtin_bit_cast(float, (unsigned)q[0]) + __builtin_bit_cast(float, (unsigned)q[1]);
}

// out[T] (T = 0..3: output units 16 T .., accumulator layout) = M * in; M as wl[k * C16P + unit] with the lane part
// (4 g rows down, j units across) already in the pointer.  Four independent accumulators per step (the instruction's
// dependent latency is 40 cycles against a 32-cycle issue), the next step's weights requested before this step's MFMAs.
__device__ __forceinline__ void critic16_layer(const float* wl, const cf32x4* in, cf32x4* out) {
#pragma unroll
    for (int t = 0; t < 4; ++t) out[t] = cf32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float w[4], wn[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) w[t] = wl[16 * t];
#pragma unroll
    for (int st = 0; st < 16; ++st) {
        if (st + 1 < 16) {
#pragma unroll
            for (int t = 0; t < 4; ++t) wn[t] = wl[(16 * ((st + 1) >> 2) + ((st + 1) & 3)) * C16P + 16 * t];
        }
        const float b = in[st >> 2][st & 3];
#pragma unroll
        for (int t = 0; t < 4; ++t) out[t] = C16MFMA(w[t], b, out[t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) w[t] = wn[t];
    }
}

// diagnostic build (-DCRITIC_STAMPS, tools/critic_stamps.py): s_memtime between the phases of block 0's first wavefront,
// summed over its tiles (never timed, never shipped; the stamps' own s_waitcnt drains the LDS queue at every boundary)
#ifdef CRITIC_STAMPS
__device__ unsigned long long critic_stamps[8 * 16];          // [wavefront of block 0][phase]
#define CSTAMP_DECL unsigned long long cs_last = __builtin_amdgcn_s_memtime(), cs_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define CSTAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); cs_acc[k] += t_ - cs_last; cs_last = t_; } while (0)
#define CSTAMP_OUT do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { for (int k_ = 0; k_ < 10; ++k_) critic_stamps[(threadIdx.x >> 6) * 16 + k_] = cs_acc[k_]; } } while (0)
extern "C" int flexnet_debug_critic_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(critic_stamps), sizeof(critic_stamps)) == hipSuccess ? 0 : -2;
}
#else
#define CSTAMP_DECL do { } while (0)
#define CSTAMP(k) do { } while (0)
#define CSTAMP_OUT do { } while (0)
#endif

#define C16_IDW (FLEXNET_MAX_AGENTS * HID)               // a wavefront's id-column sums [agent][unit] (SM)

template <bool TD, bool SM, bool LN>
__global__ __launch_bounds__(64 * C16W) void critic_tail_pgrad16_kernel(FlexCriticTailArgs a, FlexTdLossArgs td, int64_t dz_off) {
    __shared__ float w2t[HID * C16P];                    // w2t[k][j] = W2[j][k]: A operand of z2 = W2 a1
    __shared__ float w2n[HID * C16P];                    // W2 as stored [j][i]:  A operand of da1 = W2^T dz2
    __shared__ __attribute__((aligned(16))) float vec[4][HID];
    __shared__ float td_m[TD_NA], td_sc[TD_NA], td_sh[TD_NA];
    __shared__ double td_part[8][2 * TD_NA];
    __shared__ double td_sqw[C16W];
    // one pool: per wavefront the a1 tile and the dz2 tile as [row][unit] (pitch C16Q), then (SM) per wavefront the sample
    // tile's shared rows as loaded [16][64]; at the end of the kernel the pool is the fold area (four partial rows)
    constexpr int TR_FLOATS = C16W * 2 * 16 * C16Q, XS_FLOATS = SM ? C16W * 16 * HID : 0;
    constexpr int POOL_FLOATS = TR_FLOATS + XS_FLOATS > 4 * CRITIC_WS_PITCH ? TR_FLOATS + XS_FLOATS : 4 * CRITIC_WS_PITCH;
    __shared__ __attribute__((aligned(16))) float pool[POOL_FLOATS];
    __shared__ __attribute__((aligned(16))) float idt[CRITIC_IDT_FLOATS];
    __shared__ float idacc[SM ? C16W : 1][C16_IDW];      // SM: per-wavefront sums over samples of dz1, per agent
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    CSTAMP_DECL;
    for (int idx = tid; idx < HID * HID; idx += 64 * C16W) {
        const int r = idx / HID, c = idx - r * HID;
        const float w = a.fc2_w[idx];
        w2t[c * C16P + r] = w;
        w2n[r * C16P + c] = w;
    }
    critic_stage_ids(a, idt, tid, 64 * C16W);
    if constexpr (SM) {
        for (int e = tid; e < C16W * C16_IDW; e += 64 * C16W) (&idacc[0][0])[e] = 0.0f;
    }
    if constexpr (TD) {
        // the reward columns' batch statistics from the statistics pass's 64 per-block partial sums: 128 threads take 8
        // blocks each (loads in flight together), 16 threads finish — td_column_stats' loop, run by one thread per column,
        // was 128 dependent-latency loads in the prologue of every block
        if (td.normalise && tid < 128) {
            const int c = tid & 15, part = tid >> 4;
            const double* ws = reinterpret_cast<const double*>(td.workspace);
            double v = 0.0;
#pragma unroll
            for (int b = 0; b < TD_BLOCKS / 8; ++b) v += ws[(part * (TD_BLOCKS / 8) + b) * 2 * TD_NA + c];
            td_part[part][c] = v;
        }
    }
    if (tid < HID) {
        vec[0][tid] = a.layernorm ? a.ln_w[tid] : 1.0f;
        vec[1][tid] = a.layernorm ? a.ln_b[tid] : 0.0f;
        vec[2][tid] = a.fc2_b[tid];
        vec[3][tid] = a.fc3_w[tid];
    }
    __syncthreads();
    if constexpr (TD) {
        if (tid < TD_NA) {
            float m = 0.0f, sc = 1.0f, sh = 0.0f;
            if (td.normalise && tid < td.n_agents) {
                double su = 0.0, ss = 0.0;
#pragma unroll
                for (int p8 = 0; p8 < 8; ++p8) { su += td_part[p8][tid]; ss += td_part[p8][TD_NA + tid]; }
                const double nr = (double)(td.stat_rows > 0 ? td.stat_rows : (int64_t)td.rows);
                const double mean = su / nr;
                double var = ss / nr - mean * mean;                                  // biased (what the normalisation uses)
                if (var < 0.0) var = 0.0;
                m = (float)mean;
                sc = (float)(1.0 / sqrt(var + (double)td.bn_eps)) * (td.bn_weight ? td.bn_weight[tid] : 1.0f);
                sh = td.bn_bias ? td.bn_bias[tid] : 0.0f;
            }
            td_m[tid] = m; td_sc[tid] = sc; td_sh[tid] = sh;
        }
        __syncthreads();
    }
    const float* w2t_l = w2t + (4 * g) * C16P + j;
    const float* w2n_l = w2n + (4 * g) * C16P + j;
    float* t1 = pool + (2 * wave) * 16 * C16Q;
    float* t2 = pool + (2 * wave + 1) * 16 * C16Q;
    float* t1_w = t1 + j * C16Q + 4 * g;                 // this lane's row, its unit pattern (write side)
    float* t2_w = t2 + j * C16Q + 4 * g;
    const float* t1_r = t1 + g * C16Q + j;               // row 4 s + g, unit 16 T + j: read side
    const float* t2_r = t2 + g * C16Q + j;
    constexpr bool ln = LN;
    const bool composed = SM || !a.z1;

    cf32x4 dW[4][4];                                     // [tj][ti]: dW2 rows 16 tj + 4 g + r, column 16 ti + j
    cf32x4 sg[4], sb[4], sw3[4];                         // per-lane (row) partial sums of dy xhat, dy, dq h2
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int v = 0; v < 4; ++v) dW[u][v] = cf32x4{0.0f, 0.0f, 0.0f, 0.0f};
        sg[u] = cf32x4{0.0f, 0.0f, 0.0f, 0.0f}; sb[u] = sg[u]; sw3[u] = sg[u];
    }
    float cs2[4] = {0.0f, 0.0f, 0.0f, 0.0f}, sb3 = 0.0f;
    double td_sq = 0.0;
    const float td_b3 = a.fc3_b[0], td_inv = TD ? 1.0f / (float)a.rows : 0.0f;

    // Work items.  Plain: 16 consecutive rows per tile.  SM: sample tile `ot` (16 samples) x agent `ia`, agents innermost.
    const int n_ag = SM ? a.n_agents : 1;
    const int units = SM ? a.rows / a.n_agents : a.rows;                       // samples (SM) or rows
    const int n_ot = (units + 15) / 16;
    const int ot_step = gridDim.x * C16W;
    int ot = wave * gridDim.x + blockIdx.x, ia = 0;
    // the raw first-layer row of a work item (z1 row, or the sample's shared part): requested one item ahead, into the
    // registers the previous item's xhat has just vacated
    cf32x4 xn[4];
    auto request = [&](int o, cf32x4* dst) {
        const int u = min(16 * o + j, units - 1);
        const float* p0;
        if (SM) p0 = a.z_shared + (int64_t)u * HID + 4 * g;
        else if (a.z1) p0 = a.z1 + (int64_t)u * HID + 4 * g;
        else p0 = a.z_shared + (int64_t)(u / a.n_agents) * HID + 4 * g;
#pragma unroll
        for (int S = 0; S < 4; ++S) {
            const float4 t = *reinterpret_cast<const float4*>(p0 + 16 * S);
            dst[S] = cf32x4{t.x, t.y, t.z, t.w};
        }
    };
    if (!SM && ot < n_ot) request(ot, xn);
    cf32x4 dzs[SM ? 4 : 1];                              // SM: sum over the agents of dz1, this lane's sample
    float* xs_l = pool + TR_FLOATS + (SM ? wave * 16 * HID + j * HID + 4 * g : 0);   // this lane's 4 x 16 bytes of its sample's row
    // static priority for the second-dispatched half: at equal priority the older wavefront of a SIMD wins every issue
    // arbitration, finishes first and leaves the younger one to run its tail alone (67.7 -> 66.5 us for the TD backward's
    // launches at the update batch; alternating the priority per tile measured the same.  Round 3, on the sample-major
    // kernel: no priority 67.1, this 68.2, the other half raised 67.3 us — inside the noise; per-wavefront stamps show the
    // first-dispatched four waiting ~15 k of 123 k cycles at the final barrier either way: tools/critic_stamps.py)
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    CSTAMP(0);                                                                 // staging, barriers
    while (ot < n_ot) {
        // the four parameter vectors are read from LDS where they are used: 64 loop-invariant reads hoisted out of this
        // loop (which the compiler does when it can prove the offset invariant) are 64 registers this kernel does not have
        int voff = 4 * g;
        asm volatile("" : "+v"(voff));
        const float* g_l = vec[0] + voff;                // unit 16 S + 4 g + r at [16 S + r]
        const float* be_l = vec[1] + voff;
        const float* b2_l = vec[2] + voff;
        const float* w3_l = vec[3] + voff;
        const int u0 = 16 * ot;
        const int unit = min(u0 + j, units - 1);                               // sample (SM) or row
        const bool live = u0 + j < units;
        const int row = SM ? unit * n_ag + ia : unit;
        // ---- phase A ------------------------------------------------------------------------------------------
        cf32x4 xh[4], a1[4];
        {
            const float* p1 = idt + 4 * g;
            if (SM) p1 += ia * HID;
            else if (composed) p1 += (row - (row / a.n_agents) * a.n_agents) * HID;
            if constexpr (SM) {
                // the sample's shared row: from HBM for the first agent (kept in LDS as loaded), from LDS for the others —
                // no second trip to memory, and no registers holding it across the tile
                if (ia == 0) {
                    request(ot, xn);
#pragma unroll
                    for (int S = 0; S < 4; ++S)
                        *reinterpret_cast<float4*>(xs_l + 16 * S) = make_float4(xn[S][0], xn[S][1], xn[S][2], xn[S][3]);
                } else {
#pragma unroll
                    for (int S = 0; S < 4; ++S) {
                        const float4 t = *reinterpret_cast<const float4*>(xs_l + 16 * S);
                        xn[S] = cf32x4{t.x, t.y, t.z, t.w};
                    }
                }
            }
#pragma unroll
            for (int S = 0; S < 4; ++S) xh[S] = xn[S];
            if (composed) {
#pragma unroll
                for (int S = 0; S < 4; ++S) {
                    const float4 s4 = *reinterpret_cast<const float4*>(p1 + 16 * S);
                    xh[S][0] += s4.x; xh[S][1] += s4.y; xh[S][2] += s4.z; xh[S][3] += s4.w;
                }
            }
        }
        if (SM && ia == 0) {
#pragma unroll
            for (int S = 0; S < 4; ++S) dzs[S] = cf32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        float rstd = 1.0f;
        if (ln) {
            float sum = 0.0f;
#pragma unroll
            for (int S = 0; S < 4; ++S)
#pragma unroll
                for (int r = 0; r < 4; ++r) sum += xh[S][r];
            const float mean = critic16_group_sum(sum) * (1.0f / HID);
            float var = 0.0f;
#pragma unroll
            for (int S = 0; S < 4; ++S)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = xh[S][r] - mean; var = fmaf(d, d, var); }
            rstd = rsqrtf(critic16_group_sum(var) * (1.0f / HID) + a.ln_eps);
#pragma unroll
            for (int S = 0; S < 4; ++S)
#pragma unroll
                for (int r = 0; r < 4; ++r) xh[S][r] = (xh[S][r] - mean) * rstd;
        }
        unsigned amask = 0;                                                    // bit 4 S + r: a1 > 0 (<=> y > 0)
#pragma unroll
        for (int S = 0; S < 4; ++S) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a1[S][r] = fmaxf(ln ? fmaf(xh[S][r], g_l[16 * S + r], be_l[16 * S + r]) : xh[S][r], 0.0f);
                amask |= (a1[S][r] > 0.0f ? 1u : 0u) << (4 * S + r);
            }
            *reinterpret_cast<float4*>(t1_w + 16 * S) = make_float4(a1[S][0], a1[S][1], a1[S][2], a1[S][3]);
        }
        float dq = 0.0f, td_r = 0.0f, td_nq = 0.0f, td_dn = 0.0f;
        int td_j = 0;
        if constexpr (TD) {                                  // the row's TD inputs, in flight under fc2
            const int tb = SM ? unit : row / td.n_agents;
            td_j = SM ? ia : row - tb * td.n_agents;
            td_r = td.reward[row]; td_nq = td.next_q[row]; td_dn = td.done[tb];
        } else {
            dq = live ? a.dq[row] : 0.0f;                    // spare rows of the last tile contribute nothing
        }
        CSTAMP(1);                                                             // LayerNorm, a1, transposed store
        cf32x4 z2[4];
        critic16_layer(w2t_l, a1, z2);
        CSTAMP(2);                                                             // fc2 forward chain
        float qp = 0.0f;
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float zz = z2[T][r] + b2_l[16 * T + r];
                z2[T][r] = zz;
                if constexpr (TD) qp = fmaf(w3_l[16 * T + r], fmaxf(zz, 0.0f), qp);
            }
        if constexpr (TD) {
            const float q = critic16_group_sum(qp) + td_b3;
            const float rn = (td_r - td_m[td_j]) * td_sc[td_j] + td_sh[td_j];
            const float delta = rn + td.gamma * (1.0f - td_dn) * td_nq - q;
            dq = live ? -2.0f * delta * td_inv : 0.0f;
            if (live && g == 0) {
                td_sq += (double)delta * (double)delta;
                if (td.q) const_cast<float*>(td.q)[row] = q;             // (outputs here: the caller asked to see them)
                if (td.dq) td.dq[row] = dq;
            }
        }
        if (g == 0) sb3 += dq;
#pragma unroll
        for (int T = 0; T < 4; ++T) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float zz = z2[T][r];
                sw3[T][r] = fmaf(dq, fmaxf(zz, 0.0f), sw3[T][r]);
                z2[T][r] = zz > 0.0f ? dq * w3_l[16 * T + r] : 0.0f;                                       // dz2
            }
            *reinterpret_cast<float4*>(t2_w + 16 * T) = make_float4(z2[T][0], z2[T][1], z2[T][2], z2[T][3]);
        }
        CSTAMP(3);                                                             // bias, q, TD error, dz2, transposed store
        cf32x4 d[4];
        critic16_layer(w2n_l, z2, d);                                                                     // da1
        CSTAMP(4);                                                             // da1 chain
        float m1 = 0.0f, m2 = 0.0f;
#pragma unroll
        for (int U = 0; U < 4; ++U)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dy = (amask >> (4 * U + r)) & 1u ? d[U][r] : 0.0f;
                if (ln) {
                    sg[U][r] = fmaf(dy, xh[U][r], sg[U][r]);
                    sb[U][r] += dy;
                }
                const float dxh = ln ? dy * g_l[16 * U + r] : dy;
                d[U][r] = dxh;
                m1 += dxh;
                m2 = fmaf(dxh, xh[U][r], m2);
            }
        if (ln) {
            m1 = critic16_group_sum(m1) * (1.0f / HID);
            m2 = critic16_group_sum(m2) * (1.0f / HID);
#pragma unroll
            for (int U = 0; U < 4; ++U)
#pragma unroll
                for (int r = 0; r < 4; ++r) d[U][r] = rstd * (d[U][r] - m1 - xh[U][r] * m2);
        }
        // the next work item's raw row is requested here: xhat is dead, phase B covers the latency
        int ot_n = ot, ia_n = ia + 1;
        if (ia_n == n_ag) { ia_n = 0; ot_n = ot + ot_step; }
        if (!SM && ot_n < n_ot) request(ot_n, xn);
        if constexpr (SM) {
#pragma unroll
            for (int U = 0; U < 4; ++U)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = live ? d[U][r] : 0.0f;                     // (spare samples of the last tile: nothing)
                    dzs[U][r] += v;
                    const float cs = critic16_row_sum(v);                      // over the tile's 16 samples
                    if (j == 15) atomicAdd(&idacc[wave][ia * HID + 16 * U + 4 * g + r], cs);   // this wavefront's row only:
                }                                                              // in-order ds_add_f32, nobody to race with
            if (ia == n_ag - 1 && live) {
                float* out = a.d_z_shared + (int64_t)unit * HID + 4 * g;
#pragma unroll
                for (int U = 0; U < 4; ++U)
                    *reinterpret_cast<float4*>(out + 16 * U) = make_float4(dzs[U][0], dzs[U][1], dzs[U][2], dzs[U][3]);
            }
        } else {
            if (live) {
                float* out = a.dz1 + (int64_t)row * HID + 4 * g;
#pragma unroll
                for (int U = 0; U < 4; ++U)
                    *reinterpret_cast<float4*>(out + 16 * U) = make_float4(d[U][0], d[U][1], d[U][2], d[U][3]);
            }
        }
        CSTAMP(5);                                                             // ReLU / LayerNorm backward, dz1 / d_z_shared
        // ---- phase B: dW2 += dz2^T a1 over the tile's 16 rows (4 row groups), db2 from the A operands ----------------
        // (a wavefront's own LDS writes and reads execute in order: no wait between the transposed stores and these reads)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            float av[4], bv[4];
#pragma unroll
            for (int T = 0; T < 4; ++T) { av[T] = t2_r[4 * s4 * C16Q + 16 * T]; bv[T] = t1_r[4 * s4 * C16Q + 16 * T]; }
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) {
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) dW[tj][ti] = C16MFMA(av[tj], bv[ti], dW[tj][ti]);
                cs2[tj] += av[tj];
            }
        }
        __builtin_amdgcn_wave_barrier();                                       // (compiler: the next item's stores stay below)
        CSTAMP(6);                                                             // phase B
        ot = ot_n; ia = ia_n;
    }

    // ---- the wavefronts' sums -> the block's partial row [dW2 | db2 | dw3 | dg | db | db3] ---------------------------
    if constexpr (TD) {                                                        // squared TD errors: lanes, then wavefronts in order
        double v = td_sq;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) td_sqw[wave] = v;
    }
    // row-lane vectors: over the 16 rows of the lane group (DPP), valid in lane j == 15
#pragma unroll
    for (int S = 0; S < 4; ++S)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sw3[S][r] = critic16_row_sum(sw3[S][r]);
            sg[S][r] = critic16_row_sum(sg[S][r]);
            sb[S][r] = critic16_row_sum(sb[S][r]);
        }
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) cs2[tj] = critic16_group_sum(cs2[tj]);      // db2[16 tj + j], every group holds it
    float b3t = g == 0 ? sb3 : 0.0f;                                           // rows live in the g == 0 lanes
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) b3t += __shfl_xor(b3t, off, 64);
    CSTAMP(7);                                                                 // per-wavefront sums after the last tile
    __syncthreads();                                                           // all tiles done: tr becomes the fold area
    CSTAMP(8);                                                                 // ... waiting for the block's other wavefronts
    if constexpr (TD) {
        if (tid == 0) {
            double v = td_sqw[0];
#pragma unroll
            for (int w = 1; w < C16W; ++w) v += td_sqw[w];
            reinterpret_cast<double*>(td.workspace)[TD_WS_SQ + blockIdx.x] = v;
        }
    }
    if constexpr (SM) {                                                        // the block's id-column partial row, wavefronts in order
        float* dzo = a.workspace + dz_off + (int64_t)blockIdx.x * C16_IDW;
        for (int e = tid; e < C16_IDW; e += 64 * C16W) {
            float t = idacc[0][e];
#pragma unroll
            for (int w = 1; w < C16W; ++w) t += idacc[w][e];
            dzo[e] = t;
        }
    }
    // wavefronts 0-3 store their partial row into area w, wavefronts 4-7 add theirs to area w - 4, then the four areas are
    // summed in index order: a fixed order whatever the timing
    float* area = pool + (wave & 3) * CRITIC_WS_PITCH;
    for (int half = 0; half < 2; ++half) {
        if ((wave >> 2) == half) {
            const bool first = half == 0;
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int e = (16 * tj + 4 * g + r) * HID + 16 * ti + j;                     // dW2[j][i]
                        area[e] = (first ? 0.0f : area[e]) + dW[tj][ti][r];
                    }
            if (g == 0) {
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) {
                    const int e = HID * HID + 16 * tj + j;
                    area[e] = (first ? 0.0f : area[e]) + cs2[tj];
                }
            }
            if (j == 15) {
#pragma unroll
                for (int S = 0; S < 4; ++S)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int u = 16 * S + 4 * g + r;
                        float* p = area + HID * HID + HID + u;
                        p[0] = (first ? 0.0f : p[0]) + sw3[S][r];
                        p[HID] = (first ? 0.0f : p[HID]) + sg[S][r];
                        p[2 * HID] = (first ? 0.0f : p[2 * HID]) + sb[S][r];
                    }
            }
            if (lane == 0) { const int e = HID * HID + 4 * HID; area[e] = (first ? 0.0f : area[e]) + b3t; }
        }
        __syncthreads();
    }
    const float* a0 = pool;
    float* out = a.workspace + (int64_t)blockIdx.x * CRITIC_WS_PITCH;
    for (int e = tid; e < HID * HID + 4 * HID + 1; e += 64 * C16W)
        out[e] = ((a0[e] + a0[CRITIC_WS_PITCH + e]) + a0[2 * CRITIC_WS_PITCH + e]) + a0[3 * CRITIC_WS_PITCH + e];
    CSTAMP(9);                                                                 // end-of-kernel fold
    CSTAMP_OUT;
}

template <bool TD>
static void critic_launch_pgrad16(int nb, bool sm, const FlexCriticTailArgs& a, const FlexTdLossArgs& t, int64_t dz_off, hipStream_t s) {
    const dim3 grid(nb), block(64 * C16W);
    if (sm) {
        if (a.layernorm) hipLaunchKernelGGL((critic_tail_pgrad16_kernel<TD, true, true>), grid, block, 0, s, a, t, dz_off);
        else hipLaunchKernelGGL((critic_tail_pgrad16_kernel<TD, true, false>), grid, block, 0, s, a, t, dz_off);
    } else {
        if (a.layernorm) hipLaunchKernelGGL((critic_tail_pgrad16_kernel<TD, false, true>), grid, block, 0, s, a, t, dz_off);
        else hipLaunchKernelGGL((critic_tail_pgrad16_kernel<TD, false, false>), grid, block, 0, s, a, t, dz_off);
    }
}

// below this the VALU kernels (4 rows per wavefront, 8 blocks per CU) spread a batch over the chip better than
// 32-row MFMA tiles do: 8.5 vs 11.4 us forward at 20 480 rows, 43.5 vs 28.5 us at 163 840
#define CRITIC_MFMA_MIN_ROWS 65536
static int critic_mfma_grid(int rows) {
    const int cus = flex_cu_count();
    if (cus < 1) return -1;
    const int tiles = (rows + 31) / 32;
    return tiles < cus ? tiles : cus;                    // one block per CU (it stages W2 once); tiles spread over blocks first
}

// ---- composed input: dz1 folded back onto its two sources in one pass over dz1 ---------------------------------------
// d_z_shared[b] = sum_i dz1[b n + i] (stored), d_z_id[i] = sum_b dz1[b n + i] (per-block partial rows in the workspace,
// then summed over blocks in a fixed order) — instead of two library reductions that each read dz1 again.
#define DZF_W 4
static_assert(DZF_PITCH == C16_IDW, "the 16-row kernel writes the fold kernel's partial rows itself (SM)");
__global__ __launch_bounds__(64 * DZF_W) void critic_dz_fold_kernel(FlexCriticTailArgs a, int64_t ws_off) {
    __shared__ float fold[DZF_W][DZF_PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = a.n_agents, samples = a.rows / n;
    float acc[FLEXNET_MAX_AGENTS];
#pragma unroll
    for (int i = 0; i < FLEXNET_MAX_AGENTS; ++i) acc[i] = 0.0f;
    for (int b = blockIdx.x * DZF_W + wave; b < samples; b += gridDim.x * DZF_W) {
        const float* p = a.dz1 + (int64_t)b * n * HID + lane;
        float v[FLEXNET_MAX_AGENTS];
#pragma unroll
        for (int i = 0; i < FLEXNET_MAX_AGENTS; ++i) v[i] = i < n ? p[i * HID] : 0.0f;
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < FLEXNET_MAX_AGENTS; ++i) { s += v[i]; acc[i] += v[i]; }
        a.d_z_shared[(int64_t)b * HID + lane] = s;
    }
#pragma unroll
    for (int i = 0; i < FLEXNET_MAX_AGENTS; ++i) fold[wave][i * HID + lane] = acc[i];
    __syncthreads();
    float* out = a.workspace + ws_off + (int64_t)blockIdx.x * DZF_PITCH;
    for (int e = threadIdx.x; e < DZF_PITCH; e += 64 * DZF_W) {
        float t = fold[0][e];
#pragma unroll
        for (int w = 1; w < DZF_W; ++w) t += fold[w][e];
        out[e] = t;
    }
}

__global__ __launch_bounds__(64 * RED_G) void critic_dz_reduce_kernel(FlexCriticTailArgs a, int blocks) {
    critic_dz_reduce(a, a.workspace, blocks, blockIdx.x);
}

static int critic_dz_fold(const FlexCriticTailArgs& k, hipStream_t stream) {
    const int samples = k.rows / k.n_agents;
    int blocks = (samples + DZF_W - 1) / DZF_W;
    if (blocks > 256) blocks = 256;                      // one per CU: the second stage walks one partial row per block
    if ((int64_t)blocks * DZF_PITCH > k.workspace_floats) return FLEXNET_EINVAL;
    hipLaunchKernelGGL(critic_dz_fold_kernel, dim3(blocks), dim3(64 * DZF_W), 0, stream, k, (int64_t)0);
    hipLaunchKernelGGL(critic_dz_reduce_kernel, dim3(k.n_agents), dim3(64 * RED_G), 0, stream, k, blocks);
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}

// Everything that follows the two first-stage kernels of flexnet_critic_td_backward in ONE launch: the parameter gradients'
// fixed-order sums (blocks 0 .. CRITIC_RED_BLOCKS - 1), the id-column sums (one block per agent) and the loss / running
// statistics finish (last block, one wavefront).
__global__ __launch_bounds__(64 * RED_G) void critic_td_finish_kernel(CriticFinishK k) {
    critic_finish_block(k, blockIdx.x);
}

static int critic_check(const FlexCriticTailArgs* a, bool backward, bool need_dq = true) {
    if (!a || a->rows < 0) return FLEXNET_EINVAL;
    if (!a->fc2_w || !a->fc2_b || !a->fc3_w || !a->fc3_b || (a->layernorm && (!a->ln_w || !a->ln_b)))
        return FLEXNET_EINVAL;
    if (!a->z1 && (!a->z_shared || !a->z_id || a->n_agents < 1 || a->rows % a->n_agents != 0)) return FLEXNET_EINVAL;
    if (!backward && !a->q) return FLEXNET_EINVAL;
    if (backward && ((need_dq && !a->dq && !a->dq_uniform) || !a->dz1)) return FLEXNET_EINVAL;
    if (backward && ((a->d_z_shared != nullptr) != (a->d_z_id != nullptr))) return FLEXNET_EINVAL;
    if (backward && a->d_z_shared && (a->z1 || !a->workspace || a->n_agents > FLEXNET_MAX_AGENTS)) return FLEXNET_EINVAL;
    if (a->d_z_id_agent_stride < 0 || a->d_z_id_unit_stride < 0 || ((a->d_z_id_agent_stride == 0) != (a->d_z_id_unit_stride == 0)))
        return FLEXNET_EINVAL;
    if (a->z_id_agent_stride < 0 || a->z_id_unit_stride < 0 || ((a->z_id_agent_stride == 0) != (a->z_id_unit_stride == 0)))
        return FLEXNET_EINVAL;
    if (!a->z1 && a->n_agents > FLEXNET_MAX_AGENTS) return FLEXNET_EINVAL;
    // parameter gradients: all of them, or none (d_fc2_w == NULL: dz1 only)
    if (backward && a->d_fc2_w && (!a->d_fc2_b || !a->d_fc3_w || !a->d_fc3_b || (a->layernorm && (!a->d_ln_w || !a->d_ln_b))))
        return FLEXNET_EINVAL;
    return FLEXNET_OK;
}

static int critic_grid(int rows, int per_cu) {
    const int cus = flex_cu_count();
    if (cus < 1) return -1;
    const int want = (rows + CW * CRT - 1) / (CW * CRT);
    return want < cus * per_cu ? want : cus * per_cu;
}

extern "C" int flexnet_critic_tail_forward(const FlexCriticTailArgs* a, void* stream) {
    const int rc = critic_check(a, false);
    if (rc != FLEXNET_OK) return rc;
    if (a->rows == 0) return FLEXNET_OK;
    if (a->variant == 0 && a->rows >= CRITIC_MFMA_MIN_ROWS) {
        const int nb = critic_mfma_grid(a->rows);
        if (nb < 1) return FLEXNET_EHIP;
        hipLaunchKernelGGL(critic_tail_mfma_kernel<false>, dim3(nb), dim3(64 * CMW), 0, (hipStream_t)stream, *a);
        return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
    }
    const int blocks = critic_grid(a->rows, 8);
    if (blocks < 1) return FLEXNET_EHIP;
    hipLaunchKernelGGL(critic_tail_fwd_kernel, dim3(blocks), dim3(64 * CW), 0, (hipStream_t)stream, *a);
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}

static int critic_tail_backward_main(const FlexCriticTailArgs* a, void* stream);

// the matrix-core backward with parameter gradients runs on 16-row tiles in its sample-major form (it forms d_z_shared and
// the id-column partial rows itself, dz1 is not written)
static bool critic_pgrad16_sm(const FlexCriticTailArgs& k) {
    const bool two_stage = k.workspace && k.workspace_floats >= FLEXNET_CRITIC_WS_FLOATS;
    return k.d_fc2_w && two_stage && k.variant == 0 && k.rows >= CRITIC_MFMA_MIN_ROWS && k.variant_pgrad32 == 0 &&
           k.d_z_shared && !k.z1;
}

extern "C" int flexnet_critic_tail_backward(const FlexCriticTailArgs* a, void* stream) {
    const int rc = critic_check(a, true);
    if (rc != FLEXNET_OK) return rc;
    if (a->rows == 0) return FLEXNET_OK;
    const int rm = critic_tail_backward_main(a, stream);
    if (rm != FLEXNET_OK || !a->d_z_shared) return rm;
    if (critic_pgrad16_sm(*a)) {                           // the id-column partial rows are there already: second stage only
        const int nb = critic_mfma_grid(a->rows);
        FlexCriticTailArgs k = *a;
        k.workspace = a->workspace + (int64_t)nb * CRITIC_WS_PITCH;
        hipLaunchKernelGGL(critic_dz_reduce_kernel, dim3(k.n_agents), dim3(64 * RED_G), 0, (hipStream_t)stream, k, nb);
        return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
    }
    return critic_dz_fold(*a, (hipStream_t)stream);        // after the main launches: it reuses their workspace
}

// the checks and the launch geometry both halves of flexnet_critic_td_backward share (csrc/critic_finish.h)
int critic_finish_prepare(const FlexCriticTailArgs* a, const FlexTdLossArgs* t, CriticFinishK* out) {
    const int rc = critic_check(a, true, false);
    if (rc != FLEXNET_OK) return rc;
    if (!t || t->rows < 1 || t->n_agents < 1 || !t->reward || !t->done || !t->next_q || !t->loss || !t->workspace ||
        t->workspace_floats < FLEXNET_TD_WS_FLOATS || (reinterpret_cast<uintptr_t>(t->workspace) & 7) != 0)
        return FLEXNET_EINVAL;
    if ((int64_t)t->rows * t->n_agents != a->rows) return FLEXNET_EINVAL;
    if (t->n_agents > TD_NA) return FLEXNET_EUNSUPPORTED;
    const bool two_stage = a->workspace && a->workspace_floats >= FLEXNET_CRITIC_WS_FLOATS;
    if (!a->d_fc2_w || !two_stage || a->variant != 0 || a->rows < CRITIC_MFMA_MIN_ROWS) return FLEXNET_EUNSUPPORTED;
    const int nb = critic_mfma_grid(a->rows);
    if (nb < 1 || nb > 1024 || nb > TD_SQ_MAX) return FLEXNET_EHIP;
    // dz1 folded onto its sources: the partial rows of the id-column sums go behind the backward kernel's, so ONE launch
    // finishes both.  16-row kernel on a composed input (SM): it forms d_z_shared and those partial rows itself; otherwise
    // the fold kernel reads dz1 back.
    const int64_t dz_off = (int64_t)nb * CRITIC_WS_PITCH;
    const bool sm = a->variant_pgrad32 == 0 && a->d_z_shared && !a->z1;
    if (sm && (t->n_agents != a->n_agents || dz_off + (int64_t)nb * DZF_PITCH > a->workspace_floats)) return FLEXNET_EINVAL;
    int dz_blocks = 0;
    if (sm) {
        dz_blocks = nb;
    } else if (a->d_z_shared) {
        const int samples = a->rows / a->n_agents;
        dz_blocks = (samples + DZF_W - 1) / DZF_W;
        if (dz_blocks > 256) dz_blocks = 256;
        if (dz_off + (int64_t)dz_blocks * DZF_PITCH > a->workspace_floats) return FLEXNET_EINVAL;
    }
    out->a = *a; out->td = *t; out->nb = nb; out->dz_blocks = dz_blocks; out->dz_off = dz_off; out->pad = 0;
    out->blocks = CRITIC_RED_BLOCKS + (dz_blocks > 0 ? a->n_agents : 0) + 1;
    return FLEXNET_OK;
}

// The value loss and the critic's backward in one pass (maddpg.py:100-123 + mlp_critic.py:25-33): reward statistics, then
// the matrix-core backward forming q, the TD error, dLoss/dq and the loss partial sums itself, the fixed-order second
// stage, the loss / running-statistics finish, and (composed input) dz1 folded onto its sources.
// phases: 1 = (statistics pass unless stats_ready) + the backward kernel, 2 = the finish launch, 3 = both (include/flexnet.h)
extern "C" int flexnet_critic_td_backward_phases(const FlexCriticTailArgs* a, const FlexTdLossArgs* t, int32_t phases, void* stream) {
    if (phases < 1 || phases > 3) return FLEXNET_EINVAL;
    CriticFinishK k;
    const int rc = critic_finish_prepare(a, t, &k);
    if (rc != FLEXNET_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    if ((phases & 1) && t->normalise && !t->stats_ready) flex_td_launch_stats(*t, s);
    const bool sm = a->variant_pgrad32 == 0 && a->d_z_shared && !a->z1;
    if (phases & 1) {
        if (a->variant_pgrad32 == 1)  // the 32-row kernel (one wavefront per SIMD), kept as the cross-check and for A/B timing
            hipLaunchKernelGGL(critic_tail_pgrad_mfma_kernel<true>, dim3(k.nb), dim3(64 * CPW), 0, s, *a, *t);
        else
            critic_launch_pgrad16<true>(k.nb, sm, *a, *t, k.dz_off, s);
        if (!sm && a->d_z_shared)
            hipLaunchKernelGGL(critic_dz_fold_kernel, dim3(k.dz_blocks), dim3(64 * DZF_W), 0, s, *a, k.dz_off);
    }
    if (phases & 2)
        hipLaunchKernelGGL(critic_td_finish_kernel, dim3(k.blocks), dim3(64 * RED_G), 0, s, k);
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}

extern "C" int flexnet_critic_td_backward(const FlexCriticTailArgs* a, const FlexTdLossArgs* t, void* stream) {
    return flexnet_critic_td_backward_phases(a, t, 3, stream);
}

static int critic_tail_backward_main(const FlexCriticTailArgs* a, void* stream) {
    FlexCriticTailArgs k = *a;
    if (!k.d_fc2_w && k.variant == 0 && k.rows >= CRITIC_MFMA_MIN_ROWS) {
        const int nb = critic_mfma_grid(k.rows);
        if (nb < 1) return FLEXNET_EHIP;
        if (k.q_mean_out && (!k.workspace || k.workspace_floats < 2 * (int64_t)nb ||
                             (reinterpret_cast<uintptr_t>(k.workspace) & 7) != 0)) return FLEXNET_EINVAL;
        if (k.dq_uniform != (k.q_mean_out != nullptr)) return FLEXNET_EUNSUPPORTED;      // (the two come together)
        if (k.q_mean_out) {
            hipLaunchKernelGGL((critic_tail_mfma_kernel<true, true>), dim3(nb), dim3(64 * CMW), 0, (hipStream_t)stream, k);
            hipLaunchKernelGGL(critic_qmean_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, k, nb);
        } else {
            hipLaunchKernelGGL(critic_tail_mfma_kernel<true>, dim3(nb), dim3(64 * CMW), 0, (hipStream_t)stream, k);
        }
        return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
    }
    if (k.dq_uniform || k.q_mean_out) return FLEXNET_EUNSUPPORTED;       // the matrix-core dz1-only backward's extras
    if (!k.d_fc2_w) {
        const int nb = critic_grid(k.rows, 4);
        if (nb < 1) return FLEXNET_EHIP;
        hipLaunchKernelGGL(critic_tail_bwd_kernel<false>, dim3(nb), dim3(64 * CW), 0, (hipStream_t)stream, k);
        return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
    }
    const bool two_stage = k.workspace && k.workspace_floats >= FLEXNET_CRITIC_WS_FLOATS;
    if (k.overwrite_grads && !two_stage) return FLEXNET_EINVAL;
    if (two_stage && k.variant == 0 && k.rows >= CRITIC_MFMA_MIN_ROWS) {
        const int nb = critic_mfma_grid(k.rows);
        if (nb < 1 || nb > 1024) return FLEXNET_EHIP;
        const int64_t dz_off = (int64_t)nb * CRITIC_WS_PITCH;
        const bool sm = critic_pgrad16_sm(k);
        if (sm && dz_off + (int64_t)nb * DZF_PITCH > k.workspace_floats) return FLEXNET_EINVAL;
        if (k.variant_pgrad32 == 1)
            hipLaunchKernelGGL(critic_tail_pgrad_mfma_kernel<false>, dim3(nb), dim3(64 * CPW), 0, (hipStream_t)stream, k, FlexTdLossArgs{});
        else
            critic_launch_pgrad16<false>(nb, sm, k, FlexTdLossArgs{}, dz_off, (hipStream_t)stream);
        hipLaunchKernelGGL(critic_reduce_kernel, dim3((HID * HID + 4 * HID + 1 + 63) / 64), dim3(64 * RED_G), 0, (hipStream_t)stream, k, nb);
        return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
    }
    if (!two_stage) k.workspace = nullptr;
    // deterministic path: up to 1024 blocks of partial sums; atomic path: one block per CU (each ends with 4 k atomics)
    const int blocks = critic_grid(k.rows, two_stage ? 4 : 1);
    if (blocks < 1 || blocks > 1024) return FLEXNET_EHIP;
    hipLaunchKernelGGL(critic_tail_bwd_kernel<true>, dim3(blocks), dim3(64 * CW), 0, (hipStream_t)stream, k);
    if (two_stage)
        hipLaunchKernelGGL(critic_reduce_kernel, dim3((HID * HID + 4 * HID + 1 + 63) / 64), dim3(64 * RED_G), 0, (hipStream_t)stream, k, blocks);
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}
