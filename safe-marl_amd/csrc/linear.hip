// linear.hip — the centralised critic's first layer on the matrix cores (gfx950), in exact fp32.
// Boundary: include/flexnet.h (FlexLinear2Args, flexnet_linear2).  Reference: madrl/critics/mlp_critic.py:25-26 (fc1) on the
// input madrl/models/maddpg.py:33-54 assembles — every agent's observation and action, the same for the n rows of a sample
// but for the id column — so the shared part of fc1's output is ONE skinny product per sample:
//     out[b, :] = bias + x1[b, :] W[:, c1 : c1 + k1]^T + x2[b, :] W[:, c2 : c2 + k2]^T        (x1 = observations, x2 = actions)
// Rounds 1-3 ran it as two library GEMMs (rocBLAS MT64x128x32, ~100 TFLOP/s: 30.3 + 9.0 us at 32 768 samples, the second one
// re-reading and re-writing the [b, 64] result).  Here: one launch, v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain per
// output element), evaluated transposed like the policy kernels (csrc/actor_r16.h): D[unit][row] = W X^T with the weights as
// the A operand from LDS ([k][unit], pitch 68) and a lane's 16-byte loads of its row as B operands of four k-steps — MFMA
// step (q, r) contracts over columns {16 q + 4 g + r : g = 0..3}, the weights are read in the matching order.  The weight
// matrix (64 x 740 fp32 = 189 KB) does not fit a CU's LDS: it is staged in chunks of 96 k-columns (26 KB), double-buffered,
// chunk c + 1 (and the wavefront's own rows of it) requested before the MFMAs of chunk c and written behind them — the first
// version requested a chunk's inputs at the head of its own MFMAs and sat through the memory latency eight times: 48.4 us
// at 32 768 rows against 39.8 us for the library pair.  A wavefront carries two 16-row tiles through the K loop (every A
// operand read serves both), 8 wavefronts per block, one block per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "flexnet.h"
#include "flex_launch.h"

#define L2_WAVES 8
#define L2_P 68
#define L2_KC 96                     // k-columns per staged chunk = 6 groups of 16
#define L2_GC (L2_KC / 16)
#define L2_NT 2                      // 16-row tiles per wavefront and round
#define L2_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x4f32((a_), (b_), (c_), 0, 0, 0)
typedef float l2_f4 __attribute__((ext_vector_type(4)));

struct __attribute__((aligned(16))) Lin2Lds { float w[2][L2_KC * L2_P]; };

__global__ __launch_bounds__(64 * L2_WAVES, 1) void linear2_kernel(FlexLinear2Args a) {
    __shared__ Lin2Lds s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int kg1 = (a.k1 + 15) >> 4, kg2 = (a.k2 + 15) >> 4, KG = kg1 + kg2;
    const int n_chunks = (KG + L2_GC - 1) / L2_GC;
    const int n_tiles = (int)((a.rows + 15) / 16);
    const int waves_total = gridDim.x * L2_WAVES;
    // buffer descriptors: 16-byte loads at dword alignment, zeros for offset -1 (columns past a block's end, rows past the batch)
    const int64_t wbytes = (int64_t)FLEXNET_HID * a.ldw * 4, x1b = a.rows * (int64_t)a.ld1 * 4, x2b = a.rows * (int64_t)a.ld2 * 4;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)wbytes, 0x00027000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x1), 0,
                                                                        x1b > 0x7ffffff0ll ? 0x7ffffff0 : (int)x1b, 0x00027000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x2 ? a.x2 : a.x1), 0,
                                                                        a.x2 ? (x2b > 0x7ffffff0ll ? 0x7ffffff0 : (int)x2b) : 0, 0x00027000);
    // ---- weight staging: 4 x 4 blocks (four units x four k-columns): four 16-byte reads along k, four 16-byte LDS writes
    //      along the units of the transposed image; 384 blocks per chunk, one per thread of the first six wavefronts
    l2_f4 wv[4];
    const int se = tid, srest = se >> 6;
    const int sub = 8 * (srest & 1) + (se & 7), sk4 = 8 * (srest >> 1) + ((se >> 3) & 7);
    const bool stager = se < 16 * (L2_KC / 4);
    auto stage_load = [&](int c) {
        const int kk = c * L2_KC + 4 * sk4;                                 // first of this block's four k-columns
        int col = -1;
        if (stager) {
            if (kk < 16 * kg1) col = kk < a.k1 ? a.c1 + kk : -1;
            else { const int k2 = kk - 16 * kg1; col = k2 < a.k2 ? a.c2 + k2 : -1; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            wv[i] = __builtin_bit_cast(l2_f4, __builtin_amdgcn_raw_buffer_load_b128(
                rw, col >= 0 ? ((4 * sub + i) * a.ldw + col) * 4 : -1, 0, 0));
    };
    auto stage_store = [&](int buf) {
        if (stager) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                *reinterpret_cast<l2_f4*>(&s.w[buf][(4 * sk4 + kk) * L2_P + 4 * sub]) = l2_f4{wv[0][kk], wv[1][kk], wv[2][kk], wv[3][kk]};
        }
    };

    for (int round = 0; (int64_t)round * waves_total * L2_NT < n_tiles; ++round) {
        // this wavefront's tiles of the round: spread over all wavefronts first (a small batch uses every SIMD)
        int row[L2_NT];
        bool live[L2_NT], in[L2_NT];
        l2_f4 acc[L2_NT][4];
#pragma unroll
        for (int i = 0; i < L2_NT; ++i) {
            // pairs of tiles go to wavefront 0 of every block first, then wavefront 1, ...: at 32 768 rows four wavefronts per
            // CU (one per SIMD) carry two tiles each and the other four stay idle — every A-operand read from LDS then feeds
            // two products; with one tile on each of eight wavefronts the LDS reads (256 B per 32-cycle product and
            // wavefront, two-way bank conflicts) were the limiter: 47 us against the library's 40
            const int tile = L2_NT * ((round * L2_WAVES + wave) * (int)gridDim.x + (int)blockIdx.x) + i;
            live[i] = tile < n_tiles;                                       // wavefront-uniform
            row[i] = tile * 16 + j;
            in[i] = live[i] && row[i] < a.rows;
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                const float4 b = *reinterpret_cast<const float4*>(a.bias + 16 * T + 4 * g);
                acc[i][T] = l2_f4{b.x, b.y, b.z, b.w};
            }
        }
        // a lane's 16-byte pieces of its rows for chunk c (column group q = c * 6 + qq of the [x1 | x2] space)
        auto load_x = [&](int c, l2_f4 (&xq)[L2_NT][L2_GC]) {
#pragma unroll
            for (int i = 0; i < L2_NT; ++i) {
#pragma unroll
                for (int qq = 0; qq < L2_GC; ++qq) {
                    const int q = c * L2_GC + qq, col = (q < kg1 ? 16 * q : 16 * (q - kg1)) + 4 * g;
                    if (q < kg1) xq[i][qq] = __builtin_bit_cast(l2_f4, __builtin_amdgcn_raw_buffer_load_b128(
                        r1, in[i] && col < a.k1 ? (row[i] * a.ld1 + col) * 4 : -1, 0, 0));
                    else xq[i][qq] = __builtin_bit_cast(l2_f4, __builtin_amdgcn_raw_buffer_load_b128(
                        r2, in[i] && q < KG && col < a.k2 ? (row[i] * a.ld2 + col) * 4 : -1, 0, 0));
                }
            }
        };
        auto multiply = [&](int buf, const l2_f4 (&xq)[L2_NT][L2_GC]) {
            const float* w_l = s.w[buf] + (4 * g) * L2_P + j;
            float w[4], wn[4];
#pragma unroll
            for (int T = 0; T < 4; ++T) w[T] = w_l[16 * T];
#pragma unroll
            for (int st = 0; st < 4 * L2_GC; ++st) {
                if (st + 1 < 4 * L2_GC) {
#pragma unroll
                    for (int T = 0; T < 4; ++T) wn[T] = w_l[(16 * ((st + 1) >> 2) + ((st + 1) & 3)) * L2_P + 16 * T];
                }
#pragma unroll
                for (int i = 0; i < L2_NT; ++i) {
                    if (live[i]) {
                        const float b = xq[i][st >> 2][st & 3];
#pragma unroll
                        for (int T = 0; T < 4; ++T) acc[i][T] = L2_MFMA(w[T], b, acc[i][T]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int T = 0; T < 4; ++T) w[T] = wn[T];
            }
        };
        l2_f4 xa[L2_NT][L2_GC], xb[L2_NT][L2_GC];
        __syncthreads();                                                    // (the previous round's last chunk has been read)
        stage_load(0);
        load_x(0, xa);
        stage_store(0);
        __syncthreads();
        for (int c = 0; c < n_chunks; c += 2) {                             // two chunks per trip: the input registers alternate
            if (c + 1 < n_chunks) { stage_load(c + 1); load_x(c + 1, xb); }
            multiply(0, xa);
            if (c + 1 < n_chunks) stage_store(1);
            __syncthreads();
            if (c + 1 < n_chunks) {
                if (c + 2 < n_chunks) { stage_load(c + 2); load_x(c + 2, xa); }
                multiply(1, xb);
                if (c + 2 < n_chunks) stage_store(0);
                __syncthreads();
            }
        }
#pragma unroll
        for (int i = 0; i < L2_NT; ++i) {
            if (in[i]) {
#pragma unroll
                for (int T = 0; T < 4; ++T)
                    *reinterpret_cast<float4*>(a.out + (int64_t)row[i] * FLEXNET_HID + 16 * T + 4 * g) =
                        make_float4(acc[i][T][0], acc[i][T][1], acc[i][T][2], acc[i][T][3]);
            }
        }
    }
}

extern "C" int flexnet_linear2(const FlexLinear2Args* a, void* stream) {
    if (!a || a->rows < 0 || !a->x1 || !a->w || !a->bias || !a->out || a->k1 < 4 || a->k2 < 0 || (a->k2 > 0 && !a->x2) ||
        a->ld1 < a->k1 || (a->k2 > 0 && a->ld2 < a->k2) || a->c1 < 0 || a->c2 < 0 || a->ldw < a->c1 + a->k1 ||
        (a->k2 > 0 && a->ldw < a->c2 + a->k2))
        return FLEXNET_EINVAL;
    if (a->rows == 0) return FLEXNET_OK;
    // 16-byte units along k; 32-bit byte offsets; 16-byte aligned bias and output rows
    if ((a->k1 & 3) || (a->k2 & 3) || a->rows * (int64_t)a->ld1 * 4 >= 0x7ffffff0ll || a->rows * (int64_t)a->ld2 * 4 >= 0x7ffffff0ll ||
        ((reinterpret_cast<uintptr_t>(a->bias) | reinterpret_cast<uintptr_t>(a->out)) & 15) ||
        ((reinterpret_cast<uintptr_t>(a->x1) | reinterpret_cast<uintptr_t>(a->x2) | reinterpret_cast<uintptr_t>(a->w)) & 3))
        return FLEXNET_EUNSUPPORTED;
    const int cus = flex_cu_count();
    if (cus < 1) return FLEXNET_EHIP;
    const int64_t tiles = (a->rows + 15) / 16;
    const int64_t want = (tiles + L2_WAVES - 1) / L2_WAVES;
    const int blocks = (int)(want < cus ? want : cus);
    hipLaunchKernelGGL(linear2_kernel, dim3(blocks), dim3(64 * L2_WAVES), 0, (hipStream_t)stream, *a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fprintf(stderr, "[flexnet] linear2 launch failed: %s\n", hipGetErrorString(e));
        return FLEXNET_EHIP;
    }
    return FLEXNET_OK;
}
