// linear.hip — the shared part of the centralised critic's first layer in ONE launch (gfx950).  Boundary: include/flexnet.h
// (flexnet_linear2).  madrl/critics/mlp_critic.py:25-26 on the input madrl/models/maddpg.py:33-54 assembles:
//     out[b, 0:64] = bias + x1[b, 0:k1] W[:, c1:c1+k1]^T + x2[b, 0:k2] W[:, c2:c2+k2]^T          (exact fp32)
// with x1 every agent's stacked observation (k1 = 720), x2 every agent's action (k2 = 20), b = the update batch (32 768 to
// 135 168 rows): 2 * 64 * 740 flop per 2 960 bytes of input — 77 us of the fp32 matrix pipe against 47 us of HBM at 131 072
// rows, so the matrix pipe is the roofline and the inputs must arrive without taking issue slots or LDS bandwidth from it.
//
// WEIGHT-STATIONARY IN REGISTERS.  Round 4's kernel staged the weights through LDS in slabs and reached 0.45 of the pipe
// (two LDS operand reads per MFMA, a barrier per slab); the library's pair of GEMMs runs at ~0.65 and pays a second pass
// over `out` for the 20 action columns.  Here the 64 x 740 weights never leave the register file: a block is eight
// wavefronts on one CU (two per SIMD), wavefront v owns the input columns [32 SS v, 32 SS (v + 1))
// for ALL 64 output units — 2 x 16 x SS A-operand registers of v_mfma_f32_32x32x2_f32, loaded once per launch — and every
// layer is evaluated transposed (D[unit][row] = W X^T, as in csrc/actor.hip), so the B operand is the INPUT: lane (row r,
// half h) reads columns 8 q + 4 h .. + 3 of row r with one 16-byte global load and feeds them to MFMA steps (q, 0..3),
// whose k-pairs are (8 q + j, 8 q + 4 + j).  No LDS and no barrier on the operand path; each input byte is loaded once,
// by one wavefront, straight from HBM into the register it is consumed from.  The eight K-partial [64 x 32] tiles of a row
// tile meet in LDS (8 KB each, double-buffered: ONE barrier per 32 rows), are summed in a fixed order (bit-reproducible),
// get the bias and leave as 32-byte stores.  The reduction of tile t is issued inside tile t + 1's MFMA stream.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flexnet.h"
#include "flex_launch.h"

typedef float l2_f16 __attribute__((ext_vector_type(16)));
typedef float l2_f4 __attribute__((ext_vector_type(4)));
#define L2_WAVES 8               // wavefronts per block = K shares; two per SIMD: one wavefront's load latency is the other's MFMA time
#define L2_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x2f32((a_), (b_), (c_), 0, 0, 0)

struct Linear2K {
    FlexLinear2Args a;
    int32_t tiles;               // row tiles of 32
};

// Input pieces.  A piece is the 8 input columns [kp, kp + 8) of a row as two 16-byte loads, one per lane half (columns
// kp + 4 h ..).  k1 is a multiple of 8, so a piece lies wholly in the observation block (kp < k1), in the action block, or past
// the last column — a WAVEFRONT-UNIFORM property (kp depends on the wavefront and the step only): the source pointer is
// chosen by scalar compares, nothing per piece lives in a vector register across the loop.  Action pieces clamp their column
// to the block's last piece and pieces past the end re-read the row's first columns: both meet zero weights, every address
// is valid in every lane, every load unconditional (a conditional load is a branch with a full wait behind it), and nothing
// is done to the loaded value before the MFMAs read it, so no wait sits behind the request.
struct L2Rows { const float *obs, *obs_k, *act; };          // row base, row base + this lane's first column, action row base
__device__ __forceinline__ l2_f4 l2_load(const Linear2K& p, const L2Rows& r, int kp, int rel, int hf) {
    const int k1 = p.a.k1, kt = p.a.k1 + p.a.k2;
    const float* src;
    if (kp < k1) src = r.obs_k + rel;                                            // (uniform selects, no branch)
    else if (kp < kt) src = r.act + min(kp - k1 + 4 * hf, p.a.k2 - 4);
    else src = r.obs;
    return *reinterpret_cast<const l2_f4*>(src);
}

template <int SS>
__global__ __launch_bounds__(64 * L2_WAVES, 1) void linear2_wreg_kernel(Linear2K p) {
    __shared__ float red[2][L2_WAVES][64 * 32];                 // K-partials of two row tiles: red[buf][wavefront][unit][row]
    __shared__ float sbias[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rb = lane & 31, hf = lane >> 5;
    const FlexLinear2Args& a = p.a;
    const int k1 = a.k1, kt = a.k1 + a.k2;
    const int kk = wave * (32 * SS) + 4 * hf;                   // this lane's first input column; step (s, q, j) adds 32 s + 8 q + j

    // ---- weights into registers: A operand of step (s, q, j), unit tile u: W[32 u + rb][column of input kk + 32 s + 8 q + j]
    // (four consecutive inputs = four consecutive columns of W's row — k1 is a multiple of 8, a group never straddles the
    //  blocks — fetched as ONE 16-byte load at dword alignment: W's row pitch need not be a multiple of four floats.  As 96
    //  scalar loads per lane, 64 scattered requests each, this prologue cost ~18 us per launch.)
    struct __attribute__((packed, aligned(4))) W4 { float v[4]; };
    float w[2][SS][16];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const float* wrow = a.w + (int64_t)(32 * u + rb) * a.ldw;
#pragma unroll
        for (int s = 0; s < SS; ++s)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = kk + 32 * s + 8 * q;
                const int col = k < k1 ? a.c1 + k : (k < kt ? a.c2 + (k - k1) : a.c1);
                const W4 g = *reinterpret_cast<const W4*>(wrow + col);
#pragma unroll
                for (int j = 0; j < 4; ++j) w[u][s][4 * q + j] = k < kt ? g.v[j] : 0.0f;
            }
    }

    const int G = gridDim.x;
    int t = blockIdx.x;
    if (t >= p.tiles) return;                                   // (block-uniform: every wavefront of the block leaves)
    // Input ring.  A row tile is H = 2 SS half-steps of 16 input columns (two pieces, 16 MFMAs); xb holds R of them, R | H:
    // as soon as half-step h has issued its MFMAs its registers are re-requested for half-step h + R (of this row tile, or of
    // the next one) — R - 1 half-steps ahead of use, with the block's other seven wavefronts (two per SIMD) filling what is left
    // of the latency.  (A ring of whole row tiles kept the register count at the 256 limit and spilled address registers into
    // the loop.)
    constexpr int H = 2 * SS, R = SS == 3 ? 3 : (SS == 2 ? 4 : 2);
    static_assert(H % R == 0, "the ring must close over a row tile");
    l2_f4 xb[R][2];
    const int kw = __builtin_amdgcn_readfirstlane(wave * (32 * SS));       // this wavefront's first input column (scalar)
    L2Rows rw;
    const float* const x1 = a.x1 + (a.x1_row_cell ? *a.x1_row_cell * a.ld1 : 0);      // (in-place window of a row store)
    auto set_rows = [&](int tile) {
        int64_t row = (int64_t)tile * 32 + rb;
        if (row >= a.rows) row = a.rows - 1;
        rw.obs = x1 + row * a.ld1;
        rw.obs_k = rw.obs + kk;
        rw.act = a.x2 ? a.x2 + row * a.ld2 : rw.obs;
    };
    set_rows(t);
#pragma unroll
    for (int h = 0; h < R; ++h)
#pragma unroll
        for (int e = 0; e < 2; ++e) xb[h][e] = l2_load(p, rw, kw + 16 * h + 8 * e, 16 * h + 8 * e, hf);

    // the K-partial tiles of a row tile -> out: wavefront v finishes units [8 v, 8 v + 8) of the 32 rows; lane -> row lane / 2,
    // four consecutive units (16 bytes of the output row).  The bias waits in LDS: as a register loaded up front its arrival
    // would be waited for with the vector-memory counter inside the loop, behind the requests in flight.
    if (tid < 64) sbias[tid] = a.bias[tid];
    // The reduction of the previous row tile is spread over this tile's half-steps: the eight partials of output e are
    // requested from LDS after the MFMAs of half-step e and summed after those of half-step e + 1 — their latency lies under
    // matrix work of the same wavefront instead of idling the pipe of a SIMD whose two wavefronts reach the barrier together.
    const int srow = lane >> 1, ub = 8 * wave + 4 * (lane & 1);
    float part[L2_WAVES], o[4] = {0.0f, 0.0f, 0.0f, 0.0f};

    int it = 0, prev_tile = -1;
    for (; t < p.tiles; t += G, ++it) {
        l2_f16 acc0, acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
        const int tn = t + G;
        const bool more = tn < p.tiles;
        const int pbuf = (it - 1) & 1;
#pragma unroll
        for (int h = 0; h < H; ++h) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float b = xb[h % R][e][j];
                    acc0 = L2_MFMA(w[0][h >> 1][8 * (h & 1) + 4 * e + j], b, acc0);
                    acc1 = L2_MFMA(w[1][h >> 1][8 * (h & 1) + 4 * e + j], b, acc1);
                }
            __builtin_amdgcn_sched_barrier(0);
            const int hn = h + R;                               // the half-step these registers hold next
            if (hn < H) {
#pragma unroll
                for (int e = 0; e < 2; ++e) xb[h % R][e] = l2_load(p, rw, kw + 16 * hn + 8 * e, 16 * hn + 8 * e, hf);
            } else {
                if (hn == H && more) set_rows(tn);              // (the current row tile has no request left to make)
                if (more) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) xb[h % R][e] = l2_load(p, rw, kw + 16 * (hn - H) + 8 * e, 16 * (hn - H) + 8 * e, hf);
                }
            }
            __builtin_amdgcn_sched_barrier(0);                  // the requests stay HERE: not sunk to their uses
            if (prev_tile >= 0) {
                if (H >= 5) {
                    if (h == 0) __syncthreads();
                    if (h >= 1 && h <= 4) {                     // sum what half-step h - 1 asked for
                        float sum = part[0];
#pragma unroll
                        for (int v = 1; v < L2_WAVES; ++v) sum += part[v];
                        o[h - 1] = sum + sbias[ub + h - 1];
                    }
                    if (h <= 3) {
#pragma unroll
                        for (int v = 0; v < L2_WAVES; ++v) part[v] = red[pbuf][v][(ub + h) * 32 + srow];
                    }
                    if (h == 4) {
                        const int64_t row = (int64_t)prev_tile * 32 + srow;
                        if (row < a.rows)
                            __builtin_nontemporal_store(l2_f4{o[0], o[1], o[2], o[3]}, reinterpret_cast<l2_f4*>(a.out + row * 64 + ub));
                    }
                } else if (h == 0) {                            // short row tiles (SS < 3): the reduction in one piece
                    __syncthreads();
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float sum = red[pbuf][0][(ub + e) * 32 + srow];
#pragma unroll
                        for (int v = 1; v < L2_WAVES; ++v) sum += red[pbuf][v][(ub + e) * 32 + srow];
                        o[e] = sum + sbias[ub + e];
                    }
                    const int64_t row = (int64_t)prev_tile * 32 + srow;
                    if (row < a.rows)
                        __builtin_nontemporal_store(l2_f4{o[0], o[1], o[2], o[3]}, reinterpret_cast<l2_f4*>(a.out + row * 64 + ub));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // accumulator register i of lane (rb, hf) is D[unit 8 (i / 4) + 4 hf + i % 4][row rb] (csrc/actor.hip)
        float* dst = red[it & 1][wave];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int unit = 8 * (i >> 2) + 4 * hf + (i & 3);
            dst[unit * 32 + rb] = acc0[i];
            dst[(32 + unit) * 32 + rb] = acc1[i];
        }
        prev_tile = t;
    }
    __syncthreads();
    {
        const int pbuf = (it - 1) & 1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float sum = red[pbuf][0][(ub + e) * 32 + srow];
#pragma unroll
            for (int v = 1; v < L2_WAVES; ++v) sum += red[pbuf][v][(ub + e) * 32 + srow];
            o[e] = sum + sbias[ub + e];
        }
        const int64_t row = (int64_t)prev_tile * 32 + srow;
        if (row < a.rows)
            __builtin_nontemporal_store(l2_f4{o[0], o[1], o[2], o[3]}, reinterpret_cast<l2_f4*>(a.out + row * 64 + ub));
    }
}

extern "C" int flexnet_linear2(const FlexLinear2Args* a, void* stream) {
    if (!a || a->rows < 1 || !a->x1 || !a->w || !a->bias || !a->out || a->k1 < 4 || a->k2 < 0 || (a->k2 > 0 && !a->x2))
        return FLEXNET_EINVAL;
    if (a->ld1 < a->k1 || (a->k2 > 0 && a->ld2 < a->k2) || a->c1 < 0 || a->c2 < 0 || a->c1 + a->k1 > a->ldw ||
        a->c2 + a->k2 > a->ldw)
        return FLEXNET_EINVAL;
    // 16-byte input pieces: block widths and row pitches in whole pieces, bases aligned
    if ((a->k1 & 7) || (a->k2 & 3) || (a->ld1 & 3) || (a->k2 > 0 && (a->ld2 & 3)) || ((uintptr_t)a->x1 & 15) ||
        (a->k2 > 0 && ((uintptr_t)a->x2 & 15)) || ((uintptr_t)a->out & 15))
        return FLEXNET_EUNSUPPORTED;
    const int steps = (a->k1 + a->k2 + 31) / 32;                 // super-steps of 32 input columns
    int ss = (steps + L2_WAVES - 1) / L2_WAVES;
    const int cus = flex_cu_count();
    if (cus < 1) return FLEXNET_EHIP;
    Linear2K p;
    p.a = *a;
    const int64_t tiles = (a->rows + 31) / 32;
    if (tiles > 0x7fffffff) return FLEXNET_EUNSUPPORTED;
    p.tiles = (int32_t)tiles;
    // one block per CU (64 KB of LDS, one wavefront per SIMD); the last round of row tiles is spread evenly
    const int64_t rounds = (tiles + cus - 1) / cus;
    const int grid = (int)((tiles + rounds - 1) / rounds);
    hipStream_t s = (hipStream_t)stream;
    switch (ss) {
        case 1: hipLaunchKernelGGL(linear2_wreg_kernel<1>, dim3(grid), dim3(64 * L2_WAVES), 0, s, p); break;
        case 2: hipLaunchKernelGGL(linear2_wreg_kernel<2>, dim3(grid), dim3(64 * L2_WAVES), 0, s, p); break;
        case 3: hipLaunchKernelGGL(linear2_wreg_kernel<3>, dim3(grid), dim3(64 * L2_WAVES), 0, s, p); break;
        default: return FLEXNET_EUNSUPPORTED;                  // more than 768 input columns: the weights no longer fit the registers of eight wavefronts
    }
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}
