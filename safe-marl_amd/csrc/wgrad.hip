// wgrad.hip — weight gradients of the learner's linear layers: C[m, n] = sum_k A[k, m] * B[k, n] with k = the batch
// (gfx950).  Boundary: include/flexnet.h (flexnet_wgrad).
//
// The reference's update (madrl/utils/trainer.py:62-111 -> loss.backward()) spends its GEMM time on exactly this
// shape: dW = dY^T X for fc1 / GRUCell / fc2 of rnn_agent.py:13-33 and fc1 of mlp_critic.py:5-34, where the summed
// dimension is the batch (32 768 samples, x agents = 163 840 rows) and the output is at most 192 x 745.  Library
// split-K kernels reach 10-15 TFLOP/s there.  Here both operands are read in their stored [k, .] layout straight
// into the operand registers of v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation): lane l of a
// wavefront supplies row k0 + (l >> 5) and columns MT * (l & 31) .. + MT - 1 of A — one vector load, coalesced over
// the half-wavefront — so register t of that load is the A operand of M-tile t (the tile's row index i is column
// MT * i + t: a permutation undone when the result is written).  B alike.  A wavefront owns the whole
// [32 MT, 32 NT] output block over its share of k, so every element of A and B is read once per column chunk and the
// kernel runs at the HBM / matrix-core balance point (7 loaded floats per 10 MFMAs for [64, 160]).
// Partial blocks are folded over the thread block's four wavefronts in LDS (fixed order), written as register images
// to the workspace and summed over thread blocks in a fixed order by wgrad_reduce_kernel: bit-reproducible.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flexnet.h"
#include "flex_reduce.h"
#include "critic_finish.h"

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define WG_WAVES 4
#ifndef WG_UNROLL
#define WG_UNROLL(MT, NT) ((MT) * (NT) >= 10 ? 2 : 4)   // k steps (2 rows each) whose loads are issued together
#endif
#ifndef WG_DEPTH
#define WG_DEPTH(MT, NT) 2                              // register buffers in rotation
#endif
#define WG_IMG(MT, NT) ((MT) * (NT) * 1024)

#define WG_CS_FLOATS (520 * 192)      // head of the workspace: per-block column sums of A

struct WgradK {
    const float* a;
    const float* b;
    float* c;
    float* ws;                       // register images, after the column-sum area
    float* cs;                       // column-sum partials [slabs][32 MT], or NULL
    float* colsum;
    int64_t k, lda, ldb, a_floats, b_floats;
    int32_t m, n, rows_per_block, slabs, accumulate, ldc;
    // second input block (TWO instantiations): B = [b | b2], columns n .. n + n2 - 1 of the result go to c2
    const float* b2;
    float* c2;
    int64_t ldb2, b2_floats;
    int32_t n2, ldc2;
    const int64_t* b_cell;           // B's first row inside a larger row store (device cell), or NULL
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wg_rsrc(const float* base, int64_t floats) {
    const int64_t bytes = floats > 0 ? floats * 4 : 0;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes > 0x7ffffff0ll ? 0x7ffffff0 : (int)bytes, 0x00027000);
}

// W consecutive floats at byte offset `off` (out of range: zeros)
template <int W>
__device__ __forceinline__ void wg_load(__amdgpu_buffer_rsrc_t r, int off, float* out) {
    if constexpr (W == 1) {
        out[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
    } else if constexpr (W == 2) {
        const v2f v = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
        out[0] = v.x; out[1] = v.y;
    } else if constexpr (W >= 4) {
        const v4f v = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
        if constexpr (W > 4) wg_load<W - 4>(r, off + 16, out + 4);
    } else {
        wg_load<2>(r, off, out);
        wg_load<1>(r, off + 8, out + 2);
    }
}

template <int MT, int NT, bool CS, bool TWO = false>
__global__ __launch_bounds__(64 * WG_WAVES, 2) void wgrad_kernel(WgradK p) {
    __shared__ float fold[WG_IMG(MT, NT)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t k0 = (int64_t)blockIdx.x * p.rows_per_block;
    const int64_t left = p.k - k0;
    const int rows = left < p.rows_per_block ? (int)left : p.rows_per_block;
    const int n0 = blockIdx.y * (32 * NT);
    // buffer views of this block's rows: anything past them (other blocks' rows, the end of the allocation) reads 0
    int64_t fa = (int64_t)rows * p.lda, fb = (int64_t)rows * p.ldb - n0;
    const int64_t ea = p.a_floats - k0 * p.lda, eb = p.b_floats - k0 * p.ldb - n0;
    if (ea < fa) fa = ea;
    if (eb < fb) fb = eb;
    const __amdgpu_buffer_rsrc_t ra = wg_rsrc(p.a + k0 * p.lda, fa);
    const float* const pb = p.b + (p.b_cell ? *p.b_cell * p.ldb : 0);
    const __amdgpu_buffer_rsrc_t rb = wg_rsrc(pb + k0 * p.ldb + n0, fb);
    // TWO: a lane's NT columns lie in b (virtual column < n) or in b2 (n is a multiple of NT: never astride); it loads
    // from both views every step with the offset of the other one out of range (-> zeros) and keeps its own
    int64_t fb2 = 0;
    if (TWO) {
        fb2 = (int64_t)rows * p.ldb2;
        const int64_t eb2 = p.b2_floats - k0 * p.ldb2;
        if (eb2 < fb2) fb2 = eb2;
    }
    const __amdgpu_buffer_rsrc_t rb2 = wg_rsrc(TWO ? p.b2 + k0 * p.ldb2 : p.b, TWO ? fb2 : 0);

    v16f acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // step s of the block: rows 2 s, 2 s + 1; wavefront w takes steps w, w + 4, ...
    const int col = lane & 31, half = lane >> 5;
    const int steps = (rows + 1) >> 1;
    const int lda4 = (int)p.lda * 4, ldb4 = (int)p.ldb * 4;
    int offa = (2 * wave + half) * lda4 + col * (MT * 4);
    int offb = (2 * wave + half) * ldb4 + col * (NT * 4);
    const int stepa = 2 * WG_WAVES * lda4, stepb = 2 * WG_WAVES * ldb4;
    // (offsets are advanced as unsigned numbers: an out-of-range lane starts at 2^31 and stays past every view, whose
    // size is below 2^31 bytes by the dispatcher's check on rows_per_block)
    unsigned offb2 = 0x80000000u, stepb2 = 0;
    bool in_b2 = false;
    if (TWO) {
        const int vcol = n0 + NT * col - p.n;            // this lane's first column, counted from the start of b2
        in_b2 = vcol >= 0;
        stepb2 = 2 * WG_WAVES * (unsigned)p.ldb2 * 4u;
        if (in_b2) {
            offb2 = (unsigned)((2 * wave + half) * (int)p.ldb2 * 4 + vcol * 4);
            offb = (int)0x80000000u;
        }
    }
    const int mine = steps > wave ? (steps - wave + WG_WAVES - 1) / WG_WAVES : 0;

    constexpr int U = WG_UNROLL(MT, NT);
    constexpr int D = WG_DEPTH(MT, NT);
    float av[D][U][MT], bv[D][U][NT];
    float bw[TWO ? D : 1][TWO ? U : 1][NT];
    float csum[MT];                  // this lane's share of sum_k A[k, MT * col + i] (the bias gradient)
#pragma unroll
    for (int i = 0; i < MT; ++i) csum[i] = 0.0f;
    const bool want_cs = CS && blockIdx.y == 0;
    if (TWO) {
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < NT; ++j) { bv[d][u][j] = 0.0f; bw[d][u][j] = 0.0f; }
    }
    auto issue = [&](int buf) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            wg_load<MT>(ra, offa, av[buf][u]);
            offa += stepa;
            if constexpr (TWO) {
                // both views, unconditionally: the view a column chunk does not touch is out of range for every lane and its
                // loads return zeros without memory traffic.  (Skipping them with `if (has1)` / `if (has2)` put uniform
                // branches — and the waits behind them — into the prefetch: 51.2 -> 47.2 us at [32 768, 64] x [720 | 20].)
                wg_load<NT>(rb, offb, bv[buf][u]);
                wg_load<NT>(rb2, (int)offb2, bw[buf][u]);
                offb = (int)((unsigned)offb + (unsigned)stepb);
                offb2 += stepb2;
            } else {
                wg_load<NT>(rb, offb, bv[buf][u]);
                offb += stepb;
            }
        }
    };
    auto multiply = [&](int buf) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float bsel[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) bsel[j] = (TWO && in_b2) ? bw[TWO ? buf : 0][TWO ? u : 0][j] : bv[buf][u][j];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][u][i], bsel[j], acc[i][j], 0, 0, 0);
                }
        }
        if (CS) {                    // (every column chunk adds; only chunk 0 stores)
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < MT; ++i) csum[i] += av[buf][u][i];
        }
    };
    // D register buffers in rotation, D - 1 groups of loads in flight behind the one being multiplied.  Rows past the
    // block's end lie outside the buffer views and load 0 (0 * 0 adds nothing), so the trip count is rounded up to a
    // whole rotation and the loads issued past the last group are harmless.
    const int groups = (mine + U - 1) / U;
    if (groups > 0) {
#pragma unroll
        for (int d = 0; d < D - 1; ++d) issue(d);
        for (int g = 0; g < groups; g += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                // the loads of the group D - 1 ahead go out BEFORE this group's products and stay there: left alone, the
                // instruction scheduler sinks them to a few MFMAs before their use (to shorten register lifetimes), which
                // turns the prefetch distance from a whole group (~1500 cycles of MFMA) into ~500 and the loop latency-bound
                issue((d + D - 1) % D);
                __builtin_amdgcn_sched_barrier(0);
                multiply(d);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    if (want_cs) {                   // lanes l and l + 32 hold the same columns; then the four wavefronts in order
        __shared__ float cfold[WG_WAVES][32 * MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const float s = csum[i] + __shfl_xor(csum[i], 32);
            if (half == 0) cfold[wave][col * MT + i] = s;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 32 * MT; e += 64 * WG_WAVES) {
            float s = cfold[0][e];
#pragma unroll
            for (int w = 1; w < WG_WAVES; ++w) s += cfold[w][e];
            p.cs[(int64_t)blockIdx.x * (32 * MT) + e] = s;
        }
    }

    // fold the four wavefronts' register images in LDS, wavefront 0 first
#pragma unroll
    for (int w = 0; w < WG_WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int e = ((i * NT + j) * 16 + r) * 64 + lane;
                        if (w == 0) fold[e] = acc[i][j][r];
                        else if (w < WG_WAVES - 1) fold[e] += acc[i][j][r];
                        else acc[i][j][r] += fold[e];
                    }
        }
        __syncthreads();
    }
    if (wave != WG_WAVES - 1) return;
    float* out = p.ws + ((int64_t)blockIdx.y * p.slabs + blockIdx.x) * WG_IMG(MT, NT);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[((i * NT + j) * 16 + r) * 64 + lane] = acc[i][j][r];
}

// element e of the slabs' register images, summed in a fixed order, stored at its place in C; the thread blocks past
// the image (column chunk 0 only, when column sums were asked for) sum the blocks' column-sum partials the same way
// RIDER (flexnet_wgrad_critic_finish): the finish blocks of flexnet_critic_td_backward ride behind this launch's own blocks
// (row 0 of the grid) — same thread-block shape, nothing shared with them but the launch
#define WG_RED FLEX_RED_G
template <int MT, int NT, bool RIDER = false>
__global__ __launch_bounds__(64 * WG_RED) void wgrad_reduce_kernel(WgradK p, CriticFinishK f) {
    constexpr int IMG_BLOCKS = WG_IMG(MT, NT) / 64;
    float sum;
    if constexpr (RIDER) {
        const int own = IMG_BLOCKS + (p.cs ? (32 * MT + 63) / 64 : 0);
        if ((int)blockIdx.x >= own) {
            if (blockIdx.y == 0 && (int)blockIdx.x - own < f.blocks) critic_finish_block(f, blockIdx.x - own);
            return;
        }
    }
    if (blockIdx.x >= IMG_BLOCKS) {
        if (blockIdx.y != 0) return;
        const int m = (blockIdx.x - IMG_BLOCKS) * 64 + (threadIdx.x & 63);
        if (!flex_reduce_rows(p.cs + m, 32 * MT, p.slabs, m < 32 * MT, sum) || m >= p.m) return;
        p.colsum[m] = p.accumulate ? p.colsum[m] + sum : sum;
        return;
    }
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);        // < WG_IMG: the grid covers it exactly
    if (!flex_reduce_rows(p.ws + (int64_t)blockIdx.y * p.slabs * WG_IMG(MT, NT) + e, WG_IMG(MT, NT), p.slabs, true, sum)) return;
    // register image -> matrix position (v_mfma_f32_32x32x2_f32 result layout, tile rows / columns interleaved)
    const int l = e & 63, r = (e >> 6) & 15, t = e >> 10;
    const int ti = t / NT, tj = t - ti * NT;
    const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;
    const int m = MT * i + ti, n = blockIdx.y * (32 * NT) + NT * j + tj;
    if (m >= p.m) return;
    float* dst;
    if (n < p.n) dst = p.c + (int64_t)m * p.ldc + n;
    else if (n - p.n < p.n2) dst = p.c2 + (int64_t)m * p.ldc2 + (n - p.n);       // (n2 = 0 without a second block)
    else return;
    *dst = p.accumulate ? *dst + sum : sum;
}

template <int MT, int NT>
static int wgrad_launch(WgradK p, int chunks, hipStream_t s, const CriticFinishK* rider) {
    if (p.b2) {
        if constexpr (MT == 2 && NT == 5) {
            if (p.cs) hipLaunchKernelGGL((wgrad_kernel<MT, NT, true, true>), dim3(p.slabs, chunks), dim3(64 * WG_WAVES), 0, s, p);
            else hipLaunchKernelGGL((wgrad_kernel<MT, NT, false, true>), dim3(p.slabs, chunks), dim3(64 * WG_WAVES), 0, s, p);
        } else return FLEXNET_EUNSUPPORTED;
    } else if (p.cs) hipLaunchKernelGGL((wgrad_kernel<MT, NT, true>), dim3(p.slabs, chunks), dim3(64 * WG_WAVES), 0, s, p);
    else hipLaunchKernelGGL((wgrad_kernel<MT, NT, false>), dim3(p.slabs, chunks), dim3(64 * WG_WAVES), 0, s, p);
    // the column-sum blocks ride at the end of every grid row; those of column chunks > 0 return at once
    const int cs_blocks = p.cs ? (32 * MT + 63) / 64 : 0;
    if (rider) {
        if constexpr (MT == 2 && NT == 5)
            hipLaunchKernelGGL((wgrad_reduce_kernel<MT, NT, true>), dim3(WG_IMG(MT, NT) / 64 + cs_blocks + rider->blocks, chunks),
                               dim3(64 * WG_RED), 0, s, p, *rider);
        else return FLEXNET_EUNSUPPORTED;
    } else {
        CriticFinishK none;
        none.blocks = 0;                                          // (never read without RIDER)
        hipLaunchKernelGGL((wgrad_reduce_kernel<MT, NT>), dim3(WG_IMG(MT, NT) / 64 + cs_blocks, chunks), dim3(64 * WG_RED), 0, s, p, none);
    }
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}

static int wgrad_run(const FlexWgradArgs* a, void* stream, const CriticFinishK* rider) {
    if (!a || !a->a || !a->b || !a->c || !a->workspace || a->k < 0 || a->m < 1 || a->n < 1) return FLEXNET_EINVAL;
    if (a->lda < a->m || a->ldb < a->n || (a->ldc != 0 && a->ldc < a->n)) return FLEXNET_EINVAL;
    if (a->m > 192 || a->lda >= (1 << 24) || a->ldb >= (1 << 24)) return FLEXNET_EUNSUPPORTED;
    const bool two = a->b2 != nullptr || a->n2 != 0;
    if (two && (!a->b2 || !a->c2 || a->n2 < 1 || a->ldb2 < a->n2 || (a->ldc2 != 0 && a->ldc2 < a->n2))) return FLEXNET_EINVAL;
    const int mt = a->m <= 32 ? 1 : a->m <= 64 ? 2 : 6;
    const int nt = a->n <= 32 ? 1 : a->n <= 64 ? 2 : (mt == 6 ? 2 : 5);
    if (two && (mt != 2 || nt != 5 || a->n % nt != 0 || a->ldb2 >= (1 << 24))) return FLEXNET_EUNSUPPORTED;
    const int n_all = a->n + (two ? a->n2 : 0);
    const int chunks = (n_all + 32 * nt - 1) / (32 * nt);
    const int64_t img = (int64_t)mt * nt * 1024;
    // thread blocks: about two per CU over all column chunks, at least 64 rows each, within the workspace
    const int64_t ws_floats = a->workspace_floats - WG_CS_FLOATS;
    if (ws_floats < chunks * img) return FLEXNET_EINVAL;
    int64_t slabs = (a->k + 63) / 64;
    // never more than two blocks per CU in total: with 515 blocks the last three ran as a second round (62 -> 3x us at
    // [32 768, 64] x [32 768, 720])
    const int64_t want = 512 / chunks > 0 ? 512 / chunks : 1;
    if (slabs > want) slabs = want;
    if (slabs * chunks * img > ws_floats) slabs = ws_floats / (chunks * img);
    if (slabs > 520) slabs = 520;
    if (slabs < 1) slabs = 1;
    int64_t rpb = (a->k + slabs - 1) / slabs;
    rpb = (rpb + 7) & ~(int64_t)7;
    if (rpb < 8) rpb = 8;
    int64_t ldmax = a->lda > a->ldb ? a->lda : a->ldb;
    if (two && a->ldb2 > ldmax) ldmax = a->ldb2;
    if (rpb * ldmax * 4 >= 0x7fffffffll) return FLEXNET_EUNSUPPORTED;
    slabs = a->k > 0 ? (a->k + rpb - 1) / rpb : 1;
    WgradK p;
    p.a = a->a; p.b = a->b; p.c = a->c; p.ws = a->workspace + WG_CS_FLOATS;
    p.cs = a->colsum ? a->workspace : nullptr; p.colsum = a->colsum;
    p.k = a->k; p.lda = a->lda; p.ldb = a->ldb;
    p.a_floats = a->k > 0 ? (a->k - 1) * a->lda + a->m : 0;
    p.b_floats = a->k > 0 ? (a->k - 1) * a->ldb + a->n : 0;
    p.m = a->m; p.n = a->n; p.rows_per_block = (int)rpb; p.slabs = (int)slabs; p.accumulate = a->accumulate;
    p.ldc = a->ldc > 0 ? a->ldc : a->n;
    p.b2 = two ? a->b2 : nullptr; p.c2 = two ? a->c2 : nullptr; p.ldb2 = two ? a->ldb2 : 0;
    p.b2_floats = two && a->k > 0 ? (a->k - 1) * a->ldb2 + a->n2 : 0;
    p.n2 = two ? a->n2 : 0; p.ldc2 = two ? (a->ldc2 > 0 ? a->ldc2 : a->n2) : 0;
    p.b_cell = a->b_row_cell;
    hipStream_t s = (hipStream_t)stream;
    switch (mt * 10 + nt) {
        case 11: return wgrad_launch<1, 1>(p, chunks, s, rider);
        case 12: return wgrad_launch<1, 2>(p, chunks, s, rider);
        case 15: return wgrad_launch<1, 5>(p, chunks, s, rider);
        case 21: return wgrad_launch<2, 1>(p, chunks, s, rider);
        case 22: return wgrad_launch<2, 2>(p, chunks, s, rider);
        case 25: return wgrad_launch<2, 5>(p, chunks, s, rider);
        case 61: return wgrad_launch<6, 1>(p, chunks, s, rider);
        case 62: return wgrad_launch<6, 2>(p, chunks, s, rider);
    }
    return FLEXNET_EUNSUPPORTED;
}

extern "C" int flexnet_wgrad(const FlexWgradArgs* a, void* stream) { return wgrad_run(a, stream, nullptr); }

// The critic's first-layer weight gradient with the finish of flexnet_critic_td_backward riding in its second-stage launch
// (include/flexnet.h): call flexnet_critic_td_backward_phases(critic, td, 1, stream) first — this is its phase 2 plus
// flexnet_wgrad(w), one launch fewer.  Only for the [64, 5 n] form the critic's first layer takes (else FLEXNET_EUNSUPPORTED,
// nothing launched: the caller falls back to the two separate calls).
extern "C" int flexnet_wgrad_critic_finish(const FlexWgradArgs* w, const FlexCriticTailArgs* critic, const FlexTdLossArgs* td, void* stream) {
    if (!w || !critic || !td) return FLEXNET_EINVAL;
    const int mt = w->m <= 32 ? 1 : w->m <= 64 ? 2 : 6;
    const int nt = w->n <= 32 ? 1 : w->n <= 64 ? 2 : (mt == 6 ? 2 : 5);
    if (mt != 2 || nt != 5) return FLEXNET_EUNSUPPORTED;
    CriticFinishK k;
    const int rc = critic_finish_prepare(critic, td, &k);
    if (rc != FLEXNET_OK) return rc;
    return wgrad_run(w, stream, &k);
}
