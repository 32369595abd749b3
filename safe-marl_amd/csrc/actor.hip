// actor.hip — fused inference pass of the reference's actor (gfx950).  Boundary: include/flexnet.h.
//
// madrl/agents/rnn_agent.py:25-33 (fc1 -> LayerNorm -> ReLU -> GRUCell -> fc2) as called by
// madrl/models/model.py:102-116 on the [b * n, obs (+ one-hot id)] reshape, for every row in one launch.
//
// Mapping: ONE LANE PER HIDDEN UNIT (hid_size = 64 = one wavefront), RT = 4 rows per wavefront at a time.  All
// weights live in LDS (141 KB of the CU's 160 KB: one 8-wavefront block per CU), transposed so that lane j reads
// W[j][i] at [i][j] (consecutive lanes, consecutive banks; the row pitch is odd so that the transposing fill is
// conflict-free too); the rows' inputs are staged [i][row] per wavefront, so one broadcast ds_read_b128 hands every
// lane the i-th input of all four rows.  Per input i a lane then does 4 FMAs (fc1) or 24 (the six GRU gate products
// of four rows): 34 k MAC per row in fp32 VALU FMAs.  Measured on one MI355X (tools/actor_bench.py): 20 480 rows
// (4096 envs x 5 agents) 59 us, 163 840 rows (a 32 768-sample update batch) 338 us = 33 TFLOP/s, against 118 / 506 us
// for the PyTorch module's ten kernels; two wavefronts per SIMD matter more than a bigger row tile (8 rows at one
// wavefront per SIMD: 69 / 409 us) because the loop is LDS-latency-, not LDS-bandwidth-bound.
// fp32 throughout (the reference's dtype); the summation order differs from rocBLAS, so results agree with the
// PyTorch module to ~1e-6 relative, not bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "flexnet.h"
#include "flex_launch.h"

// diagnostic build (-DACTOR_STAMPS): s_memtime at the phase boundaries of block 0's first wavefront (tools: scratch only)
#ifdef ACTOR_STAMPS
__device__ unsigned long long actor_stamps[16];
#define ASTAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) actor_stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
#define ASTAMP_C(k) do { if (blockIdx.x == 0 && threadIdx.x == 256) actor_stamps[8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int flexnet_debug_actor_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(actor_stamps), sizeof(actor_stamps)) == hipSuccess ? 0 : -2;
}
#else
#define ASTAMP(k) do { } while (0)
#define ASTAMP_C(k) do { } while (0)
#endif

#define HID FLEXNET_HID
#include "actor_r16.h"
#ifndef RT
#define RT 4                       // rows per wavefront tile (multiple of 4)
#endif
#ifndef AW
#define AW 8                       // wavefronts per block (two per SIMD; the block owns the CU's LDS)
#endif
#ifndef UNR
#define UNR 4
#endif
#define KC 64                      // fc1 input columns staged at a time
#define P1 (HID + 1)               // row pitch of W1T   [obs_dim][HID]
#define PG (3 * HID + 1)           // row pitch of WihT / WhhT   [HID][3 HID]

struct ActorLds {
    float w1t[FLEXNET_MAX_OBS * P1];
    float wih[HID * PG];
    float whh[HID * PG];
    float b1[HID], lnw[HID], lnb[HID];
    float w1id[FLEXNET_MAX_AGENTS * HID];
    float bih[3 * HID], bhh[3 * HID];
    float w2[FLEXNET_MAX_ACT * HID];
    float b2[FLEXNET_MAX_ACT];
    float stage[AW][2 * HID * RT];  // per wavefront: [i][row] inputs (fc1 chunk, then x | h, then h')
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ void st_rows(float* p, const float (&v)[RT]) {
#pragma unroll
    for (int q = 0; q < RT / 4; ++q) *reinterpret_cast<float4*>(p + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}
__device__ __forceinline__ void ld_rows(const float* p, float (&v)[RT]) {
#pragma unroll
    for (int q = 0; q < RT / 4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(p + 4 * q);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(64 * AW, 1) void actor_forward_kernel(FlexActorArgs a) {
    __shared__ ActorLds s;
    if (a.cursor) { const int64_t p = *a.cursor; if (!a.obs_pushed) a.obs += p * a.obs_slab_stride; a.hidden_in += p * a.hid_slab_stride; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int od = a.obs_dim, na = a.n_agents, ad = a.act_dim;
    const int ld1 = od + (a.agent_id ? na : 0);
    // ---- weights -> LDS (coalesced global reads, transposing conflict-free writes) -------------------------------
#pragma unroll 8
    for (int idx = tid; idx < HID * od; idx += 64 * AW) {
        const int j = idx / od, i = idx - j * od;
        s.w1t[i * P1 + j] = a.fc1_w[(int64_t)j * ld1 + i];
    }
#pragma unroll 8
    for (int idx = tid; idx < 3 * HID * HID; idx += 64 * AW) {
        const int gj = idx / HID, i = idx - gj * HID;                 // gj = gate * 64 + unit
        s.wih[i * PG + gj] = a.w_ih[idx];
        s.whh[i * PG + gj] = a.w_hh[idx];
    }
    for (int idx = tid; idx < 3 * HID; idx += 64 * AW) { s.bih[idx] = a.b_ih[idx]; s.bhh[idx] = a.b_hh[idx]; }
    if (tid < HID) {
        s.b1[tid] = a.fc1_b[tid];
        s.lnw[tid] = a.layernorm ? a.ln_w[tid] : 1.0f;
        s.lnb[tid] = a.layernorm ? a.ln_b[tid] : 0.0f;
    }
    for (int idx = tid; idx < FLEXNET_MAX_AGENTS * HID; idx += 64 * AW) {
        const int ag = idx / HID, j = idx - ag * HID;
        s.w1id[idx] = (a.agent_id && ag < na) ? a.fc1_w[(int64_t)j * ld1 + od + ag] : 0.0f;    // model.py:105-108
    }
    for (int idx = tid; idx < ad * HID; idx += 64 * AW) s.w2[idx] = a.fc2_w[idx];
    if (tid < ad) s.b2[tid] = a.fc2_b[tid];
    __syncthreads();

    float* st = s.stage[wave];
    const int n_tiles = (a.rows + RT - 1) / RT;
    for (int tile = wave * gridDim.x + blockIdx.x; tile < n_tiles; tile += gridDim.x * AW) {
        const int r0 = tile * RT;
        int row[RT], xo[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) { row[r] = min(r0 + r, a.rows - 1); xo[r] = actor_obs_off(a, row[r]); }   // spare rows of the last tile recompute a valid one
        // ---- fc1: acc[r] = sum_i W1[j][i] obs[r][i], inputs staged KC columns at a time ----------------------
        float acc[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r] = 0.0f;
        for (int c0 = 0; c0 < od; c0 += KC) {
            const int kc = min(KC, od - c0);
            float v[RT];
#pragma unroll
            for (int r = 0; r < RT; ++r) v[r] = lane < kc ? a.obs[xo[r] + c0 + lane] : 0.0f;
            st_rows(st + lane * RT, v);
            __builtin_amdgcn_wave_barrier();
#pragma unroll UNR
            for (int i = 0; i < kc; ++i) {
                const float w = s.w1t[(c0 + i) * P1 + lane];
                float ov[RT];
                ld_rows(st + i * RT, ov);
#pragma unroll
                for (int r = 0; r < RT; ++r) acc[r] = fmaf(w, ov[r], acc[r]);
            }
            __builtin_amdgcn_wave_barrier();
        }
        // ---- + bias (+ id column), LayerNorm over the 64 units, ReLU (rnn_agent.py:26-29) --------------------
        float x[RT], h[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int ag = row[r] % na;
            float z1 = acc[r] + s.b1[lane] + s.w1id[ag * HID + lane];
            if (a.layernorm) {
                const float mean = wave_sum(z1) * (1.0f / HID);
                const float d = z1 - mean;
                const float var = wave_sum(d * d) * (1.0f / HID);               // biased, like nn.LayerNorm
                z1 = d * rsqrtf(var + a.ln_eps) * s.lnw[lane] + s.lnb[lane];
            }
            x[r] = fmaxf(z1, 0.0f);
            h[r] = a.hidden_in[(int64_t)row[r] * HID + lane];
        }
        // ---- GRUCell (rnn_agent.py:30-31; torch gate order r, z, n) ------------------------------------------
        st_rows(st + lane * RT, x);
        st_rows(st + HID * RT + lane * RT, h);
        __builtin_amdgcn_wave_barrier();
        float ar[RT], az[RT], gin[RT], ghn[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) { ar[r] = 0.0f; az[r] = 0.0f; gin[r] = 0.0f; ghn[r] = 0.0f; }
#pragma unroll UNR
        for (int i = 0; i < HID; ++i) {
            const float wr = s.wih[i * PG + lane], wz = s.wih[i * PG + HID + lane], wn = s.wih[i * PG + 2 * HID + lane];
            const float ur = s.whh[i * PG + lane], uz = s.whh[i * PG + HID + lane], un = s.whh[i * PG + 2 * HID + lane];
            float xv[RT], hv[RT];
            ld_rows(st + i * RT, xv);
            ld_rows(st + HID * RT + i * RT, hv);
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                ar[r] = fmaf(wr, xv[r], fmaf(ur, hv[r], ar[r]));
                az[r] = fmaf(wz, xv[r], fmaf(uz, hv[r], az[r]));
                gin[r] = fmaf(wn, xv[r], gin[r]);
                ghn[r] = fmaf(un, hv[r], ghn[r]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        float hn[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const float rg = sigmoidf_(ar[r] + s.bih[lane] + s.bhh[lane]);
            const float zg = sigmoidf_(az[r] + s.bih[HID + lane] + s.bhh[HID + lane]);
            const float ng = tanhf(gin[r] + s.bih[2 * HID + lane] + rg * (ghn[r] + s.bhh[2 * HID + lane]));
            hn[r] = ng + zg * (h[r] - ng);                                       // (1 - z) n + z h
            if (r0 + r < a.rows) a.hidden_out[(int64_t)(r0 + r) * HID + lane] = hn[r];
        }
        // ---- fc2 (rnn_agent.py:32): lane (row, output) takes one 64-long dot product -------------------------
        st_rows(st + lane * RT, hn);
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < RT * ad; e += 64) {
            const int r = e / ad, k = e - r * ad;
            float o = s.b2[k];
            for (int i = 0; i < HID; ++i) o = fmaf(s.w2[k * HID + i], st[i * RT + r], o);
            if (r0 + r < a.rows) {
                const int64_t at = (int64_t)(r0 + r) * ad + k;
                a.means[at] = o;
                if (a.noise) {                                                    // util.py:57-64, 125-128
                    const float act = tanhf(o + a.std * a.noise[at]);
                    a.action[at] = act;
                    a.env_action[at] = 0.5f * (fminf(fmaxf(act, a.action_low), a.action_high) + 1.0f) * (a.action_high - a.action_low) + a.action_low;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Matrix-core version.  v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD = twice v_fma_f32) computes
// D[32 x 32] += A[32 x 2] B[2 x 32]; here A = WEIGHTS (rows = output units) and B = ACTIVATIONS (columns = the 32
// batch rows of a wavefront's tile), i.e. every layer is evaluated transposed, D[unit][row].  That choice makes the
// layers chain THROUGH REGISTERS: lane (row r, half h) ends a layer holding units 8q + 4h + j (reg = 4q + j) of
// its row — exactly what the next layer's B operand wants if the k-pair of MFMA step (q, j) is taken as
// (8q + j, 8q + 4 + j) instead of two consecutive inputs, and the A operand (weights, from LDS) is read in the
// matching order.  The network inputs need no staging either: lane (row, half) reads columns 8q + 4 half .. + 3 of its
// observation row (and units of its previous hidden state) with one 16-byte load per group of eight and uses them as
// the B operands of steps (q, 0..3).  The gate biases are what the accumulators start from, fc2 is one more transposed
// product (outputs padded to 32), so the gate epilogue touches no LDS; LayerNorm is an in-lane sum plus one exchange
// between the halves.
// Per 32 rows: 144 + 384 + 32 MFMAs (36 k cycles of a SIMD at the issue rate), two wavefronts per SIMD.
// ---------------------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MW 8                       // wavefronts per block

struct ActorLdsM {
    float w1t[FLEXNET_MAX_OBS * P1];
    float wih[HID * PG];
    float whh[HID * PG];
    float b1[HID], lnw[HID], lnb[HID];
    float w1id[FLEXNET_MAX_AGENTS * HID];
    float gb[4 * HID];                  // gate biases as the accumulators start from them: r, z (b_ih + b_hh), n_x, n_h
    float w2p[HID * 32];                // fc2.weight transposed and padded to 32 outputs: w2p[unit][k], zero for k >= act_dim
    float b2[FLEXNET_MAX_ACT];
};

#define DU0(i) (8 * ((i) >> 2) + ((i) & 3))       // unit of accumulator register i within a 32-unit tile, without the half's + 4 hf
#define MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x2f32((a_), (b_), (c_), 0, 0, 0)

__global__ __launch_bounds__(64 * MW, 2) void actor_forward_mfma_kernel(FlexActorArgs a) {
    __shared__ ActorLdsM s;
    ASTAMP(0);
    if (a.cursor) {
        const int64_t p = *a.cursor;
        if (!a.obs_pushed) a.obs += p * a.obs_slab_stride;
        a.hidden_in += p * a.hid_slab_stride;
        if (a.cursor_out && blockIdx.x == 0 && threadIdx.x == 0) *a.cursor_out = p;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rb = lane & 31, hf = lane >> 5;
    const int od = a.obs_dim, na = a.n_agents, ad = a.act_dim;
    const int ld1 = od + (a.agent_id ? na : 0);
    // observations through a buffer descriptor: 16-byte loads at dword alignment, zeros past the end of the tensor.
    // The first tile's first six column groups are requested before the weights are staged, every later tile's at the
    // end of the previous tile's fc1: the load latency never sits at the head of a tile.
    typedef float v4f_ __attribute__((ext_vector_type(4)));
    constexpr int QB = 6;
    const int64_t obs_bytes = actor_obs_bytes(a);
    const __amdgpu_buffer_rsrc_t robs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.obs), 0, obs_bytes > 0x7ffffff0ll ? 0x7ffffff0 : (int)obs_bytes, 0x00027000);
    const int nq = (od + 7) >> 3;
    const int n_tiles = (a.rows + 31) / 32;
    // tile t goes to block t % grid, wavefront (t / grid) % MW: a small batch spreads over all CUs first
    int tile = wave * gridDim.x + blockIdx.x;
    v4f_ xc[QB];
    {
        const int xoff0 = (actor_obs_off(a, min(tile * 32 + rb, a.rows - 1)) + 4 * hf) * 4;
#pragma unroll
        for (int e = 0; e < QB; ++e)
            xc[e] = __builtin_bit_cast(v4f_, __builtin_amdgcn_raw_buffer_load_b128(
                robs, tile < n_tiles && e < nq ? xoff0 + 32 * e : -1, 0, 0));
    }
    // Weights -> LDS (transposed).  All global reads of a thread are issued before the first LDS write — 12 + up to 5
    // 16-byte loads in one round trip instead of ~66 scalar loads in rounds of 6-8: the staging was a third of a
    // rollout-sized call (20 480 rows: 37 -> 2x us).  fc1's rows (pitch ld1 floats, any dword alignment) go through a
    // buffer descriptor, whose 16-byte loads need dword alignment only and return 0 past the end.
    {
        constexpr int NG = 3 * HID * HID / 4 / (64 * MW);                 // 6 float4 per thread and matrix
        static_assert(NG * 4 * 64 * MW == 3 * HID * HID, "gate matrices split evenly");
        float4 vi[NG], vh[NG];
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            vi[t] = reinterpret_cast<const float4*>(a.w_ih)[tid + 64 * MW * t];
            vh[t] = reinterpret_cast<const float4*>(a.w_hh)[tid + 64 * MW * t];
        }
        constexpr int NF = (HID * FLEXNET_MAX_OBS / 4 + 64 * MW - 1) / (64 * MW);   // 5
        const int q4 = (od + 3) >> 2;                                      // float4 groups per fc1 row (the last may be ragged)
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.fc1_w), 0, HID * ld1 * 4, 0x00027000);
        typedef float v4f_ __attribute__((ext_vector_type(4)));
        v4f_ vf[NF];
#pragma unroll
        for (int t = 0; t < NF; ++t) {
            const int g = tid + 64 * MW * t, j = g / q4, i4 = g - j * q4;
            vf[t] = __builtin_bit_cast(v4f_, __builtin_amdgcn_raw_buffer_load_b128(r1, j < HID ? (j * ld1 + 4 * i4) * 4 : -1, 0, 0));
        }
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            const int idx = 4 * (tid + 64 * MW * t), gj = idx / HID, i = idx - gj * HID;
            s.wih[i * PG + gj] = vi[t].x; s.wih[(i + 1) * PG + gj] = vi[t].y;
            s.wih[(i + 2) * PG + gj] = vi[t].z; s.wih[(i + 3) * PG + gj] = vi[t].w;
            s.whh[i * PG + gj] = vh[t].x; s.whh[(i + 1) * PG + gj] = vh[t].y;
            s.whh[(i + 2) * PG + gj] = vh[t].z; s.whh[(i + 3) * PG + gj] = vh[t].w;
        }
#pragma unroll
        for (int t = 0; t < NF; ++t) {
            const int g = tid + 64 * MW * t, j = g / q4, i = 4 * (g - j * q4);
            if (j < HID) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (i + e < od) s.w1t[(i + e) * P1 + j] = vf[t][e];
            }
        }
    }
    if (tid < HID) {
        s.gb[tid] = a.b_ih[tid] + a.b_hh[tid];
        s.gb[HID + tid] = a.b_ih[HID + tid] + a.b_hh[HID + tid];
        s.gb[2 * HID + tid] = a.b_ih[2 * HID + tid];
        s.gb[3 * HID + tid] = a.b_hh[2 * HID + tid];
    }
    for (int idx = tid; idx < (((od + 7) & ~7) - od) * HID; idx += 64 * MW)        // fc1 runs over 8-column groups
        s.w1t[(od + idx / HID) * P1 + (idx % HID)] = 0.0f;
    if (tid < HID) {
        s.b1[tid] = a.fc1_b[tid];
        s.lnw[tid] = a.layernorm ? a.ln_w[tid] : 1.0f;
        s.lnb[tid] = a.layernorm ? a.ln_b[tid] : 0.0f;
    }
    for (int idx = tid; idx < FLEXNET_MAX_AGENTS * HID; idx += 64 * MW) {
        const int ag = idx / HID, j = idx - ag * HID;
        s.w1id[idx] = (a.agent_id && ag < na) ? a.fc1_w[(int64_t)j * ld1 + od + ag] : 0.0f;
    }
    for (int idx = tid; idx < HID * 32; idx += 64 * MW) {
        const int u = idx >> 5, k = idx & 31;
        s.w2p[idx] = k < ad ? a.fc2_w[k * HID + u] : 0.0f;
    }
    if (tid < ad) s.b2[tid] = a.fc2_b[tid];
    __syncthreads();
    ASTAMP(1);

    // Lane-dependent parts of every LDS index go into a base pointer per array, the rest is a compile-time constant that
    // fits the 16-bit offset field of the ds_read: otherwise the compiler materialises one address register per
    // unrolled access, hoists all of them out of the tile loop and spills.
    const float* wi_l = s.wih + (4 * hf) * PG + rb;
    const float* wh_l = s.whh + (4 * hf) * PG + rb;
    const float* gb_l = s.gb + 4 * hf;
    const float* b1_l = s.b1 + 4 * hf;
    const float* lnw_l = s.lnw + 4 * hf;
    const float* lnb_l = s.lnb + 4 * hf;
    const float* w2p_l = s.w2p + (4 * hf) * 32 + rb;
    const float* w1_l = s.w1t + (4 * hf) * P1 + rb;
    const uint64_t rng_seed = a.rng_state ? a.rng_state[0] : 0ull, rng_step = a.rng_state ? a.rng_state[1] : 0ull;
    for (; tile < n_tiles; tile += gridDim.x * MW) {
        const int r0 = tile * 32;
        const int row = min(r0 + rb, a.rows - 1);                    // this lane's batch row (both halves share it)
        // ---- fc1: z1[unit][row] ----------------------------------------------------------------------------------
        f32x16 z1[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) z1[u][i] = 0.0f;
        // Lane (row, half) reads columns 8q + 4 half .. + 3 of its row with ONE 16-byte load per group q — the k-pair of
        // MFMA step (q, j) is (8q + j, 8q + 4 + j), as in the layers behind — so the observation goes from row-major memory
        // straight into B operands: no LDS hand-over, no barriers.  Six groups (24 registers) in flight, the next six
        // requested before the current ones are multiplied.
        const int xoff = (actor_obs_off(a, row) + 4 * hf) * 4;
        v4f_ xn[QB];
        for (int q0 = 0; q0 < nq; q0 += QB) {
#pragma unroll
            for (int e = 0; e < QB; ++e)
                xn[e] = __builtin_bit_cast(v4f_, __builtin_amdgcn_raw_buffer_load_b128(
                    robs, q0 + QB + e < nq ? xoff + 32 * (q0 + QB + e) : -1, 0, 0));
            const float* wq = w1_l + (8 * q0) * P1;
#pragma unroll
            for (int e = 0; e < QB; ++e) {
                if (q0 + e < nq) {                                   // wavefront-uniform
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float a0 = wq[(8 * e + j) * P1], a1 = wq[(8 * e + j) * P1 + 32];
                        z1[0] = MFMA(a0, xc[e][j], z1[0]);
                        z1[1] = MFMA(a1, xc[e][j], z1[1]);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < QB; ++e) xc[e] = xn[e];
        }
        {   // the NEXT tile's first six groups go out now and land underneath LayerNorm and the GRU
            const int nt = tile + gridDim.x * MW;
            const int nxoff = (actor_obs_off(a, min(nt * 32 + rb, a.rows - 1)) + 4 * hf) * 4;
#pragma unroll
            for (int e = 0; e < QB; ++e)
                xc[e] = __builtin_bit_cast(v4f_, __builtin_amdgcn_raw_buffer_load_b128(
                    robs, nt < n_tiles && e < nq ? nxoff + 32 * e : -1, 0, 0));
        }
        ASTAMP(2);
        const bool save = a.save_z1 != nullptr && r0 + rb < a.rows;       // training forward: keep what the backward needs
        const int64_t so = (int64_t)(r0 + rb) * HID + 4 * hf;              // + 32 u + 8 q: this lane's float4 groups
        if (save) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(a.save_z1 + so + 32 * u + 8 * q) =
                        make_float4(z1[u][4 * q], z1[u][4 * q + 1], z1[u][4 * q + 2], z1[u][4 * q + 3]);
        }
        // ---- + bias (+ id column), LayerNorm over the row's 64 units (32 here, 32 in the other half), ReLU ----
        const int ag = row % na;
        const float* w1id_l = s.w1id + ag * HID + 4 * hf;
        float sum = 0.0f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int cu = 32 * u + DU0(i);                            // + 4 hf is in the base pointers
                z1[u][i] += b1_l[cu] + w1id_l[cu];
                sum += z1[u][i];
            }
        if (a.layernorm) {
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * (1.0f / HID);
            float var = 0.0f;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) { const float d = z1[u][i] - mean; var = fmaf(d, d, var); }
            var += __shfl_xor(var, 32, 64);
            const float rstd = rsqrtf(var * (1.0f / HID) + a.ln_eps);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int cu = 32 * u + DU0(i);
                    z1[u][i] = (z1[u][i] - mean) * rstd * lnw_l[cu] + lnb_l[cu];
                }
        }
        // previous hidden state of this row in the same register layout
        f32x16 hv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = *reinterpret_cast<const float4*>(a.hidden_in + (int64_t)row * HID + 32 * u + 8 * q + 4 * hf);
                hv[u][4 * q] = t.x; hv[u][4 * q + 1] = t.y; hv[u][4 * q + 2] = t.z; hv[u][4 * q + 3] = t.w;
            }
        // ---- GRUCell: six gate products, x part and h part, chained through registers; one 32-unit output tile at a
        //      time (64 accumulator registers live instead of 128), gates and fc2 partial sums right behind it ----------
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) z1[u][i] = fmaxf(z1[u][i], 0.0f);     // x = ReLU(LayerNorm(z1))
        if (save) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(a.save_x + so + 32 * u + 8 * q) =
                        make_float4(z1[u][4 * q], z1[u][4 * q + 1], z1[u][4 * q + 2], z1[u][4 * q + 3]);
        }
        f32x16 hnew[2];
        ASTAMP(3);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 ar, az, gin, ghn;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int cu = 32 * t + DU0(i);
                ar[i] = gb_l[cu]; az[i] = gb_l[HID + cu]; gin[i] = gb_l[2 * HID + cu]; ghn[i] = gb_l[3 * HID + cu];
            }
            // software pipeline: the six weights of step s + 1 are requested before the MFMAs of step s; the scheduling
            // barrier keeps the compiler from hoisting ALL 192 LDS reads above the loop (it spilled doing so)
            float w[6];
            {
                const int o = DU0(0) * PG + 32 * t;
                w[0] = wi_l[o]; w[1] = wi_l[o + HID]; w[2] = wi_l[o + 2 * HID];
                w[3] = wh_l[o]; w[4] = wh_l[o + HID]; w[5] = wh_l[o + 2 * HID];
            }
#pragma unroll
            for (int st_ = 0; st_ < 32; ++st_) {
                const int u = st_ >> 4, i = st_ & 15;
                float wn[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                if (st_ + 1 < 32) {
                    const int o = (32 * ((st_ + 1) >> 4) + DU0((st_ + 1) & 15)) * PG + 32 * t;
                    wn[0] = wi_l[o]; wn[1] = wi_l[o + HID]; wn[2] = wi_l[o + 2 * HID];
                    wn[3] = wh_l[o]; wn[4] = wh_l[o + HID]; wn[5] = wh_l[o + 2 * HID];
                }
                const float bx = z1[u][i], bh = hv[u][i];               // this half's input unit 32 u + DU0(i) + 4 hf
                ar = MFMA(w[0], bx, ar);
                az = MFMA(w[1], bx, az);
                gin = MFMA(w[2], bx, gin);
                ar = MFMA(w[3], bh, ar);
                az = MFMA(w[4], bh, az);
                ghn = MFMA(w[5], bh, ghn);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 6; ++e) w[e] = wn[e];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float rg = fast_sigmoid(ar[i]);
                const float zg = fast_sigmoid(az[i]);
                const float ng = fast_tanh(gin[i] + rg * ghn[i]);
                hnew[t][i] = ng + zg * (hv[t][i] - ng);                  // (1 - z) n + z h
                ar[i] = rg; az[i] = zg; gin[i] = ng;                     // the gates themselves, for the stores below
            }
            if (save) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t at = so + 32 * t + 8 * q;
                    *reinterpret_cast<float4*>(a.save_r + at) = make_float4(ar[4 * q], ar[4 * q + 1], ar[4 * q + 2], ar[4 * q + 3]);
                    *reinterpret_cast<float4*>(a.save_z + at) = make_float4(az[4 * q], az[4 * q + 1], az[4 * q + 2], az[4 * q + 3]);
                    *reinterpret_cast<float4*>(a.save_n + at) = make_float4(gin[4 * q], gin[4 * q + 1], gin[4 * q + 2], gin[4 * q + 3]);
                    *reinterpret_cast<float4*>(a.save_hn + at) = make_float4(ghn[4 * q], ghn[4 * q + 1], ghn[4 * q + 2], ghn[4 * q + 3]);
                }
            }
            if (r0 + rb < a.rows) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(a.hidden_out + (int64_t)(r0 + rb) * HID + 32 * t + 8 * q + 4 * hf) =
                        make_float4(hnew[t][4 * q], hnew[t][4 * q + 1], hnew[t][4 * q + 2], hnew[t][4 * q + 3]);
            }
        }
        ASTAMP(4);
        // ---- fc2 (rnn_agent.py:32) as one more transposed product: means[k][row], k padded to 32; lane (row, half) ends
        //      with actions 4 half .. 4 half + 3 of its row in the first four accumulator registers ----------------------
        f32x16 mo;
#pragma unroll
        for (int i = 0; i < 16; ++i) mo[i] = 0.0f;
#pragma unroll
        for (int st_ = 0; st_ < 32; ++st_)
            mo = MFMA(w2p_l[(32 * (st_ >> 4) + DU0(st_ & 15)) * 32], hnew[st_ >> 4][st_ & 15], mo);
        if (r0 + rb < a.rows) {
            float zr[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (!a.noise && a.rng_state && 4 * hf < ad) actor_noise4(rng_seed, rng_step, (uint32_t)(r0 + rb), (uint32_t)hf, zr);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 4 * hf + r;
                if (k < ad) {
                    const float o = mo[r] + s.b2[k];
                    const int64_t at = (int64_t)(r0 + rb) * ad + k;
                    a.means[at] = o;
                    if (a.action) {                                           // util.py:57-64, 125-128
                        const float act = tanhf(o + a.std * (a.noise ? a.noise[at] : zr[r]));
                        a.action[at] = act;
                        a.env_action[at] = 0.5f * (fminf(fmaxf(act, a.action_low), a.action_high) + 1.0f) * (a.action_high - a.action_low) + a.action_low;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        ASTAMP(5);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Rollout-size batches (round 3): 16-row tiles on v_mfma_f32_16x16x4_f32, FIVE tiles per CU and round.
//
// A rollout call at 4096 environments is 20 480 rows = 640 of the 32-row tiles above on 1 024 SIMDs: one tile's latency
// (560 MFMAs x 64 cycles = 36 k cycles of a 57 k-cycle call) with 38 % of the SIMDs idle, and any split into equal
// tasks that does not divide 20 rows per SIMD evenly leaves some SIMD with two of them.  What divides evenly: a CU takes 80
// rows = five 16-row tiles (544 MFMAs x 32 cycles = 17.4 k cycles each).  Wavefronts 0-3 — one per SIMD — run one tile each,
// start to finish, exactly as above with the 16x16 register maps (lane = row j + 16 g, register r of unit tile S = unit
// 16 S + 4 g + r; the layers chain through registers).  Wavefronts 4-7 — the second wavefront of each SIMD — share the
// FIFTH tile by OUTPUT UNITS: wavefront 4 + q computes units 16 q .. 16 q + 15 of fc1 (36 MFMAs), of the three GRU gates
// (96) and wavefront 4 fc2 (16) — a quarter of a tile's matrix work each, 21.9 k MFMA cycles per SIMD in total — and they
// hand fc1's output and the new hidden state to each other through 8 KB of LDS with two rendezvous (an LDS counter
// each; the block's other four wavefronts never wait).  Every output unit is still one MFMA chain over the inputs in
// the same order, so a row's result does not depend on which kind of wavefront computed it.
// Weights in LDS (156 KB): pitches 68 / 388 / 20 floats put the four lane groups of an A-operand read 16 banks apart.
// Inference only (no saved activations: the update batches run the 32-row kernel).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * R16_W) void actor_rollout16_kernel(FlexActorArgs a) {
    __shared__ ActorLds16 s;
    actor_r16_body(a, s);
}


extern "C" int flexnet_actor_forward(const FlexActorArgs* a, void* stream) {
    if (!a || a->rows < 0) return FLEXNET_EINVAL;
    if (a->rows == 0) return FLEXNET_OK;
    if (!a->obs || !a->hidden_in || !a->fc1_w || !a->fc1_b || !a->w_ih || !a->w_hh || !a->b_ih || !a->b_hh || !a->fc2_w ||
        !a->fc2_b || !a->means || !a->hidden_out || (a->layernorm && (!a->ln_w || !a->ln_b)) ||
        ((a->noise || a->rng_state) && (!a->action || !a->env_action)))
        return FLEXNET_EINVAL;
    if (!a->noise && !a->rng_state && a->action) return FLEXNET_EINVAL;      // exploration outputs need a noise source
    if (a->rng_state && !a->noise && a->variant == 1) return FLEXNET_EUNSUPPORTED;
    if (a->obs_dim < 1 || a->obs_dim > FLEXNET_MAX_OBS || a->n_agents < 1 || a->n_agents > FLEXNET_MAX_AGENTS ||
        a->act_dim < 1 || a->act_dim > FLEXNET_MAX_ACT)
        return FLEXNET_EUNSUPPORTED;
    if (a->obs_pushed) {                                  // observations read in place from an environment's history
        if (a->obs_slots < 1 || a->obs_slot_w < 1 || a->obs_slots * a->obs_slot_w != a->obs_dim || a->obs_pushed_stride < 1 ||
            a->obs_row_stride < 2 * a->obs_dim || a->rows % a->n_agents != 0) return FLEXNET_EINVAL;
        if ((int64_t)a->rows * a->obs_row_stride * 4 >= 0x7ffffff0ll) return FLEXNET_EUNSUPPORTED;
    }
    if ((int64_t)a->rows * a->obs_dim * 4 >= 0x7ffffff0ll) return FLEXNET_EUNSUPPORTED;   // observations are addressed with 32-bit byte offsets
    // (obs_slab_stride: one launch's rows for the rollout's slab ring; a stride of ONE sample row — n_agents * obs_dim — makes
    //  the cursor cell the first row of an in-place window of the replay's stacked-observation ring: nets.RING_VIEWS)
    if (a->cursor && ((!a->obs_pushed && a->obs_slab_stride < (int64_t)a->n_agents * a->obs_dim) || a->hid_slab_stride < 0)) return FLEXNET_EINVAL;
    if (a->cursor_out && (!a->cursor || a->variant == 1 || a->cursor_out == a->cursor)) return FLEXNET_EINVAL;
    {
        const int saves = (a->save_z1 != nullptr) + (a->save_x != nullptr) + (a->save_r != nullptr) + (a->save_z != nullptr) +
                          (a->save_n != nullptr) + (a->save_hn != nullptr);
        if (saves != 0 && saves != 6) return FLEXNET_EINVAL;                  // all six or none
        if (saves && a->variant != 0 && a->variant != 3) return FLEXNET_EUNSUPPORTED;
    }
    const int cus = flex_cu_count();                      // one block per CU owns that CU's LDS
    if (cus < 1) return FLEXNET_EHIP;
    const bool saves = a->save_z1 != nullptr;
    // Inference batches take the five-tiles-per-CU kernel where it is faster.  Measured (tools/actor_bench.py, graph
    // replays, 256 CUs): it needs ~7 + 13 r us for r = ceil(rows / (80 CUs)) rounds, the 32-row kernel ~7 + 19 w us for
    // w = ceil(rows / (128 CUs)) tiles per SIMD — 20 480 rows: 20.6 vs 26.1 us, 30 720: 32.4 vs 27.2, 40 960 (SAFEMADDPG's
    // rollout at 8192 envs): 33.6 vs 44.2, 81 920: 58.9 vs 63.9, 163 840: 108 vs 101.  variant 2 asks for it whatever the
    // size, variant 3 for the 32-row kernel whatever the size (tests, tools/actor_bench.py).
    const int64_t r16 = (a->rows + 80 * (int64_t)cus - 1) / (80 * (int64_t)cus), w32 = (a->rows + 128 * (int64_t)cus - 1) / (128 * (int64_t)cus);
    const bool rollout16 = !saves && (a->variant == 2 || (a->variant == 0 && 13 * r16 < 19 * w32));
    if (rollout16) {
        const int rounds = ((a->rows + 15) / 16 + 4) / 5;
        const int blocks = rounds < cus ? rounds : cus;
        hipLaunchKernelGGL(actor_rollout16_kernel, dim3(blocks), dim3(64 * R16_W), 0, (hipStream_t)stream, *a);
    } else if (a->variant == 0 || a->variant == 2 || a->variant == 3) {
        const int tiles = (a->rows + 31) / 32;
        const int blocks = tiles < cus ? tiles : cus;
        hipLaunchKernelGGL(actor_forward_mfma_kernel, dim3(blocks), dim3(64 * MW), 0, (hipStream_t)stream, *a);
    } else {
        const int tiles = (a->rows + RT - 1) / RT;
        const int blocks = tiles < cus ? tiles : cus;
        hipLaunchKernelGGL(actor_forward_kernel, dim3(blocks), dim3(64 * AW), 0, (hipStream_t)stream, *a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fprintf(stderr, "[flexnet] actor_forward launch failed: %s\n", hipGetErrorString(e));
        return FLEXNET_EHIP;
    }
    return FLEXNET_OK;
}
