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
    if (a.cursor) { const int64_t p = *a.cursor; a.obs += p * a.obs_slab_stride; a.hidden_in += p * a.hid_slab_stride; }
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
        int row[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) row[r] = min(r0 + r, a.rows - 1);             // spare rows of the last tile recompute a valid one
        // ---- fc1: acc[r] = sum_i W1[j][i] obs[r][i], inputs staged KC columns at a time ----------------------
        float acc[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r] = 0.0f;
        for (int c0 = 0; c0 < od; c0 += KC) {
            const int kc = min(KC, od - c0);
            float v[RT];
#pragma unroll
            for (int r = 0; r < RT; ++r) v[r] = lane < kc ? a.obs[(int64_t)row[r] * od + c0 + lane] : 0.0f;
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

// v_exp_f32 / v_rcp_f32 forms (about 1 ulp each): sigmoid(x) = 1 / (1 + 2^(-x log2 e)), tanh(x) = 1 - 2 / (2^(2x log2 e) + 1)
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * x)); }
__device__ __forceinline__ float fast_tanh(float x) {
    const float xc = fminf(fmaxf(x, -15.0f), 15.0f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.88539008f * xc) + 1.0f);
}
// Exploration noise drawn in the kernel (FlexActorArgs::rng_state): four standard normal numbers for actions
// 4 group .. 4 group + 3 of one row.  Philox4x32-10 (the generator of the env's reset stream, flex_device.h), counter =
// (row, group, step lo, tag ^ step hi), key = seed; uniforms from the top 24 bits, (x + 0.5) 2^-24 in (0, 1); Box-Muller.
// Restated for the tests in tests/test_actor_gpu.py.
#define ACTOR_NOISE_TAG 0xAC70A5E1u
__device__ __forceinline__ void actor_noise4(uint64_t seed, uint64_t step, uint32_t row, uint32_t group, float* z) {
    uint32_t c0 = row, c1 = group, c2 = (uint32_t)step, c3 = ACTOR_NOISE_TAG ^ (uint32_t)(step >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const float s24 = 1.0f / 16777216.0f;
    const float u0 = ((float)(c0 >> 8) + 0.5f) * s24, u1 = ((float)(c1 >> 8) + 0.5f) * s24;
    const float u2 = ((float)(c2 >> 8) + 0.5f) * s24, u3 = ((float)(c3 >> 8) + 0.5f) * s24;
    const float ra = sqrtf(-2.0f * logf(u0)), rb_ = sqrtf(-2.0f * logf(u2));
    z[0] = ra * cospif(2.0f * u1); z[1] = ra * sinpif(2.0f * u1);
    z[2] = rb_ * cospif(2.0f * u3); z[3] = rb_ * sinpif(2.0f * u3);
}

#define DU0(i) (8 * ((i) >> 2) + ((i) & 3))       // unit of accumulator register i within a 32-unit tile, without the half's + 4 hf
#define MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x2f32((a_), (b_), (c_), 0, 0, 0)

__global__ __launch_bounds__(64 * MW, 2) void actor_forward_mfma_kernel(FlexActorArgs a) {
    __shared__ ActorLdsM s;
    ASTAMP(0);
    if (a.cursor) {
        const int64_t p = *a.cursor;
        a.obs += p * a.obs_slab_stride; a.hidden_in += p * a.hid_slab_stride;
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
    const int64_t obs_bytes = (int64_t)a.rows * od * 4;
    const __amdgpu_buffer_rsrc_t robs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.obs), 0, obs_bytes > 0x7ffffff0ll ? 0x7ffffff0 : (int)obs_bytes, 0x00027000);
    const int nq = (od + 7) >> 3;
    const int n_tiles = (a.rows + 31) / 32;
    // tile t goes to block t % grid, wavefront (t / grid) % MW: a small batch spreads over all CUs first
    int tile = wave * gridDim.x + blockIdx.x;
    v4f_ xc[QB];
    {
        const int xoff0 = (min(tile * 32 + rb, a.rows - 1) * od + 4 * hf) * 4;
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
        const int xoff = (row * od + 4 * hf) * 4;
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
            const int nxoff = (min(nt * 32 + rb, a.rows - 1) * od + 4 * hf) * 4;
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
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define R16_W 8
#define R16_P1 68                       // fc1^T   [k][64 units]
#define R16_PG 388                      // gates^T [k][W_ih r z n (192) | W_hh r z n (192)]
#define R16_P2 20                       // fc2^T   [k][16 outputs, zero past act_dim]
#define R16_PX 68                       // exchange tiles [row][unit]
#define MFMA16(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x4f32((a_), (b_), (c_), 0, 0, 0)

struct __attribute__((aligned(16))) ActorLds16 {
    float w1t[FLEXNET_MAX_OBS * R16_P1];
    float wg[HID * R16_PG];
    float w2p[HID * R16_P2];
    float b1[HID], lnw[HID], lnb[HID];
    float w1id[FLEXNET_MAX_AGENTS * HID];
    float gb[4 * HID];
    float b2[FLEXNET_MAX_ACT];
    float xz[16 * R16_PX];              // cooperative tile: fc1 output (bias and id column added) of all 64 units
    float xh[16 * R16_PX];              // cooperative tile: the new hidden state
    int sync[2];
};

// sum over the four lane groups holding one row (lanes j, j + 16, j + 32, j + 48): same bits in all four
__device__ __forceinline__ float r16_group_sum(float v) {
    const unsigned x = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    const unsigned y = __builtin_bit_cast(unsigned, v);
    auto q = __builtin_amdgcn_permlane32_swap(y, y, false, false);
    return __builtin_bit_cast(float, (unsigned)q[0]) + __builtin_bit_cast(float, (unsigned)q[1]);
}

// rendezvous of the four cooperating wavefronts: everything this wavefront wrote to LDS before is visible to whoever sees
// the count (a wavefront's LDS operations execute in order); never called by wavefronts 0-3
__device__ __forceinline__ void r16_rendezvous(int* cnt, int target, int lane) {
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
}

// LayerNorm (+ ReLU) of a row held as 4 x 4 registers per lane, in place: rnn_agent.py:27-28
__device__ __forceinline__ void r16_ln_relu(f32x4* z, bool layernorm, float eps, const float* lnw_l, const float* lnb_l) {
    if (layernorm) {
        float sum = 0.0f;
#pragma unroll
        for (int S = 0; S < 4; ++S)
#pragma unroll
            for (int r = 0; r < 4; ++r) sum += z[S][r];
        const float mean = r16_group_sum(sum) * (1.0f / HID);
        float var = 0.0f;
#pragma unroll
        for (int S = 0; S < 4; ++S)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = z[S][r] - mean; var = fmaf(d, d, var); }
        const float rstd = rsqrtf(r16_group_sum(var) * (1.0f / HID) + eps);
#pragma unroll
        for (int S = 0; S < 4; ++S)
#pragma unroll
            for (int r = 0; r < 4; ++r) z[S][r] = (z[S][r] - mean) * rstd * lnw_l[16 * S + r] + lnb_l[16 * S + r];
    }
#pragma unroll
    for (int S = 0; S < 4; ++S)
#pragma unroll
        for (int r = 0; r < 4; ++r) z[S][r] = fmaxf(z[S][r], 0.0f);
}

// the three gates of output-unit tile T from x and h (16 k-steps, six products), and the new hidden state of those units
__device__ __forceinline__ f32x4 r16_gru_tile(const float* wg_l, const float* gb_l, int T, const f32x4* x, const f32x4* hv) {
    f32x4 ar, az, gin, ghn;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        ar[r] = gb_l[16 * T + r]; az[r] = gb_l[HID + 16 * T + r]; gin[r] = gb_l[2 * HID + 16 * T + r]; ghn[r] = gb_l[3 * HID + 16 * T + r];
    }
    float w[6], wn[6];
    {
        const int o = 16 * T;
        w[0] = wg_l[o]; w[1] = wg_l[o + HID]; w[2] = wg_l[o + 2 * HID];
        w[3] = wg_l[o + 3 * HID]; w[4] = wg_l[o + 4 * HID]; w[5] = wg_l[o + 5 * HID];
    }
#pragma unroll
    for (int st = 0; st < 16; ++st) {
        if (st + 1 < 16) {
            const int o = (16 * ((st + 1) >> 2) + ((st + 1) & 3)) * R16_PG + 16 * T;
            wn[0] = wg_l[o]; wn[1] = wg_l[o + HID]; wn[2] = wg_l[o + 2 * HID];
            wn[3] = wg_l[o + 3 * HID]; wn[4] = wg_l[o + 4 * HID]; wn[5] = wg_l[o + 5 * HID];
        }
        const float bx = x[st >> 2][st & 3], bh = hv[st >> 2][st & 3];
        ar = MFMA16(w[0], bx, ar);
        az = MFMA16(w[1], bx, az);
        gin = MFMA16(w[2], bx, gin);
        ar = MFMA16(w[3], bh, ar);
        az = MFMA16(w[4], bh, az);
        ghn = MFMA16(w[5], bh, ghn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 6; ++e) w[e] = wn[e];
    }
    f32x4 hn;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float rg = fast_sigmoid(ar[r]);
        const float zg = fast_sigmoid(az[r]);
        const float ng = fast_tanh(gin[r] + rg * ghn[r]);
        hn[r] = ng + zg * (hv[T][r] - ng);                                  // (1 - z) n + z h
    }
    return hn;
}

__global__ __launch_bounds__(64 * R16_W) void actor_rollout16_kernel(FlexActorArgs a) {
    __shared__ ActorLds16 s;
    ASTAMP(0); ASTAMP_C(0);
    // the slab cursor (inputs in a slab ring): requested here, USED only after the weight loads below are in flight — they
    // do not depend on it, and the cell was written by the launch before this one (a cold scalar load, ~1 us, that used to
    // sit in front of everything)
    const int64_t cur_p = a.cursor ? *a.cursor : 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int od = a.obs_dim, na = a.n_agents, ad = a.act_dim;
    const int ld1 = od + (a.agent_id ? na : 0);
    const int nq = (od + 15) >> 4;                                         // 16-column groups of an observation row
    const int n_tiles = (a.rows + 15) / 16;
    const bool coop = wave >= 4;
    const int cq = wave & 3;                                               // cooperative wavefronts: their unit tile
    // this wavefront's tile in round `rnd`: 5 rnd + wave for wavefronts 0-3, 5 rnd + 4 for the cooperating four
    int rnd = blockIdx.x;
    auto tile_of = [&](int r_) { return 5 * r_ + (coop ? 4 : wave); };
    f32x4 xq[FLEXNET_MAX_OBS / 16];
    __amdgpu_buffer_rsrc_t robs;
    auto load_obs = [&](int tile) {
        const int row = min(tile * 16 + j, a.rows - 1);
        const int xoff = (row * od + 4 * g) * 4;
#pragma unroll
        for (int q = 0; q < FLEXNET_MAX_OBS / 16; ++q)
            xq[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                robs, (q < nq && tile < n_tiles) ? xoff + 64 * q : -1, 0, 0));
    };
    // the previous hidden state of a tile's rows, requested with its observations (one round ahead)
    f32x4 hv[4];
    auto load_hid = [&](int tile) {
        const int row = min(min(tile, n_tiles - 1) * 16 + j, a.rows - 1);
#pragma unroll
        for (int S = 0; S < 4; ++S) {
            const float4 t = *reinterpret_cast<const float4*>(a.hidden_in + (int64_t)row * HID + 16 * S + 4 * g);
            hv[S] = f32x4{t.x, t.y, t.z, t.w};
        }
    };
    // ---- weights -> LDS (transposed); every global read of a thread is issued before its first LDS write.  A thread
    //      takes a 4 x 4 block (four output units x four inputs): four 16-byte reads, one per unit row, and four 16-byte
    //      LDS writes, one per input row of the transposed image — a quarter of the LDS write instructions of a scalar
    //      scatter.  Lanes: eight consecutive unit blocks x eight consecutive input groups per wavefront, so that a read
    //      instruction covers 128 contiguous bytes of each of its rows and the eight lanes of a write cover all 32 banks
    //      (the pitches are multiples of 4 for the A-operand reads: with consecutive lanes on consecutive inputs — how
    //      the 32-row kernel stages its odd-pitch images — a wavefront's scalar stores land on 8 banks, 14.1 k cycles of
    //      staging; with consecutive lanes on consecutive units the reads are 64 separate 16-byte requests, 17.9 k) -------
    //      TWO STAGES: a CU receives ~12 B per cycle when every CU stages at once, so the 156 KB are ~13 k cycles however they
    //      are requested.  fc1 needs only its own 39 KB: those (and the small vectors) are requested first — loads return in
    //      order — and published with a first barrier; the gate weights, requested right behind, arrive while the first
    //      tile's fc1 runs and are published by a second barrier after it.
    const uint64_t rng_seed = a.rng_state ? a.rng_state[0] : 0ull, rng_step = a.rng_state ? a.rng_state[1] : 0ull;
    const bool draws = !a.noise && a.rng_state && 4 * g < ad && (!coop || cq == 0);
    float zr[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int NGB = 2 * (3 * HID / 4) * (HID / 4) / (64 * R16_W);      // 3 blocks per thread over both gate matrices
    static_assert(NGB * 64 * R16_W == 2 * (3 * HID / 4) * (HID / 4), "gate blocks split evenly");
    float4 vg[NGB][4];
    {
        const int q4 = (od + 3) >> 2;
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.fc1_w), 0, HID * ld1 * 4, 0x00027000);
        f32x4 vf[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int e = tid + 64 * R16_W * t, rest = e >> 6;
            const int ub = 8 * (rest & 1) + (e & 7), k4 = 8 * (rest >> 1) + ((e >> 3) & 7);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                vf[t][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    r1, k4 < q4 ? ((4 * ub + i) * ld1 + 4 * k4) * 4 : -1, 0, 0));
        }
        // the small vectors too: every address valid in every thread (clamped index, buffer descriptor), no branch around a
        // load — as separate `if (tid < 64)` blocks behind the big stores they were four more memory round trips
        const int tu = tid & (HID - 1);
        const float s_bih0 = a.b_ih[tu], s_bih1 = a.b_ih[HID + tu], s_bih2 = a.b_ih[2 * HID + tu];
        const float s_bhh0 = a.b_hh[tu], s_bhh1 = a.b_hh[HID + tu], s_bhh2 = a.b_hh[2 * HID + tu];
        const float s_b1 = a.fc1_b[tu];
        const float s_lnw = a.layernorm ? a.ln_w[tu] : 1.0f, s_lnb = a.layernorm ? a.ln_b[tu] : 0.0f;
        static_assert(FLEXNET_MAX_AGENTS * HID == 64 * R16_W, "one id-column element per thread");
        const int id_ag = tid / HID;
        const float s_id = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
            r1, (a.agent_id && id_ag < na) ? (tu * ld1 + od + id_ag) * 4 : -1, 0, 0));
        static_assert(HID * 16 == 2 * 64 * R16_W, "two fc2 elements per thread");
        const int w2k0 = tid >> 4, w2o = tid & 15, w2oc = w2o < ad ? w2o : ad - 1;
        const float s_w2a = a.fc2_w[w2oc * HID + w2k0], s_w2b = a.fc2_w[w2oc * HID + w2k0 + 32];
        const float s_b2 = a.fc2_b[tid < ad ? tid : 0];
        // the first tile's observations, behind fc1's weights in the queue (what fc1 needs first) and in front of the gate
        // weights; this is where the slab cursor is first needed
        if (a.cursor) {
            a.obs += cur_p * a.obs_slab_stride; a.hidden_in += cur_p * a.hid_slab_stride;
            if (a.cursor_out && blockIdx.x == 0 && threadIdx.x == 0) *a.cursor_out = cur_p;
        }
        {
            const int64_t obs_bytes = (int64_t)a.rows * od * 4;
            robs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.obs), 0,
                                                     obs_bytes > 0x7ffffff0ll ? 0x7ffffff0 : (int)obs_bytes, 0x00027000);
        }
        load_obs(tile_of(rnd));
#pragma unroll
        for (int t = 0; t < NGB; ++t) {                                    // the gate weights: requested last, stored after fc1
            const int e = tid + 64 * R16_W * t, rest = e >> 6;
            const int ub = 8 * (rest % 6) + (e & 7), k4 = 8 * ((rest / 6) & 1) + ((e >> 3) & 7);
            const float* src = (rest >= 12 ? a.w_hh : a.w_ih) + (int64_t)(4 * ub) * HID + 4 * k4;
#pragma unroll
            for (int i = 0; i < 4; ++i) vg[t][i] = *reinterpret_cast<const float4*>(src + i * HID);
        }
        // the exploration noise of this wavefront's first tile depends on nothing but (seed, step, row): drawn HERE, with
        // every load in flight and before the first wait — Philox + Box-Muller behind fc2 was 2.8 us of a 22.8 us call, and
        // drawn between the stage-1 stores and their barrier it still delayed the barrier (tools/actor_bench.py)
        if (draws && tile_of(rnd) < n_tiles) actor_noise4(rng_seed, rng_step, (uint32_t)(tile_of(rnd) * 16 + j), (uint32_t)g, zr);
        asm volatile("" : "+v"(zr[0]), "+v"(zr[1]), "+v"(zr[2]), "+v"(zr[3]));        // (the draws stay here)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int e = tid + 64 * R16_W * t, rest = e >> 6;
            const int ub = 8 * (rest & 1) + (e & 7), k4 = 8 * (rest >> 1) + ((e >> 3) & 7);
            if (k4 < q4) {
                float* dst = s.w1t + (4 * k4) * R16_P1 + 4 * ub;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (4 * k4 + c < od)
                        *reinterpret_cast<float4*>(dst + c * R16_P1) = make_float4(vf[t][0][c], vf[t][1][c], vf[t][2][c], vf[t][3][c]);
            }
        }
        if (tid < HID) {
            s.gb[tid] = s_bih0 + s_bhh0;
            s.gb[HID + tid] = s_bih1 + s_bhh1;
            s.gb[2 * HID + tid] = s_bih2;
            s.gb[3 * HID + tid] = s_bhh2;
            s.b1[tid] = s_b1; s.lnw[tid] = s_lnw; s.lnb[tid] = s_lnb;
        }
        s.w1id[tid] = s_id;
        s.w2p[w2k0 * R16_P2 + w2o] = w2o < ad ? s_w2a : 0.0f;
        s.w2p[(w2k0 + 32) * R16_P2 + w2o] = w2o < ad ? s_w2b : 0.0f;
        if (tid < ad) s.b2[tid] = s_b2;
    }
    for (int idx = tid; idx < (16 * nq - od) * HID; idx += 64 * R16_W)     // fc1 runs over 16-column groups: zero rows behind obs_dim
        s.w1t[(od + idx / HID) * R16_P1 + (idx % HID)] = 0.0f;
    if (tid < 2) s.sync[tid] = 0;
    __syncthreads();
    ASTAMP(1); ASTAMP_C(1);

    const float* w1_l = s.w1t + (4 * g) * R16_P1 + j;
    const float* wg_l = s.wg + (4 * g) * R16_PG + j;
    const float* w2_l = s.w2p + (4 * g) * R16_P2 + j;
    const float* gb_l = s.gb + 4 * g;
    const float* b1_l = s.b1 + 4 * g;
    const float* lnw_l = s.lnw + 4 * g;
    const float* lnb_l = s.lnb + 4 * g;
    int passes = 0;                                                        // rendezvous passed so far (cooperating wavefronts)
    f32x4 x[4];                                                            // fc1 output -> GRU input of the current tile
    f32x4 zq = f32x4{0.0f, 0.0f, 0.0f, 0.0f};                              // cooperating wavefronts: fc1 output of their 16 units
    // fc1 of this wavefront's tile of round `rnd` from the observation groups in xq (bias and id column added)
    auto fc1 = [&](int tile) {
        const int ag = min(tile * 16 + j, a.rows - 1) % na;
        const float* w1id_l = s.w1id + ag * HID + 4 * g;
        if (!coop) {
            // ---- fc1, all 64 units: four independent chains ------------------------------------------------------------
#pragma unroll
            for (int T = 0; T < 4; ++T) x[T] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            // the four weights of step s + 1 are requested before the MFMAs of step s (as in the GRU below)
            float w[4], wn[4];
#pragma unroll
            for (int T = 0; T < 4; ++T) w[T] = w1_l[16 * T];
#pragma unroll
            for (int q = 0; q < FLEXNET_MAX_OBS / 16; ++q) {
                if (q < nq) {                                              // wavefront-uniform
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int nx = 4 * q + r + 1;                      // next step: its rows of w1t are zero past obs_dim
                        if (nx < 4 * (FLEXNET_MAX_OBS / 16)) {
#pragma unroll
                            for (int T = 0; T < 4; ++T) wn[T] = w1_l[(16 * (nx >> 2) + (nx & 3)) * R16_P1 + 16 * T];
                        }
                        const float b = xq[q][r];
#pragma unroll
                        for (int T = 0; T < 4; ++T) x[T] = MFMA16(w[T], b, x[T]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int T = 0; T < 4; ++T) w[T] = wn[T];
                    }
                }
            }
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int r = 0; r < 4; ++r) x[T][r] += b1_l[16 * T + r] + w1id_l[16 * T + r];
        } else {
            // ---- fc1, this wavefront's 16 units (one chain: the order the full-tile wavefronts sum in) -------------------
            zq = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int q = 0; q < FLEXNET_MAX_OBS / 16; ++q) {
                if (q < nq) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) zq = MFMA16(w1_l[(16 * q + r) * R16_P1 + 16 * cq], xq[q][r], zq);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) zq[r] += b1_l[16 * cq + r] + w1id_l[16 * cq + r];
        }
    };
    // the first tile's fc1 runs between the two staging stages, while the gate weights are still arriving
    bool peeled = tile_of(rnd) < n_tiles;
    if (peeled) fc1(tile_of(rnd));
    load_hid(tile_of(rnd));                                                // (the observation registers are free now)
    ASTAMP(2); ASTAMP_C(2);
#pragma unroll
    for (int t = 0; t < NGB; ++t) {
        const int e = tid + 64 * R16_W * t, rest = e >> 6;
        const int ub = 8 * (rest % 6) + (e & 7), k4 = 8 * ((rest / 6) & 1) + ((e >> 3) & 7);
        float* dst = s.wg + (4 * k4) * R16_PG + (rest >= 12 ? 3 * HID : 0) + 4 * ub;
        *reinterpret_cast<float4*>(dst) = make_float4(vg[t][0].x, vg[t][1].x, vg[t][2].x, vg[t][3].x);
        *reinterpret_cast<float4*>(dst + R16_PG) = make_float4(vg[t][0].y, vg[t][1].y, vg[t][2].y, vg[t][3].y);
        *reinterpret_cast<float4*>(dst + 2 * R16_PG) = make_float4(vg[t][0].z, vg[t][1].z, vg[t][2].z, vg[t][3].z);
        *reinterpret_cast<float4*>(dst + 3 * R16_PG) = make_float4(vg[t][0].w, vg[t][1].w, vg[t][2].w, vg[t][3].w);
    }
    __syncthreads();
    for (; 5 * rnd < n_tiles; rnd += gridDim.x) {
        const int tile = tile_of(rnd);
        if (tile >= n_tiles) break;                                        // (uniform per wavefront; for the cooperating four: all of them)
        const int r0 = tile * 16;
        const int row = min(r0 + j, a.rows - 1);
        const bool live = r0 + j < a.rows;
        f32x4 hnew[4];
        if (!peeled) fc1(tile);
        peeled = false;
        if (coop) {
            *reinterpret_cast<float4*>(s.xz + j * R16_PX + 16 * cq + 4 * g) = make_float4(zq[0], zq[1], zq[2], zq[3]);
            ++passes;
            r16_rendezvous(&s.sync[0], 4 * passes, lane);
#pragma unroll
            for (int S = 0; S < 4; ++S) {
                const float4 t = *reinterpret_cast<const float4*>(s.xz + j * R16_PX + 16 * S + 4 * g);
                x[S] = f32x4{t.x, t.y, t.z, t.w};
            }
        }
        // the next round's observations go out now and land underneath LayerNorm and the GRU
        if (5 * (rnd + gridDim.x) < n_tiles) load_obs(tile_of(rnd + gridDim.x));
        r16_ln_relu(x, a.layernorm != 0, a.ln_eps, lnw_l, lnb_l);
        ASTAMP(3); ASTAMP_C(3);
        if (!coop) {
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                hnew[T] = r16_gru_tile(wg_l, gb_l, T, x, hv);
                if (live)
                    *reinterpret_cast<float4*>(a.hidden_out + (int64_t)(r0 + j) * HID + 16 * T + 4 * g) =
                        make_float4(hnew[T][0], hnew[T][1], hnew[T][2], hnew[T][3]);
            }
            if (5 * (rnd + gridDim.x) < n_tiles) load_hid(tile_of(rnd + gridDim.x));     // (hv is dead: the next round's)
        } else {
            f32x4 hq;
            // (the tile index must be a compile-time constant of the inlined GRU body: one copy per unit tile)
            switch (cq) {
                case 0: hq = r16_gru_tile(wg_l, gb_l, 0, x, hv); break;
                case 1: hq = r16_gru_tile(wg_l, gb_l, 1, x, hv); break;
                case 2: hq = r16_gru_tile(wg_l, gb_l, 2, x, hv); break;
                default: hq = r16_gru_tile(wg_l, gb_l, 3, x, hv); break;
            }
            if (5 * (rnd + gridDim.x) < n_tiles) load_hid(tile_of(rnd + gridDim.x));
            if (live)
                *reinterpret_cast<float4*>(a.hidden_out + (int64_t)(r0 + j) * HID + 16 * cq + 4 * g) = make_float4(hq[0], hq[1], hq[2], hq[3]);
            *reinterpret_cast<float4*>(s.xh + j * R16_PX + 16 * cq + 4 * g) = make_float4(hq[0], hq[1], hq[2], hq[3]);
            r16_rendezvous(&s.sync[1], 4 * passes, lane);
            if (cq != 0) continue;                                         // fc2 and the action epilogue: wavefront 4 alone
#pragma unroll
            for (int S = 0; S < 4; ++S) {
                const float4 t = *reinterpret_cast<const float4*>(s.xh + j * R16_PX + 16 * S + 4 * g);
                hnew[S] = f32x4{t.x, t.y, t.z, t.w};
            }
        }
        ASTAMP(4); ASTAMP_C(4);
        // ---- fc2 (rnn_agent.py:32): means[k][row], k padded to 16; lane (row, g) ends with actions 4 g .. 4 g + 3 ---------
        f32x4 mo = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int st = 0; st < 16; ++st)
            mo = MFMA16(w2_l[(16 * (st >> 2) + (st & 3)) * R16_P2], hnew[st >> 2][st & 3], mo);
        if (live) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 4 * g + r;
                if (k < ad) {
                    const float o = mo[r] + s.b2[k];
                    const int64_t at = (int64_t)(r0 + j) * ad + k;
                    a.means[at] = o;
                    if (a.action) {                                           // util.py:57-64, 125-128
                        const float act = fast_tanh(o + a.std * (a.noise ? a.noise[at] : zr[r]));     // (~1 ulp, as the gates)
                        a.action[at] = act;
                        a.env_action[at] = 0.5f * (fminf(fmaxf(act, a.action_low), a.action_high) + 1.0f) * (a.action_high - a.action_low) + a.action_low;
                    }
                }
            }
        }
        ASTAMP(5); ASTAMP_C(5);
        // (a block with more than one round: the next tile's draws, now)
        if (draws && tile_of(rnd + gridDim.x) < n_tiles && 5 * (rnd + gridDim.x) < n_tiles)
            actor_noise4(rng_seed, rng_step, (uint32_t)(tile_of(rnd + gridDim.x) * 16 + j), (uint32_t)g, zr);
    }
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
    if ((int64_t)a->rows * a->obs_dim * 4 >= 0x7ffffff0ll) return FLEXNET_EUNSUPPORTED;   // observations are addressed with 32-bit byte offsets
    if (a->cursor && (a->obs_slab_stride < (int64_t)a->rows * a->obs_dim || a->hid_slab_stride < 0)) return FLEXNET_EINVAL;
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
