// actor_r16.h — the policy network on 16-row tiles (v_mfma_f32_16x16x4_f32), five tiles per CU: shared by csrc/actor.hip
// (flexnet_actor_forward at inference sizes) and csrc/flexenv.hip (flexenv_rollout_burst: policy and environment step of a
// block's 16 environments in ONE persistent launch, the weights staged once per burst).  Boundary: include/flexnet.h.
#ifndef ACTOR_R16_H
#define ACTOR_R16_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flexnet.h"

#ifndef HID
#define HID FLEXNET_HID
#endif
#ifndef ASTAMP
#define ASTAMP(k) do { } while (0)
#define ASTAMP_C(k) do { } while (0)
#endif

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


// float offset of row r's observation behind FlexActorArgs::obs: [rows, obs_dim], or — obs_pushed — the window of the
// environment's mirror ring (include/flexnet.h)
__device__ __forceinline__ int actor_obs_off(const FlexActorArgs& a, int row) {
    if (!a.obs_pushed) return row * a.obs_dim;
    const int c = a.obs_pushed[(int64_t)(row / a.n_agents) * a.obs_pushed_stride];
    const int s = c > 0 ? (c - 1) % a.obs_slots : 0;
    return row * a.obs_row_stride + (s + 1) * a.obs_slot_w;
}
__device__ __forceinline__ int64_t actor_obs_bytes(const FlexActorArgs& a) {
    return (int64_t)a.rows * (a.obs_pushed ? a.obs_row_stride : a.obs_dim) * 4;
}

// fc1 of one 16-row tile, all 64 units (four independent chains over the observation's 16-column groups in xq), bias and
// the agent's id column added: rnn_agent.py:26 with model.py:105-108's one-hot columns folded into w1id
__device__ __forceinline__ void r16_fc1_tile(const float* w1_l, const float* b1_l, const float* w1id_l, const f32x4* xq, int nq,
                                             f32x4* x) {
#pragma unroll
    for (int T = 0; T < 4; ++T) x[T] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // the four weights of step s + 1 are requested before the MFMAs of step s (as in the GRU)
    float w[4], wn[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) w[T] = w1_l[16 * T];
#pragma unroll
    for (int q = 0; q < FLEXNET_MAX_OBS / 16; ++q) {
        if (q < nq) {                                                      // wavefront-uniform
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nx = 4 * q + r + 1;                              // next step: its rows of w1t are zero past obs_dim
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
}

// The stand-alone kernel's body (flexnet_actor_forward at rollout sizes): rounds of five tiles over the grid.
__device__ __forceinline__ void actor_r16_body(FlexActorArgs a, ActorLds16& s) {
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
    constexpr int tpr = 5;
    auto tile_of = [&](int r_) { return tpr * r_ + (coop ? 4 : wave); };
    f32x4 xq[FLEXNET_MAX_OBS / 16];
    __amdgpu_buffer_rsrc_t robs;
    auto load_obs = [&](int tile) {
        const int row = min(tile * 16 + j, a.rows - 1);
        const int xoff = (actor_obs_off(a, row) + 4 * g) * 4;
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
            if (!a.obs_pushed) a.obs += cur_p * a.obs_slab_stride;
            a.hidden_in += cur_p * a.hid_slab_stride;
            if (a.cursor_out && blockIdx.x == 0 && threadIdx.x == 0) *a.cursor_out = cur_p;
        }
        {
            const int64_t obs_bytes = actor_obs_bytes(a);
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
            r16_fc1_tile(w1_l, b1_l, w1id_l, xq, nq, x);
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
    for (rnd = blockIdx.x; tpr * rnd < n_tiles; rnd += gridDim.x) {
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
        if (tpr * (rnd + gridDim.x) < n_tiles) load_obs(tile_of(rnd + gridDim.x));
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
            if (tpr * (rnd + gridDim.x) < n_tiles) load_hid(tile_of(rnd + gridDim.x));     // (hv is dead: the next round's)
        } else {
            f32x4 hq;
            // (the tile index must be a compile-time constant of the inlined GRU body: one copy per unit tile)
            switch (cq) {
                case 0: hq = r16_gru_tile(wg_l, gb_l, 0, x, hv); break;
                case 1: hq = r16_gru_tile(wg_l, gb_l, 1, x, hv); break;
                case 2: hq = r16_gru_tile(wg_l, gb_l, 2, x, hv); break;
                default: hq = r16_gru_tile(wg_l, gb_l, 3, x, hv); break;
            }
            if (tpr * (rnd + gridDim.x) < n_tiles) load_hid(tile_of(rnd + gridDim.x));
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
        if (draws && tile_of(rnd + gridDim.x) < n_tiles && tpr * (rnd + gridDim.x) < n_tiles)
            actor_noise4(rng_seed, rng_step, (uint32_t)(tile_of(rnd + gridDim.x) * 16 + j), (uint32_t)g, zr);
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// The policy inside the rollout burst (csrc/flexenv.hip, flexenv_rollout_burst).  A block owns sixteen environments for the
// whole burst and is TWO independent groups of four wavefronts — group q: environments 16 b + 8 q .. + 7, their 8 n_agents
// <= 40 policy rows as up to three 16-row tiles (wavefront w < 3 of the group: rows 16 w ..), wavefront 3 of the group draws
// the NEXT step's exploration noise for those tiles into LDS meanwhile (Philox + Box-Muller cost a tile's wavefront 6.7 k
// cycles per step when it drew its own).  Each group alternates policy and environment step (env_step: its four wavefronts x
// two environments) on its own, meeting only its own four wavefronts in between: the two groups free-run, so that most of
// the time one wavefront of a SIMD is in the policy (matrix pipe, LDS) and the other in the environment step (fp64 VALU,
// memory latency) instead of both competing for the same pipe in lockstep.  Same arithmetic per row as actor_r16_body: a
// row's result does not depend on the tile it sits in (tests/test_actor_gpu.py), so the burst equals the stand-alone
// launches bit for bit (tests/test_rollout_gpu.py).
// ------------------------------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) ActorLds16B {
    float w1t[FLEXNET_MAX_OBS * R16_P1];
    float wg[HID * R16_PG];
    float w2p[HID * R16_P2];
    float b1[HID], lnw[HID], lnb[HID];
    float w1id[FLEXNET_MAX_AGENTS * HID];
    float gb[4 * HID];
    float b2[FLEXNET_MAX_ACT];
    float noise[2][2][3][16][FLEXNET_MAX_ACT];   // [step parity][group][tile][row][action]
    int gsync[2];                                // arrivals of each group's wavefronts (monotonic)
};

// the LDS image actor_r16_body stages (same values in the same places), without its choreography: once per burst
__device__ __forceinline__ void r16_stage_simple(const FlexActorArgs& a, ActorLds16B& s, int tid) {
    const int od = a.obs_dim, na = a.n_agents, ad = a.act_dim;
    const int ld1 = od + (a.agent_id ? na : 0);
    const int nq = (od + 15) >> 4;
    for (int idx = tid; idx < HID * ld1; idx += 64 * R16_W) {
        const int u = idx / ld1, k = idx - u * ld1;
        if (k < od) s.w1t[k * R16_P1 + u] = a.fc1_w[idx];
    }
    for (int idx = tid; idx < (16 * nq - od) * HID; idx += 64 * R16_W) s.w1t[(od + idx / HID) * R16_P1 + (idx % HID)] = 0.0f;
    for (int idx = tid; idx < 3 * HID * HID; idx += 64 * R16_W) {
        const int u = idx / HID, k = idx % HID;
        s.wg[k * R16_PG + u] = a.w_ih[idx];
        s.wg[k * R16_PG + 3 * HID + u] = a.w_hh[idx];
    }
    for (int idx = tid; idx < 16 * HID; idx += 64 * R16_W) {
        const int o = idx / HID, k = idx % HID;
        s.w2p[k * R16_P2 + o] = o < ad ? a.fc2_w[o * HID + k] : 0.0f;
    }
    if (tid < HID) {
        s.gb[tid] = a.b_ih[tid] + a.b_hh[tid];
        s.gb[HID + tid] = a.b_ih[HID + tid] + a.b_hh[HID + tid];
        s.gb[2 * HID + tid] = a.b_ih[2 * HID + tid];
        s.gb[3 * HID + tid] = a.b_hh[2 * HID + tid];
        s.b1[tid] = a.fc1_b[tid];
        s.lnw[tid] = a.layernorm ? a.ln_w[tid] : 1.0f;
        s.lnb[tid] = a.layernorm ? a.ln_b[tid] : 0.0f;
    }
    static_assert(FLEXNET_MAX_AGENTS * HID == 64 * R16_W, "one id-column element per thread");
    {
        const int ag = tid / HID, u = tid % HID;
        s.w1id[tid] = (a.agent_id && ag < na) ? a.fc1_w[u * ld1 + od + ag] : 0.0f;
    }
    if (tid < ad) s.b2[tid] = a.fc2_b[tid];
}

template <typename EnvStep, typename Mark>
__device__ __forceinline__ void actor_r16_burst(const FlexActorArgs& a, ActorLds16B& s, const int n_steps, EnvStep&& env_step,
                                                Mark&& mark) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int grp = wave >> 2, w = wave & 3;
    const int od = a.obs_dim, na = a.n_agents, ad = a.act_dim;
    const int nq = (od + 15) >> 4;
    const uint64_t rng_seed = a.rng_state[0], rng_step = a.rng_state[1];
    // this group's rows, this wavefront's tile
    const int grow0 = (16 * (int)blockIdx.x + 8 * grp) * na;
    const int grows = min(8 * na, a.rows - grow0);                         // (<= 0: a tail block's empty group)
    const bool has_tile = w < 3 && 16 * w < grows;
    const int r0 = grow0 + 16 * w;
    const bool live = has_tile && 16 * w + j < grows;
    const int row = live ? r0 + j : max(min(r0 + j, grow0 + grows - 1), 0);   // (idle lanes compute on a valid row, store nothing)
    // wavefront 3: the noise of `for_step` for the group's tiles — lane (tile, row) = (lane >> 4, lane & 15), all action groups
    auto draw = [&](uint64_t for_step) {
        const int tt = lane >> 4;
        if (tt < 3 && 16 * tt + j < grows) {
            float* dst = s.noise[for_step & 1][grp][tt][j];
#pragma unroll
            for (int q = 0; q < FLEXNET_MAX_ACT / 4; ++q) {
                if (4 * q < ad) {
                    float z[4];
                    actor_noise4(rng_seed, for_step, (uint32_t)(grow0 + 16 * tt + j), (uint32_t)q, z);
                    *reinterpret_cast<float4*>(dst + 4 * q) = make_float4(z[0], z[1], z[2], z[3]);
                }
            }
        }
    };
    r16_stage_simple(a, s, tid);
    if (w == 3) draw(rng_step);
    if (tid < 2) s.gsync[tid] = 0;
    __syncthreads();

    const float* w1_l = s.w1t + (4 * g) * R16_P1 + j;
    const float* wg_l = s.wg + (4 * g) * R16_PG + j;
    const float* w2_l = s.w2p + (4 * g) * R16_P2 + j;
    const float* gb_l = s.gb + 4 * g;
    const float* b1_l = s.b1 + 4 * g;
    const float* lnw_l = s.lnw + 4 * g;
    const float* lnb_l = s.lnb + 4 * g;
    const float* w1id_l = s.w1id + (row % na) * HID + 4 * g;
    int64_t slab = *a.cursor;
    int meets = 0;
    // (Both groups start together and drift apart on their own — their steps take different times.  Holding group 1 back
    //  by half a period at the start was measured: the same 31.6 us per step in a long burst, 0.6 us per step more in a
    //  16-step one.)
    for (int step = 0; step < n_steps; ++step) {
        mark(12, step);
        if (has_tile) {
            const float* hid_p = a.hidden_in + slab * a.hid_slab_stride + (int64_t)row * HID + 4 * g;
            f32x4 xq[FLEXNET_MAX_OBS / 16], hv[4], x[4], hnew[4];
            {
                // (the loads of actor_r16_body: whole-slab descriptor, a 16-column group behind obs_dim reads what follows the
                //  row — against zero weights — or zero behind the slab)
                // (ring of stacked observations: slab `slab`; in place — obs_pushed — the environment's own history, whose
                //  push count this block's environment step left a rendezvous ago)
                const int64_t obs_bytes = actor_obs_bytes(a);
                const __amdgpu_buffer_rsrc_t robs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(a.obs + (a.obs_pushed ? 0 : slab * a.obs_slab_stride)), 0,
                    obs_bytes > 0x7ffffff0ll ? 0x7ffffff0 : (int)obs_bytes, 0x00027000);
                const int xoff = (actor_obs_off(a, row) + 4 * g) * 4;
#pragma unroll
                for (int q = 0; q < FLEXNET_MAX_OBS / 16; ++q)
                    xq[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(robs, q < nq ? xoff + 64 * q : -1, 0, 0));
            }
#pragma unroll
            for (int S = 0; S < 4; ++S) {
                const float4 t = *reinterpret_cast<const float4*>(hid_p + 16 * S);
                hv[S] = f32x4{t.x, t.y, t.z, t.w};
            }
            r16_fc1_tile(w1_l, b1_l, w1id_l, xq, nq, x);
            r16_ln_relu(x, a.layernorm != 0, a.ln_eps, lnw_l, lnb_l);
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                hnew[T] = r16_gru_tile(wg_l, gb_l, T, x, hv);
                if (live)
                    *reinterpret_cast<float4*>(a.hidden_out + (int64_t)row * HID + 16 * T + 4 * g) =
                        make_float4(hnew[T][0], hnew[T][1], hnew[T][2], hnew[T][3]);
            }
            f32x4 mo = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int st = 0; st < 16; ++st)
                mo = MFMA16(w2_l[(16 * (st >> 2) + (st & 3)) * R16_P2], hnew[st >> 2][st & 3], mo);
            if (live && 4 * g < ad) {
                const float4 zv = *reinterpret_cast<const float4*>(&s.noise[(rng_step + step) & 1][grp][w][j][4 * g]);
                const float zr[4] = {zv.x, zv.y, zv.z, zv.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 4 * g + r;
                    if (k < ad) {
                        const float o = mo[r] + s.b2[k];
                        const int64_t at = (int64_t)row * ad + k;
                        a.means[at] = o;
                        const float act = fast_tanh(o + a.std * zr[r]);                       // util.py:57-64, 125-128
                        a.action[at] = act;
                        a.env_action[at] = 0.5f * (fminf(fmaxf(act, a.action_low), a.action_high) + 1.0f) * (a.action_high - a.action_low) + a.action_low;
                    }
                }
            }
        } else if (w == 3 && step + 1 < n_steps) {
            draw(rng_step + step + 1);                                     // (the other parity: read in the next policy phase)
        }
        // policy outputs (global memory: env action, action, new hidden state) and the noise (LDS) -> the group's other
        // wavefronts: release / acquire at work-group scope on the group's counter (one CU, one vector L1)
        mark(8, step);
        ++meets;
        r16_rendezvous(&s.gsync[grp], 4 * meets, lane);
        mark(9, step);
        env_step(slab);
        mark(10, step);
        // ... and the step's outputs (observation, masked hidden state: slab + 1) -> the next policy phase
        ++meets;
        r16_rendezvous(&s.gsync[grp], 4 * meets, lane);
        mark(11, step);
        slab = slab + 1 >= a.ring_slabs ? 0 : slab + 1;
    }
}

#endif
