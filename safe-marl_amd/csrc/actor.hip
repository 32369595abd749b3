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
    for (int tile = blockIdx.x * AW + wave; tile < n_tiles; tile += gridDim.x * AW) {
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

extern "C" int flexnet_actor_forward(const FlexActorArgs* a, void* stream) {
    if (!a || a->rows < 0) return FLEXNET_EINVAL;
    if (a->rows == 0) return FLEXNET_OK;
    if (!a->obs || !a->hidden_in || !a->fc1_w || !a->fc1_b || !a->w_ih || !a->w_hh || !a->b_ih || !a->b_hh || !a->fc2_w ||
        !a->fc2_b || !a->means || !a->hidden_out || (a->layernorm && (!a->ln_w || !a->ln_b)) ||
        (a->noise && (!a->action || !a->env_action)))
        return FLEXNET_EINVAL;
    if (a->obs_dim < 1 || a->obs_dim > FLEXNET_MAX_OBS || a->n_agents < 1 || a->n_agents > FLEXNET_MAX_AGENTS ||
        a->act_dim < 1 || a->act_dim > FLEXNET_MAX_ACT)
        return FLEXNET_EUNSUPPORTED;
    static int cus = 0;                                   // one block per CU owns that CU's LDS
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
            return FLEXNET_EHIP;
        cus = n;
    }
    const int tiles = (a->rows + RT - 1) / RT;
    const int blocks = (tiles + AW - 1) / AW < cus ? (tiles + AW - 1) / AW : cus;
    hipLaunchKernelGGL(actor_forward_kernel, dim3(blocks), dim3(64 * AW), 0, (hipStream_t)stream, *a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fprintf(stderr, "[flexnet] actor_forward launch failed: %s\n", hipGetErrorString(e));
        return FLEXNET_EHIP;
    }
    return FLEXNET_OK;
}
