// rollout.hip — replay bookkeeping of the vectorised rollout (gfx950): one vector step's transition into the slab ring at
// a device-side cursor, hidden-state hand-over and statistics in one launch (rollout_pack_kernel), and the replay-window
// refresh as one multi-job row copy (gather_rows_kernel).
// Boundary: include/flexnet.h (FlexRolloutPackArgs, FlexGatherArgs).  Reference: madrl/models/model.py:230-262,
// utils/replay_buffer.py:14-27.
// Pure data movement, HBM-bound.  Every observation is stored once (next_state of slab k = obs of slab k + 1), so a step
// moves obs_next + hid_new in and ring + hand-over out: 4 096 envs x (2 880 + 1 280 read, 2 880 + 2 x 1 280 + 112 written) =
// 39 MB, against 154 MB for the round-1 layout (packed record with state AND next_state, then a ring copy).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flexnet.h"
#include "flex_td.h"
#include "window_refresh.h"

#define PACK_THREADS 256
#define PACK_ENVS 4                // environments per copy block: all their loads are in flight before the first store
#define PACK_STATS 10               // statistics blocks at the head of the grid

typedef float pack_f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(PACK_THREADS) void rollout_pack_kernel(FlexRolloutPackArgs a) {
    const int tid = threadIdx.x;
    // Two cursor cells, each written by ONE kernel of the step and read only by others: cursor[0] = slab the policy reads
    // (written here, at the end, by one lane; read by the policy and the environment kernel), cursor[1] = slab the step
    // fills next (advanced by the environment's step kernel BEFORE this launch when cursor_stepped, read here).  Without
    // an environment counter this launch reads cursor[0] and rollout_cursor_kernel advances both cells AFTER it.
    int64_t cur, nxt;
    if (a.cursor_stepped) { nxt = a.cursor[1]; cur = nxt == 0 ? a.slabs - 1 : nxt - 1; }
    else { cur = a.cursor[0]; nxt = cur + 1 >= a.slabs ? 0 : cur + 1; }
    if (blockIdx.x < PACK_STATS) {
        // the first ten blocks own one statistic each (info columns, reward, failures): a block reduction over all
        // environments in a fixed order and one plain += — no atomics on the sums, bit-reproducible
        __shared__ double red[PACK_THREADS];
        const int q = blockIdx.x;
        const bool is_info = q < 8;
        const bool live = is_info ? (q < a.info_w && a.info && a.info_sum) : (q == 8 ? true : (a.failed && a.fail_sum));
        if (live) {
            // eight loads in flight per thread (the statistics blocks are the launch's long pole otherwise: sixteen
            // dependent round trips at 4096 environments), summed in a fixed order
            double v = 0.0;
            for (int e0 = tid; e0 < a.n_envs; e0 += 8 * PACK_THREADS) {
                double x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + u * PACK_THREADS;
                    const int ec = e < a.n_envs ? e : a.n_envs - 1;
                    const double t = is_info ? a.info[(int64_t)ec * a.info_w + q] : (q == 8 ? a.reward[ec] : (a.failed[ec] ? 1.0 : 0.0));
                    x[u] = e < a.n_envs ? t : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) v += x[u];
            }
            red[tid] = v;
            __syncthreads();
            for (int sft = PACK_THREADS / 2; sft > 0; sft >>= 1) {
                if (tid < sft) red[tid] += red[tid + sft];
                __syncthreads();
            }
            if (tid == 0) {
                if (is_info) a.info_sum[q] += red[0];
                else if (q == 8) {                                    // this block always runs
                    *a.rew_sum += red[0];
                    if (a.rng_state) a.rng_state[1] += 1;
                    if (a.cursor_stepped) a.cursor[0] = nxt;          // no block of this launch reads cursor[0] in this mode
                }
                else *a.fail_sum += red[0];
            }
        }
    } else {
        const int no = a.n_agents * a.obs_dim, na = a.n_agents * a.act_dim, nh = a.n_agents * FLEXNET_HID;
        const int e0 = (blockIdx.x - PACK_STATS) * PACK_ENVS;
        const int no4 = no >> 2, nh4 = nh >> 2;
        constexpr int MO = (FLEXNET_MAX_AGENTS * FLEXNET_MAX_OBS / 4 + PACK_THREADS - 1) / PACK_THREADS;      // 2
        constexpr int MH = (FLEXNET_MAX_AGENTS * FLEXNET_HID / 4 + PACK_THREADS - 1) / PACK_THREADS;          // 1
        pack_f4 ob[PACK_ENVS][MO], hb[PACK_ENVS][MH];
        float keep[PACK_ENVS];
        // loads first, then stores (hid_state may alias hid_new): 16-byte units, n * obs_dim and 64 are multiples of 4
#pragma unroll
        for (int j = 0; j < PACK_ENVS; ++j) {
            const int e = e0 + j < a.n_envs ? e0 + j : a.n_envs - 1;          // clamped: loads stay in bounds, stores are skipped
            keep[j] = a.done[e] ? 0.0f : 1.0f;
            const pack_f4* on = reinterpret_cast<const pack_f4*>(a.obs_next + (int64_t)e * no);
            const pack_f4* hn = reinterpret_cast<const pack_f4*>(a.hid_new + (int64_t)e * nh);
            if (a.obs_next) {                                             // NULL: the environment kernel wrote the slab itself
#pragma unroll
                for (int t = 0; t < MO; ++t) { const int i = tid + PACK_THREADS * t; if (i < no4) ob[j][t] = on[i]; }
            }
#pragma unroll
            for (int t = 0; t < MH; ++t) { const int i = tid + PACK_THREADS * t; if (i < nh4) hb[j][t] = hn[i]; }
        }
#pragma unroll
        for (int j = 0; j < PACK_ENVS; ++j) {
            const int e = e0 + j;
            if (e >= a.n_envs) break;
            pack_f4* o_ring = reinterpret_cast<pack_f4*>(a.obs_ring + (nxt * a.n_envs + e) * (int64_t)no);
            pack_f4* h_ring = reinterpret_cast<pack_f4*>(a.hid_ring + (nxt * a.n_envs + e) * (int64_t)nh);
            pack_f4* h_state = reinterpret_cast<pack_f4*>(a.hid_state + (int64_t)e * nh);
            if (a.obs_next) {
#pragma unroll
                for (int t = 0; t < MO; ++t) {
                    const int i = tid + PACK_THREADS * t;
                    if (i < no4) __builtin_nontemporal_store(ob[j][t], &o_ring[i]);    // model.py:236,262: next_state = the next state
                }
            }
#pragma unroll
            for (int t = 0; t < MH; ++t) {
                const int i = tid + PACK_THREADS * t;
                if (i < nh4) {                                                          // fresh hidden state after a terminal step
                    const pack_f4 h = hb[j][t] * keep[j];
                    h_ring[i] = h;                     // read back by the next policy launch when it takes its input from the ring
                    if (a.hid_state) h_state[i] = h;
                }
            }
            float* sm = a.small_ring + (cur * a.n_envs + e) * (int64_t)a.small_w;
            for (int i = tid; i < na; i += PACK_THREADS) sm[i] = a.action[(int64_t)e * na + i];     // model.py:232
            if (tid < a.n_agents) sm[na + tid] = (float)a.reward[e];                                // model.py:235: one reward, n copies
            if (tid == 0) { sm[na + a.n_agents] = 1.0f - keep[j]; sm[na + a.n_agents + 1] = 1.0f - keep[j]; }
        }
    }
}

// cursor[0] += 1 as a launch of its own: every block of rollout_pack_kernel reads the cursor, so the increment has to wait
// for all of them.  (A last-block ticket inside the pack kernel was tried first: 2 058 device-scope atomics on one address
// — executed memory-side, past the per-XCD L2s — took longer than the 39 MB of copies: 34.8 us per launch.)
__global__ void rollout_cursor_kernel(int64_t* cursor, int slabs) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int64_t nxt = cursor[0] + 1 >= slabs ? 0 : cursor[0] + 1;
        cursor[0] = nxt;
        cursor[1] = nxt;
    }
}

extern "C" int flexnet_rollout_pack(const FlexRolloutPackArgs* a, void* stream) {
    if (!a || a->n_envs < 0) return FLEXNET_EINVAL;
    if (a->n_envs == 0) return FLEXNET_OK;
    if (!a->action || !a->reward || !a->done || !a->hid_new || !a->obs_ring || !a->hid_ring ||
        !a->small_ring || !a->cursor || !a->rew_sum || a->info_w < 0 || a->info_w > 8 || a->n_agents < 1 ||
        a->n_agents > FLEXNET_MAX_AGENTS || a->obs_dim < 1 || a->obs_dim > FLEXNET_MAX_OBS || a->act_dim < 1 ||
        a->act_dim > FLEXNET_MAX_ACT || a->slabs < 2 || a->small_w < a->n_agents * a->act_dim + a->n_agents + 2)
        return FLEXNET_EINVAL;
    if (((a->n_agents * a->obs_dim) & 3) != 0) return FLEXNET_EUNSUPPORTED;       // 16-byte units (obs_dim = 6 * history)
    const uintptr_t align = reinterpret_cast<uintptr_t>(a->obs_next) | reinterpret_cast<uintptr_t>(a->hid_new) |
                            reinterpret_cast<uintptr_t>(a->obs_ring) | reinterpret_cast<uintptr_t>(a->hid_ring) |
                            reinterpret_cast<uintptr_t>(a->hid_state);
    if (align & 15) return FLEXNET_EINVAL;
    const int blocks = (a->n_envs + PACK_ENVS - 1) / PACK_ENVS + PACK_STATS;
    hipLaunchKernelGGL(rollout_pack_kernel, dim3(blocks), dim3(PACK_THREADS), 0, (hipStream_t)stream, *a);
    if (!a->cursor_stepped) hipLaunchKernelGGL(rollout_cursor_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a->cursor, a->slabs);
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}

// ---- replay-window refresh ------------------------------------------------------------------------------------------------
// Job j copies rows[j] rows of width[j] floats from src (row pitch src_stride) to dst (row pitch dst_stride).  The grid is
// cut over the jobs in proportion to their size (host side); inside a job a block walks 16-byte units when pitch, width and
// base allow, single floats otherwise (the small record's reward / done columns).
// ---- agent-summed exploration (flexnet_agent_sum_explore): one thread per (environment, action component) ---------------
__global__ __launch_bounds__(256) void agent_sum_explore_kernel(FlexAgentSumArgs a) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= a.n_envs * a.act_dim) return;
    const int e = tid / a.act_dim, k = tid - e * a.act_dim;
    const int n = a.n_agents, ad = a.act_dim;
    const float* m = a.means + (int64_t)e * n * ad + k;
    float s = m[0];
    for (int i = 1; i < n; ++i) s = __fadd_rn(s, m[i * ad]);                            // ((m0 + m1) + m2) + ...
    // eps == NULL: no exploration — the agent-summed mean itself goes to every agent (util.py:75-77 behind matd3.py:94-96)
    const float y = a.eps ? tanhf(__fadd_rn(s, __fmul_rn(a.eps[tid], a.std[k]))) : s;  // tanh(loc + eps * scale)
    const float span = a.act_high - a.act_low;
    const float c = fminf(fmaxf(y, a.act_low), a.act_high);
    const float ev = __fadd_rn(__fmul_rn(__fmul_rn(0.5f, __fadd_rn(c, 1.0f)), span), a.act_low);
    float* ao = a.action + (int64_t)e * n * ad + k;
    for (int i = 0; i < n; ++i) ao[i * ad] = y;
    if (a.env_action) {
        float* eo = a.env_action + (int64_t)e * n * ad + k;
        for (int i = 0; i < n; ++i) eo[i * ad] = ev;
    }
}

extern "C" int flexnet_agent_sum_explore(const FlexAgentSumArgs* a, void* stream) {
    if (!a || a->n_envs < 0 || a->n_agents < 1 || a->act_dim < 1 || !a->means || (a->eps && !a->std) || !a->action ||
        !(a->act_high >= a->act_low))
        return FLEXNET_EINVAL;
    if (a->n_envs == 0) return FLEXNET_OK;
    const int64_t tot = (int64_t)a->n_envs * a->act_dim;
    if (tot > 0x7fffffff) return FLEXNET_EUNSUPPORTED;
    hipLaunchKernelGGL(agent_sum_explore_kernel, dim3((int)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a);
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}

// ---- stacked observations of a replay window from the row ring (flexnet_gather_window) -------------------------------------
// A block owns 16 consecutive (env, agent) pairs for a run of up to 16 consecutive slabs of the window: it stages the
// 16 + history - 1 slabs of 32-byte records those outputs reach into in LDS ONCE (each record is used by up to `history`
// outputs: 2.4 records read per stacked row written instead of 24) and writes, per slab, the 16 pairs' stacked
// observations — 16 x 576 B contiguous in the output — as coalesced 8-byte stores.  Round-4 history of this launch at the
// update batch (36 864 rows): one thread per 24-byte chunk with lanes on consecutive history slots (64 different slabs per
// wavefront load) 36 us; lanes on consecutive pairs + an LDS transpose, every record still read `history` times from
// cache, 31.8 us (the reads cross the fabric: 141 MB next to 106 MB of stores); time-blocked as here: see DESIGN.md §4.8.
#define WINDOW_THREADS 256
#define WINDOW_PAIRS 16
#define WINDOW_SLABS 16
#define WINDOW_MAX_H 32
#define WINDOW_PITCH (WINDOW_PAIRS * 8 + 2)          // floats per staged slab: + 2 keeps the history slots of a store on different banks
typedef float win_f2 __attribute__((ext_vector_type(2)));
// HC: history as a compile-time constant (24: the reference's), 0: a.history at run time.  All index arithmetic of the
// store loop is hoisted out of it (the first time-blocked version divided by run-time values per 8-byte store and was
// instruction-bound: 48.9 us).
template <int HC>
__global__ __launch_bounds__(WINDOW_THREADS) void gather_window_kernel(FlexWindowArgs a) {
    __shared__ float rec[(WINDOW_SLABS + WINDOW_MAX_H - 1) * WINDOW_PITCH];
    __shared__ int64_t obase[WINDOW_PAIRS];          // (row * n + agent) of pair p at the block's first slab
    __shared__ int penv[WINDOW_PAIRS];
    const int H = HC ? HC : a.history, n = a.n_agents, tid = threadIdx.x;
    const int64_t npairs = (int64_t)a.n_envs * n;                      // (env, agent) pairs of one slab
    const int64_t q0 = (int64_t)blockIdx.x * WINDOW_PAIRS;
    const int np = npairs - q0 < WINDOW_PAIRS ? (int)(npairs - q0) : WINDOW_PAIRS;
    const int64_t sl_lo = a.first_slot / a.n_envs, sl_hi = (a.first_slot + a.rows - 1) / a.n_envs;
    const int64_t sl0 = sl_lo + (int64_t)blockIdx.y * WINDOW_SLABS;     // first slab (counter) of this block's run
    const int nt = sl_hi - sl0 + 1 < WINDOW_SLABS ? (int)(sl_hi - sl0 + 1) : WINDOW_SLABS;
    if (tid < np) {
        const int64_t q = q0 + tid, env = q / n;
        penv[tid] = (int)env;
        obase[tid] = (sl0 * a.n_envs + env - a.first_slot) * n + (q - env * n);
    }
    // stage slabs sl0 - (H - 1) .. sl0 + nt - 1: 16 pairs x 32 B contiguous per slab, as 16-byte loads
    const int ns = nt + H - 1, per_stage = np * 2;
    for (int s = tid / 32, r = tid % 32; s < ns; s += WINDOW_THREADS / 32) {
        if (r < per_stage) {
            int64_t sl = (sl0 - (H - 1) + s) % a.slabs;
            sl = sl < 0 ? sl + a.slabs : sl;
            const pack_f4 v = *reinterpret_cast<const pack_f4*>(a.row_ring + (sl * npairs + q0) * 8 + 4 * r);
            float* d = rec + s * WINDOW_PITCH + 4 * r;
            *reinterpret_cast<win_f2*>(d) = win_f2{v.x, v.y};
            *reinterpret_cast<win_f2*>(d + 2) = win_f2{v.z, v.w};
        }
    }
    __syncthreads();
    const int w2 = 3 * H;                                              // 8-byte units of one stacked observation
    const int per_slab = np * w2;
    constexpr int KMAX = (WINDOW_PAIRS * 3 * (HC ? HC : WINDOW_MAX_H) + WINDOW_THREADS - 1) / WINDOW_THREADS;
    int src_off[KMAX], old_off[KMAX], back[KMAX], pp[KMAX];
    int64_t dst_off[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int j = tid + WINDOW_THREADS * k;
        const bool in = j < per_slab;
        const int jc = in ? j : 0;
        const int p = jc / w2, c = jc - p * w2, h = c / 3, f2 = c - 3 * h;
        pp[k] = in ? p : -1;
        src_off[k] = h * WINDOW_PITCH + p * 8 + 2 * f2;
        old_off[k] = (H - 1) * WINDOW_PITCH + p * 8 + 6;
        back[k] = H - 1 - h;
        dst_off[k] = obase[p] * (int64_t)(6 * H) + 2 * c;
    }
    const int64_t slab_floats = npairs * (int64_t)(6 * H);
    for (int ts = 0; ts < nt; ++ts) {
        const float* rs = rec + ts * WINDOW_PITCH;
        const int64_t row0 = (sl0 + ts) * a.n_envs - a.first_slot;    // + env = output row
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (pp[k] < 0) continue;
            const int64_t row = row0 + penv[pp[k]];
            if (row < 0 || row >= a.rows) continue;
            const win_f2 v = *reinterpret_cast<const win_f2*>(rs + src_off[k]);
            const bool live = (float)back[k] <= rs[old_off[k]];
            *reinterpret_cast<win_f2*>(a.dst + dst_off[k] + ts * slab_floats) = live ? v : win_f2{0.0f, 0.0f};
        }
    }
}

extern "C" int flexnet_gather_window(const FlexWindowArgs* a, void* stream) {
    if (!a || !a->row_ring || !a->dst || a->rows < 0 || a->first_slot < 0 || a->n_envs < 1 || a->n_agents < 1 ||
        a->n_agents > FLEXNET_MAX_AGENTS || a->history < 1 || a->history > WINDOW_MAX_H || a->slabs < a->history ||
        ((reinterpret_cast<uintptr_t>(a->row_ring) & 15) | (reinterpret_cast<uintptr_t>(a->dst) & 7)))
        return FLEXNET_EINVAL;
    if (a->rows == 0) return FLEXNET_OK;
    const int64_t npairs = (int64_t)a->n_envs * a->n_agents;
    const int64_t bx = (npairs + WINDOW_PAIRS - 1) / WINDOW_PAIRS;
    const int64_t nslabs = (a->first_slot + a->rows - 1) / a->n_envs - a->first_slot / a->n_envs + 1;
    const int64_t by = (nslabs + WINDOW_SLABS - 1) / WINDOW_SLABS;
    if (bx > 0x7fffffffll || by > 65535) return FLEXNET_EUNSUPPORTED;
    if (a->history == 24) hipLaunchKernelGGL(gather_window_kernel<24>, dim3((unsigned)bx, (unsigned)by), dim3(WINDOW_THREADS), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL(gather_window_kernel<0>, dim3((unsigned)bx, (unsigned)by), dim3(WINDOW_THREADS), 0, (hipStream_t)stream, *a);
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}

#define GATHER_THREADS 256
struct GatherPlan { int first_block[FLEXNET_GATHER_MAX_JOBS + 1]; };

// TD: the reward-statistics pass of the value loss (csrc/flex_td.h: td_stats_block) rides behind the copy blocks, reading the
// reward rows from where job td_job (and td_job + 1, a window that wraps the ring) copies them FROM
static_assert(GATHER_THREADS == TD_THREADS, "the statistics blocks ride in the gather launch");
struct GatherTd { FlexTdLossArgs td; int32_t job, jobs; };
template <bool TD>
__global__ __launch_bounds__(GATHER_THREADS) void gather_rows_kernel(FlexGatherArgs a, GatherPlan p, GatherTd t) {
    int bx = blockIdx.x;
    if constexpr (TD) {
        // (the statistics blocks come FIRST: they are the launch's longest dependent chain and start with it)
        if (bx < TD_BLOCKS) {
            const int j1 = t.jobs > 1 ? t.job + 1 : t.job;
            const TdRewardRows rr = {a.src[t.job], a.src[j1], a.rows[t.job], a.src_stride[t.job], a.src_stride[j1]};
            td_stats_block(t.td, rr, bx);
            return;
        }
        bx -= TD_BLOCKS;
    }
    int j = 0;
    while (j + 1 < a.n_jobs && bx >= p.first_block[j + 1]) ++j;
    const int nb = p.first_block[j + 1] - p.first_block[j], b = bx - p.first_block[j];
    const float* __restrict__ src = a.src[j];
    float* __restrict__ dst = a.dst[j];
    const int w = a.width[j], ss = a.src_stride[j], ds = a.dst_stride[j];
    const int64_t rows = a.rows[j];
    const bool vec = ((w | ss | ds) & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
    if (vec) {
        const int w4 = w >> 2;
        const int64_t total = rows * w4;
        const bool dense = ss == w && ds == w;
        for (int64_t i = (int64_t)b * GATHER_THREADS + threadIdx.x; i < total; i += (int64_t)nb * GATHER_THREADS) {
            int64_t so = i, dof = i;
            if (!dense) { const int64_t r = i / w4; const int c = (int)(i - r * w4); so = r * (ss >> 2) + c; dof = r * (ds >> 2) + c; }
            const pack_f4 v = __builtin_nontemporal_load(reinterpret_cast<const pack_f4*>(src) + so);
            reinterpret_cast<pack_f4*>(dst)[dof] = v;
        }
    } else if (w <= 8) {
        // narrow columns (reward: n_agents floats, done: one): a thread per ROW — no 64-bit division per element (round 5: the
        // refresh of a 32 768-row batch's small columns was 13 us of launch for 3.7 MB)
        for (int64_t r = (int64_t)b * GATHER_THREADS + threadIdx.x; r < rows; r += (int64_t)nb * GATHER_THREADS) {
            const float* sp = src + r * ss;
            float* dp = dst + r * ds;
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = c < w ? sp[c] : 0.0f;
#pragma unroll
            for (int c = 0; c < 8; ++c) if (c < w) dp[c] = v[c];
        }
    } else {
        const int64_t total = rows * w;
        for (int64_t i = (int64_t)b * GATHER_THREADS + threadIdx.x; i < total; i += (int64_t)nb * GATHER_THREADS) {
            const int64_t r = i / w;
            const int c = (int)(i - r * w);
            dst[r * ds + c] = src[r * ss + c];
        }
    }
}

static int gather_rows_run(const FlexGatherArgs* a, const GatherTd* t, void* stream) {
    if (!a || a->n_jobs < 0 || a->n_jobs > FLEXNET_GATHER_MAX_JOBS) return FLEXNET_EINVAL;
    if (a->n_jobs == 0) return t ? FLEXNET_EINVAL : FLEXNET_OK;
    GatherPlan p;
    int blocks = 0;
    for (int j = 0; j < a->n_jobs; ++j) {
        if (!a->src[j] || !a->dst[j] || a->rows[j] < 0 || a->width[j] < 1 || a->src_stride[j] < a->width[j] ||
            a->dst_stride[j] < a->width[j])
            return FLEXNET_EINVAL;
        p.first_block[j] = blocks;
        // ~16 KB per block, at least one block per job, at most 4096 per job (the loop strides)
        const int64_t bytes = a->rows[j] * (int64_t)a->width[j] * 4;
        int64_t nb = (bytes + 16383) / 16384;
        nb = nb < 1 ? 1 : (nb > 4096 ? 4096 : nb);
        blocks += (int)nb;
    }
    p.first_block[a->n_jobs] = blocks;
    for (int j = a->n_jobs + 1; j <= FLEXNET_GATHER_MAX_JOBS; ++j) p.first_block[j] = blocks;
    if (t) hipLaunchKernelGGL(gather_rows_kernel<true>, dim3(blocks + TD_BLOCKS), dim3(GATHER_THREADS), 0, (hipStream_t)stream, *a, p, *t);
    else {
        GatherTd none;
        none.job = none.jobs = 0;                                // (never read without TD)
        hipLaunchKernelGGL(gather_rows_kernel<false>, dim3(blocks), dim3(GATHER_THREADS), 0, (hipStream_t)stream, *a, p, none);
    }
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}

extern "C" int flexnet_gather_rows(const FlexGatherArgs* a, void* stream) { return gather_rows_run(a, nullptr, stream); }

// flexnet_gather_rows with the statistics pass of flexnet_td_loss / flexnet_critic_td_backward (= flexnet_td_stats(td)) riding
// in the launch: jobs reward_job .. reward_job + reward_jobs - 1 (one, or two for a window that wraps the ring) copy the
// [td->rows, td->n_agents] reward rows into td->reward; the statistics are taken from the rows those jobs READ, so the caller
// hands `td` on with stats_ready = 1.
extern "C" int flexnet_gather_rows_td(const FlexGatherArgs* a, int32_t reward_job, int32_t reward_jobs, const FlexTdLossArgs* td,
                                      void* stream) {
    if (!a || !td || reward_jobs < 1 || reward_jobs > 2 || reward_job < 0 || reward_job + reward_jobs > a->n_jobs)
        return FLEXNET_EINVAL;
    if (td->rows < 1 || td->n_agents < 1 || td->n_agents > TD_NA || !td->workspace || td->workspace_floats < FLEXNET_TD_WS_FLOATS ||
        (reinterpret_cast<uintptr_t>(td->workspace) & 7) != 0)
        return FLEXNET_EINVAL;
    int64_t rows = 0;
    for (int j = reward_job; j < reward_job + reward_jobs; ++j) {
        if (a->width[j] != td->n_agents || a->rows[j] < 0) return FLEXNET_EINVAL;
        rows += a->rows[j];
    }
    if (rows != td->rows) return FLEXNET_EINVAL;
    GatherTd t;
    t.td = *td; t.job = reward_job; t.jobs = reward_jobs;
    return gather_rows_run(a, &t, stream);
}

// ---- the same refresh with the window's first slot read from device memory (flexnet_window_refresh, csrc/window_refresh.h) ---
template <bool TD>
__global__ __launch_bounds__(WINDOW_THREADS) void window_refresh_kernel(FlexWindowRefreshArgs a, WindowPlan p, FlexTdLossArgs td) {
    int bx = blockIdx.x;
    if constexpr (TD) {
        if (bx < TD_BLOCKS) { window_refresh_td_block(a, td, bx); return; }      // (first in the grid: gather_rows_kernel)
        bx -= TD_BLOCKS;
    }
    window_refresh_copy_block(a, p, bx);
}

extern "C" int flexnet_window_refresh(const FlexWindowRefreshArgs* a, const FlexTdLossArgs* td, void* stream) {
    WindowPlan p;
    int blocks = 0;
    const int rc = window_refresh_prepare(a, td, &p, &blocks);
    if (rc != FLEXNET_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (td) hipLaunchKernelGGL(window_refresh_kernel<true>, dim3(blocks + TD_BLOCKS), dim3(WINDOW_THREADS), 0, s, *a, p, *td);
    else hipLaunchKernelGGL(window_refresh_kernel<false>, dim3(blocks), dim3(WINDOW_THREADS), 0, s, *a, p, FlexTdLossArgs{});
    return hipGetLastError() == hipSuccess ? FLEXNET_OK : FLEXNET_EHIP;
}
