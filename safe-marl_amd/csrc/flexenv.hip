// flexenv.hip — kernels and C-ABI host side of the flexibility-provision hot path (gfx950).
// Boundary: include/flexenv.h.  Reference semantics: SURVEY.md §8(a) a1-a12; every kernel cites
// the reference lines (under /root/reference) whose behaviour it reproduces.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include "flex_device.h"
#include "flexnet.h"
#include "actor_r16.h"

#define FLEX_MAX_DEVICES 16
#define HIP_TRY(x) do { hipError_t _e = (x); if (_e != hipSuccess) { \
    fprintf(stderr, "[flexenv] %s failed: %s (%s:%d)\n", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
    return FLEX_EHIP; } } while (0)

// ------------------------------------------------------------------------------------------------
// Per-environment state (device, owned by the handle).  Layout is array-of-records per FIELD with
// the environment as the slow index, so each wavefront touches a few short contiguous runs.
// ------------------------------------------------------------------------------------------------
// Agent fields of an environment's record: the six a step rewrites come first and are contiguous — [field][agent] with
// the agents packed (field stride = n_agents, not FLEX_MAX_AGENTS), 240 B for the reference's five buildings — in a record
// padded to whole 64-byte lines (agent_rec_doubles).  Until round 4 every field sat in its own 64-byte line and a step wrote
// seven 40-byte pieces into seven lines.  AF_EINIT is written by reset only: initial_ess_energy differs
// from current_ess_energy exactly between a reset and the first step (env:100,147 vs env:354), i.e. while
// steps == 1, so a step reads `steps == 1 ? EINIT : E` and stores E alone; flexenv_peek(E_INIT) applies the same rule.
enum { AF_E = 0, AF_PRED, AF_CH, AF_DIS, AF_Q, AF_PCT, AF_EINIT, AF_COUNT };          // agent fields
// the four a step reads in one 16-byte load come first; the two solver statistics share an aligned 8-byte store
enum { IF_STEPS = 0, IF_START, IF_ROW, IF_OBSCNT, IF_ITERS, IF_SWEEPS, IF_EPISODE, IF_COUNT = 8 }; // int fields
__host__ __device__ __forceinline__ int agent_rec_doubles(int n_agents) { return (AF_COUNT * n_agents + 7) & ~7; }

struct DevState {
    double* vm;        // [N, n_bus]   |V| in BUS order           (current_voltage, env:146,310)
    uint32_t* vw;      // [N, LW]      (Re V - 1, Im V) in group-LANE order as an fp16 pair (LW = 32 lanes per environment on feeders
                       //              of <= 32 PQ buses, else 64): the warm start of the next solve — an initial guess only (the
                       //              solve ends on an fp64 mismatch test).  fp16 of the OFFSET from the flat start keeps 5e-5 pu,
                       //              a twentieth of what the loads move V between steps: 4 bytes per bus (fp32 pairs: 8, round 4)
    double* agent;     // [N, agent_rec_doubles(n_agents)]   record = [AF_*][n_agents], see the enum
    double* cumrew;    // [N]
    int32_t* ienv;     // [N, IF_COUNT]
    float* ring;       // [N, n_agents, 2 * history, 6]  observation history (env:387-401) as a MIRROR ring: the row pushed k-th
                       //              in its episode sits in slots k mod H and H + k mod H, so the last H rows are always ONE
                       //              contiguous run — slots (k mod H) + 1 .. (k mod H) + H — and the stacked observation of
                       //              env:387-401 is that run, read in place (no rotation); slots 1 .. H-1 are zeroed at
                       //              every episode start: the zero left-padding of SURVEY A16 is physically there
};

struct FlexEnv {
    FlexCfg cfg;
    int32_t n_envs, n_bus, device;
    SeriesTab series;
    DevNet* net;       // device
    DevNet hnet;       // host copy
    DevState st;
    int64_t* step_counter;   // device cell flexenv_step bumps once per launch (flexenv_set_step_counter), or NULL
    int64_t step_modulo;     // ... wrapping to 0 here (0 = never)
    const int64_t* obs_cursor;   // FLEX_STEP_OBS_RING (flexenv_set_obs_ring)
    int64_t obs_slab_stride;
    int32_t obs_slabs;
    FlexReplaySink sink;         // FLEX_STEP_REPLAY_SINK (flexenv_set_replay_sink)
    int32_t has_sink;
};

// Diagnostic build only (-DFLEX_STAMPS): per-phase s_memtime stamps, lane 0 of each wave, written to a
// buffer nothing else reads (cdna_hip_programming.md §7 "In-kernel stamps").  Never timed, never shipped.
#ifdef FLEX_STAMPS
#define FLEX_STAMP(slot) do { unsigned long long _t; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    if (a.stamps && (lane & (64 / EPW - 1)) == 0) a.stamps[(int64_t)env * 16 + (slot)] = _t; } while (0)
#define FLEX_STAMP_RT(slot) do { unsigned long long _t; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    if (a.stamps && (lane & (64 / EPW - 1)) == 0) a.stamps[(int64_t)env * 16 + (slot)] = _t; } while (0)
#else
#define FLEX_STAMP(slot) do { } while (0)
#define FLEX_STAMP_RT(slot) do { } while (0)
#endif

struct KArgs {
    unsigned long long* stamps;
    FlexCfg cfg;
    const DevNet* net;
    DevState st;
    const double* series;
    int64_t rows;
    int32_t cols, n_envs, n_bus;
    // derived on the host once per launch so that no lane spends a division on them
    int32_t row_bytes;         // cols * 8; the series table is addressed with 32-bit byte offsets (checked at create)
    float inv_h, inv_h3;       // 1/history, 1/(3*history) for the observation plan's small_mod
    double inv_eta_ch, inv_eta_dis;
    int64_t* step_counter;     // += 1 per flex_step_kernel launch (one lane), or NULL
    int64_t step_modulo;       // the counter wraps to 0 here (0 = never)
    const int64_t* obs_cursor; // FLEX_STEP_OBS_RING: the observation goes to slab (obs_cursor[0] + 1) mod obs_slabs ...
    int64_t obs_slab_stride;   // ... of a ring whose slabs are this many elements apart
    int32_t obs_slabs;         // 0: `obs` is the output buffer itself
    FlexReplaySink sink;       // FLEX_STEP_REPLAY_SINK: read only by the SINK instantiation's epilogue
};

__device__ __forceinline__ double load_action(const void* p, int dtype, int64_t i) {
    return dtype == FLEX_F32 ? (double)((const float*)p)[i] : ((const double*)p)[i];
}

__device__ __forceinline__ int64_t clamp_row(int64_t r, int64_t rows) { return r < 0 ? 0 : (r >= rows ? rows - 1 : r); }

// Which environment does this lane group serve?  `valid` is false for the spare group of an odd batch; such a
// group computes on environment n_envs-1's inputs (no out-of-bounds reads) and stores nothing.
template <int EPW>
struct EnvSlot {
    int lane, env;
    bool valid;
    __device__ __forceinline__ EnvSlot(int n_envs) {
        constexpr int LW = FLEX_WAVE / EPW;
        lane = threadIdx.x & 63;
        const int wave = blockIdx.x * FLEX_WAVES_PER_BLOCK + (threadIdx.x >> 6);
        const int e = wave * EPW + lane / LW;
        valid = e < n_envs;
        env = valid ? e : n_envs - 1;
    }
    // true when the whole wavefront has nothing to do
    __device__ __forceinline__ bool wave_idle(int n_envs) const {
        const int wave = blockIdx.x * FLEX_WAVES_PER_BLOCK + (threadIdx.x >> 6);
        return wave * EPW >= n_envs;
    }
};

// Reward terms, env:679-706.  Building terms live in building lanes, voltages in PQ lanes; the slack bus
// (|V| = 1 exactly) contributes max(0, 1 - v_max, v_min - 1) to the penalty over ALL buses.
struct RewardOut { double reward, revenue, der, ess, disc, vpen; };     // (valid in the last lane of each group)
template <int EPW>
__device__ __forceinline__ RewardOut reward_terms(const FlexCfg& c, const LaneNet& ln, bool is_bld, double price,
                                                  double pred, double ch, double dis, double q, double v) {
    RewardOut r;
    double t[5] = {is_bld ? price * pred : 0.0,
                   is_bld ? c.pv_cost * q : 0.0,                                    // signed: SURVEY A6
                   is_bld ? c.ess_cost * (ch + dis) : 0.0,
                   is_bld ? c.discomfort_coeff * pred * pred : 0.0,
                   ln.pq ? c.voltage_coeff * fmax(0.0, fmax(v - c.v_max, c.v_min - v)) : 0.0};
    grp_sum5<EPW, false>(t, ln.grp);                          // valid in the group's LAST lane only: that lane stores them
    const double slack_pen = c.voltage_coeff * fmax(0.0, fmax(1.0 - c.v_max, c.v_min - 1.0));
    r.revenue = t[0]; r.der = t[1]; r.ess = t[2]; r.disc = t[3]; r.vpen = t[4] + slack_pen;
    r.reward = r.revenue - r.der - r.ess - r.disc - r.vpen;
    return r;
}

// Per-step results and state are written with nontemporal stores: this launch never reads them back, and lines
// that do not sit dirty in the per-XCD L2s shorten the write-back that ends the launch (measured: +6 % env-steps/s).
template <typename T> __device__ __forceinline__ void st_nt(T* p, T v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void st_nt2(float2* p, float2 v) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f t; t.x = v.x; t.y = v.y; __builtin_nontemporal_store(t, reinterpret_cast<v2f*>(p));
}

// Warm-start record of a bus (DevState::vw): fp16 pair of (Re V - 1, Im V).
typedef _Float16 flex_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_warm(double e, double f) {
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz((float)(e - 1.0), (float)f));
}
__device__ __forceinline__ void unpack_warm(uint32_t u, double& e, double& f) {
    const flex_h2 h = __builtin_bit_cast(flex_h2, u);
    e = 1.0 + (double)(float)h.x; f = (double)(float)h.y;
}

// The kernel's first parameter (KArgs, by value) sits at offset 0 of the kernarg segment.  Re-deriving its address
// through an opaque asm makes later reads fresh scalar loads at the point of use instead of values kept live
// (and spilled to VGPR lanes) from kernel entry.
// `base`: the kernarg segment's address — __builtin_amdgcn_kernarg_segment_ptr() in a kernel; a called function is HANDED
// it (the builtin is null outside kernels).
template <typename T> __device__ __forceinline__ const T* relaunder_kernarg(unsigned long long base) {
    unsigned long long v = base;
    asm volatile("" : "+s"(v));
    return (const T*)(const __attribute__((address_space(4))) T*)v;
}

// Addressing: everything a wavefront touches hangs off a handful of wavefront-uniform base pointers (scalar
// registers, computed on the scalar unit from the kernel arguments and the wavefront's first environment) plus a
// 32-bit per-lane byte offset — one `global_load v, v_off, s[base]` per access instead of a 64-bit multiply-add
// chain per lane and access.
template <typename T> __device__ __forceinline__ T ld_at(const void* base, uint32_t off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + off);
}
template <typename T> __device__ __forceinline__ void st_at(void* base, uint32_t off, T v) {
    __builtin_nontemporal_store(v, reinterpret_cast<T*>(reinterpret_cast<char*>(base) + off));
}
__device__ __forceinline__ void st_at2(void* base, uint32_t off, float2 v) {
    st_nt2(reinterpret_cast<float2*>(reinterpret_cast<char*>(base) + off), v);
}
__device__ __forceinline__ int clamp_row32(int r, int rows) { return r < 0 ? 0 : (r >= rows ? rows - 1 : r); }

// get_obs (env:370-403): the stacked [n_agents, history*6] observation is, per agent, the last `history` feature rows
// with zero left-padding (A16).  The environment keeps them as a mirror ring (DevState::ring): a contiguous run.
// x mod m for small non-negative x (< 2^20): one float multiply by the host's 1/m and a correction.
__device__ __forceinline__ int small_mod(int x, int m, float inv_m) {
    int r = x - (int)((float)x * inv_m) * m;
    r = r < 0 ? r + m : r;
    return r >= m ? r - m : r;
}

// Row-ring record (FLEX_STEP_OBS_RING): what a step files per (environment, agent) where the replay keeps observations —
// the newest feature row and the number of OLDER rows of its episode that belong to its stacked observation:
//     [Pd, Qd, Ppv, V | price, E, older, 0]        (two 16-byte stores; include/flexenv.h)
#define FLEX_ROW_W FLEX_ROW_FLOATS

// Stacked-output mode only (flexenv_step with an output buffer, no FLEX_STEP_OBS_ROWS): the history part of the stacked
// observation — everything except this step's own row — depends on nothing the solve produces: its loads are requested
// before the solve, its stores go out right after it.
//
// Work split: in float2 units an agent's block is 3*history units long; group lane l owns the units l, l+LW, l+2LW
// of EVERY agent's block (three classes of 32 lanes, or two of 64, cover history <= 32 resp. 42 rows).  With the mirror
// ring the source of unit `rem` is ring unit 3 * ((k mod H) + 1) + rem of the same agent — the same offset for every agent,
// no wrap, no padding test (the zeros are in the ring).
//
// Both directions go through buffer descriptors that cover just this wavefront's environments: a raw buffer access
// is never turned into a branch by the compiler (a conditional global access is — for a load with a full s_waitcnt
// behind every one of them), an out-of-range offset (-1) reads zeros or drops the store, and the 32-bit range limit of a
// descriptor applies per wavefront, not to the whole batch.  The slots of this step's own row are written too (with
// whatever the ring holds there): obs_store_new overwrites them afterwards, in program order.  The stores are
// nontemporal: nothing in this launch reads them back.
typedef int flex_v2i __attribute__((ext_vector_type(2)));
typedef int flex_v4i __attribute__((ext_vector_type(4)));
#define FLEX_BUF_FLAGS 0x00020000
#define FLEX_AUX_NT 2            // gfx940+ cache policy: bit 1 = nt
#define FLEX_OBS_CLASSES(EPW_) ((EPW_) == 2 ? 3 : 2)
#define FLEX_OBS_AGENTS_SMALL 5  // the reference's five buildings: agent loop without a per-agent predicate
#define FLEX_OBS_AGENTS_LARGE FLEX_MAX_AGENTS

template <int EPW, int NA_CAP, typename OutT>
struct ObsHist {
    static constexpr int LW = FLEX_WAVE / EPW, CLS = FLEX_OBS_CLASSES(EPW);
    static constexpr bool NA_EXACT = NA_CAP == FLEX_OBS_AGENTS_SMALL;            // host dispatch guarantees na == NA_CAP
    float2 buf[NA_CAP][CLS];
    int dst[CLS];

    __device__ __forceinline__ void load(const KArgs& a, int env0, int g, bool valid, bool enable, const LaneNet& ln, int k) {
        const int H = a.cfg.history, na = a.cfg.n_agents, H3 = 3 * H;
        const int ring_floats = na * H * 12, out_floats = na * H * 6;            // per environment
        const int span = enable ? EPW * ring_floats : 0;                        // `enable` is launch-uniform
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.st.ring + (int64_t)env0 * ring_floats), 0, span * 4, FLEX_BUF_FLAGS);
        const int first = 3 * (small_mod(k, H, a.inv_h) + 1);                   // ring unit of the oldest row of the window
        int src[CLS];
#pragma unroll
        for (int i = 0; i < CLS; ++i) {
            const int rem = ln.l + LW * i;
            const bool have = valid && rem < H3;
            src[i] = have ? g * ring_floats * 4 + 8 * (first + rem) : -1;
            dst[i] = have ? g * out_floats * (int)sizeof(OutT) + 2 * (int)sizeof(OutT) * rem : -1;
        }
#pragma unroll
        for (int ag = 0; ag < NA_CAP; ++ag) {
            const bool live = NA_EXACT || ag < na;           // scalar
#pragma unroll
            for (int i = 0; i < CLS; ++i) {
                const flex_v2i r = __builtin_amdgcn_raw_buffer_load_b64(rin, live ? src[i] : -1, ag * H3 * 16, 0);
                buf[ag][i] = make_float2(__int_as_float(r.x), __int_as_float(r.y));
            }
        }
    }
    __device__ __forceinline__ void store(const KArgs& a, int env0, bool enable, OutT* __restrict__ out) const {
        const int H = a.cfg.history, na = a.cfg.n_agents, H3 = 3 * H;
        const int env_floats = na * H * 6;
        const int span = enable ? EPW * env_floats : 0;
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(out + (int64_t)env0 * env_floats), 0, span * (int)sizeof(OutT), FLEX_BUF_FLAGS);
#pragma unroll
        for (int ag = 0; ag < NA_CAP; ++ag) {
            const bool live = NA_EXACT || ag < na;
#pragma unroll
            for (int i = 0; i < CLS; ++i) {
                const int d = live ? dst[i] : -1;
                if constexpr (sizeof(OutT) == 4) {
                    flex_v2i w; w.x = __float_as_int(buf[ag][i].x); w.y = __float_as_int(buf[ag][i].y);
                    __builtin_amdgcn_raw_buffer_store_b64(w, rout, d, ag * H3 * 8, FLEX_AUX_NT);
                } else {
                    const double dx = (double)buf[ag][i].x, dy = (double)buf[ag][i].y;
                    flex_v4i w; w.x = __double2loint(dx); w.y = __double2hiint(dx); w.z = __double2loint(dy); w.w = __double2hiint(dy);
                    __builtin_amdgcn_raw_buffer_store_b128(w, rout, d, ag * H3 * 16, FLEX_AUX_NT);
                }
            }
        }
    }
};

// This step's row [Pd, Qd, Ppv, V, price, E] (env:377-382), pushed as row k of its episode: both mirror slots of the
// environment's ring (building lanes; 24 contiguous bytes, three 8-byte stores each), the record of the replay's row ring
// when one is given (`rows`: this launch's slab, [N, n_agents, FLEX_ROW_W]), the newest slot of a stacked output when
// one is given (`out`).
// The push count (IF_OBSCNT = k + 1) is the caller's to store: the step kernel folds it into the one 16-byte store of its
// integer fields.
template <int EPW, typename OutT>
__device__ __forceinline__ void obs_store_new(const KArgs& a, int env0, int g, bool valid, const LaneNet& ln, int k,
                                              double pd, double qd, double ppv, double v, double price, double e,
                                              OutT* __restrict__ out, float* __restrict__ rows) {
    const int H = a.cfg.history, na = a.cfg.n_agents;
    const int ring_floats = na * H * 12, env_floats = na * H * 6;
    float* const ring = a.st.ring + (int64_t)env0 * ring_floats;          // wavefront-uniform bases
    const int slot = small_mod(k, H, a.inv_h);
    if (valid && ln.agent >= 0) {
        const float2 r0 = make_float2((float)pd, (float)qd), r1 = make_float2((float)ppv, (float)v);
        const float2 r2 = make_float2((float)price, (float)e);
        const uint32_t ro = (uint32_t)(g * ring_floats + (ln.agent * 2 * H + slot) * 6) * 4, mo = ro + (uint32_t)H * 24;
        st_at2(ring, ro, r0); st_at2(ring, ro + 8, r1); st_at2(ring, ro + 16, r2);
        st_at2(ring, mo, r0); st_at2(ring, mo + 8, r1); st_at2(ring, mo + 16, r2);
        if (rows) {
            typedef float rw_f4 __attribute__((ext_vector_type(4)));
            float* const rb = rows + (int64_t)env0 * (na * FLEX_ROW_W);
            const uint32_t wo = (uint32_t)((g * na + ln.agent) * FLEX_ROW_W) * 4;
            const float older = (float)(k < H - 1 ? k : H - 1);
            st_at<rw_f4>(rb, wo, rw_f4{r0.x, r0.y, r1.x, r1.y});
            st_at<rw_f4>(rb, wo + 16, rw_f4{r2.x, r2.y, older, 0.0f});
        }
        if (out) {
            OutT* const o = out + (int64_t)env0 * env_floats;
            const uint32_t oo = (uint32_t)(g * env_floats + (ln.agent * H + (H - 1)) * 6) * (uint32_t)sizeof(OutT);
            if constexpr (sizeof(OutT) == 4) {
                st_at2(o, oo, r0); st_at2(o, oo + 8, r1); st_at2(o, oo + 16, r2);
            } else {
                st_at<double>(o, oo, pd); st_at<double>(o, oo + 8, qd); st_at<double>(o, oo + 16, ppv);
                st_at<double>(o, oo + 24, v); st_at<double>(o, oo + 32, price); st_at<double>(o, oo + 40, e);
            }
        }
    }
}

// Episode start (reset, restart inside a step): slots 1 .. H-1 of every agent's ring are zeroed — the rows "before the
// episode began" of every window of the new episode (A16); slot 0 and the mirror half are overwritten before they are read.
template <int EPW>
__device__ __forceinline__ void obs_ring_clear(const KArgs& a, int env, bool valid, const LaneNet& ln) {
    constexpr int LW = FLEX_WAVE / EPW;
    const int H = a.cfg.history, na = a.cfg.n_agents;
    float* ring = a.st.ring + (int64_t)env * na * H * 12;
    const int per_agent = (H - 1) * 6;
    if (valid)
        for (int idx = ln.l; idx < na * per_agent; idx += LW) {
            const int ag = idx / per_agent, rem = idx - ag * per_agent;
            ring[(ag * 2 * H + 1) * 6 + rem] = 0.0f;
        }
}

// General path (history/agent counts beyond the register budget, the stand-alone get_obs(), the episode start): the same
// push, and the stacked output read from the ring afterwards.  `fresh`: row 0 of an episode whose ring this wavefront has
// just cleared (obs_ring_clear) — the older rows of its window are zeros, written out directly.
template <int EPW, typename OutT>
__device__ __forceinline__ void push_and_emit_obs(const KArgs& a, int env, bool valid, const LaneNet& ln, int k,
                                                  double pd, double qd, double ppv, double v, double price,
                                                  double e, OutT* __restrict__ out, float* __restrict__ rows, bool fresh) {
    constexpr int LW = FLEX_WAVE / EPW;
    const int H = a.cfg.history, na = a.cfg.n_agents;
    int32_t* ie = a.st.ienv + (int64_t)env * IF_COUNT;
    const int slot = k % H;
    float* ring = a.st.ring + (int64_t)env * na * H * 12;
    OutT* o = out ? out + (int64_t)env * na * H * 6 : nullptr;
    if (valid && ln.agent >= 0) {
        const double feat[6] = {pd, qd, ppv, v, price, e};
        float* r = ring + (ln.agent * 2 * H + slot) * 6;
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            r[t] = (float)feat[t];
            r[H * 6 + t] = (float)feat[t];
            if (o) o[(ln.agent * H + (H - 1)) * 6 + t] = (OutT)feat[t];
        }
        if (rows) {
            float* w = rows + ((int64_t)env * na + ln.agent) * FLEX_ROW_W;
#pragma unroll
            for (int t = 0; t < 6; ++t) w[t] = (float)feat[t];
            w[6] = (float)(k < H - 1 ? k : H - 1);
            w[7] = 0.0f;
        }
    }
    if (o && valid) {
        // the H - 1 older rows of the window: ring slots slot + 1 .. slot + H - 1 (none of them written above); the first
        // observation of an episode is zeros there (A16) — written out directly, not read back from the zero-fill
        const int total = na * H * 6;
        for (int idx = ln.l; idx < total; idx += LW) {
            const int ag = idx / (H * 6), rem = idx - ag * (H * 6);
            if (rem >= (H - 1) * 6) continue;
            o[idx] = fresh ? (OutT)0 : (OutT)ring[(ag * 2 * H + slot + 1) * 6 + rem];
        }
    }
    if (valid && ln.l == 0) ie[IF_OBSCNT] = k + 1;
}

// -------------------------------------------------------------------------------------------------
// reset()/manual_reset(): env:74-155, 157-239
// -------------------------------------------------------------------------------------------------
// draw d of the reset stream lives in group lane d/2 (u0 for even d, u1 for odd d).  Both halves are pulled
// unconditionally and selected afterwards — a shuffle inside a lane-divergent branch would read lanes that are
// masked off there.
__device__ __forceinline__ double flex_draw(double u0, double u1, int d, int base) {
    const double a = __shfl(u0, (d >> 1) + base, FLEX_WAVE), b = __shfl(u1, (d >> 1) + base, FLEX_WAVE);
    return (d & 1) ? b : a;
}

struct DevResetSpec { const int32_t *day, *hour, *interval; const double *e0, *a0; };

// The episode (re)start of the lane groups flagged `valid`; shared by the reset kernel and by the step kernel's
// auto-reset tail.  `ln0` is the group's LaneNet with pq not yet masked.
template <int EPW, typename ObsT>
__device__ __forceinline__ void flex_reset_body(const KArgs& a, int env, bool valid, const LaneNet& ln0,
                                                const DevResetSpec& inj, ObsT* __restrict__ obs, int want_obs,
                                                float* __restrict__ rows, uint8_t* __restrict__ failed, bool or_failed) {
    const FlexCfg& c = a.cfg;
    LaneNet ln = ln0;
    ln.pq = ln.pq && valid;
    const int nb = a.n_bus, na = c.n_agents;
    int32_t* ie = a.st.ienv + (int64_t)env * IF_COUNT;
    const bool is_bus = ln.bus >= 0, is_bld = ln.agent >= 0;
    const int ag = is_bld ? ln.agent : 0;
    double* agst = a.st.agent + (int64_t)env * agent_rec_doubles(na);
    const bool all_injected = inj.day && inj.hour && inj.interval && inj.e0 && inj.a0;
    const int max_attempts = all_injected ? 1 : 8;
    uint32_t episode = (uint32_t)ie[IF_EPISODE];

    bool ok = false;
    int start = 0, iters = 0, sweeps = 0;
    int64_t row = 0;
    double e = 1.0, f = 0.0, e0 = 0.0, pd = 0.0, qd = 0.0, ppv = 0.0, price = 0.0, e_new = 0.0;
    FlexAct act = {0, 0, 0, 0, 0};
    for (int attempt = 0; attempt < max_attempts; ++attempt) {
        // groups that already hold a solvable episode sit out the re-draw (env:83,150-153)
        const bool need = valid && !ok;
        if (__ballot(need) == 0ull) break;
        // group lane j holds Philox block j of this attempt: draws 2j and 2j+1
        double u0, u1;
        philox_pair((uint32_t)ln.l, episode, (uint32_t)env, c.seed, u0, u1);
        const double uh = flex_draw(u0, u1, 0, ln.base), ud = flex_draw(u0, u1, 1, ln.base);
        const double ui = flex_draw(u0, u1, 2, ln.base);
        const int de = 3 + ag, da = 3 + na + 4 * ag;
        const double ue = flex_draw(u0, u1, de, ln.base);
        const double ua0 = flex_draw(u0, u1, da, ln.base), ua1 = flex_draw(u0, u1, da + 1, ln.base);
        const double ua2 = flex_draw(u0, u1, da + 2, ln.base), ua3 = flex_draw(u0, u1, da + 3, ln.base);
        const int hour = inj.hour ? inj.hour[env] : (int)(uh * 24.0);                          // env:85,412
        const int day = inj.day ? inj.day[env] : (int)(ud * (double)c.n_start_days);           // env:86,424
        const int interval = inj.interval ? inj.interval[env] : (int)(ui * (double)c.per_hour);  // env:87,416
        const int start_n = interval + hour * c.per_hour + day * 24 * c.per_hour;               // env:477
        const int64_t row_n = clamp_row((int64_t)start_n + 1, a.rows);                          // steps = 1: env:76,98
        const double* sr = a.series + row_n * a.cols;
        const double pd_n = is_bus ? sr[ln.bus] : 0.0, qd_n = is_bus ? sr[nb + ln.bus] : 0.0;
        const double ppv_n = is_bld ? sr[2 * nb + ag] : 0.0, price_n = sr[2 * nb + na];
        const double lo = 0.9 * (c.e_max / 2), hi = 1.1 * (c.e_max / 2);                        // env:100
        double e0_n = 0.0;
        FlexAct act_n = {0, 0, 0, 0, 0};
        if (is_bld) {
            e0_n = inj.e0 ? inj.e0[(int64_t)env * na + ag] : lo + (hi - lo) * ue;
            const double* ia = inj.a0 ? inj.a0 + ((int64_t)env * na + ag) * 4 : nullptr;
            const double span = c.action_high - c.action_low;                                   // env:716-719
            const double av0 = ia ? ia[0] : c.action_low + span * ua0, av1 = ia ? ia[1] : c.action_low + span * ua1;
            const double av2 = ia ? ia[2] : c.action_low + span * ua2, av3 = ia ? ia[3] : c.action_low + span * ua3;
            act_n = parse_actions(c, a.inv_eta_ch, a.inv_eta_dis, false, av0, av1, av2, av3, pd_n, ppv_n, e0_n);             // env:113-130
        }
        const double pnet = pd_n - act_n.pred - ppv_n + act_n.ch - act_n.dis;
        const double qnet = qd_n - act_n.q;
        double e_n = 1.0, f_n = 0.0;
        int it_n = 0, sw_n = 0;
        LaneNet lt = ln;
        lt.pq = ln.pq && need;                       // finished groups do not hold up the convergence ballot
        const double e_new_n = e0_n + c.dt * (c.eta_ch * act_n.ch - (1.0 / c.eta_dis) * act_n.dis);   // pf.py:96-98
        // pf.py:45: E_next is a NonNegativeReals variable pinned by that equality; outside the domain the NLP is infeasible
        // and pf.py:104-105 raises (the draw is repeated, env:150-153)
        bool wave_dom;
        const bool dom_bad = grp_any<EPW>(is_bld && need && e_new_n < -FLEX_DOMAIN_EPS, ln.grp, wave_dom);
        const bool ok_n = pf_solve<EPW>(a.net, lt, c.solver, need ? pnet : 0.0, need ? qnet : 0.0, e_n, f_n, c.pf_tol,
                                        c.pf_max_iter, it_n, sw_n) && !dom_bad;                 // env:134-144
        if (need) {
            ok = ok_n; start = start_n; row = row_n; pd = pd_n; qd = qd_n; ppv = ppv_n; price = price_n;
            e0 = e0_n; act = act_n; e = e_n; f = f_n; iters = it_n; sweeps = sw_n;
            e_new = e_new_n;
            ++episode;
        }
    }
    const double v = sqrt(e * e + f * f);
    if (ln.pq) {
        a.st.vm[(int64_t)env * nb + ln.bus] = v;
        a.st.vw[(int64_t)env * (FLEX_WAVE / EPW) + ln.l] = ok ? pack_warm(e, f) : pack_warm(1.0, 0.0);
    }
    if (is_bld && valid) {
        agst[AF_E * na + ag] = e_new;          // env:147
        agst[AF_EINIT * na + ag] = e0;         // A5: stays the pre-solve draw
        agst[AF_PRED * na + ag] = act.pred;
        agst[AF_CH * na + ag] = act.ch;
        agst[AF_DIS * na + ag] = act.dis;
        agst[AF_Q * na + ag] = act.q;
        agst[AF_PCT * na + ag] = act.pct;
    }
    if (ln.l == 0 && valid) {
        a.st.vm[(int64_t)env * nb + a.net->slack_bus] = 1.0;   // pf.py:53: Vsqr[slack] = 1
        ie[IF_STEPS] = 1;                                   // env:76
        ie[IF_START] = start;
        ie[IF_ROW] = (int32_t)row;
        ie[IF_OBSCNT] = 0;                                  // env:79-80
        ie[IF_EPISODE] = (int32_t)episode;
        ie[IF_ITERS] = iters;
        ie[IF_SWEEPS] = sweeps;
        if (failed) { if (or_failed) { if (!ok) failed[env] = 1; } else failed[env] = ok ? 0 : 1; }
    }
    if (ln.l == FLEX_WAVE / EPW - 1 && valid) a.st.cumrew[env] = 0.0;   // env:77 (the lane that writes it in the step, too)
    obs_ring_clear<EPW>(a, env, valid, ln);               // env:79-80: the history starts empty
    if (want_obs) push_and_emit_obs<EPW, ObsT>(a, env, valid, ln, 0, pd, qd, ppv, v, price, e_new, obs, rows, true);
}

template <int EPW, typename ObsT>
__global__ __launch_bounds__(FLEX_WAVE * FLEX_WAVES_PER_BLOCK)
void flex_reset_kernel(KArgs a, const uint8_t* __restrict__ mask, DevResetSpec inj, ObsT* __restrict__ obs,
                       int want_obs, uint8_t* __restrict__ failed) {
    EnvSlot<EPW> slot(a.n_envs);
    if (slot.wave_idle(a.n_envs)) return;
    // a group takes part when its env exists and is selected; the wavefront leaves when no group does
    const bool valid = slot.valid && (!mask || mask[slot.env] != 0);
    if (__ballot(valid) == 0ull) return;
    LaneNet ln;
    load_lane_net<EPW>(a.net, slot.lane, ln);
    flex_reset_body<EPW, ObsT>(a, slot.env, valid, ln, inj, obs, want_obs, nullptr, failed, false);
}


// -------------------------------------------------------------------------------------------------
// step(): env:241-356 for one environment per lane group, get_obs() optionally fused (model.py:220-223)
// -------------------------------------------------------------------------------------------------
// Residency: 4096 envs are 2048 wavefronts at EPW = 2 (2 per SIMD, <= 256 VGPRs) or 4096 at EPW = 1 (4 per SIMD,
// <= 128 VGPRs); in both cases the whole batch must be co-resident, otherwise the last blocks start only when the
// first ones retire and the launch takes twice as long (measured: profiles/).
// Both instantiations are built for two wavefronts per SIMD (256 registers).  A feeder with more than 32 PQ buses takes a
// whole wavefront per environment (EPW = 1): until round 3 that build was capped at 128 registers so that 4096 environments
// stayed co-resident at four wavefronts per SIMD, and spilled 136 registers (~310 B of scratch per lane) doing so.
// Measured on a 45-bus feeder (tools/epw1_bench.py, profiles/r03_epw1_bench.txt): the 256-register build, which spills
// nothing, is faster at EVERY batch size — 10.1 vs 12.2 us at 1024 environments, 12.2 vs 15.8 at 2048, 21.3 vs 23.8 at 4096
// (two rounds of wavefronts) and 38.2 vs 40.7 at 8192 — so it is the only one.
// The body of a step for the environments of wavefront `wave` (of the whole batch).  `slab` >= 0: the replay ring's slab
// index is handed in (flex_rollout_burst_kernel, which walks it itself) instead of read from the cursor cell; `cells`: this
// call maintains the device-side cursor / counter cells (the one-step kernel does, through one lane of the grid).
// ROWS (FLEX_STEP_OBS_ROWS / FLEX_STEP_OBS_RING): get_obs() is the push of this step's feature row — 120 B per environment
// into its mirror ring (where the policy kernels read the stacked observation in place) and, with a row ring registered, a
// 160-B record into the replay's slab — instead of a [n_agents, 6 * history] copy of the whole window per step (2 760 B
// read + 2 880 B written per env-step: 0.65 x the kernel's algorithmic bytes, 15 of its 53 vector loads and 15 of its 58
// stores).  `obs` is then the row ring's base (or NULL).  !ROWS: the stacked copy into `obs`.
// MANY (flex_step_many_kernel: consecutive steps of one launch): `mc` carries, in registers, what the step before has just
// stored — the lane's network rows (loaded once per launch) and its environment's integer record, the head of the prologue's
// only two-level load chain (ienv -> series row) — unless an environment of this wavefront restarted in that step.
#ifndef FLEX_MANY_CARRY_NET
#define FLEX_MANY_CARRY_NET 1
#endif
struct StepCarry {
    flex_v4i iv;                 // steps, start, row, pushes: the 16 bytes the step wrote to ienv
    bool have;                   // wavefront-uniform: everything below is current
    LaneNet ln;
    // the lane's inputs of the NEXT step, as that step would load them: its bus's cells of the series row the step advanced to
    // (read by every lane for get_obs() anyway: env:340 / env:377-382), the ESS state, the warm-start word, the running return
    double pd, qd, ppv, price, e_cur, e_init, cum;
    uint32_t vw;
};
template <int EPW, typename ObsT, typename ActT, int NA_CAP, bool SINK, bool ROWS, bool MANY = false>
__device__ __forceinline__ void flex_step_body(const KArgs& a, const int wave, const ActT* __restrict__ actions,
                                               double* __restrict__ reward, uint8_t* __restrict__ done, double* __restrict__ info,
                                               uint8_t* __restrict__ failed, ObsT* __restrict__ obs, int want_obs, int auto_reset,
                                               const int64_t slab, const bool cells, const unsigned long long kbase,
                                               StepCarry* mc = nullptr) {
    constexpr int LW = FLEX_WAVE / EPW;
    const int lane = threadIdx.x & 63;
    const int env0 = wave * EPW;                               // first environment of this wavefront
    if (env0 >= a.n_envs) return;
    // (tried in round 3: s_setprio 1 for one of the two hardware wave slots of a SIMD, either parity — 13.5-13.8 us per
    //  launch with and without: the end spread of this kernel is not an arbitration effect)
    float* rowrec = nullptr;
    if constexpr (ROWS) {
        if (a.obs_slabs > 0 && obs) {
            // FLEX_STEP_OBS_RING (launch-uniform): the row records go straight into the replay's slab ring, one slab past
            // the one the policy is reading; nobody in this launch writes the cursor
            const int64_t p = (slab >= 0 ? slab : *a.obs_cursor) + 1;
            rowrec = reinterpret_cast<float*>(obs) + (p >= a.obs_slabs ? 0 : p) * a.obs_slab_stride;
        }
    }
    // the spare group of an odd batch computes on env0's inputs (no out-of-bounds reads) and stores nothing
    const bool valid = env0 + lane / LW < a.n_envs;
    const int g = valid ? lane / LW : 0;
    const int env = env0 + g;
    FLEX_STAMP_RT(5);
    FLEX_STAMP(0);
    const FlexCfg& c = a.cfg;
    // the head of the only dependent chain (ienv -> series row) goes out before the 19 table loads
    int32_t* const b_ienv = a.st.ienv + (int64_t)env0 * IF_COUNT;
    const uint32_t o_ienv = g * (IF_COUNT * 4);
    int4 iv;                                                      // steps, start, row, obs_cnt in one load
    LaneNet ln;
    if constexpr (MANY) {
        if (mc->have) iv = int4{mc->iv.x, mc->iv.y, mc->iv.z, mc->iv.w};
        else iv = ld_at<int4>(b_ienv, o_ienv);
#if FLEX_MANY_CARRY_NET
        ln = mc->ln;
#else
        load_lane_net<EPW>(a.net, lane, ln);
#endif
    } else {
        iv = ld_at<int4>(b_ienv, o_ienv);
        load_lane_net<EPW>(a.net, lane, ln);
    }
    ln.pq = ln.pq && valid;
    const int nb = a.n_bus, na = c.n_agents;
    const bool is_bus = ln.bus >= 0, is_bld = ln.agent >= 0;
    const int ag = is_bld ? ln.agent : 0, busi = is_bus ? ln.bus : 0;

    // wavefront-uniform bases and per-lane byte offsets
    const int arec = agent_rec_doubles(na);
    double* const b_agent = a.st.agent + (int64_t)env0 * arec;
    uint32_t* const b_vw = a.st.vw + (int64_t)env0 * LW;
    const ActT* const b_act = actions + (int64_t)env0 * (na * 4);
    const uint32_t o_agent = (g * arec + ag) * 8;                                    // + field * AFB
    const uint32_t o_volt = (g * LW + ln.l) * 4;
    const uint32_t o_bus = busi * 8, o_qbus = (nb + busi) * 8, o_pv = (2 * nb + ag) * 8, o_price = (2 * nb + na) * 8;
    const uint32_t AFB = na * 8;                                                     // bytes between agent fields

    const int steps = iv.x, start = iv.y, obs_cnt = iv.w;
    const int rows = (int)a.rows;
    const uint32_t row_off = (uint32_t)clamp_row32(iv.z, rows) * (uint32_t)a.row_bytes;

    // Every load below uses an address that is valid in EVERY lane (idle lanes read bus 0 / agent 0) and is
    // issued unconditionally: a conditional load compiles to a branch with a full s_waitcnt behind it, and a
    // handful of those in a row serialise the prologue into as many memory round trips.
    // 1) what the solve needs: current data row (env:340, A2), ESS state, actions, previous voltages
    // (MANY, wavefront-uniform: the step before left all of the state below in registers — only the actions are loaded)
    const bool from_regs = MANY && mc->have;
    const uint32_t o_act = (g * na + ag) * (4 * (uint32_t)sizeof(ActT));
    ActT av[4];
    if constexpr (sizeof(ActT) == 4) {
        const float4 t = ld_at<float4>(b_act, o_act);
        av[0] = t.x; av[1] = t.y; av[2] = t.z; av[3] = t.w;
    } else {
        const double2 t0 = ld_at<double2>(b_act, o_act), t1 = ld_at<double2>(b_act, o_act + 16);
        av[0] = t0.x; av[1] = t0.y; av[2] = t1.x; av[3] = t1.y;
    }
    double pd_r, qd_r, ppv_r, price, e_cur_r, e_init_r, cum_before;
    uint32_t vw_word;
    if (from_regs) {
        pd_r = mc->pd; qd_r = mc->qd; ppv_r = mc->ppv; price = mc->price;
        e_cur_r = mc->e_cur; e_init_r = mc->e_init; cum_before = mc->cum; vw_word = mc->vw;
    } else {
        pd_r = ld_at<double>(a.series, row_off + o_bus); qd_r = ld_at<double>(a.series, row_off + o_qbus);
        ppv_r = ld_at<double>(a.series, row_off + o_pv); price = ld_at<double>(a.series, row_off + o_price);
        e_cur_r = ld_at<double>(b_agent, o_agent + AF_E * AFB);
        e_init_r = ld_at<double>(b_agent, o_agent + AF_EINIT * AFB);
        vw_word = ld_at<uint32_t>(b_vw, o_volt);
        // (the epilogue's only read-modify-write: requested here, not behind the solve — a memory round trip per launch)
        cum_before = ld_at<double>(a.st.cumrew + env0, g * 8);
    }
    double we, wf;
    unpack_warm(vw_word, we, wf);
    // 2) what only get_obs() needs: the row env:340 will load (start + steps, A2) and the history part of the
    //    stacked observation, which is copied right away and drains underneath the solve; an environment that
    //    turns out to restart below rewrites its whole observation afterwards (same wavefront, program order)
    const int new_row = clamp_row32(start + steps, rows);
    const uint32_t nrow_off = (uint32_t)new_row * (uint32_t)a.row_bytes;
    // (only building lanes use the next row's Pd / Qd / Ppv: every other lane reads the price cell instead — an address this
    //  wavefront fetches anyway — so the observation touches 5 + 5 + 2 sectors of the next row, not all 18)
    // (MANY: every lane reads its OWN bus's cells — they are its inputs of the launch's next step, StepCarry)
    double n_pd = ld_at<double>(a.series, nrow_off + (is_bld || MANY ? o_bus : o_price));
    double n_qd = ld_at<double>(a.series, nrow_off + (is_bld || MANY ? o_qbus : o_price));
    double n_ppv = ld_at<double>(a.series, nrow_off + (is_bld || MANY ? o_pv : o_price));
    const double n_price = ld_at<double>(a.series, nrow_off + o_price);
    const bool obs_fast = want_obs && (ROWS || (na <= NA_CAP && 3 * c.history <= FLEX_OBS_CLASSES(EPW) * LW));
    ObsHist<EPW, ROWS ? 1 : NA_CAP, ObsT> hist;
    if constexpr (!ROWS) hist.load(a, env0, g, valid, obs_fast, ln, obs_cnt);

    const bool warm = c.warm_start != 0 && ln.pq;
    double e = warm ? we : 1.0, f = warm ? wf : 0.0;

    const double pd = is_bus ? pd_r : 0.0, qd = is_bus ? qd_r : 0.0, ppv = is_bld ? ppv_r : 0.0;
    // initial_ess_energy is current_ess_energy except between a reset and the first step (A5; see the AF_ enum)
    const double e_cur = is_bld ? e_cur_r : 0.0, e_init = is_bld ? (steps == 1 ? e_init_r : e_cur_r) : 0.0;
    // actions -> physical set-points (env:260-293); computed in every lane, kept in building lanes
    FlexAct act = parse_actions(c, a.inv_eta_ch, a.inv_eta_dis, c.raw_actions != 0, (double)av[0], (double)av[1],
                                (double)av[2], (double)av[3], pd, ppv, e_cur);
    if (!is_bld) { act.pct = 0.0; act.pred = 0.0; act.ch = 0.0; act.dis = 0.0; act.q = 0.0; }
    // net load per bus (pf.py:69-73, 81-82)
    const double pnet = pd - act.pred - ppv + act.ch - act.dis;
    const double qnet = qd - act.q;
    // FLEX_STEP_REPLAY_SINK: what the epilogue files into the replay ring but does not compute — the policy's action and
    // its new recurrent state — is requested HERE, last in the prologue's load queue (loads return in order: nothing the
    // solve waits for queues behind these), and lands underneath the solve; issued in the epilogue these loads were a full
    // memory round trip between the solve and the stores (17.6 vs 13.9 us per launch against the kernel without the sink)
    typedef float sk_f4 __attribute__((ext_vector_type(4)));
    sk_f4 sk_hv[SINK ? 3 : 1];
    float sk_av = 0.0f;
    int64_t sk_p = 0;                                                           // the slab this step files into
    if constexpr (SINK) {
        sk_p = slab >= 0 ? slab : *a.obs_cursor;
        const FlexReplaySink& sk = a.sink;
        const sk_f4* hs = reinterpret_cast<const sk_f4*>(sk.hid_new + (int64_t)env * sk.hid_w);
        const int h4 = sk.hid_w >> 2;
#pragma unroll
        for (int r = 0; r < 3; ++r) { const int i = ln.l + LW * r; sk_hv[r] = hs[i < h4 ? i : 0]; }
        sk_av = sk.policy_action[(int64_t)env * sk.act_w + (ln.l < sk.act_w ? ln.l : 0)];
    }

    // power flow (pf.py:10-113)
    int iters = 0, sweeps = 0;
#ifdef FLEX_STAMPS
    asm volatile("" :: "v"(pnet), "v"(qnet), "v"(e), "v"(f));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    FLEX_STAMP(1);
    bool ok = pf_solve<EPW>(a.net, ln, c.solver, pnet, qnet, e, f, c.pf_tol, c.pf_max_iter, iters, sweeps);
#ifdef FLEX_STAMPS
    asm volatile("" :: "v"(e), "v"(f));
#endif
    FLEX_STAMP(2);

    // the epilogue re-reads its configuration and rebuilds its bases from fresh kernarg loads (see relaunder_kernarg)
    const KArgs& z = *relaunder_kernarg<KArgs>(kbase);
    const FlexCfg& cz = z.cfg;
    // stacked-output mode: the history copy's stores go out HERE, its loads having been requested before the solve — issued
    // in the prologue (round 2) the stores made the wavefront wait for those loads, a second memory round trip in front of
    // the solve, to let them drain underneath it: 13.63 -> 13.18 us per launch (round 3)
    if constexpr (!ROWS) hist.store(z, env0, obs_fast, obs);
    double* const e_agent = z.st.agent + (int64_t)env0 * agent_rec_doubles(cz.n_agents);
    uint32_t* const e_vw = z.st.vw + (int64_t)env0 * LW;
    {
        // pf.py:45: E_next (pf.py:96-98) is declared NonNegativeReals; a negative value makes the reference's NLP infeasible,
        // pf.py:104-105 raises and the step takes the failure path of env:314-337 whatever the network equations say.
        // (Tested AFTER the solve, from values the epilogue needs anyway: nothing extra is live across the solve.)
        bool wave_dom;
        const bool dom_bad = grp_any<EPW>(is_bld && valid && e_init + cz.dt * (cz.eta_ch * act.ch - z.inv_eta_dis * act.dis) < -FLEX_DOMAIN_EPS,
                                          ln.grp, wave_dom);
        ok = ok && !dom_bad;
    }
    double* const e_vm = z.st.vm + (int64_t)env0 * z.n_bus;
    int32_t* const e_ienv = z.st.ienv + (int64_t)env0 * IF_COUNT;
    const uint32_t o_vm = (g * z.n_bus + busi) * 8;
    double v, pred, ch, dis, q, e_new;
    if (ok) {
        v = sqrt(e * e + f * f);                                                       // pf.py:108
        pred = act.pred; ch = act.ch; dis = act.dis; q = act.q;
        e_new = e_init + cz.dt * (cz.eta_ch * ch - z.inv_eta_dis * dis);                 // pf.py:96-98
        if (ln.pq) {
            st_at<double>(e_vm, o_vm, v);
            st_at<uint32_t>(e_vw, o_volt, pack_warm(e, f));
        }
        if (is_bld && valid) {
            st_at<double>(e_agent, o_agent + AF_PRED * AFB, pred);
            st_at<double>(e_agent, o_agent + AF_CH * AFB, ch);
            st_at<double>(e_agent, o_agent + AF_DIS * AFB, dis);
            st_at<double>(e_agent, o_agent + AF_Q * AFB, q);
        }
    } else {                                                                           // env:314-328
        v = is_bus ? ld_at<double>(e_vm, o_vm) : 1.0;
        pred = is_bld ? ld_at<double>(e_agent, o_agent + AF_PRED * AFB) : 0.0;
        ch = is_bld ? ld_at<double>(e_agent, o_agent + AF_CH * AFB) : 0.0;
        dis = is_bld ? ld_at<double>(e_agent, o_agent + AF_DIS * AFB) : 0.0;
        q = is_bld ? ld_at<double>(e_agent, o_agent + AF_Q * AFB) : 0.0;
        e_new = e_cur;
    }
    if (is_bld && valid) {
        st_at<double>(e_agent, o_agent + AF_PCT * AFB, act.pct);
        st_at<double>(e_agent, o_agent + AF_E * AFB, e_new);                           // env:354: EINIT == E from here on (AF_ enum)
    }

    RewardOut rw = reward_terms<EPW>(cz, ln, is_bld, price, pred, ch, dis, q, v);        // env:330-335
    double* const b_cum = z.st.cumrew + env0;
    double rwd = rw.reward;
    if (!ok) rwd -= cz.fail_penalty;                                                    // env:336
    const int new_steps = steps + 1;                                                   // env:342
    const bool term = (new_steps >= cz.episode_limit) || !ok;                           // env:345
    // FLEX_STEP_AUTORESET: an environment that just terminated restarts inside this launch; its observation row
    // then holds the first observation of the new episode (the terminal observation is not materialised)
    const bool restart = auto_reset && term && valid;
    // The group's LAST lane holds the reward terms (reward_terms: the scan's totals stay where they fall) and writes what is
    // made of them; lane 0 writes the rest.  cumrew is the one cell a restart below writes as well — from the same lane.
    if (ln.l == LW - 1 && valid) {
        st_at<double>(reward + env0, g * 8, rwd);
        if (info) {
            double* const io = info + (int64_t)env0 * FLEX_INFO_W;
            const uint32_t oi = g * (FLEX_INFO_W * 8);
            st_at<double>(io, oi, rw.reward); st_at<double>(io, oi + 8, rw.revenue); st_at<double>(io, oi + 16, rw.der);
            st_at<double>(io, oi + 24, rw.ess); st_at<double>(io, oi + 32, rw.disc); st_at<double>(io, oi + 40, rw.vpen);
            st_at<double>(io, oi + 48, cum_before);                                    // A9
        }
        st_at<double>(b_cum, g * 8, cum_before + rwd);                                 // env:343
    }
    const int cnt_after = obs_cnt + ((want_obs && obs_fast && !restart) ? 1 : 0);
    if constexpr (MANY) {
        // (a restart rewrites all of this in memory behind here: the next step of the launch then loads it)
        mc->iv = flex_v4i{new_steps, start, new_row, cnt_after};
        mc->have = !(auto_reset && __ballot(restart) != 0ull);
        mc->pd = n_pd; mc->qd = n_qd; mc->ppv = n_ppv; mc->price = n_price;
        mc->e_cur = e_new; mc->e_init = e_init_r; mc->cum = cum_before + rwd;
        mc->vw = (ok && ln.bus >= 0) ? pack_warm(e, f) : vw_word;        // (the spare group of an odd batch mirrors environment 0)
    }
    if (ln.l == 0 && valid) {
        done[env] = term ? 1 : 0;
        if (failed) failed[env] = ok ? 0 : 1;
        // steps (env:342), start, row (env:340 reads row `steps`: A2) and the push count of the fast observation paths in ONE
        // 16-byte store, the solver statistics in one 8-byte store (five 4-byte stores until round 4); an environment that
        // restarts below overwrites them afterwards — same lane, program order
        st_at<flex_v4i>(e_ienv, o_ienv, flex_v4i{new_steps, start, new_row, cnt_after});
        st_at<flex_v2i>(e_ienv, o_ienv + IF_ITERS * 4, flex_v2i{iters, sweeps});
    }
    if constexpr (SINK) {
        // FLEX_STEP_REPLAY_SINK: this step's transition goes into the consumer's slab ring from here (include/flexenv.h) —
        // the small record into the slab the policy read, the masked recurrent state into the next one, the episode
        // statistics into per-environment running sums.  Everything is re-read from the kernarg segment: nothing of it
        // is live across the solve.
        const KArgs& zs = *relaunder_kernarg<KArgs>(kbase);
        const FlexReplaySink& sk = zs.sink;
        const int64_t p = sk_p;
        const int64_t pn = p + 1 >= zs.obs_slabs ? 0 : p + 1;
        const float keep = term ? 0.0f : 1.0f;
        if (valid) {
            sk_f4* hd = reinterpret_cast<sk_f4*>(sk.hid_ring + (pn * zs.n_envs + env) * (int64_t)sk.hid_w);
            const int h4 = sk.hid_w >> 2;
            float* const sm = sk.small_ring + (p * zs.n_envs + env) * (int64_t)sk.small_w;
#pragma unroll
            for (int r = 0; r < 3; ++r) { const int i = ln.l + LW * r; if (i < h4) hd[i] = sk_hv[r] * keep; }
            if (ln.l < sk.act_w) sm[ln.l] = sk_av;                                      // model.py:232
            if (ln.l == LW - 1) {                                                       // (the lane that holds the reward terms)
                const int na_ = cz.n_agents;
                for (int j = 0; j < na_; ++j) sm[sk.act_w + j] = (float)rwd;            // model.py:235: one reward, n copies
                sm[sk.act_w + na_] = 1.0f - keep;
                sm[sk.act_w + na_ + 1] = 1.0f - keep;
                // running sums: fp64 atomics without return (fire and forget; one lane per address and launch, so the order of
                // the additions is the launch order) — as plain += they were nine loads behind the solve and a round trip
                double* const ac = sk.acc + (int64_t)env * 10;
                unsafeAtomicAdd(ac + 0, rw.reward); unsafeAtomicAdd(ac + 1, rw.revenue); unsafeAtomicAdd(ac + 2, rw.der);
                unsafeAtomicAdd(ac + 3, rw.ess); unsafeAtomicAdd(ac + 4, rw.disc); unsafeAtomicAdd(ac + 5, rw.vpen);
                unsafeAtomicAdd(ac + 6, cum_before); unsafeAtomicAdd(ac + 7, rwd); unsafeAtomicAdd(ac + 8, ok ? 0.0 : 1.0);
            }
        }
        if (cells && blockIdx.x == 0 && threadIdx.x == 0) {
            if (sk.cursor_out) *sk.cursor_out = pn;
            if (sk.aux_counter) *sk.aux_counter += 1;
        }
    }
#ifdef FLEX_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    FLEX_STAMP(3);
    if (want_obs) {
        n_pd = is_bld ? n_pd : 0.0; n_qd = is_bld ? n_qd : 0.0; n_ppv = is_bld ? n_ppv : 0.0;
        const bool emit = valid && !restart;
        ObsT* const out = ROWS ? nullptr : obs;
        if (obs_fast) obs_store_new<EPW, ObsT>(z, env0, g, emit, ln, obs_cnt, n_pd, n_qd, n_ppv, v, n_price, e_new, out, rowrec);
        else push_and_emit_obs<EPW, ObsT>(z, env, emit, ln, obs_cnt, n_pd, n_qd, n_ppv, v, n_price, e_new, out, rowrec, false);
    }
    if (auto_reset && __ballot(restart) != 0ull) {       // wavefront-uniform, taken once per episode
        LaneNet ln0;
        if constexpr (MANY && FLEX_MANY_CARRY_NET) ln0 = mc->ln;      // (the carried rows: nothing second is live in the restart's solve)
        else load_lane_net<EPW>(a.net, lane, ln0);
        const DevResetSpec none = {nullptr, nullptr, nullptr, nullptr, nullptr};
        // the restart reads its configuration through a freshly "discovered" kernarg pointer: otherwise the compiler
        // loads every field the (rare) restart needs at kernel entry and carries them — spilled — across the hot path
        const KArgs& ar = *relaunder_kernarg<KArgs>(kbase);
        flex_reset_body<EPW, ObsT>(ar, env, restart, ln0, none, ROWS ? nullptr : obs, want_obs, rowrec, failed, true);
    }
    // launch counter for consumers that index by vector step (flexnet_rollout_pack's ring cursor): one lane of the whole
    // grid, pointer re-read from the kernarg segment so that it is not carried across the solve
    if (cells && blockIdx.x == 0 && threadIdx.x == 0) {
        const KArgs* const ac = relaunder_kernarg<KArgs>(kbase);
        int64_t* const sc = ac->step_counter;
        if (sc) {
            const int64_t nx = *sc + 1;
            *sc = (ac->step_modulo > 0 && nx >= ac->step_modulo) ? 0 : nx;
        }
    }
#ifdef FLEX_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    FLEX_STAMP(4);
    FLEX_STAMP_RT(6);
}

template <int EPW, typename ObsT, typename ActT, int NA_CAP, bool SINK = false, bool ROWS = false>
__global__ __launch_bounds__(FLEX_WAVE * FLEX_WAVES_PER_BLOCK, 8 / FLEX_WAVES_PER_BLOCK)
void flex_step_kernel(KArgs a, const ActT* __restrict__ actions, double* __restrict__ reward,
                      uint8_t* __restrict__ done, double* __restrict__ info, uint8_t* __restrict__ failed,
                      ObsT* __restrict__ obs, int want_obs, int auto_reset) {
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * FLEX_WAVES_PER_BLOCK + (threadIdx.x >> 6));
    flex_step_body<EPW, ObsT, ActT, NA_CAP, SINK, ROWS>(a, wave, actions, reward, done, info, failed, obs, want_obs, auto_reset, -1, true,
                                                  (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr());
}

// -------------------------------------------------------------------------------------------------
// flexenv_step_many: `n_steps` consecutive steps on a GIVEN action sequence in ONE launch — the vectorised form of the
// reference's open-loop episode runner (run_env.py:78-92: sampled actions, step(), per-step records).  Environments do not
// interact, so nothing orders one wavefront's steps against another's: each wavefront walks its own environments through the
// whole sequence — no launch boundary per step (2.6 us of drain + dispatch, and a launch ends with its slowest wavefront:
// a sequence of launches takes the SUM over steps of the per-step maximum, this kernel the maximum over wavefronts of their
// own sums).  The body is flex_step_body's: same loads, same stores, same arithmetic per step, state through HBM between
// steps (bit-identical to n_steps calls of flexenv_step: tests/test_step_many_gpu.py); what a wavefront carries in registers
// is StepCarry.  Step k reads action slab k mod act_period and writes row k of reward / done / info / failed.
// KArgs first: the step body re-reads it through relaunder_kernarg.
// -------------------------------------------------------------------------------------------------
struct ManyArgs {
    KArgs k;
    const void* actions;        // [act_period][N, n_agents, 4]
    double* reward;             // [n_steps][N]
    uint8_t* done;              // [n_steps][N]
    double* info;               // [n_steps][N, FLEX_INFO_W] or NULL
    uint8_t* failed;            // [n_steps][N] or NULL
    int32_t n_steps, act_period, auto_reset, carry;
};

template <int EPW, typename ActT>
__global__ __launch_bounds__(FLEX_WAVE * FLEX_WAVES_PER_BLOCK, 8 / FLEX_WAVES_PER_BLOCK)
void flex_step_many_kernel(ManyArgs m) {
    const unsigned long long kb = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * FLEX_WAVES_PER_BLOCK + (threadIdx.x >> 6));
    if (wave * EPW >= m.k.n_envs) return;
    StepCarry mc;
    mc.have = false;
    mc.iv = flex_v4i{0, 0, 0, 0};
    load_lane_net<EPW>(m.k.net, threadIdx.x & 63, mc.ln);
    const int n_steps = m.n_steps;
    int slot = 0;
    for (int k = 0; k < n_steps; ++k) {
        // (every step reads its arguments through a freshly derived kernarg pointer, as the step body's epilogue does:
        //  nothing of them is carried — spilled — across the solve)
        const ManyArgs& b = *relaunder_kernarg<ManyArgs>(kb);
        const int64_t ne = b.k.n_envs;
        const ActT* act = reinterpret_cast<const ActT*>(b.actions) + (int64_t)slot * ne * (b.k.cfg.n_agents * 4);
        if (!b.carry) mc.have = false;
        flex_step_body<EPW, float, ActT, FLEX_OBS_AGENTS_SMALL, false, true, true>(
            b.k, wave, act, b.reward + k * ne, b.done + k * ne, b.info ? b.info + k * ne * FLEX_INFO_W : nullptr,
            b.failed ? b.failed + k * ne : nullptr, nullptr, 1, b.auto_reset, -1, false, kb, &mc);
        slot = slot + 1 == b.act_period ? 0 : slot + 1;
        // this wavefront's own stores (state, history; a restart's) are what its next step loads: complete and visible
        // within the CU before those loads go out (work-group scope: one vector L1, write-through)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {                 // the launch counter, as n_steps single launches leave it
        const ManyArgs& b = *relaunder_kernarg<ManyArgs>(kb);
        int64_t* const sc = b.k.step_counter;
        if (sc) {
            int64_t nx = *sc + n_steps;
            if (b.k.step_modulo > 0) nx %= b.k.step_modulo;
            *sc = nx;
        }
    }
}


// -------------------------------------------------------------------------------------------------
// get_obs(): env:370-403 standalone
// -------------------------------------------------------------------------------------------------
template <int EPW, typename ObsT>
__global__ __launch_bounds__(FLEX_WAVE * FLEX_WAVES_PER_BLOCK)
void flex_obs_kernel(KArgs a, ObsT* __restrict__ obs) {
    EnvSlot<EPW> slot(a.n_envs);
    if (slot.wave_idle(a.n_envs)) return;
    const int lane = slot.lane, env = slot.env;
    LaneNet ln;
    load_lane_net<EPW>(a.net, lane, ln);
    const int nb = a.n_bus, na = a.cfg.n_agents;
    const int32_t* ie = a.st.ienv + (int64_t)env * IF_COUNT;
    const double* sr = a.series + clamp_row(ie[IF_ROW], a.rows) * a.cols;
    const bool is_bus = ln.bus >= 0, is_bld = ln.agent >= 0;
    const int ag = is_bld ? ln.agent : 0;
    const double* agst = a.st.agent + (int64_t)env * agent_rec_doubles(na);
    push_and_emit_obs<EPW, ObsT>(a, env, slot.valid, ln, ie[IF_OBSCNT], is_bus ? sr[ln.bus] : 0.0,
                                 is_bus ? sr[nb + ln.bus] : 0.0, is_bld ? sr[2 * nb + ag] : 0.0,
                                 is_bus ? a.st.vm[(int64_t)env * nb + ln.bus] : 0.0, sr[2 * nb + na],
                                 is_bld ? agst[AF_E * na + ag] : 0.0, obs, nullptr, false);
}

// The stacked observation the last push left (env:387-401), WITHOUT pushing a row: what step(FLEX_STEP_OBS_ROWS) + a read
// of the environment's mirror ring amounts to, materialised [N, n_agents, 6 * history] for consumers that want the copy
// (the N = 1 drop-in view, tests, evaluation).  One thread per output element; not a hot kernel.
template <typename ObsT>
__global__ __launch_bounds__(256) void flex_obs_view_kernel(KArgs a, ObsT* __restrict__ obs) {
    const int H = a.cfg.history, na = a.cfg.n_agents, w = H * 6;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)a.n_envs * na * w) return;
    const int64_t row = idx / w;                       // env * na + agent
    const int rem = (int)(idx - row * w);
    const int cnt = a.st.ienv[(row / na) * IF_COUNT + IF_OBSCNT];
    const int start = cnt > 0 ? (cnt - 1) % H + 1 : 0;
    obs[idx] = cnt > 0 ? (ObsT)a.st.ring[row * (2 * w) + start * 6 + rem] : (ObsT)0;
}

// get_state(): env:358-368  [Pd | Qd | Ppv | V | price | E]   (one wavefront per environment; not a hot kernel)
__global__ __launch_bounds__(FLEX_WAVE * FLEX_WAVES_PER_BLOCK)
void flex_state_kernel(KArgs a, double* __restrict__ state) {
    const int lane = threadIdx.x & 63;
    const int env = blockIdx.x * FLEX_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (env >= a.n_envs) return;
    const int nb = a.n_bus, na = a.cfg.n_agents;
    const int32_t* ie = a.st.ienv + (int64_t)env * IF_COUNT;
    const double* sr = a.series + clamp_row(ie[IF_ROW], a.rows) * a.cols;
    const int width = 3 * nb + 2 * na + 1;
    double* o = state + (int64_t)env * width;
    const double* agst = a.st.agent + (int64_t)env * agent_rec_doubles(na);
    for (int i = lane; i < width; i += FLEX_WAVE) {
        double v;
        if (i < 2 * nb + na) v = sr[i];
        else if (i < 3 * nb + na) v = a.st.vm[(int64_t)env * nb + (i - 2 * nb - na)];
        else if (i == 3 * nb + na) v = sr[2 * nb + na];
        else v = agst[AF_E * na + (i - 3 * nb - na - 1)];
        o[i] = v;
    }
}

// -------------------------------------------------------------------------------------------------
// power_flow_solver_simplified on a batch: utils/pf.py:115-192
// -------------------------------------------------------------------------------------------------
template <int EPW>
__global__ __launch_bounds__(FLEX_WAVE * FLEX_WAVES_PER_BLOCK)
void pf_batch_kernel(const DevNet* __restrict__ net, int n, int nb, const double* __restrict__ pnet,
                     const double* __restrict__ qnet, double* __restrict__ v, double* __restrict__ isqr,
                     double* __restrict__ pl, double* __restrict__ ql, int32_t* __restrict__ iters_out,
                     uint8_t* __restrict__ failed, double tol, int max_iter, int solver) {
    EnvSlot<EPW> slot(n);
    if (slot.wave_idle(n)) return;
    const int lane = slot.lane, i = slot.env;
    LaneNet ln;
    load_lane_net<EPW>(net, lane, ln);
    ln.pq = ln.pq && slot.valid;
    const double p = ln.pq ? pnet[(int64_t)i * nb + ln.bus] : 0.0;
    const double q = ln.pq ? qnet[(int64_t)i * nb + ln.bus] : 0.0;
    double e = 1.0, f = 0.0;
    int iters = 0, sweeps = 0;
    const bool ok = pf_solve<EPW>(net, ln, solver, p, q, e, f, tol, max_iter, iters, sweeps);
    // line quantities in the receiving-end convention of pf.py:85-88
    const double ep0 = __shfl(e, ln.par, FLEX_WAVE), fp0 = __shfl(f, ln.par, FLEX_WAVE);
    if (ln.pq) {
        const double ep = ln.par_slack ? 1.0 : ep0, fp = ln.par_slack ? 0.0 : fp0;
        v[(int64_t)i * nb + ln.bus] = sqrt(e * e + f * f);
        const double de = ep - e, df = fp - f;
        const double jr = ln.g * de - ln.b * df, ji = ln.b * de + ln.g * df;
        if (isqr) isqr[(int64_t)i * nb + ln.bus] = jr * jr + ji * ji;
        if (pl) pl[(int64_t)i * nb + ln.bus] = e * jr + f * ji;
        if (ql) ql[(int64_t)i * nb + ln.bus] = f * jr - e * ji;
    }
    if (ln.l == 0 && slot.valid) {
        const int sb = net->slack_bus;
        v[(int64_t)i * nb + sb] = 1.0;
        if (isqr) isqr[(int64_t)i * nb + sb] = 0.0;
        if (pl) pl[(int64_t)i * nb + sb] = 0.0;
        if (ql) ql[(int64_t)i * nb + sb] = 0.0;
        if (iters_out) iters_out[i] = iters + 1000 * sweeps;   // Newton steps + 1000 * sweeps
        if (failed) failed[i] = ok ? 0 : 1;
    }
}

// The dense-LU Newton variant of the same solve (FLEX_SOLVER_DENSE): one wavefront per environment, the 64 x 64
// Jacobian in 32 KB of LDS.  Reference variant for the measurement table only.
__global__ __launch_bounds__(FLEX_WAVE)
void pf_batch_dense_kernel(const DevNet* __restrict__ net, int n, int nb, const double* __restrict__ pnet,
                           const double* __restrict__ qnet, double* __restrict__ v, int32_t* __restrict__ iters_out,
                           uint8_t* __restrict__ failed, double tol, int max_iter) {
    __shared__ double A[64 * 64];
    const int lane = threadIdx.x & 63, i = blockIdx.x;
    if (i >= n) return;
    LaneNet ln;
    load_lane_net<1>(net, lane, ln);
    const double p = ln.pq ? pnet[(int64_t)i * nb + ln.bus] : 0.0;
    const double q = ln.pq ? qnet[(int64_t)i * nb + ln.bus] : 0.0;
    double e = 1.0, f = 0.0;
    int iters = 0;
    const bool ok = pf_newton_dense(net, ln, p, q, e, f, tol, max_iter, iters, A);
    if (ln.pq) v[(int64_t)i * nb + ln.bus] = sqrt(e * e + f * f);
    if (lane == 0) {
        v[(int64_t)i * nb + net->slack_bus] = 1.0;
        if (iters_out) iters_out[i] = iters;
        if (failed) failed[i] = ok ? 0 : 1;
    }
}

// -------------------------------------------------------------------------------------------------
// Safety layer, madrl/models/safemaddpg.py:142-299, one lane per (env, building).
// with V_pred(bus) = sP*P_net[bus] + sQ*Q_net[bus] + beta (own-bus only,
// safemaddpg.py:266,272) the QP separates per building into
//     min |x - x0|^2 + rho*(s_lo + s_up)   s.t.  v_min - s_lo <= c.x + d <= v_max + s_up,
//     x = (pr, ch, dis, q), pr,ch,dis >= 0, q free, s >= 0,
//     c = (-sP*Pd, sP, -sP, sQ),  d = sP*Pd + sQ*Qd + beta.
// For each side of the slab the minimiser is found by active-set enumeration over the three
// sign constraints (8 subsets), each subset solved in closed form; the exact L1 penalty caps the
// multiplier at rho (beyond it the slack absorbs the rest).
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void qp_side(const double x0[4], const double cvec[4], double d, double bound,
                                        double sign /* +1: c.x+d <= bound ; -1: c.x+d >= bound */,
                                        double rho, double xout[4], bool& changed) {
    // constraint h(x) = sign*(c.x + d - bound) <= slack, slack >= 0 with cost rho*slack.
    // Start from the projection of x0 on the sign constraints alone.
    double best[4], bestobj = 1e300;
    bool found = false;
    double xfree[4] = {fmax(x0[0], 0.0), fmax(x0[1], 0.0), fmax(x0[2], 0.0), x0[3]};
    double h0 = sign * (cvec[0] * xfree[0] + cvec[1] * xfree[1] + cvec[2] * xfree[2] + cvec[3] * xfree[3] + d - bound);
    if (h0 <= 0.0) {
        for (int t = 0; t < 4; ++t) xout[t] = xfree[t];
        changed = false;
        return;
    }
    changed = true;
    // enumerate active sets A (bits over x0..x2 fixed at 0)
    for (int A = 0; A < 8; ++A) {
        double cc = 0.0, cx = 0.0;
        for (int t = 0; t < 4; ++t) {
            const bool fixed = (t < 3) && ((A >> t) & 1);
            if (!fixed) { cc += cvec[t] * cvec[t]; cx += cvec[t] * x0[t]; }
        }
        // x = x0 - lam*sign*c/2 on free coords; h(x) = sign*(cx + d - bound) - lam*cc/2 = slack
        const double hfree = sign * (cx + d - bound);
        double lam;               // multiplier of the slab constraint, 0 <= lam <= rho
        if (cc > 0.0) lam = fmin(fmax(2.0 * hfree / cc, 0.0), rho);
        else lam = (hfree > 0.0) ? rho : 0.0;
        double x[4];
        bool feas = true;
        for (int t = 0; t < 4; ++t) {
            const bool fixed = (t < 3) && ((A >> t) & 1);
            x[t] = fixed ? 0.0 : x0[t] - 0.5 * lam * sign * cvec[t];
            if (t < 3 && x[t] < -1e-15) feas = false;
        }
        if (!feas) continue;
        double h = sign * (cvec[0] * x[0] + cvec[1] * x[1] + cvec[2] * x[2] + cvec[3] * x[3] + d - bound);
        const double slack = fmax(h, 0.0);
        double obj = rho * slack;
        for (int t = 0; t < 4; ++t) obj += (x[t] - x0[t]) * (x[t] - x0[t]);
        if (obj < bestobj) { bestobj = obj; found = true; for (int t = 0; t < 4; ++t) best[t] = x[t]; }
    }
    for (int t = 0; t < 4; ++t) xout[t] = found ? fmax(best[t], t < 3 ? 0.0 : -1e300) : xfree[t];
}

// one (environment, building) of the safety layer; the kernel below maps threads to them, the rollout burst its wavefronts' lanes
__device__ __forceinline__ void flex_safety_one(const KArgs& a, const int env, const int ag, const void* __restrict__ proposed,
                                                int dtype, const double* __restrict__ s_p, const double* __restrict__ s_q,
                                                const double* __restrict__ beta, double v_min, double v_max, double rho,
                                                double* __restrict__ adjusted, uint8_t* __restrict__ intervened,
                                                float* __restrict__ env_action, float act_low, float act_span) {
    const int na = a.cfg.n_agents;
    const FlexCfg& c = a.cfg;
    const int nb = a.n_bus;
    const int32_t* ie = a.st.ienv + (int64_t)env * IF_COUNT;
    const double* sr = a.series + clamp_row(ie[IF_ROW], a.rows) * a.cols;
    const int bus = a.net->bus_of_lane[a.net->lane_of_agent[ag]];
    const double pd = sr[bus], qd = sr[nb + bus], ppv = sr[2 * nb + ag];
    const double* agst = a.st.agent + (int64_t)env * agent_rec_doubles(na);
    const double e_cur = agst[AF_E * na + ag];
    const int64_t base = ((int64_t)env * na + ag) * 4;
    // parse_actions, safemaddpg.py:142-174: always the scaled branch, clip vs current_ess_energy
    FlexAct p = parse_actions(c, a.inv_eta_ch, a.inv_eta_dis, false, load_action(proposed, dtype, base), load_action(proposed, dtype, base + 1),
                              load_action(proposed, dtype, base + 2), load_action(proposed, dtype, base + 3),
                              pd, ppv, e_cur);
    const double x0[4] = {p.pct, p.ch, p.dis, p.q};
    const double sp = s_p[ag], sq = s_q[ag];
    const double cvec[4] = {-sp * pd, sp, -sp, sq};
    const double d = sp * pd + sq * qd + beta[ag];
    double x[4];
    bool ch_up = false, ch_lo = false;
    qp_side(x0, cvec, d, v_max, +1.0, rho, x, ch_up);
    if (!ch_up) qp_side(x0, cvec, d, v_min, -1.0, rho, x, ch_lo);
    double* o = adjusted + (int64_t)env * 4 * na;          // type-major, safemaddpg.py:297 (A13)
    o[0 * na + ag] = x[0]; o[1 * na + ag] = x[1]; o[2 * na + ag] = x[2]; o[3 * na + ag] = x[3];
    if (env_action) {
        // what the caller would feed env.step: translate_action (util.py:125-128) of the fp32 cast of the same flat vector,
        // 0.5 (clamp(a, low, high) + 1) (high - low) + low, every step rounded to fp32 as the tensor ops round it
        float* eo = env_action + (int64_t)env * 4 * na;
        const float act_high = act_low + act_span;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float c = fminf(fmaxf((float)x[t], act_low), act_high);
            eo[t * na + ag] = __fadd_rn(__fmul_rn(__fmul_rn(0.5f, __fadd_rn(c, 1.0f)), act_span), act_low);
        }
    }
    if (intervened && (ch_up || ch_lo)) intervened[env] = 1;
}

__global__ void flex_safety_kernel(KArgs a, const void* __restrict__ proposed, int dtype,
                                   const double* __restrict__ s_p, const double* __restrict__ s_q,
                                   const double* __restrict__ beta, double v_min, double v_max, double rho,
                                   double* __restrict__ adjusted, uint8_t* __restrict__ intervened,
                                   float* __restrict__ env_action, float act_low, float act_span) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int na = a.cfg.n_agents;
    if (tid >= a.n_envs * na) return;
    const int env = tid / na;
    flex_safety_one(a, env, tid - env * na, proposed, dtype, s_p, s_q, beta, v_min, v_max, rho, adjusted, intervened, env_action,
                    act_low, act_span);
}

// -------------------------------------------------------------------------------------------------
// A burst of rollout steps in ONE persistent launch (round 3): block b owns environments 16 b .. 16 b + 15 as two groups
// of four wavefronts (two environments per wavefront for the step, up to three 16-row tiles per group for the policy:
// csrc/actor_r16.h, actor_r16_burst), each group alternating policy evaluation and environment step for `n_steps` steps.  The policy's weights (156 KB) are staged
// into the CU's LDS once per burst instead of once per step (a third of a rollout-size policy call), nothing is launched
// between the two halves of a step, and the hand-overs (env action, new hidden state, observation) stay in the CU's L2.
// The arithmetic of both halves is the code of the two stand-alone kernels: same results bit for bit
// (tests/test_rollout_gpu.py).  KArgs is the first parameter: the step body re-reads it through relaunder_kernarg.
// -------------------------------------------------------------------------------------------------
// Producer and consumer of every hand-over are wavefronts of ONE work-group, i.e. of one CU, whose vector L1 they share
// (write-through: a store updates the line in place): work-group scope is all the ordering needed — stores issued and
// complete (release: s_waitcnt), everybody there (barrier).  An AGENT-scope fence here writes back the XCD's whole L2
// (buffer_wbl2) twice per step: measured 180 us per vector step instead of 30.
__device__ __forceinline__ void burst_handover() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Everything the burst needs, as ONE kernel parameter: the environment step below is a real call (its own register
// allocation: inlined into the policy's body it shared 256 VGPRs with it, spilled 150 of them, and the reloads — which wait
// on the memory counter — serialised the step's load phases: 54 k cycles instead of the stand-alone kernel's 26 k) and reads
// its arguments from the kernarg segment with scalar loads instead of receiving them in vector registers.
struct BurstArgs {
    KArgs k;                    // first: flex_step_body re-reads it through relaunder_kernarg
    FlexActorArgs act;
    double* reward;
    uint8_t* done;
    double* info;
    uint8_t* failed;
    float* obs_ring;
    int n_steps;
    // SAFEMADDPG (safemaddpg.py:90-111): the safety layer between policy and environment — each wavefront projects the
    // proposed actions of its own two environments (flex_safety_one) and steps them on the result
    int safety;                 // 0: the step reads the policy's env action
    const double* s_p;
    const double* s_q;
    const double* beta;
    double v_min, v_max, rho;
    double* adjusted;           // [N, 4 n_agents] (the replay keeps the policy's own action: nothing reads this)
    float* safe_env_action;     // [N, 4 n_agents] translate_action of the projected vector: what the step reads
    float act_low, act_span;
};

// (Inlined again since the policy became actor_r16_burst, which keeps almost nothing live across the step: 12 spilled
//  registers, none in the step's loops, against the 64 the called version saved and restored per step — 34.4 -> 31.8 us per
//  vector step.  The argument plumbing of the called version stays: it is what keeps the step's scalars out of the policy's.)
template <int NA_CAP, bool SAFE>
__device__ __forceinline__ void flex_burst_env_step(int slab_v, unsigned kbase_lo, unsigned kbase_hi) {
    // (arguments of a call arrive in vector registers: back to scalars)
    const unsigned long long kbase = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)kbase_lo) |
                                     ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)kbase_hi) << 32);
    const BurstArgs& b = *relaunder_kernarg<BurstArgs>(kbase);
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * R16_W + (threadIdx.x >> 6));
    const int64_t slab = __builtin_amdgcn_readfirstlane(slab_v);
    // the step is one long dependent chain of short fp64 operations; the wavefront it shares its SIMD with is (mostly) in the
    // other group's policy phase, a dense stream of matrix and LDS instructions that fills every issue slot it is given:
    // the chain goes first, the stream takes the gaps
    __builtin_amdgcn_s_setprio(3);
    const float* actions = b.act.env_action;
    if constexpr (SAFE) {
        const int na = b.k.cfg.n_agents, lane = threadIdx.x & 63;
        const int env = 2 * wave + lane / na;
        if (lane < 2 * na && env < b.k.n_envs)
            flex_safety_one(b.k, env, lane % na, b.act.action, FLEX_F32, b.s_p, b.s_q, b.beta, b.v_min, b.v_max, b.rho, b.adjusted,
                            nullptr, b.safe_env_action, b.act_low, b.act_span);
        // (this wavefront's own stores, read back by its own step: complete before the loads go out)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        actions = b.safe_env_action;
    }
    flex_step_body<2, float, float, NA_CAP, true, true>(b.k, wave, actions, b.reward, b.done, b.info, b.failed, b.obs_ring, 1, 1,
                                                  slab, false, kbase);
    __builtin_amdgcn_s_setprio(0);
}

template <int NA_CAP, bool SAFE>
__global__ __launch_bounds__(64 * R16_W)
void flex_rollout_burst_kernel(BurstArgs b) {
    __shared__ ActorLds16B s;
    const unsigned long long kb = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    auto env_step = [&](int64_t slab) { flex_burst_env_step<NA_CAP, SAFE>((int)slab, (unsigned)kb, (unsigned)(kb >> 32)); };
#ifdef FLEX_STAMPS
    // diagnostic build: phase boundaries of the LAST step, per wavefront, in slots 8-12 of its first environment's row
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * R16_W + (threadIdx.x >> 6));
    actor_r16_burst(b.act, s, b.n_steps, env_step, [&](int slot, int step) {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (b.k.stamps && (threadIdx.x & 63) == 0 && 2 * wave < b.k.n_envs && step == b.n_steps - 1)
            b.k.stamps[(int64_t)(2 * wave) * 16 + slot] = t;
    });
#else
    actor_r16_burst(b.act, s, b.n_steps, env_step, [](int, int) {});
#endif
}

// the cells a burst leaves as `steps` single steps would: cell 0 = the slab the policy reads next, cell 1 = the slab the last
// step filed into, the noise stream's step counter advanced
__global__ void flex_burst_finish_kernel(int64_t* cell0, int64_t* cell1, uint64_t* rng_state, int steps, int slabs) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int64_t p0 = *cell0;
    *cell0 = (p0 + steps) % slabs;
    *cell1 = (p0 + steps - 1) % slabs;
    if (rng_state) rng_state[1] += (uint64_t)steps;
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
static int build_devnet(const NetFix* nf, int n_agents, DevNet* dn) {
    const int n = nf->n_bus;
    if (n < 2 || n > FLEX_MAX_BUS || n_agents < 1 || n_agents > FLEX_MAX_AGENTS) return FLEX_EINVAL;
    if (nf->max_children < 1 || nf->max_children > FLEX_MAX_CHILDREN) return FLEX_EINVAL;
    if (nf->slack < 0 || nf->slack >= n) return FLEX_EINVAL;
    memset(dn, 0, sizeof(*dn));
    const int npq = n - 1;
    dn->n_bus = n; dn->n_pq = npq; dn->n_levels = nf->n_levels; dn->n_agents = n_agents; dn->slack_bus = nf->slack;
    dn->epw = (npq <= FLEX_WAVE / 2) ? 2 : 1;
    dn->sweep_tol_frac = 0.5f;
    if (const char* fr = getenv("FLEX_SWEEP_TOL_FRAC")) {         // measurement knob, not part of the ABI
        const float v = (float)atof(fr);
        if (v > 0.0f && v <= 1.0f) dn->sweep_tol_frac = v;
    }
    // depth-first preorder of the PQ buses (the slack gets no lane): the first child of a bus lands in the next lane
    std::vector<int> order, stack;
    for (int k = nf->max_children - 1; k >= 0; --k) {
        const int cidx = nf->child[nf->slack * nf->max_children + k];
        if (cidx >= 0) { if (cidx >= n) return FLEX_EINVAL; stack.push_back(cidx); }
    }
    while (!stack.empty()) {
        const int u = stack.back(); stack.pop_back();
        order.push_back(u);
        if ((int)order.size() > npq) return FLEX_EINVAL;
        for (int k = nf->max_children - 1; k >= 0; --k) {
            const int cidx = nf->child[u * nf->max_children + k];
            if (cidx >= 0) { if (cidx >= n || cidx == nf->slack) return FLEX_EINVAL; stack.push_back(cidx); }
        }
    }
    if ((int)order.size() != npq) return FLEX_EINVAL;
    for (int l = 0; l < FLEX_MAX_BUS; ++l) {
        dn->bus_of_lane[l] = -1; dn->lane_of_bus[l] = -1; dn->par_lane[l] = l; dn->par_slack[l] = 0; dn->level[l] = -1;
        dn->agent_of_lane[l] = -1; dn->gd[l] = 1.0; dn->sub_end[l] = l;
        dn->seg_start[l] = l; dn->seg_par[l] = l; dn->seg_depth[l] = 0;
        for (int k = 0; k < FLEX_MAX_CHILDREN; ++k) dn->child_lane[k][l] = -1;
        for (int k = 0; k < FLEX_JUMP_ROUNDS; ++k) dn->anc[k][l] = -1;
    }
    for (int l = 0; l < npq; ++l) { dn->bus_of_lane[l] = order[l]; dn->lane_of_bus[order[l]] = l; }
    if (nf->parent[nf->slack] != -1 || nf->level[nf->slack] != 0) return FLEX_EINVAL;
    int maxlev = 0;
    for (int l = 0; l < npq; ++l) {
        const int b = order[l], p = nf->parent[b];
        if (nf->level[b] < 1 || nf->level[b] >= FLEX_MAX_BUS) return FLEX_EINVAL;
        if (p < 0 || p >= n || nf->level[p] != nf->level[b] - 1) return FLEX_EINVAL;
        dn->level[l] = nf->level[b];
        if (nf->level[b] > maxlev) maxlev = nf->level[b];
        if (p == nf->slack) { dn->par_slack[l] = 1; dn->par_lane[l] = l; }
        else {
            dn->par_lane[l] = dn->lane_of_bus[p];
            if (dn->par_lane[l] >= l) return FLEX_EINVAL;     // preorder: parents come first
        }
        const double r = nf->r[b], x = nf->x[b], z2 = r * r + x * x;
        if (!(z2 > 0.0)) return FLEX_EINVAL;
        dn->r[l] = r; dn->x[l] = x;
        dn->g[l] = r / z2; dn->b[l] = -x / z2;
    }
    if (maxlev + 1 != nf->n_levels) return FLEX_EINVAL;
    for (int l = 0; l < npq; ++l) { dn->gd[l] = dn->g[l]; dn->bd[l] = dn->b[l]; }
    int maxc = 1;
    for (int l = 0; l < npq; ++l) {
        const int b = order[l];
        int used = 0;
        for (int k = 0; k < nf->max_children; ++k) {
            const int cidx = nf->child[b * nf->max_children + k];
            if (cidx < 0) continue;
            if (nf->parent[cidx] != b) return FLEX_EINVAL;
            const int cl = dn->lane_of_bus[cidx];
            dn->child_lane[used++][l] = cl;
            dn->gd[l] += dn->g[cl]; dn->bd[l] += dn->b[cl];
        }
        if (used > maxc) maxc = used;
        const int lev = dn->level[l];
        if (lev + 1 < FLEX_MAX_BUS && used > dn->slots_at_level[lev + 1]) dn->slots_at_level[lev + 1] = used;
    }
    dn->max_children = maxc;
    // sweep-solver tables: subtree ranges (preorder => contiguous), chain segments, 2^k-th ancestors
    {
        std::vector<int> size(FLEX_MAX_BUS, 1);
        for (int l = npq - 1; l >= 0; --l) if (!dn->par_slack[l]) size[dn->par_lane[l]] += size[l];
        int max_depth = 0;
        for (int l = 0; l < npq; ++l) {
            dn->sub_end[l] = l + size[l] - 1;
            const bool cont = (l > 0) && !dn->par_slack[l] && (dn->par_lane[l] == l - 1);
            if (cont) {
                dn->seg_start[l] = dn->seg_start[l - 1]; dn->seg_par[l] = dn->seg_par[l - 1];
                dn->seg_depth[l] = dn->seg_depth[l - 1];
            } else if (dn->par_slack[l]) {
                dn->seg_start[l] = l; dn->seg_par[l] = l; dn->seg_depth[l] = 0;
            } else {
                dn->seg_start[l] = l; dn->seg_par[l] = dn->par_lane[l];
                dn->seg_depth[l] = dn->seg_depth[dn->par_lane[l]] + 1;
            }
            if (dn->seg_depth[l] > max_depth) max_depth = dn->seg_depth[l];
            dn->anc[0][l] = dn->par_slack[l] ? -1 : dn->par_lane[l];
        }
        dn->n_seg_rounds = max_depth;
        for (int k = 1; k < FLEX_JUMP_ROUNDS; ++k)
            for (int l = 0; l < npq; ++l) {
                const int h = dn->anc[k - 1][l];
                dn->anc[k][l] = (h >= 0) ? dn->anc[k - 1][h] : -1;
            }
        int rounds = 0;
        while ((1 << rounds) < nf->n_levels - 1) ++rounds;
        dn->n_jump_rounds = rounds;
        if (rounds > FLEX_JUMP_ROUNDS) return FLEX_EINVAL;
    }
    for (int a = 0; a < n_agents; ++a) {
        const int b = nf->agent_bus[a];
        if (b < 0 || b >= n || b == nf->slack) return FLEX_EINVAL;
        const int l = dn->lane_of_bus[b];
        if (dn->agent_of_lane[l] >= 0) return FLEX_EINVAL;   // one building per bus
        dn->agent_of_lane[l] = a; dn->lane_of_agent[a] = l;
    }
    return FLEX_OK;
}

// Calibration of pf_sweep's two-sweep extrapolation (flex_device.h): for the loading `pnet`, `qnet` (per bus, pu) solve the
// feeder on the host (dense Z-bus fixed point, n <= 64), take the dominant eigenvalue mu_1 of the two-sweep error map
// A^2 = M conj(M), M = -Z diag(conj(S) / conj(V)^2), by power iteration, and express it through the largest voltage drop:
// acc_lane = argmax |1 - V|, acc_kappa = mu_1 / |1 - V[acc_lane]|^2.  The ratio depends on how the load is distributed over
// the feeder, hardly on its level; the kernel multiplies it by the drop it sees.  Leaves the extrapolation off (kappa = 0)
// when the loading is degenerate (no drop, no convergence).
#include <complex>
static void calibrate_sweep_accel(DevNet* dn, const double* pnet, const double* qnet) {
    typedef std::complex<double> cd;
    const int n = dn->n_pq;
    dn->acc_lane = 0; dn->acc_kappa = 0.0f;
    if (n < 1 || dn->n_seg_rounds > 2) return;
    std::vector<cd> Z((size_t)n * n), S(n), V(n, cd(1.0, 0.0)), I(n), x(n), y(n), t(n);
    // Z[i][j] = impedance of the common part of the paths slack -> i and slack -> j (lanes are in preorder: ancestors first)
    std::vector<std::vector<char>> on_path(n, std::vector<char>(n, 0));
    for (int i = 0; i < n; ++i) {
        int l = i;
        for (;;) { on_path[i][l] = 1; if (dn->par_slack[l]) break; l = dn->par_lane[l]; }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            cd z(0.0, 0.0);
            for (int l = 0; l < n; ++l) if (on_path[i][l] && on_path[j][l]) z += cd(dn->r[l], dn->x[l]);
            Z[(size_t)i * n + j] = z;
        }
    double tot = 0.0;
    for (int i = 0; i < n; ++i) { S[i] = -cd(pnet[dn->bus_of_lane[i]], qnet[dn->bus_of_lane[i]]); tot += std::abs(S[i]); }
    if (!(tot > 0.0)) return;
    auto matvec = [&](const std::vector<cd>& in, std::vector<cd>& out) {
        for (int i = 0; i < n; ++i) { cd a(0.0, 0.0); for (int j = 0; j < n; ++j) a += Z[(size_t)i * n + j] * in[j]; out[i] = a; }
    };
    for (int it = 0; it < 60; ++it) {                          // V <- 1 + Z conj(S / V)
        for (int i = 0; i < n; ++i) I[i] = std::conj(S[i] / V[i]);
        matvec(I, t);
        for (int i = 0; i < n; ++i) V[i] = cd(1.0, 0.0) + t[i];
    }
    for (int i = 0; i < n; ++i) if (!std::isfinite(V[i].real()) || !std::isfinite(V[i].imag()) || !(std::abs(V[i]) > 0.3)) return;
    // A(e) = -Z conj(S e / V^2)  (anti-linear); power iteration on A^2
    auto apply_a = [&](const std::vector<cd>& in, std::vector<cd>& out) {
        for (int i = 0; i < n; ++i) I[i] = -std::conj(S[i] * in[i] / (V[i] * V[i]));
        matvec(I, out);
    };
    for (int i = 0; i < n; ++i) x[i] = cd(1.0, 0.1 * (i % 3));
    double mu = 0.0;
    for (int it = 0; it < 200; ++it) {
        apply_a(x, t); apply_a(t, y);
        cd num(0.0, 0.0); double den = 0.0, ny = 0.0;
        for (int i = 0; i < n; ++i) { num += std::conj(x[i]) * y[i]; den += std::norm(x[i]); ny += std::norm(y[i]); }
        mu = num.real() / den;
        if (!(ny > 0.0)) return;
        const double sc = 1.0 / std::sqrt(ny);
        for (int i = 0; i < n; ++i) x[i] = y[i] * sc;
    }
    int best = 0; double drop = 0.0;
    for (int i = 0; i < n; ++i) { const double d = std::norm(cd(1.0, 0.0) - V[i]); if (d > drop) { drop = d; best = i; } }
    if (!(mu > 0.0) || !(drop > 1e-8) || !(mu < 0.25)) return;
    dn->acc_lane = best;
    dn->acc_kappa = (float)(mu / drop);
}

static unsigned long long* g_stamps = nullptr;
extern "C" void flexenv_debug_set_stamps(unsigned long long* dev) { g_stamps = dev; }

static KArgs make_args(const FlexEnv* e) {
    KArgs k;
    k.stamps = g_stamps;
    k.cfg = e->cfg; k.net = e->net; k.st = e->st; k.series = e->series.table;
    k.rows = e->series.rows; k.cols = e->series.cols; k.n_envs = e->n_envs; k.n_bus = e->n_bus;
    k.row_bytes = e->series.cols * 8;
    k.inv_h = 1.0f / (float)e->cfg.history; k.inv_h3 = 1.0f / (float)(3 * e->cfg.history);
    k.inv_eta_ch = 1.0 / e->cfg.eta_ch; k.inv_eta_dis = 1.0 / e->cfg.eta_dis;
    k.step_counter = nullptr;                      // only flexenv_step hands these on
    k.step_modulo = 0; k.obs_cursor = nullptr; k.obs_slab_stride = 0; k.obs_slabs = 0;
    memset(&k.sink, 0, sizeof(k.sink));
    return k;
}

// n environments at `epw` environments per wavefront, FLEX_WAVES_PER_BLOCK wavefronts per block
static inline dim3 env_grid(int n, int epw = 1) {
    const int waves = (n + epw - 1) / epw;
    return dim3((waves + FLEX_WAVES_PER_BLOCK - 1) / FLEX_WAVES_PER_BLOCK);
}
static inline dim3 env_block() { return dim3(FLEX_WAVE * FLEX_WAVES_PER_BLOCK); }

extern "C" {

const char* flexenv_version(void) { return "flexenv-hip 0.2 (gfx950)"; }
int32_t flexenv_abi_version(void) { return FLEX_ABI_VERSION; }

int flexenv_create(const FlexCfg* cfg, const NetFix* net, const SeriesTab* series, int32_t n_envs,
                   int32_t device, FlexEnv** out) {
    if (!cfg || !net || !series || !out || n_envs < 1) return FLEX_EINVAL;
    if (cfg->n_agents < 1 || cfg->n_agents > FLEX_MAX_AGENTS || cfg->history < 1 || cfg->episode_limit < 1)
        return FLEX_EINVAL;
    if (cfg->solver != FLEX_SOLVER_TREE && cfg->solver != FLEX_SOLVER_SWEEP) return FLEX_EINVAL;
    if (series->cols != 2 * net->n_bus + cfg->n_agents + 1 || series->rows < 2 || !series->table) return FLEX_EINVAL;
    if (cfg->per_hour < 1 || cfg->n_start_days < 1) return FLEX_EINVAL;
    // the step kernel addresses the series table with 32-bit byte offsets (4 GB = two centuries of 15-minute rows)
    if (series->rows >= (1LL << 31) || (int64_t)series->rows * series->cols * 8 >= (1LL << 32)) return FLEX_EINVAL;
    // every reachable row must exist: start + 1 + (episode_limit + history)
    const int64_t max_start = (int64_t)(cfg->per_hour - 1) + 23LL * cfg->per_hour +
                              (int64_t)(cfg->n_start_days - 1) * 24 * cfg->per_hour;
    if (max_start + cfg->episode_limit + 1 >= series->rows) return FLEX_EINVAL;
    FlexEnv* e = new (std::nothrow) FlexEnv();
    if (!e) return FLEX_ENOMEM;
    memset(e, 0, sizeof(*e));
    e->cfg = *cfg; e->n_envs = n_envs; e->n_bus = net->n_bus; e->device = device; e->series = *series;
    int rc = build_devnet(net, cfg->n_agents, &e->hnet);
    if (rc != FLEX_OK) { delete e; return rc; }
    // allocate everything or nothing
    const int64_t N = n_envs;
    const size_t sz_vm = N * net->n_bus * sizeof(double), sz_v = N * 64 * sizeof(uint32_t);   // LW <= 64 per environment
    const size_t sz_ag = N * agent_rec_doubles(cfg->n_agents) * sizeof(double), sz_cr = N * sizeof(double);
    const size_t sz_ie = N * IF_COUNT * sizeof(int32_t), sz_ring = N * cfg->n_agents * cfg->history * 12 * sizeof(float);
    hipError_t err = hipSetDevice(device);
    if (err == hipSuccess && !cfg->no_sweep_accel && cfg->solver == FLEX_SOLVER_SWEEP) {
        // sweep acceleration: calibrated on the mean loading of eight rows spread over one day in the middle of the series
        // (demand columns only: what the agents and the PV do moves the level of the loading, which the kernel follows
        // through the voltage drop it sees, much more than its distribution over the feeder)
        const int nb = net->n_bus, per_day = 24 * (cfg->per_hour > 0 ? cfg->per_hour : 4);
        std::vector<double> row(series->cols), p(nb, 0.0), q(nb, 0.0);
        for (int k = 0; k < 8 && err == hipSuccess; ++k) {
            int64_t r = series->rows / 2 + (int64_t)k * per_day / 8;
            if (r >= series->rows) r = series->rows - 1;
            err = hipMemcpy(row.data(), series->table + r * series->cols, series->cols * sizeof(double), hipMemcpyDeviceToHost);
            for (int b = 0; b < nb; ++b) { p[b] += row[b] / 8.0; q[b] += row[nb + b] / 8.0; }
        }
        if (err == hipSuccess) calibrate_sweep_accel(&e->hnet, p.data(), q.data());
    }
    if (err == hipSuccess) err = hipMalloc(&e->net, sizeof(DevNet));
    if (err == hipSuccess) err = hipMemcpy(e->net, &e->hnet, sizeof(DevNet), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMalloc(&e->st.vm, sz_vm);
    if (err == hipSuccess) err = hipMalloc(&e->st.vw, sz_v);
    if (err == hipSuccess) err = hipMalloc(&e->st.agent, sz_ag);
    if (err == hipSuccess) err = hipMalloc(&e->st.cumrew, sz_cr);
    if (err == hipSuccess) err = hipMalloc(&e->st.ienv, sz_ie);
    if (err == hipSuccess) err = hipMalloc(&e->st.ring, sz_ring);
    if (err == hipSuccess) err = hipMemset(e->st.vm, 0, sz_vm);
    if (err == hipSuccess) err = hipMemset(e->st.vw, 0, sz_v);
    if (err == hipSuccess) err = hipMemset(e->st.agent, 0, sz_ag);
    if (err == hipSuccess) err = hipMemset(e->st.cumrew, 0, sz_cr);
    if (err == hipSuccess) err = hipMemset(e->st.ienv, 0, sz_ie);
    if (err == hipSuccess) err = hipMemset(e->st.ring, 0, sz_ring);
    if (err != hipSuccess) {
        fprintf(stderr, "[flexenv] flexenv_create: %s\n", hipGetErrorString(err));
        flexenv_destroy(e);            // frees whatever was allocated (null pointers are skipped)
        return err == hipErrorOutOfMemory ? FLEX_ENOMEM : FLEX_EHIP;
    }
    *out = e;
    return FLEX_OK;
}

void flexenv_destroy(FlexEnv* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    void* ptrs[] = {e->net, e->st.vm, e->st.vw, e->st.agent, e->st.cumrew, e->st.ienv, e->st.ring};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete e;
}

int32_t flexenv_num_envs(const FlexEnv* e) { return e ? e->n_envs : 0; }
int flexenv_set_step_counter(FlexEnv* e, int64_t* counter, int64_t modulo) {
    if (!e || modulo < 0) return FLEX_EINVAL;
    e->step_counter = counter;
    e->step_modulo = modulo;
    return FLEX_OK;
}
int flexenv_set_replay_sink(FlexEnv* e, const FlexReplaySink* sink) {
    if (!e) return FLEX_EINVAL;
    if (!sink) { e->has_sink = 0; return FLEX_OK; }
    if (!sink->policy_action || !sink->hid_new || !sink->small_ring || !sink->hid_ring || !sink->acc || sink->act_w < 1 ||
        sink->act_w > FLEX_WAVE / 2 || sink->hid_w < 4 || (sink->hid_w & 3) || sink->hid_w > 3 * 4 * (FLEX_WAVE / 2) ||
        sink->small_w < sink->act_w + e->cfg.n_agents + 2 ||
        ((reinterpret_cast<uintptr_t>(sink->hid_new) | reinterpret_cast<uintptr_t>(sink->hid_ring)) & 15))
        return FLEX_EINVAL;
    e->sink = *sink;
    e->has_sink = 1;
    return FLEX_OK;
}
int flexenv_set_obs_ring(FlexEnv* e, const int64_t* cursor, int64_t slab_stride, int32_t slabs) {
    if (!e || slabs < 0 || (slabs > 0 && (!cursor || slabs < 2 || slab_stride < (int64_t)e->n_envs * e->cfg.n_agents * FLEX_ROW_W)))
        return FLEX_EINVAL;
    e->obs_cursor = slabs > 0 ? cursor : nullptr;
    e->obs_slab_stride = slab_stride;
    e->obs_slabs = slabs;
    return FLEX_OK;
}
int32_t flexenv_obs_size(const FlexEnv* e) { return e ? 6 * e->cfg.history : 0; }
int32_t flexenv_state_size(const FlexEnv* e) { return e ? 3 * e->n_bus + 2 * e->cfg.n_agents + 1 : 0; }

int flexenv_reset(FlexEnv* e, const uint8_t* mask, const ResetSpec* inj, void* obs, int32_t obs_dtype,
                  uint8_t* failed, void* stream) {
    if (!e) return FLEX_EINVAL;
    if (obs && obs_dtype != FLEX_F32 && obs_dtype != FLEX_F64) return FLEX_EINVAL;
    // a partial restart outside flexenv_step would leave the replay's row records (older counts) behind the histories
    if (mask && e->obs_slabs > 0) return FLEX_EINVAL;
    DevResetSpec d = {nullptr, nullptr, nullptr, nullptr, nullptr};
    if (inj) { d.day = inj->day; d.hour = inj->hour; d.interval = inj->interval; d.e0 = inj->e0; d.a0 = inj->a0; }
    KArgs k = make_args(e);
    hipStream_t s = (hipStream_t)stream;
    const int epw = e->hnet.epw;
    const dim3 grid = env_grid(e->n_envs, epw);
    const bool f64 = obs && obs_dtype == FLEX_F64;
    const int want = obs ? 1 : 0;
    if (epw == 2) {
        if (f64) hipLaunchKernelGGL((flex_reset_kernel<2, double>), grid, env_block(), 0, s, k, mask, d, (double*)obs, want, failed);
        else hipLaunchKernelGGL((flex_reset_kernel<2, float>), grid, env_block(), 0, s, k, mask, d, (float*)obs, want, failed);
    } else {
        if (f64) hipLaunchKernelGGL((flex_reset_kernel<1, double>), grid, env_block(), 0, s, k, mask, d, (double*)obs, want, failed);
        else hipLaunchKernelGGL((flex_reset_kernel<1, float>), grid, env_block(), 0, s, k, mask, d, (float*)obs, want, failed);
    }
    HIP_TRY(hipGetLastError());
    return FLEX_OK;
}

int flexenv_step(FlexEnv* e, const void* actions, int32_t act_dtype, double* reward, uint8_t* done, double* info,
                 uint8_t* failed, void* obs, int32_t obs_dtype, int32_t flags, void* stream) {
    const int auto_reset = (flags & FLEX_STEP_AUTORESET) ? 1 : 0;
    if (!e || !actions || !reward || !done) return FLEX_EINVAL;
    if (act_dtype != FLEX_F32 && act_dtype != FLEX_F64) return FLEX_EINVAL;
    if (obs && obs_dtype != FLEX_F32 && obs_dtype != FLEX_F64) return FLEX_EINVAL;
    if (flags & ~(FLEX_STEP_AUTORESET | FLEX_STEP_OBS_RING | FLEX_STEP_REPLAY_SINK | FLEX_STEP_OBS_ROWS)) return FLEX_EINVAL;   // incl. ABI 1's ring flag (2)
    KArgs k = make_args(e);
    k.step_counter = e->step_counter; k.step_modulo = e->step_modulo;
    const bool rows = (flags & (FLEX_STEP_OBS_RING | FLEX_STEP_OBS_ROWS)) != 0;
    if (flags & FLEX_STEP_OBS_RING) {
        if (!obs || e->obs_slabs < 2 || obs_dtype != FLEX_F32) return FLEX_EINVAL;      // row ring registered (fp32 records)
        k.obs_cursor = e->obs_cursor; k.obs_slab_stride = e->obs_slab_stride; k.obs_slabs = e->obs_slabs;
    } else if (rows) {
        obs = nullptr;                               // FLEX_STEP_OBS_ROWS alone: the environment's own ring is the observation
    }
    const bool sink = (flags & FLEX_STEP_REPLAY_SINK) != 0;
    if (sink) {
        // the sink instantiation exists for the vectorised fp32 path: two environments per wavefront, fp32 actions
        if (!(flags & FLEX_STEP_OBS_RING) || !e->has_sink || e->hnet.epw != 2 || act_dtype != FLEX_F32) return FLEX_EINVAL;
        k.sink = e->sink;
    }
    hipStream_t s = (hipStream_t)stream;
    const int epw = e->hnet.epw;
    const dim3 grid = env_grid(e->n_envs, epw);
    if (rows) {
        // get_obs() as a row push (ROWS instantiations: no ObsHist, the observation dtype plays no part)
#define FLEX_LAUNCH_ROWS(EPW_, ACT_, SINK_) hipLaunchKernelGGL((flex_step_kernel<EPW_, float, ACT_, FLEX_OBS_AGENTS_SMALL, SINK_, true>), \
            grid, env_block(), 0, s, k, (const ACT_*)actions, reward, done, info, failed, (float*)obs, 1, auto_reset)
        if (sink) FLEX_LAUNCH_ROWS(2, float, true);
        else if (epw == 2 && act_dtype == FLEX_F32) FLEX_LAUNCH_ROWS(2, float, false);
        else if (epw == 2) FLEX_LAUNCH_ROWS(2, double, false);
        else if (act_dtype == FLEX_F32) FLEX_LAUNCH_ROWS(1, float, false);
        else FLEX_LAUNCH_ROWS(1, double, false);
#undef FLEX_LAUNCH_ROWS
        HIP_TRY(hipGetLastError());
        return FLEX_OK;
    }
    const bool f64 = obs && obs_dtype == FLEX_F64;
    const int want = obs ? 1 : 0;
#define FLEX_LAUNCH_STEP(EPW_, OBS_, ACT_) do { \
        if (small_obs) hipLaunchKernelGGL((flex_step_kernel<EPW_, OBS_, ACT_, FLEX_OBS_AGENTS_SMALL>), grid, env_block(), 0, s, k, \
            (const ACT_*)actions, reward, done, info, failed, (OBS_*)obs, want, auto_reset); \
        else hipLaunchKernelGGL((flex_step_kernel<EPW_, OBS_, ACT_, FLEX_OBS_AGENTS_LARGE>), grid, env_block(), 0, s, k, \
            (const ACT_*)actions, reward, done, info, failed, (OBS_*)obs, want, auto_reset); } while (0)
    const bool small_obs = e->cfg.n_agents == FLEX_OBS_AGENTS_SMALL && 3 * e->cfg.history <= FLEX_OBS_CLASSES(epw) * (FLEX_WAVE / epw);
    const int variant = (epw == 2 ? 4 : 0) + (f64 ? 2 : 0) + (act_dtype == FLEX_F64 ? 1 : 0);
    switch (variant) {
        case 0: FLEX_LAUNCH_STEP(1, float, float); break;
        case 1: FLEX_LAUNCH_STEP(1, float, double); break;
        case 2: FLEX_LAUNCH_STEP(1, double, float); break;
        case 3: FLEX_LAUNCH_STEP(1, double, double); break;
        case 4: FLEX_LAUNCH_STEP(2, float, float); break;
        case 5: FLEX_LAUNCH_STEP(2, float, double); break;
        case 6: FLEX_LAUNCH_STEP(2, double, float); break;
        default: FLEX_LAUNCH_STEP(2, double, double); break;
    }
#undef FLEX_LAUNCH_STEP
    HIP_TRY(hipGetLastError());
    return FLEX_OK;
}

int flexenv_step_many(FlexEnv* e, const void* actions, int32_t act_dtype, int32_t act_period, int32_t steps,
                      double* reward, uint8_t* done, double* info, uint8_t* failed, int32_t flags, void* stream) {
    if (!e || !actions || !reward || !done || steps < 1 || act_period < 1) return FLEX_EINVAL;
    if (act_dtype != FLEX_F32 && act_dtype != FLEX_F64) return FLEX_EINVAL;
    // get_obs() is the row push (the environment's own history); a registered row ring / replay sink belongs to the
    // closed-loop forms (flexenv_step, flexenv_rollout_burst), whose cursor cells a launch of many steps does not walk
    if (flags & ~(FLEX_STEP_AUTORESET | FLEX_STEP_OBS_ROWS | FLEX_STEP_MANY_NO_CARRY)) return FLEX_EINVAL;
    if (!(flags & FLEX_STEP_OBS_ROWS) || e->obs_slabs > 0) return FLEX_EINVAL;
    ManyArgs m;
    m.k = make_args(e);
    m.k.step_counter = e->step_counter; m.k.step_modulo = e->step_modulo;
    m.actions = actions; m.reward = reward; m.done = done; m.info = info; m.failed = failed;
    m.n_steps = steps; m.act_period = act_period; m.auto_reset = (flags & FLEX_STEP_AUTORESET) ? 1 : 0;
    m.carry = (flags & FLEX_STEP_MANY_NO_CARRY) ? 0 : 1;
    hipStream_t s = (hipStream_t)stream;
    const int epw = e->hnet.epw;
    const dim3 grid = env_grid(e->n_envs, epw);
    if (epw == 2 && act_dtype == FLEX_F32) hipLaunchKernelGGL((flex_step_many_kernel<2, float>), grid, env_block(), 0, s, m);
    else if (epw == 2) hipLaunchKernelGGL((flex_step_many_kernel<2, double>), grid, env_block(), 0, s, m);
    else if (act_dtype == FLEX_F32) hipLaunchKernelGGL((flex_step_many_kernel<1, float>), grid, env_block(), 0, s, m);
    else hipLaunchKernelGGL((flex_step_many_kernel<1, double>), grid, env_block(), 0, s, m);
    HIP_TRY(hipGetLastError());
    return FLEX_OK;
}

int flexenv_rollout_burst(FlexEnv* e, const FlexActorArgs* actor, double* reward, uint8_t* done, double* info,
                          uint8_t* failed, float* obs_ring, int32_t steps, const FlexBurstSafety* safety, void* stream) {
    if (!e || !actor || !reward || !done || !obs_ring || steps < 1) return FLEX_EINVAL;
    const FlexActorArgs& p = *actor;
    const int na = e->cfg.n_agents;
    // the configuration of the two-launch sink step (flexenv_step with FLEX_STEP_OBS_RING | FLEX_STEP_REPLAY_SINK behind
    // flexnet_actor_forward in ring mode), and the policy's tiles of a block = the block's sixteen environments
    if (!e->has_sink || e->obs_slabs < 2 || e->hnet.epw != 2 || na > 5) return FLEX_EINVAL;
    if (p.rows != (int64_t)e->n_envs * na || p.n_agents != na || p.obs_dim < 1 || p.obs_dim > FLEXNET_MAX_OBS || (p.obs_dim & 3) ||
        p.act_dim < 1 || p.act_dim > FLEXNET_MAX_ACT) return FLEX_EINVAL;
    if (!p.hidden_in || !p.hidden_out || !p.means || !p.action || !p.env_action || !p.cursor || !p.cursor_out ||
        !p.fc1_w || !p.fc1_b || !p.w_ih || !p.w_hh || !p.b_ih || !p.b_hh || !p.fc2_w || !p.fc2_b ||
        (p.layernorm && (!p.ln_w || !p.ln_b)) || (!p.noise && !p.rng_state)) return FLEX_EINVAL;
    if (p.ring_slabs != e->obs_slabs || steps >= e->obs_slabs || p.obs_dim != 6 * e->cfg.history) return FLEX_EINVAL;
    if (p.cursor_out != e->obs_cursor || p.cursor != e->sink.cursor_out || p.hidden_out != e->sink.hid_new ||
        p.action != e->sink.policy_action || p.noise) return FLEX_EINVAL;
    // the burst draws the noise of steps rng_state[1] .. rng_state[1] + steps - 1 in the kernel and the finish launch advances the
    // step counter through the sink's aux_counter: the two must be the same cell, or the next burst would repeat the draws
    // (a NULL aux_counter used to pass)
    if (!p.rng_state || !e->sink.aux_counter || (const void*)e->sink.aux_counter != (const void*)(p.rng_state + 1)) return FLEX_EINVAL;
    if (p.save_z1) return FLEX_EINVAL;
    KArgs k = make_args(e);
    k.obs_cursor = e->obs_cursor; k.obs_slab_stride = e->obs_slab_stride; k.obs_slabs = e->obs_slabs;
    k.sink = e->sink;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((e->n_envs + 15) / 16), block(64 * R16_W);
    const bool small_obs = na == FLEX_OBS_AGENTS_SMALL && 3 * e->cfg.history <= FLEX_OBS_CLASSES(2) * (FLEX_WAVE / 2);
    BurstArgs b;
    b.k = k; b.act = p;
    // the policy reads the stacked observation IN PLACE from this environment's history (what flexenv_obs_source hands out)
    b.act.obs = e->st.ring; b.act.obs_pushed = e->st.ienv + IF_OBSCNT;
    b.act.obs_row_stride = e->cfg.history * 12; b.act.obs_pushed_stride = IF_COUNT; b.act.obs_slots = e->cfg.history; b.act.obs_slot_w = 6;
    b.reward = reward; b.done = done; b.info = info; b.failed = failed; b.obs_ring = obs_ring; b.n_steps = (int)steps;
    b.safety = 0; b.s_p = b.s_q = b.beta = nullptr; b.v_min = b.v_max = b.rho = 0.0; b.adjusted = nullptr; b.safe_env_action = nullptr;
    b.act_low = b.act_span = 0.0f;
    if (safety) {
        if (!safety->s_p || !safety->s_q || !safety->beta || !safety->adjusted || !safety->env_action ||
            !(safety->act_high >= safety->act_low) || p.act_dim != 4) return FLEX_EINVAL;      // (four controls per building)
        b.safety = 1; b.s_p = safety->s_p; b.s_q = safety->s_q; b.beta = safety->beta;
        b.v_min = safety->v_min; b.v_max = safety->v_max; b.rho = safety->penalty;
        b.adjusted = safety->adjusted; b.safe_env_action = safety->env_action;
        b.act_low = safety->act_low; b.act_span = safety->act_high - safety->act_low;
    }
    if (b.safety) {
        if (small_obs) hipLaunchKernelGGL((flex_rollout_burst_kernel<FLEX_OBS_AGENTS_SMALL, true>), grid, block, 0, s, b);
        else hipLaunchKernelGGL((flex_rollout_burst_kernel<FLEX_OBS_AGENTS_LARGE, true>), grid, block, 0, s, b);
    } else {
        if (small_obs) hipLaunchKernelGGL((flex_rollout_burst_kernel<FLEX_OBS_AGENTS_SMALL, false>), grid, block, 0, s, b);
        else hipLaunchKernelGGL((flex_rollout_burst_kernel<FLEX_OBS_AGENTS_LARGE, false>), grid, block, 0, s, b);
    }
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(flex_burst_finish_kernel, dim3(1), dim3(64), 0, s, const_cast<int64_t*>(p.cursor), p.cursor_out,
                       e->sink.aux_counter ? const_cast<uint64_t*>(p.rng_state) : nullptr, (int)steps, (int)e->obs_slabs);
    HIP_TRY(hipGetLastError());
    return FLEX_OK;
}

int flexenv_obs(FlexEnv* e, void* obs, int32_t obs_dtype, void* stream) {
    if (!e || !obs || (obs_dtype != FLEX_F32 && obs_dtype != FLEX_F64)) return FLEX_EINVAL;
    if (e->obs_slabs > 0) return FLEX_EINVAL;        // a push the row ring would not see (include/flexenv.h)
    KArgs k = make_args(e);
    hipStream_t s = (hipStream_t)stream;
    const int epw = e->hnet.epw;
    const dim3 grid = env_grid(e->n_envs, epw);
    if (epw == 2) {
        if (obs_dtype == FLEX_F64) hipLaunchKernelGGL((flex_obs_kernel<2, double>), grid, env_block(), 0, s, k, (double*)obs);
        else hipLaunchKernelGGL((flex_obs_kernel<2, float>), grid, env_block(), 0, s, k, (float*)obs);
    } else {
        if (obs_dtype == FLEX_F64) hipLaunchKernelGGL((flex_obs_kernel<1, double>), grid, env_block(), 0, s, k, (double*)obs);
        else hipLaunchKernelGGL((flex_obs_kernel<1, float>), grid, env_block(), 0, s, k, (float*)obs);
    }
    HIP_TRY(hipGetLastError());
    return FLEX_OK;
}

int flexenv_obs_view(FlexEnv* e, void* obs, int32_t obs_dtype, void* stream) {
    if (!e || !obs || (obs_dtype != FLEX_F32 && obs_dtype != FLEX_F64)) return FLEX_EINVAL;
    KArgs k = make_args(e);
    const int64_t tot = (int64_t)e->n_envs * e->cfg.n_agents * e->cfg.history * 6;
    const dim3 grid((unsigned)((tot + 255) / 256));
    if (obs_dtype == FLEX_F64) hipLaunchKernelGGL(flex_obs_view_kernel<double>, grid, dim3(256), 0, (hipStream_t)stream, k, (double*)obs);
    else hipLaunchKernelGGL(flex_obs_view_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, k, (float*)obs);
    HIP_TRY(hipGetLastError());
    return FLEX_OK;
}

int flexenv_obs_source(const FlexEnv* e, FlexObsSource* out) {
    if (!e || !out) return FLEX_EINVAL;
    out->ring = e->st.ring;
    out->pushed = e->st.ienv + IF_OBSCNT;
    out->row_stride = e->cfg.history * 12;
    out->pushed_stride = IF_COUNT;
    out->slots = e->cfg.history;
    out->slot_w = 6;
    return FLEX_OK;
}

int flexenv_state(FlexEnv* e, double* state, void* stream) {
    if (!e || !state) return FLEX_EINVAL;
    KArgs k = make_args(e);
    hipLaunchKernelGGL(flex_state_kernel, env_grid(e->n_envs), env_block(), 0, (hipStream_t)stream, k, state);
    HIP_TRY(hipGetLastError());
    return FLEX_OK;
}

// strided field <-> dense [N, width] copies for peek/poke
__global__ void flex_gather_f64(const double* __restrict__ src, int64_t stride, int width, int n, double* __restrict__ dst) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n * width) return;
    dst[t] = src[(t / width) * stride + (t % width)];
}
__global__ void flex_scatter_f64(double* __restrict__ dst, int64_t stride, int width, int n, const double* __restrict__ src) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n * width) return;
    dst[(t / width) * stride + (t % width)] = src[t];
}
// initial_ess_energy (env:100,354): the reset's pre-solve draw until the first step has run (steps == 1), the
// current energy afterwards — the step kernel stores E alone (AF_ enum)
__global__ void flex_gather_einit(const double* __restrict__ agent, const int32_t* __restrict__ ienv, int rec, int na, int n,
                                  double* __restrict__ dst) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n * na) return;
    const int64_t env = t / na;
    const int ag = (int)(t - env * na);
    const bool fresh = ienv[env * IF_COUNT + IF_STEPS] == 1;
    dst[t] = agent[env * rec + (fresh ? AF_EINIT : AF_E) * na + ag];
}
__global__ void flex_gather_i32(const int32_t* __restrict__ src, int64_t stride, int n, int32_t* __restrict__ dst) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    dst[t] = src[t * stride];
}

static int field_f64(FlexEnv* e, int field, double** base, int64_t* stride, int* width) {
    const int na = e->cfg.n_agents;
    switch (field) {
        case FLEX_PEEK_V: *base = e->st.vm; *stride = e->n_bus; *width = e->n_bus; return 1;
        case FLEX_PEEK_CUMREW: *base = e->st.cumrew; *stride = 1; *width = 1; return 1;
        case FLEX_PEEK_E: case FLEX_PEEK_E_INIT: case FLEX_PEEK_PRED: case FLEX_PEEK_CH: case FLEX_PEEK_DIS:
        case FLEX_PEEK_QPV: case FLEX_PEEK_PCT: {
            static const int map[8] = {-1, AF_E, AF_EINIT, AF_PRED, AF_CH, AF_DIS, AF_Q, AF_PCT};
            *base = e->st.agent + map[field] * na;
            *stride = agent_rec_doubles(na); *width = na; return 1;
        }
        default: return 0;
    }
}

int flexenv_peek(FlexEnv* e, int32_t field, void* dev_out, void* stream) {
    if (!e || !dev_out) return FLEX_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    double* base; int64_t stride; int width;
    if (field == FLEX_PEEK_E_INIT) {
        const int na = e->cfg.n_agents;
        const int64_t tot = (int64_t)e->n_envs * na;
        hipLaunchKernelGGL(flex_gather_einit, dim3((tot + 255) / 256), dim3(256), 0, s, e->st.agent, e->st.ienv,
                           agent_rec_doubles(na), na, e->n_envs, (double*)dev_out);
        HIP_TRY(hipGetLastError());
        return FLEX_OK;
    }
    if (field_f64(e, field, &base, &stride, &width)) {
        const int64_t tot = (int64_t)e->n_envs * width;
        hipLaunchKernelGGL(flex_gather_f64, dim3((tot + 255) / 256), dim3(256), 0, s, base, stride, width, e->n_envs, (double*)dev_out);
        HIP_TRY(hipGetLastError());
        return FLEX_OK;
    }
    int col;
    switch (field) {
        case FLEX_PEEK_STEPS: col = IF_STEPS; break;
        case FLEX_PEEK_ROW: col = IF_ROW; break;
        case FLEX_PEEK_START: col = IF_START; break;
        case FLEX_PEEK_PF_ITERS: col = IF_ITERS; break;
        case FLEX_PEEK_EPISODE: col = IF_EPISODE; break;
        case FLEX_PEEK_PF_SWEEPS: col = IF_SWEEPS; break;
        default: return FLEX_EINVAL;
    }
    hipLaunchKernelGGL(flex_gather_i32, dim3((e->n_envs + 255) / 256), dim3(256), 0, s, e->st.ienv + col, (int64_t)IF_COUNT, e->n_envs, (int32_t*)dev_out);
    HIP_TRY(hipGetLastError());
    return FLEX_OK;
}

int flexenv_poke(FlexEnv* e, int32_t field, const void* dev_in, void* stream) {
    if (!e || !dev_in) return FLEX_EINVAL;
    double* base; int64_t stride; int width;
    if (!field_f64(e, field, &base, &stride, &width)) return FLEX_EINVAL;
    const int64_t tot = (int64_t)e->n_envs * width;
    hipLaunchKernelGGL(flex_scatter_f64, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, base, stride, width, e->n_envs, (const double*)dev_in);
    HIP_TRY(hipGetLastError());
    return FLEX_OK;
}

int pf_solve_batch(const NetFix* net, int32_t n, const double* pnet, const double* qnet, double* v, double* isqr,
                   double* pl, double* ql, int32_t* iters, uint8_t* failed, double tol, int32_t max_iter,
                   int32_t solver, void* stream) {
    if (!net || n < 1 || !pnet || !qnet || !v) return FLEX_EINVAL;
    if (solver != FLEX_SOLVER_TREE && solver != FLEX_SOLVER_SWEEP && solver != FLEX_SOLVER_DENSE) return FLEX_EINVAL;
    if (solver == FLEX_SOLVER_DENSE && (net->n_bus - 1 > 32 || isqr || pl || ql)) return FLEX_EINVAL;   // 64 x 64 Jacobian, |V| only
    DevNet h;
    int32_t dummy_agent = -1;
    // agents are irrelevant for a bare solve: give build_devnet one placeholder building off the slack
    NetFix nf = *net;
    for (int b = 0; b < net->n_bus; ++b) if (b != net->slack) { dummy_agent = b; break; }
    nf.agent_bus = &dummy_agent;
    int rc = build_devnet(&nf, 1, &h);
    if (rc != FLEX_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    // the device copy of the network tables is cached per device: repeated solves on the same feeder (the normal
    // case) pay no allocation, copy or synchronisation
    static DevNet cached_host[FLEX_MAX_DEVICES];
    static DevNet* cached_dev[FLEX_MAX_DEVICES] = {nullptr};
    int dev_id = 0;
    HIP_TRY(hipGetDevice(&dev_id));
    if (dev_id < 0 || dev_id >= FLEX_MAX_DEVICES) return FLEX_EINVAL;
    if (!cached_dev[dev_id] || memcmp(&cached_host[dev_id], &h, sizeof(DevNet)) != 0) {
        if (!cached_dev[dev_id]) HIP_TRY(hipMalloc((void**)&cached_dev[dev_id], sizeof(DevNet)));
        HIP_TRY(hipStreamSynchronize(s));                 // nothing in flight may still read the old tables
        HIP_TRY(hipMemcpy(cached_dev[dev_id], &h, sizeof(DevNet), hipMemcpyHostToDevice));
        cached_host[dev_id] = h;
    }
    DevNet* d = cached_dev[dev_id];
    if (solver == FLEX_SOLVER_DENSE)
        hipLaunchKernelGGL(pf_batch_dense_kernel, dim3(n), dim3(FLEX_WAVE), 0, s, d, n, net->n_bus, pnet, qnet, v, iters,
                           failed, tol, max_iter);
    else if (h.epw == 2)
        hipLaunchKernelGGL(pf_batch_kernel<2>, env_grid(n, 2), env_block(), 0, s, d, n, net->n_bus, pnet, qnet, v, isqr, pl, ql,
                           iters, failed, tol, max_iter, solver);
    else
        hipLaunchKernelGGL(pf_batch_kernel<1>, env_grid(n, 1), env_block(), 0, s, d, n, net->n_bus, pnet, qnet, v, isqr, pl, ql,
                           iters, failed, tol, max_iter, solver);
    HIP_TRY(hipGetLastError());
    return FLEX_OK;
}

int flexenv_safety_project_env(FlexEnv* e, const void* proposed, int32_t dtype, const double* s_p, const double* s_q,
                               const double* beta, double v_min, double v_max, double penalty, double* adjusted,
                               uint8_t* intervened, float act_low, float act_high, float* env_action, void* stream) {
    if (!e || !proposed || !s_p || !s_q || !beta || !adjusted) return FLEX_EINVAL;
    if (dtype != FLEX_F32 && dtype != FLEX_F64) return FLEX_EINVAL;
    if (env_action && !(act_high >= act_low)) return FLEX_EINVAL;
    KArgs k = make_args(e);
    hipStream_t s = (hipStream_t)stream;
    if (intervened) HIP_TRY(hipMemsetAsync(intervened, 0, e->n_envs, s));
    const int tot = e->n_envs * e->cfg.n_agents;
    hipLaunchKernelGGL(flex_safety_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, k, proposed, dtype, s_p, s_q, beta,
                       v_min, v_max, penalty, adjusted, intervened, env_action, act_low, act_high - act_low);
    HIP_TRY(hipGetLastError());
    return FLEX_OK;
}

int flexenv_safety_project(FlexEnv* e, const void* proposed, int32_t dtype, const double* s_p, const double* s_q,
                           const double* beta, double v_min, double v_max, double penalty, double* adjusted,
                           uint8_t* intervened, void* stream) {
    return flexenv_safety_project_env(e, proposed, dtype, s_p, s_q, beta, v_min, v_max, penalty, adjusted, intervened, 0.0f, 0.0f,
                                      nullptr, stream);
}

}  // extern "C"
