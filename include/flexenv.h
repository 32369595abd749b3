/*
 * flexenv.h — C ABI of the MI355X-native flexibility-provision hot path.
 *
 * The reference (kosmylo/Safe-MARL) is pure Python and has no FFI layer: its
 * boundary is the duck-typed object protocol of SURVEY.md §8(b).  This header
 * is the boundary a binding would use instead; each entry point names the
 * reference interface it replaces (paths under /root/reference).  The Python
 * host side (safe-marl_amd/flex_env.py) binds it with ctypes and presents the
 * reference's FlexibilityProvisionEnv / TransReplayBuffer surface on top.
 *
 * Conventions
 *  - plain C types only; `stream` is a hipStream_t passed as void*.
 *  - every `dev` pointer is device memory owned by the caller; the library owns
 *    only the FlexEnv handle and its internal per-env state.
 *  - all calls are asynchronous on `stream`; one FlexEnv per device; a handle is
 *    not thread-safe.
 *  - return 0 on success, negative FLEX_E* for API errors (bad argument, HIP
 *    error).  Per-env numerical failure is DATA (`failed[i]`), never an error
 *    code — mirroring env:314-337, where a failed solve is caught, penalised
 *    and reported in info["solver_failed"].
 */
#ifndef FLEXENV_H
#define FLEXENV_H

#include <stdint.h>
#include "flexnet.h"           /* FlexActorArgs: flexenv_rollout_burst runs the policy inside the environment's launch */

#ifdef __cplusplus
extern "C" {
#endif

#define FLEX_MAX_BUS 64      /* one lane per bus in a 64-wide wavefront */
#define FLEX_MAX_AGENTS 8
#define FLEX_MAX_CHILDREN 8
#define FLEX_INFO_W 7        /* reward, revenue, der_cost, ess_cost, discomfort, voltage_penalty, cumulative_reward (env:696-704) */

/* Bumped whenever the meaning of an argument, flag or registered buffer changes (flexenv_abi_version() returns the
 * value the library was built with).  2 = round 4's row-ring contract: FLEX_STEP_OBS_RING slabs hold FLEX_ROW_FLOATS
 * records, not stacked observations (ABI 1), and the flag moved from value 2 to 16 so that a caller built against ABI 1
 * gets FLEX_EINVAL from flexenv_step instead of row records in a ring it reads as stacked observations. */
#define FLEX_ABI_VERSION 2

/* pf.py:41-45 declares Vsqr, Isqr and E_next NonNegativeReals; E_next is pinned by the equality pf.py:96-98, so a step
 * or reset whose E_next falls below -FLEX_DOMAIN_EPS is an infeasible NLP in the reference (IPOPT status != ok,
 * pf.py:104-105) and takes the solver_failed path here.  The margin is IPOPT's default bound relaxation
 * (bound_relax_factor 1e-8; the reference sets no solver options, pf.py:101). */
#define FLEX_DOMAIN_EPS 1e-8

#define FLEX_OK 0
#define FLEX_EINVAL (-22)
#define FLEX_ENOMEM (-12)
#define FLEX_EHIP (-5)

/* dtype tags for action input / observation output */
#define FLEX_F64 0
#define FLEX_F32 1

/* linear solver used for the Newton step */
#define FLEX_SOLVER_TREE 0   /* leaf->root 2x2 block elimination on the radial Jacobian (no fill-in) */
#define FLEX_SOLVER_DENSE 1  /* dense LU of the 2(n-1) x 2(n-1) Jacobian, kept as the reference variant */
#define FLEX_SOLVER_SWEEP 2  /* backward/forward sweeps (Z-bus Gauss via wavefront scans), then the TREE Newton
                                solver verifies the Ybus mismatch and finishes/falls back if needed */

/* Scalars of madrl/args/env_args/flex_provision.yaml:3-33 plus solver controls. */
typedef struct FlexCfg {
    int32_t n_agents;          /* len(buildings), env:66 */
    int32_t history;           /* env:65 */
    int32_t episode_limit;     /* env:61 */
    int32_t per_hour;          /* 60 // time_delta, env:416,477 */
    int32_t n_start_days;      /* pv_days - episode_days, env:421-424 */
    int32_t raw_actions;       /* 1 = the safemaddpg branch env:268-274 (no scaling) */
    int32_t pf_max_iter;       /* Newton cap; beyond it the env reports solver_failed */
    int32_t solver;            /* FLEX_SOLVER_* */
    int32_t warm_start;        /* 1 = start Newton from the previous solution instead of 1∠0 */
    int32_t no_sweep_accel;    /* 1 = plain sweeps (FLEX_SOLVER_SWEEP without the two-sweep extrapolation calibrated at create) */
    double v_min, v_max;       /* env:685 */
    double e_min, e_max;       /* env:637,647 */
    double p_ch_max, p_dis_max;/* env:630-631 */
    double eta_ch, eta_dis;    /* env:634; pf.py:37-38 */
    double tan_phi;            /* tan(acos(cos_phi_max)), env:623 */
    double max_power_reduction;/* env:278,677 */
    double pv_cost, ess_cost, discomfort_coeff, voltage_coeff; /* env:682-685 */
    double dt;                 /* 24 / episode_limit, pf.py:23-24 */
    double fail_penalty;       /* 200, env:336 */
    double pf_tol;             /* inf-norm power mismatch, pu */
    double action_low, action_high; /* env:716-719 */
    uint64_t seed;             /* key of the Philox reset stream */
} FlexCfg;

/* Rooted radial feeder, HOST arrays in bus-index order (utils/create_net.py:8-39 flattened). */
typedef struct NetFix {
    int32_t n_bus;
    int32_t slack;               /* bus index with bus_types == 1 (pf.py:51-53) */
    int32_t n_levels;            /* 1 + max distance from the slack */
    int32_t max_children;
    const int32_t* parent;       /* [n_bus], -1 at the slack */
    const int32_t* level;        /* [n_bus] */
    const int32_t* child;        /* [n_bus * max_children], -1 padded */
    const double* r;             /* [n_bus] pu resistance of the line to parent (0 at slack) */
    const double* x;             /* [n_bus] */
    const int32_t* agent_bus;    /* [n_agents] bus index of each building */
} NetFix;

/* Device-resident series table: rows x cols fp64 row-major,
 * cols = [Pd(n_bus) | Qd(n_bus) | Ppv(n_agents) | price]  (env:473-547 slices of it). */
typedef struct SeriesTab {
    const double* table;         /* dev */
    int64_t rows;
    int32_t cols;
} SeriesTab;

/* Injected episode draws (parity mode).  All device arrays; any may be NULL to fall back to the
 * Philox stream for that item. */
typedef struct ResetSpec {
    const int32_t* day;          /* dev [N]  env:86  */
    const int32_t* hour;         /* dev [N]  env:85  */
    const int32_t* interval;     /* dev [N]  env:87  */
    const double* e0;            /* dev [N, n_agents]    env:100 */
    const double* a0;            /* dev [N, 4*n_agents]  env:103 */
} ResetSpec;

typedef struct FlexEnv FlexEnv;

/* fields of flexenv_peek — the reference's _get_* accessors (env:740-778) and attributes */
enum FlexField {
    FLEX_PEEK_V = 0,            /* f64 [N, n_bus]   current_voltage           env:740 */
    FLEX_PEEK_E = 1,            /* f64 [N, n_agents] current_ess_energy       env:760 */
    FLEX_PEEK_E_INIT = 2,       /* f64 [N, n_agents] initial_ess_energy       env:100,354 — differs from E only between a
                                   reset and the first step; a poke of it is honoured in that interval only */
    FLEX_PEEK_PRED = 3,         /* f64 [N, n_agents] power_reduction          env:764 */
    FLEX_PEEK_CH = 4,           /* f64 [N, n_agents] ess_charging             env:768 */
    FLEX_PEEK_DIS = 5,          /* f64 [N, n_agents] ess_discharging          env:772 */
    FLEX_PEEK_QPV = 6,          /* f64 [N, n_agents] q_pv                     env:756 */
    FLEX_PEEK_PCT = 7,          /* f64 [N, n_agents] percentage_reduction     env:284 */
    FLEX_PEEK_CUMREW = 8,       /* f64 [N]          cumulative_reward         env:343 */
    FLEX_PEEK_STEPS = 9,        /* i32 [N]          steps                     env:342 */
    FLEX_PEEK_ROW = 10,         /* i32 [N]          absolute series row of the current data, env:609-619 */
    FLEX_PEEK_START = 11,       /* i32 [N]          episode start row         env:477 */
    FLEX_PEEK_PF_ITERS = 12,    /* i32 [N]          Newton iterations of the last solve */
    FLEX_PEEK_EPISODE = 13,     /* i32 [N]          reset-attempt counter of the Philox stream */
    FLEX_PEEK_PF_SWEEPS = 14    /* i32 [N]          sweeps the environment's wavefront executed in the last solve
                                 *                   (FLEX_SOLVER_SWEEP; with two environments per wavefront both run until the slower
                                 *                   one's test has passed) */
};

/* Replaces FlexibilityProvisionEnv.__init__ (env:34-72) minus its reset; `series.table` must stay
 * valid for the life of the handle. */
int flexenv_create(const FlexCfg* cfg, const NetFix* net, const SeriesTab* series,
                   int32_t n_envs, int32_t device, FlexEnv** out);
void flexenv_destroy(FlexEnv* env);

/* Replaces reset()/manual_reset() (env:74-155, 157-239) for every env whose mask byte is non-zero
 * (mask == NULL: all).  Draws come from `inj` where given, else from the Philox reset stream; a
 * failed initial solve re-draws (env:83,150-153) up to 8 times, then sets failed[i].  If `obs` is
 * non-NULL the first observation of each reset env is pushed and written (rows of other envs are
 * left untouched) — the get_obs() of env:155. */
int flexenv_reset(FlexEnv* env, const uint8_t* mask /*dev [N] or NULL*/, const ResetSpec* inj /*or NULL*/,
                  void* obs /*dev [N, n_agents, 6*history] or NULL*/, int32_t obs_dtype,
                  uint8_t* failed /*dev [N] or NULL*/, void* stream);

/* Replaces step() (env:241-356).  `actions` is [N, n_agents, 4] in the dtype given (the reference
 * hands float32 values that NumPy 1.26 promotes to float64 before any arithmetic, util.py:184 +
 * env:278).  If `obs` is non-NULL the get_obs() that model.py:223 issues right after step() is fused
 * into the same launch. */
#define FLEX_STEP_AUTORESET 1   /* an env that terminates in this step restarts (Philox stream) inside the same
                                  launch; its `obs` row then holds the FIRST observation of the new episode */
#define FLEX_STEP_OBS_RING 16   /* (value 2 until ABI 1, now rejected) implies FLEX_STEP_OBS_ROWS; `obs` is the BASE of the ROW ring registered with flexenv_set_obs_ring:
                                  this launch also writes one record per (env, agent) into slab (cursor[0] + 1) mod slabs,
                                  read on the device (replayable HIP graph):
                                      [Pd, Qd, Ppv, V, price, E, older, 0]   (FLEX_ROW_FLOATS fp32)
                                  the step's feature row (env:377-382) and `older` = min(rows pushed before it in its
                                  episode, history - 1): the stacked observation of env:387-401 at that slab is the rows
                                  of the `older` slabs before it, this one, and zeros in front */
#define FLEX_STEP_REPLAY_SINK 4 /* with FLEX_STEP_OBS_RING: the launch also files this step's transition in the consumer's slab ring
                                  (flexenv_set_replay_sink) — no bookkeeping launch follows the step */
#define FLEX_STEP_OBS_ROWS 8    /* get_obs() as a ROW PUSH: the step appends its feature row (120 B per env) to the env's
                                  observation history and writes no stacked copy; `obs` is ignored.  Consumers read the
                                  stacked observation in place (flexenv_obs_source: the policy kernels of include/flexnet.h
                                  do) or materialise it with flexenv_obs_view.  Same history, same values as `obs` non-NULL. */
#define FLEX_ROW_FLOATS 8
int flexenv_step(FlexEnv* env, const void* actions, int32_t act_dtype,
                 double* reward /*dev [N]*/, uint8_t* done /*dev [N]*/,
                 double* info /*dev [N, FLEX_INFO_W] or NULL*/, uint8_t* failed /*dev [N] or NULL*/,
                 void* obs /*dev [N, n_agents, 6*history] or NULL*/, int32_t obs_dtype, int32_t flags, void* stream);

/* `steps` consecutive step()s on a GIVEN action sequence in ONE launch: the vectorised form of the reference's open-loop
 * episode runner (run_env.py:78-92 — sampled actions, step(), per-step records of reward and info).  Step k reads action slab
 * k mod act_period of `actions` ([act_period][N, n_agents, 4], dtype as in flexenv_step) and writes row k of `reward`, `done`,
 * `info`, `failed` ([steps][...], per-row layout as in flexenv_step); state, history and every output are bit-identical to
 * `steps` calls of flexenv_step with the same flags.  get_obs() is the row push: FLEX_STEP_OBS_ROWS is required
 * (FLEX_STEP_AUTORESET optional), FLEX_EINVAL otherwise and while a row ring is registered (flexenv_set_obs_ring: the
 * closed-loop forms own its cursor).  Environments are independent, so a wavefront walks its own environments through the
 * whole sequence without a launch boundary per step.  FLEX_STEP_MANY_NO_CARRY (diagnostic): every step re-loads its integer
 * record from memory instead of carrying it in registers. */
#define FLEX_STEP_MANY_NO_CARRY 32
int flexenv_step_many(FlexEnv* env, const void* actions, int32_t act_dtype, int32_t act_period, int32_t steps,
                      double* reward /*dev [steps, N]*/, uint8_t* done /*dev [steps, N]*/,
                      double* info /*dev [steps, N, FLEX_INFO_W] or NULL*/, uint8_t* failed /*dev [steps, N] or NULL*/,
                      int32_t flags, void* stream);

/* Replaces get_obs() (env:370-403): stateful, appends to the history on every call.  FLEX_EINVAL while a row ring is
 * registered (flexenv_set_obs_ring): a push outside flexenv_step would leave the replay's row records behind the history.
 * The same holds for a MASKED flexenv_reset; a full reset (mask NULL) restarts every history and is allowed. */
int flexenv_obs(FlexEnv* env, void* obs /*dev [N, n_agents, 6*history]*/, int32_t obs_dtype, void* stream);

/* The stacked observation as the last push left it (env:387-401), WITHOUT appending to the history: flexenv_step(...,
 * FLEX_STEP_OBS_ROWS) followed by flexenv_obs_view writes what flexenv_step(..., obs) writes. */
int flexenv_obs_view(FlexEnv* env, void* obs /*dev [N, n_agents, 6*history]*/, int32_t obs_dtype, void* stream);
/* Where the observation history lives, for consumers that read it in place.  Row r = env * n_agents + agent owns
 * `row_stride` floats at ring + r * row_stride: 2 * slots slots of slot_w floats, a MIRROR ring — the k-th row pushed in an
 * episode sits in slots k mod slots and slots + k mod slots, slots 1 .. slots-1 are zeroed at every episode start — so
 * with c = pushed[env * pushed_stride] (rows pushed so far, >= 1) the stacked observation is the slots * slot_w
 * CONTIGUOUS floats starting at float ((c - 1) mod slots + 1) * slot_w of the row.  Device pointers owned by the handle. */
typedef struct {
    const float* ring;
    const int32_t* pushed;
    int32_t row_stride, pushed_stride, slots, slot_w;
} FlexObsSource;
int flexenv_obs_source(const FlexEnv* env, FlexObsSource* out);

/* Replaces get_state() (env:358-368): [Pd | Qd | Ppv | V | price | E]. */
int flexenv_state(FlexEnv* env, double* state /*dev [N, 3*n_bus + 2*n_agents + 1]*/, void* stream);

/* The reference's attribute reads and _get_* accessors (env:740-778). */
int flexenv_peek(FlexEnv* env, int32_t field, void* dev_out, void* stream);

/* Overwrite a peekable f64 field (tests and checkpoint restore): same shapes as flexenv_peek. */
int flexenv_poke(FlexEnv* env, int32_t field, const void* dev_in, void* stream);

int32_t flexenv_num_envs(const FlexEnv* env);
/* Optional launch counter: every later flexenv_step adds 1 to *counter (device int64, caller-owned, NULL switches it
 * off) from one lane of the launch, wrapping to 0 at `modulo` (0 = never).  Lets a consumer that indexes by vector step
 * — the replay ring cursor of flexnet_rollout_pack (include/flexnet.h) — follow the steps of a replayed HIP graph
 * without a launch of its own. */
int flexenv_set_step_counter(FlexEnv* env, int64_t* counter, int64_t modulo);
/* Row ring for FLEX_STEP_OBS_RING: `cursor` (device int64, caller-owned, never written by this library) holds the
 * slab the consumer is reading; a step writes slab (cursor[0] + 1) mod slabs of a ring [slabs][N][n_agents][FLEX_ROW_FLOATS]
 * whose slabs are `slab_stride` floats apart (>= N * n_agents * FLEX_ROW_FLOATS) — every feature row is stored once, where
 * the replay keeps it; a stacked observation is never copied.  slabs = 0 switches it off. */
int flexenv_set_obs_ring(FlexEnv* env, const int64_t* cursor, int64_t slab_stride, int32_t slabs);
/* Replay sink for FLEX_STEP_REPLAY_SINK (madrl/models/model.py:230-262 + utils/replay_buffer.py:23-27 inside the step
 * launch): with p = the obs-ring cursor and p' = (p + 1) mod slabs, every environment e of the launch writes
 *     small_ring[p][e]  = [policy_action[e] (act_w floats) | reward x n_agents | done | last_step = done]
 *     hid_ring[p'][e]   = hid_new[e] * (1 - done)          (hid_w floats; the recurrent state the next step starts from)
 *     acc[e][0..8]     += info[0..6], reward, solver_failed  (per-environment running sums, fp64: episode statistics)
 * and one lane of the launch sets *cursor_out = p' and adds 1 to *aux_counter (either may be NULL).  The launch never
 * reads cursor_out / aux_counter and never writes the obs-ring cursor, so it replays from a HIP graph next to a policy
 * kernel that reads cursor_out and writes the obs-ring cursor.  All pointers are device memory owned by the caller. */
typedef struct {
    const float* policy_action; /* [N, act_w] */
    const float* hid_new;       /* [N, hid_w], hid_w a multiple of 4, 16-byte aligned */
    float* small_ring;          /* [slabs, N, small_w], small_w >= act_w + n_agents + 2 */
    float* hid_ring;            /* [slabs, N, hid_w] */
    double* acc;                /* [N, 10] */
    int64_t* cursor_out;
    int64_t* aux_counter;
    int32_t act_w, hid_w, small_w, pad0;
} FlexReplaySink;
int flexenv_set_replay_sink(FlexEnv* env, const FlexReplaySink* sink /* NULL: off */);
/* A burst of `steps` rollout steps — policy evaluation (madrl/models/model.py:102-143, agents/rnn_agent.py:25-33, exploration
 * utils/util.py:57-64, translate_action utils/util.py:125-128), environment step with restart (env:218-276, model.py:255-262),
 * transition into the replay ring (model.py:230-254, utils/replay_buffer.py:23-27) — in ONE launch instead of 2 * steps: block
 * b owns environments 16 b .. 16 b + 15 for the whole burst and stages the policy's weights into its CU once.  `actor` is
 * the FlexActorArgs (include/flexnet.h) of the ring-mode flexnet_actor_forward call this replaces, with ring_slabs set; the
 * env must be configured as for flexenv_step(FLEX_STEP_AUTORESET | FLEX_STEP_OBS_RING | FLEX_STEP_REPLAY_SINK) behind that
 * call (obs ring on actor->cursor_out, sink reading actor->action / actor->hidden_out and writing actor->cursor, its
 * aux_counter = actor->rng_state + 1: the noise stream's step counter, which the burst advances by `steps`), n_agents <= 5,
 * steps < slabs; `info` and `failed` may be NULL.  Afterwards every buffer, both cursor cells and the noise position
 * hold what `steps` repetitions of the two launches leave, bit for bit (tests/test_rollout_gpu.py). FLEX_EINVAL otherwise. */
/* `safety` non-NULL: SAFEMADDPG's step (madrl/models/safemaddpg.py:90-111) — between policy and environment every
 * environment's proposed action (actor->action) goes through flexenv_safety_project_env's arithmetic (same arguments, same
 * results) and the environment steps on `env_action`; the replay keeps the policy's own action (model.py:232). */
typedef struct {
    const double* s_p;         /* dev [n_agents] */
    const double* s_q;
    const double* beta;
    double v_min, v_max, penalty;
    double* adjusted;          /* dev [N, 4 n_agents] out */
    float* env_action;         /* dev [N, 4 n_agents] out: what the step of the same launch reads */
    float act_low, act_high;
} FlexBurstSafety;
int flexenv_rollout_burst(FlexEnv* env, const FlexActorArgs* actor, double* reward, uint8_t* done,
                          double* info, uint8_t* failed, float* obs_ring, int32_t steps,
                          const FlexBurstSafety* safety /* NULL: plain MADDPG */, void* stream);
int32_t flexenv_obs_size(const FlexEnv* env);    /* 6*history, env:71 */
int32_t flexenv_state_size(const FlexEnv* env);  /* env:72 */

/* Replaces power_flow_solver_simplified (utils/pf.py:115-192) on a batch: net loads per bus in,
 * |V| per bus out.  Optional outputs (NULL to skip) are indexed by the bus at the far end of each
 * line from the slack: isqr = |I|^2, pl/ql = power received at that bus (pf.py:85-88 convention). */
int pf_solve_batch(const NetFix* net, int32_t n, const double* pnet /*dev [n, n_bus]*/,
                   const double* qnet /*dev [n, n_bus]*/, double* v /*dev [n, n_bus]*/,
                   double* isqr, double* pl, double* ql /*dev [n, n_bus] or NULL*/,
                   int32_t* iters /*dev [n] or NULL*/, uint8_t* failed /*dev [n] or NULL*/,
                   double tol, int32_t max_iter, int32_t solver, void* stream);

/* Safety layer of SAFEMADDPG.safety_layer_optimization (madrl/models/safemaddpg.py:176-299) as the
 * separable closed-form projection of SURVEY.md App. D, one lane per (env, building).
 * proposed: policy output [N, n_agents, 4] (f32/f64) -> parse_actions (safemaddpg.py:142-174) against
 * the env's current data -> per-building QP -> `adjusted` [N, 4*n_agents] f64 in the reference's
 * type-major order [pr x n | ch x n | dis x n | q x n] (safemaddpg.py:297). */
int flexenv_safety_project(FlexEnv* env, const void* proposed, int32_t dtype,
                           const double* s_p /*dev [n_agents] row sums of W_P*/,
                           const double* s_q /*dev [n_agents]*/, const double* beta /*dev [n_agents]*/,
                           double v_min, double v_max, double penalty,
                           double* adjusted /*dev [N, 4*n_agents]*/,
                           uint8_t* intervened /*dev [N] or NULL*/, void* stream);

/* The same, also handing over what env.step() is fed next (safemaddpg.py:106-111 -> model.py:217-218): env_action
 * [N, 4*n_agents] f32 = translate_action (utils/util.py:125-128: 0.5 (clamp(a, low, high) + 1) (high - low) + low) of the
 * fp32 cast of `adjusted`, in the same flat order (which env.step re-reads agent-major) — bit for bit what
 * the cast and the five tensor operations produce, without their six launches.  env_action NULL = the call above. */
int flexenv_safety_project_env(FlexEnv* env, const void* proposed, int32_t dtype, const double* s_p, const double* s_q,
                               const double* beta, double v_min, double v_max, double penalty, double* adjusted,
                               uint8_t* intervened /*dev [N] or NULL*/, float act_low, float act_high,
                               float* env_action /*dev [N, 4*n_agents] or NULL*/, void* stream);

const char* flexenv_version(void);
int32_t flexenv_abi_version(void);   /* FLEX_ABI_VERSION of the build */

#ifdef __cplusplus
}
#endif
#endif /* FLEXENV_H */
