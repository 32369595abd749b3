/* flexnet.h — C ABI of the learner-side kernels of the flexibility-provision hot path (gfx950).
 *
 * The reference's actor is a PyTorch module evaluated once per environment step and agent:
 *     madrl/agents/rnn_agent.py:13-33  RNNAgent: fc1 -> LayerNorm -> ReLU -> GRUCell -> fc2
 *     madrl/models/model.py:102-140    Model.policy: obs (+ one-hot agent id) -> means, hidden
 * flexnet_actor_forward is that forward pass (inference: no autograd graph) for a whole batch of rows in ONE launch:
 * row r is agent r % n_agents of sample r / n_agents, exactly the [b * n, .] reshape of model.py:110-112.
 * Weights are passed in the layout of the reference's state_dict (row-major [out, in]); the kernel re-lays them out
 * in LDS itself, so a state_dict loaded from the reference works unchanged.
 *
 * Plain pointers and sizes only; every pointer is DEVICE memory (fp32); asynchronous on `stream` (a hipStream_t).
 * Return 0 or a negative FLEXNET_E* code for API errors.
 */
#ifndef FLEXNET_H
#define FLEXNET_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLEXNET_OK 0
#define FLEXNET_EINVAL (-1)
#define FLEXNET_EHIP (-2)
#define FLEXNET_EUNSUPPORTED (-3)  /* shape outside what the fused kernel holds in LDS: the caller keeps its own path */

#define FLEXNET_HID 64             /* hid_size of madrl/args/default.yaml:33; the kernel maps one lane to one unit */
#define FLEXNET_MAX_OBS 144        /* obs_size = 6 * history (env:370-403), history <= 24 */
#define FLEXNET_MAX_AGENTS 8
#define FLEXNET_MAX_ACT 8

typedef struct {
    int32_t rows;              /* b * n_agents */
    int32_t n_agents;
    int32_t obs_dim;           /* <= FLEXNET_MAX_OBS */
    int32_t act_dim;           /* <= FLEXNET_MAX_ACT */
    int32_t agent_id;          /* 1: fc1 has n_agents one-hot id columns after the obs_dim observation columns (model.py:105-108) */
    int32_t layernorm;         /* args.layernorm (rnn_agent.py:19-20,27-28) */
    float ln_eps;              /* nn.LayerNorm default 1e-5 */
    int32_t variant;           /* 0: matrix-core kernel (v_mfma_f32_32x32x2_f32); 1: the VALU kernel (kept as cross-check) */
    const float* obs;          /* [rows, obs_dim] */
    const float* hidden_in;    /* [rows, 64] */
    const float* fc1_w;        /* [64, obs_dim (+ n_agents)] */
    const float* fc1_b;        /* [64] */
    const float* ln_w;         /* [64] (ignored without layernorm) */
    const float* ln_b;         /* [64] */
    const float* w_ih;         /* [192, 64]  GRUCell.weight_ih, gate order r, z, n */
    const float* w_hh;         /* [192, 64] */
    const float* b_ih;         /* [192] */
    const float* b_hh;         /* [192] */
    const float* fc2_w;        /* [act_dim, 64] */
    const float* fc2_b;        /* [act_dim] */
    float* means;              /* out [rows, act_dim]  (fc2 output: the action mean, rnn_agent.py:32) */
    float* hidden_out;         /* out [rows, 64]       (GRU state, rnn_agent.py:31) */
    /* optional exploration epilogue (all three NULL to skip): utils/util.py:57-64 select_action in training mode with
     * the fixed std of model.py:121-123, and utils/util.py:125-128 translate_action's arithmetic */
    const float* noise;        /* [rows, act_dim] standard normal draws */
    float* action;             /* out [rows, act_dim]  tanh(mean + std * noise) */
    float* env_action;         /* out [rows, act_dim]  0.5 * (clamp(action, low, high) + 1) * (high - low) + low */
    float std, action_low, action_high, pad1;
    /* Alternative to `noise` (leave that NULL): the kernel draws the standard normal numbers itself — Philox4x32-10 keyed
     * by rng_state[0] (seed) with counter (row, action group, rng_state[1] = step), Box-Muller on 24-bit uniforms.  The
     * caller (or flexnet_rollout_pack, given the same pointer) advances rng_state[1] between launches.  variant 0 only. */
    const uint64_t* rng_state; /* device [2] or NULL */
    /* Inputs in a slab ring (the replay ring of flexnet_rollout_pack, written in place by the environment kernel):
     * with `cursor` non-NULL the launch reads obs + cursor[0] * obs_slab_stride and hidden_in + cursor[0] *
     * hid_slab_stride (strides in floats; cursor[0] is read on the device, so the launch replays from a HIP graph). */
    const int64_t* cursor;     /* device [1] or NULL */
    int64_t obs_slab_stride;
    int64_t hid_slab_stride;
    int64_t* cursor_out;       /* device [1] or NULL: one lane writes the slab index this launch read there (never read by
                                  this launch) — the cell the environment's step launch takes its slab from when it files
                                  the transition itself (FLEX_STEP_REPLAY_SINK, include/flexenv.h) */
    /* Training forward (variant 0): what the backward pass of rnn_agent.py:25-33 needs, stored on the way ([rows, 64]
     * each, all six or none): fc1's raw output (before bias / id column / LayerNorm), the GRU input x = ReLU(LayerNorm(.)),
     * the gates r, z, n and the hidden part of the candidate, W_hn h + b_hn.  flexnet_gru_backward consumes them. */
    float* save_z1;
    float* save_x;
    float* save_r;
    float* save_z;
    float* save_n;
    float* save_hn;
    int64_t ring_slabs;        /* flexenv_rollout_burst only (include/flexenv.h): slabs of the two input rings, for the wrap of
                                  the slab index from one step of the burst to the next; 0 elsewhere */
    /* Observations read IN PLACE from an environment's history (include/flexenv.h: flexenv_obs_source; FLEX_STEP_OBS_ROWS):
     * with obs_pushed non-NULL row r's observation is the obs_dim CONTIGUOUS floats at
     *     obs + r * obs_row_stride + ((obs_pushed[(r / n_agents) * obs_pushed_stride] - 1) mod obs_slots + 1) * obs_slot_w
     * (the last obs_slots rows of a mirror ring: no wrap, zeros where the episode had not begun) instead of obs + r * obs_dim;
     * `cursor` / the slab strides then apply to hidden_in only.  obs_slots * obs_slot_w == obs_dim. */
    const int32_t* obs_pushed; /* device, or NULL: obs is [rows, obs_dim] */
    int32_t obs_row_stride, obs_pushed_stride, obs_slots, obs_slot_w;
} FlexActorArgs;

/* rnn_agent.py:25-33 + model.py:102-116 for all rows. */
int flexnet_actor_forward(const FlexActorArgs* args, void* stream);

/* Pointwise backward of the GRUCell and of fc2's input (rnn_agent.py:30-32) for a training batch, from the tensors
 * flexnet_actor_forward saved: with h' = (1 - z) n + z h,
 *     dh'  = d_means @ fc2_w (+ d_hidden_out)
 *     dn~  = dh' (1 - z) (1 - n^2)      dz~ = dh' (h - n) z (1 - z)      dr~ = dn~ (W_hn h + b_hn) r (1 - r)
 *     d_gi = [dr~ | dz~ | dn~]  [rows, 192]   (gradient of W_ih x + b_ih; W_hh's first 128 rows see the same dr~, dz~)
 *     d_gh = [dr~ | dz~ | dn~ r] [rows, 192]  (gradient of W_hh h + b_hh)
 * Weight and bias gradients are then flexnet_wgrad products of d_gi / d_gh with x and h; dx = d_gi @ W_ih.  The previous
 * hidden state takes no gradient (a replayed tensor). */
typedef struct {
    int32_t rows;
    int32_t act_dim;           /* <= FLEXNET_MAX_ACT */
    const float* d_means;      /* [rows, act_dim] */
    const float* d_hidden;     /* [rows, 64] or NULL: gradient arriving at the new hidden state directly */
    const float* fc2_w;        /* [act_dim, 64] */
    const float* r;            /* saved by flexnet_actor_forward */
    const float* z;
    const float* n;
    const float* hn;
    const float* h_prev;       /* [rows, 64] */
    float* d_gi;               /* out [rows, 192] */
    float* d_gh;               /* out [rows, 192] */
    /* Round 3, optional (dz NULL: the pointwise pass above): the first layer's backward in the same launch —
     * dx = d_gi @ W_ih on the matrix cores (never stored), then the backward of x = ReLU(LayerNorm(z1 + fc1.bias + id column
     * of row % n_agents)) (rnn_agent.py:25-29; flexnet_lnrelu_backward's arithmetic): dz [rows, 64] (the gradient of fc1's
     * raw output: flexnet_wgrad against the observations gives fc1's weight gradient) and the four parameter gradients, the
     * id-column sums written through strides (e.g. straight into fc1's gradient: agent stride 1, unit stride its row pitch). */
    const float* w_ih;         /* [192, 64] */
    const float* z1;           /* [rows, 64]  save_z1 of flexnet_actor_forward */
    const float* x;            /* [rows, 64]  save_x, or NULL: ReLU's mask is then recomputed from z1 */
    const float* fc1_w;        /* [64, fc1_ld]: the id columns are read in place (agent_id) */
    const float* fc1_b;        /* [64] or NULL */
    const float* ln_w;         /* [64] (layernorm) */
    const float* ln_b;
    float* dz;                 /* out [rows, 64] */
    float* d_ln_w;             /* out [64] or NULL */
    float* d_ln_b;
    float* d_fc1_b;
    float* d_id;               /* out, n_agents x 64 elements through the two strides below, or NULL */
    float* workspace;          /* >= FLEXNET_LNRELU_WS_FLOATS floats */
    int64_t workspace_floats;
    int64_t d_id_agent_stride, d_id_unit_stride;
    int32_t fc1_ld, obs_dim, n_agents, agent_id, layernorm;
    float ln_eps;
} FlexGruBwdArgs;

int flexnet_gru_backward(const FlexGruBwdArgs* args, void* stream);

/* The centralised critic (madrl/critics/mlp_critic.py:5-34: fc1 -> LayerNorm -> ReLU -> fc2 -> ReLU -> fc3) AFTER its
 * first layer: the caller forms fc1's output z1 from column blocks (maddpg.py:38-54 repeats every agent's observation
 * n times in fc1's input; one GEMM per sample instead of n), these two entry points do the rest — forward, and the
 * backward pass that autograd would otherwise spread over some fifteen kernels.  One lane per hidden unit.
 * Gradient outputs of the backward call are ACCUMULATED into (atomics): the caller zeroes them. */
typedef struct {
    int32_t rows;              /* b * n_agents */
    int32_t layernorm;         /* args.layernorm (mlp_critic.py:12-13,27-28) */
    float ln_eps;
    int32_t variant;           /* 0: matrix-core kernels for the forward and the dz1-only backward; 1: the VALU kernels */
    const float* z1;           /* [rows, 64]  fc1 output; or NULL: z1[r] = z_shared[r / n_agents] + z_id[r % n_agents] */
    const float* ln_w;         /* [64] */
    const float* ln_b;         /* [64] */
    const float* fc2_w;        /* [64, 64] */
    const float* fc2_b;        /* [64] */
    const float* fc3_w;        /* [1, 64] */
    const float* fc3_b;        /* [1] */
    float* q;                  /* out [rows]  (mlp_critic.py:31: v) */
    /* backward only (NULL for the forward call) */
    const float* dq;           /* [rows]      dLoss/dq */
    float* dz1;                /* out [rows, 64] */
    float* d_ln_w;             /* += [64]   (d_fc2_w == NULL: no parameter gradients at all, dz1 only) */
    float* d_ln_b;             /* += [64] */
    float* d_fc2_w;            /* += [64, 64] */
    float* d_fc2_b;            /* += [64] */
    float* d_fc3_w;            /* += [64] */
    float* d_fc3_b;            /* += [1] */
    const float* z_shared;     /* [rows / n_agents, 64]  the part of fc1's output all agents of a sample share (maddpg.py:38-54) */
    const float* z_id;         /* [n_agents, 64]         fc1's one-hot id columns, transposed */
    int32_t n_agents;          /* used with z_shared */
    int32_t overwrite_grads;   /* backward, fixed-order path only: the six parameter gradients are STORED, not added to
                                * (no zero fill by the caller); with the atomic path set: FLEXNET_EINVAL */
    float* d_z_shared;         /* backward, composed input: out [rows / n_agents, 64] = sum over a sample's agents of dz1, or NULL
                                * (set: the contents of dz1 after the call are unspecified, see variant_pgrad32) */
    float* d_z_id;             /* backward, composed input: out [n_agents, 64] = sum over samples of dz1, or NULL (both or none;
                                * needs the workspace) */
    float* workspace;          /* backward: scratch for the per-block partial sums, or NULL */
    int64_t workspace_floats;  /* >= FLEXNET_CRITIC_WS_FLOATS: the parameter gradients are then reduced in a fixed order
                                * (bit-reproducible) by a second small launch; otherwise with atomics */
    int32_t d_z_id_agent_stride;   /* d_z_id element (agent i, unit u) is stored at i * agent_stride + u * unit_stride floats; */
    int32_t d_z_id_unit_stride;    /* 0, 0 = the dense [n_agents, 64] layout (64, 1).  (1, fc1 row length) writes the id */
                                   /* columns of fc1.weight's gradient in place */
    int32_t z_id_agent_stride;     /* the same for the INPUT z_id: (1, fc1 row length) reads the id columns of fc1.weight where */
    int32_t z_id_unit_stride;      /* they are (every kernel stages the [n_agents, 64] table in LDS once per block) */
    /* dz1-only matrix-core backward (d_fc2_w == NULL, variant 0, rows >= 65 536) of a loss that is a scaled mean of q
     * (the policy loss -mean Q(s, pi(s)), maddpg.py:104-107): every row's dLoss/dq is `dq_value` (`dq` is not read), and
     * the kernel — which recomputes the forward anyway — also returns q_mean_scale * sum(q) in *q_mean_out (fixed-order
     * sums, fp64 partials in the workspace).  The two come together; set elsewhere: FLEXNET_EUNSUPPORTED. */
    int32_t dq_uniform;
    float dq_value;
    float q_mean_scale;
    int32_t variant_pgrad32;       /* backward WITH parameter gradients on the matrix cores (rows >= 65 536, fixed-order workspace):
                                    * 0 = 16-row tiles, two wavefronts per SIMD (round 3) — on a composed input with d_z_shared
                                    *     set, in its sample-major form: d_z_shared and d_z_id come out of the backward kernel
                                    *     itself and `dz1` is NOT written (scratch the caller still provides);
                                    * 1 = the 32-row kernel, one wavefront per SIMD, dz1 stored and folded by a second kernel;
                                    * 2 = 16-row tiles over consecutive rows, dz1 stored and folded (cross-checks, A/B) */
    float* q_mean_out;             /* NULL: no sum of q */
} FlexCriticTailArgs;

#define FLEXNET_CRITIC_WS_FLOATS (1024 * 4416)

int flexnet_critic_tail_forward(const FlexCriticTailArgs* args, void* stream);
int flexnet_critic_tail_backward(const FlexCriticTailArgs* args, void* stream);

/* Weight gradient of a linear layer over a tall batch: C[m, n] = sum_k A[k, m] * B[k, n] (A = dLoss/dY, B = the
 * layer's input, k = the batch rows) — what loss.backward() of madrl/utils/trainer.py:62-111 computes for fc1 / the
 * GRUCell / fc2 of rnn_agent.py:13-33 and fc1 of mlp_critic.py:5-34.  fp32 products and sums on the matrix cores
 * (v_mfma_f32_32x32x2_f32), summed over k in a fixed order: bit-reproducible.  m <= 192; else FLEXNET_EUNSUPPORTED. */
typedef struct {
    int64_t k;                 /* rows summed over */
    int64_t lda, ldb;          /* row pitch of A and B in floats (>= m, n): column slices of wider records are fine */
    int64_t workspace_floats;  /* >= FLEXNET_WGRAD_WS_FLOATS */
    int32_t m, n;
    int32_t accumulate;        /* 1: C += result */
    int32_t ldc;               /* row pitch of C in floats; 0 = n (dense).  A column block of a wider weight gradient works */
    const float* a;            /* [k, m] */
    const float* b;            /* [k, n] */
    float* c;                  /* out [m, n] */
    float* workspace;
    float* colsum;             /* out [m] = sum_k A[k, m] (the layer's bias gradient), or NULL */
    /* optional second input block (NULL / 0: none): B is then the column concatenation [b | b2] read in place from its two
     * homes — the critic's observation block and action block, mlp_critic.py:25 after maddpg.py:47-54 — and the result
     * columns n .. n + n2 - 1 go to c2.  One launch instead of two.  n must be a multiple of the kernel's lane width
     * for this shape (5 for 32 < m <= 64 and n > 64; anything else: FLEXNET_EUNSUPPORTED, make two calls). */
    const float* b2;           /* [k, n2] */
    float* c2;                 /* out [m, n2] */
    int64_t ldb2;              /* row pitch of b2 */
    int32_t n2;
    int32_t ldc2;              /* row pitch of c2; 0 = n2 */
    /* optional (NULL: none): B's rows are read in place from a larger row store — the replay's stacked-observation ring —
     * starting at row *b_row_cell (device int64, read by the kernel: a replayed HIP graph follows the sampled window without
     * a copy of it); `b` is then the store's base, rows *b_row_cell .. + k - 1 must lie inside it. */
    const int64_t* b_row_cell;
} FlexWgradArgs;

#define FLEXNET_WGRAD_WS_FLOATS (520 * 12288 + 520 * 192)

int flexnet_wgrad(const FlexWgradArgs* args, void* stream);

/* The actor's first-layer epilogue at update batches (rnn_agent.py:25-29 with the one-hot id columns of
 * model.py:105-108 folded in): out = relu(LayerNorm(z + bias + id_cols[r % n_agents])) for z = obs @ W_obs^T, and the
 * backward of that chain with its four parameter gradients (OVERWRITTEN, summed in a fixed order). */
typedef struct {
    int32_t rows, n_agents;
    int32_t layernorm;         /* args.layernorm */
    float ln_eps;
    const float* z;            /* [rows, 64] */
    const float* bias;         /* [64] or NULL */
    const float* id_cols;      /* [n_agents, 64] (fc1.weight[:, obs:obs+n] transposed) or NULL */
    const float* ln_w;         /* [64] */
    const float* ln_b;         /* [64] */
    float* out;                /* forward: [rows, 64] */
    /* backward only */
    const float* dout;         /* [rows, 64] */
    float* dz;                 /* out [rows, 64] */
    float* d_bias;             /* out [64] or NULL */
    float* d_id;               /* out [n_agents, 64] or NULL */
    float* d_ln_w;             /* out [64] or NULL */
    float* d_ln_b;             /* out [64] or NULL */
    float* workspace;
    int64_t workspace_floats;  /* >= FLEXNET_LNRELU_WS_FLOATS */
} FlexLnReluArgs;

#define FLEXNET_LNRELU_WS_FLOATS (1024 * 704)

int flexnet_lnrelu_forward(const FlexLnReluArgs* args, void* stream);
int flexnet_lnrelu_backward(const FlexLnReluArgs* args, void* stream);

/* clip_grad_norm_(params, max_norm) followed by torch.optim.RMSprop.step() (alpha, eps; no momentum, not centred, no
 * weight decay — madrl/utils/trainer.py:34-35,86-90,103-107) for all tensors of one network in two small launches. */
#define FLEXNET_OPT_MAX_TENSORS 16
#define FLEXNET_OPT_MAX_ELEMENTS (1 << 20)
#define FLEXNET_OPT_WS_FLOATS 64
typedef struct {
    int32_t n_tensors;         /* <= FLEXNET_OPT_MAX_TENSORS; tensors whose gradient is None are simply not listed */
    float lr, alpha, eps;
    float max_norm;            /* <= 0: no clipping */
    int32_t pad0;
    float* total_norm;         /* out [1]: the pre-clip 2-norm over all listed gradients, or NULL */
    float* workspace;          /* FLEXNET_OPT_WS_FLOATS floats of scratch */
    int64_t numel[FLEXNET_OPT_MAX_TENSORS];
    float* param[FLEXNET_OPT_MAX_TENSORS];
    float* grad[FLEXNET_OPT_MAX_TENSORS];        /* scaled in place by the clip factor, as clip_grad_norm_ does */
    float* square_avg[FLEXNET_OPT_MAX_TENSORS];  /* RMSprop state */
    float* step[FLEXNET_OPT_MAX_TENSORS];        /* RMSprop's per-tensor step counter (fp32 scalar, += 1) or NULL */
} FlexClipRmspropArgs;

int flexnet_clip_rmsprop(const FlexClipRmspropArgs* args, void* stream);

/* The value loss of madrl/models/maddpg.py:100-123 behind model.py:308-323's reward normalisation, and its gradient:
 *     r = BatchNorm1d(n)(reward) [train mode];  ret = r + gamma (1 - done) next_q;  loss = mean((ret - q)^2);
 *     dq = dLoss/dq = -2 (ret - q) / (rows n).   The module's running statistics move as nn.BatchNorm1d moves them.
 * With q == NULL (and next_q, dq, loss NULL, normalise = 1) only the statistics pass runs: the running statistics are
 * updated, nothing else — a get_loss call that does not use the value loss still owes the module that update. */
typedef struct {
    int32_t rows, n_agents;    /* [rows, n_agents] tensors; n_agents <= FLEXNET_MAX_AGENTS */
    int32_t normalise;         /* args.reward_normalisation */
    float gamma, bn_eps, bn_momentum;
    const float* reward;       /* [rows, n] raw */
    const float* done;         /* [rows] 0/1 */
    const float* next_q;       /* [rows, n] bootstrap values (no gradient) */
    const float* q;            /* [rows, n] */
    const float* bn_weight;    /* [n] or NULL (1) */
    const float* bn_bias;      /* [n] or NULL (0) */
    float* running_mean;       /* [n] updated in place, or NULL */
    float* running_var;        /* [n] */
    int64_t* num_batches_tracked; /* += 1, or NULL */
    float* dq;                 /* out [rows, n] */
    float* loss;               /* out [1] */
    float* workspace;          /* 8-byte aligned */
    int64_t workspace_floats;  /* >= FLEXNET_TD_WS_FLOATS */
    /* Reward statistics over MORE than this call's rows (data-parallel ranks): run flexnet_td_stats, sum the
     * first FLEXNET_TD_STAT_DOUBLES doubles of the workspace over the ranks (one all-reduce of 8 KB), then make the call
     * with stats_ready = 1 and stat_rows = the rows those sums cover.  0 / 0: the call computes its own statistics. */
    int32_t stats_ready;
    int32_t pad0;
    int64_t stat_rows;
} FlexTdLossArgs;

#define FLEXNET_TD_WS_FLOATS (2 * (64 * 2 * 8 + 1024))
#define FLEXNET_TD_STAT_DOUBLES (64 * 2 * 8)   /* per-block partial sums and sums of squares of the reward columns */

int flexnet_td_loss(const FlexTdLossArgs* args, void* stream);
/* The statistics pass alone (args->reward, rows, n_agents, workspace): per-block partial column sums into the workspace. */
int flexnet_td_stats(const FlexTdLossArgs* args, void* stream);

/* The value loss AND the critic's backward in one pass (maddpg.py:100-123 over mlp_critic.py:25-33): the matrix-core
 * backward kernel recomputes the tail's forward anyway, so it forms q, the TD error, dLoss/dq and the loss partial sums
 * itself — no forward launch of the tail, no q / dq round trip.  `critic`: as for flexnet_critic_tail_backward, `dq`
 * unused; `td`: as for flexnet_td_loss with rows * n_agents == critic->rows; td->q and td->dq are optional OUTPUTS here
 * ([rows * n_agents] each, NULL = not stored).  Covers what the matrix-core backward covers (variant 0, parameter
 * gradients through the fixed-order workspace path, critic->rows >= 65 536); otherwise FLEXNET_EUNSUPPORTED and the caller
 * makes the two separate calls around a forward. */
int flexnet_critic_td_backward(const FlexCriticTailArgs* critic, const FlexTdLossArgs* td, void* stream);
/* The same in separately launchable phases, for callers that overlap the small launches with independent work on another
 * stream (round 5: the statistics pass beside the first-layer GEMM, the finish beside the first layer's weight gradient):
 * phases = 1: the statistics pass (unless td->stats_ready) and the backward kernel; 2: the finish launch (parameter
 * gradients from the partial rows, loss, running statistics) — it needs phase 1 of the same arguments complete on its stream
 * and nothing of flexnet_wgrad's; 3: both, = flexnet_critic_td_backward. */
int flexnet_critic_td_backward_phases(const FlexCriticTailArgs* critic, const FlexTdLossArgs* td, int32_t phases, void* stream);
/* Phase 2 of the above riding in the second-stage launch of the critic's first-layer weight gradient (round 5): what
 * flexnet_critic_td_backward_phases(critic, td, 2, stream) followed by flexnet_wgrad(w, stream) compute, in one launch fewer
 * (model.py:43-50 runs ten value sub-updates per event; a replayed graph's small launches each wait for the one before).
 * Needs phase 1 of the same (critic, td) complete on `stream`.  Only the weight gradient of a 64-unit layer over more than
 * 64 input columns carries the rider: FLEXNET_EUNSUPPORTED otherwise, with nothing launched. */
int flexnet_wgrad_critic_finish(const FlexWgradArgs* w, const FlexCriticTailArgs* critic, const FlexTdLossArgs* td, void* stream);

/* out[0] = scale * sum(x[0 .. n)) in a FIXED order (fp64 partial sums of 64 blocks, one-wavefront finish): the scalar
 * means the losses report — policy_loss = -Q(s, pi(s)).mean() (madrl/models/maddpg.py:107), the entropy of
 * utils/trainer.py:48-56 — without ATen's multi-block reduction, whose global semaphore does not survive a HIP-graph
 * replay on this stack.  No memset, no atomics. */
typedef struct {
    int64_t n;                 /* elements, >= 1 */
    float scale;               /* 1 / n for a mean */
    int32_t pad0;
    const float* x;            /* [n] contiguous */
    float* out;                /* out [1] */
    float* workspace;          /* 8-byte aligned, >= FLEXNET_SUM_WS_FLOATS floats */
    int64_t workspace_floats;
} FlexSumArgs;

#define FLEXNET_SUM_WS_FLOATS (2 * 64)

int flexnet_scaled_sum(const FlexSumArgs* args, void* stream);

/* One vector step's bookkeeping of the rollout (madrl/models/model.py:230-262 per environment, utils/replay_buffer.py:
 * 23-27) in ONE launch.  The replay ring is slab-structured — slab k holds, for every environment of vector step k,
 * the observation the policy acted on, the hidden state it started from, and (once the step has run) its action,
 * reward, done and last_step flags:
 *     obs_ring   [slabs, N, n * obs_dim]      obs(k)
 *     hid_ring   [slabs, N, n * 64]           hidden state going INTO step k (model.py:211,259; zero after a terminal step)
 *     small_ring [slabs, N, small_w]          [action n*act | reward n | done | last_step | pad]
 * Every observation is stored ONCE: the next_state of a transition in slab k is obs(k + 1), the `hid` of model.py:241 is
 * the hidden state of slab k + 1 (zeroed where done = 1 — where the TD target multiplies the bootstrap value by zero
 * anyway, maddpg.py:117) and `last_hid` is the slab's own.  This launch, after step k ran:
 *     slab k     <- action, reward, done (last_step = done; the caller flags the final step of a horizon, model.py:229)
 *     slab k + 1 <- obs_next (skipped when NULL: the environment kernel wrote it there itself, FLEX_STEP_OBS_RING of
 *                   include/flexenv.h), hid_new * (1 - done)
 *     hid_state  <- hid_new * (1 - done)      (hand-over to a policy that does not read the ring; skipped when NULL)
 *     statistics += this step's info / reward / failure sums (fixed-order block sums)
 * Slab indices live on the device (the launch replays from a HIP graph) in two cells, each written by exactly one kernel
 * of the step so that no launch reads a cell it also writes: cursor[0] = the slab the policy reads (advanced HERE, by one
 * lane, when cursor_stepped), cursor[1] = the slab being filled next (advanced by the environment's step kernel,
 * flexenv_set_step_counter, before this launch).  Without an environment counter (cursor_stepped = 0) this launch reads
 * cursor[0] and a one-thread launch behind it advances both cells. */
typedef struct {
    int32_t n_envs, n_agents, obs_dim, act_dim;
    int32_t slabs;             /* ring capacity in slabs, >= 2 */
    int32_t small_w;           /* floats per small record: >= n_agents * act_dim + n_agents + 2 */
    int32_t info_w;            /* columns of `info` (7) */
    int32_t cursor_stepped;    /* 1: cursor[1] was already advanced for this step by flexenv_step; 0: see above */
    const float* action;       /* [N, n, act_dim]  what the replay keeps (model.py:232) */
    const double* reward;      /* [N]   one reward per environment, stored once per agent */
    const float* obs_next;     /* [N, n, obs_dim], or NULL (already in the ring) */
    const uint8_t* done;       /* [N] */
    const float* hid_new;      /* [N, n, 64] */
    const double* info;        /* [N, info_w] or NULL */
    const uint8_t* failed;     /* [N] or NULL */
    float* obs_ring;
    float* hid_ring;
    float* small_ring;
    float* hid_state;          /* out [N, n, 64] or NULL; may alias hid_new */
    int64_t* cursor;           /* [2]: physical slab indices, see above */
    double* info_sum;          /* [info_w] or NULL */
    double* rew_sum;           /* [1] */
    double* fail_sum;          /* [1] or NULL */
    int64_t* rng_state;        /* [2] or NULL: {seed, step}; step += 1 (the actor kernel's noise stream, flexnet_actor_forward) */
} FlexRolloutPackArgs;

int flexnet_rollout_pack(const FlexRolloutPackArgs* args, void* stream);

/* Agent-summed exploration of MATD3 / IDDPG (matd3.py:92-97 + utils/util.py:57-64, iddpg.py:66-71): the policy means of a
 * sample are summed over the AGENT axis, ONE action tanh(sum + std * eps) is drawn and every agent receives it; the env is
 * fed translate_action of it (util.py:125-128).  One launch for what the tensor composition spreads over ~40:
 *     y[e, k]          = tanh(((m[e,0,k] + m[e,1,k]) + ...) + eps[e, k] * std[k])
 *     action[e, i, k]  = y[e, k]
 *     env_action[e, i, k] = 0.5 (clamp(y, low, high) + 1) (high - low) + low
 * every step rounded to fp32 in the order the tensor operations round it (bit-identical to them). */
typedef struct {
    int32_t n_envs, n_agents, act_dim, pad0;
    float act_low, act_high;
    const float* means;        /* [N, n, a] */
    const float* eps;          /* [N, a] standard normal draws; NULL: no exploration, action = the summed mean (no tanh) */
    const float* std;          /* [a] exp of the agent-summed log-std */
    float* action;             /* out [N, n, a] */
    float* env_action;         /* out [N, n, a], or NULL (bootstrap actions of a value loss: no environment behind them) */
} FlexAgentSumArgs;

int flexnet_agent_sum_explore(const FlexAgentSumArgs* args, void* stream);

/* The shared part of the centralised critic's first layer (madrl/critics/mlp_critic.py:25-26 on the input of
 * madrl/models/maddpg.py:33-54) as ONE launch on the matrix cores, exact fp32 (csrc/linear.hip: weights stationary in
 * registers, inputs straight from memory into MFMA operands):
 *     out[b, 0:64] = bias + x1[b, 0:k1] W[:, c1:c1+k1]^T + x2[b, 0:k2] W[:, c2:c2+k2]^T
 * x1 = every agent's observation ([b, n * obs_dim], row pitch ld1), x2 = every agent's action ([b, n * act_dim]; k2 = 0:
 * none), W = fc1.weight as stored ([64, ldw], the id columns between the two blocks are skipped).  k1, k2, ld1, ld2
 * multiples of 4, x1 / x2 / out 16-byte aligned, k1 + k2 <= 1280; otherwise FLEXNET_EUNSUPPORTED (the caller keeps the two
 * library GEMMs).  Bit-reproducible: every output is a fixed-order sum of four matrix-core accumulations. */
typedef struct {
    int64_t rows;
    int32_t k1, k2, ld1, ld2, ldw, c1, c2, pad0;
    const float* x1;           /* [rows, ld1] */
    const float* x2;           /* [rows, ld2] or NULL */
    const float* w;            /* [64, ldw] */
    const float* bias;         /* [64] */
    float* out;                /* out [rows, 64] */
    const int64_t* x1_row_cell; /* optional (NULL: none): as FlexWgradArgs.b_row_cell — x1 is the base of a row store, the rows
                                  of this call start at row *x1_row_cell (device int64) */
} FlexLinear2Args;
int flexnet_linear2(const FlexLinear2Args* args, void* stream);

/* Stacked observations of a replay window out of the ROW ring (include/flexenv.h: FLEX_STEP_OBS_RING) — the gather of the
 * window IS the im2col: the replay keeps every feature row once, [slabs][N][n_agents][FLEX_ROW_FLOATS] =
 * [Pd, Qd, Ppv, V, price, E, older, 0], and output row i (global slot first_slot + i: slab (slot / N) mod slabs, environment
 * slot mod N) becomes env:387-401's [n_agents, 6 * history] observation of that slab:
 *     dst[i][a][h * 6 + f] = row_ring[slab - (history - 1 - h)][env][a][f]   if history - 1 - h <= older(slab, env, a)
 *                          = 0                                              otherwise (before the episode began)
 * (slab indices modulo `slabs`).  The caller keeps the history - 1 slabs behind every window it asks for intact. */
typedef struct {
    const float* row_ring;
    float* dst;                /* out [rows, n_agents * 6 * history], 8-byte aligned */
    int64_t rows;
    int64_t first_slot;
    int32_t n_envs, n_agents, history, slabs;
} FlexWindowArgs;
int flexnet_gather_window(const FlexWindowArgs* args, void* stream);

/* Replay-window refresh: up to FLEXNET_GATHER_MAX_JOBS strided row copies in one launch.  A sampled window of the slab
 * ring (utils/replay_buffer.py:17-21: consecutive transitions) becomes the contiguous static batch a captured sub-update
 * reads; each field is one job (two where the window crosses the ring's seam). */
#define FLEXNET_GATHER_MAX_JOBS 12
typedef struct {
    int32_t n_jobs, pad0;
    const float* src[FLEXNET_GATHER_MAX_JOBS];
    float* dst[FLEXNET_GATHER_MAX_JOBS];
    int64_t rows[FLEXNET_GATHER_MAX_JOBS];
    int32_t width[FLEXNET_GATHER_MAX_JOBS];        /* floats per row */
    int32_t src_stride[FLEXNET_GATHER_MAX_JOBS];   /* floats between rows */
    int32_t dst_stride[FLEXNET_GATHER_MAX_JOBS];
} FlexGatherArgs;

int flexnet_gather_rows(const FlexGatherArgs* args, void* stream);
/* The same launch carrying the reward-statistics pass of the value loss (= flexnet_td_stats(td), model.py:308-323) as extra
 * blocks (round 5): jobs reward_job .. reward_job + reward_jobs - 1 (one, or two for a window that wraps the ring's seam)
 * are the copies of the [td->rows, td->n_agents] reward rows; the statistics are taken from the rows those jobs read, in the
 * partition flexnet_td_stats uses (bit-identical sums), into td->workspace.  The consumer then runs with stats_ready = 1. */
int flexnet_gather_rows_td(const FlexGatherArgs* args, int32_t reward_job, int32_t reward_jobs, const FlexTdLossArgs* td, void* stream);

/* The refresh of a captured sub-update's static batch with the window's first slot read from DEVICE memory (round 5): the
 * form of flexnet_gather_rows a HIP graph can hold — one graph then runs ALL the value sub-updates of an update event
 * (model.py:47-50), each refreshing its batch from its own cell of a device array of window starts, instead of one graph
 * launch per sub-update with a host-side refresh in between (8 us of graph hand-over each).
 * Job j copies rows [start + row_off[j], + rows[j]) (taken modulo ring_rows: a window may wrap the ring's seam) of a ring
 * whose rows are src_stride[j] floats apart, columns base[j] .. + width[j] - 1, into the contiguous dst[j]; cell[k] <-
 * start modulo cell_mod[k] (the in-place windows' first-row cells, nets.RING_VIEWS).  td != NULL: the reward-statistics pass
 * rides in the launch as in flexnet_gather_rows_td, on job reward_job. */
#define FLEXNET_WINDOW_MAX_JOBS 8
#define FLEXNET_WINDOW_MAX_CELLS 4
typedef struct {
    int32_t n_jobs, n_cells;
    const int64_t* start;                                  /* device: the window's first global slot (>= 0) */
    int64_t ring_rows;                                     /* rows of every ring the jobs read (slabs * n_envs) */
    const float* base[FLEXNET_WINDOW_MAX_JOBS];            /* ring row 0, first column of the job */
    float* dst[FLEXNET_WINDOW_MAX_JOBS];
    int64_t rows[FLEXNET_WINDOW_MAX_JOBS];
    int64_t row_off[FLEXNET_WINDOW_MAX_JOBS];
    int32_t width[FLEXNET_WINDOW_MAX_JOBS];
    int32_t src_stride[FLEXNET_WINDOW_MAX_JOBS];
    int64_t* cell[FLEXNET_WINDOW_MAX_CELLS];
    int64_t cell_mod[FLEXNET_WINDOW_MAX_CELLS];
    int32_t reward_job, pad0;                              /* used with td only */
} FlexWindowRefreshArgs;

int flexnet_window_refresh(const FlexWindowRefreshArgs* args, const FlexTdLossArgs* td /* NULL: none */, void* stream);
/* flexnet_clip_rmsprop(opt) followed by flexnet_window_refresh(refresh, td) without a launch for the refresh: inside an update
 * event's graph the optimiser step that ends one value sub-update is followed by the refresh for the next, which depends on
 * nothing the step computes — its copy blocks ride in the norm launch, its statistics blocks in the step launch, where they
 * read the rewards from the copy just made: td->reward must be dst[reward_job] (FLEXNET_EINVAL otherwise).  The caller
 * guarantees that nothing still reads what the refresh writes (trainer.py:81-108 order: the loss's kernels are done). */
int flexnet_clip_rmsprop_refresh(const FlexClipRmspropArgs* opt, const FlexWindowRefreshArgs* refresh, const FlexTdLossArgs* td, void* stream);

#ifdef __cplusplus
}
#endif
#endif
