/* myobatch.h — C ABI of libmyobatch: batched MyoSuite-class muscle-tendon physics + the
 * Baoding task layer, resident on one MI355X.
 *
 * The reference has no native boundary; the seam this library sits behind is the per-process
 * gym API that stable-baselines3's SubprocVecEnv drives over pipes:
 *     SubprocVecEnv([thunk]*n)            /root/reference/src/main_baoding.py:56-65
 *     env.reset() / env.step(a)           /root/reference/src/main_eval.py:92,104
 *     EnvironmentFactory.create(name,**kw) /root/reference/src/envs/environment_factory.py:8-63
 * Each entry point below names the reference interface it replaces.  All pointers marked
 * "dev" are device (HBM) pointers owned by the caller (torch tensors); `stream` is a
 * hipStream_t passed as void* (NULL = default stream).  Every function returns 0 on success
 * or a negative MYO_E_* code; myo_last_error() gives the thread-local message.  A numerical
 * blow-up of one environment is not an error: the env reports done=1 and is reset
 * (MuJoCo's mj_checkPos/Vel/Acc -> mj_resetData behaviour).
 *
 * There is no CPU execution path in this library.
 */
#ifndef MYOBATCH_H
#define MYOBATCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MYO_OK 0
#define MYO_E_ARG -1      /* bad argument / malformed blob */
#define MYO_E_UNSUPPORTED -2 /* model uses a feature the stepper lacks or exceeds its limits */
#define MYO_E_DEVICE -3   /* HIP runtime error */
#define MYO_E_STATE -4    /* call sequence error */

typedef struct myo_model myo_model;
typedef struct myo_batch myo_batch;

/* arithmetic of the stepper.  MYO_F64: everything in fp64 (the reference's mjtNum).  MYO_MIXED: fp64 for what
 * decides whether a trajectory stays on the fp64 one — the state (qpos, qvel, act, time) and its integration, the
 * kinematic chain, contact / joint-limit distances, the tendon-wrap predicates, the Newton cost, the observation —
 * and fp32 for the rest of the dynamics.  MYO_F32 is round 1's name for value 1. */
enum { MYO_F64 = 0, MYO_MIXED = 1, MYO_F32 = 1 };
enum { MYO_TASK_NONE = 0, MYO_TASK_BAODING_P1 = 1, MYO_TASK_BAODING_P2 = 2, MYO_TASK_REORIENT = 3 };
enum { MYO_WHICH_HOLD = 0, MYO_WHICH_CW = 1, MYO_WHICH_CCW = 2 }; /* MyoSuite Task enum */
enum { MYO_CHOICE_FIXED = 0, MYO_CHOICE_CW = 1, MYO_CHOICE_CCW = 2, MYO_CHOICE_RANDOM = 3 };

#define MYO_N_RWD 8 /* Baoding: pos_dist_1,pos_dist_2,act_reg,alive,sparse,solved,done,dense; die reorient: pos_dist,rot_dist,... */
#define MYO_ROT_CHOICE_MAX 4 /* entries of a goal_rot_x/y/z range list */
#define MYO_OBJG_MAX 20      /* geoms of the per-env object group (the die) */

/* Task configuration = the kwargs of CustomBaodingEnv._setup / CustomBaodingP2Env._setup
 * (/root/reference/src/envs/baoding.py:210-227,300-324) lowered to plain numbers, plus the
 * ids the reference resolves by name (baoding.py:264-269,371-378) and the TimeLimit horizon
 * (/root/reference/src/envs/__init__.py:15,61). */
typedef struct myo_task_cfg {
  int32_t kind;              /* MYO_TASK_* */
  int32_t frame_skip;        /* 10 */
  int32_t max_episode_steps; /* 200 */
  int32_t n_hand;            /* 23: obs hand_pos = qpos[:n_hand] */
  int32_t obj1_sid, obj2_sid, target1_sid, target2_sid;
  int32_t obj1_bid, obj2_bid, obj1_gid, obj2_gid;
  int32_t task_choice;       /* MYO_CHOICE_*: P1 `task` (None->FIXED=CCW,"cw","ccw","random"); P2 `task_choice` */
  int32_t enable_rsi, balls_overlap, limit_init_angle_on, beta_init_angle_on, beta_ball_size_on,
      beta_ball_mass_on;
  double drop_th, proximity_th;
  double center_pos[2];
  double weights[7];         /* pos_dist_1,pos_dist_2,act_reg,alive,sparse,solved,done */
  double goal_time_period[2], goal_xrange[2], goal_yrange[2];
  double rsi_probability, overlap_probability;
  double noise_palm, noise_fingers, noise_balls;
  double limit_init_angle, beta_init_angle[2], beta_ball_size[2], beta_ball_mass[2];
  double obj_size_range[2], obj_mass_range[2], obj_friction_change[3];
  double init_qpos0;         /* -1.57 (baoding.py:283,401); die reorient: -1.5 (reorient.py:121) */
  /* -- kind MYO_TASK_REORIENT: the kwargs of CustomReorientEnv._setup (/root/reference/src/envs/reorient.py:58-122) and what
   * _setup reads from the model.  Ids: obj1_sid = site object_o, target1_sid = site target_o, obj1_bid = body Object,
   * [obj1_gid, obj2_gid) = the die's geoms (body_geomadr .. +body_geomnum); n_hand = nq - 7 (the die's free joint is last).
   * drop_th, obj_friction_change, enable_rsi, frame_skip, max_episode_steps as above.  The other Baoding fields are unused. */
  double ro_weights[9];      /* pos_dist, rot_dist, pos_dist_diff, rot_dist_diff, alive, act_reg, sparse, solved, done */
  double ro_goal_pos[2], ro_goal_rot[2];        /* goal_pos / goal_rot ranges (low, high) */
  int32_t ro_n_rot_choice[3], ro_pad_;          /* lengths of goal_rot_x / _y / _z (0 = None: goal_rot applies) */
  double ro_rot_choice[3][MYO_ROT_CHOICE_MAX][2];
  double ro_obj_size_change, ro_pos_th, ro_rot_th;
  double ro_goal_init_pos[3], ro_goal_obj_offset[3];   /* site_xpos[target_o] and site_xpos[target_o] - site_xpos[object_o] at setup */
  double ro_rsi_distance_pos, ro_rsi_distance_rot;
} myo_task_cfg;

/* -- model ------------------------------------------------------------------------------
 * replaces gym.make(..., model_path=...) -> mj_loadModel (src/envs/__init__.py:17,63).
 * `blob` is the named-array format of include/myo_model_blob.h (host memory). */
int myo_model_from_blob(const void* blob, size_t nbytes, myo_model** out);
/* Load a MuJoCo 2.1 binary model file, what the reference's gym registrations hand to MuJoCo's mj_loadModel
 * (`model_path=…/myo_hand_baoding.mjb`, /root/reference/src/envs/__init__.py:17,63).  The C reader (csrc/myo_mjb.h) decodes the
 * file, applies the feature checks and derives the static tables exactly as the Python route (mjb.load_mjb + model.compile_model
 * + myo_model_from_blob) does.  integrator: -1 keeps the model's, 0 Euler, 1 RK4.  unsupported_contacts: 0 = a model with
 * colliding geom pairs that have no narrow phase here is refused (MYO_E_UNSUPPORTED), 1 = compiled without those pairs; | 2 = a model
 * whose opt.solver is PGS / CG or that asks for noslip iterations is stepped with this stepper's Newton solver (no noslip pass) instead
 * of being refused — an explicit opt-in; every other unsupported feature is refused whatever the flags. */
int myo_model_load_mjb(const char* path, int integrator, int unsupported_contacts, myo_model** out);
void myo_model_destroy(myo_model* m);
int myo_model_size(const myo_model* m, const char* name); /* nq nv nu na nbody ... ; -1 unknown */

/* -- batch ------------------------------------------------------------------------------
 * replaces SubprocVecEnv([thunk]*n) + TimeLimit + Monitor (src/main_baoding.py:56-65).
 * `cfg` may be NULL (kind NONE: physics only).  dtype: MYO_F64 | MYO_MIXED. */
int myo_batch_create(const myo_model* m, const myo_task_cfg* cfg, int n_envs, int device,
                     uint64_t seed, int dtype, myo_batch** out);
void myo_batch_destroy(myo_batch* b);
int myo_batch_num_envs(const myo_batch* b);
int myo_batch_obs_dim(const myo_batch* b);
int myo_batch_lds_bytes(const myo_batch* b); /* LDS footprint of one env's working set */

/* env.reset() for the envs selected by mask (dev uint8[N] or NULL = all); writes obs
 * (dev float[N,obs_dim], may be NULL).  baoding.py:146-208 / :494-647. */
int myo_batch_reset(myo_batch* b, const uint8_t* mask, float* obs, void* stream);

/* env.step(a) for every env with SubprocVecEnv auto-reset semantics: BaodingEnvV1.step ->
 * BaseV0.step -> mj_step x frame_skip -> get_obs/get_reward_dict (baoding.py:24-94,403-467),
 * TimeLimit truncation, reset on done.  act: dev float[N,nu].  Outputs (dev): obs[N,obs_dim]
 * (post-reset obs where done), rew[N], done[N], trunc[N] (info["TimeLimit.truncated"]),
 * term_obs[N,obs_dim] (info["terminal_observation"], valid where done), comps[N,MYO_N_RWD]
 * (rwd_dict of the step), ep_info[N,2] (Monitor's episode r,l; valid where done).
 * Any output pointer except obs/rew/done may be NULL. */
int myo_batch_step(myo_batch* b, const float* act, float* obs, float* rew, uint8_t* done,
                   uint8_t* trunc, float* term_obs, float* comps, float* ep_info, void* stream);

/* Test hook: set the generation counter of the step plan (k_step's part protocol; it wraps after 2^28 steps). */
int myo_batch_set_step_generation(myo_batch* b, unsigned int gen);

/* Health counters of a batch (all 0 in a healthy one; synchronises the device).
 * out[0]: k_step workgroups that found their env's hand-off state in another generation than the launch's (a failed launch,
 *         or one batch stepped from two unsynchronised streams; such a step writes nothing and flags the env in bad_state);
 * out[1]: substeps in which an env had more contacts than its scratch holds (24; 22 in the fp64 stepper; 48 for models with
 *         extended collision pairs or a die) — the surplus was dropped, as MuJoCo drops contacts beyond nconmax with a warning;
 * out[2]: substeps in which an env had more joint-limit / tendon-limit / friction-loss rows than the scratch's row capacity
 *         (MYO_NLIM_MAX = 56) — the surplus was dropped;
 * out[3]: the largest number of contact slots any substep counted in out[1] asked for (what capacity would have been enough). */
int myo_batch_health(myo_batch* b, int out[4]);

/* Device check of the premise of k_step's per-wave-slot workspace (csrc/wave.h: myo_wave_slot): launches n_workgroups one-wave
 * workgroups with lds_bytes of dynamic LDS that each occupy their hardware wave slot's counter for ~20 us.
 * out[0]: workgroups that found their slot occupied by another resident workgroup (0 on a device whose HW_ID layout is the expected one);
 * out[1]: distinct slots seen; out[2]: largest slot index; out[3]: bit mask of the XCC ids seen. */
int myo_debug_wave_slots(int device, int n_workgroups, int lds_bytes, int32_t out[4]);

/* env.step(a) of the UNWRAPPED env for the envs selected by mask (dev uint8[N], NULL = all): no
 * TimeLimit / Monitor accounting, no auto-reset.  This is the `self.step(action)` that
 * MixtureModelBaodingEnv.reset runs with its base policy for the first n_steps_base_model steps
 * (baoding.py:700-711).  obs rows of unselected envs are left untouched; done (may be NULL) receives
 * the env's own done flag (ball dropped) for the selected envs. */
int myo_batch_step_inner(myo_batch* b, const uint8_t* mask, const float* act, float* obs, uint8_t* done, void* stream);

/* The same step for a LIST of envs in compact form: idx = dev int32[n_idx] (env numbers; -1 = empty slot; no env twice), act = dev
 * float[n_idx, nu], obs = dev float[n_idx, obs_dim], done = dev uint8[n_idx] or NULL — row r belongs to env idx[r].  One workgroup
 * per slot instead of one per env of the batch: the base phase of MixtureModelBaodingEnv (baoding.py:700-711) touches the few
 * envs that were just reset. */
int myo_batch_step_inner_idx(myo_batch* b, const int* idx, int n_idx, const float* act, float* obs, uint8_t* done, void* stream);

/* Whole env records from one batch into another of the same model / task kind / device: dst env dst_idx[r] <- src env src_idx[r],
 * r < k (dev int32 arrays; out-of-range entries are skipped).  A record is everything an env is between two steps, so the
 * destination env continues where the source env stood (MixtureModelBaodingEnv: pre-played episodes from a pool batch). */
int myo_batch_copy_envs(myo_batch* dst, const int* dst_idx, const myo_batch* src, const int* src_idx, int k, void* stream);

/* raw physics: apply ctrl (dev double[N,nu]) and run `nsub` mj_step substeps; no task layer.
 * Used by parity tests on arbitrary models. */
int myo_batch_physics_step(myo_batch* b, const double* ctrl, int nsub, void* stream);

/* state access (sim.data.qpos/qvel/act/time; env.set_state; robot.reset).  dev double
 * arrays [N,nq],[N,nv],[N,na],[N]; any may be NULL. */
int myo_batch_get_state(myo_batch* b, double* qpos, double* qvel, double* act, double* time,
                        void* stream);
int myo_batch_set_state(myo_batch* b, const double* qpos, const double* qvel, const double* act,
                        const double* time, void* stream);
/* Numerical blow-up of an env (non-finite / huge qpos, qvel or qacc: MuJoCo's mj_checkPos/Vel/Acc, which warn and
 * reset the data) is NOT an error of myo_batch_step: the env ends its episode (done = 1, trunc = 0, reward 0, reward
 * components 0 except `done`), is reset at once, and its terminal observation is the reset observation, so no NaN
 * leaves the kernel.  Register a caller-owned device buffer uint8[N] here to be told which envs it happened to:
 * every myo_batch_step writes 0 / 1 per env.  NULL (default) = not reported. */
int myo_batch_set_bad_state_buffer(myo_batch* b, uint8_t* bad_state);

/* qacc_warmstart (dev double[N,nv]): the remaining piece of MuJoCo's integration state (mjData.qacc_warmstart
 * seeds the Newton solver, so two steppers only retrace each other when it is copied along with qpos/qvel/act).
 * Either pointer may be NULL. */
int myo_batch_warmstart(myo_batch* b, double* get_qacc_warmstart, const double* set_qacc_warmstart, void* stream);

/* per-env task parameters, explicit injection for parity tests (what reset() samples):
 * task_i  dev int32[N,2]  = which_task, counter
 * task_d  dev double[N,9] = start_angle1, start_angle2, x_radius, y_radius, time_period,
 *                           target1_x, target1_y, target2_x, target2_y (palm-frame site xy)
 * ball_d  dev double[N,10]= mass1, mass2, friction1[3], friction2[3], size1, size2
 * MYO_TASK_REORIENT batches: task_i = 0, episode step counter; task_d = goal_pos[3], goal_quat[4], pos_dist, rot_dist
 * (the shaping state of reorient.py:207-210); ball_d[8] = the die's size delta, the rest unused. */
int myo_batch_set_task(myo_batch* b, const int32_t* task_i, const double* task_d,
                       const double* ball_d, void* stream);
int myo_batch_get_task(myo_batch* b, int32_t* task_i, double* task_d, double* ball_d, void* stream);

/* Per-env physical randomisation of an object made of several geoms (the die of CustomReorientEnv.reset,
 * /root/reference/src/envs/reorient.py:136-147): each of the geoms [gid0, gidn) (at most MYO_OBJG_MAX) has its own per-env
 * friction triple (myo_batch_object_friction; the model's values until set) and the group a per-env size delta in
 * ball_d[8] (size1): every geom centre of the group moves outward by the delta along each non-zero local
 * coordinate, capsule half-lengths grow by it.  (-1, -1) clears the group.  A MYO_TASK_REORIENT batch owns its
 * group (the die, from the task cfg) and draws both at every reset; this call is for physics-only batches. */
int myo_batch_set_object_group(myo_batch* b, int gid0, int gidn);
/* friction of the object group's geoms: dev double[N, gidn-gid0, 3]; either pointer may be NULL (set, then get).  The stepper
 * builds condim-3 contacts only, which read the sliding coefficient [.,.,0]; a MYO_TASK_REORIENT reset draws all three
 * per geom in the reference's order and keeps the sliding one (the other two slots stay at the model's values). */
int myo_batch_object_friction(myo_batch* b, const double* set_fric, double* get_fric, void* stream);

/* The model / task parameters of ONE batch per device sit in __constant__ memory; every launching entry point
 * re-uploads them (on its stream) when another batch used the device in between.  A caller that REPLAYS a
 * captured graph containing this batch's launches must call this first whenever other batches of the same
 * process may have launched since (no-op if this batch is still the bound one). */
int myo_batch_bind_constants(myo_batch* b, void* stream);

/* Order of the tendon stage's geom wraps (sphere / cylinder wraps; the solver runs 64 of them per pass and a pass none of whose wraps
 * engages skips its tangent solve): counts which wraps engage in the envs' PRESENT states and sorts the batch's wrap order by it, so
 * that the wraps that rarely engage share the last pass.  Runs by itself: after a myo_batch_reset of all envs (mask = NULL), 16
 * myo_batch_step calls later, then every 256 calls (not while the stream is being captured); a caller may also run it between any
 * two steps — two small kernels on the stream, no host synchronisation, safe beside captured graphs of steps (the tables are
 * rewritten in place).  No result bit depends on the order.  No-op for models with at
 * most 64 geom wraps, and with MYO_NO_WRAP_ORDER set in the environment when the batch is made (A/B switch). */
int myo_batch_tune_wrap_order(myo_batch* b, void* stream);

/* forward dynamics of the current state with intermediates dumped for stage-wise parity
 * tests: out is dev double[N, myo_batch_dump_size()] ; layout by myo_batch_dump_offset(name).
 * names: ten_length ten_J(nt*nv) M(nv*nv) qfrc_bias qfrc_passive qfrc_actuator qacc_smooth
 *        qacc actuator_force act_dot counts(ncon,nefc,iter) efc_aref efc_D site_xpos subtree_com */
int myo_batch_forward_dump(myo_batch* b, const double* ctrl, double* out, void* stream);
int myo_batch_dump_size(const myo_batch* b);
int myo_batch_dump_offset(const myo_batch* b, const char* name);

/* average duration (ms) of the step kernel over the launches since the last call, measured
 * with HIP events on the launch stream; resets the accumulator.  Returns <0 if no launch. */
double myo_batch_kernel_ms(myo_batch* b);
int myo_batch_enable_timing(myo_batch* b, int on);

/* Elementwise part of one PPO minibatch step in a single launch (SB3 PPO.train loss terms; the
 * reference runs them as separate torch ops inside sb3_contrib.RecurrentPPO.train,
 * /root/reference/src/train/trainer.py:66-71).  All pointers dev float32: mean[B,A] values[B]
 * actions[B,A] old_logp[B] adv[B] (raw) returns[B] log_std[A] adv_stats[2]={mean,std of adv}.
 * Outputs: dmean[B,A], dvalue[B] = d loss/d(policy mean, value); acc[2A+3] = {sum_i dlogp_i (z^2-1)
 * per action dim (entropy term not included) [A], policy loss, value loss, sum_i dmean[i,:] [A],
 * sum_i dvalue[i]} (the last two are the bias gradients of the action / value heads).
 * dmean_bf16 / dvalue_bf16: optional (NULL) bfloat16 copies of dmean / dvalue for the backward GEMMs.
 * work: dev float32 [ceil(B/64) * (2A+3)] scratch; the sums are formed from per-block partials in
 * block order by a second small launch (deterministic, no float atomics).
 * in_bf16 != 0: `mean` and `values` point to bfloat16 data (the GEMM outputs) instead of float32.
 * Optional direct gradient outputs (NULL to skip): g_log_std[A] = acc[0..A) - ent_coef,
 * g_bias_pi[A] = acc[A+2..2A+2), g_bias_vf[1] = acc[2A+2]. */
int myo_ppo_loss_grad(const float* mean, const float* values, const float* actions, const float* old_logp,
                      const float* adv, const float* returns, const float* log_std, const float* adv_stats,
                      int B, int A, float clip, float vf_coef, float* dmean, float* dvalue, float* acc,
                      uint16_t* dmean_bf16, uint16_t* dvalue_bf16, float* work, int in_bf16, float ent_coef,
                      float* g_log_std, float* g_bias_pi, float* g_bias_vf, void* stream);

/* Minibatch gather of one optimiser step (SB3 RolloutBuffer.get + the per-minibatch advantage
 * normalisation statistics of PPO.train): rows idx[0..bs) (dev int64) of obs[N,obs_dim] act[N,act_dim]
 * oldlp[N] adv[N] ret[N] (dev float32) -> obs_bf16 [copies, bs, obs_dim] (bfloat16), act_mb, oldlp_mb,
 * adv_mb, ret_mb, adv_stats[2] = {mean, unbiased std} of adv_mb.  work: dev float32
 * [2*ceil(bs/16)] scratch (block moments, merged in block order by a second small launch). */
int myo_ppo_gather(const float* obs, const float* act, const float* oldlp, const float* adv, const float* ret,
                   const int64_t* idx, int bs, int obs_dim, int act_dim, uint16_t* obs_bf16, int copies,
                   float* act_mb, float* oldlp_mb, float* adv_mb, float* ret_mb, float* adv_stats /* NULL: not computed */, float* work,
                   void* stream);

/* h <- max(h + bias, 0) in place: bfloat16 h[groups, rows, cols], bias[groups, cols] (cols even). */
int myo_bias_relu_bf16(uint16_t* h, const uint16_t* bias, int groups, int rows, int cols, void* stream);

/* Finishes a split-K product: out[g, j] = sum_k part[g, k, j]; part dev [groups, splits, n] of
 * bfloat16 (part_is_bf16 != 0) or float32, n even; out dev float32 [groups, n]. */
int myo_splitk_reduce(const void* part, int part_is_bf16, float* out, int groups, int splits, int n, void* stream);
/* Two such reductions in one launch (a layer's weight-gradient partials and its bias partials; the two head
 * gradients): same arguments and the same arithmetic as two myo_splitk_reduce calls. */
int myo_splitk_reduce2(const void* part_a, int a_is_bf16, float* out_a, int groups_a, int splits_a, int n_a,
                       const void* part_b, int b_is_bf16, float* out_b, int groups_b, int splits_b, int n_b, void* stream);

/* ReLU backward in place on bfloat16 dy[rows, cols] (dy *= act > 0) plus column sums of the result
 * over blocks of 32 rows: partial dev float32 [rows/32, cols] (finish with myo_splitk_reduce: the
 * bias gradient).  rows % 32 == 0, cols/2 divides 256. */
int myo_relu_bwd_colsum_bf16(uint16_t* dy, const uint16_t* act, int rows, int cols, float* partial, void* stream);

/* ---- per-step rollout plumbing (between two myo_batch_step launches) ------------------------------
 * `t_idx`: dev int32 = rollout-buffer row of the current step; `draw_counter`: dev uint64[2] (Philox
 * stream position, [1] is the pending value).  Buffers are dev float32, row-major [T, N, ...].
 *
 * policy input: obs [N,O] -> obs_buf[t] (may be NULL) and `copies` stacked bfloat16 copies. */
int myo_rollout_policy_input(const float* obs, int N, int O, float* obs_buf, uint16_t* x_bf16, int copies,
                             const int32_t* t_idx, void* stream);
/* SB3 DiagGaussianDistribution sample + log_prob (RecurrentPPO.collect_rollouts ->
 * policy.forward, /root/reference/src/train/trainer.py:66-71): a = mean + exp(log_std) * eps, eps from
 * Philox4x32-10(seed, draw_counter[0]); writes act_buf[t], val_buf[t] (from value_bf16), logp_buf[t]
 * and clipped = clip(a, -1, 1) [N,A] (the env input). */
int myo_rollout_sample(const uint16_t* mean_bf16, const uint16_t* value_bf16, const float* log_std, int N, int A,
                       uint64_t seed, uint64_t* draw_counter, const int32_t* t_idx, float* act_buf, float* val_buf,
                       float* logp_buf, float* clipped, int deterministic, void* stream);
/* stable_baselines3 VecNormalize.step_wait for one batched step (VecNormalize.load / .normalize_obs:
 * /root/reference/src/main_baoding.py:75, src/metrics/custom_callbacks.py:34): running mean/var of obs
 * and of the discounted return (Chan merge, fp64, in place: obs_mean[O] obs_var[O] obs_count[1],
 * ret_stats[3] = mean,var,count, returns[N]), normalised + clipped obs -> nobs [N,O], reward ->
 * rew_buf[t], terminal obs -> term_buf[t] (may be NULL), trunc -> trunc_buf[t] (may be NULL),
 * start_buf[t] <- starts, starts <- done.  work: dev double [ceil(N/128) * 2 * (O+1)]. */
int myo_vecnorm_step(const float* obs, const float* rew, const uint8_t* done, const uint8_t* trunc, const float* term_obs,
                     int N, int O, double* obs_mean, double* obs_var, double* obs_count, double* ret_stats,
                     double* returns, double gamma, double eps, double clip_obs, double clip_rew, int training,
                     int norm_obs, int norm_reward, float* nobs, float* starts, const int32_t* t_idx, float* rew_buf,
                     float* start_buf, float* term_buf, float* trunc_buf, double* work, void* stream);
/* t_idx <- (t_idx + 1) mod T, commit the Philox position. */
/* gSDE action sampling (stable-baselines3 StateDependentNoiseDistribution as the reference's RecurrentPPO(use_sde=True,
 * sde_sample_freq=-1) uses it, /root/reference/docs/summary.md:100): mean f32[N,A], latent f32[N,L] (latent_pi), exploration_mat
 * f32[N,L,A] (one matrix per env, drawn once per rollout), log_std f32[L,A] -> actions (unclipped), clipped, logp f32[N]. */
int myo_rollout_sample_sde(const float* mean, const float* latent, const float* exploration_mat, const float* log_std, int N, int L,
                           int A, float* actions, float* clipped, float* logp, int deterministic, void* stream);

/* myo_vecnorm_step split where N ranks exchange their batch moments (one VecNormalize over the envs of ALL ranks,
 * /root/reference/src/main_baoding.py:75): batch_moments leaves batch[0] = n, batch[1..O+1] = sum x (column O = the
 * discounted returns), batch[O+2..2O+2] = sum x^2 of this rank's step in a caller-owned dev double[2 O + 3]; the
 * caller all-reduces it (SUM); finish then updates the running statistics from it and normalises as myo_vecnorm_step. */
int myo_vecnorm_batch_moments(const float* obs, const float* rew, int N, int O, double* returns, double gamma, int training,
                              double* work, double* batch, void* stream);
int myo_vecnorm_finish(const float* obs, const float* rew, const uint8_t* done, const uint8_t* trunc, const float* term_obs,
                       int N, int O, double* obs_mean, double* obs_var, double* obs_count, double* ret_stats, double* returns,
                       double eps, double clip_obs, double clip_rew, int training, int norm_obs, int norm_reward, float* nobs,
                       float* starts, const int32_t* t_idx, float* rew_buf, float* start_buf, float* term_buf, float* trunc_buf,
                       const double* batch, void* stream);
int myo_rollout_advance(int32_t* t_idx, int T, uint64_t* draw_counter, void* stream);

/* One time step of G stacked one-layer LSTMs (gate order i, f, g, o) around the recurrent GEMM: the pointwise
 * part of RecurrentActorCriticPolicy's lstm_actor / lstm_critic in collect_rollouts / train
 * (/root/reference/src/train/trainer.py:49-71; sb3-contrib _process_sequence).  Rows r = g * N + n, R = G * N.
 * Storage float32 (is_bf16 = 0) or bfloat16 (1); arithmetic fp32.  keep_next: dev float[N], 0 where the NEXT
 * step starts an episode (NULL = all ones).
 * fwd: gx, gh [R,4H] pre-activations (input and recurrent parts), c_prev [R,H] (already masked) ->
 *      out_h, c_new [R,H] (unmasked), hm_next, cm_next [R,H] (times keep_next: inputs of the next step),
 *      ws [R,4H] (activated gates, for the backward pass).
 * bwd: dout [R,H] (gradient of out_h, may be NULL), dhm_next / dcm_next [R,H] (gradients of hm_next / cm_next
 *      from the later step, NULL at the last step) -> dgates [R,4H] (gradient of gx and of gh), dc_prev [R,H]. */
int myo_lstm_cell_fwd(const void* gx, const void* gh, const void* c_prev, const float* keep_next, int R, int N, int H,
                      int is_bf16, void* out_h, void* hm_next, void* cm_next, void* c_new, void* ws, void* stream);
int myo_lstm_cell_bwd(const void* dout, const void* dhm_next, const void* dcm_next, const float* keep_next,
                      const void* c_prev, const void* c_new, const void* ws, int R, int N, int H, int is_bf16,
                      void* dgates, void* dc_prev, void* stream);

/* The same time step with the recurrent product inside (one launch instead of GEMM + cell kernel): bfloat16 storage, fp32
 * accumulation on the matrix cores; H in {32, 64, 128, 256} (myo_lstm_step_supported).  Arrays are [G, N, .] contiguous except
 *   gx    element (g, n, col) at gx[g*gx_sg + n*gx_sr + col]  (col < 4H; e.g. one [N, G*4H] GEMM output: gx_sg = 4H, gx_sr = G*4H),
 *   out_h element (g, n, u) at out_h[g*out_sg + n*H + u],  dout likewise with dout_sg  (strides in elements, multiples of 4).
 * fwd: h_prev, c_prev [G,N,H] (already masked), w_hh [G,4H,H] (weight_hh_l0 of each LSTM) -> out_h, hm_next, cm_next, c_new, ws
 *      as myo_lstm_cell_fwd; c_new / ws may be NULL (rollout).  The CELL state may travel in float32 beside (or instead of) its
 *      bfloat16 arrays: c_prev32 float32 [G,N,H] (non-NULL: read instead of c_prev), cm_next32 float32 [G,N,H] (non-NULL: written,
 *      unrounded; cm_next may then be NULL) — what an LSTM carries through an episode is c (stock nn.LSTM state is float32,
 *      /root/reference/src/main_reorient.py:53-71); h is the matrix cores' operand and stays bfloat16.
 * bwd: dgates_next [G,N,4H] = the dgates of step t+1 (NULL at the last step), w_hh_t [G,H,4H] = W_hh transposed; the product
 *      dgates_next . W_hh replaces dhm_next of myo_lstm_cell_bwd -> dgates [G,N,4H], dc_prev [G,N,H]. */
int myo_lstm_step_supported(int H);
int myo_lstm_step_fwd(const void* gx, long long gx_sg, long long gx_sr, const void* h_prev, const void* c_prev, const void* w_hh,
                      const float* keep_next, int G, int N, int H, void* out_h, long long out_sg, void* hm_next, void* cm_next,
                      void* c_new, void* ws, const float* c_prev32, float* cm_next32, void* stream);
int myo_lstm_step_bwd(const void* dout, long long dout_sg, const void* dgates_next, const void* dcm_next, const void* w_hh_t,
                      const float* keep_next, const void* c_prev, const void* c_new, const void* ws, int G, int N, int H,
                      void* dgates, void* dc_prev, void* stream);

/* ALL T time steps of the same G stacked LSTMs over N sequences in ONE launch per direction (csrc/myo_lstm_seq.h): what the recurrent
 * PPO minibatch step runs instead of T myo_lstm_step_fwd + T myo_lstm_step_bwd launches (RecurrentPPO.train,
 * /root/reference/src/train/trainer.py:49-71; sb3-contrib _process_sequence).  H in {128, 256} (myo_lstm_seq_supported), N a
 * multiple of 16 (MYO_E_ARG otherwise: the caller keeps the step kernels).  Same roundings as the step kernels.
 * row_split RS in {1, 2} (and 4 at H = 256): a workgroup owns 16 / RS sequences — more, smaller workgroups for small minibatches.
 * The recurrent weights are passed FRAGMENT-MAJOR (one contiguous KB per wave load; permuted from W_hh [G,4H,H] by the caller,
 * once per minibatch step — rl/fused_lstm.py lstm_seq_weights(W_hh, RS)).  With UT = H/128 tiles of 16 units per wave, eight waves,
 * CL = 4*UT/RS, b = i%4 and unit(w, ut, i) = 16*UT*w + 4*UT*(i/4) + CL*(b/(4/RS)) + (4/RS)*ut + b%(4/RS):
 *   w_frag  bf16 [G][8][H/32][4][UT][64][8]:  w_frag[g][w][kk][q][ut][l][j] = W_hh[g][q*H + unit(w, ut, l&15)][32*kk + 8*(l>>4) + j]
 *   wt_frag bf16 [G][8][4H/32][UT][64][8]:   wt_frag[g][w][kk][ut][l][j]   = W_hh[g][32*kk + 8*(l>>4) + j][unit(w, ut, l&15)]
 * The arrays only these two calls exchange — c_new, ws, and cm from slot 1 on — are TILE-MAJOR (same sizes as the row-major
 * shapes below; [(g, r/(16/RS))][wave 8][gate][lane 64][CL], csrc/myo_lstm_seq.h; rl/fused_lstm.py lstm_seq_rows converts).
 * fwd: gx element (t, g, r, col) at gx[t*gx_st + g*gx_sg + r*gx_sr + col]; hm / cm bf16 [T+1,G,N,H]: slot 0 = the masked state
 *      entering step 0 (given, row-major), slots 1..T written (hm row-major, cm tile-major); keep float32 [T,N] (0 where an episode
 *      starts at that step: step t masks its outgoing state with keep[t+1], the last step with 1); out_h element (g, t, r, u) at
 *      out_h[g*out_sg + t*out_st + r*H + u]; c_new [T,G,N,H], ws [T,G,N,4H] (gate activations) kept for the backward pass.
 *      c0_32 float32 [G,N,H] (may be NULL): the masked CELL state entering step 0, unrounded; with it the cell state is carried from
 *      step to step in float32 (the bfloat16 cm slots are still written: the backward pass reads them), without it every step
 *      re-reads its cell state from the bfloat16 slot the previous step wrote (the roundings of T myo_lstm_step_fwd calls on bf16 c).
 * bwd: dout laid out like out_h; keep / cm / c_new / ws as the forward pass left them -> dgates [T,G,N,4H] (row-major). */
int myo_lstm_seq_supported(int H);
int myo_lstm_seq_fwd(const void* gx, long long gx_st, long long gx_sg, long long gx_sr, void* hm, void* cm, const void* w_frag,
                     const float* keep, int G, int N, int H, int T, int row_split, void* out_h, long long out_sg, long long out_st,
                     void* c_new, void* ws, const float* c0_32, void* stream);
int myo_lstm_seq_bwd(const void* dout, long long dout_sg, long long dout_st, const void* wt_frag, const float* keep, const void* cm,
                     const void* c_new, const void* ws, int G, int N, int H, int T, int row_split, void* dgates, void* stream);

/* GAE(gamma, lambda) backward scan = SB3 RolloutBuffer.compute_returns_and_advantage (run by
 * RecurrentPPO.learn, /root/reference/src/train/trainer.py:66-71).  dev float32 [T,N] row-major:
 * rew, val, starts (episode_starts), outputs adv, ret; last_val[N], last_done[N]. */
int myo_gae(const float* rew, const float* val, const float* starts, const float* last_val,
            const float* last_done, int T, int N, float gamma, float lam, float* adv, float* ret,
            void* stream);

/* One PPO minibatch step of the MLP actor-critic (two hidden layers of 256 per net, ReLU) on the matrix cores: what SB3's
 * PPO.train runs per minibatch through autograd (evaluate_actions -> clipped surrogate + vf_coef * MSE - ent_coef * entropy ->
 * backward; /root/reference/src/train/trainer.py:57-71 hands that loop to sb3-contrib).  Gathers rows
 * idx[0..B) of the rollout arrays, runs forward, loss gradient and backward, and leaves d(loss)/d(param) in `grads` (same flat
 * layout as `params`; offsets in ELEMENTS, [0] actor / [1] critic) and acc[2A+3] = {d log_std[A], policy loss, value loss,
 * action-head bias gradient[A], value-head bias gradient}.  compute_adv_stats != 0: adv_stats <- {mean, unbiased std} of the
 * minibatch advantages (SB3's per-minibatch normalisation); 0: the caller supplies them ({0, 1} = no normalisation).
 * workspace: device memory of myo_ppo_mlp_workspace_bytes() bytes, ZERO-FILLED ONCE by the caller and then owned by these
 * calls (weight images, feature-major activations, split-K slabs).  Deterministic: no float atomics.  Returns
 * MYO_E_UNSUPPORTED for other shapes (hidden != 256, obs > 128, act > 48, B not a multiple of 1024). */
typedef struct myo_ppo_mlp_desc {
  const float *obs, *act, *oldlp, *adv, *ret;
  const int64_t* idx;
  int32_t B, O, A, hidden;
  const float* params;
  float* grads;
  int64_t G;
  int64_t off_W1[2], off_b1[2], off_W2[2], off_b2[2], off_Wh[2], off_bh[2], off_log_std;
  float clip, vf_coef, ent_coef;
  int32_t compute_adv_stats;
  float* adv_stats;
  float* acc;
  void* workspace;
  int64_t workspace_bytes;
  /* single-rank tail (both NULL: off): the squares of the finished gradient, in myo_ppo_mlp_sqnorm_parts(act_dim) partial sums
   * -> sqnorm_part, and Adam's step counter advanced (adam_step: the int32[2] of myo_adam_clip_step) — then myo_adam_apply
   * instead of myo_adam_clip_step.  Not for N > 1 ranks: there the gradient changes (all-reduce) before it is clipped. */
  float* sqnorm_part;
  int32_t* adam_step;
} myo_ppo_mlp_desc;
int myo_ppo_mlp_sqnorm_parts(int act_dim);
long long myo_ppo_mlp_workspace_bytes(int B, int obs_dim, int act_dim, int hidden, long long G);
int myo_ppo_mlp_step(const myo_ppo_mlp_desc* d, void* stream);

/* One env step's policy call of the rollout for the same MLP actor-critic (collect_rollouts: ActorCriticPolicy.forward,
 * DiagGaussianDistribution.sample / log_prob, RolloutBuffer.add; /root/reference/src/train/trainer.py:66-71 -> agent.learn) in one
 * launch: rows of the policy input obs f32[N,O] -> both trunks and heads on the matrix cores -> actions = mean + exp(log_std) *
 * N(0,1) (Philox, the counters of myo_rollout_sample), log pi, value -> row *t_idx of act_buf [T,N,A], val_buf, logp_buf [T,N]
 * (+ obs_buf [T,N,O] if not NULL), clipped f32[N,A] for the env.  It replaces myo_rollout_policy_input + the trunk / head GEMMs
 * + myo_rollout_sample.  The bf16 weight images it reads live in `workspace` (myo_ppo_mlp_rollout_workspace_bytes, caller-owned);
 * myo_ppo_mlp_rollout_refresh rebuilds them from `params` and has to run after every change of the parameters (once per
 * PPO update).  N must be a multiple of 32; other limits as myo_ppo_mlp_step. */
typedef struct myo_ppo_mlp_rollout_desc {
  const float* obs;
  int32_t N, O, A, hidden;
  const float* params;
  int64_t off_W1[2], off_b1[2], off_W2[2], off_b2[2], off_Wh[2], off_bh[2], off_log_std;
  uint64_t seed;
  uint64_t* draw_counter;
  const int32_t* t_idx;
  float *obs_buf, *act_buf, *val_buf, *logp_buf, *clipped;
  int32_t deterministic;
  void* workspace;
  int64_t workspace_bytes;
} myo_ppo_mlp_rollout_desc;
long long myo_ppo_mlp_rollout_workspace_bytes(int obs_dim, int act_dim, int hidden);
int myo_ppo_mlp_rollout_refresh(const myo_ppo_mlp_rollout_desc* d, void* stream);
int myo_ppo_mlp_rollout(const myo_ppo_mlp_rollout_desc* d, void* stream);

/* clip_grad_norm_(max_norm) followed by one torch.optim.Adam step over a flat fp32 parameter vector
 * (what RecurrentPPO.train does per minibatch; /root/reference/src/train/trainer.py:66-71, SB3 Adam
 * eps 1e-5).  g is multiplied by grad_scale first (1/world after an all-reduce SUM).  step: dev
 * int32[2], zero-initialised ([1] = number of steps taken); scratch: dev float[64] (per-block sums
 * of g^2, added in a fixed order: deterministic).  p_bf16 (may be NULL): bfloat16 copy of p, rewritten with
 * the new parameters in the same pass (the operand the bf16 GEMMs read). */
int myo_adam_clip_step(float* p, const float* g, float* m, float* v, int n, float lr, float b1, float b2,
                       float eps, float max_norm, float grad_scale, int* step, float* scratch, uint16_t* p_bf16,
                       void* stream);
/* The second half of myo_adam_clip_step alone: |g|^2 = the sum of scratch[0 .. nparts) (left there by myo_ppo_mlp_step with
 * sqnorm_part set, which has also advanced `step`), then clip + Adam as above. */
int myo_adam_apply(float* p, const float* g, float* m, float* v, int n, float lr, float b1, float b2,
                   float eps, float max_norm, float grad_scale, const int* step, const float* scratch, int nparts,
                   uint16_t* p_bf16, void* stream);

const char* myo_last_error(void);
const char* myo_version(void);

#ifdef __cplusplus
}
#endif
#endif
