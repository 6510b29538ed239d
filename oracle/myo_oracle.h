/* myo_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar fp64 CPU restatement of the hot path the reference executes through third-party
 * code: MyoSuite's BaodingEnvV1.step -> MuJoCo 2.1 mj_step x frame_skip
 * (call stack: SURVEY.md §3.3; reference call sites /root/reference/src/envs/baoding.py:179,
 * 183,206,608,625,632 and the inherited step).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.
 *
 * PARITY STATUS
 *   - statics at qpos0 (FK, tendon wrapping, inertia, M^-1, moment arms): PINNED against the
 *     MuJoCo-computed constants embedded in the three .mjb files the reference ships
 *     (tests/test_oracle_statics.py).
 *   - task layer (obs layout, reward, termination, goal schedule): PINNED against golden
 *     vectors produced by importing /root/reference/src/envs/baoding.py (tools/make_golden.py).
 *   - stepping (constraint solver, contacts, integrators): PARITY UNPINNED.  MuJoCo 2.1.x
 *     (free-mujoco-py==2.1.6), MyoSuite==1.2.3 are absent from /root/reference and from this
 *     image, and the reference has no tests or trajectories; the algorithm below restates the
 *     published MuJoCo 2.1 pipeline and is checked by self-consistency
 *     (energy, KKT residual, finite differences) only.
 */
#ifndef MYO_ORACLE_H
#define MYO_ORACLE_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrcModel OrcModel;
typedef struct OrcData OrcData;

OrcModel* orc_model_from_blob(const void* blob, size_t nbytes, char* err, int errlen);
void orc_model_free(OrcModel* m);
int orc_model_int(const OrcModel* m, const char* name); /* nq, nv, nu, ... */

OrcData* orc_data_new(const OrcModel* m);
void orc_data_free(OrcData* d);
void orc_reset(const OrcModel* m, OrcData* d); /* qpos0, zero velocity/activation */

/* named access to per-env arrays (qpos, qvel, act, ctrl, xpos, ten_length, ten_J, M, ...) */
double* orc_ptr(OrcData* d, const char* name);
int orc_count(const OrcData* d, const char* name);
int orc_get_int(const OrcData* d, const char* name); /* ncon, nefc, solver_iter, bad */

void orc_fwd_position(const OrcModel* m, OrcData* d);
void orc_forward(const OrcModel* m, OrcData* d);
void orc_step(const OrcModel* m, OrcData* d);
/* flop counter of the algorithm as restated here: out[orc_flops_stages()] = counts per stage since the
 * last reset, in the order kinematics, com, tendon(+transmission), crb(+factor M), collision, constraint rows,
 * velocity, actuation, acceleration (qacc_smooth), newton, integrate, task.  One counter per process. */
void orc_flops(double* out, int reset);
int orc_flops_stages(void);
/* kinematics only (what MyoSuite's observation sim does before get_obs) */
void orc_kinematics(const OrcModel* m, OrcData* d);

/* ---- Baoding task layer ---------------------------------------- */
typedef struct OrcBaodingCfg {
  int frame_skip;
  int obj1_sid, obj2_sid, target1_sid, target2_sid;
  int obj1_bid, obj2_bid, obj1_gid, obj2_gid;
  int n_hand; /* qpos[:n_hand] = hand_pos */
  double drop_th, proximity_th;
  double center_pos[2];
  /* weights in the order pos_dist_1,pos_dist_2,act_reg,alive,sparse,solved,done */
  double w[7];
} OrcBaodingCfg;

typedef struct OrcBaodingState {
  int which_task; /* 0 hold, 1 cw, 2 ccw */
  int counter;
  double start_angle[2];
  double x_radius, y_radius, time_period;
} OrcBaodingState;

/* one env.step(a): goal placement, sigmoid action map, frame_skip physics steps, obs, reward.
 * comps = pos_dist_1,pos_dist_2,act_reg,alive,sparse,solved,done,dense */
void orc_baoding_step(const OrcModel* m, OrcData* d, const OrcBaodingCfg* cfg,
                      OrcBaodingState* st, const float* action, double* obs, double* comps);
void orc_baoding_obs(const OrcModel* m, OrcData* d, const OrcBaodingCfg* cfg, double* obs);
void orc_baoding_reward(const OrcBaodingCfg* cfg, int na, const double* obs, double* comps);

/* ---- die-reorient task layer (/root/reference/src/envs/reorient.py) ---------- */
typedef struct OrcReorientCfg {
  int frame_skip, n_hand;
  int object_sid, goal_sid, object_bid, gid0, gidn; /* sites object_o / target_o, body Object, the die's geoms [gid0, gidn) */
  double drop_th, pos_th, rot_th;
  double goal_obj_offset[3];
  double w[9]; /* pos_dist, rot_dist, pos_dist_diff, rot_dist_diff, alive, act_reg, sparse, solved, done */
} OrcReorientCfg;
typedef struct OrcReorientState {
  double goal_pos[3], goal_quat[4]; /* body_pos / body_quat of `target` for this episode */
  double pos_dist, rot_dist;        /* self.pos_dist / self.rot_dist */
} OrcReorientState;
void orc_euler2quat(const double* e, double* q);
void orc_reorient_set_die(const OrcModel* m, OrcData* d, const OrcReorientCfg* cfg, const double* friction, double del_size);
void orc_reorient_obs(const OrcModel* m, OrcData* d, const OrcReorientCfg* cfg, const OrcReorientState* st, double* obs);
/* c[10] = the rwd_dict in its own order: pos_dist, rot_dist, pos_dist_diff, rot_dist_diff, alive, act_reg, sparse, solved, done, dense */
void orc_reorient_reward(const OrcReorientCfg* cfg, int na, const double* pos_err, const double* rot_err, const double* act,
                         double prev_pos_dist, double prev_rot_dist, double* c);
void orc_reorient_step(const OrcModel* m, OrcData* d, const OrcReorientCfg* cfg, OrcReorientState* st, const float* action,
                       double* obs, double* comps);
void orc_reorient_reset_dists(const OrcModel* m, OrcData* d, const OrcReorientCfg* cfg, OrcReorientState* st, double* obs);

/* Newton line search: 0 = safeguarded Newton on p'(alpha) (default), 1 = the bracketing structure of MuJoCo 2.1's PrimalSearch
   [3P-RECALL] (myo_oracle.c: primal_search).  Both stop inside the same gradient tolerance; tools/oracle_linesearch.py measures how far
   apart their results are.  Environment: MYO_ORACLE_LS=primal. */
void orc_set_line_search(int primal);

#ifdef __cplusplus
}
#endif
#endif
