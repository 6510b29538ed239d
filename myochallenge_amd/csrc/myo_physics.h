// myo_physics.h — the per-environment forward dynamics + integrators, one env per wavefront.
//
// Restates the MuJoCo 2.1 mj_step pipeline that the reference runs through
// MyoSuite -> mujoco_py (reference call sites
// /root/reference/src/envs/baoding.py:179,183,206,608,625,632) as wave-parallel phases:
//
//   kinematics   lanes = bodies, level by level          (mj_kinematics)
//   com/cdof     lanes = bodies / joints                  (mj_comPos)
//   tendon       lanes = tendons, wrap walk + moment arms (mj_tendon, mju_wrap)
//   crb          lanes = tree-sparse M entries            (mj_crb)
//   collision    lanes = candidate geom pairs             (mj_collision)
//   constraint   lanes = joints / tendons / contacts      (mj_makeConstraint, mj_makeImpedance)
//   velocity     lanes = bodies / dofs                    (mj_comVel, mj_passive, mj_rne)
//   actuation    lanes = muscles                          (mj_fwdActuation, mju_muscle*)
//   solve        packed 35x35 Cholesky in LDS + Newton with exact line search (mj_solNewton)
//   integrate    Euler with implicit joint damping / RK4  (mj_Euler, mj_RungeKutta)
//
// The constraint Jacobian is never materialised: J*v and J'*f go through per-body spatial
// vectors (6 numbers per body), and the Newton Hessian M + J' D J is assembled per contact from
// 3x3 blocks.  Everything an env touches between load and store lives in LDS (Scratch<T, NC>).
#pragma once
#include "myo_model_dev.h"
#include "wave.h"

#define MYO_MINVAL ((T)1e-15)

// Precision plan.  HP ("high precision") is fp64 in EVERY build.  k_step<double> computes everything in fp64.
// k_step<float> is the MIXED stepper: what decides whether a trajectory stays on the reference's (fp64 MuJoCo)
// trajectory is kept in HP, the bulk of the arithmetic runs in fp32:
//   HP : the state (qpos, qvel, act, time) and its integration; the kinematic chain qpos -> body poses
//        (xpos, xquat); contact and joint-limit DISTANCES (penetrations are ~1e-4 m on world coordinates of
//        ~1.5 m, and MuJoCo's contact activation at dist < margin is discontinuous: a distance error of eps
//        flips an activation with probability ~eps / (v h)); goal / target positions and the observation.
//   T  : everything else — the fp32 stages work in coordinates SHIFTED by the env's origin O (the HP world
//        position of the model's first tree root), so their positions are ~0.1 m and differences of positions
//        (moment arms, contact offsets, tendon segments) keep ~1e-8 m.  The dynamics only sees differences
//        of positions, so the shift changes nothing else.
// Measured on the lane-serial build against the fp64 oracle: rounding every fp32-stage OUTPUT of the fp64
// stepper to float moves a 60-env-step contact-rich trajectory by 5e-7; rounding xpos or xmat alone by 3e-3.
typedef double HP;

// Stage timers: compiled only into the diagnostic build (-DMYO_PROF, libmyobatch_prof.so); the
// product build contains no stamp.  Lane 0 accumulates s_memtime deltas per stage in LDS.
#define MYO_NPROF 32
#if defined(MYO_PROF) && !defined(MYO_EMU)
#define PROF(s, k) { if (threadIdx.x == 0) { unsigned long long _t = clock64(); (s).prof[k] += _t - (s).prof_t; (s).prof_t = _t; } }
#else
#define PROF(s, k)
#endif
// Stage instruction counts: the diagnostic build -DMYO_STAGECOUNT (tools/dev/stage_counts.sh) runs ONE stage of a substep twice — the
// stage whose bit is set in c_dbg_repeat (MYO_DBG_REPEAT in the environment when the constants are bound) — and the difference of the
// launch's SQ_INSTS_* counters to a run without a bit is that stage's instruction count.  Only stages that may be repeated (they read what
// they do not write) carry a REP; the tool checks the state checksum of every pass against the plain run.  The product build contains
// none of it.
#if defined(MYO_STAGECOUNT) && !defined(MYO_EMU)
__constant__ int c_dbg_repeat;
#define REP(bit, ...) for (int _r = 0, _n = 1 + ((c_dbg_repeat >> (bit)) & 1); _r < _n; ++_r) { __VA_ARGS__; }
#else
#define REP(bit, ...) { __VA_ARGS__; }
#endif
// Storage of the dense system matrix H: lower triangle, row i padded to the next multiple of four columns
// (rows 4q..4q+3 hold 4(q+1) entries each), so that a lane still fetches / stores its row with 16-byte
// accesses: 720 entries for 36 dofs instead of 1296.  myo_hrow(i) = first entry of row i.
#define MYO_H_SIZE (8 * (MYO_NV_MAX / 4) * (MYO_NV_MAX / 4 + 1))
static_assert(MYO_NV_MAX % 4 == 0, "H rows are stored in groups of four");
#ifdef MYO_EMU
static inline int myo_hrow(int i) { const int q = i >> 2; return ((q * (q + 1)) << 3) + (((i & 3) * (q + 1)) << 2); }
#else
__device__ __forceinline__ constexpr int myo_hrow(int i) { const int q = i >> 2; return ((q * (q + 1)) << 3) + (((i & 3) * (q + 1)) << 2); }
#endif
#define MYO_HIDX(i, j) (myo_hrow(i) + (j)) /* i >= j */

/* doubles per WAVE SLOT of TaskDev::ctrl_ws (whole 128-byte lines): the controls, the tendon moment arms [slot][tendon], the tendon
   lengths, the activation rates and the reward terms (the last three are used by the 48-slot fp64 scratch only: Scratch::SPILL), the warm
   start.  The workspace belongs to the HARDWARE wave slot the workgroup runs in (myo_wave_slot, wave.h), not to the env: nothing in it
   outlives a workgroup (every entry is written before it is read, load_env .. store_env), two workgroups that run at the same time sit in
   different slots by construction, and a slot's lines are only ever touched from ONE XCD — so they stay in that XCD's L2 from launch to
   launch (2048 live slots x 3.9 KB) instead of being dirtied once per (env, part) wherever the part happens to run, and no XCD can hold a
   stale dirty copy of a line another XCD is using (the per-env workspace was written by the parts of a step from different
   XCDs with nothing ordering the write-backs). */
#define MYO_WS_TENJ MYO_NU_MAX
#define MYO_WS_TLEN (MYO_NU_MAX + MYO_NT_MAX * MYO_TJ_MAX)
#define MYO_WS_ADOT (MYO_WS_TLEN + MYO_NT_MAX)
#define MYO_WS_RWD (MYO_WS_ADOT + MYO_NU_MAX)
#define MYO_WS_WARM (MYO_WS_RWD + 8)            /* the solver's warm start while a workgroup holds the env (load_env .. store_env) */
#define MYO_ENVWS_N ((MYO_WS_WARM + MYO_NV_MAX + 15) / 16 * 16)
#define MYO_TASK_REORIENT_K 3   // == MYO_TASK_REORIENT of include/myobatch.h (checked in myobatch.hip)
struct TaskDev {  // device copy of myo_task_cfg (ids = -1 when there is no task layer)
  int kind, frame_skip, max_episode_steps, n_hand;
  int obj1_sid, obj2_sid, target1_sid, target2_sid, obj1_bid, obj2_bid, obj1_gid, obj2_gid;
  int objg_gid0, objg_gidn;   // geom group with per-env, per-geom friction (objg_fric) and a size delta (ball_size[0]); empty = -1,-1
  int task_choice, enable_rsi, balls_overlap, limit_init_angle_on, beta_init_angle_on,
      beta_ball_size_on, beta_ball_mass_on;
  double drop_th, proximity_th, center_pos[2], weights[7];
  double goal_time_period[2], goal_xrange[2], goal_yrange[2];
  double rsi_probability, overlap_probability, noise_palm, noise_fingers, noise_balls;
  double limit_init_angle, beta_init_angle[2], beta_ball_size[2], beta_ball_mass[2];
  double obj_size_range[2], obj_mass_range[2], obj_friction_change[3], init_qpos0;
  // die reorient (kind 3; include/myobatch.h ro_*)
  double ro_weights[9], ro_goal_pos[2], ro_goal_rot[2], ro_rot_choice[3][MYO_ROT_CHOICE_MAX][2];
  double ro_obj_size_change, ro_pos_th, ro_rot_th, ro_goal_init_pos[3], ro_goal_obj_offset[3];
  int ro_n_rot_choice[3], ro_obj_bid;
  void* rk_ws;                // RkScratch<T>[n_envs] in global memory (RK4 models), else null
  int objf_off;               // env record: doubles from the warm start to the object group's friction triples (Scratch::SPILL reads them in place)
  int* slot_map;              // int[MYO_WS_SLOTS]: owner flags of the workspace blocks (myo_ws_acquire / myo_ws_release, wave.h); emulation: null
  char* big_ws;               // char[MYO_WS_SLOTS][MYO_BIGWS_BYTES]: the wave slots' blocks of the 48-slot fp64 scratch (Scratch::SPILL: contact records, wrap results); else null
  double* ctrl_ws;            // double[MYO_WS_SLOTS][MYO_ENVWS_N] in global memory, fp64 stepper: the wave slots' workspaces (one per device, shared by its batches; emulation: one per env)
  int* health;                // int[4] in global memory (myo_batch_health): [0] hand-off states of another generation met by k_step, [1] substeps that dropped contacts beyond the scratch's capacity, [2] substeps that dropped joint / tendon limit or friction-loss rows beyond MYO_NLIM_MAX
  unsigned long long seed;
};

#define MYO_LIM_UPPER ((short)0x8000)
#define MYO_LIM_FRIC ((short)0x4000)            /* a friction-loss row (J = +dof / +tendon moment arms, force clamped to +- frictionloss) */
DEV int lim_index(int id) { return id & 0x3fff; }
DEV bool lim_is_fric(int id) { return (id & 0x4000) != 0; }
// cost and force of a friction-loss row at jar = x (mj_constraintUpdate): quadratic inside |x| < R f, linear outside
template <typename T> DEV T fric_cost(T D, T f, T x, T* force, int* quad) {
  const T Rf = f / D;
  if (x <= -Rf) { *force = f; *quad = 0; return f * (-(T)0.5 * Rf - x); }
  if (x >= Rf) { *force = -f; *quad = 0; return f * (-(T)0.5 * Rf + x); }
  *force = -D * x; *quad = 1; return (T)0.5 * D * x * x;
}
template <typename T> DEV T lim_sign(int id) { return id < 0 ? (T)-1 : (T)1; }

template <typename T>
struct alignas(8) ContactRec {     // 80 B (fp32) / 136 B (fp64).  The STRIDE is chosen for the LDS banks: lanes = constraint rows read one field of 8 (fp64) or
  // 16 (fp32) consecutive records in one access, and a 128-byte stride (32 banks) would put every second record on the same banks — a 4-way
  // conflict on each of those reads; 34 / 20 banks apart they are all distinct (the padding words below buy that)
  // nrm: the contact normal.  The frame's first tangent is not stored: it is (e - nrm[a] nrm) |tinv|, e = the y axis (tinv > 0, a = 1) or
  // the z axis (tinv < 0, a = 2) — make_frame's construction with its reciprocal norm kept (con_frame rebuilds all six numbers in 6 flops);
  // the second tangent is their cross product (con_t2).  muA / muB: friction of the slot's two row pairs.  The rows' reference-acceleration
  // coefficients B and K imp (dist - margin) are not here either: the emitting lane leaves them in the rows' efc_force / efc_jar entries
  // (S_ROW_B / S_ROW_KIP), where efc_reference consumes them.
  T nrm[3], tinv, muA, muB, D;
  T r1[3], r2[3];                 // contact point relative to the reference point of body1's / body2's tree
  int pk;                         // body 1 | body 2 << 8 | support size << 16 | slot kind << 24 (con_b1, con_b2, con_nsup, con_kind)
  // the dofs either body can move, ascending; bit 6 / bit 7 of an entry: the dof is an ancestor dof of body 1 / body 2 (con_sup_*).
  // (The bodies' 64-bit ancestor masks themselves are not kept here: J' f, whose lanes are dofs, takes them from the model's
  // body_dofmask through the scalar cache — the contact index is wave-uniform there.)
  alignas(4) unsigned char sup[MYO_CS_MAX];
  int pad_[sizeof(T) == 8 ? 3 : 2];
};
static_assert(sizeof(ContactRec<double>) == 136 && sizeof(ContactRec<float>) == 80, "contact record strides (LDS banks)");
static_assert(MYO_NV_MAX <= 64, "ContactRec::sup entries: dof index in bits 0-5");
// the contact frame (normal, first tangent) of a record, as make_frame built it
template <typename C, typename T> DEV void con_frame(const C& c, T* f) {      // (C: a ContactRec in LDS or — Scratch::SPILL — in global memory)
  f[0] = c.nrm[0]; f[1] = c.nrm[1]; f[2] = c.nrm[2];
  const bool z = c.tinv < 0;
  const T i = z ? -c.tinv : c.tinv, t = z ? f[2] : f[1];
  f[3] = ((T)0 - t * f[0]) * i; f[4] = ((z ? (T)0 : (T)1) - t * f[1]) * i; f[5] = ((z ? (T)1 : (T)0) - t * f[2]) * i;
}
DEV int con_sup_dof(int e) { return e & 63; }
DEV int con_sup_on1(int e) { return (e >> 6) & 1; }
DEV int con_sup_on2(int e) { return (e >> 7) & 1; }
// A contact of condim d is stored as one, two or three SLOTS of four constraint rows each (mj_makeConstraint's pyramid rows
// J_normal +- friction[k] J_k, k < d - 1, in MuJoCo's order); every slot is a full record (same point, frame, D), its KIND says
// what the two row pairs are:
//   0  translation along tangent 1 / tangent 2                      (condim 3, and the first slot of condim 4 / 6)
//   1  rotation about the normal (torsional) / about tangent 1       (second slot of condim 6)
//   2  rotation about tangent 2 (rolling) / --                       (third slot of condim 6)
//   3  the normal row alone (muA = 0) / --, --, --                   (condim 1: frictionless)
//   4  rotation about the normal / --                                (second slot of condim 4)
// "--" rows are padding: J = 0 and aref = -1, so they are never active and carry no force.
template <typename C> DEV int con_kind(const C& c) { return c.pk >> 24; }
template <typename C> DEV int con_nsup(const C& c) { return (c.pk >> 16) & 255; }
template <typename C> DEV int con_b1(const C& c) { return c.pk & 255; }
template <typename C> DEV int con_b2(const C& c) { return (c.pk >> 8) & 255; }
static_assert(MYO_NB_MAX <= 256 && MYO_CS_MAX <= 255, "ContactRec::pk fields are 8 bits");
DEV bool con_pad(int kind, int e) { return (kind == 2 || kind == 4) ? e >= 2 : (kind == 3 ? e >= 1 : false); }
DEV int con_rows(int kind) { return kind == 3 ? 1 : ((kind == 2 || kind == 4) ? 2 : 4); }

template <typename T>
struct RkScratch {                // RK4 stage storage: the start state and the running weighted sum of the stage derivatives (the
  HP x0[MYO_NQ_MAX + MYO_NV_MAX + MYO_NU_MAX];      // next stage only needs the latest derivative, which goes straight into S_RKDX).
  T Fsum[2 * MYO_NV_MAX + MYO_NU_MAX];              // 1,360 B in the mixed stepper: behind the scratch in LDS where eight workgroups
};                                                  // per CU still fit (Scratch<float, 24>), else one per workgroup in GLOBAL memory

// World poses of the bodies (HP), the origin O of the fp32 stages' coordinates, the tree reference points, and the fp32 copies the
// mixed stepper's stages read.  Mixed stepper: members of their own.  fp64 stepper: no copies, and the poses themselves are
// POSITION-STAGE data (kinematics .. collision; the task layer reads them after a kinematics pass of its own), so they live in
// the solver's seven dof vectors, which are first written after the collision stage — accessors S_XPOS / S_XQUAT / S_ORIGIN /
// S_COM below (1.9 KB of the fp64 scratch: with the 16-slot contact capacity it fits seven workgroups per CU).
template <typename T> struct ScratchPoses {
  HP xpos[MYO_NB_MAX * 3], xquat[MYO_NB_MAX * 4];     // world poses of the bodies
  HP origin[3];                                        // O: the fp32 stages' coordinates are world - O
  T com[MYO_NB_MAX * 3];
  T qvelT_[MYO_NV_MAX];                                // qvel as the fp32 stages read it (S_QVELT)
  T xposT_[MYO_NB_MAX * 3];                            // xpos - O as the fp32 stages read it (S_XPOST)
  T xmat_[MYO_NB_MAX * 9];                             // rotation matrices as the fp32 stages read them; the fp64 stepper derives them from xquat (body_rot)
  T ctrl_[MYO_NU_MAX], qacc_warm_[MYO_NV_MAX];         // controls, warm start (ctrl_get / warm_get below)
  T ten_J[MYO_NT_MAX * MYO_TJ_MAX];                    // tendon moment arms, [tendon][slot] (tenj_get below)
  int pub;                                             // record stores of this part are write-through (st_pub, wave.h)
};
// fp64 stepper: the controls (read once per substep) and the solver's warm start (read once, written once) stay in global memory —
// the warm start in the env's record, where the next part of the step / the next step finds it anyway, the controls in the batch's ctrl_ws
// ... and so do the tendon moment arms (2.5 KB: the difference between seven and eight workgroups per CU): accumulated in LDS during the
// tendon stage (S_TENJ_STAGE, storage the constraint rows take over later), written out once, [slot][tendon] so that tendon-per-lane
// reads coalesce; read back by the tendon-velocity / actuator-moment phases and, for tendon-limit rows only, by the solver
// big_g (Scratch::SPILL only): the wave slot's block of the BIG workspace (TaskDev::big_ws) — the contact records and the tendon
// stage's wrap results of the 48-slot fp64 scratch live there (CON / S_TWRES below)
#ifdef MYO_WARM_LDS      /* diagnostic build: the fp64 stepper's warm start in LDS (280 B: seven workgroups per CU) — is the warm start's trip through global memory what differs between two runs? */
#ifdef MYO_TENJ_LDS     /* ... and the moment arms (2.5 KB more) */
template <> struct ScratchPoses<double> { double* warm_g; double* ctrl_g; double* tenj_g; char* big_g; int pub; int ws_idx; double qacc_warm_[MYO_NV_MAX]; double ten_J_[MYO_NT_MAX * MYO_TJ_MAX]; };
#else
template <> struct ScratchPoses<double> { double* warm_g; double* ctrl_g; double* tenj_g; char* big_g; int pub; int ws_idx; double qacc_warm_[MYO_NV_MAX]; };
#endif
#else
template <> struct ScratchPoses<double> { double* warm_g; double* ctrl_g; double* tenj_g; char* big_g; int pub; int ws_idx; };
#endif
/* bytes per wave slot of TaskDev::big_ws: MYO_NCON_BIG contact records, then 7 doubles per geom wrap (MYO_BIGWS_GW of them) */
#define MYO_BIGWS_GW 96
#define MYO_BIGWS_WRES (MYO_NCON_BIG * 136)
#define MYO_BIGWS_BYTES ((MYO_BIGWS_WRES + MYO_BIGWS_GW * 7 * 8 + 127) / 128 * 128)

template <typename T, int NC = MYO_NCON_MAX>
struct Scratch : ScratchPoses<T> {
  // per-env friction coefficients kept per geom of an object group: all three (sliding, torsional, rolling) in the big scratch that
  // batches with a die get, the sliding one in the base scratch (an object group on a base batch: torsional / rolling stay nominal)
  static constexpr int OBJG_NF = NC >= MYO_NCON_BIG ? 3 : 1;
  // Round 6: the big fp64 scratch holds 48 contact slots (config E's rollouts ask for up to 45) and STILL fits eight workgroups per CU
  // (20,320 B), because its contact RECORDS (6.5 KB) and the tendon stage's wrap results (4 KB of staging that used to sit in con[])
  // live in the wave slot's block of a global workspace (TaskDev::big_ws, L2-resident: 2048 live slots x 12 KB): CON(s, ci), S_TWRES.
  // What the solver reads per ROW stays in LDS (the rows' arrays, a contact's D: conD).
  // (Round 5, then with 34 slots:) the big fp64 scratch keeps what a substep touches ONCE in GLOBAL memory — the object group's friction triples and the muscle
  // activations in the env record, where they live anyway; the tendon lengths, the activation rates and the reward terms in the env's
  // workspace (ctrl_ws): 1,072 B, the difference between six and seven workgroups per CU (24,064 -> 22,992 B).  Accessors: S_OBJF,
  // S_ACT / act_set, S_TEN_LENGTH, S_ACT_DOT, S_RWD below; one wave per workgroup, so a lane's global store is visible to the loads of a
  // later phase in program order.  (A first cut moved the limit rows' D / ids and the rows' active flags instead — the solver's own
  // arrays: 2.98 -> 2.80 ms for the die's k_step where a 29-slot capacity experiment had said 2.62.)
  static constexpr bool SPILL = sizeof(T) == sizeof(HP) && NC >= MYO_NCON_BIG;
  static_assert(NC >= MYO_NCON_F64 && MYO_NLIM_MAX + 4 * NC <= 256, "contact capacity: at least the smallest base (the aliases below are sized for it), at most four constraint rows per lane");
  // ---- state (HP in every build)
  HP qpos[MYO_NQ_MAX], qvel[MYO_NV_MAX], act[SPILL ? 1 : MYO_NU_MAX];      // (act: S_ACT)
  HP time;
  // ---- per-env parameters
  HP ball_size[2];
  union {                         // the task's per-env numbers = the record's taskd block, in this order
    struct { HP start_angle[2], x_radius, y_radius, time_period, target_xy[4]; };    // Baoding
    struct { HP goal_pos[3], goal_quat[4], pos_dist, rot_dist; };                    // die reorient
    HP taskd[MYO_TASKD_N];
  };
  union {                         // a batch has either the two Baoding balls or an object group (the die), never both
    struct { HP target_w[6]; T ball_mass[2], ball_fric[6]; };
    T objg_fric[SPILL ? 1 : MYO_OBJG_MAX * (NC >= MYO_NCON_BIG ? 3 : 1)];   // friction of the object group's geoms, OBJG_NF coefficients each (S_OBJF)
  };
  T ep_ret;
  int which_task, counter, elapsed, episode, ep_len;
  // ---- position stage
  // (short-lived arrays alias longer-lived storage, see the S_* accessors below)
  T cdof[MYO_NV_MAX * 6];
  union {                         // tendon lengths (tendon stage .. actuation) and activation rates (actuation .. advance) share a slot:
    HP ten_length[SPILL ? 1 : MYO_NT_MAX];    // fwd_actuation reads every length before it stores the first rate.  HP: what muscle forces and
    T act_dot[SPILL ? 1 : MYO_NU_MAX];        // tendon limits are made of (S_TEN_LENGTH, S_ACT_DOT)
  };
                                  // (ten_vel, act_force: S_TEN_VEL / S_ACT_FORCE below; the moment arms: ScratchPoses)
  alignas(16) T H[MYO_H_SIZE];   // dense system matrix / its Cholesky factor (packed lower triangle, MYO_HIDX); hosts short-lived arrays too
  // ---- constraints
  int ncon, nefc, nl, ntl, bad, solver_iter;   // ncon: contact SLOTS (four rows each, ContactRec)
  unsigned char hperm[MYO_NV_MAX];   // dof -> row of the Newton system (DevModel::hperm; identity unless the block-arrow solver is on)
  // (from con[] on: one contiguous block, the staging area of the tendon stage — S_TWP / S_TWRES — which runs before any of it is live.
  //  Mixed stepper: up to rk.  fp64 stepper: up to qfrc_smooth, where its body poses live during the position stage.)
  // NREC record slots over the 4 NC contact rows of the efc_* arrays: the fp64 base scratch has 22 slots for 64 contact rows, because the
  // rows are SHARED with the limit rows (capacity MYO_NLIM_MAX, ~11 in use on the hand): a substep holds min(NREC, (rows - limit rows) / 4)
  // contacts (contacts_emit_*), at least NC.  Measured on the hand with P2's ball sizes: up to 19 contacts (oracle, 32 episodes).
  static constexpr int NREC = (sizeof(T) == sizeof(HP) && NC == MYO_NCON_F64) ? MYO_NREC_F64 : NC;
  alignas(16) ContactRec<T> con[SPILL ? 1 : NREC];                     // (SPILL: in the big workspace — CON below; the one record here is never used)
  alignas(16) short lim_id[MYO_NLIM_MAX];                              // dof (joint rows) / tendon (tendon rows); bit 15: the upper limit (row sign -1)
  T efc_D[MYO_NLIM_MAX];                                               // limit rows only; contact rows: con[] / conD
  T conD[SPILL ? NREC : 1];                                            // SPILL: D of every contact slot (CON_D)
  // bvec, efc_jv, efc_force: contiguous, in this order — the linear solves stage their operands from bvec on (S_SOLVE_STAGE)
  alignas(16) T bvec[MYO_NB_MAX * 6];
  T efc_jv[MYO_NLIM_MAX + 4 * NC], efc_force[MYO_NLIM_MAX + 4 * NC], efc_jar[MYO_NLIM_MAX + 4 * NC];
  unsigned char efc_active[MYO_NLIM_MAX + 4 * NC];
  // the seven dof vectors: contiguous, in this order (fp64 stepper: the position stage's poses live here, see ScratchPoses)
  alignas(16) T qfrc_smooth[MYO_NV_MAX], qacc_smooth[MYO_NV_MAX], qacc[MYO_NV_MAX], qfrc_constraint[MYO_NV_MAX];
  T Ma[MYO_NV_MAX], search[MYO_NV_MAX], Mv[MYO_NV_MAX];   // search: -gradient between update_constraint and the solve, then the Newton direction
  T qM[MYO_NM_MAX];               // tree-sparse inertia matrix (written by crb, i.e. after the tendon stage: the last piece of its staging area)
  RkScratch<T>* rk;               // null unless the model integrates with RK4
  // ---- task layer
  T rwd[SPILL ? 1 : 8];              // (S_RWD)
#ifdef MYO_PROF
  unsigned long long prof[MYO_NPROF], prof_t;
#endif
};

// body poses, origin, tree reference points: ScratchPoses members (mixed stepper) / inside the seven dof vectors (fp64 stepper)
#define MYO_POSE_ACC(NAME, TYPE, MEMBER, OFF)                                                                                         \
  template <typename T, int NC> DEV TYPE* NAME(Scratch<T, NC>& s) {                                                                   \
    if constexpr (sizeof(T) == sizeof(HP)) return reinterpret_cast<TYPE*>(s.qfrc_smooth) + (OFF); else return s.MEMBER;               \
  }                                                                                                                                   \
  template <typename T, int NC> DEV const TYPE* NAME(const Scratch<T, NC>& s) {                                                       \
    if constexpr (sizeof(T) == sizeof(HP)) return reinterpret_cast<const TYPE*>(s.qfrc_smooth) + (OFF); else return s.MEMBER;         \
  }
MYO_POSE_ACC(S_XPOS, HP, xpos, 0)
MYO_POSE_ACC(S_XQUAT, HP, xquat, MYO_NB_MAX * 3)
MYO_POSE_ACC(S_ORIGIN, HP, origin, MYO_NB_MAX * 7)
MYO_POSE_ACC(S_COM, T, com, MYO_NB_MAX * 7 + 3)
#undef MYO_POSE_ACC
static_assert(MYO_NB_MAX * 10 + 3 <= 7 * MYO_NV_MAX && (MYO_NB_MAX * 3) % 2 == 0, "fp64 stepper: xpos, xquat, origin, com fit in the seven dof vectors; xquat stays 16-byte aligned");

// the fp32 stages' view of the two HP arrays they read every substep: a float copy in the mixed stepper,
// the HP array itself in the fp64 stepper (O = 0 there)
template <typename T, int NC> DEV T* S_QVELT(Scratch<T, NC>& s) { if constexpr (sizeof(T) == sizeof(HP)) return reinterpret_cast<T*>(s.qvel); else return s.qvelT_; }
template <typename T, int NC> DEV const T* S_QVELT(const Scratch<T, NC>& s) { if constexpr (sizeof(T) == sizeof(HP)) return reinterpret_cast<const T*>(s.qvel); else return s.qvelT_; }
template <typename T, int NC> DEV T* S_XPOST(Scratch<T, NC>& s) { if constexpr (sizeof(T) == sizeof(HP)) return reinterpret_cast<T*>(S_XPOS(s)); else return s.xposT_; }
template <typename T, int NC> DEV const T* S_XPOST(const Scratch<T, NC>& s) { if constexpr (sizeof(T) == sizeof(HP)) return reinterpret_cast<const T*>(S_XPOS(s)); else return s.xposT_; }

// Aliases: arrays whose lifetime ends before the buffer they live in is next written.
//   H is only live from qacc_smooth to the end of the solver / Euler solve.  Before that
//   it hosts: cinert (com_pos..RNE), crb (CRB only) then cdof_dot (velocity stage), the passive /
//   bias / actuator force vectors (velocity..actuation), and after the physics the observation.
//   xanchor / xaxis (until cdof is built) -> efc_jar / efc_jv; the kinematics stage keeps its HP joint
//   anchors / axes (parent frame) in con[], which is dead until the collision stage
//   xipos (until cinert is built) -> efc_force;  the compaction prefix npre -> efc_jv;  cfrcb (RNE) -> bvec
//   every row's B / K imp (pos - margin) (constraint_limits, collision .. efc_reference) -> the row's efc_force / efc_jar entry;  the reference acceleration aref
//   (efc_reference .. the solver's warm-start choice, where efc_jar = J qacc - aref replaces it) -> efc_jar
#define S_CINERT(s) ((s).H)
#define S_CRB(s) ((s).H + MYO_NB_MAX * 10)
#define S_CDOFDOT(s) ((s).H + MYO_NB_MAX * 10)
#define S_QFRC_PASSIVE(s) ((s).H + MYO_NB_MAX * 20)
#define S_QFRC_BIAS(s) ((s).H + MYO_NB_MAX * 20 + MYO_NV_MAX)
#define S_QFRC_ACTUATOR(s) ((s).H + MYO_NB_MAX * 20 + 2 * MYO_NV_MAX)
#define S_OBS(s) ((s).H + MYO_NB_MAX * 20 + 3 * MYO_NV_MAX)
#define S_TWP(s) (reinterpret_cast<T*>((s).con))   /* tendon stage, mixed stepper: fp32 position of every path element (con[] is dead until the collision stage) */
/* tendon stage: HP wrap results (7 per geom wrap) behind the T path points, running on through the limit-row and efc_* arrays.
   The fp64 stepper stages no path points (its moment arms take the HP points the lengths are made of): the results start at con[] */
template <typename T, int NC> DEV auto S_TWRES(Scratch<T, NC>& s, int nwrap) {
  if constexpr (Scratch<T, NC>::SPILL) { (void)nwrap; return (GPTR(HP))((GPTR(char))s.big_g + MYO_BIGWS_WRES); }      // (the big workspace)
  else return reinterpret_cast<HP*>(reinterpret_cast<char*>(s.con) + (sizeof(T) == sizeof(HP) ? (size_t)0 : ((3 * (size_t)nwrap * sizeof(T) + 7) & ~(size_t)7)));
}
/* operand stage of the linear solves (chol_factor_solve_reg, arrow_eliminate_blocks): from bvec on through efc_jv, efc_force —
   body vectors, J v and the row forces are all rebuilt after a solve */
#define S_SOLVE_STAGE(s) ((s).bvec)
#define S_ACT_GF(s) (static_cast<T*>((s).Ma))   /* gear * actuator force (actuation stage, NU_MAX entries through Ma, search: the body velocities that live there are dead after efc_reference) */
/* HP [2][MYO_NJ_MAX * 3]: in con[] — or, for the scratch that keeps no records in LDS (SPILL), from lim_id on through efc_D, conD, bvec */
template <typename T, int NC> DEV HP* S_KTMP(Scratch<T, NC>& s) {
  if constexpr (Scratch<T, NC>::SPILL) {
    typedef Scratch<T, NC> S;
    static_assert(offsetof(S, efc_jv) - offsetof(S, lim_id) >= 2 * MYO_NJ_MAX * 3 * sizeof(HP) && offsetof(S, lim_id) % 8 == 0, "kinematics temporaries fit before efc_jv");
    return reinterpret_cast<HP*>(s.lim_id);
  } else return reinterpret_cast<HP*>(s.con);
}
#define S_XANCHOR(s) ((s).efc_jar)
#define S_XAXIS(s) ((s).efc_jv)
#define S_XIPOS(s) ((s).efc_force)
#define S_NPRE(s) (reinterpret_cast<int*>((s).efc_jv))   /* 64 ints; efc_jv is free from the end of com_pos (S_XAXIS) to efc_reference */
#define S_CFRCB(s) ((s).bvec)
// RK4's combined stage derivative (2 nv + nu numbers): in efc_jv, dead between two forward() calls (J v of the last line search)
#define S_RKDX(s) ((s).efc_jv)
#define S_ROW_B(s) ((s).efc_force)      /* every row's B and K imp (pos - margin), from the lane that builds the row to efc_reference */
#define S_ROW_KIP(s) ((s).efc_jar)
#define S_AREF(s) ((s).efc_jar)
/* world-frame force of every contact (3 per contact), staged by J' f: in efc_jv, dead between the line search that consumed J v and the next J v */
#define S_CONF(s) (static_cast<T*>((s).efc_jv))
/* ... and the world torque of the slots whose rows are rotations (condim 4 / 6): in bvec, dead between J v and the next body_vectors */
#define S_CONTQ(s) (static_cast<T*>((s).bvec))
#define S_CVEL(s) (static_cast<T*>((s).qfrc_constraint))   /* body velocities (velocity stage) live in the solver vectors qfrc_constraint,Ma,search,Mv */
/* tendon velocities (velocity stage .. actuation) and actuator forces (actuation; read by the stage dump) live in qacc_smooth, qacc and the first
   entries of qfrc_constraint, which the solver writes later (the body velocities above are dead when the actuation stage writes the forces) */
#define S_TEN_VEL(s) (static_cast<T*>((s).qacc_smooth))
#define S_ACT_FORCE(s) (static_cast<T*>((s).qacc_smooth) + MYO_NT_MAX)
static_assert(MYO_NB_MAX * 20 + 3 * MYO_NV_MAX + MYO_OBS_MAX <= MYO_H_SIZE && MYO_NU_MAX <= 2 * MYO_NV_MAX, "H aliases; S_ACT_GF");
static_assert(2 * MYO_NLIM_MAX <= MYO_NEFC_MAX && 64 * sizeof(int) <= (MYO_NLIM_MAX + 4 * MYO_NCON_F64) * sizeof(float), "S_NPRE in efc_jv");
static_assert(MYO_NT_MAX * MYO_TJ_MAX <= 1024 && MYO_NU_MAX <= 64, "packed actuator gather entries are 10 + 6 bits");
static_assert(MYO_NV_MAX * 6 <= MYO_NB_MAX * 10, "cdof_dot fits where crb was");
static_assert(MYO_NB_MAX * 6 <= 4 * MYO_NV_MAX, "cvel fits in qfrc_constraint..Mv");
static_assert(MYO_NT_MAX <= MYO_NV_MAX + 4 && MYO_NT_MAX + MYO_NU_MAX <= 3 * MYO_NV_MAX, "S_TEN_VEL ends before qfrc_constraint (cvel), S_ACT_FORCE inside qacc_smooth..qfrc_constraint");
static_assert(2 * MYO_NV_MAX + MYO_NU_MAX <= MYO_NEFC_MAX, "the RK4 stage derivative fits in efc_jv");
static_assert(MYO_NJ_MAX * 3 <= MYO_NEFC_MAX && MYO_NB_MAX * 3 <= MYO_NEFC_MAX && 64 <= MYO_NEFC_MAX, "efc aliases");
static_assert(2 * MYO_NJ_MAX * 3 * sizeof(HP) <= MYO_NCON_MAX * sizeof(ContactRec<float>) && 2 * MYO_NJ_MAX * 3 * sizeof(HP) <= MYO_NCON_F64 * sizeof(ContactRec<double>), "kinematics temporaries fit in con[]");
// controls and warm start of the env: LDS members (mixed stepper) / global memory (fp64 stepper, ScratchPoses<double>).  A lane only ever
// reads entries it wrote itself (lane i <-> entry i, i + 64, ...), so the global copies need no fence inside a workgroup.
// moment arm of tendon t with respect to the slot-th dof it moves
template <typename T, int NC> DEV T tenj_get(const Scratch<T, NC>& s, int t, int slot) {
#if defined(MYO_TENJ_LDS) && defined(MYO_WARM_LDS)
  if constexpr (sizeof(T) == sizeof(HP)) return (T)s.ten_J_[slot * MYO_NT_MAX + t]; else return s.ten_J[t * MYO_TJ_MAX + slot];
#else
  if constexpr (sizeof(T) == sizeof(HP)) return (T)((GPTR(const double))s.tenj_g)[slot * MYO_NT_MAX + t]; else return s.ten_J[t * MYO_TJ_MAX + slot];
#endif
}
// all MYO_TJ_MAX of them (requested together: the fp64 stepper's come from global memory)
template <typename T, int NC> DEV void tenj_row(const Scratch<T, NC>& s, int t, T* j) {
#pragma unroll
  for (int k = 0; k < MYO_TJ_MAX; ++k) j[k] = tenj_get(s, t, k);
}
// where the tendon stage accumulates them: the member itself (mixed) / the constraint-row arrays, free until constraint_limits (fp64)
template <typename T, int NC> DEV T* S_TENJ_STAGE(Scratch<T, NC>& s) { if constexpr (sizeof(T) == sizeof(HP)) return s.efc_jv; else return s.ten_J; }
static_assert(MYO_NT_MAX * MYO_TJ_MAX <= 3 * (MYO_NLIM_MAX + 4 * MYO_NCON_F64), "fp64 stepper: the moment-arm stage fits in efc_jv, efc_force, efc_jar");
template <typename T, int NC> DEV T ctrl_get(const Scratch<T, NC>& s, int i) { if constexpr (sizeof(T) == sizeof(HP)) return (T)((GPTR(const double))s.ctrl_g)[i]; else return s.ctrl_[i]; }
template <typename T, int NC> DEV void ctrl_set(Scratch<T, NC>& s, int i, T v) { if constexpr (sizeof(T) == sizeof(HP)) ((GPTR(double))s.ctrl_g)[i] = (double)v; else s.ctrl_[i] = v; }
// (fp64 stepper: in the wave slot's workspace between load_env and store_env, which copy it from / to the env's record — warm_g)
#ifdef MYO_WARM_LDS
template <typename T, int NC> DEV T warm_get(const Scratch<T, NC>& s, int i) { return (T)s.qacc_warm_[i]; }
template <typename T, int NC> DEV void warm_set(Scratch<T, NC>& s, int i, T v) { s.qacc_warm_[i] = v; }
#else
template <typename T, int NC> DEV T warm_get(const Scratch<T, NC>& s, int i) { if constexpr (sizeof(T) == sizeof(HP)) return (T)((GPTR(const double))s.ctrl_g)[MYO_WS_WARM + i]; else return s.qacc_warm_[i]; }
template <typename T, int NC> DEV void warm_set(Scratch<T, NC>& s, int i, T v) { if constexpr (sizeof(T) == sizeof(HP)) ((GPTR(double))s.ctrl_g)[MYO_WS_WARM + i] = (double)v; else s.qacc_warm_[i] = v; }
#endif
#define MYO_NEFC_MIN (MYO_NLIM_MAX + 4 * MYO_NCON_F64)   /* rows of the smallest scratch: every alias of an efc_* array must fit in this many */
static_assert(2 * MYO_NLIM_MAX <= MYO_NEFC_MIN && 2 * MYO_NV_MAX + MYO_NU_MAX <= MYO_NEFC_MIN && MYO_NJ_MAX * 3 <= MYO_NEFC_MIN && MYO_NB_MAX * 3 <= MYO_NEFC_MIN && 64 <= MYO_NEFC_MIN,
              "efc aliases (S_RKDX, S_XANCHOR / S_XAXIS, S_XIPOS) in the smallest scratch");

// ---- phase functions are real (non-inlined) functions in the gfx950 build: each gets its own
// register allocation (the fully inlined kernel spilled ~170 VGPRs and was several MB of code).
// They do not receive the model / task / scratch through arguments: the model and task live in
// __constant__ memory (scalar loads), the scratch is THE dynamic LDS block of the workgroup, so
// every access keeps its address space (ds_* / s_load_*).  LDS vectors are passed as byte offsets.
#ifdef MYO_EMU
#define DEVFN static inline
#define MYO_BIND_M(T) const DevModel<T>& M = M_in;
#define MYO_BIND_K const TaskDev& K = K_in;
#define MYO_BIND_S(T) auto& s = s_in;
#define LREF(T) T*
#define LCREF(T) const T*
#define LOFF(s, p) (p)
#define LPTR(T, r) (r)
#define LNULL(T) ((T*)0)
#define LISNULL(r) ((r) == 0)
#else
#define DEVFN __device__ __noinline__
extern __shared__ __align__(16) unsigned char myo_lds[];
__constant__ DevModel<float> c_model_f;
__constant__ DevModel<double> c_model_d;
__constant__ TaskDev c_task;
template <typename T> __device__ __forceinline__ const DevModel<T>& myo_cmodel();
template <> __device__ __forceinline__ const DevModel<float>& myo_cmodel<float>() { return c_model_f; }
template <> __device__ __forceinline__ const DevModel<double>& myo_cmodel<double>() { return c_model_d; }
#define MYO_BIND_M(T) const DevModel<T>& M = myo_cmodel<T>(); (void)M_in;
#define MYO_BIND_K const TaskDev& K = c_task; (void)K_in;
#define MYO_BIND_S(T) Scratch<T, NC>& s = *reinterpret_cast<Scratch<T, NC>*>(myo_lds); (void)s_in;
#define LREF(T) int
#define LCREF(T) int
#define LOFF(s, p) ((int)((const char*)(p) - (const char*)&(s)))
#define LPTR(T, r) (reinterpret_cast<T*>(myo_lds + (r)))
#define LNULL(T) (-1)
#define LISNULL(r) ((r) < 0)
#endif

// what Scratch::SPILL keeps in global memory (the env record: act = the na doubles in front of the warm start; the env workspace)
// (SPILL: typed global pointers — GPTR, wave.h — hence `auto`)
template <typename T, int NC> DEV auto S_OBJF(const TaskDev& K, Scratch<T, NC>& s) { if constexpr (Scratch<T, NC>::SPILL) return (GPTR(T))s.warm_g + K.objf_off; else return (T*)s.objg_fric; }
template <typename T, int NC> DEV auto S_OBJF(const TaskDev& K, const Scratch<T, NC>& s) { if constexpr (Scratch<T, NC>::SPILL) return (GPTR(const T))s.warm_g + K.objf_off; else return (const T*)s.objg_fric; }
template <typename T, int NC> DEV auto S_ACT(const DevModel<T>& M, const Scratch<T, NC>& s) { if constexpr (Scratch<T, NC>::SPILL) return (GPTR(const HP))s.warm_g - M.na; else return (const HP*)s.act; }
// (a part of a step that hands the record on writes it through, like the warm start: warm_set)
template <typename T, int NC> DEV void act_set(const DevModel<T>& M, Scratch<T, NC>& s, int i, HP v) {
  if constexpr (Scratch<T, NC>::SPILL) st_pub((GPTR(double))s.warm_g - M.na + i, (double)v, UNI(s.pub)); else s.act[i] = v;
}
template <typename T, int NC> DEV auto S_TEN_LENGTH(Scratch<T, NC>& s) { if constexpr (Scratch<T, NC>::SPILL) return (GPTR(HP))s.ctrl_g + MYO_WS_TLEN; else return (HP*)s.ten_length; }
template <typename T, int NC> DEV auto S_TEN_LENGTH(const Scratch<T, NC>& s) { if constexpr (Scratch<T, NC>::SPILL) return (GPTR(const HP))s.ctrl_g + MYO_WS_TLEN; else return (const HP*)s.ten_length; }
template <typename T, int NC> DEV auto S_ACT_DOT(Scratch<T, NC>& s) { if constexpr (Scratch<T, NC>::SPILL) return (GPTR(T))s.ctrl_g + MYO_WS_ADOT; else return (T*)s.act_dot; }
template <typename T, int NC> DEV auto S_ACT_DOT(const Scratch<T, NC>& s) { if constexpr (Scratch<T, NC>::SPILL) return (GPTR(const T))s.ctrl_g + MYO_WS_ADOT; else return (const T*)s.act_dot; }
template <typename T, int NC> DEV auto S_RWD(Scratch<T, NC>& s) { if constexpr (Scratch<T, NC>::SPILL) return (GPTR(T))s.ctrl_g + MYO_WS_RWD; else return (T*)s.rwd; }
template <typename T, int NC> DEV auto S_RWD(const Scratch<T, NC>& s) { if constexpr (Scratch<T, NC>::SPILL) return (GPTR(const T))s.ctrl_g + MYO_WS_RWD; else return (const T*)s.rwd; }
// contact slot ci's record: the LDS member, or the wave slot's block of the big workspace (Scratch::SPILL)
template <typename T, int NC> DEV auto& CON(Scratch<T, NC>& s, int ci) {
  if constexpr (Scratch<T, NC>::SPILL) return ((GPTR(ContactRec<T>))s.big_g)[ci]; else return s.con[ci];
}
template <typename T, int NC> DEV const auto& CON(const Scratch<T, NC>& s, int ci) {
  if constexpr (Scratch<T, NC>::SPILL) return ((GPTR(const ContactRec<T>))s.big_g)[ci]; else return s.con[ci];
}
// ... and its D (SPILL: kept in LDS, the solver reads it per row)
template <typename T, int NC> DEV T CON_D(const Scratch<T, NC>& s, int ci) { if constexpr (Scratch<T, NC>::SPILL) return s.conD[ci]; else return s.con[ci].D; }
template <typename T, int NC> DEV void CON_SET_D(Scratch<T, NC>& s, int ci, T v) { if constexpr (Scratch<T, NC>::SPILL) s.conD[ci] = v; else s.con[ci].D = v; }
// the record's support list as words (same address space as the record)
#define CON_SUP_WORDS(c) (reinterpret_cast<decltype(&(c).pk)>((c).sup))
template <typename T, int NC> DEV T row_D(const Scratch<T, NC>& s, int r, int nlim) { return r < nlim ? s.efc_D[r] : CON_D(s, (r - nlim) >> 2); }

// ------------------------------------------------------------------------------------------
// small math
template <typename T> DEV T dot3(const T* a, const T* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <typename T> DEV void cross3(T* r, const T* a, const T* b) {
  T x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
template <typename T> DEV T norm3(const T* a) { return sqrt(dot3(a, a)); }
template <typename T> DEV T normalize3(T* a) {
  T n = norm3(a);
  if (n < MYO_MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; } else { T i = 1 / n; a[0] *= i; a[1] *= i; a[2] *= i; }
  return n;
}
template <typename T> DEV void mulmatvec3(T* r, const T* R, const T* v) {
  T x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
    z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
template <typename T> DEV void mulmatTvec3(T* r, const T* R, const T* v) {
  T x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2], y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2],
    z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
template <typename T> DEV void mulmat3(T* r, const T* A, const T* B) {
  T t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  for (int i = 0; i < 9; ++i) r[i] = t[i];
}
template <typename T> DEV void quat2mat(T* R, const T* q) {
  T w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = w * w - x * x - y * y + z * z;
}
template <typename T> DEV void mulquat(T* r, const T* a, const T* b) {
  T t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
    t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
}
template <typename T> DEV void normalize4(T* q) {
  T n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MYO_MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; } else { T i = 1 / n; q[0] *= i; q[1] *= i; q[2] *= i; q[3] *= i; }
}
template <typename T> DEV void axisangle2quat(T* q, const T* axis, T angle) {
  T s = sin(angle * (T)0.5);
  q[0] = cos(angle * (T)0.5); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
template <typename T> DEV T tmax(T a, T b) { return a > b ? a : b; }
template <typename T> DEV T tmin(T a, T b) { return a < b ? a : b; }
template <typename T> DEV T tclamp(T x, T lo, T hi) { return x < lo ? lo : (x > hi ? hi : x); }

// (J v)[row e of the slot] from the relative linear / angular velocity of the two bodies at the contact
template <typename C, typename T> DEV T con_row_val(const C& c, int kind, int e, const T* rel_lin, const T* rel_ang) {
  T t2[3], fr[6];
  con_frame(c, fr);
  cross3(t2, fr, fr + 3);
  const int second = e >> 1;
  const T mu = second ? c.muB : c.muA;
  // the pair's axis: kind 0: t1 | t2 (on the linear velocity); 1: n | t1, 2: t2, 4: n (on the angular velocity); 3: none (mu = 0)
  const bool ax_n = (kind == 1 && !second) || kind == 4, ax_t1 = (kind == 0 && !second) || (kind == 1 && second);
  const T ax[3] = {ax_n ? fr[0] : (ax_t1 ? fr[3] : t2[0]), ax_n ? fr[1] : (ax_t1 ? fr[4] : t2[1]), ax_n ? fr[2] : (ax_t1 ? fr[5] : t2[2])};
  const T* rel = kind == 0 ? rel_lin : rel_ang;
  const T val = dot3(fr, rel_lin) + ((e & 1) ? -mu : mu) * dot3(ax, rel);
  return con_pad(kind, e) ? (T)0 : val;
}

// spatial algebra in MuJoCo's com-based convention: motion = [ang; lin], force = [torque; force]
template <typename T> DEV void mul_inert_vec(T* r, const T* I, const T* v) {
  const T* w = v; const T* l = v + 3; const T* h = I + 6; T mass = I[9]; T t[3];
  r[0] = I[0] * w[0] + I[3] * w[1] + I[4] * w[2]; r[1] = I[3] * w[0] + I[1] * w[1] + I[5] * w[2];
  r[2] = I[4] * w[0] + I[5] * w[1] + I[2] * w[2];
  cross3(t, h, l); r[0] += t[0]; r[1] += t[1]; r[2] += t[2];
  cross3(t, h, w); r[3] = mass * l[0] - t[0]; r[4] = mass * l[1] - t[1]; r[5] = mass * l[2] - t[2];
}
template <typename T> DEV void cross_motion(T* r, const T* v, const T* m) {
  T a[3], b[3], c[3];
  cross3(a, v, m); cross3(b, v, m + 3); cross3(c, v + 3, m);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
template <typename T> DEV void cross_force(T* r, const T* v, const T* f) {
  T a[3], b[3], c[3];
  cross3(a, v, f); cross3(b, v + 3, f + 3); cross3(c, v, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}

// per-env overrides of model constants (P2 randomisation writes ball mass / size / friction
// into the model, /root/reference/src/envs/baoding.py:559-604)
template <typename T, int NC> DEV T body_mass_of(const DevModel<T>& M, const TaskDev& K, const Scratch<T, NC>& s, int b) {
  if (b == K.obj1_bid) return s.ball_mass[0];
  if (b == K.obj2_bid) return s.ball_mass[1];
  return M.body_mass[b];
}
template <typename T, int NC> DEV T geom_size0_of(const DevModel<T>& M, const TaskDev& K, const Scratch<T, NC>& s, int g) {
  if (g == K.obj1_gid) return (T)s.ball_size[0];
  if (g == K.obj2_gid) return (T)s.ball_size[1];
  if (g < K.objg_gidn && g >= K.objg_gidn - 3 && g >= K.objg_gid0) return M.geom_size[3 * g] + (T)s.ball_size[0];
  return M.geom_size[3 * g];
}
template <typename T, int NC> DEV HP geom_size0_hp(const DevModel<T>& M, const TaskDev& K, const Scratch<T, NC>& s, int g) {
  if (g == K.obj1_gid) return s.ball_size[0];
  if (g == K.obj2_gid) return s.ball_size[1];
  if (g < K.objg_gidn && g >= K.objg_gidn - 3 && g >= K.objg_gid0) return M.h_geom_size[3 * g] + s.ball_size[0];
  return M.h_geom_size[3 * g];
}
// friction coefficient k (0 sliding, 1 torsional, 2 rolling) of geom g for this env: the per-env values of the Baoding balls / of
// the object group's geoms (P2 / reorient randomisation), else `nominal` (the model's, from the pair record)
template <typename T, int NC> DEV T geom_fric_of(const DevModel<T>& M, const TaskDev& K, const Scratch<T, NC>& s, int g, int k, T nominal) {
  if (K.objg_gidn > 0) {
    if (g >= K.objg_gid0 && g < K.objg_gidn && k < Scratch<T, NC>::OBJG_NF) return S_OBJF(K, s)[(g - K.objg_gid0) * Scratch<T, NC>::OBJG_NF + k];
    return nominal;
  }
  if (g == K.obj1_gid) return s.ball_fric[k];
  if (g == K.obj2_gid) return s.ball_fric[3 + k];
  return nominal;
}

// geometry of a geom of the per-env object group (the die of the reorient task, reorient.py:136-147): every geom
// centre moves outward by the env's size delta; size[1] (a capsule's half-length) of every geom grows by it, and so
// does size[0] of the group's last three geoms (the reference adds the delta to all sizes of those)
template <typename T, int NC> DEV void geom_lpos_of(const DevModel<T>& M, const TaskDev& K, const Scratch<T, NC>& s, int g, T* out) {
  out[0] = M.geom_pos[3 * g]; out[1] = M.geom_pos[3 * g + 1]; out[2] = M.geom_pos[3 * g + 2];
  if (g >= K.objg_gid0 && g < K.objg_gidn) {
    const T del = (T)s.ball_size[0];
    for (int e = 0; e < 3; ++e) if (out[e] != 0) out[e] += out[e] > 0 ? del : -del;
  }
}
template <typename T, int NC> DEV void geom_lpos_hp(const DevModel<T>& M, const TaskDev& K, const Scratch<T, NC>& s, int g, HP* out) {
  out[0] = M.h_geom_pos[3 * g]; out[1] = M.h_geom_pos[3 * g + 1]; out[2] = M.h_geom_pos[3 * g + 2];
  if (g >= K.objg_gid0 && g < K.objg_gidn) {
    const HP del = s.ball_size[0];
    for (int e = 0; e < 3; ++e) if (out[e] != 0) out[e] += out[e] > 0 ? del : -del;
  }
}
template <typename T, int NC> DEV T geom_size1_of(const DevModel<T>& M, const TaskDev& K, const Scratch<T, NC>& s, int g) {
  const T v = M.geom_size[3 * g + 1];
  return (g >= K.objg_gid0 && g < K.objg_gidn) ? v + (T)s.ball_size[0] : v;
}
template <typename T, int NC> DEV HP geom_size1_hp(const DevModel<T>& M, const TaskDev& K, const Scratch<T, NC>& s, int g) {
  const HP v = M.h_geom_size[3 * g + 1];
  return (g >= K.objg_gid0 && g < K.objg_gidn) ? v + s.ball_size[0] : v;
}
// rotation matrix of body b.  Mixed stepper: the float copy the kinematics stage left.  fp64 stepper: recomputed from the
// body's quaternion — the same quat2mat on the same numbers as the kinematics stage's, so the same bits — which keeps 1.7 KB
// of matrices out of LDS (the fp64 scratch fits six workgroups per CU without them)
template <typename T, int NC> DEV void body_rot(const Scratch<T, NC>& s, int b, T* R) {
  if constexpr (sizeof(T) == sizeof(HP)) quat2mat(R, reinterpret_cast<const T*>(S_XQUAT(s)) + 4 * b);
  else { for (int k = 0; k < 9; ++k) R[k] = s.xmat_[9 * b + k]; }
}
// position of a point given in body coordinates, in the fp32 stages' frame (world - O)
template <typename T, int NC> DEV void body_point(const Scratch<T, NC>& s, int b, const T* local, T* out) {
  const T* xp = S_XPOST(s) + 3 * b;
  T Rb[9];
  body_rot(s, b, Rb);
  mulmatvec3(out, Rb, local);
  out[0] += xp[0]; out[1] += xp[1]; out[2] += xp[2];
}
// the same in HP and in WORLD coordinates (contact distances, observation)
template <typename T, int NC> DEV void body_point_hp(const Scratch<T, NC>& s, int b, const HP* local, HP* out) {
  HP R[9];
  quat2mat(R, S_XQUAT(s) + 4 * b);
  mulmatvec3(out, R, local);
  out[0] += S_XPOS(s)[3 * b]; out[1] += S_XPOS(s)[3 * b + 1]; out[2] += S_XPOS(s)[3 * b + 2];
}
// translational Jacobian column of dof d for a world point p: cdof_lin + cdof_ang x (p - com)
template <typename T, int NC> DEV void jac_col(const DevModel<T>& M, const Scratch<T, NC>& s, int d, const T* p, T* col) {
  const T* c = S_COM(s) + 3 * M.dof_rootbody[d]; const T* cd = s.cdof + 6 * d;
  T off[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]}, t[3];
  cross3(t, cd, off);
  col[0] = cd[3] + t[0]; col[1] = cd[4] + t[1]; col[2] = cd[5] + t[2];
}

// ------------------------------------------------------------------------------------------
// P2: kinematics (mj_kinematics).  HP throughout (see the precision plan at the top): the chain qpos -> poses
// is what contact distances are made of.  Leaves xpos / xquat (HP, world), and for the fp32 stages xmat, the
// shifted positions S_XPOST / S_XIPOS and the shifted joint anchors / axes.
template <typename T, int NC>
DEVFN void kinematics(const DevModel<T>& M_in, Scratch<T, NC>& s_in) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  WAVE_FN
  PHASE {
    if (lane == 0) {
      S_XPOS(s)[0] = S_XPOS(s)[1] = S_XPOS(s)[2] = 0; S_XQUAT(s)[0] = 1; S_XQUAT(s)[1] = S_XQUAT(s)[2] = S_XQUAT(s)[3] = 0;
      if constexpr (sizeof(T) != sizeof(HP)) { for (int k = 0; k < 9; ++k) s.xmat_[k] = (k % 4 == 0) ? (T)1 : (T)0; }
    }
    if constexpr (sizeof(T) != sizeof(HP)) { if (lane < M.nv) s.qvelT_[lane] = (T)s.qvel[lane]; }
  }
  SYNC();
  // Three stages instead of one walk down the tree with everything inside it (lane = body throughout):
  //  1. parent-independent: every body composes, IN ITS PARENT'S FRAME, its offset (body_pos / body_quat) with
  //     the rotations / translations of its own joints (the sin / cos of the joint angles, the anchor and axis
  //     of every joint).  All table loads and all transcendentals happen here, for all bodies at once.
  //  2. the level loop only chains poses: q = q_parent * q_local, p = p_parent + R(q_parent) p_local — one LDS
  //     round trip and ~40 flops per tree level instead of the joint code.
  //  3. parallel again: xmat, xipos, and the world anchors / axes from the parent's final pose.
  // (mj_kinematics composes in the world frame; the two are the same maps, rounded differently.)
  HP* const kanchor = S_KTMP(s);
  HP* const kaxis = kanchor + MYO_NJ_MAX * 3;
  LANE_VAR(int, k_depth); LANE_VAR(int, k_par); LANE_VAR(int, k_jn); LANE_VAR(int, k_ja); LANE_VAR(int, k_free);
  LANE_VAR(HP, k_p0); LANE_VAR(HP, k_p1); LANE_VAR(HP, k_p2);
  LANE_VAR(HP, k_q0); LANE_VAR(HP, k_q1); LANE_VAR(HP, k_q2); LANE_VAR(HP, k_q3);
  PHASE {
    const int b = lane;
    LV(k_depth) = -1; LV(k_par) = 0; LV(k_jn) = 0; LV(k_ja) = 0; LV(k_free) = 0;
    LV(k_p0) = LV(k_p1) = LV(k_p2) = 0; LV(k_q0) = 1; LV(k_q1) = LV(k_q2) = LV(k_q3) = 0;
    // the body's whole record — tree position, offset pose, and qpos address / type / anchor / axis / reference angle of its (<= 3)
    // joints — in ONE level of table loads, requested together (bk_i / bk_f: host-packed, myobatch.hip).  Rounds 1-5 walked
    // body -> joint -> qpos0 with a dependent vector load at every step and inside the joint loop (15 memory round trips a call).
    const int bb = (b > 0 && b < M.nbody) ? b : 0;
    int bi[MYO_BK_I];
    HP bf[MYO_BK_F];
#pragma unroll
    for (int k = 0; k < MYO_BK_I; ++k) bi[k] = M.bk_i[MYO_BK_I * bb + k];
#pragma unroll
    for (int k = 0; k < MYO_BK_F; ++k) bf[k] = M.h_bk_f[MYO_BK_F * bb + k];
#pragma unroll
    for (int k = 0; k < MYO_BK_I; ++k) MYO_PIN(bi[k]);
#pragma unroll
    for (int k = 0; k < MYO_BK_F; ++k) MYO_PIN(bf[k]);
    if (b > 0 && b < M.nbody) {
      const int jn = bi[2], ja = bi[3];
      LV(k_depth) = bi[0]; LV(k_par) = bi[1]; LV(k_jn) = jn; LV(k_ja) = ja;
      HP p[3] = {bf[0], bf[1], bf[2]};
      HP q[4] = {bf[3], bf[4], bf[5], bf[6]};
      if (bi[4]) {                                   // free joint: the pose is the state (parent = world)
        const int qa = bi[5];
        HP qq[4] = {s.qpos[qa + 3], s.qpos[qa + 4], s.qpos[qa + 5], s.qpos[qa + 6]};
        normalize4(qq);
        for (int k = 0; k < 4; ++k) { s.qpos[qa + 3 + k] = qq[k]; q[k] = qq[k]; }
        for (int k = 0; k < 3; ++k) { p[k] = s.qpos[qa + k]; kanchor[3 * ja + k] = p[k]; }
        kaxis[3 * ja] = 0; kaxis[3 * ja + 1] = 0; kaxis[3 * ja + 2] = 1;
        LV(k_free) = 1;
      } else {
#pragma unroll
        for (int k = 0; k < MYO_BK_NJ; ++k) {
          if (k >= jn) break;
          const int j = ja + k;
          const int qa = bi[5 + k], jtype = bi[8 + k];
          const HP jpos[3] = {bf[7 + 7 * k], bf[8 + 7 * k], bf[9 + 7 * k]};
          const HP jaxis[3] = {bf[10 + 7 * k], bf[11 + 7 * k], bf[12 + 7 * k]};
          HP R[9], anchor[3], axis[3];
          quat2mat(R, q);
          mulmatvec3(anchor, R, jpos);
          anchor[0] += p[0]; anchor[1] += p[1]; anchor[2] += p[2];
          mulmatvec3(axis, R, jaxis);
          for (int e = 0; e < 3; ++e) { kanchor[3 * j + e] = anchor[e]; kaxis[3 * j + e] = axis[e]; }   // parent frame for now
          const HP ang = s.qpos[qa] - bf[13 + 7 * k];
          if (jtype == 2) {
            p[0] += axis[0] * ang; p[1] += axis[1] * ang; p[2] += axis[2] * ang;
          } else {
            HP ql[4], R2[9], t2[3];
            axisangle2quat(ql, jaxis, ang);
            mulquat(q, q, ql);
            quat2mat(R2, q);
            mulmatvec3(t2, R2, jpos);
            p[0] = anchor[0] - t2[0]; p[1] = anchor[1] - t2[1]; p[2] = anchor[2] - t2[2];
          }
        }
      }
      LV(k_p0) = p[0]; LV(k_p1) = p[1]; LV(k_p2) = p[2];
      LV(k_q0) = q[0]; LV(k_q1) = q[1]; LV(k_q2) = q[2]; LV(k_q3) = q[3];
    }
  }
  for (int level = 1; level <= M.maxdepth; ++level) {
    PHASE {
      const int b = lane;
      if (LV(k_depth) == level) {
        const int par = LV(k_par);
        const HP pl[3] = {LV(k_p0), LV(k_p1), LV(k_p2)}, ql[4] = {LV(k_q0), LV(k_q1), LV(k_q2), LV(k_q3)};
        const HP qp[4] = {S_XQUAT(s)[4 * par], S_XQUAT(s)[4 * par + 1], S_XQUAT(s)[4 * par + 2], S_XQUAT(s)[4 * par + 3]};
        HP Rp[9], t[3], q[4];
        quat2mat(Rp, qp);
        mulmatvec3(t, Rp, pl);
        mulquat(q, qp, ql);
        normalize4(q);
        for (int k = 0; k < 3; ++k) S_XPOS(s)[3 * b + k] = S_XPOS(s)[3 * par + k] + t[k];
        for (int k = 0; k < 4; ++k) S_XQUAT(s)[4 * b + k] = q[k];
      }
    }
    SYNC();
  }
  // O = world position of the first tree root (body 1), fp32-representable so that xpos - O is exact in HP
  PHASE {
    if (lane < 3) S_ORIGIN(s)[lane] = (sizeof(T) != sizeof(HP) && M.nbody > 1) ? (HP)(float)S_XPOS(s)[3 + lane] : (HP)0;
  }
  SYNC();
  PHASE {
    const int b = lane;
    if (b < M.nbody) {
      if constexpr (sizeof(T) != sizeof(HP)) { for (int k = 0; k < 3; ++k) s.xposT_[3 * b + k] = (T)(S_XPOS(s)[3 * b + k] - S_ORIGIN(s)[k]); }
    }
    if (b > 0 && b < M.nbody) {
      const int par = LV(k_par), jn = LV(k_jn), ja = LV(k_ja);
      const HP q[4] = {S_XQUAT(s)[4 * b], S_XQUAT(s)[4 * b + 1], S_XQUAT(s)[4 * b + 2], S_XQUAT(s)[4 * b + 3]};
      HP R[9], t[3];
      quat2mat(R, q);
      if constexpr (sizeof(T) != sizeof(HP)) { for (int k = 0; k < 9; ++k) s.xmat_[9 * b + k] = (T)R[k]; }
      const HP ipos[3] = {(HP)M.body_ipos[3 * b], (HP)M.body_ipos[3 * b + 1], (HP)M.body_ipos[3 * b + 2]};
      mulmatvec3(t, R, ipos);
      for (int k = 0; k < 3; ++k) S_XIPOS(s)[3 * b + k] = (T)(S_XPOS(s)[3 * b + k] - S_ORIGIN(s)[k] + t[k]);
      if (LV(k_free)) {
        for (int e = 0; e < 3; ++e) { S_XANCHOR(s)[3 * ja + e] = (T)(kanchor[3 * ja + e] - S_ORIGIN(s)[e]); S_XAXIS(s)[3 * ja + e] = (T)kaxis[3 * ja + e]; }
      } else if (jn > 0) {                           // anchors / axes: parent frame -> world
        const HP qp[4] = {S_XQUAT(s)[4 * par], S_XQUAT(s)[4 * par + 1], S_XQUAT(s)[4 * par + 2], S_XQUAT(s)[4 * par + 3]};
        const HP pp[3] = {S_XPOS(s)[3 * par] - S_ORIGIN(s)[0], S_XPOS(s)[3 * par + 1] - S_ORIGIN(s)[1], S_XPOS(s)[3 * par + 2] - S_ORIGIN(s)[2]};
        HP Rp[9];
        quat2mat(Rp, qp);
        for (int k = 0; k < jn; ++k) {
          const int j = ja + k;
          const HP al[3] = {kanchor[3 * j], kanchor[3 * j + 1], kanchor[3 * j + 2]};
          const HP xl[3] = {kaxis[3 * j], kaxis[3 * j + 1], kaxis[3 * j + 2]};
          HP aw[3], xw[3];
          mulmatvec3(aw, Rp, al);
          mulmatvec3(xw, Rp, xl);
          for (int e = 0; e < 3; ++e) { S_XANCHOR(s)[3 * j + e] = (T)(pp[e] + aw[e]); S_XAXIS(s)[3 * j + e] = (T)xw[e]; }
        }
      }
    }
  }
  SYNC();
}

// P2: mj_comPos — tree reference points, body inertias about them, dof motion axes
template <typename T, int NC>
DEVFN void com_pos(const DevModel<T>& M_in, const TaskDev& K_in, Scratch<T, NC>& s_in) {
  MYO_BIND_M(T) MYO_BIND_K MYO_BIND_S(T)
  WAVE_FN
  // tree reference points.  Two phases: every body lane leaves its mass (the env's own for the Baoding balls: body_mass_of) and its
  // tree root in LDS, then each ROOT lane walks the bodies in index order — the same sums in the same order as before, but the walk
  // reads LDS broadcasts only.  (Rounds 1-5 read the root id and the mass of body o through the scalar cache INSIDE the walk: two
  // dependent scalar loads and a branch per body, 23 times — 11 k of a substep's 250 k cycles for three lanes' worth of sums.)
  T* const tmp_m = s.bvec;                                        // body vectors are not live before the velocity stage
  int* const tmp_r = reinterpret_cast<int*>(s.bvec + MYO_NB_MAX);
  static_assert(2 * MYO_NB_MAX <= MYO_NB_MAX * 6 && sizeof(T) >= sizeof(int), "per-body mass and root in bvec");
  LANE_VAR(int, my_root);
  PHASE {
    const int b = lane;
    const int bc = b < M.nbody ? b : 0;
    int root = M.body_rootid[bc];
    T mt = M.body_mass[bc];
    MYO_PIN(root); MYO_PIN(mt);
    LV(my_root) = root;
    if (b < M.nbody) {
      tmp_m[b] = (b == K.obj1_bid) ? s.ball_mass[0] : ((b == K.obj2_bid) ? s.ball_mass[1] : mt);
      tmp_r[b] = root;
    }
    if (b == 0) { S_COM(s)[0] = S_COM(s)[1] = S_COM(s)[2] = 0; }
  }
  SYNC();
  PHASE {
    const int b = lane;
    if (b > 0 && b < M.nbody && LV(my_root) == b) {
      T mass = 0, c[3] = {0, 0, 0};
#pragma unroll 8
      for (int o = 1; o < M.nbody; ++o) {
        // wave-uniform LDS reads issued unconditionally, membership applied as a select
        const T mo = (tmp_r[o] == b) ? tmp_m[o] : (T)0;
        const T x = S_XIPOS(s)[3 * o], y = S_XIPOS(s)[3 * o + 1], z = S_XIPOS(s)[3 * o + 2];
        mass += mo; c[0] += mo * x; c[1] += mo * y; c[2] += mo * z;
      }
      if (mass < MYO_MINVAL) { c[0] = S_XIPOS(s)[3 * b]; c[1] = S_XIPOS(s)[3 * b + 1]; c[2] = S_XIPOS(s)[3 * b + 2]; }
      else { c[0] /= mass; c[1] /= mass; c[2] /= mass; }
      S_COM(s)[3 * b] = c[0]; S_COM(s)[3 * b + 1] = c[1]; S_COM(s)[3 * b + 2] = c[2];
    }
  }
  SYNC();
  PHASE {
    const int b = lane;
    // (every table word of the lane's body and joint first, together: see fwd_actuation)
    const int bc = b < M.nbody ? b : 0, jc = lane < M.njnt ? lane : 0;
    T imat[9], I[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) imat[k] = M.body_imat[9 * bc + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) I[k] = M.body_inertia[3 * bc + k];
    int jrec[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) jrec[k] = M.jk_i[4 * jc + k];
#pragma unroll
    for (int k = 0; k < 9; ++k) MYO_PIN(imat[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) MYO_PIN(I[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) MYO_PIN(jrec[k]);
    if (b < M.nbody) {
      T* ci = S_CINERT(s) + 10 * b;
      if (b == 0) { for (int k = 0; k < 10; ++k) ci[k] = 0; }
      else {
        T R[9], Rb[9];
        body_rot(s, b, Rb);
        mulmat3(R, Rb, imat);
        const T* c = S_COM(s) + 3 * LV(my_root);
        const T off[3] = {S_XIPOS(s)[3 * b] - c[0], S_XIPOS(s)[3 * b + 1] - c[1], S_XIPOS(s)[3 * b + 2] - c[2]};
        const T mb = tmp_m[b];                             // (body_mass_of, staged above)
        ci[0] = R[0] * R[0] * I[0] + R[1] * R[1] * I[1] + R[2] * R[2] * I[2] + mb * (off[1] * off[1] + off[2] * off[2]);
        ci[1] = R[3] * R[3] * I[0] + R[4] * R[4] * I[1] + R[5] * R[5] * I[2] + mb * (off[0] * off[0] + off[2] * off[2]);
        ci[2] = R[6] * R[6] * I[0] + R[7] * R[7] * I[1] + R[8] * R[8] * I[2] + mb * (off[0] * off[0] + off[1] * off[1]);
        ci[3] = R[0] * R[3] * I[0] + R[1] * R[4] * I[1] + R[2] * R[5] * I[2] - mb * off[0] * off[1];
        ci[4] = R[0] * R[6] * I[0] + R[1] * R[7] * I[1] + R[2] * R[8] * I[2] - mb * off[0] * off[2];
        ci[5] = R[3] * R[6] * I[0] + R[4] * R[7] * I[1] + R[5] * R[8] * I[2] - mb * off[1] * off[2];
        ci[6] = mb * off[0]; ci[7] = mb * off[1]; ci[8] = mb * off[2]; ci[9] = mb;
      }
    }
    const int j = lane;
    if (j < M.njnt) {
      // (jk_i: the joint's body, dof address, type and tree root behind one index)
      const int b = jrec[0], da = jrec[1], jtype = jrec[2];
      const T* c = S_COM(s) + 3 * jrec[3];
      const T off[3] = {c[0] - S_XANCHOR(s)[3 * j], c[1] - S_XANCHOR(s)[3 * j + 1], c[2] - S_XANCHOR(s)[3 * j + 2]};
      if (jtype == 0) {
        for (int k = 0; k < 3; ++k) { T* cd = s.cdof + 6 * (da + k); for (int e = 0; e < 6; ++e) cd[e] = 0; cd[3 + k] = 1; }
        T Rb[9];
        body_rot(s, b, Rb);
        for (int k = 0; k < 3; ++k) {
          T* cd = s.cdof + 6 * (da + 3 + k);
          const T ax[3] = {Rb[k], Rb[3 + k], Rb[6 + k]};
          cd[0] = ax[0]; cd[1] = ax[1]; cd[2] = ax[2];
          cross3(cd + 3, ax, off);
        }
      } else if (jtype == 2) {
        T* cd = s.cdof + 6 * da; cd[0] = cd[1] = cd[2] = 0;
        cd[3] = S_XAXIS(s)[3 * j]; cd[4] = S_XAXIS(s)[3 * j + 1]; cd[5] = S_XAXIS(s)[3 * j + 2];
      } else {
        T* cd = s.cdof + 6 * da; const T ax[3] = {S_XAXIS(s)[3 * j], S_XAXIS(s)[3 * j + 1], S_XAXIS(s)[3 * j + 2]};
        cd[0] = ax[0]; cd[1] = ax[1]; cd[2] = ax[2];
        cross3(cd + 3, ax, off);
      }
    }
  }
  SYNC();
}

// ------------------------------------------------------------------------------------------
// P3: spatial tendons with sphere / cylinder wrapping (mj_tendon + mju_wrap)
// do the segments p1-p2 and p3-p4 cross?  (mju_wrap's is_intersect: both line parameters a = na / det, b = nb / det in
// [0, 1]; tested on the numerators — 0 <= n <= det or det <= n <= 0 — which needs no division)
template <typename T> DEV int seg_intersect(const T* p1, const T* p2, const T* p3, const T* p4) {
  const T det = (p4[1] - p3[1]) * (p2[0] - p1[0]) - (p4[0] - p3[0]) * (p2[1] - p1[1]);
  if (fabs(det) < MYO_MINVAL) return 0;
  const T na = (p4[0] - p3[0]) * (p1[1] - p3[1]) - (p4[1] - p3[1]) * (p1[0] - p3[0]);
  const T nb = (p2[0] - p1[0]) * (p1[1] - p3[1]) - (p2[1] - p1[1]) * (p1[0] - p3[0]);
  const T lo = det > 0 ? (T)0 : det, hi = det > 0 ? det : (T)0;
  return (na >= lo && na <= hi && nb >= lo && nb <= hi);
}

template <typename T> DEV T wrap_circle(T* pnt, const T* dd, const T* sd, int has_side, T rad) {
  const T sqlen0 = dd[0] * dd[0] + dd[1] * dd[1], sqlen1 = dd[2] * dd[2] + dd[3] * dd[3], sqrad = rad * rad;
  const T dif[2] = {dd[2] - dd[0], dd[3] - dd[1]};
  const T dsq = dif[0] * dif[0] + dif[1] * dif[1];
  if (sqlen0 < sqrad || sqlen1 < sqrad || rad < MYO_MINVAL) return -1;
  if (dsq < MYO_MINVAL) return -1;
  T a = -(dif[0] * dd[0] + dif[1] * dd[1]) / dsq;
  a = tclamp(a, (T)0, (T)1);
  const T tmp[2] = {a * dif[0] + dd[0], a * dif[1] + dd[1]};
  if (tmp[0] * tmp[0] + tmp[1] * tmp[1] > sqrad && (!has_side || sd[0] * tmp[0] + sd[1] * tmp[1] >= 0)) return -1;
  T sol[2][4], good[2];
  const T sqrt0 = sqrt(sqlen0 - sqrad), sqrt1 = sqrt(sqlen1 - sqrad);
  const T inv0 = 1 / sqlen0, inv1 = 1 / sqlen1;     // one reciprocal per end point instead of eight divisions
  for (int i = 0; i < 2; ++i) {
    const T sgn = (i == 0) ? (T)1 : (T)-1;
    sol[i][0] = (dd[0] * sqrad + sgn * rad * dd[1] * sqrt0) * inv0;
    sol[i][1] = (dd[1] * sqrad - sgn * rad * dd[0] * sqrt0) * inv0;
    sol[i][2] = (dd[2] * sqrad - sgn * rad * dd[3] * sqrt1) * inv1;
    sol[i][3] = (dd[3] * sqrad + sgn * rad * dd[2] * sqrt1) * inv1;
    if (has_side) {
      const T t[2] = {sol[i][0] + sol[i][2], sol[i][1] + sol[i][3]};
      const T n = sqrt(t[0] * t[0] + t[1] * t[1]);
      good[i] = n < MYO_MINVAL ? sd[0] : (t[0] * sd[0] + t[1] * sd[1]) / n;      // (t / |t|) . sd
    } else {
      const T t[2] = {sol[i][0] - sol[i][2], sol[i][1] - sol[i][3]};
      good[i] = -(t[0] * t[0] + t[1] * t[1]);
    }
    if (seg_intersect(dd, sol[i], dd + 2, sol[i] + 2)) good[i] = -10000;
  }
  const bool first = good[0] > good[1];          // selects, not sol[i]: a run-time index would move sol to private memory
  for (int k = 0; k < 4; ++k) pnt[k] = first ? sol[0][k] : sol[1][k];
  if (seg_intersect(dd, pnt, dd + 2, pnt + 2)) return -1;
  // arc angle between the two tangent points (both on the circle): atan2(|cross|, dot) — the same angle as
  // mju_wrap's acos(dot / r^2), without its square-root loss of precision at small wrap angles (fp32)
  return rad * atan2(fabs(pnt[0] * pnt[3] - pnt[1] * pnt[2]), pnt[0] * pnt[2] + pnt[1] * pnt[3]);
}

template <typename T>
DEV T wrap_geom(T* wpnt, const T* x0, const T* x1, const T* gpos, const T* gmat, T radius, int type,
                const T* side, int has_side) {
  T p0[3], p1[3], t[3];
  for (int k = 0; k < 3; ++k) t[k] = x0[k] - gpos[k];
  mulmatTvec3(p0, gmat, t);
  for (int k = 0; k < 3; ++k) t[k] = x1[k] - gpos[k];
  mulmatTvec3(p1, gmat, t);
  if (dot3(p0, p0) < MYO_MINVAL * MYO_MINVAL || dot3(p1, p1) < MYO_MINVAL * MYO_MINVAL) return -1;   // |p| < mjMINVAL without the roots
  T axis0[3], axis1[3];
  if (type == 4) {
    T normal[3];
    axis0[0] = p0[0]; axis0[1] = p0[1]; axis0[2] = p0[2];
    normalize3(axis0);
    cross3(normal, p0, p1);
    const T nrm = norm3(normal);
    if (nrm < MYO_MINVAL) {
      int imin = 0;
      T amin = fabs(axis0[0]);
      if (fabs(axis0[1]) < amin) { imin = 1; amin = fabs(axis0[1]); }
      if (fabs(axis0[2]) < amin) imin = 2;
      const T e[3] = {imin == 0 ? (T)1 : (T)0, imin == 1 ? (T)1 : (T)0, imin == 2 ? (T)1 : (T)0};
      cross3(normal, axis0, e);
      normalize3(normal);
    } else { normal[0] /= nrm; normal[1] /= nrm; normal[2] /= nrm; }
    cross3(axis1, normal, axis0);
    normalize3(axis1);
  } else {
    axis0[0] = 1; axis0[1] = 0; axis0[2] = 0; axis1[0] = 0; axis1[1] = 1; axis1[2] = 0;
  }
  const T s2[4] = {dot3(p0, axis0), dot3(p0, axis1), dot3(p1, axis0), dot3(p1, axis1)};
  T sd[2] = {0, 0};
  if (has_side) {
    T ps[3];
    for (int k = 0; k < 3; ++k) t[k] = side[k] - gpos[k];
    mulmatTvec3(ps, gmat, t);
    sd[0] = dot3(ps, axis0); sd[1] = dot3(ps, axis1);
    const T n = sqrt(sd[0] * sd[0] + sd[1] * sd[1]);
    if (n < MYO_MINVAL) { sd[0] = radius; sd[1] = 0; } else { const T sc = radius / n; sd[0] *= sc; sd[1] *= sc; }
  }
  T pnt[4];
  T wlen = wrap_circle(pnt, s2, sd, has_side, radius);
  if (wlen < 0) return -1;
  T r0[3], r1[3];
  for (int k = 0; k < 3; ++k) { r0[k] = axis0[k] * pnt[0] + axis1[k] * pnt[1]; r1[k] = axis0[k] * pnt[2] + axis1[k] * pnt[3]; }
  if (type == 5) {
    const T L0 = sqrt((s2[0] - pnt[0]) * (s2[0] - pnt[0]) + (s2[1] - pnt[1]) * (s2[1] - pnt[1]));
    const T L1 = sqrt((s2[2] - pnt[2]) * (s2[2] - pnt[2]) + (s2[3] - pnt[3]) * (s2[3] - pnt[3]));
    const T dz = (p1[2] - p0[2]) / (L0 + wlen + L1);
    r0[2] = p0[2] + dz * L0;
    r1[2] = p0[2] + dz * (L0 + wlen);
    const T height = fabs(r1[2] - r0[2]);
    wlen = sqrt(wlen * wlen + height * height);
  }
  mulmatvec3(wpnt, gmat, r0);
  mulmatvec3(wpnt + 3, gmat, r1);
  for (int k = 0; k < 3; ++k) { wpnt[k] += gpos[k]; wpnt[3 + k] += gpos[k]; }
  return wlen;
}

// moment contribution of one straight tendon segment p0 (on body b0) -> p1 (on body b1).
// Dofs that move both bodies contribute axis x (p1-p0) . u = 0, so only the symmetric
// difference of the two ancestor-dof masks is visited.
// moment contribution of one straight tendon segment p0 (on a body with dof mask m0, tree root
// r0) -> p1 (m1, r1) with unit direction u: the dofs that move exactly one of the two end points
template <typename T, int NC>
DEV void tendon_segment_moment(const Scratch<T, NC>& s, T* Jrow, unsigned long long tmask, unsigned long long m0, int r0,
                               const T* p0, unsigned long long m1, int r1, const T* p1, const T* u, T inv_div) {
  unsigned long long x = (m0 ^ m1) & tmask;
  while (x) {
    const int d = myo_ffsll(x);
    x &= x - 1;
    const int on1 = (int)((m1 >> d) & 1ull);
    const T* p = on1 ? p1 : p0;
    const T* c = S_COM(s) + 3 * (on1 ? r1 : r0);
    const T* cd = s.cdof + 6 * d;
    const T off[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]};
    T t[3];
    cross3(t, cd, off);
    const T col[3] = {cd[3] + t[0], cd[4] + t[1], cd[5] + t[2]};
    const T v = dot3(col, u) * inv_div;
    const int slot = myo_popcll(tmask & ((1ull << d) - 1ull));
    lds_add(Jrow + slot, on1 ? v : -v);             // several path elements (lanes) of one tendon add into its row
  }
}

// HP position (relative to O, like every position of the fp32 stages) of a point given in body coordinates
template <typename T, int NC> DEV void wrap_point_hp(const DevModel<T>& M, const Scratch<T, NC>& s, int body, const HP* local, HP* out) {
  (void)M;
  body_point_hp(s, body, local, out);
  out[0] -= S_ORIGIN(s)[0]; out[1] -= S_ORIGIN(s)[1]; out[2] -= S_ORIGIN(s)[2];
}

// One wrap object as the tendon stage sees it (resolved on the host, see upload: wr_i / wr_p / wr_m /
// wr_mask): a single level of table loads per path element instead of type -> objid -> body -> pos.
template <typename T> struct WrapRec { int type, body, geom, side_body, root, side_root; T pos[3], prm; unsigned long long mask; };
template <typename T> DEV void load_wrap(const DevModel<T>& M, int w, WrapRec<T>& r) {
  r.type = M.wr_i[8 * w]; r.body = M.wr_i[8 * w + 1]; r.geom = M.wr_i[8 * w + 2]; r.side_body = M.wr_i[8 * w + 3];
  r.root = M.wr_i[8 * w + 4]; r.side_root = M.wr_i[8 * w + 5];
  r.pos[0] = M.wr_p[4 * w]; r.pos[1] = M.wr_p[4 * w + 1]; r.pos[2] = M.wr_p[4 * w + 2]; r.prm = M.wr_p[4 * w + 3];
  r.mask = M.wr_mask[w];
}

template <typename T, int NC>
DEVFN void tendon(const DevModel<T>& M_in, const TaskDev& K_in, Scratch<T, NC>& s_in) {
  MYO_BIND_M(T) MYO_BIND_K MYO_BIND_S(T)
  WAVE_FN
  // Three phases instead of one divergent walk per tendon (a wave whose 39 lanes sit at different
  // path elements executes the site branch AND the cylinder-wrap branch every iteration):
  //  A  lane = path element : position of every site / wrap-geom centre (T, relative to O)  -> S_TWP[3w]   (mixed stepper only)
  //  B  lane = geom wrap    : wrap_geom for all sphere/cylinder wraps in lockstep -> wres[7k] = len, 2 points (HP)
  //  C  lane = tendon       : lengths and moment arms from the staged points (cheap, little divergence)
  // Precision: the LENGTH of a tendon is HP end to end — the points it is made of are recomputed from the HP
  // body poses, the wrap solver runs in HP, the segment norms are HP sums — because a Hill muscle turns a length
  // error dl into a force error F0 fpmax dl / (L0 (lmax - 1) / 2): ~1e4 N per metre for MyoSuite's finger muscles,
  // so the 2e-8 m of a float path put 2e-5 relative error into qacc on the first substep (measured on
  // myo_finger_v0).  The MOMENT ARMS (a Jacobian, 1e-7 relative is plenty) come from the T points of phase A.
  // Staging: T points in con[] (dead until the collision stage); HP wrap results behind them, through the
  // limit-row and efc_* arrays (contiguous; constraint rows are only built after this stage).
  T* wp = S_TWP(s);
  auto wres = S_TWRES(s, M.nwrap);
  (void)wres;
  PHASE {
    for (int i = lane; i < M.ntendon * MYO_TJ_MAX; i += 64) S_TENJ_STAGE(s)[i] = 0;     // phase C accumulates into it
    if constexpr (sizeof(T) != sizeof(HP)) {
      for (int w = lane; w < M.nwrap; w += 64) {
        const int body = M.wr_i[8 * w + 1];
        if (body >= 0) {
          const T lp[3] = {M.wr_p[4 * w], M.wr_p[4 * w + 1], M.wr_p[4 * w + 2]};
          body_point(s, body, lp, wp + 3 * w);
        }
      }
    }
  }
  SYNC();
  (void)wp;
}

// phase B of the tendon stage, one geom wrap per lane (called once per 64 wraps from kernel level: a leaf function
// without a loop, so that the HP wrap solver has a register allocation of its own)
template <typename T, int NC>
DEVFN void tendon_wrap_pass(const DevModel<T>& M_in, const TaskDev& K_in, Scratch<T, NC>& s_in, int base) {
  MYO_BIND_M(T) MYO_BIND_K MYO_BIND_S(T)
  WAVE_FN
  auto wres = S_TWRES(s, M.nwrap);
  PHASE {
    const int k = base + lane;
    if (k < M.ngw) {
      const int w = M.gw_elem[k];
      const int body = M.wr_i[8 * w + 1], type = M.wr_i[8 * w], geom = M.wr_i[8 * w + 2], side_body = M.wr_i[8 * w + 3];
      // The wrap solver's branches are geometric predicates (point inside the circle, tangent segments crossing,
      // which of the two tangent pairs) that sit on a knife edge exactly when a wrap engages — the wrap length is
      // continuous there, but an fp32 evaluation of the predicates picks the wrong tangent pair (measured:
      // seg_intersect flips on a 7e-7 m wrap and the path goes the long way round).
      HP gmat[9], bm[9], side[3] = {0, 0, 0}, pts[6], x0[3], x1[3], gp[3];
      quat2mat(bm, S_XQUAT(s) + 4 * body);
      mulmat3(gmat, bm, M.h_wr_m + 12 * w);
      if (side_body >= 0) wrap_point_hp(M, s, side_body, M.h_wr_m + 12 * w + 9, side);
      wrap_point_hp(M, s, M.wr_i[8 * (w - 1) + 1], M.h_wr_p + 4 * (w - 1), x0);
      wrap_point_hp(M, s, M.wr_i[8 * (w + 1) + 1], M.h_wr_p + 4 * (w + 1), x1);
      wrap_point_hp(M, s, body, M.h_wr_p + 4 * w, gp);
      const HP wlen = wrap_geom(pts, x0, x1, gp, gmat, geom_size0_hp(M, K, s, geom), type, side, side_body >= 0);
      wres[7 * k] = wlen;
      for (int e = 0; e < 6; ++e) wres[7 * k + 1 + e] = pts[e];
    }
  }
  if constexpr (Scratch<T, NC>::SPILL) { SYNC_G(); } else { SYNC(); }      // (SPILL: the wrap results went to the big workspace in global memory)
}

// phase C of the tendon stage: lengths and moment arms.  One lane per PATH ELEMENT (site -> site, or site -> wrap geom ->
// site; the walk along each tendon is resolved on the host: te_i / te_div), 64 elements per call from kernel level (a
// loop-free leaf, like the wrap pass).  A lane computes the HP length of its element into a staging slot and adds the
// element's moment-arm contributions into its tendon's row of ten_J with LDS adds; tendon_length_sums then adds the
// element lengths of each tendon in path order.  (The first version walked each tendon in one lane — 39 lanes chaining
// ~5 elements of fp64 site positions, square roots and moment gathers: 15.6 k of the substep's 229 k cycles.)
#define S_TELEN(s) (reinterpret_cast<HP*>((s).H + MYO_NB_MAX * 10))   /* behind cinert; crb / qfrc_* come after the tendon stage */
template <typename T, int NC>
DEVFN void tendon_element_pass(const DevModel<T>& M_in, const TaskDev& K_in, Scratch<T, NC>& s_in, int base) {
  MYO_BIND_M(T) MYO_BIND_K MYO_BIND_S(T)
  WAVE_FN
  (void)K;
  T* wp = S_TWP(s);
  auto wres = S_TWRES(s, M.nwrap);
  PHASE {
    const int e = base + lane;
    if (e < M.nte) {
      const int i0 = M.te_i[4 * e], iend = M.te_i[4 * e + 1], ig = M.te_i[4 * e + 2], t = M.te_i[4 * e + 3];
      const unsigned long long tmask = M.tendon_dofmask[t];
      T* J = S_TENJ_STAGE(s) + t * MYO_TJ_MAX;
      const T inv_div = 1 / M.te_div[e];
      const int is_geom = ig >= 0;
      WrapRec<T> w0, w1, we;
      load_wrap(M, i0, w0);
      load_wrap(M, iend, we);
      load_wrap(M, is_geom ? ig : i0, w1);
      HP q0[3], q1[3];
      wrap_point_hp(M, s, w0.body, M.h_wr_p + 4 * i0, q0);
      wrap_point_hp(M, s, we.body, M.h_wr_p + 4 * iend, q1);
      // straight segments of this path element: site -> site, or site -> wrap point, (arc), wrap point -> site.
      // Selected with scalars (no run-time indexed local arrays: they would live in private memory).
      // (fp64 stepper: the HP end points themselves — body_point and wrap_point_hp are the same arithmetic there, O = 0 — no staged copy)
      T p0[3], x1[3];
      if constexpr (sizeof(T) == sizeof(HP)) { for (int k = 0; k < 3; ++k) { p0[k] = q0[k]; x1[k] = q1[k]; } (void)wp; }
      else { for (int k = 0; k < 3; ++k) { p0[k] = wp[3 * i0 + k]; x1[k] = wp[3 * iend + k]; } }
      HP wlen = -1, h0[3] = {0, 0, 0}, h1[3] = {0, 0, 0};
      if (is_geom) {
        const auto r = wres + 7 * M.wr_i[8 * ig + 6];
        wlen = r[0];
        for (int k = 0; k < 3; ++k) { h0[k] = r[1 + k]; h1[k] = r[4 + k]; }
      }
      const T g0[3] = {(T)h0[0], (T)h0[1], (T)h0[2]}, g1[3] = {(T)h1[0], (T)h1[1], (T)h1[2]};
      const bool wrapped = wlen >= 0;
      HP len = 0;
      for (int sg = 0; sg < 2; ++sg) {
        if (sg == 1 && !wrapped) break;
        const bool to_wrap = (sg == 0) && wrapped;          // this segment ends on the wrap geom
        T pa[3], pb[3];
        HP da[3];
        for (int k = 0; k < 3; ++k) {
          pa[k] = sg == 0 ? p0[k] : g1[k]; pb[k] = to_wrap ? g0[k] : x1[k];
          da[k] = (to_wrap ? h0[k] : q1[k]) - (sg == 0 ? q0[k] : h1[k]);
        }
        const unsigned long long ma = sg == 0 ? w0.mask : w1.mask, mb = to_wrap ? w1.mask : we.mask;
        const int ra = sg == 0 ? w0.root : w1.root, rb = to_wrap ? w1.root : we.root;
        const int ba = sg == 0 ? w0.body : w1.body, bb = to_wrap ? w1.body : we.body;
        T dif[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
        const T dn = norm3(dif);
        len += norm3(da) * (HP)inv_div;
        if (ba != bb && dn > MYO_MINVAL) {
          dif[0] /= dn; dif[1] /= dn; dif[2] /= dn;
          tendon_segment_moment(s, J, tmask, ma, ra, pa, mb, rb, pb, dif, inv_div);
        }
        if (to_wrap) len += wlen * (HP)inv_div;
      }
      S_TELEN(s)[e] = len;
    }
  }
  SYNC();
}
template <typename T, int NC>
DEVFN void tendon_length_sums(const DevModel<T>& M_in, Scratch<T, NC>& s_in) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  WAVE_FN
  PHASE {
    const int t = lane;
    if (t < M.ntendon) {
      const int e0 = M.tendon_eadr[t], n = M.tendon_enum[t];
      HP len = 0;
      for (int k = 0; k < n; ++k) len += S_TELEN(s)[e0 + k];
      S_TEN_LENGTH(s)[t] = len;
    }
    if constexpr (sizeof(T) == sizeof(HP)) {      // the finished moment arms leave LDS: [slot][tendon]
      for (int o = lane; o < MYO_TJ_MAX * MYO_NT_MAX; o += 64) {
        const int slot = o / MYO_NT_MAX, tt = o - slot * MYO_NT_MAX;
#if defined(MYO_TENJ_LDS) && defined(MYO_WARM_LDS)
        if (tt < M.ntendon) s.ten_J_[o] = (double)S_TENJ_STAGE(s)[tt * MYO_TJ_MAX + slot];
#else
        if (tt < M.ntendon) ((GPTR(double))s.tenj_g)[o] = (double)S_TENJ_STAGE(s)[tt * MYO_TJ_MAX + slot];
#endif
      }
    }
  }
  SYNC_G();       // (the moment arms / SPILL's tendon lengths went to global memory: other lanes read them)
}

// ------------------------------------------------------------------------------------------
// P5: composite rigid body inertia -> tree-sparse M (mj_crb)
template <typename T, int NC>
DEVFN void crb(const DevModel<T>& M_in, Scratch<T, NC>& s_in) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  WAVE_FN
  PHASE {
    const int b = lane;
    if (b > 0 && b < M.nbody) {
      T acc[10];
      for (int k = 0; k < 10; ++k) acc[k] = 0;
      const unsigned long long sub = M.body_submask[b];
      // all bodies, membership as a select (wave-uniform reads, all in flight; see the RNE bias loop)
#pragma unroll 4
      for (int c = 1; c < M.nbody; ++c) {
        const T in = ((sub >> c) & 1ull) ? (T)1 : (T)0;
        for (int k = 0; k < 10; ++k) acc[k] += in * S_CINERT(s)[10 * c + k];
      }
      for (int k = 0; k < 10; ++k) S_CRB(s)[10 * b + k] = acc[k];
    }
  }
  SYNC();
  PHASE {
    // (row, column, body) of each entry packed in one table word, fetched for all of a lane's entries up front
    constexpr int NE = (MYO_NM_MAX + 63) / 64;
    int pk[NE];
#pragma unroll
    for (int q = 0; q < NE; ++q) pk[q] = M.M_pk[(lane + 64 * q) < MYO_NM_MAX ? (lane + 64 * q) : 0];
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      const int e = lane + 64 * q;
      if (e < M.nM) {
        const int i = pk[q] & 255, j = (pk[q] >> 8) & 255, bi = pk[q] >> 16;
        T buf[6];
        mul_inert_vec(buf, S_CRB(s) + 10 * bi, s.cdof + 6 * i);
        T v = 0;
        for (int k = 0; k < 6; ++k) v += s.cdof[6 * j + k] * buf[k];
        if (i == j) v += M.dof_armature[i];
        s.qM[e] = v;
      }
    }
  }
  SYNC();
}

// out = M * v   (lanes = dofs; static CSR pattern of the symmetric tree-sparse matrix)
template <typename T, int NC>
DEV void mul_M(const DevModel<T>& M_in, const Scratch<T, NC>& s_in, LREF(T) out_r, LCREF(T) v_r) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  T* out = LPTR(T, out_r); const T* v = LPTR(const T, v_r);
  WAVE_FN_K
  PHASE {
    const int i = lane;
    if (i < M.nv) {
      // the row's (qM index, column) pairs arrive in one go (3 x 16-byte loads), not one dependent
      // table load per non-zero
      unsigned w[MYO_MV_ROW / 2];
#pragma unroll
      for (int q = 0; q < MYO_MV_ROW / 2; ++q) w[q] = (unsigned)M.mv_pack[i * (MYO_MV_ROW / 2) + q];
      const int len = M.mv_len[i];
      T acc = 0;
#pragma unroll
      for (int k = 0; k < MYO_MV_ROW; ++k) {
        // unconditional LDS reads (padding entries are 0 -> qM[0], v[0], valid) and a select: no exec-mask
        // branch per entry, so all the reads are in flight together
        const unsigned ent = (k & 1) ? (w[k / 2] >> 16) : (w[k / 2] & 0xffffu);
        const T q = s.qM[ent >> 6], x = v[ent & 63u];
        acc += (k < len) ? q * x : (T)0;
      }
      out[i] = acc;
    }
  }
  SYNC();
}

// packed dense H <- M (+ diag).  perm = 1: rows in the Newton system's order (hperm; M_pkh holds the packed offsets)
template <typename T, int NC>
DEV void load_H_from_M(const DevModel<T>& M_in, Scratch<T, NC>& s_in, const T* diag_add, T diag_scale, int perm = 0) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  WAVE_FN_K
  PHASE {
    // flat clear with 16-byte stores
    struct alignas(16) Q4 { T a, b, c, d; };
    Q4* h4 = reinterpret_cast<Q4*>(s.H);
#pragma unroll
    for (int k = 0; k < (MYO_H_SIZE / 4 + 63) / 64; ++k) { const int e = lane + 64 * k; if (e < MYO_H_SIZE / 4) h4[e] = Q4{0, 0, 0, 0}; }
  }
  SYNC();
  PHASE {
    // identity on the rows that hold no dof
    if (lane < MYO_NV_MAX && (perm ? (int)((M.arrow_pad >> lane) & 1ull) : (lane >= M.nv))) s.H[MYO_HIDX(lane, lane)] = 1;
    constexpr int NE = (MYO_NM_MAX + 63) / 64;
    int pk[NE], ph[NE];
#pragma unroll
    for (int q = 0; q < NE; ++q) { pk[q] = M.M_pk[(lane + 64 * q) < MYO_NM_MAX ? (lane + 64 * q) : 0]; ph[q] = M.M_pkh[(lane + 64 * q) < MYO_NM_MAX ? (lane + 64 * q) : 0]; }
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      const int e = lane + 64 * q;
      if (e < M.nM) {
        const int i = pk[q] & 255, j = (pk[q] >> 8) & 255;
        T v = s.qM[e];
        if (i == j && diag_add) v += diag_scale * diag_add[i];
        s.H[perm ? ph[q] : MYO_HIDX(i, j)] = v;
      }
    }
  }
  SYNC();
}

// in-place Cholesky of the packed lower triangle (right-looking; lanes tile the trailing block)
template <typename T, int NC>
DEV void chol_factor(Scratch<T, NC>& s, int n) {
  WAVE_FN_K
  for (int k = 0; k < n; ++k) {
    PHASE {
      // every lane derives the pivot; lanes k+1.. scale their column entry
      T piv = s.H[MYO_HIDX(k, k)];
      if (piv < MYO_MINVAL) piv = MYO_MINVAL;
      const T dk = sqrt(piv);
      const int i = k + 1 + lane;
      if (i < n) s.H[MYO_HIDX(i, k)] = s.H[MYO_HIDX(i, k)] / dk;
      if (lane == 63) s.H[MYO_HIDX(k, k)] = dk;  // last lane: never a column lane (n <= 36 < 64)
    }
    SYNC();
    PHASE {
      const int li = lane >> 3, lj = lane & 7;
      for (int i = k + 1 + li; i < n; i += 8) {
        const T lik = s.H[MYO_HIDX(i, k)];
        for (int j = k + 1 + lj; j <= i; j += 8) s.H[MYO_HIDX(i, j)] -= lik * s.H[MYO_HIDX(j, k)];
      }
    }
    SYNC();
  }
}

// solve L L' x = b in place (x in scratch vector)
template <typename T, int NC>
DEV void chol_solve(Scratch<T, NC>& s, T* x, int n) {
  WAVE_FN_K
  for (int k = 0; k < n; ++k) {
    PHASE {
      const T xk = x[k] / s.H[MYO_HIDX(k, k)];
      const int i = k + 1 + lane;
      if (i < n) x[i] -= s.H[MYO_HIDX(i, k)] * xk;
      if (lane == 63) x[k] = xk;
    }
    SYNC();
  }
  for (int k = n - 1; k >= 0; --k) {
    PHASE {
      const T xk = x[k] / s.H[MYO_HIDX(k, k)];
      const int i = lane;
      if (i < k) x[i] -= s.H[MYO_HIDX(k, i)] * xk;
      if (lane == 63) x[k] = xk;
    }
    SYNC();
  }
}

#ifndef MYO_EMU
// ---- blocked Cholesky on the matrix cores + register-resident substitutions (gfx950 build).
// FACTOR (right-looking, panels of four columns = the K of v_mfma_*_16x16x4): the matrix lives in MFMA accumulator
// tiles (16x16, lower tiles only; 36 dofs -> 3x3 tiles, the 24-wide variant 2x2).  Per panel: the four panel
// columns leave the accumulator layout through a small LDS stage and come back row-per-lane (every lane gets the
// four entries of "its" row of each tile row), the 4x4 diagonal block is broadcast with v_readlane and factored
// redundantly by all lanes, every lane finishes its rows of the panel by a four-step substitution, the finished L
// columns go to H (packed lower triangle) for the substitution phase, and the trailing matrix is updated with one
// v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64 per remaining lower tile (operands: the panel rows, the same
// register as A and as B).  The one-column algorithm this replaces exchanged a column through LDS 36 times per
// factorisation and spent 53 k of a substep's 241 k cycles (mixed) / 156 k of 425 k (fp64) in the three solves.
// The arithmetic per element is a k-ordered FMA chain as in the LDS version the MYO_EMU build keeps.
// SOLVE: lane i holds row i of L in VGPRs; the substitutions broadcast with v_readlane.
template <typename T> __device__ __forceinline__ T lane_bcast(T v, int src);
template <> __device__ __forceinline__ float lane_bcast<float>(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
template <> __device__ __forceinline__ double lane_bcast<double>(double v, int src) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float myo_rsqrt(float v) { return __frsqrt_rn(v); }
__device__ __forceinline__ double myo_rsqrt(double v) { return rsqrt(v); }
#define MYO_OPAQUE_LANE(v) int v = (int)threadIdx.x; asm volatile("" : "+v"(v));
// accumulator layouts of the two 16x16x4 MFMAs (cdna_hip_programming.md §3): column = lane & 15 in both,
// row = 4 (lane >> 4) + reg for f32, (lane >> 4) + 4 reg for f64; A / B operand: [lane & 15][k = lane >> 4]
template <typename T> struct MyoMfma;
template <> struct MyoMfma<float> {
  typedef float V4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ V4 mma(float a, float b, V4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
};
template <> struct MyoMfma<double> {
  typedef double V4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ V4 mma(double a, double b, V4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <typename T, int N, int NC>
__device__ __noinline__ void chol_factor_solve_reg(int x_r, int n) {
  static_assert(N % 4 == 0 && N <= MYO_NV_MAX, "row vectors are loaded 4 at a time");
  typedef T V2 __attribute__((ext_vector_type(2)));
  typedef T V4 __attribute__((ext_vector_type(4)));
  typedef MyoMfma<T> MM;
  constexpr int NT = (N + 15) / 16;                  // tiles per side
  Scratch<T, NC>& s = *reinterpret_cast<Scratch<T, NC>*>(myo_lds);
  T* x = LPTR(T, x_r);
  const int lane = threadIdx.x;
  const int lc = lane & 15, lq = lane >> 4;
  const int row = lane < N ? lane : N - 1;          // lanes >= N shadow the last row and never store
  V2 a2[N / 2];
  T b = (lane < n) ? x[lane] : (T)0;
  T invd = 1;                                        // 1 / L[lane][lane]
  // ---- factor.  fp64: blocked on the matrix cores (below).  fp32 keeps the one-column register algorithm: measured on
  // the same build, the blocked MFMA factorisation is 1.5x FASTER than the register version in fp64 (103 k instead of
  // 156 k cycles per substep for the three solves: half the registers, no spills, 31 MFMAs instead of ~1300 v_fma_f64)
  // and 1.7x SLOWER in fp32 (90 k instead of 54 k): there v_pk_fma_f32 already does two entries per instruction, the
  // column exchange is hidden by software pipelining, and what remains of the blocked version is its serial chain per
  // panel (stage round trip, 10 v_readlane, four dependent rsqrt, 40-cycle MFMA latency) nine times per factorisation.
#ifdef MYO_CHOL_NO_MFMA      /* diagnostic build: the fp64 factorisation without the matrix cores (the fp32 stepper's register algorithm) */
  constexpr bool reg_alg = true;
#else
  constexpr bool reg_alg = sizeof(T) == 4;
#endif
  if constexpr (reg_alg) {
    {
      // row `row` of H (packed, MYO_HIDX): N/4 vectors from the row's first entry; what lies beyond the row's own
      // 4(row/4 + 1) entries belongs to later rows (always inside H) and lands above the diagonal
      const V4* hr = reinterpret_cast<const V4*>(s.H + myo_hrow(row));
#pragma unroll
      for (int q = 0; q < N / 4; ++q) { const V4 v = hr[q]; a2[2 * q] = V2{v.x, v.y}; a2[2 * q + 1] = V2{v.z, v.w}; }
    }
    // fp32: one column per step.  (The two-column scheme below is 14 % faster end to end in fp64, where the
    // factorisation is a quarter of the kernel, but measured 2.6 % SLOWER in fp32: its serial 2x2 pivot chain
    // outweighs the saved round trips once the per-column exchange is already pipelined.)
    // (Broadcasting column k with v_readlane into SGPR pairs instead — no LDS round trip on the chain, same bits — measured
    // 1 % SLOWER end to end, 2.788 vs 2.761 ms at 4096 envs: 630 v_readlane + their SGPR hazard nops cost more issue slots
    // than the LDS latency they remove costs a SIMD that has a second wave to run.)
    // Column exchange through LDS, software-pipelined: while the trailing update of step k is still
    // running, column k+1 (updated first) is already written and read back for step k+1.  Two buffers
    // alternate; a workgroup is one wavefront, so its LDS operations execute in program order and no
    // barrier is needed between the write and the reads.
    T* colbuf0 = s.efc_jv;                             // >= 128 entries, free while a system is solved
    T* colbuf1 = s.efc_jv + 64;
    V2 cb2[2][N / 2];                                 // column k lives in cb2[k & 1] (compile-time parity: no copies between steps)
    colbuf0[lane] = a2[0].x;
    {
      const V2* cb = reinterpret_cast<const V2*>(colbuf0);
#pragma unroll
      for (int p = 0; p < N / 2; ++p) cb2[0][p] = cb[p];
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
      V2* const c2 = cb2[k & 1];
      V2* const c2n = cb2[(k + 1) & 1];
      const T ak = (k & 1) ? a2[k / 2].y : a2[k / 2].x;
      T akk = (k & 1) ? c2[k / 2].y : c2[k / 2].x;
      akk = akk < MYO_MINVAL ? MYO_MINVAL : akk;
      const T inv = myo_rsqrt(akk);
      const T lik = ak * inv;                           // lane k: akk / sqrt(akk) = sqrt(akk)
      const V2 m = V2{-lik * inv, -lik * inv};
      const int p1 = (k + 1) / 2;
      if (k + 1 < N) {
        a2[p1] = __builtin_elementwise_fma(m, c2[p1], a2[p1]);      // the pair holding column k+1 goes first
        T* nb = ((k + 1) & 1) ? colbuf1 : colbuf0;
        nb[lane] = ((k + 1) & 1) ? a2[p1].y : a2[p1].x;
        const V2* cb = reinterpret_cast<const V2*>(nb);
#pragma unroll
        for (int p = p1; p < N / 2; ++p) c2n[p] = cb[p];
      }
#pragma unroll
      for (int p = p1 + 1; p < N / 2; ++p) a2[p] = __builtin_elementwise_fma(m, c2[p], a2[p]);
      if (k & 1) a2[k / 2].y = lik; else a2[k / 2].x = lik;
      {
        MYO_OPAQUE_LANE(l)
        if (k == l) invd = inv;                        // 1/sqrt(pivot) = 1/L[k][k]
      }
    }
    // transpose through LDS: lane i needs column i of L for the backward substitution
    if (lane < N) {
      V4* hw = reinterpret_cast<V4*>(s.H + myo_hrow(lane));
#pragma unroll
      for (int q = 0; q < N / 4; ++q) if (q <= (lane >> 2)) hw[q] = V4{a2[2 * q].x, a2[2 * q].y, a2[2 * q + 1].x, a2[2 * q + 1].y};
    }
    SYNC();
  } else {
    {
      // LDS pointers of this function: taken ONCE and made opaque.  In a non-kernel function the address of the dynamic
      // LDS block is a look-up in llvm.amdgcn.dynlds.offset.table (s_getpc + s_load, or a global load when the
      // compiler re-derives it inside a divergent region: measured 26 look-ups and +22 k cycles in one call of this
      // function before the pointers were pinned)
      typedef __attribute__((address_space(3))) T* lds_t;
      lds_t stage = (lds_t)S_SOLVE_STAGE(s);           // [NT * 16 rows][4] + 64 dump entries; bvec, efc_jv and efc_force are free while a system is solved
      lds_t Hp = (lds_t)s.H;
      asm volatile("" : "+v"(stage), "+v"(Hp));
      static_assert(NT * 16 * 4 + 64 <= MYO_NB_MAX * 6 + 2 * (MYO_NLIM_MAX + 4 * NC) && 4 * 64 <= MYO_NB_MAX * 6 + 2 * (MYO_NLIM_MAX + 4 * NC), "panel stage and dump area fit in bvec + efc_jv + efc_force");
      typedef Scratch<T, NC> ScratchT;
      static_assert(offsetof(ScratchT, efc_jv) - offsetof(ScratchT, bvec) == MYO_NB_MAX * 6 * sizeof(T) &&
                    offsetof(ScratchT, efc_force) - offsetof(ScratchT, efc_jv) == (MYO_NLIM_MAX + 4 * NC) * sizeof(T), "bvec, efc_jv and efc_force are contiguous");
      typename MM::V4 acc[NT * (NT + 1) / 2];
      // symmetric fill of the lower tiles from the packed lower triangle; identity beyond N
  #pragma unroll
      for (int I = 0; I < NT; ++I)
  #pragma unroll
        for (int J = 0; J <= I; ++J)
  #pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * I + MM::crow(lane, r), j = 16 * J + lc;
            const int hi = i > j ? i : j, lo = i > j ? j : i;
            // (no exec-mask branches in this function: inside a divergent region the compiler re-derives the base of
            // the dynamic LDS block — a table lookup in a non-kernel function, ~500 cycles each; measured +22 k cycles)
            const T h = Hp[hi < N ? MYO_HIDX(hi, lo) : 0];
            acc[I * (I + 1) / 2 + J][r] = hi < N ? h : ((i == j) ? (T)1 : (T)0);
          }
  #pragma unroll
      for (int kb = 0; kb < N / 4; ++kb) {
        constexpr int dummy = 0; (void)dummy;
        const int c0 = 4 * kb, J = c0 / 16, jc = c0 % 16;
        // 1. the four panel columns: accumulator layout -> stage[row][k]  (the other lanes store into a dump area)
        {
          const bool mine = lc >= jc && lc < jc + 4;
  #pragma unroll
          for (int I = J; I < NT; ++I)
  #pragma unroll
            for (int r = 0; r < 4; ++r)
              stage[mine ? (16 * I + MM::crow(lane, r)) * 4 + (lc - jc) : NT * 64 + lane] = acc[I * (I + 1) / 2 + J][r];
        }
        SYNC();
        // 2. row-per-lane: the four panel entries of row 16 I + lc (the same in all four lane quarters)
        T xr[NT][4];
  #pragma unroll
        for (int I = J; I < NT; ++I) {
          typedef __attribute__((address_space(3))) const V4* lds_v4;
          const V4 v = *(lds_v4)(stage + (16 * I + lc) * 4);
          xr[I][0] = v.x; xr[I][1] = v.y; xr[I][2] = v.z; xr[I][3] = v.w;
        }
        // diagonal block (rows c0 .. c0+3 sit in tile row J at lanes jc .. jc+3), factored by every lane
        T d00 = lane_bcast<T>(xr[J][0], jc);
        const T d10 = lane_bcast<T>(xr[J][0], jc + 1), d20 = lane_bcast<T>(xr[J][0], jc + 2), d30 = lane_bcast<T>(xr[J][0], jc + 3);
        T d11 = lane_bcast<T>(xr[J][1], jc + 1);
        const T d21 = lane_bcast<T>(xr[J][1], jc + 2), d31 = lane_bcast<T>(xr[J][1], jc + 3);
        T d22 = lane_bcast<T>(xr[J][2], jc + 2);
        const T d32 = lane_bcast<T>(xr[J][2], jc + 3);
        T d33 = lane_bcast<T>(xr[J][3], jc + 3);
        d00 = d00 < MYO_MINVAL ? MYO_MINVAL : d00;
        const T i0 = myo_rsqrt(d00);
        const T l10 = d10 * i0, l20 = d20 * i0, l30 = d30 * i0;
        d11 -= l10 * l10; d11 = d11 < MYO_MINVAL ? MYO_MINVAL : d11;
        const T i1 = myo_rsqrt(d11);
        const T l21 = (d21 - l20 * l10) * i1, l31 = (d31 - l30 * l10) * i1;
        d22 -= l20 * l20 + l21 * l21; d22 = d22 < MYO_MINVAL ? MYO_MINVAL : d22;
        const T i2 = myo_rsqrt(d22);
        const T l32 = (d32 - l30 * l20 - l31 * l21) * i2;
        d33 -= l30 * l30 + l31 * l31 + l32 * l32; d33 = d33 < MYO_MINVAL ? MYO_MINVAL : d33;
        const T i3 = myo_rsqrt(d33);
        // 3. every row of the panel: L21 = A21 L11^-T  (rows of the diagonal block come out as L11 itself)
  #pragma unroll
        for (int I = J; I < NT; ++I) {
          const int i = 16 * I + lc;
          T y0 = xr[I][0] * i0;
          T y1 = (xr[I][1] - y0 * l10) * i1;
          T y2 = (xr[I][2] - y0 * l20 - y1 * l21) * i2;
          T y3 = (xr[I][3] - y0 * l30 - y1 * l31 - y2 * l32) * i3;
          if (i == c0) { y1 = 0; y2 = 0; y3 = 0; }      // above the diagonal inside the block
          if (i == c0 + 1) { y2 = 0; y3 = 0; }
          if (i == c0 + 2) y3 = 0;
          if (i < c0) { y0 = 0; y1 = 0; y2 = 0; y3 = 0; }   // rows factored earlier take no part
          // finished L columns -> H (packed rows; c0 is a multiple of 4 and so is every row start: 16-byte store)
          // (lanes that hold no row of L store into the dump area: stage rows are free until the next panel)
          {
            const bool mine = lq == 0 && i >= c0 && i < N;
            typedef __attribute__((address_space(3))) V4* lds_v4w;
            lds_t const dst = mine ? Hp + myo_hrow(i) + c0 : stage + 4 * lane;
            *(lds_v4w)dst = V4{y0, y1, y2, y3};
          }
          xr[I][0] = y0; xr[I][1] = y1; xr[I][2] = y2; xr[I][3] = y3;
        }
        // 4. trailing update  A22 -= L21 L21'  : one MFMA per remaining lower tile, operand = the panel row of this lane
        T op[NT];
  #pragma unroll
        for (int I = J; I < NT; ++I) op[I] = lq == 0 ? xr[I][0] : (lq == 1 ? xr[I][1] : (lq == 2 ? xr[I][2] : xr[I][3]));
        if (kb + 1 < N / 4) {
  #pragma unroll
          for (int I = J; I < NT; ++I)
  #pragma unroll
            for (int Jt = J; Jt <= I; ++Jt) acc[I * (I + 1) / 2 + Jt] = MM::mma(-op[I], op[Jt], acc[I * (I + 1) / 2 + Jt]);
        }
        SYNC();                                 // the stage is rewritten by the next panel
      }
    }
    {
      // row `row` of L from H
      const V4* hr = reinterpret_cast<const V4*>(s.H + myo_hrow(row));
#pragma unroll
      for (int q = 0; q < N / 4; ++q) { const V4 v = hr[q]; a2[2 * q] = V2{v.x, v.y}; a2[2 * q + 1] = V2{v.z, v.w}; }
    }
    invd = (T)1 / s.H[MYO_HIDX(row, row)];
  }
  // forward substitution  L y = b.  Lane j's b is complete when step j starts (every k < j has been subtracted); it keeps
  // the unscaled value and y_j = b_j / L_jj is formed for the broadcast only — the lane's own scaling happens once, after
  // the loop (the same product, so the same bits), which takes a compare + select out of each of the N serial steps
#pragma unroll
  for (int j = 0; j < N; ++j) {
    MYO_OPAQUE_LANE(l)
    const T aj = (j & 1) ? a2[j / 2].y : a2[j / 2].x;
    const T yj = lane_bcast<T>(b * invd, j);
    b = (l > j) ? b - aj * yj : b;
    __builtin_amdgcn_sched_barrier(0);
  }
  b *= invd;
  // lane i needs column i of L for the backward substitution (L is in H)
  T c[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    MYO_OPAQUE_LANE(l)
    const T v = s.H[myo_hrow(j) + row];               // L[j][row]; lanes beyond row j's storage read a later row and drop it
    c[j] = (j > l) ? v : (T)0;
  }
  // backward substitution  L' x = y
#pragma unroll
  for (int j = N - 1; j >= 0; --j) {
    const T xj = lane_bcast<T>(b * invd, j);
    b = b - c[j] * xj;                                // c[j] = 0 for lanes >= j: they keep their (unscaled) value
    __builtin_amdgcn_sched_barrier(0);
  }
  b *= invd;
  if (lane < n) x[lane] = b;
  SYNC();
}
#endif

// `lead` = number of leading dofs that are coupled; dofs >= lead have a diagonal-only row in the
// matrix at hand (free bodies whose inertial frame is the body frame: the two balls), so their
// solve is a division.  M-only solves and contact-free Newton steps use the 24-wide variant.
#define MYO_CHOL_SMALL 24
template <typename T, int NC>
DEV void chol_factor_solve(Scratch<T, NC>& s, T* x, int n, int lead) {
#ifdef MYO_EMU
  (void)lead;
  chol_factor(s, n);
  chol_solve(s, x, n);
#else
  if (lead < n && lead <= MYO_CHOL_SMALL) {
    const int lane = threadIdx.x;
    if (lane >= lead && lane < n) x[lane] = x[lane] / s.H[MYO_HIDX(lane, lane)];
    chol_factor_solve_reg<T, MYO_CHOL_SMALL, NC>(LOFF(s, x), lead);
  } else {
    chol_factor_solve_reg<T, MYO_NV_MAX, NC>(LOFF(s, x), n);
  }
#endif
}

#include "myo_sparse_ldl.h"
#include "myo_arrow_chol.h"

// x <- (M + [damped] h diag(b))^-1 x : the two M-only systems of a substep.  Tree-sparse L'DL where the model's tables
// provide it, else the dense Cholesky of the Newton step.
template <typename T, int NC>
DEV void solve_M(const DevModel<T>& M_in, Scratch<T, NC>& s_in, T* x, int damped) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  // measured at 4096 envs (tools/dev/kab.py, MYO_DENSE_MSOLVE=1 for the dense side): fp64 4.345 -> 4.029 ms per env step with the sparse
  // solves; the mixed stepper 2.229 -> 2.289 ms — its 24-wide register Cholesky (packed fp32 FMAs, pipelined column exchange) is
  // already shorter than the ~23 LDS round trips of the level schedule, so fp32 keeps the dense path
  if (sizeof(T) == sizeof(HP) && M.ld_nsq >= 0) {
    ldl_factor_solve(M, s, LOFF(s, x), damped);
  } else {
    load_H_from_M(M, s, damped ? (const T*)M.dof_damping : (const T*)0, damped ? M.timestep : (T)0);
    chol_factor_solve(s, x, M.nv, M.nlead);
  }
}

// ------------------------------------------------------------------------------------------
// P6: collision (narrow phase per candidate pair; lanes = pairs).  HP and world coordinates: the contact
// distance decides activation (dist < margin, discontinuous in MuJoCo's soft-contact model) and the spring
// term, and it is a 1e-4 m difference of 1.5 m coordinates.  What leaves this stage for the fp32 stages is
// the distance itself, the normal, and the contact point relative to O.
struct ContactTmp { HP dist[2], pos[6], nrm[6]; int n; };

// normal f[0..2] -> unit normal; returns the record's `tinv`: the reciprocal norm of (e - (e . n) n), e = y (returned positive) unless the
// normal is within 60 degrees of y, then z (returned negative).  The norm is >= 0.5 either way.  con_frame rebuilds the tangent from it.
template <typename T> DEV T make_frame(T* f) {
  normalize3(f);
  const bool z = !(f[1] < (T)0.5 && f[1] > (T)-0.5);
  const T t = z ? f[2] : f[1];
  const T u[3] = {(T)0 - t * f[0], (z ? (T)0 : (T)1) - t * f[1], (z ? (T)1 : (T)0) - t * f[2]};
  const T i = 1 / norm3(u);
  return z ? -i : i;
}
// second tangent of a contact frame (what make_frame's caller used to store behind the first): normal x tangent 1
template <typename T> DEV void con_t2(const T* f, T* t2) { cross3(t2, f, f + 3); }
DEV int sphere_sphere(HP* dist, HP* pos, HP* n, const HP* c1, HP r1, const HP* c2, HP r2, HP margin) {
  const HP dif[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]};
  const HP cd = norm3(dif);
  if (cd - r1 - r2 > margin) return 0;
  if (cd < (HP)1e-15) { n[0] = 1; n[1] = 0; n[2] = 0; } else { { const HP i_cd = (HP)1 / cd; n[0] = dif[0] * i_cd; n[1] = dif[1] * i_cd; n[2] = dif[2] * i_cd; } }
  *dist = cd - r1 - r2;
  for (int k = 0; k < 3; ++k) pos[k] = c1[k] + n[k] * (r1 + (HP)0.5 * (*dist));
  return 1;
}
DEV void seg_nearest(HP* out, const HP* c, const HP* axis, HP half, const HP* p) {
  HP t = (p[0] - c[0]) * axis[0] + (p[1] - c[1]) * axis[1] + (p[2] - c[2]) * axis[2];
  t = tclamp(t, -half, half);
  for (int k = 0; k < 3; ++k) out[k] = c[k] + t * axis[k];
}

template <typename T, int NC> DEV HP geom_size2_hp(const DevModel<T>& M, const TaskDev& K, const Scratch<T, NC>& s, int g) {
  const HP v = M.h_geom_size[3 * g + 2];      // (the reference adds the size delta to ALL sizes of the group's last three geoms)
  return (g < K.objg_gidn && g >= K.objg_gidn - 3 && g >= K.objg_gid0) ? v + s.ball_size[0] : v;
}
// world poses of the two geoms of a pair (HP), their (per-env) sizes, and mj_collideGeoms' bounding-sphere filter, EXACT (the
// MODEL's rbound): with P2's per-episode ball radius above the nominal one (the reference never refreshes geom_rbound,
// baoding.py:586-604) this test, not the narrow phase, decides when a ball contact switches on — the callers' fp32
// pre-filter only rejects pairs that are clearly apart.  Returns 0 when the filter rejects the pair.
template <typename T, int NC>
DEV int pair_poses(const DevModel<T>& M, const TaskDev& K, const Scratch<T, NC>& s, int g1, int g2, HP margin, HP* p1, HP* p2, HP* R1, HP* R2,
                   HP* s1, HP* s2) {
  const int b1 = M.geom_bodyid[g1], b2 = M.geom_bodyid[g2];
  HP B1[9], B2[9], l1[3], l2[3];
  geom_lpos_hp(M, K, s, g1, l1);
  geom_lpos_hp(M, K, s, g2, l2);
  quat2mat(B1, S_XQUAT(s) + 4 * b1);
  quat2mat(B2, S_XQUAT(s) + 4 * b2);
  mulmatvec3(p1, B1, l1);
  mulmatvec3(p2, B2, l2);
  for (int k = 0; k < 3; ++k) { p1[k] += S_XPOS(s)[3 * b1 + k]; p2[k] += S_XPOS(s)[3 * b2 + k]; }
  mulmat3(R1, B1, M.h_geom_mat + 9 * g1);
  mulmat3(R2, B2, M.h_geom_mat + 9 * g2);
  s1[0] = geom_size0_hp(M, K, s, g1); s1[1] = geom_size1_hp(M, K, s, g1); s1[2] = geom_size2_hp(M, K, s, g1);
  s2[0] = geom_size0_hp(M, K, s, g2); s2[1] = geom_size1_hp(M, K, s, g2); s2[2] = geom_size2_hp(M, K, s, g2);
  const HP rb1 = M.h_geom_rbound[g1], rb2 = M.h_geom_rbound[g2];
  const HP df[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  const HP bound = rb1 + rb2 + margin;
  return !(rb1 > 0 && rb2 > 0 && dot3(df, df) > bound * bound);
}

template <typename T, int NC>
DEV void collide_pair(const DevModel<T>& M, const TaskDev& K, const Scratch<T, NC>& s, int g1, int g2, HP margin,
                      ContactTmp& o) {
  const int t1 = M.geom_type[g1], t2 = M.geom_type[g2];
  HP p1[3], p2[3], R1[9], R2[9], s1[3], s2[3];
  o.n = 0;
  if (!pair_poses(M, K, s, g1, g2, margin, p1, p2, R1, R2, s1, s2)) return;
  if (t1 == 0 && t2 == 2) {
    const HP n[3] = {R1[2], R1[5], R1[8]};
    const HP dd = (p2[0] - p1[0]) * n[0] + (p2[1] - p1[1]) * n[1] + (p2[2] - p1[2]) * n[2] - s2[0];
    if (dd > margin) return;
    o.dist[0] = dd;
    for (int k = 0; k < 3; ++k) { o.nrm[k] = n[k]; o.pos[k] = p2[k] - n[k] * (s2[0] + (HP)0.5 * dd); }
    o.n = 1;
  } else if (t1 == 0 && t2 == 3) {
    // both capsule ends against the plane; results are placed with compile-time slot indices (a run-time
    // slot index would put the whole ContactTmp into private memory)
    const HP n[3] = {R1[2], R1[5], R1[8]}, ax[3] = {R2[2], R2[5], R2[8]};
    HP c0[3], c1[3];
    for (int k = 0; k < 3; ++k) { c0[k] = p2[k] + s2[1] * ax[k]; c1[k] = p2[k] - s2[1] * ax[k]; }
    const HP d0 = (c0[0] - p1[0]) * n[0] + (c0[1] - p1[1]) * n[1] + (c0[2] - p1[2]) * n[2] - s2[0];
    const HP d1 = (c1[0] - p1[0]) * n[0] + (c1[1] - p1[1]) * n[1] + (c1[2] - p1[2]) * n[2] - s2[0];
    const int v0 = !(d0 > margin), v1 = !(d1 > margin);
    const HP da = v0 ? d0 : d1;
    o.dist[0] = da;
    o.dist[1] = d1;
    for (int k = 0; k < 3; ++k) {
      o.nrm[k] = n[k]; o.nrm[3 + k] = n[k];
      o.pos[k] = (v0 ? c0[k] : c1[k]) - n[k] * (s2[0] + (HP)0.5 * da);
      o.pos[3 + k] = c1[k] - n[k] * (s2[0] + (HP)0.5 * d1);
    }
    o.n = v0 + v1;
  } else if (t1 == 2 && t2 == 2) {
    o.n = sphere_sphere(o.dist, o.pos, o.nrm, p1, s1[0], p2, s2[0], margin);
  } else if (t1 == 2 && t2 == 3) {
    const HP ax[3] = {R2[2], R2[5], R2[8]};
    HP q[3];
    seg_nearest(q, p2, ax, s2[1], p1);
    o.n = sphere_sphere(o.dist, o.pos, o.nrm, p1, s1[0], q, s2[0], margin);
  } else if (t1 == 3 && t2 == 3) {
    const HP a1[3] = {R1[2], R1[5], R1[8]}, a2[3] = {R2[2], R2[5], R2[8]};
    const HP dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    const HP ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2);
    const HP u = -dot3(a1, dif), v = dot3(a2, dif);
    const HP det = ma * mc - mb * mb;
    HP x1, x2;
    if (fabs(det) < (HP)1e-12) { x1 = 0; x2 = v / mc; } else { x1 = (mc * u - mb * v) / det; x2 = (ma * v - mb * u) / det; }
    x1 = tclamp(x1, -s1[1], s1[1]);
    x2 = tclamp(x2, -s2[1], s2[1]);
    (void)x2;
    HP q1[3], q2[3];
    for (int k = 0; k < 3; ++k) q1[k] = p1[k] + x1 * a1[k];
    seg_nearest(q2, p2, a2, s2[1], q1);
    seg_nearest(q1, p1, a1, s1[1], q2);
    o.n = sphere_sphere(o.dist, o.pos, o.nrm, q1, s1[0], q2, s2[0], margin);
  } else if (t1 == 2 && t2 == 6) {
    const HP t[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    HP c[3], cl[3];
    mulmatTvec3(c, R2, t);
    int inside = 1;
    for (int k = 0; k < 3; ++k) {
      cl[k] = c[k];
      if (cl[k] > s2[k]) { cl[k] = s2[k]; inside = 0; } else if (cl[k] < -s2[k]) { cl[k] = -s2[k]; inside = 0; }
    }
    HP nl[3], dd;
    if (!inside) {
      const HP df[3] = {cl[0] - c[0], cl[1] - c[1], cl[2] - c[2]};
      const HP dn = norm3(df);
      dd = dn - s1[0];
      if (dd > margin) return;
      { const HP i_dn = (HP)1 / dn; nl[0] = df[0] * i_dn; nl[1] = df[1] * i_dn; nl[2] = df[2] * i_dn; }
    } else {
      int kb = 0;
      HP best = (HP)1e30;
      for (int k = 0; k < 3; ++k) { const HP e = s2[k] - fabs(c[k]); if (e < best) { best = e; kb = k; } }
      const HP ckb = kb == 0 ? c[0] : (kb == 1 ? c[1] : c[2]);
      for (int k = 0; k < 3; ++k) nl[k] = (k == kb) ? (ckb > 0 ? (HP)-1 : (HP)1) : (HP)0;
      dd = -best - s1[0];
    }
    mulmatvec3(o.nrm, R2, nl);
    o.dist[0] = dd;
    for (int k = 0; k < 3; ++k) o.pos[k] = p1[k] + o.nrm[k] * (s1[0] + (HP)0.5 * dd);
    o.n = 1;
  }
}

// ---- narrow phases beyond MuJoCo's sphere / capsule primitives (capsule-box, box-box vertex contacts, sphere / capsule against
// cylinder and ellipsoid, plane against ellipsoid and cylinder).  Same geometry as oracle/myo_oracle.c (point_box, point_cylinder,
// point_ellipsoid, SEG_ARGMIN) — see the note there on what MuJoCo does for these pairs (libccd MPR / multi-contact primitives)
// and where the contact sets can differ.  Written without run-time indexed local arrays (they would live in private memory):
// results go to compile-time slots.  These run in their own leaf function (collision_pass_ext) over the model's trailing
// "extended" pairs, so the register allocation of the Baoding hand's collision pass is untouched.
// d/dt of the signed distance from the point c + t a to the solid, up to a positive factor (see the oracle's dsd_box / dsd_cylinder:
// the nearest point of a capsule's segment is found by bisection on this sign, to the last bit)
DEV HP dsd_box_hp(const HP* c, const HP* a, HP t, const HP* sz) {
  const HP x0 = c[0] + t * a[0], x1 = c[1] + t * a[1], x2 = c[2] + t * a[2];
  HP g = 0;
  int inside = 1;
  if (x0 > sz[0]) { g += (x0 - sz[0]) * a[0]; inside = 0; } else if (x0 < -sz[0]) { g += (x0 + sz[0]) * a[0]; inside = 0; }
  if (x1 > sz[1]) { g += (x1 - sz[1]) * a[1]; inside = 0; } else if (x1 < -sz[1]) { g += (x1 + sz[1]) * a[1]; inside = 0; }
  if (x2 > sz[2]) { g += (x2 - sz[2]) * a[2]; inside = 0; } else if (x2 < -sz[2]) { g += (x2 + sz[2]) * a[2]; inside = 0; }
  if (!inside) return g;
  const HP e0 = sz[0] - fabs(x0), e1 = sz[1] - fabs(x1), e2 = sz[2] - fabs(x2);
  HP best = e0, xk = x0, ak = a[0];
  if (e1 < best) { best = e1; xk = x1; ak = a[1]; }
  if (e2 < best) { best = e2; xk = x2; ak = a[2]; }
  return xk > 0 ? ak : -ak;
}
DEV HP dsd_cyl_hp(const HP* c, const HP* a, HP t, HP R, HP h) {
  const HP x0 = c[0] + t * a[0], x1 = c[1] + t * a[1], x2 = c[2] + t * a[2];
  const HP rho = sqrt(x0 * x0 + x1 * x1), az = fabs(x2);
  const HP ux = rho > (HP)1e-15 ? x0 / rho : (HP)1, uy = rho > (HP)1e-15 ? x1 / rho : (HP)0;
  if (rho > R || az > h) {
    const HP qr = rho < R ? rho : R, qz = x2 > h ? h : (x2 < -h ? -h : x2);
    return (x0 - ux * qr) * a[0] + (x1 - uy * qr) * a[1] + (x2 - qz) * a[2];
  }
  if (h - az < R - rho) return x2 > 0 ? a[2] : -a[2];
  return ux * a[0] + uy * a[1];
}
// the three point-vs-solid routines write contact slot SLOT of o and return 1 when the contact is within the margin
template <int SLOT>
DEV int point_finish(ContactTmp& o, const HP* c, HP r, const HP* Rg, const HP* nl, HP dd, HP flip) {
  HP nw[3];
  mulmatvec3(nw, Rg, nl);
  o.dist[SLOT] = dd;
  for (int k = 0; k < 3; ++k) { o.nrm[3 * SLOT + k] = flip * nw[k]; o.pos[3 * SLOT + k] = c[k] + nw[k] * (r + (HP)0.5 * dd); }
  return 1;
}
template <int SLOT, int SMOOTH_INSIDE>
DEV int point_box_hp(ContactTmp& o, const HP* c, HP r, const HP* pb, const HP* Rb, const HP* sb, HP margin, HP flip) {
  const HP t[3] = {c[0] - pb[0], c[1] - pb[1], c[2] - pb[2]};
  HP x[3], cl[3], nl[3], dd;
  mulmatTvec3(x, Rb, t);
  int inside = 1;
  for (int k = 0; k < 3; ++k) {
    cl[k] = x[k];
    if (cl[k] > sb[k]) { cl[k] = sb[k]; inside = 0; } else if (cl[k] < -sb[k]) { cl[k] = -sb[k]; inside = 0; }
  }
  if (!inside) {
    const HP df[3] = {cl[0] - x[0], cl[1] - x[1], cl[2] - x[2]};
    const HP dn = norm3(df);
    dd = dn - r;
    if (dd > margin) return 0;
    { const HP i_dn = (HP)1 / dn; nl[0] = df[0] * i_dn; nl[1] = df[1] * i_dn; nl[2] = df[2] * i_dn; }
  } else {
    int kb = 0;
    HP best = (HP)1e300;
    for (int k = 0; k < 3; ++k) { const HP e = sb[k] - fabs(x[k]); if (e < best) { best = e; kb = k; } }
    const HP xkb = kb == 0 ? x[0] : (kb == 1 ? x[1] : x[2]);
    for (int k = 0; k < 3; ++k) nl[k] = (k == kb) ? (xkb > 0 ? (HP)-1 : (HP)1) : (HP)0;
    if (SMOOTH_INSIDE) {      // capsule axis inside the box: normal from the smooth field x_k / s_k^2 (see the oracle's point_box)
      const HP g[3] = {-x[0] / (sb[0] * sb[0]), -x[1] / (sb[1] * sb[1]), -x[2] / (sb[2] * sb[2])};
      const HP gn = norm3(g);
      if (gn > (HP)1e-15) { { const HP i_gn = (HP)1 / gn; nl[0] = g[0] * i_gn; nl[1] = g[1] * i_gn; nl[2] = g[2] * i_gn; } }
    }
    dd = -best - r;
    if (dd > margin) return 0;
  }
  return point_finish<SLOT>(o, c, r, Rb, nl, dd, flip);
}
template <int SLOT>
DEV int point_cyl_hp(ContactTmp& o, const HP* c, HP r, const HP* pc, const HP* Rc, HP R, HP h, HP margin) {
  const HP t[3] = {c[0] - pc[0], c[1] - pc[1], c[2] - pc[2]};
  HP x[3], nl[3], dd;
  mulmatTvec3(x, Rc, t);
  const HP rho = sqrt(x[0] * x[0] + x[1] * x[1]), az = fabs(x[2]);
  const HP ux = rho > (HP)1e-15 ? x[0] / rho : (HP)1, uy = rho > (HP)1e-15 ? x[1] / rho : (HP)0;
  if (rho > R || az > h) {
    const HP qr = rho < R ? rho : R, qz = x[2] > h ? h : (x[2] < -h ? -h : x[2]);
    const HP df[3] = {ux * qr - x[0], uy * qr - x[1], qz - x[2]};
    const HP dn = norm3(df);
    dd = dn - r;
    if (dd > margin) return 0;
    { const HP i_dn = (HP)1 / dn; nl[0] = df[0] * i_dn; nl[1] = df[1] * i_dn; nl[2] = df[2] * i_dn; }
  } else {
    const HP e_side = R - rho, e_cap = h - az;
    if (e_cap < e_side) { nl[0] = 0; nl[1] = 0; nl[2] = x[2] > 0 ? (HP)-1 : (HP)1; dd = -e_cap - r; }
    else { nl[0] = -ux; nl[1] = -uy; nl[2] = 0; dd = -e_side - r; }
    if (dd > margin) return 0;
  }
  return point_finish<SLOT>(o, c, r, Rc, nl, dd, (HP)1);
}
template <int SLOT>
DEV int point_ell_hp(ContactTmp& o, const HP* c, HP r, const HP* pe, const HP* Re, const HP* sz, HP margin) {
  const HP t[3] = {c[0] - pe[0], c[1] - pe[1], c[2] - pe[2]};
  HP x[3], q[3];
  mulmatTvec3(x, Re, t);
  const HP lev = (x[0] / sz[0]) * (x[0] / sz[0]) + (x[1] / sz[1]) * (x[1] / sz[1]) + (x[2] / sz[2]) * (x[2] / sz[2]);
  HP dn, sign;
  if (lev > 1) {
    HP smax = sz[0] > sz[1] ? sz[0] : sz[1]; if (sz[2] > smax) smax = sz[2];
    HP lo = 0, hi = norm3(x) * smax;
    for (int it = 0; it < 64; ++it) {
      const HP mid = (HP)0.5 * (lo + hi);
      HP F = -1;
      for (int k = 0; k < 3; ++k) { const HP v = sz[k] * x[k] / (mid + sz[k] * sz[k]); F += v * v; }
      if (F > 0) lo = mid; else hi = mid;
    }
    const HP tt = (HP)0.5 * (lo + hi);
    for (int k = 0; k < 3; ++k) q[k] = sz[k] * sz[k] * x[k] / (tt + sz[k] * sz[k]);
    const HP df[3] = {q[0] - x[0], q[1] - x[1], q[2] - x[2]};
    dn = norm3(df); sign = 1;
  } else {
    const HP sc = lev > (HP)1e-15 ? 1 / sqrt(lev) : (HP)0;
    if (sc == 0) { q[0] = sz[0]; q[1] = 0; q[2] = 0; } else { q[0] = x[0] * sc; q[1] = x[1] * sc; q[2] = x[2] * sc; }
    const HP df[3] = {q[0] - x[0], q[1] - x[1], q[2] - x[2]};
    dn = norm3(df); sign = -1;
  }
  const HP dd = sign * dn - r;
  if (dd > margin) return 0;
  const HP g[3] = {-q[0] / (sz[0] * sz[0]), -q[1] / (sz[1] * sz[1]), -q[2] / (sz[2] * sz[2])};
  const HP gn = norm3(g);
  const HP ign = (HP)1 / gn;
  const HP nl[3] = {g[0] * ign, g[1] * ign, g[2] * ign};
  return point_finish<SLOT>(o, c, r, Re, nl, dd, (HP)1);
}

template <typename T, int NC>
DEV void collide_pair_ext(const DevModel<T>& M, const TaskDev& K, const Scratch<T, NC>& s, int g1, int g2, int sub, HP margin, ContactTmp& o) {
  const int t1 = M.geom_type[g1], t2 = M.geom_type[g2];
  HP p1[3], p2[3], R1[9], R2[9], s1[3], s2[3];
  o.n = 0;
  if (!pair_poses(M, K, s, g1, g2, margin, p1, p2, R1, R2, s1, s2)) return;
  if (t1 == 0 && t2 == 4) {                       // plane - ellipsoid: the deepest point along -n (support mapping)
    const HP n[3] = {R1[2], R1[5], R1[8]};
    HP w[3], q[3];
    mulmatTvec3(w, R2, n);
    const HP sw[3] = {s2[0] * w[0], s2[1] * w[1], s2[2] * w[2]};
    const HP L = norm3(sw);
    const HP ql[3] = {-s2[0] * sw[0] / L, -s2[1] * sw[1] / L, -s2[2] * sw[2] / L};
    mulmatvec3(q, R2, ql);
    for (int k = 0; k < 3; ++k) q[k] += p2[k];
    const HP dd = (q[0] - p1[0]) * n[0] + (q[1] - p1[1]) * n[1] + (q[2] - p1[2]) * n[2];
    if (dd > margin) return;
    o.dist[0] = dd;
    for (int k = 0; k < 3; ++k) { o.nrm[k] = n[k]; o.pos[k] = q[k] - n[k] * (HP)0.5 * dd; }
    o.n = 1;
  } else if (t1 == 0 && t2 == 5) {                // plane - cylinder: the lowest rim point of each cap (its centre when the cap is level)
    const HP n[3] = {R1[2], R1[5], R1[8]}, ax[3] = {R2[2], R2[5], R2[8]};
    const HP na = dot3(n, ax);
    const HP np_[3] = {n[0] - na * ax[0], n[1] - na * ax[1], n[2] - na * ax[2]};
    const HP L = norm3(np_);
    HP q0[3], q1[3];
    for (int k = 0; k < 3; ++k) {
      const HP rim = L > (HP)1e-12 ? s2[0] * np_[k] / L : (HP)0;
      q0[k] = p2[k] + s2[1] * ax[k] - rim; q1[k] = p2[k] - s2[1] * ax[k] - rim;
    }
    const HP d0 = (q0[0] - p1[0]) * n[0] + (q0[1] - p1[1]) * n[1] + (q0[2] - p1[2]) * n[2];
    const HP d1 = (q1[0] - p1[0]) * n[0] + (q1[1] - p1[1]) * n[1] + (q1[2] - p1[2]) * n[2];
    const int v0 = !(d0 > margin), v1 = !(d1 > margin);
    const HP da = v0 ? d0 : d1;
    o.dist[0] = da; o.dist[1] = d1;
    for (int k = 0; k < 3; ++k) {
      o.nrm[k] = n[k]; o.nrm[3 + k] = n[k];
      o.pos[k] = (v0 ? q0[k] : q1[k]) - n[k] * (HP)0.5 * da;
      o.pos[3 + k] = q1[k] - n[k] * (HP)0.5 * d1;
    }
    o.n = v0 + v1;
  } else if (t1 == 2 && t2 == 5) {
    o.n = point_cyl_hp<0>(o, p1, s1[0], p2, R2, s2[0], s2[1], margin);
  } else if (t1 == 2 && t2 == 4) {
    o.n = point_ell_hp<0>(o, p1, s1[0], p2, R2, s2, margin);
  } else if (t1 == 3 && (t2 == 4 || t2 == 5 || t2 == 6)) {
    // the point of the capsule's segment nearest to geom 2 (bisection on the slope of the convex signed distance; closed
    // form in the ellipsoid's own metric), then a sphere of the capsule's radius there
    const HP ax[3] = {R1[2], R1[5], R1[8]}, t[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    HP c[3], a[3], ts;
    mulmatTvec3(c, R2, t); mulmatTvec3(a, R2, ax);
    if (t2 == 4) {
      HP num = 0, den = 0;
      for (int k = 0; k < 3; ++k) { num += c[k] * a[k] / (s2[k] * s2[k]); den += a[k] * a[k] / (s2[k] * s2[k]); }
      ts = den > (HP)1e-15 ? -num / den : (HP)0;
      ts = tclamp(ts, -s1[1], s1[1]);
    } else {
      HP lo = -s1[1], hi = s1[1];
      const HP glo = t2 == 6 ? dsd_box_hp(c, a, lo, s2) : dsd_cyl_hp(c, a, lo, s2[0], s2[1]);
      const HP ghi = t2 == 6 ? dsd_box_hp(c, a, hi, s2) : dsd_cyl_hp(c, a, hi, s2[0], s2[1]);
      if (glo >= 0) ts = lo;
      else if (ghi <= 0) ts = hi;
      else {
        for (int it = 0; it < 60; ++it) {
          const HP tt = (HP)0.5 * (lo + hi);
          const HP g = t2 == 6 ? dsd_box_hp(c, a, tt, s2) : dsd_cyl_hp(c, a, tt, s2[0], s2[1]);
          if (g > 0) hi = tt; else lo = tt;
        }
        ts = (HP)0.5 * (lo + hi);
      }
    }
    const HP q[3] = {p1[0] + ts * ax[0], p1[1] + ts * ax[1], p1[2] + ts * ax[2]};
    if (t2 == 6) {
      // a capsule lying on a face rests on its two ends (both within the margin: two contacts, like plane-capsule); otherwise one
      // contact at the nearest point of the segment
      const HP qa[3] = {p1[0] + s1[1] * ax[0], p1[1] + s1[1] * ax[1], p1[2] + s1[1] * ax[2]};
      const HP qb[3] = {p1[0] - s1[1] * ax[0], p1[1] - s1[1] * ax[1], p1[2] - s1[1] * ax[2]};
      int both = point_box_hp<0, 1>(o, qa, s1[0], p2, R2, s2, margin, (HP)1);
      if (both) both += point_box_hp<1, 1>(o, qb, s1[0], p2, R2, s2, margin, (HP)1);
      o.n = both == 2 ? 2 : point_box_hp<0, 1>(o, q, s1[0], p2, R2, s2, margin, (HP)1);
    } else if (t2 == 5) o.n = point_cyl_hp<0>(o, q, s1[0], p2, R2, s2[0], s2[1], margin);
    else o.n = point_ell_hp<0>(o, q, s1[0], p2, R2, s2, margin);
  } else if (t1 == 6 && t2 == 6 && sub == 17) {  // box - box: the edge-edge candidate (oracle/myo_oracle.c: collide_pair, sub == 17)
    // separating-axis test over the 6 face normals and the 9 edge cross products; when the axis of largest separation is an edge
    // pair: one contact at the midpoint of the two edges' closest points, normal = that axis (box 1 -> box 2)
    const HP t[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    HP A[3][3], B[3][3];
    for (int k = 0; k < 3; ++k) for (int e = 0; e < 3; ++e) { A[k][e] = R1[3 * e + k]; B[k][e] = R2[3 * e + k]; }
    HP best_face = (HP)-1e300, best_edge = (HP)-1e300, Ln[3] = {0, 0, 0};
    int bi = -1, bj = -1;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const HP* L = k < 3 ? A[k] : B[k - 3];
      HP ra = 0, rb = 0;
      for (int e = 0; e < 3; ++e) { ra += s1[e] * fabs(dot3(A[e], L)); rb += s2[e] * fabs(dot3(B[e], L)); }
      const HP sep = fabs(dot3(t, L)) - ra - rb;
      best_face = sep > best_face ? sep : best_face;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        HP L[3];
        cross3(L, A[i], B[j]);
        const HP len = sqrt(dot3(L, L));
        const bool ok = !(len < (HP)1e-6);
        const HP inv = ok ? (HP)1 / len : (HP)0;
        for (int e = 0; e < 3; ++e) L[e] *= inv;
        HP ra = 0, rb = 0;
        for (int e = 0; e < 3; ++e) { ra += s1[e] * fabs(dot3(A[e], L)); rb += s2[e] * fabs(dot3(B[e], L)); }
        const HP sep = fabs(dot3(t, L)) - ra - rb;
        if (ok && sep > best_edge) { best_edge = sep; bi = i; bj = j; Ln[0] = L[0]; Ln[1] = L[1]; Ln[2] = L[2]; }
      }
    if (bi < 0 || best_edge > margin || best_face > margin) return;
    if (!(best_edge > best_face + (HP)1e-9 * ((HP)1 + fabs(best_face)))) return;
    if (dot3(t, Ln) < 0) { Ln[0] = -Ln[0]; Ln[1] = -Ln[1]; Ln[2] = -Ln[2]; }
    HP ea[3] = {p1[0], p1[1], p1[2]}, eb[3] = {p2[0], p2[1], p2[2]}, Ai[3] = {0, 0, 0}, Bj[3] = {0, 0, 0}, ha = 0, hb = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {                       // (compile-time k; bi / bj select: no run-time indexed local arrays)
      const HP sga = dot3(A[k], Ln) > 0 ? (HP)1 : (HP)-1, sgb = dot3(B[k], Ln) > 0 ? (HP)-1 : (HP)1;
      for (int e = 0; e < 3; ++e) {
        ea[e] += (k != bi) ? sga * s1[k] * A[k][e] : (HP)0;
        eb[e] += (k != bj) ? sgb * s2[k] * B[k][e] : (HP)0;
        Ai[e] = (k == bi) ? A[k][e] : Ai[e];
        Bj[e] = (k == bj) ? B[k][e] : Bj[e];
      }
      ha = (k == bi) ? s1[k] : ha; hb = (k == bj) ? s2[k] : hb;
    }
    const HP w[3] = {ea[0] - eb[0], ea[1] - eb[1], ea[2] - eb[2]};
    const HP ab = dot3(Ai, Bj), aw = dot3(Ai, w), bw = dot3(Bj, w), den = (HP)1 - ab * ab;
    HP u = den > (HP)1e-12 ? (ab * bw - aw) / den : (HP)0, v = den > (HP)1e-12 ? (bw - ab * aw) / den : (HP)0;
    u = tclamp(u, -ha, ha); v = tclamp(v, -hb, hb);
    for (int e = 0; e < 3; ++e) { o.pos[e] = (HP)0.5 * ((ea[e] + u * Ai[e]) + (eb[e] + v * Bj[e])); o.nrm[e] = Ln[e]; }
    o.dist[0] = best_edge;
    o.n = 1;
  } else if (t1 == 6 && t2 == 6 && sub >= 1) {   // box - box: one vertex-face candidate (see the pair table)
    const int v = (sub - 1) & 7, second = sub > 8;
    HP loc[3], q[3];
    if (!second) {
      loc[0] = (v & 1) ? s1[0] : -s1[0]; loc[1] = (v & 2) ? s1[1] : -s1[1]; loc[2] = (v & 4) ? s1[2] : -s1[2];
      mulmatvec3(q, R1, loc);
      for (int k = 0; k < 3; ++k) q[k] += p1[k];
      o.n = point_box_hp<0, 0>(o, q, (HP)0, p2, R2, s2, margin, (HP)1);
    } else {
      loc[0] = (v & 1) ? s2[0] : -s2[0]; loc[1] = (v & 2) ? s2[1] : -s2[1]; loc[2] = (v & 4) ? s2[2] : -s2[2];
      mulmatvec3(q, R2, loc);
      for (int k = 0; k < 3; ++k) q[k] += p2[k];
      o.n = point_box_hp<0, 0>(o, q, (HP)0, p1, R1, s1, margin, (HP)-1);    // the normal runs from geom 1 to geom 2
    }
  }
}

// impedance / reference parameters of one constraint row (mj_makeImpedance, getsolparam).  What depends on the model alone is resolved on
// the host (myobatch.hip: sol_precompute — the arrays keep their names and sizes): kb = (K, B) of the row's solref (refsafe applied),
// imp5 = (d0, d1, 1 / width or 0 when the impedance is constant, midpoint, power) of its solimp, clamped as getsolparam clamps them.  Left for
// the device: the impedance at this position — for MuJoCo's default power 2, one division.  (Round 5: the fp64 leaves that call this held
// 44 + 41 fp64 divisions per substep, ~12 instructions and a dependent chain each; `constraint_limits` 6.8 k -> see DESIGN section 5.)
template <typename T>
DEV void sol_param(const DevModel<T>& M, const T* kb, const T* imp5, T pos_minus_margin, T* Kp, T* Bp, T* Ip) {
  (void)M;
  const T d0 = imp5[0], d1 = imp5[1], iw = imp5[2];
  T imp;
  if (iw == 0) imp = (T)0.5 * (d0 + d1);
  else {
    const T x = fabs(pos_minus_margin) * iw;
    if (x >= 1) imp = d1;
    else if (x <= 0) imp = d0;
    else {
      const T mid = imp5[3], power = imp5[4];
      T y;
      if (power == 1) y = x;
      else if (power == 2 && mid == (T)0.5) y = (x <= mid) ? 2 * x * x : 1 - 2 * (1 - x) * (1 - x);   // MuJoCo's default solimp (midpoint 0.5, power 2): the same numbers without a division
      else if (power == 2) y = (x <= mid) ? x * x / mid : 1 - (1 - x) * (1 - x) / (1 - mid);   // no pow()
      else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
      else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
      imp = d0 + y * (d1 - d0);
    }
  }
  *Kp = kb[0]; *Bp = kb[1]; *Ip = imp;
}
// D = 1 / R of a row with R = max(MINVAL, (1 - imp) w / imp): one division
template <typename T> DEV T sol_D(T imp, T w) {
  const T num = (1 - imp) * w;
  return num >= MYO_MINVAL * imp ? imp / num : 1 / MYO_MINVAL;
}

// D of a pyramidal contact's rows (mj_makeImpedance): R0 = max(MINVAL, (1 - imp)(tran + mu0^2 tran) / imp) of the first row's diagApprox,
// R_py = max(MINVAL, 2 mu'^2 R0), mu' = friction[0] / sqrt(impratio) (1 / sqrt(impratio): host, M.isqrt_impratio) — one division while
// neither floor is active, the formula as written otherwise
template <typename T> DEV T con_D_pyramid(const DevModel<T>& M, T imp, T tran, T fr0) {
  const T mu = fr0 * M.isqrt_impratio, m2 = 2 * mu * mu;
  const T R0n = (1 - imp) * (tran + fr0 * fr0 * tran);
  if (R0n >= MYO_MINVAL * imp && m2 * R0n >= MYO_MINVAL * imp) return imp / (m2 * R0n);
  const T R0 = tmax(MYO_MINVAL, R0n / imp);
  return 1 / tmax(MYO_MINVAL, m2 * R0);
}

// friction-loss rows (mj_instantiateFriction), models that have them only (M.any_floss, chosen at kernel level): tendon = 0, BEFORE
// constraint_limits: one row per dof with frictionloss > 0 at the head of the joint-row range (J = +dof, like a lower joint limit);
// tendon = 1, AFTER it: one row per tendon with frictionloss > 0 behind the tendon-limit rows (J = +moment arms).  pos = margin = 0, so
// aref = -B v; R = (1 - d0) / d0 * invweight; the solver treats the rows by their MYO_LIM_FRIC flag (fric_update, fric_linesearch).
template <typename T, int NC>
DEVFN void friction_rows(const DevModel<T>& M_in, const TaskDev& K_in, Scratch<T, NC>& s_in, int tendon) {
  MYO_BIND_M(T) MYO_BIND_K MYO_BIND_S(T)
  WAVE_FN
  LANE_VAR(int, cnt);
  int total = 0;
  PHASE {
    const int i = lane;
    LV(cnt) = tendon ? (i < M.ntendon && M.tendon_frictionloss[i] > 0) : (i < M.nv && M.dof_frictionloss[i] > 0);
  }
  WAVE_EXSCAN(LV(cnt), S_NPRE(s), total);
  const int base = tendon ? s.nl + s.ntl : 0;
  PHASE {
    const int i = lane, r = base + S_NPRE(s)[lane];
    if (LV(cnt) && r < MYO_NLIM_MAX) {
      T Kc, Bc, Ic;
      if (tendon) sol_param(M, M.tendon_solref_fri + 2 * i, M.tendon_solimp_fri + 5 * i, (T)0, &Kc, &Bc, &Ic);
      else sol_param(M, M.dof_solref + 2 * i, M.dof_solimp + 5 * i, (T)0, &Kc, &Bc, &Ic);
      s.lim_id[r] = i | MYO_LIM_FRIC;
      s.efc_D[r] = sol_D(Ic, tendon ? M.tendon_invweight0[i] : M.dof_invweight0[i]); S_ROW_B(s)[r] = Bc; S_ROW_KIP(s)[r] = 0;
    }
    if (lane == 0) {
      const int n = base + total < MYO_NLIM_MAX ? base + total : MYO_NLIM_MAX;
      if (tendon) { s.ntl = n - s.nl; s.nefc = n; } else s.nl = n;
      if (base + total > MYO_NLIM_MAX && K.health) myo_count(K.health + 2);      // rows beyond the capacity are dropped: counted (myo_batch_health)
    }
  }
  SYNC();
}

template <typename T, int NC>
DEVFN void constraint_limits(const DevModel<T>& M_in, const TaskDev& K_in, Scratch<T, NC>& s_in) {
  MYO_BIND_M(T) MYO_BIND_K MYO_BIND_S(T)
  WAVE_FN
  // ---- limit rows: joints (lanes = joints), then tendons (lanes = tendons)
  LANE_VAR(int, cnt);
  LANE_VAR(T, dlo);
  LANE_VAR(T, dhi);
  int total = 0;
  const int r0 = M.any_floss ? s.nl : 0;       // rows [0, r0): the dofs' friction-loss rows (friction_rows, models that have them)
  PHASE {
    const int j = lane;
    int c = 0;
    T a = 0, b = 0;
    if (j < M.njnt && M.jnt_limited[j] && M.jnt_type[j] != 0) {
      // distances and the activation tests in HP, from the HP state
      const HP q = s.qpos[M.jnt_qposadr[j]];
      const HP ah = q - M.h_jnt_range[2 * j], bh = M.h_jnt_range[2 * j + 1] - q, mh = M.h_jnt_margin[j];
      c = (ah < mh ? 1 : 0) + (bh < mh ? 1 : 0);
      a = (T)(ah - mh); b = (T)(bh - mh);      // kept as dist - margin
    }
    LV(cnt) = c; LV(dlo) = a; LV(dhi) = b;
  }
  WAVE_EXSCAN(LV(cnt), S_NPRE(s), total);
  PHASE {
    const int j = lane;
    if (LV(cnt) > 0) {
      int r = r0 + S_NPRE(s)[lane];
      const HP q = s.qpos[M.jnt_qposadr[j]], mh = M.h_jnt_margin[j];
      const int on_lo = (q - M.h_jnt_range[2 * j]) < mh, on_hi = (M.h_jnt_range[2 * j + 1] - q) < mh;
      for (int side = 0; side < 2; ++side) {
        const T dm = side ? LV(dhi) : LV(dlo);         // dist - margin
        if ((side ? on_hi : on_lo) && r < MYO_NLIM_MAX) {
          T Kc, Bc, Ic;
          sol_param(M, M.jnt_solref + 2 * j, M.jnt_solimp + 5 * j, dm, &Kc, &Bc, &Ic);
          s.lim_id[r] = M.jnt_dofadr[j] | (side ? MYO_LIM_UPPER : 0);   // joint rows keep the DOF index
          s.efc_D[r] = sol_D(Ic, M.dof_invweight0[M.jnt_dofadr[j]]); S_ROW_B(s)[r] = Bc; S_ROW_KIP(s)[r] = Kc * Ic * dm;
          r++;
        }
      }
    }
    if (lane == 0) {
      s.nl = r0 + total < MYO_NLIM_MAX ? r0 + total : MYO_NLIM_MAX;
      if (r0 + total > MYO_NLIM_MAX && K.health) myo_count(K.health + 2);        // limit rows beyond the capacity are dropped: counted (myo_batch_health)
    }
  }
  SYNC();
  const int nl = s.nl;
  PHASE {
    const int t = lane;
    int c = 0;
    T a = 0, b = 0;
    if (t < M.ntendon && M.tendon_limited[t]) {
      const HP L = S_TEN_LENGTH(s)[t], mh = M.h_tendon_margin[t];
      const HP ah = L - M.h_tendon_range[2 * t], bh = M.h_tendon_range[2 * t + 1] - L;
      c = (ah < mh ? 1 : 0) + (bh < mh ? 1 : 0);
      a = (T)(ah - mh); b = (T)(bh - mh);      // kept as dist - margin
    }
    LV(cnt) = c; LV(dlo) = a; LV(dhi) = b;
  }
  WAVE_EXSCAN(LV(cnt), S_NPRE(s), total);
  PHASE {
    const int t = lane;
    if (LV(cnt) > 0) {
      int r = nl + S_NPRE(s)[lane];
      const HP L = S_TEN_LENGTH(s)[t], mh = M.h_tendon_margin[t];
      const int on_lo = (L - M.h_tendon_range[2 * t]) < mh, on_hi = (M.h_tendon_range[2 * t + 1] - L) < mh;
      for (int side = 0; side < 2; ++side) {
        const T dm = side ? LV(dhi) : LV(dlo);         // dist - margin
        if ((side ? on_hi : on_lo) && r < MYO_NLIM_MAX) {
          T Kc, Bc, Ic;
          sol_param(M, M.tendon_solref_lim + 2 * t, M.tendon_solimp_lim + 5 * t, dm, &Kc, &Bc, &Ic);
          s.lim_id[r] = t | (side ? MYO_LIM_UPPER : 0);
          s.efc_D[r] = sol_D(Ic, M.tendon_invweight0[t]); S_ROW_B(s)[r] = Bc; S_ROW_KIP(s)[r] = Kc * Ic * dm;
          r++;
        }
      }
    }
    if (lane == 0) { int n = nl + total; s.ntl = (n < MYO_NLIM_MAX ? n : MYO_NLIM_MAX) - nl; if (n > MYO_NLIM_MAX && K.health) myo_count(K.health + 2); }
  }
  SYNC();
  PHASE { if (lane == 0) { s.ncon = 0; s.nefc = s.nl + s.ntl; } }
  SYNC();
}

// fp32 bounding-sphere PRE-filter with the MODEL rbound (the reference rewrites geom_size per episode without refreshing rbound,
// baoding.py:586-604 — the stale value gates contacts): rejects only pairs that are apart by more than its rounding; the narrow
// phase repeats the test exactly, in HP (pair_poses)
template <typename T, int NC>
DEV int pair_far_apart(const DevModel<T>& M, const TaskDev& K, const Scratch<T, NC>& s, int g1, int g2, HP margin) {
  const T rb1 = M.geom_rbound[g1], rb2 = M.geom_rbound[g2];
  if (!(rb1 > 0 && rb2 > 0)) return 0;
  T c1[3], c2[3], l1[3], l2[3];
  geom_lpos_of(M, K, s, g1, l1);
  geom_lpos_of(M, K, s, g2, l2);
  body_point(s, M.geom_bodyid[g1], l1, c1);
  body_point(s, M.geom_bodyid[g2], l2, c2);
  const T df[3] = {c1[0] - c2[0], c1[1] - c2[1], c1[2] - c2[2]};
  const T bound = rb1 + rb2 + (T)margin;
  return dot3(df, df) > bound * bound * (T)1.0001;
}
// keep only contacts that enter the constraint set (dist < margin - gap); static slot indices
template <typename T>
DEV void pair_keep_included(const DevModel<T>& M, int g1, int g2, HP margin, ContactTmp& ct, int p = -1) {
  const HP inc = p >= 0 ? M.h_pair_mg[2 * p + 1] : margin - tmax(M.h_geom_gap[g1], M.h_geom_gap[g2]);      // (p >= 0: the pair record's margin - gap)
  const int k0 = ct.n > 0 && ct.dist[0] < inc, k1 = ct.n > 1 && ct.dist[1] < inc;
  if (!k0 && k1) {
    ct.dist[0] = ct.dist[1];
    for (int e = 0; e < 3; ++e) { ct.pos[e] = ct.pos[3 + e]; ct.nrm[e] = ct.nrm[3 + e]; }
  }
  ct.n = k0 + k1;
}
#ifdef MYO_EMU
#define LANE_ARG(T, name) T* name
#else
#define LANE_ARG(T, name) T& name
#endif
// contact records from the narrow-phase results of one pass (ballot / prefix compaction over the 64 lanes, constraint parameters
// from the host-resolved pair records); shared by the two collision passes.  pbase = pair index of lane 0.
// condim-3-only models (M.any_gen == 0, wave-uniform): one slot of kind 0 per contact — the lean path the bench workload runs
template <typename T, int NC>
DEV void contacts_emit_c3(const DevModel<T>& M_in, const TaskDev& K_in, Scratch<T, NC>& s_in, int pbase, LANE_ARG(ContactTmp, ct), int& ncon) {
  MYO_BIND_M(T) MYO_BIND_K MYO_BIND_S(T)
  WAVE_FN_K
  int total = 0;
  const int base = pbase;
  // contact slots this substep can hold: the record slots, and what the limit rows leave of the constraint-row arrays (four rows a slot)
  const int nlim = s.nl + s.ntl;
  const int cap = tmin((int)Scratch<T, NC>::NREC, (MYO_NLIM_MAX + 4 * NC - nlim) >> 2);
  {
    WAVE_EXSCAN(LV(ct).n, S_NPRE(s), total);
    PHASE {
      const int p = base + lane;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int ci = ncon + S_NPRE(s)[lane] + k;
        if (k >= LV(ct).n || ci >= cap) break;
        const int g1 = M.pair_geom1[p], g2 = M.pair_geom2[p];
        auto& c = CON(s, ci);
        T cpos[3], cnrm[3];                // contact point relative to O: only r1 / r2 below are made of it
        for (int e = 0; e < 3; ++e) { cpos[e] = (T)(LV(ct).pos[3 * k + e] - S_ORIGIN(s)[e]); cnrm[e] = (T)LV(ct).nrm[3 * k + e]; }
        c.tinv = make_frame(cnrm);
        for (int e = 0; e < 3; ++e) c.nrm[e] = cnrm[e];
        // everything that depends on the two geoms only comes from the host-resolved pair record (pc_*):
        // bodies, tree roots, dof masks + support list, mixed solref / solimp (mj_contactParam), static
        // friction, margin and gap, inverse-weight sum; the balls' per-env friction is patched in here
        const int b1 = M.pc_i[8 * p], b2 = M.pc_i[8 * p + 1], root1 = M.pc_i[8 * p + 2], root2 = M.pc_i[8 * p + 3];
        const int ns = M.pc_i[8 * p + 4], fsel = M.pc_i[8 * p + 7];
        T F[16];
        for (int e = 0; e < 16; ++e) F[e] = M.pc_f[16 * p + e];
        // condim 3: both tangential directions use friction[0]; torsional / rolling coefficients are not used
        T fa = F[9], fb = F[12];
        if (K.objg_gidn > 0) {
          if (g1 >= K.objg_gid0 && g1 < K.objg_gidn) fa = S_OBJF(K, s)[(g1 - K.objg_gid0) * Scratch<T, NC>::OBJG_NF];
          if (g2 >= K.objg_gid0 && g2 < K.objg_gidn) fb = S_OBJF(K, s)[(g2 - K.objg_gid0) * Scratch<T, NC>::OBJG_NF];
        } else {
          fa = g1 == K.obj1_gid ? s.ball_fric[0] : (g1 == K.obj2_gid ? s.ball_fric[3] : fa);
          fb = g2 == K.obj1_gid ? s.ball_fric[0] : (g2 == K.obj2_gid ? s.ball_fric[3] : fb);
        }
        const T fr0 = (fsel == 0) ? tmax(fa, fb) : (fsel == 1 ? fa : fb);
        c.muA = fr0; c.muB = fr0;
        const T dmi = (T)(LV(ct).dist[k] - (tmax(M.h_geom_margin[g1], M.h_geom_margin[g2]) - tmax(M.h_geom_gap[g1], M.h_geom_gap[g2])));   // dist - (margin - gap), HP difference
        T Kc, Bc, Ic;
        sol_param(M, F + 2, F + 4, dmi, &Kc, &Bc, &Ic);
        const T tran = F[15];
        CON_SET_D(s, ci, con_D_pyramid(M, Ic, tran, fr0));
        { const int r0 = nlim + 4 * ci; const T kip = Kc * Ic * dmi; for (int e = 0; e < 4; ++e) { S_ROW_B(s)[r0 + e] = Bc; S_ROW_KIP(s)[r0 + e] = kip; } }
        {
          const T* c1 = S_COM(s) + 3 * root1; const T* c2 = S_COM(s) + 3 * root2;
          for (int e = 0; e < 3; ++e) { c.r1[e] = cpos[e] - c1[e]; c.r2[e] = cpos[e] - c2[e]; }
        }
        for (int e = 0; e < MYO_CS_MAX / 4; ++e) CON_SUP_WORDS(c)[e] = M.pc_sup[4 * p + e];
        c.pk = b1 | (b2 << 8) | (ns << 16);
      }
    }
    SYNC();
    ncon += total;      // (runs past the capacity: slots beyond it are not written, contacts_clamp counts the substep and cuts the count)
  }
}

// any condim (1, 3, 4, 6)
template <typename T, int NC>
DEV void contacts_emit_gen(const DevModel<T>& M_in, const TaskDev& K_in, Scratch<T, NC>& s_in, int pbase, LANE_ARG(ContactTmp, ct), int& ncon) {
  MYO_BIND_M(T) MYO_BIND_K MYO_BIND_S(T)
  WAVE_FN_K
  int total = 0;
  const int base = pbase;
  // contact slots this substep can hold: the record slots, and what the limit rows leave of the constraint-row arrays (four rows a slot)
  const int nlim = s.nl + s.ntl;
  const int cap = tmin((int)Scratch<T, NC>::NREC, (MYO_NLIM_MAX + 4 * NC - nlim) >> 2);
  {
    // slots per lane: a contact of condim 1 / 3 takes one, condim 4 two, condim 6 three (ContactRec kinds)
    LANE_VAR(int, nslot);
    PHASE {
      const int p = base + lane;
      const int dim = LV(ct).n > 0 ? (M.pc_i[8 * p + 6] & 255) : 3;
      LV(nslot) = LV(ct).n * (dim == 6 ? 3 : (dim == 4 ? 2 : 1));
    }
    WAVE_EXSCAN6(LV(nslot), S_NPRE(s), total);
    PHASE {
      const int p = base + lane;
      if (LV(ct).n > 0) {
        const int g1 = M.pair_geom1[p], g2 = M.pair_geom2[p];
        // everything that depends on the two geoms only comes from the host-resolved pair record (pc_*):
        // bodies, tree roots, dof masks + support list, mixed solref / solimp (mj_contactParam), static
        // friction, margin and gap, inverse-weight sums, condim; the per-env friction of the balls / the die is patched in here
        const int b1 = M.pc_i[8 * p], b2 = M.pc_i[8 * p + 1], root1 = M.pc_i[8 * p + 2], root2 = M.pc_i[8 * p + 3];
        const int ns = M.pc_i[8 * p + 4], dim = M.pc_i[8 * p + 6] & 255, xpair = M.pc_i[8 * p + 6] >> 8, fsel = M.pc_i[8 * p + 7];
        const int per = dim == 6 ? 3 : (dim == 4 ? 2 : 1);
        T F[16];
        for (int e = 0; e < 16; ++e) F[e] = M.pc_f[16 * p + e];
        // friction (sliding, torsional, rolling): the larger of the two geoms', or the higher-priority geom's
        T fr[3];
        for (int k = 0; k < 3; ++k) {
          if (k > 0 && dim <= 3) { fr[k] = 0; continue; }
          // (an explicit <pair> has its own coefficients: the geoms' — and their per-env values — do not enter)
          const T fa = xpair ? F[9 + k] : geom_fric_of(M, K, s, g1, k, F[9 + k]), fb = xpair ? F[12 + k] : geom_fric_of(M, K, s, g2, k, F[12 + k]);
          fr[k] = (fsel == 0) ? tmax(fa, fb) : (fsel == 1 ? fa : fb);
        }
        const T fr0 = fr[0];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          if (k >= LV(ct).n) break;
          T cpos[3], frame[3];                 // contact point relative to O: only r1 / r2 below are made of it
          for (int e = 0; e < 3; ++e) { cpos[e] = (T)(LV(ct).pos[3 * k + e] - S_ORIGIN(s)[e]); frame[e] = (T)LV(ct).nrm[3 * k + e]; }
          const T tinv = make_frame(frame);
          const T dmi = (T)(LV(ct).dist[k] - M.h_pair_mg[2 * p + 1]);   // dist - (margin - gap), HP difference
          T Kc, Bc, Ic;
          sol_param(M, F + 2, F + 4, dmi, &Kc, &Bc, &Ic);
          const T tran = F[15];
          // mj_makeImpedance: R of the first row from its diagApprox (tran + mu^2 tran; tran alone for the frictionless row), then every
          // pyramid row of the contact gets Rpy = 2 mu'^2 R, mu' = friction[0] / sqrt(impratio)
          T D;
          if (dim == 1) D = sol_D(Ic, tran);
          else D = con_D_pyramid(M, Ic, tran, fr0);
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int ci = ncon + S_NPRE(s)[lane] + k * per + j;
            if (j >= per || ci >= cap) break;
            auto& c = CON(s, ci);
            const int kind = dim == 1 ? 3 : (j == 0 ? 0 : (dim == 4 ? 4 : j));
            for (int e = 0; e < 3; ++e) c.nrm[e] = frame[e];
            c.tinv = tinv;
            c.muA = kind == 0 ? fr0 : (kind == 1 || kind == 4 ? fr[1] : (kind == 2 ? fr[2] : (T)0));
            c.muB = kind == 0 ? fr0 : (kind == 1 ? fr[2] : (T)0);
            CON_SET_D(s, ci, D);
            { const int r0 = nlim + 4 * ci; const T kip = Kc * Ic * dmi; for (int e = 0; e < 4; ++e) { S_ROW_B(s)[r0 + e] = Bc; S_ROW_KIP(s)[r0 + e] = kip; } }
            {
              const T* c1 = S_COM(s) + 3 * root1; const T* c2 = S_COM(s) + 3 * root2;
              for (int e = 0; e < 3; ++e) { c.r1[e] = cpos[e] - c1[e]; c.r2[e] = cpos[e] - c2[e]; }
            }
            for (int e = 0; e < MYO_CS_MAX / 4; ++e) CON_SUP_WORDS(c)[e] = M.pc_sup[4 * p + e];
            c.pk = b1 | (b2 << 8) | (ns << 16) | (kind << 24);
          }
        }
      }
    }
    SYNC();
    ncon += total;      // (runs past the capacity: slots beyond it are not written, contacts_clamp counts the substep and cuts the count)
  }
}

// after the last collision pass: a substep that wanted more contact slots than the scratch holds keeps the first `cap` (the surplus —
// the last candidate pairs — was never written), as MuJoCo drops beyond nconmax with a warning.  Counted (myo_batch_health [1]) together
// with the largest number of slots any substep wanted ([3]), so that the host sees both that it happened and what would have been enough.
template <typename T, int NC>
DEV void contacts_clamp(const TaskDev& K_in, Scratch<T, NC>& s_in) {
  MYO_BIND_K MYO_BIND_S(T)
  WAVE_FN_K
  const int nlim = s.nl + s.ntl, ncon = s.ncon;
  const int cap = tmin((int)Scratch<T, NC>::NREC, (MYO_NLIM_MAX + 4 * NC - nlim) >> 2);
  if (ncon > cap) {
    PHASE {
      if (lane == 0) {
        if (K.health) { myo_count(K.health + 1); myo_max(K.health + 3, ncon); }
        s.ncon = cap; s.nefc = nlim + 4 * cap;
      }
    }
    SYNC();
  }
  if constexpr (Scratch<T, NC>::SPILL) { SYNC_G(); }      // (the contact records went to global memory: other lanes read them from here on)
}

template <bool GEN, typename T, int NC>
DEV void contacts_emit(const DevModel<T>& M_in, const TaskDev& K_in, Scratch<T, NC>& s_in, int pbase, LANE_ARG(ContactTmp, ct), int& ncon) {
  if constexpr (GEN) contacts_emit_gen(M_in, K_in, s_in, pbase, ct, ncon);
  else contacts_emit_c3(M_in, K_in, s_in, pbase, ct, ncon);
}

// contacts: one pass over 64 candidate pairs (lanes = pairs), called from kernel level once per 64 pairs — a leaf
// function without the limit rows and without a loop, so that the HP narrow phase has a register allocation of its
// own.  s.ncon / s.nefc carry the running counts between passes.  The model's pairs are ordered: first the pairs of MuJoCo's
// sphere / capsule / plane primitives and sphere-box (this pass), then the "extended" pairs (collision_pass_ext).
// GEN: the model has pairs whose condim is not 3 (M.any_gen): the general slot emission; a separate instantiation, chosen at
// kernel level, so that the condim-3 pass is exactly the code it was before condim 1 / 4 / 6 existed
template <bool GEN, typename T, int NC>
DEVFN void collision_pass(const DevModel<T>& M_in, const TaskDev& K_in, Scratch<T, NC>& s_in, int base) {
  MYO_BIND_M(T) MYO_BIND_K MYO_BIND_S(T)
  WAVE_FN
  const int nlim = s.nl + s.ntl;
  int ncon = s.ncon;
  LANE_VAR(ContactTmp, ct);
  PHASE {
    const int p = base + lane;
    LV(ct).n = 0;
    if (p < M.npair_std) {
      const int g1 = M.pair_geom1[p], g2 = M.pair_geom2[p];
      const HP margin = GEN ? M.h_pair_mg[2 * p] : tmax(M.h_geom_margin[g1], M.h_geom_margin[g2]);      // (GEN: explicit <pair>s carry their own)
      if (!pair_far_apart(M, K, s, g1, g2, margin)) {
        collide_pair(M, K, s, g1, g2, margin, LV(ct));
        pair_keep_included(M, g1, g2, margin, LV(ct), GEN ? p : -1);
      }
    }
  }
  contacts_emit<GEN>(M, K, s, base, ct, ncon);
  PHASE {
    if (lane == 0) { s.ncon = ncon; s.nefc = nlim + 4 * ncon; }
  }
  SYNC();
}
// the same for the extended pairs (capsule-box, box-box vertex candidates, cylinders, ellipsoids): base counts from the first of them
template <bool GEN, typename T, int NC>
DEVFN void collision_pass_ext(const DevModel<T>& M_in, const TaskDev& K_in, Scratch<T, NC>& s_in, int base) {
  MYO_BIND_M(T) MYO_BIND_K MYO_BIND_S(T)
  WAVE_FN
  const int nlim = s.nl + s.ntl;
  int ncon = s.ncon;
  LANE_VAR(ContactTmp, ct);
  PHASE {
    const int p = M.npair_std + base + lane;
    LV(ct).n = 0;
    if (p < M.npair) {
      const int g1 = M.pair_geom1[p], g2 = M.pair_geom2[p];
      const HP margin = GEN ? M.h_pair_mg[2 * p] : tmax(M.h_geom_margin[g1], M.h_geom_margin[g2]);
      if (!pair_far_apart(M, K, s, g1, g2, margin)) {
        collide_pair_ext(M, K, s, g1, g2, M.pc_i[8 * p + 5], margin, LV(ct));
        pair_keep_included(M, g1, g2, margin, LV(ct), GEN ? p : -1);
      }
    }
  }
  contacts_emit<GEN>(M, K, s, M.npair_std + base, ct, ncon);
  PHASE {
    if (lane == 0) { s.ncon = ncon; s.nefc = nlim + 4 * ncon; }
  }
  SYNC();
}

// ------------------------------------------------------------------------------------------
// matrix-free constraint Jacobian products
// body spatial vectors V_b(v) = sum_{d in ancestors(b)} cdof_d v_d   (lanes = bodies)
template <typename T, int NC>
DEV void body_vectors(const DevModel<T>& M_in, const Scratch<T, NC>& s_in, LCREF(T) v_r, LREF(T) out_r) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  const T* v = LPTR(const T, v_r); T* out = LPTR(T, out_r);
  WAVE_FN_K
  PHASE {
    const int b = lane;
    if (b < M.nbody) {
      T acc[6] = {0, 0, 0, 0, 0, 0};
      unsigned long long m = M.body_dofmask[b];
      while (m) {
        const int d = myo_ffsll(m);
        m &= m - 1;
        const T vd = v[d];
        for (int e = 0; e < 6; ++e) acc[e] += s.cdof[6 * d + e] * vd;
      }
      for (int e = 0; e < 6; ++e) out[6 * b + e] = acc[e];
    }
  }
  SYNC();
}
template <typename T, typename P>
DEV void point_vel(const T* bv, int b, P off, T* out) {       // (P: pointer to the offset, LDS or global)
  const T* V = bv + 6 * b;
  const T o[3] = {off[0], off[1], off[2]};
  T t[3];
  cross3(t, V, o);
  out[0] = V[3] + t[0]; out[1] = V[4] + t[1]; out[2] = V[5] + t[2];
}
// Jacobian column of dof d at a contact, given the contact offset from the dof's tree reference point
template <typename T, int NC, typename P> DEV void con_col(const Scratch<T, NC>& s, int d, P off, T* col) {
  const T* cd = s.cdof + 6 * d;
  const T o[3] = {off[0], off[1], off[2]};
  T t[3];
  cross3(t, cd, o);
  col[0] = cd[3] + t[0]; col[1] = cd[4] + t[1]; col[2] = cd[5] + t[2];
}

// J_times for models with condim 1 / 4 / 6 pairs (M.any_gen): a leaf of its own, the condim-3 version below stays what it was
template <typename T, int NC>
DEVFN void J_times_gen(const DevModel<T>& M_in, const Scratch<T, NC>& s_in, LCREF(T) v_r, LCREF(T) bv_r, LREF(T) out_r) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  const T* v = LPTR(const T, v_r); const T* bv = LPTR(const T, bv_r); T* out = LPTR(T, out_r);
  WAVE_FN
  const int nl = s.nl, nlim = s.nl + s.ntl, nefc = s.nefc;
  PHASE {
    for (int r = lane; r < nefc; r += 64) {
      T val;
      if (r < nl) val = lim_sign<T>(s.lim_id[r]) * v[lim_index(s.lim_id[r])];
      else if (r < nlim) {
        const int t = lim_index(s.lim_id[r]);
        unsigned long long m = M.tendon_dofmask[t];
        T acc = 0, tj[MYO_TJ_MAX];
        tenj_row(s, t, tj);
#pragma unroll
        for (int k = 0; k < MYO_TJ_MAX; ++k) if (m) { const int d = myo_ffsll(m); m &= m - 1; acc += tj[k] * v[d]; }
        val = lim_sign<T>(s.lim_id[r]) * acc;
      } else {
        const int ci = (r - nlim) >> 2, e = (r - nlim) & 3;
        const auto& c = CON(s, ci);
        T v1[3], v2[3];
        point_vel(bv, con_b1(c), c.r1, v1);
        point_vel(bv, con_b2(c), c.r2, v2);
        const T rel[3] = {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]};
        const T* w1 = bv + 6 * con_b1(c); const T* w2 = bv + 6 * con_b2(c);
        const T ra[3] = {w2[0] - w1[0], w2[1] - w1[1], w2[2] - w1[2]};
        val = con_row_val(c, con_kind(c), e, rel, ra);
      }
      out[r] = val;
    }
  }
  SYNC();
}

// out[r] = (J v)[r] for every constraint row; bv = body vectors of v (already computed)
template <typename T, int NC>
DEV void J_times(const DevModel<T>& M_in, const Scratch<T, NC>& s_in, LCREF(T) v_r, LCREF(T) bv_r, LREF(T) out_r) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  const T* v = LPTR(const T, v_r); const T* bv = LPTR(const T, bv_r); T* out = LPTR(T, out_r);
  WAVE_FN_K
  const int nl = s.nl, nlim = s.nl + s.ntl, nefc = s.nefc;
  if (M.any_gen) { J_times_gen(M, s, v_r, bv_r, out_r); return; }     // (scalar branch: the flag lives in constant memory)
  PHASE {
    for (int r = lane; r < nefc; r += 64) {
      T val;
      if (r < nl) val = lim_sign<T>(s.lim_id[r]) * v[lim_index(s.lim_id[r])];
      else if (r < nlim) {
        const int t = lim_index(s.lim_id[r]);
        unsigned long long m = M.tendon_dofmask[t];
        T acc = 0, tj[MYO_TJ_MAX];
        tenj_row(s, t, tj);
#pragma unroll
        for (int k = 0; k < MYO_TJ_MAX; ++k) if (m) { const int d = myo_ffsll(m); m &= m - 1; acc += tj[k] * v[d]; }
        val = lim_sign<T>(s.lim_id[r]) * acc;
      } else {
        const int ci = (r - nlim) >> 2, e = (r - nlim) & 3;
        const auto& c = CON(s, ci);
        T v1[3], v2[3];
        point_vel(bv, con_b1(c), c.r1, v1);
        point_vel(bv, con_b2(c), c.r2, v2);
        const T rel[3] = {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]};
                                          // condim-3 models: translation along tangent 1 | 2
        T t2[3], fr[6];
        con_frame(c, fr);
        con_t2(fr, t2);
        const T vn = dot3(fr, rel), vt = (e >> 1) ? dot3(t2, rel) : dot3(fr + 3, rel);
        val = vn + ((e & 1) ? -c.muA : c.muA) * vt;
      }
      out[r] = val;
    }
  }
  SYNC();
}

// body_vectors / J_times for TWO vectors at once (the solver's warm-start comparison evaluates J on
// qacc_warmstart and on qacc_smooth): the cdof entries and contact records are read once.
template <typename T, int NC>
DEVFN void body_vectors2(const DevModel<T>& M_in, const Scratch<T, NC>& s_in, LCREF(T) va_r, LCREF(T) vb_r, LREF(T) outa_r, LREF(T) outb_r) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  const T* va = LPTR(const T, va_r); const T* vb = LPTR(const T, vb_r);
  T* outa = LPTR(T, outa_r); T* outb = LPTR(T, outb_r);
  WAVE_FN
  PHASE {
    const int b = lane;
    if (b < M.nbody) {
      T acca[6] = {0, 0, 0, 0, 0, 0}, accb[6] = {0, 0, 0, 0, 0, 0};
      unsigned long long m = M.body_dofmask[b];
      while (m) {
        const int d = myo_ffsll(m);
        m &= m - 1;
        const T xa = va[d], xb = vb[d];
        for (int e = 0; e < 6; ++e) { const T cd = s.cdof[6 * d + e]; acca[e] += cd * xa; accb[e] += cd * xb; }
      }
      for (int e = 0; e < 6; ++e) { outa[6 * b + e] = acca[e]; outb[6 * b + e] = accb[e]; }
    }
  }
  SYNC();
}
template <bool GEN, typename T, int NC>
DEVFN void J_times2(const DevModel<T>& M_in, const Scratch<T, NC>& s_in, LCREF(T) va_r, LCREF(T) bva_r, LREF(T) outa_r,
                    LCREF(T) vb_r, LCREF(T) bvb_r, LREF(T) outb_r) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  const T* va = LPTR(const T, va_r); const T* bva = LPTR(const T, bva_r); T* outa = LPTR(T, outa_r);
  const T* vb = LPTR(const T, vb_r); const T* bvb = LPTR(const T, bvb_r); T* outb = LPTR(T, outb_r);
  WAVE_FN
  const int nl = s.nl, nlim = s.nl + s.ntl, nefc = s.nefc;
  PHASE {
    for (int r = lane; r < nefc; r += 64) {
      T vala, valb;
      if (r < nl) { const T sg = lim_sign<T>(s.lim_id[r]); const int id = lim_index(s.lim_id[r]); vala = sg * va[id]; valb = sg * vb[id]; }
      else if (r < nlim) {
        const int t = lim_index(s.lim_id[r]);
        unsigned long long m = M.tendon_dofmask[t];
        T acca = 0, accb = 0, tj[MYO_TJ_MAX];
        tenj_row(s, t, tj);
#pragma unroll
        for (int k = 0; k < MYO_TJ_MAX; ++k) if (m) { const int d = myo_ffsll(m); m &= m - 1; acca += tj[k] * va[d]; accb += tj[k] * vb[d]; }
        vala = lim_sign<T>(s.lim_id[r]) * acca; valb = lim_sign<T>(s.lim_id[r]) * accb;
      } else {
        const int ci = (r - nlim) >> 2, e = (r - nlim) & 3;
        const auto& c = CON(s, ci);
        T v1[3], v2[3];
        point_vel(bva, con_b1(c), c.r1, v1);
        point_vel(bva, con_b2(c), c.r2, v2);
        const T rela[3] = {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]};
        point_vel(bvb, con_b1(c), c.r1, v1);
        point_vel(bvb, con_b2(c), c.r2, v2);
        const T relb[3] = {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]};
        if constexpr (!GEN) {
          T t2[3], fr[6];
          con_frame(c, fr);
          con_t2(fr, t2);
          const T* fn = fr;
          const T ft[3] = {(e >> 1) ? t2[0] : fr[3], (e >> 1) ? t2[1] : fr[4], (e >> 1) ? t2[2] : fr[5]};
          const T mu = (e & 1) ? -c.muA : c.muA;
          vala = dot3(fn, rela) + mu * dot3(ft, rela);
          valb = dot3(fn, relb) + mu * dot3(ft, relb);
        } else {
          const int kind = con_kind(c);
          const T* a1 = bva + 6 * con_b1(c); const T* a2 = bva + 6 * con_b2(c); const T* b1 = bvb + 6 * con_b1(c); const T* b2 = bvb + 6 * con_b2(c);
          const T raa[3] = {a2[0] - a1[0], a2[1] - a1[1], a2[2] - a1[2]}, rab[3] = {b2[0] - b1[0], b2[1] - b1[1], b2[2] - b1[2]};
          vala = con_row_val(c, kind, e, rela, raa);
          valb = con_row_val(c, kind, e, relb, rab);
        }
      }
      outa[r] = vala; outb[r] = valb;
    }
  }
  SYNC();
}

// out = J' f  (lanes = dofs; contacts act as a world force at the contact point)
template <typename T, int NC>
DEV void JT_times(const DevModel<T>& M_in, Scratch<T, NC>& s_in, LCREF(T) f_r, LREF(T) out_r) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  const T* f = LPTR(const T, f_r); T* out = LPTR(T, out_r);
  WAVE_FN_K
  const int nl = s.nl, nlim = s.nl + s.ntl, ncon = s.ncon, nrot = M.any_rot;      // any_rot: the model has condim 4 / 6 pairs (wave-uniform)
  PHASE {
    for (int ci = lane; ci < ncon; ci += 64) {
      const auto& c = CON(s, ci);
      const T* fe = f + nlim + 4 * ci;
      const T fn = fe[0] + fe[1] + fe[2] + fe[3], fa = c.muA * (fe[0] - fe[1]), fb = c.muB * (fe[2] - fe[3]);      // (padding rows carry no force)
      T t2[3], fr[6];
      con_frame(c, fr);
      con_t2(fr, t2);
      // world force at the contact point; slots whose pairs are rotations (kinds 1, 2, 4) add a world TORQUE instead of the tangential force
      if (!M.any_gen) {
        for (int k = 0; k < 3; ++k) S_CONF(s)[3 * ci + k] = fr[k] * fn + fr[3 + k] * fa + t2[k] * fb;
      } else {
        const int kind = con_kind(c);
        for (int k = 0; k < 3; ++k) S_CONF(s)[3 * ci + k] = fr[k] * fn + (kind == 0 ? fr[3 + k] * fa + t2[k] * fb : (T)0);
        if (nrot > 0)
          for (int k = 0; k < 3; ++k)
            S_CONTQ(s)[3 * ci + k] = kind == 1 ? fr[k] * fa + fr[3 + k] * fb : (kind == 2 ? t2[k] * fa : (kind == 4 ? fr[k] * fa : (T)0));
      }
    }
  }
  SYNC();
  PHASE {
    const int d = lane;
    if (d < M.nv) {
      T acc = 0;                                 // (joint-limit / friction-loss rows: below, one lane per row)
      // (four rows at a time: the fp64 stepper's moment arms come from global memory, so the reads of a group are requested together;
      //  the sum keeps its row order)
      for (int r0 = nl; r0 < nlim; r0 += 4) {
        T jv[4], fv[4];
        int on[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = r0 + k < nlim ? r0 + k : r0;                          // (rows beyond the last: a valid read, dropped below)
          const int t = lim_index(s.lim_id[r]);
          const unsigned long long m = M.tendon_dofmask[t];
          on[k] = (r0 + k < nlim) && ((m >> d) & 1ull);
          const int slot = on[k] ? myo_popcll(m & ((1ull << d) - 1ull)) : 0;  // slot 0 is always a valid read
          jv[k] = tenj_get(s, t, slot);
          fv[k] = lim_sign<T>(s.lim_id[r]) * f[r];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { const T v = fv[k] * jv[k]; acc += on[k] ? v : (T)0; }
      }
      out[d] = acc;
    }
  }
  SYNC();
  // contacts: one lane per (contact, support entry) — four contacts a pass — adding the entry's share  +- column . force  into its dof
  // with an LDS add.  (Rounds 1-5: every dof lane walked all contacts, one LDS round trip + two scalar loads of the bodies' dof masks
  // per contact; the support list carries the same "moves body 1 / body 2" bits.)
  static_assert(MYO_CS_MAX == 16, "four contacts x sixteen support entries per pass");
  PHASE {
    for (int r = lane; r < nl; r += 64) lds_add(&out[lim_index(s.lim_id[r])], lim_sign<T>(s.lim_id[r]) * f[r]);      // a joint row moves one dof
    const int e = lane & 15;
    for (int ci = lane >> 4; ci < ncon; ci += 4) {
      const auto& c = CON(s, ci);
      if (e < con_nsup(c)) {
        const int sd = c.sup[e];
        const int d = con_sup_dof(sd), on1 = con_sup_on1(sd), on2 = con_sup_on2(sd);
        if (on1 != on2) {
          T col[3];
          con_col(s, d, on2 ? c.r2 : c.r1, col);
          T v = dot3(col, S_CONF(s) + 3 * ci);
          if (nrot > 0) v += dot3(s.cdof + 6 * d, S_CONTQ(s) + 3 * ci);      // (scalar branch) torques of the rotational slots
          lds_add(&out[d], on2 ? v : -v);
        }
      }
    }
  }
  SYNC();
}

// ------------------------------------------------------------------------------------------
// P8: velocity stage (mj_fwdVelocity): tendon/actuator velocity, comVel, passive, RNE bias, aref
template <typename T, int NC>
DEVFN void fwd_velocity(const DevModel<T>& M_in, const TaskDev& K_in, Scratch<T, NC>& s_in) {
  MYO_BIND_M(T) MYO_BIND_K MYO_BIND_S(T)
  WAVE_FN
  const T* const qv = S_QVELT(s);
  // (the caller has run body_vectors(qvel -> S_CVEL): called from kernel level so that this function stays a leaf)
  PHASE {
    const int t = lane;
    const int d = lane;
    // (every table word and moment arm of this lane first, together: see fwd_actuation)
    const int tt = t < M.ntendon ? t : 0, dc = d < M.nv ? d : 0;
    unsigned long long tmask = M.tendon_dofmask[tt], pmask = M.dof_prevmask[dc];
    T tj[MYO_TJ_MAX];
    tenj_row(s, tt, tj);
    T damp = M.dof_damping[dc], spr_k = M.dof_spr[2 * dc], spr_q0 = M.dof_spr[2 * dc + 1];
    int spr_qa = M.dk_i[2 * dc];
    MYO_PIN(tmask); MYO_PIN(pmask); MYO_PIN(damp); MYO_PIN(spr_k); MYO_PIN(spr_q0); MYO_PIN(spr_qa);
#pragma unroll
    for (int k = 0; k < MYO_TJ_MAX; ++k) MYO_PIN(tj[k]);
    if (t < M.ntendon) {
      unsigned long long m = tmask;
      T acc = 0;
#pragma unroll
      for (int k = 0; k < MYO_TJ_MAX; ++k) if (m) { const int d = myo_ffsll(m); m &= m - 1; acc += tj[k] * qv[d]; }
      S_TEN_VEL(s)[t] = acc;
    }
    if (d < M.nv) {
      // cdof_dot = (velocity accumulated before this dof) x cdof   (mj_comVel)
      T cv[6] = {0, 0, 0, 0, 0, 0};
      unsigned long long m = pmask;
      const int is_free_trans = (int)((m >> 63) & 1ull);
      m &= ~(1ull << 63);
      while (m) {
        const int o = myo_ffsll(m);
        m &= m - 1;
        const T vo = qv[o];
        for (int e = 0; e < 6; ++e) cv[e] += s.cdof[6 * o + e] * vo;
      }
      if (is_free_trans) { for (int e = 0; e < 6; ++e) S_CDOFDOT(s)[6 * d + e] = 0; }
      else cross_motion(S_CDOFDOT(s) + 6 * d, cv, s.cdof + 6 * d);
      // passive joint forces (dof_spr: the joint's stiffness — 0 for none / a free joint — and spring reference, dk_i: its qpos address)
      T p = -damp * qv[d];
      if (spr_k != 0) p -= spr_k * ((T)s.qpos[spr_qa] - spr_q0);
      S_QFRC_PASSIVE(s)[d] = p;
    }
  }
  SYNC();
  if (M.any_tendon_passive) {
    PHASE {
      const int d = lane;
      if (d < M.nv) {
        T acc = 0;
        for (int t = 0; t < M.ntendon; ++t) {
          const T k = M.tendon_stiffness[t], b = M.tendon_damping[t];
          const unsigned long long m = M.tendon_dofmask[t];
          if ((k != 0 || b != 0) && ((m >> d) & 1ull)) {
            const T f = -k * ((T)S_TEN_LENGTH(s)[t] - M.tendon_lengthspring[t]) - b * S_TEN_VEL(s)[t];
            acc += tenj_get(s, t, myo_popcll(m & ((1ull << d) - 1ull))) * f;
          }
        }
        S_QFRC_PASSIVE(s)[d] += acc;
      }
    }
    SYNC();
  }
  // RNE with zero acceleration: body forces
  PHASE {
    const int b = lane;
    if (b < M.nbody) {
      T a[6] = {0, 0, 0, -M.gravity[0], -M.gravity[1], -M.gravity[2]};
      unsigned long long m = M.body_dofmask[b];
      while (m) {
        const int d = myo_ffsll(m);
        m &= m - 1;
        const T vd = qv[d];
        for (int e = 0; e < 6; ++e) a[e] += S_CDOFDOT(s)[6 * d + e] * vd;
      }
      T t1[6], t2[6], t3[6];
      mul_inert_vec(t1, S_CINERT(s) + 10 * b, a);
      mul_inert_vec(t2, S_CINERT(s) + 10 * b, S_CVEL(s) + 6 * b);
      cross_force(t3, S_CVEL(s) + 6 * b, t2);
      for (int e = 0; e < 6; ++e) S_CFRCB(s)[6 * b + e] = (b == 0) ? (T)0 : t1[e] + t3[e];
    }
  }
  SYNC();
  PHASE {
    const int d = lane;
    if (d < M.nv) {
      T f[6] = {0, 0, 0, 0, 0, 0};
      const unsigned long long sub = M.dof_submask[d];
      // all bodies, membership as a select: the reads are wave-uniform and all in flight together, instead of
      // one dependent round trip per subtree body on the lanes of the root dofs (18 bodies for the wrist)
#pragma unroll 4
      for (int c = 1; c < M.nbody; ++c) {
        const T in = ((sub >> c) & 1ull) ? (T)1 : (T)0;
        for (int e = 0; e < 6; ++e) f[e] += in * S_CFRCB(s)[6 * c + e];
      }
      T acc = 0;
      for (int e = 0; e < 6; ++e) acc += s.cdof[6 * d + e] * f[e];
      S_QFRC_BIAS(s)[d] = acc;
    }
  }
  SYNC();
  (void)K;
}

// reference acceleration of every constraint row: aref = -B vel - K imp (pos - margin).  Inlined at kernel
// level (needs S_CVEL from body_vectors(qvel)), so that fwd_velocity and J_times are both leaf functions.
template <typename T, int NC>
DEV void efc_reference(const DevModel<T>& M_in, Scratch<T, NC>& s_in) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  WAVE_FN_K
  if (s.nefc > 0) {
    J_times(M, s, LOFF(s, S_QVELT(s)), LOFF(s, S_CVEL(s)), LOFF(s, s.efc_jv));
    const int gen = M.any_gen;
    PHASE {
      const int nlim_ = s.nl + s.ntl;
      for (int r = lane; r < s.nefc; r += 64) {
        const T Bc = S_ROW_B(s)[r], kp = S_ROW_KIP(s)[r];      // (left in the row's own entries by the lane that built the row)
        const bool pad = gen && r >= nlim_ && con_pad(con_kind(CON(s, (r - nlim_) >> 2)), (r - nlim_) & 3);
        S_AREF(s)[r] = pad ? (T)-1 : -Bc * s.efc_jv[r] - kp;       // a padding row: J = 0, so J a - aref = 1 > 0, never active
      }
    }
    SYNC();
  }
}

// ------------------------------------------------------------------------------------------
// P9: muscle actuation (mj_fwdActuation, mju_muscleDynamics/Gain/Bias)
template <typename T> DEV T muscle_FL(T L, T lmin, T lmax) {
  if (L < lmin || L > lmax) return 0;
  const T a = (T)0.5 * (lmin + 1), b = (T)0.5 * (1 + lmax);
  T x;
  if (L <= a) { x = (L - lmin) / tmax(MYO_MINVAL, a - lmin); return (T)0.5 * x * x; }
  if (L <= 1) { x = (1 - L) / tmax(MYO_MINVAL, 1 - a); return 1 - (T)0.5 * x * x; }
  if (L <= b) { x = (L - 1) / tmax(MYO_MINVAL, b - 1); return 1 - (T)0.5 * x * x; }
  x = (lmax - L) / tmax(MYO_MINVAL, lmax - b);
  return (T)0.5 * x * x;
}

template <typename T, int NC>
DEVFN void fwd_actuation(const DevModel<T>& M_in, Scratch<T, NC>& s_in) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  WAVE_FN
  LANE_VAR(T, adot);               // act_dot shares its LDS slot with ten_length: kept per lane until every length has been read
  PHASE {
    const int i = lane;
    LV(adot) = 0;
    // Every table word of this lane is requested HERE, unconditionally and together (lanes beyond nu read actuator 0), and pinned
    // (MYO_PIN) so that the compiler does not sink the loads back into the branches that use them: a lane-indexed table read is a
    // vector memory load, ~500 cycles, and rounds 1-5 issued ~35 of them one after the other — each inside the branch of the muscle
    // curves that needs it, each followed by its own s_waitcnt: 12 k of a substep's 250 k cycles for ~200 flops.
    const int ii = i < M.nu ? i : 0;
    int ctrllim = M.actuator_ctrllimited[ii], dyntype = M.actuator_dyntype[ii], tid = M.actuator_tendon[ii], gaintype = M.actuator_gaintype[ii];
    int biastype = M.actuator_biastype[ii], forcelim = M.actuator_forcelimited[ii];
    T cr0 = M.actuator_ctrlrange[2 * ii], cr1 = M.actuator_ctrlrange[2 * ii + 1], fr0 = M.actuator_forcerange[2 * ii], fr1 = M.actuator_forcerange[2 * ii + 1];
    T gear = M.actuator_gear[6 * ii], gear0 = M.act_gear0[ii], dp0 = M.actuator_dynprm[10 * ii], dp1 = M.actuator_dynprm[10 * ii + 1];
    T gp0 = M.actuator_gainprm[10 * ii], bp0 = M.actuator_biasprm[10 * ii], bp1 = M.actuator_biasprm[10 * ii + 1], bp2 = M.actuator_biasprm[10 * ii + 2];
    HP ap[MYO_ACT_PRE];
#pragma unroll
    for (int k = 0; k < MYO_ACT_PRE; ++k) ap[k] = M.h_act_pre[MYO_ACT_PRE * ii + k];
    T ctrl = ctrl_get(s, ii);
    MYO_PIN(ctrllim); MYO_PIN(dyntype); MYO_PIN(tid); MYO_PIN(gaintype); MYO_PIN(biastype); MYO_PIN(forcelim);
    MYO_PIN(cr0); MYO_PIN(cr1); MYO_PIN(fr0); MYO_PIN(fr1); MYO_PIN(gear); MYO_PIN(gear0); MYO_PIN(dp0); MYO_PIN(dp1);
    MYO_PIN(gp0); MYO_PIN(bp0); MYO_PIN(bp1); MYO_PIN(bp2); MYO_PIN(ctrl);
#pragma unroll
    for (int k = 0; k < MYO_ACT_PRE; ++k) MYO_PIN(ap[k]);
    const int ia = i - (M.nu - M.na);                      // the actuator's activation (muscles: the last na actuators)
    const T act_in = (T)S_ACT(M, s)[(ia >= 0 && ia < M.na) ? ia : 0];
    const HP ten_len = S_TEN_LENGTH(s)[tid];
    const T ten_vel = S_TEN_VEL(s)[tid];
    if (i < M.nu) {
      if (ctrllim) ctrl = tclamp(ctrl, cr0, cr1);
      T input = ctrl;
      // The constants of mju_muscleGain / Bias / Dynamics that depend on the parameters alone — peak force, L0 and the normalised-length
      // map, the reciprocals of every constant denominator of the force-length / force-velocity / passive curves — are resolved on the
      // host (act_pre, MYO_ACT_PRE per actuator: myobatch.hip); round 4's version divided 16 times per lane here.
      if (dyntype == 3) {
        const T act = act_in;
        const T cc = tclamp(ctrl, (T)0, (T)1), ac = tclamp(act, (T)0, (T)1);
        const T w = (T)0.5 + (T)1.5 * ac, ideact = (T)ap[22];
        // tau = tau_act w (activating) | tau_deact / w (deactivating); act_dot = (ctrl - act) / tau
        if (cc > act) LV(adot) = (cc - act) / tmax(MYO_MINVAL, dp0 * w);
        else if (ideact != 0) LV(adot) = (cc - act) * w * ideact;
        else LV(adot) = (cc - act) / tmax(MYO_MINVAL, dp1 / w);
        input = act;
      }
      // muscle length normalisation and the force-length curves in HP from the HP tendon length (the curves are
      // piecewise quadratics of DIFFERENCES like L - 1 and L - lmin); the force-velocity factor in T
      const HP lenh = (HP)gear * ten_len;
      const T len = (T)lenh, vel = gear * ten_vel;
      T gain, bias = 0;
      if (gaintype == 1) {
        const HP L = ap[1] + (lenh - ap[3]) * ap[2];
        const T V = vel * (T)ap[4];
        const HP lmin = ap[5], lmax = ap[6], a = ap[7], b = ap[8];
        HP FLh;
        if (L < lmin || L > lmax) FLh = 0;
        else if (L <= a) { const HP x = (L - lmin) * ap[9]; FLh = (HP)0.5 * x * x; }
        else if (L <= 1) { const HP x = (1 - L) * ap[10]; FLh = 1 - (HP)0.5 * x * x; }
        else if (L <= b) { const HP x = (L - 1) * ap[11]; FLh = 1 - (HP)0.5 * x * x; }
        else { const HP x = (lmax - L) * ap[12]; FLh = (HP)0.5 * x * x; }
        const T FL = (T)FLh, y = (T)ap[13], fvmax = (T)ap[15];
        T FV;
        if (V <= -1) FV = 0;
        else if (V <= 0) FV = (V + 1) * (V + 1);
        else if (V <= y) FV = fvmax - (y - V) * (y - V) * (T)ap[14];
        else FV = fvmax;
        gain = -(T)ap[0] * FL * FV;
      } else gain = gp0;
      if (biastype == 2) {
        const HP L = ap[17] + (lenh - ap[3]) * ap[18];
        const HP b = ap[19];
        const T force = (T)ap[16], fpmax = (T)ap[21];
        if (L <= 1) bias = 0;
        else if (L <= b) { const T x = (T)((L - 1) * ap[20]); bias = -force * fpmax * (T)0.5 * x * x; }
        else { const T x = (T)((L - b) * ap[20]); bias = -force * fpmax * ((T)0.5 + x); }
      } else if (biastype == 1) {
        bias = bp0 + bp1 * len + bp2 * vel;
      }
      T f = gain * input + bias;
      if (forcelim) f = tclamp(f, fr0, fr1);
      S_ACT_FORCE(s)[i] = f;
      S_ACT_GF(s)[i] = gear0 * f;
    }
  }
  SYNC();
  // qfrc_actuator = moment' * force.  lane = dof; its (ten_J offset, actuator) pairs arrive in one
  // wide load (host-built dof-major table, actuator order), then two LDS reads per entry
  if constexpr (sizeof(T) == sizeof(HP)) {
    // fp64 stepper (moment arms in global memory): one lane per (actuator, slot) — the reads coalesce — adding into the dof's entry
    PHASE {
      if (lane < M.nu && M.actuator_dyntype[lane < M.nu ? lane : 0] == 3) S_ACT_DOT(s)[lane - (M.nu - M.na)] = LV(adot);
      if (lane < M.nv) S_QFRC_ACTUATOR(s)[lane] = 0;
    }
    if constexpr (Scratch<T, NC>::SPILL) { SYNC_G(); } else { SYNC(); }      // (SPILL: the rates are in global memory)
    PHASE {
      // (table words of all the lane's items first, then the moment arms, then the adds: two memory round trips, not two per item)
      constexpr int NIT = (MYO_TJ_MAX * MYO_NU_MAX + 63) / 64;
      static_assert(MYO_TJ_MAX * MYO_NU_MAX % 64 == 0, "whole passes");
      int dd[NIT], tt[NIT];
      T jj[NIT];
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int o = lane + 64 * k, slot = o / MYO_NU_MAX, i = o - slot * MYO_NU_MAX;
        dd[k] = M.act_sd[i * MYO_TJ_MAX + slot];                  // dof of the actuator's tendon's slot, -1: none
        tt[k] = M.act_tj[i];
      }
#pragma unroll
      for (int k = 0; k < NIT; ++k) { MYO_PIN(dd[k]); MYO_PIN(tt[k]); }
#pragma unroll
      for (int k = 0; k < NIT; ++k) { const int o = lane + 64 * k, slot = o / MYO_NU_MAX; jj[k] = tenj_get(s, tt[k] / MYO_TJ_MAX, slot); }
#pragma unroll
      for (int k = 0; k < NIT; ++k) MYO_PIN(jj[k]);
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int o = lane + 64 * k, slot = o / MYO_NU_MAX, i = o - slot * MYO_NU_MAX;
        if (dd[k] >= 0) lds_add(S_QFRC_ACTUATOR(s) + dd[k], jj[k] * S_ACT_GF(s)[i]);
      }
    }
    SYNC();
    PHASE {
      const int d = lane;
      if (d < M.nv) s.qfrc_smooth[d] = S_QFRC_PASSIVE(s)[d] - S_QFRC_BIAS(s)[d] + S_QFRC_ACTUATOR(s)[d];
    }
    SYNC();
    return;
  }
  PHASE {
    if (lane < M.nu && M.actuator_dyntype[lane < M.nu ? lane : 0] == 3) S_ACT_DOT(s)[lane - (M.nu - M.na)] = LV(adot);
    const int d = lane;
    if (d < M.nv) {
      unsigned w[MYO_AQ_ROW / 2];
#pragma unroll
      for (int q = 0; q < MYO_AQ_ROW / 2; ++q) w[q] = (unsigned)M.aq_pack[d * (MYO_AQ_ROW / 2) + q];
      const int len = M.aq_len[d];
      T acc = 0;
#pragma unroll
      for (int k = 0; k < MYO_AQ_ROW; ++k) {
        const unsigned ent = (k & 1) ? (w[k / 2] >> 16) : (w[k / 2] & 0xffffu);     // padding entries are 0: valid reads
        const T j = tenj_get(s, (int)(ent >> 6) / MYO_TJ_MAX, (int)(ent >> 6) % MYO_TJ_MAX), f = S_ACT_GF(s)[ent & 63u];
        acc += (k < len) ? j * f : (T)0;
      }
      S_QFRC_ACTUATOR(s)[d] = acc;
      s.qfrc_smooth[d] = S_QFRC_PASSIVE(s)[d] - S_QFRC_BIAS(s)[d] + acc;
    }
  }
  SYNC();
}

// ------------------------------------------------------------------------------------------
// P10: Newton solver on the primal problem (mj_solNewton)
// The COST is accumulated in HP in every build: Newton's termination test is a cost DIFFERENCE of ~1e-8 on a
// cost of ~1e2, and near the minimum the cost is flat to second order, so an fp32 cost stalls while the
// iterate is still sqrt(eps) away.  The iterates, gradient and search direction stay T.
// what a friction-loss row's cost exceeds the ordinary rows' 0.5 D min(x, 0)^2 by (0 for the other rows): the sums below stay as
// they are and models with friction loss add this on top
template <typename T, int NC> DEV T fric_excess(const DevModel<T>& M, const Scratch<T, NC>& s, int r, T x) {
  if (!lim_is_fric(s.lim_id[r])) return (T)0;
  const T fl = r < s.nl ? M.dof_frictionloss[lim_index(s.lim_id[r])] : M.tendon_frictionloss[lim_index(s.lim_id[r])];
  T force; int quad;
  const T c = fric_cost(s.efc_D[r], fl, x, &force, &quad);
  const T xm = tmin(x, (T)0);
  return c - (T)0.5 * s.efc_D[r] * xm * xm;
}

// friction-loss rows after a change of jar: clamped force, "active" = the quadratic zone (what the Hessian sees); returns the
// rows' excess cost (fric_excess) over the ordinary formula
template <typename T, int NC>
DEVFN HP fric_update(const DevModel<T>& M_in, Scratch<T, NC>& s_in) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  WAVE_FN
  const int nlim = s.nl + s.ntl;
  PHASE {
    for (int r = lane; r < nlim; r += 64) {
      if (!lim_is_fric(s.lim_id[r])) continue;
      const T fl = r < s.nl ? M.dof_frictionloss[lim_index(s.lim_id[r])] : M.tendon_frictionloss[lim_index(s.lim_id[r])];
      T force; int quad;
      (void)fric_cost(s.efc_D[r], fl, s.efc_jar[r], &force, &quad);
      s.efc_force[r] = force; s.efc_active[r] = (unsigned char)quad;
    }
  }
  SYNC();
  WAVE_SUM_N(HP, fc, nlim, r, (HP)fric_excess(M, s, r, s.efc_jar[r]));
  return fc;
}
// sum of the friction-loss rows' excess cost at jar = x - aref (the warm-start comparison)
template <typename T, int NC>
DEVFN HP fric_excess_sum(const DevModel<T>& M_in, const Scratch<T, NC>& s_in, LCREF(T) x_r) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  WAVE_FN
  const T* x = LPTR(const T, x_r);
  const int nlim = s.nl + s.ntl;
  WAVE_SUM_N(HP, fc, nlim, r, (HP)fric_excess(M, s, r, x[r] - S_AREF(s)[r]));
  return fc;
}
// the exact line search for models with friction-loss rows: the same safeguarded Newton iteration on p'(alpha) as newton_solve's, the
// rows read from LDS every trip (the register-resident version is the condim-3 / no-friction-loss fast path)
template <typename T, int NC>
DEVFN T fric_linesearch(const DevModel<T>& M_in, const Scratch<T, NC>& s_in, T q1, T q2, T gtol) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  WAVE_FN
  const int nefc = s.nefc, nlim = s.nl + s.ntl;
  T alpha = 0, lo = 0, hi = -1;
  for (int li = 0; li < 50; ++li) {
    T a1 = 0, a2 = 0;
    PHASE { (void)lane; }
    WAVE_SUM3_N(T, e1, e2, e3unused, nefc, r, {
      const T v = s.efc_jv[r], x = s.efc_jar[r] + alpha * v, D = row_D(s, r, nlim);
      if (r < nlim && lim_is_fric(s.lim_id[r])) {
        const T fl = r < s.nl ? M.dof_frictionloss[lim_index(s.lim_id[r])] : M.tendon_frictionloss[lim_index(s.lim_id[r])];
        T force; int quad;
        (void)fric_cost(D, fl, x, &force, &quad);
        _e1 = -force * v; _e2 = quad ? D * v * v : (T)0;
      } else if (x < 0) { _e1 = D * x * v; _e2 = D * v * v; }
    });
    (void)e3unused; (void)a1; (void)a2;
    const T d1 = 2 * alpha * q2 + q1 + e1, d2 = 2 * q2 + e2;
    if (fabs(d1) < gtol) break;
    if (d1 < 0) lo = alpha; else hi = alpha;
    T next = alpha - d1 / d2;
    if (hi >= 0 && (next <= lo || next >= hi)) next = (T)0.5 * (lo + hi);
    if (next == alpha) break;
    alpha = next;
  }
  return alpha;
}

template <typename T, int NC>
DEV HP update_constraint(const DevModel<T>& M_in, Scratch<T, NC>& s_in) {   // inlined into the solver loop (kernel level): JT_times stays a leaf call
  MYO_BIND_M(T) MYO_BIND_S(T)
  // forces / active set from jar, cost, qfrc_constraint, gradient
  WAVE_FN_K
  const int nefc = s.nefc, nlim = s.nl + s.ntl;
  PHASE {
    for (int r = lane; r < nefc; r += 64) {
      const T x = s.efc_jar[r];
      const unsigned char a = x < 0;
      s.efc_active[r] = a;
      const T Dr = row_D(s, r, nlim);
      s.efc_force[r] = a ? -Dr * x : (T)0;
    }
  }
  SYNC();
  PROF(s, 21)
  HP fcorr = 0;
  if (M.any_floss) fcorr = fric_update(M, s);      // (scalar branch; a leaf of its own: the kernel-level solver loop is at its register budget)
  JT_times(M, s, LOFF(s, s.efc_force), LOFF(s, s.qfrc_constraint));
  PROF(s, 22)
  WAVE_SUM_N(HP, ccost0, nefc, r, ((HP)0.5 * (HP)row_D(s, r, nlim) * (HP)tmin(s.efc_jar[r], (T)0) * (HP)tmin(s.efc_jar[r], (T)0)));   // active <=> jar < 0
  const HP ccost = ccost0 + fcorr;
  WAVE_SUM_N(HP, gcost, M.nv, c, (((HP)s.Ma[c] - (HP)s.qfrc_smooth[c]) * ((HP)s.qacc[c] - (HP)s.qacc_smooth[c])));
  PHASE {
    const int c = lane;
    if (c < M.nv) s.search[c] = -(s.Ma[c] - s.qfrc_smooth[c] - s.qfrc_constraint[c]);      // -gradient: the right-hand side of the next Newton system
  }
  SYNC();
  PROF(s, 23)
  return ccost + (HP)0.5 * gcost;
}

template <typename T, int NC>
DEV void build_hessian(const DevModel<T>& M_in, Scratch<T, NC>& s_in) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  WAVE_FN_K
  // (the caller has loaded M into H: load_H_from_M is called from kernel level so that this function stays a leaf)
  const int nl = s.nl, nlim = s.nl + s.ntl;
  // joint-limit rows touch one diagonal entry each: lanes = rows, LDS adds (a dof's lower and upper limit row — or its friction-loss
  // row — meet in one entry; rounds 1-5 had every dof lane walk all nl rows, one LDS round trip a row)
  PHASE {
    for (int r = lane; r < nl; r += 64) {
      if (s.efc_active[r]) { const int pd = s.hperm[lim_index(s.lim_id[r])]; lds_add(&s.H[MYO_HIDX(pd, pd)], s.efc_D[r]); }
    }
  }
  SYNC();
  // tendon-limit rows: rank-1 blocks over the tendon's dofs (one row at a time)
  for (int r = nl; r < nlim; ++r) {
    if (!s.efc_active[r]) continue;
    PHASE {
      const int t = lim_index(s.lim_id[r]);
      const unsigned long long m = M.tendon_dofmask[t];
      const int n = myo_popcll(m);
      const int a = lane >> 3, b = lane & 7;
      if (a < n && b <= a) {
        // a-th / b-th set bits
        unsigned long long ma = m, mb = m;
        for (int k = 0; k < a; ++k) ma &= ma - 1;
        for (int k = 0; k < b; ++k) mb &= mb - 1;
        const int da = myo_ffsll(ma), db = myo_ffsll(mb);
        const int pa = s.hperm[da], pb = s.hperm[db];
        s.H[MYO_HIDX(pa > pb ? pa : pb, pa > pb ? pb : pa)] += s.efc_D[r] * tenj_get(s, t, a) * tenj_get(s, t, b);
      }
    }
    SYNC();
  }
  // contacts: H += Jp' (R' A R) Jp with A = D sum_active w w'  (3x3 per contact, frame coordinates).
  // Two stages per contact: (A) the (<= 16) lanes of its support set put its Jacobian columns, rotated into the contact frame and
  // signed, into a staging buffer — each column is evaluated once per contact instead of once per pair; (B) the pair lanes add the
  // block.  bvec is free here (body vectors are rebuilt after the solve).
  T* stage = s.bvec;                       // 2 x MYO_CS_MAX x 4: the column in frame coordinates and, in the fourth slot, its row of H
  static_assert(2 * MYO_CS_MAX * 4 <= MYO_NB_MAX * 6 && sizeof(T) >= sizeof(int), "staging fits in bvec");
  const int ncon = s.ncon, gen = M.any_gen;
  // the lower-triangular pairs (a >= b) of a support set, enumerated linearly: q = a(a+1)/2 + b — the same for every contact, so
  // a lane decodes its (at most three) pairs once
  static_assert(MYO_CS_MAX * (MYO_CS_MAX + 1) / 2 <= 192, "three pairs per lane");
  LANE_VAR(int, pa0); LANE_VAR(int, pa1); LANE_VAR(int, pa2);      // a | b << 8
  PHASE {
    int pk3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int q = lane + 64 * k;
      int a = (int)((sqrtf((float)(8 * q + 1)) - 1.0f) * 0.5f);
      if ((a + 1) * (a + 2) / 2 <= q) a++;           // float sqrt rounding guard
      if (a * (a + 1) / 2 > q) a--;
      pk3[k] = a | ((q - a * (a + 1) / 2) << 8);
    }
    LV(pa0) = pk3[0]; LV(pa1) = pk3[1]; LV(pa2) = pk3[2];
  }
  // Two contacts a trip: stage A for both at once (lanes 0-15 / 16-31: a support set has at most MYO_CS_MAX = 16 entries), then their
  // stage B one after the other — H takes the contacts' blocks in index order, as before.
  static_assert(MYO_CS_MAX <= 16, "two support sets side by side in stage A");
  for (int c0 = 0; c0 < ncon; c0 += 2) {
    PHASE {
      // ---- stage A for contacts c0, c0 + 1
      const int half = lane >> 4, sl = lane & 15, cn = c0 + half;
      if (lane < 32 && cn < ncon && sl < con_nsup(CON(s, cn))) {
        const auto& c = CON(s, cn);
        const int sd = c.sup[sl];
        const int d = con_sup_dof(sd), on2 = con_sup_on2(sd), on1 = con_sup_on1(sd);
        T col[3];
        con_col(s, d, on2 ? c.r2 : c.r1, col);
        T t2[3], fr[6];
        con_frame(c, fr);
        con_t2(fr, t2);
        T j[3] = {dot3(fr, col), dot3(fr + 3, col), dot3(t2, col)};
        if (gen) {                                             // (scalar branch) rotational / frictionless slots
          const int kind = con_kind(c);
          const T* ang = s.cdof + 6 * d;                       // angular part of the dof's motion axis
          j[1] = kind == 0 ? j[1] : ((kind == 1 || kind == 4) ? dot3(fr, ang) : (kind == 2 ? dot3(t2, ang) : (T)0));
          j[2] = kind == 0 ? j[2] : (kind == 1 ? dot3(fr + 3, ang) : (T)0);
        }
        if (!on2) { j[0] = -j[0]; j[1] = -j[1]; j[2] = -j[2]; }
        if (on1 == on2) { j[0] = 0; j[1] = 0; j[2] = 0; }   // moves both bodies or neither: no relative motion (its pairs add 0)
        T* dst = stage + half * (MYO_CS_MAX * 4) + 4 * sl;
        dst[0] = j[0]; dst[1] = j[1]; dst[2] = j[2];
        *reinterpret_cast<int*>(dst + 3) = (int)s.hperm[d];   // the dof's row of H: stage B reads it with the column, not through sup -> hperm
      }
    }
    SYNC();
    // ---- stage B for contacts c0, c0 + 1 (their columns were staged above), one after the other
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      PHASE {
        const int ci = c0 + half;
        if (ci < ncon) {
          const auto& c = CON(s, ci);
          const unsigned char* act = s.efc_active + nlim + 4 * ci;
          T nn = 0, n1 = 0, n2 = 0, a11 = 0, a22 = 0;
          if (act[0]) { nn += 1; n1 += c.muA; a11 += c.muA * c.muA; }
          if (act[1]) { nn += 1; n1 -= c.muA; a11 += c.muA * c.muA; }
          if (act[2]) { nn += 1; n2 += c.muB; a22 += c.muB * c.muB; }
          if (act[3]) { nn += 1; n2 -= c.muB; a22 += c.muB * c.muB; }
          if (nn != 0) {
            const T cD = CON_D(s, ci);
            const T A0 = cD * nn, A1 = cD * n1, A2 = cD * n2, A3 = cD * a11, A4 = cD * a22;
            const int ns = con_nsup(c);
            const T* jc = stage + half * (MYO_CS_MAX * 4);
            const int npair = ns * (ns + 1) / 2;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const int q = lane + 64 * k;
              if (q < npair) {
                const int pk = k == 0 ? LV(pa0) : (k == 1 ? LV(pa1) : LV(pa2));
                const int a = pk & 255, b = pk >> 8;
                const T ja[3] = {jc[4 * a], jc[4 * a + 1], jc[4 * a + 2]};
                const T jb[3] = {jc[4 * b], jc[4 * b + 1], jc[4 * b + 2]};
                const int pa = *reinterpret_cast<const int*>(jc + 4 * a + 3), pb = *reinterpret_cast<const int*>(jc + 4 * b + 3);
                const T Ajb0 = A0 * jb[0] + A1 * jb[1] + A2 * jb[2];
                const T Ajb1 = A1 * jb[0] + A3 * jb[1];
                const T Ajb2 = A2 * jb[0] + A4 * jb[2];
                // (an LDS add: the pairs of one contact are distinct entries, contacts follow each other in program order)
                lds_add(&s.H[MYO_HIDX(pa > pb ? pa : pb, pa > pb ? pb : pa)], ja[0] * Ajb0 + ja[1] * Ajb1 + ja[2] * Ajb2);
              }
            }
          }
        }
      }
      SYNC();
    }
  }
}

template <typename T, int NC>
DEV void newton_solve(const DevModel<T>& M_in, Scratch<T, NC>& s_in) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  WAVE_FN_K
  const int nv = M.nv, nefc = s.nefc;
  // ---- warm start: cheaper of qacc_warmstart and qacc_smooth.  The warm start is fetched into qacc, where the chosen vector ends up anyway
  PHASE { const int c = lane; if (c < nv) s.qacc[c] = warm_get(s, c); }
  SYNC();
  // (the second set of body vectors borrows Ma..Mv, which the solver has not started to use)
  body_vectors2(M, s, LOFF(s, s.qacc), LOFF(s, s.qacc_smooth), LOFF(s, s.bvec), LOFF(s, S_CVEL(s)));
  // (J qacc_warm lands in efc_force, which the solver writes only after the choice: efc_jar still holds aref)
  if (M.any_gen) J_times2<true>(M, s, LOFF(s, s.qacc), LOFF(s, s.bvec), LOFF(s, s.efc_force), LOFF(s, s.qacc_smooth), LOFF(s, S_CVEL(s)), LOFF(s, s.efc_jv));
  else J_times2<false>(M, s, LOFF(s, s.qacc), LOFF(s, s.bvec), LOFF(s, s.efc_force), LOFF(s, s.qacc_smooth), LOFF(s, S_CVEL(s)), LOFF(s, s.efc_jv));
  mul_M(M, s, LOFF(s, s.Ma), LOFF(s, s.qacc));
  const int nlim = s.nl + s.ntl;
  // cost of the violated rows, branch-free: 0.5 D min(x, 0)^2  (no conditional around the D load)
  WAVE_SUM_N(HP, costw_c, nefc, r, ((HP)0.5 * (HP)row_D(s, r, nlim) * (HP)tmin(s.efc_force[r] - S_AREF(s)[r], (T)0) * (HP)tmin(s.efc_force[r] - S_AREF(s)[r], (T)0)));
  WAVE_SUM_N(HP, costs, nefc, r, ((HP)0.5 * (HP)row_D(s, r, nlim) * (HP)tmin(s.efc_jv[r] - S_AREF(s)[r], (T)0) * (HP)tmin(s.efc_jv[r] - S_AREF(s)[r], (T)0)));
  WAVE_SUM_N(HP, gw, nv, c, (((HP)s.Ma[c] - (HP)s.qfrc_smooth[c]) * ((HP)s.qacc[c] - (HP)s.qacc_smooth[c])));
  HP fw = 0, fs = 0;
  if (M.any_floss) { fw = fric_excess_sum(M, s, LOFF(s, s.efc_force)); fs = fric_excess_sum(M, s, LOFF(s, s.efc_jv)); }
  const int use_warm = (costw_c + fw + (HP)0.5 * gw) < costs + fs;
  PHASE {
    const int c = lane;
    if (c < nv) {
      if (!use_warm) s.qacc[c] = s.qacc_smooth[c];
      if (!use_warm) s.Ma[c] = s.qfrc_smooth[c];  // M qacc_smooth = qfrc_smooth
    }
    for (int r = lane; r < nefc; r += 64) s.efc_jar[r] = (use_warm ? s.efc_force[r] : s.efc_jv[r]) - S_AREF(s)[r];
  }
  SYNC();
  if (!use_warm) mul_M(M, s, LOFF(s, s.Ma), LOFF(s, s.qacc));  // keep Ma exactly consistent with M*qacc
  PROF(s, 24)
  HP cost = update_constraint(M, s);
  const T scale = 1 / (M.meaninertia * (T)(nv > 1 ? nv : 1));
  int iter = 0;
  while (iter < M.iterations) {
    PROF(s, 11)
    REP(12,      // (the Newton system: H = M + J'DJ, factor, solve; bit 13: load M + the Hessian alone)
    REP(13,
    load_H_from_M(M, s, (const T*)0, (T)0, 1);
    PROF(s, 25)
    build_hessian(M, s);
    )
    PROF(s, 9)
#ifndef MYO_EMU
    if (M.arrow_nf > 0) {
      arrow_eliminate_blocks<T, NC>(LOFF(s, s.search));
      PROF(s, 26)
      chol_factor_solve_reg<T, MYO_ARROW_S, NC>(LOFF(s, s.Mv), MYO_ARROW_S);
      PROF(s, 27)
      arrow_finish<T, NC>(LOFF(s, s.search));
    } else
#endif
    chol_factor_solve(s, s.search, nv, (s.ncon == 0 && s.ntl == 0) ? M.nlead : nv);
    )
    PROF(s, 10)
    REP(14,      // (M search, body vectors, J search)
    mul_M(M, s, LOFF(s, s.Mv), LOFF(s, s.search));
    PROF(s, 28)
    body_vectors(M, s, LOFF(s, s.search), LOFF(s, s.bvec));
    PROF(s, 29)
    J_times(M, s, LOFF(s, s.search), LOFF(s, s.bvec), LOFF(s, s.efc_jv));
    )
    PROF(s, 30)
    WAVE_SUM3_N(T, q1, q2, sn2, nv, c, { _e1 = s.search[c] * (s.Ma[c] - s.qfrc_smooth[c]); _e2 = (T)0.5 * s.search[c] * s.Mv[c]; _e3 = s.search[c] * s.search[c]; });
    const T snorm = sqrt(sn2);
    PROF(s, 0)
#ifdef MYO_EMU_DEBUG
    printf("newton iter %d cost %g q1 %g q2 %g snorm %g nefc %d\n", iter, (double)cost, (double)q1, (double)q2, (double)snorm, nefc);
#endif
    if (snorm < MYO_MINVAL) break;
    const T gtol = M.tolerance * (T)0.01 * snorm / scale;
    // exact 1-D minimisation of the convex piecewise-quadratic: safeguarded Newton on p'(alpha)
    T alpha = 0, lo = 0, hi = -1;
    // each lane keeps its (<= 3) rows' jar, jv and D in registers for the whole line search
    static_assert(MYO_NLIM_MAX + 4 * NC <= 256, "four rows per lane");
    constexpr bool LS4 = MYO_NLIM_MAX + 4 * NC > 192;       // (the 48-slot scratch: a fourth row per lane)
    LANE_VAR(T, ls_x0); LANE_VAR(T, ls_x1); LANE_VAR(T, ls_x2); LANE_VAR(T, ls_x3);
    LANE_VAR(T, ls_v0); LANE_VAR(T, ls_v1); LANE_VAR(T, ls_v2); LANE_VAR(T, ls_v3);
    LANE_VAR(T, ls_d0); LANE_VAR(T, ls_d1); LANE_VAR(T, ls_d2); LANE_VAR(T, ls_d3);
    PHASE {
      const int r0 = lane, r1 = lane + 64, r2 = lane + 128, r3 = lane + 192;
      LV(ls_x3) = (LS4 && r3 < nefc) ? s.efc_jar[LS4 ? r3 : 0] : (T)0; LV(ls_v3) = (LS4 && r3 < nefc) ? s.efc_jv[LS4 ? r3 : 0] : (T)0;
      LV(ls_d3) = (LS4 && r3 < nefc) ? row_D(s, LS4 ? r3 : 0, nlim) : (T)0;
      LV(ls_x0) = r0 < nefc ? s.efc_jar[r0] : (T)0; LV(ls_v0) = r0 < nefc ? s.efc_jv[r0] : (T)0;
      LV(ls_d0) = r0 < nefc ? row_D(s, r0, nlim) : (T)0;
      LV(ls_x1) = r1 < nefc ? s.efc_jar[r1] : (T)0; LV(ls_v1) = r1 < nefc ? s.efc_jv[r1] : (T)0;
      LV(ls_d1) = r1 < nefc ? row_D(s, r1, nlim) : (T)0;
      LV(ls_x2) = r2 < nefc ? s.efc_jar[r2] : (T)0; LV(ls_v2) = r2 < nefc ? s.efc_jv[r2] : (T)0;
      LV(ls_d2) = r2 < nefc ? row_D(s, r2, nlim) : (T)0;
    }
    const int nls = M.any_floss ? 0 : 50;      // (models with friction-loss rows: fric_linesearch below)
    for (int li = 0; li < nls; ++li) {
      WAVE_SUM2_LANES(T, e1, e2, {
        const T xa = LV(ls_x0) + alpha * LV(ls_v0), xb = LV(ls_x1) + alpha * LV(ls_v1), xc = LV(ls_x2) + alpha * LV(ls_v2);
        if (xa < 0) { _e1 += LV(ls_d0) * xa * LV(ls_v0); _e2 += LV(ls_d0) * LV(ls_v0) * LV(ls_v0); }
        if (xb < 0) { _e1 += LV(ls_d1) * xb * LV(ls_v1); _e2 += LV(ls_d1) * LV(ls_v1) * LV(ls_v1); }
        if (xc < 0) { _e1 += LV(ls_d2) * xc * LV(ls_v2); _e2 += LV(ls_d2) * LV(ls_v2) * LV(ls_v2); }
        if constexpr (LS4) { const T xd = LV(ls_x3) + alpha * LV(ls_v3); if (xd < 0) { _e1 += LV(ls_d3) * xd * LV(ls_v3); _e2 += LV(ls_d3) * LV(ls_v3) * LV(ls_v3); } }
      });
      const T d1 = 2 * alpha * q2 + q1 + e1, d2 = 2 * q2 + e2;
#ifdef MYO_EMU_DEBUG
      printf("   ls %d alpha %g d1 %g d2 %g gtol %g e1 %g e2 %g\n", li, (double)alpha, (double)d1, (double)d2, (double)gtol, (double)e1, (double)e2);
#endif
      if (fabs(d1) < gtol) break;
      if (d1 < 0) lo = alpha; else hi = alpha;
      T next = alpha - d1 / d2;
      if (hi >= 0 && (next <= lo || next >= hi)) next = (T)0.5 * (lo + hi);
      if (next == alpha) break;
      alpha = next;
#ifdef MYO_EMU_DEBUG
      printf("   ls %d alpha %g d1 %g d2 %g gtol %g\n", li, (double)alpha, (double)d1, (double)d2, (double)gtol);
#endif
    }
    if (M.any_floss) alpha = fric_linesearch(M, s, q1, q2, gtol);
    PROF(s, 14)
    if (alpha == 0) break;
    PHASE {
      const int c = lane;
      if (c < nv) { s.qacc[c] += alpha * s.search[c]; s.Ma[c] += alpha * s.Mv[c]; }
      for (int r = lane; r < nefc; r += 64) s.efc_jar[r] += alpha * s.efc_jv[r];
    }
    SYNC();
    const HP oldcost = cost;
    REP(15, cost = update_constraint(M, s);)
    iter++;
    WAVE_SUM3_N(T, gn, qn2, fn, nv, c, { _e1 = s.search[c] * s.search[c]; _e2 = s.qacc[c] * s.qacc[c];
                                         _e3 = s.qfrc_smooth[c] * s.qfrc_smooth[c] + s.qfrc_constraint[c] * s.qfrc_constraint[c] + s.Ma[c] * s.Ma[c]; });
    const HP improvement = (HP)scale * (oldcost - cost);
    const T gradient = scale * sqrt(gn);
#ifdef MYO_EMU_DEBUG
    printf("  it %d improvement*scale %.3e gradient*scale %.3e step/qacc %.3e  g/f %.3e\n", iter, (double)improvement, (double)gradient, (double)(alpha * snorm / sqrt(qn2)), sqrt((double)gn / (double)fn));
#endif
    if (improvement < (HP)M.tolerance || gradient < M.tolerance) break;
    // Mixed stepper: MuJoCo's two tests sit below the rounding noise of an fp32 gradient (the fp64 solver
    // usually leaves through `gradient` right after the iteration that lands in the final active set, with
    // |grad| ~ 1e-12).  The T gradient  Ma - qfrc_smooth - qfrc_constraint  cannot get below ~1e-7 of its terms;
    // measured over contact-rich steps: once |grad| < 1e-6 |terms| the NEXT Newton step would move qacc by
    // <= 2e-6 |qacc| (median 1e-7), while every unconverged state has |grad| > 1e-5 |terms|.  The second test
    // is the same statement about the step just taken.  Iteration counts then equal the fp64 solver's.
    if (sizeof(T) != sizeof(HP) && (gn <= (T)1e-12 * fn || alpha * snorm <= (T)1e-5 * sqrt(qn2))) break;
  }
  PHASE { if (lane == 0) s.solver_iter = iter; }
  SYNC();
  PROF(s, 11)
}

template <typename T, int NC>
DEV void fwd_acceleration(const DevModel<T>& M_in, Scratch<T, NC>& s_in) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  WAVE_FN_K
  REP(11,
  PHASE { const int c = lane; if (c < M.nv) s.qacc_smooth[c] = s.qfrc_smooth[c]; }
  SYNC();
  solve_M(M, s, s.qacc_smooth, 0);
  )
  PROF(s, 8)
  if (s.nefc == 0) {
    PHASE {
      const int c = lane;
      if (c < M.nv) { s.qacc[c] = s.qacc_smooth[c]; s.qfrc_constraint[c] = 0; }
      if (lane == 0) s.solver_iter = 0;
    }
    SYNC();
    return;
  }
  newton_solve(M, s);
}

template <typename T, int NC>
DEV void forward(const DevModel<T>& M_in, const TaskDev& K_in, Scratch<T, NC>& s_in) {
  MYO_BIND_M(T) MYO_BIND_K MYO_BIND_S(T)
  PROF(s, 15)
  REP(0, kinematics(M, s);)
  PROF(s, 1)
  REP(1, com_pos(M, K, s);)
  PROF(s, 2)
  REP(2,      // (the whole tendon stage; bit 3: its geom-wrap passes alone)
  tendon(M, K, s);
  PROF(s, 16)
  REP(3, for (int base = 0; base < M.ngw; base += 64) tendon_wrap_pass(M, K, s, base);)
  PROF(s, 17)
  for (int base = 0; base < M.nte; base += 64) tendon_element_pass(M, K, s, base);
  tendon_length_sums(M, s);
  )
  PROF(s, 3)
  REP(4, crb(M, s);)
  PROF(s, 4)
  REP(5,      // (limit rows, friction-loss rows, contacts: the constraint set is rebuilt from its counters)
  if (M.any_floss) friction_rows(M, K, s, 0);
  constraint_limits(M, K, s);
  if (M.any_floss) friction_rows(M, K, s, 1);
  PROF(s, 18)
  REP(6,
  if (M.any_gen) {
    for (int base = 0; base < M.npair_std; base += 64) collision_pass<true>(M, K, s, base);
    for (int base = M.npair_std; base < M.npair; base += 64) collision_pass_ext<true>(M, K, s, base - M.npair_std);
  } else {
    for (int base = 0; base < M.npair_std; base += 64) collision_pass<false>(M, K, s, base);
    for (int base = M.npair_std; base < M.npair; base += 64) collision_pass_ext<false>(M, K, s, base - M.npair_std);
  }
  contacts_clamp(K, s);
  )
  )
  PROF(s, 5)
  REP(7, body_vectors(M, s, LOFF(s, S_QVELT(s)), LOFF(s, S_CVEL(s)));)
  PROF(s, 19)
  REP(8, fwd_velocity(M, K, s);)
  PROF(s, 20)
  REP(9, efc_reference(M, s);)
  PROF(s, 6)
  REP(10, fwd_actuation(M, s);)
  PROF(s, 7)
  fwd_acceleration(M, s);
}

// ------------------------------------------------------------------------------------------
// P11: integrators.  The state and its update are HP; the rates (qacc, act_dot, RK4 stage derivatives) are T.
template <typename T, int NC>
DEV void integrate_pos(const DevModel<T>& M_in, Scratch<T, NC>& s_in, LCREF(T) vel_r, HP h_in) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  // vel_r = null: the (HP) velocity state itself; otherwise a T vector (RK4 stage combination)
  const T* velT = LISNULL(vel_r) ? (const T*)0 : LPTR(const T, vel_r);
  const HP h = h_in;
  WAVE_FN_K
  PHASE {
    const int j = lane;
    if (j < M.njnt) {
      const int qa = M.jnt_qposadr[j], da = M.jnt_dofadr[j];
      if (M.jnt_type[j] == 0) {
        HP v[6];
        for (int k = 0; k < 6; ++k) v[k] = velT ? (HP)velT[da + k] : s.qvel[da + k];
        for (int k = 0; k < 3; ++k) s.qpos[qa + k] += h * v[k];
        HP w[3] = {v[3], v[4], v[5]};
        const HP ang = h * norm3(w);
        HP q[4] = {s.qpos[qa + 3], s.qpos[qa + 4], s.qpos[qa + 5], s.qpos[qa + 6]};
        if (ang > 0) {
          HP qr[4];
          normalize3(w);
          axisangle2quat(qr, w, ang);
          mulquat(q, q, qr);
        }
        normalize4(q);
        for (int k = 0; k < 4; ++k) s.qpos[qa + 3 + k] = q[k];
      } else s.qpos[qa] += h * (velT ? (HP)velT[da] : s.qvel[da]);
    }
  }
  SYNC();
}

template <typename T, int NC>
DEV void advance(const DevModel<T>& M_in, Scratch<T, NC>& s_in, LCREF(T) act_dot_r, LCREF(T) qacc_r, LCREF(T) vel_r) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  const bool own_rates = LISNULL(act_dot_r);                                            // (null: the scratch's own, S_ACT_DOT)
  const T* act_dot = own_rates ? (const T*)0 : LPTR(const T, act_dot_r);
  const T* qacc = LPTR(const T, qacc_r);
  WAVE_FN_K
  const HP h = M.h_timestep;
  PHASE {
    const int i = lane;
    if (i < M.na) {
      HP a = S_ACT(M, s)[i] + h * (HP)(own_rates ? S_ACT_DOT(s)[i] : act_dot[i]);
      if (M.actuator_dyntype[i + (M.nu - M.na)] == 3) a = tclamp(a, (HP)0, (HP)1);
      act_set(M, s, i, a);
    }
    if (i < M.nv) { s.qvel[i] += h * (HP)qacc[i]; warm_set(s, i, s.qacc[i]); }
    if (lane == 0) s.time += h;
  }
  SYNC();
  integrate_pos(M, s, vel_r, M.h_timestep);
}

template <typename T, int NC>
DEV void check_state(const DevModel<T>& M_in, Scratch<T, NC>& s_in, int check_acc) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  WAVE_FN_K
  // mj_checkPos / mj_checkVel / mj_checkAcc: any non-finite or huge entry marks the env bad.  No reduction:
  // every lane tests its own entries and stores the flag itself (all writers store 1; the flag is read
  // after later barriers, at the end of the env step).
  static_assert(MYO_NQ_MAX <= 64 && MYO_NV_MAX <= 64, "one entry per lane");
  PHASE {
    const int i = lane;
    int bad = 0;
    if (i < M.nq) bad |= !(isfinite(s.qpos[i]) && fabs(s.qpos[i]) < (HP)1e10);
    if (i < M.nv) {
      bad |= !(isfinite(s.qvel[i]) && fabs(s.qvel[i]) < (HP)1e10);
      if (check_acc) bad |= !(isfinite(s.qacc[i]) && fabs(s.qacc[i]) < (T)1e10);
    }
    if (bad) s.bad = 1;
  }
}

// RKM: the integrator when the caller knows it at compile time (the kernels' RK parameter: 1 = RK4, 0 = Euler; the other branch is
// then not compiled), -1 = read the model's
template <typename T, int RKM = -1, int NC>
DEV void mj_step(const DevModel<T>& M_in, const TaskDev& K_in, Scratch<T, NC>& s_in) {
  MYO_BIND_M(T) MYO_BIND_K MYO_BIND_S(T)
  WAVE_FN_K
  check_state(M, s, 0);
  forward(M, K, s);
  check_state(M, s, 1);
  PROF(s, 15)
  const bool rk4 = RKM < 0 ? (M.integrator == 1) : (RKM != 0);
  if (rk4) {
    // (SYNC_G in this branch: the stage storage s.rk is reached through a generic pointer — LDS behind the scratch or global memory)
    // RK4 (mj_RungeKutta): tableau 1/2,1/2,1; weights 1/6,1/3,1/3,1/6
    const int nq = M.nq, nv = M.nv, na = M.na, nf = 2 * nv + na;
    const HP h = M.h_timestep; const HP t0 = s.time;
    // (F0 + 2 F1 + 2 F2 + F3) / 6 is accumulated as it is written, left to right — ((F0 + 2 F1) + 2 F2) + F3 — so only the running
    // sum is kept; stage st's derivative, scaled by the next stage's tableau entry, goes straight into S_RKDX
    PHASE {
      for (int i = lane; i < nq; i += 64) s.rk->x0[i] = s.qpos[i];
      for (int i = lane; i < nv; i += 64) {
        s.rk->x0[nq + i] = s.qvel[i];
        const T f0 = (T)s.qvel[i], f1 = s.qacc[i];
        s.rk->Fsum[i] = f0; s.rk->Fsum[nv + i] = f1;
        S_RKDX(s)[i] = (T)0.5 * f0; S_RKDX(s)[nv + i] = (T)0.5 * f1;
      }
      for (int i = lane; i < na; i += 64) {
        s.rk->x0[nq + nv + i] = S_ACT(M, s)[i];
        const T f2 = S_ACT_DOT(s)[i];
        s.rk->Fsum[2 * nv + i] = f2; S_RKDX(s)[2 * nv + i] = (T)0.5 * f2;
      }
    }
    SYNC_G();
    for (int st = 1; st < 4; ++st) {
      const T a = (st == 3) ? (T)1 : (T)0.5;
      PHASE {
        for (int i = lane; i < nq; i += 64) s.qpos[i] = s.rk->x0[i];
      }
      SYNC_G();
      integrate_pos(M, s, LOFF(s, S_RKDX(s)), h);
      PHASE {
        for (int i = lane; i < nv; i += 64) s.qvel[i] = s.rk->x0[nq + i] + h * (HP)S_RKDX(s)[nv + i];
        for (int i = lane; i < na; i += 64) act_set(M, s, i, s.rk->x0[nq + nv + i] + h * (HP)S_RKDX(s)[2 * nv + i]);
        if (lane == 0) s.time = t0 + h * (HP)a;
      }
      SYNC_G();
      forward(M, K, s);
      PHASE {
        const T wgt = (st == 3) ? (T)1 : (T)2;          // weight of this stage in the sum
        const T an = (st == 2) ? (T)1 : (T)0.5;         // tableau entry of the NEXT stage (unused after the last)
        for (int i = lane; i < nv; i += 64) {
          const T f0 = (T)s.qvel[i], f1 = s.qacc[i];
          s.rk->Fsum[i] = s.rk->Fsum[i] + wgt * f0; s.rk->Fsum[nv + i] = s.rk->Fsum[nv + i] + wgt * f1;
          S_RKDX(s)[i] = an * f0; S_RKDX(s)[nv + i] = an * f1;
        }
        for (int i = lane; i < na; i += 64) {
          const T f2 = S_ACT_DOT(s)[i];
          s.rk->Fsum[2 * nv + i] = s.rk->Fsum[2 * nv + i] + wgt * f2; S_RKDX(s)[2 * nv + i] = an * f2;
        }
      }
      SYNC_G();
    }
    PHASE {
      for (int i = lane; i < nf; i += 64) S_RKDX(s)[i] = s.rk->Fsum[i] / 6;
      for (int i = lane; i < nq; i += 64) s.qpos[i] = s.rk->x0[i];
      for (int i = lane; i < nv; i += 64) s.qvel[i] = s.rk->x0[nq + i];
      for (int i = lane; i < na; i += 64) act_set(M, s, i, s.rk->x0[nq + nv + i]);
      if (lane == 0) s.time = t0;
    }
    SYNC_G();
    advance(M, s, LOFF(s, S_RKDX(s) + 2 * nv), LOFF(s, S_RKDX(s) + nv), LOFF(s, S_RKDX(s)));
  } else if (M.any_damping) {
    // Euler, implicit in joint damping: (M + h diag(b)) qacc' = qfrc_smooth + qfrc_constraint
    REP(16,
    PHASE { const int c = lane; if (c < M.nv) s.search[c] = s.qfrc_smooth[c] + s.qfrc_constraint[c]; }   // (search: dead after the solver)
    SYNC();
    solve_M(M, s, s.search, 1);
    )
    PROF(s, 12)
    advance(M, s, LNULL(const T), LOFF(s, s.search), LNULL(const T));
    PROF(s, 13)
  } else {
    advance(M, s, LNULL(const T), LOFF(s, s.qacc), LNULL(const T));
  }
}
