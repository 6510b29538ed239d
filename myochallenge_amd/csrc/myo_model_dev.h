// myo_model_dev.h — device-side view of a compiled model (shared by the HIP build and the
// MYO_EMU test build).  All arrays live in HBM, are read-only during stepping and are shared by
// every environment (they stay L2/L1 resident: ~25 KB for the 35-dof hand).
#pragma once
#include <stdint.h>

// compile-time capacity of one environment's scratch (LDS).  A model that exceeds any of these
// is rejected at myo_batch_create with MYO_E_UNSUPPORTED.
#define MYO_NB_MAX 24     // bodies
#define MYO_NJ_MAX 28     // joints
#define MYO_NV_MAX 36     // dofs (<= 64 required by the 64-bit dof masks)
#define MYO_NQ_MAX 38
#define MYO_NT_MAX 40     // tendons
#define MYO_NU_MAX 40     // actuators
#define MYO_AQ_ROW 40     // actuator moment entries per dof (packed qfrc_actuator gather rows)
#define MYO_MV_ROW 24     // non-zeros in one row of the tree-sparse inertia matrix (packed M*v rows)
#define MYO_TJ_MAX 8      // dofs one tendon can move
#define MYO_NM_MAX 176    // tree-sparse inertia entries
#define MYO_NCON_MAX 24   // contacts (base capacity of the scratch: Scratch<T, NC = MYO_NCON_MAX>)
#define MYO_NCON_F64 16   // contacts, base capacity of the fp64 stepper's scratch (the bench workload peaks at 11 - 12 contacts; a substep with more than the capacity drops the surplus and is counted, myo_batch_health)
#define MYO_NREC_F64 22   // contact record slots of the fp64 base scratch (its 56 + 4 * 16 constraint rows are shared between limit and contact rows)
#define MYO_NCON_BIG 48   // contact slots, scratch of models with extended collision pairs / a die / condim 4, 6 pairs (56 + 4 * 48 = 248 rows, four per lane).
                          // 34 until round 5: config E's rollouts asked for up to 45 slots (myo_batch_health [3]) and the surplus — penetrating, active contacts — was dropped
#define MYO_CS_MAX 16     // dofs one contact can move
#define MYO_NLIM_MAX 56   // joint-limit + tendon-limit rows
#define MYO_NEFC_MAX (MYO_NLIM_MAX + 4 * MYO_NCON_MAX)
#define MYO_OBS_MAX 104
#define MYO_ARROW_S 16    // separator rows of the block-arrow Newton system (one MFMA tile)
#define MYO_ARROW_B 4     // rows per leaf block (= the K of v_mfma_*_16x16x4)
#define MYO_ARROW_NF ((MYO_NV_MAX - MYO_ARROW_S) / MYO_ARROW_B)
#define MYO_ACT_PRE 24    // host-resolved muscle constants per actuator (act_pre; myobatch.hip fills them, fwd_actuation reads them)
// host-packed per-lane records (one level of table loads per stage instead of index -> index -> value chains; myobatch.hip fills them):
//   bk_i[MYO_BK_I b ..] = depth, parent, jntnum, jntadr, is-free-joint, qposadr of joints 0..2, type of joints 0..2, 0
//   bk_f[MYO_BK_F b ..] = body_pos[3], body_quat[4], then per joint k < 3: jnt_pos[3], jnt_axis[3], qpos0[qposadr]        (kinematics)
//   jk_i[4 j ..]        = body, dofadr, type, root body of the joint's body                                                (com_pos)
//   dk_i[2 d ..]        = qposadr of the dof's joint, 0;  dof_spr[2 d ..] = joint stiffness (0: none / free joint), qpos_spring   (passive forces)
//   dof_submask[d]      = body_submask[dof_bodyid[d]]                                                                       (RNE bias)
#define MYO_BK_I 12
#define MYO_BK_F 28
#define MYO_BK_NJ 3
#define MYO_LD_FQ 12      // 64-item chunks of the tree-sparse L'DL factorisation (myo_sparse_ldl.h); more: the dense path
#define MYO_LD_SQ 8       // 64-item chunks of its substitutions
#ifndef MYO_OBJG_MAX
#define MYO_OBJG_MAX 20     // geoms of the per-env object group (the die of the reorient task); include/myobatch.h
#endif
#ifndef MYO_ROT_CHOICE_MAX
#define MYO_ROT_CHOICE_MAX 4
#endif

// X-macro lists: (type, name).  I = int32, U = uint64, R = real (T)
#define MYO_MODEL_INT_ARRAYS(X)                                                                  \
  X(body_parentid) X(body_rootid) X(body_jntnum) X(body_jntadr) X(body_dofnum) X(body_dofadr)    \
  X(body_depth) X(jnt_type) X(jnt_qposadr) X(jnt_dofadr) X(jnt_bodyid) X(jnt_limited)            \
  X(dof_bodyid) X(dof_jntid) X(dof_parentid) X(dof_rootbody) X(dof_Madr) X(geom_type)            \
  X(geom_bodyid) X(geom_priority) X(site_bodyid) X(tendon_adr) X(tendon_num) X(tendon_limited)   \
  X(wrap_type) X(wrap_objid) X(wrap_side) X(actuator_dyntype) X(actuator_gaintype)               \
  X(actuator_biastype) X(actuator_tendon) X(actuator_ctrllimited) X(actuator_forcelimited)       \
  X(pair_geom1) X(pair_geom2) X(M_i) X(M_j) X(mv_adr) X(mv_col) X(mv_e) X(act_tj) X(act_sd) X(wr_i) X(mv_pack) X(mv_len) X(aq_pack) X(aq_len) X(gw_elem) X(pc_i) X(pc_sup) X(M_pk) X(te_i) X(tendon_eadr) X(tendon_enum) X(ld_fac) X(ld_sol) X(hperm) X(M_pkh) X(bk_i) X(jk_i) X(dk_i)
#define MYO_MODEL_U64_ARRAYS(X) X(body_dofmask) X(body_submask) X(dof_prevmask) X(tendon_dofmask) X(act_dofmask) X(wr_mask) X(pc_mask) X(dof_submask)
#define MYO_MODEL_REAL_ARRAYS(X)                                                                 \
  X(qpos0) X(qpos_spring) X(body_pos) X(body_quat) X(body_ipos) X(body_imat) X(body_mass)        \
  X(body_inertia) X(body_invweight0) X(jnt_solref) X(jnt_solimp) X(jnt_pos) X(jnt_axis)          \
  X(jnt_stiffness) X(jnt_range) X(jnt_margin) X(dof_armature) X(dof_damping) X(dof_invweight0)   \
  X(geom_solmix) X(geom_solref) X(geom_solimp) X(geom_size) X(geom_rbound) X(geom_pos)           \
  X(geom_mat) X(geom_friction) X(geom_margin) X(geom_gap) X(site_pos) X(tendon_solref_lim)       \
  X(tendon_solimp_lim) X(tendon_range) X(tendon_margin) X(tendon_stiffness) X(tendon_damping)    \
  X(tendon_lengthspring) X(tendon_invweight0) X(wrap_prm) X(actuator_dynprm)                     \
  X(actuator_gainprm) X(actuator_biasprm) X(actuator_ctrlrange) X(actuator_forcerange)           \
  X(actuator_gear) X(actuator_acc0) X(actuator_lengthrange) X(act_gear0) X(act_pre) X(wr_p) X(wr_m) X(pc_f) X(te_div)               \
  X(pair_mg) X(dof_frictionloss) X(dof_solref) X(dof_solimp) X(tendon_frictionloss) X(tendon_solref_fri) X(tendon_solimp_fri) X(bk_f) X(dof_spr)

// geometry tables of the HP stages (kinematic chain, contact / limit distances, observation): fp64 copies in
// every build, named h_<array> (the fp64 stepper's h_ tables are its ordinary tables)
#define MYO_MODEL_HP_ARRAYS(X)                                                                   \
  X(qpos0) X(body_pos) X(body_quat) X(jnt_pos) X(jnt_axis) X(jnt_range) X(jnt_margin)            \
  X(geom_pos) X(geom_mat) X(geom_size) X(geom_margin) X(geom_gap) X(geom_rbound) X(site_pos) X(wr_p) X(wr_m)       \
  X(actuator_lengthrange) X(actuator_gainprm) X(actuator_biasprm) X(act_pre) X(tendon_range) X(tendon_margin) X(pair_mg) X(bk_f)

// Model table handle.  The tables are immutable for the lifetime of a batch, so the gfx950 build
// reads them through the CONSTANT address space: a load with a wave-uniform index becomes a scalar
// (SMEM) load instead of a 64-lane vector load, and LLVM may treat every table load as invariant
// (hoisting / CSE across LDS stores and calls).  `arr + k` decays to a plain pointer as before.
template <typename U>
struct MyoCArr {
  const U* p;
#if defined(__HIP_DEVICE_COMPILE__)
  template <typename I> __device__ __attribute__((always_inline)) U operator[](I i) const {
    typedef const U __attribute__((address_space(4))) * cptr;
    return ((cptr)(unsigned long long)p)[i];
  }
  __device__ __attribute__((always_inline)) operator const U*() const { return p; }
#else
  template <typename I> U operator[](I i) const { return p[i]; }
  operator const U*() const { return p; }
#endif
};

template <typename T>
struct DevModel {
  int nq, nv, nu, na, nbody, njnt, ngeom, nsite, ntendon, nwrap, npair, nM, maxdepth;
  int integrator, iterations, disableflags, any_damping, any_tendon_passive, nlead, ngw, nte;
  int npair_std;                // pairs [0, npair_std): collision_pass; [npair_std, npair): collision_pass_ext
  int any_floss;                // some dof / tendon has friction loss: friction rows exist (the solver's Huber branches run)
  int any_gen;                  // some collision pair has a condim other than 3: the general contact-slot code runs (else the lean condim-3 path)
  int any_rot;                  // some collision pair has condim 4 / 6: contact slots with rotational rows exist (J' f stages torques)
  int arrow_nf;                 // block-arrow Newton system (myo_arrow_chol.h): number of 4-row leaf blocks behind the 16-row separator; 0: dense
  unsigned long long arrow_pad; // rows of the (permuted) 36-row system that hold no dof: identity
  int ld_nfq, ld_nsq;           // chunks of ld_fac / ld_sol in use (ld_nsq < 0: the model's M-only solves take the dense path)
  T timestep, tolerance, impratio, gravity[3], meaninertia;
  T isqrt_impratio;             // 1 / sqrt(impratio)
  double h_timestep;            // the integration step of the HP state update
#define X(n) MyoCArr<int> n;
  MYO_MODEL_INT_ARRAYS(X)
#undef X
#define X(n) MyoCArr<unsigned long long> n;
  MYO_MODEL_U64_ARRAYS(X)
#undef X
#define X(n) MyoCArr<T> n;
  MYO_MODEL_REAL_ARRAYS(X)
#undef X
#define X(n) MyoCArr<double> h_##n;
  MYO_MODEL_HP_ARRAYS(X)
#undef X
};

// Per-env state record in HBM (always fp64, env-major: one env's record is contiguous, so the
// 64 lanes of the env's wavefront read consecutive scalars).
struct EnvRecordLayout {
  int nq, nv, na;
  int off_qpos, off_qvel, off_act, off_warm, off_time, off_taskd, off_balld, off_misc, off_objfric;
  int stride;  // doubles per env
};
// taskd: start_angle[2], x_radius, y_radius, time_period, target_xy[4]  (9); die reorient: goal_pos[3], goal_quat[4], pos_dist, rot_dist
// objfric: friction[MYO_OBJG_MAX][3] of the object group's geoms (read only by batches that have a group)
// balld: mass[2], friction[2][3], size[2]  (10)
// misc : which_task, counter, elapsed_steps, episode_index, ep_return, ep_len (6, stored as double)
#define MYO_TASKD_N 9
#define MYO_BALLD_N 10
#define MYO_MISC_N 8      // which_task, counter, elapsed, episode, ep_ret, ep_len, bad (mid-step hand-off only), spare
